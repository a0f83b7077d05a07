"""coarsegrainingvae_amd -- MI355X-native (gfx950) message-passing hot path of CoarseGrainingVAE.

Public names follow the reference package (``CoarseGrainingVAE.{modules,conv,cgvae,data}``) so
``from coarsegrainingvae_amd import CGequiVAE, EquiEncoder, ...`` is a drop-in for that path.
The compute lives in ``libcgvae_hip.so`` (hand-written HIP, C ABI in include/cgvae_hip.h);
there is no CPU or eager fallback.
"""
from .primitives import (CosineEnvelope, Dense, DistanceEmbed, PainnRadialBasis, Swish, layer_types,
                         shifted_softplus, to_module)
from .graph import BatchGraph, EdgeGeometry, EdgePlan, get_neighbor_list, make_directed, radius_graph
from .ops import scatter_add, scatter_mean
from .blocks import (ContractiveMessageBlock, EquiMessageBlock, EquiMessageCross, EquiMessagePsuedo, InvariantMessage,
                     PseudoUpdateBlock, UpdateBlock, preprocess_r)
from .model import CGequiVAE, CGprior, EquiEncoder, EquivariantDecoder, EquivariantPsuedoDecoder
from .data import (CG_collate, CGDataset, batch_to, build_dataset, get_high_order_edge, get_higher_order_adj_matrix,
                   prepare_batch, random_rotation_matrices, synthetic_batch)
from .train import KL, build_model, loop, loss_terms, train_step

__version__ = "0.1.0"
