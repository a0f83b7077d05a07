"""Explicit A/B switches (nothing is read from the environment).

``set(name, value)`` takes both kinds: launcher options of the C ABI (``cgv_set_option``, include/cgvae_hip.h: which
kernel computes an entry point) and the host mirror's own dispatch choices below.  Defaults are what every documented
number was measured with; the alternatives exist for A/B measurements and for the parity tests of the other kernels."""
from __future__ import annotations

from . import _lib

HOST = {
    "fwd_group": -1,        # receivers per group of the shared-source forward: -1 rule of graph.receiver_group_size, 0 / 2 / 4
    "wgrad_kernel": 0,      # grouped weight gradients of <= 64 rows: 0 weight-streaming VALU kernel, 1 MFMA tiles for all
    "table_upload": 0,      # record tables of a captured step: 0 copied once after capture, 1 a copy node in every replay
    "wgrad_tile": 64,       # output tile edge of the grouped MFMA weight-gradient launch: 64 or 128
    "rank_update": 1,       # Trainer: 1 rank update of the bead-level layers where it pays, 0 every gradient materialised (A/B)
    "rank_flat": 2,         # rank update of <= 16-row layers: contiguous ranges of rank_flat x 2048 float4 of a weight per block (0: 64 rows x one k tile)
    "rank_mixed": 1,        # ... and the layers that stay tiled (more rows) in the SAME launch, their blocks dealt among the flat ones (0: a second launch)
    "rank_rows_mfma": -1,   # single process: rows up to which layers beyond 40 rows take the MFMA rank update (-1: Trainer.RANK_ROWS_MFMA)
    "rank_gram_rows": -1,   # MFMA rank update: rows up to which the norm comes from the Gram launch instead of a tile pass (-1: Trainer.RANK_GRAM_ROWS)
    "concurrent_prior": 0,  # CGequiVAE.forward: 1 the prior net (bead graph) on a side stream beside the encoder -- a forked branch of the captured step
    "quad_heads": 1,        # CGequiVAE.forward: 1 layer j of the prior's and the encoder's four heads in one launch, 0 one launch pair per (mu, sigma) pair
    "fused_loss_tail": 1,   # Trainer: 1 decoder tail + ELBO + their backward in one launch (cgv_loss_tail), 0 reconstruct_fwd / elbo_fwd / reconstruct_bwd
    "paired_embeddings": 1, # CGequiVAE.forward: 1 the encoder's and the prior's embedding lookups in one launch (and their weight gradients in one), 0 one launch each
    "pair_sum2": 1,         # pair launches, shared input: 1 both layers' input gradients (+ alias, + parked segment gradient) as ONE two-source product, 0 a chain of two backward-input launches
    "head_pairs": 1,        # (mu, sigma) heads on more than 16 bead rows: 1 layer j of both heads as a pair launch of the tile kernels, 0 one launch per layer (second head through the first one's fork)
    "encoder_pairs": 1,     # EquiEncoder: 1 node MLPs of contractive block i and message block i + 1 (same input) as pair launches of the tile kernels (layer by layer), 0 one launch per layer
    "wgrad_split": 1,       # weight gradients of layers with > 128 operand rows: 1 bf16 matrix path with split operands (3 bf16 terms per fp32 value, 6 products, fp32 accumulation: fp32-class accuracy), 0 fp32 MFMA tiles
    "fused_bead_mean": 1,   # EquiEncoder layer 0: 1 H, V = scatter_mean(h), scatter_mean(v) from one launch inside the contractive block, gradient through its first Dense's epilogue; 0 two launches + broadcast + add
    "fused_prior": 1,       # CGprior: 1 the message-block loop of a small bead graph on the channel-group kernels (prior_fused.py), 0 per-block path
    "update_fused_fwd": 0,  # UpdateBlock forward on 17..96 bead rows: 0 (default) product + element-wise launch each (5 launches), 1 norm / gate in the epilogues of channel-group products (cgv_update_*_fwd_fused: 3 launches) -- measured SLOWER: dipeptide 2.645 / 2.675 against 2.636 / 2.637 ms, 2000 atoms 5.996 / 5.968 (a 96-row channel-group block is 900 fp32 MFMAs on one CU; the tile kernels spread the same product over the chip)
    "fwd_parts": 1,         # shared-source message forward (rb = 2): blocks per (group, channel tile), 1 (default) .. 4 (cgv_equi_msg_fwd_grouped_parts; measured level with 1 or slower)
    "fwd_balanced": 0,      # shared-source message forward (rb = 2): 0 (default) one block per (group, channel tile) (cgv_equi_msg_fwd_grouped), 1 equal edge ranges per wave on a resident grid (cgv_equi_msg_fwd_balanced) -- measured slower: chignolin 45.6 against 42.7 us, 2000 atoms 678 / 600
    "act_downstream": 1,    # Dense(act) -> Dense chains on the tile kernels (> 64 rows): 1 the second layer's backward-input launch multiplies its output by act'(z) of the first (once per element), whose own backward then runs without an activation; 0 act'(z) in the operand loads of the first layer's backward launches (every column-tile block evaluates it again: 704 x 600 x 600 17.7 against 11.2 us)
    "update_fused_bwd": 1,  # UpdateBlock backward on more than 32 bead rows: 1 the norm / stack backward in the store epilogue of s_dense.0's backward-input product (cgv_tile_linear_bwd_input_norm_stack: one launch less per layer), 0 product + element-wise launch
    "strip_split": 1,       # weight gradients of layers with 32 .. 96 operand rows (bead-level layers of a large bead batch, gathered rows of 4 - 8 ranks): 1 from 64 rows on strips on the bf16 matrix path with split operands (x split once per problem, g once per strip), 0 fp32 MFMA strips, 2 split strips at every row count (tests)
    "decoder_dense": 0,     # full-width products of the fused decoder loop: 0 four-column blocks (cgv_decoder_dense_fwd), 1 skinny_fwd_k
}
_DEFAULTS = dict(HOST)


def set(name: str, value: int) -> None:      # noqa: A001  (module-level API: options.set / options.get)
    if name in HOST:
        HOST[name] = int(value)
    elif name in _lib.OPTIONS:
        _lib.set_option(name, int(value))
    else:
        raise KeyError(f"unknown option {name!r}; known: {sorted(HOST) + sorted(_lib.OPTIONS)}")


def get(name: str) -> int:
    return HOST[name] if name in HOST else _lib.get_option(name)


def reset() -> None:
    HOST.update(_DEFAULTS)
    _lib.reset_options()


def apply(specs) -> None:
    """``["name=value", ...]`` (bench.py --option, tools)."""
    for spec in specs or ():
        name, _, value = spec.partition("=")
        set(name.strip(), int(value))


def pop_cli(argv) -> list:
    """Strip ``--option name=value`` pairs out of an argv list, apply them, return the rest (tools/*.py)."""
    rest, i = [], 0
    while i < len(argv):
        if argv[i] == "--option" and i + 1 < len(argv):
            apply([argv[i + 1]])
            i += 2
        else:
            rest.append(argv[i])
            i += 1
    return rest
