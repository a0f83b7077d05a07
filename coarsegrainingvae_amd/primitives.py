"""Node-level primitives with the reference's names and state_dict layout
(CoarseGrainingVAE/modules.py): ``Dense``, ``Swish``, ``PainnRadialBasis``, ``CosineEnvelope``,
``DistanceEmbed``, ``layer_types``.

On the hot path the radial basis, envelope and the distance filter GEMV are fused into the
edge kernels (csrc/geometry.hip, csrc/equi_msg.hip); the module ``forward``s below exist for
API completeness (they materialise ``[E, feat]`` like the reference) and run as ordinary
device tensor ops.  ``DistanceEmbed`` only *owns* the filter parameters the kernels read.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as Fn
from torch import nn


class Swish(nn.Module):
    """x * sigmoid(x) (modules.py:16-21), as one fused device op."""

    def forward(self, x):
        return Fn.silu(x)


class shifted_softplus(nn.Module):
    def forward(self, x):
        return Fn.softplus(x) - np.log(2.0)


# activation registry, same keys as modules.py:32-42
layer_types = {
    "linear": nn.Linear, "Tanh": nn.Tanh, "ReLU": nn.ReLU, "shifted_softplus": shifted_softplus,
    "sigmoid": nn.Sigmoid, "Dropout": nn.Dropout, "LeakyReLU": nn.LeakyReLU, "ELU": nn.ELU, "swish": Swish,
}


def to_module(activation: str) -> nn.Module:
    return layer_types[activation]()


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b with a backward that can write the weight / bias gradients straight into the
    trainer's gradient arena (``param.grad`` is a view of it, trainer.py) instead of producing
    temporaries that autograd then adds in: per step that removes one [out,in] allocation and one
    read-modify-write pass per layer (~160 launches and ~0.8 GB of traffic at n_basis=600)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.params = (weight, bias)
        return Fn.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        w_param, b_param = ctx.params
        gx = gy.matmul(weight) if ctx.needs_input_grad[0] else None
        x2, gy2 = x.reshape(-1, x.shape[-1]), gy.reshape(-1, gy.shape[-1])
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = _direct_grad(w_param, lambda out: torch.mm(gy2.t(), x2, out=out), lambda: gy2.t().mm(x2))
        if b_param is not None and ctx.needs_input_grad[2]:
            gb = _direct_grad(b_param, lambda out: torch.sum(gy2, 0, out=out), lambda: gy2.sum(0))
        return gx, gw, gb


def _direct_grad(param, write_into, compute):
    """Arena-managed parameter (trainer.ParamArena sets ``_cgv_direct``): the first gradient of a
    step is written in place into ``param.grad`` (no zero-fill needed), later ones are added.
    Otherwise return the gradient to autograd as usual."""
    if getattr(param, "_cgv_direct", False) and param.grad is not None:
        if param._cgv_pending:
            write_into(param.grad)
            param._cgv_pending = False
        else:
            param.grad.add_(compute())
        return None
    return compute()


def mark_direct_grad(*params):
    """Declare that these parameters' gradients are produced by a backward that honours
    ``_direct_grad`` (so the arena may skip zero-filling them)."""
    for p in params:
        if p is not None:
            p._cgv_direct_ok = True


def linear(x, weight, bias=None):
    return _LinearFn.apply(x, weight, bias)


class Linear(nn.Linear):
    """torch.nn.Linear (same init, same parameter names) on the arena-aware linear function."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias)
        mark_direct_grad(self.weight, self.bias)

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias)


class Dense(nn.Linear):
    """Linear layer with xavier-uniform weights, zero bias, optional activation
    (modules.py:75-114).  ``nn.Linear.__init__`` calls ``reset_parameters`` exactly once, so the
    RNG stream matches the reference's for same-seed initialisation."""

    def __init__(self, in_features, out_features, bias=True, activation=None, dropout_rate=0.0):
        self.activation = None
        super().__init__(in_features, out_features, bias)
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_rate)   # parameter-free; kept so module trees line up
        self.dropout_rate = dropout_rate
        mark_direct_grad(self.weight, self.bias)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, inputs):
        y = _LinearFn.apply(inputs, self.weight, self.bias)
        if self.dropout_rate > 0.0:
            y = self.dropout(y)
        return self.activation(y) if self.activation is not None else y


class PainnRadialBasis(nn.Module):
    """sin(n pi d / cut) / d (modules.py:139-172); ``n`` is a plain attribute, not a buffer."""

    def __init__(self, n_rbf, cutoff):
        super().__init__()
        self.n = torch.arange(1, n_rbf + 1).float()
        self.cutoff = cutoff

    def forward(self, dist):
        d = dist.unsqueeze(-1)
        coef = (self.n * np.pi / self.cutoff).to(d.device)
        safe = torch.where(d == 0, torch.ones_like(d), d)
        val = torch.where(d == 0, coef.expand(d.shape[0], -1), torch.sin(coef * d)) / safe
        return torch.where(d >= self.cutoff, torch.zeros_like(val), val)


class CosineEnvelope(nn.Module):
    """0.5 (cos(pi d / cut) + 1), zero from the cutoff on (modules.py:45-58)."""

    def __init__(self, cutoff):
        super().__init__()
        self.cutoff = cutoff

    def forward(self, d):
        env = 0.5 * (torch.cos(np.pi * d / self.cutoff) + 1)
        return torch.where(d >= self.cutoff, torch.zeros_like(env), env)


class DistanceEmbed(nn.Module):
    """Owner of the distance-filter parameters ``block.1.{weight [feat,R], bias [feat]}``
    (modules.py:175-197).  The fused kernels read them directly through :meth:`filter_params`."""

    def __init__(self, n_rbf, cutoff, feat_dim, dropout):
        super().__init__()
        self.block = nn.Sequential(PainnRadialBasis(n_rbf=n_rbf, cutoff=cutoff),
                                   Dense(in_features=n_rbf, out_features=feat_dim, bias=True, dropout_rate=dropout))
        self.f_cut = CosineEnvelope(cutoff=cutoff)
        self.n_rbf, self.cutoff = n_rbf, cutoff

    def filter_params(self):
        return self.block[1].weight, self.block[1].bias

    def forward(self, dist):
        return self.block(dist) * self.f_cut(dist).reshape(-1, 1)
