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

from . import _lib


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


def _skinny_ok(x2, weight):
    """Bead-level products (M <= 64 rows) go to the weight-streaming HIP kernels (csrc/skinny_gemm.hip);
    atom-level ones (hundreds of rows) are ordinary GEMMs and stay with hipBLASLt."""
    if not (x2.is_cuda and x2.dtype == torch.float32 and weight.dtype == torch.float32):
        return False
    M, K = x2.shape
    N = weight.shape[0]
    return bool(_lib.load().cgv_skinny_supported(M, N, K)) and weight.is_contiguous() and weight.data_ptr() % 16 == 0


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b.  Forward / backward run on the skinny-GEMM kernels when the row count is small,
    and the backward can write the weight / bias gradients straight into the trainer's gradient
    arena (``param.grad`` is a view of it, trainer.py) instead of producing temporaries that
    autograd then adds in: per step that removes one [out,in] allocation and one read-modify-write
    pass per layer (~160 launches and ~0.8 GB of traffic at n_basis=600)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.params = (weight, bias)
        ctx.skinny = _skinny_ok(x2, weight) and (bias is None or bias.data_ptr() % 16 == 0)
        if ctx.skinny:
            x2 = x2.contiguous()
            M, K = x2.shape
            N = weight.shape[0]
            y = torch.empty(M, N, dtype=torch.float32, device=x.device)
            _lib.call("cgv_skinny_linear_fwd", _lib.ptr(x2), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), M, N, K,
                      _lib.stream_ptr())
            ctx.save_for_backward(x2, weight)
            return y.reshape(x.shape[:-1] + (N,))
        ctx.save_for_backward(x, weight)
        return Fn.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        w_param, b_param = ctx.params
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = b_param is not None and ctx.needs_input_grad[2]
        x2, gy2 = x.reshape(-1, x.shape[-1]), gy.reshape(-1, gy.shape[-1])
        if not ctx.skinny:
            gx = gy.matmul(weight) if need_x else None
            gw = _direct_grad(w_param, lambda out: torch.mm(gy2.t(), x2, out=out), lambda: gy2.t().mm(x2)) if need_w else None
            gb = _direct_grad(b_param, lambda out: torch.sum(gy2, 0, out=out), lambda: gy2.sum(0)) if need_b else None
            return gx, gw, gb
        gy2 = gy2.contiguous()
        M, K = x2.shape
        N = weight.shape[0]
        st = _lib.stream_ptr()
        gx = None
        if need_x:
            gx = torch.empty(M, K, dtype=torch.float32, device=gy.device)
            ws_bytes = int(_lib.load().cgv_skinny_bwd_input_workspace_bytes(M, N, K))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=gy.device)
            _lib.call("cgv_skinny_linear_bwd_input", _lib.ptr(gy2), _lib.ptr(weight), _lib.ptr(gx), M, N, K, _lib.ptr(ws),
                      ws_bytes, st)
            gx = gx.reshape(gy.shape[:-1] + (K,))
        gw = gb = None
        if need_w or need_b:
            tw, acc_w, gw = _grad_target(w_param, weight) if need_w else (None, False, None)
            tb, acc_b, gb = _grad_target(b_param, b_param) if need_b else (None, False, None)
            if tw is None:                                   # bias only (never on this path): plain reduction
                gb = _direct_grad(b_param, lambda out: torch.sum(gy2, 0, out=out), lambda: gy2.sum(0))
            else:
                if tb is not None and acc_b != acc_w:        # mixed first/second write: keep the kernel's flag for W
                    gb = _direct_grad_after(b_param, gy2.sum(0), acc_b)
                    tb = None
                _lib.call("cgv_skinny_linear_bwd_weight", _lib.ptr(gy2), _lib.ptr(x2), _lib.ptr(tw), _lib.ptr(tb), M, N, K,
                          int(acc_w), st)
        return gx, gw, gb


def _is_direct(param):
    return param is not None and getattr(param, "_cgv_direct", False) and param.grad is not None


def _grad_target(param, like):
    """(tensor the kernel writes, accumulate?, value returned to autograd) for one parameter."""
    if _is_direct(param) and param.grad.is_contiguous():
        acc = not param._cgv_pending
        param._cgv_pending = False
        return param.grad, acc, None
    out = torch.empty_like(like)
    return out, False, out


def _direct_grad_after(param, value, accumulate):
    """Finish a direct-mode gradient whose target was already claimed by ``_grad_target``."""
    if _is_direct(param):
        if accumulate:
            param.grad.add_(value)
        else:
            param.grad.copy_(value)
        return None
    return value


def _direct_grad(param, write_into, compute):
    """Arena-managed parameter (trainer.ParamArena sets ``_cgv_direct``): the first gradient of a
    step is written in place into ``param.grad`` (no zero-fill needed), later ones are added.
    Otherwise return the gradient to autograd as usual."""
    if _is_direct(param):
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
