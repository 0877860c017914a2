"""Autograd wiring of the HIP kernels (``torch.autograd.Function`` per kernel group).

Used by the module-level API (``guide.forward`` / ``model.forward`` / predictor), where the
caller composes the pieces itself.  The training step ``loss_module.loss`` does not go
through autograd at all: it calls the fused C step (``cradle_vae_amd.step``).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

LEAKY_SLOPE = 0.01


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) (+ residual) on the fp32 MFMA GEMM (mlp.py:76-80,111-119)."""

    @staticmethod
    def forward(ctx, x, W, b, act, residual):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, N = x2.shape[0], W.shape[0]
        res2 = residual.reshape(M, N) if residual is not None else None
        act_out = torch.empty((M, N), device=x.device, dtype=torch.float32) if (act and residual is not None) else None
        # an output width that is not a multiple of 4 (the decoder's logits over an arbitrary gene count) gets rows of
        # the next multiple: the NB kernels (K1, K9) then take the result as it is
        out = ops.empty_gene_rows(M, N, x.device) if (N % 4 and not act and residual is None) else None
        y = ops.gemm(ops.NT, x2, W, out, bias=b, act=act, act_out=act_out, residual=res2)
        ctx.act = act
        ctx.has_res = residual is not None
        ctx.has_bias = b is not None
        ctx.save_for_backward(x2, W, act_out if act_out is not None else (y if act else None))
        ctx.x_shape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W, r = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.stride(1) != 1 or ctx.act:
            dy2 = dy2.contiguous()
        da = dy2 * torch.where(r > 0, 1.0, LEAKY_SLOPE) if ctx.act else dy2
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(ops.NN, da, W).reshape(ctx.x_shape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            db_buf = torch.empty((W.shape[0],), device=dy.device, dtype=torch.float32) if ctx.has_bias else None
            dW = ops.gemm(ops.TN, da, x2, extra_col=db_buf)
            db = db_buf
        dres = dy if ctx.has_res else None
        return dx, dW, db, None, dres


def linear(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], act: bool = False,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _Linear.apply(x, W, b, act, residual)


class _NBLogProb(torch.autograd.Function):
    """sum_g log GammaPoisson(X | softmax(logits), library, exp(log_c)) per (particle, cell)."""

    @staticmethod
    def forward(ctx, logits, X, lib, log_c, P):
        gc = ops.nb_gene_consts(log_c)
        out = ops.nb_logprob_fwd(logits, X, lib, gc, P)
        ctx.save_for_backward(logits, X, lib, gc)
        ctx.P = P
        return out

    @staticmethod
    def backward(ctx, grow):
        logits, X, lib, gc = ctx.saved_tensors
        _, dlogits, dlog_c = ops.nb_logprob_fwd_bwd(logits, X, lib, gc, grow.contiguous(), ctx.P)
        return dlogits, None, None, dlog_c, None


def nb_logprob(logits: torch.Tensor, X: torch.Tensor, lib: torch.Tensor, log_c: torch.Tensor, P: int) -> torch.Tensor:
    """logits (P*N, G) -> (P*N,) summed log-likelihood (loss_modules.py:134-135,143)."""
    if logits.stride(-1) != 1:
        logits = logits.contiguous()  # (row-strided views pass: ops.nb_logprob_* read the leading dimension)
    return _NBLogProb.apply(logits, X.contiguous(), lib.contiguous(), log_c.contiguous(), P)


# ---------------------------------------------------------------------------------------------
# K4: matmul(D, table) through the non-zeros of D
# ---------------------------------------------------------------------------------------------
class DosagePlan:
    """ELL form of a dense (N, T) dosage / indicator matrix (<= ``kmax`` non-zeros per row).  ``overflow`` is a
    device flag (non-zero: some row has more, the gathers below are then wrong and the caller must redo the product
    densely -- ``check()`` raises, costing one sync)."""

    def __init__(self, D: torch.Tensor, kmax: int = 8, flags: Optional[torch.Tensor] = None):
        self.N, self.T = D.shape
        self.idx, self.val, self.overflow = ops.dosage_rows(D, kmax, flags)
        self.kmax = self.idx.shape[1]

    def rows(self, s0: int, s1: int) -> "DosagePlan":
        out = object.__new__(DosagePlan)
        out.N, out.T, out.kmax, out.overflow = s1 - s0, self.T, self.kmax, self.overflow
        out.idx, out.val = self.idx[s0:s1].contiguous(), self.val[s0:s1].contiguous()
        return out

    def check(self):
        if int(self.overflow.item()) & 1:
            raise RuntimeError(f"a row of the dosage matrix has more than kmax={self.kmax} non-zeros")


class _DosageMatmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, plan, P):
        ctx.plan, ctx.shape = plan, table.shape
        return ops.gather_sum(table, plan.idx, plan.val, P)

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        d = ops.scatter_sum(dout, plan.idx, plan.val, plan.T)  # (P, T, L)
        if len(ctx.shape) == 2:
            d = d.sum(0)
        elif ctx.shape[0] == 1 and d.shape[0] != 1:
            d = d.sum(0, keepdim=True)
        return d, None, None


def dosage_matmul(plan: DosagePlan, table: torch.Tensor, n_particles: Optional[int] = None) -> torch.Tensor:
    """``torch.matmul(D, table)`` for D (N, T) given as its ``DosagePlan`` and table (P, T, L) or (T, L)
    -> (P, N, L) (or (N, L) for a 2-D table without ``n_particles``)."""
    squeeze = table.dim() == 2 and n_particles is None
    P = n_particles if n_particles is not None else (table.shape[0] if table.dim() == 3 else 1)
    if not table.is_contiguous():
        table = table.contiguous()
    out = _DosageMatmul.apply(table, plan, P)
    return out[0] if squeeze else out


class _DenseDosageMatmul(torch.autograd.Function):
    """``matmul(D, table)`` as one fp32 MFMA GEMM per particle (NN layout; backward: D^T dout, TN layout) -- the dense
    form, for a dosage matrix whose rows hold more non-zeros than the gather plans for.  D is data: no gradient."""

    @staticmethod
    def forward(ctx, D, table, P):
        N, T = D.shape
        L = table.shape[-1]
        tab = table if table.dim() == 3 else table.unsqueeze(0)
        out = torch.empty((P, N, L), device=D.device, dtype=torch.float32)
        for p in range(P):
            ops.gemm(ops.NN, D, tab[p if tab.shape[0] > 1 else 0], out[p])
        ctx.save_for_backward(D)
        ctx.shape = table.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (D,) = ctx.saved_tensors
        P, N, L = dout.shape
        dout = dout.contiguous()
        d = torch.empty((P, D.shape[1], L), device=D.device, dtype=torch.float32)
        for p in range(P):
            ops.gemm(ops.TN, D, dout[p], d[p])
        if len(ctx.shape) == 2:
            d = d.sum(0)
        elif ctx.shape[0] == 1 and P != 1:
            d = d.sum(0, keepdim=True)
        return None, d, None


def dense_dosage_matmul(D: torch.Tensor, table: torch.Tensor, n_particles: Optional[int] = None) -> torch.Tensor:
    """``torch.matmul(D, table)`` for D (N, T) and table (P, T, L) or (T, L) -> (P, N, L), on the HIP GEMM."""
    P = n_particles if n_particles is not None else (table.shape[0] if table.dim() == 3 else 1)
    return _DenseDosageMatmul.apply(D.contiguous().float(), table.contiguous(), P)


class _OneHotLinear(torch.autograd.Function):
    """First layer of the embedding encoder on the input [mask | one_hot(t)] (correlated_normal_guide.py:
    147-153) without the one-hot: y[p, j, :] = LeakyReLU(mask[p, j] W[:, :L]^T + W[:, L + t_j] + b), the one-hot
    block entering as a per-row bias of the GEMM epilogue.  ``t0``: the rows are treatments t0 .. t0 + Ts - 1."""

    @staticmethod
    def forward(ctx, mask, W, b, t0):
        Pn, Ts, L = mask.shape
        x2 = mask.reshape(Pn * Ts, L)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        table = W[:, L + t0:L + t0 + Ts].t().contiguous()  # (Ts, H): weights only
        y = ops.gemm(ops.NT, x2, W[:, :L], bias=b, act=True, rowbias=table, rowbias_mode=2, rowbias_n=Ts)
        ctx.save_for_backward(x2, W, y)
        ctx.dims = (Pn, Ts, L, t0)
        return y.reshape(Pn, Ts, -1)

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        Pn, Ts, L, t0 = ctx.dims
        H = W.shape[0]
        da = dy.reshape(Pn * Ts, H) * torch.where(y > 0, 1.0, LEAKY_SLOPE)
        dmask = ops.gemm(ops.NN, da, W[:, :L]).reshape(Pn, Ts, L) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1]:
            dW = torch.zeros_like(W)
            db = torch.empty((H,), device=W.device, dtype=torch.float32)
            ops.gemm(ops.TN, da, x2, out=dW[:, :L], extra_col=db)
            dW[:, L + t0:L + t0 + Ts] = da.reshape(Pn, Ts, H).sum(0).t()
        return dmask, dW, db, None


def onehot_linear(mask: torch.Tensor, W: torch.Tensor, b: torch.Tensor, t0: int = 0) -> torch.Tensor:
    return _OneHotLinear.apply(mask, W, b, t0)
