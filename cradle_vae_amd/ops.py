"""Thin tensor-level wrappers over the C ABI (one function per entry point).

These take/return torch CUDA tensors purely as device-memory handles; all arithmetic is in
the HIP kernels.  Autograd wiring lives in ``cradle_vae_amd.functional``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import hip

# Optional HIP-event timing of individual launches (bench.py: roofline of K1).  Events are
# recorded on the stream the kernel is launched on (torch's current stream).
KERNEL_TIMING = False  # True: K1 only (HIP events recorded by the C code); "all": also the _timed wrappers below
KERNEL_TIMERS = []  # (name, hip.EventPair) for K1; (name, start_event, end_event) for _timed regions


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if KERNEL_TIMING == "all":
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *exc):
        if KERNEL_TIMING == "all":
            self.e.record()
            KERNEL_TIMERS.append((self.name, self.s, self.e))


def _gp(G: int) -> int:
    """Row length the NB kernels run a gene axis of G at: the next multiple of 4 (16-byte vectors)."""
    return (G + 3) & ~3


def _gene_rows(t: torch.Tensor, fresh: bool = False) -> torch.Tensor:
    """``t`` (rows, G) as a view whose rows are 16-byte aligned and at least _gp(G) floats apart -- itself when it already
    is one (G % 4 == 0 and contiguous, or a [:, :G] view of a padded buffer as F_hip.linear makes for such widths),
    else a copy into a padded buffer.  The kernels never use what lies between G and the row stride."""
    rows, G = t.shape
    if (not fresh and t.stride(1) == 1 and t.stride(0) >= _gp(G) and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
            and t.dtype == torch.float32):
        return t
    buf = torch.empty((rows, _gp(G)), device=t.device, dtype=torch.float32)
    view = buf[:, :G]
    if not fresh:
        view.copy_(t)
    return view


def _rp(t: Optional[torch.Tensor]) -> Optional[int]:
    """Pointer of a _gene_rows view (rows contiguous, row stride checked there)."""
    return None if t is None else t.data_ptr()


def empty_gene_rows(rows: int, G: int, device) -> torch.Tensor:
    """Uninitialised (rows, G) fp32 view in the layout _gene_rows asks for."""
    return torch.empty((rows, _gp(G)), device=device, dtype=torch.float32)[:, :G]


def nb_gene_consts(log_concentration: torch.Tensor) -> torch.Tensor:
    """(G,) log-concentration -> (3, Gp) [c, lgamma(c), digamma(c)], Gp = G rounded up to 4  (mlp.py:160)."""
    G = log_concentration.numel()
    out = torch.empty((3, _gp(G)), device=log_concentration.device, dtype=torch.float32)
    hip.check(hip.load().cradle_nb_gene_consts(hip.ptr(log_concentration), hip.ptr(out), G, hip.stream()), "nb_gene_consts")
    return out


def _nb_check(logits, X, lib, P, N, G):
    if logits.shape != (P * N, G) or X.shape != (N, G) or lib.numel() != N:
        raise hip.CradleHipError(
            f"nb_logprob: shape mismatch logits {tuple(logits.shape)} X {tuple(X.shape)} lib {tuple(lib.shape)} "
            f"for P={P} N={N} G={G}"
        )


def nb_logprob_fwd(logits: torch.Tensor, X: torch.Tensor, lib: torch.Tensor, gene_consts: torch.Tensor, P: int) -> torch.Tensor:
    """sum_g log GammaPoisson(X | softmax(logits), lib, c) per (particle, cell) row -> (P*N,)."""
    N, G = X.shape
    _nb_check(logits, X, lib, P, N, G)
    logits, X = _gene_rows(logits), _gene_rows(X)
    out = torch.empty((P * N,), device=X.device, dtype=torch.float32)
    hip.check(
        hip.load().cradle_nb_logprob_fwd(
            _rp(logits), logits.stride(0), _rp(X), X.stride(0), hip.ptr(lib), hip.ptr(gene_consts),
            hip.ptr(out), P, N, G, hip.stream()),
        "nb_logprob_fwd",
    )
    return out


def nb_logprob_fwd_bwd(
    logits: torch.Tensor, X: torch.Tensor, lib: torch.Tensor, gene_consts: torch.Tensor, grow: torch.Tensor, P: int,
    dlog_c: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None, inplace: bool = False,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (out (P*N,), dlogits (P*N,G), dlog_concentration (G,)); ``inplace`` overwrites logits."""
    N, G = X.shape
    _nb_check(logits, X, lib, P, N, G)
    lib_ = hip.load()
    logits, X = _gene_rows(logits), _gene_rows(X)  # (inplace then overwrites the padded copy, if one was made)
    out = torch.empty((P * N,), device=X.device, dtype=torch.float32)
    dlogits = logits if inplace else empty_gene_rows(P * N, G, X.device)
    accumulate = dlog_c is not None
    if dlog_c is None:
        dlog_c = torch.empty((G,), device=X.device, dtype=torch.float32)
    need = lib_.cradle_nb_workspace_floats(P, N, G)
    if workspace is None:
        workspace = torch.empty((need,), device=X.device, dtype=torch.float32)
    ev = None
    if KERNEL_TIMING:  # HIP events right around the rows kernel, on the launch stream
        ev = hip.EventPair()
        KERNEL_TIMERS.append(("nb_logprob_fwd_bwd", ev))
    hip.check(
        lib_.cradle_nb_logprob_fwd_bwd_timed(
            _rp(logits), logits.stride(0), _rp(dlogits), dlogits.stride(0), _rp(X), X.stride(0),
            hip.ptr(lib), hip.ptr(gene_consts), hip.ptr(grow), hip.ptr(out), hip.ptr(dlog_c), int(accumulate),
            hip.ptr(workspace), workspace.numel(), P, N, G, ev.begin if ev else None, ev.end if ev else None,
            hip.stream()),
        "nb_logprob_fwd_bwd",
    )
    return out, dlogits, dlog_c


METHODS = {"mean": 0, "perturbseq": 1}
SAMPLERS = {"nb": 0, "mean": 1}


def nb_sample_accumulate(
    logits: torch.Tensor, lib: torch.Tensor, gene_consts: torch.Tensor, P: int, C: int, *, sums: Optional[torch.Tensor] = None,
    method: str = "mean", sampler: str = "nb", seed: int = 0, row_offset: int = 0, return_samples: bool = False,
) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """K9: draws x ~ GammaPoisson for every (particle, condition) row of ``logits`` (P*C, G)
    and adds the (optionally perturbseq-normalised) draws of each condition, summed over its
    particles, to ``sums`` (C, G).  -> (sums, samples or None)."""
    G = logits.shape[1]
    if logits.shape[0] != P * C or lib.numel() not in (1, C) or gene_consts.shape[-1] != _gp(G):
        raise hip.CradleHipError(
            f"nb_sample: shape mismatch logits {tuple(logits.shape)} lib {tuple(lib.shape)} for P={P} C={C}")
    if sums is None:
        sums = torch.zeros((C, _gp(G)), device=logits.device, dtype=torch.float32)[:, :G]
    if sums.shape != (C, G):
        raise hip.CradleHipError(f"nb_sample: sums has shape {tuple(sums.shape)}, expected {(C, G)}")
    # the kernel's rows of sums / samples are _gp(G) floats long: a caller's sums in another layout is added to afterwards
    logits = _gene_rows(logits)
    own_sums = sums.stride(1) == 1 and sums.stride(0) == _gp(G) and sums.data_ptr() % 16 == 0
    ksums = sums if own_sums else torch.zeros((C, _gp(G)), device=logits.device, dtype=torch.float32)[:, :G]
    samples = empty_gene_rows(P * C, G, logits.device) if return_samples else None
    lib_ = hip.load()
    need = lib_.cradle_nb_sample_workspace_floats(P, C, G)
    ws = _sample_workspace(logits.device, need) if need > 0 else None
    with _timed("nb_sample_accumulate"):  # (bench.py: events on the launch stream around K9 and its partial-sum pass)
        hip.check(
            lib_.cradle_nb_sample_accumulate_ws(
                _rp(logits), logits.stride(0), hip.ptr(lib), lib.numel(), hip.ptr(gene_consts), _rp(ksums),
                _rp(samples), P, C, G, METHODS[method], SAMPLERS[sampler], seed & (2**64 - 1), row_offset & 0xFFFFFFFF,
                hip.ptr(ws), need if ws is not None else 0, hip.stream()),
            "nb_sample_accumulate",
        )
    if not own_sums:
        sums += ksums
    return sums, samples


_SAMPLE_WS = {}


def _sample_workspace(device, floats: int) -> torch.Tensor:
    """Per-device scratch of the sampler's partial sums, grown on demand (stream-ordered reuse)."""
    ws = _SAMPLE_WS.get(device)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(floats, device=device, dtype=torch.float32)
        _SAMPLE_WS[device] = ws
    return ws


def dosage_rows(D: torch.Tensor, kmax: int, flags: Optional[torch.Tensor] = None):
    """Dense (N, T) dosage / indicator matrix -> ELL rows (idx (N, kmax) int32, val (N, kmax)); ``flags`` (int32
    device word) gets bit 0 OR-ed in when a row has more than kmax non-zeros."""
    if D.dim() != 2:
        raise hip.CradleHipError("dosage_rows: expected a 2-D matrix")
    D = D.contiguous().float()
    N, T = D.shape
    kmax = max(1, min(int(kmax), T, 64))
    idx = torch.empty((N, kmax), device=D.device, dtype=torch.int32)
    val = torch.empty((N, kmax), device=D.device, dtype=torch.float32)
    if flags is None:
        flags = torch.zeros(1, device=D.device, dtype=torch.int32)
    hip.check(hip.load().cradle_dosage_rows(hip.ptr(D), N, T, kmax, idx.data_ptr(), hip.ptr(val), flags.data_ptr(),
                                            hip.stream()), "dosage_rows")
    return idx, val, flags


def _table_strides(table: torch.Tensor, P: int):
    """(stride_p, stride_t) of a (T, L) or (P, T, L) table whose rows are contiguous."""
    if table.stride(-1) != 1 or table.dtype != torch.float32 or not table.is_cuda:
        raise hip.CradleHipError("gather/scatter tables must be fp32 CUDA tensors with contiguous rows")
    if table.dim() == 2:
        return 0, table.stride(0)
    if table.dim() == 3 and table.shape[0] in (1, P):
        return (0 if table.shape[0] == 1 and P > 1 else table.stride(0)), table.stride(1)
    raise hip.CradleHipError(f"table of shape {tuple(table.shape)} does not broadcast to {P} particles")


def gather_sum(table: torch.Tensor, idx: torch.Tensor, val: torch.Tensor, P: Optional[int] = None) -> torch.Tensor:
    """out[p, n, :] = sum_k val[n, k] * table[p, idx[n, k], :]  -> (P, N, L)."""
    if P is None:
        P = table.shape[0] if table.dim() == 3 else 1
    sp, st = _table_strides(table, P)
    N, kmax = idx.shape
    L = table.shape[-1]
    out = torch.empty((P, N, L), device=table.device, dtype=torch.float32)
    hip.check(hip.load().cradle_gather_sum(table.data_ptr(), sp, st, idx.data_ptr(), hip.ptr(val), P, N, kmax, L,
                                           hip.ptr(out), out.stride(0), out.stride(1), hip.stream()), "gather_sum")
    return out


def scatter_sum(dout: torch.Tensor, idx: torch.Tensor, val: torch.Tensor, T: int) -> torch.Tensor:
    """dtable[p, t, :] = sum over (n, k) with idx[n, k] == t of val[n, k] * dout[p, n, :]  -> (P, T, L)."""
    dout = dout.contiguous()
    P, N, L = dout.shape
    kmax = idx.shape[1]
    dtable = torch.empty((P, T, L), device=dout.device, dtype=torch.float32)
    hip.check(hip.load().cradle_scatter_sum(hip.ptr(dout), dout.stride(0), dout.stride(1), idx.data_ptr(), hip.ptr(val), P, N,
                                            kmax, T, L, hip.ptr(dtable), dtable.stride(0), dtable.stride(1), hip.stream()),
              "scatter_sum")
    return dtable


def mask_sample(logits: torch.Tensor, uniforms: torch.Tensor, temperature: float) -> torch.Tensor:
    """Hard relaxed-Bernoulli straight-through sample (no gradient): logits (Ts, L), uniforms (P, Ts, L) -> (P, Ts, L) in
    {0, 1}; what ``GumbelSoftmaxBernoulliStraightThrough.rsample`` returns, in one launch (inference calls only)."""
    P, Ts, L = uniforms.shape
    if tuple(logits.shape) != (Ts, L) or not uniforms.is_cuda or uniforms.dtype != torch.float32:
        raise hip.CradleHipError(f"mask_sample: logits {tuple(logits.shape)} / uniforms {tuple(uniforms.shape)} on {uniforms.device}")
    lg, u = logits.detach().contiguous().float(), uniforms.contiguous()
    out = torch.empty_like(u)
    hip.check(hip.load().cradle_mask_sample(hip.ptr(lg), hip.ptr(u), float(temperature), hip.ptr(out), P, Ts, L, hip.stream()),
              "mask_sample")
    return out


def gauss_sample(mean_raw: torch.Tensor, log_var_raw: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """nan_to_num(mean_raw) + sqrt(nan_to_num(|exp(log_var_raw) + 1e-4|)) * eps, all (rows, L) (rows may be strided):
    the Gaussian head's reparameterised draw of mlp.py:131-141 in one launch (inference calls only)."""
    L = mean_raw.shape[-1]
    m, lv = mean_raw.reshape(-1, L), log_var_raw.reshape(-1, L)
    e = eps.reshape(-1, L).contiguous()
    if m.shape != lv.shape or m.shape != e.shape or m.stride(1) != 1 or lv.stride(1) != 1 or not (m.is_cuda and lv.is_cuda) \
            or m.dtype != torch.float32 or lv.dtype != torch.float32:
        raise hip.CradleHipError("gauss_sample: fp32 CUDA tensors of one shape with contiguous rows expected")
    out = torch.empty((m.shape[0], L), device=m.device, dtype=torch.float32)
    # (row-strided views -- the two halves of one head buffer -- are taken as they are: pointer + leading dimension)
    hip.check(hip.load().cradle_gauss_sample(m.data_ptr(), m.stride(0), lv.data_ptr(), lv.stride(0), hip.ptr(e), hip.ptr(out),
                                             m.shape[0], L, hip.stream()), "gauss_sample")
    return out


NT, NN, TN = 0, 1, 2


def _mat(t: torch.Tensor):
    """(pointer, leading dimension) of a 2-D fp32 CUDA tensor view whose rows are contiguous."""
    if t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda or t.dtype != torch.float32:
        raise hip.CradleHipError("gemm operands must be 2-D fp32 CUDA tensors with unit column stride")
    return t.data_ptr(), t.stride(0)


PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2}
_default_precision = "f32"


class gemm_precision:
    """``with ops.gemm_precision("bf16x3"):`` -- the module path's GEMMs (guide / model / predictor: ATE, IWELBO) of
    >= 1 GFLOP run in the optional split-bf16 mode inside the block (``cradle_gemm_desc.precision``); the fused training
    step has its own switch (``loss_module.precision``).  Default "f32": the reference's arithmetic."""

    def __init__(self, precision: str):
        if precision not in PRECISIONS:
            raise KeyError(precision)
        self.precision = precision

    def __enter__(self):
        global _default_precision
        self.previous, _default_precision = _default_precision, self.precision
        return self

    def __exit__(self, *exc):
        global _default_precision
        _default_precision = self.previous


def gemm(
    layout: int, A: torch.Tensor, B: torch.Tensor, out: Optional[torch.Tensor] = None, *, beta: float = 0.0,
    bias: Optional[torch.Tensor] = None, act: bool = False, act_out: Optional[torch.Tensor] = None,
    residual: Optional[torch.Tensor] = None, dact_src: Optional[torch.Tensor] = None,
    out2: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None, rowbias_mode: int = 0,
    rowbias_n: int = 0, rowscale: Optional[torch.Tensor] = None, extra_col: Optional[torch.Tensor] = None,
    workspace: Optional[torch.Tensor] = None, repeat: int = 1, precision: Optional[str] = None, k_valid: int = 0,
) -> torch.Tensor:
    """fp32 MFMA GEMM with the fused epilogue of ``cradle_gemm_f32`` (see the header).
    ``repeat`` > 1 relaunches the same GEMM from C (benchmarking aid).  ``precision``: "f32" (exact fp32 MFMA) or
    "bf16x3" (split-bf16, three products per element pair, for products of >= 1 GFLOP).  ``k_valid`` > 0: the
    inner dimension has k_valid real entries and is padded by the caller to the next multiple of 4 (see
    ``cradle_gemm_desc.k_valid``): an operand may then show either extent along its inner axis."""
    if layout == NT:
        (M, K), (N, K2) = A.shape, B.shape
    elif layout == NN:
        (M, K), (K2, N) = A.shape, B.shape
    else:
        (K, M), (K2, N) = A.shape, B.shape
    if k_valid > 0:
        if {K, K2} - {k_valid, (k_valid + 3) & ~3}:
            raise hip.CradleHipError(f"gemm: inner dimensions {K}, {K2} are neither k_valid={k_valid} nor its padding")
        K = K2 = (k_valid + 3) & ~3
    if K != K2:
        raise hip.CradleHipError(f"gemm: inner dimensions differ ({K} vs {K2})")
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    if out.shape != (M, N):
        raise hip.CradleHipError(f"gemm: out has shape {tuple(out.shape)}, expected {(M, N)}")
    d = hip.GemmDesc()
    d.layout, d.M, d.N, d.K = layout, M, N, K
    d.A, d.lda = _mat(A)
    d.B, d.ldb = _mat(B)
    d.C, d.ldc = _mat(out)
    d.beta, d.act = beta, int(act)
    d.precision = PRECISIONS[_default_precision if precision is None else precision]
    d.k_valid = k_valid
    d.bias = hip.ptr(bias)
    if rowbias is not None:
        d.rowbias, d.ld_rowbias = _mat(rowbias)
        d.rowbias_mode, d.rowbias_n = rowbias_mode, rowbias_n
    if rowscale is not None:
        d.rowscale, d.rowscale_n = hip.ptr(rowscale), rowscale.numel()
    for name, t in (("act_out", act_out), ("residual", residual), ("dact_src", dact_src)):
        if t is not None:
            if t.shape != (M, N):
                raise hip.CradleHipError(f"gemm: {name} has shape {tuple(t.shape)}, expected {(M, N)}")
            ptr_, ld = _mat(t)
            setattr(d, name, ptr_)
            setattr(d, "ld_" + name, ld)
    if out2 is not None:
        d.C2, d.ldc2 = _mat(out2)
    if extra_col is not None:
        if extra_col.numel() != M:
            raise hip.CradleHipError("gemm: extra_col must have M elements")
        d.extra_col = hip.ptr(extra_col)
    ws_ptr, ws_n = (hip.ptr(workspace), workspace.numel()) if workspace is not None else (None, 0)
    import ctypes

    if repeat > 1:
        hip.check(hip.load().cradle_gemm_f32_repeat(ctypes.byref(d), ws_ptr, ws_n, repeat, hip.stream()), "gemm_f32")
    else:
        with _timed(("gemm", M, N, K)):
            hip.check(hip.load().cradle_gemm_f32(ctypes.byref(d), ws_ptr, ws_n, hip.stream()), "gemm_f32")
    return out


# ---------------------------------------------------------------------------------------------------------------
# input preparation on the device (csrc/dataprep.hip)
# ---------------------------------------------------------------------------------------------------------------
COUNT_TYPES = {torch.float32: 0, torch.uint8: 1, torch.int16: 2, torch.int32: 3}


def _counts(X: torch.Tensor):
    if not X.is_cuda or X.dim() != 2 or X.stride(1) != 1 or X.dtype not in COUNT_TYPES:
        raise hip.CradleHipError("expected a 2-D CUDA count matrix (fp32 / uint8 / int16 / int32) with unit column stride; "
                                 f"got {X.dtype} on {X.device}")
    return X.data_ptr(), COUNT_TYPES[X.dtype]


def log_stats(X: torch.Tensor, shift: Optional[torch.Tensor] = None, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> (2, G) fp64: per-gene sum over the rows of X of y and y^2, y = log(x + 1) - shift[g] evaluated in fp32
    (``cradle_log_stats``).  ``sums`` given: accumulated into (feed a matrix in row slabs)."""
    ptr_, ty = _counts(X)
    rows, G = X.shape
    lib_ = hip.load()
    accumulate = sums is not None
    if sums is None:
        sums = torch.empty((2, G), device=X.device, dtype=torch.float64)
    if sums.shape != (2, G) or sums.dtype != torch.float64 or not sums.is_contiguous():
        raise hip.CradleHipError("log_stats: sums must be a contiguous (2, G) fp64 tensor")
    if rows == 0:
        return sums if accumulate else sums.zero_()
    need = lib_.cradle_log_stats_workspace_doubles(rows, G)
    ws = torch.empty(need, device=X.device, dtype=torch.float64)
    hip.check(lib_.cradle_log_stats(ptr_, ty, rows, G, X.stride(0), hip.ptr(shift), sums.data_ptr(), int(accumulate),
                                    ws.data_ptr(), need, hip.stream()), "log_stats")
    return sums


def obs_counts_dense(M: torch.Tensor) -> torch.Tensor:
    """(M != 0).sum(0) of a dense (rows, C) fp32 matrix -> (C,) int64."""
    M = M.contiguous().float()
    out = torch.empty(M.shape[1], device=M.device, dtype=torch.int64)
    hip.check(hip.load().cradle_obs_counts_dense(hip.ptr(M), M.shape[0], M.shape[1], out.data_ptr(), hip.stream()),
              "obs_counts_dense")
    return out


def obs_counts_pairs(idx: torch.Tensor, val: torch.Tensor, n_treatments: int) -> torch.Tensor:
    """Per-treatment count of non-zero dosages of a matrix stored as (rows, kmax) (index, dosage) pairs -> (T,) int64."""
    idx, val = idx.contiguous(), val.contiguous()
    out = torch.empty(n_treatments, device=idx.device, dtype=torch.int64)
    hip.check(hip.load().cradle_obs_counts_pairs(hip.ptr(idx), hip.ptr(val), idx.shape[0], idx.shape[1], n_treatments,
                                                 out.data_ptr(), hip.stream()), "obs_counts_pairs")
    return out


def peer_median(src: torch.Tensor, peers: torch.Tensor) -> torch.Tensor:
    """out[n, g] = lower median over the leading non-negative entries of peers[n] of src[peer, g] (``cradle_peer_median``);
    a cell without peers gets a zero row.  src: stored counts, peers: (n, kmax) int32 rows of src, -1 = unused."""
    ptr_, ty = _counts(src)
    if not src.is_contiguous():
        raise hip.CradleHipError("peer_median: src must be contiguous")
    peers = peers.to(torch.int32).contiguous()
    n, kmax = peers.shape
    out = torch.empty((n, src.shape[1]), device=src.device, dtype=src.dtype)
    flags = torch.zeros(1, device=src.device, dtype=torch.int32)
    hip.check(hip.load().cradle_peer_median(ptr_, ty, src.shape[0], src.shape[1], peers.data_ptr(), n, kmax,
                                            out.data_ptr(), flags.data_ptr(), hip.stream()), "peer_median")
    if int(flags.item()) & 1:
        raise hip.CradleHipError("peer_median: a peer index lies outside the source matrix")
    return out
