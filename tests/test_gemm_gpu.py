"""GPU parity of the fp32 MFMA GEMM (all layouts, epilogues, tails, split-K) vs torch fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64).float().to(DEV)


def _close(got, ref, K):
    # one fp32 fmaf chain of length K per element: error <~ eps * sqrt(K) * magnitude
    tol = 2e-7 * max(1.0, K ** 0.5) * max(1.0, float(ref.abs().max()))
    err = float((got.double().cpu() - ref.cpu()).abs().max())
    assert err < tol, (err, tol)


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (2560, 400, 400), (525, 400, 200), (100, 30, 37), (1, 8, 8),
                                   (1024, 400, 5000), (300, 5000, 400), (5, 400, 600), (129, 65, 17)])
def test_nt_bias_act_residual(M, N, K):
    from cradle_vae_amd import ops

    x, W, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3), _rand(M, N, seed=4)
    act_out = torch.empty(M, N, device=DEV)
    y = ops.gemm(ops.NT, x, W, bias=b, act=True, act_out=act_out, residual=r)
    pre = torch.nn.functional.leaky_relu(x.double() @ W.double().T + b.double(), 0.01)
    _close(act_out, pre, K)
    _close(y, pre + r.double(), K)
    y2 = ops.gemm(ops.NT, x, W)
    _close(y2, x.double() @ W.double().T, K)


@pytest.mark.parametrize("M,N,K", [(2560, 400, 5000), (2560, 600, 400), (77, 33, 129), (1050, 200, 400)])
def test_nn_dgrad_with_leaky_backward(M, N, K):
    from cradle_vae_amd import ops

    dy, W, res, src = _rand(M, K, seed=1), _rand(K, N, seed=2), _rand(M, N, seed=3), _rand(M, N, seed=4)
    out2 = torch.empty(M, N, device=DEV)
    dx = ops.gemm(ops.NN, dy, W, residual=res, dact_src=src, out2=out2)
    ref = dy.double() @ W.double() + res.double()
    _close(dx, ref, K)
    _close(out2, ref * torch.where(src.double() > 0, 1.0, 0.01), K)


@pytest.mark.parametrize("M,N,K,ws", [(400, 400, 2560, True), (400, 400, 7680, True), (5000, 400, 2560, False),
                                      (400, 5000, 1024, True), (33, 17, 300, True), (400, 200, 5120, False),
                                      (8, 8, 24, True)])
def test_tn_wgrad_bias_column_splitk(M, N, K, ws):
    from cradle_vae_amd import hip, ops

    dy, x = _rand(K, M, seed=1), _rand(K, N, seed=2)
    db = torch.empty(M, device=DEV)
    workspace = None
    if ws:
        workspace = torch.empty(hip.load().cradle_gemm_workspace_floats(ops.TN, M, N, K, 1), device=DEV)
    dW = ops.gemm(ops.TN, dy, x, extra_col=db, workspace=workspace)
    _close(dW, dy.double().T @ x.double(), K)
    _close(db, dy.double().sum(0), K)
    # beta = 1 accumulates
    dW0 = dW.clone()
    ops.gemm(ops.TN, dy, x, out=dW, beta=1.0, workspace=workspace)
    _close(dW, 2 * dW0.double(), K)


def test_strided_views_and_rowbias():
    from cradle_vae_amd import ops

    P, T, L, H = 3, 7, 8, 12
    W0 = _rand(H, L + T, seed=1)  # embedding-encoder first layer [mask | one-hot]
    mask = (_rand(P * T, L, seed=2) > 0).float()
    b = _rand(H, seed=3)
    W0t_T = W0[:, L:].t().contiguous()  # (T, H)
    y = ops.gemm(ops.NT, mask, W0[:, :L], bias=b, act=True, rowbias=W0t_T, rowbias_mode=2, rowbias_n=T)
    onehot = torch.eye(T, device=DEV).repeat(P, 1)
    ref = torch.nn.functional.leaky_relu(torch.cat([mask, onehot], 1).double() @ W0.double().T + b.double(), 0.01)
    _close(y, ref, L + T)
    # mode 1 with a row scale: decoder first layer, bad-variant shift per particle
    N = 5
    z, Wd = _rand(P * N, 2 * L, seed=4), _rand(H, 3 * L, seed=5)
    wq, live = _rand(P, H, seed=6), (_rand(N, seed=7) > 0).float()
    y = ops.gemm(ops.NT, z, Wd[:, : 2 * L], rowbias=wq, rowbias_mode=1, rowbias_n=N, rowscale=live)
    ref = z.double() @ Wd[:, : 2 * L].double().T + live.double().repeat(P).unsqueeze(1) * wq.double().repeat_interleave(N, 0)
    _close(y, ref, 2 * L)


CONFIGS = [(2, 2, 16, 2, 1), (1, 1, 16, 2, 1), (1, 1, 16, 4, 1), (1, 1, 32, 2, 1), (1, 1, 32, 2, 2), (1, 1, 32, 4, 2),
           (1, 2, 32, 2, 2), (2, 2, 32, 2, 2), (1, 1, 32, 2, 4), (1, 1, 32, 4, 4), (1, 1, 64, 2, 4), (1, 2, 32, 2, 4),
           (2, 2, 32, 2, 4)]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_every_kernel_configuration_all_layouts(cfg):
    """Each instantiated tile / K-step / wave arrangement (incl. the single-wave-tile kernels with a
    four-way K split inside the workgroup), forced through the tuning hook, ragged sizes, with split-K."""
    from cradle_vae_amd import hip, ops

    lib = hip.load()
    g = torch.Generator().manual_seed(3)
    try:
        for layout, (M, N, K) in [(ops.NT, (203, 140, 333)), (ops.NN, (203, 140, 332)), (ops.TN, (204, 140, 333)),
                                  (ops.NT, (5, 400, 400)), (ops.TN, (400, 200, 1050))]:
            for sk in (1, 3):
                if layout == ops.NT:
                    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
                    ref = A.double() @ B.double().T
                elif layout == ops.NN:
                    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
                    ref = A.double() @ B.double()
                else:
                    A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
                    ref = A.double().T @ B.double()
                bias = torch.randn(N, generator=g)
                ref = ref + bias.double()
                ws = torch.empty(lib.cradle_gemm_workspace_floats(layout, M, N, K, 0), device=DEV)
                lib.cradle_gemm_tune(*cfg, sk)
                out = ops.gemm(layout, A.to(DEV), B.to(DEV), bias=bias.to(DEV), workspace=ws)
                err = float((out.double().cpu() - ref).abs().max())
                assert err <= 2e-7 * (K ** 0.5) * float(ref.abs().max()) + 1e-6, (cfg, layout, sk, err)
    finally:
        lib.cradle_gemm_tune(-1, -1, -1, -1, -1, -1)


# ---- optional split-bf16 mode (precision="bf16x3"): same entry point, same epilogues, products >= 1 GFLOP
def _close_b3(got, ref, K):
    # per product ~3 * 2^-18 relative to |a||b| (dropped lo*lo term + two split residuals), random sign: the error of a
    # length-K dot product of N(0,1) operands is ~1.2e-5 * sqrt(K) at worst; the bar leaves a factor of ~3 over what is seen
    tol = 1.2e-5 * max(1.0, K ** 0.5) * max(1.0, float(ref.abs().max()) / max(1.0, K ** 0.5))
    err = float((got.double().cpu() - ref.cpu()).abs().max())
    rel = float((got.double().cpu() - ref.cpu()).norm() / ref.cpu().norm())
    assert err < tol, (err, tol)
    assert rel < 1e-5, rel


@pytest.mark.parametrize("M,N,K", [(2560, 5000, 400), (1024, 400, 5000), (7680, 400, 400), (2500, 4996, 120), (130, 4100, 1000)])
def test_bf16x3_nt_bias_act_residual(M, N, K):
    from cradle_vae_amd import hip, ops

    x, W, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3), _rand(M, N, seed=4)
    act_out = torch.empty(M, N, device=DEV)
    ws = torch.empty(hip.load().cradle_gemm_workspace_floats(ops.NT, M, N, K, 0), device=DEV)
    y = ops.gemm(ops.NT, x, W, bias=b, act=True, act_out=act_out, residual=r, workspace=ws, precision="bf16x3")
    pre = torch.nn.functional.leaky_relu(x.double() @ W.double().T + b.double(), 0.01)
    _close_b3(act_out, pre, K)
    _close_b3(y, pre + r.double(), K)
    y32 = ops.gemm(ops.NT, x, W, bias=b, act=True, residual=r)
    assert not torch.equal(y, y32)  # the mode was really taken


@pytest.mark.parametrize("M,N,K", [(2560, 400, 5000), (7680, 400, 400), (2500, 404, 1236)])
def test_bf16x3_nn_dgrad_with_leaky_backward(M, N, K):
    from cradle_vae_amd import hip, ops

    dy, W, res, src = _rand(M, K, seed=1), _rand(K, N, seed=2), _rand(M, N, seed=3), _rand(M, N, seed=4)
    out2 = torch.empty(M, N, device=DEV)
    ws = torch.empty(hip.load().cradle_gemm_workspace_floats(ops.NN, M, N, K, 0), device=DEV)
    dx = ops.gemm(ops.NN, dy, W, residual=res, dact_src=src, out2=out2, workspace=ws, precision="bf16x3")
    ref = dy.double() @ W.double() + res.double()
    _close_b3(dx, ref, K)
    _close_b3(out2, ref * torch.where(src.double() > 0, 1.0, 0.01), K)


@pytest.mark.parametrize("M,N,K,ws", [(5000, 400, 2560, True), (400, 5000, 1024, True), (400, 400, 7680, True),
                                      (5000, 400, 2560, False), (404, 1236, 2500, True)])
def test_bf16x3_tn_wgrad_bias_column_splitk(M, N, K, ws):
    from cradle_vae_amd import hip, ops

    dy, x = _rand(K, M, seed=1), _rand(K, N, seed=2)
    db = torch.empty(M, device=DEV)
    workspace = None
    if ws:
        workspace = torch.empty(hip.load().cradle_gemm_workspace_floats(ops.TN, M, N, K, 1), device=DEV)
    dW = ops.gemm(ops.TN, dy, x, extra_col=db, workspace=workspace, precision="bf16x3")
    _close_b3(dW, dy.double().T @ x.double(), K)
    _close_b3(db, dy.double().sum(0), K)


def test_bf16x3_small_or_unaligned_products_stay_fp32():
    from cradle_vae_amd import ops

    x, W = _rand(300, 402, seed=1), _rand(200, 402, seed=2)  # K % 4 != 0 and far below 1 GFLOP
    assert torch.equal(ops.gemm(ops.NT, x, W, precision="bf16x3"), ops.gemm(ops.NT, x, W))
    with pytest.raises(KeyError):
        ops.gemm(ops.NT, x, W, precision="fp8")


def _view(rows, cols, ld, offset, seed, tail=float("nan")):
    """(rows, cols) view with leading dimension ld starting `offset` floats into a NaN-filled buffer: whatever a
    kernel reads outside the view and lets reach a stored result shows."""
    buf = torch.full((offset + rows * ld + 8,), tail, device=DEV)
    v = buf[offset:offset + rows * ld].view(rows, ld)[:, :cols]
    v.copy_(_rand(rows, cols, seed=seed))
    return v


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_ragged_operands_odd_offsets_and_leading_dimensions(precision):
    """Operands whose rows start at odd float offsets and whose M / N extents are not multiples of 4 (a weight block
    inside a wider matrix, n_phenos = 4k+1..3) -- the 16-byte loader takes them (dword alignment is all
    buffer_load_dwordx4 needs on gfx950); the slack it may read behind a row never reaches a stored element."""
    from cradle_vae_amd import hip, ops

    close = _close if precision == "f32" else _close_b3
    M, N, K = 1283, 403, 1000  # 1 GFLOP: the split-bf16 kernels take it when asked
    x, W = _view(M, K, K + 3, 1, 1), _view(N, K, K + 5, 3, 2)
    close(ops.gemm(ops.NT, x, W, precision=precision), x.double() @ W.double().T, K)
    Wn = _view(K, N, N + 2, 1, 3)  # NN: B is MN-major, 403 columns in rows of 405
    close(ops.gemm(ops.NN, x, Wn, precision=precision), x.double() @ Wn.double(), K)
    dy, xin = _view(K, M, M + 1, 2, 4), _view(K, N, N + 1, 1, 5)  # TN: both MN-major
    db = torch.empty(M, device=DEV)
    ws = torch.empty(hip.load().cradle_gemm_workspace_floats(ops.TN, M, N, K, 1), device=DEV)
    dW = ops.gemm(ops.TN, dy, xin, extra_col=db, workspace=ws, precision=precision)
    close(dW, dy.double().T @ xin.double(), K)
    close(db, dy.double().sum(0), K)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("G", [1201, 1202, 1203])
def test_padded_inner_dimension_with_k_valid(G, precision):
    """The decoder's output layer at a gene count that is not a multiple of 4: d logits rows are padded with zeros to
    Gp, the weight has G rows and NaNs behind them -- the range check, not the memory, supplies the padding."""
    from cradle_vae_amd import ops

    close = _close if precision == "f32" else _close_b3
    Gp, M, H = (G + 3) & ~3, 1100, 400
    dl = torch.zeros(M, Gp, device=DEV)
    dl[:, :G] = _rand(M, G, seed=1)
    Wh = _view(G, H, H, 0, 2)  # contiguous (G, H), NaN behind the last row
    ref = dl[:, :G].double() @ Wh.double()
    close(ops.gemm(ops.NN, dl, Wh, k_valid=G, precision=precision), ref, G)
    # x Wx^T: the weight's rows continue with other (finite) columns, which meet the zeros of the padded input
    Wx = _view(H, G + 40, G + 40, 0, 3)[:, :G]
    close(ops.gemm(ops.NT, dl, Wx, k_valid=G, precision=precision), dl[:, :G].double() @ Wx.double().T, G)
    with pytest.raises(ops.hip.CradleHipError):
        ops.gemm(ops.NN, dl, Wh[: G - 1], k_valid=G)


@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
def test_plain_bf16_mode_equals_a_product_of_rounded_operands(layout):
    """precision="bf16": the split kernels without the two correction products -- the result is the fp32-accumulated
    product of the operands rounded to bf16 (checked against exactly that in fp64), ~3e-3 relative to the exact one."""
    from cradle_vae_amd import hip, ops

    M, N, K = 1284, 400, 2560  # (an MN-major extent must be a multiple of 4 or have row slack for these kernels)
    lay = getattr(ops, layout)
    a = _rand(*((K, M) if layout == "TN" else (M, K)), seed=1)
    b = _rand(*((N, K) if layout == "NT" else (K, N)), seed=2)
    ws = torch.empty(hip.load().cradle_gemm_workspace_floats(lay, M, N, K, 0), device=DEV)
    y = ops.gemm(lay, a, b, workspace=ws, precision="bf16")
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    A = ar.T if layout == "TN" else ar
    B = br.T if layout == "NT" else br
    _close(y, A @ B, K)  # fp32 accumulation of exact bf16 x bf16 products
    exact = (a.double().T if layout == "TN" else a.double()) @ (b.double().T if layout == "NT" else b.double())
    rel = float((y.double() - exact).norm() / exact.norm())
    assert 1e-4 < rel < 1e-2, rel
