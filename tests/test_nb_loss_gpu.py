"""GPU parity of K1 (NB log-likelihood fwd / fused fwd+bwd) against the fp64 oracle."""
import pytest
import torch

from oracle import cradle_oracle as co

pytestmark = pytest.mark.gpu


def _case(P, N, G, seed, count_scale=1.0, big_counts=False, tiny_c=False, tiny_c_head=0):
    g = torch.Generator().manual_seed(seed)
    logits = 2.0 * torch.randn(P * N, G, generator=g, dtype=torch.float64)
    lam = torch.exp(-1.0 + 1.5 * torch.randn(G, generator=g, dtype=torch.float64)) * count_scale
    X = torch.poisson(lam.expand(N, G).contiguous(), generator=g)
    if big_counts:
        X[:, : G // 8] = torch.poisson(3000.0 * torch.rand(N, G // 8, generator=g, dtype=torch.float64), generator=g)
    lib = X.sum(1) + 1.0
    log_c = 1.2 * torch.randn(G, generator=g, dtype=torch.float64)
    if tiny_c:  # concentrations far below a millionth of the mean: rate = c / (mu + 1e-4) sits on its clamp (mlp.py:163)
        log_c[::7] = -30.0
        log_c[3::11] = -18.0
    if tiny_c_head:  # only the first wave of a 256-thread workgroup meets clamped genes: its pass keeps the clamp, the others' drop it
        log_c[:tiny_c_head:3] = -30.0
    grow = torch.randn(P * N, generator=g, dtype=torch.float64)
    return logits.float().double(), X, lib, log_c.float().double(), grow.float().double()


def _oracle(logits, X, lib, log_c, grow, P):
    N, G = X.shape
    logits = logits.clone().requires_grad_(True)
    log_c = log_c.clone().requires_grad_(True)
    mu = lib.repeat(P).unsqueeze(1) * torch.softmax(logits, -1)
    c = torch.exp(log_c)
    rate = torch.clamp(c / (mu + co.MU_EPS), min=co.RATE_MIN)
    out = co.gamma_poisson_log_prob(X.repeat(P, 1), c, rate).sum(-1)
    (out * grow).sum().backward()
    return out.detach(), logits.grad, log_c.grad


@pytest.mark.parametrize(
    "P,N,G,kw",
    [
        (3, 8, 32, {}),
        (5, 64, 300, {}),
        (5, 96, 5000, {}),
        (2, 16, 5000, dict(big_counts=True)),
        (4, 32, 8000, dict(count_scale=5.0)),
        (7, 5, 2052, {}),
        (100, 6, 1000, {}),
        (2, 4, 16384, {}),
        (5, 64, 301, {}),   # gene counts that are not a multiple of 4: rows are run as the next multiple,
        (2, 9, 1202, {}),   # the tail is masked in the kernel
        (3, 17, 5003, dict(big_counts=True)),
        (4, 24, 1000, dict(tiny_c=True)),  # the rate clamp active on real genes (K1's single-reciprocal form, nb_math.h)
        (5, 12, 5000, dict(tiny_c_head=256)),  # ... in one wave of each workgroup only: both passes in one launch
    ],
)
def test_nb_logprob_matches_oracle(P, N, G, kw):
    from cradle_vae_amd import ops

    logits, X, lib, log_c, grow = _case(P, N, G, seed=P * 1000 + N, **kw)
    ref_out, ref_dl, ref_dc = _oracle(logits, X, lib, log_c, grow, P)
    dev = "cuda:0"
    f = lambda t: t.float().to(dev).contiguous()  # noqa: E731
    gc = ops.nb_gene_consts(f(log_c))
    out_f = ops.nb_logprob_fwd(f(logits), f(X), f(lib), gc, P)
    out, dl, dc = ops.nb_logprob_fwd_bwd(f(logits), f(X), f(lib), gc, f(grow), P)
    torch.cuda.synchronize()
    # per-row log-likelihood: 1e-5 relative to the row magnitude (north star: ELBO within 1e-4)
    scale = ref_out.abs().clamp(min=1.0)
    assert float(((out_f.cpu().double() - ref_out).abs() / scale).max()) < 1e-5
    # fwd-only and fused kernels are separate instantiations: same math, last-bit differences allowed
    assert torch.allclose(out.cpu(), out_f.cpu(), rtol=2e-6, atol=1e-5)
    # gradients: relative to the largest entry of each row / of the vector
    row_scale = ref_dl.abs().amax(dim=1, keepdim=True).clamp(min=1e-20)
    assert float(((dl.cpu().double() - ref_dl).abs() / row_scale).max()) < 2e-5
    assert float((dc.cpu().double() - ref_dc).abs().max() / ref_dc.abs().max()) < 2e-5
    # in place + accumulate variant gives the same numbers
    lg = ops.empty_gene_rows(P * N, G, dev)  # the layout the kernel takes as it is, whatever G
    lg.copy_(f(logits))
    dc2 = torch.ones(G, device=dev)
    out2, dl2, dc2 = ops.nb_logprob_fwd_bwd(lg, f(X), f(lib), gc, f(grow), P, dlog_c=dc2, inplace=True)
    assert dl2.data_ptr() == lg.data_ptr()
    assert torch.equal(dl2.cpu(), dl.cpu()) and torch.equal(out2.cpu(), out.cpu())
    assert torch.allclose(dc2.cpu(), dc.cpu() + 1.0, rtol=1e-6, atol=1e-6)


def test_nb_ragged_gene_count_never_reads_the_row_padding():
    """G = 4k+1..3 handed over as [:, :G] views of NaN-filled buffers with rows of the next multiple of 4 (what
    F_hip.linear produces for such widths): the NaNs behind column G-1 reach neither the row sums nor a gradient,
    d logits is written as zeros there, and d log_concentration ends at G."""
    from cradle_vae_amd import ops

    dev = "cuda:0"
    for G in (29, 30, 31):
        P, N = 3, 6
        logits, X, lib, log_c, grow = _case(P, N, G, seed=G)
        ref_out, ref_dl, ref_dc = _oracle(logits, X, lib, log_c, grow, P)

        def padded(t):
            buf = torch.full((t.shape[0], 32), float("nan"), device=dev)
            buf[:, :G] = t.float().to(dev)
            return buf, buf[:, :G]

        lbuf, lg = padded(logits)
        _, xv = padded(X)
        f = lambda t: t.float().to(dev).contiguous()  # noqa: E731
        gc = ops.nb_gene_consts(f(log_c))
        assert gc.shape == (3, 32)
        out_f = ops.nb_logprob_fwd(lg, xv, f(lib), gc, P)
        dcbuf = torch.full((32,), 7.0, device=dev)
        out, dl, dc = ops.nb_logprob_fwd_bwd(lg, xv, f(lib), gc, f(grow), P, dlog_c=dcbuf[:G], inplace=True)
        torch.cuda.synchronize()
        assert dl.data_ptr() == lbuf.data_ptr()  # no copy was made
        scale = ref_out.abs().clamp(min=1.0)
        assert float(((out_f.cpu().double() - ref_out).abs() / scale).max()) < 1e-5
        assert float(((out.cpu().double() - ref_out).abs() / scale).max()) < 1e-5
        row_scale = ref_dl.abs().amax(dim=1, keepdim=True).clamp(min=1e-20)
        assert float(((dl.cpu().double() - ref_dl).abs() / row_scale).max()) < 2e-5
        assert torch.equal(lbuf[:, G:], torch.zeros_like(lbuf[:, G:]))
        assert float((dcbuf[:G].cpu().double() - 7.0 - ref_dc).abs().max() / ref_dc.abs().max()) < 2e-5
        assert torch.equal(dcbuf[G:], torch.full((32 - G,), 7.0, device=dev))


def test_nb_rejects_bad_arguments():
    from cradle_vae_amd import hip, ops

    dev = "cuda:0"
    gc = ops.nb_gene_consts(torch.zeros(30, device=dev))
    with pytest.raises(hip.CradleHipError):  # counts of another width than the logits
        ops.nb_logprob_fwd(torch.zeros(4, 30, device=dev), torch.zeros(4, 28, device=dev), torch.ones(4, device=dev), gc, 1)
    with pytest.raises(hip.CradleHipError):
        ops.nb_logprob_fwd(torch.zeros(4, 32), torch.zeros(4, 32), torch.ones(4), gc.cpu(), 1)
