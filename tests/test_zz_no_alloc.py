"""cradle_nb_logprob_fwd_bwd_static -- the entry point that allocates nothing (its scratch is a device array
built into the library) -- against cradle_nb_logprob_fwd_bwd with a caller workspace, through the C ABI:
bit-identical outputs on two calls in a row, the second with a smaller input (the array keeps the first
call's contents), and a clear refusal beyond the built-in size."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(P, N, G, seed):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(P * N, G, generator=g).to(DEV)
    X = torch.poisson(3.0 * torch.rand(N, G, generator=g)).to(DEV)
    lib = X.sum(1).clamp(min=1.0).contiguous()
    log_c = (0.5 * torch.randn(G, generator=g)).to(DEV)
    grow = (-torch.rand(P * N, generator=g) / (P * N)).to(DEV)
    return logits, X, lib, log_c, grow


def _call(lib_, static, logits, X, lib, consts, grow, P, N, G):
    from cradle_vae_amd import hip

    out = torch.empty(P * N, device=DEV)
    dlogits = torch.empty_like(logits)
    dlog_c = torch.empty(G, device=DEV)
    common = (hip.ptr(logits), G, hip.ptr(dlogits), G, hip.ptr(X), G, hip.ptr(lib), hip.ptr(consts), hip.ptr(grow),
              hip.ptr(out), hip.ptr(dlog_c), 0)
    if static:
        rc = lib_.cradle_nb_logprob_fwd_bwd_static(*common, P, N, G, hip.stream())
    else:
        need = lib_.cradle_nb_workspace_floats(P, N, G)
        ws = torch.empty(need, device=DEV)
        rc = lib_.cradle_nb_logprob_fwd_bwd(*common, hip.ptr(ws), need, P, N, G, hip.stream())
    hip.check(rc, "nb_logprob_fwd_bwd")
    torch.cuda.synchronize()
    return out, dlogits, dlog_c


def test_static_scratch_entry_equals_workspace_entry_on_two_calls():
    from cradle_vae_amd import hip, ops

    lib_ = hip.load()
    assert lib_.cradle_nb_static_scratch_floats() == 4 << 20
    assert lib_.cradle_nb_static_init() == 0  # the address look-up, made eagerly (what a capturing caller does first)
    for (P, N, G, seed) in ((5, 96, 1200, 1), (3, 40, 300, 2)):  # the second call is the smaller one
        logits, X, lib, log_c, grow = _inputs(P, N, G, seed)
        consts = ops.nb_gene_consts(log_c)
        ref = _call(lib_, False, logits, X, lib, consts, grow, P, N, G)
        got = _call(lib_, True, logits, X, lib, consts, grow, P, N, G)
        for a, b, name in zip(got, ref, ("out", "dlogits", "dlog_concentration")):
            assert torch.equal(a, b), (P, N, G, name, float((a - b).abs().max()))
        assert torch.isfinite(got[0]).all() and float(got[2].abs().max()) > 0


def test_static_scratch_entry_rejects_what_does_not_fit():
    from cradle_vae_amd import hip

    lib_ = hip.load()
    P, N, G = 2, 1024, 8192  # 8 Mi floats of partial rows > the built-in 4 Mi
    assert lib_.cradle_nb_workspace_floats(P, N, G) > lib_.cradle_nb_static_scratch_floats()
    t = torch.zeros(16, device=DEV)  # never dereferenced: the size check comes first
    rc = lib_.cradle_nb_logprob_fwd_bwd_static(hip.ptr(t), G, hip.ptr(t), G, hip.ptr(t), G, hip.ptr(t), hip.ptr(t),
                                               hip.ptr(t), hip.ptr(t), hip.ptr(t), 0, P, N, G, hip.stream())
    assert rc == 1 and b"exceed the built-in" in lib_.cradle_last_error()
