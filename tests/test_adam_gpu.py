"""cradle_adam_step (K10) against what the reference's trainer does to the same gradients:
``torch.nn.utils.clip_grad_norm_(params, 100)`` (Lightning's ``gradient_clip_val=100``,
train_norman.py:96) followed by ``torch.optim.Adam(lr, weight_decay=1e-6).step()``
(cradle_vae/models/utils/perturbation_lightning_module.py:147-150), run in fp64 on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference_run(p0, grads, lr, wd, max_norm, grad_scale):
    """fp64 torch: the DDP mean (grad_scale = 1/world), clip_grad_norm_, Adam with L2 weight decay."""
    # several parameter tensors, as in the module, so that clip_grad_norm_ really takes a global norm
    cuts = [0, 7, 400, p0.numel()]
    params = [torch.nn.Parameter(p0[a:b].clone().double()) for a, b in zip(cuts[:-1], cuts[1:])]
    opt = torch.optim.Adam(params, lr=lr, weight_decay=wd)
    hist, norms = [], []
    for g in grads:
        for p, (a, b) in zip(params, zip(cuts[:-1], cuts[1:])):
            p.grad = g[a:b].double() * grad_scale
        if max_norm is not None:
            norms.append(float(torch.nn.utils.clip_grad_norm_(params, max_norm)))
        else:
            norms.append(float(torch.cat([p.grad.reshape(-1) for p in params]).norm()))
        opt.step()
        hist.append(torch.cat([p.detach().reshape(-1) for p in params]).clone())
    return hist, norms


@pytest.mark.parametrize("n", [1003, 4096])  # 1003: not a multiple of 4 (scalar tail of the vector loads)
@pytest.mark.parametrize("gmag,max_norm", [(50.0, 100.0), (1e-2, 100.0), (50.0, None)])  # clip active / inactive / off
@pytest.mark.parametrize("grad_scale", [1.0, 0.5])
def test_adam_step_matches_torch_clip_and_adam(n, gmag, max_norm, grad_scale):
    from cradle_vae_amd import hip

    lib = hip.load()
    g = torch.Generator().manual_seed(n + int(gmag))
    steps, lr, wd = 24, 1e-3, 1e-6
    p0 = 0.1 * torch.randn(n, generator=g)
    grads = [gmag * torch.randn(n, generator=g) * (1.0 + 0.1 * i) for i in range(steps)]
    ref_hist, ref_norms = _reference_run(p0, grads, lr, wd, max_norm, grad_scale)
    if max_norm is not None:  # the case really is what its label says
        assert (min(ref_norms) > max_norm) == (gmag > 1.0)

    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ws = torch.zeros(lib.cradle_adam_workspace_floats(), device=DEV)
    flags = torch.zeros(1, device=DEV, dtype=torch.int32)
    for i, gr in enumerate(grads):
        gd = gr.to(DEV)
        hip.check(lib.cradle_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, 0.9, 0.999, 1e-8,
                                       wd, i + 1, -1.0 if max_norm is None else max_norm, grad_scale, ws.data_ptr(),
                                       flags.data_ptr(), hip.stream()), "cradle_adam_step")
        torch.cuda.synchronize()
        ref = ref_hist[i]
        # Adam's update is ~lr per entry whatever the gradient scale: the bar is a small fraction of one update
        err = float((p.cpu().double() - ref).abs().max())
        assert err <= 2e-3 * lr * (i + 1) ** 0.5 + 1e-7, (i, err)
        assert abs(float(ws[-1]) - ref_norms[i]) <= 1e-5 * ref_norms[i], (i, float(ws[-1]), ref_norms[i])
    # the moments too (first moment of the clipped, decayed gradient)
    assert float((p.cpu().double() - ref_hist[-1]).norm()) <= 1e-5 * float(ref_hist[-1].norm())


def test_adam_step_skips_the_update_when_flags_are_set():
    from cradle_vae_amd import hip

    lib = hip.load()
    n = 777
    p = torch.randn(n, device=DEV)
    gd = torch.randn(n, device=DEV)
    m, v = torch.rand(n, device=DEV), torch.rand(n, device=DEV)
    ws = torch.zeros(lib.cradle_adam_workspace_floats(), device=DEV)
    before = p.clone(), m.clone(), v.clone()
    for flag, moved in ((2, False), (0, True)):
        flags = torch.full((1,), flag, device=DEV, dtype=torch.int32)
        hip.check(lib.cradle_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8,
                                       1e-6, 1, 100.0, 1.0, ws.data_ptr(), flags.data_ptr(), hip.stream()), "adam")
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(before, (p, m, v))) != moved
    hip.check(lib.cradle_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8,
                                   1e-6, 2, 100.0, 1.0, ws.data_ptr(), None, hip.stream()), "adam")  # NULL flags: allowed
