"""Two ranks on one GPU (gloo, device tensors staged through the host): the data-parallel training step
with the bucketed all-reduce that overlaps the backward pass must give exactly the parameters of the
single all-reduce after the step, identical on both ranks.  RCCL across ranks needs >= 2 GPUs (the driver's
scaling run covers it); the second test runs the same step over the nccl backend with a one-rank group, which
exercises the bucket events, the communication stream and RCCL's stream ordering on the one GPU available."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, overlap, out, sync="allreduce", buckets="2"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo",
                      CRADLE_OVERLAP_ALLREDUCE="1" if overlap else "0", CRADLE_GRAD_SYNC=sync, CRADLE_BUCKETS=buckets)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from cradle_vae_amd import synthetic as syn
        from cradle_vae_amd.trainer import TrainStep
        from tests import golden_io as gio
        from tests.test_modular_gpu import build_loss_module

        torch.cuda.set_device(0)
        case = gio.load_elbo("mid", torch.float32, "cuda:0")
        lm = build_loss_module(case)
        st = TrainStep(lm, lr=1e-3, world_size=WORLD, seed=7)
        assert st.fused is not None and st.overlap == overlap and st.grad_sync == sync
        shape = syn.MID
        batch = syn.make_batch(shape, seed=50 + rank, device="cuda:0")  # each rank its own cells
        losses = [float(st(batch, n_particles=shape.n_particles)) for _ in range(4)]
        torch.cuda.synchronize()
        out[rank] = (st.fused.arena.flat.detach().cpu().clone(), losses)
    finally:
        dist.destroy_process_group()


def _run(overlap, sync="allreduce", buckets="2"):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, port, overlap, out, sync, buckets)) for r in range(WORLD)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=150)
        for p in procs:
            if p.is_alive():
                p.kill()
        assert sorted(out.keys()) == [0, 1], "a rank failed"
        return {r: out[r] for r in range(WORLD)}


def test_overlapped_bucket_allreduce_equals_single_allreduce():
    a = _run(overlap=True)
    b = _run(overlap=False)
    assert torch.equal(a[0][0], a[1][0]), "ranks diverged with the bucketed all-reduce"
    assert torch.equal(b[0][0], b[1][0]), "ranks diverged with the single all-reduce"
    assert torch.equal(a[0][0], b[0][0]), "bucketed and single all-reduce disagree"
    assert a[0][1] != a[1][1]  # different cells per rank: different local losses
    # reduce-scatter + clip/Adam on each rank's half of every bucket + all-gather of the parameters
    c = _run(overlap=True, sync="rs_ag")
    assert torch.equal(c[0][0], c[1][0]), "ranks diverged with reduce-scatter / sharded Adam / all-gather"
    assert torch.equal(c[0][0], a[0][0]), "sharded and full optimiser updates disagree"
    assert c[0][1] == a[0][1]
    # four buckets (finer overlap, more collectives), and the default of one (everything after the step, on the
    # step's own stream): other partitions of the clip norm's sum, so equal up to its rounding, and again the same
    # on both ranks
    for nb in ("4", "1"):
        d = _run(overlap=True, sync="rs_ag", buckets=nb)
        assert torch.equal(d[0][0], d[1][0]), nb
        assert torch.allclose(d[0][0], c[0][0], rtol=1e-5, atol=1e-7), nb


def _reject_worker(rank, port, out, overlap, sync, buckets):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo",
                      CRADLE_OVERLAP_ALLREDUCE="1" if overlap else "0", CRADLE_GRAD_SYNC=sync, CRADLE_BUCKETS=buckets)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from cradle_vae_amd import hip
        from cradle_vae_amd import synthetic as syn
        from cradle_vae_amd.trainer import TrainStep
        from tests import golden_io as gio
        from tests.test_modular_gpu import build_loss_module

        torch.cuda.set_device(0)
        lm = build_loss_module(gio.load_elbo("mid", torch.float32, "cuda:0"))
        lm.max_dosages_per_cell = 2
        st = TrainStep(lm, lr=1e-3, world_size=WORLD, seed=7)
        shape = syn.MID
        good = syn.make_batch(shape, seed=50 + rank, device="cuda:0")
        bad = dict(good)
        if rank == 1:  # only this rank's batch is invalid: a cell with three dosages where the step plans for two
            bad["D"] = good["D"].clone()
            bad["D"][0] = 0.0
            bad["D"][0, :3] = 1.0
        for _ in range(2):
            st(good, n_particles=shape.n_particles)
        torch.cuda.synchronize()
        before = [t.clone() for t in (st.fused.arena.flat, st.exp_avg, st.exp_avg_sq)]
        loss_bad = st(bad, n_particles=shape.n_particles)  # (no host check on this call: the state was seen before)
        torch.cuda.synchronize()
        loss_was_nan = bool(torch.isnan(loss_bad))  # (a view of the step's output buffer: read before the next step)
        unchanged = all(torch.equal(a, b) for a, b in zip(before, (st.fused.arena.flat, st.exp_avg, st.exp_avg_sq)))
        msg = None
        try:
            st.check_flags()
        except hip.CradleHipError as e:
            msg = str(e)
        st(good, n_particles=shape.n_particles)
        torch.cuda.synchronize()
        moved = not torch.equal(before[0], st.fused.arena.flat)
        st.check_flags()  # reported once; the valid step leaves nothing behind
        out[rank] = (unchanged, msg, moved, loss_was_nan, st.fused.arena.flat.detach().cpu().clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,sync,buckets", [(True, "rs_ag", "1"), (True, "rs_ag", "2"), (True, "allreduce", "2"),
                                                  (False, "allreduce", "1")])
def test_a_batch_rejected_on_one_rank_is_skipped_on_all(overlap, sync, buckets):
    """One rank's batch is invalid (more dosages in a cell than the step plans for): its gradients are garbage.  The
    flags word is local, the summed gradients are not -- so the rejecting rank poisons its gradients with a NaN, the
    squared norm is non-finite everywhere, and EVERY rank skips the update (parameters and moments bit-identical to
    before, and across ranks); the host check then raises on both ranks in the same call, and training goes on."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_reject_worker, args=(r, port, out, overlap, sync, buckets)) for r in range(WORLD)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=150)
        for p in procs:
            if p.is_alive():
                p.kill()
        assert sorted(out.keys()) == [0, 1], "a rank failed or hung"
        res = {r: out[r] for r in range(WORLD)}
    for r in range(WORLD):
        unchanged, msg, moved, _, _ = res[r]
        assert unchanged, f"rank {r} applied an update that held a rejected batch's gradients"
        assert msg is not None and "skipped on all 2 ranks" in msg, (r, msg)
        assert moved
    assert "kmax" in res[1][1] and "another rank" in res[0][1]
    assert res[1][3] and not res[0][3]  # the loss is NaN where the input was rejected, finite elsewhere
    assert torch.equal(res[0][4], res[1][4]), "ranks diverged"


def _oracle_worker(rank, port, out, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    for k in ("CRADLE_OVERLAP_ALLREDUCE", "CRADLE_GRAD_SYNC", "CRADLE_BUCKETS"):
        os.environ.pop(k, None)  # the defaults: what bench.py --gpus N and train.py run
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from cradle_vae_amd import synthetic as syn
        from cradle_vae_amd.trainer import TrainStep
        from tests import golden_io as gio
        from tests.test_modular_gpu import build_loss_module

        torch.cuda.set_device(0)
        lm = build_loss_module(gio.load_elbo("mid", torch.float32, "cuda:0"))
        st = TrainStep(lm, lr=1e-3, weight_decay=1e-6, gradient_clip_norm=5.0, world_size=WORLD, seed=11)
        fs = st.fused
        shape = syn.MID
        batch = syn.make_batch(shape, seed=60 + rank, device="cuda:0")
        names = {id(p): k for k, p in lm.named_parameters()}
        rec = []
        for _ in range(steps):
            before = {names[id(p)]: p.detach().cpu().clone() for p in fs.arena.params()}
            loss = st(batch, n_particles=shape.n_particles)
            torch.cuda.synchronize()
            noise = {k: fs.intermediate(k).cpu() for k in ("u_mask", "eps_E", "eps_qc_E", "eps_z")}
            gsum = {names[id(p)]: g.detach().cpu().clone() for p, g in zip(fs.arena.params(), fs.arena.grad_views())}
            after = {names[id(p)]: p.detach().cpu().clone() for p in fs.arena.params()}
            rec.append(dict(before=before, after=after, gsum=gsum, noise=noise, loss=float(loss),
                            norm=float(st.norm_out[0])))
        out[rank] = rec
    finally:
        dist.destroy_process_group()


def test_two_rank_update_equals_the_oracles_averaged_gradients():
    """What Lightning DDP computes on the reference (train_norman.py:89-97): every rank the ELBO gradients of its own
    cells, their mean over ranks, clip_grad_norm_, Adam -- rebuilt in fp64 on the CPU from the oracle and compared with
    the two-rank step on the GPU (default gradient-exchange form; gloo stands in for RCCL): the summed gradients every
    rank holds after the exchange, the norm the clip uses, and the parameters after each update.  The step draws its
    own noise; it is read back out of the workspace and injected into the oracle (plated draws identical on both ranks,
    z_basal's per rank)."""
    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co
    from tests import golden_io as gio

    steps, lr, clip = 3, 1e-3, 5.0
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_oracle_worker, args=(r, port, out, steps)) for r in range(WORLD)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=200)
        for p in procs:
            if p.is_alive():
                p.kill()
        assert sorted(out.keys()) == [0, 1], "a rank failed"
        rec = {r: out[r] for r in range(WORLD)}
    case = gio.load_elbo("mid", torch.float64, "cpu")
    cfg, shape = case["cfg"], syn.MID
    P = shape.n_particles
    fixed = {k: v for k, v in case["params"].items() if k not in co.trainable_keys(case["params"])}
    keys = co.trainable_keys(case["params"])
    leaves = {k: torch.zeros_like(case["params"][k]).requires_grad_(True) for k in keys}
    opt = torch.optim.Adam([leaves[k] for k in keys], lr=lr, weight_decay=1e-6)
    f64 = lambda d: {k: (v.double() if v.is_floating_point() else v) for k, v in d.items()}
    clipped = 0
    for s in range(steps):
        for k in keys:  # both ranks hold the same parameters; the reference optimiser restarts from them every step
            assert torch.equal(rec[0][s]["before"]["" + k], rec[1][s]["before"][k]), (s, k)
            leaves[k].data.copy_(rec[0][s]["before"][k].double())
        params = dict(fixed)
        params.update({k: leaves[k].detach() for k in keys})
        mean = {k: torch.zeros_like(leaves[k]) for k in keys}
        for r in range(WORLD):
            nz = f64(rec[r][s]["noise"])
            assert torch.equal(rec[r][s]["noise"]["u_mask"], rec[0][s]["noise"]["u_mask"])       # plated: shared
            assert r == 0 or not torch.equal(rec[r][s]["noise"]["eps_z"], rec[0][s]["noise"]["eps_z"])  # local: own
            T, L, Q, N = cfg["n_treatments"], cfg["n_latent"], cfg["n_qc"], shape.n_cells
            main = dict(u_mask=nz["u_mask"].view(2 * P, T, L)[:P], eps_E=nz["eps_E"].view(2 * P, T, L)[:P],
                        eps_qc_E=nz["eps_qc_E"].view(2 * P, Q, L)[:P], eps_z=nz["eps_z"].view(P, N, L))
            cf = dict(u_mask=nz["u_mask"].view(2 * P, T, L)[P:], eps_E=nz["eps_E"].view(2 * P, T, L)[P:],
                      eps_qc_E=nz["eps_qc_E"].view(2 * P, Q, L)[P:], eps_z=torch.zeros(P, N, L, dtype=torch.float64))
            b = f64(syn.make_batch(shape, seed=60 + r))
            loss, _, _, grads = co.elbo_loss_and_grads(
                params, X=b["X"], D=b["D"], qc=b["qc"], cfX=b["cfX"], library_size=b["library_size"],
                D_obs_counts=b["D_obs_counts"], qc_obs_counts=b["qc_obs_counts"], noise=main, cf_noise=cf,
                n_particles=P, beta=cfg["beta"], alpha=cfg["alpha"])
            assert abs(rec[r][s]["loss"] - float(loss)) <= 1e-4 * abs(float(loss)), (s, r)
            for k in keys:
                mean[k] += grads[k] / WORLD
        for k in keys:  # the exchange: every rank holds the sum of both ranks' gradients
            for r in range(WORLD):
                got = rec[r][s]["gsum"][k].double() / WORLD
                assert float((got - mean[k]).norm()) <= 1e-4 * max(float(mean[k].norm()), 1e-12), (s, r, k)
        norm = float(torch.sqrt(sum((g ** 2).sum() for g in mean.values())))
        clipped += norm > clip
        for r in range(WORLD):
            assert abs(rec[r][s]["norm"] - norm) <= 1e-5 * norm, (s, r, rec[r][s]["norm"], norm)
        for k in keys:
            leaves[k].grad = mean[k].clone()
        torch.nn.utils.clip_grad_norm_([leaves[k] for k in keys], clip)
        opt.step()
        # Adam's update is lr * m / (sqrt(v) + eps): where a gradient entry is within rounding of zero its sign, and
        # with it the whole lr-sized step, is undetermined between fp32 and fp64 -- so: (almost) every entry within
        # 1e-3 of a step, none further off than the step itself allows
        total = off = 0
        for k in keys:
            for r in range(WORLD):
                assert torch.equal(rec[r][s]["after"][k], rec[0][s]["after"][k]), (s, k)
            diff = (rec[0][s]["after"][k].double() - leaves[k].detach()).abs()
            assert float(diff.max()) <= 2.2 * lr, (s, k, float(diff.max()))
            total += diff.numel()
            off += int((diff > 1e-3 * lr).sum())
        assert off <= 2e-3 * total, (s, off, total)
    assert clipped >= 1  # the clip was active: its coefficient came from the all-rank norm


def _rccl_worker(port, overlap, out, sync="allreduce"):
    """One rank, backend nccl (= RCCL): the data-parallel step exactly as bench.py --gpus N runs it,
    on the only GPU a test box has.  world_size 2 is what TrainStep is told, so the bucket events, the
    communication stream and RCCL's own stream ordering are all exercised; the group itself has one rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CRADLE_OVERLAP_ALLREDUCE="1" if overlap else "0",
                      CRADLE_GRAD_SYNC=sync, CRADLE_BUCKETS="2")  # two buckets: the communication stream is in use
    import torch.distributed as dist

    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    except Exception as e:  # no usable network interface for RCCL's bootstrap on this box: environment, not code
        out["skip"] = f"{type(e).__name__}: {e}"
        return
    try:
        from cradle_vae_amd import synthetic as syn
        from cradle_vae_amd.trainer import TrainStep
        from tests import golden_io as gio
        from tests.test_modular_gpu import build_loss_module

        lm = build_loss_module(gio.load_elbo("mid", torch.float32, "cuda:0"))
        st = TrainStep(lm, lr=1e-3, world_size=2, seed=7)  # grads are halved in Adam either way
        assert st.fused is not None and st.overlap == overlap and st.grad_sync == sync
        batch = syn.make_batch(syn.MID, seed=50, device="cuda:0")
        losses = [float(st(batch, n_particles=syn.MID.n_particles)) for _ in range(4)]
        dist.barrier()
        torch.cuda.synchronize()
        if sync == "rs_ag":  # the four parameter all-gathers went out as one coalesced RCCL launch, not by the fallback
            from cradle_vae_amd import parallel

            assert parallel._COALESCE["ok"]
        out[(int(overlap), sync)] = (st.fused.arena.flat.detach().cpu().clone(), losses)
    finally:
        dist.destroy_process_group()


def test_rccl_backend_bucketed_allreduce_single_rank():
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        for overlap, sync in ((True, "allreduce"), (False, "allreduce"), (True, "rs_ag")):
            p = ctx.Process(target=_rccl_worker, args=(_free_port(), overlap, out, sync))
            p.start()
            p.join(timeout=150)
            if p.is_alive():
                p.kill()
            if "skip" in out:
                pytest.skip("RCCL process group could not be created here: " + out["skip"])
            assert p.exitcode == 0, f"RCCL worker (overlap={overlap}, {sync}) failed"
        a, b, c = out[(1, "allreduce")], out[(0, "allreduce")], out[(1, "rs_ag")]
    assert torch.equal(a[0], b[0]) and a[1] == b[1], "bucketed and single RCCL all-reduce disagree"
    assert torch.equal(a[0], c[0]) and a[1] == c[1], "RCCL reduce-scatter / all-gather form disagrees"
    assert all(x == x for x in a[1])


def _ate_kwargs(case):
    ate = case["ate"]
    Q = case["cfg"]["n_qc"]
    return dict(dosages_alt=ate["D_alt"], dosages_control=ate["D_control"], quality=torch.ones(1, Q),
                method="perturbseq", n_particles=24, condition_values=dict(library_size=1e4 * torch.ones(1)),
                batch_size=4, seed=5)


def _ate_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from tests import golden_io as gio
        from tests.test_predictor_gpu import _predictor

        torch.cuda.set_device(0)
        case = gio.load_predictor("tiny", torch.float32, "cuda:0")
        res = _predictor(case).estimate_average_treatment_effects(**_ate_kwargs(case))
        out[rank] = torch.from_numpy(res.X).clone()
    finally:
        dist.destroy_process_group()


def test_particle_parallel_ate_equals_single_process():
    """ATE with its particle chunks sharded over two ranks (real NB sampler, counter-based draws keyed by
    chunk, one sum all-reduce) against the same call in one process: equal up to the fp32 order of the sum."""
    from tests import golden_io as gio
    from tests.test_predictor_gpu import _predictor

    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_ate_worker, args=(r, port, out)) for r in range(WORLD)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=150)
        for p in procs:
            if p.is_alive():
                p.kill()
        assert sorted(out.keys()) == [0, 1], "a rank failed"
        a, b = out[0], out[1]
    assert torch.equal(a, b), "ranks disagree after the all-reduce"
    case = gio.load_predictor("tiny", torch.float32, "cuda:0")
    single = torch.from_numpy(_predictor(case).estimate_average_treatment_effects(**_ate_kwargs(case)).X)
    assert torch.isfinite(single).all() and float(single.abs().max()) > 0
    assert torch.allclose(a, single, rtol=1e-5, atol=1e-5 * float(single.abs().max()))
