"""World-size-2 test of the N>1 host path on CPU (gloo): noise generators, the flat gradient
all-reduce both step implementations use, work dealing for the ATE particle chunks.

The HIP step itself cannot run here; the per-rank gradients come from the CPU oracle (test
infrastructure), which is what the GPU parity tests pin the HIP step to.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_inputs(rank):
    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co

    shape = syn.TINY
    b = syn.make_batch(shape, seed=1000 + rank)
    b0 = syn.make_batch(shape, seed=1000)  # normalisation statistics and init are those of rank 0's view
    params = co.init_params(
        n_latent=shape.n_latent, n_treatments=shape.n_treatments, n_phenos=shape.n_phenos, n_qc=shape.n_qc,
        decoder_n_layers=shape.decoder_n_layers, decoder_n_hidden=shape.decoder_n_hidden,
        basal_encoder_n_layers=shape.basal_encoder_n_layers, basal_encoder_n_hidden=shape.basal_encoder_n_hidden,
        embedding_encoder_n_layers=shape.embedding_encoder_n_layers,
        embedding_encoder_n_hidden=shape.embedding_encoder_n_hidden, mask_prior_prob=shape.mask_prior_prob,
        log_x_mean=b0["log_x_mean"], log_x_std=b0["log_x_std"], seed=5)
    return shape, b, params


def _rank_grads(shape, b, params, noise, cf_noise):
    from oracle import cradle_oracle as co

    _, _, _, grads = co.elbo_loss_and_grads(
        params, X=b["X"], D=b["D"], qc=b["qc"], cfX=b["cfX"], library_size=b["library_size"],
        D_obs_counts=b["D_obs_counts"], qc_obs_counts=b["qc_obs_counts"], noise=noise, cf_noise=cf_noise,
        n_particles=shape.n_particles, beta=shape.beta)
    return grads


def _noise_for(rank, shape, seed=3):
    from cradle_vae_amd import parallel

    gp, gl = parallel.make_generators(seed, rank, "cpu")
    args = (gp, gl, shape.n_particles, shape.n_cells, shape.n_treatments, shape.n_latent, shape.n_qc, "cpu")
    return parallel.draw_noise(*args), parallel.draw_noise(*args)


def _worker(rank, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from cradle_vae_amd import parallel
        from oracle import cradle_oracle as co

        assert parallel.world_and_rank() == (WORLD, rank)
        shape, b, params = _shard_inputs(rank)
        noise, cf_noise = _noise_for(rank, shape)

        # 1. plated draws agree across ranks, local draws differ
        for nz in (noise, cf_noise):
            for k in parallel.PLATED_KEYS + parallel.LOCAL_KEYS:
                gathered = [torch.empty_like(nz[k]) for _ in range(WORLD)]
                dist.all_gather(gathered, nz[k])
                same = torch.equal(gathered[0], gathered[1])
                assert same == (k in parallel.PLATED_KEYS), k

        # 2. gradient averaging: per-parameter (modular path) and flat arena (fused path)
        grads = _rank_grads(shape, b, params, noise, cf_noise)
        keys = co.trainable_keys(params)
        leaves = [torch.nn.Parameter(params[k].clone()) for k in keys]
        for p, k in zip(leaves, keys):
            p.grad = grads[k].clone()
        parallel.allreduce_mean_grads_(leaves, WORLD)
        arena = torch.cat([grads[k].reshape(-1) for k in keys])
        parallel.allreduce_sum_(arena, WORLD)
        arena /= WORLD  # what cradle_adam_step's grad_scale does on the GPU
        # expectation: both shards' gradients recomputed locally (deterministic) and averaged
        expect = None
        for r in range(WORLD):
            sh, br, pr = _shard_inputs(r)
            nr, cr = _noise_for(r, sh)
            gr = _rank_grads(sh, br, pr, nr, cr)
            flat = torch.cat([gr[k].reshape(-1) for k in keys])
            expect = flat if expect is None else expect + flat
        expect /= WORLD
        got = torch.cat([p.grad.reshape(-1) for p in leaves])
        tol = 1e-6 * float(expect.abs().max())
        assert float((got - expect).abs().max()) <= tol
        assert float((arena - expect).abs().max()) <= tol

        # 3. identical update on every rank
        opt = torch.optim.Adam(leaves, lr=3e-4, weight_decay=1e-6)
        torch.nn.utils.clip_grad_norm_(leaves, 100.0)
        opt.step()
        new = torch.cat([p.detach().reshape(-1) for p in leaves])
        gathered = [torch.empty_like(new) for _ in range(WORLD)]
        dist.all_gather(gathered, new)
        assert torch.equal(gathered[0], gathered[1])

        # 4. ATE particle chunks: every chunk on exactly one rank; chunk noise independent of the rank
        mine = torch.zeros(7)
        mine[parallel.my_chunks(7, WORLD, rank)] = 1
        dist.all_reduce(mine)
        assert torch.equal(mine, torch.ones(7))
        from cradle_vae_amd.ate import _chunk_generator

        draw = torch.randn(4, generator=_chunk_generator("cpu", 0, 3))
        gathered = [torch.empty_like(draw) for _ in range(WORLD)]
        dist.all_gather(gathered, draw)
        assert torch.equal(gathered[0], gathered[1])

        # 5. rank-sharded data storage: each rank keeps (about) half of the cells, the halves are disjoint and
        #    cover the set, every epoch visits the rank's own cells in a new order, batches equal the source rows
        from cradle_vae_amd.data import DeviceBatchLoader, SCRNASeqTensorPerturbationDataset, SparseDosage

        g = torch.Generator().manual_seed(0)  # same data set on both ranks
        n, G, T = 101, 6, 9
        X = torch.poisson(3 * torch.rand(n, G, generator=g))
        D = torch.zeros(n, T)
        D[torch.arange(n), torch.randint(0, T, (n,), generator=g)] = 1.0
        qc = (torch.rand(n, 1, generator=g) < 0.2).float()
        ds = SCRNASeqTensorPerturbationDataset(X, SparseDosage.from_dense(D), qc, X.flip(0))
        whole = DeviceBatchLoader(ds, 8, shuffle=True, seed=2, shard_storage=False, rank=rank, world_size=WORLD)
        ld = DeviceBatchLoader(ds, 8, shuffle=True, drop_last=True, seed=2, rank=rank, world_size=WORLD)
        assert ld.shard_storage and ld._t["X"].shape[0] == n // WORLD and ld.stored_bytes() * 2 <= whole.stored_bytes() + 64
        e0, e1 = [], []
        for ep in (e0, e1):
            for bt in ld:
                i = bt["idx"]
                assert torch.equal(bt["X"], X[i]) and torch.equal(bt["D"], D[i]) and torch.equal(bt["cfX"], X.flip(0)[i])
                assert torch.equal(bt["library_size"], X[i].sum(1))
                ep.append(i)
        e0, e1 = torch.cat(e0), torch.cat(e1)
        assert not torch.equal(e0, e1) and set(e0.tolist()) <= set(ld.owned.tolist())
        owned = torch.full((n // WORLD,), -1, dtype=torch.long)
        owned[:len(ld.owned)] = ld.owned
        gathered = [torch.empty_like(owned) for _ in range(WORLD)]
        dist.all_gather(gathered, owned)
        a, b2 = set(gathered[0].tolist()), set(gathered[1].tolist())
        assert not (a & b2) and len(a | b2) == 2 * (n // WORLD)
        # the sharded-update collectives (reduce-scatter -> update of this rank's piece -> all-gather of the pieces),
        # in the forms the gloo backend gets: same sums and the same final buckets on every rank
        from cradle_vae_amd import parallel as par

        buckets = [torch.arange(8, dtype=torch.float32) * (rank + 1), torch.ones(4) * (rank + 2)]
        for bk in buckets:
            par.reduce_scatter_inplace_(bk, rank, WORLD)
        half = buckets[0][rank * 4:(rank + 1) * 4]
        assert torch.equal(half, torch.arange(8, dtype=torch.float32)[rank * 4:(rank + 1) * 4] * 3)  # (1 + 2) x
        sums = par.all_gather_small(torch.tensor([float(rank), 10.0 + rank]), WORLD)
        assert sums.tolist() == [0.0, 10.0, 1.0, 11.0]
        for bk in buckets:  # "update": every rank rewrites its own piece only
            n = bk.numel() // WORLD
            bk[rank * n:(rank + 1) * n] += 100.0 * (rank + 1)
            bk[(1 - rank) * n:(2 - rank) * n] = -1.0  # stale everywhere else
        par.all_gather_inplace_many_(buckets, rank, WORLD)
        assert buckets[0].tolist() == [100.0, 103.0, 106.0, 109.0, 212.0, 215.0, 218.0, 221.0]
        assert buckets[1].tolist() == [105.0, 105.0, 205.0, 205.0]
        results[rank] = "ok"
    except BaseException as e:  # noqa: BLE001
        results[rank] = f"{type(e).__name__}: {e}"
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_host_path():
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, port, results)) for r in range(WORLD)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
        for p in procs:
            if p.is_alive():
                p.kill()
        assert dict(results) == {0: "ok", 1: "ok"}, dict(results)
