"""One training step = ELBO loss + backward + gradient all-reduce + norm clip + Adam.

What Lightning does around ``training_step`` in the reference (``pl.Trainer(gradient_clip_val=
100)`` train_norman.py:89-97, ``configure_optimizers`` perturbation_lightning_module.py:
147-150: Adam with L2 weight decay 1e-6), as an explicit loop body that works without
pytorch_lightning and with one process per GPU (``torch.distributed``, RCCL).
"""
from __future__ import annotations

from typing import Dict, Optional

import os

import torch

from . import parallel


class TrainStep:
    def __init__(self, loss_module, lr: float, weight_decay: float = 1e-6,
                 gradient_clip_norm: Optional[float] = 100.0, world_size: int = 1, seed: int = 0):
        self.lm = loss_module
        self.world_size = world_size
        self.clip = gradient_clip_norm
        self.params = [p for p in loss_module.parameters() if p.requires_grad]
        self.optimizer = torch.optim.Adam(self.params, lr=lr, weight_decay=weight_decay)
        self.device = self.params[0].device
        # plated draws (mask, E, qc_E) must agree across ranks, local draws (z_basal) must not:
        # two generators, the first seeded identically everywhere
        rank = torch.distributed.get_rank() if world_size > 1 else 0
        self.gen_plated, self.gen_local = parallel.make_generators(seed, rank, self.device)
        self.impl = "modular"
        self.decoder_variants = 2
        self.lr, self.weight_decay = lr, weight_decay
        self.step_count = 0
        self.fused = None
        # "device": the fused step draws its own noise (Philox, one launch); "torch": torch generators (4 launches)
        self.rng = os.environ.get("CRADLE_STEP_RNG", "device")
        if loss_module._use_fused(None):
            from . import hip

            self.fused = loss_module._fused_step()
            self.fused.seed_plated, self.fused.seed_local = seed, seed + 7919 * (rank + 1)
            self.impl = "fused"
            self.decoder_variants = 1
            a = self.fused.arena
            self.exp_avg = torch.zeros_like(a.flat)
            self.exp_avg_sq = torch.zeros_like(a.flat)
            self.norm_ws = torch.zeros(hip.load().cradle_adam_workspace_floats(), device=self.device)
            # N > 1: gradients are synchronised in buckets (CRADLE_BUCKETS: one by default, after the step; with 2 or
            # 4, in the order the backward pass finishes them, on a communication stream that runs under the remaining
            # backward kernels).  Two forms (CRADLE_GRAD_SYNC):
            #   "rs_ag" (default): reduce-scatter every bucket, clip + Adam on this rank's 1/N of each bucket only,
            #           all-gather the updated parameters -- the same bytes on the wire as an all-reduce, 1/N of the
            #           optimiser's HBM traffic, and collectives RCCL can run as direct exchanges over the xGMI mesh;
            #   "allreduce": all-reduce every bucket, full update on every rank (the reference's DDP semantics as
            #           they are usually implemented).  CRADLE_OVERLAP_ALLREDUCE=0: one all-reduce after the step.
            # All three give bit-identical parameters (tests/test_ddp_gpu.py).
            self.overlap = world_size > 1 and os.environ.get("CRADLE_OVERLAP_ALLREDUCE", "1") != "0"
            self.group_world = parallel.group_size() if world_size > 1 else 1
            self.group_rank = torch.distributed.get_rank() if self.group_world > 1 else 0
            shardable = 64 % (4 * self.group_world) == 0  # arena segments are multiples of 64 floats
            self.grad_sync = os.environ.get("CRADLE_GRAD_SYNC", "rs_ag" if shardable else "allreduce")
            if self.grad_sync == "rs_ag" and (not shardable or not self.overlap):
                self.grad_sync = "allreduce"
            if world_size > 1:
                self._setup_regions(a, shardable)
                import warnings

                queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
                if queues < 6:  # measured (profiles/r02_ddp_one_rank_emulation.txt): 6 and more run at full speed
                    warnings.warn("multi-GPU step with GPU_MAX_HW_QUEUES=%s: the step's streams share hardware queues "
                                  "with the communicator's and run one after the other (0.2-0.3 ms per step); export "
                                  "GPU_MAX_HW_QUEUES=16 before the process starts (bench.py and train.py do)"
                                  % os.environ.get("GPU_MAX_HW_QUEUES", "4 (the runtime's default)"))
                if queues >= 8 and int(os.environ.get("CRADLE_BUCKETS", "1")) > 1:
                    warnings.warn("CRADLE_BUCKETS > 1 with GPU_MAX_HW_QUEUES >= 8: in the one-rank emulation on this "
                                  "runtime every form that reduces a bucket on the communication stream took twice "
                                  "the step time in that combination (DESIGN 6); measure before relying on it")
            if self.overlap:
                self.fused.bucket_events = (hip.EventPair(), hip.EventPair())
                self.fused.bucket_events_used = {i for i in self.bucket_waits if i is not None}
                self.comm_stream = torch.cuda.Stream(device=self.device)

    def _setup_regions(self, a, shardable):
        """Buckets in backward-completion order and their per-rank pieces; region index = rank * 4 + bucket."""
        import ctypes

        from . import hip

        o_enc, o_dec, o_out = a.offsets["enc.W0"], a.offsets["dec.W0"], a.offsets["dec.Wh"]
        self.buckets = [(o_out, a.numel), (o_dec, o_out), (o_enc, o_dec), (0, o_enc)]
        self.bucket_waits = [0, 1, 2, None]  # index into (ev[0].begin, ev[0].end, ev[1].begin); None: end of the step
        # Default ONE bucket: all of it after the step, on the step's own stream.  CRADLE_BUCKETS=2 reduces the output
        # layer (a third of the parameters, final 0.3 ms before the step ends) under the rest of the backward pass on
        # the communication stream, 4 cuts per layer group.  On this runtime the communication stream's wait for the
        # step's mid-backward event stalls the step (2.1 instead of 1.0 ms) whenever that stream has a hardware queue
        # to itself, and runs well (1.02 ms) only under some sharings of queues (GPU_MAX_HW_QUEUES=6 with the step on
        # the null stream): measured with a one-rank RCCL group, tools/bench_ddp_one_rank.py, DESIGN 6.  One bucket
        # never touches that stream: 0.98 ms at 6 ... 32 queues.
        nb = int(os.environ.get("CRADLE_BUCKETS", "1"))
        if nb == 2:
            self.buckets, self.bucket_waits = [(o_out, a.numel), (0, o_out)], [0, None]
        elif nb == 1:
            self.buckets, self.bucket_waits = [(0, a.numel)], [None]
        gw = self.group_world if shardable else 1
        self.piece_world = gw
        pieces = [[(s + r * ((e - s) // gw), (e - s) // gw) for (s, e) in self.buckets] for r in range(gw)]
        flat = [pc for rank_pieces in pieces for pc in rank_pieces]
        arr = lambda vals: (ctypes.c_int64 * len(vals))(*vals)
        self._all_regions = (arr([o for o, _ in flat]), arr([n for _, n in flat]), len(flat))
        mine = pieces[self.group_rank if shardable else 0]
        self._my_regions = (arr([o for o, _ in mine]), arr([n for _, n in mine]), len(mine))
        lib = hip.load()
        self.region_ws = torch.zeros(lib.cradle_adam_regions_workspace_floats(), device=self.device)
        self.region_sums = torch.zeros(len(flat), device=self.device)
        self.norm_out = torch.zeros(2, device=self.device)  # [|g|, updates skipped on every rank since the last look]
        self.fused.defer_flag_check = True  # ranks raise together (check_flags), never one alone inside FusedElboStep.run

    def _adam_regions(self, a, regions, all_sums):
        from . import hip

        off, ln, n = regions
        hip.check(
            hip.load().cradle_adam_step_regions(
                a.flat.data_ptr(), a.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), off, ln, n,
                all_sums.data_ptr(), all_sums.numel(), self.lr, 0.9, 0.999, 1e-8, self.weight_decay, self.step_count,
                float(self.clip) if self.clip is not None else -1.0, 1.0 / self.world_size, self.norm_out.data_ptr(),
                self.fused.flags.data_ptr(), hip.stream()),
            "cradle_adam_step_regions")

    def _region_sumsq(self, a, regions, out):
        from . import hip

        off, ln, n = regions
        hip.check(hip.load().cradle_grad_region_sumsq(a.grad.data_ptr(), off, ln, n, self._all_regions[2], out.data_ptr(),
                                                      self.region_ws.data_ptr(), hip.stream()), "cradle_grad_region_sumsq")

    def draw_noise(self, P, N):
        g = self.lm.guide
        args = (self.gen_plated, self.gen_local, P, N, g.n_treatments, g.n_latent, g.n_qc, self.device)
        return parallel.draw_noise(*args), parallel.draw_noise(*args)

    def allreduce_grads(self):
        parallel.allreduce_mean_grads_(self.params, self.world_size)

    def _fused_call(self, batch, n_particles):
        from . import hip

        f = self.fused
        out = f.run(
            X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"],
            qc_obs_counts=batch["qc_obs_counts"], cfX=batch["cfX"], library_size=batch["library_size"],
            n_particles=n_particles, treatments_present=batch.get("treatments_present"),
            **({"gen_plated": self.gen_plated, "gen_local": self.gen_local} if self.rng == "torch" else {}))
        a = f.arena
        self.step_count += 1
        if self.world_size > 1:
            self._sync_and_update(a)
            self.last_out = out
            return out[0]
        hip.check(
            hip.load().cradle_adam_step(
                a.flat.data_ptr(), a.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), a.numel,
                self.lr, 0.9, 0.999, 1e-8, self.weight_decay, self.step_count,
                float(self.clip) if self.clip is not None else -1.0, 1.0 / self.world_size, self.norm_ws.data_ptr(),
                f.flags.data_ptr(), hip.stream()),
            "cradle_adam_step")
        self.last_out = out
        return out[0]

    def _sync_and_update(self, a):
        """Gradient synchronisation + clip + Adam for more than one rank (see __init__ for the three forms)."""
        from . import hip

        gw, rk = self.piece_world, self.group_rank
        shard = self.grad_sync == "rs_ag"
        # A rank whose step rejected its batch (flags != 0: invalid gradients) puts a NaN into the first element of
        # what is summed at the end of the step; the squared norm is then not finite on EVERY rank and
        # cradle_adam_step_regions skips the update everywhere -- no rank applies a sum that holds invalid gradients,
        # and no rank updates alone.
        lib = hip.load()
        last = [s for (s, e), w in zip(self.buckets, self.bucket_waits) if w is None] if self.overlap else [0]
        hip.check(lib.cradle_grad_poison_if_flagged(a.grad.data_ptr() + 4 * last[0], self.fused.flags.data_ptr(),
                                                    hip.stream()), "cradle_grad_poison_if_flagged")
        if not self.overlap:
            parallel.allreduce_sum_(a.grad, self.world_size)  # one RCCL call on the flat gradient arena
        else:
            # buckets in the order the backward pass finishes them (decoder output layer, rest of the decoder, basal
            # encoder, the rest), each as soon as the event the step recorded for it has fired
            ev = self.fused.bucket_events
            evs = (ev[0].begin, ev[0].end, ev[1].begin)
            waits = [None if i is None else evs[i] for i in self.bucket_waits]
            cur = torch.cuda.current_stream()
            sync = (lambda g: parallel.reduce_scatter_inplace_(g, rk, gw)) if shard else parallel._all_reduce
            early = [(b, w) for b, w in zip(self.buckets, waits) if w is not None]
            if early:
                with torch.cuda.stream(self.comm_stream):
                    for (s, e), w in early:
                        hip.check(lib.cradle_stream_wait_event(self.comm_stream.cuda_stream, w), "stream_wait_event")
                        sync(a.grad[s:e])
            # the bucket that is final only when the step is (q(M), q(E|M), q(qc_E)) has nothing to run under: it goes
            # on the step's own stream, two stream hand-overs fewer than through the communication stream (the process
            # group runs its collectives in issue order whatever stream they come from)
            for (s, e), w in zip(self.buckets, waits):
                if w is None:
                    sync(a.grad[s:e])
            if early:
                cur.wait_stream(self.comm_stream)
        # |g|^2 = in-order sum of the region sums (bucket x rank), whoever computes them
        if shard:
            mine = self.region_sums[:self._my_regions[2]]
            self._region_sumsq(a, self._my_regions, mine)
            all_sums = parallel.all_gather_small(mine, gw)
            self._adam_regions(a, self._my_regions, all_sums)
            parallel.all_gather_inplace_many_([a.flat[s:e] for (s, e) in self.buckets], rk, gw)
        else:
            self._region_sumsq(a, self._all_regions, self.region_sums)
            self._adam_regions(a, self._all_regions, self.region_sums)

    def check_flags(self):
        """Raises -- on EVERY rank, in the same call -- if any rank's step rejected its input since the last look (a cell
        with more than kmax dosages, qc outside {0,1}): such a step's update was skipped on all ranks.  One rank: the
        fused step's own check.  Collective for N > 1: every rank must call it at the same point (the step does so
        on its first training step and every 256th; trainer.fit at the end of every epoch and after validation)."""
        if self.fused is None:
            return
        if self.world_size == 1:
            self.fused.check_flags()
            return
        from . import hip

        mine = None
        try:
            self.fused.check_flags()
        except hip.CradleHipError as e:
            mine = str(e)
        skipped = float(self.norm_out[1].item())
        self.norm_out[1] = 0.0
        t = torch.tensor([1.0 if mine else 0.0, skipped], device=self.device)
        parallel._all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        if float(t[0]) > 0 or float(t[1]) > 0:
            raise hip.CradleHipError(
                f"rank {self.group_rank}: " + (mine or "another rank's step rejected its input") +
                f"; {int(t[1])} parameter update(s) were skipped on all {self.world_size} ranks")

    def __call__(self, batch: Dict[str, torch.Tensor], n_particles: int) -> torch.Tensor:
        if self.fused is not None:
            loss = self._fused_call(batch, n_particles)
            # collective: every rank must get here in the same call, so the schedule hangs on the count of TRAINING steps
            # (equal on all ranks: the loaders cut equal shards) and not on the fused step's own call counter, which
            # also counts a rank's validation batches (shards of a validation set can differ by one batch)
            if self.world_size > 1 and (self.step_count == 1 or self.step_count % 256 == 0):
                self.check_flags()
            return loss
        noise, cf_noise = self.draw_noise(n_particles, batch["X"].shape[0])
        self.optimizer.zero_grad(set_to_none=False)
        loss, metrics = self.lm.loss(
            X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"],
            qc_obs_counts=batch["qc_obs_counts"], cfX=batch["cfX"],
            condition_values={"library_size": batch["library_size"]}, n_particles=n_particles, noise=noise,
            cf_noise=cf_noise)
        loss.backward()
        self.allreduce_grads()
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip)
        self.optimizer.step()
        self.last_metrics = metrics
        return loss.detach()


def fit(module, train_loader, val_loader=None, *, max_epochs: int = 1, max_steps: Optional[int] = None,
        gradient_clip_val: Optional[float] = 100.0, seed: int = 0, log_every: int = 0, callbacks=(), step_callbacks=()):
    """What ``pl.Trainer(...).fit(lightning_module, datamodule)`` does in train_norman.py:89-104
    for this module, without pytorch_lightning: epochs over ``train_loader`` with ``TrainStep``
    (fused HIP step + RCCL all-reduce + clip + Adam), then ``validation_step`` over
    ``val_loader``.  Returns a history list of dicts (``epoch``, ``train/loss``, ``val/loss`` ...).

    ``module`` is a ``PerturbationLightningModule``; loaders yield batch dicts on the module's
    device (``cradle_vae_amd.data.DeviceBatchLoader``).  ``callbacks``: callables
    ``cb(module, epoch, record)`` run after each epoch (checkpointing, early stopping by raising
    StopIteration); every key of ``record`` is the same on all ranks (train/loss and val/loss are all-reduced), so
    ranks that act on it act alike.  ``step_callbacks``: callables ``cb(module, epoch, steps)`` run after every training
    step (ModelCheckpoint(every_n_train_steps=...), train_norman.py:240-247)."""
    from . import parallel

    world, _ = parallel.world_and_rank()
    if getattr(module, "use_scheduler", False):
        raise NotImplementedError("use_scheduler=True (SGD + CyclicLR) is never enabled by the reference's entry "
                                  "points (train_norman.py:82); drive configure_optimizers() yourself for it")
    # the data set knows how many dosages a cell can carry: lets the fused step compact its treatment axis
    kmax = [ld.dataset.max_dosages_per_cell() for ld in (train_loader, val_loader)
            if hasattr(getattr(ld, "dataset", None), "max_dosages_per_cell")]
    if kmax and hasattr(module.loss_module, "max_dosages_per_cell"):
        module.loss_module.max_dosages_per_cell = max(kmax)
    stepper = TrainStep(module.loss_module, lr=module.lr, weight_decay=module.weight_decay,
                        gradient_clip_norm=gradient_clip_val, world_size=world, seed=seed)
    history, steps = [], 0
    for epoch in range(max_epochs):
        module.train()
        if hasattr(train_loader, "set_epoch"):
            train_loader.set_epoch(epoch)
        tot, cnt = None, 0
        for batch in train_loader:
            b = dict(batch)
            b["D_obs_counts"], b["qc_obs_counts"] = module.D_obs_counts_train, module.qc_obs_counts_train
            loss = stepper(b, n_particles=module.n_particles)
            tot = loss.detach().clone() if tot is None else tot + loss.detach()
            cnt += 1
            steps += 1
            if log_every and steps % log_every == 0:
                print(f"epoch {epoch} step {steps} train/loss {float(loss):.4f}", flush=True)
            for cb in step_callbacks:
                cb(module, epoch, steps)
            if max_steps is not None and steps >= max_steps:
                break
        stepper.check_flags()  # no epoch ends with a rejected batch unreported (all ranks raise together)
        train_loss = None
        if tot is not None or world > 1:  # the mean over every rank's steps: a monitored key must not be rank-local
            t = torch.stack([tot if tot is not None else torch.zeros((), device=stepper.device),
                             torch.tensor(float(cnt), device=stepper.device)])
            parallel.allreduce_sum_(t, world)
            train_loss = float(t[0] / t[1]) if float(t[1]) > 0 else None
        record = {"epoch": epoch, "steps": steps, "train/loss": train_loss}
        if val_loader is not None:
            module.eval()
            vtot, vcnt = None, 0
            for batch in val_loader:
                v = module.validation_step(batch, vcnt).detach() * batch["X"].shape[0]
                vtot = v if vtot is None else vtot + v
                vcnt += batch["X"].shape[0]
            stepper.check_flags()  # forward-only steps reject inputs too
            if vtot is not None or world > 1:  # (a rank whose validation shard is empty still takes part in the sum)
                t = torch.stack([vtot if vtot is not None else torch.zeros((), device=stepper.device),
                                 torch.tensor(float(vcnt), device=stepper.device)])
                parallel.allreduce_sum_(t, world)
                if float(t[1]) > 0:
                    record["val/loss"] = float(t[0] / t[1])
        history.append(record)
        try:
            for cb in callbacks:
                cb(module, epoch, record)
        except StopIteration:
            break
        if max_steps is not None and steps >= max_steps:
            break
    return history
