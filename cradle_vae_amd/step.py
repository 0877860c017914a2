"""Host side of the fused ELBO step (``cradle_elbo_step`` in ``csrc/step.hip``).

* ``ParamArena`` keeps every trainable tensor of the loss module as a view into ONE flat fp32
  buffer (and its gradient as a view into a second one): the gradient all-reduce is a single
  RCCL call on one contiguous buffer and clip + Adam are two kernels (``cradle_adam_step``).
  The ``mean_encoder`` / ``log_var_encoder`` pairs are laid out back to back so that both heads
  are one (2L, H) GEMM.  State-dict keys are untouched.
* ``FusedElboStep`` fills the C descriptor and launches the step; ``loss()`` of the loss module
  returns its result wrapped in an autograd node so that ``loss.backward()`` (Lightning) still
  works -- the gradients were already computed by the HIP backward.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Tuple

import torch

from . import hip

ALIGN = 64  # floats (256 B): keeps every segment 16-byte aligned for the vector-load GEMM path
OUT_KEYS = ["loss", "loss_term_reconstruction", "loss_term_mask", "loss_term_E", "loss_term_qc_E", "loss_term_z_basal",
            "cf_KLdivergence"]


def unsupported_reason(lm):
    """None when the fused step runs this loss module, else what keeps it on the modular path (one sentence)."""
    from .guide import cradle_vae_CorrelatedNormalGuide
    from .mlp import LibraryGammaPoissonSharedConcentrationResidualMLP
    from .model import cradle_vae_Model

    m, g = lm.model, lm.guide
    if not isinstance(m, cradle_vae_Model) or not isinstance(g, cradle_vae_CorrelatedNormalGuide):
        return f"model / guide classes {type(m).__name__} / {type(g).__name__} are not the ones it implements"
    if m.sum_mode:
        return "sum_mode=True"
    if not isinstance(m.decoder, LibraryGammaPoissonSharedConcentrationResidualMLP):
        return f"likelihood {getattr(m, 'likelihood_key', '?')!r} (it implements library_nb)"
    if m.n_phenos < 4:
        return f"n_phenos={m.n_phenos} < 4"
    if not next(lm.parameters()).is_cuda:
        raise hip.CradleHipError("cradle_vae_amd runs on the GPU only: move the module to a cuda device (no CPU path)")
    return None


def supports(lm) -> bool:
    return unsupported_reason(lm) is None


class ParamArena:
    def __init__(self, lm):
        self.lm = lm
        g, m = lm.guide, lm.model
        self.groups: List[Tuple[str, List[torch.nn.Parameter]]] = []

        def gauss(name, mlp):
            lins = mlp.mlp.linears()
            for k, lin in enumerate(lins):
                self.groups.append((f"{name}.W{k}", [lin.weight]))
                self.groups.append((f"{name}.b{k}", [lin.bias]))
            self.groups.append((f"{name}.Wh", [mlp.mean_encoder.weight, mlp.log_var_encoder.weight]))
            self.groups.append((f"{name}.bh", [mlp.mean_encoder.bias, mlp.log_var_encoder.bias]))

        self.groups.append(("q_mask_logits", [g.param_dict["q_mask_logits"]]))
        gauss("emb", g.embedding_encoder)
        gauss("qce", g.qc_encoder)
        gauss("enc", g.z_basal_encoder)
        dec = m.decoder
        for k, lin in enumerate(dec.mlp.linears()):
            self.groups.append((f"dec.W{k}", [lin.weight]))
            self.groups.append((f"dec.b{k}", [lin.bias]))
        self.groups.append(("dec.Wh", [dec.normalized_mean_decoder[0].weight]))
        self.groups.append(("dec.bh", [dec.normalized_mean_decoder[0].bias]))
        self.groups.append(("log_c", [dec.log_concentration]))
        listed = {id(p) for _, ps in self.groups for p in ps}
        extra = [n for n, p in lm.named_parameters() if id(p) not in listed]
        if extra:
            raise hip.CradleHipError(f"ParamArena: parameters outside the known layout: {extra}")
        self.flat = self.grad = None
        self.offsets: Dict[str, int] = {}
        self._views: List[Tuple[torch.nn.Parameter, int]] = []
        self.rebuild()

    def rebuild(self):
        dev = next(self.lm.parameters()).device
        off = 0
        self.offsets.clear()
        self._views.clear()
        for name, ps in self.groups:
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            self.offsets[name] = off
            for p in ps:
                self._views.append((p, off))
                off += p.numel()
        self.numel = (off + ALIGN - 1) // ALIGN * ALIGN
        flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        for p, o in self._views:
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).to(torch.float32))
        self.flat = flat
        self.grad = torch.zeros_like(flat)
        for p, o in self._views:
            p.data = flat[o:o + p.numel()].view(p.shape)
        self.base = flat.data_ptr()
        self.gbase = self.grad.data_ptr()

    def valid(self) -> bool:
        base = self.base
        for p, o in self._views:
            if p.data_ptr() != base + 4 * o:
                return False
        return True

    def ptr(self, name: str) -> int:
        return self.base + 4 * self.offsets[name]

    def gptr(self, name: str) -> int:
        return self.gbase + 4 * self.offsets[name]

    def grad_views(self):
        return [self.grad[o:o + p.numel()].view(p.shape) for p, o in self._views]

    def params(self):
        return [p for p, _ in self._views]


class _FusedLoss(torch.autograd.Function):
    """Autograd node for Lightning-style callers: forward value = the loss computed by the HIP
    step, backward = the gradients the HIP step already wrote, scaled by the incoming gradient."""

    @staticmethod
    def forward(ctx, loss, arena, *params):
        ctx.arena = arena
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        grads = tuple(g * grad_out for g in ctx.arena.grad_views())
        return (None, None) + grads


class FusedElboStep:
    def __init__(self, lm, kmax: int = 8):
        self.lm = lm
        self.lib = hip.load()
        self.arena = ParamArena(lm)
        self.kmax = kmax
        self.desc = hip.StepDesc()
        self.workspace = None
        self.out = None
        self._noise = None
        self._calls = 0
        self._shape_key = None
        self._states = {}
        self.k1_events = None  # list collecting hip.EventPair around K1 when profiling (bench.py)
        self.gemm_events = None  # same around the decoder output-layer GEMM (the largest of the step)
        # timed HIP events are barrier packets (~5 us each between dependent kernels): bench.py samples
        # every `event_every`-th step instead of paying four of them in every step
        self.event_every = 1
        # hip.EventPair whose two events fire when the decoder / basal-encoder gradients are final
        # (trainer.TrainStep starts their all-reduce while the rest of the backward pass runs)
        self.bucket_events = None
        self.bucket_events_used = None  # indices of ev_bucket the caller waits for (None: all three)
        self._checked_states = set()
        # More than one data-parallel rank: a rank must not raise on its own (the others would wait for it in the next
        # collective until the watchdog fires).  trainer.TrainStep sets this and checks on all ranks at once instead
        # (TrainStep.check_flags: local flags + the count of updates every rank skipped, met in one small all-reduce).
        self.defer_flag_check = False
        # Random draws of the step.  Default: the HIP step draws its own noise (Philox4x32-10 in one launch,
        # csrc/elementwise.hip draw_noise_kernel) keyed by these seeds -- seed_plated identical on every
        # data-parallel rank (mask, E, qc_E are shared latent variables), seed_local per rank (z_basal) -- and by
        # rng_step, advanced every call.  Passing torch generators to run() draws with torch instead (four more
        # launches on the step's critical path); passing noise tensors injects them (parity tests).
        self.seed_plated, self.seed_local, self.rng_step = 0, 1, 0

    # ------------------------------------------------------------------ descriptor
    def _select_state(self, P, N):
        """Per-(P, N) descriptor, outputs and noise buffers are kept, so that alternating shapes
        (training / validation batches) do not rebuild them."""
        key = (P, N, self.arena.base)
        st = self._states.get(key)
        if st is None:
            if len(self._states) >= 4:
                self._states.pop(next(iter(self._states)))
            self.desc = hip.StepDesc()
            self._fill_static(P, N)
            st = dict(desc=self.desc, out=self.out, flags=self.flags, noise=self._noise, bufs=self._bufs)
            self._states[key] = st
        else:
            self.desc, self.out, self.flags, self._noise, self._bufs = (
                st["desc"], st["out"], st["flags"], st["noise"], st["bufs"])
        self._shape_key = key
        if self.desc.workspace != self.workspace.data_ptr():  # the shared workspace grew meanwhile
            self.desc.workspace, self.desc.workspace_bytes = self.workspace.data_ptr(), self.workspace.numel() * 4

    def _fill_static(self, P, N):
        lm, d, a = self.lm, self.desc, self.arena
        g, m = lm.guide, lm.model
        d.P, d.N, d.T, d.L, d.Q, d.G = P, N, g.n_treatments, g.n_latent, g.n_qc, g.n_phenos
        d.He = g.z_basal_encoder.mean_encoder.in_features
        d.Hm = g.embedding_encoder.mean_encoder.in_features
        d.Hd = m.decoder.normalized_mean_decoder[0].in_features
        d.kmax = min(self.kmax, d.T)
        d.q_mask_logits, d.d_q_mask_logits = a.ptr("q_mask_logits"), a.gptr("q_mask_logits")
        d.log_concentration, d.d_log_concentration = a.ptr("log_c"), a.gptr("log_c")
        for name, mlp in (("emb", g.embedding_encoder), ("qce", g.qc_encoder), ("enc", g.z_basal_encoder),
                          ("dec", m.decoder)):
            n_layers = len(mlp.mlp.layers)
            if n_layers > hip.MAX_LAYERS:
                raise hip.CradleHipError(f"{name}: {n_layers} layers exceed CRADLE_MAX_LAYERS")
            for tgt, fn in ((getattr(d, name), a.ptr), (getattr(d, "d_" + name), a.gptr)):
                tgt.n_layers = n_layers
                for k in range(n_layers):
                    tgt.W[k], tgt.b[k] = fn(f"{name}.W{k}"), fn(f"{name}.b{k}")
                tgt.W_head, tgt.b_head = fn(f"{name}.Wh"), fn(f"{name}.bh")
        self._bufs = [t.contiguous().float() for t in (
            m.p_E_loc, m.p_E_scale, m.p_mask_probs, m.p_qc_E_loc, m.p_qc_E_scale,
            g.normalization_module.log_mean, g.normalization_module.log_scale)]
        (d.p_E_loc, d.p_E_scale, d.p_mask_probs, d.p_qc_E_loc, d.p_qc_E_scale, d.log_mean,
         d.log_scale) = [hip.ptr(t) for t in self._bufs]
        d.temperature = float(g.gs_temperature.item())
        dev = a.flat.device
        d.workspace, d.workspace_bytes = None, 0
        need = self.lib.cradle_elbo_step_workspace_bytes(ctypes.byref(d))
        if need <= 0:
            raise hip.CradleHipError("cradle_elbo_step_workspace_bytes failed")
        if self.workspace is None or self.workspace.numel() * 4 < need:
            self.workspace = torch.empty((need + 3) // 4, device=dev, dtype=torch.float32)
        d.workspace, d.workspace_bytes = self.workspace.data_ptr(), self.workspace.numel() * 4
        self.out = torch.zeros(8, device=dev, dtype=torch.float32)
        self.flags = torch.zeros(1, device=dev, dtype=torch.int32)
        d.out, d.flags = self.out.data_ptr(), self.flags.data_ptr()
        T, L, Q = d.T, d.L, d.Q
        self._noise = dict(
            u_mask=torch.empty((2 * P, T, L), device=dev), eps_E=torch.empty((2 * P, T, L), device=dev),
            eps_qc_E=torch.empty((2 * P, Q, L), device=dev), eps_z=torch.empty((P, N, L), device=dev))

    def _set_noise(self, noise, cf_noise, gen_plated=None, gen_local=None, rng_key=None):
        nz, d = self._noise, self.desc
        P = d.P
        if noise is None and gen_plated is None and gen_local is None:
            d.u_mask = d.eps_E = d.eps_qc_E = d.eps_z = None
            if rng_key is not None:  # (seed_plated, seed_local, step) of this call, given by the caller
                d.rng_seed_plated, d.rng_seed_local, d.rng_step = (int(v) & (2**64 - 1) for v in rng_key)
                return
            d.rng_seed_plated, d.rng_seed_local = self.seed_plated & (2**64 - 1), self.seed_local & (2**64 - 1)
            d.rng_step = self.rng_step
            self.rng_step += 1
            return
        if noise is None:
            nz["u_mask"].uniform_(generator=gen_plated)
            nz["eps_E"].normal_(generator=gen_plated)
            nz["eps_qc_E"].normal_(generator=gen_plated)
            nz["eps_z"].normal_(generator=gen_local)
        else:
            for k in ("u_mask", "eps_E", "eps_qc_E"):
                nz[k][:P].copy_(noise[k])
                nz[k][P:].copy_(cf_noise[k])
            nz["eps_z"].copy_(noise["eps_z"])
        d.u_mask, d.eps_E, d.eps_qc_E, d.eps_z = (nz[k].data_ptr() for k in ("u_mask", "eps_E", "eps_qc_E", "eps_z"))

    # ------------------------------------------------------------------ launch
    def run(self, *, X, D, D_obs_counts, qc, qc_obs_counts, cfX, library_size, n_particles, noise=None, cf_noise=None,
            compute_grads=True, gen_plated=None, gen_local=None, rng_key=None, treatments_present=None):
        """Launches the step; returns the (8,) device tensor [loss, 5 loss terms, cf KL, -].  ``treatments_present``:
        the number of distinct treatments with a non-zero dosage in ``D`` if the caller has counted them (a host
        integer, e.g. ``DeviceBatchLoader``'s ``batch["treatments_present"]``): the treatment axis of the step is then
        that long (rounded up to 32) instead of N*kmax -- cradle_step_desc.max_present."""
        if not self.arena.valid():
            self.arena.rebuild()  # e.g. after module.to(device) or load_state_dict(assign=True)
            self._states.clear()
            self._shape_key = None
        P, N = n_particles, X.shape[0]
        self._select_state(P, N)
        d = self.desc
        if X.shape[1] != d.G or D.shape != (N, d.T) or qc.shape != (N, d.Q) or cfX.shape != X.shape:
            raise hip.CradleHipError("elbo_step: batch tensor shapes do not match the model")
        ins = (X, cfX, D, qc, library_size.reshape(-1), D_obs_counts, qc_obs_counts)
        self._keep = [t.contiguous().float() for t in ins]
        (d.X, d.cfX, d.D, d.qc, d.library_size, d.D_obs_counts, d.qc_obs_counts) = [hip.ptr(t) for t in self._keep]
        self._set_noise(noise, cf_noise, gen_plated, gen_local, rng_key)
        d.beta = float(self.lm.beta)
        d.alpha = -1.0 if self.lm.alpha is None else float(self.lm.alpha)
        d.compute_grads = int(compute_grads)
        d.max_present = 0 if treatments_present is None else max(32, (int(treatments_present) + 31) & ~31)
        d.precision = {"f32": 0, "bf16x3": 1, "bf16": 2}[getattr(self.lm, "precision", "f32")]
        sample = self._calls % max(1, self.event_every) == 0
        if self.k1_events is not None and sample:
            ev = hip.EventPair()
            self.k1_events.append(ev)
            d.ev_k1_begin, d.ev_k1_end = ev.begin, ev.end
        else:
            d.ev_k1_begin, d.ev_k1_end = None, None
        be = self.bucket_events if compute_grads else None
        for i in range(3):  # bucket_events: two hip.EventPair = (out layer, decoder) and (basal encoder, unused)
            used = be is not None and (self.bucket_events_used is None or i in self.bucket_events_used)
            d.ev_bucket[i] = (be[0].begin, be[0].end, be[1].begin)[i] if used else None
        if self.gemm_events is not None and sample:
            ev = hip.EventPair()
            self.gemm_events.append(ev)
            d.ev_gemm_begin, d.ev_gemm_end = ev.begin, ev.end
        else:
            d.ev_gemm_begin, d.ev_gemm_end = None, None
        hip.check(self.lib.cradle_elbo_step(ctypes.byref(d), hip.stream()), "cradle_elbo_step")
        self._calls += 1
        # A rejected input (flags != 0) is harmless on the device: the step's loss is NaN and cradle_adam_step,
        # handed the same flags word, skips its update.  The host only has to say why, which costs a sync: on
        # the first call of every (P, N) state (a new loader), every 256th call, in forward-only (validation /
        # test) calls, and whenever trainer.fit finishes an epoch (check_flags).
        self.check_due = self._shape_key not in self._checked_states or self._calls % 256 == 0 or not compute_grads
        if self.check_due:
            self._checked_states.add(self._shape_key)
            if not self.defer_flag_check:
                self.check_flags()
        return self.out

    def intermediate(self, name: str) -> torch.Tensor:
        """Copy of an intermediate of the last step ("mask", "E", "qc_E", "z_offset", "logits"), read out of
        the workspace through cradle_elbo_step_intermediate -- for parity tests."""
        p = ctypes.c_void_p()
        n = self.lib.cradle_elbo_step_intermediate(ctypes.byref(self.desc), name.encode(), ctypes.byref(p))
        if n < 0:
            raise hip.CradleHipError(f"unknown intermediate {name!r}")
        off = (p.value - self.workspace.data_ptr()) // 4
        torch.cuda.synchronize()
        return self.workspace[off:off + n].clone()

    def check_flags(self):
        """Raises if the last step -- or, by the sticky count in out[7], any step since the last look -- rejected
        its input.  Synchronises."""
        f, rejected = int(self.flags.item()), float(self.out[7].item())
        if rejected and not f:
            self.out[7] = 0.0
            raise hip.CradleHipError(
                f"{int(rejected)} earlier step(s) rejected their input (a cell with more than kmax={self.desc.kmax} "
                "non-zero dosages, or qc outside {0,1}); their loss was NaN and their parameter update was skipped")
        if f:
            self.out[7] = 0.0
        if f & 1:
            raise hip.CradleHipError(
                f"a cell has more than kmax={self.desc.kmax} non-zero dosages (raise kmax)"
                + (f", or the batch holds more distinct treatments than the {self.desc.max_present} slots it was "
                   "announced with (treatments_present)" if self.desc.max_present else ""))
        if f & 2:
            raise hip.CradleHipError("qc has entries outside {0,1}: the fused step scores one decoder variant per cell; "
                                     "use loss_module.step = 'modular'")

    def __call__(self, *, X, D, D_obs_counts, qc, qc_obs_counts, cfX, library_size, n_particles, noise=None,
                 cf_noise=None):
        """``loss_module.loss`` semantics: (loss with autograd node, metrics dict of device scalars)."""
        need_grads = torch.is_grad_enabled()
        # Callers of loss() (Lightning's training_step) own the randomness through torch's global generator, as in
        # the reference: the key of this call's device draws is taken from it (two host integers, no launch), so
        # torch.manual_seed / seed_everything make a run reproducible and, as under Lightning DDP, ranks seeded
        # alike draw alike.
        key = None
        if noise is None:
            r = torch.randint(0, 2**62, (2,))
            key = (int(r[0]), int(r[1]), 0)
        out = self.run(X=X, D=D, D_obs_counts=D_obs_counts, qc=qc, qc_obs_counts=qc_obs_counts, cfX=cfX,
                       library_size=library_size, n_particles=n_particles, noise=noise, cf_noise=cf_noise,
                       compute_grads=need_grads, rng_key=key)
        vals = out.clone()
        metrics = {k: vals[i] for i, k in enumerate(OUT_KEYS) if i > 0}
        if need_grads:
            loss = _FusedLoss.apply(vals[0], self.arena, *self.arena.params())
        else:
            loss = vals[0]
        return loss, metrics
