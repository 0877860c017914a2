"""Set-up passes of the data module on a device-resident data set (csrc/dataprep.hip) against the torch forms they
replace: per-gene log statistics, observation counts, counterfactual inputs (per-gene median over peers).
python tools/bench_dataprep.py [cells] [genes] [treatments] [peers]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import data, ops

n, G, T, K = (int(a) for a in (sys.argv[1:5] + ["32768", "5000", "105", "50"][len(sys.argv) - 1:]))
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
X = torch.poisson(torch.rand(n, G, device=dev, generator=g) * 3).to(torch.int16)
codes = torch.randint(0, T, (n,), device=dev, generator=g)
fail = torch.rand(n, device=dev, generator=g) < 0.3


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, out


def torch_stats():
    y = torch.log(X.float() + 1)
    return y.mean(0), y.std(0)


t_hip, (m1, s1) = timed(lambda: data.log_mean_std(X))
t_ref, (m0, s0) = timed(torch_stats)
print(f"log statistics {n} x {G} ({X.dtype}): HIP one pass {t_hip:.2f} ms ({X.numel() * X.element_size() / t_hip / 1e6:.0f} GB/s of "
      f"stored counts), torch log/mean/std {t_ref:.2f} ms; max |mean diff| {float((m1 - m0).abs().max()):.2e}, "
      f"max |std diff| {float((s1 - s0).abs().max()):.2e}")

D = torch.zeros(n, T, device=dev)
D[torch.arange(n, device=dev), codes] = 1.0
t_hip, c1 = timed(lambda: ops.obs_counts_dense(D))
t_ref, c0 = timed(lambda: (D != 0).sum(0))
print(f"observation counts {n} x {T}: HIP {t_hip:.3f} ms, torch {t_ref:.3f} ms, equal {bool((c1 == c0).all())}")

# peers table as build_counterfactual_inputs makes it (draws from one generator), then the kernel alone against the
# torch form of the same table (gather in slabs, torch.median over the peer axis: lower median, as the reference's)
peers = torch.full((n, K), -1, device=dev, dtype=torch.int32)
for t in range(T):
    members = torch.nonzero(codes == t).squeeze(1)
    fails = torch.nonzero((codes == t) & fail).squeeze(1)
    if fails.numel():
        k = min(K, fails.numel())
        peers[members, :k] = fails[torch.randint(0, fails.numel(), (members.numel(), k), device=dev, generator=g)].int()


def torch_median():
    out = torch.zeros((n, G), device=dev, dtype=X.dtype)
    step = max(1, (1 << 28) // (K * G))
    for s in range(0, n, step):
        p = peers[s:s + step].long()
        valid = p[:, 0] >= 0  # (every row here has either K peers or none)
        if valid.any():
            out[s:s + step][valid] = X[p[valid]].median(dim=1).values
    return out


t_hip, o1 = timed(lambda: ops.peer_median(X, peers), reps=3)
t_ref, o0 = timed(torch_median, reps=3)
print(f"counterfactual inputs {n} cells x {K} peers x {G} genes: HIP peer_median {t_hip:.2f} ms "
      f"({n * K * G * X.element_size() / t_hip / 1e6:.0f} GB/s of gathered counts), torch gather + median {t_ref:.2f} ms, "
      f"equal {bool((o1 == o0).all())}")
t_all, _ = timed(lambda: data.build_counterfactual_inputs(X, codes, fail, n_peers=K, generator=g), reps=3)
print(f"build_counterfactual_inputs end to end (peer table by {T} treatments + kernel): {t_all:.2f} ms")
