"""Race hunt: every reduction of the fused step and of the sampler has a fixed order, so the same inputs and the same
RNG key must give the same bits however the three streams and the workgroups interleave.  Repeats each case and counts
repeats whose outputs differ from the first.  python tools/determinism_soak.py [repeats]"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cradle_vae_amd import ops, synthetic as syn

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
bad_total = 0
cases = [("norman", {}, "f32"), ("norman", {}, "bf16x3"), ("norman", dict(n_phenos=5003), "f32"), ("replogle", {}, "f32"),
         ("replogle", dict(n_phenos=8001, n_latent=98), "f32"), ("adamson", {}, "f32")]
for base, repl, precision in cases:
    shape = dataclasses.replace(syn.SHAPES[base], **repl)
    lm, batch = bench.build(shape, dev, seed=7)
    lm.precision = precision
    fs = lm._fused_step()
    kw = dict(X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"], qc_obs_counts=batch["qc_obs_counts"],
              cfX=batch["cfX"], library_size=batch["library_size"], n_particles=shape.n_particles, rng_key=(3, 4, 9),
              treatments_present=batch["treatments_present"])
    first, bad = None, 0
    for i in range(reps):
        out = fs.run(**kw)
        torch.cuda.synchronize()
        cur = (out[:7].clone(), fs.arena.grad.clone())
        if first is None:
            first = cur
        elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])):
            bad += 1
    bad_total += bad
    print(f"{base} {repl} {precision}: {reps} repeats, {bad} differ from the first; loss {float(first[0][0]):.4f}", flush=True)
    del fs, lm, batch
g = torch.Generator(device=dev).manual_seed(0)
for G, C, P, logc in ((5000, 237, 64, 0.0), (5001, 100, 64, -1.0), (301, 30, 40, 0.0), (16384, 16, 32, 0.0)):
    logits = 0.5 * torch.randn(P * C, G, device=dev, generator=g)
    consts = ops.nb_gene_consts(logc + 0.5 * torch.randn(G, device=dev, generator=g))
    lib = 1e4 * torch.ones(1, device=dev)
    first, bad = None, 0
    for i in range(max(20, reps // 4)):
        s, x = ops.nb_sample_accumulate(logits, lib, consts, P, C, method="perturbseq", sampler="nb", seed=5, return_samples=True)
        torch.cuda.synchronize()
        cur = (s.clone(), x.clone())
        if first is None:
            first = cur
        elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])):
            bad += 1
    bad_total += bad
    print(f"sampler G={G} C={C} P={P} log c {logc}: {bad} repeats differ; mean draw {float(first[1].mean()):.4f}", flush=True)
print("TOTAL differing repeats:", bad_total)
sys.exit(1 if bad_total else 0)
