"""Phase times of the fused step on the GPU's own clock, un-profiled and in the steady state: bench.py is run once per
checkpoint with CRADLE_PROBE=k (csrc/step.hip), which moves the event pair bench.py reads as `largest_gemm.launch_us`
to [checkpoint k, end of the output layer] (forward, k = 0..6) or [checkpoint k, all streams joined] (backward,
k = 10..16; 17 / 18: how long the caller's stream waits for side stream 1 / 2 after its own last kernel).
python tools/probe_sweep.py [steps] [shape]"""
import json
import os
import subprocess
import sys

steps = sys.argv[1] if len(sys.argv) > 1 else "200"
shape = sys.argv[2] if len(sys.argv) > 2 else "norman"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {0: "step start", 1: "noise drawn, forks made", 2: "q(M) sample + embedding encoder done", 3: "gather + offset GEMM done, x Wx^T and qc_E Wqc^T met",
         4: "encoder first layer combine + heads GEMM done", 5: "z head done, side streams met", 6: "decoder hidden layers done (output layer starts)",
         10: "K1 done (backward starts)", 11: "decoder data gradients done", 12: "[dz|doff] GEMM + z head backward done", 13: "encoder heads data gradient done",
         14: "combine backward + offsets data gradient done", 15: "q(E|M) backward + embedding encoder data gradients done",
         16: "d mask GEMM + q(M) backward + last leaves done (caller's stream has nothing left)", 17: "wait for side stream 1", 18: "wait for side stream 2"}
res = {}
for k in list(range(0, 7)) + list(range(10, 19)):
    env = dict(os.environ, CRADLE_PROBE=str(k))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", steps, "--warmup", "20", "--no-cpu-baseline", "--no-secondary", "--shape", shape],
                         env=env, capture_output=True, text=True, timeout=300)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    res[k] = (d["mfma"]["largest_gemm"]["launch_us"], d["ms_per_step"])
    print(f"probe {k:2d}: {res[k][0]:7.1f} us to the end point   (step {res[k][1]:.4f} ms)   <- {names[k]}", flush=True)
print("\nforward phases (us):")
for a, b in zip(range(0, 6), range(1, 7)):
    print(f"  {names[a]:70s} -> {res[a][0] - res[b][0]:6.1f}")
print(f"  {'decoder output layer':70s} -> {res[6][0]:6.1f}")
print("backward phases (us):")
for a, b in zip(range(10, 16), range(11, 17)):
    print(f"  {names[a]:70s} -> {res[a][0] - res[b][0]:6.1f}")
print(f"  {'tail: caller stream waits for the side streams':70s} -> {res[16][0]:6.1f}   (stream 1 alone {res[17][0]:.1f}, stream 2 alone {res[18][0]:.1f})")
