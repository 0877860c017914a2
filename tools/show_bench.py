"""One line of what matters from a bench.py JSON line.  python tools/show_bench.py FILE"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d["mfma"]["largest_gemm"]
line = (f"{d['ms_per_step']:.4f} ms/step  {d['value']/1e3:.1f} k cells/s  K1 {d['roofline']['launch_us']:.2f} us "
        f"(frac {d['roofline']['frac']:.3f})  largest GEMM {g['launch_us']:.1f} us ({g['frac']:.3f})  mfma {d['mfma']['frac']:.3f}")
if "ate" in d:
    line += f"  ATE {d['ate']['ms']:.1f} ms (K9 {d['ate']['roofline']['launch_us']:.0f} us)"
if "loss_rel_err_vs_oracle" in d:
    line += f"  loss err {d['loss_rel_err_vs_oracle']:.1e} terms {d['loss_terms_max_rel_err_vs_oracle']:.1e} flips {d['mask_bits_differing_from_oracle']}"
print(line)
