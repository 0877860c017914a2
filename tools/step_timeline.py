"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV (one step between two adam_kernel launches).
python tools/step_timeline.py gpurun_out/prof/r_kernel_trace.csv [step_index] [--full]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else len(idx) // 2
step = rows[idx[k] + 1: idx[k + 1] + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"step {k}: wall {(t1 - t0) / 1e3:.1f} us, GPU busy (union) {busy / 1e3:.1f} us, sum of kernels {sum(e - s for s, e in iv) / 1e3:.1f} us, {len(step)} kernels")
agg = defaultdict(lambda: [0, 0.0])
for r in step:
    name = r["Kernel_Name"].split("(")[0][:60]
    agg[name][0] += 1
    agg[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t:8.1f} us  x{n:3d}  {name}")
if "--full" in sys.argv:
    for r in step:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3
        e = (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{s:8.1f} {e:8.1f} {e - s:6.1f} q{r['Queue_Id']:>2s} wg{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} {r['Kernel_Name'][:80]}")
