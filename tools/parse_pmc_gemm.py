"""Per-dispatch summary of the two rocprofv3 --pmc passes over tools/pmc_gemm.py (SQ counters).
python tools/parse_pmc_gemm.py DIR_PASS1 DIR_PASS2"""
import csv, glob, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:3]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "gemm_f32" not in k:
                continue
            key = (k[k.index("<"):k.index(">") + 1], row["Grid_Size"], row.get("Workgroup_Size"))
            a = acc[key][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
print("gemm_f32_kernel<A_KMAJOR,B_KMAJOR,TM,TN,BK,D,WK,VEC,WS>, grid threads, workgroup threads; means over the dispatches of tools/pmc_gemm.py")
print("(shapes in launch order: NT 2560x5000x400, NN 2560x400x5000 [split-K], TN 5000x400x2560 [split-K], NT 7680x400x400, NT 2560x400x400, NT 1050x400x400)")
for key, cs in acc.items():
    v = {c: s / n for c, (s, n) in cs.items()}
    wc = v["SQ_WAVE_CYCLES"]
    print(key)
    print("  waves %d  SQ_WAVE_CYCLES(quad-cycles) %.3g  SQ_BUSY_CYCLES %.3g" % (v["SQ_WAVES"], wc, v["SQ_BUSY_CYCLES"]))
    print("  fractions of wave cycles: WAIT_ANY %.2f  WAIT_INST_ANY %.2f  ACTIVE_INST_ANY %.2f  WAIT_INST_LDS %.2f" % (
        v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc, v["SQ_WAIT_INST_LDS"] / wc))
    print("  SQ_VALU_MFMA_BUSY_CYCLES %.3g (= 64 x %d MFMAs)  SQ_LDS_BANK_CONFLICT %.3g  SQ_ACTIVE_INST_LDS %.3g  SQ_ACTIVE_INST_VALU %.3g" % (
        v["SQ_VALU_MFMA_BUSY_CYCLES"], v["SQ_INSTS_MFMA"], v["SQ_LDS_BANK_CONFLICT"], v["SQ_ACTIVE_INST_LDS"], v["SQ_ACTIVE_INST_VALU"]))
    print("  instructions: VALU %.3g  MFMA %.3g  LDS %.3g  VMEM_RD %.3g  SALU %.3g" % (
        v["SQ_INSTS_VALU"], v["SQ_INSTS_MFMA"], v["SQ_INSTS_LDS"], v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_SALU"]))
