"""Timeline of the software-pipelined train loop from a rocprofv3 kernel trace of tools/pipelined_steps.py: per steady-state step, when
the trainable chain (everything that is not an encoder kernel) runs, how long its kernels take against their plain-step durations, and how
long the chain waits between its kernels.   usage: python tools/pipeline_timeline.py <rocprof dir>"""
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ENC = ("conv", "stem_pool", "bn_act_reg", "nchw_to_s2d", "stem_weight", "avgpool", "bn_update_all", "bn_reduce", "igemm_s3")
is_enc = lambda n: any(k in n for k in ENC) and "igemm_kernel" not in n
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
print("steps seen (sgd launches):", len(sgd))
for a, b in zip(sgd[6:10], sgd[7:11]):
    seg = rows[a + 1:b + 1]
    tr = [r for r in seg if not is_enc(r["Kernel_Name"])]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr) / 1e3
    gaps = [(int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3 for x, y in zip(tr[:-1], tr[1:])]
    first = (int(tr[0]["Start_Timestamp"]) - t0) / 1e3
    print("step: %.1f us sgd-to-sgd; trainable chain: %d kernels, %.1f us of kernel time (sum of durations), first kernel %.1f us after the previous sgd, "
          "gaps between chain kernels: sum %.1f us, max %.1f us; encoder kernels in the window: %d"
          % ((t1 - t0) / 1e3, len(tr), busy, first, sum(gaps), max(gaps), len(seg) - len(tr)))
# slowdown of chain kernels by family
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for r in rows[sgd[6]:sgd[10]]:
    n = r["Kernel_Name"]
    if is_enc(n): continue
    k = re.sub(r"\(anonymous namespace\)::|^void |_ZN12_GLOBAL__N_1\d+", "", n)[:60]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("chain kernels over 4 steps (launches, mean us):")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-62s %4d  %7.1f" % (k, c, d / c))
