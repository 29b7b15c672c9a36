#!/bin/bash
# Kernel trace of the TRAINABLE part of one plain train step (head + decoder forward, loss, backward, optimizer): everything between
# the encoder's average pool and the next step's input layout kernel.   usage (GPU box): bash tools/prof_trainable.sh <outdir>
out=${1:-gpurun_out/prof_trainable}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o tr -- python3 $GRAFT_REPO_ROOT/tools/trainable_step.py 8 > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $out <<'PY'
import csv, glob, os, re, sys
from collections import OrderedDict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:90]
pools = [i for i, r in enumerate(rows) if "avgpool_kernel" in r["Kernel_Name"]]
s2d = [i for i, r in enumerate(rows) if "nchw_to_s2d" in r["Kernel_Name"]]
a = pools[-3]; b = min(i for i in s2d if i > a)
t0 = int(rows[a]["End_Timestamp"])
agg = OrderedDict(); tot = 0.0
for r in rows[a + 1:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = name(r); agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += d; tot += d
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print("# trainable part of one plain step: %d launches, %.1f us of kernel time, span %.1f us" % (b - a - 1, tot, span))
print("kernel,launches,total_us")
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%.1f' % (k, n, d))
PY
