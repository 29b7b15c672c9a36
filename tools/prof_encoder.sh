#!/bin/bash
# rocprofv3 kernel trace of the ResNet-101 train-mode forward (tools/time_encoder.py): per-kernel totals -> stdout
# usage (GPU box): tools/prof_encoder.sh <outdir>
out=${1:-gpurun_out/prof_enc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o enc -- python3 $GRAFT_REPO_ROOT/tools/time_encoder.py > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the run")
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/1e3:10.0f} us {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:8.1f} us avg {float(r["Percentage"]):5.1f} %  {r["Name"][:110]}')
PY
