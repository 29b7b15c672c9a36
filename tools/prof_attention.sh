#!/bin/bash
# rocprofv3 kernel trace of the soft-attention training step (tools/time_attention.py, BASELINE configs[2]): per-kernel totals -> stdout
out=${1:-gpurun_out/prof_attn}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o attn -- python3 $GRAFT_REPO_ROOT/tools/time_attention.py > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu $out/run.log | grep -E "encoder|decoder"
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/1e3:10.0f} us {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:8.1f} us avg {float(r["Percentage"]):5.1f} %  {r["Name"][:100]}')
PY
