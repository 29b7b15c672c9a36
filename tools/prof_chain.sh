#!/bin/bash
# rocprofv3 kernel trace of tools/chain_bench.py <variants>: per-kernel averages -> stdout.  usage: tools/prof_chain.sh <outdir> <variants...>
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o chain -- python3 $GRAFT_REPO_ROOT/tools/chain_bench.py "$@" > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/1e3:10.0f} us {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:8.1f} us avg {float(r["Percentage"]):5.1f} %  {r["Name"][:100]}')
PY
