#!/bin/bash
# Round profile on the GPU box: (1) rocprofv3 kernel trace + stats of bench.py, (2) the two PMC passes (FETCH_SIZE, WRITE_SIZE:
# separate runs, counters only) of the unpipelined bench and of the greedy decode, aggregated by tools/pmc_traffic.py.
# usage: tools/profile_round.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/profile_round}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- $B > $out/bench_line_profiled.json 2> $out/trace.err
echo "[profile] trace done"
P="python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pipeline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- $P > $out/pmc_fetch.log 2>&1
echo "[profile] fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- $P > $out/pmc_write.log 2>&1
echo "[profile] write pass done"
D="python3 $GRAFT_REPO_ROOT/tools/time_decode.py"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/dec_fetch -o f -- $D > $out/dec_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/dec_write -o w -- $D > $out/dec_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/dec_trace -o dec -- $D > $out/dec_trace.log 2>&1
echo "[profile] decode passes done"
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write > $out/pmc_hbm_traffic.csv
python3 tools/pmc_traffic.py $out/dec_fetch $out/dec_write > $out/pmc_decode_hbm_traffic.csv
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
cp $(find $out/dec_trace -name "*kernel_stats.csv" | head -1) $out/decode_kernel_stats.csv
rm -rf $out/pmc_fetch $out/pmc_write $out/dec_fetch $out/dec_write $out/trace/*trace.csv $out/dec_trace
head -12 $out/pmc_hbm_traffic.csv; head -8 $out/bench_kernel_stats.csv | cut -c1-160
