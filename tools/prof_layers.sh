#!/bin/bash
# Per-layer table of the routed kernels (profiles/rNN_layer_table_{train,eval}.csv): one rocprofv3 kernel trace of tools/time_encoder.py
# with the engine's launch log switched on, joined by tools/layer_table.py.
# usage (GPU box): bash tools/prof_layers.sh <outdir> [tag]
out=${1:-gpurun_out/prof_layers}
tag=${2:-r03}
mkdir -p $out
export ST_LAYER_LOG=$GRAFT_REPO_ROOT/$out/layer_log.csv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o enc -- python3 $GRAFT_REPO_ROOT/tools/time_encoder.py > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
unset ST_LAYER_LOG
python3 tools/layer_table.py $out $out/layer_log.csv 6 > $out/${tag}_layer_table_train.csv
python3 tools/layer_table.py $out $out/layer_log.csv 6 --eval > $out/${tag}_layer_table_eval.csv
python3 tools/forward_timeline.py $out 6 > $out/${tag}_encoder_forward_timeline.csv 2>/dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $out/${tag}_encoder_forward_kernel_stats.csv
tail -3 $out/run.log
grep "^#" $out/${tag}_layer_table_train.csv | head -3
grep "^#" $out/${tag}_layer_table_eval.csv | head -3
