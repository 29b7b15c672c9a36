#!/bin/bash
# sweep the (ntw, tms) forms and prefetch depths of st_conv1x1_wreg over the pointwise layers (tools/bench_conv_pw.py)
for d in 2 3; do
for cfg in 1,2 1,4 2,2 2,4 4,4; do
  echo "== ST_PW_CFG=$cfg D=$d"
  ST_PW_DEPTH=$d ST_PW_CFG=$cfg timeout -k 10 100 python tools/bench_conv_pw.py 2>&1 | grep "1x1" | sed 's/|.*//'
done
done
