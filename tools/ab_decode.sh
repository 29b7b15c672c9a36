#!/bin/bash
# A/B harness for decode-kernel variants on ONE GPU box (box-to-box noise is larger than most kernel-level effects).
# Build each variant as show-tell_amd/lib/var_<name>.so here (objects of the unchanged sources come from
# show-tell_amd/build/, only the edited file is recompiled), then on the box:
#   gpurun -- 'bash tools/ab_decode.sh base mt2 base mt2'
# "base" is the library as built by `make`; every run prints tools/time_decode.py's bf16 greedy and beam lines.
set -e
L="$(dirname "$0")/../show-tell_amd/lib"
cp "$L/libshowtell_hip.so" /tmp/ab_base.so
trap 'cp /tmp/ab_base.so "$L/libshowtell_hip.so"' EXIT
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/ab_base.so "$L/libshowtell_hip.so"; else cp "$L/var_$v.so" "$L/libshowtell_hip.so"; fi
  echo "== $v"
  timeout -k 5 180 python "$(dirname "$0")/time_decode.py" 2>&1 | grep -E "greedy torch.bfloat16|beam" | cut -c1-110
done
