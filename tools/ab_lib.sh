#!/bin/bash
# A/B harness for library variants on ONE GPU box: bash tools/ab_lib.sh "<command>" base v1 base v1 ...
# Variants are show-tell_amd/lib/var_<name>.so (see tools/ab_decode.sh for how to build one); "base" is the `make` library.
set -e
L="$(dirname "$0")/../show-tell_amd/lib"
CMD="$1"; shift
cp "$L/libshowtell_hip.so" /tmp/ab_base.so
trap 'cp /tmp/ab_base.so "$L/libshowtell_hip.so"' EXIT
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/ab_base.so "$L/libshowtell_hip.so"; else cp "$L/var_$v.so" "$L/libshowtell_hip.so"; fi
  echo "== $v"
  timeout -k 5 300 bash -c "$CMD" 2>&1 | tail -3 | cut -c1-160
done
