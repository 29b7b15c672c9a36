"""Per-layer table of ONE ResNet-101 forward as the engine routes it (B = 128, bf16): which kernel took which layer, its rocprofv3
duration, and the floor that bounds it -- max(MFMA floor at 2.5 PFLOP/s, HBM floor at 6 TB/s (the achievable streaming rate,
MI355X_MICROARCH) of the layer's algorithmic bytes).

Inputs (both written by ONE run of tools/time_encoder.py under rocprofv3, see tools/prof_layers.sh):
  * the kernel trace (*kernel_trace.csv) -- start / end of every launch;
  * the engine's launch log (ST_LAYER_LOG=<file>, csrc/resnet.cpp:log_launch) -- one line per conv / normalise launch in launch order:
    kernel family, what it computes, geometry, algorithmic FLOPs and bytes.
The two are joined forward by forward (a forward starts at its stem kernel) and launch by launch (the next trace row whose kernel name
contains the logged family).

usage: python tools/layer_table.py <rocprof dir> <layer log> [forward index, default 6] [--eval] > profiles/rNN_layer_table.csv
Rows: every launch of the forward; then a summary per (kernel family, geometry).  `span_us` (first workgroup entry -> last exit, from
tools/{img,as,ks}_stamps.py) is merged from an optional JSON file given with --stamps."""
import csv
import glob
import json
import os
import re
import sys
from collections import OrderedDict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
want_eval = "--eval" in sys.argv
stamps = {}
for a in sys.argv[1:]:
    if a.startswith("--stamps="):
        stamps = json.load(open(a.split("=", 1)[1]))
prof_dir, log_file = args[0], args[1]
fwd = int(args[2]) if len(args) > 2 else 6

f = glob.glob(os.path.join(prof_dir, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"^void ", "", n)
    return n


for r in rows:
    r["name"] = short(r["Kernel_Name"])
    r["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
is_stem = lambda n: "stem_pool_kernel" in n
# train-mode forwards come first in tools/time_encoder.py (stem_pool_kernel<false> = no eval-mode affine), then eval (<true>)
stem_rows = [i for i, r in enumerate(rows) if is_stem(r["name"]) and (("<true>" in r["name"]) == want_eval)]
log = [l.rstrip("\n").split(",") for l in open(log_file) if l.strip()]
log_starts = [i for i, l in enumerate(log) if l[0] == "stem_pool"]
# the log does not say train / eval: forwards appear in the same order as in the trace, so index them through ALL stem launches
all_stems = [i for i, r in enumerate(rows) if is_stem(r["name"])]
k_all = all_stems.index(stem_rows[fwd])
a, b = all_stems[k_all], (all_stems[k_all + 1] if k_all + 1 < len(all_stems) else len(rows))
la, lb = log_starts[k_all], (log_starts[k_all + 1] if k_all + 1 < len(log_starts) else len(log))

FAM = {"igemm": ("igemm_kernel", "igemm_s3b_kernel", "igemm_s3_kernel"), "conv3x3_img": ("conv3x3_img_kernel",), "conv1x1_wreg": ("conv1x1_wreg_kernel",),
       "conv1x1_astat": ("conv1x1_astat_kernel", "conv1x1_cstat_kernel"), "conv1x1_kstream": ("conv1x1_kstream_kernel",), "conv_b2b": ("conv_b2b_kernel",), "conv_c3c1": ("conv_c3c1_kernel",), "conv3x3_s2": ("conv3x3s2_kstream_kernel",),
       "bn_act": ("bn_act_reg_kernel", "bn_act_kernel"), "stem_pool": ("stem_pool_kernel",), "bn_reduce_replicas": ("bn_reduce_replicas_kernel",)}
PEAK, HBM = 2.5e15, 6.0e12
out = []
ti = a
t0 = int(rows[a]["Start_Timestamp"])
for l in log[la:lb]:
    fam, what, cin, cout, k, stride, hin, win, flops, byts = l[0], l[1], int(l[2]), int(l[3]), int(l[4]), int(l[5]), int(l[6]), int(l[7]), float(l[8]), float(l[9])
    while ti < b and not any(key in rows[ti]["name"] for key in FAM[fam]):
        ti += 1
    if ti >= b:
        raise SystemExit("launch log and kernel trace disagree at %r" % (l,))
    r = rows[ti]; ti += 1
    mf, hb = flops / PEAK * 1e6, byts / HBM * 1e6
    out.append(dict(start_us=(int(r["Start_Timestamp"]) - t0) / 1e3, kernel=r["name"].split("(")[0][:70], what=what, cin=cin, cout=cout, k=k, stride=stride,
                    hin=hin, dur=r["dur"], grid=int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), gflop=flops / 1e9, mb=byts / 1e6, mfma_floor=mf, hbm_floor=hb))

print("# one %s-mode ResNet-101 forward, B=128 bf16: launch, kernel, layer, rocprofv3 duration, floors (MFMA 2.5 PFLOP/s | HBM 6 TB/s)" % ("eval" if want_eval else "train"))
print("start_us,duration_us,workgroups,kernel,what,cin,cout,k,stride,hin,GFLOP,MB,mfma_floor_us,hbm_floor_us,floor_us,x_floor")
tot = totf = 0.0
for o in out:
    fl = max(o["mfma_floor"], o["hbm_floor"])
    tot += o["dur"]; totf += fl
    print('%.1f,%.1f,%d,"%s","%s",%d,%d,%d,%d,%d,%.2f,%.1f,%.1f,%.1f,%.1f,%s' % (o["start_us"], o["dur"], o["grid"], o["kernel"], o["what"], o["cin"], o["cout"], o["k"],
          o["stride"], o["hin"], o["gflop"], o["mb"], o["mfma_floor"], o["hbm_floor"], fl, ("%.2f" % (o["dur"] / fl)) if fl > 0 else ""))
other = sum(r["dur"] for r in rows[a:b]) - tot
span = (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3
print("# logged launches %d: %.1f us against a floor of %.1f us (x %.2f); unlogged kernels (layout, pooling, running-buffer update) %.1f us; forward span %.1f us"
      % (len(out), tot, totf, tot / max(totf, 1e-9), other, span))
print("#")
print("# summary per (kernel family, layer geometry): launches, mean duration, floor, ratio, total; span_us = first workgroup entry -> last exit (in-kernel stamps)")
print("kernel,what,cin,cout,k,stride,hin,launches,mean_us,span_us,floor_us,bound,x_floor,total_us")
grp = OrderedDict()
for o in out:
    key = (o["kernel"].split("<")[0], o["what"], o["cin"], o["cout"], o["k"], o["stride"], o["hin"])
    grp.setdefault(key, []).append(o)
for key, v in sorted(grp.items(), key=lambda kv: -sum(o["dur"] for o in kv[1])):
    mean = sum(o["dur"] for o in v) / len(v)
    mf, hb = v[0]["mfma_floor"], v[0]["hbm_floor"]
    fl = max(mf, hb)
    sp = stamps.get("%s_%d_%d" % (key[0], key[2], key[3]), "")
    print('"%s","%s",%d,%d,%d,%d,%d,%d,%.1f,%s,%.1f,%s,%s,%.0f' % (key[0], key[1], key[2], key[3], key[4], key[5], key[6], len(v), mean, sp, fl, "mfma" if mf >= hb else "hbm",
          ("%.2f" % (mean / fl)) if fl > 0 else "", mean * len(v)))
