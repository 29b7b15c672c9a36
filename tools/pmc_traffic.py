"""Aggregates two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, csv output) into the per-kernel HBM
traffic table kept under profiles/ (mean bytes per launch, 2*FETCH_SIZE + WRITE_SIZE in KB: the gfx950 correction of
MI355X_MICROARCH.md).   python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/rNN_pmc_hbm_traffic.csv"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"igemm_kernelIDF16bLi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        return "igemm_kernel<bf16,%s,%s,%s,%s,%s,mode%s>" % m.groups()
    m = re.search(r"(conv3x3_img_kernel_occ2|conv3x3_img_kernel|conv1x1_wreg_kernel|conv1x1_kstream_kernel|conv1x1_astat_kernel|conv1x1_cstat_kernel|conv1x1_kfuse_kernel|conv_b2b_kernel|conv_c3c1_kernel|conv3x3s2_kstream_kernel)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    for k in ("igemm_s3b_kernel", "igemm_s3_kernel", "bn_act_reg_kernel", "bn_act_kernel", "vocab_argmax_lds_kernel", "maxpool_kernel",
              "decode_pipe_kernel", "transpose_kernel", "ce_kernel", "sgd_kernel", "nchw_to_s2d_kernel", "bn_reduce_replicas_kernel", "stem_pool_kernel"):
        if k in name:
            return k
    m = re.search(r"rnn_gemm_kernel.*", name)
    return m.group(0)[:60] if m else name[:60]


def load(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
print("kernel,launches,mean_FETCH_SIZE_KB_raw,mean_WRITE_SIZE_KB,mean_hbm_MB_per_launch(2*FETCH+WRITE)")
rows = []
for k in ft:
    if k in wt and fc[k] and wc[k]:
        f, w = ft[k] / fc[k], wt[k] / wc[k]
        rows.append((fc[k] * (2 * f + w), k, fc[k], f, w))
for _, k, n, f, w in sorted(rows, reverse=True)[:24]:
    print('"%s",%d,%.1f,%.1f,%.2f' % (k, n, f, w, (2 * f + w) / 1e3))
