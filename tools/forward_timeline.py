"""One train-mode ResNet-101 forward as a kernel timeline, from the rocprofv3 kernel trace of tools/time_encoder.py
(tools/prof_encoder.sh): start offset, duration, grid, kernel of every launch between two stem launches.
usage: python tools/forward_timeline.py <dir with *kernel_trace.csv> [index of the forward, default 8] > profiles/rNN_encoder_forward_timeline.csv"""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
stem = [i for i, r in enumerate(rows) if "stem_pool_kernel" in r["Kernel_Name"] or "Li128ELi64ELi4ELi2ELi8ELi1" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a, b = stem[k], stem[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
print("start_us,duration_us,grid_threads,kernel")
tot = 0.0
for r in rows[a:b]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"^void ", "", n)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print('%.1f,%.1f,%s,"%s"' % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r["Grid_Size_X"], n[:100]))
print('# launches %d, kernel time %.1f us, span %.1f us' % (b - a, tot, (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3))
