"""Break one training step of a rocprofv3 kernel trace (bench.py run) into encoder / rest and list the kernels of the rest.
usage: python tools/step_breakdown.py <kernel_trace.csv> [step_index]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nchw_to_s2d' in r['Kernel_Name'] or 'nchw_to_nhwc' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seg = rows[idx[k]:idx[k + 1]]
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
def short(n):
    m = re.search(r'igemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)', n)
    if m: return f"igemm<{m.group(1)},{m.group(2)}x{m.group(3)}>"
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n); n = re.sub(r'void \(anonymous namespace\)::', '', n); n = re.sub(r'\(anonymous namespace\)::', '', n)
    return n[:64]
enc_end = max(i for i, r in enumerate(seg) if 'avgpool' in r['Kernel_Name'])
te = int(seg[enc_end]['End_Timestamp'])
print(f"step span {(t1-t0)/1e3:.1f} us, {len(seg)} kernels; encoder {(te-t0)/1e3:.1f} us ({enc_end+1} kernels); rest {(t1-te)/1e3:.1f} us")
for name, part in (("encoder", seg[:enc_end + 1]), ("rest", seg[enc_end + 1:])):
    d = defaultdict(lambda: [0, 0])
    for r in part:
        n = short(r['Kernel_Name']); d[n][0] += 1; d[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print(f"-- {name}")
    for n, (c, t) in sorted(d.items(), key=lambda x: -x[1][1])[:14]:
        print(f"{t/1e3:9.1f} us {c:4d}  {n}")
