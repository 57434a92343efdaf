#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters:  python tools/pmc_summary.py <counter_collection.csv> [...]  [--match substr]"""
import csv
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
match = None
files = []
it = iter(sys.argv[1:])
for a in it:
    if a == '--match':
        match = next(it)
    else:
        files.append(a)
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for f in files:
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if match and match not in k:
            continue
        if len(k) > 90:
            k = k[:60] + '...' + k[-27:]
        k = f'{k} [grid {r["Grid_Size"]}, wg {r["Workgroup_Size"]}, vgpr {r["VGPR_Count"]}+{r["Accum_VGPR_Count"]}, lds {r["LDS_Block_Size"]}]'
        a = acc[k][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
        key = (f, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            d = dur[k]
            d[0] += 1
            d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, cs in sorted(acc.items(), key=lambda kv: -dur[kv[0]][1]):
    n, t = dur[k]
    print(f'{k}\n    launches {n}  avg {t / max(n, 1):.1f} us (profiled)')
    wc = cs.get('SQ_WAVE_CYCLES')
    for c, (m, v) in sorted(cs.items()):
        avg = v / m
        extra = ''
        if wc and c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM'):
            extra = f'  = {100.0 * avg / (wc[1] / wc[0]):5.1f} % of SQ_WAVE_CYCLES'
        print(f'    {c:28s} {avg:16.0f}{extra}')
