"""Per-step kernel table from a rocprofv3 --kernel-trace --stats run: python tools/prof_summary.py <kernel_stats.csv> <steps incl. warmup>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r['Name'][:84]
    c = int(r['Calls'])
    print(f"{n:84s} calls/step={c / steps:5.1f} avg={float(r['AverageNs']) / 1e3:8.1f}us  per-step={float(r['TotalDurationNs']) / 1e3 / steps:8.1f}us {float(r['Percentage']):5.1f}%")
print(f"total per step: {tot / 1e3 / steps:.1f} us")
