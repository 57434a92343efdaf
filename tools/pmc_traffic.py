"""HBM-side bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB-like units of the memory-side
request counters; FETCH_SIZE tallies 128-B read requests at 64 B, so it is doubled.  Usage:
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        a = acc[r['Kernel_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return acc


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(set(fetch) | set(write)):
    nf, vf = fetch.get(k, [0, 0.0])
    nw, vw = write.get(k, [0, 0.0])
    # rocprofv3 reports both in KB (x1024 bytes)
    rd = 2.0 * 1024.0 * vf / max(nf, 1)
    wr = 1024.0 * vw / max(nw, 1)
    out[k] = {'launches': max(nf, nw), 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'bytes_per_launch': rd + wr}
from bench import csrc_sha  # noqa: E402  (stamp: the kernel sources these counters were collected on; bench.py refuses a mismatch)
out['_csrc_sha'] = csrc_sha()
json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
del out['_csrc_sha']
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['bytes_per_launch'] * kv[1]['launches'])[:16]:
    print(f"{k[:70]:70s} n={v['launches']:4d} read={v['read_bytes_per_launch'] / 1e6:9.1f} MB write={v['write_bytes_per_launch'] / 1e6:9.1f} MB")
