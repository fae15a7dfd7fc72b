"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-launch HBM traffic of the dominant
kernels (profiles/rNN_traffic.json).  Counters are in KB (TCC_EA0_RDREQ/WRREQ based, MI355X_MICROARCH.md section HBM):
bytes = counter * 1024.  The conv kernels stage with 4-byte-per-lane loads, an access width for which FETCH_SIZE is
NOT calibrated on gfx950 (the guide only pins the 2x under-count of 16-B/lane streams), so the read side is quoted raw
and as-is; ratios between kernel versions are unaffected."""
import collections
import csv
import json
import re
import sys

fetch_csv, write_csv, out = sys.argv[1:4]


def load(path, counter):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = re.sub(r'<.*', '', re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''))
        tot[k] += float(r['Counter_Value'])
        n[k].add(r['Dispatch_Id'])
    return tot, {k: len(v) for k, v in n.items()}


f, fn = load(fetch_csv, 'FETCH_SIZE')
w, wn = load(write_csv, 'WRITE_SIZE')
res = {}
for k in f:
    if not k.startswith('pnsfm::'):
        continue
    fb = f[k] * 1024 / fn[k]
    wb = w.get(k, 0.0) * 1024 / max(wn.get(k, 1), 1)
    res[k] = {'launches_in_pass': fn[k], 'fetch_bytes_per_launch': fb, 'write_bytes_per_launch': wb,
              'hbm_bytes_per_launch': fb + wb}
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches_in_pass'])[:12]:
    print('%-40s launches %5d  fetch %8.2f MB  write %8.2f MB per launch' % (k, v['launches_in_pass'], v['fetch_bytes_per_launch'] / 1e6, v['write_bytes_per_launch'] / 1e6))
