"""Per-step kernel breakdown from a rocprofv3 kernel trace CSV of bench.py: takes the LAST full step (between the last two
bursts of fused-Adam kernels) and prints time per kernel name."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'FusedAdam' in r['Kernel_Name']]
bursts, prev = [], None
for i in adam:
    if prev is None or i - prev > 50:
        bursts.append([i, i])
    else:
        bursts[-1][1] = i
    prev = i
b0, b1 = bursts[-2][1] + 1, bursts[-1][1] + 1
win = rows[b0:b1]
wall = (win[-1]['e'] - win[0]['s']) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    n = re.sub(r'\(.*', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)[:80]
    agg[n][0] += 1
    agg[n][1] += r['e'] - r['s']
tot = sum(v[1] for v in agg.values()) / 1e6
print('last step: wall %.2f ms, kernel-sum %.2f ms, %d launches' % (wall, tot, len(win)))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%7.3f ms %5d  %s' % (v[1] / 1e6, v[0], n))
