"""GPU timeline of one training step from a rocprofv3 kernel trace CSV of bench.py (side streams ON): wall time of the step, time
with at least one kernel running (union of the kernel intervals), with >= 2 running, idle time, and where the idle time sits
(the kernels that start after the largest gaps).  Steps are delimited by the Adam launches, as in step_breakdown.py.
usage: step_timeline.py trace.csv [K]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_flat_kernel' in r['Kernel_Name']]
bursts, prev = [], None
for i in adam:
    if prev is None or i - prev > 50:
        bursts.append([i, i])
    else:
        bursts[-1][1] = i
    prev = i
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
b0, b1 = bursts[K - 1][1] + 1, bursts[K][1] + 1
win = rows[b0:b1]
t0, t1 = win[0]['s'], max(r['e'] for r in win)
ev = sorted([(r['s'], 1) for r in win] + [(r['e'], -1) for r in win])
busy1 = busy2 = 0
depth, last = 0, t0
for t, d in ev:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth += d
    last = t
wall = t1 - t0
qkey = 'Queue_Id' if 'Queue_Id' in win[0] else None
print('step %d: wall %.2f ms, >=1 kernel running %.2f ms, >=2 running %.2f ms, idle %.2f ms (%.1f %%), kernel-sum %.2f ms, %d launches'
      % (K, wall / 1e6, busy1 / 1e6, busy2 / 1e6, (wall - busy1) / 1e6, 100.0 * (wall - busy1) / wall, sum(r['e'] - r['s'] for r in win) / 1e6, len(win)))
if qkey:
    per = {}
    for r in win:
        per.setdefault(r[qkey], [0, 0])
        per[r[qkey]][0] += 1
        per[r[qkey]][1] += r['e'] - r['s']
    for q, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print('   queue %s: %d launches, %.2f ms of kernels' % (q, v[0], v[1] / 1e6))
# gaps: time before a kernel start during which nothing was running
gaps = []
end_so_far = win[0]['e']
for r in win[1:]:
    if r['s'] > end_so_far:
        gaps.append((r['s'] - end_so_far, re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:70]))
    end_so_far = max(end_so_far, r['e'])
gaps.sort(reverse=True)
print('gaps: %d, total %.2f ms; > 20 us: %d (%.2f ms); 5-20 us: %d (%.2f ms); < 5 us: %d (%.2f ms)' % (
    len(gaps), sum(g for g, _ in gaps) / 1e6,
    sum(1 for g, _ in gaps if g > 20000), sum(g for g, _ in gaps if g > 20000) / 1e6,
    sum(1 for g, _ in gaps if 5000 < g <= 20000), sum(g for g, _ in gaps if 5000 < g <= 20000) / 1e6,
    sum(1 for g, _ in gaps if g <= 5000), sum(g for g, _ in gaps if g <= 5000) / 1e6))
for g, n in gaps[:15]:
    print('   %7.1f us before %s' % (g / 1e3, n))
