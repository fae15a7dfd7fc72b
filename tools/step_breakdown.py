"""Per-step kernel breakdown from a rocprofv3 kernel trace CSV of bench.py: takes step number K (default 5: inside the timed
region of `--warmup 2 --steps 6`; steps are delimited by the bursts of Adam kernels) and prints time per kernel name.
usage: step_breakdown.py trace.csv [top] [K]"""
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
adam = [i for i, r in enumerate(rows) if 'FusedAdam' in r['Kernel_Name'] or 'adam_flat_kernel' in r['Kernel_Name']
        or 'adam_pack_table_kernel' in r['Kernel_Name'] or 'adam_segments_kernel' in r['Kernel_Name']]      # (round 5: the fused optimizer tail)
bursts, prev = [], None
for i in adam:
    if prev is None or i - prev > 50:
        bursts.append([i, i])
    else:
        bursts[-1][1] = i
    prev = i
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
b0, b1 = bursts[K - 1][1] + 1, bursts[K][1] + 1
win = rows[b0:b1]
wall = (win[-1]['e'] - win[0]['s']) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    n = re.sub(r'\(.*', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)[:80]
    agg[n][0] += 1
    agg[n][1] += r['e'] - r['s']
tot = sum(v[1] for v in agg.values()) / 1e6
print('step %d of %d: wall' % (K, len(bursts)) + ' wall %.2f ms, kernel-sum %.2f ms, %d launches' % (wall, tot, len(win)))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%7.3f ms %5d  %s' % (v[1] / 1e6, v[0], n))
# optional 4th argument: a regular expression -- every launch of the step whose kernel name matches, one line each
if len(sys.argv) > 4:
    pat = re.compile(sys.argv[4])
    print('-- launches matching %r in this step (name, workgroups x threads, us)' % sys.argv[4])
    for r in win:
        if pat.search(r['Kernel_Name']):
            n = re.sub(r'^void ', '', re.sub(r'\(.*', '', r['Kernel_Name']))[:60]
            try:
                wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1)
                nb = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1) // max(wg, 1)
            except (KeyError, ValueError):
                wg = nb = -1
            print('%-60s %6d x %4d  %8.2f' % (n, nb, wg, (r['e'] - r['s']) / 1e3))
