"""Summarise a per-launch conv table written by `bench.py --layer-table` (per step, sorted by time)."""
import collections
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(path)))
agg = collections.OrderedDict()
for r in rows:
    k = (r['kind'], r['B'], r['Cin'], r['Cout'], r['H'], r['W'], r['ks'], r['split'], r['blocks'])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(r['ms'])
    a[2] += float(r['gflop'])
tot = sum(a[1] for a in agg.values())
print('total conv ms per step %.2f   (%.1f TFLOP/s overall)' % (tot / steps, sum(a[2] for a in agg.values()) / tot))
print('kind   B   Cin  Cout    HW/H    W ks split blocks  n  ms/step  GF/launch    TF  share')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%4s %3s %5s %5s %7s %5s %2s %5s %6s %3d %8.3f %9.1f %6.1f %5.1f%%' % (
        k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], a[0] // steps, a[1] / steps, a[2] / a[0], a[2] / a[1], 100 * a[1] / tot))
