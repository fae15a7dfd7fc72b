"""Sum one rocprofv3 --pmc counter per kernel name: usage pmc_sum.py <counter_collection.csv> <COUNTER> [scale]"""
import collections, csv, re, sys
tot, n = collections.defaultdict(float), collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] != sys.argv[2]:
        continue
    k = re.sub(r'<.*', '', re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''))
    tot[k] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
sc = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
for k in sorted(tot, key=lambda k: -tot[k])[:8]:
    print('%-44s launches %4d  %s total %.1f  per launch %.2f' % (k, len(n[k]), sys.argv[2], tot[k] * sc, tot[k] * sc / len(n[k])))
