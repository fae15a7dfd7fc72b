"""FETCH_SIZE / WRITE_SIZE calibration factors from two rocprofv3 --pmc passes over tools/micro/fetch_calib (each kernel moves exactly
2 GiB): factor = true bytes / (counter * 1024).  usage: pmc_calib.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys
TRUE = float(2 << 30)


def load(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        tot[k] += float(r['Counter_Value'])
        n[k].add(r['Dispatch_Id'])
    return {k: tot[k] * 1024.0 / len(n[k]) for k in tot}


f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {'true_bytes_per_launch': TRUE,
       'fetch_factor': {k: round(TRUE / v, 4) for k, v in f.items() if k.startswith('read_') and v > 0},
       'write_factor': {k: round(TRUE / v, 4) for k, v in w.items() if k.startswith('write_') and v > 0},
       'reported_fetch_bytes': f, 'reported_write_bytes': w,
       '_note': 'factor = bytes really moved / bytes the counter reports (counter * 1024); multiply a kernel\'s raw counter bytes by the '
                'factor of its access width.  read_dword = the conv patch staging, read_ldsdma = the conv weight stream, write_dword = '
                'the conv epilogue.'}
json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
print(json.dumps({'fetch_factor': out['fetch_factor'], 'write_factor': out['write_factor']}))
