"""MFMA-pipe utilisation of the conv kernels from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass over
bench.py (profiles/rNN_mfma_busy.json).  usage: pmc_mfma_busy.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys

src, out = sys.argv[1:3]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(src)):
    k = re.sub(r'<.*', '', re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''))
    if not k.startswith('pnsfm::conv2d'):
        continue
    tot[k][r['Counter_Name']] += float(r['Counter_Value'])
    launches[k].add(r['Dispatch_Id'])
res = {}
for k, c in tot.items():
    gui = c.get('GRBM_GUI_ACTIVE', 0.0)
    res[k] = {'launches': len(launches[k]), **{n: v for n, v in c.items()},
              'mfma_busy_fraction': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui / 8 * 1024) if gui else None}
res['_note'] = ('rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 2 --warmup 1` (primed tuning '
                'database), summed over the launches of each kernel. GRBM_GUI_ACTIVE is reported per XCD and summed over the 8 XCDs, so '
                'mfma_busy_fraction = MFMA_BUSY_CYCLES / (GUI_ACTIVE/8 * 1024 SIMDs): the share of SIMD-cycles with the matrix pipe busy '
                '(counts padded rows/columns of partial tiles; SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_bf16, 64 per '
                'v_mfma_f32_32x32x2_f32). Counter collection serialises kernels: these are the kernels in isolation.')
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    if isinstance(v, dict):
        print('%-36s launches %4d  mfma busy %.3f' % (k, v['launches'], v['mfma_busy_fraction'] or 0))
