"""Where the conv kernels' wave-cycles go, from a rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES pass over bench.py (profiles/rNN_sq_waits.json).
usage: pmc_sq_waits.py <counter_collection.csv> <out.json>"""
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
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    la = c.get('SQ_LDS_IDX_ACTIVE', 0.0)
    if not wc:
        continue
    res[k] = {'launches': len(launches[k]),
              'wait_any_frac': c.get('SQ_WAIT_ANY', 0.0) / wc, 'wait_inst_any_frac': c.get('SQ_WAIT_INST_ANY', 0.0) / wc,
              'wait_inst_lds_frac': c.get('SQ_WAIT_INST_LDS', 0.0) / wc, 'active_inst_any_frac': c.get('SQ_ACTIVE_INST_ANY', 0.0) / wc,
              'lds_active_over_wave_cycles': la / wc,
              'lds_bank_conflict_over_lds_active': (c.get('SQ_LDS_BANK_CONFLICT', 0.0) / la) if la else None,
              'mfma_busy_cycles_per_wave_quadcycle': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / wc}
res['_note'] = ('rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE '
                'SQ_VALU_MFMA_BUSY_CYCLES over `bench.py --steps 2 --warmup 1 --no-prof --no-extra` (shipped tuning database), summed over the launches '
                'of each conv kernel; fractions are of SQ_WAVE_CYCLES (quad-cycles summed over waves). wait_inst_any = waves stalled waiting for an '
                'instruction to issue (MFMA pipe busy / dependencies), wait_any = any wait incl. memory and barriers.')
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    if isinstance(v, dict):
        print('%-34s launches %4d  wait_inst %.3f  wait_any %.3f  issuing %.3f  lds active %.3f  of which conflicts %.3f'
              % (k, v['launches'], v['wait_inst_any_frac'], v['wait_any_frac'], v['active_inst_any_frac'], v['lds_active_over_wave_cycles'],
                 v['lds_bank_conflict_over_lds_active'] or 0))
