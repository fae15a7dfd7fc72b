"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-launch HBM traffic of the dominant
kernels (profiles/rNN_traffic.json).  Counters are in KB (TCC_EA0_RDREQ/WRREQ based, MI355X_MICROARCH.md section HBM):
bytes = counter * 1024 * calibration factor.  Round 4 calibrated the factors on this part with kernels that move a known 2 GiB
in the conv kernels' own access widths (tools/micro/fetch_calib.hip -> profiles/r04_pmc_calibration.json): FETCH_SIZE reports
exactly half of the bytes for 4-byte-per-lane buffer loads, 16-byte-per-lane loads and 16-byte LDS-DMA alike (factor 2.0),
WRITE_SIZE is exact (1.0).  The output records the factors and the sha of the kernel sources the passes ran on (bench.py only
quotes a profile whose sha matches the running build)."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
try:
    CAL = json.load(open(os.path.join(ROOT, 'profiles', 'r04_pmc_calibration.json')))
    FETCH_FACTOR = float(CAL['fetch_factor']['read_dword'])        # (== read_ldsdma == read_dwordx4 on gfx950)
    WRITE_FACTOR = float(CAL['write_factor']['write_dword'])
except Exception:
    FETCH_FACTOR, WRITE_FACTOR = 2.0, 1.0

fetch_csv, write_csv, out = sys.argv[1:4]
layer_csv = sys.argv[4] if len(sys.argv) > 4 else None   # bench.py --layer-table: adds the algorithmic bytes


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
    fraw = f[k] * 1024 / fn[k]
    fb = fraw * FETCH_FACTOR
    wb = w.get(k, 0.0) * 1024 / max(wn.get(k, 1), 1) * WRITE_FACTOR
    res[k] = {'launches_in_pass': fn[k], 'fetch_bytes_per_launch': fb, 'fetch_bytes_per_launch_raw_counter': fraw,
              'write_bytes_per_launch': wb, 'hbm_bytes_per_launch': fb + wb}
if layer_csv:
    # algorithmic bytes of a conv launch = its input + output + weight tensors once (fp32)
    acc = {}
    for r in csv.DictReader(open(layer_csv)):
        B, Cin, Cout, H, W, ks = (int(r[k]) for k in ('B', 'Cin', 'Cout', 'H', 'W', 'ks'))
        px = H if r['kind'] == '1' else H * W          # the wgrad rows carry H*W in the H column
        by = 4.0 * (B * Cin * px + B * Cout * px + Cout * Cin * ks * ks)
        # which kernel family ran the launch: the split-bf16 kernels take >= 16 K-channels and k >= 3 (the autotuner may
        # still have preferred an f32 kernel for a few weight-gradient shapes, e.g. W = 20)
        split = Cin >= 16 and (r['kind'] == '0' or Cout >= 16)       # (1x1 layers joined the split kernels in round 3)
        name = {('0', True): 'pnsfm::conv2d_bx3_kernel', ('0', False): 'pnsfm::conv2d_mfma_kernel',
                ('1', True): 'pnsfm::conv2d_wgrad3_kernel', ('1', False): 'pnsfm::conv2d_wgrad_kernel'}[(r['kind'], split)]
        a = acc.setdefault(name, [0.0, 0])
        a[0] += by
        a[1] += 1
        # round 6: the table says which kernel ran the launch (`kernel` column) -- the algorithmic bytes of the ping-pong kernel and of the
        # nine-taps weight gradient are recorded under their own names as well (the family averages above stay for bench.py)
        own = {('0', '7'): 'pnsfm::conv2d_bx3pp_kernel', ('0', '8'): 'pnsfm::conv1x1_bx3_kernel', ('1', '4'): 'pnsfm::conv2d_wgrad4_kernel',
               ('1', '2'): 'pnsfm::conv2d_wgrad2_kernel'}.get((r['kind'], r.get('kernel', '')))
        if Cin == 3 and ks == 5 and r.get('kernel') == '0':      # the stem's own kernels run under the generic kernel ids
            own = 'pnsfm::conv2d_stem5_kernel' if r['kind'] == '0' else 'pnsfm::conv2d_wgrad_stem5_kernel'
        if own is not None:
            a = acc.setdefault(own, [0.0, 0])
            a[0] += by
            a[1] += 1
        elif r.get('kernel') is not None and (r['kind'], split) in (('0', True), ('1', True)):
            a = acc.setdefault(name + ' (own launches)', [0.0, 0])
            a[0] += by
            a[1] += 1
    for name, a in acc.items():
        if name.endswith(' (own launches)'):
            base = name[:-len(' (own launches)')]
            if base in res and a[1]:
                res[base]['algorithmic_bytes_per_own_launch'] = a[0] / a[1]
        elif name in res and a[1]:
            res[name]['algorithmic_bytes_per_launch'] = a[0] / a[1]
res['workload_shape'] = [int(v) for v in sys.argv[5].split(',')] if len(sys.argv) > 5 else [192, 640, 4]      # H, W, batch of the bench.py run the passes were collected on (bench.py only quotes
                                           # these numbers for that workload)
res['fetch_factor'], res['write_factor'] = FETCH_FACTOR, WRITE_FACTOR
sys.path.insert(0, ROOT)
try:
    import bench
    res['csrc_sha'] = bench.csrc_sha()
except Exception as e:
    res['csrc_sha'] = None
    print('csrc_sha unavailable:', e)
res['_note'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` with a '
                'primed tuning database (PNSFM_TUNE_DB: every launch is a training-step launch, no autotune candidates); '
                'bytes = counter*1024*factor averaged over the launches of each kernel (FETCH_SIZE x 2.0, WRITE_SIZE x 1.0: '
                'profiles/r04_pmc_calibration.json). algorithmic = (input + output + weight) bytes of the layer, averaged '
                'over the launches of bench.py --layer-table.')
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
for k, v in sorted(((k, v) for k, v in res.items() if isinstance(v, dict) and 'hbm_bytes_per_launch' in v), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches_in_pass'])[:12]:
    print('%-40s launches %5d  fetch %8.2f MB  write %8.2f MB per launch' % (k, v['launches_in_pass'], v['fetch_bytes_per_launch'] / 1e6, v['write_bytes_per_launch'] / 1e6))
