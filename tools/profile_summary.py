''' Turns the rocprofv3 outputs of tools/gpu_round.sh (gpurun_out/<tag>/{stats,fetch,write,sq}) into the
    small files kept under profiles/ and read by bench.py:
        <tag>_bench_kernel_stats.csv    rocprofv3 --stats summary (copied)
        <tag>_hbm_traffic.json          HBM bytes per launch of the integration kernel (FETCH_SIZE + WRITE_SIZE,
                                        separate passes; corrections as MI355X_MICROARCH.md prescribes)
        <tag>_bench_sq_counters.json    SQ counters of the same kernel per launch (VALU / SALU instructions,
                                        wave cycles, issue and wait cycles)
    usage: python tools/profile_summary.py gpurun_out/<tag>      (writes into gpurun_out/<tag>/profiles_out/) '''
import csv
import json
import os
import shutil
import sys
import collections

src = sys.argv[1].rstrip('/')
tag = os.path.basename(src)
out = os.path.join(src, 'profiles_out')
os.makedirs(out, exist_ok=True)
KERNEL = 'sonic_integrate_quad_kernel'


def counters(sub):
    f = os.path.join(src, sub, f'{sub}_counter_collection.csv')
    per = collections.defaultdict(list)
    if not os.path.isfile(f):
        return per
    for r in csv.DictReader(open(f)):
        if KERNEL in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    return per


st = os.path.join(src, 'stats', 'stats_kernel_stats.csv')
if os.path.isfile(st):
    shutil.copy(st, os.path.join(out, f'{tag}_bench_kernel_stats.csv'))
    for r in csv.DictReader(open(st)):
        if KERNEL in r['Name']:
            print('kernel stats:', r['Name'][:60], 'calls', r['Calls'], 'avg ms', float(r['AverageNs']) * 1e-6)

fe, wr = counters('fetch'), counters('write')
if fe and wr:
    fetch_kib = sum(fe['FETCH_SIZE']) / len(fe['FETCH_SIZE'])
    write_kib = sum(wr['WRITE_SIZE']) / len(wr['WRITE_SIZE'])
    bench = {}
    bj = os.path.join(src, 'bench.json')
    if os.path.isfile(bj):
        bench = json.loads(open(bj).read().strip().split('\n')[-1])
    d = {'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 3 --warmup 1 '
                   f'--no-cpu-baseline --no-extras`, kernel {KERNEL}<false>, mean over {len(fe["FETCH_SIZE"])} dispatches',
         'fetch_size_kib': fetch_kib, 'write_size_kib': write_kib,
         'corrections': 'WRITE_SIZE is exact for 16-B-per-lane stores (MI355X_MICROARCH.md, HBM); FETCH_SIZE under-reports '
                        'wide coalesced reads by 2x on gfx950: the reads here are per-lane gathers of lookup records (L2 hits '
                        'after the first touch) and of schedule entries, counted as is',
         'hbm_bytes_per_launch': (fetch_kib + write_kib) * 1024.0,
         'algorithmic_bytes_per_launch': bench.get('roofline', {}).get('algorithmic_bytes_per_launch')}
    json.dump(d, open(os.path.join(out, f'{tag}_hbm_traffic.json'), 'w'), indent=1)
    print('hbm traffic per launch: %.1f MB' % (d['hbm_bytes_per_launch'] / 1e6))

sq = counters('sq')
if sq:
    m = {k: sum(v) / len(v) for k, v in sq.items()}
    d = {'source': f'rocprofv3 --pmc {" ".join(sorted(m))} on `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras`, '
                   f'kernel {KERNEL}<false>, mean over {len(next(iter(sq.values())))} dispatches',
         **m, 'valu_wave_insts_per_launch': m.get('SQ_INSTS_VALU')}
    if 'SQ_WAVE_CYCLES' in m:
        d['fraction_of_wave_cycles_issuing_valu'] = m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_WAVE_CYCLES']
        d['fraction_of_wave_cycles_issuing_scalar'] = m.get('SQ_ACTIVE_INST_SCA', 0) / m['SQ_WAVE_CYCLES']
        d['fraction_of_wave_cycles_waiting'] = m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']
    json.dump(d, open(os.path.join(out, f'{tag}_bench_sq_counters.json'), 'w'), indent=1)
    print('sq counters per launch:', {k: '%.3g' % v for k, v in m.items()})

# profiles/CURRENT.json: which summaries describe the benchmarked kernel, and the digest of the native sources they
# were measured on (bench.py quotes the counters only when that digest is this build's)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd.build import source_hash          # noqa: E402
if fe and wr and sq:
    json.dump({'source_hash': source_hash(), 'hbm_traffic': f'{tag}_hbm_traffic.json',
               'sq_counters': f'{tag}_bench_sq_counters.json', 'kernel_stats': f'{tag}_bench_kernel_stats.csv',
               'kernel': KERNEL, 'made_by': 'tools/profile_summary.py (rocprofv3 passes of tools/gpu_round.sh)'},
              open(os.path.join(out, 'CURRENT.json'), 'w'), indent=1)
