#!/bin/bash
# Development (GPU box): rocprofv3 kernel stats and an SQ counter pass of tools/bench_configs.py <configs...>.
# usage: bash tools/profile_configs.sh <tag> <configs...>      -> gpurun_out/<tag>/{stats,sq}/..., summary.json
set -o pipefail
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- python3 $R/tools/bench_configs.py "$@" > $O/stats.log 2>&1 || { echo stats failed; tail -5 $O/stats.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES -d $O/sq -o sq --output-format csv -- python3 $R/tools/bench_configs.py "$@" > $O/sq.log 2>&1 || { echo sq failed; tail -5 $O/sq.log; exit 1; }
cd $R && python3 - "$O" "$*" <<'PY'
import csv, json, sys, collections, os
O, cfgs = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(os.path.join(O, 'sq', 'sq_counter_collection.csv'))):
    per[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
out = {'command': 'rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES '
                  f'-- python3 tools/bench_configs.py {cfgs}',
       'note': 'raw counter values summed over the launches of each kernel',
       'kernels': {k: dict(v) for k, v in per.items() if 'kernel' in k and 'rocclr' not in k}}
for k, v in out['kernels'].items():
    if v.get('SQ_INSTS_VALU') and v.get('SQ_WAVE_CYCLES'):
        v['valu_cycles_per_inst'] = v['SQ_ACTIVE_INST_VALU'] / v['SQ_INSTS_VALU']
        v['valu_share_of_wave_cycles'] = v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES']
json.dump(out, open(os.path.join(O, 'sq_counters.json'), 'w'), indent=1)
print(json.dumps(out['kernels'], indent=1)[:1500])
PY
