#!/bin/bash
# GPU box: the detailed-model kernels in one call -- GPU tests of the full / hybrid paths, BASELINE
# config 5 (tools/bench_configs.py 5), rocprofv3 kernel stats and SQ counters of the cooperative kernel
# on a short batch of the same 256 configurations. usage: bash tools/gpu_full_round.sh <tag>
set -o pipefail
tag=${1:-x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_full.py tests/test_native_abi.py -m gpu -x -q > $O/full_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/full_tests.log
timeout -k 10 300 python tools/bench_configs.py 5 --tstim-full 1e-3 > $O/config5.json 2> $O/config5.err || { echo config5 failed; tail -5 $O/config5.err; exit 1; }
cat $O/config5.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- python3 $R/tools/full_probe.py --sizes 256 --kernels 2 > $O/stats.log 2>&1 || { echo stats failed; tail -5 $O/stats.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/sq1 -o sq1 --output-format csv -- python3 $R/tools/full_probe.py --sizes 256 --kernels 2 > $O/sq1.log 2>&1 || { echo sq1 failed; tail -5 $O/sq1.log; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/sq2 -o sq2 --output-format csv -- python3 $R/tools/full_probe.py --sizes 256 --kernels 2 > $O/sq2.log 2>&1 || { echo sq2 failed; tail -5 $O/sq2.log; }
find $O -name "*.csv" | head -20
