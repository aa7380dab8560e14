#!/bin/bash
# Development (GPU box): tests, bench, rocprofv3 kernel stats and HBM-traffic PMC passes in one call.
# usage: bash tools/gpu_round.sh <tag>
set -o pipefail
tag=${1:-x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
tail -1 $O/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1 || { echo stats failed; tail -5 $O/stats.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 || { echo write failed; exit 1; }
ls $O/stats $O/fetch $O/write
