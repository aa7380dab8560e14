#!/bin/bash
# Development (GPU box): tests, bench, rocprofv3 kernel stats, HBM-traffic and SQ counter passes in one call.
# usage: bash tools/gpu_round.sh <tag> [notests]
set -o pipefail
tag=${1:-x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
if [ "$2" != "notests" ]; then
  timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
tail -1 $O/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- $P --steps 10 --warmup 2 > $O/stats.log 2>&1 || { echo stats failed; tail -5 $O/stats.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- $P --steps 3 --warmup 1 > $O/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- $P --steps 3 --warmup 1 > $O/write.log 2>&1 || { echo write failed; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVES -d $O/sq -o sq --output-format csv -- $P --steps 3 --warmup 1 > $O/sq.log 2>&1 || { echo sq failed; tail -3 $O/sq.log; }
cd $R && python3 tools/profile_summary.py $O > $O/summary.log 2>&1; cat $O/summary.log
# the bench line again, now that profiles/CURRENT.json describes THIS build (bench.py quotes roofline.traffic and the
# VALU rate only from counter summaries whose source digest matches): the line to commit as profiles/<tag>_bench.json
if [ -f $O/profiles_out/CURRENT.json ]; then
  cp $O/profiles_out/* $R/profiles/
  timeout -k 10 400 python bench.py > $O/profiles_out/${tag}_bench.json 2> $O/bench_final.err || echo "second bench run failed"
fi
