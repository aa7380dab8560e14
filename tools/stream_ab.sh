#!/bin/bash
# Development (GPU box): the 4096-cell map on the quad kernel built without (default) and with (PYSONIC_AMD_STREAM=1)
# the work-queue switch inside its step loop: kernel ms from bench.py, wavefront instruction counts from one SQ
# counter pass each.   usage: bash tools/stream_ab.sh <tag>
set -o pipefail
tag=${1:-x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-extras > $O/bench_plain.json 2> $O/bench_plain.err || exit 1
PYSONIC_AMD_STREAM=1 python bench.py --no-cpu-baseline --no-extras > $O/bench_stream.json 2> $O/bench_stream.err || exit 1
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVES"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/sq_plain -o sq --output-format csv -- $P > $O/sq_plain.log 2>&1 || { echo sq plain failed; exit 1; }
export PYSONIC_AMD_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/sq_stream -o sq --output-format csv -- $P > $O/sq_stream.log 2>&1 || { echo sq stream failed; exit 1; }
unset PYSONIC_AMD_STREAM
cd $R && python3 - <<PY > $O/summary.txt
import json, csv, glob, collections
for n in ("plain", "stream"):
    d = json.loads(open("$O/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "kernel_ms", d["roofline"]["kernel_ms"], "max_steps", d["roofline"]["max_steps_per_config"])
    acc = collections.defaultdict(float); nk = 0
    for f in glob.glob("$O/sq_%s/**/*counter_collection.csv" % n, recursive=True):
        for r in csv.DictReader(open(f)):
            if "quad_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                nk += r["Counter_Name"] == "SQ_WAVES"
    print(n, {k: "%.4g" % (v / max(nk, 1)) for k, v in sorted(acc.items())}, "launches", nk)
PY
cat $O/summary.txt
