#!/bin/bash
# Development: per-source-line instruction counts of a kernel in sonic_lib.hip (pattern = $1)
set -e
mkdir -p /tmp/isa && cd /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -S --cuda-device-only -o sonic_lib_g.s /root/repo/pysonic_amd/csrc/sonic_lib.hip 2>/dev/null
awk "/^$1/,/s_endpgm/" sonic_lib_g.s > kern.s
python3 - <<'PY'
import re, collections
files={}
for line in open('/tmp/isa/sonic_lib_g.s'):
    m=re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',line)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2)).split('/')[-1]
cur=None; cnt=collections.Counter(); ops=collections.Counter()
for line in open('/tmp/isa/kern.s'):
    m=re.match(r'\s+\.loc\s+(\d+)\s+(\d+)',line)
    if m: cur=(files.get(int(m.group(1)),m.group(1)),int(m.group(2))); continue
    m=re.match(r'\s+((v_|s_|global_|ds_|scratch_)\w+)',line)
    if m: cnt[cur]+=1; ops[m.group(1)]+=1
print('total',sum(cnt.values()), 'valu', sum(c for o,c in ops.items() if o.startswith('v_')))
print(ops.most_common(22))
for (f,l),c in sorted(cnt.items(), key=lambda kv:-kv[1])[:40]: print(f,l,c)
PY
