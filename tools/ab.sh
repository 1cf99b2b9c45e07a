#!/bin/bash
# A/B on one GPU box: the same bench through several builds of the library (PT_HIP_LIB; same ABI), interleaved twice.
#   AB_ARGS="--opt name=value ..." adds bench.py arguments to every run.
#   tools/ab.sh CONFIG LIB [LIB ...]     (LIB = path of a libpt_hip.so build; results: gpurun_out/ab_<name>.log)
set -e
cfg=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    PT_HIP_LIB=$PWD/$lib timeout -k 10 150 python bench.py --config $cfg --steps 60 --no-extra $AB_ARGS > gpurun_out/ab_$name.log 2>&1
    python - <<PY
import json
l=[x for x in open('gpurun_out/ab_$name.log') if x.startswith('{')][-1]; d=json.loads(l)
print('$name', 'config $cfg', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'parity', d.get('cpu_baseline',{}).get('parity_vs_oracle'), flush=True)
PY
  done
done
