#!/bin/bash
# tools/env_sweep.sh "VAR=a VAR2=b" "VAR=c ..." -- one probe of config 2 per environment setting (library knobs read from the environment)
out=gpurun_out/env_sweep.log; : > $out
for setting in "$@"; do
  echo "== $setting" >> $out
  env $setting timeout -k 10 90 python -u tools/probe.py ${PROBE_ARGS:-} 2>&1 | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['crc'], d['kernel_ms'], 'bal', d['balance'], 'iters', d['wave_iterations'])" >> $out 2>&1 || { cat $out; exit 1; }
done
cat $out
