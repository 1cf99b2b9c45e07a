#!/bin/bash
# tools/env_sweep.sh "--opt park_high=48 --opt park_wait=6" "--opt ..." -- one probe of config 2 per option setting (pt_set_option)
out=gpurun_out/env_sweep.log; : > $out
for setting in "$@"; do
  echo "== $setting" >> $out
  timeout -k 10 90 python -u tools/probe.py ${PROBE_ARGS:-} $setting 2>&1 | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['crc'], d['kernel_ms'], 'bal', d['balance'], 'iters', d['wave_iterations'])" >> $out 2>&1 || { cat $out; exit 1; }
done
cat $out
