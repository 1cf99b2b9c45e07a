#!/bin/bash
# tools/split_sweep.sh -- variant 7 (specialised waves) knobs on config 2; one line per setting
out=gpurun_out/split_sweep.log; : > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 90 python -u tools/probe.py --variant ${V:-7} 2>&1 | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['crc'], d['kernel_ms'], 'bal', d['balance'], 'iters', d['wave_iterations'], 'wd', d['mesh_tri_tests'])" >> $out 2>&1 || exit 1; }
V=6 run X=1
for nw in ${NWS:-10 9}; do for nv in ${NVS:-1 2 3}; do for ph in ${PHS:-0}; do for pr in ${PRS:-0 3}; do
  run PT_SPLIT_WAVES=$nw PT_SPLIT_NV=$nv PT_PARK_HIGH=$ph PT_PARK_EXIT=$pr PT_SPLIT_REFILL=${RF:-16} PT_SPLIT_STEPS=${ST:-4}
done; done; done; done
cat $out
