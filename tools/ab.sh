#!/bin/bash
# tools/ab.sh -- A/B of library variants x passes-per-item on the GPU box (same ABI, PT_HIP_LIB selects the build)
for lib in "" $(ls path_tracer_amd/variants/*.so 2>/dev/null); do
  for c in ${CHUNKS:-2 4 8}; do
    for w in ${WAVES:-12}; do
    r=$(PT_HIP_LIB=$lib timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --passes-per-item $c --waves-per-cu $w --variant ${VARIANT:-0} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['roofline']['work_per_launch']; print('%.2f ms/step kernel %.2f ms frac %.3f iters %.3g balance %.3f primary/mesh %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], w.get('wave_iterations',0), d['roofline'].get('wave_balance',0), w.get('primary_mesh_segments',0)/max(1,w['mesh_segments'])))")
    echo "lib=${lib:-default} chunk=$c waves=$w : $r"
    done
  done
done
