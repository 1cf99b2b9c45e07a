#!/bin/bash
# tools/insitu_cost.sh -- price instruction forms INSIDE the dominant kernel: builds of libpt_hip.so with 200 extra instructions
# of one kind per loop iteration of the traversal kernel's direction phase (-DPT_DUMMY_VALU=200 -DPT_DUMMY_KIND=k), config 2;
# (time - base time) / (200 x wave iterations) x 1024 SIMDs x clock = SIMD cycles per wave64 instruction.
# Build step (CPU): tools/insitu_cost.sh build     Run step (GPU box): tools/insitu_cost.sh run
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
KINDS="0:v_fma_f32 1:v_min_f32 2:v_cmp_lt_f32 3:v_cndmask_b32 4:v_mul_f32_sgpr 5:v_mul_lo_u32 6:v_lshrrev_b32 7:v_rcp_f32 8:v_mul_f32 9:v_max3_f32 10:s_and_b64 11:v_xor_b32"
mkdir -p path_tracer_amd/variants
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc $F -o path_tracer_amd/variants/base.so path_tracer_amd/csrc/pt_api.hip &
  for k in $KINDS; do /opt/rocm/bin/hipcc $F -DPT_DUMMY_VALU=200 -DPT_DUMMY_KIND=${k%%:*} -o path_tracer_amd/variants/dk${k%%:*}.so path_tracer_amd/csrc/pt_api.hip & done
  wait; ls path_tracer_amd/variants
else
  echo "== tools/insitu_cost.sh (MI355X): config 2, kernel ms of 3 launches, then SIMD cycles per inserted instruction at 2.4 GHz"
  base=$(PT_HIP_LIB=path_tracer_amd/variants/base.so timeout -k 10 100 python -u tools/probe.py --reps 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(min(d['kernel_ms']), d['wave_iterations'])")
  echo "base: $base"
  for k in $KINDS; do
    PT_HIP_LIB=path_tracer_amd/variants/dk${k%%:*}.so timeout -k 10 100 python -u tools/probe.py --reps 3 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); b,it='$base'.split(); b=float(b); ms=min(d['kernel_ms'])
print('%-16s %s  +%.3f ms  -> %.2f cycles per instruction' % ('${k#*:}', d['kernel_ms'], ms-b, (ms-b)*1e-3*2.4e9*1024/(200*d['wave_iterations'])))"
  done
fi
