set -e
for lib in t1 t2; do for v in 6 7; do PT_HIP_LIB=path_tracer_amd/variants/$lib.so timeout -k 10 120 python -u tools/probe.py --variant $v --reps 2 2>&1 | tail -1; done; done
for lib in t1 t2; do for v in 5 8; do PT_HIP_LIB=path_tracer_amd/variants/$lib.so timeout -k 10 200 python -u tools/probe.py --scene hf708 --cam 0.2 1.6 2.2 95 35 --variant $v --reps 1 2>&1 | tail -1; done; done
