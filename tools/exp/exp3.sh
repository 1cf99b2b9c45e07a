set -e
for o in "--opt park_high=60 --opt park_wait=32 --opt park_exit=6 --opt park_blocked=48" "--opt park_high=64 --opt park_wait=64 --opt park_exit=6 --opt park_blocked=64" "--opt park_high=56 --opt park_wait=16 --opt park_exit=5 --opt park_blocked=32" "--opt park_high=48 --opt park_wait=16 --opt park_exit=4 --opt park_blocked=24"; do
PT_HIP_LIB=path_tracer_amd/variants/t2.so timeout -k 10 120 python -u tools/probe.py --variant 7 --reps 2 $o 2>&1 | tail -1
done
for o in "--opt park_high=48 --opt park_wait=8 --opt park_exit=5" "--opt park_high=56 --opt park_wait=8 --opt park_exit=6" "--opt park_high=32 --opt park_wait=4 --opt park_exit=4"; do
PT_HIP_LIB=path_tracer_amd/variants/t2.so timeout -k 10 120 python -u tools/probe.py --variant 6 --reps 2 $o 2>&1 | tail -1
done
