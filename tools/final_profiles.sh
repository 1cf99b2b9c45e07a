set -u
tools/steps.sh build/ab/r3_pt2.so > gpurun_out/steps_r3.txt 2>&1
for c in "" "--scene mixed --cam 1.5 0.8 1.5 45 20 --bounces 16" "--scene hf708 --cam 0.2 1.6 2.2 95 35"; do PT_HIP_LIB=$PWD/build/ab/r3_pt1.so timeout -k 10 120 python -u tools/probe.py $c --reps 1 2>&1 | tail -n 1 >> gpurun_out/phase_cycles_r3.txt; done
bash tools/profile.sh r3a 2 > gpurun_out/prof_r3a.log 2>&1 && python3 tools/summarize_profile.py gpurun_out/prof_r3a pt_trace_items_bvh_lds > gpurun_out/prof_r3a/summary.json
bash tools/profile.sh c3_r3a 3 > gpurun_out/prof_c3_r3a.log 2>&1 && python3 tools/summarize_profile.py gpurun_out/prof_c3_r3a pt_trace_items_bvh > gpurun_out/prof_c3_r3a/summary.json
bash tools/profile.sh c5_r3a 5 > gpurun_out/prof_c5_r3a.log 2>&1 && python3 tools/summarize_profile.py gpurun_out/prof_c5_r3a pt_trace_items_bvh_lds > gpurun_out/prof_c5_r3a/summary.json
timeout -k 10 300 python bench.py > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err
timeout -k 10 300 python bench.py --config 4 --steps 10 > gpurun_out/bench_c4_r3a.json 2> gpurun_out/bench_c4_r3a.err
cat gpurun_out/steps_r3.txt; for d in r3a c3_r3a c5_r3a; do python3 -c "
import json,sys; s=json.load(open('gpurun_out/prof_$d/summary.json')); print('$d', json.dumps(s['derived']))"; done
