set -u
# the round's closing profile set: rocprofv3 kernel trace + PMC passes of configs 2, 3, 5 and the two bench lines.  TAG=r3e
TAG=${1:-r3e}
bash tools/profile.sh $TAG 2 > gpurun_out/prof_$TAG.log 2>&1 && python3 tools/summarize_profile.py gpurun_out/prof_$TAG pt_trace_items_bvh_lds > gpurun_out/prof_$TAG/summary.json
bash tools/profile.sh c3_$TAG 3 > gpurun_out/prof_c3_$TAG.log 2>&1 && python3 tools/summarize_profile.py gpurun_out/prof_c3_$TAG pt_trace_items_bvh > gpurun_out/prof_c3_$TAG/summary.json
bash tools/profile.sh c5_$TAG 5 > gpurun_out/prof_c5_$TAG.log 2>&1 && python3 tools/summarize_profile.py gpurun_out/prof_c5_$TAG pt_trace_items_bvh_lds > gpurun_out/prof_c5_$TAG/summary.json
timeout -k 10 300 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
timeout -k 10 300 python bench.py --config 4 --steps 10 > gpurun_out/bench_c4_$TAG.json 2> gpurun_out/bench_c4_$TAG.err
for d in $TAG c3_$TAG c5_$TAG; do python3 -c "
import json,sys; s=json.load(open('gpurun_out/prof_$d/summary.json')); print('$d', json.dumps(s['derived']))"; done
