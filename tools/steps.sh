#!/bin/bash
# tools/steps.sh LIB... -- traversal-phase event counts (builds with -DPT_PHASE_TIMERS=2) on configs 2, 5 and 3:
# phases entered, node steps, leaf steps, rays at phase entry; lanes per node step = bvh_node_visits / node steps
for lib in "$@"; do
  for cfg in "c2|" "c5|--scene mixed --cam 1.5 0.8 1.5 45 20 --bounces 16" "c3|--scene hf708 --cam 0.2 1.6 2.2 95 35"; do
    name=${cfg%%|*}; args=${cfg#*|}
    PT_HIP_LIB=$PWD/$lib timeout -k 10 120 python -u tools/probe.py $args --reps 1 2>&1 | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['phase_cycles']
print('$(basename $lib .so) $name: kernel ms', d['kernel_ms'], '| traversal phases %.3g, rays at entry %.1f per phase | node steps %.4g with %.1f lanes | leaf steps %.4g with %.1f lanes (x2 triangles) | node visits %.4g, triangle tests %.4g, wave iterations %.4g'
      % (p[0], p[3] / max(1, p[0]), p[1], d['bvh_node_visits'] / max(1, p[1]), p[2], d['bvh_tri_tests'] / 2 / max(1, p[2]), d['bvh_node_visits'], d['bvh_tri_tests'], d['wave_iterations']))" || exit 1
  done
done
