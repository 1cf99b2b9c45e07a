#!/bin/bash
# tools/sweep3.sh -- one-at-a-time sweep of the traversal's scheduling options on configs 2, 5 and 3 (tools/probe.py, kernel ms)
S2="park_high=24 park_high=40 park_high=48 park_exit=2 park_exit=4 park_wait=2 park_wait=6 park_wait=8 refill_min=2 refill_min=8"
S3="park_high=32 park_high=40 park_high=56 park_exit=2 park_exit=4 park_wait=4 park_wait=8 refill_min=2 refill_min=8"
run() { # name, probe args, settings
  echo "== $1 defaults"; timeout -k 10 120 python -u tools/probe.py $2 2>&1 | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], 'bal', d['balance'], 'iters', d['wave_iterations'], 'nodes', d['bvh_node_visits'])" || exit 1
  for s in $3; do
    echo "== $1 $s"; timeout -k 10 120 python -u tools/probe.py $2 --opt $s 2>&1 | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], 'bal', d['balance'], 'iters', d['wave_iterations'])" || exit 1
  done
}
run c2 "" "$S2" && run c5 "--scene mixed --cam 1.5 0.8 1.5 45 20 --bounces 16" "$S2" && run c3 "--scene hf708 --cam 0.2 1.6 2.2 95 35" "$S3"
