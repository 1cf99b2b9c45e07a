set -e
for v in 6 7; do timeout -k 10 120 python -u tools/probe.py --variant $v --reps 3 2>&1 | tail -1; done
for h in 32 48 56; do for w in 4 8 16; do timeout -k 10 120 python -u tools/probe.py --variant 7 --reps 2 --opt park_high=$h --opt park_wait=$w 2>&1 | tail -1; done; done
for v in 6 7; do timeout -k 10 120 python -u tools/probe.py --scene mixed --cam 1.5 0.8 1.5 45 20 --bounces 16 --variant $v --reps 2 2>&1 | tail -1; done
for v in 5 8; do timeout -k 10 200 python -u tools/probe.py --scene hf708 --cam 0.2 1.6 2.2 95 35 --variant $v --reps 2 2>&1 | tail -1; done
