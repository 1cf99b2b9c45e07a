#!/usr/bin/env python3
"""Register, spill and scratch usage of every trace kernel as hipcc reports it (-Rpass-analysis=kernel-resource-usage).
    python tools/kernel_regs.py [substring of the demangled kernel name]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

want = sys.argv[1] if len(sys.argv) > 1 else "pt_trace_items_bvh"
r = subprocess.run([g.HIPCC, *g.HIP_FLAGS, "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/pt_regs.so",
                    os.path.join(ROOT, "path_tracer_amd", "csrc", "pt_api.hip")], capture_output=True, text=True, cwd=ROOT)
rows, cur = {}, None
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    m = re.search(r"remark: (?:\S+: )?\s*([A-Za-z][A-Za-z \[\]/]*?): (\d+)\s", line + " ")
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if want in k:
        print(f"{k:60s} VGPR {v.get('VGPRs')} SGPR {v.get('SGPRs')} sgpr-spill {v.get('SGPRs Spill')} vgpr-spill {v.get('VGPRs Spill')} "
              f"scratch {v.get('ScratchSize [bytes/lane]')} occupancy {v.get('Occupancy [waves/SIMD]')}")
