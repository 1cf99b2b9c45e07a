#!/usr/bin/env python3
"""Dump the ISA of one trace kernel (hipcc --save-temps in /tmp/pt_isa) and print instruction counts per basic block, or
the text of a line range.   python tools/isa_loop.py [kernel-substring] [first_line last_line]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
d = "/tmp/pt_isa"
os.makedirs(d, exist_ok=True)
subprocess.run([g.HIPCC, *g.HIP_FLAGS, "--save-temps", "-o", d + "/x.so", os.path.join(ROOT, "path_tracer_amd", "csrc", "pt_api.hip")], check=True, cwd=d, capture_output=True)
s = open(d + "/pt_api-hip-amdgcn-amd-amdhsa-gfx950.s").read()
want = sys.argv[1] if len(sys.argv) > 1 else "pt_trace_items_bvh_ldsILb0ELb0ELb0E"
m = re.search(r"^(_ZN2pt\S*" + re.escape(want) + r"\S*):", s, re.M)
i = m.start(); j = s.index(".Lfunc_end", i)
lines = s[i:j].splitlines()
open(d + "/kernel.s", "w").write("\n".join(lines))
if len(sys.argv) > 3:
    a, b = int(sys.argv[2]), int(sys.argv[3])
    print("\n".join(f"{n+a:5d} {l[:110]}" for n, l in enumerate(lines[a:b])))
    sys.exit(0)
cur = None
for n, ln in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", ln)
    if m:
        if cur and cur[2] >= 8: print(cur)
        cur = [m.group(1), n, 0, 0, 0, 0, m.group(2).strip()[:50]]
        continue
    t = ln.strip()
    if not t or t[0] in ";." or cur is None: continue
    op = t.split()[0]
    cur[2] += 1
    cur[3] += op.startswith("v_"); cur[4] += op.startswith("s_"); cur[5] += op.startswith("ds_")
print("total lines", len(lines), "| columns: label, line, instructions, VALU, SALU, LDS")
