"""tools/hip_order_diag.py ORDER -- which HIP runtime(s) end up in the process when torch and libpt_hip.so are loaded in
either order, and whether both see the GPU (diagnostic for 'no HIP device available' after torch has used the card)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})
def use_torch():
    import torch
    t = torch.zeros(4, device="cuda"); torch.cuda.synchronize(); print("torch ok", t.sum().item(), flush=True)
def use_capi():
    from path_tracer_amd import capi
    try:
        r = capi.Renderer(64, 40, device=0); print("capi ok", flush=True); r.close()
    except Exception as e:
        print("capi FAILED:", e, flush=True)
for step in order.split(","):
    {"torch": use_torch, "capi": use_capi, "load": lambda: __import__("path_tracer_amd.capi", fromlist=["x"]).load()}[step]()
    print(step, "->", maps(), flush=True)
