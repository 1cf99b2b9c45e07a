#!/usr/bin/env python3
"""bench.py -- samples/sec of the path-tracing hot path on N MI355X (BASELINE.json metric).

A "step" is one full render of the workload BASELINE.json's metric is quoted on (configs[1]):
default res/meshes scene (bunny1 656 tris + light + ground), 1920x1080, 64 spp per GPU, 8 bounces,
bench camera (1.5, 0.4, 0) yaw 0 pitch 0.  Inputs (scene, camera) are resident in HBM before the timed
region; the result stays in HBM on rank 0.

N = 1: reference semantics exactly (running mean, rt.comp:337).  N > 1: samples-per-pixel split, one
process per GPU (torch.distributed, backend nccl == RCCL over xGMI): rank g renders passes
g*64+1 .. (g+1)*64 of the same image as a plain sum, the per-GPU RGBA32F buffers are summed onto rank 0
with ONE reduce per render, rank 0 scales by 1/(64*N).  Per-GPU work is fixed -> weak scaling.

Prints one JSON line on rank 0 (see the repo contract).

Rehearsal of the N > 1 path on a one-GPU box (not a measurement; both ranks time-slice cuda:0, gloo instead of RCCL):
  PT_BENCH_REHEARSAL=1 GLOO_SOCKET_IFNAME=lo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H, SPP, BOUNCES = 1920, 1080, 64, 8
FP32_VALU_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md "Peak FP32 (vector)"
# algorithmic FLOP per unit of work: Moeller-Trumbore test 52, slab test 27, shaded hit ~150
FLOP_TRI, FLOP_AABB, FLOP_SPHERE, FLOP_SHADE = 52, 27, 35, 150


def brute_force_flops(c, sc):
    """FLOP the reference's own algorithm (every mesh triangle behind the root-AABB gate, rt.comp:213-218) needs for
    the segments the kernel counted (padded register slots are NOT counted)."""
    return (FLOP_TRI * (c["mesh_segments"] * len(sc.mesh_tris) + c["segments"] * len(sc.triangles))
            + FLOP_SPHERE * c["segments"] * len(sc.spheres) + FLOP_AABB * c["segments"] + FLOP_SHADE * c["hits"])


def executed_flops(c, sc):
    """FLOP of the tests the kernel actually executed: for the BVH traversal the node (two slab tests each) and
    triangle tests its counters report (BVH configs); for the brute-force kernels the same as above."""
    if not c["bvh_node_visits"]:
        return brute_force_flops(c, sc)
    return (FLOP_TRI * (c["bvh_tri_tests"] + c["segments"] * len(sc.triangles)) + 2 * FLOP_AABB * c["bvh_node_visits"]
            + FLOP_SPHERE * c["segments"] * len(sc.spheres) + FLOP_AABB * c["segments"] + FLOP_SHADE * c["hits"])


def cpu_baseline(sc, cam, gpu_image=None):
    """The oracle (C restatement of the reference, oracle/pt_oracle.c) timed on this host's cores on a bounded
    sample of the SAME workload: every 2nd row of the 1080p frame (540 rows x 1920 px), 64 spp, 8 bounces.
    The rows it renders are also compared, bit for bit, with the image the GPU produced in the timed steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    o = O.Oracle(sc, cam, W, H, BOUNCES)
    rows = list(range(1, H, 2))
    o.render_region(1, 1, 0, rows[0], W, 1)          # warm up (page in, thread pool)
    t0 = time.perf_counter()
    got = []
    for y in rows:
        got.append(o.render_region(1, SPP, 0, y, W, 1)[0])
    dt = time.perf_counter() - t0
    n = len(rows) * W * SPP
    parity = None
    if gpu_image is not None:
        want = np.stack(got)                                             # (540, W, 4)
        have = gpu_image[rows]
        diff = (want.view(np.uint32) != have.view(np.uint32)).any(axis=-1)
        parity = {"rows_checked": len(rows), "pixels_checked": int(diff.size), "pixels_differing_bitwise": int(diff.sum())}
    return {"parity_vs_oracle": parity, "value": n / dt, "unit": "samples/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"oracle/pt_oracle.c (OpenMP) on rows 1,3,..,{rows[-1]} of the 1920x1080 frame ({len(rows)} rows x {W} px), "
                      f"{SPP} spp, {BOUNCES} bounces = {n} samples in {dt:.1f} s"}


def profiled_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/<dir>/summary.json, tools/profile.sh: FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE).  bench.py
    cannot collect PMC counters itself; null when the newest profile is of another kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*", "summary.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("kernel") != kernel_name:
                break
            dv = d["derived"]
            rd = next(v for k, v in dv.items() if k.startswith("hbm_read_bytes_per_launch"))
            wr = next(v for k, v in dv.items() if k.startswith("hbm_write_bytes_per_launch"))
            return {"traffic": rd + wr, "traffic_unit": "bytes per launch (read %.3g + write %.3g)" % (rd, wr),
                    "traffic_source": os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))}
        except (OSError, ValueError, KeyError, StopIteration):
            break
    return {"traffic": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-brute-force", action="store_true", help="skip the extra (untimed) brute-force-kernel measurement")
    ap.add_argument("--passes-per-item", type=int, default=0, help="tuning: passes per work item (0 = library default)")
    ap.add_argument("--waves-per-cu", type=int, default=0, help="tuning: persistent waves per CU (0 = library default)")
    ap.add_argument("--variant", type=int, default=0, help="tuning: kernel variant (0 = library default)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from path_tracer_amd import capi, distributed as D, scene as S

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: path_tracer_amd has no CPU path")
    # rehearsal on a one-GPU box (not a measurement): PT_BENCH_REHEARSAL=1 puts every rank on cuda:0 and swaps RCCL,
    # which refuses two ranks on one device, for gloo -- the rest of the N > 1 path is the real one
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # PT_BENCH_FORCE_SPLIT=1: take the N > 1 code path (process group on RCCL, sum render, reduce, scale) with whatever
    # world size the launcher gave, 1 included -- how the RCCL leg is exercised on a one-GPU box (not a measurement)
    split = world > 1 or os.environ.get("PT_BENCH_FORCE_SPLIT") == "1"
    if split and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
    if split:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    sc, cam = S.default_scene(), S.bench_camera()
    r = capi.Renderer(W, H, device=local_rank)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", BOUNCES)
    if a.passes_per_item or a.waves_per_cu or a.variant:
        r.set_tuning(a.passes_per_item, 0, a.waves_per_cu, a.variant)
    # one explicit stream for everything in a step (memset, trace, accumulate, RCCL reduce, scale): torch's
    # legacy default stream has handle 0, which pt_set_stream reads as "use the context's own stream"
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")      # device memory owned by torch: the reduce reads it
    r.set_accum_buffer(accum.data_ptr(), accum.numel() * 4)
    r.set_stream(stream.cuda_stream)

    def step(check=False):
        if not split:
            accum.zero_()
            r.render(1, SPP)                                                 # reference semantics, bit-exact
        else:
            # rank g traces passes g*SPP+1 .. (g+1)*SPP as a sum; ONE reduce onto rank 0 (RCCL over xGMI); rank 0 scales
            D.render_split(r, accum, SPP * world, check=check)       # the sample-count check synchronises: warm-up only

    def fence():
        torch.cuda.synchronize()
        if split:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(check=True)
    fence()
    r.reset_counters()
    durs = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        # HIP events recorded on the launch stream around pt_trace_tiles (pt_last_render_ms waits for the kernel
        # only: ~10 us of host time against a step of hundreds of ms)
        durs.append(r.last_render_ms())
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if split:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    c = r.counters()
    k_ms = float(np.mean(durs))
    flops_per_launch = executed_flops(c, sc) / a.steps
    bf_flops_per_launch = brute_force_flops(c, sc) / a.steps
    achieved = flops_per_launch / (k_ms * 1e-3) / 1e12
    kernel_name = {0: capi.load().pt_kernel_name().decode(), 1: "pt_trace_items", 5: "pt_trace_items_bvh", 6: "pt_trace_items_bvh_lds"}.get(a.variant, "pt_trace_items_lds")

    final_image = accum.cpu().numpy() if (rank == 0 and not split and not a.no_cpu_baseline) else None
    brute = None
    if rank == 0 and not split and not a.no_brute_force and kernel_name.startswith("pt_trace_items_bvh"):
        # outside the timed region: the same render on the brute-force kernel (exact by construction), to show what the
        # BVH buys and that both give the same image bit for bit
        img = accum.clone()
        r.set_tuning(0, 0, 0, 4)
        r.reset_counters()
        bd = []
        for _ in range(3):
            step()
            bd.append(r.last_render_ms())
        torch.cuda.synchronize()
        bc = r.counters()
        b_ms = float(np.mean(bd[1:]))
        b_flops = brute_force_flops(bc, sc) / 3
        brute = {"kernel": "pt_trace_items_lds", "kernel_ms": b_ms, "samples_per_s": W * H * SPP / (b_ms * 1e-3),
                 "algorithmic_flop_per_launch": b_flops, "achieved_tflops": b_flops / (b_ms * 1e-3) / 1e12,
                 "frac": b_flops / (b_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                 "image_bit_identical_to_bvh_path": bool(torch.equal(img.view(torch.int32), accum.view(torch.int32)))}

    if rank == 0:
        total_samples = W * H * SPP * world * a.steps
        out = {
            "metric": "samples/sec (1080p, 8 bounces)", "value": total_samples / dt, "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL (all ranks on one GPU over gloo): not a measurement",
            "config": {"workload": "configs[1]: default res/meshes scene (bunny1 656 tris + 2 scene tris), 1920x1080, "
                                   f"{SPP} spp per GPU, {BOUNCES} bounces, camera (1.5,0.4,0) yaw 0 pitch 0",
                       "width": W, "height": H, "spp_per_gpu": SPP, "spp_total": SPP * world, "max_bounces": BOUNCES,
                       "parallelism": "spp-split x%d + 1 RCCL reduce/render" % world if split else "single GPU"},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_VALU_PEAK_TFLOPS, **profiled_traffic(kernel_name),
                         "kernel": kernel_name, "kernel_ms": k_ms,
                         "algorithmic_flop_per_launch": flops_per_launch,
                         # the same segments traced the reference's way (every mesh triangle behind the gate)
                         "brute_force_equivalent_flop_per_launch": bf_flops_per_launch,
                         "brute_force_equivalent_tflops": bf_flops_per_launch / (k_ms * 1e-3) / 1e12,
                         "note": "FP32 vector peak (FMA-counted); the parity contract forbids FMA contraction, so unfused mul/add "
                                 "can reach at most 50% of it. achieved counts the tests EXECUTED (BVH node + triangle "
                                 "tests from in-kernel counters); brute_force_equivalent_* prices the same segments the "
                                 "reference's way; brute_force_kernel is that path measured",
                         "work_per_launch": {k: v / a.steps for k, v in c.items() if k not in ("wave_ticks_max", "phase_cycles")},
                         # mean wave span / longest wave span of the persistent grid (1.0 = no tail)
                         "wave_balance": c["wave_ticks_sum"] / max(1, c["waves"] * c["wave_ticks_max"])},
        }
        if brute:
            out["brute_force_kernel"] = brute
        if not split and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, cam, final_image)
        print(json.dumps(out), flush=True)
    if split:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
