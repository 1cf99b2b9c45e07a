#!/usr/bin/env python3
"""bench.py -- samples/sec of the path-tracing hot path on N MI355X (BASELINE.json metric).

A "step" is one full render of the workload BASELINE.json's metric is quoted on (configs[1]):
default res/meshes scene (bunny1 656 tris + light + ground), 1920x1080, 64 spp per GPU, 8 bounces,
bench camera (1.5, 0.4, 0) yaw 0 pitch 0.  Inputs (scene, camera) are resident in HBM before the timed
region; the result stays in HBM on rank 0.  `--config 3` / `--config 5` time BASELINE's configs[2] (1,002,528-triangle
height field) and configs[4] (mixed materials, 145 spheres, 16 bounces) the same way, `--config 4` one GPU's share of configs[3]
(3840x2160, 128 of the 1024 spp; at --gpus 8 it is configs[3] itself); the default run also carries a short
measurement of both under the `extra` key.

N = 1: reference semantics exactly (running mean, rt.comp:337).  N > 1: samples-per-pixel split, one
process per GPU (torch.distributed, backend nccl == RCCL over xGMI): rank g renders passes
g*64+1 .. (g+1)*64 of the same image as a plain sum, the per-GPU RGBA32F buffers are summed onto rank 0
with ONE reduce per render, rank 0 scales by 1/(64*N).  Per-GPU work is fixed -> weak scaling.

`python bench.py --gpus N` starts its N ranks itself (path_tracer_amd/launch.py: the parent touches no GPU and only
spawns children); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it uses the ranks it is given.

Prints one JSON line on rank 0 (see the repo contract).

Rehearsal of the N > 1 path on a one-GPU box (not a measurement; all ranks time-slice cuda:0, gloo instead of RCCL):
  PT_BENCH_REHEARSAL=1 GLOO_SOCKET_IFNAME=lo python bench.py --gpus 2 --steps 3 --warmup 1
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

FP32_VALU_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md "Peak FP32 (vector)"
# algorithmic FLOP per unit of work: Moeller-Trumbore test 52, slab test 27, shaded hit ~150
FLOP_TRI, FLOP_AABB, FLOP_SPHERE, FLOP_SHADE = 52, 27, 35, 150


def workload(config):
    """(description, scene, camera, W, H, spp, bounces) of a BASELINE.json config (numbered as in SURVEY §8: C2, C3, C5)."""
    from path_tracer_amd import scene as S
    if config == 2:
        return ("configs[1]: default res/meshes scene (bunny1 656 tris + 2 scene tris), 1920x1080, 64 spp per GPU, 8 bounces, "
                "camera (1.5,0.4,0) yaw 0 pitch 0", S.default_scene(), S.bench_camera(), 1920, 1080, 64, 8)
    if config == 3:
        return ("configs[2]: synthetic height field y = 0.1 sin(7x) cos(5z) + 0.2, 708 x 708 cells = 1,002,528 mesh triangles, + 2 scene "
                "tris, 1920x1080, 64 spp per GPU, 8 bounces, camera (0.2,1.6,2.2) yaw 95 pitch 35",
                S.default_scene(S.heightfield_mesh(708, 3)), S.heightfield_camera(), 1920, 1080, 64, 8)
    if config == 5:
        return ("configs[4]: mixed materials (bunny1 diffuse+specular, 144 glass spheres, 1 mirror-like sphere, emissive + ground "
                "triangles), 1920x1080, 64 spp per GPU, 16 bounces, camera (1.5,0.8,1.5) yaw 45 pitch 20",
                S.mixed_scene(), S.mixed_camera(), 1920, 1080, 64, 16)
    if config == 4:
        return ("configs[3], one GPU's share: default res/meshes scene, 3840x2160, 128 spp per GPU (1024 spp over 8 GPUs), 8 bounces, "
                "camera (1.5,0.4,0) yaw 0 pitch 0", S.default_scene(), S.bench_camera(), 3840, 2160, 128, 8)
    raise SystemExit(f"--config {config}: choose 2, 3, 4 or 5 (config 1 is the CPU reference case)")


def brute_force_flops(c, sc):
    """FLOP the reference's own algorithm (every mesh triangle behind the root-AABB gate, rt.comp:213-218) needs for
    the segments the kernel counted (padded register slots are NOT counted)."""
    return (FLOP_TRI * (c["mesh_segments"] * len(sc.mesh_tris) + c["segments"] * len(sc.triangles))
            + FLOP_SPHERE * c["segments"] * len(sc.spheres) + FLOP_AABB * c["segments"] + FLOP_SHADE * c["hits"])


def executed_flops(c, sc):
    """FLOP of the tests the kernel actually executed: for the BVH traversal the node (two slab tests each),
    triangle and sphere tests its counters report (BVH configs); for the brute-force kernels the same as above."""
    if not c["bvh_node_visits"]:
        return brute_force_flops(c, sc)
    sph_in_tree = c.get("bvh_sphere_tests", 0)
    sph = FLOP_SPHERE * (sph_in_tree if sph_in_tree else c["segments"] * len(sc.spheres))
    return (FLOP_TRI * (c["bvh_tri_tests"] + c["segments"] * len(sc.triangles)) + c.get("bvh_box_tests", 2 * c["bvh_node_visits"]) * FLOP_AABB
            + sph + FLOP_AABB * c["segments"] + FLOP_SHADE * c["hits"])


def reference_llvmpipe():
    """The reference itself -- rt.comp under Mesa llvmpipe -- cannot run on the GPU box (its shader source lives only in the
    reference checkout and is never copied), so the figure recorded when the golden fixtures were generated in the build
    container travels with them (tests/golden/meta.json, written by oracle/gen_goldens.py)."""
    try:
        e = json.load(open(os.path.join(ROOT, "tests", "golden", "meta.json")))["entries"]["radiance_default_b4_640x360"]
        t = e["llvmpipe_timing"]
        return {"value": t["steady_samples_per_s"], "unit": "samples/s", "threads": e["host_cpus"], "LP_NUM_THREADS": e["lp_threads"],
                "host": "build container (%d CPUs), not this machine" % e["host_cpus"],
                "workload": "BASELINE configs[0]: %dx%d, %d spp, 4 bounces, default scene; passes 2..%d (steady state) %.2f s, pass 1 with the shader JIT %.2f s"
                            % (t["width"], t["height"], t["passes"], t["passes"], t["steady_s"], t["first_pass_s"]),
                "source": "tests/golden/meta.json (oracle/llvmpipe_runner.c running the reference's res/shaders/rt.comp)"}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(config, sc, cam, W, H, spp, bounces, gpu_image=None):
    """The oracle (C restatement of the reference, oracle/pt_oracle.c) timed on this host's cores on a bounded sample of
    the SAME workload, and the pixels it renders compared bit for bit with the image the GPU produced in the timed steps.
    Config 2: every 2nd row of the frame.  Config 4 (4K, 128 spp): every 16th row.  Config 5 (twice the bounces, 147 more
    primitives): every 6th row.  Config 3: the
    oracle scans all 1,002,528 triangles per gate-passing segment, so two windows of 24 x 4 pixels."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    o = O.Oracle(sc, cam, W, H, bounces)
    if config == 3:
        regions = [(940, 500, 24, 4), (300, 700, 24, 4)]
        what = "two windows of 24 x 4 px at (940,500) and (300,700)"
    else:
        step = {2: 2, 4: 16, 5: 6}[config]
        regions = [(0, y, W, 1) for y in range(1, H, step)]
        what = f"rows 1,{1 + step},..,{regions[-1][1]} of the frame ({len(regions)} rows x {W} px)"
    o.render_region(1, 1, 0, regions[0][1], 8, 1)          # warm up (page in, thread pool)
    t0 = time.perf_counter()
    got = [o.render_region(1, spp, x0, y0, w, h) for (x0, y0, w, h) in regions]
    dt = time.perf_counter() - t0
    n = sum(w * h for (_, _, w, h) in regions) * spp
    parity = None
    if gpu_image is not None:
        bad = sum(int((g.view(np.uint32) != np.ascontiguousarray(gpu_image[y0:y0 + h, x0:x0 + w]).view(np.uint32)).any(axis=-1).sum())
                  for g, (x0, y0, w, h) in zip(got, regions))
        parity = {"pixels_checked": n // spp, "pixels_differing_bitwise": bad}
    out = {"parity_vs_oracle": parity, "value": n / dt, "unit": "samples/s", "cores": O.num_threads(), "kind": "port",
           "sample": f"oracle/pt_oracle.c (OpenMP) on {what}, {spp} spp, {bounces} bounces = {n} samples in {dt:.1f} s"}
    ref = reference_llvmpipe()
    if ref:
        out["reference_llvmpipe"] = ref
    return out


def profiled_traffic(kernel_name, config):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/<dir>/summary.json, tools/profile.sh: FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE).  bench.py
    cannot collect PMC counters itself; null when the newest profile of this config is of another kernel."""
    import glob
    pat = {2: "r[0-9]*", 3: "c3_r*", 4: "c4_r*", 5: "c5_r*"}[config]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pat, "summary.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("kernel", "").split("<")[0] != kernel_name.split("<")[0]:
                break
            dv = d["derived"]
            rd = dv["hbm_read_bytes_per_launch"] if "hbm_read_bytes_per_launch" in dv else next(v for k, v in dv.items() if k.startswith("hbm_read_bytes_per_launch"))
            wr = next(v for k, v in dv.items() if k.startswith("hbm_write_bytes_per_launch"))
            return {"traffic": rd + wr, "traffic_unit": "bytes per launch (read %.3g + write %.3g)" % (rd, wr),
                    "traffic_source": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, KeyError, StopIteration):
            break
    return {"traffic": None}


def roofline_of(c, sc, k_ms, n_launches, kernel_name, config):
    flops = executed_flops(c, sc) / n_launches
    bf = brute_force_flops(c, sc) / n_launches
    ach = flops / (k_ms * 1e-3) / 1e12
    return {"bound": "valu", "achieved": ach, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_VALU_PEAK_TFLOPS,
            **profiled_traffic(kernel_name, config), "kernel": kernel_name, "kernel_ms": k_ms,
            "algorithmic_flop_per_launch": flops,
            # the same segments traced the reference's way (every mesh triangle behind the gate)
            "brute_force_equivalent_flop_per_launch": bf, "brute_force_equivalent_tflops": bf / (k_ms * 1e-3) / 1e12,
            "work_per_launch": {k: v / n_launches for k, v in c.items() if k not in ("wave_ticks_max", "phase_cycles")},
            # mean wave span / longest wave span of the persistent grid (1.0 = no tail)
            "wave_balance": c["wave_ticks_sum"] / max(1, c["waves"] * c["wave_ticks_max"])}


def extra_config(config, capi, torch, stream, check_oracle=True):
    """A short, untimed-by-the-contract measurement of another BASELINE config on this GPU (3 renders, the first a warm-up),
    with a small oracle window compared bit for bit: the default line carries it under `extra` so that the driver's
    record holds numbers for configs 3 and 5 as well."""
    desc, sc, cam, W, H, spp, bounces = workload(config)
    r = capi.Renderer(W, H, device=torch.cuda.current_device())
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", bounces)
    r.set_stream(stream.cuda_stream)
    ms = []
    for i in range(3):
        r.reset()
        if i == 1:
            r.reset_counters()
        r.render(1, spp)
        ms.append(r.last_render_ms())
    c = r.counters()
    k_ms = float(np.mean(ms[1:]))
    out = {"workload": desc, "kernel_ms": k_ms, "samples_per_s_kernel": W * H * spp / (k_ms * 1e-3),
           "roofline": roofline_of(c, sc, k_ms, 2, r.kernel_name(), config)}
    if check_oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as O
        img = r.read_accum()
        # config 3: one window of 32 x 8 px (the oracle scans all 1,002,528 triangles per gate-passing segment); config 5: the
        # glass grid and the bunny, 128 x 4 px each
        windows = [(940, 500, 32, 8)] if config == 3 else [(896, 528, 128, 4), (856, 544, 128, 4)]
        o = O.Oracle(sc, cam, W, H, bounces)
        bad = 0
        for (x0, y0, w, h) in windows:
            want = o.render_region(1, spp, x0, y0, w, h)
            bad += int((want.view(np.uint32) != np.ascontiguousarray(img[y0:y0 + h, x0:x0 + w]).view(np.uint32)).any(-1).sum())
        out["parity_vs_oracle"] = {"windows": [list(w) for w in windows], "pixels_checked": sum(w * h for (_, _, w, h) in windows),
                                   "pixels_differing_bitwise": bad}
    r.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed renders (default 150: a timed region of about one second)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, help="BASELINE config to time: 2 (default; the metric's), 3, 4 (one GPU's share: 4K, 128 spp) or 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-brute-force", action="store_true", help="skip the extra (untimed) brute-force-kernel measurement")
    ap.add_argument("--no-extra", action="store_true", help="skip the short config 3 / config 5 measurements of the default run")
    ap.add_argument("--passes-per-item", type=int, default=0, help="tuning: passes per work item (0 = library default)")
    ap.add_argument("--waves-per-cu", type=int, default=0, help="tuning: persistent waves per CU (0 = library default)")
    ap.add_argument("--variant", type=int, default=0, help="tuning: kernel variant (0 = library default)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="pt_set_option (repeatable)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the ranks ourselves.  Nothing above has touched a GPU (no torch import, no HIP
        # call); the parent only waits for its children and passes rank 0's line through.
        from path_tracer_amd import launch
        raise SystemExit(launch.launch(a.gpus, [sys.executable, os.path.abspath(__file__), *sys.argv[1:]],
                                       timeout_s=float(os.environ.get("PT_LAUNCH_TIMEOUT", "3000"))))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PT_BENCH_TRACEBACK_ON_TERM") == "1":     # a rank stopped by the launcher's time limit says where it was
        import faulthandler
        import signal
        faulthandler.register(signal.SIGTERM, all_threads=True, chain=True)

    import torch
    import torch.distributed as dist
    from path_tracer_amd import distributed as D

    # PT_BENCH_STUB=<module> (tests/test_bench_cpu.py only): the module's Renderer stands in for capi.Renderer and every
    # tensor lives on the CPU, so that the N > 1 bookkeeping of this file -- process group, pass split, reduce_and_finish,
    # reduce timing, max-over-ranks time, rank 0's line -- runs at world 2 over gloo on a machine without a GPU.  It renders
    # nothing and measures nothing; the line says so.
    stub = os.environ.get("PT_BENCH_STUB")
    if stub:
        import importlib
        capi = importlib.import_module(stub)
    else:
        from path_tracer_amd import capi
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: path_tracer_amd has no CPU path")
    dev = "cpu" if stub else "cuda"
    # rehearsal on a one-GPU box (not a measurement): PT_BENCH_REHEARSAL=1 puts every rank on cuda:0 and swaps RCCL,
    # which refuses two ranks on one device, for gloo -- the rest of the N > 1 path is the real one
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    # how the rehearsal orders the ranks on the shared device (diagnostic switches; both on by default)
    reh_turns = os.environ.get("PT_REHEARSAL_TURNS", "1") == "1"
    reh_host = os.environ.get("PT_REHEARSAL_HOST_REDUCE", "1") == "1"
    if rehearsal:
        local_rank = 0
    if not stub:
        torch.cuda.set_device(local_rank)
    # PT_BENCH_FORCE_SPLIT=1: take the N > 1 code path (process group on RCCL, sum render, reduce, scale) with whatever
    # world size the launcher gave, 1 included -- how the RCCL leg is exercised on a one-GPU box (not a measurement)
    split = world > 1 or os.environ.get("PT_BENCH_FORCE_SPLIT") == "1"
    if split and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
    if split:
        if rehearsal or stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    desc, sc, cam, W, H, SPP, BOUNCES = workload(a.config)
    r = capi.Renderer(W, H, device=local_rank)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", BOUNCES)
    if a.passes_per_item or a.waves_per_cu or a.variant:
        r.set_tuning(a.passes_per_item, 0, a.waves_per_cu, a.variant)
    for kv in a.opt:
        k, v = kv.split("=")
        r.set_option(k, int(v))
    # one explicit stream for everything in a step (memset, trace, accumulate, RCCL reduce, scale): torch's
    # legacy default stream has handle 0, which pt_set_stream reads as "use the context's own stream"
    if stub:
        stream = capi.HostStream()
        new_event = capi.HostEvent
        device_sync = lambda: None
    else:
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        new_event = lambda: torch.cuda.Event(enable_timing=True)
        device_sync = torch.cuda.synchronize
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)         # device memory owned by torch: the reduce reads it
    r.set_accum_buffer(accum.data_ptr(), accum.numel() * 4)
    r.set_stream(stream.cuda_stream)
    red_ev = []                                                              # (before, after) events around the reduce + finish

    def step(check=False, time_reduce=False):
        if not split:
            accum.zero_()
            r.render(1, SPP)                                                 # reference semantics, bit-exact
            return
        # rank g traces passes g*SPP+1 .. (g+1)*SPP as a sum; ONE reduce onto rank 0 (RCCL over xGMI); rank 0 scales
        first, count = D.pass_range(rank, world, SPP * world, 1)
        if rehearsal and reh_turns:
            # one GPU, several processes: the ranks take the device in turn, so that no two processes ever have a kernel
            # in flight at once (what stalled was the device time-slicing two processes' persistent grids)
            for turn in range(world):
                if turn == rank:
                    accum.zero_()
                    if count:
                        r.render_sum(first, count)
                    stream.synchronize()
                dist.barrier()
        else:
            accum.zero_()
            if count:
                r.render_sum(first, count)
        if rehearsal:
            D.reduce_and_finish(accum, SPP * world, 0, None, r, check, stage_on_host=reh_host)
            if reh_turns:
                stream.synchronize()
                dist.barrier()
            return
        if time_reduce and not rehearsal:
            e0, e1 = new_event(), new_event()
            e0.record(stream)
        D.reduce_and_finish(accum, SPP * world, 0, None, r, check)           # the sample-count check synchronises: warm-up only
        if time_reduce and not rehearsal:
            e1.record(stream)
            red_ev.append((e0, e1))

    def fence():
        device_sync()
        if split:
            dist.barrier()
        device_sync()

    for _ in range(a.warmup):
        step(check=True)
    fence()
    r.reset_counters()
    durs = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(time_reduce=True)
        # HIP events recorded on the launch stream around the trace kernel (pt_last_render_ms waits for that kernel
        # only: ~10 us of host time against a step of milliseconds)
        durs.append(r.last_render_ms())
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if (rehearsal or stub) else "cuda")     # gloo reduces host memory
    if split:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    c = r.counters()
    k_ms = float(np.mean(durs))
    kernel_name = r.kernel_name()                 # the trace kernel the timed renders launched

    final_image = accum.cpu().numpy() if (rank == 0 and not split and not a.no_cpu_baseline) else None
    brute = None
    if rank == 0 and not split and not a.no_brute_force and a.config == 2 and kernel_name.startswith("pt_trace_items_bvh"):
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
    r.close()

    if rank == 0:
        total_samples = W * H * SPP * world * a.steps
        bounces_txt = "%d bounces" % BOUNCES
        out = {
            "metric": "samples/sec (%s, %s)" % ("1080p" if H == 1080 else "%dx%d" % (W, H), bounces_txt), "value": total_samples / dt, "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic; STUB renderer on the CPU (tests): not a measurement" if stub else
                     "synthetic" if not rehearsal else "synthetic; REHEARSAL (all ranks on one GPU over gloo): not a measurement"),
            "config": {"workload": desc, "baseline_config": a.config,
                       "width": W, "height": H, "spp_per_gpu": SPP, "spp_total": SPP * world, "max_bounces": BOUNCES,
                       "parallelism": "spp-split x%d + 1 RCCL reduce/render" % world if split else "single GPU"},
            "roofline": {**roofline_of(c, sc, k_ms, a.steps, kernel_name, a.config),
                         "note": "FP32 vector peak (FMA-counted); the parity contract forbids FMA contraction, so unfused mul/add "
                                 "can reach at most 50% of it. achieved counts the tests EXECUTED (BVH box + triangle + sphere "
                                 "tests from in-kernel counters); brute_force_equivalent_* prices the same segments the "
                                 "reference's way; brute_force_kernel is that path measured"},
        }
        if red_ev:
            red_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in red_ev]))
            out["config"]["reduce_ms"] = red_ms
            out["config"]["reduce_fraction_of_step"] = red_ms / (dt / a.steps * 1e3)
        if brute:
            out["brute_force_kernel"] = brute
        if not split and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.config, sc, cam, W, H, SPP, BOUNCES, final_image)
        if not split and not a.no_extra and a.config == 2:
            out["extra"] = {"config3": extra_config(3, capi, torch, stream), "config5": extra_config(5, capi, torch, stream)}
        print(json.dumps(out), flush=True)
    if split:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
