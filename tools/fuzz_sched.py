#!/usr/bin/env python3
"""Randomised check of everything between pt_render's arguments and the trace kernel's work items, on the GPU box: for random
image sizes (ragged edges included), pass counts, first passes, passes per work item, workspace limits (one to many batches),
waves per CU, scheduling options and kernel variants, the image must equal -- bit for bit -- the one the reference's frame
loop produces: one dispatch per pass (pt_render_pass) with the library's defaults.  pt_render_static is put through the same
inputs where its static array can hold them.   python tools/fuzz_sched.py [--cases N] [--first S]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd import capi, scene as S  # noqa: E402


def scenes():
    mixed_cam = S.Camera(position=np.array([1.5, 0.8, 1.5], np.float32), yaw=45.0, pitch=20.0).UpdateView()
    out = [("default", S.default_scene(), S.bench_camera())]
    if hasattr(S, "mixed_scene"):
        out.append(("mixed", S.mixed_scene(), mixed_cam))
    hf = S.default_scene(S.heightfield_mesh(40, 1))
    out.append(("hf40", hf, S.Camera(position=np.array([0.2, 1.6, 2.2], np.float32), yaw=95.0, pitch=35.0).UpdateView()))
    return out


def run_case(case, sc_list):
    """one seeded case; returns the number of routes whose image differs from the frame loop's (and prints them)"""
    bad = 0
    if True:
        rng = np.random.default_rng(case)
        name, sc, cam = sc_list[int(rng.integers(0, len(sc_list)))]
        W, H = int(rng.integers(8, 420)), int(rng.integers(8, 260))
        spp, first = int(rng.choice([1, 2, 3, 5, 8, 13, 24])), int(rng.choice([1, 1, 2, 7, 100]))
        bounces = int(rng.choice([1, 3, 8]))
        ref = capi.Renderer(W, H)
        ref.upload_scene(sc); ref.SetUniformCamera(cam); ref.SetUniform1i("maxBounces", bounces)
        ref.set_tuning(0, 0, 0, 1)                       # brute force, one dispatch per pass: the reference's frame loop
        # a running mean that starts at pass `first` continues an image: start both from the same (cleared) image
        for p in range(first, first + spp):
            ref.render_pass(p)
        want = ref.read_accum()
        ref.close()
        variant = int(rng.choice([0, 0, 5, 6, 1, 4]))
        ppi = int(rng.choice([0, 1, 2, 3, 7]))
        ws = int(rng.choice([0, 0, 1, 2, 5])) * W * H * 12           # bytes: 0 = default, else room for 1 / 2 / 5 passes
        wpc = int(rng.choice([0, 0, 4, 8, 16]))
        opts = {}
        if rng.random() < 0.5:
            opts["park_high"] = int(rng.choice([1, 8, 32, 64]))
        if rng.random() < 0.3:
            opts["park_wait"] = int(rng.choice([1, 2, 16]))
        if rng.random() < 0.3:
            opts["park_exit"] = int(rng.integers(1, 9))
        if rng.random() < 0.3:
            opts["refill_min"] = int(rng.choice([1, 16, 64]))
        if rng.random() < 0.3:
            opts["no_fused"] = 1
        if rng.random() < 0.3:
            opts["lds_block"] = int(rng.choice([512, 1024]))
        if rng.random() < 0.2:
            opts["wgs_per_cu"] = int(rng.integers(1, 5))
        for route in ("render", "static", "split"):
            r = capi.Renderer(W, H)
            r.upload_scene(sc); r.SetUniformCamera(cam); r.SetUniform1i("maxBounces", bounces)
            r.set_tuning(ppi, ws, wpc, variant)
            for k, v in opts.items():
                r.set_option(k, v)
            try:
                if route == "render":
                    r.render(first, spp)
                elif route == "static":
                    r.render_static(first, spp)
                else:                                    # the same passes in two calls
                    cut = int(rng.integers(0, spp + 1))
                    if cut:
                        r.render(first, cut)
                    if spp - cut:
                        r.render(first + cut, spp - cut)
            except capi.PtError as e:
                if route == "static" and e.code == capi.PT_E_LIMIT:
                    r.close()
                    continue
                raise
            got = r.read_accum()
            n = int((got.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
            if n:
                bad += 1
                print(f"case {case}: {route} differs in {n} pixels: scene {name} {W}x{H} passes {first}+{spp} bounces {bounces} variant {variant} "
                      f"ppi {ppi} ws {ws} wpc {wpc} opts {opts} kernel {r.kernel_name()}", flush=True)
            r.close()
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    capi.load()
    sc_list = scenes()
    bad = 0
    for case in range(a.first, a.first + a.cases):
        bad += run_case(case, sc_list)
        if (case - a.first) % 25 == 24:
            print(f"... {case - a.first + 1} cases, {bad} differences so far", flush=True)
    print(f"fuzz_sched: {a.cases} cases from {a.first}: {bad} differing renders", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
