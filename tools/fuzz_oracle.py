#!/usr/bin/env python3
"""Randomised parity check against the CPU oracle on the GPU box: the scenes of tools/fuzz_bvh.py, rendered by the library's
default kernel choice and by the brute-force kernel, compared bit for bit with oracle/pt_oracle.c (the restatement of
rt.comp pinned against the reference under llvmpipe).  tools/fuzz_bvh.py compares GPU kernels with each other, which share
their arithmetic helpers (exact reciprocal / square root / division forms with fast paths and fallbacks); this one puts
those helpers against plain IEEE C on inputs far from the fixtures: coordinates of 30,000, edges of 1e-4, random materials.
   python tools/fuzz_oracle.py [--seeds N] [--first S] [--preset 0|1]      exit status 1 on any difference."""
import argparse
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
spec = importlib.util.spec_from_file_location("fuzz_bvh", os.path.join(ROOT, "tools", "fuzz_bvh.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
import oracle_py as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--size", type=int, nargs=2, default=[96, 56])
    ap.add_argument("--spp", type=int, default=2)
    ap.add_argument("--preset", type=int, default=0)
    ap.add_argument("--extreme-materials", action="store_true", help="material parameters at the ends of their ranges (own random stream)")
    ap.add_argument("--many-scene-tris", action="store_true")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--max-tris", type=int, default=2000, help="skip scenes whose mesh the oracle would take seconds over")
    a = ap.parse_args()
    fz.capi.load()
    W, H = a.size
    bad = done = skipped = nonfinite = 0
    for seed in range(a.first, a.first + a.seeds):
        sc, cam, bounces = fz.random_scene(seed, None, None, a.many_scene_tris, a.scale)
        if a.extreme_materials:
            r2 = np.random.default_rng(seed + 10**9)
            m = sc.materials
            for name, choices in (("smoothness", [0.0, 1.0, 0.5]), ("specularSmoothness", [0.0, 1.0]), ("specularChance", [0.0, 1.0, 0.5]),
                                  ("refractionAmount", [0.0, 1.0, 0.5]), ("ior", [0.25, 0.999, 1.0, 1.001, 1.5, 4.0]), ("emissionStrength", [0.0, 1e-3, 50.0])):
                m[name] = r2.choice(choices, len(m)).astype(np.float32)
        if len(sc.mesh_tris) > a.max_tris:
            skipped += 1
            continue
        want = O.Oracle(sc, cam, W, H, bounces, preset=a.preset).render(1, a.spp)
        nonfinite += int(not np.isfinite(want).all())
        for variant in (0, 1):
            got, name = fz.render(sc, cam, W, H, bounces, a.spp, variant, 0, False, a.preset)
            n = int((got.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
            if n:
                bad += 1
                yx = np.argwhere((got.view(np.uint32) != want.view(np.uint32)).any(-1))[0]
                print(f"seed {seed}: variant {variant} ({name}) differs from the oracle in {n} pixels, first at x={yx[1]} y={yx[0]}: "
                      f"gpu {got[yx[0], yx[1]]} oracle {want[yx[0], yx[1]]}; mesh {len(sc.mesh_tris)} tris, {len(sc.spheres)} spheres, "
                      f"{len(sc.triangles)} scene tris, {bounces} bounces", flush=True)
        done += 1
        if done % 25 == 0:
            print(f"... {done} scenes, {bad} differences so far", flush=True)
    print(f"fuzz_oracle: {done} scenes from seed {a.first} ({skipped} skipped for size), {W}x{H}, {a.spp} spp, preset {a.preset}: {bad} differing renders; "
          f"{nonfinite} oracle images with non-finite values", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
