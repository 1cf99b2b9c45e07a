"""CPU, world_size 2 over gloo: the samples-per-pixel split (path_tracer_amd/distributed.py).  Each rank's
"renderer" is the oracle summing 1-spp radiances of its pass range (test infrastructure standing in for
pt_render_sum); the reduce + finish logic under test is the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from path_tracer_amd import distributed as D


def test_pass_ranges_partition_the_passes():
    for world in (1, 2, 3, 4, 8):
        for spp in (0, 1, 7, 64, 1024, 1025):
            seen = []
            for r in range(world):
                f, c = D.pass_range(r, world, spp, first_pass=5)
                seen += list(range(f, f + c))
            assert seen == list(range(5, 5 + spp))
            counts = [D.pass_range(r, world, spp)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        D.pass_range(2, 2, 8)


def _worker(rank, world, port, W, H, spp, bounces, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py as O
    from path_tracer_amd import scene as S

    class OracleSumRenderer:        # same contract as capi.Renderer.render_sum on an external accumulation buffer
        def __init__(self, buf):
            self.o, self.buf = O.Oracle(S.default_scene(), S.bench_camera(), W, H, bounces), buf

        def render_sum(self, first, count):
            for p in range(first, first + count):
                self.buf[..., :3] += torch.from_numpy(self.o.radiance_pass(p))
                self.buf[..., 3] += 1.0

    accum = torch.zeros((H, W, 4), dtype=torch.float32)
    is_root = D.render_split(OracleSumRenderer(accum), accum, spp)
    if is_root:
        q.put(accum.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spp", [8, 9])
def test_two_rank_split_matches_single_process_mean(spp):
    W, H, bounces, world = 64, 40, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + spp) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, spp, bounces, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from path_tracer_amd import scene as S
    want = O.Oracle(S.default_scene(), S.bench_camera(), W, H, bounces).render(1, spp)      # the reference's running mean
    assert np.all(got[..., 3] == 1.0)
    # tolerance: float32 sum in a different order than the running mean; radiance is O(1)
    np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=2e-6, atol=1e-6)
