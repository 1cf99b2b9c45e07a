"""Samples-per-pixel split across GPUs (BASELINE.json north_star; no counterpart in the reference, which is
single-GPU).  One process per GPU under torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo"
in the CPU tests).  Every rank holds the full scene and traces its own contiguous pass range of the SAME image
as a plain float sum (pt_render_sum); the per-rank RGBA32F buffers meet in ONE reduce onto rank 0 per render,
and rank 0 scales by 1/spp.  Samples are pure functions of (pixel, pass) (rt.comp:326-327), so the union of the
ranks' sample sets is exactly the single-GPU set; only the summation order differs from the reference's
running mean (toleranced in the tests).

A direct reduce (every peer -> root over its own xGMI link) moves W*H*16 bytes per rank once per render:
33 MB at 1080p, 133 MB at 4K ~ 0.9 ms at ~150 GB/s per link -- negligible beside the trace, so it is done
once per render, never per pass.
"""
from __future__ import annotations


def pass_range(rank: int, world: int, total_spp: int, first_pass: int = 1):
    """Contiguous, balanced split of passes first_pass .. first_pass+total_spp-1: returns (first, count) of `rank`.
    The first (total_spp % world) ranks get one extra pass; ranges are disjoint and cover every pass once."""
    if not (0 <= rank < world) or total_spp < 0:
        raise ValueError("bad rank/world/spp")
    base, extra = divmod(total_spp, world)
    count = base + (1 if rank < extra else 0)
    first = first_pass + rank * base + min(rank, extra)
    return first, count


def reduce_and_finish(accum, total_spp: int, dst: int = 0, group=None, renderer=None, check: bool = True, stage_on_host: bool = False):
    """Sum the per-rank sum buffers onto `dst` (one collective) and turn the sum into the mean there.
    `accum`: (H, W, 4) float32 tensor holding this rank's sum (alpha = number of passes it traced).
    `renderer`: if it offers `scale_accum` (capi.Renderer), the finish is ONE library kernel (pt_scale_accum) on the
    renderer's stream -- which must be the stream torch issues the reduce on (bench.py sets both); otherwise torch ops.
    `check`: read the summed sample count back and compare it with total_spp (a host synchronisation: callers in a
    timed loop switch it off after their warm-up).
    `stage_on_host`: for a host-side backend (gloo) with a device tensor -- the one-GPU rehearsal of bench.py: the caller's
    stream is drained, the image crosses to the host with one plain blocking copy, the backend reduces host memory, and the
    root copies the sum back.  gloo's own path for device tensors does the same staging on streams of its own (a
    high-priority pool stream per call, a pinned cache, a worker thread that synchronises that stream); keeping it out
    leaves exactly one stream per process on the device.  Returns True on the rank that now holds the finished image."""
    import torch.distributed as dist

    if dist.is_initialized():
        if stage_on_host and accum.is_cuda:
            import torch
            torch.cuda.current_stream(accum.device).synchronize()
            host = accum.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            if dist.get_rank(group) == dst:
                accum.copy_(host)
        else:
            dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)      # a group of one still goes through the backend
        is_root = dist.get_rank(group) == dst
    else:
        is_root = True
    if is_root:
        if check and int(round(float(accum[0, 0, 3]))) != total_spp:
            raise RuntimeError(f"sample count after reduce is {float(accum[0, 0, 3])}, expected {total_spp}")
        scale = getattr(renderer, "scale_accum", None)
        if scale is not None:
            scale(1.0 / total_spp, True)
        else:
            accum[..., :3] *= 1.0 / total_spp          # python scalar: rounded to float32 once, no host tensor
            accum[..., 3] = 1.0
    return is_root


def render_split(renderer, accum, total_spp: int, first_pass: int = 1, group=None, check: bool = True):
    """One multi-GPU render: zero, trace this rank's passes as a sum into `accum` (the tensor registered with
    renderer.set_accum_buffer), reduce, finish on rank 0.  Returns True on rank 0."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    first, count = pass_range(rank, world, total_spp, first_pass)
    accum.zero_()
    if count:
        renderer.render_sum(first, count)
    return reduce_and_finish(accum, total_spp, 0, group, renderer, check)
