"""Stand-in for path_tracer_amd.capi in bench.py's PT_BENCH_STUB mode (tests/test_bench_cpu.py): same method surface as
capi.Renderer, no kernels, CPU memory.  render_sum adds a known value per pass, so the reduced and scaled image can be checked."""
import ctypes as C
import time

import numpy as np


class HostStream:
    cuda_stream = 0

    def synchronize(self):
        pass


class HostEvent:
    def __init__(self):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class Renderer:
    def __init__(self, width, height, device=0):
        self.W, self.H, self.buf, self.samples, self.launches = width, height, None, 0, 0

    def upload_scene(self, scene):
        pass

    def SetUniformCamera(self, cam):
        pass

    def SetUniform1i(self, name, value):
        pass

    def set_tuning(self, *a):
        pass

    def set_option(self, name, value):
        pass

    def set_stream(self, stream_ptr):
        pass

    def set_accum_buffer(self, ptr, nbytes):
        assert nbytes == self.W * self.H * 16
        self.buf = np.ctypeslib.as_array((C.c_float * (self.W * self.H * 4)).from_address(ptr)).reshape(self.H, self.W, 4)

    def render_sum(self, first, count):
        # "radiance" of pass p = p / 1024 in every channel: the mean over passes 1..S is (S + 1) / 2048
        self.buf[..., :3] += np.float32(sum(range(first, first + count)) / 1024.0)
        import os
        broken = os.environ.get("BENCH_STUB_BREAK_RANK") == os.environ.get("RANK")      # a rank that loses a pass (failure test)
        self.buf[..., 3] += np.float32(count - 1 if broken else count)
        self.samples += self.W * self.H * count
        self.launches += 1

    def render(self, first, count):
        raise AssertionError("the stub only serves the split path")

    def scale_accum(self, scale, set_alpha_one=True):
        self.buf[..., :3] *= np.float32(scale)
        if set_alpha_one:
            self.buf[..., 3] = 1.0

    def last_render_ms(self):
        return 1.0

    def kernel_name(self):
        return "stub"

    def reset_counters(self):
        self.samples = 0

    def counters(self):
        keys = ("samples", "segments", "mesh_segments", "mesh_tri_tests", "hits", "wave_iterations", "primary_mesh_segments", "waves",
                "wave_ticks_sum", "wave_ticks_max", "bvh_node_visits", "bvh_tri_tests", "bvh_sphere_tests", "bvh_box_tests")
        d = {k: 0 for k in keys}
        d.update(samples=self.samples, segments=2 * self.samples, hits=self.samples, waves=1, wave_ticks_sum=1, wave_ticks_max=1)
        d["phase_cycles"] = [0, 0, 0, 0]
        return d

    def close(self):
        pass
