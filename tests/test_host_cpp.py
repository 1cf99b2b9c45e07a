"""C++ host classes (path_tracer_amd/host): Camera on CPU, the headless driver pt_main on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT, crc

PT_MAIN = os.path.join(ROOT, "path_tracer_amd", "host", "pt_main")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(PT_MAIN):     # normally built by __graft_entry__.build(); g++ is enough (links libpt_hip.so)
        subprocess.run(["make", "-C", os.path.dirname(PT_MAIN), "-j8"], check=True, capture_output=True)


def test_cpp_camera_update_view_matches_reference_glm():
    """Camera::UpdateView (host/Camera.cpp, own math) vs matrices from the reference's vendored GLM."""
    for e in json.load(open(os.path.join(GOLD, "camera.json")))["cameras"]:
        args = [f"{v:.9g}" for v in e["position"]] + [f"{e['yaw']:.9g}", f"{e['pitch']:.9g}"]
        out = subprocess.run([PT_MAIN, "--print-camera", "--cam", *args], check=True, capture_output=True, text=True).stdout.split()
        assert out == e["inverse_view_bits"], (e["yaw"], e["pitch"])


def test_pt_main_fails_loudly_without_gpu_or_mesh(tmp_path):
    r = subprocess.run([PT_MAIN, "--mesh", str(tmp_path / "missing.obj"), "--width", "64", "--height", "40", "--spp", "1"],
                       capture_output=True, text=True)
    assert r.returncode != 0


def _write_obj(path):
    z = np.load(os.path.join(ROOT, "path_tracer_amd", "data", "bunny1.npz"))
    with open(path, "w") as fh:
        fh.write("o bunny\n")
        for v in z["vertices"]:
            fh.write("v %.9g %.9g %.9g\n" % tuple(float(c) for c in v))
        for f in z["faces"]:
            fh.write("f %d %d %d\n" % tuple(int(i) + 1 for i in f))


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [False, True])
def test_pt_main_renders_config1_like_the_reference(tmp_path, meta, batch):
    """main()'s scene through the C++ classes: OBJ load -> Mesh/Scene upload -> per-frame loop; image CRC must be
    the reference's (640x360, 16 spp, 4 bounces)."""
    obj, out = tmp_path / "bunny1.obj", tmp_path / "accum.f32"
    _write_obj(obj)
    cmd = [PT_MAIN, "--mesh", str(obj), "--width", "640", "--height", "360", "--spp", "16", "--bounces", "4", "--out", str(out),
           "--ppm", str(tmp_path / "img.ppm")] + (["--batch"] if batch else [])
    subprocess.run(cmd, check=True, capture_output=True, timeout=120)
    img = np.fromfile(out, np.float32).reshape(360, 640, 4)
    assert crc(img[:, :, :3]) == meta["radiance_default_b4_640x360"]["crc_mean16"]
    assert os.path.getsize(tmp_path / "img.ppm") == 640 * 360 * 3 + len(b"P6\n640 360\n255\n")


@pytest.mark.gpu
def test_pt_main_mixed_scene(tmp_path, meta):
    obj, out = tmp_path / "bunny1.obj", tmp_path / "accum.f32"
    _write_obj(obj)
    subprocess.run([PT_MAIN, "--mesh", str(obj), "--scene", "mixed", "--width", "640", "--height", "360", "--spp", "4", "--bounces", "16",
                    "--batch", "--out", str(out)], check=True, capture_output=True, timeout=120)
    img = np.fromfile(out, np.float32).reshape(360, 640, 4)
    assert crc(img[:, :, :3]) == meta["radiance_mixed_b16_640x360"]["crc_mean4"]


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [False, True])
def test_pt_main_legacy_shader_name_selects_the_rt_frag_literals(tmp_path, meta, batch):
    """ComputeProgram("res/shaders/rt.frag"): the image must be what rt.frag itself draws (12 bounces, no mesh, sqrt output ...)"""
    obj, out = tmp_path / "bunny1.obj", tmp_path / "accum.f32"
    _write_obj(obj)
    subprocess.run([PT_MAIN, "--mesh", str(obj), "--scene", "mixed", "--shader", "res/shaders/rt.frag", "--width", "640", "--height", "360",
                    "--spp", "4", "--out", str(out)] + (["--batch"] if batch else []), check=True, capture_output=True, timeout=120)
    img = np.fromfile(out, np.float32).reshape(360, 640, 4)
    assert crc(img[:, :, :3]) == meta["radiance_frag_mixed_b12_640x360"]["crc_mean4"]


@pytest.mark.gpu
def test_pt_main_scripted_camera_loop(tmp_path):
    """the interactive loop's stand-in (src/Main.cpp:108-150): the camera moves every frame, each frame restarts the
    accumulation (pt_reset) and renders 2 passes with the short-launch kernels; the last frame must be the image a fresh
    process renders from that final camera pose"""
    import re
    obj, a, b = tmp_path / "bunny1.obj", tmp_path / "orbit.f32", tmp_path / "still.f32"
    _write_obj(obj)
    common = [PT_MAIN, "--mesh", str(obj), "--width", "320", "--height", "184", "--spp", "2", "--bounces", "4"]
    r = subprocess.run(common + ["--orbit", "5", "7.5", "--out", str(a)], check=True, capture_output=True, text=True, timeout=120)
    m = re.search(r"Frames: 5 x 2 spp .* final camera (\S+) (\S+) (\S+) yaw (\S+)", r.stdout)
    assert m, r.stdout
    x, y, z, yaw = m.groups()
    subprocess.run(common + ["--cam", x, y, z, yaw, "0", "--out", str(b)], check=True, capture_output=True, timeout=120)
    ia, ib = np.fromfile(a, np.float32), np.fromfile(b, np.float32)
    assert ia.size == 320 * 184 * 4 and np.array_equal(ia.view(np.uint32), ib.view(np.uint32))
    assert float(ia.reshape(184, 320, 4)[:, :, :3].max()) > 0


@pytest.mark.gpu
def test_pt_main_group_of_one_matches_batch_within_tolerance(tmp_path):
    """--group RANK WORLD IDFILE at world 1 (all a one-GPU box can hold): id file hand-over, pt_group_init/render/finalize;
    sum/spp vs the running mean of --batch, same tolerance as test_sum_mode_matches_mean_within_tolerance"""
    obj, a, b, idf = tmp_path / "bunny1.obj", tmp_path / "group.f32", tmp_path / "batch.f32", tmp_path / "rccl.id"
    _write_obj(obj)
    common = [PT_MAIN, "--mesh", str(obj), "--width", "320", "--height", "184", "--spp", "16", "--bounces", "4"]
    subprocess.run(common + ["--group", "0", "1", str(idf), "--out", str(a)], check=True, capture_output=True, timeout=120)
    subprocess.run(common + ["--batch", "--out", str(b)], check=True, capture_output=True, timeout=120)
    assert not os.path.exists(idf), "rank 0 removes the id file once the group exists (host/GroupHandshake.h)"
    ia, ib = np.fromfile(a, np.float32).reshape(184, 320, 4), np.fromfile(b, np.float32).reshape(184, 320, 4)
    assert np.all(ia[:, :, 3] == 1.0)
    np.testing.assert_allclose(ia[:, :, :3], ib[:, :, :3], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_pt_main_merges_added_meshes(tmp_path):
    """--add-mesh PATH MATERIAL: the C++ Mesh(parts) constructor + the `f a/b/c` form of the OBJ reader; the image must be
    the one the Python host renders from the same merged mesh, bit for bit"""
    from path_tracer_amd import capi, scene as S
    from test_gpu_parity import _octahedron
    obj, octa, out = tmp_path / "bunny1.obj", tmp_path / "octa.obj", tmp_path / "merged.f32"
    _write_obj(obj)
    part, v, f = _octahedron((0.3, 0.35, 0.7), 0.25, 0)
    with open(octa, "w") as fh:
        for p in v:
            fh.write("v %.9g %.9g %.9g\n" % tuple(float(c) for c in p))
        for t in f:
            fh.write("f %d/1/1 %d/2/2 %d/3/3\n" % tuple(int(i) + 1 for i in t))
    subprocess.run([PT_MAIN, "--mesh", str(obj), "--add-mesh", str(octa), "0", "--width", "320", "--height", "184", "--spp", "4",
                    "--bounces", "6", "--batch", "--out", str(out)], check=True, capture_output=True, timeout=120)
    got = np.fromfile(out, np.float32).reshape(184, 320, 4)
    sc = S.default_scene(S.Mesh.merged([S.bunny_mesh(3), part]))
    r = capi.Renderer(320, 184, device=0)
    r.upload_scene(sc)
    r.SetUniformCamera(S.bench_camera())
    r.SetUniform1i("maxBounces", 6)
    r.render(1, 4)
    want = r.read_accum()
    r.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_cmake_build_produces_library_host_and_driver(tmp_path):
    """north_star: 'a Linux CMake build' in place of the reference's Windows project: configure + build out of tree (hipcc
    cross-compiles gfx950 without a GPU), then the built library must export the whole C ABI and the driver must run"""
    import ctypes
    import re
    import shutil
    if not (shutil.which("cmake") and shutil.which("ninja")):
        pytest.skip("cmake / ninja not installed")
    b = tmp_path / "build"
    subprocess.run(["cmake", "-S", ROOT, "-B", str(b), "-G", "Ninja"], check=True, capture_output=True, timeout=300)
    subprocess.run(["cmake", "--build", str(b), "-j", "8"], check=True, capture_output=True, timeout=900)
    lib = ctypes.CDLL(str(b / "libpt_hip.so"))
    hdr = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    for name in set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", hdr)):
        assert hasattr(lib, name), name
    e = json.load(open(os.path.join(GOLD, "camera.json")))["cameras"][0]
    args = [f"{v:.9g}" for v in e["position"]] + [f"{e['yaw']:.9g}", f"{e['pitch']:.9g}"]
    out = subprocess.run([str(b / "pt_main"), "--print-camera", "--cam", *args], check=True, capture_output=True, text=True).stdout.split()
    assert out == e["inverse_view_bits"]


def _handshake(tmp_path, world, stale=(), delays=None, timeout="20"):
    """run `pt_main --handshake-only` once per rank (host-only: returns before any GPU call) and collect the printed ids"""
    import time
    idfile = tmp_path / "group.id"
    for name, content in stale:
        (tmp_path / name).write_bytes(content)
    procs = []
    for rank in (delays or range(world)):
        procs.append((rank, subprocess.Popen([PT_MAIN, "--handshake-only", str(rank), str(world), str(idfile), "--handshake-timeout", timeout],
                                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        if delays:
            time.sleep(0.3)
    outs = {}
    for rank, p in procs:
        o, e = p.communicate(timeout=60)
        outs[rank] = (p.returncode, o.strip(), e.strip())
    return idfile, outs


@pytest.mark.parametrize("world", [1, 2, 3])
def test_group_id_file_handshake_world_n(tmp_path, world):
    """the file protocol `pt_main --group` uses to hand the RCCL id from rank 0 to the others (host/GroupHandshake.h), run as
    `world` real processes with a stand-in id: every rank ends up with rank 0's 128 bytes, nothing is left behind"""
    idfile, outs = _handshake(tmp_path, world)
    assert all(rc == 0 for rc, _, _ in outs.values()), outs
    ids = {o.split(" id ")[1] for _, o, _ in outs.values()}
    assert len(ids) == 1 and len(ids.pop()) == 256
    assert sorted(os.listdir(tmp_path)) == [], os.listdir(tmp_path)


def test_group_id_file_handshake_ignores_files_of_a_dead_run(tmp_path):
    """an id file and request files left by a run that died must not be taken for this run's (a stale IDFILE made
    ncclCommInitRank hang); the late starters go first so that rank 0 meets the stale request of rank 2 before the live one"""
    stale = [("group.id", b"\xee" * (128 + 16)), ("group.id.req1", b"\x11" * 8), ("group.id.req2", b"\x22" * 8)]
    idfile, outs = _handshake(tmp_path, 3, stale=stale, delays=[1, 0, 2])
    assert all(rc == 0 for rc, _, _ in outs.values()), outs
    ids = {o.split(" id ")[1] for _, o, _ in outs.values()}
    assert len(ids) == 1 and ids.pop() != "ee" * 128
    assert sorted(os.listdir(tmp_path)) == []


def test_group_id_file_handshake_times_out_without_peers(tmp_path):
    idfile, outs = _handshake(tmp_path, 2, delays=[1], timeout="1")
    rc, _, err = outs[1]
    assert rc != 0 and "no id for rank 1" in err
    assert sorted(os.listdir(tmp_path)) == []


def _srgb8_expected(v):
    """GL_FRAMEBUFFER_SRGB's encoding (src/Renderer.cpp:135) in float64: clamp, linear -> sRGB, round to 8 bits; also returns
    the distance of 255*s from the nearest rounding boundary, so that values a float32 evaluation may round the other way
    can be left out of an exact comparison"""
    v = np.clip(np.nan_to_num(np.asarray(v, np.float64), nan=0.0), 0.0, 1.0)
    s = np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(v, 1.0 / 2.4) - 0.055)
    x = s * 255.0
    return np.floor(x + 0.5).astype(np.int64), np.abs((x + 0.5) - np.round(x + 0.5))


def test_present_values_follow_the_srgb_transfer_and_gl_row_order(tmp_path):
    """the display step (ShaderProgram::WritePPM <- src/Renderer.cpp:135 GL_FRAMEBUFFER_SRGB + compute_accum.frag:9-11): exact
    8-bit values for a ramp, out-of-range and non-finite inputs, alpha ignored, and image row 0 (GL's bottom row) written last"""
    W, H = 256, 20
    img = np.zeros((H, W, 4), np.float32)
    ramp = (np.arange(W * (H - 4), dtype=np.float64) / (W * (H - 4) - 1)).astype(np.float32).reshape(H - 4, W)
    img[:H - 4, :, 0] = ramp                            # full ramp 0..1, 4096 steps
    img[:H - 4, :, 1] = ramp * np.float32(0.0031308 * 2)  # the linear segment and the knee
    img[:H - 4, :, 2] = ramp ** 3                        # dense near black
    img[H - 4, :, :3] = np.float32(-0.5)                 # below range -> 0
    img[H - 3, :, :3] = np.float32(7.25)                 # above range -> 255
    img[H - 2, :, 0], img[H - 2, :, 1], img[H - 2, :, 2] = np.float32("nan"), np.float32("inf"), np.float32("-inf")
    img[H - 1, :, :3] = np.float32(0.5)                  # top row in GL terms
    img[:, :, 3] = np.linspace(-3, 3, W, dtype=np.float32)[None, :]     # alpha must not matter
    raw, ppm = tmp_path / "ramp.f32", tmp_path / "ramp.ppm"
    img.tofile(raw)
    subprocess.run([PT_MAIN, "--present-file", str(raw), str(W), str(H), str(ppm)], check=True, capture_output=True)
    data = open(ppm, "rb").read()
    header = b"P6\n%d %d\n255\n" % (W, H)
    assert data.startswith(header) and len(data) == len(header) + W * H * 3
    px = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)[::-1]          # back to GL row order
    want, margin = _srgb8_expected(img[:, :, :3])
    sure = margin > 1e-3                                  # 255*s at least 1e-3 away from x.5: no float32 evaluation rounds the other way
    assert sure.mean() > 0.99
    assert np.array_equal(px[sure], want[sure].astype(np.uint8))
    assert np.all(np.abs(px.astype(int) - want) <= 1)
    assert np.all(px[H - 4] == 0) and np.all(px[H - 3] == 255)
    assert np.all(px[H - 2, :, 0] == 0) and np.all(px[H - 2, :, 1] == 255) and np.all(px[H - 2, :, 2] == 0)
    assert np.all(px[H - 1] == 188)                      # 0.5 linear = 0.7354 sRGB = 187.5 -> 188 (255 * 0.73536 = 187.52)
    # the file's FIRST row is GL's top row
    first = np.frombuffer(data[len(header):len(header) + W * 3], np.uint8)
    assert np.all(first == 188)
    # anchors of the transfer function itself
    assert _srgb8_expected(np.array([0.0, 1.0, 0.0031308, 0.2158605, 0.5]))[0].tolist() == [0, 255, 10, 128, 188]


# ---- the input path of the interactive loop (src/Camera.cpp:28-59, src/Renderer.cpp:142-166, src/Main.cpp:108-150) ----------

def _print_input(start, script):
    out = subprocess.run([PT_MAIN, "--print-input", script, "--cam", *[f"{v:.9g}" for v in start]], check=True, capture_output=True, text=True)
    return [ln.split() for ln in out.stdout.strip().splitlines()]


def test_process_input_matches_the_reference_camera():
    """Camera::ProcessInput + UpdateView (host/Camera.cpp) against the reference's own src/Camera.cpp, compiled where it lies
    (oracle/camera_input_ref.cpp -> tests/golden/camera_input.json): every key, key pairs (W wins over S, A over D, E over Q),
    no key, dt = 0; position / forward / right / up / inverseView bit for bit and the `moving` flag after each frame"""
    d = json.load(open(os.path.join(GOLD, "camera_input.json")))
    assert len(d["scripts"]) >= 3
    for sc in d["scripts"]:
        got = _print_input(sc["start"], ",".join(sc["steps"]))
        assert got == sc["frames"], sc["steps"]
    # W moves along -forward (src/Camera.cpp:30-34), E up, Q down; a frame without keys leaves `moving` cleared
    f = lambda w: np.array([int(x, 16) for x in w], np.uint32).view(np.float32)
    fr = d["scripts"][0]["frames"]
    assert d["scripts"][0]["steps"][0] == "W:0.5" and f(fr[0][:3])[0] == np.float32(1.5) - np.float32(2.0) * np.float32(0.5)
    assert fr[0][-1] == "1" and fr[2][-1] == "0"


def test_mouse_look_follows_renderer_mouse_callback():
    """Renderer::MouseCallback / Camera::MouseLook against a float32 restatement of src/Renderer.cpp:142-166: offsets taken in
    double and narrowed, sensitivity 0.1, yaw += x, pitch -= (lastY - y), clamp to +-89 degrees, `moving` set by any motion"""
    f32 = np.float32
    moves = [(10.0, 5.0), (10.0, 5.0), (13.25, -40.5), (2000.0, -3000.0), (1999.5, 9000.0), (1999.5, 8999.0)]
    script = ",".join("@%.9g:%.9g" % m for m in moves)
    got = _print_input((0.3, 0.2, 0.1, 12.5, -3.0), script)
    yaw, pitch, lx, ly = f32(12.5), f32(-3.0), f32(0.0), f32(0.0)
    for (x, y), line in zip(moves, got):
        xo, yo = f32(x - float(lx)), f32(float(ly) - y)              # :144-145
        lx, ly = f32(x), f32(y)
        moving = abs(xo) > 0 or abs(yo) > 0                          # :149
        xo, yo = f32(xo * f32(0.1)), f32(yo * f32(0.1))              # :151-153
        yaw, pitch = f32(yaw + xo), f32(pitch - yo)                  # :155-156
        pitch = min(max(pitch, f32(-89.0)), f32(89.0))               # :158-159
        assert line[0] == "mouse" and f32(line[1]) == yaw and f32(line[2]) == pitch, (x, y, line)
        # `moving` stays set until the frame loop clears it; the second event repeats the position: no motion on its own
        assert line[3] == "1" or not moving
    # the cursor going UP the screen (smaller y) gives a positive yOffset, and `pitch -= yOffset` (:156) turns the camera DOWN
    assert f32(got[3][2]) == f32(-89.0) and f32(got[4][2]) == f32(89.0)
    # a mouse move alone restarts the count: the next frame sees moving = 1 and clears it
    got = _print_input((0, 0, 0, 0, 0), "@1:0,-:0.016,-:0.016")
    assert got[0][3] == "1" and got[1][-1] == "0" and got[2][-1] == "0"


@pytest.mark.gpu
def test_pt_main_keys_loop_keeps_the_reference_frame_order(tmp_path):
    """`pt_main --keys`: src/Main.cpp:108-150 with scripted input.  Emulated frame by frame with the oracle: the frame in which
    W goes down still blends into the old image as pass 3, the count restarts one frame later WITHOUT clearing the image
    (pass 1 = accum + 1.0 * (colour - accum), rt.comp:337), N toggles debugNormal on the key's edge only"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from path_tracer_amd import scene as S
    W, H, B = 64, 40, 4
    script = "-:0.016,-:0.016,W:0.05,-:0.016,-:0.016,N:0.016,N:0.016,-:0.016,@3:-2,-:0.016"
    obj, out = tmp_path / "bunny1.obj", tmp_path / "keys.f32"
    _write_obj(obj)
    r = subprocess.run([PT_MAIN, "--mesh", str(obj), "--width", str(W), "--height", str(H), "--bounces", str(B), "--keys", script,
                        "--out", str(out)], check=True, capture_output=True, text=True, timeout=120)
    assert "Input frames: 9 | Samples: 1 " in r.stdout and "debugNormal 1" in r.stdout, r.stdout
    got = np.fromfile(out, np.float32).reshape(H, W, 4)
    frames = _print_input((1.5, 0.4, 0.0, 0.0, 0.0), script)
    sc = S.default_scene()
    img = np.zeros((H, W, 4), np.float32)
    passes = [1, 2, 3, 1, 2, 3, 4, 5, 1]                 # W in frame 3 -> restart in frame 4; the mouse move -> restart in the last frame
    debug = [0, 0, 0, 0, 0, 1, 1, 1, 1]
    cams = [ln for ln in frames if ln[0] != "mouse"]
    assert len(cams) == len(passes)
    for ln, p, dbg in zip(cams, passes, debug):
        w = np.array([int(x, 16) for x in ln[:-1]], np.uint32).view(np.float32)
        cam = S.Camera(position=w[:3].copy())
        cam.inverseView = w[12:28].reshape(4, 4).copy()
        O.Oracle(sc, cam, W, H, B, debug_normal=bool(dbg)).render(p, 1, accum=img)
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))
