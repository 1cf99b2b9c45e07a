// Main.cpp — headless counterpart of the reference's main() (src/Main.cpp:9-174): same five materials, same
// mesh + light + ground scene, same per-frame sequence (currAccumPass++, SetUniform1i, SetUniformCamera,
// dispatch), with the window loop replaced by "render --spp passes, write the image".
//   pt_main --mesh bunny1.obj [--scene default|mixed] [--width 1920 --height 1080] [--spp 64] [--bounces 8]
//           [--cam x y z yaw pitch] [--debug-normal] [--batch] [--out accum.f32] [--ppm image.ppm] [--device 0]
//           [--orbit FRAMES DEGREES]   scripted stand-in for the interactive loop (src/Main.cpp:108-150): every frame the
//                                      camera turns by DEGREES of yaw around the scene centre, which restarts the
//                                      accumulation (currAccumPass = 0 + clear, src/Main.cpp:110-117), then --spp passes
//           [--keys SCRIPT]            the interactive loop itself (src/Main.cpp:108-150) with scripted input, one frame per
//                                      comma-separated step: KEYS:DT (KEYS = letters out of WASDEQN or '-', DT = the frame's
//                                      deltaTime in seconds) or @X:Y (a mouse move to X,Y delivered before the next frame,
//                                      Renderer::MouseCallback).  Per frame, in the reference's order: a camera that moved in
//                                      the previous frame restarts the count (currAccumPass = 0; the image itself is NOT
//                                      cleared: pass 1 blends with weight 1/1, rt.comp:337), ProcessInput, UpdateView, N toggles
//                                      debugNormal on the key's edge, currAccumPass++, dispatch.  --spp is ignored.
//           [--print-input SCRIPT]     host-only: the same loop without rendering; prints per frame position, forward, right,
//                                      up, inverseView as hex float32 words and `moving` (tests against the reference's Camera)
//           [--add-mesh PATH MATERIAL]  (repeatable) further OBJ files, merged with --mesh into one traced mesh
//           [--group RANK WORLD IDFILE] one process per GPU (give each its --device): the --spp passes are split over the
//                                      WORLD ranks, one RCCL reduce brings the sums to rank 0, which writes the outputs
//                                      (pt_group_*, include/pt_hip.h); IDFILE is a path all ranks can reach
//           [--present-file ACCUM.f32 W H OUT.ppm]  host-only: the display step alone (ShaderProgram::WritePPM) on a raw
//                                      RGBA32F image, row 0 = bottom row, as --out writes it
//           [--handshake-only RANK WORLD IDFILE]  host-only: run just the id-file protocol of --group (GroupHandshake.h) with
//                                      a stand-in id made on rank 0 and print the id every rank ends up with (tests)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "GroupHandshake.h"
#include "Renderer.h"

using namespace ptm;

static void default_materials(Scene &scene)
{
    Material specular;
    specular.baseColor = vec4(0.2, 0.9, 0.1, 1.0); specular.smoothness = 0.9f; specular.emissionColor = vec4(0);
    scene.materials.push_back(specular);
    Material diffuse;
    diffuse.baseColor = vec4(0.9, 0.1, 0.1, 1.0); diffuse.smoothness = 0.2f; diffuse.specularSmoothness = 0.99f;
    diffuse.specularChance = 0.1f; diffuse.emissionColor = vec4(0);
    scene.materials.push_back(diffuse);
    Material glass;
    glass.baseColor = vec4(0.9, 0.1, 0.1, 1.0); glass.smoothness = 0.99f; glass.specularSmoothness = 0.99f;
    glass.specularChance = 0.2f; glass.emissionColor = vec4(0); glass.refractionAmount = 0.8f; glass.ior = 1.05f;
    scene.materials.push_back(glass);
    Material ground;
    ground.baseColor = vec4(0.9, 0.9, 0.9, 1.0); ground.emissionColor = vec4(0);
    scene.materials.push_back(ground);
    Material sun;
    sun.baseColor = vec4(1.0f); sun.emissionColor = vec4(1.0f); sun.emissionStrength = 5.0f;
    scene.materials.push_back(sun);
}

// One frame of scripted input (--keys / --print-input): a key set held during the frame with the frame's deltaTime, or a
// mouse move delivered by glfwPollEvents before it.
struct InputStep { bool mouse = false; unsigned keys = 0; bool n_key = false; float dt = 0.0f; double mx = 0.0, my = 0.0; };

static bool parse_input_script(const std::string &script, std::vector<InputStep> &out)
{
    size_t pos = 0;
    while (pos <= script.size()) {
        const size_t comma = std::min(script.find(',', pos), script.size());
        const std::string tok = script.substr(pos, comma - pos);
        pos = comma + 1;
        if (tok.empty()) { if (comma >= script.size()) break; continue; }
        InputStep st;
        const size_t colon = tok.find(':');
        if (colon == std::string::npos) return false;
        if (tok[0] == '@') {
            st.mouse = true;
            st.mx = std::strtod(tok.c_str() + 1, nullptr); st.my = std::strtod(tok.c_str() + colon + 1, nullptr);
        } else {
            for (size_t k = 0; k < colon; k++)
                switch (tok[k]) {
                case 'W': st.keys |= KEY_W; break; case 'S': st.keys |= KEY_S; break; case 'A': st.keys |= KEY_A; break;
                case 'D': st.keys |= KEY_D; break; case 'E': st.keys |= KEY_E; break; case 'Q': st.keys |= KEY_Q; break;
                case 'N': st.n_key = true; break; case '-': break;
                default: return false;
                }
            st.dt = std::strtof(tok.c_str() + colon + 1, nullptr);
        }
        out.push_back(st);
        if (comma >= script.size()) break;
    }
    return !out.empty();
}

static void print_camera_state(const Camera &c)
{
    auto put = [](float f) { unsigned u; std::memcpy(&u, &f, 4); std::printf("%08x ", u); };
    for (const vec3 &v : {c.position, c.forward, c.right, c.up}) { put(v.x); put(v.y); put(v.z); }
    for (int k = 0; k < 16; k++) put(c.inverseView.data()[k]);
    std::printf("%d\n", c.moving ? 1 : 0);
}

int main(int argc, char **argv)
{
    std::string mesh_path = "res/meshes/bunny1.obj", scene_name = "default", out_path, ppm_path, shader_path = "res/shaders/rt.comp";
    int width = 1920, height = 1080, spp = 64, bounces = 0, device = 0;
    bool debugNormal = false, batch = false, cam_given = false, print_camera = false;
    float cam[5] = {1.5f, 0.4f, 0.0f, 0.0f, 0.0f};
    int orbit_frames = 0; float orbit_deg = 0.0f;
    std::string keys_script; bool print_input = false;
    std::vector<std::pair<std::string, int>> more_meshes;               // --add-mesh PATH MATERIAL (repeatable)
    int group_rank = 0, group_world = 0; std::string group_id_path;     // --group: one process per GPU, samples split over RCCL
    bool handshake_only = false; double handshake_timeout = 60.0;
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--mesh")) mesh_path = next();
        else if (!std::strcmp(argv[i], "--scene")) scene_name = next();
        else if (!std::strcmp(argv[i], "--shader")) shader_path = next();    // .../rt.comp (default) or .../rt.frag: which copy of the kernel to reproduce
        else if (!std::strcmp(argv[i], "--width")) width = std::atoi(next());
        else if (!std::strcmp(argv[i], "--height")) height = std::atoi(next());
        else if (!std::strcmp(argv[i], "--spp")) spp = std::atoi(next());
        else if (!std::strcmp(argv[i], "--bounces")) bounces = std::atoi(next());
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--out")) out_path = next();
        else if (!std::strcmp(argv[i], "--ppm")) ppm_path = next();
        else if (!std::strcmp(argv[i], "--debug-normal")) debugNormal = true;
        else if (!std::strcmp(argv[i], "--batch")) batch = true;
        else if (!std::strcmp(argv[i], "--print-camera")) print_camera = true;
        else if (!std::strcmp(argv[i], "--orbit")) { orbit_frames = std::atoi(next()); orbit_deg = std::strtof(next(), nullptr); }
        else if (!std::strcmp(argv[i], "--keys")) keys_script = next();
        else if (!std::strcmp(argv[i], "--print-input")) { keys_script = next(); print_input = true; }
        else if (!std::strcmp(argv[i], "--add-mesh")) { const char *p = next(); more_meshes.emplace_back(p, std::atoi(next())); }
        else if (!std::strcmp(argv[i], "--group")) { group_rank = std::atoi(next()); group_world = std::atoi(next()); group_id_path = next(); }
        else if (!std::strcmp(argv[i], "--handshake-only")) { group_rank = std::atoi(next()); group_world = std::atoi(next()); group_id_path = next(); handshake_only = true; }
        else if (!std::strcmp(argv[i], "--present-file")) {
            const char *in = next(); const int w = std::atoi(next()), h = std::atoi(next()); const char *out = next();
            if (w < 1 || h < 1) { std::fprintf(stderr, "bad size\n"); return 2; }
            std::vector<float> img((size_t)w * h * 4);
            FILE *f = std::fopen(in, "rb");
            if (!f || std::fread(img.data(), sizeof(float), img.size(), f) != img.size()) { std::perror(in); return 1; }
            std::fclose(f);
            return ShaderProgram::WritePPM(img.data(), w, h, out) ? 0 : 1;
        }
        else if (!std::strcmp(argv[i], "--handshake-timeout")) handshake_timeout = std::strtod(next(), nullptr);
        else if (!std::strcmp(argv[i], "--cam")) { for (int k = 0; k < 5; k++) cam[k] = std::strtof(next(), nullptr); cam_given = true; }
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (print_camera) {     // host-only check of Camera::UpdateView (no GPU needed): 16 hex words, column-major
        Camera c(vec3(cam[0], cam[1], cam[2]));
        c.yaw = cam[3]; c.pitch = cam[4];
        c.UpdateView();
        for (int k = 0; k < 16; k++) { unsigned u; std::memcpy(&u, c.inverseView.data() + k, 4); std::printf("%08x%c", u, k == 15 ? '\n' : ' '); }
        return 0;
    }
    std::vector<InputStep> input;
    if (!keys_script.empty() && !parse_input_script(keys_script, input)) { std::fprintf(stderr, "bad input script '%s'\n", keys_script.c_str()); return 2; }
    if (print_input) {      // host-only: Camera::ProcessInput / MouseLook / UpdateView in the frame loop's order, no rendering
        Camera &c = Renderer::camera;
        c.position = vec3(cam[0], cam[1], cam[2]); c.yaw = cam[3]; c.pitch = cam[4];
        c.UpdateView();
        for (const InputStep &st : input) {
            if (st.mouse) { Renderer::MouseCallback(st.mx, st.my); std::printf("mouse %.9g %.9g %d\n", c.yaw, c.pitch, c.moving ? 1 : 0); continue; }
            if (c.moving) c.moving = false;
            c.ProcessInput(st.keys, st.dt);
            c.UpdateView();
            print_camera_state(c);
        }
        return 0;
    }
    if (group_world > 0 && (group_rank < 0 || group_rank >= group_world)) { std::fprintf(stderr, "bad rank %d of %d\n", group_rank, group_world); return 2; }
    if (handshake_only) {   // no GPU: the stand-in id is 128 bytes derived from rank 0's nonce
        char id[PT_GROUP_ID_BYTES];
        auto make = [](char *out) { uint64_t x = handshake_nonce(); for (int k = 0; k < PT_GROUP_ID_BYTES; k++) { x = x * 6364136223846793005ull + 1442695040888963407ull; out[k] = (char)(x >> 56); } return true; };
        if (!handshake_exchange(group_rank, group_world, group_id_path, id, sizeof id, make, handshake_timeout)) return 1;
        handshake_cleanup(group_rank, group_id_path);
        std::printf("rank %d id ", group_rank);
        for (int k = 0; k < PT_GROUP_ID_BYTES; k++) std::printf("%02x", (unsigned char)id[k]);
        std::printf("\n");
        return 0;
    }
    const bool mixed = scene_name == "mixed";
    if (mixed && !cam_given) { cam[0] = 1.5f; cam[1] = 0.8f; cam[2] = 1.5f; cam[3] = 45.0f; cam[4] = 20.0f; }

    if (Renderer::Init(width, height, device)) return 1;
    ComputeProgram computeProgram(shader_path.c_str());
    if (bounces <= 0) bounces = computeProgram.legacy() ? 12 : 8;       // rt.frag:337 | this driver's default
    ShaderProgram computeAccumProgram;

    default_materials(Renderer::scene);
    Mesh mesh(mesh_path.c_str(), mixed ? 1 : 3);
    if (!mesh.ok()) return 1;
    // further meshes: traced together with the first as one merged mesh (the reference would keep only the last one)
    std::vector<std::unique_ptr<Mesh>> extra;
    std::unique_ptr<Mesh> merged;
    if (!more_meshes.empty()) {
        std::vector<const Mesh *> parts{&mesh};
        for (const auto &mm : more_meshes) {
            extra.emplace_back(new Mesh(mm.first.c_str(), (uint32_t)mm.second));
            if (!extra.back()->ok()) return 1;
            parts.push_back(extra.back().get());
        }
        merged.reset(new Mesh(parts));
        if (!merged->ok()) return 1;
    }
    Triangle(Renderer::scene, vec3(-1.0, 0.0, 3.0), vec3(1.0, 0.0, 3.0), vec3(0.0, 1.4, 3.0), 4);
    Triangle(Renderer::scene, vec3(5000.0, 0.0, 5000.0), vec3(-5000.0, 0.0, 5000.0), vec3(0.0, 0.0, -5000.0), 3);
    if (mixed) {    // the sphere grid main() carries commented out
        for (int i = -6; i < 6; i++)
            for (int k = -6; k < 6; k++)
                Sphere(Renderer::scene, vec3(((float)i) / 3.0, ((float)0) + 0.101, ((float)k) / 3.0), 0.1f, 2);
        Sphere(Renderer::scene, vec3(-0.2, 1.0, 0.0), 0.5f, 0);
    }
    Renderer::scene.SetupSSBOs();

    Renderer::camera.position = vec3(cam[0], cam[1], cam[2]);
    Renderer::camera.yaw = cam[3];
    Renderer::camera.pitch = cam[4];
    Renderer::camera.UpdateView();
    computeProgram.SetUniform1i("debugNormal", debugNormal);
    computeProgram.SetUniform1i("maxBounces", bounces);
    computeProgram.SetUniformCamera(Renderer::camera);

    const auto t0 = std::chrono::steady_clock::now();
    if (orbit_frames > 0) {
        // the camera moves every frame: like the reference while a key is held, each frame starts a new accumulation
        const float radius = std::sqrt(cam[0] * cam[0] + cam[2] * cam[2]);
        const float start = std::atan2(cam[2], cam[0]);
        for (int frame = 1; frame <= orbit_frames; frame++) {
            const float a = start + radians(orbit_deg * (float)frame);
            Renderer::camera.position = vec3(radius * std::cos(a), cam[1], radius * std::sin(a));
            Renderer::camera.yaw = cam[3] + orbit_deg * (float)frame;       // keeps looking at the y axis
            Renderer::camera.UpdateView();
            if (!Renderer::Check(pt_reset(Renderer::ctx), "pt_reset")) return 1;
            for (int currAccumPass = 1; currAccumPass <= spp; currAccumPass++) {
                computeProgram.SetUniform1i("currAccumPass", currAccumPass);
                computeProgram.SetUniformCamera(Renderer::camera);
                computeProgram.Dispatch();
            }
        }
        if (!Renderer::Check(pt_synchronize(Renderer::ctx), "pt_synchronize")) return 1;
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("Frames: %d x %d spp | %.3f ms per frame | %.1f frames/s | final camera %.9g %.9g %.9g yaw %.9g\n", orbit_frames, spp,
                    dt * 1e3 / orbit_frames, orbit_frames / dt, Renderer::camera.position.x, Renderer::camera.position.y,
                    Renderer::camera.position.z, Renderer::camera.yaw);
    } else if (!input.empty()) {
        // src/Main.cpp:108-150 with scripted input.  Kept from the reference on purpose: the frame in which a key goes down
        // still blends into the old image (the count restarts one frame later), and a restart does not clear the image --
        // pass 1 is accum + 1.0 * (colour - accum) (rt.comp:337), which the short-launch kernels evaluate as written.
        int currAccumPass = 0, frames = 0;
        bool keyPressed = false;
        for (const InputStep &st : input) {
            if (st.mouse) { Renderer::MouseCallback(st.mx, st.my); continue; }      // glfwPollEvents at the end of the previous frame
            if (Renderer::camera.moving) { currAccumPass = 0; Renderer::camera.moving = false; }
            Renderer::camera.ProcessInput(st.keys, st.dt);
            Renderer::camera.UpdateView();
            if (st.n_key) {
                if (!keyPressed) { debugNormal = !debugNormal; computeProgram.SetUniform1i("debugNormal", debugNormal); }
                keyPressed = true;
            } else {
                keyPressed = false;
            }
            currAccumPass++;
            computeProgram.SetUniform1i("currAccumPass", currAccumPass);
            computeProgram.SetUniformCamera(Renderer::camera);
            computeProgram.Dispatch();
            frames++;
        }
        if (!Renderer::Check(pt_synchronize(Renderer::ctx), "pt_synchronize")) return 1;
        std::printf("Input frames: %d | Samples: %d | final camera %.9g %.9g %.9g yaw %.9g pitch %.9g | debugNormal %d\n", frames, currAccumPass,
                    Renderer::camera.position.x, Renderer::camera.position.y, Renderer::camera.position.z, Renderer::camera.yaw,
                    Renderer::camera.pitch, debugNormal ? 1 : 0);
    } else if (group_world > 0) {
        // rank 0 makes the RCCL id; it travels through IDFILE (GroupHandshake.h: nonce-checked, so a file left by a run that
        // died is never taken for this run's)
        char id[PT_GROUP_ID_BYTES];
        auto make = [](char *out) { return Renderer::Check(pt_group_unique_id(out), "pt_group_unique_id"); };
        if (!handshake_exchange(group_rank, group_world, group_id_path, id, sizeof id, make, handshake_timeout)) return 1;
        if (!Renderer::Check(pt_group_init(Renderer::ctx, id, group_rank, group_world), "pt_group_init")) return 1;
        handshake_cleanup(group_rank, group_id_path);
        if (!Renderer::Check(pt_group_render(Renderer::ctx, 1, spp), "pt_group_render")) return 1;
        if (!Renderer::Check(pt_synchronize(Renderer::ctx), "pt_synchronize")) return 1;
        if (!Renderer::Check(pt_group_finalize(Renderer::ctx), "pt_group_finalize")) return 1;
        if (group_rank != 0) { Renderer::Shutdown(); return 0; }       // the finished image lives on rank 0
    } else if (batch) {
        if (!Renderer::Check(pt_render(Renderer::ctx, 1, spp), "pt_render")) return 1;
    } else {
        int currAccumPass = 0;
        for (int frame = 0; frame < spp; frame++) {
            currAccumPass++;
            computeProgram.SetUniform1i("currAccumPass", currAccumPass);
            computeProgram.SetUniformCamera(Renderer::camera);
            computeProgram.Dispatch();
        }
    }
    if (!Renderer::Check(pt_synchronize(Renderer::ctx), "pt_synchronize")) return 1;
    if (orbit_frames <= 0 && input.empty()) {
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("Samples: %d | %.3f ms | %.1f Msamples/s\n", spp, dt * 1e3, (double)width * height * spp / dt * 1e-6);
    }

    if (!out_path.empty()) {
        std::vector<float> img((size_t)width * height * 4);
        if (!Renderer::Check(pt_read_accum(Renderer::ctx, img.data(), img.size() * sizeof(float)), "pt_read_accum")) return 1;
        FILE *f = std::fopen(out_path.c_str(), "wb");
        if (!f) { std::perror(out_path.c_str()); return 1; }
        std::fwrite(img.data(), sizeof(float), img.size(), f);
        std::fclose(f);
    }
    if (!ppm_path.empty() && !computeAccumProgram.Present(ppm_path.c_str())) return 1;
    Renderer::Shutdown();
    return 0;
}
