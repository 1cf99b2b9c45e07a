/*
 * llvmpipe_runner — TEST INFRASTRUCTURE (oracle), not product code.
 *
 * Runs a GLSL compute shader headless under Mesa llvmpipe by driving the DRI
 * swrast interface directly (no X, no EGL, no OSMesa).  Two uses:
 *
 *  1. --reference-dir DIR : load DIR/res/shaders/rt.comp (the reference's own
 *     path-tracing kernel, never copied into this repo), apply the
 *     arithmetic-neutral compile patch of SURVEY.md App. C.2 *in memory*
 *     (plus optional bounce-count literal and hit-id instrumentation,
 *     App. C.3), upload a scene blob to SSBO bindings 1..5 exactly as
 *     src/Object.cpp:3-41,97-110 do, and dispatch it the way
 *     src/Main.cpp:139-150 does (one dispatch == one accumulation pass).
 *  2. --shader-file FILE  : run one of our own probe shaders (oracle/probes/)
 *     that pin llvmpipe's arithmetic lowering.
 *
 * Output: raw little-endian float32 RGBA images (W*H*4 floats per dump),
 * row 0 = bottom row (GL texel origin), appended to --out in dump order.
 *
 * Build: see oracle/Makefile (gcc, -ldl).  Only this container has the
 * reference checkout; the GPU box uses the committed fixtures in
 * tests/golden/ produced by oracle/gen_goldens.py from this program.
 */
#define _GNU_SOURCE
#define GL_GLEXT_PROTOTYPES 0
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---------------------------------------------------------------- DRI glue */
static void ldr_getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16; }
static void ldr_putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ldr_getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ldr_putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void ldr_getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }

static const __DRIswrastLoaderExtension swrast_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = ldr_getDrawableInfo,
    .putImage = ldr_putImage,
    .getImage = ldr_getImage,
    .putImage2 = ldr_putImage2,
    .getImage2 = ldr_getImage2,
};
static const __DRIextension *loader_exts[] = { &swrast_loader.base, NULL };

typedef void (*glproc)(void);
static glproc (*get_proc)(const char *);
#define GLF(ret, name, ...) static ret (*p_##name)(__VA_ARGS__)
GLF(const GLubyte *, glGetString, GLenum);
GLF(GLenum, glGetError, void);
GLF(GLuint, glCreateShader, GLenum);
GLF(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
GLF(void, glCompileShader, GLuint);
GLF(void, glGetShaderiv, GLuint, GLenum, GLint *);
GLF(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(GLuint, glCreateProgram, void);
GLF(void, glAttachShader, GLuint, GLuint);
GLF(void, glLinkProgram, GLuint);
GLF(void, glGetProgramiv, GLuint, GLenum, GLint *);
GLF(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(void, glUseProgram, GLuint);
GLF(GLint, glGetUniformLocation, GLuint, const GLchar *);
GLF(void, glUniform1i, GLint, GLint);
GLF(void, glUniform3f, GLint, GLfloat, GLfloat, GLfloat);
GLF(void, glUniformMatrix4fv, GLint, GLsizei, GLboolean, const GLfloat *);
GLF(void, glGenBuffers, GLsizei, GLuint *);
GLF(void, glBindBuffer, GLenum, GLuint);
GLF(void, glBufferData, GLenum, GLsizeiptr, const void *, GLenum);
GLF(void, glBindBufferBase, GLenum, GLuint, GLuint);
GLF(void, glGenTextures, GLsizei, GLuint *);
GLF(void, glBindTexture, GLenum, GLuint);
GLF(void, glTexParameteri, GLenum, GLenum, GLint);
GLF(void, glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
GLF(void, glBindImageTexture, GLuint, GLuint, GLint, GLboolean, GLint, GLenum, GLenum);
GLF(void, glDispatchCompute, GLuint, GLuint, GLuint);
GLF(void, glMemoryBarrier, GLbitfield);
GLF(void, glFinish, void);
GLF(void, glGetTexImage, GLenum, GLint, GLenum, GLenum, void *);
/* fragment-shader mode (--frag): fullscreen quad into an RGBA32F colour attachment */
GLF(void, glGenFramebuffers, GLsizei, GLuint *);
GLF(void, glBindFramebuffer, GLenum, GLuint);
GLF(void, glFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint);
GLF(GLenum, glCheckFramebufferStatus, GLenum);
GLF(void, glGenVertexArrays, GLsizei, GLuint *);
GLF(void, glBindVertexArray, GLuint);
GLF(void, glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void *);
GLF(void, glEnableVertexAttribArray, GLuint);
GLF(void, glDrawElements, GLenum, GLsizei, GLenum, const void *);
GLF(void, glViewport, GLint, GLint, GLsizei, GLsizei);
GLF(void, glActiveTexture, GLenum);
GLF(void, glDisable, GLenum);
#define LOAD(name) do { p_##name = (void *)get_proc(#name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); exit(2); } } while (0)

static void die(const char *m) { fprintf(stderr, "llvmpipe_runner: %s\n", m); exit(2); }

static void make_context(void)
{
    void *h = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) die(dlerror());
    const __DRIextension **(*getext)(void) = dlsym(h, "__driDriverGetExtensions_swrast");
    if (!getext) die("no __driDriverGetExtensions_swrast");
    const __DRIextension **exts = getext();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *swrast = NULL;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const void *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const void *)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) die("DRI_Core/DRI_SWRast v4 not found");
    const __DRIconfig **configs = NULL;
    __DRIscreen *scr = swrast->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!scr || !configs || !configs[0]) die("createNewScreen2 failed");
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5 };
    unsigned err = 0;
    __DRIcontext *ctx = swrast->createContextAttribs(scr, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) { fprintf(stderr, "createContextAttribs error %u\n", err); exit(2); }
    __DRIdrawable *draw = swrast->createNewDrawable(scr, configs[0], NULL);
    if (!draw) die("createNewDrawable failed");
    if (!core->bindContext(ctx, draw, draw)) die("bindContext failed");

    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) die(dlerror());
    get_proc = dlsym(glapi, "_glapi_get_proc_address");
    if (!get_proc) die("no _glapi_get_proc_address");
    LOAD(glGetString); LOAD(glGetError); LOAD(glCreateShader); LOAD(glShaderSource); LOAD(glCompileShader);
    LOAD(glGetShaderiv); LOAD(glGetShaderInfoLog); LOAD(glCreateProgram); LOAD(glAttachShader);
    LOAD(glLinkProgram); LOAD(glGetProgramiv); LOAD(glGetProgramInfoLog); LOAD(glUseProgram);
    LOAD(glGetUniformLocation); LOAD(glUniform1i); LOAD(glUniform3f); LOAD(glUniformMatrix4fv);
    LOAD(glGenBuffers); LOAD(glBindBuffer); LOAD(glBufferData); LOAD(glBindBufferBase);
    LOAD(glGenTextures); LOAD(glBindTexture); LOAD(glTexParameteri); LOAD(glTexImage2D);
    LOAD(glBindImageTexture); LOAD(glDispatchCompute); LOAD(glMemoryBarrier); LOAD(glFinish);
    LOAD(glGetTexImage);
    LOAD(glGenFramebuffers); LOAD(glBindFramebuffer); LOAD(glFramebufferTexture2D); LOAD(glCheckFramebufferStatus);
    LOAD(glGenVertexArrays); LOAD(glBindVertexArray); LOAD(glVertexAttribPointer); LOAD(glEnableVertexAttribArray);
    LOAD(glDrawElements); LOAD(glViewport); LOAD(glActiveTexture); LOAD(glDisable);
}

/* ------------------------------------------------------------ text helpers */
static char *read_file(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *b = malloc(n + 1);
    if (fread(b, 1, n, f) != (size_t)n) die("short read");
    b[n] = 0; fclose(f);
    if (len) *len = n;
    return b;
}

/* replace the first (all=0) or every (all=1) occurrence of `from` at or after
 * byte offset `start`; returns number of replacements; *ps is re-allocated */
static int subst(char **ps, size_t start, const char *from, const char *to, int all)
{
    int n = 0; size_t fl = strlen(from), tl = strlen(to);
    size_t pos = start;
    for (;;) {
        char *s = *ps; char *hit = strstr(s + pos, from);
        if (!hit) break;
        size_t off = hit - s, sl = strlen(s);
        char *r = malloc(sl - fl + tl + 1);
        memcpy(r, s, off); memcpy(r + off, to, tl); strcpy(r + off + tl, hit + fl);
        free(s); *ps = r; n++; pos = off + tl;
        if (!all) break;
    }
    return n;
}
static void must(int n, const char *what) { if (n < 1) { fprintf(stderr, "patch step failed: %s\n", what); exit(2); } }

/* The arithmetic-neutral compile patch: Mesa rejects
 * #version 460, global `const` with non-constant initialisers, and
 * mix(vec3,vec3,bool).  bvec3 selects are exact, so no arithmetic changes. */
static char *patch_reference_shader(char *src, int bounces, int instrument)
{
    must(subst(&src, 0, "#version 460", "#version 450", 0), "version");
    must(subst(&src, 0, "const ivec2 texelCoord", "ivec2 texelCoord", 0), "const texelCoord");
    must(subst(&src, 0, "const vec2 viewport ", "vec2 viewport ", 0), "const viewport");
    must(subst(&src, 0, "const vec2 viewportCenter", "vec2 viewportCenter", 0), "const viewportCenter");
    must(subst(&src, 0, "const vec2 accumTexSize", "vec2 accumTexSize", 0), "const accumTexSize");
    must(subst(&src, 0, "const vec2 pixelPos", "vec2 pixelPos", 0), "const pixelPos");
    must(subst(&src, 0, "const int maxBounces", "int maxBounces", 0), "const maxBounces");
    /* the two bool-mix sites inside RayTrace */
    char *rt = strstr(src, "vec3 RayTrace(");
    if (!rt) die("RayTrace not found");
    size_t rto = rt - src;
    must(subst(&src, rto, "\t\t\t\t\tisRefracted\n", "\t\t\t\t\tbvec3(isRefracted)\n", 0), "mix bool #1");
    must(subst(&src, rto, "isSpecularBounce), hitInfo.hitMaterial.baseColor.rgb, isRefracted)",
                          "bvec3(isSpecularBounce)), hitInfo.hitMaterial.baseColor.rgb, bvec3(isRefracted))", 0), "mix bool #2");
    if (bounces > 0) {
        char to[64]; snprintf(to, sizeof to, "debugNormal ? 1 : %d;", bounces);
        must(subst(&src, 0, "debugNormal ? 1 : 3;", to, 0), "bounce literal");
    }
    if (instrument) {
        /* App. C.3: canonical primitive ids (mesh i, scene tri Nm+i, sphere Nm+Nt+i, miss -1)
         * and a pre-RayTrace dump selected by the uniform `ptDump`:
         *   1 -> (primId, hitDist, frontFace, rngState-after-jitter as float bits)
         *   2 -> (ray.direction.xyz, pixelIndex as float bits) */
        must(subst(&src, 0, "bool frontFace; Material hitMaterial; };", "bool frontFace; Material hitMaterial; int primId; };", 0), "HitInfo.primId");
        must(subst(&src, 0, "uniform bool debugNormal;", "uniform bool debugNormal;\nuniform int ptDump;\nuniform int ptBounce;\nvec4 ptOut = vec4(-7.0);", 0), "ptDump uniform");
        /*   3 -> state after bounce `ptBounce` of a path that hit: (ray.origin.xyz, rng state bits)
         *   4 -> (ray.direction.xyz, hitDist of that bounce); paths that ended earlier keep -7 */
        must(subst(&src, 0, "ray.origin += ray.direction * 0.0002;",
            "ray.origin += ray.direction * 0.0002;\n"
            "\t\t\tif (ptDump == 3 && i == ptBounce) ptOut = vec4(ray.origin, uintBitsToFloat(state));\n"
            "\t\t\tif (ptDump == 4 && i == ptBounce) ptOut = vec4(ray.direction, hitInfo.hitDist);\n", 0), "bounce dump");
        must(subst(&src, 0, "\tvec3 finalColor = mix(", "\tif (ptDump >= 3) { imageStore(accumImage, texelCoord, ptOut); return; }\n\tvec3 finalColor = mix(", 0), "bounce dump store");
        must(subst(&src, 0, "\tHitInfo tempHit;\n", "\tHitInfo tempHit;\n\tclosestHit.primId = -1;\n", 0), "init primId");
        must(subst(&src, 0, "tempHit = HitTriangle(meshTriangles[i], ray);\n", "tempHit = HitTriangle(meshTriangles[i], ray); tempHit.primId = i;\n", 0), "mesh primId");
        must(subst(&src, 0, "tempHit = HitTriangle(sceneTriangles[i], ray);\n", "tempHit = HitTriangle(sceneTriangles[i], ray); tempHit.primId = meshTriangles.length() + i;\n", 0), "tri primId");
        must(subst(&src, 0, "tempHit = HitSphere(sceneSpheres[i], ray);\n", "tempHit = HitSphere(sceneSpheres[i], ray); tempHit.primId = meshTriangles.length() + sceneTriangles.length() + i;\n", 0), "sphere primId");
        must(subst(&src, 0, "\tvec3 rayTraceColor = RayTrace(",
            "\tif (ptDump == 1) { HitInfo h = CalculateRay(ray); bool ok = h.primId >= 0 && h.hitDist < INFINITY;\n"
            "\t\timageStore(accumImage, texelCoord, vec4(ok ? float(h.primId) : -1.0, h.hitDist, (ok && h.frontFace) ? 1.0 : 0.0, uintBitsToFloat(rngState))); return; }\n"
            "\tif (ptDump == 2) { imageStore(accumImage, texelCoord, vec4(ray.direction, uintBitsToFloat(pixelIndex))); return; }\n"
            "\tvec3 rayTraceColor = RayTrace(", 0), "dump hook");
    }
    return src;
}

/* The legacy fragment-shader copy of the kernel (res/shaders/rt.frag): same kind of patch -- version
 * line, the two mix(vec3, vec3, bool) sites inside RayTrace, optionally the bounce literal (12 as shipped). */
static char *patch_reference_frag(char *src, int bounces)
{
    must(subst(&src, 0, "#version 460", "#version 450", 0), "version");
    char *rt = strstr(src, "vec3 RayTrace(");
    if (!rt) die("RayTrace not found");
    size_t rto = rt - src;
    must(subst(&src, rto, "\t\t\t\t\tisRefracted\n", "\t\t\t\t\tbvec3(isRefracted)\n", 0), "mix bool #1");
    must(subst(&src, rto, "isSpecularBounce), hitInfo.hitMaterial.baseColor.rgb, isRefracted)",
                          "bvec3(isSpecularBounce)), hitInfo.hitMaterial.baseColor.rgb, bvec3(isRefracted))", 0), "mix bool #2");
    if (bounces > 0) {
        char to[64]; snprintf(to, sizeof to, "debugNormal ? 1 : %d;", bounces);
        must(subst(&src, 0, "debugNormal ? 1 : 12;", to, 0), "bounce literal");
    }
    return src;
}

static GLuint compile(GLenum kind, const char *src, const char *what)
{
    GLuint sh = p_glCreateShader(kind);
    const GLchar *srcs[1] = { src };
    p_glShaderSource(sh, 1, srcs, NULL);
    p_glCompileShader(sh);
    GLint ok = 0; char log[8192];
    p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) { p_glGetShaderInfoLog(sh, sizeof log, NULL, log); fprintf(stderr, "%s: compile failed:\n%s\n", what, log); exit(3); }
    return sh;
}

/* ------------------------------------------------------------------- main */
struct scene_hdr { uint32_t magic, n_mesh_tris, n_nodes, n_mats, n_spheres, n_scene_tris; };
#define SCENE_MAGIC 0x31435350u /* "PSC1" */

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static GLuint make_ssbo(GLuint binding, const void *data, size_t bytes)
{
    GLuint b; p_glGenBuffers(1, &b);
    p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, b);
    p_glBufferData(GL_SHADER_STORAGE_BUFFER, bytes, bytes ? data : NULL, GL_DYNAMIC_DRAW);
    p_glBindBufferBase(GL_SHADER_STORAGE_BUFFER, binding, b);
    p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, 0);
    return b;
}

int main(int argc, char **argv)
{
    const char *refdir = NULL, *shader_file = NULL, *scene_file = NULL, *out_file = NULL, *dump_src = NULL;
    int W = 64, H = 40, passes = 1, first_pass = 1, bounces = 0, debug_normal = 0, dump = 0, dump_bounce = 0, every = 0, quiet = 0, frag = 0;
    const char *probe_from[8], *probe_to[8]; int n_probe = 0;
    float cam_pos[3] = { 0, 0, 0 };
    float inv_view[16] = { 0, 0, 1, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 0, 1 }; /* default camera, column-major */
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
#define NEXT() (i + 1 < argc ? argv[++i] : (die("missing value"), ""))
        if (!strcmp(a, "--reference-dir")) refdir = NEXT();
        else if (!strcmp(a, "--shader-file")) shader_file = NEXT();
        else if (!strcmp(a, "--scene")) scene_file = NEXT();
        else if (!strcmp(a, "--out")) out_file = NEXT();
        else if (!strcmp(a, "--dump-patched-source")) dump_src = NEXT(); /* debugging only; never commit the output */
        else if (!strcmp(a, "--width")) W = atoi(NEXT());
        else if (!strcmp(a, "--height")) H = atoi(NEXT());
        else if (!strcmp(a, "--passes")) passes = atoi(NEXT());
        else if (!strcmp(a, "--first-pass")) first_pass = atoi(NEXT());
        else if (!strcmp(a, "--bounces")) bounces = atoi(NEXT());
        else if (!strcmp(a, "--debug-normal")) debug_normal = atoi(NEXT());
        else if (!strcmp(a, "--dump")) dump = atoi(NEXT());          /* 0 radiance, 1 primary ids, 2 primary dirs, 3/4 per-bounce state */
        else if (!strcmp(a, "--dump-bounce")) dump_bounce = atoi(NEXT()); /* bounce index for --dump 3/4 */
        else if (!strcmp(a, "--dump-every-pass")) every = 1;         /* else only the final accumulation image */
        else if (!strcmp(a, "--quiet")) quiet = 1;
        else if (!strcmp(a, "--probe-subst")) {                      /* debugging only: one more textual substitution, after the patch */
            if (n_probe == 8) die("too many --probe-subst");
            probe_from[n_probe] = NEXT(); probe_to[n_probe++] = NEXT();
        }
        else if (!strcmp(a, "--frag")) frag = 1;                     /* run res/shaders/rt.frag (+ rt.vert) instead of rt.comp */
        else if (!strcmp(a, "--cam-pos")) { for (int k = 0; k < 3; k++) cam_pos[k] = strtof(NEXT(), NULL); }
        else if (!strcmp(a, "--inv-view")) { for (int k = 0; k < 16; k++) inv_view[k] = strtof(NEXT(), NULL); }
        else { fprintf(stderr, "unknown arg %s\n", a); return 2; }
    }
    if (!refdir && !shader_file) die("need --reference-dir or --shader-file");
    if (!out_file) die("need --out");

    make_context();
    if (!quiet) fprintf(stderr, "GL_VERSION %s | GL_RENDERER %s | LP_NUM_THREADS=%s\n", p_glGetString(GL_VERSION),
            p_glGetString(GL_RENDERER), getenv("LP_NUM_THREADS") ? getenv("LP_NUM_THREADS") : "(unset)");

    char *src;
    GLuint prog = p_glCreateProgram();
    GLint ok = 0; char log[8192];
    if (frag) {
        if (!refdir || dump) die("--frag needs --reference-dir and supports --dump 0 only");
        char path[4096];
        snprintf(path, sizeof path, "%s/res/shaders/rt.vert", refdir);
        char *vsrc = read_file(path, NULL);
        must(subst(&vsrc, 0, "#version 460", "#version 450", 0), "vert version");
        snprintf(path, sizeof path, "%s/res/shaders/rt.frag", refdir);
        src = patch_reference_frag(read_file(path, NULL), bounces);
        for (int k = 0; k < n_probe; k++) must(subst(&src, 0, probe_from[k], probe_to[k], 0), "--probe-subst");
        if (dump_src) { FILE *f = fopen(dump_src, "w"); fputs(src, f); fclose(f); }
        p_glAttachShader(prog, compile(GL_VERTEX_SHADER, vsrc, "rt.vert"));
        p_glAttachShader(prog, compile(GL_FRAGMENT_SHADER, src, "rt.frag"));
    } else {
        if (refdir) {
            char path[4096]; snprintf(path, sizeof path, "%s/res/shaders/rt.comp", refdir);
            src = patch_reference_shader(read_file(path, NULL), bounces, dump != 0);
            for (int k = 0; k < n_probe; k++) must(subst(&src, 0, probe_from[k], probe_to[k], 0), "--probe-subst");
        } else {
            src = read_file(shader_file, NULL);
        }
        if (dump_src) { FILE *f = fopen(dump_src, "w"); fputs(src, f); fclose(f); }
        p_glAttachShader(prog, compile(GL_COMPUTE_SHADER, src, "compute shader"));
    }
    p_glLinkProgram(prog);
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { p_glGetProgramInfoLog(prog, sizeof log, NULL, log); fprintf(stderr, "link failed:\n%s\n", log); return 3; }
    p_glUseProgram(prog);

    if (scene_file) {
        size_t len; char *blob = read_file(scene_file, &len);
        struct scene_hdr h; memcpy(&h, blob, sizeof h);
        if (h.magic != SCENE_MAGIC) die("bad scene blob magic");
        const char *p = blob + sizeof h;
        size_t need = sizeof h + 64ul * h.n_mesh_tris + 48ul * h.n_nodes + 80ul * h.n_mats + 32ul * h.n_spheres + 64ul * h.n_scene_tris;
        if (need != len) die("scene blob size mismatch");
        make_ssbo(1, p, 64ul * h.n_mesh_tris); p += 64ul * h.n_mesh_tris;   /* Triangle[]  src/Object.cpp:102-105 */
        make_ssbo(2, p, 48ul * h.n_nodes);     p += 48ul * h.n_nodes;       /* Node[]      src/Object.cpp:107-110 */
        make_ssbo(3, p, 80ul * h.n_mats);      p += 80ul * h.n_mats;        /* Material[]  src/Object.cpp:9-12   */
        make_ssbo(4, p, 32ul * h.n_spheres);   p += 32ul * h.n_spheres;     /* Sphere[]    src/Object.cpp:14-17  */
        make_ssbo(5, p, 64ul * h.n_scene_tris);                              /* Triangle[]  src/Object.cpp:19-22  */
        free(blob);
    }

    /* accumulation image: RGBA32F, zero-initialised (src/Renderer.cpp:117-128 leaves it undefined).  Fragment mode: two of
     * them, sampled / rendered to in turn (the shader reads texture(accumTexture) and writes fragColor, rt.frag:340-342). */
    GLuint tex[2]; p_glGenTextures(2, tex);
    float *zero = calloc((size_t)W * H * 4, sizeof(float));
    for (int k = 0; k < (frag ? 2 : 1); k++) {
        p_glBindTexture(GL_TEXTURE_2D, tex[k]);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, GL_RGBA, GL_FLOAT, zero);
    }
    GLuint fbo = 0;
    if (!frag) {
        p_glBindImageTexture(0, tex[0], 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
    } else {
        /* the fullscreen quad of src/Renderer.h:11-28 / src/Renderer.cpp:76-95 */
        static const float quad[20] = { -1, -1, 0,  1, -1, 0,  1, 1, 0,  -1, 1, 0,   0, 0,  1, 0,  1, 1,  0, 1 };
        static const unsigned idx[6] = { 0, 1, 2, 2, 3, 0 };
        GLuint vao, vbo, ebo;
        p_glGenVertexArrays(1, &vao); p_glBindVertexArray(vao);
        p_glGenBuffers(1, &vbo); p_glBindBuffer(GL_ARRAY_BUFFER, vbo);
        p_glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
        p_glGenBuffers(1, &ebo); p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
        p_glBufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof idx, idx, GL_STATIC_DRAW);
        p_glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 0, (void *)0);
        p_glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 0, (void *)(12 * sizeof(float)));
        p_glEnableVertexAttribArray(0); p_glEnableVertexAttribArray(1);
        p_glGenFramebuffers(1, &fbo);
        p_glViewport(0, 0, W, H);
        p_glDisable(GL_DEPTH_TEST); p_glDisable(GL_BLEND); p_glDisable(GL_DITHER);
        GLint u_tex = p_glGetUniformLocation(prog, "accumTexture");
        if (u_tex >= 0) p_glUniform1i(u_tex, 0);
        p_glActiveTexture(GL_TEXTURE0);
    }
    GLuint tex_result = tex[0];

    GLint u_pass = p_glGetUniformLocation(prog, "currAccumPass");
    GLint u_dbg = p_glGetUniformLocation(prog, "debugNormal");
    GLint u_pos = p_glGetUniformLocation(prog, "cam.position");
    GLint u_inv = p_glGetUniformLocation(prog, "cam.inverseView");
    GLint u_dump = p_glGetUniformLocation(prog, "ptDump");
    if (u_dbg >= 0) p_glUniform1i(u_dbg, debug_normal);
    if (u_pos >= 0) p_glUniform3f(u_pos, cam_pos[0], cam_pos[1], cam_pos[2]);
    if (u_inv >= 0) p_glUniformMatrix4fv(u_inv, 1, GL_FALSE, inv_view);      /* src/Shader.cpp:72-76 */
    if (u_dump >= 0) p_glUniform1i(u_dump, dump);
    GLint u_bounce = p_glGetUniformLocation(prog, "ptBounce");
    if (u_bounce >= 0) p_glUniform1i(u_bounce, dump_bounce);

    FILE *out = fopen(out_file, "wb");
    if (!out) die("cannot open --out");
    float *img = malloc((size_t)W * H * 4 * sizeof(float));
    double t_first = 0, t_rest = 0;
    for (int k = 0; k < passes; k++) {
        int pass = first_pass + k;
        double t0 = now_s();
        if (u_pass >= 0) p_glUniform1i(u_pass, pass);
        if (!frag) {
            p_glDispatchCompute(W / 8, H / 8, 1);                            /* src/Main.cpp:148 */
            p_glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
        } else {
            const int src_t = k & 1, dst_t = src_t ^ 1;
            p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
            p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex[dst_t], 0);
            if (k == 0 && p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("framebuffer incomplete");
            p_glBindTexture(GL_TEXTURE_2D, tex[src_t]);
            p_glDrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, (void *)0);
            p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
            tex_result = tex[dst_t];
        }
        p_glFinish();
        double dt = now_s() - t0;
        if (k == 0) t_first = dt; else t_rest += dt;
        if (every || k == passes - 1) {
            p_glBindTexture(GL_TEXTURE_2D, tex_result);
            p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, img);
            fwrite(img, sizeof(float), (size_t)W * H * 4, out);
        }
    }
    fclose(out);
    GLenum e = p_glGetError();
    if (e) fprintf(stderr, "GL error 0x%x\n", e);
    /* machine-readable timing line on stdout */
    printf("{\"width\": %d, \"height\": %d, \"passes\": %d, \"first_pass_s\": %.6f, \"steady_s\": %.6f, "
           "\"steady_samples_per_s\": %.1f}\n", W, H, passes, t_first, t_rest,
           passes > 1 && t_rest > 0 ? (double)W * H * (passes - 1) / t_rest : 0.0);
    return e ? 4 : 0;
}
