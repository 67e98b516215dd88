/*
 * TEST INFRASTRUCTURE (oracle side) -- not part of the shipped product.
 *
 * Headless runner for the reference GLSL shaders on Mesa llvmpipe.  It opens a GL 4.5 core
 * context by loading swrast_dri.so through the DRI swrast interface (no X server, no EGL, no
 * OSMesa), compiles a vertex + fragment shader pair that it READS FROM DISK AT RUN TIME
 * (by default the reference's own files where they lie:
 *   /root/reference/app/assets/ray_tracer.vert and ray_tracer.frag),
 * binds the four SSBOs, the cubemap and the uniforms exactly as the reference's render pass does
 * (/root/reference/app/core/src/App.cpp:96-140, Framebuffer.cpp:6-12, Skybox.cpp:46-57,
 * Buffer.cpp:3-14), draws the fullscreen triangle and dumps the RGBA32F colour attachment and the
 * 24-bit depth attachment.
 *
 * Optionally (--overlay ...) it then draws the pass that follows in the reference's frame (App.cpp:142-166): indexed meshes with the
 * reference's Vert_PosNormTex.vert / Frag_LightingSimple.frag (read from disk at run time as well), blending off, depth test
 * GL_LEQUAL against what the ray-tracing draw left in the depth attachment, one draw per instance matrix, a mip-mapped RGBA8
 * texture set up as GLUtils.cpp:193-224 does.  --no-rt skips the ray-tracing draw (colour 0, depth cleared to 1).
 *
 * --present also performs the blit onto the window's framebuffer (Framebuffer.cpp:33-35) and dumps its 8-bit pixels.
 *
 * No reference source text is contained in this file; the shader text never enters the repo.
 * The binary is written to oracle/_ref/ (git-ignored).  It only works where /root/reference
 * exists (this container), so it is used to generate the data fixtures under tests/golden/ and to
 * validate the CPU restatement in oracle/rt_oracle.c -- never on the GPU box.
 *
 * Uniform file layout (70 little-endian 32-bit words, see oracle/refrun.py):
 *   [0..15]  inv_view_proj_mat (column-major)   [16..18] camera_position
 *   [19]     max_recursion_limit (uint32)       [20] time   [21] blur_amount
 *   [22..24] portal_position_1  [25..27] portal_direction_1
 *   [28..30] portal_position_2  [31..33] portal_direction_2
 *   [34] portal_width [35] portal_height
 *   [36..51] portal_1_to_2  [52..67] portal_2_to_1   [68] z_near [69] z_far
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

static int g_w = 64, g_h = 64;

static void getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) { (void)d; (void)p; *x = 0; *y = 0; *w = g_w; *h = g_h; }
static void putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }

static const __DRIswrastLoaderExtension swrast_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = getDrawableInfo, .putImage = putImage, .getImage = getImage,
    .putImage2 = putImage2, .getImage2 = getImage2,
};
static const __DRIextension *loader_exts[] = {&swrast_loader.base, NULL};

typedef void *(*getproc_t)(const char *);
static getproc_t gp;
#define GLF(type, name) type name = (type)gp(#name); if (!name) { fprintf(stderr, "missing GL entry point %s\n", #name); return 2; }

static void *slurp(const char *path, size_t *n_out) {
    FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(3); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *s = malloc((size_t)n + 1); if (fread(s, 1, (size_t)n, f) != (size_t)n) { perror(path); exit(3); }
    s[n] = 0; fclose(f); if (n_out) *n_out = (size_t)n; return s;
}
static void dump(const char *prefix, const char *suffix, const void *d, size_t n) {
    char p[4096]; snprintf(p, sizeof p, "%s%s", prefix, suffix);
    FILE *f = fopen(p, "wb"); if (!f) { perror(p); exit(3); }
    fwrite(d, 1, n, f); fclose(f);
}
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; }

int main(int argc, char **argv) {
    const char *vert = "/root/reference/app/assets/ray_tracer.vert";
    const char *frag = "/root/reference/app/assets/ray_tracer.frag";
    const char *ssbo[4] = {0, 0, 0, 0}, *uniforms = 0, *bounds = 0, *sky = 0, *times = 0, *out = "refout";
    int sky_w = 0, sky_h = 0, blend = 1, bench = 0, quiet = 0, no_rt = 0, present = 0;
    const char *ov_vert = "/root/reference/app/assets/Vert_PosNormTex.vert", *ov_frag = "/root/reference/app/assets/Frag_LightingSimple.frag";
    const char *ov_vbo = 0, *ov_ibo = 0, *ov_uniforms = 0, *ov_tex = 0; int ov_tex_w = 0, ov_tex_h = 0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
#define NEXT (i + 1 < argc ? argv[++i] : (fprintf(stderr, "missing value for %s\n", a), exit(2), ""))
        if (!strcmp(a, "--w")) g_w = atoi(NEXT); else if (!strcmp(a, "--h")) g_h = atoi(NEXT);
        else if (!strcmp(a, "--vert")) vert = NEXT; else if (!strcmp(a, "--frag")) frag = NEXT;
        else if (!strcmp(a, "--ssbo0")) ssbo[0] = NEXT; else if (!strcmp(a, "--ssbo1")) ssbo[1] = NEXT;
        else if (!strcmp(a, "--ssbo2")) ssbo[2] = NEXT; else if (!strcmp(a, "--ssbo3")) ssbo[3] = NEXT;
        else if (!strcmp(a, "--uniforms")) uniforms = NEXT; else if (!strcmp(a, "--bounds")) bounds = NEXT;
        else if (!strcmp(a, "--sky")) { sky = NEXT; sky_w = atoi(NEXT); sky_h = atoi(NEXT); }
        else if (!strcmp(a, "--times")) times = NEXT; else if (!strcmp(a, "--out")) out = NEXT;
        else if (!strcmp(a, "--noblend")) blend = 0; else if (!strcmp(a, "--bench")) bench = atoi(NEXT);
        else if (!strcmp(a, "--quiet")) quiet = 1;
        else if (!strcmp(a, "--overlay")) { ov_vbo = NEXT; ov_ibo = NEXT; ov_uniforms = NEXT; }
        else if (!strcmp(a, "--overlay-tex")) { ov_tex = NEXT; ov_tex_w = atoi(NEXT); ov_tex_h = atoi(NEXT); }
        else if (!strcmp(a, "--overlay-vert")) ov_vert = NEXT; else if (!strcmp(a, "--overlay-frag")) ov_frag = NEXT;
        else if (!strcmp(a, "--no-rt")) no_rt = 1;
        else if (!strcmp(a, "--present")) present = 1;
        else { fprintf(stderr, "unknown option %s\n", a); return 2; }
    }
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) { fprintf(stderr, "glapi: %s\n", dlerror()); return 1; }
    void *drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) { fprintf(stderr, "drv: %s\n", dlerror()); return 1; }
    gp = (getproc_t)dlsym(glapi, "_glapi_get_proc_address");
    const __DRIextension **(*getext)(void) = dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!gp || !getext) { fprintf(stderr, "DRI symbols missing\n"); return 1; }
    const __DRIextension **drv_exts = getext();
    const __DRIcoreExtension *core = NULL; const __DRIswrastExtension *swrast = NULL;
    for (int i = 0; drv_exts[i]; i++) {
        if (!strcmp(drv_exts[i]->name, __DRI_CORE)) core = (const void *)drv_exts[i];
        if (!strcmp(drv_exts[i]->name, __DRI_SWRAST)) swrast = (const void *)drv_exts[i];
    }
    if (!core || !swrast) { fprintf(stderr, "no DRI core/swrast extension\n"); return 1; }
    const __DRIconfig **configs = NULL;
    __DRIscreen *screen = swrast->createNewScreen2(0, loader_exts, drv_exts, &configs, NULL);
    if (!screen) { fprintf(stderr, "createNewScreen2 failed\n"); return 1; }
    const __DRIconfig *cfg = NULL;
    for (int i = 0; configs[i]; i++) {
        unsigned r = 0, al = 0, z = 0, db = 0;
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &al);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DEPTH_SIZE, &z);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        if (r == 8 && al == 8 && z == 24 && !db) { cfg = configs[i]; break; }
    }
    if (!cfg) cfg = configs[0];
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5};
    unsigned err = 0;
    __DRIcontext *ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!ctx) { fprintf(stderr, "createContextAttribs failed err=%u\n", err); return 1; }
    __DRIdrawable *draw = swrast->createNewDrawable(screen, cfg, NULL);
    if (!draw || !core->bindContext(ctx, draw, draw)) { fprintf(stderr, "drawable/bind failed\n"); return 1; }

    const GLubyte *(*pglGetString)(GLenum) = gp("glGetString");
    if (!quiet) fprintf(stderr, "GL_VERSION %s | GL_RENDERER %s\n", pglGetString(GL_VERSION), pglGetString(GL_RENDERER));

    GLF(PFNGLCREATESHADERPROC, glCreateShader); GLF(PFNGLSHADERSOURCEPROC, glShaderSource);
    GLF(PFNGLCOMPILESHADERPROC, glCompileShader); GLF(PFNGLGETSHADERIVPROC, glGetShaderiv);
    GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog); GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram);
    GLF(PFNGLATTACHSHADERPROC, glAttachShader); GLF(PFNGLLINKPROGRAMPROC, glLinkProgram);
    GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv); GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog);
    GLF(PFNGLUSEPROGRAMPROC, glUseProgram); GLF(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
    GLF(PFNGLUNIFORM1FPROC, glUniform1f); GLF(PFNGLUNIFORM1IPROC, glUniform1i); GLF(PFNGLUNIFORM1UIPROC, glUniform1ui);
    GLF(PFNGLUNIFORM3FVPROC, glUniform3fv); GLF(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv);
    GLF(PFNGLCREATEBUFFERSPROC, glCreateBuffers); GLF(PFNGLNAMEDBUFFERSTORAGEPROC, glNamedBufferStorage);
    GLF(PFNGLBINDBUFFERBASEPROC, glBindBufferBase); GLF(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays);
    GLF(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray); GLF(PFNGLCREATETEXTURESPROC, glCreateTextures);
    GLF(PFNGLTEXTURESTORAGE2DPROC, glTextureStorage2D); GLF(PFNGLTEXTURESUBIMAGE3DPROC, glTextureSubImage3D);
    GLF(PFNGLTEXTUREPARAMETERIPROC, glTextureParameteri); GLF(PFNGLCREATEFRAMEBUFFERSPROC, glCreateFramebuffers);
    GLF(PFNGLNAMEDFRAMEBUFFERTEXTUREPROC, glNamedFramebufferTexture); GLF(PFNGLCREATERENDERBUFFERSPROC, glCreateRenderbuffers);
    GLF(PFNGLNAMEDRENDERBUFFERSTORAGEPROC, glNamedRenderbufferStorage);
    GLF(PFNGLNAMEDFRAMEBUFFERRENDERBUFFERPROC, glNamedFramebufferRenderbuffer);
    GLF(PFNGLCHECKNAMEDFRAMEBUFFERSTATUSPROC, glCheckNamedFramebufferStatus); GLF(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer);
    GLF(PFNGLACTIVETEXTUREPROC, glActiveTexture); GLF(PFNGLBLENDCOLORPROC, glBlendColor);
    GLF(PFNGLCLEARNAMEDFRAMEBUFFERFVPROC, glClearNamedFramebufferfv);
    void (*pglViewport)(GLint, GLint, GLsizei, GLsizei) = gp("glViewport");
    void (*pglEnable)(GLenum) = gp("glEnable"); void (*pglDisable)(GLenum) = gp("glDisable");
    void (*pglClear)(GLbitfield) = gp("glClear"); void (*pglDepthFunc)(GLenum) = gp("glDepthFunc");
    void (*pglBlendFunc)(GLenum, GLenum) = gp("glBlendFunc"); void (*pglBindTexture)(GLenum, GLuint) = gp("glBindTexture");
    void (*pglDrawArrays)(GLenum, GLint, GLsizei) = gp("glDrawArrays"); void (*pglFinish)(void) = gp("glFinish");
    void (*pglReadPixels)(GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *) = gp("glReadPixels");
    void (*pglCullFace)(GLenum) = gp("glCullFace");
    GLenum (*pglGetError)(void) = gp("glGetError");
    GLF(PFNGLGENBUFFERSPROC, glGenBuffers); GLF(PFNGLBINDBUFFERPROC, glBindBuffer); GLF(PFNGLBUFFERDATAPROC, glBufferData);
    GLF(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray); GLF(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer);
    GLF(PFNGLGENERATEMIPMAPPROC, glGenerateMipmap);
    void (*pglGenTextures)(GLsizei, GLuint *) = gp("glGenTextures");
    void (*pglTexImage2D)(GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *) = gp("glTexImage2D");
    void (*pglTexParameteri)(GLenum, GLenum, GLint) = gp("glTexParameteri");
    void (*pglDrawElements)(GLenum, GLsizei, GLenum, const void *) = gp("glDrawElements");
    void (*pglClearDepth)(GLdouble) = gp("glClearDepth");

    /* the overlay's program, mesh and texture */
    GLuint ov_prog = 0, ov_vao = 0, ov_texid = 0; GLsizei ov_count = 0; float *OU = NULL; size_t ov_n_inst = 0;
    if (ov_vbo) {
        const char *vsrc2 = slurp(ov_vert, NULL), *fsrc2 = slurp(ov_frag, NULL);
        GLuint vs2 = glCreateShader(GL_VERTEX_SHADER), fs2 = glCreateShader(GL_FRAGMENT_SHADER);
        glShaderSource(vs2, 1, &vsrc2, NULL); glShaderSource(fs2, 1, &fsrc2, NULL); glCompileShader(vs2); glCompileShader(fs2);
        GLint ok2 = 0; static char log2[16384];
        glGetShaderiv(vs2, GL_COMPILE_STATUS, &ok2); if (!ok2) { glGetShaderInfoLog(vs2, sizeof log2, NULL, log2); fprintf(stderr, "overlay VS compile failed: %s\n", log2); return 4; }
        glGetShaderiv(fs2, GL_COMPILE_STATUS, &ok2); if (!ok2) { glGetShaderInfoLog(fs2, sizeof log2, NULL, log2); fprintf(stderr, "overlay FS compile failed: %s\n", log2); return 4; }
        ov_prog = glCreateProgram(); glAttachShader(ov_prog, vs2); glAttachShader(ov_prog, fs2); glLinkProgram(ov_prog);
        glGetProgramiv(ov_prog, GL_LINK_STATUS, &ok2); if (!ok2) { glGetProgramInfoLog(ov_prog, sizeof log2, NULL, log2); fprintf(stderr, "overlay link failed: %s\n", log2); return 4; }
        size_t nv, ni, nu; void *vd = slurp(ov_vbo, &nv), *id = slurp(ov_ibo, &ni); OU = slurp(ov_uniforms, &nu);
        if (nu < 26 * 4 || (nu - 26 * 4) % (32 * 4)) { fprintf(stderr, "overlay uniform file: 26 words + 32 per instance\n"); return 6; }
        ov_n_inst = (nu - 26 * 4) / (32 * 4); ov_count = (GLsizei)(ni / 4);
        GLuint vbo, ibo; glGenVertexArrays(1, &ov_vao); glBindVertexArray(ov_vao);
        glGenBuffers(1, &vbo); glBindBuffer(GL_ARRAY_BUFFER, vbo); glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)nv, vd, GL_STATIC_DRAW);
        glGenBuffers(1, &ibo); glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ibo); glBufferData(GL_ELEMENT_ARRAY_BUFFER, (GLsizeiptr)ni, id, GL_STATIC_DRAW);
        /* App.cpp:65-69: position, normal, texcoord of the 32-byte Vertex (GLUtils.hpp:29-34) */
        glEnableVertexAttribArray(0); glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 32, (const void *)0);
        glEnableVertexAttribArray(1); glVertexAttribPointer(1, 3, GL_FLOAT, GL_FALSE, 32, (const void *)12);
        glEnableVertexAttribArray(2); glVertexAttribPointer(2, 2, GL_FLOAT, GL_FALSE, 32, (const void *)24);
        glBindVertexArray(0);
        if (ov_tex) {
            size_t nt2; void *td = slurp(ov_tex, &nt2);
            if (nt2 != (size_t)ov_tex_w * ov_tex_h * 4) { fprintf(stderr, "overlay texture file size mismatch\n"); return 6; }
            pglGenTextures(1, &ov_texid); pglBindTexture(GL_TEXTURE_2D, ov_texid);
            pglTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, ov_tex_w, ov_tex_h, 0, GL_RGBA, GL_UNSIGNED_BYTE, td);      /* GLUtils.cpp:193-202 */
            glGenerateMipmap(GL_TEXTURE_2D);                                                                    /* GLUtils.cpp:215 */
            pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR); pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR_MIPMAP_LINEAR);
            pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT); pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
            pglBindTexture(GL_TEXTURE_2D, 0);
        }
    }

    const char *vsrc = slurp(vert, NULL), *fsrc = slurp(frag, NULL);
    GLuint vs = glCreateShader(GL_VERTEX_SHADER), fs = glCreateShader(GL_FRAGMENT_SHADER);
    glShaderSource(vs, 1, &vsrc, NULL); glShaderSource(fs, 1, &fsrc, NULL);
    glCompileShader(vs); glCompileShader(fs);
    GLint ok = 0; static char log[16384];
    glGetShaderiv(vs, GL_COMPILE_STATUS, &ok); if (!ok) { glGetShaderInfoLog(vs, sizeof log, NULL, log); fprintf(stderr, "VS compile failed: %s\n", log); return 4; }
    glGetShaderiv(fs, GL_COMPILE_STATUS, &ok); if (!ok) { glGetShaderInfoLog(fs, sizeof log, NULL, log); fprintf(stderr, "FS compile failed: %s\n", log); return 4; }
    GLuint prog = glCreateProgram(); glAttachShader(prog, vs); glAttachShader(prog, fs); glLinkProgram(prog);
    glGetProgramiv(prog, GL_LINK_STATUS, &ok); if (!ok) { glGetProgramInfoLog(prog, sizeof log, NULL, log); fprintf(stderr, "link failed: %s\n", log); return 4; }

    /* FBO: RGBA32F colour texture + DEPTH24 renderbuffer (Framebuffer.cpp:6-12). */
    GLuint tex, fbo, rbo;
    glCreateTextures(GL_TEXTURE_2D, 1, &tex); glTextureStorage2D(tex, 1, GL_RGBA32F, g_w, g_h);
    glCreateFramebuffers(1, &fbo); glNamedFramebufferTexture(fbo, GL_COLOR_ATTACHMENT0, tex, 0);
    glCreateRenderbuffers(1, &rbo); glNamedRenderbufferStorage(rbo, GL_DEPTH_COMPONENT24, g_w, g_h);
    glNamedFramebufferRenderbuffer(fbo, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, rbo);
    if (glCheckNamedFramebufferStatus(fbo, GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { fprintf(stderr, "FBO incomplete\n"); return 5; }
    { const float zero[4] = {0, 0, 0, 0}; glClearNamedFramebufferfv(fbo, GL_COLOR, 0, zero); }

    GLuint b[4]; glCreateBuffers(4, b);
    for (int i = 0; i < 4; i++) if (ssbo[i]) {
        size_t n; void *d = slurp(ssbo[i], &n);
        static const uint32_t pad[4] = {0, 0, 0, 0};
        glNamedBufferStorage(b[i], n ? (GLsizeiptr)n : 16, n ? d : (const void *)pad, 0);   /* Buffer.cpp:5 */
        glBindBufferBase(GL_SHADER_STORAGE_BUFFER, i, b[i]);                                /* Buffer.cpp:13 */
        free(d);
    }
    GLuint cube = 0;
    if (sky) {
        size_t n; uint8_t *d = slurp(sky, &n);
        if (n != (size_t)sky_w * sky_h * 4 * 6) { fprintf(stderr, "sky file size mismatch\n"); return 6; }
        glCreateTextures(GL_TEXTURE_CUBE_MAP, 1, &cube); glTextureStorage2D(cube, 1, GL_RGBA8, sky_w, sky_h);   /* Skybox.cpp:46 */
        for (int f = 0; f < 6; f++) glTextureSubImage3D(cube, 0, 0, 0, f, sky_w, sky_h, 1, GL_RGBA, GL_UNSIGNED_BYTE, d + (size_t)f * sky_w * sky_h * 4);
        glTextureParameteri(cube, GL_TEXTURE_MIN_FILTER, GL_LINEAR); glTextureParameteri(cube, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
        glTextureParameteri(cube, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE); glTextureParameteri(cube, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
        glTextureParameteri(cube, GL_TEXTURE_WRAP_R, GL_CLAMP_TO_EDGE);
        free(d);
    }

    GLuint vao; glGenVertexArrays(1, &vao);
    glBindFramebuffer(GL_FRAMEBUFFER, fbo); pglViewport(0, 0, g_w, g_h);
    /* GL state of App.cpp:46-56 */
    pglEnable(GL_CULL_FACE); pglCullFace(GL_BACK);
    pglEnable(GL_DEPTH_TEST); pglDepthFunc(GL_LEQUAL);
    if (blend) { pglEnable(GL_BLEND); pglBlendFunc(GL_CONSTANT_ALPHA, GL_ONE_MINUS_CONSTANT_ALPHA); } else pglDisable(GL_BLEND);
    glUseProgram(prog); glBindVertexArray(vao);
    if (cube) { glActiveTexture(GL_TEXTURE0); pglBindTexture(GL_TEXTURE_CUBE_MAP, cube); }
#define UL(n) glGetUniformLocation(prog, n)
    glUniform1i(UL("skyboxTexture"), 0);
    float U[70]; memset(U, 0, sizeof U);
    if (uniforms) { size_t n; void *d = slurp(uniforms, &n); if (n != sizeof U) { fprintf(stderr, "uniform file must be 280 bytes\n"); return 6; } memcpy(U, d, sizeof U); free(d); }
    float B[6] = {0, 0, 0, 0, 0, 0};
    if (bounds) { size_t n; void *d = slurp(bounds, &n); if (n != sizeof B) { fprintf(stderr, "bounds file must be 24 bytes\n"); return 6; } memcpy(B, d, sizeof B); free(d); }
    uint32_t maxrec; memcpy(&maxrec, &U[19], 4);
    glUniformMatrix4fv(UL("inv_view_proj_mat"), 1, GL_FALSE, U);
    glUniform3fv(UL("camera_position"), 1, U + 16);
    glUniform1f(UL("width"), (float)g_w); glUniform1f(UL("height"), (float)g_h);
    glUniform3fv(UL("octree_min_bounds"), 1, B); glUniform3fv(UL("octree_max_bounds"), 1, B + 3);
    glUniform1ui(UL("max_recursion_limit"), maxrec);
    glUniform1f(UL("time"), U[20]); glUniform1f(UL("blur_amount"), U[21]);
    glUniform3fv(UL("portal_position_1"), 1, U + 22); glUniform3fv(UL("portal_direction_1"), 1, U + 25);
    glUniform3fv(UL("portal_position_2"), 1, U + 28); glUniform3fv(UL("portal_direction_2"), 1, U + 31);
    glUniform1f(UL("portal_width"), U[34]); glUniform1f(UL("portal_height"), U[35]);
    glUniformMatrix4fv(UL("portal_1_to_2"), 1, GL_FALSE, U + 36); glUniformMatrix4fv(UL("portal_2_to_1"), 1, GL_FALSE, U + 52);
    glUniform1f(UL("z_near"), U[68]); glUniform1f(UL("z_far"), U[69]);

    /* Frames: one per entry of the times file (default: a single frame at U[20]); frame k is blended
       with constant alpha 1/(k+1) as App.cpp:104-105 does with m_still_frame_counter. */
    size_t nt = 1; float *T = NULL;
    if (times) { size_t n; T = slurp(times, &n); nt = n / 4; }
    for (size_t k = 0; k < nt; k++) {
        pglClear(GL_DEPTH_BUFFER_BIT);                                        /* App.cpp:98 */
        if (blend) glBlendColor(0.0f, 0.0f, 0.0f, 1.0f / (float)(k + 1));     /* App.cpp:104 */
        if (T) glUniform1f(UL("time"), T[k]);
        if (!no_rt) pglDrawArrays(GL_TRIANGLES, 0, 3);                        /* App.cpp:140 */
        if (ov_prog) {                                                        /* App.cpp:142-166 */
            pglDisable(GL_BLEND); pglEnable(GL_DEPTH_TEST);
            glUseProgram(ov_prog); glBindVertexArray(ov_vao);
            glActiveTexture(GL_TEXTURE0); pglBindTexture(GL_TEXTURE_2D, ov_texid);
            glUniform1i(glGetUniformLocation(ov_prog, "texImage"), 0);
            glUniformMatrix4fv(glGetUniformLocation(ov_prog, "viewProj"), 1, GL_FALSE, OU);
            glUniform3fv(glGetUniformLocation(ov_prog, "light_dir"), 1, OU + 16);
            glUniform3fv(glGetUniformLocation(ov_prog, "La"), 1, OU + 19);
            glUniform3fv(glGetUniformLocation(ov_prog, "Ld"), 1, OU + 22);
            for (size_t q = 0; q < ov_n_inst; q++) {
                glUniformMatrix4fv(glGetUniformLocation(ov_prog, "world"), 1, GL_FALSE, OU + 26 + 32 * q);
                glUniformMatrix4fv(glGetUniformLocation(ov_prog, "worldIT"), 1, GL_FALSE, OU + 26 + 32 * q + 16);
                pglDrawElements(GL_TRIANGLES, ov_count, GL_UNSIGNED_INT, 0);
            }
            glUseProgram(prog); glBindVertexArray(vao);
            if (cube) { glActiveTexture(GL_TEXTURE0); pglBindTexture(GL_TEXTURE_CUBE_MAP, cube); }
            if (blend) pglEnable(GL_BLEND);
        }
        pglFinish();
    }
    float *px = malloc(sizeof(float) * 4 * (size_t)g_w * g_h), *dz = malloc(sizeof(float) * (size_t)g_w * g_h);
    pglReadPixels(0, 0, g_w, g_h, GL_RGBA, GL_FLOAT, px);
    pglReadPixels(0, 0, g_w, g_h, GL_DEPTH_COMPONENT, GL_FLOAT, dz);
    GLenum e = pglGetError(); if (e) { fprintf(stderr, "glGetError=0x%x\n", e); return 7; }
    dump(out, ".rgba.f32", px, sizeof(float) * 4 * (size_t)g_w * g_h);
    dump(out, ".depth.f32", dz, sizeof(float) * (size_t)g_w * g_h);
    if (present) {      /* Framebuffer::Blit (Framebuffer.cpp:33-35): the colour attachment onto the window's 8-bit framebuffer */
        GLF(PFNGLBLITNAMEDFRAMEBUFFERPROC, glBlitNamedFramebuffer);
        glBlitNamedFramebuffer(fbo, 0, 0, 0, g_w, g_h, 0, 0, g_w, g_h, GL_COLOR_BUFFER_BIT, GL_LINEAR);
        glBindFramebuffer(GL_FRAMEBUFFER, 0);
        pglFinish();
        uint8_t *p8 = malloc((size_t)g_w * g_h * 4);
        pglReadPixels(0, 0, g_w, g_h, GL_RGBA, GL_UNSIGNED_BYTE, p8);
        GLenum e2 = pglGetError(); if (e2) { fprintf(stderr, "present: glGetError=0x%x\n", e2); return 7; }
        dump(out, ".present.u8", p8, (size_t)g_w * g_h * 4);
        glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    }

    if (bench > 0) {   /* steady-state timing of the last frame's state (CPU baseline numbers) */
        double best = 1e30, sum = 0;
        for (int k = 0; k < bench; k++) {
            double t0 = now_ms(); pglClear(GL_DEPTH_BUFFER_BIT); pglDrawArrays(GL_TRIANGLES, 0, 3); pglFinish(); double t1 = now_ms();
            if (t1 - t0 < best) best = t1 - t0; sum += t1 - t0;
        }
        printf("{\"w\": %d, \"h\": %d, \"frames\": %d, \"ms_best\": %.3f, \"ms_mean\": %.3f}\n", g_w, g_h, bench, best, sum / bench);
    }
    return 0;
}
