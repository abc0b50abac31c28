// refgl -- executes the reference's GLSL fragment shaders VERBATIM on Mesa
// llvmpipe through a headless GL 4.3 core context.
//
// TEST INFRASTRUCTURE ONLY (oracle side): builds into oracle/_ref/librefgl.so,
// is used to generate tests/golden/* and as the "reference" cpu_baseline.
// Nothing in the product path may link or call this.
//
// What is the reference's and what is restated here:
//   * reference, used as-is: the GLSL text (raw string literals of
//     src/filter/*.cpp and src/Shader.h) composed by the reference's own
//     Shader::builder / Shader::getCode (src/Shader.cpp is compiled in from
//     /root/reference; the builder expressions are lifted as text at build
//     time by extract_shaders.py into oracle/_ref/gen_shaders.cpp).
//   * restated (this file): Texture (src/Texture.cpp:21-42,56-67,81-93),
//     the fullscreen quad (src/Project.cpp:125-141, src/Vbo.cpp:8-40),
//     Project::apply (src/Project.cpp:284-292) and setCanvasUniforms
//     (src/Project.cpp:456-461).  Sampler binding is by explicit
//     (sampler name -> texture) pairs instead of the reference's
//     unordered_set walk (src/Project.cpp:294-309), which pins the reference's
//     same-named-sampler ambiguity to
//     "read = current field, write = scratch".
#include "dri_headless.h"
#include <GL/gl.h>
#include <GL/glext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>
#include <chrono>

#include "Shader.h"   // reference: /root/reference/src/Shader.h (via -I)

void refgl_register_shaders(std::map<std::string, Shader *> &reg);  // generated

#define GLF(ret, name, ...) typedef ret (*PFN_##name)(__VA_ARGS__); static PFN_##name p_##name;
GLF(void, glGenTextures, GLsizei, GLuint *)
GLF(void, glDeleteTextures, GLsizei, const GLuint *)
GLF(void, glBindTexture, GLenum, GLuint)
GLF(void, glTexParameteri, GLenum, GLenum, GLint)
GLF(void, glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *)
GLF(void, glTexSubImage2D, GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void *)
GLF(void, glGetTexImage, GLenum, GLint, GLenum, GLenum, void *)
GLF(void, glActiveTexture, GLenum)
GLF(void, glGenFramebuffers, GLsizei, GLuint *)
GLF(void, glBindFramebuffer, GLenum, GLuint)
GLF(void, glFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint)
GLF(void, glViewport, GLint, GLint, GLsizei, GLsizei)
GLF(GLuint, glCreateShader, GLenum)
GLF(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *)
GLF(void, glCompileShader, GLuint)
GLF(void, glGetShaderiv, GLuint, GLenum, GLint *)
GLF(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *)
GLF(GLuint, glCreateProgram, void)
GLF(void, glAttachShader, GLuint, GLuint)
GLF(void, glLinkProgram, GLuint)
GLF(void, glGetProgramiv, GLuint, GLenum, GLint *)
GLF(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *)
GLF(void, glUseProgram, GLuint)
GLF(GLint, glGetUniformLocation, GLuint, const GLchar *)
GLF(void, glUniform1i, GLint, GLint)
GLF(void, glUniform1f, GLint, GLfloat)
GLF(void, glUniform2f, GLint, GLfloat, GLfloat)
GLF(void, glUniform1fv, GLint, GLsizei, const GLfloat *)
GLF(void, glUniform3f, GLint, GLfloat, GLfloat, GLfloat)
GLF(void, glUniformMatrix3fv, GLint, GLsizei, GLboolean, const GLfloat *)
GLF(void, glGenVertexArrays, GLsizei, GLuint *)
GLF(void, glBindVertexArray, GLuint)
GLF(void, glGenBuffers, GLsizei, GLuint *)
GLF(void, glBindBuffer, GLenum, GLuint)
GLF(void, glBufferData, GLenum, GLsizeiptr, const void *, GLenum)
GLF(void, glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void *)
GLF(void, glEnableVertexAttribArray, GLuint)
GLF(void, glDrawElements, GLenum, GLsizei, GLenum, const void *)
GLF(void, glFinish, void)
GLF(GLenum, glGetError, void)
GLF(const GLubyte *, glGetString, GLenum)
GLF(void, glPixelStorei, GLenum, GLint)

namespace {
struct Tex { GLuint id; int w, h; GLenum internal; int ncomp; };
struct Prog { GLuint id; };
std::vector<Tex> texs;
std::vector<Prog> progs;
std::map<std::string, Shader *> registry;
GLuint fbo, vao;
float canvas_coords[4] = { -1.5707963267948966f, 1.5707963267948966f, -3.14159265358979f, 3.14159265358979f };
float canvas_circ = 42000.0f;
std::string last_err;
bool inited = false;

GLuint compile(GLenum type, const std::string &src)
{
    GLuint s = p_glCreateShader(type);
    const char *c = src.c_str();
    p_glShaderSource(s, 1, &c, nullptr);
    p_glCompileShader(s);
    GLint ok = 0;
    p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[4096]; GLsizei n = 0;
        p_glGetShaderInfoLog(s, sizeof log, &n, log);
        last_err = std::string("shader compile failed: ") + log;
        return 0;
    }
    return s;
}
}  // namespace

extern "C" {

const char *refgl_last_error(void) { return last_err.c_str(); }

int refgl_init(void)
{
    if (inited) return 0;
    // legacy texture2D() under "#version 430" (BlurMenu.cpp:54-60, cpufilter.cpp:52...)
    // is only accepted by Mesa with this driconf option.
    setenv("force_compat_shaders", "true", 0);
    char err[512];
    if (dri_headless_init(err, sizeof err)) { last_err = err; return -1; }
#define LOAD(name) p_##name = (PFN_##name)dri_headless_getproc(#name); if (!p_##name) { last_err = "missing GL symbol " #name; return -1; }
    LOAD(glGenTextures) LOAD(glDeleteTextures) LOAD(glBindTexture) LOAD(glTexParameteri) LOAD(glTexImage2D)
    LOAD(glTexSubImage2D) LOAD(glGetTexImage) LOAD(glActiveTexture) LOAD(glGenFramebuffers)
    LOAD(glBindFramebuffer) LOAD(glFramebufferTexture2D) LOAD(glViewport) LOAD(glCreateShader)
    LOAD(glShaderSource) LOAD(glCompileShader) LOAD(glGetShaderiv) LOAD(glGetShaderInfoLog)
    LOAD(glCreateProgram) LOAD(glAttachShader) LOAD(glLinkProgram) LOAD(glGetProgramiv)
    LOAD(glGetProgramInfoLog) LOAD(glUseProgram) LOAD(glGetUniformLocation) LOAD(glUniform1i)
    LOAD(glUniform1f) LOAD(glUniform2f) LOAD(glUniform1fv) LOAD(glUniform3f) LOAD(glUniformMatrix3fv) LOAD(glGenVertexArrays) LOAD(glBindVertexArray)
    LOAD(glGenBuffers) LOAD(glBindBuffer) LOAD(glBufferData) LOAD(glVertexAttribPointer)
    LOAD(glEnableVertexAttribArray) LOAD(glDrawElements) LOAD(glFinish) LOAD(glGetError) LOAD(glGetString)
    LOAD(glPixelStorei)
#undef LOAD
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);   // Project.cpp:76
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glGenFramebuffers(1, &fbo);               // Project.cpp:123
    // quad: Project.cpp:125-141, Vbo.cpp:8-34
    const float positions[] = { -1, -1, 0, -1, 1, 0, 1, 1, 0, 1, -1, 0 };
    const float texcoords[] = { 0, 0, 0, 1, 1, 1, 1, 0 };
    const int indices[] = { 0, 1, 3, 2, 3, 1 };
    p_glGenVertexArrays(1, &vao);
    p_glBindVertexArray(vao);
    GLuint b[3];
    p_glGenBuffers(3, b);
    p_glBindBuffer(GL_ARRAY_BUFFER, b[0]);
    p_glBufferData(GL_ARRAY_BUFFER, sizeof positions, positions, GL_STATIC_DRAW);
    p_glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 0, nullptr);
    p_glEnableVertexAttribArray(0);
    p_glBindBuffer(GL_ARRAY_BUFFER, b[1]);
    p_glBufferData(GL_ARRAY_BUFFER, sizeof texcoords, texcoords, GL_STATIC_DRAW);
    p_glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 0, nullptr);
    p_glEnableVertexAttribArray(1);
    p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, b[2]);
    p_glBufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof indices, indices, GL_STATIC_DRAW);
    p_glBindVertexArray(0);
    refgl_register_shaders(registry);
    registry["Shader.h:copy_img"] = copy_img;         // Shader.h:179-183
    registry["Shader.h:fragment_set"] = fragment_set; // Shader.h:185-191
    inited = true;
    return 0;
}

const char *refgl_renderer(void)
{
    static std::string s;
    s = std::string((const char *)p_glGetString(GL_RENDERER)) + " | " + (const char *)p_glGetString(GL_VERSION);
    return s.c_str();
}

int refgl_num_shaders(void) { return (int)registry.size(); }
const char *refgl_shader_key(int i)
{
    auto it = registry.begin();
    std::advance(it, i);
    return it->first.c_str();
}
// Full fragment shader text as the reference's Shader::getCode() assembles it.
const char *refgl_shader_code(const char *key)
{
    static std::string s;
    auto it = registry.find(key);
    if (it == registry.end()) { last_err = std::string("unknown shader key ") + key; return nullptr; }
    s = it->second->getCode();
    return s.c_str();
}

void refgl_set_canvas(const float *coords4, float circumference)
{
    memcpy(canvas_coords, coords4, sizeof canvas_coords);
    canvas_circ = circumference;
}

// Texture::Texture (Texture.cpp:21-36). ncomp 1=GL_R32F 2=GL_RG32F 4=GL_RGBA32F; linear 0=GL_NEAREST 1=GL_LINEAR
int refgl_tex_create(int w, int h, int ncomp, int linear)
{
    Tex t; t.w = w; t.h = h; t.ncomp = ncomp;
    t.internal = ncomp == 1 ? GL_R32F : ncomp == 2 ? GL_RG32F : GL_RGBA32F;
    p_glGenTextures(1, &t.id);
    p_glActiveTexture(GL_TEXTURE0);
    p_glBindTexture(GL_TEXTURE_2D, t.id);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    GLint f = linear ? GL_LINEAR : GL_NEAREST;
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, f);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, f);
    p_glTexImage2D(GL_TEXTURE_2D, 0, t.internal, w, h, 0, GL_RGB, GL_UNSIGNED_BYTE, nullptr);
    texs.push_back(t);
    return (int)texs.size() - 1;
}

static GLenum fmt_of(int ncomp) { return ncomp == 1 ? GL_RED : ncomp == 2 ? GL_RG : GL_RGBA; }

// Texture::uploadData (Texture.cpp:38-42)
int refgl_tex_upload(int t, const float *data)
{
    p_glActiveTexture(GL_TEXTURE0);
    p_glBindTexture(GL_TEXTURE_2D, texs[t].id);
    p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, texs[t].w, texs[t].h, fmt_of(texs[t].ncomp), GL_FLOAT, data);
    return p_glGetError() == GL_NO_ERROR ? 0 : -1;
}

// Texture::downloadDataRAW (Texture.cpp:56-67), all components
int refgl_tex_download(int t, float *data)
{
    p_glActiveTexture(GL_TEXTURE0);
    p_glBindTexture(GL_TEXTURE_2D, texs[t].id);
    p_glGetTexImage(GL_TEXTURE_2D, 0, fmt_of(texs[t].ncomp), GL_FLOAT, data);
    return p_glGetError() == GL_NO_ERROR ? 0 : -1;
}

// Texture::swap (Texture.cpp:81-93): exchanges id/width/height, not format
void refgl_tex_swap(int a, int b)
{
    std::swap(texs[a].id, texs[b].id);
    std::swap(texs[a].w, texs[b].w);
    std::swap(texs[a].h, texs[b].h);
}

void refgl_tex_delete(int t) { p_glDeleteTextures(1, &texs[t].id); texs[t].id = 0; }

// ShaderProgram::builder().addShader(vertex2D).addShader(<key>).link() (ShaderProgram.cpp:22-72)
int refgl_program(const char *key)
{
    auto it = registry.find(key);
    if (it == registry.end()) { last_err = std::string("unknown shader key ") + key; return -1; }
    GLuint vs = compile(GL_VERTEX_SHADER, vertex2D->getCode());
    if (!vs) return -1;
    GLuint fs = compile(GL_FRAGMENT_SHADER, it->second->getCode());
    if (!fs) { last_err += "\n--- key " + std::string(key); return -1; }
    GLuint p = p_glCreateProgram();
    p_glAttachShader(p, vs);
    p_glAttachShader(p, fs);
    p_glLinkProgram(p);
    GLint ok = 0;
    p_glGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[4096]; GLsizei n = 0;
        p_glGetProgramInfoLog(p, sizeof log, &n, log);
        last_err = std::string("link failed: ") + log;
        return -1;
    }
    progs.push_back({ p });
    return (int)progs.size() - 1;
}

// probe programs written by the test tooling itself (e.g. to sample llvmpipe's sin/cos/asin/atan): vertex2D + src
int refgl_program_source(const char *frag_src)
{
    GLuint vs = compile(GL_VERTEX_SHADER, vertex2D->getCode());
    if (!vs) return -1;
    GLuint fs = compile(GL_FRAGMENT_SHADER, frag_src);
    if (!fs) return -1;
    GLuint p = p_glCreateProgram();
    p_glAttachShader(p, vs);
    p_glAttachShader(p, fs);
    p_glLinkProgram(p);
    GLint ok = 0;
    p_glGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) { last_err = "probe link failed"; return -1; }
    progs.push_back({ p });
    return (int)progs.size() - 1;
}

int refgl_uniform1f(int prog, const char *name, float v)
{
    p_glUseProgram(progs[prog].id);
    GLint id = p_glGetUniformLocation(progs[prog].id, name);
    if (id < 0) return -1;
    p_glUniform1f(id, v);
    return 0;
}
int refgl_uniform2f(int prog, const char *name, float a, float b)
{
    p_glUseProgram(progs[prog].id);
    GLint id = p_glGetUniformLocation(progs[prog].id, name);
    if (id < 0) return -1;
    p_glUniform2f(id, a, b);
    return 0;
}

int refgl_uniform1i(int prog, const char *name, int v)
{
    p_glUseProgram(progs[prog].id);
    GLint id = p_glGetUniformLocation(progs[prog].id, name);
    if (id < 0) return -1;
    p_glUniform1i(id, v);
    return 0;
}

// GradientNoise.cpp:489-491, the reference's own statements run against this image's libstdc++
void refgl_noise_seed_offset(int seed, float *out3)
{
    std::mt19937 engine{(unsigned)seed};
    std::uniform_real_distribution<float> dis(0.0, 10000.0);
    out3[0] = dis(engine); out3[1] = dis(engine); out3[2] = dis(engine);
}

int refgl_uniform3f(int prog, const char *name, float a, float b, float c)
{
    p_glUseProgram(progs[prog].id);
    GLint id = p_glGetUniformLocation(progs[prog].id, name);
    if (id < 0) return -1;
    p_glUniform3f(id, a, b, c);
    return 0;
}
// glUniformMatrix3fv(id, 1, GL_FALSE, m): 9 floats, column-major (Plate.cpp:37-42)
int refgl_uniform_mat3(int prog, const char *name, const float *m)
{
    p_glUseProgram(progs[prog].id);
    GLint id = p_glGetUniformLocation(progs[prog].id, name);
    if (id < 0) return -1;
    p_glUniformMatrix3fv(id, 1, GL_FALSE, m);
    return 0;
}

// program->bind(); p->setCanvasUniforms(program); p->apply(program, target)
// samplers: n pairs (names[i] -> tex ids[i]); every pass sets the canvas uniforms
// like the reference's hot loops do (OceanCurrents.cpp:503-505).
int refgl_apply(int prog, int target, int n, const char *const *names, const int *ids)
{
    GLuint pid = progs[prog].id;
    p_glUseProgram(pid);
    GLint loc = p_glGetUniformLocation(pid, "cornerCoords");
    if (loc >= 0) p_glUniform1fv(loc, 4, canvas_coords);
    loc = p_glGetUniformLocation(pid, "circumference");
    if (loc >= 0) p_glUniform1f(loc, canvas_circ);
    p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glViewport(0, 0, texs[target].w, texs[target].h);
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, texs[target].id, 0);
    for (int i = 0; i < n; i++) {
        GLint l = p_glGetUniformLocation(pid, names[i]);
        if (l >= 0) {
            p_glActiveTexture(GL_TEXTURE0 + i);
            p_glBindTexture(GL_TEXTURE_2D, texs[ids[i]].id);
            p_glUniform1i(l, i);
        }
    }
    p_glBindVertexArray(vao);
    p_glDrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, nullptr);
    p_glBindVertexArray(0);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    return 0;
}

// The reference's ping-pong hot loop shape (OceanCurrents.cpp:500-508, 701-707):
//   for i<iters { bind; setCanvasUniforms; apply(prog, scratch); field.swap(scratch) }
// pp_slot = index into names/ids of the sampler that reads the ping-ponged field;
// ids[pp_slot] is the field, `scratch` the other buffer.  Returns wall seconds
// (glFinish-bracketed) so the same call serves as the llvmpipe CPU baseline.
double refgl_pingpong(int prog, int scratch, int iters, int n, const char *const *names, int *ids, int pp_slot)
{
    p_glFinish();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; i++) {
        refgl_apply(prog, scratch, n, names, ids);
        refgl_tex_swap(ids[pp_slot], scratch);
    }
    p_glFinish();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void refgl_finish(void) { p_glFinish(); }
int refgl_gl_error(void) { return (int)p_glGetError(); }

}  // extern "C"
