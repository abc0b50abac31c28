/* See dri_headless.h.  TEST INFRASTRUCTURE ONLY. */
#include "dri_headless.h"
#include <GL/gl.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static void *h_dri, *h_glapi;
static const __DRIcoreExtension *core;
static const __DRIswrastExtension *swrast;
static __DRIscreen *screen;
static __DRIcontext *ctx;
static __DRIdrawable *drawable;
static void *(*glapi_getproc)(const char *);

static void ld_getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16; }
static void ld_putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void ld_putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void ld_getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }

static const __DRIswrastLoaderExtension loader_ext = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = ld_getDrawableInfo,
    .putImage = ld_putImage,
    .getImage = ld_getImage,
    .putImage2 = ld_putImage2,
    .getImage2 = ld_getImage2,
};
static const __DRIextension *loader_exts[] = { &loader_ext.base, NULL };

#define FAIL(...) do { snprintf(err, errlen, __VA_ARGS__); return 1; } while (0)

int dri_headless_init(char *err, int errlen)
{
    const char *paths[] = {
        getenv("DMG_SWRAST_DRI") ? getenv("DMG_SWRAST_DRI") : "",
        "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so",
        "/usr/lib64/dri/swrast_dri.so",
        "/usr/lib/dri/swrast_dri.so",
    };
    h_glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!h_glapi) FAIL("dlopen libglapi.so.0: %s", dlerror());
    for (unsigned i = 0; i < sizeof paths / sizeof *paths && !h_dri; i++)
        if (paths[i][0]) h_dri = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h_dri) FAIL("dlopen swrast_dri.so: %s", dlerror());

    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(h_dri, "__driDriverGetExtensions_swrast");
    if (!get_exts) FAIL("no __driDriverGetExtensions_swrast");
    const __DRIextension **drv = get_exts();
    for (int i = 0; drv[i]; i++) {
        if (!strcmp(drv[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)drv[i];
        if (!strcmp(drv[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)drv[i];
    }
    if (!core || !swrast || swrast->base.version < 4) FAIL("DRI_Core/DRI_SWRast(v4) missing");

    const __DRIconfig **configs = NULL;
    screen = swrast->createNewScreen2(0, loader_exts, drv, &configs, NULL);
    if (!screen || !configs || !configs[0]) FAIL("createNewScreen2 failed");

    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3 };
    unsigned e = 0;
    ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL,
                                       2, attribs, &e, NULL);
    if (!ctx) FAIL("createContextAttribs(core 4.3) failed, error %u", e);
    drawable = swrast->createNewDrawable(screen, configs[0], NULL);
    if (!drawable) FAIL("createNewDrawable failed");
    if (!core->bindContext(ctx, drawable, drawable)) FAIL("bindContext failed");

    glapi_getproc = (void *(*)(const char *))dlsym(h_glapi, "_glapi_get_proc_address");
    if (!glapi_getproc) FAIL("no _glapi_get_proc_address");
    return 0;
}

void *dri_headless_getproc(const char *name) { return glapi_getproc ? glapi_getproc(name) : NULL; }

void dri_headless_shutdown(void)
{
    if (core && ctx) { core->unbindContext(ctx); core->destroyContext(ctx); ctx = NULL; }
    if (core && drawable) { core->destroyDrawable(drawable); drawable = NULL; }
    if (core && screen) { core->destroyScreen(screen); screen = NULL; }
}
