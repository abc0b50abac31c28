/* Headless OpenGL 4.3 core context on Mesa llvmpipe WITHOUT X/EGL/OSMesa.
 *
 * TEST INFRASTRUCTURE ONLY (oracle side).  Not part of the product path.
 *
 * Drives Mesa's DRI "swrast" loader interface directly
 * (/usr/include/GL/internal/dri_interface.h): dlopen swrast_dri.so ->
 * __driDriverGetExtensions_swrast -> DRI_Core + DRI_SWRast ->
 * createNewScreen2 / createContextAttribs(core 4.3) / createNewDrawable ->
 * bindContext.  GL entry points come from libglapi's _glapi_get_proc_address.
 * Recipe recorded in SURVEY.md section 8c.
 */
#ifndef DMG_DRI_HEADLESS_H
#define DMG_DRI_HEADLESS_H
#ifdef __cplusplus
extern "C" {
#endif

/* Returns 0 on success; on failure returns nonzero and writes a message to err. */
int dri_headless_init(char *err, int errlen);
/* GL symbol lookup (valid after init). */
void *dri_headless_getproc(const char *name);
void dri_headless_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
