// Entry points declared in include/demiurge_hip.h whose kernels are not written yet.
// They fail loudly; nothing falls back to the CPU.
#include "ocean_internal.h"
#define NOT_YET(name) return dmg_fail(DMG_ESTATE, name ": kernel not implemented in this build")
extern "C" {
int dmg_blur(dmg_ctx *, const dmg_grid *, float *, float) { NOT_YET("dmg_blur"); }
int dmg_temperature_steps(dmg_ctx *, const dmg_grid *, float *, const float *, int, int) { NOT_YET("dmg_temperature_steps"); }
int dmg_erosion_pass(dmg_ctx *, const dmg_grid *, float *, const float *, const float *, float, float) { NOT_YET("dmg_erosion_pass"); }
}
