// The host side of the streaming Jacobi kernel: how a launch is cut into units (ocean_jacobi_stream_plan.hip), for the launcher
// (ocean_jacobi_stream.hip) and the deep form (ocean_jacobi_deep.hip).
#pragma once
#include "ocean_internal.h"

struct st_features { int fast = 0, coast = 0, tested = 0; int t_open = 0, t_mixed = 0, t_land = 0;     // tested: steps past the fill; t_*: their level-rows by class
                     int f_open = 0, f_mixed = 0, f_land = 0;                                       // the fill steps' level-rows
                     int bad = 0; };                                                                // steps past the fill that are tested only because a row without an exact divisor pair is in flight
// bad: the band's rows without an exact two-operation divisor pair ([L] flags in the band's own numbering, dmg_ocean::rowbad_host), or null
st_features st_unit_features(const unsigned char *cls, int L, int y_first, int H, int r0, int r1, bool flip, bool twin, const unsigned char *bad = nullptr);
int st_unit_rows();
// the unit list of a launch geometry (cached per handle); nullptr: st_units_failed() says whether a message has been set
const dmg_ocean::tile_list *st_units(dmg_ocean *o, int variant, int lo, int hi, int part, int zone_lo, int zone_hi, int seg);
bool st_units_failed();
