// Shared by the temporally blocked kernels (ocean_jacobi_tb.hip, ocean_diffusion_tb.hip): tile geometry and the
// lane-to-lane / code-bit helpers.
#pragma once
#include <hip/hip_runtime.h>

#define TB_T 8            // halo depth = max sweeps per launch
#define TB_EW 256         // extended tile width  (64 lanes x 4)
#define TB_OW (TB_EW - 2 * TB_T)      // 240 output columns per tile

__device__ __forceinline__ float dpp_from_prev_lane(float v)   // lane i <- lane i-1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_next_lane(float v)   // lane i <- lane i+1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// 0 or -1 from bit BIT of a packed neighbour code.  `tick` is the sweep counter: an unused asm input that keeps the
// extraction inside the sweep loop (left to itself the compiler hoists the lane masks of every row out of the loop
// into VGPR lanes and pays two v_readlane + s_nop per select).
template <int BIT>
__device__ __forceinline__ int code_mask(unsigned code, int tick)
{
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(code), "n"(BIT), "s"(tick));
    return m;
}
