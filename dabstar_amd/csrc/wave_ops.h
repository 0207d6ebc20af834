// wave_ops.h -- butterfly reductions over the 64 lanes of a wavefront that stay in the VALU.
//
// A reduction written with __shfl_xor compiles to ds_bpermute_b32: six DEPENDENT trips through the LDS pipeline per value
// (~100 cycles each with a quiet LDS, more next to the transform's exchange traffic), and every wave of a block takes them
// at the same point -- just before a barrier.  On gfx950 all six exchanges have VALU forms: v_permlane32_swap /
// v_permlane16_swap for the two that cross a 16-lane row, DPP row_ror / row_half_mirror / quad_perm for the rest.
// Partners are combined at distance 32, 16, 8, 4, 2, 1 in this order (the order of the loops these helpers replace: for a
// float sum it is part of the result's rounding).  Every lane must be active; every lane gets the result.
#pragma once
#include <hip/hip_runtime.h>

namespace dabx {

template <class Op> __device__ __forceinline__ unsigned wave_butterfly_u32(unsigned v, Op op)
{
  {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // [0]: lanes l % 32, [1]: lanes 32 + l % 32
    v = op(r[0], r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);      // [0]: the even row of each row pair, [1]: the odd one
    v = op(r[0], r[1]);
  }
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true));                                  // xor 8: row_ror:8
  {
    const int hm = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);                                      // xor 4 = row_half_mirror (i -> 7 - i)
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, hm, 0x1B, 0xF, 0xF, true));                                      //         then quad_perm [3,2,1,0] (i -> i ^ 3)
  }
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));                                   // xor 2: quad_perm [2,3,0,1]
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));                                   // xor 1: quad_perm [1,0,3,2]
  return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
  return __builtin_bit_cast(float, wave_butterfly_u32(__builtin_bit_cast(unsigned, v), [](unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b)); }));
}
__device__ __forceinline__ unsigned wave_xor(unsigned v) { return wave_butterfly_u32(v, [](unsigned a, unsigned b) { return a ^ b; }); }
__device__ __forceinline__ int wave_sum_int(int v)
{
  return (int)wave_butterfly_u32((unsigned)v, [](unsigned a, unsigned b) { return a + b; });
}
__device__ __forceinline__ int wave_min_int(int v)
{
  return (int)wave_butterfly_u32((unsigned)v, [](unsigned a, unsigned b) { return (unsigned)((int)a < (int)b ? (int)a : (int)b); });
}

}  // namespace dabx
