// acq_walk.h -- the two sample-serial float recurrences of the null-symbol search (k_acquire, pipeline.hip), one lane each.
//     sLevel += 0.00001f * (|x| - sLevel)      sample_reader.cpp:245-248
//     level  += d                              timesyncer.cpp:64-66, 78-80
#pragma once
#include <hip/hip_runtime.h>

namespace dabx {

// The serial loops: one lane, 16 samples per iteration, operands as four 16-byte LDS reads requested one half-iteration ahead,
// results as four 16-byte writes, waits counted so that neither is ever waited for (LDS operations complete in order: at each
// wait the two newest reads and the two newest writes may still be on their way).  A lone wave issues an instruction every
// ~6 cycles whether it depends on the last one or not, so the loop is nothing but the recurrence.  Written out as one asm
// block: the compiler's version of this loop waits for its own stores at the loop head (77 cycles per sample), and its
// preferred v_pk_add_f32 for the two sums has four times the latency of v_add_f32 (tools/acq_walk_bench.hip).
// a / out: LDS, 16-byte aligned, readable / writable up to n16 * 16 + 16 floats.  Only in k_acquire (two waves per SIMD: the
// block's 34 fixed registers v200..v233 lie within its budget).
#define DABX_ACQ_SKELETON(STEP)                                                                                                 \
  "ds_read_b128 v[200:203], %[ap]\n\t"                                                                                         \
  "ds_read_b128 v[204:207], %[ap] offset:16\n\t"                                                                               \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "1:\n\t"                                                                                                                      \
  "ds_read_b128 v[208:211], %[ap] offset:32\n\t"                                                                               \
  "ds_read_b128 v[212:215], %[ap] offset:48\n\t"                                                                               \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                                    \
  STEP("v200", "v216", "%[x]") STEP("v201", "v217", "v216") STEP("v202", "v218", "v217") STEP("v203", "v219", "v218")           \
  "ds_write_b128 %[op], v[216:219]\n\t"                                                                                        \
  STEP("v204", "v220", "v219") STEP("v205", "v221", "v220") STEP("v206", "v222", "v221") STEP("v207", "v223", "v222")           \
  "ds_write_b128 %[op], v[220:223] offset:16\n\t"                                                                              \
  "ds_read_b128 v[200:203], %[ap] offset:64\n\t"                                                                               \
  "ds_read_b128 v[204:207], %[ap] offset:80\n\t"                                                                               \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                                    \
  STEP("v208", "v226", "v223") STEP("v209", "v227", "v226") STEP("v210", "v228", "v227") STEP("v211", "v229", "v228")           \
  "ds_write_b128 %[op], v[226:229] offset:32\n\t"                                                                              \
  STEP("v212", "v230", "v229") STEP("v213", "v231", "v230") STEP("v214", "v232", "v231") STEP("v215", "v233", "v232")           \
  "ds_write_b128 %[op], v[230:233] offset:48\n\t"                                                                              \
  "v_mov_b32 %[x], v233\n\t"                                                                                                   \
  "v_add_u32 %[ap], 64, %[ap]\n\t"                                                                                             \
  "v_add_u32 %[op], 64, %[op]\n\t"                                                                                             \
  "s_sub_u32 %[n], %[n], 1\n\t"                                                                                                \
  "s_cmp_lg_u32 %[n], 0\n\t"                                                                                                   \
  "s_cbranch_scc1 1b\n\t"                                                                                                      \
  "s_waitcnt lgkmcnt(0)\n\t"
#define DABX_ACQ_CLOBBERS "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", \
                          "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", \
                          "v228", "v229", "v230", "v231", "v232", "v233", "scc", "memory"
// sLevel += 0.00001f * (|x| - sLevel), sample_reader.cpp:248 (three roundings, no contraction)
#define DABX_ACQ_STEP_S(A, R, P) "v_sub_f32 v224, " A ", " P "\n\tv_mul_f32 v224, %[c], v224\n\tv_add_f32 " R ", " P ", v224\n\t"
// level += d, timesyncer.cpp:66, 80
#define DABX_ACQ_STEP_L(A, R, P) "v_add_f32 " R ", " P ", " A "\n\t"
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p;
}
__device__ __forceinline__ float acq_walk_S(const float *a, float *out, int n16, float S)   // returns sLevel after the last sample
{
  unsigned ap = lds_addr(a), op = lds_addr(out);
  asm volatile(DABX_ACQ_SKELETON(DABX_ACQ_STEP_S) : [ap] "+v"(ap), [op] "+v"(op), [n] "+s"(n16), [x] "+v"(S) : [c] "s"(0.00001f) : DABX_ACQ_CLOBBERS);
  return S;
}
// sLevel alone, results not stored (the exact in-lock tracker, k_level_exact): the same loop without its writes
#define DABX_ACQ_SKELETON_NOSTORE(STEP)                                                                                         \
  "ds_read_b128 v[200:203], %[ap]\n\t"                                                                                         \
  "ds_read_b128 v[204:207], %[ap] offset:16\n\t"                                                                               \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "1:\n\t"                                                                                                                      \
  "ds_read_b128 v[208:211], %[ap] offset:32\n\t"                                                                               \
  "ds_read_b128 v[212:215], %[ap] offset:48\n\t"                                                                               \
  "s_waitcnt lgkmcnt(2)\n\t"                                                                                                    \
  STEP("v200", "v216", "%[x]") STEP("v201", "v217", "v216") STEP("v202", "v218", "v217") STEP("v203", "v219", "v218")           \
  STEP("v204", "v220", "v219") STEP("v205", "v221", "v220") STEP("v206", "v222", "v221") STEP("v207", "v223", "v222")           \
  "ds_read_b128 v[200:203], %[ap] offset:64\n\t"                                                                               \
  "ds_read_b128 v[204:207], %[ap] offset:80\n\t"                                                                               \
  "s_waitcnt lgkmcnt(2)\n\t"                                                                                                    \
  STEP("v208", "v226", "v223") STEP("v209", "v227", "v226") STEP("v210", "v228", "v227") STEP("v211", "v229", "v228")           \
  STEP("v212", "v230", "v229") STEP("v213", "v231", "v230") STEP("v214", "v232", "v231") STEP("v215", "%[x]", "v232")           \
  "v_add_u32 %[ap], 64, %[ap]\n\t"                                                                                             \
  "s_sub_u32 %[n], %[n], 1\n\t"                                                                                                \
  "s_cmp_lg_u32 %[n], 0\n\t"                                                                                                   \
  "s_cbranch_scc1 1b\n\t"                                                                                                      \
  "s_waitcnt lgkmcnt(0)\n\t"
// The recurrence with a CHECKPOINT per 16 samples instead of every value (k_acquire): ck[k] = the state after sample 16 k + 15.
// The values in between are recomputed where they are needed, 64 groups of 16 in parallel (acquire_stream, wave 1): one 4-byte
// LDS write per 16 samples instead of four 16-byte ones -- sLevel walks at ~17 instead of 23.5 cycles per sample, level at ~9
// instead of 15.5.
#define DABX_ACQ_SKELETON_CKPT(STEP)                                                                                            \
  "ds_read_b128 v[200:203], %[ap]\n\t"                                                                                         \
  "ds_read_b128 v[204:207], %[ap] offset:16\n\t"                                                                               \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "1:\n\t"                                                                                                                      \
  "ds_read_b128 v[208:211], %[ap] offset:32\n\t"                                                                               \
  "ds_read_b128 v[212:215], %[ap] offset:48\n\t"                                                                               \
  "s_waitcnt lgkmcnt(3)\n\t"                                                                                                    \
  STEP("v200", "v216", "%[x]") STEP("v201", "v217", "v216") STEP("v202", "v218", "v217") STEP("v203", "v219", "v218")           \
  STEP("v204", "v220", "v219") STEP("v205", "v221", "v220") STEP("v206", "v222", "v221") STEP("v207", "v223", "v222")           \
  "ds_read_b128 v[200:203], %[ap] offset:64\n\t"                                                                               \
  "ds_read_b128 v[204:207], %[ap] offset:80\n\t"                                                                               \
  "s_waitcnt lgkmcnt(2)\n\t"                                                                                                    \
  STEP("v208", "v226", "v223") STEP("v209", "v227", "v226") STEP("v210", "v228", "v227") STEP("v211", "v229", "v228")           \
  STEP("v212", "v230", "v229") STEP("v213", "v231", "v230") STEP("v214", "v232", "v231") STEP("v215", "%[x]", "v232")           \
  "ds_write_b32 %[op], %[x]\n\t"                                                                                               \
  "v_add_u32 %[ap], 64, %[ap]\n\t"                                                                                             \
  "v_add_u32 %[op], 4, %[op]\n\t"                                                                                              \
  "s_sub_u32 %[n], %[n], 1\n\t"                                                                                                \
  "s_cmp_lg_u32 %[n], 0\n\t"                                                                                                   \
  "s_cbranch_scc1 1b\n\t"                                                                                                      \
  "s_waitcnt lgkmcnt(0)\n\t"
__device__ __forceinline__ float acq_walk_S_ckpt(const float *a, float *ck, int n16, float S)
{
  unsigned ap = lds_addr(a), op = lds_addr(ck);
  asm volatile(DABX_ACQ_SKELETON_CKPT(DABX_ACQ_STEP_S) : [ap] "+v"(ap), [op] "+v"(op), [n] "+s"(n16), [x] "+v"(S) : [c] "s"(0.00001f) : DABX_ACQ_CLOBBERS);
  return S;
}
__device__ __forceinline__ float acq_walk_L_ckpt(const float *d, float *ck, int n16, float L)
{
  unsigned ap = lds_addr(d), op = lds_addr(ck);
  asm volatile(DABX_ACQ_SKELETON_CKPT(DABX_ACQ_STEP_L) : [ap] "+v"(ap), [op] "+v"(op), [n] "+s"(n16), [x] "+v"(L) : : DABX_ACQ_CLOBBERS);
  return L;
}
__device__ __forceinline__ float acq_walk_S_only(const float *a, int n16, float S)
{
  unsigned ap = lds_addr(a);
  asm volatile(DABX_ACQ_SKELETON_NOSTORE(DABX_ACQ_STEP_S) : [ap] "+v"(ap), [n] "+s"(n16), [x] "+v"(S) : [c] "s"(0.00001f) : DABX_ACQ_CLOBBERS);
  return S;
}
__device__ __forceinline__ void acq_walk_L(const float *d, float *out, int n16, float L)
{
  unsigned ap = lds_addr(d), op = lds_addr(out);
  asm volatile(DABX_ACQ_SKELETON(DABX_ACQ_STEP_L) : [ap] "+v"(ap), [op] "+v"(op), [n] "+s"(n16), [x] "+v"(L) : : DABX_ACQ_CLOBBERS);
}
}  // namespace dabx
