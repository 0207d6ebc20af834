// level_par.h -- SampleReader's level recurrence (sample_reader.cpp:245-248), 1024 samples at a time, bit for bit.
//
//     sLevel += 0.00001f * (|x| - sLevel)          three float operations per sample, each rounded
//
// is sample-serial: 16.5 cycles per sample for a lone wave (acq_walk.h), 1.4 ms per frame and stream.  What makes it
// parallel all the same: two trajectories that start one float apart stay exactly one float apart, step after step, unless a
// rounding merges them (probability ~1e-5 per sample and pair) or both additions are exact ties (2^-16); and a trajectory
// can be predicted to a few floats -- the same recurrence in real arithmetic from an exact start value drifts away from the
// float one by the accumulated roundings only (0.3 sqrt(n) floats rms).  So, per block of 64 groups of 16 samples, one group
// per lane:
//   1. guess G_g, the level before group g: real-arithmetic recurrence from the block's exact start value S0, as an
//      offset r_g = R_g - S0 (|r| < 1e-2 S: float arithmetic is exact enough by two orders), a weighted prefix scan over the
//      lanes (DPP, no LDS);
//   2. walk the group twice, from G_g - K and G_g + K floats (K = 32), and check at EVERY step that the two are still
//      2 K floats apart, that neither addition was a tie (Fast2Sum residual == half an ulp), and -- over the group -- that
//      everything stayed inside one binade and no sample exceeded 1024 x the level (|d| <= |S| for Fast2Sum).  Then EVERY start
//      value in [G_g - K, G_g + K] is mapped by the same shift (see below): end = end_lo + (start - start_lo);
//   3. the true start of group g + 1 is the true end of group g: an integer prefix sum over the lanes of D_g = E_g - G_{g+1}
//      gives every group its offset k_g from its guess;
//   4. the first group whose checks failed or whose k_g lies outside [-K, K] is walked from its true start (known: everything
//      before it is settled), the offsets behind it are corrected by what that changed; repeat until no group is left.
//      About one group per block for receiver input (tools/level_bracket_sim.c: 0.65 - 1.3).  If more than six groups need it,
//      steps 2 - 4 are repeated with K = 128 (exact zeros: the rounding errors of successive steps are correlated over a thousand
//      samples there and the trajectory leaves a +-32 bracket; merges are four times as frequent with the wide one, 2 - 3 groups
//      per block); and if more than twelve still do -- a constant envelope, which parks the float recurrence in its dead zone
//      away from the real one; a NaN -- the rest of the block goes to the serial walker and costs what it always did.
// Why 2 holds.  One step maps S to fl(S + d(S)), d(S) = fl(c fl(a - S)) non-increasing in S (rounding is monotone).  For the
// 2 K + 1 lattice points S_j between the two walked ones, inside one binade with spacing u: S_j + d_j = S_j + m u + e_j where
// m u = r_lo - S_lo is what the low walk added after rounding and e_j = d_j - m u its residual.  e_hi <= e_j <= e_lo, and the
// step checks say: r_hi - S_hi = m u too (the distance is kept), |e_lo| < u / 2 and |e_hi| < u / 2 (no tie at either end).  So
// every |e_j| < u / 2, every S_j + d_j rounds to S_j + m u: the step is the same shift for all of them.  Induction over the steps.
// The model in tools/level_bracket_sim.c checks the scheme (and, with -DBRUTE, claim 2 for every start value of every group)
// against the serial recurrence on 140 M samples of seven kinds of input; tools/level_par_check.hip does the same with
// this code on the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include "acq_walk.h"

namespace dabx {

constexpr int LVL_K = 32, LVL_K_WIDE = 128;                // half-width of the brackets, first and second tier
constexpr int LVL_MAX_WALKED = 6, LVL_MAX_WALKED_WIDE = 12; // groups settled one by one per tier before the next tier / the serial walker takes over

#ifdef DABX_LV_DEBUG
__device__ unsigned *dabx_lv_debug;
#endif
struct LevelPar {
  float q, p1, p2;                       // per lane: alpha^lane - 1, alpha^(lane % 16 + 1), alpha^(lane % 32 + 1); alpha = (1 - c)^16: a group's slope
  static constexpr double C = (double)0.00001f;
  __device__ static double ipow(double b, int n) { double r = 1.0; for (; n; n >>= 1, b *= b) if (n & 1) r *= b; return r; }
  __device__ void init(int lane)
  {
    const double alpha = ipow(1.0 - C, 16);
    q = (float)(ipow(alpha, lane) - 1.0);
    p1 = (float)ipow(alpha, (lane & 15) + 1);
    p2 = (float)ipow(alpha, (lane & 31) + 1);
  }
  template <int CTRL, int ROW_MASK> __device__ static __forceinline__ float dpp(float v)
  {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
  }
  template <int CTRL, int ROW_MASK> __device__ static __forceinline__ unsigned dppu(unsigned v)
  {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
  }
  __device__ static __forceinline__ unsigned rl(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }

  // a: the block's magnitudes in LDS (16-byte aligned, ng * 16 of them); ng: groups, 1..64, wave-uniform; S0: the level before the
  // block (the same in every lane).  Returns the level after the block in every lane; ck (LDS, or nullptr): ck[g] = the level before
  // group g, ck[ng] = after the last.  Every lane of the wave must be active.  fallbacks (if given) += groups walked in step 4.
  __device__ __forceinline__ float block(const float *a, int ng, float S0, float *ck, int lane, int *fallbacks = nullptr) const
  {
    float x[16];
    {
      const float4 *p = reinterpret_cast<const float4 *>(a + 16 * (lane < ng ? lane : 0));
      const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
      x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
      x[8] = v2.x; x[9] = v2.y; x[10] = v2.z; x[11] = v2.w; x[12] = v3.x; x[13] = v3.y; x[14] = v3.z; x[15] = v3.w;
    }
    // 1. the guess
    constexpr float W[16] = {9.998499852e-06f, 9.998599838e-06f, 9.998699825e-06f, 9.998799813e-06f, 9.998899802e-06f, 9.998999792e-06f,
                             9.999099783e-06f, 9.999199775e-06f, 9.999299768e-06f, 9.999399762e-06f, 9.999499757e-06f, 9.999599753e-06f,
                             9.999699750e-06f, 9.999799748e-06f, 9.999899747e-06f, 9.999999747e-06f};      // c (1 - c)^(15 - k)
    float B = 0.f, amax = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) { B = __builtin_fmaf(W[k], x[k], B); amax = fmaxf(amax, x[k]); }
    constexpr float A1 = 0.99984001200348138f, A2 = (float)(0.99984001200348138 * 0.99984001200348138),
                    A4 = (float)(0.99984001200348138 * 0.99984001200348138 * 0.99984001200348138 * 0.99984001200348138),
                    A8 = (float)((0.99984001200348138 * 0.99984001200348138 * 0.99984001200348138 * 0.99984001200348138) *
                                 (0.99984001200348138 * 0.99984001200348138 * 0.99984001200348138 * 0.99984001200348138));
    B = __builtin_fmaf(A1, dpp<0x111, 0xF>(B), B);        // row_shr:1 (lanes without a source get 0)
    B = __builtin_fmaf(A2, dpp<0x112, 0xF>(B), B);        // row_shr:2
    B = __builtin_fmaf(A4, dpp<0x114, 0xF>(B), B);        // row_shr:4
    B = __builtin_fmaf(A8, dpp<0x118, 0xF>(B), B);        // row_shr:8
    B = __builtin_fmaf(p1, dpp<0x142, 0xA>(B), B);        // row_bcast:15 into rows 1 and 3
    B = __builtin_fmaf(p2, dpp<0x143, 0xC>(B), B);        // row_bcast:31 into rows 2 and 3
    const float Bex = dpp<0x138, 0xF>(B);                 // wave_shr:1: everything before this lane's group
    const float G = S0 + __builtin_fmaf(q, S0, Bex);      // (lane 0: S0 itself)
    const unsigned Gb = __builtin_bit_cast(unsigned, G);
    const unsigned long long valid = ng >= 64 ? ~0ull : ((1ull << ng) - 1ull);
    // Two tiers: brackets of +-32 floats first; if more groups than can be walked singly end up outside theirs -- an input whose float
    // trajectory runs away from the real one faster than the roundings' random walk: exact zeros, where the error of successive steps is
    // correlated over a thousand samples -- once more with +-128 (merges are four times as frequent then: 2-3 groups per block) before
    // the rest of the block goes to the serial walker.
#pragma nounroll
    for (int tier = 0;; tier++) {
    const unsigned K = tier ? LVL_K_WIDE : LVL_K;
    const int max_walked = tier ? LVL_MAX_WALKED_WIDE : LVL_MAX_WALKED;
    // 2. the two walks
    float lo = __builtin_bit_cast(float, Gb - K), hi = __builtin_bit_cast(float, Gb + K);
    const float lo0 = lo;
    float mn = lo, mx = hi, tmax = 0.f;
    unsigned dmin = 2u * K, dmax = 2u * K;                // distance of the two walks after each step: must stay 2 K
    // Two samples per asm block, 14 instructions per sample and lane pair of walks; one block so that the compiler puts nothing in
    // between (it pads dependent asm statements with s_nop, and a lone wave pays full price for every instruction it issues):
    //   t = x - S; t = c t; S' = S + t;  z = t - (S' - S)  [what the addition rounded away: Fast2Sum, |t| <= |S|]
#define DABX_LV_STEP(X, LO, HI, NL, NH, ZL, ZH)                                                                                 \
    "v_sub_f32 %[tl], " X ", " LO "\n\tv_sub_f32 %[th], " X ", " HI "\n\t"                                                       \
    "v_mul_f32 %[tl], 0x3727c5ac, %[tl]\n\tv_mul_f32 %[th], 0x3727c5ac, %[th]\n\t"                                               \
    "v_add_f32 " NL ", " LO ", %[tl]\n\tv_add_f32 " NH ", " HI ", %[th]\n\t"                                                     \
    "v_sub_f32 " ZL ", " NL ", " LO "\n\tv_sub_f32 " ZH ", " NH ", " HI "\n\t"                                                   \
    "v_sub_f32 " ZL ", %[tl], " ZL "\n\tv_sub_f32 " ZH ", %[th], " ZH "\n\t"                                                     \
    "v_max3_f32 %[tm], %[tm], |" ZL "|, |" ZH "|\n\t"                                                                            \
    "v_sub_u32 " ZL ", " NH ", " NL "\n\t"
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      float l2, h2, tl, th, z0, z1, z2, z3;
      asm(DABX_LV_STEP("%[x0]", "%[lo]", "%[hi]", "%[l2]", "%[h2]", "%[z0]", "%[z1]")
          DABX_LV_STEP("%[x1]", "%[l2]", "%[h2]", "%[lo]", "%[hi]", "%[z2]", "%[z3]")
          "v_min3_u32 %[dmin], %[dmin], %[z0], %[z2]\n\t"
          "v_max3_u32 %[dmax], %[dmax], %[z0], %[z2]\n\t"
          "v_min3_f32 %[mn], %[mn], %[l2], %[lo]\n\t"
          "v_max3_f32 %[mx], %[mx], %[h2], %[hi]"
          : [lo] "+v"(lo), [hi] "+v"(hi), [tm] "+v"(tmax), [dmin] "+v"(dmin), [dmax] "+v"(dmax), [mn] "+v"(mn), [mx] "+v"(mx),
            [l2] "=&v"(l2), [h2] "=&v"(h2), [tl] "=&v"(tl), [th] "=&v"(th), [z0] "=&v"(z0), [z1] "=&v"(z1), [z2] "=&v"(z2), [z3] "=&v"(z3)
          : [x0] "v"(x[k]), [x1] "v"(x[k + 1]));
    }
#undef DABX_LV_STEP
    const unsigned dacc = (dmin ^ (2u * K)) | (dmax ^ (2u * K));
    const unsigned e0 = __builtin_bit_cast(unsigned, lo0) >> 23;
    const float half_ulp = __builtin_bit_cast(float, (e0 - 24u) << 23);
    const bool safe = Gb > (25u << 23) + K && Gb < 0x7f000000u && dacc == 0 && tmax != half_ulp &&
                      (__builtin_bit_cast(unsigned, mn) >> 23) == (__builtin_bit_cast(unsigned, mx) >> 23) && amax <= 1024.f * lo0;
    const unsigned E = __builtin_bit_cast(unsigned, lo) + K;           // where the walk from G itself would have ended, if the group is safe
    // 3. offsets from the guesses: k_g = sum over the groups before g of (E_j - G_{j+1}), i.e. the inclusive prefix sum of
    //    D_g = E_{g-1} - G_g, D_0 = 0 (mod 2^32 throughout: unsafe groups put garbage in, step 4 takes it out again)
    unsigned Eprev;                                                      // wave_shr:1 as an instruction of its own: folded into the subtraction
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(Eprev) : "v"(E));   // (v_subrev_u32_dpp) it shifted the wrong operand
    unsigned k = lane == 0 ? 0u : Eprev - Gb;
    k += dppu<0x111, 0xF>(k);
    k += dppu<0x112, 0xF>(k);
    k += dppu<0x114, 0xF>(k);
    k += dppu<0x118, 0xF>(k);
    k += dppu<0x142, 0xA>(k);
    k += dppu<0x143, 0xC>(k);
#ifdef DABX_LV_DEBUG
    if (dabx_lv_debug) { unsigned *o = dabx_lv_debug + 8 * lane; o[0] = Gb; o[1] = __builtin_bit_cast(unsigned, lo); o[2] = __builtin_bit_cast(unsigned, hi); o[3] = E; o[4] = k; o[5] = safe; o[6] = __builtin_bit_cast(unsigned, B); o[7] = __builtin_bit_cast(unsigned, x[0]); }
#endif
    // 4. settle the groups in order
    bool walked = false;
    unsigned tend = 0;
    int n_walked = 0, serial_from = 64;
    float S_serial = 0.f;
    unsigned long long todo = __ballot(!(safe && k + K <= 2u * K)) & valid;
    if (tier == 0 && __builtin_popcountll(todo) > max_walked) continue;
    while (todo) {
      const int gs = __builtin_ctzll(todo);
      if (++n_walked > max_walked) {
        // an input that defeats the guess (a constant envelope parks the float recurrence in its dead zone, away from the real one; a
        // NaN): the rest of the block sample by sample, from the true start of group gs -- the block then costs about what it always did
        S_serial = __builtin_bit_cast(float, rl(Gb + k, gs));
        const int rest = __builtin_amdgcn_readfirstlane(ng - gs);
        S_serial = ck ? acq_walk_S_ckpt(a + 16 * gs, ck + gs + 1, rest, S_serial) : acq_walk_S_only(a + 16 * gs, rest, S_serial);
        serial_from = gs;
        if (fallbacks && lane == 0) *fallbacks += rest;
        break;
      }
      // every lane walks its own group from where it believes it starts; only lane gs is known to be right (and is the one that is used)
      float S = __builtin_bit_cast(float, Gb + k);
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        float t;
        asm("v_sub_f32 %[t], %[x0], %[S]\n\tv_mul_f32 %[t], 0x3727c5ac, %[t]\n\tv_add_f32 %[S], %[S], %[t]\n\t"
            "v_sub_f32 %[t], %[x1], %[S]\n\tv_mul_f32 %[t], 0x3727c5ac, %[t]\n\tv_add_f32 %[S], %[S], %[t]\n\t"
            "v_sub_f32 %[t], %[x2], %[S]\n\tv_mul_f32 %[t], 0x3727c5ac, %[t]\n\tv_add_f32 %[S], %[S], %[t]\n\t"
            "v_sub_f32 %[t], %[x3], %[S]\n\tv_mul_f32 %[t], 0x3727c5ac, %[t]\n\tv_add_f32 %[S], %[S], %[t]"
            : [S] "+v"(S), [t] "=&v"(t) : [x0] "v"(x[i]), [x1] "v"(x[i + 1]), [x2] "v"(x[i + 2]), [x3] "v"(x[i + 3]));
      }
      const unsigned te = rl(__builtin_bit_cast(unsigned, S), gs);
      if (lane == gs) { walked = true; tend = te; }
      if (fallbacks && lane == 0) ++*fallbacks;
      if (gs == 63) break;
      const unsigned delta = te - rl(Gb, gs + 1) - rl(k, gs + 1);      // what group gs + 1 really starts from, against what the sum said
      if (lane > gs) k += delta;
      todo = __ballot(!(safe && k + K <= 2u * K)) & valid & (~0ull << (gs + 1));
    }
    if (ck && lane < ng && lane <= serial_from) ck[lane] = __builtin_bit_cast(float, Gb + k);
    if (serial_from < 64) return S_serial;                              // (ck[serial_from + 1 .. ng] are the walker's)
    const float S_end = __builtin_bit_cast(float, rl(walked ? tend : E + k, ng - 1));
    if (ck && lane == 0) ck[ng] = S_end;
    return S_end;
    }
  }
};

// The other recurrence of the null-symbol search, the 50-tap moving sum as the reference keeps it (timesyncer.cpp:64-66, 78-80):
//     level += d            one float addition per sample, d = |x[n]| - |x[n - 50]| (or |x[n]| while the window fills)
// stays sample-serial (acq_walk_L_ckpt).  The same scheme was built for it and measured (round 4): d does not depend on the level, so
// one walk per group suffices -- but d is a difference of two magnitudes and has only six or seven bits below the level's last place, so
// one addition in a hundred is an exact tie (round-to-even then sends odd and even neighbours different ways: one group in six), and a
// level near a power of two (the moving sum of noise wanders +-7 %) crosses the binade every few groups: 13 000 - 17 000 cycles per block
// against the serial walk's 9 000 - 11 000.  What is kept is the one case that is free: a segment of exact zeros at level zero (a
// drop-out, a squelched input) needs no walk at all.
// d: the increments in LDS (16-byte aligned, ng * 16 of them, readable 16 floats further); ng: 1..64, wave-uniform; L0: the level before them;
// ck: ck[g] = the level before group g, ck[ng] = after the last.  Every lane of the wave must be active.
__device__ __forceinline__ void level_sum_block(const float *d, int ng, float L0, float *ck, int lane)
{
  const float4 *p = reinterpret_cast<const float4 *>(d + 16 * (lane < ng ? lane : 0));
  const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
  const float amax = fmaxf(fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))), fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)))),
                           fmaxf(fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))), fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)))));
  const unsigned long long valid = ng >= 64 ? ~0ull : ((1ull << ng) - 1ull);
  if (L0 == 0.f && (__ballot(amax != 0.f) & valid) == 0) {              // 0 + 0 = 0
    if (lane <= ng) ck[lane] = 0.f;
    if (ng == 64 && lane == 0) ck[64] = 0.f;
    return;
  }
  if (lane == 0) ck[0] = L0;
  acq_walk_L_ckpt(d, ck + 1, ng, L0);
}

}  // namespace dabx
