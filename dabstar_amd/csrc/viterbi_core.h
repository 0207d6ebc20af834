// viterbi_core.h -- K=7 rate-1/4 Viterbi decoder, one wavefront (64 lanes) per trellis.
//
// Replaces ViterbiSpiral::deconvolve (base/support/viterbi_spiral/viterbi_spiral.cpp:95-126, canonical
// scalar body viterbi_scalar.h:9-94) with a CDNA4 formulation that is bit-identical to it:
//
//  * lane = trellis state, in-place butterflies.  At step t lane j holds state rotl6(j, t mod 6), so the
//    butterfly partner (state ^ 32) is always lane j ^ (1 << ((5 - t) mod 6)): the exchange is a DPP
//    quad_perm / row_ror or a ds_swizzle, never a general permutation, and no metric ever moves.
//  * path metrics are kept doubled and centred: the reference adds metric / (1020 - metric) with
//    metric = sum(Branch ^ sym); subtracting the common 510 and doubling gives w = sum(+-(2 sym - 255)),
//    so both branches are own + w and partner - w.  Decisions depend only on metric differences and the
//    tie rule (tie -> predecessor i, not i + 32) is kept, so every decision bit equals the reference's.
//    int32 metrics, no renormalisation (growth <= 1020 * 9222 < 2^31), exactly like the scalar reference.
//  * the branch metric has only 8 signed values per step (polynomials 109 and 109 coincide): they are
//    computed once per step by the lane that fetched the step's 4 symbols and parked in LDS; each lane
//    then reads its own value with one ds_read_i16.
//  * decision bits are accumulated per lane (acc = 2 acc + d) and streamed to an HBM/L2-resident scratch
//    as one coalesced 256-B store per 30 steps, so LDS does not limit occupancy.  Chain-back runs on the
//    scalar unit: the lane index of the surviving path is a wave-uniform value, each step is one
//    v_readlane plus a few SALU ops, and the per-step exchange bit is the only bit of it that changes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_ops.h"

namespace dabx {

constexpr int VIT_BLK = 60;   // trellis steps per LDS block (multiple of 6, <= 64)
constexpr int VIT_DW = 30;    // steps per 32-bit decision word (multiple of 6, <= 32)

__host__ __device__ inline int vit_words(int nbits) { return (nbits + 6 + VIT_DW - 1) / VIT_DW; }
__host__ __device__ inline int vit_blocks(int nbits) { return (nbits + 6 + VIT_BLK - 1) / VIT_BLK; }
// decision scratch per trellis: u32[vit_blocks*2][64]
__host__ __device__ inline size_t vit_scratch_words(int nbits) { return (size_t)vit_blocks(nbits) * 2 * 64; }

struct VitSyms { int x0, x1, x2, x3; };   // 2*sym - 255 of the step's four soft symbols

__device__ __forceinline__ int vit_sym_from_i16(int16_t s)
{
  // viterbi_scalar.h:34-40: i16 tmp = s; tmp += 127 (wraps); clamp to 0..255
  int v = (int16_t)(s + 127);
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  return 2 * v - 255;
}
__device__ __forceinline__ int vit_sym_from_u8(uint8_t sym) { return 2 * (int)sym - 255; }
__device__ __forceinline__ int vit_sym_from_i16_sat(int16_t s)      // viterbi_16way.h:73-76: saturating +127, then 0..255
{
  int v = (int)s + 127;
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  return 2 * v - 255;
}

__device__ __forceinline__ int rotl6(int x, int c) { return ((x << c) | (x >> (6 - c))) & 63; }

// Byte offset (pattern index * 2) into a step's 8-entry branch-metric row, for lane j in step class c.
__device__ __forceinline__ int vit_pat_off(int lane, int c)
{
  const int i = rotl6(lane, c) & 31;
  // Branch[p][i] = parity(2i & poly_p): viterbi_spiral.cpp:27-37 with polys 109,79,83,109
  const int c0 = ((i >> 1) ^ (i >> 2) ^ (i >> 4)) & 1;   // polys 0 and 3
  const int c1 = (i ^ (i >> 1) ^ (i >> 2)) & 1;          // poly 1
  const int c2 = (i ^ (i >> 3)) & 1;                     // poly 2
  return (c0 * 4 + c1 * 2 + c2) * 2;
}

template <int C> __device__ __forceinline__ int vit_exchange(int m, int lane)
{
  // partner lane = lane ^ (1 << ((5 - C) % 6)); all six exchanges stay in the VALU (DPP, and gfx950's permlane swaps for the
  // two that cross a 16-lane row) -- no trip through the LDS crossbar on the add-compare-select chain
  if constexpr (C == 5) return __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, true);        // xor 1  quad_perm [1,0,3,2]
  else if constexpr (C == 4) return __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, true);   // xor 2  quad_perm [2,3,0,1]
  else if constexpr (C == 3) {                                                                 // xor 4 = (i ^ 7) ^ 3
    const int hm = __builtin_amdgcn_update_dpp(0, m, 0x141, 0xF, 0xF, true);                   //   row_half_mirror: i -> 7 - i
    return __builtin_amdgcn_update_dpp(0, hm, 0x1B, 0xF, 0xF, true);                           //   quad_perm [3,2,1,0]: i -> i ^ 3
  }
  else if constexpr (C == 2) return __builtin_amdgcn_update_dpp(0, m, 0x128, 0xF, 0xF, true);  // xor 8  row_ror:8
  else if constexpr (C == 1) {                                                                 // xor 16
    const auto r = __builtin_amdgcn_permlane16_swap(m, m, false, false);                       //   odd rows of [0] <-> even rows of [1]
    return (lane & 16) ? (int)r[0] : (int)r[1];
  }
  else {                                                                                       // xor 32
    const auto r = __builtin_amdgcn_permlane32_swap(m, m, false, false);                       //   upper half of [0] <-> lower half of [1]
    return (lane & 32) ? (int)r[0] : (int)r[1];
  }
}

struct VitLaneConst {
  int pat[6];        // LDS byte offset of this lane's branch metric within a step row, per class
  int sgn[6];        // +1 where this lane holds the lower predecessor of its butterfly in class c, -1 where the upper
  int lane;
  unsigned inv30;    // canonical body: bits of a 30-step decision word this lane accumulates INVERTED (see vit_step)
};

__device__ __forceinline__ VitLaneConst vit_lane_const(int lane)
{
  VitLaneConst k;
  k.inv30 = 0;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    k.pat[c] = vit_pat_off(lane, c);
    const int upper = (lane >> ((5 - c) % 6)) & 1;
    k.sgn[c] = upper ? -1 : 1;
    if (c >= 2 && upper)                               // DPP classes only (the swap classes produce the decision directly)
#pragma unroll
      for (int s = c; s < VIT_DW; s += 6) k.inv30 |= 1u << (VIT_DW - 1 - s);
  }
  k.lane = lane;
  return k;
}

// One add-compare-select step of class C.  viterbi_scalar.h:25-26: decision = (value through predecessor i) > (value through
// predecessor i + 32), ties -> predecessor i.  A lower lane of the pair owns predecessor i (co = own + w), an upper lane owns
// predecessor i + 32, so the test is co > cp below and cp > co above.  What sits on the wave's dependent-instruction chain is
// only  m -> partner's m (DPP operand of the subtraction, or a permlane swap) -> min:
//  * DPP classes (C = 2..5): cp = dpp(m) - w and  g = (co + u > cp)  with u = 1 in upper lanes: below that is the decision,
//    above it is (co >= cp), the decision's complement -- the lane accumulates g and the complement is undone with one xor
//    per 30-step word (inv30).  co + u = m + (w + u): w + u does not depend on m.
//  * swap classes (C = 0, 1: partner 32 / 16 lanes away): v_permlane{32,16}_swap(m, m) leaves {own, partner} in r[0], r[1] in
//    lane-dependent order (own first in the lower lanes).  With the lane's signed w:  x0 = r[0] + sgn w, x1 = r[1] - sgn w  is
//    (co, cp) below and (cp, co) above -- min(x0, x1) is the new metric and x0 > x1 is the decision in BOTH halves: no select
//    after the swap, no complement.
// (Round 3: the select after the swap and the chain  co - cp -> * sgn -> compare -> add-with-carry  made a lone wave's step 72
// cycles; profiles/r03_fic_phase_timing.txt.)
template <int C>
__device__ __forceinline__ void vit_step(int &m, unsigned &acc, const char *wrow, const VitLaneConst &k)
{
  const int w = *reinterpret_cast<const int16_t *>(wrow + k.pat[C]);
  if constexpr (C >= 2) {
    const int w2 = w - (k.sgn[C] >> 1);                  // + 1 in upper lanes (sgn = -1), loop invariant
    const int co = m + w, co2 = m + w2;
    const int cp = vit_exchange<C>(m, k.lane) - w;
    m = co < cp ? co : cp;
    asm("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(co2), "v"(cp) : "vcc");
  } else {
    const int ws = __mul24(w, k.sgn[C]);
    int x0, x1;
    if constexpr (C == 1) { const auto r = __builtin_amdgcn_permlane16_swap(m, m, false, false); x0 = (int)r[0] + ws; x1 = (int)r[1] - ws; }
    else { const auto r = __builtin_amdgcn_permlane32_swap(m, m, false, false); x0 = (int)r[0] + ws; x1 = (int)r[1] - ws; }
    m = x0 < x1 ? x0 : x1;
    asm("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(x0), "v"(x1) : "vcc");
  }
}

// ---- the reference's SIMD builds (VITERBI_AVX2: viterbi_16way.h; VITERBI_SSE2 / NEON: viterbi_8way.h) -------------
// Same trellis, different arithmetic.  RULE 1 (AVX2): path metrics are uint16 with SATURATING adds (_mm256_adds_epu16,
// :36-39), the survivor is min and the decision is `survivor == m1` (:44-45), i.e. a TIE goes to predecessor i + 32 (the
// scalar body sends it to i), and after every second step the new metrics are reduced by their minimum when state 0's metric
// of the step BEFORE exceeded 60000 (renormalize, :9-25: the test reads metrics2[0], the subtraction works on new_metrics).
// RULE 2 (SSE2 / NEON, viterbi_8way.h:9-53): SIGNED int16 metrics, adds saturating at 32767, decision = m0 > m1 (a tie keeps
// predecessor i, as in the scalar body), the same renormalisation with threshold 30000.
// Reproduced here in the reference's own metric domain (start 0 / 1000, branch metric 0..1020) so that saturation and
// renormalisation fall on the same steps: bit-identical to the respective object code (tests/test_oracle_ref.py builds both).
template <int C, int RULE>
__device__ __forceinline__ void vit_step_simd(int &m, unsigned &acc, const char *wrow, const VitLaneConst &k)
{
  constexpr int CEIL = RULE == 1 ? 65535 : 32767;
  const int w = *reinterpret_cast<const int16_t *>(wrow + k.pat[C]);      // 2 * metric - 1020
  const int bm = (w + 1020) >> 1;                                         // sum (Branch ^ sym), viterbi_16way.h:30-31
  const int partner = vit_exchange<C>(m, k.lane);
  int a = m + bm, b = partner + (1020 - bm);
  a = a > CEIL ? CEIL : a; b = b > CEIL ? CEIL : b;                       // adds_epu16 / adds_epi16
  // AVX2: lower lane of the pair: decision0 = (min == m1) = (b <= a); upper lane: decision1 = (min == m3) = (a <= b)
  // SSE2: decision0 = m0 > m1 = (a > b) in the lower lane, decision1 = m2 > m3 = (b > a) in the upper lane
  const int dd = __mul24(a - b, k.sgn[C]);
  m = a < b ? a : b;
  if constexpr (RULE == 1) asm("v_cmp_le_i32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(dd) : "vcc");
  else asm("v_cmp_lt_i32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(dd) : "vcc");
}
__device__ __forceinline__ int vit_wave_min(int v) { return wave_min_int(v); }
template <int C, int RULE>
__device__ __forceinline__ void vit_pair_simd(int &m, unsigned &acc, const char *row, const VitLaneConst &k)
{
  vit_step_simd<C, RULE>(m, acc, row, k);
  const int m0_before = __builtin_amdgcn_readfirstlane(m);                // state 0 lives in lane 0 in every step class
  vit_step_simd<C + 1, RULE>(m, acc, row + 16, k);
  if (m0_before > (RULE == 1 ? 60000 : 30000)) m -= vit_wave_min(m);      // renormalize: wave-uniform, rare
}

// Forward pass.  wtab: this wave's LDS area, VIT_BLK rows of 8 int16.  dec: u32[vit_blocks*2][64].
// RULE 1 / 2: arithmetic of the reference's AVX2 / SSE2 builds (see above) instead of the canonical scalar body (0).
template <int RULE = 0, class Src>
__device__ __forceinline__ void vit_forward(const Src &src, int nsteps, char *wtab, uint32_t *dec, int lane,
                                            const VitLaneConst &k)
{
  int m = RULE ? (lane == 0 ? 0 : 1000) : (lane == 0 ? 0 : 2000);   // viterbi_spiral.cpp:98-101 (0 / 1000); doubled in the canonical form
  const int nblk = (nsteps + VIT_BLK - 1) / VIT_BLK;
  // A symbol source delivers a step's four soft symbols in three phases so that no memory latency sits between the blocks:
  //   key(t)      the step's depuncture-map entry (one load; a source without a map returns t)
  //   raw(key)    the four symbol loads, UNCONDITIONAL (a punctured position reads index 0) and unconverted: nothing waits
  //   syms(raw, key)   conversion to 2 sym - 255, punctured positions -> the neutral value
  // The key of block b + 2 and the raw symbols of block b + 1 are requested before the 60 add-compare-select steps of block b
  // run; syms() is applied afterwards.  (Round 3: with the one-call source the compiler waited for the map, then for each of
  // the four symbol loads in turn -- five serial memory latencies per 60-step block, half of the forward pass of a FIC block.)
  const bool fetcher = lane < VIT_BLK;
  auto tclamp = [&](int t) { return t < nsteps ? t : 0; };
  typename Src::Key k1 = src.key(tclamp(lane));
  typename Src::Raw rw = src.raw(k1);
  VitSyms s = src.syms(rw, k1);
  if (!(fetcher && lane < nsteps)) s = VitSyms{0, 0, 0, 0};
  k1 = src.key(tclamp(VIT_BLK + lane));
  for (int b = 0; b < nblk; b++) {
    const int tn = (b + 1) * VIT_BLK + lane;
    const typename Src::Raw rn = src.raw(k1);                      // block b + 1
    const typename Src::Key kn = k1;
    k1 = src.key(tclamp(tn + VIT_BLK));                            // block b + 2
    if (lane < VIT_BLK) {
      const int y0 = s.x0 + s.x3;
      short v[8];
#pragma unroll
      for (int q = 0; q < 8; q++)
        v[q] = (short)(((q & 4) ? -y0 : y0) + ((q & 2) ? -s.x1 : s.x1) + ((q & 1) ? -s.x2 : s.x2));
      int4 pk;
      pk.x = (unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16);
      pk.y = (unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16);
      pk.z = (unsigned short)v[4] | ((unsigned)(unsigned short)v[5] << 16);
      pk.w = (unsigned short)v[6] | ((unsigned)(unsigned short)v[7] << 16);
      *reinterpret_cast<int4 *>(wtab + lane * 16) = pk;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned acc = 0;
#pragma unroll
    for (int h = 0; h < VIT_BLK / VIT_DW; h++) {
#pragma unroll
      for (int s6 = 0; s6 < VIT_DW; s6 += 6) {
        const char *row = wtab + (h * VIT_DW + s6) * 16;
        if constexpr (RULE != 0) {
          vit_pair_simd<0, RULE>(m, acc, row + 0 * 16, k);
          vit_pair_simd<2, RULE>(m, acc, row + 2 * 16, k);
          vit_pair_simd<4, RULE>(m, acc, row + 4 * 16, k);
        } else {
          vit_step<0>(m, acc, row + 0 * 16, k);
          vit_step<1>(m, acc, row + 1 * 16, k);
          vit_step<2>(m, acc, row + 2 * 16, k);
          vit_step<3>(m, acc, row + 3 * 16, k);
          vit_step<4>(m, acc, row + 4 * 16, k);
          vit_step<5>(m, acc, row + 5 * 16, k);
        }
      }
      dec[(size_t)(b * (VIT_BLK / VIT_DW) + h) * 64 + lane] = RULE == 0 ? acc ^ k.inv30 : acc;
      acc = 0;
    }
    __builtin_amdgcn_wave_barrier();
    s = src.syms(rn, kn);
    if (!(fetcher && tn < nsteps)) s = VitSyms{0, 0, 0, 0};
  }
}

// Chain-back (viterbi_spiral.cpp:114-125) in lane space, on the scalar unit.
// The lane index j of the surviving path is wave-uniform; one trellis step is
//   k = bit (29 - s) of hist[lane j]   (v_readlane_b32 + s_bfe_u32)
//   j = (j & ~(1 << p)) | (k << p)     (the exchange bit of that step is the only bit that changes)
// The decoded bits of a 30-step word are collected LSB = earliest step into raw[wi] (LDS, one dword per
// word, written by lane 0); vit_pack_output() turns them into MSB-first packed bytes afterwards.
constexpr int VIT_RAW_WORDS = 312;          // >= vit_words(9216) + 3

// Six steps (one cycle of the step classes 5..0, exchange bits 0..5) of the chain-back.
//  * vit_tb_masks<G>: the decisions of steps 6 G + 5 .. 6 G of a 30-step word as six wave masks in SGPR pairs (a v_bfe + v_cmp
//    per step, independent of the path: issued a whole group ahead of the chain that consumes them).
//  * vit_tb_chain: on the scalar unit, per step  SCC = mask[j];  j = SCC ? j | bit : j & ~bit  -- four scalar instructions, two
//    of them dependent (s_bitcmp1 -> s_cselect); the output bits are read off j once per group.
// (Round 3, before: v_readlane of the lane's 30-step word with the path's lane as selector -- 4 wait states after the scalar
// write of j -- then bfe / shift / andn2 / or: 72 cycles per step, profiles/r03_fic_phase_timing.txt.)
struct VitMask6 { unsigned long long m5, m4, m3, m2, m1, m0; };    // named members: stay in SGPRs
template <int G> __device__ __forceinline__ VitMask6 vit_tb_masks(unsigned hist)
{
  VitMask6 r;
  r.m5 = __builtin_amdgcn_ballot_w64(((hist >> (VIT_DW - 1 - (6 * G + 5))) & 1u) != 0u);
  r.m4 = __builtin_amdgcn_ballot_w64(((hist >> (VIT_DW - 1 - (6 * G + 4))) & 1u) != 0u);
  r.m3 = __builtin_amdgcn_ballot_w64(((hist >> (VIT_DW - 1 - (6 * G + 3))) & 1u) != 0u);
  r.m2 = __builtin_amdgcn_ballot_w64(((hist >> (VIT_DW - 1 - (6 * G + 2))) & 1u) != 0u);
  r.m1 = __builtin_amdgcn_ballot_w64(((hist >> (VIT_DW - 1 - (6 * G + 1))) & 1u) != 0u);
  r.m0 = __builtin_amdgcn_ballot_w64(((hist >> (VIT_DW - 1 - (6 * G + 0))) & 1u) != 0u);
  return r;
}
__device__ __forceinline__ void vit_tb_chain(const VitMask6 &q, int &j, unsigned &acc)
{
  // After the six steps every bit of j has been rewritten once: bit p of j IS the decision of the step with exchange bit p, the
  // first one of the group (p = 0) being the oldest in `acc` order -- so the six output bits are bitreverse6(j), appended once
  // per group instead of one add-with-carry per step.
  int jc, js;
#define DABX_TB_STEP(M, BIT)                                                                                                    \
  "s_andn2_b32 %[jc], %[j], " #BIT "\n\ts_or_b32 %[js], %[j], " #BIT "\n\ts_bitcmp1_b64 %[" #M "], %[j]\n\t"                        \
  "s_cselect_b32 %[j], %[js], %[jc]\n\t"
  asm volatile(DABX_TB_STEP(m5, 1) DABX_TB_STEP(m4, 2) DABX_TB_STEP(m3, 4) DABX_TB_STEP(m2, 8) DABX_TB_STEP(m1, 16) DABX_TB_STEP(m0, 32)
               "s_brev_b32 %[jc], %[j]\n\ts_lshl_b32 %[acc], %[acc], 6\n\ts_lshr_b32 %[jc], %[jc], 26\n\ts_or_b32 %[acc], %[acc], %[jc]"
               : [jc] "=&s"(jc), [js] "=&s"(js), [acc] "+s"(acc), [j] "+s"(j)
               : [m5] "s"(q.m5), [m4] "s"(q.m4), [m3] "s"(q.m3), [m2] "s"(q.m2), [m1] "s"(q.m1), [m0] "s"(q.m0)
               : "scc");
#undef DABX_TB_STEP
}
// groups NG - 1 .. 0 of one word; `ahead` = the masks of group NG - 1 (made while the previous word's chain ran); returns the
// masks of the next word's top group (`nxt`, group 4) made while this word's last chain runs
template <int NG>
__device__ __forceinline__ VitMask6 vit_tb_word(unsigned hist, unsigned nxt, VitMask6 ahead, int &j, unsigned &acc)
{
  VitMask6 a = ahead;
  if constexpr (NG >= 5) { const VitMask6 b = vit_tb_masks<3>(hist); vit_tb_chain(a, j, acc); a = b; }
  if constexpr (NG >= 4) { const VitMask6 b = vit_tb_masks<2>(hist); vit_tb_chain(a, j, acc); a = b; }
  if constexpr (NG >= 3) { const VitMask6 b = vit_tb_masks<1>(hist); vit_tb_chain(a, j, acc); a = b; }
  if constexpr (NG >= 2) { const VitMask6 b = vit_tb_masks<0>(hist); vit_tb_chain(a, j, acc); a = b; }
  const VitMask6 b = vit_tb_masks<4>(nxt);
  vit_tb_chain(a, j, acc);
  return b;
}

__device__ __forceinline__ void vit_traceback(const uint32_t *dec, int nbits, int lane, uint32_t *raw)
{
  const int nsteps = __builtin_amdgcn_readfirstlane(nbits + 6);   // wave-uniform by contract; tell the compiler (j, acc live in SGPRs)
  int j = 0;                                   // lane of the terminal state 0
  int wi = (nsteps - 1) / VIT_DW;
  unsigned hist = dec[(size_t)wi * 64 + lane];
  VitMask6 ahead;
  {                                            // top (possibly partial) word
    const unsigned nxt = wi > 0 ? dec[(size_t)(wi - 1) * 64 + lane] : 0u;
    unsigned acc = 0;
    const int n_top = nsteps - wi * VIT_DW;    // 1 .. 30 steps
    if (n_top % 6 == 0) {                      // every legal DAB trellis (nbits + 6 is a multiple of 6): whole groups
      switch (n_top / 6) {
        case 5: ahead = vit_tb_word<5>(hist, nxt, vit_tb_masks<4>(hist), j, acc); break;
        case 4: ahead = vit_tb_word<4>(hist, nxt, vit_tb_masks<3>(hist), j, acc); break;
        case 3: ahead = vit_tb_word<3>(hist, nxt, vit_tb_masks<2>(hist), j, acc); break;
        case 2: ahead = vit_tb_word<2>(hist, nxt, vit_tb_masks<1>(hist), j, acc); break;
        default: ahead = vit_tb_word<1>(hist, nxt, vit_tb_masks<0>(hist), j, acc); break;
      }
    } else {                                   // arbitrary lengths of the stage-level entry point: step by step
      for (int s = n_top - 1; s >= 0; --s) {
        const unsigned hv = (unsigned)__builtin_amdgcn_readlane((int)hist, __builtin_amdgcn_readfirstlane(j));
        const unsigned k = (hv >> (VIT_DW - 1 - s)) & 1u;
        acc = (acc << 1) | k;
        const int p = (5 - (s % 6)) % 6;
        j = (j & ~(1 << p)) | ((int)k << p);
      }
      j = __builtin_amdgcn_readfirstlane(j);
      acc = __builtin_amdgcn_readfirstlane(acc);
      ahead = vit_tb_masks<4>(nxt);
    }
    if (lane == 0) { raw[wi] = acc; raw[wi + 1] = 0; raw[wi + 2] = 0; }
    hist = nxt;
    --wi;
  }
  for (; wi >= 0; --wi) {
    const unsigned nxt = wi > 0 ? dec[(size_t)(wi - 1) * 64 + lane] : 0u;
    unsigned acc = 0;
    ahead = vit_tb_word<5>(hist, nxt, ahead, j, acc);
    if (lane == 0) raw[wi] = acc;
    hist = nxt;
  }
}

// raw[] (bit b of word g = decision of trellis step 30 g + b = decoded bit q = 30 g + b - 6) -> output word w:
// bits q = 32 w .. 32 w + 31, first bit in the MSB of the first byte (FicDecoder / Mp4Processor byte packing).
__device__ __forceinline__ uint32_t vit_output_word(const uint32_t *raw, int w)
{
  const int t0 = 32 * w + 6, g0 = t0 / VIT_DW, b0 = t0 - g0 * VIT_DW;
  unsigned long long v = (unsigned long long)raw[g0] | ((unsigned long long)raw[g0 + 1] << 30);
  v >>= b0;
  v |= (unsigned long long)raw[g0 + 2] << (60 - b0);
  const uint32_t bits = (uint32_t)v;                 // bit i = decoded bit 32 w + i
  return __builtin_bswap32(__builtin_bitreverse32(bits));
}

}  // namespace dabx
