// vit_t.hip -- MSC decode, throughput formulation: one LANE per trellis (64 trellises per wavefront).
//
// k_msc_prep  : time de-interleave (backend.cpp:131-139) of the pending CIFs of every (stream, sub-channel) job,
//               read from the planar TDI ring, written TRANSPOSED (inT[group][dword][lane]) so that the decoder's
//               per-lane reads are coalesced.  Pure byte movement: 4 dword loads + 8 v_perm + 4 dword stores
//               per 16 soft symbols.
// k_msc_vitT  : depuncture (wave-uniform map, protection.cpp:46-59) + Viterbi (viterbi_spiral.cpp:95-126,
//               scalar tie rule) + PRBS + byte packing for 64 jobs per wave.  The 64 path metrics of a trellis
//               live in 32 VGPRs as packed int16 pairs; butterflies are v_pk_add/sub/min on register pairs in
//               place (tools/gen_vit_t.py, vit_t_gen.h); ~3 VALU per trellis step instead of ~8 for the
//               wave-per-trellis kernel (viterbi_core.h), which remains the path for small batches.
// Jobs are grouped in CLASSES of equal protection profile across all streams (pipeline.h, MscClass): one launch
// covers every class, a block finds its class from its group index, so ensembles with different and mixed
// sub-channel layouts are decoded together.
#include "pipeline.h"
#include "vit_t_gen.h"
#include <vector>

namespace dabx {

__device__ __forceinline__ int bitrev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }

// ---------------------------------------------------------------------------------------------------- prepare
// grid = ceil(jobs / PJB), 256 threads; a block prepares PJB = 32 consecutive jobs (half of a decoder wave).
// Per chunk of PCH = 64 ring positions: (1) every (job, plane) run of 64 bytes -- exactly one HBM line -- is read with
// lanes ALONG the run into LDS (32-byte pieces made the kernel fetch every line twice to four times: the two halves
// were a whole chunk iteration apart, too far for the L2); (2) thread = (job, plane group, half of the chunk):
// 4 planes x 1 dword -> 4x4 byte transpose in registers (8 v_perm) -> 4 idx-ordered dwords, stored to inT[q][lane]
// (the 32 jobs are 32 adjacent lanes: 128-B pieces of a row).
constexpr int PJB = 32;                       // jobs per block
constexpr int PCH = 64;                       // positions per chunk (bytes per plane run)
constexpr int PJS = 16 * PCH + 4;             // LDS job stride in bytes (+4: conflict-free ds_read across jobs); 33 KB per block
__device__ __forceinline__ int msc_class_of_group(const MscLaunch &L, int g)
{
  int c = 0;
  while (c + 1 < L.n && g >= L.c[c + 1].g0) c++;
  return c;
}

// cache hints of k_msc_prep's streams (pipeline.h, DABX_PREP_NT; both off: measured slower)
__device__ __forceinline__ uint32_t prep_ld(const uint32_t *p)
{
#if DABX_PREP_NT & 1
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void prep_st(uint32_t *p, uint32_t v)
{
#if DABX_PREP_NT & 2
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
constexpr int PREP_STG = PJB * 16 * (PCH / 4) / 256;             // (job, plane, dword) items per thread and chunk (32)
__device__ __forceinline__ void prep_item(int it, int cw, bool full, int &d, int &pl, int &job)
{
  int jp;
  if (full) { d = it & (PCH / 4 - 1); jp = it / (PCH / 4); }       // shifts/masks for whole chunks
  else { d = it % cw; jp = it / cw; }
  pl = jp & 15; job = jp >> 4;
}
// requests this thread's items of the chunk at ring position p0 (cw dwords per run) into registers
__device__ __forceinline__ void prep_request(uint32_t (&stage)[PREP_STG], int tid, int p0, int cw, const uint8_t *const *s_base, const long long *s_r,
                                             const int2 *s_mv /* per job: {CIFs m < x lie before the sub-channel's move, byte offset of its old address} */)
{
  const bool full = cw == PCH / 4;
  if (full) {
    // whole chunk: item r of thread tid is (job r, plane (tid >> 4) & 15, dword tid & 15) -- tid + 256 r split by 16 and 16 --
    // so the plane's part of the address is per-thread constant and only the job's base changes with r
    const int pl = (tid >> 4) & 15, d = tid & 15;
    const int brev = bitrev4(pl);
    const unsigned in_plane = (unsigned)pl * (CIF_BITS / 16) + (unsigned)p0 + 4u * (unsigned)d;
#pragma unroll
    for (int r = 0; r < PREP_STG; r++) {
      const uint8_t *base = s_base[r];
      uint32_t v = 0x7F7F7F7Fu;
      if (base) {
        const unsigned slot = (unsigned)(s_r[r] - 16 + brev) & (TDI_SLOTS - 1);      // out_r[idx] = in_{r-16+map[idx&15]}[idx], backend.cpp:129
        const int2 mv = s_mv[r];
        v = prep_ld(reinterpret_cast<const uint32_t *>(base + (brev < mv.x ? mv.y : 0) + (size_t)slot * CIF_BITS + in_plane));
      }
      stage[r] = v;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < PREP_STG; r++) {
    const int it = tid + 256 * r;
    uint32_t v = 0x7F7F7F7Fu;
    if (it < PJB * 16 * cw) {
      int d, pl, job;
      prep_item(it, cw, full, d, pl, job);
      const uint8_t *base = s_base[job];
      if (base) {
        // out_r[idx] = in_{r-16+map[idx&15]}[idx], map = 4-bit reversal (backend.cpp:129); planar ring: plane = idx & 15
        const long long cif = s_r[job] - 16 + bitrev4(pl);
        const int2 mv = s_mv[job];
        v = prep_ld(reinterpret_cast<const uint32_t *>(base + (bitrev4(pl) < mv.x ? mv.y : 0) + (size_t)(cif & (TDI_SLOTS - 1)) * CIF_BITS + (size_t)pl * (CIF_BITS / 16) + p0 + 4 * d));
      }
    }
    stage[r] = v;
  }
}

__global__ __launch_bounds__(256) void k_msc_prep(EngineDev e, int cifs, MscLaunch L)
{
  __builtin_amdgcn_s_setprio(3);      // above the decoder whose input it prepares (pipeline.hip, front_prio)
  __shared__ __attribute__((aligned(16))) uint8_t tile[PJB * PJS];
  __shared__ const uint8_t *s_base[PJB];     // per job: stream ring + cu_start*4 (nullptr = invalid job)
  __shared__ long long s_r[PJB];
  __shared__ int2 s_mv[PJB];
  const int ci = msc_class_of_group(L, blockIdx.x / (64 / PJB));
  const MscLaunchCls &cl = L.c[ci];
  const int n_in = cl.n_in;
  uint32_t *inT = cl.inT;
  const int tid = threadIdx.x, job0 = (blockIdx.x - cl.g0 * (64 / PJB)) * PJB;      // job within the class
  if (tid < PJB) {
    const MscJob q = msc_class_job(e, cl, job0 + tid, cifs);
    const SubchDev &sc = e.subch[(size_t)q.s * e.max_subch + q.j];
    s_base[tid] = q.valid ? e.tdi + (size_t)q.s * TDI_SLOTS * CIF_BITS + sc.cu_start * 4 : nullptr;
    s_r[tid] = q.r;
    s_mv[tid] = make_int2(q.valid ? msc_move_thr(sc, q.r) : 0, (sc.prev_cu_start - sc.cu_start) * 4);
  }
  const int rows = n_in / 4 + 1;
  const int jl = tid & (PJB - 1), pg = (tid >> 5) & 3, qd = tid >> 7;      // job in block, plane group, half of the chunk
  uint32_t *dst = inT + (size_t)(job0 >> 6) * rows * 64 + (job0 & 63) + jl;
  if (pg == 0 && qd == 0) dst[(size_t)(rows - 1) * 64] = 0x7F7F7F7Fu;      // punctured soft bit = 0 -> symbol 127
  __syncthreads();
  const int npos = n_in / 16;                                      // ring positions per plane of one job
  // Software pipeline over the chunks: the (job, plane, dword) items of chunk c + 1 are requested into registers before
  // chunk c is transposed and stored, so the ring reads (64-byte runs, one HBM line each: long latency, little to
  // coalesce) are in flight behind the LDS reads, permutes and stores of the chunk before instead of in front of a barrier.
  uint32_t stage[PREP_STG];
  auto chunk_dwords = [&](int p0) { return (npos - p0 < PCH ? npos - p0 : PCH) / 4; };     // npos % 4 == 0
  if (npos > 0) prep_request(stage, tid, 0, chunk_dwords(0), s_base, s_r, s_mv);
  for (int p0 = 0; p0 < npos; p0 += PCH) {
    const int cw = chunk_dwords(p0);                               // dwords per run in this chunk
    const bool full = cw == PCH / 4;
    // (1) this chunk's items into the LDS tile
    if (full) {                                                     // item r = (job r, this thread's plane and dword): see prep_request
      uint8_t *mine_w = tile + ((tid >> 4) & 15) * PCH + 4 * (tid & 15);
#pragma unroll
      for (int r = 0; r < PREP_STG; r++) *reinterpret_cast<uint32_t *>(mine_w + r * PJS) = stage[r];
    } else {
#pragma unroll
      for (int r = 0; r < PREP_STG; r++) {
        const int it = tid + 256 * r;
        if (it < PJB * 16 * cw) {
          int d, pl, job;
          prep_item(it, cw, full, d, pl, job);
          *reinterpret_cast<uint32_t *>(tile + job * PJS + pl * PCH + 4 * d) = stage[r];
        }
      }
    }
    __syncthreads();
    if (p0 + PCH < npos) prep_request(stage, tid, p0 + PCH, chunk_dwords(p0 + PCH), s_base, s_r, s_mv);
    // (2) transpose 4 planes x 4 positions, write idx-ordered dwords
    const uint8_t *mine = tile + jl * PJS + (4 * pg) * PCH;
    for (int d = qd; d < cw; d += 2) {
      const uint32_t m0 = *reinterpret_cast<const uint32_t *>(mine + 0 * PCH + 4 * d), m1 = *reinterpret_cast<const uint32_t *>(mine + 1 * PCH + 4 * d);
      const uint32_t m2 = *reinterpret_cast<const uint32_t *>(mine + 2 * PCH + 4 * d), m3 = *reinterpret_cast<const uint32_t *>(mine + 3 * PCH + 4 * d);
      const uint32_t a = __builtin_amdgcn_perm(m1, m0, 0x05010400u), b = __builtin_amdgcn_perm(m1, m0, 0x07030602u);
      const uint32_t c = __builtin_amdgcn_perm(m3, m2, 0x05010400u), f = __builtin_amdgcn_perm(m3, m2, 0x07030602u);
      const int P = p0 + 4 * d;                                    // position P + k holds idx = 16 (P + k) + plane: dword 4 (P + k) + pg
      prep_st(&dst[(size_t)(4 * (P + 0) + pg) * 64], __builtin_amdgcn_perm(c, a, 0x05040100u));
      prep_st(&dst[(size_t)(4 * (P + 1) + pg) * 64], __builtin_amdgcn_perm(c, a, 0x07060302u));
      prep_st(&dst[(size_t)(4 * (P + 2) + pg) * 64], __builtin_amdgcn_perm(f, b, 0x05040100u));
      prep_st(&dst[(size_t)(4 * (P + 3) + pg) * 64], __builtin_amdgcn_perm(f, b, 0x07060302u));
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------- decode
// Symbols of one 6-step cycle: 24 transposed dwords (one coalesced load each; the index is wave-uniform) plus the
// byte lane of every symbol (2 bits each).  Fetched one cycle ahead so no memory latency sits on the ACS chain.
struct VtCycle { uint32_t w[24]; unsigned long long sh; };

// The depuncture map is written once by the host (engine.cpp, build_msc_classes) and never by a kernel: read through the
// CONSTANT address space its entries come in over the scalar cache (s_load_dwordx2 per trellis step, lgkmcnt) instead of
// as vector loads + v_readfirstlane, which sat in the same in-order vmcnt queue as the symbol loads and decision stores and
// forced a full s_waitcnt vmcnt(0) round trip per 6-step cycle.
typedef unsigned vt_u2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) vt_u2 *vt_cmap;

// in_grp: buffer resource over the group's transposed symbols [row][64 lanes]: the row offset is wave-uniform and goes
// into the instruction's scalar offset, the lane is the only vector part of the address (no 64-bit VALU address math)
typedef __amdgpu_buffer_rsrc_t vt_rsrc;
__device__ __forceinline__ vt_rsrc vt_make_rsrc(const uint32_t *base, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(base), 0, (int)bytes, 0x00020000);   // raw buffer, dword format (gfx9 / CDNA)
}
// The lane's column of the group's decision words.  Default: a per-lane pointer, `nt` stores and loads (pipeline.h, DABX_VIT_NT).  Experiment builds
// (-DDABX_VIT_BUF=1 -DDABX_VIT_ST_AUX=a -DDABX_VIT_LD_AUX=b): raw-buffer accesses over the group's words -- the step's row is a
// scalar offset -- with the cache-policy bits free to choose: aux 0 plain, 1 sc0, 2 nt, 16 sc1 (write-through / L2 bypass), 17 sc0 sc1.
struct VtDec {
  uint2 *p;
#if DABX_VIT_BUF
  vt_rsrc rs;
  int voff;
#endif
};
__device__ __forceinline__ VtDec vt_make_dec(uint2 *group_base, int nsteps, int lane)
{
  VtDec d;
  d.p = group_base + lane;
#if DABX_VIT_BUF
  d.rs = vt_make_rsrc(reinterpret_cast<const uint32_t *>(group_base), (unsigned)nsteps * 512u);
  d.voff = lane * 8;
#else
  (void)nsteps;
#endif
  return d;
}
#ifndef DABX_VIT_ST_AUX
#define DABX_VIT_ST_AUX 2
#define DABX_VIT_LD_AUX 2
#endif
typedef unsigned vt_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void vt_dec_store(const VtDec &d, int t, unsigned acc0, unsigned acc1)
{
#if DABX_VIT_BUF
  vt_u2v v; v.x = acc0; v.y = acc1;
  __builtin_amdgcn_raw_buffer_store_b64(v, d.rs, d.voff, t * 512, DABX_VIT_ST_AUX);
#elif DABX_VIT_NT & 1        // decision words, written once and read once: past the caches (pipeline.h)
  vt_u2v v; v.x = acc0; v.y = acc1;
  __builtin_nontemporal_store(v, reinterpret_cast<vt_u2v *>(&d.p[(size_t)t * 64]));
#else
  d.p[(size_t)t * 64] = make_uint2(acc0, acc1);
#endif
}
__device__ __forceinline__ uint2 vt_dec_load(const VtDec &d, int t)
{
#if DABX_VIT_BUF
  const vt_u2v v = __builtin_amdgcn_raw_buffer_load_b64(d.rs, d.voff, t * 512, DABX_VIT_LD_AUX);
  return make_uint2(v.x, v.y);
#elif DABX_VIT_NT & 2
  const vt_u2v v = __builtin_nontemporal_load(reinterpret_cast<const vt_u2v *>(&d.p[(size_t)t * 64]));
  return make_uint2(v.x, v.y);
#else
  return d.p[(size_t)t * 64];
#endif
}
#if DABX_VIT_NT & 4          // (A/B builds: the transposed input past the caches; aux bit 1 = nt)
constexpr int VT_IN_AUX = 2;
#else
constexpr int VT_IN_AUX = 0;
#endif
__device__ __forceinline__ void vt_fetch(VtCycle &c, vt_rsrc in_grp, int lane, vt_cmap map, int t0)
{
  c.sh = 0;
#pragma unroll
  for (int s6 = 0; s6 < 6; s6++) {
    const vt_u2 m = map[t0 + s6];                                    // 4 x uint16 indices of step t0 + s6, wave-uniform
    const unsigned idx[4] = {m.x & 0xFFFFu, m.x >> 16, m.y & 0xFFFFu, m.y >> 16};
#pragma unroll
    for (int p = 0; p < 4; p++) {
      c.w[4 * s6 + p] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(in_grp, lane * 4, (int)((idx[p] >> 2) * 256u), VT_IN_AUX);
      c.sh |= (unsigned long long)(idx[p] & 3) << (2 * (4 * s6 + p));
    }
  }
}

// ---- the arithmetic of the reference's SIMD builds (cfg.viterbi_tie_mode 1 / 2) on the same register scheme -----------------
// VITERBI_AVX2 (viterbi_16way.h:9-110): uint16 metrics, saturating adds, a tie goes to predecessor i + 32, and after every
// second step `renormalize`: if state 0's metric of the step BEFORE exceeds 60000, the minimum of the new metrics is subtracted.
// VITERBI_SSE2 / NEON (viterbi_8way.h:9-120): int16 metrics saturating at 32767, threshold 30000, ties as in the scalar body.
// The lane keeps its relative, doubled and centred int16 metrics R and follows the reference's ABSOLUTE level with one int:
// 2 M_ref = R + C, Coff = C before the first step of the current 6-step cycle; C grows by 1020 per step and by `ref` at a
// re-centring, and becomes -min(R) at a renormalisation.  A candidate saturates when it exceeds lim = LIMTOP - C.  That can
// only happen if max(R) + 12240 > LIMTOP - Coff at the start of the cycle (metrics grow by at most 1020 per step while lim
// falls by 1020): only then does the cycle run the clamped step bodies (four more v_pk_min per butterfly pair).  Model and
// proof by test: tools/gen_vit_t.py (model_decode_tie against the oracle restatements and the reference's own object code).
template <int TIE> struct VtTie { static constexpr int LIMTOP = 0, REN2 = 0; };
template <> struct VtTie<1> { static constexpr int LIMTOP = 2 * 65535, REN2 = 2 * 60000; };
template <> struct VtTie<2> { static constexpr int LIMTOP = 2 * 32767, REN2 = 2 * 30000; };

template <int C, int TIE = 0, bool CLAMP = false>
__device__ __forceinline__ void vt_one(vt::s2 (&R)[32], const VtCycle &cy, int t, const VtDec &dec_lane, vt::s2 v2n, int &Coff)
{
  // the step's four symbols: dword + byte lane each (the lanes are wave-uniform and end up in SGPR selectors); the packed
  // branch metrics come straight from those (vit_t_gen.h, bm<C>): no scalar extraction, no 32-bit sums
  unsigned w[4], b[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    w[p] = cy.w[4 * C + p];
    b[p] = (unsigned)((cy.sh >> (2 * (4 * C + p))) & 3);
  }
  vt::s2 M[4];
  unsigned acc0, acc1;
  bool pre = false;
  vt::s2 lim = vt::pk(0, 0);
  if constexpr (TIE != 0) {
    if constexpr ((C & 1) != 0) pre = (int)R[0].x + Coff + 1020 * C > VtTie<TIE>::REN2;      // metrics2[0] > threshold, checked on the step before
    if constexpr (CLAMP) {
      int l = VtTie<TIE>::LIMTOP - Coff - 1020 * (C + 1);
      l = l < 32767 ? l : 32767;
      lim = vt::pk(l, l);
    }
  }
  if constexpr (C == 0) { vt::bm0(w, b, v2n, M); vt::step0<TIE, CLAMP>(R, M, acc0, acc1, lim); }
  else if constexpr (C == 1) { vt::bm1(w, b, v2n, M); vt::step1<TIE, CLAMP>(R, M, acc0, acc1, lim); }
  else if constexpr (C == 2) { vt::bm2(w, b, v2n, M); vt::step2<TIE, CLAMP>(R, M, acc0, acc1, lim); }
  else if constexpr (C == 3) { vt::bm3(w, b, v2n, M); vt::step3<TIE, CLAMP>(R, M, acc0, acc1, lim); }
  else if constexpr (C == 4) { vt::bm4(w, b, v2n, M); vt::step4<TIE, CLAMP>(R, M, acc0, acc1, lim); }
  else { vt::bm5(w, b, v2n, M); vt::step5<TIE, CLAMP>(R, M, acc0, acc1, lim); }
  vt_dec_store(dec_lane, t, acc0, acc1);
  if constexpr (TIE != 0 && (C & 1) != 0) {
    if (__builtin_amdgcn_ballot_w64(pre)) {              // some lane renormalises (wave-uniform branch; every ~100-200 steps per lane)
      const int mnv = vt::min64(R);
      if (pre) Coff = -mnv - 1020 * (C + 1);
    }
  }
}

template <int TIE, bool CLAMP>
__device__ __forceinline__ void vt_cycle6(vt::s2 (&R)[32], const VtCycle &cy, int t0, const VtDec &dec_lane, vt::s2 v2n, int &Coff)
{
  vt_one<0, TIE, CLAMP>(R, cy, t0 + 0, dec_lane, v2n, Coff);
  vt_one<1, TIE, CLAMP>(R, cy, t0 + 1, dec_lane, v2n, Coff);
  vt_one<2, TIE, CLAMP>(R, cy, t0 + 2, dec_lane, v2n, Coff);
  vt_one<3, TIE, CLAMP>(R, cy, t0 + 3, dec_lane, v2n, Coff);
  vt_one<4, TIE, CLAMP>(R, cy, t0 + 4, dec_lane, v2n, Coff);
  vt_one<5, TIE, CLAMP>(R, cy, t0 + 5, dec_lane, v2n, Coff);
}
template <int TIE, bool ALWAYS_CLAMP>
__device__ __forceinline__ void vt_cycle(vt::s2 (&R)[32], const VtCycle &cy, int t0, const VtDec &dec_lane, vt::s2 v2n, int &Coff)
{
  if constexpr (TIE == 0) vt_cycle6<0, false>(R, cy, t0, dec_lane, v2n, Coff);
  else {
    bool clamp = ALWAYS_CLAMP;
    if constexpr (!ALWAYS_CLAMP) {
      const int room = VtTie<TIE>::LIMTOP - Coff;
      // cheap filter first (the spread of the 64 metrics never exceeds 14 240 doubled units), then the exact test on the maximum
      if (__builtin_amdgcn_ballot_w64((int)R[0].x + 14280 + 12240 > room)) {
        const int mxv = vt::max64(R);
        clamp = __builtin_amdgcn_ballot_w64(mxv + 12240 > room) != 0;
      }
    }
    if (clamp) vt_cycle6<TIE, true>(R, cy, t0, dec_lane, v2n, Coff);
    else vt_cycle6<TIE, false>(R, cy, t0, dec_lane, v2n, Coff);
    Coff += 6 * 1020;
  }
}

// ---- chain-back helpers: one 6-step cycle of decision words in named registers
struct VtDec6 { uint2 w0, w1, w2, w3, w4, w5; };
__device__ __forceinline__ VtDec6 vt_load_dec(const VtDec &dec_lane, int t0)
{
  VtDec6 d;
  d.w0 = vt_dec_load(dec_lane, t0); d.w1 = vt_dec_load(dec_lane, t0 + 1); d.w2 = vt_dec_load(dec_lane, t0 + 2);
  d.w3 = vt_dec_load(dec_lane, t0 + 3); d.w4 = vt_dec_load(dec_lane, t0 + 4); d.w5 = vt_dec_load(dec_lane, t0 + 5);
  return d;
}
// step t of class C (viterbi_spiral.cpp:114-125 in label space): the decision of label L is bit pos_c[L] of the word; it
// replaces bit p = 5 - C of the label and is output bit t - 6 (PRBS, backend.cpp:155-158, and byte packing on the way out)
template <int C>
__device__ __forceinline__ void vt_back_one(const uint2 w, int t, unsigned &L, unsigned &outw, uint32_t *out, const uint32_t *prbs,
                                            const unsigned char *pos_c)
{
  constexpr unsigned p = 5 - C;                                   // VT_P[C]
  const unsigned pos = pos_c[L];
  const unsigned long long ww = ((unsigned long long)w.y << 32) | w.x;
  const unsigned bit = (unsigned)(ww >> pos) & 1u;
  const int qb = t - 6;
  outw |= bit << (((qb >> 3) & 3) * 8 + 7 - (qb & 7));
  L = (L & ~(1u << p)) | (bit << p);
  if ((qb & 31) == 0) {                                           // wave-uniform
    // PRBS word over the scalar cache (constant table): a vector load here would wait for vmcnt(0), i.e. drain the
    // prefetched decision words, every 32 steps
    const unsigned pw = ((const __attribute__((address_space(4))) uint32_t *)(const void *)prbs)[qb >> 5];
    if (out) out[qb >> 5] = outw ^ pw;
    outw = 0;
  }
}

// Diagnostic (tools/vit_timeline.py): when a buffer is registered through dabx_internal_set_vt_timeline every decoder wave
// records where it ran (HW_ID, XCC_ID) and when (100-MHz real-time counter: start, end of the forward pass, end).  Null
// in normal operation: one scalar load per wave.
__device__ unsigned long long *g_vt_timeline = nullptr;
__device__ unsigned g_vt_timeline_cap = 0;
__device__ unsigned g_vt_timeline_n = 0;

// Forward pass + chain-back of the 64 trellises of one wave.  in_grp: the group's transposed symbols; cmap: depuncture map
// with PUNCT remapped to the 0x7F row; dec_lane: this lane's column of the group's decision words; out: where the lane's
// packed, de-dispersed bytes go (nullptr: nothing is stored).
template <int TIE, bool ALWAYS_CLAMP>
__device__ __forceinline__ void vt_decode(vt_rsrc in_grp, vt_cmap cmap, int nsteps, const VtDec &dec_lane, uint32_t *out, const uint32_t *prbs,
                                          const unsigned char (*pos_tab)[64], int lane, unsigned long long &t_forward_end, bool want_time)
{
  vt::s2 R[32];
#pragma unroll
  for (int r = 0; r < 32; r++) R[r] = vt::pk(2000, 2000);          // viterbi_spiral.cpp:98-101 (0 / 1000), doubled
  R[0] = vt::pk(0, 2000);
  int Coff = 0;                                                    // tie modes: 2 M_ref = R + Coff before the cycle's first step
  // Two 6-step cycles of symbols are always in flight, fetched UNCONDITIONALLY one cycle ahead (beyond the end the last
  // cycle is fetched again): with no branch around a fetch the number of outstanding loads is static and the compiler
  // waits with s_waitcnt vmcnt(N) for exactly the cycle it is about to consume instead of draining the queue.
  const int last = nsteps - 6;                                     // nsteps is a multiple of 6, >= 12
  VtCycle ca, cb;
  vt_fetch(ca, in_grp, lane, cmap, 0);
  vt_fetch(cb, in_grp, lane, cmap, 6);
  auto recentre = [&]() {                                  // every 12 steps on the metric of label 0
    const vt::s2 ref = vt::pk(R[0].x, R[0].x);
    if constexpr (TIE != 0) Coff += (int)R[0].x;
#pragma unroll
    for (int r = 0; r < 32; r++) R[r] = R[r] - ref;
  };
  vt::s2 v2n;                                                      // (2, -2), pinned in a VGPR (VOP3P takes no literal on gfx9)
  asm volatile("v_mov_b32 %0, %1" : "=v"(v2n) : "s"(0xFFFE0002u));
  int t = 0;
  for (; t + 12 <= nsteps; t += 12) {
    recentre();
    vt_cycle<TIE, ALWAYS_CLAMP>(R, ca, t, dec_lane, v2n, Coff);
    vt_fetch(ca, in_grp, lane, cmap, t + 12 < last ? t + 12 : last);
    vt_cycle<TIE, ALWAYS_CLAMP>(R, cb, t + 6, dec_lane, v2n, Coff);
    vt_fetch(cb, in_grp, lane, cmap, t + 18 < last ? t + 18 : last);
  }
  if (t < nsteps) {                                                // odd number of cycles
    recentre();
    vt_cycle<TIE, ALWAYS_CLAMP>(R, ca, t, dec_lane, v2n, Coff);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (want_time) t_forward_end = __builtin_amdgcn_s_memrealtime();

  // chain-back per lane (viterbi_spiral.cpp:114-125 in label space) + PRBS (backend.cpp:155-158) + byte packing.
  // The chain is one LDS look-up (bit position of the label's decision in the step's word) and three VALU operations per
  // step and lane; the decision words themselves are fetched two 6-step cycles ahead of their use -- their addresses do not
  // depend on the path, only the bit that is picked does -- so no memory latency sits on the chain.  (Kept in NAMED
  // registers: as arrays indexed through a lambda the compiler had put one set in scratch and one in LDS, and the chain
  // then ran at ~1150 cycles per step, a third of the kernel.)
  unsigned L = 0, outw = 0;
  VtDec6 cur = vt_load_dec(dec_lane, nsteps - 6);
  VtDec6 nx1 = vt_load_dec(dec_lane, nsteps - 12);                // nsteps >= 18 for every legal profile (24 * 8 + 6 = 198 at least)
  for (int tc = nsteps - 6; tc >= 6; tc -= 6) {
    const int tf = tc - 12 > 6 ? tc - 12 : 6;                      // beyond the start: fetch a valid cycle again (never consumed)
    const VtDec6 nx2 = vt_load_dec(dec_lane, tf);
    vt_back_one<5>(cur.w5, tc + 5, L, outw, out, prbs, pos_tab[5]);
    vt_back_one<4>(cur.w4, tc + 4, L, outw, out, prbs, pos_tab[4]);
    vt_back_one<3>(cur.w3, tc + 3, L, outw, out, prbs, pos_tab[3]);
    vt_back_one<2>(cur.w2, tc + 2, L, outw, out, prbs, pos_tab[2]);
    vt_back_one<1>(cur.w1, tc + 1, L, outw, out, prbs, pos_tab[1]);
    vt_back_one<0>(cur.w0, tc + 0, L, outw, out, prbs, pos_tab[0]);
    cur = nx1; nx1 = nx2;
  }
}

// grid = groups, 64 threads.  TIE: cfg.viterbi_tie_mode (0 canonical, 1 VITERBI_AVX2, 2 VITERBI_SSE2 arithmetic).
template <int TIE>
__device__ __forceinline__ void msc_vitT_body(const EngineDev &e, int cifs, const MscLaunch &ML, const uint32_t *prbs)
{
  __shared__ unsigned char pos_tab[6][64];
  const int lane = threadIdx.x;
  unsigned long long *const tl = g_vt_timeline;
  unsigned long long tl0 = 0, tl1 = 0;
  if (tl) tl0 = __builtin_amdgcn_s_memrealtime();
  // Groups are ordered longest trellis first.  All waves of a launch are resident at once (<= 4 per SIMD), so the work of
  // a SIMD is the sum over the ~4 "rounds" of 1024 blocks that landed on it: walk every second round backwards
  // (boustrophedon) so that long and short trellises pair up on the same SIMD.
  for (int i = lane; i < 6 * 64; i += 64) pos_tab[i / 64][i % 64] = vt::VT_POS[i / 64][i % 64];
  int gg = blockIdx.x;
  {
    constexpr int ROUND = 1024;                                    // 256 CUs x 4 SIMDs
    const int r = gg / ROUND, base = r * ROUND;
    const int len = ML.groups - base < ROUND ? ML.groups - base : ROUND;
    if (r & 1) gg = base + (len - 1 - (gg - base));
  }
  const MscLaunchCls &cl = ML.c[msc_class_of_group(ML, gg)];
  const int g = gg - cl.g0;                                       // decoder group within the class
  const MscJob q = msc_class_job(e, cl, g * 64 + lane, cifs);
  const int nsteps = cl.nbits + 6, rows = cl.n_in / 4 + 1;
  const vt_rsrc in_grp = vt_make_rsrc(cl.inT + (size_t)g * rows * 64, (unsigned)rows * 256u);
  const VtDec dec_lane = vt_make_dec(cl.decT + (size_t)g * nsteps * 64, nsteps, lane);
  const vt_cmap cmap = (vt_cmap)(const void *)cl.map2;
  uint32_t *out = nullptr;
  if (q.valid)
    out = reinterpret_cast<uint32_t *>(e.msc_out + (((size_t)q.s * e.max_subch + q.j) * MSC_SLOTS + (size_t)(q.out_idx % MSC_SLOTS)) * e.msc_stride);
  // One wave per group, all <= 4 per SIMD resident at once.  Persistent waves at 2 or 3 per SIMD (a grid capped at 2048 / 3072
  // blocks, each walking several groups) so that the front end of the following frames could co-reside were measured in round 3:
  // -9 % / -3.5 % on the chain (profiles/r03_ab/ab4_persistent_decoder_grid_cap2048_cap3072.txt): the decoder alone gets 35-50 %
  // slower and nothing that moves in next to it pays that back.
  vt_decode<TIE, false>(in_grp, cmap, nsteps, dec_lane, out, prbs, pos_tab, lane, tl1, tl != nullptr);
  if (tl && lane == 0) {
    const unsigned slot = atomicAdd(&g_vt_timeline_n, 1u);
    if (slot < g_vt_timeline_cap) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long *o = tl + (size_t)slot * 4;
      o[0] = ((unsigned long long)xcc << 32) | hw; o[1] = tl0; o[2] = tl1; o[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
}
#ifdef DABX_VIT_WAVES                 // experiment builds only (tools/build_variant.sh -DDABX_VIT_WAVES=5 -DDABX_MSC_BATCH=9): ask for a fifth wave per SIMD
#define DABX_VIT_OCC __attribute__((amdgpu_waves_per_eu(DABX_VIT_WAVES, DABX_VIT_WAVES)))
#else
#define DABX_VIT_OCC
#endif
__global__ __launch_bounds__(64) DABX_VIT_OCC void k_msc_vitT(EngineDev e, int cifs, MscLaunch ML, const uint32_t *prbs) { msc_vitT_body<0>(e, cifs, ML, prbs); }
// the same trellises decoded with the arithmetic of the reference's VITERBI_AVX2 / VITERBI_SSE2 builds (cfg.viterbi_tie_mode)
__global__ __launch_bounds__(64) void k_msc_vitT_avx2(EngineDev e, int cifs, MscLaunch ML, const uint32_t *prbs) { msc_vitT_body<1>(e, cifs, ML, prbs); }
__global__ __launch_bounds__(64) void k_msc_vitT_sse2(EngineDev e, int cifs, MscLaunch ML, const uint32_t *prbs) { msc_vitT_body<2>(e, cifs, ML, prbs); }

// ---- stage-level access to the lane-per-trellis decoder (tests): batch trellises of nbits decoded bits, unpunctured ------------
// symT: [groups][nsteps + 1][64] transposed symbol dwords (row r = mother-code bits 4 r .. 4 r + 3 of the lane's trellis),
// map: identity (4 t, 4 t + 1, 4 t + 2, 4 t + 3), out: [groups * 64][nbits / 32] packed words, zero PRBS.
template <int TIE, bool ALWAYS_CLAMP>
__global__ __launch_bounds__(64) void k_vitT_stage(const uint32_t *symT, const uint16_t *map, int nbits, uint2 *decT, uint32_t *outw, const uint32_t *zeros)
{
  __shared__ unsigned char pos_tab[6][64];
  const int lane = threadIdx.x, g = blockIdx.x;
  for (int i = lane; i < 6 * 64; i += 64) pos_tab[i / 64][i % 64] = vt::VT_POS[i / 64][i % 64];
  const int nsteps = nbits + 6, rows = nsteps + 1;
  const vt_rsrc in_grp = vt_make_rsrc(symT + (size_t)g * rows * 64, (unsigned)rows * 256u);
  unsigned long long unused = 0;
  vt_decode<TIE, ALWAYS_CLAMP>(in_grp, (vt_cmap)(const void *)map, nsteps, vt_make_dec(decT + (size_t)g * nsteps * 64, nsteps, lane),
                               outw + ((size_t)g * 64 + lane) * (nbits / 32), zeros, pos_tab, lane, unused, false);
}

// not part of include/dabx.h (tests/test_gpu_viterbi.py): ViterbiSpiral::deconvolve on the lane-per-trellis kernel.
// soft: batch x 4 (nbits + 6) int16, bits: batch x nbits (one per byte); nbits a multiple of 96; always_clamp != 0 forces the
// saturating step bodies in every cycle (tie modes: must not change a bit).
extern "C" int dabx_internal_vitT(const int16_t *soft, int nbits, int batch, int tie_mode, int always_clamp, uint8_t *bits)
{
  if (!soft || !bits || batch <= 0 || nbits < 96 || nbits % 96 != 0 || 4 * (nbits + 6) > 65532 || tie_mode < 0 || tie_mode > 2) {
    set_error("dabx_internal_vitT: bad argument");
    return DABX_E_ARG;
  }
  const int nsteps = nbits + 6, rows = nsteps + 1, groups = (batch + 63) / 64, nw = nbits / 32;
  std::vector<uint32_t> symT((size_t)groups * rows * 64, 0x7F7F7F7Fu);
  for (int b = 0; b < batch; b++)
    for (int r = 0; r < nsteps; r++) {
      uint32_t v = 0;
      for (int k = 0; k < 4; k++) {
        const int16_t sft = soft[(size_t)b * 4 * nsteps + 4 * r + k];
        int sy = tie_mode ? (int)sft + 127 : (int)(int16_t)(sft + 127);          // viterbi_16way.h:73-76 saturates, viterbi_scalar.h:34-40 wraps
        sy = sy < 0 ? 0 : (sy > 255 ? 255 : sy);
        v |= (uint32_t)sy << (8 * k);
      }
      symT[((size_t)(b / 64) * rows + r) * 64 + (b % 64)] = v;
    }
  std::vector<uint16_t> map((size_t)4 * nsteps + 64);
  for (size_t i = 0; i < map.size(); i++) map[i] = (uint16_t)(i < (size_t)4 * nsteps ? i : 4 * nsteps);
  uint32_t *d_sym = nullptr, *d_out = nullptr, *d_zero = nullptr;
  uint16_t *d_map = nullptr;
  uint2 *d_dec = nullptr;
  int rc = DABX_E_HIP;
  std::vector<uint32_t> outw((size_t)groups * 64 * nw);
  do {
    if (hipMalloc(&d_sym, symT.size() * 4) != hipSuccess || hipMalloc(&d_map, map.size() * 2) != hipSuccess ||
        hipMalloc(&d_dec, (size_t)groups * nsteps * 64 * sizeof(uint2)) != hipSuccess || hipMalloc(&d_out, outw.size() * 4) != hipSuccess ||
        hipMalloc(&d_zero, (size_t)(nw + 1) * 4) != hipSuccess) break;
    if (hipMemcpy(d_sym, symT.data(), symT.size() * 4, hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_map, map.data(), map.size() * 2, hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemset(d_zero, 0, (size_t)(nw + 1) * 4) != hipSuccess || hipMemset(d_out, 0, outw.size() * 4) != hipSuccess) break;
#define VT_STAGE(T, A) hipLaunchKernelGGL((k_vitT_stage<T, A>), dim3(groups), dim3(64), 0, 0, d_sym, d_map, nbits, d_dec, d_out, d_zero)
    if (tie_mode == 0) VT_STAGE(0, false);
    else if (tie_mode == 1) { if (always_clamp) VT_STAGE(1, true); else VT_STAGE(1, false); }
    else { if (always_clamp) VT_STAGE(2, true); else VT_STAGE(2, false); }
#undef VT_STAGE
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
    if (hipMemcpy(outw.data(), d_out, outw.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
    rc = 0;
  } while (false);
  (void)hipFree(d_sym); (void)hipFree(d_map); (void)hipFree(d_dec); (void)hipFree(d_out); (void)hipFree(d_zero);
  if (rc) { set_error("dabx_internal_vitT: HIP error"); return rc; }
  for (int b = 0; b < batch; b++) {
    const uint8_t *pb = reinterpret_cast<const uint8_t *>(outw.data() + (size_t)b * nw);
    for (int i = 0; i < nbits; i++) bits[(size_t)b * nbits + i] = (uint8_t)((pb[i >> 3] >> (7 - (i & 7))) & 1);
  }
  return 0;
}

// not part of include/dabx.h: registers (or clears, buf == nullptr) the diagnostic wave-timeline buffer of k_msc_vitT
extern "C" int dabx_internal_set_vt_timeline(void *buf, unsigned capacity_records)
{
  unsigned long long *p = reinterpret_cast<unsigned long long *>(buf);
  unsigned zero = 0;
  DABX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_vt_timeline_cap), &capacity_records, sizeof(unsigned)));
  DABX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_vt_timeline_n), &zero, sizeof(unsigned)));
  DABX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_vt_timeline), &p, sizeof(p)));
  return 0;
}

// ---------------------------------------------------------------------------------------------------- launch
// prep on stream a (reads the TDI ring before the front end moves on), decode on stream b.
int launch_msc_prep(const EngineDev &e, int cifs, const MscLaunch &L, hipStream_t st, Marker &mk)
{
  mk.begin(6, st);
  hipLaunchKernelGGL(k_msc_prep, dim3(L.groups * (64 / PJB)), dim3(256), 0, st, e, cifs, L);
  mk.end(6, st);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_msc_vitT(const EngineDev &e, int cifs, const MscLaunch &L, hipStream_t st, Marker &mk)
{
  const DevTables *t;
  int rc = get_tables(&t);
  if (rc) return rc;
  mk.begin(7, st);
  const int grid = L.groups;
  if (e.tie_mode == 1) hipLaunchKernelGGL(k_msc_vitT_avx2, dim3(grid), dim3(64), 0, st, e, cifs, L, t->prbs_words);
  else if (e.tie_mode == 2) hipLaunchKernelGGL(k_msc_vitT_sse2, dim3(grid), dim3(64), 0, st, e, cifs, L, t->prbs_words);
  else hipLaunchKernelGGL(k_msc_vitT, dim3(grid), dim3(64), 0, st, e, cifs, L, t->prbs_words);
  mk.end(7, st);
  DABX_HIP(hipGetLastError());
  return 0;
}

}  // namespace dabx
