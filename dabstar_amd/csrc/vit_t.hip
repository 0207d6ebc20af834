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
//               wave-per-trellis kernel (viterbi_core.h), which remains the path for small or mixed batches.
#include "pipeline.h"
#include "vit_t_gen.h"

namespace dabx {

__device__ __forceinline__ int bitrev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }

// ---------------------------------------------------------------------------------------------------- prepare
// grid = groups, 256 threads: lane = job within the group, pg = plane group (planes 4pg .. 4pg+3).
__global__ __launch_bounds__(256) void k_msc_prep(EngineDev e, int cifs, int n_in, uint32_t *inT)
{
  const int g = blockIdx.x, lane = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const MscJob q = msc_job(e, g * 64 + lane, cifs);
  const int rows = n_in / 4 + 1;
  uint32_t *dst = inT + (size_t)g * rows * 64 + lane;
  if (pg == 0) dst[(size_t)(rows - 1) * 64] = 0x7F7F7F7Fu;        // punctured soft bit = 0 -> symbol 127
  if (!q.valid) return;
  const SubchDev &sc = e.subch[(size_t)q.s * e.max_subch + q.j];
  const uint8_t *tdi = e.tdi + (size_t)q.s * TDI_SLOTS * CIF_BITS;
  const uint32_t *src[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int p = 4 * pg + i;
    // out_r[idx] = in_{r-16+map[idx&15]}[idx], map = 4-bit reversal (backend.cpp:129); planar ring: plane = idx & 15
    const long long cif = q.r - 16 + bitrev4(p);
    src[i] = reinterpret_cast<const uint32_t *>(tdi + (size_t)(cif & (TDI_SLOTS - 1)) * CIF_BITS + (size_t)p * (CIF_BITS / 16) + sc.cu_start * 4);
  }
  const int nd = n_in / 64;                                        // dwords per plane
  for (int d = 0; d < nd; d++) {
    const uint32_t m0 = src[0][d], m1 = src[1][d], m2 = src[2][d], m3 = src[3][d];
    const uint32_t a = __builtin_amdgcn_perm(m1, m0, 0x05010400u), b = __builtin_amdgcn_perm(m1, m0, 0x07030602u);
    const uint32_t c = __builtin_amdgcn_perm(m3, m2, 0x05010400u), f = __builtin_amdgcn_perm(m3, m2, 0x07030602u);
    // position P = 4 d + k holds idx = 16 P + plane: dword q = idx / 4 = 4 P + pg
    dst[(size_t)(4 * (4 * d + 0) + pg) * 64] = __builtin_amdgcn_perm(c, a, 0x05040100u);
    dst[(size_t)(4 * (4 * d + 1) + pg) * 64] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    dst[(size_t)(4 * (4 * d + 2) + pg) * 64] = __builtin_amdgcn_perm(f, b, 0x05040100u);
    dst[(size_t)(4 * (4 * d + 3) + pg) * 64] = __builtin_amdgcn_perm(f, b, 0x07060302u);
  }
}

// ---------------------------------------------------------------------------------------------------- decode
__device__ __forceinline__ int vt_sym(const uint32_t *in_lane, unsigned idx)
{
  const uint32_t w = in_lane[(size_t)(idx >> 2) * 64];
  return 2 * (int)((w >> ((idx & 3) * 8)) & 0xFFu) - 255;
}

template <int C>
__device__ __forceinline__ void vt_one(vt::s2 (&R)[32], const uint32_t *in_lane, const uint16_t *map, int t, uint2 *dec_lane)
{
  const ushort4 m = *reinterpret_cast<const ushort4 *>(map + 4 * t);   // wave-uniform
  const int x0 = vt_sym(in_lane, m.x), x1 = vt_sym(in_lane, m.y), x2 = vt_sym(in_lane, m.z), x3 = vt_sym(in_lane, m.w);
  const int y0 = x0 + x3, a1 = y0 + x1, a2 = y0 - x1;
  int W[8];
  W[0] = a1 + x2; W[1] = a1 - x2; W[2] = a2 + x2; W[3] = a2 - x2;
  W[4] = -W[3]; W[5] = -W[2]; W[6] = -W[1]; W[7] = -W[0];
  unsigned acc0, acc1;
  if constexpr (C == 0) vt::step0(R, W, acc0, acc1);
  else if constexpr (C == 1) vt::step1(R, W, acc0, acc1);
  else if constexpr (C == 2) vt::step2(R, W, acc0, acc1);
  else if constexpr (C == 3) vt::step3(R, W, acc0, acc1);
  else if constexpr (C == 4) vt::step4(R, W, acc0, acc1);
  else vt::step5(R, W, acc0, acc1);
  dec_lane[(size_t)t * 64] = make_uint2(acc0, acc1);
}

// grid = groups, 64 threads.  map: depuncture map with PUNCT remapped to n_in (the 0x7F row).
__global__ __launch_bounds__(64) void k_msc_vitT(EngineDev e, int cifs, int n_in, int nbits, const uint16_t *map,
                                                 const uint32_t *inT, uint2 *decT, const uint32_t *prbs)
{
  __shared__ unsigned char pos_tab[6][64];
  const int g = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < 6 * 64; i += 64) pos_tab[i / 64][i % 64] = vt::VT_POS[i / 64][i % 64];
  const MscJob q = msc_job(e, g * 64 + lane, cifs);
  const int nsteps = nbits + 6, rows = n_in / 4 + 1;
  const uint32_t *in_lane = inT + (size_t)g * rows * 64 + lane;
  uint2 *dec_lane = decT + (size_t)g * nsteps * 64 + lane;

  vt::s2 R[32];
#pragma unroll
  for (int r = 0; r < 32; r++) R[r] = vt::pk(2000, 2000);          // viterbi_spiral.cpp:98-101 (0 / 1000), doubled
  R[0] = vt::pk(0, 2000);
  for (int t = 0; t < nsteps; t += 6) {
    if (((t / 6) & 1) == 0) {                                      // re-centre every 12 steps on the metric of label 0
      const vt::s2 ref = vt::pk(R[0].x, R[0].x);
#pragma unroll
      for (int r = 0; r < 32; r++) R[r] = R[r] - ref;
    }
    vt_one<0>(R, in_lane, map, t + 0, dec_lane);
    vt_one<1>(R, in_lane, map, t + 1, dec_lane);
    vt_one<2>(R, in_lane, map, t + 2, dec_lane);
    vt_one<3>(R, in_lane, map, t + 3, dec_lane);
    vt_one<4>(R, in_lane, map, t + 4, dec_lane);
    vt_one<5>(R, in_lane, map, t + 5, dec_lane);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  // chain-back per lane (viterbi_spiral.cpp:114-125 in label space) + PRBS (backend.cpp:155-158) + byte packing
  uint32_t *out = nullptr;
  if (q.valid)
    out = reinterpret_cast<uint32_t *>(e.msc_out + (((size_t)q.s * e.max_subch + q.j) * MSC_SLOTS + (size_t)(q.out_idx % MSC_SLOTS)) * e.msc_stride);
  int L = 0;
  unsigned outw = 0;
  for (int t = nsteps - 1; t >= 6; --t) {
    const int c = t % 6, p = 5 - c;                                // VT_P[c] = (5 - c) % 6
    const uint2 w = dec_lane[(size_t)t * 64];
    const int pos = pos_tab[c][L];
    const unsigned bit = (((pos & 32) ? w.y : w.x) >> (pos & 31)) & 1u;
    const int qb = t - 6;
    outw |= bit << (((qb >> 3) & 3) * 8 + 7 - (qb & 7));
    L = (L & ~(1 << p)) | ((int)bit << p);
    if ((qb & 31) == 0) {
      if (out) out[qb >> 5] = outw ^ prbs[qb >> 5];
      outw = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- launch
int launch_msc_vitT(const EngineDev &e, int cifs, int n_in, int nbits, const uint16_t *map2, uint32_t *inT, uint2 *decT,
                    hipStream_t st, Marker &mk)
{
  const DevTables *t;
  int rc = get_tables(&t);
  if (rc) return rc;
  const int jobs = e.n_streams * cifs * e.max_subch, groups = (jobs + 63) / 64;
  mk.begin(6, st);
  hipLaunchKernelGGL(k_msc_prep, dim3(groups), dim3(256), 0, st, e, cifs, n_in, inT);
  mk.end(6, st);
  mk.begin(7, st);
  hipLaunchKernelGGL(k_msc_vitT, dim3(groups), dim3(64), 0, st, e, cifs, n_in, nbits, map2, inT, decT, t->prbs_words);
  mk.end(7, st);
  DABX_HIP(hipGetLastError());
  return 0;
}

}  // namespace dabx
