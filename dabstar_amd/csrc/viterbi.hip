// viterbi.hip -- stage-level Viterbi / deconvolve kernels (see viterbi_core.h for the algorithm).
#include "dabx_internal.h"
#include "viterbi_core.h"

namespace dabx {

struct SrcI16 {                       // ViterbiSpiral::deconvolve input: 4*(n+6) int16, already depunctured (viterbi_core.h: key / raw / syms)
  const int16_t *soft;
  int sat;                            // AVX2 body: saturating symbol conversion
  typedef int Key;
  typedef short4 Raw;
  __device__ Key key(int t) const { return t; }
  __device__ Raw raw(Key t) const { return *reinterpret_cast<const short4 *>(soft + 4 * t); }
  __device__ VitSyms syms(Raw v, Key) const
  {
    if (sat) return {vit_sym_from_i16_sat(v.x), vit_sym_from_i16_sat(v.y), vit_sym_from_i16_sat(v.z), vit_sym_from_i16_sat(v.w)};
    return {vit_sym_from_i16(v.x), vit_sym_from_i16(v.y), vit_sym_from_i16(v.z), vit_sym_from_i16(v.w)};
  }
};

struct SrcI16Map {                    // Protection::deconvolve input: punctured int16 + depuncture map
  const int16_t *in;
  const uint16_t *map;
  typedef ushort4 Key;
  typedef short4 Raw;
  __device__ Key key(int t) const { return *reinterpret_cast<const ushort4 *>(map + 4 * t); }
  __device__ int16_t ld(uint16_t idx) const { return in[idx == PUNCT ? 0 : idx]; }
  __device__ Raw raw(Key m) const { short4 r; r.x = ld(m.x); r.y = ld(m.y); r.z = ld(m.z); r.w = ld(m.w); return r; }
  __device__ static int cv(int16_t v, uint16_t idx) { return vit_sym_from_i16(idx == PUNCT ? (int16_t)0 : v); }
  __device__ VitSyms syms(Raw v, Key m) const { return {cv(v.x, m.x), cv(v.y, m.y), cv(v.z, m.z), cv(v.w, m.w)}; }
};

template <class Src>
__device__ __forceinline__ void viterbi_wave_to_packed(const Src &src, int nbits, char *wtab, uint32_t *raw, uint32_t *dec,
                                                       uint32_t *out_words, int lane, int tie_mode)
{
  const VitLaneConst k = vit_lane_const(lane);
  if (tie_mode == 2) vit_forward<2>(src, nbits + 6, wtab, dec, lane, k);
  else if (tie_mode) vit_forward<1>(src, nbits + 6, wtab, dec, lane, k);
  else vit_forward<0>(src, nbits + 6, wtab, dec, lane, k);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);     // decision stores of this wave have left the CU before they are re-read
  vit_traceback(dec, nbits, lane, raw);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int w = lane; w < (nbits + 31) / 32; w += 64) out_words[w] = vit_output_word(raw, w);
}

__global__ __launch_bounds__(256) void k_viterbi_i16(const int16_t *soft, int nbits, int batch, uint32_t *dec,
                                                     uint32_t *packed, int words_per, int tie_mode)
{
  __shared__ __attribute__((aligned(16))) char wtab[4][VIT_BLK * 16];
  __shared__ uint32_t raw[4][VIT_RAW_WORDS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int job = blockIdx.x * 4 + wave;
  if (job >= batch) return;
  SrcI16 src{soft + (size_t)job * 4 * (nbits + 6), tie_mode};
  viterbi_wave_to_packed(src, nbits, wtab[wave], raw[wave], dec + (size_t)job * vit_scratch_words(nbits),
                         packed + (size_t)job * words_per, lane, tie_mode);
}

__global__ __launch_bounds__(256) void k_deconvolve_i16(const int16_t *in, int in_stride, const uint16_t *map, int nbits,
                                                        int batch, uint32_t *dec, uint32_t *packed, int words_per)
{
  __shared__ __attribute__((aligned(16))) char wtab[4][VIT_BLK * 16];
  __shared__ uint32_t raw[4][VIT_RAW_WORDS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int job = blockIdx.x * 4 + wave;
  if (job >= batch) return;
  SrcI16Map src{in + (size_t)job * in_stride, map};
  viterbi_wave_to_packed(src, nbits, wtab[wave], raw[wave], dec + (size_t)job * vit_scratch_words(nbits),
                         packed + (size_t)job * words_per, lane, 0);
}

// packed (MSB-first bytes) -> one bit per byte, as the reference's output convention
__global__ void k_unpack_bits(const uint32_t *packed, int words_per, int nbits, int batch, uint8_t *bits)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)batch * nbits) return;
  const int job = (int)(i / nbits), q = (int)(i % nbits);
  const uint8_t *b = reinterpret_cast<const uint8_t *>(packed + (size_t)job * words_per);
  bits[i] = (b[q >> 3] >> (7 - (q & 7))) & 1;
}

int viterbi_scratch_bytes_per_trellis(int nbits) { return (int)(vit_scratch_words(nbits) * 4); }

static int run_packed_then_unpack(int nbits, int batch, uint8_t *bits, hipStream_t st,
                                  void (*launch)(uint32_t *, uint32_t *, int, void *), void *ctx)
{
  const int words_per = (nbits + 31) / 32;
  uint32_t *dec = nullptr, *packed = nullptr;
  DABX_HIP(hipMalloc(&dec, vit_scratch_words(nbits) * 4 * (size_t)batch));
  DABX_HIP(hipMalloc(&packed, (size_t)words_per * 4 * batch));
  launch(dec, packed, words_per, ctx);
  const size_t total = (size_t)batch * nbits;
  hipLaunchKernelGGL(k_unpack_bits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, packed, words_per, nbits,
                     batch, bits);
  DABX_HIP(hipGetLastError());
  DABX_HIP(hipStreamSynchronize(st));
  DABX_HIP(hipFree(dec));
  DABX_HIP(hipFree(packed));
  return 0;
}

int launch_viterbi_i16(const int16_t *soft, int nbits, int batch, uint8_t *bits, hipStream_t st, int tie_mode)
{
  struct Ctx { const int16_t *soft; int nbits, batch; hipStream_t st; int tie; } c{soft, nbits, batch, st, tie_mode};
  return run_packed_then_unpack(nbits, batch, bits, st, [](uint32_t *dec, uint32_t *packed, int wp, void *p) {
    auto *c = (Ctx *)p;
    hipLaunchKernelGGL(k_viterbi_i16, dim3((c->batch + 3) / 4), dim3(256), 0, c->st, c->soft, c->nbits, c->batch, dec,
                       packed, wp, c->tie);
  }, &c);
}

int launch_deconvolve_i16(const int16_t *in, int in_stride, const uint16_t *map, int nbits, int batch, uint8_t *bits,
                          hipStream_t st)
{
  struct Ctx { const int16_t *in; int stride; const uint16_t *map; int nbits, batch; hipStream_t st; }
      c{in, in_stride, map, nbits, batch, st};
  return run_packed_then_unpack(nbits, batch, bits, st, [](uint32_t *dec, uint32_t *packed, int wp, void *p) {
    auto *c = (Ctx *)p;
    hipLaunchKernelGGL(k_deconvolve_i16, dim3((c->batch + 3) / 4), dim3(256), 0, c->st, c->in, c->stride, c->map,
                       c->nbits, c->batch, dec, packed, wp);
  }, &c);
}

}  // namespace dabx
