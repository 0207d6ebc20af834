// iqfile.hip -- recorded-IQ payload bytes -> cf32 at 2.048 MS/s on the GPU (HBM-bound byte work, one thread per sample).
//   k_decode_iq     container bytes -> cf32 (raw_reader.cpp:66-70,155-158; libsndfile sf_readf_float rules for
//                   wav_reader.cpp:164; xml_reader.cpp:254-398)
//   k_resample_1ms  linear interpolation of 1-ms blocks to 2048 samples (wav_reader.cpp:190-206, xml_reader.cpp:237-244)
#include "dabx_internal.h"
#include "iqfile.h"
#include <algorithm>

namespace dabx {

__device__ __forceinline__ uint32_t ld_be(const uint8_t *p, int n)
{
  uint32_t v = 0;
  for (int i = 0; i < n; i++) v = (v << 8) | p[i];
  return v;
}
__device__ __forceinline__ uint32_t ld_le(const uint8_t *p, int n)
{
  uint32_t v = 0;
  for (int i = n - 1; i >= 0; i--) v = (v << 8) | p[i];
  return v;
}

__device__ __forceinline__ float decode_one(const uint8_t *p, const IqDecode &d)
{
  switch (d.container) {
  case DABX_C_U8:
    return d.family == DABX_FAMILY_WAV ? __fmul_rn((float)((int)p[0] - 128), 1.0f / 128.0f)          // pcm.c uc2f: (x - 128) / 0x80
                                       : __fdiv_rn(__fsub_rn((float)p[0], 127.38f), 128.0f);         // raw_reader.cpp:69, xml_reader.cpp:85
  case DABX_C_S8:
    return d.family == DABX_FAMILY_UFF ? __fdiv_rn((float)(int8_t)p[0], 127.0f)                      // xml_reader.cpp:266
                                       : __fmul_rn((float)(int8_t)p[0], 1.0f / 128.0f);
  case DABX_C_I16: {
    const int16_t v = (int16_t)(d.big_endian ? ld_be(p, 2) : ld_le(p, 2));
    return __fmul_rn((float)v, d.int_scale);                                                         // x / 2^15 or x / 2^(Bits-1): exact
  }
  case DABX_C_I24: {
    int32_t v = (int32_t)(d.big_endian ? ld_be(p, 3) : ld_le(p, 3));
    if (v & 0x800000) v |= (int32_t)0xFF000000;
    return __fmul_rn((float)v, d.int_scale);
  }
  case DABX_C_I32: {
    const int32_t v = (int32_t)(d.big_endian ? ld_be(p, 4) : ld_le(p, 4));
    return __fmul_rn((float)v, d.int_scale);
  }
  default:
    return __uint_as_float(d.big_endian ? ld_be(p, 4) : ld_le(p, 4));
  }
}

// sample i of a payload that starts (on a read block, for the quirk mode) at src
__device__ __forceinline__ float2 decode_sample(const uint8_t *src, const IqDecode &d, size_t i)
{
  const uint8_t *p = src + i * (size_t)(2 * d.bytes);
  float a, b;
  if (d.quirk_i24) {
    // the reference's int24 / MSB loops, literally (xml_reader.cpp:312-325 IQ, :458-472 QI): `src` starts on a read block
    const size_t c = i / (size_t)d.quirk_block, ii = i - c * (size_t)d.quirk_block;
    const uint8_t *lbuf = src + c * (size_t)d.quirk_block * 6;
    int32_t t1 = (int32_t)((lbuf[6 * ii] << 16) | (lbuf[6 * ii + 1] << 8) | lbuf[6 * ii + 2]);
    int32_t t2 = (int32_t)((lbuf[6 * ii + 3] << 16) | (lbuf[4 * ii + 4] << 8) | lbuf[6 * ii + 5]);
    const int32_t ext = d.quirk_sign7f ? (int32_t)0x7F000000 : (int32_t)0xFF000000;
    if (t1 & 0x800000) t1 |= ext;
    if (t2 & 0x800000) t2 |= ext;
    a = __fmul_rn((float)t1, d.int_scale); b = __fmul_rn((float)t2, d.int_scale);
  } else {
    a = decode_one(p, d); b = decode_one(p + d.bytes, d);
  }
  if (d.swap_iq) { const float t = a; a = b; b = t; }
  return make_float2(a, b);
}

// ---- bulk ingest, general form: every stream its own container, rate and length, ONE slab, two launches (include/dabx.h "Bulk ingest") ----
// grid (x, S).  Pass 1: the stream's payload -> cf32, straight into its ring (2.048 MS/s) or behind the carried samples in its work row.
__global__ __launch_bounds__(256) void k_ingest_decode_multi(IngestMulti m)
{
  const int s = blockIdx.y;
  const IngestJob j = m.jobs[s];
  if (j.n == 0) return;
  const uint8_t *src = m.slab + j.src_off;
  float2 *ring = m.iq + (size_t)s * m.ring_len;
  float2 *work = m.work + (size_t)s * m.work_pitch;
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j.M) for (size_t i = t0; i < j.carry_n; i += stride) work[i] = m.carry[(size_t)s * m.carry_pitch + i];
  for (size_t i = t0; i < j.n; i += stride) {
    const float2 v = decode_sample(src, j.dec, i);
    if (j.M) work[j.carry_n + i] = v;
    else ring[(size_t)((j.dst0 + i) % (unsigned long long)m.ring_len)] = v;
  }
}
// Pass 2 (resampling streams): the 1-ms blocks of [carry | decoded] -> 2048 samples each into the ring (k_resample_1ms's arithmetic), and what
// is left over -> the stream's carry row for the next slab.
__global__ __launch_bounds__(256) void k_ingest_resample_multi(IngestMulti m)
{
  const int s = blockIdx.y;
  const IngestJob j = m.jobs[s];
  if (j.n == 0 || j.M == 0) return;
  const float2 *V = m.work + (size_t)s * m.work_pitch;
  float2 *ring = m.iq + (size_t)s * m.ring_len;
  const int16_t *ti = m.tab_int + (size_t)j.tab * 2048;
  const float *tf = m.tab_frac + (size_t)j.tab * 2048;
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n_out = (size_t)j.blocks * 2048;
  for (size_t n = t0; n < n_out; n += stride) {
    const size_t c = n >> 11;
    const int q = (int)(n & 2047), base = ti[q];
    const float r = tf[q], w = __fsub_rn(1.0f, r);
    const float2 lo = V[c * (size_t)j.M + base], hi = V[c * (size_t)j.M + base + 1];
    ring[(size_t)((j.dst0 + n) % (unsigned long long)m.ring_len)] =
        make_float2(__fadd_rn(__fmul_rn(hi.x, r), __fmul_rn(lo.x, w)), __fadd_rn(__fmul_rn(hi.y, r), __fmul_rn(lo.y, w)));
  }
  for (size_t i = t0; i < j.keep; i += stride) m.carry[(size_t)s * m.carry_pitch + i] = V[(size_t)j.blocks * j.M + i];
}
__global__ void k_commit_counts(unsigned long long *wr, const unsigned *counts, int n_streams)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_streams) wr[s] += counts[s];
}

// dst[(dst0 + i) % dst_len] = sample i ; dst_len = 0 -> linear buffer
__global__ __launch_bounds__(256) void k_decode_iq(const uint8_t *src, IqDecode d, float2 *dst, unsigned long long dst0, int dst_len, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t o = dst_len ? (size_t)((dst0 + i) % (unsigned long long)dst_len) : (size_t)(dst0 + i);
  dst[o] = decode_sample(src, d, i);
}

// Block c, output j: conv = V + c M ; out = conv[base_j + 1] * frac_j + conv[base_j] * (1 - frac_j)
__global__ __launch_bounds__(256) void k_resample_1ms(const float2 *V, int M, const int16_t *tab_int, const float *tab_frac,
                                                      float2 *dst, unsigned long long dst0, int dst_len, size_t n_out)
{
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_out) return;
  const size_t c = n >> 11;
  const int j = (int)(n & 2047);
  const int base = tab_int[j];
  const float r = tab_frac[j], q = __fsub_rn(1.0f, r);
  const float2 lo = V[c * (size_t)M + base], hi = V[c * (size_t)M + base + 1];
  const float2 v = make_float2(__fadd_rn(__fmul_rn(hi.x, r), __fmul_rn(lo.x, q)), __fadd_rn(__fmul_rn(hi.y, r), __fmul_rn(lo.y, q)));
  const size_t o = dst_len ? (size_t)((dst0 + n) % (unsigned long long)dst_len) : (size_t)(dst0 + n);
  dst[o] = v;
}

int launch_ingest_multi(const IngestMulti &m, int n_streams, unsigned max_n, unsigned max_out, hipStream_t st)
{
  if (max_n == 0) return 0;
  const unsigned bx = std::min<unsigned>((max_n + 255) / 256, 1024);
  hipLaunchKernelGGL(k_ingest_decode_multi, dim3(bx, n_streams), dim3(256), 0, st, m);
  if (max_out) hipLaunchKernelGGL(k_ingest_resample_multi, dim3(std::min<unsigned>((max_out + 255) / 256, 1024), n_streams), dim3(256), 0, st, m);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_commit_counts(unsigned long long *wr, const unsigned *counts_dev, int n_streams, hipStream_t st)
{
  hipLaunchKernelGGL(k_commit_counts, dim3((n_streams + 255) / 256), dim3(256), 0, st, wr, counts_dev, n_streams);
  DABX_HIP(hipGetLastError());
  return 0;
}

int launch_decode_iq(const uint8_t *src, const IqDecode &d, float2 *dst, unsigned long long dst0, int dst_len, size_t n, hipStream_t st)
{
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_decode_iq, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, d, dst, dst0, dst_len, n);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_resample_1ms(const float2 *V, int M, const int16_t *tab_int, const float *tab_frac, float2 *dst, unsigned long long dst0,
                        int dst_len, size_t n_out, hipStream_t st)
{
  if (n_out == 0) return 0;
  hipLaunchKernelGGL(k_resample_1ms, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, V, M, tab_int, tab_frac, dst, dst0, dst_len, n_out);
  DABX_HIP(hipGetLastError());
  return 0;
}

}  // namespace dabx
