// ofdm.hip -- stage-level kernels of the OFDM front end (FFT, PRS correlator, coarse CFO, demapper).
#include "dabx_internal.h"
#include "ofdm_core.h"

namespace dabx {

__global__ __launch_bounds__(256) void k_fft2048(const float2 *in, float2 *out, int inverse, DevTables t)
{
  __shared__ float2 lds[FFT_LDS_FLOAT2];
  const int tid = threadIdx.x;
  const float2 *src = in + (size_t)blockIdx.x * TU;
  float2 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) v[u] = src[tid + 256 * u];
  if (inverse) fft2048<true>(v, lds, t.twiddle, tid); else fft2048<false>(v, lds, t.twiddle, tid);
  float2 *dst = out + (size_t)blockIdx.x * TU;
#pragma unroll
  for (int u = 0; u < 8; u++) dst[tid + 256 * u] = v[u];
}

__global__ __launch_bounds__(256) void k_prs_correlate(const float2 *in, float threshold, int strongest, int32_t *start,
                                                       DevTables t)
{
  __shared__ float2 lds[FFT_LDS_FLOAT2];
  __shared__ float peak[TU];
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const float2 *src = in + (size_t)blockIdx.x * TU;
  float2 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) v[u] = src[tid + 256 * u];
  const int r = prs_correlate_block(v, threshold, strongest, t, lds, peak, red, tid);
  if (tid == 0) start[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_coarse_cfo(const float2 *in, int32_t *hz, DevTables t)
{
  __shared__ float2 lds[FFT_LDS_FLOAT2];
  __shared__ float mag[160];
  const int tid = threadIdx.x;
  const float2 *src = in + (size_t)blockIdx.x * TU;
  float2 X[8];
#pragma unroll
  for (int u = 0; u < 8; u++) X[u] = src[tid + 256 * u];
  const int r = coarse_cfo_block(X, t, lds, mag, tid);
  if (tid == 0) hz[blockIdx.x] = r;
}

// ---- demapper state ------------------------------------------------------------------------------------
int demap_alloc(DemapDev &d, int batch)
{
  d.batch = batch;
  d.soft_type = 1;
  d.track_mer = 0;
  DABX_HIP(hipMalloc((void **)&d.std_dev, sizeof(float) * K * (size_t)batch));
  DABX_HIP(hipMalloc((void **)&d.phase_ref, sizeof(float2) * TU * (size_t)batch));
  DABX_HIP(hipMalloc((void **)&d.integ, sizeof(float) * K * (size_t)batch));
  DABX_HIP(hipMalloc((void **)&d.mean_power, sizeof(float) * K * (size_t)batch));
  DABX_HIP(hipMalloc((void **)&d.mean_sigma, sizeof(float) * K * (size_t)batch));
  DABX_HIP(hipMalloc((void **)&d.null_power, sizeof(float) * TU * (size_t)batch));
  DABX_HIP(hipMalloc((void **)&d.null_power2, sizeof(float) * TU * (size_t)batch));
  DABX_HIP(hipMalloc((void **)&d.mean_value, sizeof(float) * (size_t)batch));
  DABX_HIP(hipMalloc((void **)&d.mean_power_all, sizeof(float) * (size_t)batch));
  DABX_HIP(hipMemset(d.phase_ref, 0, sizeof(float2) * TU * (size_t)batch));
  return 0;
}
void demap_free(DemapDev &d)
{
  (void)hipFree(d.phase_ref); (void)hipFree(d.integ); (void)hipFree(d.mean_power);
  (void)hipFree(d.mean_sigma); (void)hipFree(d.null_power); (void)hipFree(d.null_power2); (void)hipFree(d.mean_value); (void)hipFree(d.mean_power_all);
  (void)hipFree(d.std_dev);
  d = DemapDev{};
}

// OfdmDecoder::reset (ofdm_decoder.cpp:90-101); first = also the constructor defaults (mMeanValue = 1, ofdm_decoder.h:103)
__global__ void k_demap_reset(DemapDev d, int first)
{
  const int s = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < K; i += blockDim.x) {
    d.integ[(size_t)s * K + i] = 0.f; d.mean_power[(size_t)s * K + i] = 0.f; d.mean_sigma[(size_t)s * K + i] = 0.f;
    d.std_dev[(size_t)s * K + i] = 0.f;                       // :92
  }
  for (int i = tid; i < TU; i += blockDim.x) { d.null_power[(size_t)s * TU + i] = 0.f; d.null_power2[(size_t)s * TU + i] = 0.f; }
  if (first && tid == 0) d.mean_value[s] = 1.0f;
  if (tid == 0) d.mean_power_all[s] = 1.0f;                  // :98
}

__global__ void k_demap_store_ref(DemapDev d, const float2 *fft)   // ofdm_decoder.cpp:132-136
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)d.batch * TU) d.phase_ref[i] = fft[i];
}

// store_null_symbol_without_tii, ofdm_decoder.cpp:114-130: IIR alpha 0.05 on the 1536 used bins
__device__ __forceinline__ void null_power_update(float *np, float2 x)
{
  const float kMinNoisePower = (1.0f / 32767.0f) * (1.0f / 32767.0f);
  const float power = x.x * x.x + x.y * x.y + kMinNoisePower;
  *np += 0.05f * (power - *np);
}
__global__ void k_demap_store_null(DemapDev d, const float2 *fft)
{
  const int s = blockIdx.x;
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    const int idx = i - K / 2;
    const int bin = idx < 0 ? idx + TU : idx + 1;
    null_power_update(&d.null_power[(size_t)s * TU + bin], fft[(size_t)s * TU + bin]);
  }
}

// decode_symbol for n_sym consecutive symbols of one stream per block (512 threads x 3 carriers)
__global__ __launch_bounds__(512) void k_demap_symbols(DemapDev d, const float2 *fft, int n_sym, const float *clock_err,
                                                       int16_t *soft, DevTables t)
{
  __shared__ float red[8];
  const int s = blockIdx.x, tid = threadIdx.x;
  DemapCarrier c[3];
  int bin[3], rel[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int k = tid + 512 * q;
    bin[q] = t.perm_bin[k]; rel[q] = t.perm_rel[k];
    c[q].prev = d.phase_ref[(size_t)s * TU + bin[q]];
    c[q].integ = d.integ[(size_t)s * K + k];
    c[q].mean_power = d.mean_power[(size_t)s * K + k];
    c[q].mean_sigma_sq = d.mean_sigma[(size_t)s * K + k];
    c[q].null_power = d.null_power[(size_t)s * TU + bin[q]];
    c[q].std_dev_sq = d.std_dev[(size_t)s * K + k];
  }
  float mean_value = d.mean_value[s], mpa = d.mean_power_all[s];
  const float ce = clock_err[s];
  float wk[3], pacc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 3; q++) wk[q] = mpa_weight(tid + 512 * q);
  for (int l = 0; l < n_sym; l++) {
    const float2 *X = fft + ((size_t)s * n_sym + l) * TU;
    int16_t *o = soft + ((size_t)s * n_sym + l) * K2;
    const float w2 = demap_w2(mean_value, d.soft_type);
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      int16_t sr, si;
      float pw;
      part += demap_one(c[q], X[bin[q]], rel[q], ce, w2, d.soft_type, sr, si, pw);
      pacc[q] = pacc[q] * mpa_decay() + pw;                        // sum_l d^(n-1-l) p_l per carrier, reduced once below
      o[tid + 512 * q] = sr;
      o[K + tid + 512 * q] = si;
    }
    mean_value = block_sum(part, red, tid) * (1.0f / (float)K);   // :294
  }
  {                                                                // :214 over the n_sym symbols in closed form
    float wsum = 0.f;
#pragma unroll
    for (int q = 0; q < 3; q++) wsum += wk[q] * pacc[q];
    wsum = block_sum(wsum, red, tid);
    mpa = mpa * mpa_decay_n(n_sym) + wsum;
  }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int k = tid + 512 * q;
    d.integ[(size_t)s * K + k] = c[q].integ;
    d.mean_power[(size_t)s * K + k] = c[q].mean_power;
    d.mean_sigma[(size_t)s * K + k] = c[q].mean_sigma_sq;
    d.std_dev[(size_t)s * K + k] = c[q].std_dev_sq;
  }
  // :354 mPhaseReference <- last symbol (all 2048 bins)
  if (n_sym > 0) {
    const float2 *X = fft + ((size_t)s * n_sym + (n_sym - 1)) * TU;
    for (int i = tid; i < TU; i += 512) d.phase_ref[(size_t)s * TU + i] = X[i];
  }
  if (tid == 0) { d.mean_value[s] = mean_value; d.mean_power_all[s] = mpa; }
}

// SNR estimate of the LCD statistics (ofdm_decoder.cpp:326-343) from the state as it stands
__global__ __launch_bounds__(256) void k_demap_snr(DemapDev d, float *snr_db)
{
  __shared__ float red[8];
  const int s = blockIdx.x, tid = threadIdx.x;
  float ns = 0.f;
  for (int i = tid; i < K; i += 256) {
    const int idx = i - K / 2;
    ns += d.null_power[(size_t)s * TU + (idx < 0 ? idx + TU : idx + 1)];
  }
  ns = block_sum(ns, red, tid);
  if (tid == 0) snr_db[s] = snr_db_from(d.mean_power_all[s], ns);
}

// The LCD record's device-side numbers (ofdm_decoder.cpp:326-345) from the state as it stands: SNR, MER, mMeanValue (TestData1)
__global__ __launch_bounds__(256) void k_demap_lcd(DemapDev d, float *out3)
{
  __shared__ float red[32];
  const int s = blockIdx.x, tid = threadIdx.x;
  float ns = 0.f, sd = 0.f;
  for (int i = tid; i < K; i += 256) {
    const int idx = i - K / 2;
    ns += d.null_power[(size_t)s * TU + (idx < 0 ? idx + TU : idx + 1)];
    sd += d.std_dev[(size_t)s * K + i];
  }
  block_sum2w(ns, sd, red, tid);
  if (tid == 0) { out3[3 * s] = snr_db_from(d.mean_power_all[s], ns); out3[3 * s + 1] = mer_db_from(sd); out3[3 * s + 2] = d.mean_value[s]; }
}

// ---- launchers --------------------------------------------------------------------------------------------
#define GET_TABLES(t) const DevTables *t; { int rc__ = get_tables(&t); if (rc__) return rc__; }

int launch_fft2048(const float2 *in, int batch, int inverse, float2 *out, hipStream_t st)
{
  GET_TABLES(t);
  hipLaunchKernelGGL(k_fft2048, dim3(batch), dim3(256), 0, st, in, out, inverse, *t);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_prs_correlate(const float2 *v, int batch, float threshold, int strongest, int32_t *start, hipStream_t st)
{
  GET_TABLES(t);
  hipLaunchKernelGGL(k_prs_correlate, dim3(batch), dim3(256), 0, st, v, threshold, strongest, start, *t);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_coarse_cfo(const float2 *fft, int batch, int32_t *hz, hipStream_t st)
{
  GET_TABLES(t);
  hipLaunchKernelGGL(k_coarse_cfo, dim3(batch), dim3(256), 0, st, fft, hz, *t);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_demap_reset(DemapDev &d, hipStream_t st)
{
  hipLaunchKernelGGL(k_demap_reset, dim3(d.batch), dim3(256), 0, st, d, 0);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_demap_init(DemapDev &d, hipStream_t st)
{
  hipLaunchKernelGGL(k_demap_reset, dim3(d.batch), dim3(256), 0, st, d, 1);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_demap_lcd(DemapDev &d, float *out3_dev, hipStream_t st)
{
  hipLaunchKernelGGL(k_demap_lcd, dim3(d.batch), dim3(256), 0, st, d, out3_dev);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_demap_snr(DemapDev &d, float *snr_db_dev, hipStream_t st)
{
  hipLaunchKernelGGL(k_demap_snr, dim3(d.batch), dim3(256), 0, st, d, snr_db_dev);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_demap_store_ref(DemapDev &d, const float2 *fft, hipStream_t st)
{
  const size_t n = (size_t)d.batch * TU;
  hipLaunchKernelGGL(k_demap_store_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, fft);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_demap_store_null(DemapDev &d, const float2 *fft, hipStream_t st)
{
  hipLaunchKernelGGL(k_demap_store_null, dim3(d.batch), dim3(256), 0, st, d, fft);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_demap_symbols(DemapDev &d, const float2 *fft, int n_sym, const float *clock_err, int16_t *soft, hipStream_t st)
{
  GET_TABLES(t);
  hipLaunchKernelGGL(k_demap_symbols, dim3(d.batch), dim3(512), 0, st, d, fft, n_sym, clock_err, soft, *t);
  DABX_HIP(hipGetLastError());
  return 0;
}

}  // namespace dabx
