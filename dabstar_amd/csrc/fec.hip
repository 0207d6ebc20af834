// fec.hip -- stage-level kernels for RS(120,110), fire code and CRC-16 (one lane per code word / header).
#include "dabx_internal.h"
#include "fec_core.h"

namespace dabx {

__global__ __launch_bounds__(256) void k_rs_decode(const uint8_t *in, int batch, uint8_t *out, int16_t *ret, DevTables t)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  uint8_t cw[120];
  const uint32_t *src = reinterpret_cast<const uint32_t *>(in + (size_t)i * 120);
#pragma unroll
  for (int k = 0; k < 30; k++) {
    const uint32_t v = src[k];
    cw[4 * k] = v & 0xFF; cw[4 * k + 1] = (v >> 8) & 0xFF; cw[4 * k + 2] = (v >> 16) & 0xFF; cw[4 * k + 3] = v >> 24;
  }
  const Gf gf{t.gf_exp, t.gf_log};
  const int r = rs_decode_120(CwArray{cw}, gf);
  ret[i] = (int16_t)r;
  for (int k = 0; k < 110; k++) out[(size_t)i * 110 + k] = cw[k];
}

__global__ void k_firecode(uint8_t *x, int batch, int correct, uint8_t *ok, DevTables t)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  uint8_t *p = x + (size_t)i * 12;
  if (correct) ok[i] = firecode_check_and_correct(p, t.fc_crctab, t.fc_syndrome);
  else ok[i] = firecode_syndrome([&](int k) { return p[k]; }, t.fc_crctab) == 0;
}

__global__ void k_crc16_check(const uint8_t *msgs, int stride, int len, int batch, uint8_t *ok, DevTables t)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  ok[i] = crc16_check_bytes(msgs + (size_t)i * stride, len, t.crc_ccitt);
}

int launch_rs_decode(const uint8_t *in, int batch, uint8_t *out, int16_t *ret, hipStream_t st)
{
  const DevTables *t;
  int rc = get_tables(&t);
  if (rc) return rc;
  hipLaunchKernelGGL(k_rs_decode, dim3((batch + 255) / 256), dim3(256), 0, st, in, batch, out, ret, *t);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_firecode(uint8_t *x, int batch, int correct, uint8_t *ok, hipStream_t st)
{
  const DevTables *t;
  int rc = get_tables(&t);
  if (rc) return rc;
  hipLaunchKernelGGL(k_firecode, dim3((batch + 255) / 256), dim3(256), 0, st, x, batch, correct, ok, *t);
  DABX_HIP(hipGetLastError());
  return 0;
}
int launch_crc16_check(const uint8_t *msgs, int stride, int len, int batch, uint8_t *ok, hipStream_t st)
{
  const DevTables *t;
  int rc = get_tables(&t);
  if (rc) return rc;
  hipLaunchKernelGGL(k_crc16_check, dim3((batch + 255) / 256), dim3(256), 0, st, msgs, stride, len, batch, ok, *t);
  DABX_HIP(hipGetLastError());
  return 0;
}

}  // namespace dabx
