// capi_stage.cpp -- stage-level C ABI (host-pointer entry points mirroring single reference functions).
#include <algorithm>
#include "dabx_internal.h"
#include <cstring>

namespace dabx { const char *last_error(); }
using namespace dabx;

namespace {
// RAII device buffer for the host-pointer convenience entry points
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t n) { DABX_HIP(hipMalloc(&p, n ? n : 1)); return 0; }
  int from_host(const void *h, size_t n) { int rc = alloc(n); if (rc) return rc; DABX_HIP(hipMemcpy(p, h, n, hipMemcpyHostToDevice)); return 0; }
  int to_host(void *h, size_t n) { DABX_HIP(hipMemcpy(h, p, n, hipMemcpyDeviceToHost)); return 0; }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};
int need_device()
{
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { set_error("no HIP device available (libdabx has no CPU fallback)"); return DABX_E_NODEVICE; }
  return 0;
}
}  // namespace

extern "C" {

const char *dabx_last_error(void) { return dabx::last_error(); }
int dabx_abi_version(void) { return DABX_ABI_VERSION; }
#ifndef DABX_HIPMODULE
// 0: the kernels are in this library's fat binary (default build); 1 in the hipModule form (csrc/hipmodule/hipmodule_rt.cpp)
int dabx_internal_hipmodule(void) { return 0; }
#endif
int dabx_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
int dabx_set_device(int device)
{
  int rc = need_device();
  if (rc) return rc;
  DABX_HIP(hipSetDevice(device));
  return 0;
}

int dabx_viterbi(const int16_t *soft, int nbits, int batch, uint8_t *bits) { return dabx_viterbi_mode(soft, nbits, batch, 0, bits); }

int dabx_viterbi_mode(const int16_t *soft, int nbits, int batch, int tie_mode, uint8_t *bits)
{
  if (!soft || !bits || nbits <= 0 || batch <= 0 || tie_mode < 0 || tie_mode > 2) { set_error("dabx_viterbi: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  DevBuf dsoft, dbits;
  const size_t nin = (size_t)batch * 4 * (nbits + 6);
  if ((rc = dsoft.from_host(soft, nin * 2))) return rc;
  if ((rc = dbits.alloc((size_t)batch * nbits))) return rc;
  if ((rc = launch_viterbi_i16(dsoft.as<int16_t>(), nbits, batch, dbits.as<uint8_t>(), 0, tie_mode))) return rc;
  return dbits.to_host(bits, (size_t)batch * nbits);
}

int dabx_profile_input_bits(int kbps, int prot_level, int short_form)
{
  std::vector<uint16_t> m;
  int n = 0;
  const int rc = host_profile_map(kbps, prot_level, short_form, m, &n);
  return rc ? rc : n;
}

// Host only: the depuncture index list itself (mother-code bit i <- transmitted bit map[i], 0xFFFF = punctured), the table
// the device kernels gather through; length 96*kbps + 24.  Returns the number of transmitted bits.
int dabx_profile_map(int kbps, int prot_level, int short_form, uint16_t *map, int max_entries)
{
  std::vector<uint16_t> m;
  int n = 0;
  const int rc = host_profile_map(kbps, prot_level, short_form, m, &n);
  if (rc) return rc;
  if (!map || max_entries < (int)m.size()) { set_error("dabx_profile_map: need room for %d entries", (int)m.size()); return DABX_E_ARG; }
  std::copy(m.begin(), m.end(), map);
  return n;
}

int dabx_deconvolve(const int16_t *in, int in_stride, int kbps, int prot_level, int short_form, int batch, uint8_t *bits)
{
  if (!in || !bits || batch <= 0) { set_error("dabx_deconvolve: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  const uint16_t *map = nullptr;
  int n_in = 0;
  if ((rc = get_profile_map(kbps, prot_level, short_form, &map, &n_in))) return rc;
  if (in_stride < n_in) { set_error("dabx_deconvolve: stride %d < %d input bits", in_stride, n_in); return DABX_E_ARG; }
  const int nbits = 24 * kbps;
  DevBuf din, dbits;
  if ((rc = din.from_host(in, (size_t)batch * in_stride * 2))) return rc;
  if ((rc = dbits.alloc((size_t)batch * nbits))) return rc;
  if ((rc = launch_deconvolve_i16(din.as<int16_t>(), in_stride, map, nbits, batch, dbits.as<uint8_t>(), 0))) return rc;
  return dbits.to_host(bits, (size_t)batch * nbits);
}

int dabx_rs_decode(const uint8_t *in, int batch, uint8_t *out, int16_t *ret)
{
  if (!in || !out || !ret || batch <= 0) { set_error("dabx_rs_decode: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  DevBuf din, dout, dret;
  if ((rc = din.from_host(in, (size_t)batch * 120))) return rc;
  if ((rc = dout.alloc((size_t)batch * 110))) return rc;
  if ((rc = dret.alloc((size_t)batch * 2))) return rc;
  if ((rc = launch_rs_decode(din.as<uint8_t>(), batch, dout.as<uint8_t>(), dret.as<int16_t>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  if ((rc = dout.to_host(out, (size_t)batch * 110))) return rc;
  return dret.to_host(ret, (size_t)batch * 2);
}

static int firecode_common(uint8_t *x, int batch, uint8_t *ok, int correct)
{
  if (!x || !ok || batch <= 0) { set_error("dabx_firecode: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  DevBuf dx, dok;
  if ((rc = dx.from_host(x, (size_t)batch * 12))) return rc;
  if ((rc = dok.alloc((size_t)batch))) return rc;
  if ((rc = launch_firecode(dx.as<uint8_t>(), batch, correct, dok.as<uint8_t>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  if (correct && (rc = dx.to_host(x, (size_t)batch * 12))) return rc;
  return dok.to_host(ok, (size_t)batch);
}
int dabx_firecode_check(const uint8_t *x, int batch, uint8_t *ok) { return firecode_common(const_cast<uint8_t *>(x), batch, ok, 0); }
int dabx_firecode_check_and_correct(uint8_t *x, int batch, uint8_t *ok) { return firecode_common(x, batch, ok, 1); }

int dabx_crc16_check(const uint8_t *msgs, int stride, int len, int batch, uint8_t *ok)
{
  if (!msgs || !ok || batch <= 0 || len < 0 || stride < len + 2) { set_error("dabx_crc16_check: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  DevBuf dm, dok;
  if ((rc = dm.from_host(msgs, (size_t)batch * stride))) return rc;
  if ((rc = dok.alloc((size_t)batch))) return rc;
  if ((rc = launch_crc16_check(dm.as<uint8_t>(), stride, len, batch, dok.as<uint8_t>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  return dok.to_host(ok, (size_t)batch);
}

int dabx_fft2048(const dabx_cf32 *in, int batch, int inverse, dabx_cf32 *out)
{
  if (!in || !out || batch <= 0) { set_error("dabx_fft2048: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  DevBuf din, dout;
  const size_t n = (size_t)batch * TU * sizeof(float2);
  if ((rc = din.from_host(in, n))) return rc;
  if ((rc = dout.alloc(n))) return rc;
  if ((rc = launch_fft2048(din.as<float2>(), batch, inverse, dout.as<float2>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  return dout.to_host(out, n);
}

int dabx_prs_correlate(const dabx_cf32 *v, int batch, float threshold, int strongest, int32_t *start_index)
{
  if (!v || !start_index || batch <= 0) { set_error("dabx_prs_correlate: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  DevBuf din, dout;
  if ((rc = din.from_host(v, (size_t)batch * TU * sizeof(float2)))) return rc;
  if ((rc = dout.alloc((size_t)batch * 4))) return rc;
  if ((rc = launch_prs_correlate(din.as<float2>(), batch, threshold, strongest, dout.as<int32_t>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  return dout.to_host(start_index, (size_t)batch * 4);
}

int dabx_coarse_cfo(const dabx_cf32 *fft_sym0, int batch, int32_t *hz)
{
  if (!fft_sym0 || !hz || batch <= 0) { set_error("dabx_coarse_cfo: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  DevBuf din, dout;
  if ((rc = din.from_host(fft_sym0, (size_t)batch * TU * sizeof(float2)))) return rc;
  if ((rc = dout.alloc((size_t)batch * 4))) return rc;
  if ((rc = launch_coarse_cfo(din.as<float2>(), batch, dout.as<int32_t>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  return dout.to_host(hz, (size_t)batch * 4);
}

struct dabx_demap { DemapDev d; };

int dabx_demap_create(int batch, dabx_demap **out)
{
  if (!out || batch <= 0) { set_error("dabx_demap_create: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  auto *h = new dabx_demap();
  if ((rc = demap_alloc(h->d, batch))) { delete h; return rc; }
  if ((rc = launch_demap_init(h->d, 0))) { demap_free(h->d); delete h; return rc; }
  DABX_HIP(hipStreamSynchronize(0));
  *out = h;
  return 0;
}
void dabx_demap_destroy(dabx_demap *d) { if (d) { demap_free(d->d); delete d; } }
int dabx_demap_reset(dabx_demap *d)
{
  if (!d) return DABX_E_ARG;
  int rc = launch_demap_reset(d->d, 0);
  if (rc) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  return 0;
}
int dabx_demap_set_soft_bit_gen_type(dabx_demap *d, int type)
{
  if (!d || type < 1 || type > 3) return DABX_E_ARG;
  d->d.soft_type = type;
  return 0;
}
static int demap_store(dabx_demap *d, const dabx_cf32 *fft, bool null_sym)
{
  if (!d || !fft) return DABX_E_ARG;
  DevBuf din;
  int rc = din.from_host(fft, (size_t)d->d.batch * TU * sizeof(float2));
  if (rc) return rc;
  rc = null_sym ? launch_demap_store_null(d->d, din.as<float2>(), 0) : launch_demap_store_ref(d->d, din.as<float2>(), 0);
  if (rc) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  return 0;
}
int dabx_demap_store_reference_symbol_0(dabx_demap *d, const dabx_cf32 *fft) { return demap_store(d, fft, false); }
int dabx_demap_store_null_symbol_without_tii(dabx_demap *d, const dabx_cf32 *fft) { return demap_store(d, fft, true); }
int dabx_demap_get_snr_db(dabx_demap *d, float *snr_db)
{
  if (!d || !snr_db) return DABX_E_ARG;
  DevBuf dout;
  int rc;
  if ((rc = dout.alloc((size_t)d->d.batch * 4))) return rc;
  if ((rc = launch_demap_snr(d->d, dout.as<float>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  return dout.to_host(snr_db, (size_t)d->d.batch * 4);
}
int dabx_demap_get_lcd_data(dabx_demap *d, float *snr_db, float *mer_db, float *mean_value)
{
  if (!d) return DABX_E_ARG;
  const int B = d->d.batch;
  DevBuf dout;
  int rc;
  if ((rc = dout.alloc((size_t)B * 12))) return rc;
  if ((rc = launch_demap_lcd(d->d, dout.as<float>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  std::vector<float> h((size_t)B * 3);
  if ((rc = dout.to_host(h.data(), (size_t)B * 12))) return rc;
  for (int s = 0; s < B; s++) {
    if (snr_db) snr_db[s] = h[3 * (size_t)s];
    if (mer_db) mer_db[s] = h[3 * (size_t)s + 1];
    if (mean_value) mean_value[s] = h[3 * (size_t)s + 2];
  }
  return 0;
}
int dabx_demap_decode_symbols(dabx_demap *d, const dabx_cf32 *fft, int n_sym, const float *clock_err, int16_t *soft)
{
  if (!d || !fft || !clock_err || !soft || n_sym <= 0) return DABX_E_ARG;
  const int B = d->d.batch;
  DevBuf din, dce, dsoft;
  int rc;
  if ((rc = din.from_host(fft, (size_t)B * n_sym * TU * sizeof(float2)))) return rc;
  if ((rc = dce.from_host(clock_err, (size_t)B * 4))) return rc;
  if ((rc = dsoft.alloc((size_t)B * n_sym * K2 * 2))) return rc;
  if ((rc = launch_demap_symbols(d->d, din.as<float2>(), n_sym, dce.as<float>(), dsoft.as<int16_t>(), 0))) return rc;
  DABX_HIP(hipStreamSynchronize(0));
  return dsoft.to_host(soft, (size_t)B * n_sym * K2 * 2);
}

}  // extern "C"
