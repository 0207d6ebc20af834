// capi_stage.cpp -- stage-level C ABI (host-pointer entry points mirroring single reference functions).
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
int dabx_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
int dabx_set_device(int device)
{
  int rc = need_device();
  if (rc) return rc;
  DABX_HIP(hipSetDevice(device));
  return 0;
}

int dabx_viterbi(const int16_t *soft, int nbits, int batch, uint8_t *bits)
{
  if (!soft || !bits || nbits <= 0 || batch <= 0) { set_error("dabx_viterbi: bad argument"); return DABX_E_ARG; }
  int rc = need_device();
  if (rc) return rc;
  DevBuf dsoft, dbits;
  const size_t nin = (size_t)batch * 4 * (nbits + 6);
  if ((rc = dsoft.from_host(soft, nin * 2))) return rc;
  if ((rc = dbits.alloc((size_t)batch * nbits))) return rc;
  if ((rc = launch_viterbi_i16(dsoft.as<int16_t>(), nbits, batch, dbits.as<uint8_t>(), 0))) return rc;
  return dbits.to_host(bits, (size_t)batch * nbits);
}

int dabx_profile_input_bits(int kbps, int prot_level, int short_form)
{
  std::vector<uint16_t> m;
  int n = 0;
  const int rc = host_profile_map(kbps, prot_level, short_form, m, &n);
  return rc ? rc : n;
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

}  // extern "C"
