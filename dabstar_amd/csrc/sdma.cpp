// sdma.cpp -- see sdma.h.
#include "sdma.h"
#include "dabx_internal.h"
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>

namespace dabx {

namespace {
struct Find { uint32_t domain, bdf; hsa_agent_t gpu; bool found; hsa_agent_t cpu; bool have_cpu; };
hsa_status_t agent_cb(hsa_agent_t a, void *p)
{
  Find *f = static_cast<Find *>(p);
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
  if (t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
  uint32_t bdf = 0, dom = 0;
  (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
  (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
  if ((bdf & 0xFFFFu) == f->bdf && dom == f->domain) { f->gpu = a; f->found = true; }
  return HSA_STATUS_SUCCESS;
}
std::once_flag g_hsa_once;
bool g_hsa_ok = false;
}  // namespace

int sdma_open(int hip_device, Sdma *out)
{
  *out = Sdma{};
  std::call_once(g_hsa_once, [] { g_hsa_ok = hsa_init() == HSA_STATUS_SUCCESS; });      // reference counted: the HIP runtime holds its own
  if (!g_hsa_ok) { set_error("sdma: hsa_init failed"); return DABX_E_HIP; }
  char bus[64] = {0};
  DABX_HIP(hipDeviceGetPCIBusId(bus, sizeof(bus), hip_device));
  unsigned dom = 0, b = 0, d = 0, fn = 0;
  if (std::sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &fn) != 4) { set_error("sdma: cannot parse PCI bus id '%s'", bus); return DABX_E_HIP; }
  Find f{};
  f.domain = dom; f.bdf = ((b & 0xFFu) << 8) | ((d & 0x1Fu) << 3) | (fn & 7u);
  if (hsa_iterate_agents(agent_cb, &f) != HSA_STATUS_SUCCESS || !f.found) {
    set_error("sdma: no HSA agent for HIP device %d (%s)", hip_device, bus);
    return DABX_E_HIP;
  }
  if (!f.have_cpu) { set_error("sdma: the HSA runtime lists no CPU agent"); return DABX_E_HIP; }
  out->cpu_agent = f.cpu.handle;
  out->gpu_agent = f.gpu.handle;
  // one engine per direction, from the runtime's preferred set for this pair of agents (lowest bit); 0 = leave it to the runtime
  uint32_t pref = 0;
  if (hsa_amd_memory_get_preferred_copy_engine(f.cpu, f.gpu, &pref) == HSA_STATUS_SUCCESS && pref) out->engine_to_host = pref & (~pref + 1u);
  pref = 0;
  if (hsa_amd_memory_get_preferred_copy_engine(f.gpu, f.cpu, &pref) == HSA_STATUS_SUCCESS && pref) out->engine_to_dev = pref & (~pref + 1u);
  out->ok = true;
  return 0;
}

int sdma_signal_create(uint64_t *sig)
{
  hsa_signal_t s{};
  if (hsa_signal_create(0, 0, nullptr, &s) != HSA_STATUS_SUCCESS) { set_error("sdma: hsa_signal_create failed"); return DABX_E_HIP; }
  *sig = s.handle;
  return 0;
}

void sdma_signal_destroy(uint64_t sig)
{
  if (sig) (void)hsa_signal_destroy(hsa_signal_t{sig});
}

static int sdma_copy_impl(const Sdma &s, void *dst, const void *src, size_t bytes, bool to_host, uint64_t sig, bool strict_engine);
int sdma_copy(const Sdma &s, void *dst, const void *src, size_t bytes, bool to_host, uint64_t sig) { return sdma_copy_impl(s, dst, src, bytes, to_host, sig, false); }
// strict_engine (the calibration's probes): the engine asked for or an error -- no silent fall-back to the runtime's own choice, whose rate would
// otherwise be credited to an engine the runtime refuses
static int sdma_copy_impl(const Sdma &s, void *dst, const void *src, size_t bytes, bool to_host, uint64_t sig, bool strict_engine)
{
  if (!s.ok) { set_error("sdma: not open"); return DABX_E_STATE; }
  // the agent that owns the host allocation (the NUMA node hipHostMalloc took it from); any CPU agent would do for the engine choice
  hsa_agent_t host{s.cpu_agent};
  hsa_amd_pointer_info_t info;
  std::memset(&info, 0, sizeof(info));
  info.size = sizeof(info);
  const void *hp = to_host ? dst : src;
  if (hsa_amd_pointer_info(hp, &info, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS && info.type != HSA_EXT_POINTER_TYPE_UNKNOWN) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_CPU) host = info.agentOwner;
  } else {
    set_error("sdma: %p is not page-locked memory the runtime knows (hipHostMalloc / hipHostRegister)", hp);
    return DABX_E_ARG;
  }
  const hsa_agent_t gpu{s.gpu_agent};
  const hsa_signal_t done{sig};
  hsa_signal_store_relaxed(done, 1);
  const uint32_t engine = to_host ? s.engine_to_host : s.engine_to_dev;
  hsa_status_t st = HSA_STATUS_ERROR;
  if (engine)
    st = to_host ? hsa_amd_memory_async_copy_on_engine(dst, host, src, gpu, bytes, 0, nullptr, done, (hsa_amd_sdma_engine_id_t)engine, true)
                 : hsa_amd_memory_async_copy_on_engine(dst, gpu, src, host, bytes, 0, nullptr, done, (hsa_amd_sdma_engine_id_t)engine, true);
  if (st != HSA_STATUS_SUCCESS && !(strict_engine && engine))   // no preferred engine known, or the runtime refuses it: its own choice
    st = to_host ? hsa_amd_memory_async_copy(dst, host, src, gpu, bytes, 0, nullptr, done)
                 : hsa_amd_memory_async_copy(dst, gpu, src, host, bytes, 0, nullptr, done);
  if (st != HSA_STATUS_SUCCESS) {
    const char *msg = nullptr;
    (void)hsa_status_string(st, &msg);
    set_error("sdma: hsa_amd_memory_async_copy failed: %s", msg ? msg : "?");
    hsa_signal_store_relaxed(done, 0);
    return DABX_E_HIP;
  }
  return 0;
}

int sdma_calibrate(Sdma &s, void *host, void *dev, bool to_host, uint64_t sig, double *gbps)
{
  constexpr size_t N = (size_t)16 << 20;
  uint32_t &engine = to_host ? s.engine_to_host : s.engine_to_dev;
  auto measure = [&](uint32_t eng, double *out, bool strict) -> int {
    const uint32_t keep = engine;
    engine = eng;
    double best = 0;
    int rc = 0;
    for (int r = 0; r < 2 && !rc; r++) {             // the first transfer of a queue includes its creation
      const auto t0 = std::chrono::steady_clock::now();
      rc = sdma_copy_impl(s, to_host ? host : dev, to_host ? dev : host, N, to_host, sig, strict);
      if (!rc) {
        const hsa_signal_t sg{sig};
        hsa_signal_value_t v;
        while ((v = hsa_signal_load_scacquire(sg)) >= 1) {                              // 0.3 ms: spinning is the measurement
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            // the transfer is still in flight and the caller frees both buffers on an error: wait it out (bounded) before saying so
            const int late = sdma_wait(sig, 0);
            engine = keep;
            set_error(late ? "sdma: a 16-MiB probe transfer did not complete within 65 s" : "sdma: a 16-MiB probe transfer took more than 5 s");
            return DABX_E_HIP;
          }
        }
        if (v < 0) { engine = keep; set_error("sdma: a 16-MiB probe transfer failed (signal value %lld)", (long long)v); return DABX_E_HIP; }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        best = std::max(best, (double)N / dt / 1e9);
      }
    }
    engine = keep;
    *out = best;
    return rc;
  };
  double rate = 0;
  if (int rc = measure(engine, &rate, false)) return rc;
  if (rate < 35.0) {
    uint32_t best_eng = engine;
    for (int b = 0; b < 8; b++) {
      double r = 0;
      if (measure(1u << b, &r, true) == 0 && r > rate * 1.15) { rate = r; best_eng = 1u << b; }   // an engine the runtime refuses is skipped
    }
    engine = best_eng;
  }
  if (gbps) *gbps = rate;
  return 0;
}

int sdma_wait(uint64_t sig, size_t bytes_hint)
{
  // Polled, not blocked on the signal's interrupt: sleeping for the time the link needs at least and then looking every 20 us keeps no core
  // busy, adds 10 us on average and does not depend on the interrupt path.  (The 37-ms stalls first blamed on missed interrupts were the
  // Python consumer's garbage collector, docs/history/r05.md; the poll was kept because it costs nothing.)
  const hsa_signal_t s{sig};
  if (bytes_hint) std::this_thread::sleep_for(std::chrono::nanoseconds((long long)(bytes_hint / 60.0)));       // no transfer beats 60 GB/s: sleep that long first
  const auto t0 = std::chrono::steady_clock::now();
  hsa_signal_value_t v;
  while ((v = hsa_signal_load_scacquire(s)) >= 1) {
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    // (a slab is at most a few GB: seconds even on the slowest engine; a transfer that never completes must not hang the caller silently)
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { set_error("sdma: transfer did not complete within 60 s"); return DABX_E_HIP; }
  }
  if (v < 0) { set_error("sdma: the transfer failed (the runtime set its signal to %lld)", (long long)v); return DABX_E_HIP; }      // the runtime's way to report a fault
  return 0;
}

}  // namespace dabx
