// sdma_engines.hip -- rate of every SDMA engine of the GPU for a 96-MiB device <-> page-locked host transfer (hsa_amd_memory_async_copy_on_engine),
// what the runtime reports as free / preferred for that pair of agents, and what hsa_amd_memory_async_copy (engine chosen by the runtime) gets.
// One JSON line.   hipcc --offload-arch=gfx950 -O3 -o tools/_build/sdma_engines tools/sdma_engines.hip -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t a, void *)
{
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
int main(int argc, char **argv)
{
  const size_t bytes = (size_t)(argc > 1 ? std::atoi(argv[1]) : 96) << 20;
  CK(hipSetDevice(0));
  void *dev = nullptr, *pinned = nullptr;
  CK(hipMalloc(&dev, bytes)); CK(hipMemset(dev, 1, bytes));
  CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault)); std::memset(pinned, 2, bytes);
  CK(hipDeviceSynchronize());
  if (hsa_init() != HSA_STATUS_SUCCESS || hsa_iterate_agents(agent_cb, nullptr) != HSA_STATUS_SUCCESS || !g_have_gpu || !g_have_cpu) { std::fprintf(stderr, "no HSA agents\n"); return 1; }
  hsa_amd_pointer_info_t info; std::memset(&info, 0, sizeof(info)); info.size = sizeof(info);
  if (hsa_amd_pointer_info(pinned, &info, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_CPU) g_cpu = info.agentOwner;
  }
  hsa_signal_t sig;
  hsa_signal_create(1, 0, nullptr, &sig);
  uint32_t st_d2h = 0, st_h2d = 0, pref_d2h = 0, pref_h2d = 0;
  const hsa_status_t s1 = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &st_d2h);
  const hsa_status_t s2 = hsa_amd_memory_copy_engine_status(g_gpu, g_cpu, &st_h2d);
  (void)hsa_amd_memory_get_preferred_copy_engine(g_cpu, g_gpu, &pref_d2h);
  (void)hsa_amd_memory_get_preferred_copy_engine(g_gpu, g_cpu, &pref_h2d);
  std::printf("{\"MiB\": %zu, \"engine_status_d2h\": \"0x%x (rc %d)\", \"engine_status_h2d\": \"0x%x (rc %d)\", \"preferred_d2h\": \"0x%x\", \"preferred_h2d\": \"0x%x\"",
              bytes >> 20, st_d2h, (int)s1, st_h2d, (int)s2, pref_d2h, pref_h2d);
  auto run = [&](bool d2h, int engine /* < 0: runtime's choice */) {
    double best = 1e30;
    for (int r = 0; r < 4; r++) {
      hsa_signal_store_relaxed(sig, 1);
      const auto t0 = std::chrono::steady_clock::now();
      hsa_status_t st;
      void *dst = d2h ? pinned : dev; const void *src = d2h ? dev : pinned;
      hsa_agent_t da = d2h ? g_cpu : g_gpu, sa = d2h ? g_gpu : g_cpu;
      if (engine < 0) st = hsa_amd_memory_async_copy(dst, da, src, sa, bytes, 0, nullptr, sig);
      else st = hsa_amd_memory_async_copy_on_engine(dst, da, src, sa, bytes, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)(1u << engine), true);
      if (st != HSA_STATUS_SUCCESS) return -1.0;
      hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (r > 0 && dt < best) best = dt;
    }
    return (double)bytes / best / 1e9;
  };
  for (int d2h = 1; d2h >= 0; d2h--) {
    std::printf(", \"%s_GBps\": {\"runtime_choice\": %.2f", d2h ? "d2h" : "h2d", run(d2h, -1));
    for (int e = 0; e < 16; e++) {
      const double g = run(d2h, e);
      if (g >= 0) std::printf(", \"engine_%d\": %.2f", e, g);
    }
    std::printf(", \"runtime_choice_again\": %.2f}", run(d2h, -1));
  }
  std::printf("}\n");
  return 0;
}
