// copy_interference.hip -- what a 96-MiB device-to-host transfer costs the kernels that run next to it, by the means the delivery
// could use: a victim kernel (an HBM-bound device-to-device stream, 64 MiB per launch, launched back to back
// on its own HIP stream) is timed alone and then while the transfer repeats next to it:
//   hipMemcpyAsync      what the HIP runtime does for device -> page-locked host (on this image a shader copy, __amd_rocclr_copyBuffer)
//   hsa_sdma            hsa_amd_memory_async_copy between the GPU agent and the CPU agent (an SDMA engine, no CU)
//   kernel_<b>x<t>      our own copy kernel writing the mapped host buffer, b workgroups of t threads, 16 B per lane and store
// One JSON line: transfer rate of each means + the victim's slowdown.   hipcc ... -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_stream(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void k_copy_out(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(src[i], &dst[i]);
}

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
  const size_t vbytes = (size_t)64 << 20;
  CK(hipSetDevice(0));
  void *dev = nullptr, *pinned = nullptr, *pinned_dev = nullptr, *va = nullptr, *vb = nullptr;
  CK(hipMalloc(&dev, bytes)); CK(hipMemset(dev, 1, bytes));
  CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault)); std::memset(pinned, 2, bytes);
  CK(hipHostGetDevicePointer(&pinned_dev, pinned, 0));
  CK(hipMalloc(&va, vbytes)); CK(hipMalloc(&vb, vbytes)); CK(hipMemset(va, 3, vbytes));
  hipStream_t sv, sc;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  CK(hipStreamCreateWithPriority(&sv, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, lo));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const bool hsa_ok = hsa_init() == HSA_STATUS_SUCCESS && hsa_iterate_agents(agent_cb, nullptr) == HSA_STATUS_SUCCESS && g_have_gpu && g_have_cpu;
  hsa_signal_t sig{};
  if (hsa_ok) hsa_signal_create(1, 0, nullptr, &sig);

  constexpr int VN = 400;                       // victim launches per measurement (~ 64 MiB x 2 / 4 TB/s = 34 us each -> ~15 ms)
  auto victim_ms = [&](auto &&start_transfer, auto &&finish_transfer, int reps, double *xfer_gbps) {
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipEventRecord(e0, sv);
    for (int r = 0; r < reps; r++) start_transfer();            // queued back to back on the copy stream / engine
    for (int i = 0; i < VN; i++) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, sv, (const uint4 *)va, (uint4 *)vb, vbytes / 16);
    (void)hipEventRecord(e1, sv);
    finish_transfer();
    const double dt_x = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (xfer_gbps) *xfer_gbps = reps ? (double)bytes * reps / dt_x / 1e9 : 0.0;
    return (double)ms;
  };
  double alone = 1e30;
  for (int r = 0; r < 3; r++) { const double m = victim_ms([] {}, [] {}, 0, nullptr); if (m < alone) alone = m; }
  std::printf("{\"MiB\": %zu, \"victim\": \"64-MiB device-to-device stream x %d launches\", \"victim_alone_ms\": %.3f, \"victim_GBps\": %.0f, \"hsa\": %s", bytes >> 20, VN,
              alone, 2.0 * vbytes * VN / (alone * 1e-3) / 1e9, hsa_ok ? "true" : "false");
  const int REPS = 6;                           // 6 x 96 MiB at ~55 GB/s = 11 ms: most of the victim's run
  auto report = [&](const char *name, double ms, double gbps) {
    std::printf(", \"%s\": {\"transfer_GBps\": %.2f, \"victim_ms\": %.3f, \"victim_slowdown\": %.3f}", name, gbps, ms, ms / alone);
    std::fflush(stdout);
  };
  {
    double g = 0;
    const double ms = victim_ms([&] { (void)hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, sc); }, [&] { (void)hipStreamSynchronize(sc); }, REPS, &g);
    report("hipMemcpyAsync", ms, g);
  }
  if (hsa_ok) {
    double g = 0;
    const double ms = victim_ms([&] {}, [&] {
      for (int r = 0; r < REPS; r++) {
        hsa_signal_store_relaxed(sig, 1);
        if (hsa_amd_memory_async_copy(pinned, g_cpu, dev, g_gpu, bytes, 0, nullptr, sig) != HSA_STATUS_SUCCESS) { std::fprintf(stderr, "hsa copy failed\n"); break; }
        hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
      } }, REPS, &g);
    report("hsa_sdma", ms, g);
    // explicitly on one SDMA engine
    double g2 = 0;
    const double ms2 = victim_ms([&] {}, [&] {
      for (int r = 0; r < REPS; r++) {
        hsa_signal_store_relaxed(sig, 1);
        if (hsa_amd_memory_async_copy_on_engine(pinned, g_cpu, dev, g_gpu, bytes, 0, nullptr, sig, HSA_AMD_SDMA_ENGINE_0, false) != HSA_STATUS_SUCCESS) { std::fprintf(stderr, "hsa copy_on_engine failed\n"); break; }
        hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
      } }, REPS, &g2);
    report("hsa_sdma_engine0", ms2, g2);
  }
  const int grids[][2] = {{8, 256}, {16, 256}, {32, 256}, {64, 256}, {256, 256}, {16, 64}, {64, 64}};
  for (const auto &gr : grids) {
    double g = 0;
    const double ms = victim_ms([&] { hipLaunchKernelGGL(k_copy_out, dim3(gr[0]), dim3(gr[1]), 0, sc, (const v4u *)dev, (v4u *)pinned_dev, bytes / 16); },
                                [&] { (void)hipStreamSynchronize(sc); }, REPS, &g);
    char name[64];
    std::snprintf(name, sizeof(name), "kernel_%dx%d", gr[0], gr[1]);
    report(name, ms, g);
  }
  std::printf("}\n");
  return 0;
}
