// pcie_probe.hip -- what the host link of THIS box gives a 100-MB transfer, each way, by every means the library could use:
//   memcpy1    one hipMemcpyAsync between page-locked host memory and device memory (one SDMA engine)
//   memcpyN    the same buffer as N pieces on N HIP streams (several SDMA engines)
//   kernel     a copy kernel reading / writing the page-locked buffer through its device mapping (the CUs' own loads / stores over the link)
//   pageable   one hipMemcpy from / to malloc'ed memory (the runtime stages it)
// and where the GPU sits: PCI bus id, NUMA node of the device, CPUs of that node, the CPUs this process may run on.
// One JSON line.   hipcc --offload-arch=gfx950 -O3 -o tools/_build/pcie_probe tools/pcie_probe.hip ; tools/_build/pcie_probe [MiB] [numa: -1 keep | n pin to node n's CPUs]
#include <hip/hip_runtime.h>
#include <sched.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static std::string slurp(const std::string &p)
{
  std::ifstream f(p);
  std::stringstream ss;
  ss << f.rdbuf();
  std::string s = ss.str();
  while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
  return s;
}
static std::vector<int> parse_cpulist(const std::string &s)
{
  std::vector<int> out;
  std::stringstream ss(s);
  std::string tok;
  while (std::getline(ss, tok, ',')) {
    int a, b;
    if (std::sscanf(tok.c_str(), "%d-%d", &a, &b) == 2) for (int i = a; i <= b; i++) out.push_back(i);
    else if (std::sscanf(tok.c_str(), "%d", &a) == 1) out.push_back(a);
  }
  return out;
}

int main(int argc, char **argv)
{
  const size_t mib = argc > 1 ? (size_t)std::atoi(argv[1]) : 96;
  const int pin_node = argc > 2 ? std::atoi(argv[2]) : -1;
  const size_t bytes = mib << 20;
  char bus[64] = "?";
  CK(hipSetDevice(0));
  CK(hipDeviceGetPCIBusId(bus, sizeof(bus), 0));
  std::string bdf(bus);
  for (auto &c : bdf) c = (char)tolower(c);
  const std::string numa = slurp("/sys/bus/pci/devices/" + bdf + "/numa_node");
  const std::string local_cpus = slurp("/sys/bus/pci/devices/" + bdf + "/local_cpulist");
  std::string node_cpus;
  if (pin_node >= 0) {
    node_cpus = slurp("/sys/devices/system/node/node" + std::to_string(pin_node) + "/cpulist");
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : parse_cpulist(node_cpus)) CPU_SET(c, &set);
    if (sched_setaffinity(0, sizeof(set), &set) != 0) std::perror("sched_setaffinity");
  }
  cpu_set_t cur;
  CPU_ZERO(&cur);
  sched_getaffinity(0, sizeof(cur), &cur);
  int n_allowed = CPU_COUNT(&cur), first_allowed = -1;
  for (int c = 0; c < CPU_SETSIZE && first_allowed < 0; c++) if (CPU_ISSET(c, &cur)) first_allowed = c;

  void *dev = nullptr, *pinned = nullptr, *pinned_dev = nullptr, *reg = nullptr, *reg_raw = nullptr;
  CK(hipMalloc(&dev, bytes));
  CK(hipMemset(dev, 1, bytes));
  CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));       // allocated (and first touched) AFTER the affinity was set
  std::memset(pinned, 2, bytes);
  CK(hipHostGetDevicePointer(&pinned_dev, pinned, 0));
  reg_raw = std::malloc(bytes + 4096);
  reg = (void *)(((uintptr_t)reg_raw + 4095) & ~(uintptr_t)4095);
  std::memset(reg, 3, bytes);
  const bool registered = hipHostRegister(reg, bytes, hipHostRegisterDefault) == hipSuccess;
  void *pageable = std::malloc(bytes);
  std::memset(pageable, 4, bytes);
  constexpr int NS = 4;
  hipStream_t st[NS];
  for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev[NS];
  for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

  auto timeit = [&](auto &&fn) {
    double best = 1e30;
    for (int r = 0; r < 6; r++) {
      (void)hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      fn();
      (void)hipDeviceSynchronize();
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (r > 0 && dt < best) best = dt;
    }
    return (double)bytes / best / 1e9;
  };
  auto split = [&](void *dst, const void *src, hipMemcpyKind kind, int n) {
    const size_t piece = (bytes / n + 255) & ~(size_t)255;
    for (int i = 0; i < n; i++) {
      const size_t o = piece * i, len = o >= bytes ? 0 : (bytes - o < piece ? bytes - o : piece);
      if (len) (void)hipMemcpyAsync((char *)dst + o, (const char *)src + o, len, kind, st[i % NS]);
    }
  };
  std::printf("{\"MiB\": %zu, \"pci_bus_id\": \"%s\", \"gpu_numa_node\": \"%s\", \"gpu_local_cpulist\": \"%s\", \"pinned_to_node\": %d, \"allowed_cpus\": %d, \"first_allowed_cpu\": %d",
              mib, bus, numa.c_str(), local_cpus.c_str(), pin_node, n_allowed, first_allowed);
  std::printf(", \"d2h_GBps\": {\"memcpy1_hostmalloc\": %.2f", timeit([&] { (void)hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, st[0]); }));
  std::printf(", \"memcpy2_hostmalloc\": %.2f", timeit([&] { split(pinned, dev, hipMemcpyDeviceToHost, 2); }));
  std::printf(", \"memcpy4_hostmalloc\": %.2f", timeit([&] { split(pinned, dev, hipMemcpyDeviceToHost, 4); }));
  if (registered) std::printf(", \"memcpy1_hostregister\": %.2f", timeit([&] { (void)hipMemcpyAsync(reg, dev, bytes, hipMemcpyDeviceToHost, st[0]); }));
  for (int blocks : {64, 256, 1024})
    std::printf(", \"kernel_%d_blocks\": %.2f", blocks, timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st[0], (const uint4 *)dev, (uint4 *)pinned_dev, bytes / 16); }));
  std::printf(", \"pageable\": %.2f}", timeit([&] { (void)hipMemcpy(pageable, dev, bytes, hipMemcpyDeviceToHost); }));
  std::printf(", \"h2d_GBps\": {\"memcpy1_hostmalloc\": %.2f", timeit([&] { (void)hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st[0]); }));
  std::printf(", \"memcpy2_hostmalloc\": %.2f", timeit([&] { split(dev, pinned, hipMemcpyHostToDevice, 2); }));
  std::printf(", \"memcpy4_hostmalloc\": %.2f", timeit([&] { split(dev, pinned, hipMemcpyHostToDevice, 4); }));
  if (registered) std::printf(", \"memcpy1_hostregister\": %.2f", timeit([&] { (void)hipMemcpyAsync(dev, reg, bytes, hipMemcpyHostToDevice, st[0]); }));
  for (int blocks : {64, 256, 1024})
    std::printf(", \"kernel_%d_blocks\": %.2f", blocks, timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st[0], (const uint4 *)pinned_dev, (uint4 *)dev, bytes / 16); }));
  std::printf(", \"pageable\": %.2f}", timeit([&] { (void)hipMemcpy(dev, pageable, bytes, hipMemcpyHostToDevice); }));
  // both directions at once (the receiver ingests while it delivers)
  std::printf(", \"bidir_GBps_each\": {\"memcpy1\": %.2f}}\n", timeit([&] {
    (void)hipMemcpyAsync(dev, reg, bytes, hipMemcpyHostToDevice, st[0]);
    (void)hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, st[1]); }));
  return 0;
}
