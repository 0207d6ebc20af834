// sdma.h -- bulk transfers between device memory and page-locked host memory on the GPU's SDMA engines, through the HSA runtime
// (hsa_amd_memory_async_copy).  Why not hipMemcpyAsync: on this ROCm (7.2) the HIP runtime moves device <-> page-locked host buffers
// with a shader copy (__amd_rocclr_copyBuffer).  It reaches the link's rate, but while it runs the receiver's kernels stand still:
// a 96-MiB copy next to an HBM-bound kernel doubles that kernel's time, the same bytes on an SDMA engine cost it 0.5 %
// (tools/copy_interference.hip, profiles/r05_copy_interference.json).  The delivery's one copy per chunk (engine.cpp) goes this way.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace dabx {

struct Sdma {
  uint64_t gpu_agent = 0;       // hsa_agent_t::handle of the HIP device's agent
  bool ok = false;
};
// Finds the HSA agent of HIP device `hip_device` (matched by PCI domain:bus:device.function).  0, or a dabx error with set_error().
int sdma_open(int hip_device, Sdma *out);
int sdma_signal_create(uint64_t *sig);
void sdma_signal_destroy(uint64_t sig);
// Submits dst <- src (one side device memory of the Sdma's GPU, the other page-locked host memory known to the runtime: hipHostMalloc,
// hipHostRegister) and returns at once; `sig` is armed and completes when the bytes are in place.
int sdma_copy(const Sdma &s, void *dst, const void *src, size_t bytes, bool to_host, uint64_t sig);
// Blocks (no spinning) until the copy armed on `sig` is complete.
int sdma_wait(uint64_t sig);

}  // namespace dabx
