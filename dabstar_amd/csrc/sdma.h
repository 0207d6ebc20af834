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
  uint64_t cpu_agent = 0;       // ... of the first CPU agent (stands in for the owner of a host buffer the runtime cannot name)
  uint32_t engine_to_host = 0;  // hsa_amd_sdma_engine_id_t (one bit) the transfers of each direction are put on: the first of the engines the
  uint32_t engine_to_dev = 0;   // runtime names as preferred for the GPU <-> CPU pair.  Not left to hsa_amd_memory_async_copy: of the MI355X's 16
                                // engines only 0-3 reach the link's rate towards the host (56 GB/s; 4-7: 12.7, 8-15: 7-10 GB/s, tools/sdma_engines.hip),
                                // and which one the runtime takes depends on what else has opened SDMA queues before (measured: a slab took
                                // 10 ms instead of 1.8 ms after another GPU process had run, profiles/r05_sdma_engines.json)
  bool ok = false;
};
// Finds the HSA agent of HIP device `hip_device` (matched by PCI domain:bus:device.function).  0, or a dabx error with set_error().
int sdma_open(int hip_device, Sdma *out);
int sdma_signal_create(uint64_t *sig);
void sdma_signal_destroy(uint64_t sig);
// Submits dst <- src (one side device memory of the Sdma's GPU, the other page-locked host memory known to the runtime: hipHostMalloc,
// hipHostRegister) and returns at once; `sig` is armed and completes when the bytes are in place.
int sdma_copy(const Sdma &s, void *dst, const void *src, size_t bytes, bool to_host, uint64_t sig);
// Makes sure the engine the transfers of one direction go to is a fast one: times a 16-MiB transfer between `host` and `dev` the way
// sdma_copy would make it; below 35 GB/s (an engine that is not wired to the host link at full width: 7-13 GB/s on the MI355X) it tries
// the engines 0..7 one by one and keeps the fastest.  *gbps (optional) = the rate of the engine kept.  Both buffers >= 16 MiB.
int sdma_calibrate(Sdma &s, void *host, void *dev, bool to_host, uint64_t sig, double *gbps);
// Returns when the copy armed on `sig` is complete (sleeps bytes_hint / 60 GB/s, then polls the signal every 20 us: no spinning, no interrupt).
int sdma_wait(uint64_t sig, size_t bytes_hint);

}  // namespace dabx
