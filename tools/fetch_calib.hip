// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the ACCESS WIDTHS this code base
// uses.  MI355X_MICROARCH.md (HBM): FETCH_SIZE reads exactly half of a 16-B-per-lane coalesced stream and "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below moves
// exactly BYTES bytes (1 GiB, four times the Infinity Cache) once; the counter value / BYTES is the correction factor.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/cal_f --output-format csv -- build/fetch_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/cal_w --output-format csv -- build/fetch_calib
//   python3 tools/fetch_calib_summary.py gpurun_out/cal_f gpurun_out/cal_w > profiles/r02_fetch_calib.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr size_t BYTES = 1ull << 30;

// ---- reads: every lane accumulates, one store per block at the end (negligible)
template <class T> __device__ unsigned fold(T v);
template <> __device__ unsigned fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ unsigned fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <> __device__ unsigned fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <> __device__ unsigned fold<uint8_t>(uint8_t v) { return v; }

template <class T> __global__ __launch_bounds__(256) void cal_read_coalesced(const T *p, size_t n, unsigned *sink)
{
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= fold(p[i]);
  if (acc == 0x9E3779B9u) sink[0] = acc;
}
// k_msc_prep's pattern: a wave reads 16-dword (64-byte) runs, runs `stride` bytes apart (one HBM line each)
__global__ __launch_bounds__(256) void cal_read_runs64(const uint32_t *p, size_t n_runs, size_t stride_dw, unsigned *sink)
{
  unsigned acc = 0;
  const int d = threadIdx.x & 15;
  for (size_t r = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4; r < n_runs; r += ((size_t)gridDim.x * 256) >> 4) acc ^= p[r * stride_dw + d];
  if (acc == 0x9E3779B9u) sink[0] = acc;
}
// k_msc_vitT's pattern: lane-strided dwords inT[row][lane] -- coalesced 256-B rows
__global__ __launch_bounds__(64) void cal_read_rows256(const uint32_t *p, size_t rows_per_block, unsigned *sink)
{
  unsigned acc = 0;
  const uint32_t *q = p + (size_t)blockIdx.x * rows_per_block * 64 + threadIdx.x;
  for (size_t r = 0; r < rows_per_block; r++) acc ^= q[r * 64];
  if (acc == 0x9E3779B9u) sink[0] = acc;
}
// ---- writes
template <class T> __global__ __launch_bounds__(256) void cal_write_coalesced(T *p, size_t n, T v)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
// one wave writes rows of 64 x 8 B (decision words dec[t][lane]) / 64 x 4 B
template <class T> __global__ __launch_bounds__(64) void cal_write_rows(T *p, size_t rows_per_block, T v)
{
  T *q = p + (size_t)blockIdx.x * rows_per_block * 64 + threadIdx.x;
  for (size_t r = 0; r < rows_per_block; r++) q[r * 64] = v;
}
// per-lane scattered dword: lane stride `stride_dw` dwords (k_msc_vitT's packed output words)
__global__ __launch_bounds__(64) void cal_write_scatter(uint32_t *p, size_t per_lane, size_t stride_dw, uint32_t v)
{
  uint32_t *q = p + ((size_t)blockIdx.x * 64 + threadIdx.x) * stride_dw;
  for (size_t r = 0; r < per_lane; r++) q[r] = v;
}

int main()
{
  void *buf; unsigned *sink;
  CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, BYTES));
  CK(hipDeviceSynchronize());
  const int grid = 256 * 16;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(cal_read_coalesced<uint8_t>, dim3(grid), dim3(256), 0, 0, (const uint8_t *)buf, BYTES / 4, sink);      // 256 MiB only (slow pattern)
    hipLaunchKernelGGL(cal_read_coalesced<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, BYTES / 4, sink);
    hipLaunchKernelGGL(cal_read_coalesced<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2 *)buf, BYTES / 8, sink);
    hipLaunchKernelGGL(cal_read_coalesced<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4 *)buf, BYTES / 16, sink);
    hipLaunchKernelGGL(cal_read_runs64, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, BYTES / 64 / 54, (size_t)54 * 16, sink);   // 1 line in 54 (3456-B plane pitch)
    hipLaunchKernelGGL(cal_read_rows256, dim3(4096), dim3(64), 0, 0, (const uint32_t *)buf, BYTES / 256 / 4096, sink);
    hipLaunchKernelGGL(cal_write_coalesced<uint32_t>, dim3(grid), dim3(256), 0, 0, (uint32_t *)buf, BYTES / 4, 7u);
    hipLaunchKernelGGL(cal_write_coalesced<uint4>, dim3(grid), dim3(256), 0, 0, (uint4 *)buf, BYTES / 16, make_uint4(1, 2, 3, 4));
    hipLaunchKernelGGL(cal_write_rows<uint2>, dim3(4096), dim3(64), 0, 0, (uint2 *)buf, BYTES / 512 / 4096, make_uint2(5, 6));
    hipLaunchKernelGGL(cal_write_rows<uint32_t>, dim3(4096), dim3(64), 0, 0, (uint32_t *)buf, BYTES / 256 / 4096, 9u);
    hipLaunchKernelGGL(cal_write_scatter, dim3(4096), dim3(64), 0, 0, (uint32_t *)buf, (size_t)48, BYTES / 4 / (4096 * 64), 11u);   // 48 dwords per lane, lanes 4 KiB apart
    CK(hipDeviceSynchronize());
  }
  printf("{\"bytes\": {\"cal_read_coalesced<unsigned char>\": %zu, \"cal_read_coalesced<unsigned int>\": %zu, \"cal_read_coalesced<HIP_vector_type<unsigned int, 2u>>\": %zu, "
         "\"cal_read_coalesced<HIP_vector_type<unsigned int, 4u>>\": %zu, \"cal_read_runs64\": %zu, \"cal_read_rows256\": %zu, \"cal_write_coalesced<unsigned int>\": %zu, "
         "\"cal_write_coalesced<HIP_vector_type<unsigned int, 4u>>\": %zu, \"cal_write_rows<HIP_vector_type<unsigned int, 2u>>\": %zu, \"cal_write_rows<unsigned int>\": %zu, "
         "\"cal_write_scatter\": %zu}}\n",
         BYTES / 4, BYTES, BYTES, BYTES, (BYTES / 64 / 54) * 64, BYTES, BYTES, BYTES, BYTES, BYTES, (size_t)4096 * 64 * 48 * 4);
  return 0;
}
