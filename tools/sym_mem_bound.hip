// sym_mem_bound.hip -- how fast can the MEMORY side of k_symbols go?  The same grid (75 x 512 blocks of 256 threads), the same
// per-block traffic (a contiguous 2552-sample slice of a stream's ring read once, 12 of the samples per thread in one go; 1536
// float2 written contiguously), the same residency (17.4 KB of LDS per block: 8 blocks per CU) -- and no transform: a few
// FLOPs, one barrier.  Variants: (a) as above, (b) with four more barriers and an LDS round trip each (the transform's
// synchronisation skeleton without its arithmetic).  k_symbols itself takes 0.29 ms per 512-stream step.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/sym_mem_bound tools/sym_mem_bound.hip && build/sym_mem_bound
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int TU = 2048, TG = 504, TS = 2552, K = 1536, TF = 196608, S = 512, RING = 10 * TF;

template <int SKEL>
__global__ __launch_bounds__(256, 8) void k_stub(const float2 *iq, float2 *spectra, int frame)
{
  __shared__ float2 lds[2048 + 128];
  const int s = blockIdx.y, l = blockIdx.x, tid = threadIdx.x;
  const float2 *ring = iq + (size_t)s * RING;
  unsigned off = (unsigned)(((size_t)frame * TF + 2656 + (size_t)l * TS) % RING);
  auto at = [&](unsigned i) { unsigned o = off + i; if (o >= RING) o -= RING; return ring[o]; };
  const bool two = tid + 256 < TG;
  const float2 cb0 = at(tid), ca0 = at(TU + tid), cb1 = at(two ? tid + 256 : tid), ca1 = at(two ? TU + tid + 256 : TU + tid);
  float2 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) v[u] = at(TG + tid + 256 * u);
  float acc = ca0.x * cb0.x + ca0.y * cb0.y + ca1.x * cb1.y;
#pragma unroll
  for (int u = 0; u < 8; u++) { v[u].x = v[u].x * 0.5f + acc; v[u].y = v[u].y * 0.25f - acc; }
  if (SKEL) {
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
      for (int u = 0; u < 8; u++) lds[(tid * 8 + u) + ((tid * 8 + u) >> 4)] = v[u];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = tid + 256 * u; v[u] = lds[i + (i >> 4)]; }
      __syncthreads();
    }
  }
#pragma unroll
  for (int u = 0; u < 8; u++) { const int i = tid + 256 * u; if (i < K + 256) lds[i] = v[u]; }
  __syncthreads();
  float2 *dst = spectra + ((size_t)s * 75 + l) * K;
#pragma unroll
  for (int u = 0; u < K / 256; u++) dst[tid + 256 * u] = lds[tid + 256 * u];
}

// (c) persistent blocks: grid (G, S), a block walks symbols l = g, g + G, ... with the NEXT symbol's 12 samples per thread
// requested before the current one is "transformed" (WORK dependent FMAs per register stand in for the arithmetic) and stored;
// BLOCKS_PER_CU residency via the launch bound (registers) -- does the memory side still reach its floor with 4 or 5 blocks per
// CU when every block always has a symbol's loads in flight?
template <int WAVES_PER_EU, int WORK>
__global__ __launch_bounds__(256, WAVES_PER_EU) void k_stub_persistent(const float2 *iq, float2 *spectra, int frame, int G)
{
  __shared__ float2 lds[2048 + 128];
  const int s = blockIdx.y, tid = threadIdx.x;
  const float2 *ring = iq + (size_t)s * RING;
  float2 nx[12];
  auto request = [&](int l) {
    const unsigned off = (unsigned)(((size_t)frame * TF + 2656 + (size_t)l * TS) % RING);
    auto at = [&](unsigned i) { unsigned o = off + i; if (o >= RING) o -= RING; return ring[o]; };
    const bool two = tid + 256 < TG;
    nx[0] = at(tid); nx[1] = at(TU + tid); nx[2] = at(two ? tid + 256 : tid); nx[3] = at(two ? TU + tid + 256 : TU + tid);
#pragma unroll
    for (int u = 0; u < 8; u++) nx[4 + u] = at(TG + tid + 256 * u);
  };
  request(blockIdx.x);
  for (int l = blockIdx.x; l < 75; l += G) {
    float2 v[8];
    float acc = nx[1].x * nx[0].x + nx[1].y * nx[0].y + nx[3].x * nx[2].y;
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = nx[4 + u];
    if (l + G < 75) request(l + G);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; u++) { v[u].x = v[u].x * 0.5f + acc; v[u].y = v[u].y * 0.25f - acc; }
#pragma unroll 1
    for (int w = 0; w < WORK; w++)
#pragma unroll
      for (int u = 0; u < 8; u++) { v[u].x = __builtin_fmaf(v[u].x, 0.999f, v[u].y); v[u].y = __builtin_fmaf(v[u].y, 1.001f, -v[u].x); }
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
      for (int u = 0; u < 8; u++) lds[(tid * 8 + u) + ((tid * 8 + u) >> 4)] = v[u];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = tid + 256 * u; v[u] = lds[i + (i >> 4)]; }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 8; u++) { const int i = tid + 256 * u; if (i < K + 256) lds[i] = v[u]; }
    __syncthreads();
    float2 *dst = spectra + ((size_t)s * 75 + l) * K;
#pragma unroll
    for (int u = 0; u < K / 256; u++) dst[tid + 256 * u] = lds[tid + 256 * u];
    __syncthreads();
  }
}

// (d) the same traffic with 16-byte accesses: a thread loads five float4 (two consecutive samples each; the symbol's first
// sample is only 8-byte aligned in the ring, MISALIGN = 1 shifts the slice by one sample) and stores three float4.
template <int MISALIGN>
__global__ __launch_bounds__(256, 8) void k_stub16(const float2 *iq, float2 *spectra, int frame)
{
  const int s = blockIdx.y, l = blockIdx.x, tid = threadIdx.x;
  const float2 *ring = iq + (size_t)s * RING;
  const unsigned off = ((unsigned)(((size_t)frame * TF + 2656 + (size_t)l * TS) % RING) & ~1u) + MISALIGN;
  float4 v[5];
#pragma unroll
  for (int u = 0; u < 5; u++) {
    unsigned i = off + 2 * (tid + 256 * u);
    if (i + 1 >= RING) i = 0;
    v[u] = (tid + 256 * u) < TS / 2 ? *reinterpret_cast<const float4 *>(ring + i) : make_float4(0, 0, 0, 0);
  }
  const float acc = v[4].x * 1e-9f;
  float4 *dst = reinterpret_cast<float4 *>(spectra + ((size_t)s * 75 + l) * K);
#pragma unroll
  for (int u = 0; u < 3; u++) dst[tid + 256 * u] = make_float4(v[u].x * 0.5f + acc, v[u].y, v[u].z * 0.25f, v[u].w + v[(u + 1) % 5].x);
}

// (e) does the memory-side cache (256 MB) keep a group's spectra between the producer and the consumer?  k_consume reads a
// group's spectra the way the demapper does (every float2 once).  Full: producer over 512 streams, consumer over 512 streams
// (472 MB of spectra in between).  Grouped: NG groups of 512 / NG streams, each produced into the SAME spectra region and
// consumed right away (118 MB for NG = 4).  Equal times = the round trip goes to HBM either way.
__global__ __launch_bounds__(256, 8) void k_consume(const float2 *spectra, float *sink)
{
  const int s = blockIdx.y, l = blockIdx.x, tid = threadIdx.x;
  const float2 *src = spectra + ((size_t)s * 75 + l) * K;
  float acc = 0.0f;
#pragma unroll
  for (int u = 0; u < K / 256; u++) { const float2 v = src[tid + 256 * u]; acc += v.x + v.y; }
  if (acc == 12345.678f) sink[0] = acc;
}
static void run_grouped(const float2 *iq, float2 *sp, float *sink, int NG, hipEvent_t a, hipEvent_t b)
{
  std::vector<float> ms;
  const int Sg = S / NG;
  for (int it = 0; it < 24; it++) {
    CK(hipEventRecord(a));
    for (int g = 0; g < NG; g++) {
      hipLaunchKernelGGL(k_stub<0>, dim3(75, Sg), dim3(256), 0, 0, iq + (size_t)g * Sg * RING, NG == 1 ? sp : sp, it % 10);
      hipLaunchKernelGGL(k_consume, dim3(75, Sg), dim3(256), 0, 0, sp, sink);
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    if (it >= 4) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  printf("{\"variant\": \"producer + consumer, %d group(s) of %d streams through one %d-MB spectra region\", \"median_ms\": %.4f, \"min_ms\": %.4f}\n",
         NG, Sg, (int)((size_t)Sg * 75 * K * 8 >> 20), ms[ms.size() / 2], ms[0]);
}

template <int WPE, int WORK> static void run_persistent(const float2 *iq, float2 *sp, int G, hipEvent_t a, hipEvent_t b)
{
  std::vector<float> ms;
  for (int it = 0; it < 24; it++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_stub_persistent<WPE, WORK>), dim3(G, S), dim3(256), 0, 0, iq, sp, it % 10, G);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    if (it >= 4) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2], bytes = (double)S * 75 * (TS + K) * 8;
  printf("{\"variant\": \"persistent, next symbol prefetched, %d waves/SIMD bound, G = %d, work = %d x 16 dependent FMAs\", \"median_ms\": %.4f, \"min_ms\": %.4f, \"GBps\": %.0f}\n",
         WPE, G, WORK, med, ms[0], bytes / med / 1e6);
}

int main()
{
  float2 *iq, *sp;
  CK(hipMalloc(&iq, (size_t)S * RING * sizeof(float2)));
  CK(hipMalloc(&sp, (size_t)S * 75 * K * sizeof(float2)));
  CK(hipMemset(iq, 0, (size_t)S * RING * sizeof(float2)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int skel = 0; skel < 2; skel++) {
    std::vector<float> ms;
    for (int it = 0; it < 24; it++) {
      CK(hipEventRecord(a));
      if (skel) hipLaunchKernelGGL(k_stub<1>, dim3(75, S), dim3(256), 0, 0, iq, sp, it % 10);
      else hipLaunchKernelGGL(k_stub<0>, dim3(75, S), dim3(256), 0, 0, iq, sp, it % 10);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float t; CK(hipEventElapsedTime(&t, a, b));
      if (it >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2], bytes = (double)S * 75 * (TS + K) * 8;
    printf("{\"variant\": \"%s\", \"median_ms\": %.4f, \"min_ms\": %.4f, \"GBps\": %.0f, \"bytes\": %.0f}\n",
           skel ? "loads + 3 LDS exchanges with barriers + store" : "loads + store", med, ms[0], bytes / med / 1e6, bytes);
  }
  for (int mis = 0; mis < 2; mis++) {
    std::vector<float> ms;
    for (int it = 0; it < 24; it++) {
      CK(hipEventRecord(a));
      if (mis) hipLaunchKernelGGL(k_stub16<1>, dim3(75, S), dim3(256), 0, 0, iq, sp, it % 10);
      else hipLaunchKernelGGL(k_stub16<0>, dim3(75, S), dim3(256), 0, 0, iq, sp, it % 10);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float t; CK(hipEventElapsedTime(&t, a, b));
      if (it >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2], bytes = (double)S * 75 * (TS + K) * 8;
    printf("{\"variant\": \"16-byte loads and stores, %s\", \"median_ms\": %.4f, \"min_ms\": %.4f, \"GBps\": %.0f}\n",
           mis ? "slice 8-byte aligned only" : "slice 16-byte aligned", med, ms[0], bytes / med / 1e6);
  }
  float *sink;
  CK(hipMalloc(&sink, 64));
  for (int ng : {1, 2, 4, 8, 16}) run_grouped(iq, sp, sink, ng, a, b);
  run_persistent<8, 0>(iq, sp, 15, a, b);
  run_persistent<4, 0>(iq, sp, 15, a, b);
  run_persistent<4, 0>(iq, sp, 25, a, b);
  run_persistent<4, 24>(iq, sp, 15, a, b);      // ~ the transform's 400 non-load VALU instructions per thread
  run_persistent<4, 24>(iq, sp, 25, a, b);
  run_persistent<5, 24>(iq, sp, 15, a, b);
  run_persistent<8, 24>(iq, sp, 15, a, b);
  return 0;
}
