// overlap_probe.hip -- can a VALU-bound and an HBM-bound workload share the CUs of gfx950 when ONE kernel carries both block
// types?  (As separate kernels on two HIP streams the decoder and the FFT kernel time-slice by register-file capacity:
// docs/history/r01-r04_design_notebook.md 6.)  Block type A = "decoder": 4 waves of dependent-free packed-int16 / v_perm arithmetic in the mix of
// k_msc_vitT, ~100 VGPRs by launch bound, no memory traffic.  Block type B = "transform": the persistent prefetching stub
// of tools/sym_mem_bound.hip (a block walks 5 symbols of its stream, next symbol's loads in flight), 96 VGPRs max so that
// both types can be resident on one SIMD.  Kernels: A alone, B alone, and the mixed grid (blockIdx -> type by a
// repeating pattern of NA decoder blocks per transform block).  If the mixed kernel takes about max(A, B) the concept
// works; if it takes A + B the hardware serialises them anyway.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/overlap_probe tools/overlap_probe.hip && build/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int TU = 2048, TG = 504, TS = 2552, K = 1536, TF = 196608, S = 512, RING = 10 * TF, SYM_G = 15;

typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void decoder_work(int trips, unsigned *sink)
{
  // 32 "metric" registers, butterfly pairs: 4 add/sub + 2 min + 2 sub + perm + and_or per pair (the mix of vit_t_gen.h)
  s2 R[32];
#pragma unroll
  for (int i = 0; i < 32; i++) R[i] = (s2){(short)(threadIdx.x + i), (short)(blockIdx.x + 3 * i)};
  unsigned acc0 = 0, acc1 = 0;
  const s2 M = {(short)7, (short)-5};
  for (int t = 0; t < trips; t++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const s2 a0 = R[k] + M, b0 = R[k + 16] - M, a1 = R[k] - M, b1 = R[k + 16] + M;
      R[k] = __builtin_elementwise_min(a0, b0); R[k + 16] = __builtin_elementwise_min(a1, b1);
      const unsigned p = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, (s2)(b0 - a0)), __builtin_bit_cast(unsigned, (s2)(b1 - a1)), 0x0B0A0908u);
      if (k < 8) acc0 = (p & (0x01010101u << (k & 7))) | acc0; else acc1 = (p & (0x01010101u << (k & 7))) | acc1;
    }
    asm volatile("" : "+v"(acc0), "+v"(acc1));
  }
  unsigned s = acc0 ^ acc1;
#pragma unroll
  for (int i = 0; i < 32; i++) s ^= __builtin_bit_cast(unsigned, R[i]);
  if (s == 0x12345u) sink[0] = s;
}

__device__ __forceinline__ void transform_work(const float2 *iq, float2 *spectra, int frame, int g, int s, float2 *lds, int work)
{
  const int tid = threadIdx.x;
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
  request(g);
  for (int l = g; l < 75; l += SYM_G) {
    float2 v[8];
    float acc = nx[1].x * nx[0].x + nx[1].y * nx[0].y + nx[3].x * nx[2].y;
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = nx[4 + u];
    if (l + SYM_G < 75) request(l + SYM_G);
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int w = 0; w < work; w++)
#pragma unroll
      for (int u = 0; u < 8; u++) { v[u].x = __builtin_fmaf(v[u].x, 0.999f, v[u].y + acc); v[u].y = __builtin_fmaf(v[u].y, 1.001f, -v[u].x); }
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

// mode 0: decoder blocks only (grid = n_dec), 1: transform blocks only (grid = SYM_G * S), 2: mixed: every (NA + 1)-th block is a transform block
template <int NA>
__global__ __launch_bounds__(256, 5) void k_probe(const float2 *iq, float2 *spectra, unsigned *sink, int mode, int trips, int frame, int work)
{
  extern __shared__ float2 lds[];
  int b = blockIdx.x;
  bool is_tr;
  int idx;
  if (mode == 0) { is_tr = false; idx = b; }
  else if (mode == 1) { is_tr = true; idx = b; }
  else { is_tr = (b % (NA + 1)) == NA; idx = is_tr ? b / (NA + 1) : b - b / (NA + 1); }
  if (is_tr) {
    if (idx >= SYM_G * S) return;
    transform_work(iq, spectra, frame, idx % SYM_G, idx / SYM_G, lds, work);
  } else decoder_work(trips, sink);
}

template <int NA> static double run(const float2 *iq, float2 *sp, unsigned *sink, int mode, int n_dec, int trips, int work, hipEvent_t a, hipEvent_t b)
{
  const int n_tr = SYM_G * S;
  int grid = mode == 0 ? n_dec : mode == 1 ? n_tr : n_tr * (NA + 1);
  std::vector<float> ms;
  for (int it = 0; it < 14; it++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_probe<NA>, dim3(grid), dim3(256), (2048 + 128) * sizeof(float2), 0, iq, sp, sink, mode, trips, it % 10, work);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    if (it >= 4) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main()
{
  float2 *iq, *sp; unsigned *sink;
  CK(hipMalloc(&iq, (size_t)S * RING * sizeof(float2)));
  CK(hipMalloc(&sp, (size_t)S * 75 * K * sizeof(float2)));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(iq, 0, (size_t)S * RING * sizeof(float2)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int work = 24;
  // NA decoder blocks per transform block; decoder trips chosen so that the decoder part alone takes about as long as one
  // step's k_msc_vitT share (0.36 ms) when it has the chip to itself
  printf("[\n");
  {
    constexpr int NA = 2;
    const int n_tr = SYM_G * S, n_dec = n_tr * NA;
    for (int trips : {10, 20, 40}) {
      const double ta = run<NA>(iq, sp, sink, 0, n_dec, trips, work, a, b), tb = run<NA>(iq, sp, sink, 1, n_dec, trips, work, a, b),
                   tm = run<NA>(iq, sp, sink, 2, n_dec, trips, work, a, b);
      printf(" {\"decoder_blocks_per_transform_block\": %d, \"decoder_trips\": %d, \"decoder_alone_ms\": %.4f, \"transform_alone_ms\": %.4f, \"mixed_ms\": %.4f, \"sum_ms\": %.4f, \"max_ms\": %.4f},\n",
             NA, trips, ta, tb, tm, ta + tb, ta > tb ? ta : tb);
    }
  }
  {
    constexpr int NA = 4;
    const int n_tr = SYM_G * S, n_dec = n_tr * NA;
    for (int trips : {5, 10, 20}) {
      const double ta = run<NA>(iq, sp, sink, 0, n_dec, trips, work, a, b), tb = run<NA>(iq, sp, sink, 1, n_dec, trips, work, a, b),
                   tm = run<NA>(iq, sp, sink, 2, n_dec, trips, work, a, b);
      printf(" {\"decoder_blocks_per_transform_block\": %d, \"decoder_trips\": %d, \"decoder_alone_ms\": %.4f, \"transform_alone_ms\": %.4f, \"mixed_ms\": %.4f, \"sum_ms\": %.4f, \"max_ms\": %.4f},\n",
             NA, trips, ta, tb, tm, ta + tb, ta > tb ? ta : tb);
    }
  }
  printf(" {}\n]\n");
  return 0;
}
