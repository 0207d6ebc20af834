// acq_walk_bench.hip -- how fast can ONE lane walk the two float recurrences of the null-symbol search (k_acquire, pipeline.hip)?
//     S += 0.00001f * (a - S)      three dependent operations per sample (no contraction: the reference's rounding)
//     L += d                       one
// operands in LDS, results back to LDS (the block evaluates the dip comparisons from them afterwards).  Variants of the loop's
// data layout / instruction selection; prints ns and cycles per sample for a lone wave (what an out-of-lock stream gets).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/_build/acq_walk_bench tools/acq_walk_bench.hip   (__graft_entry__.build())
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../dabstar_amd/csrc/acq_walk.h"      // (5), (6): the product's loops

constexpr int CH = 1024;
typedef float v2f __attribute__((ext_vector_type(2)));

// (1) separate arrays, float4 in / out (first version of acq_walk)
__device__ __forceinline__ void walk_sep(const float *__restrict__ a, const float *__restrict__ d, float *__restrict__ So, float *__restrict__ Lo, int m, float &S, float &L)
{
  const float4 *a4 = (const float4 *)a, *d4 = (const float4 *)d;
  float4 *S4 = (float4 *)So, *L4 = (float4 *)Lo;
  const int n4 = m >> 2;
  float4 an = a4[0], dn = d4[0];
  for (int i = 0; i < n4; i++) {
    const float4 av = an, dv = dn;
    an = a4[i + 1]; dn = d4[i + 1];
    float4 s, l;
    S += 0.00001f * (av.x - S); L += dv.x; s.x = S; l.x = L;
    S += 0.00001f * (av.y - S); L += dv.y; s.y = S; l.y = L;
    S += 0.00001f * (av.z - S); L += dv.z; s.z = S; l.z = L;
    S += 0.00001f * (av.w - S); L += dv.w; s.w = S; l.w = L;
    S4[i] = s; L4[i] = l;
  }
}
// (2) interleaved pairs: in {d, a}, state and out {L, S}: one packed add per sample, operands already paired
__device__ __forceinline__ void walk_pairs(const v2f *__restrict__ ad, v2f *__restrict__ ls_out, int m, float &S, float &L)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f *in4 = (const v4f *)ad;
  v4f *out4 = (v4f *)ls_out;
  v2f ls = {L, S};
  const int n2 = m >> 1;
  v4f nx = in4[0];
  for (int i = 0; i < n2; i++) {
    const v4f cur = nx;
    nx = in4[i + 1];
    v2f inc0 = {cur.x, 0.00001f * (cur.y - ls.y)};
    ls += inc0;
    const v2f r0 = ls;
    v2f inc1 = {cur.z, 0.00001f * (cur.w - ls.y)};
    ls += inc1;
    out4[i] = (v4f){r0.x, r0.y, ls.x, ls.y};
  }
  L = ls.x; S = ls.y;
}
// (3) like (1) but the stores of group i - 1 are issued before the loads of group i + 1 (nothing outstanding at the loop top)
__device__ __forceinline__ void walk_sep_pipelined(const float *__restrict__ a, const float *__restrict__ d, float *__restrict__ So, float *__restrict__ Lo, int m, float &S, float &L)
{
  const float4 *a4 = (const float4 *)a, *d4 = (const float4 *)d;
  float4 *S4 = (float4 *)So, *L4 = (float4 *)Lo;
  const int n4 = m >> 2;
  float4 an = a4[0], dn = d4[0];
  float4 s = {0, 0, 0, 0}, l = {0, 0, 0, 0};
  for (int i = 0; i < n4; i++) {
    if (i) { S4[i - 1] = s; L4[i - 1] = l; }
    const float4 av = an, dv = dn;
    an = a4[i + 1]; dn = d4[i + 1];
    S += 0.00001f * (av.x - S); asm volatile("" : "+v"(S)); L += dv.x; s.x = S; l.x = L;
    S += 0.00001f * (av.y - S); asm volatile("" : "+v"(S)); L += dv.y; s.y = S; l.y = L;
    S += 0.00001f * (av.z - S); asm volatile("" : "+v"(S)); L += dv.z; s.z = S; l.z = L;
    S += 0.00001f * (av.w - S); asm volatile("" : "+v"(S)); L += dv.w; s.w = S; l.w = L;
  }
  S4[n4 - 1] = s; L4[n4 - 1] = l;
}
// (4) S only (what k_level_exact does), no stores: the floor of the dependent chain
__device__ __forceinline__ void walk_s_only(const float *__restrict__ a, int m, float &S)
{
  const float4 *a4 = (const float4 *)a;
  for (int i = 0; i < (m >> 2); i++) {
    const float4 av = a4[i];
    S += 0.00001f * (av.x - S);
    S += 0.00001f * (av.y - S);
    S += 0.00001f * (av.z - S);
    S += 0.00001f * (av.w - S);
  }
}
template <int V>
__global__ __launch_bounds__(256) void k_walk(const float *src, float *dst, int reps)
{
  __shared__ __attribute__((aligned(16))) float a[CH + 32], d[CH + 32], So[CH + 32], Lo[CH + 32];
  __shared__ __attribute__((aligned(16))) v2f ad[CH + 8], ls[CH + 8];
  const int tid = threadIdx.x;
  for (int i = tid; i < CH + 8; i += 256) {     // (what lies beyond is read ahead by (5), (6) and never used)
    const float x = src[(blockIdx.x * 131 + i) % 4096], y = src[(blockIdx.x * 17 + 3 * i) % 4096];
    a[i] = x; d[i] = y - x; ad[i] = (v2f){y - x, x};
  }
  __syncthreads();
  float S = 0.1f, L = 0.f;
  // variants 7..9: the same loops with ALL 64 lanes of the wave active (every lane redundantly; same-address LDS accesses)
  if (V >= 7 ? tid < 64 : tid == 0) {
    for (int r = 0; r < reps; r++) {
      if (V == 1) walk_sep(a, d, So, Lo, CH, S, L);
      if (V == 2) walk_pairs(ad, ls, CH, S, L);
      if (V == 3) walk_sep_pipelined(a, d, So, Lo, CH, S, L);
      if (V == 4) walk_s_only(a, CH, S);
      if (V == 5 || V == 8) S = dabx::acq_walk_S(a, So, CH / 16, S);
      if (V == 7) walk_s_only(a, CH, S);
      if (V == 10) S = dabx::acq_walk_S_only(a, CH / 16, S);
      if (V == 11) S = dabx::acq_walk_S_ckpt(a, So, CH / 16, S);
      if (V == 12) L = dabx::acq_walk_L_ckpt(d, Lo, CH / 16, L);
      if (V == 9) { dabx::acq_walk_L(d, Lo, CH / 16, L); L = Lo[CH - 1]; }
      if (V == 6) { dabx::acq_walk_L(d, Lo, CH / 16, L); L = Lo[CH - 1]; }
      asm volatile("" ::: "memory");
    }
    if (tid == 0) { dst[blockIdx.x * 4 + 0] = S; dst[blockIdx.x * 4 + 1] = L; }
    if (tid == 0)
    { dst[blockIdx.x * 4 + 2] = (V == 2) ? ls[CH - 1].y : So[CH - 1];
      dst[blockIdx.x * 4 + 3] = (V == 2) ? ls[CH - 1].x : Lo[CH - 1]; }
  }
}

template <int V> static void run(const char *name, const float *src, float *dst, int blocks)
{
  const int reps = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_walk<V>, dim3(blocks), dim3(256), 0, 0, src, dst, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_walk<V>, dim3(blocks), dim3(256), 0, 0, src, dst, reps);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  float h[4];
  hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost);
  const double ns = 1e6 * ms / ((double)reps * CH);
  printf("{\"variant\": \"%s\", \"blocks\": %d, \"ns_per_sample\": %.3f, \"cycles_per_sample_at_2.4GHz\": %.1f, \"ms_per_frame_196608\": %.3f, \"S\": %.9g, \"L\": %.9g, \"S_last\": %.9g, \"L_last\": %.9g}\n",
         name, blocks, ns, ns * 2.4, ns * 196608 / 1e6, h[0], h[1], h[2], h[3]);
}

int main()
{
  std::vector<float> h(4096);
  unsigned x = 12345;
  for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) / 16777216.0f; }
  float *src, *dst;
  hipMalloc(&src, 4096 * 4); hipMalloc(&dst, 4096 * 16);
  hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int blocks : {1, 512}) {
    run<1>("separate arrays, float4", src, dst, blocks);
    run<2>("interleaved pairs, packed add", src, dst, blocks);
    run<3>("separate arrays, stores before loads, no packing", src, dst, blocks);
    run<4>("S only, no stores", src, dst, blocks);
    run<5>("acq_walk_S (asm: 16 samples per iteration, counted waits)", src, dst, blocks);
    run<6>("acq_walk_L (asm)", src, dst, blocks);
    run<7>("S only, no stores, all 64 lanes active", src, dst, blocks);
    run<8>("acq_walk_S (asm), all 64 lanes active", src, dst, blocks);
    run<9>("acq_walk_L (asm), all 64 lanes active", src, dst, blocks);
    run<10>("acq_walk_S_only (asm, no stores), all 64 lanes active", src, dst, blocks);
    run<11>("acq_walk_S_ckpt (asm, one checkpoint per 16 samples), all 64 lanes active", src, dst, blocks);
    run<12>("acq_walk_L_ckpt (asm, one checkpoint per 16 samples), all 64 lanes active", src, dst, blocks);
  }
  return 0;
}
