// valu_hazards.hip -- what makes a half-rate VALU stream on gfx950 slower than its 4-cycle issue rate?
// Companion of valu_peak.hip.  Variants of a v_pk_add_i16 / v_pk_min_i16 stream at W waves per SIMD, with explicit
// physical registers (inline asm, clobbers declared) so that operand banks (vgpr index mod 4) and dependency distance
// are controlled:
//   bank_same   : src0 and src1 in the same bank            bank_diff : different banks
//   dep1 / dep2 / dep4 / dep8 : number of independent accumulator chains (dependent distance in instructions)
//   salu_mix    : one s_add_u32 + one s_lshl_b32 after every 6 VALU (the decoder loop carries ~33 SALU per 200 VALU)
//   vmem_mix    : one 256-B coalesced global store per 40 VALU
// Output: JSON rows with wave-instructions/s over the chip and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { BANK_SAME, BANK_DIFF, BANK_DST_SAME, DEP1, DEP2, DEP4, DEP8, SALU_MIX, VMEM_MIX, N_KINDS };
static const char *kNames[N_KINDS] = {"pk_add: src0, src1 same bank", "pk_add: src0, src1 different banks", "pk_add: dst, src0, src1 all same bank",
                                      "pk_add: 1 dependent chain", "pk_add: 2 chains", "pk_add: 4 chains", "pk_add: 8 chains",
                                      "pk_add 8 chains + 2 SALU per 6 VALU", "pk_add 8 chains + one 256-B store per 40 VALU"};
static const int kValuPerTrip[N_KINDS] = {64, 64, 64, 64, 64, 64, 64, 48, 40};

template <int KIND>
__global__ __launch_bounds__(256) void k_haz(unsigned *sink, unsigned long long *stamp, unsigned *scratch, int trips)
{
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned out = threadIdx.x;
  if constexpr (KIND == BANK_SAME || KIND == BANK_DIFF || KIND == BANK_DST_SAME) {
    // 8 chains on v[16..47]; bank = index mod 4
    for (int t = 0; t < trips; t++) {
      if constexpr (KIND == BANK_SAME)        // dst/src0 bank 0 (v16, v20, ...), src1 v4 (bank 0)
        asm volatile(
#define L(a) "v_pk_add_i16 v" #a ", v" #a ", v4\n\t"
#define B8 L(16) L(20) L(24) L(28) L(32) L(36) L(40) L(44)
            B8 B8 B8 B8 B8 B8 B8 B8
#undef L
            ::: "v4", "v16", "v20", "v24", "v28", "v32", "v36", "v40", "v44");
      else if constexpr (KIND == BANK_DIFF)   // dst/src0 bank 0, src1 v5 (bank 1)
        asm volatile(
#define L(a) "v_pk_add_i16 v" #a ", v" #a ", v5\n\t"
            B8 B8 B8 B8 B8 B8 B8 B8
#undef L
            ::: "v5", "v16", "v20", "v24", "v28", "v32", "v36", "v40", "v44");
      else                                     // dst bank 0, src0 = another bank-0 register, src1 bank 0: three same-bank accesses
        asm volatile(
#define L(a, b) "v_pk_add_i16 v" #a ", v" #b ", v4\n\t"
#define C8 L(16, 20) L(20, 24) L(24, 28) L(28, 32) L(32, 36) L(36, 40) L(40, 44) L(44, 16)
            C8 C8 C8 C8 C8 C8 C8 C8
#undef L
            ::: "v4", "v16", "v20", "v24", "v28", "v32", "v36", "v40", "v44");
    }
    asm volatile("v_mov_b32 %0, v16" : "=v"(out)::"v16");
  } else if constexpr (KIND == DEP1 || KIND == DEP2 || KIND == DEP4 || KIND == DEP8) {
    constexpr int NCH = KIND == DEP1 ? 1 : KIND == DEP2 ? 2 : KIND == DEP4 ? 4 : 8;
    unsigned a[8], b = threadIdx.x * 77u + 1u;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = b + i;
    for (int t = 0; t < trips; t++) {
#pragma unroll
      for (int r = 0; r < 64; r++) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[r % NCH]) : "v"(b));
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out ^= a[i];
  } else if constexpr (KIND == SALU_MIX) {
    unsigned a[8], b = threadIdx.x * 77u + 1u;
    unsigned s0 = blockIdx.x, s1 = 3;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = b + i;
    for (int t = 0; t < trips; t++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
#pragma unroll
        for (int q = 0; q < 6; q++) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[(r * 6 + q) % 8]) : "v"(b));
        asm volatile("s_add_u32 %0, %0, %1\n\ts_lshl_b32 %1, %1, 1" : "+s"(s0), "+s"(s1)::"scc");
      }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out ^= a[i];
    out ^= s0 ^ s1;
  } else {
    unsigned a[8], b = threadIdx.x * 77u + 1u;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = b + i;
    unsigned *p = scratch + ((size_t)blockIdx.x * 256 + threadIdx.x);
    for (int t = 0; t < trips; t++) {
#pragma unroll
      for (int r = 0; r < 40; r++) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[r % 8]) : "v"(b));
      p[(size_t)(t & 63) * 65536 * 4] = a[0];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out ^= a[i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (out == 0x12345678u) sink[0] = out;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *o = stamp + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    o[0] = t1 - t0; o[1] = r0; o[2] = r1; o[3] = 0;
  }
}

typedef void (*KFn)(unsigned *, unsigned long long *, unsigned *, int);
static KFn kFns[N_KINDS] = {k_haz<BANK_SAME>, k_haz<BANK_DIFF>, k_haz<BANK_DST_SAME>, k_haz<DEP1>, k_haz<DEP2>, k_haz<DEP4>, k_haz<DEP8>,
                            k_haz<SALU_MIX>, k_haz<VMEM_MIX>};

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned *sink, *scratch; unsigned long long *stamp;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&scratch, (size_t)64 * 65536 * 16 + (size_t)cus * 8 * 1024));
  CK(hipMalloc(&stamp, sizeof(unsigned long long) * (size_t)cus * 8 * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int trips = 4096;
  printf("{\"device\": \"%s\", \"cus\": %d, \"trips\": %d, \"rows\": [\n", prop.gcnArchName, cus, trips);
  bool first = true;
  for (int k = 0; k < N_KINDS; k++)
    for (int w : {1, 2, 4}) {
      const int blocks = cus * w;
      hipLaunchKernelGGL(kFns[k], dim3(blocks), dim3(256), 0, 0, sink, stamp, scratch, trips / 8);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kFns[k], dim3(blocks), dim3(256), 0, 0, sink, stamp, scratch, trips);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      std::vector<unsigned long long> raw((size_t)blocks * 16);
      CK(hipMemcpy(raw.data(), stamp, sizeof(unsigned long long) * raw.size(), hipMemcpyDeviceToHost));
      double clk = 0, life = 0; unsigned long long rmin = ~0ull, rmax = 0;
      for (size_t i = 0; i < (size_t)blocks * 4; i++) {
        clk += (double)raw[4 * i] / (double)(raw[4 * i + 2] - raw[4 * i + 1]) * 0.1;
        life += (double)(raw[4 * i + 2] - raw[4 * i + 1]);
        rmin = std::min(rmin, raw[4 * i + 1]); rmax = std::max(rmax, raw[4 * i + 2]);
      }
      clk /= (double)blocks * 4;
      const double wave_insts = (double)trips * kValuPerTrip[k] * blocks * 4;
      const double rate = wave_insts / (best * 1e-3);
      printf("%s  {\"variant\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"valu_wave_insts_per_s\": %.4e, \"in_kernel_clock_ghz\": %.3f, "
             "\"cyc_per_valu_per_simd\": %.3f, \"wave_overlap\": %.3f}",
             first ? "" : ",\n", kNames[k], w, best, rate, clk, clk * 1e9 * cus * 4 / rate, life / ((double)(rmax - rmin) * blocks * 4));
      first = false;
    }
  printf("\n]}\n");
  return 0;
}
