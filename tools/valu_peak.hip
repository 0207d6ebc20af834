// valu_peak.hip -- measured VALU issue rate of gfx950 for the instructions the decoder kernels are made of.
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak tools/valu_peak.hip && /tmp/valu_peak > profiles/r02_valu_peak.json
//
// For every instruction kind: 8 independent register chains per wave, 64 instructions per loop trip (inline asm, so the
// compiler can neither fold nor re-order them), launched as 256-thread blocks (one wave per SIMD) with W blocks per CU
// for W = 1, 2, 4, 8 waves per SIMD on all CUs.  Two clocks: the wall clock (HIP events) gives wave-instructions per
// second for the whole chip -- the number bench.py prices k_msc_vitT against -- and s_memtime inside the kernel gives
// shader cycles per wave-instruction per SIMD independent of DVFS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { PK_ADD_I16, PK_SUB_I16, PK_MIN_I16, PERM_B32, AND_OR_B32, ALIGNBIT, LSHRREV, ADD_U32, FMA_F32, PK_ADD_F32, PK_FMA_F32,
            BFE_U32, MIN_I32, CMP_ADDC, BFI_B32, PK_MAD_I16, SDWA_MIN_I16, MIX_ACS, MIX_ACS_DEC, N_KINDS };
static const char *kNames[N_KINDS] = {"v_pk_add_i16", "v_pk_sub_i16", "v_pk_min_i16", "v_perm_b32", "v_and_or_b32", "v_alignbit_b32",
                                      "v_lshrrev_b32", "v_add_u32", "v_fma_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_bfe_u32",
                                      "v_min_i32", "v_cmp_gt_i32 + v_addc_co_u32 (pair)", "v_bfi_b32", "v_pk_mad_i16", "v_min_i16 sdwa dst WORD_1",
                                      "mix: butterfly pair (4 pk_add/sub + 2 pk_min)", "mix: butterfly pair + decisions (+2 pk_sub, perm, lshr, and_or)"};
static const int kInstPerTrip[N_KINDS] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 128, 64, 64, 64, 48, 88};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(unsigned *sink, unsigned long long *cyc, int trips)
{
  unsigned a[8], b = threadIdx.x * 2654435761u + 12345u, c = blockIdx.x * 40503u + 977u;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8]; const f2 pb = {1.0001f, 0.9999f}, pc = {1e-7f, -1e-7f};
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = b + i * 0x01010101u; p[i] = (f2){(float)i, (float)(i + 1)}; }
  unsigned acc = 0;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < trips; t++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if constexpr (KIND == PK_ADD_I16) {
#define X(i) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (KIND == PK_SUB_I16) {
#define X(i) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (KIND == PK_MIN_I16) {
#define X(i) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (KIND == PERM_B32) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (KIND == AND_OR_B32) {
#define X(i) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (KIND == ALIGNBIT) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %0, 16" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == LSHRREV) {
#define X(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (KIND == BFE_U32) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == MIN_I32) {
#define X(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (KIND == CMP_ADDC) {
#define X(i) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if constexpr (KIND == BFI_B32) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (KIND == PK_MAD_I16) {
#define X(i) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (KIND == SDWA_MIN_I16) {
#define X(i) asm volatile("v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (KIND == FMA_F32) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(1.0001f), "v"(1e-7f));
        REP8(X)
#undef X
      } else if constexpr (KIND == PK_ADD_F32) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        REP8(X)
#undef X
      } else if constexpr (KIND == PK_FMA_F32) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        REP8(X)
#undef X
      } else if constexpr (KIND == MIX_ACS || KIND == MIX_ACS_DEC) {
        // one butterfly pair of the lane-per-trellis decoder (vit_t_gen.h): registers (a[2k], a[2k+1]) x 4 pairs per r,
        // with r < 2 only (6 or 11 instructions per pair)
        if (r < 2) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            unsigned a0, b0, a1, b1;
            asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(a0) : "v"(a[2 * k]), "v"(b));
            asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(b0) : "v"(a[2 * k + 1]), "v"(b));
            asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(a1) : "v"(a[2 * k]), "v"(b));
            asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(b1) : "v"(a[2 * k + 1]), "v"(b));
            asm volatile("v_pk_min_i16 %0, %1, %2" : "=v"(a[2 * k]) : "v"(a0), "v"(b0));
            asm volatile("v_pk_min_i16 %0, %1, %2" : "=v"(a[2 * k + 1]) : "v"(a1), "v"(b1));
            if constexpr (KIND == MIX_ACS_DEC) {
              unsigned d0, d1, pp;
              asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(d0) : "v"(b0), "v"(a0));
              asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(d1) : "v"(b1), "v"(a1));
              asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pp) : "v"(d0), "v"(d1), "v"(c));
              asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(pp));
              asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc) : "v"(pp), "v"(c));
            }
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int i = 0; i < 8; i++) acc ^= a[i] ^ __builtin_bit_cast(unsigned, p[i].x) ^ __builtin_bit_cast(unsigned, p[i].y);
  if (acc == 0x12345678u) sink[0] = acc;                 // never true in practice: keeps the chains alive
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *o = cyc + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    o[0] = t1 - t0; o[1] = r0; o[2] = r1; o[3] = hw;
  }
}

typedef void (*KFn)(unsigned *, unsigned long long *, int);
static KFn kFns[N_KINDS] = {k_issue<PK_ADD_I16>, k_issue<PK_SUB_I16>, k_issue<PK_MIN_I16>, k_issue<PERM_B32>, k_issue<AND_OR_B32>,
                            k_issue<ALIGNBIT>, k_issue<LSHRREV>, k_issue<ADD_U32>, k_issue<FMA_F32>, k_issue<PK_ADD_F32>,
                            k_issue<PK_FMA_F32>, k_issue<BFE_U32>, k_issue<MIN_I32>, k_issue<CMP_ADDC>, k_issue<BFI_B32>, k_issue<PK_MAD_I16>,
                            k_issue<SDWA_MIN_I16>, k_issue<MIX_ACS>, k_issue<MIX_ACS_DEC>};

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned *sink; unsigned long long *cyc;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * (size_t)cus * 8 * 4 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int trips = 4096;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d, \"trips\": %d, \"note\": \"wave64 instructions; "
         "cyc_per_inst_per_simd = median s_memtime delta / (instructions of one wave x waves_per_simd)\", \"rows\": [\n",
         prop.gcnArchName, cus, prop.clockRate / 1000, trips);
  bool first = true;
  for (int k = 0; k < N_KINDS; k++)
    for (int w : {1, 2, 4, 8}) {
      const int blocks = cus * w;
      for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kFns[k], dim3(blocks), dim3(256), 0, 0, sink, cyc, trips / 8);   // warm-up, clocks up
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kFns[k], dim3(blocks), dim3(256), 0, 0, sink, cyc, trips);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      std::vector<unsigned long long> raw((size_t)blocks * 16), h;
      CK(hipMemcpy(raw.data(), cyc, sizeof(unsigned long long) * raw.size(), hipMemcpyDeviceToHost));
      // in-kernel clock = shader cycles / (100 MHz real-time ticks); concurrency = sum of the waves' lifetimes / (span x waves)
      unsigned long long rmin = ~0ull, rmax = 0;
      double life = 0, clk = 0;
      for (size_t i = 0; i < (size_t)blocks * 4; i++) {
        h.push_back(raw[4 * i]);
        rmin = std::min(rmin, raw[4 * i + 1]); rmax = std::max(rmax, raw[4 * i + 2]);
        life += (double)(raw[4 * i + 2] - raw[4 * i + 1]);
        clk += (double)raw[4 * i] / (double)(raw[4 * i + 2] - raw[4 * i + 1]) * 0.1;      // GHz
      }
      clk /= (double)blocks * 4;
      const double overlap = life / ((double)(rmax - rmin) * blocks * 4);
      std::sort(h.begin(), h.end());
      const double per_wave = (double)trips * kInstPerTrip[k];
      const double wave_insts = per_wave * blocks * 4;
      const double med = (double)h[h.size() / 2];
      printf("%s  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_s\": %.4e, \"cyc_per_inst_per_simd\": %.3f, "
             "\"in_kernel_clock_ghz\": %.3f, \"wave_overlap\": %.3f}",
             first ? "" : ",\n", kNames[k], w, best, wave_insts / (best * 1e-3), med / per_wave / w * 1.0,
             clk, overlap);
      first = false;
    }
  printf("\n]}\n");
  return 0;
}
