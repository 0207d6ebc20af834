// dpp_probe.hip -- what the DPP controls used by csrc/level_par.h do on this GPU: every lane holds 100 + its number.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int RM, bool BC> __device__ int dpp(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xF, BC); }
__global__ void k(int *out)
{
  const int l = threadIdx.x, v = 100 + l;
  out[0 * 64 + l] = dpp<0x111, 0xF, true>(v, 0);    // row_shr:1
  out[1 * 64 + l] = dpp<0x118, 0xF, true>(v, 0);    // row_shr:8
  out[2 * 64 + l] = dpp<0x142, 0xA, true>(v, 0);    // row_bcast:15
  out[3 * 64 + l] = dpp<0x143, 0xC, true>(v, 0);    // row_bcast:31
  out[4 * 64 + l] = dpp<0x138, 0xF, true>(v, 0);    // wave_shr:1
  out[5 * 64 + l] = dpp<0x130, 0xF, true>(v, 0);    // wave_shl:1
  out[6 * 64 + l] = dpp<0x142, 0xA, false>(v, -1);  // row_bcast:15, old = -1
}
int main()
{
  int *d, h[7 * 64];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *n[] = {"row_shr:1", "row_shr:8", "row_bcast:15 rows 1,3", "row_bcast:31 rows 2,3", "wave_shr:1", "wave_shl:1", "row_bcast:15 old=-1"};
  for (int r = 0; r < 7; r++) { printf("%-24s", n[r]); for (int l = 0; l < 64; l++) printf(" %d", h[r * 64 + l]); printf("\n"); }
  return 0;
}
