// ref_vit_bind.cpp -- the reference's Viterbi alone, for its SIMD build variants (CMake options VITERBI_SSE2 / VITERBI_AVX2,
// /root/reference/CMakeLists.txt:144-145, src/base/CMakeLists.txt:46-60): compiled with the unmodified viterbi_spiral.cpp
// into oracle/_ref/libdabref_vit_{sse2,avx2}.so by oracle/ref/Makefile.  Test / CPU-baseline infrastructure only.
#include "viterbi_spiral.h"
#include <chrono>
#include <cstdint>

extern "C" {
int ref_viterbi(const int16_t * soft, int nbits, uint8_t * out)   // viterbi_spiral.h:20
{
  ViterbiSpiral v((short)nbits, true);
  v.deconvolve(soft, out);
  return 0;
}
double ref_viterbi_seconds(const int16_t * soft, int nbits, uint8_t * out, int reps)
{
  ViterbiSpiral v((short)nbits, true);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) v.deconvolve(soft, out);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (reps > 0 ? reps : 1);
}
}
