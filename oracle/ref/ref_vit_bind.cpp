// ref_vit_bind.cpp -- the reference's Viterbi alone, for its SIMD build variants (CMake options VITERBI_SSE2 / VITERBI_AVX2,
// /root/reference/CMakeLists.txt:144-145, src/base/CMakeLists.txt:46-60): compiled with the unmodified viterbi_spiral.cpp
// into oracle/_ref/libdabref_vit_{sse2,avx2}.so by oracle/ref/Makefile.  Test / CPU-baseline infrastructure only.
#include "viterbi_spiral.h"
#include <chrono>
#include <cstdint>
#include <cstring>

extern "C" {
int ref_viterbi(const int16_t * soft, int nbits, uint8_t * out)   // viterbi_spiral.h:20
{
  ViterbiSpiral v((short)nbits, true);
  v.deconvolve(soft, out);
  return 0;
}
// one decoder object per block length, kept (the receiver-level CPU baseline calls this for every FIC / MSC block);
// single-threaded use only: ViterbiSpiral keeps its path metrics in file-scope arrays (viterbi_spiral.cpp:41-42)
void ref_viterbi_cached(const int16_t * soft, int nbits, uint8_t * out)
{
  static ViterbiSpiral * cache[16] = {nullptr};
  static int lens[16] = {0};
  int i = 0;
  while (i < 16 && lens[i] != 0 && lens[i] != nbits) i++;
  if (i == 16) { ViterbiSpiral v((short)nbits, true); v.deconvolve(soft, out); return; }
  if (lens[i] == 0) { lens[i] = nbits; cache[i] = new ViterbiSpiral((short)nbits, true); }
  // the SIMD bodies read their input with aligned 128-bit loads (viterbi_16way.h:70): the caller's block may sit anywhere
  alignas(64) static int16_t aligned[4 * (9216 + 6) + 64];
  if (nbits <= 9216) { memcpy(aligned, soft, sizeof(int16_t) * 4 * (size_t)(nbits + 6)); soft = aligned; }
  cache[i]->deconvolve(soft, out);
}
double ref_viterbi_seconds(const int16_t * soft, int nbits, uint8_t * out, int reps)
{
  ViterbiSpiral v((short)nbits, true);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) v.deconvolve(soft, out);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (reps > 0 ? reps : 1);
}
}
