/* viterbi.c -- K=7 rate-1/4 Viterbi decoder (oracle; test infrastructure only).
 * Restates support/viterbi_spiral/viterbi_spiral.cpp:95-126 with the canonical scalar
 * body viterbi_scalar.h:9-94 (i32 path metrics, no renormalisation, ACS ties -> 0). */
#include "dab_oracle.h"
#include <stdlib.h>
#include <string.h>

static const int kPolys[4] = {109, 79, 83, 109};   /* viterbi_spiral.cpp:132 */

static int parity8(int x)
{
  x ^= x >> 4; x ^= x >> 2; x ^= x >> 1;
  return x & 1;
}

/* Branch table: viterbi_spiral.cpp:27-37 == 255 * parity((2*i) & poly) (Karn's
 * partab construction); rebuilt here instead of being listed. */
static int g_branch[4][32];
static int g_branch_ready = 0;
static void build_branch(void)
{
  for (int p = 0; p < 4; p++)
    for (int i = 0; i < 32; i++) g_branch[p][i] = parity8((2 * i) & kPolys[p]) ? 255 : 0;
  g_branch_ready = 1;
}

void ora_viterbi(const int16_t *soft, int nbits, uint8_t *out_bits)
{
  if (!g_branch_ready) build_branch();
  const int nsteps = nbits + 6;
  uint64_t *dec = (uint64_t *)calloc((size_t)nsteps, sizeof(uint64_t));
  int32_t m1[64], m2[64];
  int32_t *old = m1, *nw = m2;
  for (int i = 0; i < 64; i++) old[i] = 1000;      /* viterbi_spiral.cpp:98-101 */
  old[0] = 0;

  for (int t = 0; t < nsteps; t++) {
    int sym[4];
    for (int p = 0; p < 4; p++) {                   /* viterbi_scalar.h:34-40 */
      int v = (int16_t)(soft[4 * t + p] + 127);
      if (v < 0) v = 0; else if (v > 255) v = 255;
      sym[p] = v;
    }
    uint64_t d = 0;
    for (int i = 0; i < 32; i++) {                  /* viterbi_scalar.h:9-32 */
      const int metric = (g_branch[0][i] ^ sym[0]) + (g_branch[1][i] ^ sym[1]) +
                         (g_branch[2][i] ^ sym[2]) + (g_branch[3][i] ^ sym[3]);
      const int m_metric = 1020 - metric;
      const int32_t a0 = old[i] + metric, a1 = old[i + 32] + m_metric;
      const int32_t a2 = old[i] + m_metric, a3 = old[i + 32] + metric;
      const int d0 = (a0 - a1) > 0, d1 = (a2 - a3) > 0;
      nw[2 * i] = d0 ? a1 : a0;
      nw[2 * i + 1] = d1 ? a3 : a2;
      d |= (uint64_t)(d0 | (d1 << 1)) << (2 * i);
    }
    dec[t] = d;
    int32_t *tmp = old; old = nw; nw = tmp;
  }

  /* chain back: viterbi_spiral.cpp:114-125 (endstate kept as 8 bits there, >>2 = state) */
  unsigned endstate = 0;
  for (int fb = nbits - 1; fb >= 0; fb--) {
    const int k = (int)((dec[fb + 6] >> (endstate >> 2)) & 1);
    endstate = (endstate >> 1) | ((unsigned)k << 7);
    out_bits[fb] = (uint8_t)k;
  }
  free(dec);
}

/* The body the reference's VITERBI_AVX2 / VITERBI_SSE2 builds use instead (viterbi_16way.h:9-110, viterbi_8way.h):
 * uint16 path metrics with saturating adds, decision = (survivor == path through state i + 32), i.e. ties -> 1, and
 * after every second step `renormalize`: if metrics2[0] -- state 0's metric of the step BEFORE the one just computed --
 * exceeds 60000, the minimum of the new metrics is subtracted from all of them.  Pinned against the reference's AVX2
 * object code (tests/test_oracle_ref.py). */
void ora_viterbi_simd(const int16_t *soft, int nbits, uint8_t *out_bits)
{
  if (!g_branch_ready) build_branch();
  const int nsteps = nbits + 6;
  uint64_t *dec = (uint64_t *)calloc((size_t)nsteps, sizeof(uint64_t));
  uint32_t m1[64], m2[64];                          /* metrics1 / metrics2, viterbi_spiral.cpp:41-42 */
  uint32_t *old = m1, *nw = m2;
  for (int i = 0; i < 64; i++) m1[i] = 1000, m2[i] = 0;
  m1[0] = 0;
  for (int t = 0; t < nsteps; t++) {
    int sym[4];
    for (int p = 0; p < 4; p++) {                   /* viterbi_16way.h:73-76: adds_epi16(+127), min 255, max 0 */
      int v = (int)soft[4 * t + p] + 127;
      if (v > 32767) v = 32767;                     /* saturating 16-bit add */
      if (v > 255) v = 255;
      if (v < 0) v = 0;
      sym[p] = v;
    }
    uint64_t d = 0;
    for (int i = 0; i < 32; i++) {                  /* BFLY, viterbi_16way.h:27-58 */
      const int metric = (g_branch[0][i] ^ sym[0]) + (g_branch[1][i] ^ sym[1]) +
                         (g_branch[2][i] ^ sym[2]) + (g_branch[3][i] ^ sym[3]);
      const int m_metric = 1020 - metric;
      uint32_t a0 = old[i] + metric, a1 = old[i + 32] + m_metric, a2 = old[i] + m_metric, a3 = old[i + 32] + metric;
      if (a0 > 65535) a0 = 65535;                   /* adds_epu16 */
      if (a1 > 65535) a1 = 65535;
      if (a2 > 65535) a2 = 65535;
      if (a3 > 65535) a3 = 65535;
      const uint32_t s0 = a0 < a1 ? a0 : a1, s1 = a2 < a3 ? a2 : a3;
      const int d0 = s0 == a1, d1 = s1 == a3;       /* cmpeq(survivor, m1 / m3): a tie selects the i + 32 path */
      nw[2 * i] = s0; nw[2 * i + 1] = s1;
      d |= (uint64_t)(d0 | (d1 << 1)) << (2 * i);
    }
    dec[t] = d;
    if (t & 1) {                                    /* renormalize sits after the second butterfly pair of the loop body */
      /* at this point new_metrics == metrics1 and metrics2 holds the metrics of step t - 1 (pointer swaps, :93-96,:110-113) */
      const uint32_t *prev = old;                   /* `old` still points at the metrics of step t - 1 == metrics2 */
      if (prev[0] > 60000) {
        uint32_t mn = nw[0];
        for (int i = 1; i < 64; i++) if (nw[i] < mn) mn = nw[i];
        for (int i = 0; i < 64; i++) nw[i] -= mn;   /* subs_epu16: never below zero */
      }
    }
    uint32_t *tmp = old; old = nw; nw = tmp;
  }
  unsigned endstate = 0;                            /* chain back: viterbi_spiral.cpp:114-125 */
  for (int fb = nbits - 1; fb >= 0; fb--) {
    const int k = (int)((dec[fb + 6] >> (endstate >> 2)) & 1);
    endstate = (endstate >> 1) | ((unsigned)k << 7);
    out_bits[fb] = (uint8_t)k;
  }
  free(dec);
}

/* The body of the VITERBI_SSE2 (and NEON) build (viterbi_8way.h:9-120): SIGNED int16 path metrics with saturating adds
 * (_mm_adds_epi16: ceiling 32767), decision = m0 > m1 (:39,:41: a tie keeps predecessor i, like the scalar body), and after
 * every second step `renormalize` with threshold 30000 on metrics2[0] -- state 0's metric of the step BEFORE -- subtracting the
 * minimum of the new metrics (_mm_subs_epi16).  Differs from the scalar body only where a metric saturates.  Pinned against
 * the reference's SSE2 object code (tests/test_oracle_ref.py). */
void ora_viterbi_sse2(const int16_t *soft, int nbits, uint8_t *out_bits)
{
  if (!g_branch_ready) build_branch();
  const int nsteps = nbits + 6;
  uint64_t *dec = (uint64_t *)calloc((size_t)nsteps, sizeof(uint64_t));
  int32_t m1[64], m2[64];
  int32_t *old = m1, *nw = m2;
  for (int i = 0; i < 64; i++) m1[i] = 1000, m2[i] = 0;
  m1[0] = 0;
  for (int t = 0; t < nsteps; t++) {
    int sym[4];
    for (int p = 0; p < 4; p++) {                   /* viterbi_8way.h:72-75 */
      int v = (int)soft[4 * t + p] + 127;
      if (v > 32767) v = 32767;
      if (v > 255) v = 255;
      if (v < 0) v = 0;
      sym[p] = v;
    }
    uint64_t d = 0;
    for (int i = 0; i < 32; i++) {                  /* BFLY, viterbi_8way.h:27-53 */
      const int metric = (g_branch[0][i] ^ sym[0]) + (g_branch[1][i] ^ sym[1]) +
                         (g_branch[2][i] ^ sym[2]) + (g_branch[3][i] ^ sym[3]);
      const int m_metric = 1020 - metric;
      int32_t a0 = old[i] + metric, a1 = old[i + 32] + m_metric, a2 = old[i] + m_metric, a3 = old[i + 32] + metric;
      if (a0 > 32767) a0 = 32767;                   /* adds_epi16 (the operands are never negative) */
      if (a1 > 32767) a1 = 32767;
      if (a2 > 32767) a2 = 32767;
      if (a3 > 32767) a3 = 32767;
      const int d0 = a0 > a1, d1 = a2 > a3;
      nw[2 * i] = d0 ? a1 : a0; nw[2 * i + 1] = d1 ? a3 : a2;
      d |= (uint64_t)(d0 | (d1 << 1)) << (2 * i);
    }
    dec[t] = d;
    if (t & 1) {
      if (old[0] > 30000) {                         /* `old` = the metrics of step t - 1 = metrics2 */
        int32_t mn = nw[0];
        for (int i = 1; i < 64; i++) if (nw[i] < mn) mn = nw[i];
        for (int i = 0; i < 64; i++) nw[i] -= mn;
      }
    }
    int32_t *tmp = old; old = nw; nw = tmp;
  }
  unsigned endstate = 0;                            /* chain back: viterbi_spiral.cpp:114-125 */
  for (int fb = nbits - 1; fb >= 0; fb--) {
    const int k = (int)((dec[fb + 6] >> (endstate >> 2)) & 1);
    endstate = (endstate >> 1) | ((unsigned)k << 7);
    out_bits[fb] = (uint8_t)k;
  }
  free(dec);
}

/* Which body ViterbiSpiral::deconvolve was compiled with (viterbi_spiral.cpp:105-112 picks one by HAVE_VITERBI_*): the
 * receiver-level oracle (fic.c, protection.c) decodes through this switch.  0 = scalar (CMake default), 1 = AVX2, 2 = SSE2. */
static int g_viterbi_mode = 0;
static void (*g_viterbi_hook)(const int16_t *, int, uint8_t *) = 0;
void ora_set_viterbi_mode(int mode) { g_viterbi_mode = mode; }
/* CPU-baseline leg only (bench.py "reference-best"): decode through the reference's OWN AVX2 object code
 * (oracle/_ref/libdabref_vit_avx2.so, ref_viterbi_cached) instead of the plain-C restatement above. */
void ora_set_viterbi_hook(void (*fn)(const int16_t *, int, uint8_t *)) { g_viterbi_hook = fn; }
/* time spent inside the decoder (all bodies), for the CPU-baseline breakdown of bench.py */
#include <time.h>
static double g_viterbi_seconds = 0.0;
static long long g_viterbi_calls = 0;
double ora_viterbi_seconds(long long *calls) { if (calls) *calls = g_viterbi_calls; return g_viterbi_seconds; }
void ora_viterbi_seconds_reset(void) { g_viterbi_seconds = 0.0; g_viterbi_calls = 0; }
static void viterbi_build_inner(const int16_t *soft, int nbits, uint8_t *out_bits);
void ora_viterbi_build(const int16_t *soft, int nbits, uint8_t *out_bits)
{
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  viterbi_build_inner(soft, nbits, out_bits);
  clock_gettime(CLOCK_MONOTONIC, &b);
  g_viterbi_seconds += (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
  g_viterbi_calls++;
}
static void viterbi_build_inner(const int16_t *soft, int nbits, uint8_t *out_bits)
{
  if (g_viterbi_hook) {
#ifdef __AVX__
    __builtin_ia32_vzeroupper();     /* native build: leave no dirty upper vector state for the callee's 128 / 256-bit code */
#endif
    g_viterbi_hook(soft, nbits, out_bits);
#ifdef __AVX__
    __builtin_ia32_vzeroupper();     /* ... and none for the non-VEX library code that runs after it (the callee returns with its own) */
#endif
  }
  else if (g_viterbi_mode == 1) ora_viterbi_simd(soft, nbits, out_bits);
  else if (g_viterbi_mode == 2) ora_viterbi_sse2(soft, nbits, out_bits);
  else ora_viterbi(soft, nbits, out_bits);
}

/* viterbi_spiral.cpp:128-164 : re-encode and compare against hard decisions */
void ora_viterbi_ber(const int16_t *soft, const uint8_t *punct, const uint8_t *bits,
                     int nbits, int *io_bits, int *io_errors)
{
  int sr = 0;
  for (int i = 0; i < nbits + 6; i++) {
    sr = ((sr << 1) | (i < nbits ? bits[i] : 0)) & 0xff;
    for (int j = 0; j < 4; j++) {
      const int b = parity8(sr & kPolys[j]);
      if (punct[i * 4 + j]) {
        (*io_bits)++;
        if ((soft[i * 4 + j] > 0) != b) (*io_errors)++;
      }
    }
  }
}
