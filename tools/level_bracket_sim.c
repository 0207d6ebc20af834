/* CPU model of the parallel form of SampleReader's level recurrence (docs/history/r01-r04_design_notebook.md 3, "the level tracker in parallel"):
 *
 *     S <- S + 0.00001f * (a - S)              (three float operations per sample, sample_reader.cpp:245-248)
 *
 * is serial, but two trajectories that start one float apart stay one float apart until a rounding merges them (probability
 * ~1e-5 per sample), and never drift apart while they stay inside one binade.  So a block of 1024 samples is cut into 64
 * groups of 16; group g starts from a GUESS G_g (the same recurrence in real arithmetic from the block's exact start value)
 * and is walked twice, from G_g - K and G_g + K floats.  If both ends are still 2 K floats apart (and everything stayed in
 * one binade, and no sample was absurdly larger than the level) the group maps every start value in between by the same
 * shift: end = end_lo + (start - start_lo).  The true start values then follow from an integer prefix sum over the groups;
 * a group that fails a check is walked serially from its (by then known) true start.
 *
 * This program checks the scheme against the serial recurrence bit for bit on synthetic signals and counts how often the
 * fall-back is needed.        gcc -O2 -ffp-contract=off -fno-fast-math -o level_bracket_sim level_bracket_sim.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GROUP 16
#define LANES 64
#define BLOCK (GROUP * LANES)
static int K = 48;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float step(float S, float a) { volatile float t = a - S; volatile float d = 0.00001f * t; return S + d; }

static long n_brute_bad, n_blocks, n_groups, n_fallback, n_unsafe_merge, n_unsafe_binade, n_unsafe_big, n_range, n_tie, k_hist[4];

/* one block: a[ng * 16] -> exact group start values ck[0..ng] (ck[ng] = value after the block) */
static void block_bracket(const float *a, int ng, float S0, float *ck)
{
  const double cD = (double)0.00001f;
  double R = (double)S0;
  uint32_t G[LANES + 1]; int32_t E[LANES], D[LANES], k[LANES + 1]; int safe[LANES];
  for (int g = 0; g < ng; g++) {
    G[g] = f2u((float)R);
    for (int i = 0; i < GROUP; i++) R += cD * ((double)a[g * GROUP + i] - R);
  }
  G[0] = f2u(S0);
  for (int g = 0; g < ng; g++) {
    float lo = u2f(G[g] - (uint32_t)K), hi = u2f(G[g] + (uint32_t)K), mn = lo, mx = hi, amax = 0.f;
    const float lo0 = lo;
    const float half_ulp = u2f(((f2u(lo0) >> 23) - 24u) << 23);       /* of the binade the group starts in (normal numbers) */
    int shift = 1, tie = 0;
    for (int i = 0; i < GROUP; i++) {
      const float x = a[g * GROUP + i];
      volatile float tl = x - lo, th = x - hi;
      volatile float dl = 0.00001f * tl, dh = 0.00001f * th;
      volatile float rl = lo + dl, rh = hi + dh;
      volatile float zl = rl - lo, zh = rh - hi;
      volatile float el = dl - zl, eh = dh - zh;                     /* what the additions rounded away (Fast2Sum: exact while |d| <= |S|) */
      tie |= fabsf(el) == half_ulp || fabsf(eh) == half_ulp;
      lo = rl; hi = rh;
      shift &= f2u(hi) - f2u(lo) == 2u * (uint32_t)K;
      mn = fminf(mn, lo); mx = fmaxf(mx, hi); amax = fmaxf(amax, x);
    }
    const int normal = G[g] > (25u << 23) + (uint32_t)K && G[g] < 0x7f000000u;
    const int one_binade = (f2u(mn) >> 23) == (f2u(mx) >> 23);
    const int small = amax <= 1024.f * lo0;
    safe[g] = normal && one_binade && shift && small && !tie;
    if (normal && !shift) n_unsafe_merge++;
    if (normal && shift && !one_binade) n_unsafe_binade++;
    if (normal && shift && one_binade && !small) n_unsafe_big++;
    if (normal && shift && one_binade && small && tie) n_tie++;
    E[g] = (int32_t)(f2u(lo) + (uint32_t)K);
#ifdef BRUTE
    if (safe[g]) for (int kk = -K; kk <= K; kk++) {
      float S = u2f(G[g] + (uint32_t)kk);
      for (int i = 0; i < GROUP; i++) S = step(S, a[g * GROUP + i]);
      if (f2u(S) != (uint32_t)(E[g] + kk)) { n_brute_bad++; break; }
    }
#endif
  }
  k[0] = 0;
  for (int g = 0; g < ng; g++) {
    const int ok = safe[g] && abs(k[g]) <= K;
    uint32_t end;
    if (ok) end = (uint32_t)(E[g] + k[g]);
    else {
      float S = u2f(G[g] + (uint32_t)k[g]);
      for (int i = 0; i < GROUP; i++) S = step(S, a[g * GROUP + i]);
      end = f2u(S);
      n_fallback++;
      if (safe[g]) n_range++;
    }
    ck[g] = u2f(G[g] + (uint32_t)k[g]);
    if (g + 1 < ng) k[g + 1] = (int32_t)(end - G[g + 1]);
    else ck[ng] = u2f(end);
    { const int ak = abs(k[g]); k_hist[ak < 8 ? 0 : ak < 16 ? 1 : ak < 32 ? 2 : 3]++; }
    (void)D;
  }
  n_blocks++; n_groups += ng;
}

static uint64_t rs = 88172645463325252ull;
static double urand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }
static double nrand(void) { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

int main(int argc, char **argv)
{
  if (argc > 1) K = atoi(argv[1]);
  const long N = argc > 2 ? atol(argv[2]) : 40L * 1000 * 1000;
  float *a = (float *)malloc(sizeof(float) * (size_t)N);
  const char *names[] = {"ofdm-like |gauss| at 0.3", "level 1e-3 with nulls", "silence then signal", "spikes x1e4", "level steps x2 (binade crossings)",
                         "exact zeros for 9 M samples", "constant = 2^-3 (level converges onto a binade edge)"};
  long bad_total = 0;
  for (int c = 0; c < 7; c++) {
    for (long i = 0; i < N; i++) {
      const double g = hypot(nrand(), nrand());
      double v;
      switch (c) {
      case 0: v = 0.3 * g; break;
      case 1: v = ((i % 196608) < 2656 ? 1e-6 : 1e-3) * g; break;
      case 2: v = (i % 4000000) < 2000000 ? 1e-7 * g : 0.5 * g; break;
      case 3: v = (urand() < 1e-4 ? 3e3 : 0.3) * g; break;
      case 4: v = 0.25 * (1.0 + 0.9 * sin(i * 1e-5)) * g; break;
      case 5: v = i < 9000000 ? 0.0 : 0.1 * g; break;
      default: v = 0.125; break;
      }
      a[i] = (float)v;
    }
    n_blocks = n_groups = n_fallback = n_unsafe_merge = n_unsafe_binade = n_unsafe_big = n_range = n_tie = 0;
    memset(k_hist, 0, sizeof(k_hist));
    float S = 0.1f, Sref = 0.1f, ck[LANES + 1];
    long bad = 0;
    for (long p = 0; p + BLOCK <= N; p += BLOCK) {
      block_bracket(a + p, LANES, S, ck);
      for (int g = 0; g < LANES; g++) {
        if (f2u(ck[g]) != f2u(Sref)) bad++;
        for (int i = 0; i < GROUP; i++) Sref = step(Sref, a[p + g * GROUP + i]);
      }
      if (f2u(ck[LANES]) != f2u(Sref)) bad++;
      S = ck[LANES];
    }
    printf("{\"case\": \"%s\", \"K\": %d, \"samples\": %ld, \"group_starts_differing\": %ld, \"fallbacks_per_block\": %.3f, \"merge\": %ld, \"binade\": %ld, \"big\": %ld, \"tie\": %ld, "
           "\"out_of_range\": %ld, \"abs_k_lt8_lt16_lt32_ge32\": [%ld, %ld, %ld, %ld], \"level\": %.6g, \"brute_force_violations\": %ld}\n",
           names[c], K, N, bad, (double)n_fallback / (double)n_blocks, n_unsafe_merge, n_unsafe_binade, n_unsafe_big, n_tie, n_range,
           k_hist[0], k_hist[1], k_hist[2], k_hist[3], (double)S, n_brute_bad);
    bad_total += bad;
  }
  free(a);
  return bad_total != 0;
}
