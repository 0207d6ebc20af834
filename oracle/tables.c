/* tables.c -- constant tables of the DAB Mode-I path (oracle; test infrastructure only). */
#include "dab_oracle.h"
#include <math.h>
#include <string.h>

/* ETSI EN 300 401 table 13 (puncturing vectors), as held by
 * protection/protTables.cpp:36-68.  The table has a regular structure: PI_k keeps
 * (k-1)/8 + 1 bits in every group of four, and the first ((k-1)%8)+1 groups taken in
 * the order 0,4,2,6,1,5,3,7 keep one more.  Generated here, checked against the
 * reference's get_PI_codes() in tests/test_oracle_ref.py. */
static int8_t g_pi[24][32];
static int g_pi_ready = 0;

static void build_pi(void)
{
  static const int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
  for (int k = 1; k <= 24; k++) {
    const int base = (k - 1) / 8 + 1;       /* ones per group of four */
    const int extra = (k - 1) % 8 + 1;      /* groups that get one more */
    int ones[8];
    for (int g = 0; g < 8; g++) ones[g] = base;
    for (int e = 0; e < extra; e++) ones[order[e]] = base + 1;
    for (int g = 0; g < 8; g++)
      for (int j = 0; j < 4; j++) g_pi[k - 1][4 * g + j] = (j < ones[g]) ? 1 : 0;
  }
  g_pi_ready = 1;
}

const int8_t *ora_pi_codes(int pi)
{
  if (!g_pi_ready) build_pi();
  if (pi < 1 || pi > 24) return NULL;
  return g_pi[pi - 1];
}

/* ofdm/freq_interleaver.cpp:40-76 (Mode I: V1 = 511, range 256..1792, skip 1024) */
void ora_freq_interleaver(int16_t perm[ORA_K])
{
  int16_t tmp[ORA_TU];
  int idx = 0;
  tmp[0] = 0;
  for (int i = 1; i < ORA_TU; i++) tmp[i] = (int16_t)((13 * tmp[i - 1] + 511) % ORA_TU);
  for (int i = 0; i < ORA_TU; i++) {
    if (tmp[i] == ORA_TU / 2) continue;
    if (tmp[i] < 256 || tmp[i] > 256 + ORA_K) continue;
    perm[idx++] = (int16_t)(tmp[i] - ORA_TU / 2);
  }
}

/* ofdm/phasetable.cpp:35-85: (k', i, n) per block of 32 carriers; k' = -768+32*b for the
 * lower half and 1+32*b for the upper half.  i runs 0,1,2,3 (lower) / 0,3,2,1 (upper). */
static const uint8_t prs_n_lower[24] = {1,2,0,1,3,2,2,3,2,1,2,3,1,2,3,3,2,2,2,1,1,3,1,2};
static const uint8_t prs_n_upper[24] = {3,1,1,1,2,2,1,0,2,2,3,3,0,2,1,3,3,3,3,0,3,0,1,1};
/* ofdm/phasetable.cpp:103-106 (h tables, period 16) */
static const uint8_t prs_h[4][16] = {
  {0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1},
  {0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0},
  {0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3},
  {0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2}};

/* phasetable.cpp:122-135 get_phi */
static float prs_phi(int k)
{
  int b, kp, i, n;
  if (k < 0) {
    b = (k + 768) / 32; kp = -768 + 32 * b; i = b & 3; n = prs_n_lower[b];
  } else {
    b = (k - 1) / 32; kp = 1 + 32 * b; i = (4 - (b & 3)) & 3; n = prs_n_upper[b];
  }
  const float half_pi = (float)(M_PI / 2.0);
  return half_pi * (float)(prs_h[i][(k - kp) & 15] + n);
}

/* phasetable.cpp:87-101 */
void ora_phase_table(ora_cf32 ref[ORA_TU])
{
  memset(ref, 0, sizeof(ora_cf32) * ORA_TU);
  for (int i = 1; i <= ORA_K / 2; i++) {
    const float p = prs_phi(i), m = prs_phi(-i);
    ref[i].re = cosf(p);           ref[i].im = sinf(p);            /* glob_defs.h:150-154 */
    ref[ORA_TU - i].re = cosf(m);  ref[ORA_TU - i].im = sinf(m);
  }
}

/* decoder/fic_decoder.cpp:59-73 == backend/backend.cpp:72-84 */
void ora_prbs(uint8_t *out, int n)
{
  uint8_t sr[9];
  memset(sr, 1, 9);
  for (int i = 0; i < n; i++) {
    const uint8_t b = sr[8] ^ sr[4];
    for (int j = 8; j > 0; j--) sr[j] = sr[j - 1];
    sr[0] = b;
    out[i] = b;
  }
}
