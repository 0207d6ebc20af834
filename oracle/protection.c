/* protection.c -- EEP/UEP/FIC depuncturing maps (oracle; test infrastructure only). */
#include "dab_oracle.h"
#include <stdlib.h>
#include <string.h>

/* protection/eep_protection.cpp:153-167 == uep_protection.cpp:198-212:
 * L blocks of 128 mother-code bits, each punctured by PI (period 32). */
static void add_blocks(int32_t *map, int *pos, int *in_idx, int L, const int8_t *pi)
{
  for (int i = 0; i < L; i++)
    for (int j = 0; j < 128; j++) {
      map[*pos] = (pi != NULL && pi[j % 32] != 0) ? (*in_idx)++ : -1;
      (*pos)++;
    }
}

/* tail: 24 bits punctured with PI_8[0..23]  (eep_protection.cpp:137-150) */
static void add_tail(int32_t *map, int *pos, int *in_idx)
{
  const int8_t *pix = ora_pi_codes(8);
  for (int i = 0; i < 24; i++) {
    map[*pos] = pix[i] ? (*in_idx)++ : -1;
    (*pos)++;
  }
}

/* protection/eep_protection.cpp:43-150 */
int ora_eep_map(int kbps, int prot_level, int32_t *map)
{
  int L1, L2, pi1, pi2;
  const int lvl = prot_level & 3, option = (prot_level >> 2) & 1;
  if (option == 0) {                      /* A profiles, EN 300 401 11.3.2 table 18 */
    if (kbps % 8) return -1;
    const int n = kbps / 8;
    switch (lvl) {
    case 0: L1 = 6 * n - 3; L2 = 3; pi1 = 24; pi2 = 23; break;
    case 1:
      if (n == 1) { L1 = 5; L2 = 1; pi1 = 13; pi2 = 12; }
      else { L1 = 2 * n - 3; L2 = 4 * n + 3; pi1 = 14; pi2 = 13; }
      break;
    case 2: L1 = 6 * n - 3; L2 = 3; pi1 = 8; pi2 = 7; break;
    default: L1 = 4 * n - 3; L2 = 2 * n + 3; pi1 = 3; pi2 = 2; break;
    }
  } else {                                /* B profiles, table 19 */
    if (kbps % 32) return -1;
    const int n = kbps / 32;
    static const int pib[4] = {10, 6, 4, 2};
    L1 = 24 * n - 3; L2 = 3; pi1 = pib[lvl]; pi2 = pib[lvl] - 1;
  }
  int pos = 0, in_idx = 0;
  add_blocks(map, &pos, &in_idx, L1, ora_pi_codes(pi1));
  add_blocks(map, &pos, &in_idx, L2, ora_pi_codes(pi2));
  add_tail(map, &pos, &in_idx);
  return (pos == 96 * kbps + 24) ? in_idx : -2;
}

/* EN 300 401 table 8 / 11.3.1 as held by protection/uep_protection.cpp:52-134:
 * {kbps, level, L1..L4, PI1..PI4}; PI4 = 0 means "no fourth block". The row
 * {80,1,...,PI2=7} is kept as in the reference (uep_protection.cpp:81). */
static const int16_t uep_tab[][10] = {
  {32,5,3,4,17,0,5,3,2,0}, {32,4,3,3,18,0,11,6,5,0}, {32,3,3,4,14,3,15,9,6,8},
  {32,2,3,4,14,3,22,13,8,13}, {32,1,3,5,13,3,24,17,12,17}, {48,5,4,3,26,3,5,4,2,3},
  {48,4,3,4,26,3,9,6,4,6}, {48,3,3,4,26,3,15,10,6,9}, {48,2,3,4,26,3,24,14,8,15},
  {48,1,3,5,25,3,24,18,13,18}, {56,5,6,10,23,3,5,4,2,3}, {56,4,6,10,23,3,9,6,4,5},
  {56,3,6,12,21,3,16,7,6,9}, {56,2,6,10,23,3,23,13,8,13}, {64,5,6,9,31,2,5,3,2,3},
  {64,4,6,9,33,0,11,6,5,0}, {64,3,6,12,27,3,16,8,6,9}, {64,2,6,10,29,3,23,13,8,13},
  {64,1,6,11,28,3,24,18,12,18}, {80,5,6,10,41,3,6,3,2,3}, {80,4,6,10,41,3,11,6,5,6},
  {80,3,6,11,40,3,16,8,6,7}, {80,2,6,10,41,3,23,13,8,13}, {80,1,6,10,41,3,24,7,12,18},
  {96,5,7,9,53,3,5,4,2,4}, {96,4,7,10,52,3,9,6,4,6}, {96,3,6,12,51,3,16,9,6,10},
  {96,2,6,10,53,3,22,12,9,12}, {96,1,6,13,50,3,24,18,13,19}, {112,5,14,17,50,3,5,4,2,5},
  {112,4,11,21,49,3,9,6,4,8}, {112,3,11,23,47,3,16,8,6,9}, {112,2,11,21,49,3,23,12,9,14},
  {128,5,12,19,62,3,5,3,2,4}, {128,4,11,21,61,3,11,6,5,7}, {128,3,11,22,60,3,16,9,6,10},
  {128,2,11,21,61,3,22,12,9,14}, {128,1,11,20,62,3,24,17,13,19}, {160,5,11,19,87,3,5,4,2,4},
  {160,4,11,23,83,3,11,6,5,9}, {160,3,11,24,82,3,16,8,6,11}, {160,2,11,21,85,3,22,11,9,13},
  {160,1,11,22,84,3,24,18,12,19}, {192,5,11,20,110,3,6,4,2,5}, {192,4,11,22,108,3,10,6,4,9},
  {192,3,11,24,106,3,16,10,6,11}, {192,2,11,20,110,3,22,13,9,13}, {192,1,11,21,109,3,24,20,13,24},
  {224,5,12,22,131,3,8,6,2,6}, {224,4,12,26,127,3,12,8,4,11}, {224,3,11,20,134,3,16,10,7,9},
  {224,2,11,22,132,3,24,16,10,15}, {224,1,11,24,130,3,24,20,12,20}, {256,5,11,24,154,3,6,5,2,5},
  {256,4,11,24,154,3,12,9,5,10}, {256,3,11,27,151,3,16,10,7,10}, {256,2,11,22,156,3,24,14,10,13},
  {256,1,11,26,152,3,24,19,14,18}, {320,5,11,26,200,3,8,5,2,6}, {320,4,11,25,201,3,13,9,5,10},
  {320,2,11,26,200,3,24,17,9,17}, {384,5,11,27,247,3,8,6,2,7}, {384,3,11,24,250,3,16,9,7,10},
  {384,1,12,28,245,3,24,20,14,23}};

/* protection/uep_protection.cpp:136-196 */
int ora_uep_map(int kbps, int prot_level, int32_t *map)
{
  const int nrows = (int)(sizeof(uep_tab) / sizeof(uep_tab[0]));
  int row = -1;
  for (int i = 0; i < nrows; i++)
    if (uep_tab[i][0] == kbps && uep_tab[i][1] == prot_level) { row = i; break; }
  if (row < 0) return -1;   /* the reference falls back to row 1 with a qCritical; the oracle refuses */
  const int16_t *r = uep_tab[row];
  int pos = 0, in_idx = 0;
  for (int b = 0; b < 4; b++)
    add_blocks(map, &pos, &in_idx, r[2 + b], r[6 + b] ? ora_pi_codes(r[6 + b]) : NULL);
  add_tail(map, &pos, &in_idx);
  return (pos == 96 * kbps + 24) ? in_idx : -2;
}

/* decoder/fic_decoder.cpp:79-124 : 21 blocks PI_16, 3 blocks PI_15, tail PI_8 */
int ora_fic_map(int32_t map[3096])
{
  int pos = 0, in_idx = 0;
  add_blocks(map, &pos, &in_idx, 21, ora_pi_codes(16));
  add_blocks(map, &pos, &in_idx, 3, ora_pi_codes(15));
  add_tail(map, &pos, &in_idx);
  return in_idx;  /* 2304 */
}

/* protection/protection.cpp:46-59 : scatter into the zero-initialised mother-code block, decode */
void ora_deconvolve(const int16_t *in, const int32_t *map, int kbps, uint8_t *out_bits)
{
  const int n = 96 * kbps + 24;
  int16_t *blk = (int16_t *)calloc((size_t)n, sizeof(int16_t));
  for (int i = 0; i < n; i++)
    if (map[i] >= 0) blk[i] = in[map[i]];
  ora_viterbi_build(blk, 24 * kbps, out_bits);
  free(blk);
}
