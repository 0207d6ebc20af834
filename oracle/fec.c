/* fec.c -- CRC-16, DAB+ fire code, RS(120,110) (oracle; test infrastructure only). */
#include "dab_oracle.h"
#include <string.h>

/* ------------------------------------------------------------------ CRC-16-CCITT */
/* backend/crc.cpp:98-132: bit-serial LFSR, x^16+x^12+x^5+1, register all ones, the
 * last 16 message bits inverted; valid when the register ends at zero. */
int ora_check_crc_bits(const uint8_t *bits, int nbits)
{
  uint16_t reg = 0xFFFF;
  for (int i = 0; i < nbits; i++) {
    const int inv = (i >= nbits - 16) ? 1 : 0;
    const int fb = ((reg >> 15) & 1) ^ ((bits[i] ^ inv) & 1);
    reg = (uint16_t)(reg << 1);
    if (fb) reg ^= 0x1021;
  }
  return reg == 0;
}

/* backend/crc.cpp:75-86 (table driven there; same polynomial, MSB first) */
uint16_t ora_calc_crc(const uint8_t *data, int len)
{
  uint16_t crc = 0xFFFF;
  for (int i = 0; i < len; i++) {
    crc ^= (uint16_t)(data[i] << 8);
    for (int b = 0; b < 8; b++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
  }
  return (uint16_t)~crc;
}

/* backend/crc.cpp:88-96 */
int ora_check_crc_bytes(const uint8_t *msg, int len)
{
  const uint16_t acc = ora_calc_crc(msg, len);
  const uint16_t crc = (uint16_t)((msg[len] << 8) | msg[len + 1]);
  return (crc ^ acc) == 0;
}

/* ------------------------------------------------------------------ fire code */
/* backend/firecode_checker.h:53-70 : g(x) = 0x782F; the burst patterns (data). */
static const uint8_t fc_pattern[124] = {
  17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 30, 31, 34, 36, 38, 40, 42, 44, 46, 50, 52, 54,
  56, 60, 62, 68, 72, 76, 84, 88, 92, 100, 104, 108, 120, 124, 136, 152, 168, 184, 200, 216, 248,
  33, 35, 37, 39, 41, 43, 45, 49, 51, 53, 55, 57, 59, 61, 63,
  66, 70, 74, 78, 82, 86, 90, 98, 102, 106, 110, 114, 118, 122, 126,
  132, 140, 148, 156, 164, 172, 180, 196, 204, 212, 220, 228, 236, 244, 252,
  1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 29, 32, 48, 58, 64, 80, 96, 112, 116, 128,
  144, 160, 176, 192, 208, 224, 232, 240};

static uint16_t fc_syn[65536];
static int fc_ready = 0;

/* firecode_checker.cpp:146-160 : bytes 2..10 then bytes 0..1 through the CRC register */
static uint16_t fc_crc16(const uint8_t *x)
{
  uint16_t crc = 0;
  static const int order[11] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 0, 1};
  for (int k = 0; k < 11; k++) {
    crc ^= (uint16_t)(x[order[k]] << 8);
    for (int b = 0; b < 8; b++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x782F) : (uint16_t)(crc << 1);
  }
  return crc;
}

/* firecode_checker.cpp:61-144 : first writer wins for each syndrome */
static void fc_try(uint8_t *err, int bit, int pat)
{
  const uint16_t s = fc_crc16(err);
  if (fc_syn[s] == 0) fc_syn[s] = (uint16_t)((bit << 8) + pat);
}

static void fc_build(void)
{
  uint8_t err[11];
  memset(err, 0, 11);
  memset(fc_syn, 0, sizeof(fc_syn));
  for (int i = 0; i < 11; i++)                         /* aligned bursts */
    for (int j = 0; j < 124; j++) { err[i] = fc_pattern[j]; fc_try(err, i * 8, fc_pattern[j]); err[i] = 0; }
  static const struct { int sh, j0, j1; } pass[3] = {{4, 0, 45}, {2, 45, 75}, {6, 60, 90}};
  for (int p = 0; p < 3; p++)
    for (int i = 0; i < 10; i++)
      for (int j = pass[p].j0; j < pass[p].j1; j++) {
        err[i] = (uint8_t)(fc_pattern[j] >> pass[p].sh);
        err[i + 1] = (uint8_t)(fc_pattern[j] << (8 - pass[p].sh));
        fc_try(err, i * 8 + pass[p].sh, fc_pattern[j]);
        err[i] = 0; err[i + 1] = 0;
      }
  fc_ready = 1;
}

const uint16_t *ora_firecode_syndrome_table(void)
{
  if (!fc_ready) fc_build();
  return fc_syn;
}

int ora_firecode_check(const uint8_t x[11]) { return fc_crc16(x) == 0; }

/* firecode_checker.cpp:168-184 */
int ora_firecode_check_and_correct(uint8_t x[11])
{
  if (!fc_ready) fc_build();
  const uint16_t s = fc_crc16(x);
  if (s == 0) return 1;
  const uint8_t e = (uint8_t)(fc_syn[s] & 0xff);
  if (e) {
    const int bit = fc_syn[s] >> 8;
    x[bit / 8] ^= (uint8_t)(e >> (bit % 8));
    x[bit / 8 + 1] ^= (uint8_t)(e << (8 - (bit % 8)));   /* may touch x[11] when bit/8 == 10, as in the reference */
    return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------ GF(256) + RS */
/* backend/galois.cpp:37-142 : GF(2^8), poly 0x11D, alpha = 2 */
static uint8_t gf_exp[256], gf_log[256];
static int gf_ready = 0;
#define NN 255
#define NROOTS 10

static void gf_build(void)
{
  unsigned sr = 1;
  gf_log[0] = NN; gf_exp[NN] = 0;
  for (int i = 0; i < NN; i++) {
    gf_log[sr] = (uint8_t)i; gf_exp[i] = (uint8_t)sr;
    sr <<= 1;
    if (sr & 0x100) sr ^= 0x11D;
    sr &= NN;
  }
  gf_ready = 1;
}
static int modnn(int x) { while (x >= NN) { x -= NN; x = (x >> 8) + (x & NN); } return x; }
static int mul_pow(int a, int b) { return modnn(a + b); }
static int mul_poly(int a, int b) { return (a == 0 || b == 0) ? 0 : gf_exp[mul_pow(gf_log[a], gf_log[b])]; }
static int div_poly(int a, int b) { return a == 0 ? 0 : gf_exp[modnn(NN + gf_log[a] - gf_log[b])]; }
static int pow_pow(int a, int n) { return a == 0 ? 0 : (a * n) % NN; }

static uint8_t rs_gen[NROOTS + 1];   /* index form */
static int rs_ready = 0;

/* reed_solomon.cpp:40-82 with fcr = 0, prim = 1 */
static void rs_build(void)
{
  if (!gf_ready) gf_build();
  uint8_t g[NROOTS + 1];
  memset(g, 0, sizeof(g));
  g[0] = 1;
  for (int i = 0, root = 0; i < NROOTS; i++, root++) {
    g[i + 1] = 1;
    for (int j = i; j > 0; j--)
      g[j] = g[j] ? (uint8_t)(g[j - 1] ^ gf_exp[mul_pow(gf_log[g[j]], root)]) : g[j - 1];
    g[0] = gf_exp[mul_pow(root, gf_log[g[0]])];
  }
  for (int i = 0; i <= NROOTS; i++) rs_gen[i] = gf_log[g[i]];
  rs_ready = 1;
}

/* reed_solomon.cpp:85-137 (encode_rs + enc, cutlen 135) */
void ora_rs_enc(const uint8_t in[110], uint8_t out[120])
{
  if (!rs_ready) rs_build();
  uint8_t rf[NN], bb[NROOTS];
  memset(rf, 0, 135);
  memcpy(rf + 135, in, 110);
  memset(bb, 0, NROOTS);
  for (int i = 0; i < NN - NROOTS; i++) {
    const int fb = gf_log[rf[i] ^ bb[0]];
    if (fb != NN)
      for (int j = 1; j < NROOTS; j++) bb[j] ^= gf_exp[mul_pow(fb, rs_gen[NROOTS - j])];
    memmove(bb, bb + 1, NROOTS - 1);
    bb[NROOTS - 1] = (fb != NN) ? gf_exp[mul_pow(fb, rs_gen[0])] : 0;
  }
  memcpy(out, in, 110);
  memcpy(out + 110, bb, NROOTS);
}

/* reed_solomon.cpp:160-439 : syndromes (Horner), Berlekamp-Massey, Chien, Forney.
 * Kept structurally equivalent including the failure paths that leave data partly
 * modified (:223-227, :153-157). */
static int rs_decode255(uint8_t *data)
{
  uint8_t syn[NROOTS], lambda[NROOTS + 1], root_tab[NROOTS], loc_tab[NROOTS], omega[NROOTS + 1];
  int syn_err = 0;
  for (int r = 0; r < NROOTS; r++) {                     /* :254-290, root alpha^r */
    int s = data[0];
    for (int j = 1; j < NN; j++)
      s = (s == 0) ? data[j] : (data[j] ^ gf_exp[mul_pow(gf_log[s], pow_pow(mul_pow(0, r), 1))]);
    syn[r] = (uint8_t)s; syn_err |= s;
  }
  if (!syn_err) return 0;

  /* Berlekamp-Massey, :296-361 */
  uint8_t corr[NROOTS], oldl[NROOTS];
  int K = 1, L = 0, deg_lambda = 0;
  memset(corr, 0, sizeof(corr)); memset(lambda, 0, sizeof(lambda));
  int error = syn[0];
  lambda[0] = 1; corr[1] = 1;
  while (K < NROOTS) {
    memcpy(oldl, lambda, NROOTS);
    for (int i = 0; i < NROOTS; i++) lambda[i] ^= (uint8_t)mul_poly(error, corr[i]);
    if (2 * L < K && error != 0) {
      L = K - L;
      for (int i = 0; i < NROOTS; i++) corr[i] = (uint8_t)div_poly(oldl[i], error);
    }
    for (int i = NROOTS - 1; i >= 1; i--) corr[i] = corr[i - 1];
    corr[0] = 0;
    error = syn[K];
    for (int i = 1; i <= K; i++) error ^= mul_poly(syn[K - i], lambda[i]);
    K++;
  }
  for (int i = 0; i < NROOTS; i++) lambda[i] ^= (uint8_t)mul_poly(error, corr[i]);
  for (int i = 0; i < NROOTS; i++) {
    if (lambda[i] != 0) deg_lambda = i;
    lambda[i] = gf_log[lambda[i]];
  }
  /* NB the reference copies nroots+1 entries of Lambda into its work register although
   * only nroots were written (:374); entry [nroots] is never read because
   * deg_lambda <= nroots-1. */

  /* Chien search, :367-402 (iprim = 1 -> k starts at 0 and steps by 1, no modulo) */
  uint8_t work[NROOTS + 1];
  memcpy(work, lambda, NROOTS); work[NROOTS] = NN;
  int root_count = 0;
  for (int i = 1, k = 0; i <= NN; i++, k++) {
    int result = 1;
    for (int j = deg_lambda; j > 0; j--)
      if (work[j] != NN) { work[j] = (uint8_t)mul_pow(work[j], j); result ^= gf_exp[work[j]]; }
    if (result != 0) continue;
    if (root_count < NROOTS) { root_tab[root_count] = (uint8_t)i; loc_tab[root_count] = (uint8_t)k; }
    root_count++;
  }
  if (root_count != deg_lambda) return -1;

  /* omega = s*lambda mod x^nroots, :411-439 */
  int deg_omega = 0;
  for (int i = 0; i < NROOTS; i++) {
    int tmp = 0;
    for (int j = (deg_lambda < i) ? deg_lambda : i; j >= 0; j--)
      if (gf_log[syn[i - j]] != NN && lambda[j] != NN) tmp ^= gf_exp[mul_pow(gf_log[syn[i - j]], lambda[j])];
    if (tmp != 0) deg_omega = i;
    omega[i] = gf_log[tmp];
  }
  omega[NROOTS] = NN;

  /* Forney, :189-251 */
  for (int j = root_count - 1; j >= 0; j--) {
    int num1 = 0;
    for (int i = deg_omega; i >= 0; i--)
      if (omega[i] != NN) num1 ^= gf_exp[mul_pow(omega[i], pow_pow(i, root_tab[j]))];
    /* num2 = inv(X)^(fcr-1): pow_power(root, divide_power(0,1) = 254) then * codeLength (== *1) */
    const int num2 = gf_exp[mul_pow(pow_pow(root_tab[j], modnn(NN + 0 - 1)), NN)];
    int den = 0;
    const int lim = ((deg_lambda < NROOTS - 1) ? deg_lambda : NROOTS - 1) & ~1;
    for (int i = lim; i >= 0; i -= 2)
      if (lambda[i + 1] != NN) den ^= gf_exp[mul_pow(lambda[i + 1], pow_pow(i, root_tab[j]))];
    if (den == 0) return -1;
    if (num1 != 0) {
      if (loc_tab[j] >= (uint8_t)(NN - NROOTS)) root_count--;
      else {
        const int t1 = NN - gf_log[den];
        int t2 = mul_pow(gf_log[num1], gf_log[num2]);
        t2 = mul_pow(t2, t1);
        data[loc_tab[j]] ^= gf_exp[t2];
      }
    }
  }
  return root_count;
}

/* reed_solomon.cpp:140-158 with cutlen = 135 (mp4processor.cpp:203) */
int ora_rs_dec(const uint8_t in[120], uint8_t out[110])
{
  if (!rs_ready) rs_build();
  uint8_t rf[NN];
  memset(rf, 0, 135);
  memcpy(rf + 135, in, 120);
  const int ret = rs_decode255(rf);
  memcpy(out, rf + 135, 110);
  return ret;
}
