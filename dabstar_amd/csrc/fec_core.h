// fec_core.h -- device functions: CRC-16, DAB+ fire code, RS(120,110) over GF(2^8).
// Integer/byte work, one lane per code word / header; tables are read through L1/L2 (<= 130 KB total).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dabx {

// check_crc_bytes / calc_crc (base/backend/crc.cpp:75-96): CRC-16-CCITT, init 0xFFFF, complemented
__device__ __forceinline__ uint16_t crc16_ccitt(const uint8_t *d, int len, const uint16_t *tab)
{
  uint16_t crc = 0xFFFF;
  for (int i = 0; i < len; i++) crc = (uint16_t)(tab[(d[i] ^ (crc >> 8)) & 0xFF] ^ (crc << 8));
  return (uint16_t)~crc;
}
__device__ __forceinline__ bool crc16_check_bytes(const uint8_t *msg, int len, const uint16_t *tab)
{
  return crc16_ccitt(msg, len, tab) == (uint16_t)((msg[len] << 8) | msg[len + 1]);
}

// a(x) b(x) mod x^16 + x^12 + x^5 + 1 (carry-less, Horner over the bits of a): moves a CRC-16-CCITT register value over
// n message bytes when b = x^(8 n) mod P
__device__ __forceinline__ unsigned crc_mulmod(unsigned a, unsigned b)
{
  unsigned r = 0;
#pragma unroll
  for (int i = 15; i >= 0; i--) {
    r <<= 1;
    if (r & 0x10000u) r ^= 0x11021u;
    if ((a >> i) & 1u) r ^= b;
  }
  return r;
}

// FirecodeChecker::crc16 (base/backend/firecode_checker.cpp:146-160): bytes 2..10 then 0..1
template <class Get> __device__ __forceinline__ uint16_t firecode_syndrome(Get get, const uint16_t *fctab)
{
  uint16_t crc = 0;
#pragma unroll
  for (int i = 2; i < 11; i++) crc = (uint16_t)((crc << 8) ^ fctab[(crc >> 8) ^ get(i)]);
  crc = (uint16_t)((crc << 8) ^ fctab[(crc >> 8) ^ get(0)]);
  crc = (uint16_t)((crc << 8) ^ fctab[(crc >> 8) ^ get(1)]);
  return crc;
}

// check_and_correct_6bits (firecode_checker.cpp:168-184); x has >= 12 writable bytes
__device__ __forceinline__ bool firecode_check_and_correct(uint8_t *x, const uint16_t *fctab, const uint16_t *syn)
{
  const uint16_t s = firecode_syndrome([&](int i) { return x[i]; }, fctab);
  if (s == 0) return true;
  const uint16_t e = syn[s];
  if (e & 0xFF) {
    const int bit = e >> 8;
    x[bit / 8] ^= (uint8_t)((e & 0xFF) >> (bit % 8));
    x[bit / 8 + 1] ^= (uint8_t)((e & 0xFF) << (8 - (bit % 8)));
    return true;
  }
  return false;
}

// ---- GF(2^8) helpers in the reference's index ("power") arithmetic, base/backend/galois.cpp:68-142
struct Gf {
  const uint8_t *ex;   // [512] alpha^i, doubled so that i+j needs no reduction
  const uint8_t *lg;   // [256], lg[0] = 255
  __device__ int mul(int a, int b) const { return (a == 0 || b == 0) ? 0 : ex[lg[a] + lg[b]]; }
  __device__ int div(int a, int b) const { return a == 0 ? 0 : ex[255 + lg[a] - lg[b]]; }   // divide_poly: b == 0 -> lg = 255
  __device__ int exp255(int p) const { return p == 255 ? 0 : ex[p]; }                        // power2poly incl. "alpha^-inf"
};
__device__ __forceinline__ int modnn(int x) { while (x >= 255) { x -= 255; x = (x >> 8) + (x & 255); } return x; }

struct CwArray {      // code word in a private array
  uint8_t *p;
  __device__ int get(int k) const { return p[k]; }
  __device__ void xor_at(int k, uint8_t v) const { p[k] ^= v; }
};
struct CwStrided {    // code word j of an RS-interleaved super frame: bytes j + k R
  uint8_t *p;
  int stride;
  __device__ int get(int k) const { return p[k * stride]; }
  __device__ void xor_at(int k, uint8_t v) const { p[k * stride] ^= v; }
};

// ReedSolomon::dec(in, out, 135) with (8, 0435, 0, 1, 10): base/backend/reed_solomon.cpp:140-439.
// cw: 120 received bytes (data 110 + parity 10), corrected in place exactly where the reference
// corrects (including the partial corrections of its failure paths).  Returns #corrected / 0 / -1.
// The decoder in its three parts (reed_solomon.cpp:296-439), so that k_dabplus can run the middle one -- the Chien search over all 255 positions, the
// longest -- with the whole wave on one code word instead of one lane on each.
constexpr int RS_NR = 10, RS_NN = 255, RS_PAD = 135;
// :296-361 Berlekamp-Massey: syndromes -> the locator in index form (255 = zero coefficient) and its degree
__device__ inline void rs_berlekamp_massey(const uint8_t syn[RS_NR], const Gf &gf, uint8_t lambda[RS_NR + 1], int &deg_lambda)
{
  constexpr int NR = RS_NR;
  uint8_t corr[NR], oldl[NR];
  for (int i = 0; i < NR; i++) { lambda[i] = 0; corr[i] = 0; }
  lambda[NR] = 0;
  int Kk = 1, Ll = 0, error = syn[0];
  deg_lambda = 0;
  lambda[0] = 1; corr[1] = 1;
  while (Kk < NR) {
    for (int i = 0; i < NR; i++) oldl[i] = lambda[i];
    for (int i = 0; i < NR; i++) lambda[i] ^= (uint8_t)gf.mul(error, corr[i]);
    if (2 * Ll < Kk && error != 0) {
      Ll = Kk - Ll;
      for (int i = 0; i < NR; i++) corr[i] = (uint8_t)gf.div(oldl[i], error);
    }
    for (int i = NR - 1; i >= 1; i--) corr[i] = corr[i - 1];
    corr[0] = 0;
    error = syn[Kk];
    for (int i = 1; i <= Kk; i++) error ^= gf.mul(syn[Kk - i], lambda[i]);
    Kk++;
  }
  for (int i = 0; i < NR; i++) lambda[i] ^= (uint8_t)gf.mul(error, corr[i]);
  for (int i = 0; i < NR; i++) {
    if (lambda[i] != 0) deg_lambda = i;
    lambda[i] = gf.lg[lambda[i]];            // to index form; 255 = zero
  }
}
// :367-402 Chien search over all 255 positions, one lane: the first NR roots (and their locations) in the order found, and how many there are
__device__ inline int rs_chien(const uint8_t lambda[RS_NR + 1], int deg_lambda, const Gf &gf, uint8_t root_tab[RS_NR], uint8_t loc_tab[RS_NR])
{
  constexpr int NR = RS_NR, NN = RS_NN;
  uint8_t work[NR];
  for (int i = 0; i < NR; i++) work[i] = lambda[i];
  int root_count = 0;
  for (int i = 1; i <= NN; i++) {
    int result = 1;
    for (int j = deg_lambda; j > 0; j--)
      if (work[j] != NN) { work[j] = (uint8_t)modnn(work[j] + j); result ^= gf.ex[work[j]]; }
    if (result != 0) continue;
    if (root_count < NR) { root_tab[root_count] = (uint8_t)i; loc_tab[root_count] = (uint8_t)(i - 1); }
    root_count++;
  }
  return root_count;
}
// ... the same value for ONE position (what a lane of the wave-wide search evaluates): position i is a root when this is 0.  (The serial loop's
// running work[j] is (lambda[j] + i j) mod 255 at position i.)
__device__ __forceinline__ int rs_chien_at(const uint8_t *lambda, int deg_lambda, const uint8_t *ex, int i)
{
  int result = 1;
  for (int j = deg_lambda; j > 0; j--)
    if (lambda[j] != RS_NN) result ^= ex[(lambda[j] + i * j) % RS_NN];
  return result;
}
// :411-439 omega, :189-251 Forney: the corrections, in place exactly where the reference corrects (root_count == deg_lambda has been checked)
template <class CW>
__device__ inline int rs_forney(CW cw, const Gf &gf, const uint8_t syn[RS_NR], const uint8_t lambda[RS_NR + 1], int deg_lambda,
                                const uint8_t *root_tab, int root_count)
{
  constexpr int NR = RS_NR, NN = RS_NN, PAD = RS_PAD;
  uint8_t omega[NR + 1];
  int deg_omega = 0;
  for (int i = 0; i < NR; i++) {
    int tmp = 0;
    for (int j = (deg_lambda < i) ? deg_lambda : i; j >= 0; j--)
      if (syn[i - j] != 0 && lambda[j] != NN) tmp ^= gf.ex[gf.lg[syn[i - j]] + lambda[j]];
    if (tmp != 0) deg_omega = i;
    omega[i] = gf.lg[tmp];
  }
  omega[NR] = NN;
  for (int j = root_count - 1; j >= 0; j--) {
    const int root = root_tab[j], loc = root - 1;
    int num1 = 0;
    for (int i = deg_omega; i >= 0; i--)
      if (omega[i] != NN) num1 ^= gf.ex[modnn(omega[i] + (i * root) % NN)];
    const int num2 = gf.ex[(root * 254) % NN];   // pow_power(root, 254) then * alpha^255 (= 1); root >= 1
    int den = 0;
    const int lim = ((deg_lambda < NR - 1) ? deg_lambda : NR - 1) & ~1;
    for (int i = lim; i >= 0; i -= 2)
      if (lambda[i + 1] != NN) den ^= gf.ex[modnn(lambda[i + 1] + (i * root) % NN)];
    if (den == 0) return -1;
    if (num1 != 0) {
      if (loc >= NN - NR) root_count--;
      else {
        int t2 = modnn(gf.lg[num1] + gf.lg[num2]);
        t2 = modnn(t2 + NN - gf.lg[den]);
        const int k = loc - PAD;
        if (k >= 0) cw.xor_at(k, gf.ex[t2]);   // positions < 135 lie in the virtual zero padding (discarded)
      }
    }
  }
  return root_count;
}

// syn_in (optional): the ten syndromes S_r = XOR_k c_k alpha^(r (119 - k)) when the caller has them already (k_dabplus evaluates them lane-parallel for
// all code words of a super frame: the Horner recursion below is the same sum, 1200 dependent table look-ups on one lane)
template <class CW>   // CW: byte accessor with uint8_t get(int k) / void xor_at(int k, uint8_t v), k < 120
__device__ inline int rs_decode_120(CW cw, const Gf &gf, const uint8_t *syn_in = nullptr)
{
  constexpr int NR = RS_NR;
  uint8_t syn[NR];
  int syn_err = 0;
  if (syn_in) {
    for (int r = 0; r < NR; r++) { syn[r] = syn_in[r]; syn_err |= syn[r]; }
  } else {
    for (int r = 0; r < NR; r++) {           // :254-290 Horner; the 135 leading zeros contribute nothing
      int s = 0;
      for (int j = 0; j < 120; j++) { const int b = cw.get(j); s = (s == 0) ? b : (b ^ gf.ex[gf.lg[s] + r]); }
      syn[r] = (uint8_t)s;
      syn_err |= s;
    }
  }
  if (!syn_err) return 0;
  uint8_t lambda[NR + 1], root_tab[NR], loc_tab[NR];
  int deg_lambda;
  rs_berlekamp_massey(syn, gf, lambda, deg_lambda);
  const int root_count = rs_chien(lambda, deg_lambda, gf, root_tab, loc_tab);
  if (root_count != deg_lambda) return -1;
  return rs_forney(cw, gf, syn, lambda, deg_lambda, root_tab, root_count);
}

}  // namespace dabx
