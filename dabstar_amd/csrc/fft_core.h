// fft_core.h -- 2048-point complex FFT for one 256-thread workgroup, LDS-staged Stockham radix 8-8-8-4.
//
// Replaces fftwf_execute of the 2048-point c2c plans (base/main/dab_processor.cpp:63,201,276,338;
// ofdm/phasereference.cpp:51-52): unnormalised DFT, forward = e^{-j..}, backward = e^{+j..}.
// Thread `tid` enters with x[tid + 256 u] (u = 0..7: eight coalesced 2-KB loads of the T_u slice) and
// leaves with X[tid + 256 u]: input and output use the same strided register layout, so element-wise
// stages before/after the transform (NCO mix, x conj(PRS), |.|) never touch LDS.  Twiddles come from a
// table computed in double on the host (16 KB, L1/L2 resident), laid out per pass so that a wave's 64 lanes read
// consecutive entries (tables.cpp): a gather through the natural e^{-j 2 pi i/2048} order cost up to 64 cache lines
// per load in the third pass.
#pragma once
#include <hip/hip_runtime.h>

namespace dabx {

constexpr int FFT_LDS_FLOAT2 = 2048;             // exchange buffer of one transform (16 KB; swizzled, not padded: fft_pad)
constexpr int FFT_TW_P2 = 0, FFT_TW_P3 = 56, FFT_TW_P4 = 56 + 448;   // twiddle table sections: [7][8], [7][64], [3][512]

// Complex products as three packed instructions: two v_pk_mul_f32 whose op_sel picks (a.x, a.x) x (b.x, b.y) and
// (a.y, a.y) x (b.y, b.x), and one v_pk_add_f32 with a per-half negation -- the rounding of the scalar formula
// (four products rounded, then one add/subtract each).  hipcc finds the multiplies by itself but forms sum AND difference
// as two packed adds plus a move; the add is therefore spelled out.
typedef float fft_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cmul(float2 a, float2 b)        // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
{
  const fft_v2f A = {a.x, a.y}, W = {b.x, b.y};
  const fft_v2f t1 = __builtin_shufflevector(A, A, 0, 0) * W;
  const fft_v2f t2 = __builtin_shufflevector(A, A, 1, 1) * __builtin_shufflevector(W, W, 1, 0);
  fft_v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(t1), "v"(t2));
  return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)   // a * conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
{
  const fft_v2f A = {a.x, a.y}, W = {b.x, b.y};
  const fft_v2f t1 = __builtin_shufflevector(A, A, 0, 0) * W;
  const fft_v2f t2 = __builtin_shufflevector(A, A, 1, 1) * __builtin_shufflevector(W, W, 1, 0);
  fft_v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(t2), "v"(t1));
  return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
template <bool INV> __device__ __forceinline__ float2 mul_mj(float2 a)   // forward: * (-j); inverse: * (+j)
{
  return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
template <bool INV> __device__ __forceinline__ float2 tw_dir(float2 w) { return INV ? make_float2(w.x, -w.y) : w; }

template <bool INV> __device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d)
{
  const float2 s0 = cadd(a, c), d0 = csub(a, c), s1 = cadd(b, d), d1 = mul_mj<INV>(csub(b, d));
  a = cadd(s0, s1); c = csub(s0, s1); b = cadd(d0, d1); d = csub(d0, d1);
}

template <bool INV> __device__ __forceinline__ void dft8(float2 v[8])
{
  // even / odd DFT-4 then combine with W8^k
  dft4<INV>(v[0], v[2], v[4], v[6]);
  dft4<INV>(v[1], v[3], v[5], v[7]);
  const float r = 0.70710678118654752440f;
  const float2 o1 = INV ? make_float2((v[3].x - v[3].y) * r, (v[3].x + v[3].y) * r)
                        : make_float2((v[3].x + v[3].y) * r, (v[3].y - v[3].x) * r);      // * e^{-+ j pi/4}
  const float2 o2 = mul_mj<INV>(v[5]);
  const float2 o3 = INV ? make_float2((-v[7].x - v[7].y) * r, (v[7].x - v[7].y) * r)
                        : make_float2((v[7].y - v[7].x) * r, (-v[7].x - v[7].y) * r);     // * e^{-+ j 3pi/4}
  const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

// LDS slot of exchange-buffer element i: an XOR swizzle that makes EVERY access of the transform bank-conflict free under
// the gfx950 rules (MI355X_MICROARCH.md "LDS"): ds_write_b64 is serviced in four groups of 16 contiguous lanes with bank =
// dword mod 32, i.e. float2 slot mod 16; ds_read_b64 in two groups of 32 lanes with bank = float2 slot mod 32.
//   pass-1 writes  i = 8 j + t       -> bits 4..6 of i (= j >> 1 within a 16-lane group) go into slot bits 0..2, bit 6 into bit 3
//   pass-2 writes  i = 64 A + b + 8t -> b ^ const in bits 0..2, (A & 1) ^ (t & 1) in bit 3
//   pass-3 writes and all reads are contiguous in j: bits 0..3 permuted within an aligned 16-run, bit 4 untouched.
// The round-2 padding i + (i >> 4) was derived for 4-byte elements: with float2 every strided read and every pass-2 write was
// a 2-way conflict (SQ_LDS_BANK_CONFLICT = 82 % of the kernel's LDS-active cycles; model: tools/lds_conflicts.py).
__device__ __forceinline__ int fft_pad(int i) { return i ^ ((i >> 4) & 7) ^ (((i >> 6) & 1) << 3); }

// One radix-8 Stockham pass: v[t] = x[j + 256 t] in; results are scattered to LDS.  w = the pass's seven twiddles
// (tw[section + (t - 1) * NS + (j & (NS - 1))], fft_twiddles8), fetched by the caller one pass AHEAD.
template <bool INV, int NS> __device__ __forceinline__ void fft_pass8(float2 v[8], int j, float2 *lds, const float2 (&w)[7])
{
  const int k = j & (NS - 1);
  if (NS > 1) {
#pragma unroll
    for (int t = 1; t < 8; t++) v[t] = cmul(v[t], tw_dir<INV>(w[t - 1]));
  }
  dft8<INV>(v);
  const int base = (j - k) * 8 + k;
#pragma unroll
  for (int t = 0; t < 8; t++) lds[fft_pad(base + t * NS)] = v[t];
}
template <int NS> __device__ __forceinline__ void fft_twiddles8(float2 (&w)[7], int j, const float2 *tw)
{
#pragma unroll
  for (int t = 1; t < 8; t++) w[t - 1] = tw[(NS == 8 ? FFT_TW_P2 : FFT_TW_P3) + (t - 1) * NS + (j & (NS - 1))];
}

struct FftNoHook { __device__ void operator()() const {} };

// Whole transform.  All 256 threads of the block must call it; `lds` holds FFT_LDS_FLOAT2 float2.
// The twiddles of a pass are requested before the PREVIOUS pass computes -- their addresses depend on the thread index only --
// so the table look-ups (L1 / L2 hits, but a full memory round trip each) no longer sit between a barrier and the pass's
// first multiply.  before_last() runs where the last pass's twiddles are requested: callers put loads there whose results
// they need right after the transform.
template <bool INV, class Hook = FftNoHook>
__device__ __forceinline__ void fft2048(float2 v[8], float2 *lds, const float2 *tw, int tid, Hook before_last = Hook())
{
  float2 w2[7], w3[7], none[7] = {};
  fft_twiddles8<8>(w2, tid, tw);
  asm volatile("" ::: "memory");
  fft_pass8<INV, 1>(v, tid, lds, none);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; t++) v[t] = lds[fft_pad(tid + 256 * t)];
  __syncthreads();
  fft_twiddles8<64>(w3, tid, tw);
  asm volatile("" ::: "memory");
  fft_pass8<INV, 8>(v, tid, lds, w2);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; t++) v[t] = lds[fft_pad(tid + 256 * t)];
  __syncthreads();
  float2 w4[2][3];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int q = 0; q < 3; q++) w4[h][q] = tw[FFT_TW_P4 + 512 * q + tid + 256 * h];
  before_last();
  asm volatile("" ::: "memory");
  fft_pass8<INV, 64>(v, tid, lds, w3);
  __syncthreads();
  // last pass: radix 4, NS = 512, two butterflies per thread (j = tid and tid + 256); output index j + 512 t
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int j = tid + 256 * h;
    float2 a = lds[fft_pad(j)], b = lds[fft_pad(j + 512)], c = lds[fft_pad(j + 1024)], d = lds[fft_pad(j + 1536)];
    b = cmul(b, tw_dir<INV>(w4[h][0]));
    c = cmul(c, tw_dir<INV>(w4[h][1]));
    d = cmul(d, tw_dir<INV>(w4[h][2]));
    dft4<INV>(a, b, c, d);
    v[h] = a; v[h + 2] = b; v[h + 4] = c; v[h + 6] = d;     // X[j + 512 t] -> register u = 2 t + h
  }
  __syncthreads();
}

// The same transform with all twenty twiddles of the thread already in registers (they depend on the thread index only): for
// kernels that run many transforms per thread (persistent k_symbols) and must keep their vector-memory queue free of table
// look-ups -- vmcnt is in order, a twiddle fetched behind a prefetch would make the wait for it a wait for the prefetch.
struct FftTwiddles { float2 w2[7], w3[7], w4[2][3]; };
__device__ __forceinline__ void fft_load_twiddles(FftTwiddles &t, const float2 *tw, int tid)
{
  fft_twiddles8<8>(t.w2, tid, tw);
  fft_twiddles8<64>(t.w3, tid, tw);
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int q = 0; q < 3; q++) t.w4[h][q] = tw[FFT_TW_P4 + 512 * q + tid + 256 * h];
}
template <bool INV> __device__ __forceinline__ void fft2048_regs(float2 v[8], float2 *lds, const FftTwiddles &t, int tid)
{
  const float2 none[7] = {};
  fft_pass8<INV, 1>(v, tid, lds, none);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 8; u++) v[u] = lds[fft_pad(tid + 256 * u)];
  __syncthreads();
  fft_pass8<INV, 8>(v, tid, lds, t.w2);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 8; u++) v[u] = lds[fft_pad(tid + 256 * u)];
  __syncthreads();
  fft_pass8<INV, 64>(v, tid, lds, t.w3);
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int j = tid + 256 * h;
    float2 a = lds[fft_pad(j)], b = lds[fft_pad(j + 512)], c = lds[fft_pad(j + 1024)], d = lds[fft_pad(j + 1536)];
    b = cmul(b, tw_dir<INV>(t.w4[h][0]));
    c = cmul(c, tw_dir<INV>(t.w4[h][1]));
    d = cmul(d, tw_dir<INV>(t.w4[h][2]));
    dft4<INV>(a, b, c, d);
    v[h] = a; v[h + 2] = b; v[h + 4] = c; v[h + 6] = d;
  }
  __syncthreads();
}

}  // namespace dabx
