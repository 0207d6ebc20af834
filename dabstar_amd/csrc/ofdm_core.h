// ofdm_core.h -- device building blocks of the OFDM front end (256-thread workgroups, one per stream/symbol).
#pragma once
#include <hip/hip_runtime.h>
#include "dabx_internal.h"
#include "fft_core.h"
#include "wave_ops.h"

namespace dabx {

__device__ __forceinline__ float cabsf_(float2 z) { return sqrtf(z.x * z.x + z.y * z.y); }   // std::abs under -ffast-math
// |z| with the 1-ulp hardware square root: only for the signal-level tracker, which is itself applied per chunk
// (k_frame_tail) and feeds nothing but the out-of-lock dip detector
__device__ __forceinline__ float cabsf_level(float2 z) { return __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y); }

// ---- block reductions (256 threads) -------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float *red /* >= 8 floats of LDS */, int tid)
{
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float s = 0.f;
  for (int w = 0; w < nw; w++) s += red[w];
  return s;
}
// two sums in one pass for blocks of at most 4 waves (red[0..3] / red[4..7]); same summation order as block_sum
__device__ __forceinline__ void block_sum2(float &a, float &b, float *red /* >= 8 floats of LDS */, int tid)
{
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = a; red[4 + (tid >> 6)] = b; }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float sa = 0.f, sb = 0.f;
  for (int w = 0; w < nw; w++) { sa += red[w]; sb += red[4 + w]; }
  a = sa; b = sb;
}
// two sums in one pass for blocks of up to 16 waves (red[0..15] / red[16..31]); summation order of block_sum
__device__ __forceinline__ void block_sum2w(float &a, float &b, float *red /* >= 32 floats of LDS */, int tid)
{
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = a; red[16 + (tid >> 6)] = b; }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float sa = 0.f, sb = 0.f;
  for (int w = 0; w < nw; w++) { sa += red[w]; sb += red[16 + w]; }
  a = sa; b = sb;
}
__device__ __forceinline__ int block_min_int(int v, int *red, int tid)
{
  v = wave_min_int(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const int nw = blockDim.x >> 6;
  int s = red[0];
  for (int w = 1; w < nw; w++) s = red[w] < s ? red[w] : s;
  return s;
}

// ---- NCO: SampleReader::get_samples frequency shift (base/ofdm/sample_reader.cpp:274-281) -----------
// The reference steps an integer phase p -= round(f) (mod 2 048 000) per sample and multiplies by
// table[p] = (f32)cos/sin(2 pi p / 2 048 000) evaluated in double (sample_reader.cpp:44-50).
// Here the same phasor is evaluated in closed form: e^{j 2 pi ((p0 - (n+1) f) mod R) / R}, in double,
// rounded once to float -- no 16-MB table gather.  n = index of the sample within the read sequence.
struct Nco {
  double br, bi;   // phasor of this thread's first sample
  double sr, si;   // rotation per +256 samples
  __device__ void init(int phase0, int f, long long n0 /* index of this thread's first sample */)
  {
    const long long R = INPUT_RATE;
    long long p = ((long long)phase0 - (n0 + 1) * (long long)f) % R;
    if (p < 0) p += R;
    long long q = (-256LL * f) % R;
    if (q < 0) q += R;
    sincospi(2.0 * (double)p / (double)R, &bi, &br);
    sincospi(2.0 * (double)q / (double)R, &si, &sr);
  }
  // Same phasor from a block-uniform base e^{j 2 pi ((p0 - (nb+1) f) mod R)/R} (one lane evaluates it) times the
  // per-thread factor e^{-j 2 pi f tid / R} tabulated once per frame: removes two double sincospi per thread and symbol.
  __device__ void init_from(double2 base, double2 step, double2 tid_factor)
  {
    br = base.x * tid_factor.x - base.y * tid_factor.y;
    bi = base.x * tid_factor.y + base.y * tid_factor.x;
    sr = step.x; si = step.y;
  }
  __device__ static void block_consts(int phase0, int f, long long nb, double2 &base, double2 &step)
  {
    const long long R = INPUT_RATE;
    long long p = ((long long)phase0 - (nb + 1) * (long long)f) % R;
    if (p < 0) p += R;
    long long q = (-256LL * f) % R;
    if (q < 0) q += R;
    sincospi(2.0 * (double)p / (double)R, &base.y, &base.x);
    sincospi(2.0 * (double)q / (double)R, &step.y, &step.x);
  }
  __device__ static double2 tid_factor(int f, int tid)
  {
    const long long R = INPUT_RATE;
    long long p = (-(long long)tid * (long long)f) % R;
    if (p < 0) p += R;
    double2 r;
    sincospi(2.0 * (double)p / (double)R, &r.y, &r.x);
    return r;
  }
  __device__ float2 mix(float2 v) const
  {
    const float cr = (float)br, ci = (float)bi;
    return cmul(v, make_float2(cr, ci));                            // v * table[p]
  }
  __device__ void step()
  {
    const double nr = __builtin_fma(br, sr, -(bi * si)), ni = __builtin_fma(br, si, bi * sr);   // 4 DP ops instead of 6
    br = nr; bi = ni;
  }
};
__device__ __forceinline__ int nco_advance(int phase0, int f, long long n)   // phase after n samples
{
  const long long R = INPUT_RATE;
  long long p = ((long long)phase0 - n * (long long)f) % R;
  if (p < 0) p += R;
  return (int)p;
}

// ---- PRS correlator: PhaseReference::correlate_with_phase_ref_and_find_max_peak ----------------------
// (base/ofdm/phasereference.cpp:87-213).  v = T_u samples in the strided register layout of fft_core.h.
// Returns the start index (first local maximum above threshold * mean in [254, 1004)) or -1.
__device__ inline int prs_correlate_block(float2 v[8], float threshold, int strongest, const DevTables &t, float2 *lds,
                                          float *peak /* [2048] LDS */, float *red, int tid)
{
  fft2048<false>(v, lds, t.twiddle, tid);
#pragma unroll
  for (int u = 0; u < 8; u++) v[u] = cmul_conj(v[u], t.prs_ref[tid + 256 * u]);   // :97-100
  fft2048<true>(v, lds, t.twiddle, tid);
  float part = 0.f;
#pragma unroll
  for (int u = 0; u < 8; u++) { const float a = cabsf_(v[u]); peak[tid + 256 * u] = a; part += a; }   // :116-122
  float sum = block_sum(part, red, tid) / (float)TU;
  __syncthreads();
  if (sum == 0.f) return -1;
  constexpr int i0 = TG - 250, i1 = TG + 500, gap = 10;   // :136-139
  int *ired = reinterpret_cast<int *>(red);
  if (!strongest) {
    // first index whose value clears the threshold and is not exceeded within the next gap-1 samples
    int cand = 0x7fffffff;
    for (int i = i0 + tid; i < i1; i += 256) {
      const float p = peak[i];
      if (p / sum > threshold) {
        bool ok = true;
        for (int j = 1; j < gap && i + j < i1; ++j) ok = ok && !(peak[i + j] > p);
        if (ok && i < cand) cand = i;
      }
    }
    cand = block_min_int(cand, ired, tid);
    return cand == 0x7fffffff ? -1 : cand;
  }
  // strongest-peak mode keeps the reference's skip logic, which is inherently sequential (750 steps, one lane)
  __shared__ int s_res;
  if (tid == 0) {
    int max_index = -1;
    float max_l = -1000.f;
    for (int i = i0; i < i1; ++i) {
      if (peak[i] / sum > threshold) {
        bool found = true;
        for (int j = 1; j < gap && i + j < i1; ++j)
          if (peak[i + j] > peak[i]) { found = false; break; }
        if (found) {
          if (peak[i] > max_l) { max_l = peak[i]; max_index = i; }
          i += gap;
        }
      }
    }
    s_res = (max_l / sum < threshold) ? -1 : max_index;
  }
  __syncthreads();
  return s_res;
}

// ---- coarse CFO: PhaseReference::estimate_carrier_offset_from_sync_symbol_0 (:223-280) -----------------
// X = FFT of symbol 0 in the strided register layout.  Returns Hz (int) or IDX_NOT_FOUND (100000).
__device__ inline int coarse_cfo_block(const float2 X[8], const DevTables &t, float2 *lds, float *mag /* >= 160 floats LDS */,
                                       int tid)
{
  // relative phase conj(X[i]) * X[i+1]  (:282-300): neighbours live in other threads -> through LDS
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 8; u++) lds[fft_pad(tid + 256 * u)] = X[u];
  __syncthreads();
  float2 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int i = tid + 256 * u;
    if (i < TU - 1) {
      const float2 a = X[u], b = lds[fft_pad(i + 1)];
      v[u] = make_float2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
    } else v[u] = make_float2(0.f, 0.f);
  }
  __syncthreads();
  fft2048<true>(v, lds, t.twiddle, tid);
#pragma unroll
  for (int u = 0; u < 8; u++) v[u] = cmul(v[u], t.prs_arg_conj[tid + 256 * u]);
  fft2048<false>(v, lds, t.twiddle, tid);
  // |.| of bins -70..70 -> mag[0..140]
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int i = tid + 256 * u;
    if (i <= 71) mag[71 + i] = cabsf_(v[u]);            // bins 0..71 (71 only feeds the interpolation)
    if (i >= TU - 71) mag[71 + i - TU] = cabsf_(v[u]);   // bins -71..-1
  }
  __syncthreads();
  __shared__ int s_hz;
  if (tid == 0) {
    int index = 100000;
    float mx = 0.f, avg = 0.f;
    for (int i = -70; i <= 70; ++i) {
      const float val = mag[71 + i];
      if (val > mx) { mx = val; index = i; }
      avg += val;
    }
    avg /= 141.0f;
    if (mx < avg * 5) s_hz = 100000;
    else {
      const float p0 = mag[71 + index - 1], p1 = mag[71 + index], p2 = mag[71 + index + 1];
      const float psum = (0.0f + p0) + p1 + p2;
      const float offset = (float)index + (p2 - p0) / psum;
      s_hz = (int)(offset * 1000.0f);
    }
  }
  __syncthreads();
  return s_hz;
}

// ---- D-QPSK soft-bit demapper: OfdmDecoder::decode_symbol (base/ofdm/ofdm_decoder.cpp:147-355) -------
struct DemapCarrier {        // per-carrier state kept in registers across the 75 symbols of a frame
  float2 prev;               // mPhaseReference[bin]
  float integ, mean_power, mean_sigma_sq, null_power;
  float std_dev_sq;          // mStdDevSqPhaseVector[k]: the LCD record's MER only (ofdm_decoder.cpp:204-208, 331-340)
};

__device__ __forceinline__ int16_t cvt_i16_x86(float x)
{
  // (i16)(float) as x86-64 compiles it: cvttss2si ("integer indefinite" 0x80000000 when out of range / NaN),
  // then the low 16 bits (ofdm_decoder.cpp:254-255)
  if (!(fabsf(x) < 2147483648.0f)) return 0;
  return (int16_t)(uint16_t)(uint32_t)(int32_t)x;
}

// Four of them at once (the soft bits of a carrier pair).  v_cvt_i32_f32 saturates; it differs from cvttss2si's "integer
// indefinite" in the low 16 bits only for x >= 2^31 (0x7fffffff against 0x80000000; x <= -2^31 gives 0x80000000 and NaN gives
// 0 on both) -- so one v_max3 + v_max + compare guards all four and the per-value repair sits behind a branch that is
// practically never taken, instead of a compare and a select per value.
__device__ __forceinline__ int cvt_i32_sat(float x)
{
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ void cvt4_i16_x86(float a, float b, float c, float d, int16_t &ra, int16_t &rb, int16_t &rc, int16_t &rd)
{
  int ia = cvt_i32_sat(a), ib = cvt_i32_sat(b), ic = cvt_i32_sat(c), id = cvt_i32_sat(d);
  float m;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(d));
  if (__builtin_expect(m >= 2147483648.0f, 0)) {
    if (a >= 2147483648.0f) ia = 0;
    if (b >= 2147483648.0f) ib = 0;
    if (c >= 2147483648.0f) ic = 0;
    if (d >= 2147483648.0f) id = 0;
  }
  ra = (int16_t)(uint16_t)(uint32_t)ia; rb = (int16_t)(uint16_t)(uint32_t)ib;
  rc = (int16_t)(uint16_t)(uint32_t)ic; rd = (int16_t)(uint16_t)(uint32_t)id;
}

// The phase detector of the demapper (ofdm_decoder.cpp:197-202, glob_defs.h:173-182): fmod(arg(b) [+ pi if negative], pi/2)
// - pi/4, the distance of the received phase from the diagonal of its quadrant.  With r = atan(min(|x|,|y|) / max(|x|,|y|))
// in [0, pi/4] (Abramowitz-Stegun 4.4.49, |err| <= 2e-8 rad) the folded angle is r or pi/2 - r, so the result is
// -(pi/4 - r) or +(pi/4 - r): ONE magnitude and a sign = NOT[(|y| > |x|) xor (x < 0) xor (y < 0)], taken from the sign bits of
// |y| - |x|, x and y.  On an axis (min = 0, incl. b = 0 with either sign of zero) the reference's fmod returns 0, i.e. -pi/4:
// the sign is forced there.  No quadrant selects, no compares (the value only feeds the +-20 degree integrator with gain
// 1e-3: far inside the soft-bit tolerance, docs/history/r01-r04_design_notebook.md 4).
__device__ __forceinline__ float atan_octant_poly(float t)
{
  const float z = t * t;
  float p = 0.0028662257f;
  p = __builtin_fmaf(p, z, -0.0161657367f);
  p = __builtin_fmaf(p, z, 0.0429096138f);
  p = __builtin_fmaf(p, z, -0.0752896400f);
  p = __builtin_fmaf(p, z, 0.1065626393f);
  p = __builtin_fmaf(p, z, -0.1420889944f);
  p = __builtin_fmaf(p, z, 0.1999355085f);
  p = __builtin_fmaf(p, z, -0.3333314528f);
  return __builtin_fmaf(p * z, t, t);
}
__device__ __forceinline__ float phase_fold_sign(float u, float d, float x, float y, float t)
{
  // (u & 0x7fffffff) | (sign of d ^ x ^ y, forced to 1 where t == +0): two xor, an add, an or and one v_bfi_b32
  const unsigned w = (__float_as_uint(d) ^ __float_as_uint(x) ^ __float_as_uint(y)) | (__float_as_uint(t) - 1u);
  float r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0x7fffffffu), "v"(u), "v"(w));
  return r;
}
// v_max3_f32 / v_min_f32 on |x|, |y| as single instructions (fmaxf / fminf on the result of an integer `and` make the compiler
// add a canonicalising v_max x, x per operand in IEEE mode)
__device__ __forceinline__ float max3_abs(float x, float y, float floor_)
{
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(x), "v"(y), "s"(floor_));
  return r;
}
__device__ __forceinline__ float min_abs(float x, float y)
{
  float r;
  asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ float phase_offset_from_diagonal(float y, float x)
{
  // max(.., FLT_MIN): b = 0 gives t = 0 * finite = 0 instead of 0 * inf
  const float t = min_abs(x, y) * __builtin_amdgcn_rcpf(max3_abs(x, y, 1.17549435e-38f));
  const float r = atan_octant_poly(t);
  return phase_fold_sign(r - 0.78539816339744830962f, fabsf(y) - fabsf(x), x, y, t);
}

// One carrier of one symbol.  Returns |r1| (summand of mMeanValue); writes the two soft bits.
// Arithmetic follows ofdm_decoder.cpp:147-355 operation by operation; divisions and square roots use the
// hardware v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the IEEE expansions (the reference itself is built with
// -ffast-math): 2x fewer VALU instructions, soft bits differ from the IEEE evaluation by at most 1 LSB.
__device__ __forceinline__ float demap_one(DemapCarrier &c, float2 x, int rel, float clock_err, float w2,
                                           int soft_type, int16_t &soft_re, int16_t &soft_im, float &power_out)
{
  constexpr float ALPHA = 0.005f;
  const float F_PI = 3.14159265358979323846f;
  const float F_RAD_PER_DEG = 0.01745329251994329577f, F_SQRT1_2 = 0.70710678118654752440f;
  const float2 pr = c.prev;
  const float pr_sq = pr.x * pr.x + pr.y * pr.y;
  const float pr_inv = __builtin_amdgcn_rsqf(pr_sq);             // 1 / |prev| and |prev| from one v_rsq_f32
  const float pr_abs = pr_sq * pr_inv;
  float2 raw;                                                   // :188-189
  raw.x = (x.x * pr.x + x.y * pr.y) * pr_inv;
  raw.y = (x.y * pr.x - x.x * pr.y) * pr_inv;
  const float phase_err = clock_err * (F_PI / 1024.0f / (float)(K / 2)) * (float)(K / 2 - rel) + c.integ;   // :192
  const float xx = -phase_err, x2 = xx * xx;                    // cmplx_from_phase2, :70-88
  const float sine = xx * (x2 * -0.16034401953220367431640625f + 0.99903142452239990234375f);
  const float cosine = 0.9994032382965087890625f + x2 * (x2 * 3.679168224334716796875e-2f + -0.495580852031707763671875f);
  float2 b;
  b.x = raw.x * cosine - raw.y * sine;
  b.y = raw.x * sine + raw.y * cosine;
  const float off = phase_offset_from_diagonal(b.y, b.x);      // :197-201: fmod(arg(b) in [0, pi], pi/2) - pi/4
  const float lim = F_RAD_PER_DEG * 20.0f;
  c.integ = __builtin_amdgcn_fmed3f(c.integ + 0.2f * ALPHA * off, -lim, lim);   // :201-202 (limit())
  c.std_dev_sq += ALPHA * (off * off - c.std_dev_sq);           // :205-208
  const float power = b.x * b.x + b.y * b.y;                    // :211-213
  power_out = power;
  c.mean_power += ALPHA * (power - c.mean_power);
  const float mean_level = __builtin_amdgcn_sqrtf(c.mean_power);   // :217-223
  const float at_axis = mean_level * F_SQRT1_2;
  const float rd = fabsf(b.x) - at_axis, id = fabsf(b.y) - at_axis;
  const float sigma_sq = rd * rd + id * id;
  c.mean_sigma_sq += ALPHA * (sigma_sq - c.mean_sigma_sq);
  float signal_power = c.mean_power - c.null_power;             // :225-226
  if (signal_power <= 0.0f) signal_power = 0.1f;
  // r1 = b * w1 with a real weight w1 >= 0, so |r1| (:256, the summand of mMeanValue) is |b| * w1: one square root
  // serves both; the quotients of the weight share one reciprocal
  const float babs = __builtin_amdgcn_sqrtf(power);
  const float nsr = c.null_power * __builtin_amdgcn_rcpf(signal_power) + 0.7f;
  float w1;
  if (soft_type == 3) {                                         // :231-235
    w1 = pr_abs;
  } else if (soft_type == 2) {                                  // :236-242
    w1 = pr_abs * __builtin_amdgcn_rcpf(c.mean_sigma_sq * nsr);
  } else {                                                      // :243-251
    w1 = __builtin_amdgcn_sqrtf(babs * pr_abs) * mean_level * __builtin_amdgcn_rcpf(nsr * (c.mean_sigma_sq * babs));
  }
  const float2 r1 = make_float2(b.x * w1, b.y * w1);
  soft_re = cvt_i16_x86(r1.x * w2);                             // :254-255, w2 = -100 (-140) / mMeanValue
  soft_im = cvt_i16_x86(r1.y * w2);
  c.prev = x;                                                   // :354
  return babs * w1;
}

// ---- the same demapper on TWO carriers at once, as 2-vectors: every add / sub / mul of demap_one becomes one packed
// v_pk_*_f32 (IEEE, same rounding as the scalar op; -ffp-contract=off holds for vectors too), the transcendentals,
// selects and conversions stay per component.  Same operations in the same order as demap_one: identical results.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
struct DemapPair {           // state of two carriers (DemapCarrier x 2, component-wise)
  v2f prev_re, prev_im, integ, mean_power, mean_sigma_sq, null_power;
};
__device__ __forceinline__ v2f v2_rsq(v2f a) { return (v2f){__builtin_amdgcn_rsqf(a.x), __builtin_amdgcn_rsqf(a.y)}; }
__device__ __forceinline__ v2f v2_sqrt(v2f a) { return (v2f){__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)}; }
__device__ __forceinline__ v2f v2_rcp(v2f a) { return (v2f){__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ v2f v2_abs(v2f a) { return (v2f){fabsf(a.x), fabsf(a.y)}; }
__device__ __forceinline__ v2f v2_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ v2f phase_offset_from_diagonal2(v2f y, v2f x, v2f ay, v2f ax)   // ay = |y|, ax = |x| (the caller has them)
{
  const v2f mx = (v2f){max3_abs(x.x, y.x, 1.17549435e-38f), max3_abs(x.y, y.y, 1.17549435e-38f)};
  const v2f mnv = (v2f){min_abs(x.x, y.x), min_abs(x.y, y.y)};
  const v2f t = mnv * v2_rcp(mx);
  const v2f z = t * t;
  v2f p = (v2f)(0.0028662257f);
  p = v2_fma(p, z, (v2f)(-0.0161657367f));
  p = v2_fma(p, z, (v2f)(0.0429096138f));
  p = v2_fma(p, z, (v2f)(-0.0752896400f));
  p = v2_fma(p, z, (v2f)(0.1065626393f));
  p = v2_fma(p, z, (v2f)(-0.1420889944f));
  p = v2_fma(p, z, (v2f)(0.1999355085f));
  p = v2_fma(p, z, (v2f)(-0.3333314528f));
  const v2f r = v2_fma(p * z, t, t);
  const v2f u = r - 0.78539816339744830962f, d = ay - ax;
  return (v2f){phase_fold_sign(u.x, d.x, x.x, y.x, t.x), phase_fold_sign(u.y, d.y, x.y, y.y, t.y)};
}

template <int SOFT_TYPE>
__device__ __forceinline__ v2f demap_pair(DemapPair &c, v2f x_re, v2f x_im, v2f rel_f, float clock_err, float w2,
                                          int16_t (&soft_re)[2], int16_t (&soft_im)[2], v2f &power_out, v2f *std_dev_sq = nullptr)
{
  constexpr float ALPHA = 0.005f;
  const float F_PI = 3.14159265358979323846f;
  const float F_RAD_PER_DEG = 0.01745329251994329577f, F_SQRT1_2 = 0.70710678118654752440f;
  const v2f pr_re = c.prev_re, pr_im = c.prev_im;
  const v2f pr_sq = pr_re * pr_re + pr_im * pr_im;
  const v2f pr_inv = v2_rsq(pr_sq);
  const v2f pr_abs = pr_sq * pr_inv;
  const v2f raw_re = (x_re * pr_re + x_im * pr_im) * pr_inv;      // :188-189
  const v2f raw_im = (x_im * pr_re - x_re * pr_im) * pr_inv;
  const v2f phase_err = clock_err * (F_PI / 1024.0f / (float)(K / 2)) * rel_f + c.integ;   // :192, rel_f = (float)(K/2 - rel)
  const v2f xx = -phase_err, x2 = xx * xx;                        // cmplx_from_phase2, :70-88
  const v2f sine = xx * (x2 * -0.16034401953220367431640625f + 0.99903142452239990234375f);
  const v2f cosine = 0.9994032382965087890625f + x2 * (x2 * 3.679168224334716796875e-2f + -0.495580852031707763671875f);
  const v2f b_re = raw_re * cosine - raw_im * sine;
  const v2f b_im = raw_re * sine + raw_im * cosine;
  const v2f abs_re = v2_abs(b_re), abs_im = v2_abs(b_im);
  const v2f off = phase_offset_from_diagonal2(b_im, b_re, abs_im, abs_re);   // :197-201: fmod(arg(b) in [0, pi], pi/2) - pi/4
  const v2f integ = c.integ + 0.2f * ALPHA * off;                 // :201-202
  const float lim = F_RAD_PER_DEG * 20.0f;
  c.integ = (v2f){__builtin_amdgcn_fmed3f(integ.x, -lim, lim), __builtin_amdgcn_fmed3f(integ.y, -lim, lim)};
  // :205-208 mStdDevSqPhaseVector -- the LCD record's MER, no soft bit: block-uniformly on or off, its state parked in LDS (the engine's
  // demapper has no two registers to spare at six waves per SIMD)
  if (std_dev_sq) { v2f sd = *std_dev_sq; sd += ALPHA * (off * off - sd); *std_dev_sq = sd; }
  const v2f power = b_re * b_re + b_im * b_im;                    // :211-213
  power_out = power;
  c.mean_power += ALPHA * (power - c.mean_power);
  const v2f mean_level = v2_sqrt(c.mean_power);                   // :217-223
  const v2f at_axis = mean_level * F_SQRT1_2;
  const v2f rd = abs_re - at_axis, id = abs_im - at_axis;
  const v2f sigma_sq = rd * rd + id * id;
  c.mean_sigma_sq += ALPHA * (sigma_sq - c.mean_sigma_sq);
  v2f signal_power = c.mean_power - c.null_power;                 // :225-226
  signal_power = (signal_power <= 0.0f) ? (v2f)(0.1f) : signal_power;
  const v2f babs = v2_sqrt(power);
  const v2f nsr = c.null_power * v2_rcp(signal_power) + 0.7f;
  v2f w1;
  if (SOFT_TYPE == 3) w1 = pr_abs;                                // :231-235
  else if (SOFT_TYPE == 2) w1 = pr_abs * v2_rcp(c.mean_sigma_sq * nsr);   // :236-242
  else w1 = v2_sqrt(babs * pr_abs) * mean_level * v2_rcp(nsr * (c.mean_sigma_sq * babs));   // :243-251
  const v2f r1_re = b_re * w1, r1_im = b_im * w1;
  const v2f s_re = r1_re * w2, s_im = r1_im * w2;                 // :254-255
  cvt4_i16_x86(s_re.x, s_re.y, s_im.x, s_im.y, soft_re[0], soft_re[1], soft_im[0], soft_im[1]);
  c.prev_re = x_re; c.prev_im = x_im;                             // :354
  return babs * w1;
}

__device__ __forceinline__ float demap_w2(float mean_value, int soft_type)
{
  return (soft_type == 1 ? -100.0f : -140.0f) * __builtin_amdgcn_rcpf(mean_value);
}

// ---- mMeanPowerOvrAll and the SNR estimate (ofdm_decoder.cpp:214, 326-343, 358-371): display statistics only --------
// The reference runs x += a (p_k - x), a = 0.005 / 1536, over the 1536 carriers of a symbol in de-interleaved order.  In
// closed form one symbol is x <- x (1 - a)^K + sum_k a (1 - a)^(K-1-k) p_k: a weighted block sum, carriers in parallel.
// (The serial float recurrence itself wanders by ~1e-6 relative; the estimate is compared with a 0.02 dB tolerance.)
__device__ __forceinline__ float mpa_weight(int k)
{
  const float a = 0.005f / (float)K;
  return a * __expf((float)(K - 1 - k) * -3.2552135e-6f);        // ln(1 - a) = -a - a^2 / 2
}
__device__ __forceinline__ float mpa_decay() { return 0.99501247f; }                     // (1 - a)^K
__device__ __forceinline__ float mpa_decay_n(int n_sym) { return __expf((float)n_sym * (float)K * -3.2552135e-6f); }   // (1 - a)^(K n)
// snr = (mMeanPowerOvrAll - noise) / noise, <= 0 -> 0.1; noise = mean null power over the used bins, all-zero -> floor
__device__ __forceinline__ float snr_db_from(float mean_power_all, float null_sum)
{
  const float kMinNoisePower = (1.0f / 32767.0f) * (1.0f / 32767.0f);
  if (null_sum == 0.0f) null_sum = kMinNoisePower * (float)K;
  const float noise = null_sum / (float)K;
  float snr = (mean_power_all - noise) / noise;
  if (snr <= 0.0f) snr = 0.1f;
  return 10.0f * log10f(snr);
}

// MER of the LCD record (ofdm_decoder.cpp:331-340): 10 log10((pi/4)^2 / mean_k mStdDevSqPhaseVector[k])
__device__ __forceinline__ float mer_db_from(float std_dev_sq_sum)
{
  const float F_PI_4 = 0.78539816339744830962f;
  return 10.0f * log10f(F_PI_4 * F_PI_4 / (std_dev_sq_sum / (float)K));
}

__device__ __forceinline__ uint8_t soft_to_sym(int16_t s)       // viterbi_scalar.h:34-40
{
  int v = (int16_t)(s + 127);
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  return (uint8_t)v;
}

// viterbi_16way.h:73-76 (the AVX2 build): _mm_adds_epi16(+127) SATURATES where the scalar body's i16 add wraps
__device__ __forceinline__ uint8_t soft_to_sym_sat(int16_t s)
{
  int v = (int)s + 127;
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  return (uint8_t)v;
}
__device__ __forceinline__ uint8_t soft_to_sym_mode(int16_t s, int tie_mode) { return tie_mode ? soft_to_sym_sat(s) : soft_to_sym(s); }

}  // namespace dabx
