/* ofdm.c -- OFDM front end: DFT, PRS correlator, coarse CFO, D-QPSK soft-bit demapper
 * (oracle; test infrastructure only; PARITY UNPINNED for this file, see dab_oracle.h).
 * Float semantics: the reference is built with -ffast-math, so std::abs(cf32) is
 * sqrt(re^2+im^2) and complex products are the plain 4-multiply form; that is what is
 * written out here.  libm atan2f/cosf/sinf/fmodf stand for std::arg/cos/sin/fmod. */
#include "dab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ DFT */
/* FFTW3f (third party, un-vendored; fftwf_plan_dft_1d(2048, dir, FFTW_ESTIMATE) at
 * main/dab_processor.cpp:63, ofdm/phasereference.cpp:51-52) computes the unnormalised
 * DFT  X[k] = sum x[n] e^{-/+ j 2 pi n k / N}.  Restated as a radix-2 transform carried
 * out in double precision and rounded once to float. */
static double g_tw_re[ORA_TU / 2], g_tw_im[ORA_TU / 2];
static int g_tw_ready = 0;

void ora_fft2048(const ora_cf32 *in, ora_cf32 *out, int inverse)
{
  enum { N = ORA_TU, LOGN = 11 };
  if (!g_tw_ready) {
    for (int i = 0; i < N / 2; i++) { g_tw_re[i] = cos(2.0 * M_PI * i / N); g_tw_im[i] = -sin(2.0 * M_PI * i / N); }
    g_tw_ready = 1;
  }
  double re[ORA_TU], im[ORA_TU];          /* on the stack: receivers may run on several threads (bench cpu_baseline) */
  for (int i = 0; i < N; i++) {
    unsigned r = 0;
    for (int b = 0; b < LOGN; b++) r |= ((i >> b) & 1u) << (LOGN - 1 - b);
    re[r] = in[i].re; im[r] = in[i].im;
  }
  for (int len = 2; len <= N; len <<= 1) {
    const int half = len / 2, step = N / len;
    for (int s = 0; s < N; s += len)
      for (int k = 0; k < half; k++) {
        const double wr = g_tw_re[k * step], wi = inverse ? -g_tw_im[k * step] : g_tw_im[k * step];
        const double xr = re[s + k + half], xi = im[s + k + half];
        const double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
        re[s + k + half] = re[s + k] - tr; im[s + k + half] = im[s + k] - ti;
        re[s + k] += tr; im[s + k] += ti;
      }
  }
  for (int i = 0; i < N; i++) { out[i].re = (float)re[i]; out[i].im = (float)im[i]; }
}

static inline float cabs_f(ora_cf32 z) { return sqrtf(z.re * z.re + z.im * z.im); }

/* (i16)(float) as compiled for x86-64: cvttss2si to 32 bits ("integer indefinite"
 * 0x80000000 when out of range or NaN), then the low 16 bits (ofdm_decoder.cpp:254-255). */
static inline int16_t cvt_i16(float x)
{
  if (!(fabsf(x) < 2147483648.0f)) return 0;
  return (int16_t)(uint16_t)(uint32_t)(int32_t)x;
}

/* ------------------------------------------------------------------ PRS correlator */
/* phasereference.cpp:282-300 */
static void relative_phase(ora_cf32 *o, const ora_cf32 *f)
{
  for (int i = 0; i < ORA_TU - 1; i++) {   /* conj(f[i]) * f[i+1] */
    o[i].re = f[i].re * f[i + 1].re + f[i].im * f[i + 1].im;
    o[i].im = f[i].re * f[i + 1].im - f[i].im * f[i + 1].re;
  }
  o[ORA_TU - 1].re = 0; o[ORA_TU - 1].im = 0;
}

/* phasereference.cpp:47-69 */
void ora_phaseref_init(ora_phaseref *p)
{
  ora_cf32 tmp[ORA_TU], t2[ORA_TU];
  ora_phase_table(p->ref);
  relative_phase(tmp, p->ref);
  ora_fft2048(tmp, t2, 1);
  for (int i = 0; i < ORA_TU; i++) { p->ref_arg_conj[i].re = t2[i].re; p->ref_arg_conj[i].im = -t2[i].im; }
  p->strongest = 0;
}

/* phasereference.cpp:87-213 */
int ora_phaseref_correlate(ora_phaseref *p, const ora_cf32 *v, float threshold)
{
  ora_cf32 a[ORA_TU], b[ORA_TU];
  float peak[ORA_TU];
  ora_fft2048(v, a, 0);
  for (int i = 0; i < ORA_TU; i++) {       /* X * conj(ref), :97-100 */
    b[i].re = a[i].re * p->ref[i].re + a[i].im * p->ref[i].im;
    b[i].im = a[i].im * p->ref[i].re - a[i].re * p->ref[i].im;
  }
  ora_fft2048(b, a, 1);
  float sum = 0;
  for (int i = 0; i < ORA_TU; i++) { peak[i] = cabs_f(a[i]); sum += peak[i]; }   /* :116-122 */
  sum /= (float)ORA_TU;
  if (sum == 0) return -1;
  int max_index = -1, first = -1;
  float max_l = -1000;
  const int gap = 10, i0 = ORA_TG - 250, i1 = ORA_TG + 500;                     /* :136-139 */
  for (int i = i0; i < i1; ++i) {
    if (peak[i] / sum > threshold) {
      int found = 1;
      for (int j = 1; j < gap && i + j < i1; ++j)
        if (peak[i + j] > peak[i]) { found = 0; break; }
      if (found) {
        if (first < 0) first = i;
        if (peak[i] > max_l) { max_l = peak[i]; max_index = i; }
        i += gap;
      }
    }
  }
  if (max_l / sum < threshold) return -1;                                       /* :173-176 */
  return p->strongest ? max_index : first;                                      /* :203-212 */
}

/* phasereference.cpp:223-280 */
int ora_phaseref_coarse_cfo(ora_phaseref *p, const ora_cf32 *fft_sym0)
{
  ora_cf32 a[ORA_TU], b[ORA_TU];
  int index = ORA_IDX_NOT_FOUND;
  float max = 0, avg = 0;
  relative_phase(a, fft_sym0);
  ora_fft2048(a, b, 1);
  for (int i = 0; i < ORA_TU; i++) {       /* b * refArgConj */
    a[i].re = b[i].re * p->ref_arg_conj[i].re - b[i].im * p->ref_arg_conj[i].im;
    a[i].im = b[i].re * p->ref_arg_conj[i].im + b[i].im * p->ref_arg_conj[i].re;
  }
  ora_fft2048(a, b, 0);
  const int range = 140;                   /* phasereference.h:61 */
  for (int i = -range / 2; i <= range / 2; ++i) {
    const float v = cabs_f(b[(ORA_TU + i) % ORA_TU]);
    if (v > max) { max = v; index = i; }
    avg += v;
  }
  avg /= (float)(range + 1);
  if (max < avg * 5) return ORA_IDX_NOT_FOUND;
  float pk[3], pk_sum = 0.0f;
  for (int i = 0; i < 3; ++i) { pk[i] = cabs_f(b[(ORA_TU + index + i - 1) % ORA_TU]); pk_sum += pk[i]; }
  const float offset = (float)index + (pk[2] - pk[0]) / pk_sum;
  return (int32_t)(offset * 1000.0f);
}

/* ------------------------------------------------------------------ demapper */
static const float kMinNoisePower = (1.0f / 32767.0f) * (1.0f / 32767.0f);   /* ofdm_decoder.cpp:40-41 */
static const float F_PI = (float)M_PI, F_PI_4 = (float)(M_PI / 4.0), F_PI_2 = (float)(M_PI / 2.0);
static const float F_RAD_PER_DEG = (float)(M_PI / 180.0);
static const float F_SQRT1_2 = 0.70710678118654752440084436210485f;

void ora_demap_reset(ora_demap *d)         /* ofdm_decoder.cpp:90-101 */
{
  memset(d->std_dev_sq, 0, sizeof(d->std_dev_sq));
  memset(d->integ_abs_phase, 0, sizeof(d->integ_abs_phase));
  memset(d->mean_power, 0, sizeof(d->mean_power));
  memset(d->mean_sigma_sq, 0, sizeof(d->mean_sigma_sq));
  memset(d->mean_null_power, 0, sizeof(d->mean_null_power));
  d->mean_power_ovr_all = 1.0f;
}

void ora_demap_init(ora_demap *d)          /* ofdm_decoder.cpp:43-66, ofdm_decoder.h:101-104 */
{
  memset(d, 0, sizeof(*d));
  ora_freq_interleaver(d->perm);
  d->mean_value = 1.0f;
  d->soft_bit_type = 1;
  ora_demap_reset(d);
}

void ora_demap_store_ref(ora_demap *d, const ora_cf32 *fft) { memcpy(d->phase_ref, fft, sizeof(d->phase_ref)); }

/* ofdm_decoder.cpp:114-130 ; fft_shift_skip_dc: idx<0 -> idx+Tu, idx>=0 -> idx+1 */
void ora_demap_store_null(ora_demap *d, const ora_cf32 *fft)
{
  for (int idx = -ORA_K / 2; idx < ORA_K / 2; ++idx) {
    const int bin = idx < 0 ? idx + ORA_TU : idx + 1;
    const float power = fft[bin].re * fft[bin].re + fft[bin].im * fft[bin].im + kMinNoisePower;
    d->mean_null_power[bin] += 0.05f * (power - d->mean_null_power[bin]);
  }
}

/* ofdm_decoder.cpp:147-355 (soft-bit relevant part; display paths omitted) */
void ora_demap_symbol(ora_demap *d, const ora_cf32 *fft, float clock_err, int16_t out[ORA_2K])
{
  float sum = 0.0f;
  const float ALPHA = 0.005f;
  for (int k = 0; k < ORA_K; ++k) {
    int bin = d->perm[k];
    int rel = bin;
    if (bin < 0) { rel += ORA_K / 2; bin += ORA_TU; } else rel += ORA_K / 2 - 1;     /* :171-179 */

    const ora_cf32 pr = d->phase_ref[bin], x = fft[bin];
    const float pr_abs = cabs_f(pr);
    ora_cf32 raw;                                      /* x * conj(pr) / |pr| , :188-189 */
    raw.re = (x.re * pr.re + x.im * pr.im) / pr_abs;
    raw.im = (x.im * pr.re - x.re * pr.im) / pr_abs;

    float *integ = &d->integ_abs_phase[k];
    const float phase_err = clock_err / 1024.0f * F_PI * (float)(ORA_K / 2 - rel) / (float)(ORA_K / 2) + *integ; /* :192 */

    /* cmplx_from_phase2(-phase_err), :70-88 */
    const float xx = -phase_err, x2 = xx * xx;
    const float sine = xx * (x2 * -0.16034401953220367431640625f + 0.99903142452239990234375f);
    const float cosine = 0.9994032382965087890625f + x2 * (x2 * 3.679168224334716796875e-2f + -0.495580852031707763671875f);
    ora_cf32 b;
    b.re = raw.re * cosine - raw.im * sine;
    b.im = raw.re * sine + raw.im * cosine;

    float ph = atan2f(b.im, b.re);                     /* :197 */
    if (ph < 0.0f) ph += F_PI;                         /* glob_defs.h:173-182 */
    const float aph = fmodf(ph, F_PI_2);

    *integ += 0.2f * ALPHA * (aph - F_PI_4);           /* :201-202 */
    if (*integ > F_RAD_PER_DEG * 20.0f) *integ = F_RAD_PER_DEG * 20.0f;
    else if (*integ < -(F_RAD_PER_DEG * 20.0f)) *integ = -(F_RAD_PER_DEG * 20.0f);

    const float sdd = aph - F_PI_4;                    /* :205-208 */
    d->std_dev_sq[k] += ALPHA * (sdd * sdd - d->std_dev_sq[k]);

    const float power = b.re * b.re + b.im * b.im;     /* :211-214 */
    d->mean_power[k] += ALPHA * (power - d->mean_power[k]);
    d->mean_power_ovr_all += (ALPHA / (float)ORA_K) * (power - d->mean_power_ovr_all);

    const float mean_level = sqrtf(d->mean_power[k]);  /* :217-223 */
    const float at_axis = mean_level * F_SQRT1_2;
    const float rd = fabsf(b.re) - at_axis, id = fabsf(b.im) - at_axis;
    const float sigma_sq = rd * rd + id * id;
    d->mean_sigma_sq[k] += ALPHA * (sigma_sq - d->mean_sigma_sq[k]);

    float signal_power = d->mean_power[k] - d->mean_null_power[bin];   /* :225-226 */
    if (signal_power <= 0.0f) signal_power = 0.1f;

    ora_cf32 r1; float w2;
    if (d->soft_bit_type == 3) {                       /* :231-235 */
      r1.re = b.re * pr_abs; r1.im = b.im * pr_abs; w2 = -140 / d->mean_value;
    } else if (d->soft_bit_type == 2) {                /* :236-242 */
      float w1 = pr_abs / d->mean_sigma_sq[k];
      w1 /= (d->mean_null_power[bin] / signal_power) + 0.7f;
      r1.re = b.re * w1; r1.im = b.im * w1; w2 = -140 / d->mean_value;
    } else {                                           /* :243-251 */
      const float babs = sqrtf(power);
      float w1 = sqrtf(babs * pr_abs) * mean_level;
      w1 /= (d->mean_null_power[bin] / signal_power) + 0.7f;
      w1 /= d->mean_sigma_sq[k] * babs;
      r1.re = b.re * w1; r1.im = b.im * w1; w2 = -100 / d->mean_value;
    }
    /* :254-255 */
    out[k] = cvt_i16(r1.re * w2);
    out[ORA_K + k] = cvt_i16(r1.im * w2);
    if (!(fabsf(r1.re * w2) < 32768.0f)) d->overflow_count++;
    if (!(fabsf(r1.im * w2) < 32768.0f)) d->overflow_count++;
    sum += cabs_f(r1);                                 /* :256 */
  }
  d->mean_value = sum / (float)ORA_K;                  /* :294 */
  memcpy(d->phase_ref, fft, sizeof(d->phase_ref));     /* :354 */
}

/* ofdm_decoder.cpp:326-343 (LCD statistics) with _compute_noise_Power, :358-371 */
float ora_demap_snr_db(const ora_demap *d)
{
  float sum_noise = 0.0f;
  for (int idx = -ORA_K / 2; idx < ORA_K / 2; ++idx) sum_noise += d->mean_null_power[idx < 0 ? idx + ORA_TU : idx + 1];
  if (sum_noise == 0.0f) sum_noise = kMinNoisePower * ORA_K;
  const float noise = sum_noise / (float)ORA_K;
  float snr = (d->mean_power_ovr_all - noise) / noise;
  if (snr <= 0.0f) snr = 0.1f;
  return 10.0f * log10f(snr);
}

/* ofdm_decoder.cpp:331-340 (LCD statistics): MER from the per-carrier phase-deviation IIR of :204-208 */
float ora_demap_mer_db(const ora_demap *d)
{
  float std_dev_sq_ovr_all = 0.0f;
  for (int idx = 0; idx < ORA_K; idx++) std_dev_sq_ovr_all += d->std_dev_sq[idx];
  std_dev_sq_ovr_all /= (float)ORA_K;
  return 10.0f * log10f(F_PI_4 * F_PI_4 / std_dev_sq_ovr_all);
}

/* heap helpers for ctypes-based tests */
ora_demap *ora_demap_new(void) { ora_demap *d = (ora_demap *)malloc(sizeof(ora_demap)); ora_demap_init(d); return d; }
void ora_demap_free(ora_demap *d) { free(d); }
ora_phaseref *ora_phaseref_new(void) { ora_phaseref *p = (ora_phaseref *)malloc(sizeof(ora_phaseref)); ora_phaseref_init(p); return p; }
void ora_phaseref_free(ora_phaseref *p) { free(p); }
void ora_phaseref_set_strongest(ora_phaseref *p, int on) { p->strongest = on; }
void ora_demap_set_type(ora_demap *d, int type) { d->soft_bit_type = type; }
float ora_demap_mean_value(const ora_demap *d) { return d->mean_value; }                     /* mMeanValue: SLcdData::TestData1, :344 */
const float *ora_demap_std_dev_sq(const ora_demap *d) { return d->std_dev_sq; }             /* mStdDevSqPhaseVector (tests) */
