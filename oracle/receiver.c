/* receiver.c -- sample reader + null-dip detector + DabProcessor state machine over an
 * in-memory IQ buffer (oracle; test infrastructure only; PARITY UNPINNED, see dab_oracle.h).
 * Restates ofdm/sample_reader.cpp:102-297 (scalar build), ofdm/timesyncer.cpp:40-90 and
 * main/dab_processor.cpp:110-442 for file-player input (no settle discard, :118-121 of
 * sample_reader.cpp; no real-time pacing). */
#include "dab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct ora_receiver {
  /* input */
  const ora_cf32 *iq; size_t n_iq, pos; int eof;
  /* SampleReader state (sample_reader.h:91-108) */
  int32_t cur_phase; float s_level, peak_level;
  int dc_iq_mode;                    /* 0 off, 1 mDoDcOrIqCorr, 2 + mDoIqCorr (set_dc_and_iq_correction, sample_reader.cpp:334-346) */
  float mean_i, mean_q, mean_ii, mean_qq, mean_iq;
  /* DabProcessor state (dab_processor.h:129-138) */
  float phase_offs_cp, freq_offs_sync, freq_offs_bb, clock_err;
  float threshold; int sync_strongest;
  ora_phaseref pr; ora_demap dm; ora_fic fic;
  int n_back; ora_backend *back;
  long cifs_done;                    /* CIFs handed to the back ends so far */
  long *move_cif; int *move_to;      /* per back end: from CIF move_cif on its capacity units start at move_to (ora_rx_move_subch) */
  int16_t cif[ORA_CIF_BITS];
  ora_cf32 buf[ORA_TN];
  int16_t bits[ORA_2K];
  /* TiiDetector::mNullSymbolBufferVec + DabProcessor::mTiiCounter (dab_processor.cpp:287-299, never processed here) */
  ora_cf32 tii_acc[ORA_TU]; int tii_count;
  /* capture */
  ora_rx_capture cap; int cap_alloc; int want_soft;
};

static ora_cf32 *g_osc = NULL;   /* sample_reader.cpp:44-50 : 2 048 000-entry oscillator table */
static void build_osc(void)
{
  g_osc = (ora_cf32 *)malloc(sizeof(ora_cf32) * ORA_INPUT_RATE);
  for (int i = 0; i < ORA_INPUT_RATE; i++) {
    g_osc[i].re = (float)cos(2.0 * M_PI * i / ORA_INPUT_RATE);
    g_osc[i].im = (float)sin(2.0 * M_PI * i / ORA_INPUT_RATE);
  }
}

/* sample_reader.cpp:218-243: DC removal and IQ-imbalance correction of one sample; st = {meanI, meanQ, meanII, meanQQ, meanIQ} */
void ora_dciq_sample(ora_cf32 *v, int mode, float *st)
{
  const float ALPHA = 1.0f / (float)ORA_INPUT_RATE / 1.00f;
  const float v_i = v->re, v_q = v->im;
  st[0] += ALPHA * (v_i - st[0]);
  st[1] += ALPHA * (v_q - st[1]);
  if (mode == 2) {
    const float x_i = v_i - st[0], x_q = v_q - st[1];
    st[2] += ALPHA * (x_i * x_i - st[2]);
    st[4] += ALPHA * (x_i * x_q - st[4]);
    const float phi = st[4] / st[2];
    const float x_q_corr = x_q - phi * x_i;
    st[3] += ALPHA * (x_q_corr * x_q_corr - st[3]);
    const float gain_q = sqrtf(st[2] / st[3]);
    v->re = x_i; v->im = x_q_corr * gain_q;
  } else {
    v->re = v_i - st[0]; v->im = v_q - st[1];
  }
}
/* the same over a buffer (test helper): state in / out as above */
void ora_dciq_buffer(ora_cf32 *iq, size_t n, int mode, float *st)
{
  for (size_t i = 0; i < n; i++) ora_dciq_sample(&iq[i], mode, st);
}

/* the same filters with double-precision states (test helper: what the recurrence converges to without float rounding
 * noise -- with ALPHA = 4.9e-7 every float update of a mean near 1 rounds away up to 6 % of its increment) */
void ora_dciq_buffer_f64(ora_cf32 *iq, size_t n, int mode, double *st)
{
  const double ALPHA = (double)(1.0f / (float)ORA_INPUT_RATE);
  for (size_t i = 0; i < n; i++) {
    const double v_i = iq[i].re, v_q = iq[i].im;
    st[0] += ALPHA * (v_i - st[0]);
    st[1] += ALPHA * (v_q - st[1]);
    const double x_i = v_i - st[0], x_q = v_q - st[1];
    if (mode == 2) {
      st[2] += ALPHA * (x_i * x_i - st[2]);
      st[4] += ALPHA * (x_i * x_q - st[4]);
      const double x_q_corr = x_q - st[4] / st[2] * x_i;
      st[3] += ALPHA * (x_q_corr * x_q_corr - st[3]);
      iq[i].re = (float)x_i; iq[i].im = (float)(x_q_corr * sqrt(st[2] / st[3]));
    } else { iq[i].re = (float)x_i; iq[i].im = (float)x_q; }
  }
}

/* sample_reader.cpp:102-297, scalar branch :212-283 (DC/IQ correction off by default: configuration.cpp:75-76) */
static int get_samples(ora_receiver *r, ora_cf32 *dst, int n, float freq_bb)
{
  if (r->pos + (size_t)n > r->n_iq) { r->eof = 1; return 0; }   /* :108-113 -> throw 20 */
  const int32_t f = (int32_t)roundf(freq_bb);                  /* :211 std::round */
  const ora_cf32 *src = r->iq + r->pos;
  for (int i = 0; i < n; i++) {
    ora_cf32 v = src[i];
    if (r->dc_iq_mode) ora_dciq_sample(&v, r->dc_iq_mode, &r->mean_i);   /* :218-243 */
    const float a = sqrtf(v.re * v.re + v.im * v.im);          /* :245-248 */
    if (a > r->peak_level) r->peak_level = a;
    r->s_level += 0.00001f * (a - r->s_level);
    r->cur_phase -= f;                                          /* :274-281 */
    r->cur_phase = (r->cur_phase + ORA_INPUT_RATE) % ORA_INPUT_RATE;
    const ora_cf32 o = g_osc[r->cur_phase];
    dst[i].re = v.re * o.re - v.im * o.im;
    dst[i].im = v.re * o.im + v.im * o.re;
  }
  r->pos += (size_t)n;
  return n;
}

/* sample_reader.cpp:245-248 alone: the level after n samples read in order, from s0 (the operations of get_samples above) */
float ora_level_walk(const ora_cf32 *x, size_t n, float s0)
{
  float s = s0;
  for (size_t i = 0; i < n; i++) {
    const float a = sqrtf(x[i].re * x[i].re + x[i].im * x[i].im);
    s += 0.00001f * (a - s);
  }
  return s;
}

/* timesyncer.cpp:40-90 ; returns 1 established, 0 otherwise, -1 eof */
static int time_sync(ora_receiver *r)
{
  enum { SEARCH = 50, BUFSZ = 4096, MASK = BUFSZ - 1 };
  float env[BUFSZ], level = 0;
  int idx = 0;
  ora_cf32 s;
  for (int i = 0; i < SEARCH; i++) {
    if (!get_samples(r, &s, 1, 0)) return -1;
    env[idx] = sqrtf(s.re * s.re + s.im * s.im);
    level += env[idx];
    ++idx;
  }
  int counter = 0;
  while (level / SEARCH > 0.55f * r->s_level) {
    if (!get_samples(r, &s, 1, 0)) return -1;
    env[idx] = sqrtf(s.re * s.re + s.im * s.im);
    level += env[idx] - env[(unsigned)(idx - SEARCH) & MASK];
    idx = (idx + 1) & MASK;
    if (++counter > ORA_TF) return 0;
  }
  counter = 0;
  while (level / SEARCH < 0.75f * r->s_level) {
    if (!get_samples(r, &s, 1, 0)) return -1;
    env[idx] = sqrtf(s.re * s.re + s.im * s.im);
    level += env[idx] - env[(unsigned)(idx - SEARCH) & MASK];
    idx = (idx + 1) & MASK;
    if (++counter > ORA_TN + SEARCH + 20) return 0;
  }
  return 1;
}

ora_receiver *ora_rx_create(const ora_subch_desc *subch, int n_subch)
{
  if (!g_osc) build_osc();
  ora_receiver *r = (ora_receiver *)calloc(1, sizeof(*r));
  r->s_level = 0.1f; r->peak_level = -1.0e6f;
  r->mean_ii = 1.0f; r->mean_qq = 1.0f;        /* sample_reader.h:102-106 */
  r->threshold = 3.0f;                         /* main/dabradio.cpp:92 */
  ora_phaseref_init(&r->pr);
  ora_demap_init(&r->dm);
  ora_fic_init(&r->fic);
  r->n_back = n_subch;
  r->back = (ora_backend *)calloc((size_t)(n_subch > 0 ? n_subch : 1), sizeof(ora_backend));
  for (int i = 0; i < n_subch; i++)
    if (ora_backend_init(&r->back[i], &subch[i]) != 0) { ora_rx_destroy(r); return NULL; }
  return r;
}

/* ProcessParams::threshold (dabradio.cpp:92), sync on strongest peak (configuration.cpp:65), soft-bit type (glob_enums.h:49-56) */
void ora_rx_configure(ora_receiver *r, float threshold, int sync_strongest, int soft_bit_type)
{
  r->threshold = threshold;
  r->sync_strongest = sync_strongest;
  ora_phaseref_set_strongest(&r->pr, sync_strongest);
  ora_demap_set_type(&r->dm, soft_bit_type);
}

void ora_rx_set_dc_iq(ora_receiver *r, int mode) { r->dc_iq_mode = mode; }

void ora_rx_destroy(ora_receiver *r)
{
  if (!r) return;
  for (int i = 0; i < r->n_back; i++) ora_backend_free(&r->back[i]);
  free(r->move_cif); free(r->move_to);
  free(r->back);
  free(r->cap.fibs); free(r->cap.fib_crc); free(r->cap.soft); free(r->cap.start_idx); free(r->cap.fbb); free(r->cap.sym0_pos);
  free(r->cap.fbb_end); free(r->cap.clock_err); free(r->cap.fic_ratio); free(r->cap.snr_db); free(r->cap.fic_overflow); free(r->cap.msc_overflow);
  free(r->cap.s_level); free(r->cap.peak_level); free(r->cap.fic_ber_bits); free(r->cap.fic_ber_errors);
  free(r->cap.mer_db);
  free(r);
}

void ora_rx_enable_soft_capture(ora_receiver *r, int on) { r->want_soft = on; }
const ora_rx_capture *ora_rx_get_capture(ora_receiver *r) { return &r->cap; }
ora_backend *ora_rx_backend(ora_receiver *r, int i) { return (i >= 0 && i < r->n_back) ? &r->back[i] : NULL; }
ora_fic *ora_rx_fic(ora_receiver *r) { return &r->fic; }

static void cap_reserve(ora_receiver *r, int n)
{
  if (n <= r->cap_alloc) return;
  int na = r->cap_alloc ? r->cap_alloc * 2 : 16;
  while (na < n) na *= 2;
  r->cap.fibs = (uint8_t *)realloc(r->cap.fibs, (size_t)na * 12 * 32);
  r->cap.fib_crc = (uint8_t *)realloc(r->cap.fib_crc, (size_t)na * 12);
  r->cap.start_idx = (int32_t *)realloc(r->cap.start_idx, sizeof(int32_t) * (size_t)na);
  r->cap.fbb = (float *)realloc(r->cap.fbb, sizeof(float) * (size_t)na);
  r->cap.sym0_pos = (int32_t *)realloc(r->cap.sym0_pos, sizeof(int32_t) * (size_t)na);
  r->cap.fbb_end = (float *)realloc(r->cap.fbb_end, sizeof(float) * (size_t)na);
  r->cap.clock_err = (float *)realloc(r->cap.clock_err, sizeof(float) * (size_t)na);
  r->cap.fic_ratio = (int32_t *)realloc(r->cap.fic_ratio, sizeof(int32_t) * (size_t)na);
  r->cap.snr_db = (float *)realloc(r->cap.snr_db, sizeof(float) * (size_t)na);
  r->cap.mer_db = (float *)realloc(r->cap.mer_db, sizeof(float) * (size_t)na);
  r->cap.fic_overflow = (int32_t *)realloc(r->cap.fic_overflow, sizeof(int32_t) * (size_t)na);
  r->cap.msc_overflow = (int32_t *)realloc(r->cap.msc_overflow, sizeof(int32_t) * (size_t)na);
  r->cap.s_level = (float *)realloc(r->cap.s_level, sizeof(float) * (size_t)na);
  r->cap.peak_level = (float *)realloc(r->cap.peak_level, sizeof(float) * (size_t)na);
  r->cap.fic_ber_bits = (int32_t *)realloc(r->cap.fic_ber_bits, sizeof(int32_t) * (size_t)na);
  r->cap.fic_ber_errors = (int32_t *)realloc(r->cap.fic_ber_errors, sizeof(int32_t) * (size_t)na);
  if (r->want_soft) r->cap.soft = (int16_t *)realloc(r->cap.soft, sizeof(int16_t) * (size_t)na * 75 * ORA_2K);
  r->cap_alloc = na;
}

static void limit_sym(float *v, float lim) { if (*v > lim) *v = lim; else if (*v < -lim) *v = -lim; }

/* main/msc_handler.cpp:148-168 */
static void msc_process_block(ora_receiver *r, const int16_t *bits, int blk)
{
  const int cur = (blk - 4) % 18;
  memcpy(&r->cif[cur * ORA_2K], bits, sizeof(int16_t) * ORA_2K);
  if (cur < 17) return;
  for (int i = 0; i < r->n_back; i++) {
    /* A sub-channel that only moves to other capacity units at a reconfiguration: the Backend object keeps running (its
     * de-interleaver history is its own, backend.cpp:131-139), MscHandler hands it its slice from the new address on */
    if (r->move_cif && r->move_cif[i] >= 0 && r->cifs_done >= r->move_cif[i]) { r->back[i].d.cu_start = r->move_to[i]; r->move_cif[i] = -1; }
    ora_backend_process(&r->back[i], &r->cif[r->back[i].d.cu_start * 64]);
  }
  r->cifs_done++;
}
void ora_rx_move_subch(ora_receiver *r, int i, int new_cu_start, long at_cif)
{
  if (i < 0 || i >= r->n_back) return;
  if (!r->move_cif) {
    r->move_cif = (long *)malloc(sizeof(long) * (size_t)r->n_back);
    r->move_to = (int *)calloc((size_t)r->n_back, sizeof(int));
    for (int k = 0; k < r->n_back; k++) r->move_cif[k] = -1;
  }
  r->move_cif[i] = at_cif; r->move_to[i] = new_cu_start;
}

/* dab_processor.cpp:191-265 + :304-367 + :267-302 ; returns 0 at end of input */
static int process_rest_of_frame(ora_receiver *r, int *sample_count, int frame_no)
{
  ora_cf32 fin[ORA_TU], fout[ORA_TU];
  memcpy(fin, r->buf, sizeof(fin));
  ora_fft2048(fin, fout, 0);
  ora_demap_store_ref(&r->dm, fout);                              /* :199-202 */

  int correction = 0;
  if (r->fic.success_ratio * 10 < 30) {                           /* :205-224 */
    correction = ora_phaseref_coarse_cfo(&r->pr, fout);
    if (correction != ORA_IDX_NOT_FOUND) {
      r->freq_offs_sync += (float)correction;
      if (fabsf(r->freq_offs_sync) > 35000.0f) r->freq_offs_sync = 0.0f;
    }
    if (correction != 0) r->clock_err = 0.0f;
    r->freq_offs_bb = r->freq_offs_sync;
  }
  r->cap.fbb[frame_no] = r->freq_offs_bb;

  /* _process_ofdm_symbols_1_to_L, :304-367 */
  float fc_re = 0, fc_im = 0;
  r->cap.fic_overflow[frame_no] = r->cap.msc_overflow[frame_no] = 0;
  for (int sym = 1; sym < ORA_L; sym++) {
    if (!get_samples(r, r->buf, ORA_TS, r->freq_offs_bb)) return 0;
    *sample_count += ORA_TS;
    for (int i = ORA_TU; i < ORA_TS; i++) {                       /* :330-333  x[i]*conj(x[i-Tu]) */
      const ora_cf32 a = r->buf[i], b = r->buf[i - ORA_TU];
      fc_re += a.re * b.re + a.im * b.im;
      fc_im += a.im * b.re - a.re * b.im;
    }
    memcpy(fin, &r->buf[ORA_TG], sizeof(fin));
    ora_fft2048(fin, fout, 0);
    const long long ovf0 = r->dm.overflow_count;
    ora_demap_symbol(&r->dm, fout, r->clock_err, r->bits);        /* :342 */
    if (sym <= 3) r->cap.fic_overflow[frame_no] += (int32_t)(r->dm.overflow_count - ovf0);
    else r->cap.msc_overflow[frame_no] += (int32_t)(r->dm.overflow_count - ovf0);
    if (r->want_soft) memcpy(&r->cap.soft[((size_t)frame_no * 75 + (sym - 1)) * ORA_2K], r->bits, sizeof(r->bits));
    if (sym <= 3) ora_fic_process_block(&r->fic, r->bits, sym);   /* :347-350 */
    if (sym > 3) msc_process_block(r, r->bits, sym);              /* :357-360 */
  }
  r->cap.snr_db[frame_no] = ora_demap_snr_db(&r->dm);
  r->cap.mer_db[frame_no] = ora_demap_mer_db(&r->dm);
  r->phase_offs_cp = atan2f(fc_im, fc_re);                        /* :366 */

  limit_sym(&r->phase_offs_cp, 20.0f * (float)(M_PI / 180.0));    /* :240-242 */
  r->freq_offs_sync += r->phase_offs_cp / (float)(2 * M_PI) * 1000.0f;
  r->freq_offs_bb = r->freq_offs_sync;

  /* _process_null_symbol, :267-302 */
  if (!get_samples(r, r->buf, ORA_TN, r->freq_offs_bb)) return 0;
  *sample_count += ORA_TN;
  const int is_tii = (r->fic.cif_count & 7) >= 4;
  memcpy(fin, &r->buf[ORA_TG], sizeof(fin));
  ora_fft2048(fin, fout, 0);
  if (!is_tii) ora_demap_store_null(&r->dm, fout);
  else {                                                          /* :282-300 add_to_tii_buffer */
    for (int i = 0; i < ORA_TU; i++) { r->tii_acc[i].re += fout[i].re; r->tii_acc[i].im += fout[i].im; }
    r->tii_count++;
  }

  if (correction == 0) {                                          /* :246-251 */
    float ce = (float)ORA_INPUT_RATE * ((float)*sample_count / (float)ORA_TF - 1.0f);
    limit_sym(&ce, 307.2f);
    r->clock_err += 0.1f * (ce - r->clock_err);
  }
  r->cap.fbb_end[frame_no] = r->freq_offs_bb;
  r->cap.clock_err[frame_no] = r->clock_err;
  r->cap.fic_ratio[frame_no] = r->fic.success_ratio * 10;
  r->cap.fic_ber_bits[frame_no] = r->fic.fic_bits; r->cap.fic_ber_errors[frame_no] = r->fic.fic_errors;
  r->cap.s_level[frame_no] = r->s_level; r->cap.peak_level[frame_no] = r->peak_level;
  return 1;
}

int ora_rx_run(ora_receiver *r, const ora_cf32 *iq, size_t n_samples, int max_frames)
{
  enum { WAIT_SYNC, EVAL_SYNC, REST } state = WAIT_SYNC;
  r->iq = iq; r->n_iq = n_samples; r->pos = 0; r->eof = 0;
  float sync_thr = 0; int sample_count = 0, frames = 0;
  r->freq_offs_bb = 0; r->freq_offs_sync = 0; r->fic.success_ratio = 0;   /* :119-124 */
  for (int i = 0; i < 20; i++)                                   /* :139-142 */
    if (!get_samples(r, r->buf, ORA_TU, 0)) return 0;
  while (!r->eof && frames < max_frames) {
    switch (state) {
    case WAIT_SYNC: {                                            /* :146-160 */
      ora_demap_reset(&r->dm);
      memset(r->tii_acc, 0, sizeof(r->tii_acc)); r->tii_count = 0;   /* mTiiDetector.reset(), :150-152 */
      sample_count = 0; sync_thr = r->threshold;
      const int ok = time_sync(r);
      if (ok < 0) return frames;
      state = ok ? EVAL_SYNC : WAIT_SYNC;
      r->clock_err = 0.0f;
      break;
    }
    case EVAL_SYNC: {                                            /* :389-414 */
      if (!get_samples(r, r->buf, ORA_TU, r->freq_offs_bb)) return frames;
      const int start = ora_phaseref_correlate(&r->pr, r->buf, sync_thr);
      if (start < 0) { state = WAIT_SYNC; break; }
      const int next = ORA_TU - start;
      memmove(r->buf, &r->buf[start], sizeof(ora_cf32) * (size_t)next);
      if (!get_samples(r, &r->buf[next], ORA_TU - next, r->freq_offs_bb)) return frames;
      sample_count = start + ORA_TU;
      cap_reserve(r, frames + 1);
      r->cap.start_idx[frames] = start;
      r->cap.sym0_pos[frames] = (int32_t)(r->pos - ORA_TU);
      state = REST;
      break;
    }
    case REST: {                                                 /* :173-180 */
      if (!process_rest_of_frame(r, &sample_count, frames)) return frames;
      for (int i = 0; i < 12; i++) {
        uint8_t *dst = &r->cap.fibs[((size_t)frames * 12 + i) * 32];
        for (int b = 0; b < 32; b++) {
          uint8_t t = 0;
          for (int k = 0; k < 8; k++) t = (uint8_t)((t << 1) | (r->fic.fib_bits[i * 256 + b * 8 + k] & 1));
          dst[b] = t;
        }
        r->cap.fib_crc[frames * 12 + i] = r->fic.fib_crc[i];
      }
      frames++;
      r->cap.n_frames = frames;
      state = EVAL_SYNC;
      sync_thr = 2 * r->threshold;
      break;
    }
    }
  }
  return frames;
}

/* The per-symbol calls DabProcessor makes on OfdmDecoder, FicDecoder and MscHandler for one frame
 * (dab_processor.cpp:199-202, :336-360, :267-286), driven with externally supplied FFT outputs instead of the sample
 * reader: spectra = [n_frames][76][2048] (symbol 0 = phase reference), nulls = [n_frames][2048], clock_err[n_frames].
 * Test driver for the class-level shims (tests/cxx/shim_symbols.cpp makes the same calls on the HIP classes). */
int ora_rx_run_spectra(ora_receiver *r, const ora_cf32 *spectra, const ora_cf32 *nulls, const float *clock_err, int n_frames)
{
  for (int f = 0; f < n_frames; f++) {
    const ora_cf32 *sp = spectra + (size_t)f * ORA_L * ORA_TU;
    cap_reserve(r, f + 1);
    ora_demap_store_ref(&r->dm, sp);
    r->cap.fic_overflow[f] = r->cap.msc_overflow[f] = 0;
    for (int sym = 1; sym < ORA_L; sym++) {
      const long long ovf0 = r->dm.overflow_count;
      ora_demap_symbol(&r->dm, sp + (size_t)sym * ORA_TU, clock_err[f], r->bits);
      if (sym <= 3) r->cap.fic_overflow[f] += (int32_t)(r->dm.overflow_count - ovf0);
      else r->cap.msc_overflow[f] += (int32_t)(r->dm.overflow_count - ovf0);
      if (r->want_soft) memcpy(&r->cap.soft[((size_t)f * 75 + (sym - 1)) * ORA_2K], r->bits, sizeof(r->bits));
      if (sym <= 3) ora_fic_process_block(&r->fic, r->bits, sym);
      if (sym > 3) msc_process_block(r, r->bits, sym);
    }
    r->cap.mer_db[f] = ora_demap_mer_db(&r->dm);
    r->cap.snr_db[f] = ora_demap_snr_db(&r->dm);                  /* after symbol 75, before the null symbol (as process_rest_of_frame) */
    if (!((r->fic.cif_count & 7) >= 4)) ora_demap_store_null(&r->dm, nulls + (size_t)f * ORA_TU);
    for (int i = 0; i < 12; i++) {
      uint8_t *dst = &r->cap.fibs[((size_t)f * 12 + i) * 32];
      for (int b = 0; b < 32; b++) {
        uint8_t t = 0;
        for (int k = 0; k < 8; k++) t = (uint8_t)((t << 1) | (r->fic.fib_bits[i * 256 + b * 8 + k] & 1));
        dst[b] = t;
      }
      r->cap.fib_crc[f * 12 + i] = r->fic.fib_crc[i];
    }
    r->cap.start_idx[f] = 0; r->cap.fbb[f] = 0; r->cap.sym0_pos[f] = 0;
    r->cap.fbb_end[f] = 0; r->cap.clock_err[f] = clock_err[f]; r->cap.fic_ratio[f] = r->fic.success_ratio * 10;
    r->cap.s_level[f] = r->s_level; r->cap.peak_level[f] = r->peak_level;
    r->cap.fic_ber_bits[f] = r->fic.fic_bits; r->cap.fic_ber_errors[f] = r->fic.fic_errors;
    r->cap.n_frames = f + 1;
  }
  return n_frames;
}

/* accumulated TII null-symbol spectrum (2048 cf32) and the number of null symbols in it; clears both */
int ora_rx_take_tii(ora_receiver *r, ora_cf32 *out)
{
  const int n = r->tii_count;
  memcpy(out, r->tii_acc, sizeof(r->tii_acc));
  memset(r->tii_acc, 0, sizeof(r->tii_acc));
  r->tii_count = 0;
  return n;
}
