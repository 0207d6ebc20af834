/*
 * dab_oracle.h -- CPU oracle for the DAB Mode-I demodulation + FEC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithms of
 * tomneda/DABstar (reference tree: /root/reference, v5.7.0) for the path
 *   IQ -> sync -> 2048-FFT -> D-QPSK soft bits -> FIC/MSC depuncture + K=7 r=1/4
 *   Viterbi -> energy de-dispersal -> CRC / fire code / RS(120,110).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product (dabstar_amd/, libdabx.so) never links it.
 *
 * Pinning status:
 *   - integer stages (Viterbi, puncture tables, EEP/UEP depuncture, RS, Galois,
 *     fire code, CRC, frequency interleaver, PRS table) are PINNED: they are checked
 *     bit-for-bit against the reference's own object code built by oracle/ref/Makefile
 *     into oracle/_ref/libdabref.so and against the fixtures in tests/golden/.
 *   - float front end (NCO sample reader, null-dip detector, PRS correlator, coarse
 *     CFO, D-QPSK demapper) and the Qt-entangled glue (FicDecoder, Backend,
 *     Mp4Processor, DabProcessor FSM) are restated line by line but PARITY UNPINNED:
 *     those reference classes need the GUI header dabradio.h / generated ui_*.h and
 *     FFTW3f (third party, un-vendored, not in this image), so they cannot be built
 *     here without stand-ins.  The FFT is restated as the mathematical DFT computed in
 *     double and rounded to float (FFTW3f, any 3.x, computes the same DFT in float).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/src/).
 */
#ifndef DAB_ORACLE_H
#define DAB_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Mode-I constants: common/glob_defs.h:40-55 ---- */
#define ORA_L      76
#define ORA_K      1536
#define ORA_TN     2656
#define ORA_TF     196608
#define ORA_TS     2552
#define ORA_TU     2048
#define ORA_TG     504
#define ORA_2K     3072
#define ORA_FIC_IN 2304
#define ORA_FIC_OUT 768
#define ORA_CIF_BITS 55296
#define ORA_INPUT_RATE 2048000
#define ORA_IDX_NOT_FOUND 100000

typedef struct { float re, im; } ora_cf32;

/* ---------------- tables (tables.c) ---------------- */
/* protection/protTables.cpp:36-68 : 24 puncturing vectors x 32 (1 = transmitted) */
const int8_t *ora_pi_codes(int pi /*1..24*/);
/* ofdm/freq_interleaver.cpp:40-76 : k in [0,1536) -> fft bin offset in [-768,768]\{0} */
void ora_freq_interleaver(int16_t perm[ORA_K]);
/* ofdm/phasetable.cpp:35-135 : PRS reference in the frequency domain (2048 cf32) */
void ora_phase_table(ora_cf32 ref[ORA_TU]);
/* decoder/fic_decoder.cpp:59-73, backend/backend.cpp:72-84 : PRBS x^9+x^5+1, all-ones */
void ora_prbs(uint8_t *out, int n);

/* ---------------- Viterbi (viterbi.c) ---------------- */
/* support/viterbi_spiral/viterbi_spiral.cpp:95-126 + viterbi_scalar.h:9-94.
 * input: 4*(nbits+6) soft bits (i16), output nbits bytes (1 bit/byte).
 * tie_mode 0 = scalar/SSE2 tie rule (decision 0), 1 = AVX2 tie rule (viterbi_16way.h). */
void ora_viterbi(const int16_t *soft, int nbits, uint8_t *out_bits);
void ora_viterbi_simd(const int16_t *soft, int nbits, uint8_t *out_bits);
void ora_viterbi_sse2(const int16_t *soft, int nbits, uint8_t *out_bits);
double ora_viterbi_seconds(long long *calls);   /* wall time spent in ora_viterbi_build since the last reset (single-threaded use) */
void ora_viterbi_seconds_reset(void);   /* viterbi_8way.h body (VITERBI_SSE2 / NEON builds) */
void ora_set_viterbi_mode(int mode);     /* 0 scalar body (default), 1 AVX2 body: what fic.c / protection.c decode with */
void ora_viterbi_build(const int16_t *soft, int nbits, uint8_t *out_bits);
void ora_set_viterbi_hook(void (*fn)(const int16_t *, int, uint8_t *));   /* bench.py: the reference's AVX2 object as the decoder */   /* body of the VITERBI_AVX2 / _SSE2 builds, viterbi_16way.h */
/* viterbi_spiral.cpp:128-164 */
void ora_viterbi_ber(const int16_t *soft, const uint8_t *punct_table, const uint8_t *bits,
                     int nbits, int *io_bits, int *io_errors);

/* ---------------- protection (protection.c) ---------------- */
/* Builds the depuncture index list: for each of the 4*(24*kbps+6) mother-code
 * positions, map[pos] = index into the punctured input or -1 if punctured.
 * returns number of transmitted bits (== CU size * 64 for legal profiles) or <0.
 * eep: protection/eep_protection.cpp:43-167 ; uep: protection/uep_protection.cpp:52-212 */
int ora_eep_map(int kbps, int prot_level /* 0..3, +4 => option B */, int32_t *map /* 96*kbps+24 */);
int ora_uep_map(int kbps, int prot_level /* 1..5 */, int32_t *map);
/* decoder/fic_decoder.cpp:79-124 */
int ora_fic_map(int32_t map[3096]);
/* protection/protection.cpp:46-59 : depuncture + viterbi */
void ora_deconvolve(const int16_t *in, const int32_t *map, int kbps, uint8_t *out_bits);

/* ---------------- CRC / fire code / RS ---------------- */
int ora_check_crc_bits(const uint8_t *bits, int nbits);          /* backend/crc.cpp:98-132 */
uint16_t ora_calc_crc(const uint8_t *data, int len);               /* backend/crc.cpp:75-86 */
int ora_check_crc_bytes(const uint8_t *msg, int len);              /* backend/crc.cpp:88-96 */
int ora_firecode_check(const uint8_t x[11]);                       /* backend/firecode_checker.cpp:162-165 */
int ora_firecode_check_and_correct(uint8_t x[11]);                 /* backend/firecode_checker.cpp:168-184 */
const uint16_t *ora_firecode_syndrome_table(void);                 /* 65536 entries */
/* backend/reed_solomon.cpp:140-158 (dec) with (8,0435,0,1,10), cutlen 135: 120 -> 110 bytes.
 * returns #corrected, 0 clean, -1 uncorrectable. */
int ora_rs_dec(const uint8_t in[120], uint8_t out[110]);
void ora_rs_enc(const uint8_t in[110], uint8_t out[120]);          /* reed_solomon.cpp:115-137 */

/* ---------------- FIC (fic.c) ---------------- */
typedef struct {
  int32_t map[3096];
  uint8_t punct[3096];
  uint8_t prbs[768];
  int16_t vit_in[3096];
  int16_t soft[ORA_FIC_IN];
  uint8_t fib_bits[4 * 768];   /* one bit per byte, after de-dispersal */
  uint8_t fic_valid[4];
  uint8_t fib_crc[12];
  int index, fic_idx;
  int fic_block, fic_errors, fic_bits;
  int success_ratio;           /* 0..10 */
  /* minimal FIB walk (decoder/fib_decoder.cpp:59-110, fib_decoder_fig0.cpp:89-101) */
  int cif_count, cif_hi, cif_lo;
} ora_fic;
void ora_fic_init(ora_fic *f);
/* decoder/fic_decoder.cpp:143-167 ; sym_idx in 1..3 */
void ora_fic_process_block(ora_fic *f, const int16_t soft[ORA_2K], int sym_idx);

/* ---------------- MSC back end (msc.c) ---------------- */
typedef struct {
  int subch_id, cu_start, cu_size, kbps, prot_level, short_form /* 1 = UEP */;
} ora_subch_desc;

typedef struct {
  ora_subch_desc d;
  int frag;                  /* cu_size*64 */
  int16_t *hist;             /* 16 * frag */
  int16_t *tmp;
  int32_t *map;
  uint8_t *prbs;
  uint8_t *outv;             /* 24*kbps bits */
  int cnt, idx;
  /* DAB+ super frame (backend/audio/mp4processor.cpp:96-241) */
  int rs_dims;
  uint8_t *frame_bytes;      /* rs_dims*120 */
  uint8_t *out_vec;          /* rs_dims*110 */
  int block_fill, blocks_in_buf, sf_sync;
  long n_cif_out;            /* number of CIFs decoded (after warm-up) */
  long n_sf_ok, n_sf_fail, n_rs_corr, n_rs_fail, n_fc_corr, n_au_ok, n_au_bad;
  /* sinks (optional, grown by the oracle) */
  uint8_t *msc_bytes; size_t msc_len, msc_cap;     /* 3*kbps bytes per CIF */
  uint8_t *sf_bytes;  size_t sf_len, sf_cap;       /* 110*kbps/8 bytes per good super frame */
  uint8_t *sfi_bytes; size_t sfi_len, sfi_cap;     /* 32 bytes per good super frame: stream parameters, AU table, per-AU CRC verdicts,
                                                      RS / fire-code corrections (mp4processor.cpp:249-333; layout = dabx_superframe_info) */
} ora_backend;
int  ora_backend_init(ora_backend *b, const ora_subch_desc *d);
void ora_backend_free(ora_backend *b);
/* backend/backend.cpp:129-161 ; in = CIF soft bits of this sub-channel (frag i16) */
void ora_backend_process(ora_backend *b, const int16_t *in);
const uint8_t *ora_backend_msc_bytes(const ora_backend *b, size_t *len);
const uint8_t *ora_backend_sf_bytes(const ora_backend *b, size_t *len);
const uint8_t *ora_backend_sfi_bytes(const ora_backend *b, size_t *len);
void ora_backend_stats(const ora_backend *b, long out[8]);

/* FIB/FIG subset (fib.c): FIG 0/0, 0/1, 0/2 -- decoder/fib_decoder.cpp:59-110, fib_decoder_fig0.cpp */
int ora_parse_fibs(const uint8_t *fib_bytes, const uint8_t *crc_ok, int n_fibs, ora_subch_desc *out, int *dab_plus, int max_out,
                   int *cif_count);
/* the same as a running decoder with a current and a next configuration (fib_decoder_fig0.cpp:102-111) */
typedef struct ora_fibdec ora_fibdec;
ora_fibdec *ora_fibdec_new(void);
void ora_fibdec_free(ora_fibdec *t);
int ora_fibdec_process(ora_fibdec *t, const uint8_t *fib_bytes, const uint8_t *crc_ok, int n_fibs);   /* returns swaps made */
void ora_fibdec_info(const ora_fibdec *t, long long info[10]);
int ora_fibdec_subchannels(const ora_fibdec *t, int next, ora_subch_desc *out, int *dab_plus, int max_out);

/* ETI(NI) frame of one CIF (eti.c): eti_generator.cpp:169-199, :207-308; returns the bytes used before the 0x55 padding */
int ora_eti_frame(int hi, int lo, int minor, const ora_subch_desc *sc, int nst, const uint8_t *fic96, const uint8_t *const *msc, uint8_t *eti);

/* recorded-IQ payload -> cf32 at 2.048 MS/s (iqfile.c); family 0 raw / 1 wav / 2 uff, container 0 u8 1 s8 2 i16 3 i24 4 i32 5 f32 */
long long ora_iq_convert(int family, int container, int big_endian, int swap_iq, int bits, int rate, const uint8_t *bytes,
                         long long n_bytes, float *out, long long max_out);

/* ---------------- OFDM front end (ofdm.c) ---------------- */
void ora_fft2048(const ora_cf32 *in, ora_cf32 *out, int inverse); /* unnormalised DFT */

typedef struct {
  ora_cf32 phase_ref[ORA_TU];
  float integ_abs_phase[ORA_K], mean_power[ORA_K], mean_sigma_sq[ORA_K], std_dev_sq[ORA_K];
  float mean_null_power[ORA_TU];
  float mean_power_ovr_all;
  float mean_value;
  int16_t perm[ORA_K];
  int soft_bit_type;   /* 1,2,3 = SOFTDEC1..3 (glob_enums.h:49-56), default 1 */
  long long overflow_count;   /* test bookkeeping: soft values that left the int16 range before the (i16) cast -- undefined
                                 behaviour in the reference (ofdm_decoder.cpp:254-255), wrapped here as x86-64 does it */
} ora_demap;
void ora_demap_init(ora_demap *d);                                   /* ofdm_decoder.cpp:43-66 */
void ora_demap_reset(ora_demap *d);                                  /* ofdm_decoder.cpp:90-101 */
void ora_demap_store_ref(ora_demap *d, const ora_cf32 *fft);         /* ofdm_decoder.cpp:132-145 */
void ora_demap_store_null(ora_demap *d, const ora_cf32 *fft);        /* ofdm_decoder.cpp:114-130 */
void ora_demap_symbol(ora_demap *d, const ora_cf32 *fft, float clock_err, int16_t out[ORA_2K]); /* :147-355 */
float ora_demap_snr_db(const ora_demap *d);                       /* :326-343, :358-371 (SNR of the LCD statistics) */
const int16_t *ora_interleave_map(void);                          /* the 16-entry time-de-interleaver map, backend.cpp:129 / eti_generator.cpp:22 */
float ora_demap_mean_value(const ora_demap *d);
const float *ora_demap_std_dev_sq(const ora_demap *d);
float ora_demap_mer_db(const ora_demap *d);                       /* :331-340 (MER of the LCD statistics, from the :204-208 IIR) */

/* PRS correlator / coarse CFO: ofdm/phasereference.cpp */
typedef struct {
  ora_cf32 ref[ORA_TU];
  ora_cf32 ref_arg_conj[ORA_TU];
  int strongest;
} ora_phaseref;
void ora_phaseref_init(ora_phaseref *p);
int  ora_phaseref_correlate(ora_phaseref *p, const ora_cf32 *v, float threshold); /* :87-213 */
int  ora_phaseref_coarse_cfo(ora_phaseref *p, const ora_cf32 *fft_sym0);          /* :223-280 */

ora_demap *ora_demap_new(void);
void ora_demap_free(ora_demap *d);
void ora_demap_set_type(ora_demap *d, int type);
ora_phaseref *ora_phaseref_new(void);
void ora_phaseref_free(ora_phaseref *p);
void ora_phaseref_set_strongest(ora_phaseref *p, int on);

/* ---------------- whole receiver (receiver.c) : main/dab_processor.cpp ---------------- */
typedef struct ora_receiver ora_receiver;
ora_receiver *ora_rx_create(const ora_subch_desc *subch, int n_subch);
void ora_rx_destroy(ora_receiver *r);
void ora_rx_move_subch(ora_receiver *r, int i, int new_cu_start, long at_cif);   /* back end i reads its slice at new_cu_start from CIF at_cif on */
void ora_rx_set_dc_iq(ora_receiver *r, int mode);   /* SampleReader::set_dc_and_iq_correction: 0 off, 1 DC, 2 DC + IQ */
void ora_dciq_sample(ora_cf32 *v, int mode, float *st5);
float ora_level_walk(const ora_cf32 *x, size_t n, float s0);   /* SampleReader's sLevel after reading x[0..n) from s0 (sample_reader.cpp:245-248) */
void ora_dciq_buffer(ora_cf32 *iq, size_t n, int mode, float *st5);
void ora_dciq_buffer_f64(ora_cf32 *iq, size_t n, int mode, double *st5);
void ora_rx_configure(ora_receiver *r, float threshold, int sync_strongest, int soft_bit_type);   /* defaults 3.0, 0, 1 */
/* Feed a finite cf32 buffer (file-player mode, no pacing); returns #frames processed. */
int ora_rx_run(ora_receiver *r, const ora_cf32 *iq, size_t n_samples, int max_frames);
/* capture buffers */
typedef struct {
  int n_frames;
  uint8_t *fibs;       /* n_frames * 12 * 32 bytes (packed) */
  uint8_t *fib_crc;    /* n_frames * 12 */
  int16_t *soft;       /* optional: n_frames * 75 * 3072 (NULL unless enabled) */
  int32_t *start_idx;  /* n_frames */
  float   *fbb;        /* n_frames : BB freq offset used for symbols 1..75 */
  int32_t *sym0_pos;   /* n_frames : absolute sample index of symbol-0 T_u start */
  /* scalars as they stand when the frame is complete (after the null symbol): what the next frame's NCO / demapper use */
  float   *fbb_end;    /* n_frames : mFreqOffsBBHz after the fine-CFO update, dab_processor.cpp:236-242 */
  float   *clock_err;  /* n_frames : mClockErrHz, :246-251 */
  int32_t *fic_ratio;  /* n_frames : FicDecoder::get_fic_decode_ratio_percent */
  float   *snr_db;     /* n_frames : SNR of the LCD statistics after symbol 75, ofdm_decoder.cpp:326-343 */
  int32_t *fic_overflow;  /* n_frames : soft values of symbols 1..3 that left the int16 range (see ora_demap.overflow_count) */
  int32_t *msc_overflow;  /* n_frames : the same for symbols 4..75 */
  float   *s_level;       /* n_frames : SampleReader::sLevel after the frame's last sample (sample_reader.cpp:245-248) */
  float   *peak_level;    /* n_frames : SampleReader::peakLevel likewise */
  int32_t *fic_ber_bits;   /* n_frames : FicDecoder::mFicBits / mFicErrors after the frame's four FIC blocks (fic_decoder.cpp:199-210) */
  int32_t *fic_ber_errors;
  float   *mer_db;        /* n_frames : MER of the LCD statistics after symbol 75, ofdm_decoder.cpp:331-340 */
} ora_rx_capture;
void ora_rx_enable_soft_capture(ora_receiver *r, int on);
const ora_rx_capture *ora_rx_get_capture(ora_receiver *r);
int ora_rx_run_spectra(ora_receiver *r, const ora_cf32 *spectra, const ora_cf32 *nulls, const float *clock_err, int n_frames);   /* per-symbol class calls on given FFT outputs */
int ora_rx_take_tii(ora_receiver *r, ora_cf32 *out2048);   /* TII null-symbol sum + count since the last call */
ora_backend *ora_rx_backend(ora_receiver *r, int i);
ora_fic *ora_rx_fic(ora_receiver *r);

#ifdef __cplusplus
}
#endif
#endif
