// dabx_internal.h -- shared declarations of libdabx (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/dabx.h"

namespace dabx {

constexpr int L = 76, K = 1536, TN = 2656, TF = 196608, TS = 2552, TU = 2048, TG = 504;
constexpr int K2 = 3072, FIC_IN = 2304, FIC_OUT = 768, CIF_BITS = 55296, INPUT_RATE = 2048000;
constexpr int MAX_SUBCH = 64;
constexpr int TDI_SLOTS = 64;          // time-deinterleaver ring depth in CIFs (16 history + 4*MSC_BATCH_FRAMES new, pow2)
constexpr uint16_t PUNCT = 0xFFFF;     // depuncture map entry of a punctured mother-code bit

void set_error(const char *fmt, ...);
const char *last_error();
int hip_fail(hipError_t e, const char *what, const char *file, int line);
#define DABX_HIP(x)                                                                  \
  do {                                                                               \
    hipError_t e__ = (x);                                                            \
    if (e__ != hipSuccess) return dabx::hip_fail(e__, #x, __FILE__, __LINE__);       \
  } while (0)

// ---- constant tables resident in HBM (one set per device) ---------------------------------------
struct DevTables {
  uint16_t *perm_bin;       // [1536] carrier k -> FFT bin (0..2047)                 freq_interleaver.cpp:40-76
  int16_t *bin_to_k;        // [2048] inverse of perm_bin: FFT bin -> carrier index k, -1 for the unused bins
  int16_t *bin_to_slot8;    // [256][8] register layout of fft_core.h: [tid][u] = LDS slot of the carrier of bin tid + 256 u (-1 unused):
                            //          slot(k) = (k & ~15) | sigma(k), sigma a permutation within each run of 16 carriers (tables.cpp)
  uint32_t *carrier_slot_rd;// [256] sigma(tid + 256 u), u = 0..5, four bits each: what thread tid of k_symbols reads back
  int16_t *perm_rel;        // [1536] realCarrRelIdx                                   ofdm_decoder.cpp:171-179
  float2 *prs_ref;          // [2048] phase reference symbol                           phasetable.cpp:87-101
  float2 *prs_arg_conj;     // [2048] conj(IFFT(relative phase of PRS))                phasereference.cpp:58-66
  float2 *twiddle;          // [2048] e^{-j 2 pi i / 2048}, computed in double
  uint16_t *fic_map;        // [3096]                                                  fic_decoder.cpp:79-124
  uint32_t *prbs_words;     // [288] PRBS packed MSB-first per byte, little-endian words
  uint16_t *fc_syndrome;    // [65536] fire-code burst table                           firecode_checker.cpp:61-144
  uint16_t *fc_crctab;      // [256] CRC table poly 0x782F
  uint16_t *crc_ccitt;      // [256] CRC table poly 0x1021
  uint16_t *crc_xpow;       // [1024] x^(8 m) mod the CCITT polynomial (a CRC state advanced over m zero bytes)
  uint8_t *gf_exp;          // [512] alpha^i (doubled), [255] = 0 handled in code
  uint8_t *gf_log;          // [256]
};
int get_tables(const DevTables **out);          // for the current device; builds on first use

// depuncture map of an MSC profile, cached per device. *n_in = #transmitted bits (cu_size*64).
int get_profile_map(int kbps, int prot_level, int short_form, const uint16_t **dev_map, int *n_in);
// host-side builders (tables.cpp)
int host_profile_map(int kbps, int prot_level, int short_form, std::vector<uint16_t> &map, int *n_in);
void host_fic_map(std::vector<uint16_t> &map);
bool fib_cif_count(const uint8_t *fib, int *hi, int *lo);   // fib.cpp

// ---- demapper state (SoA over the stream axis), OfdmDecoder members ofdm_decoder.h:88-104 ----------
struct DemapDev {
  float2 *phase_ref;    // [B][2048] mPhaseReference
  float *integ;         // [B][1536] mIntegAbsPhaseVector
  float *mean_power;    // [B][1536] mMeanPowerVector
  float *mean_sigma;    // [B][1536] mMeanSigmaSqVector
  float *null_power;    // [B][2048] mMeanNullPowerWithoutTII
  float *null_power2;   // [B][2048] second buffer: the engine's frame tail writes the one the demapper of the frame in flight does not read
  float *mean_value;    // [B]       mMeanValue
  float *mean_power_all;// [B]       mMeanPowerOvrAll (display / SNR only, ofdm_decoder.cpp:214)
  float *std_dev;       // [B][1536] mStdDevSqPhaseVector (ofdm_decoder.cpp:204-208): feeds the LCD record's MER (:331-340) and no soft bit
  int batch;
  int soft_type;        // 1..3
  int track_mer;        // the phase-deviation IIR is advanced (per-symbol handles: always; the engine: dabx_set_lcd_statistics)
};
int demap_alloc(DemapDev &d, int batch);
void demap_free(DemapDev &d);

// ---- kernel launchers (device pointers, asynchronous on `st`) -------------------------------------
// viterbi.hip
int launch_viterbi_i16(const int16_t *soft, int nbits, int batch, uint8_t *bits_1perbyte, hipStream_t st, int tie_mode = 0);
int launch_deconvolve_i16(const int16_t *in, int in_stride, const uint16_t *map, int nbits, int batch,
                          uint8_t *bits_1perbyte, hipStream_t st);
int viterbi_scratch_bytes_per_trellis(int nbits);
// fec.hip
int launch_rs_decode(const uint8_t *in, int batch, uint8_t *out, int16_t *ret, hipStream_t st);
int launch_firecode(uint8_t *x, int batch, int correct, uint8_t *ok, hipStream_t st);
int launch_crc16_check(const uint8_t *msgs, int stride, int len, int batch, uint8_t *ok, hipStream_t st);
// ofdm.hip
int launch_fft2048(const float2 *in, int batch, int inverse, float2 *out, hipStream_t st);
int launch_prs_correlate(const float2 *v, int batch, float threshold, int strongest, int32_t *start, hipStream_t st);
int launch_coarse_cfo(const float2 *fft, int batch, int32_t *hz, hipStream_t st);
int launch_demap_reset(DemapDev &d, hipStream_t st);
int launch_demap_init(DemapDev &d, hipStream_t st);
int launch_demap_store_ref(DemapDev &d, const float2 *fft, hipStream_t st);
int launch_demap_store_null(DemapDev &d, const float2 *fft, hipStream_t st);
int launch_demap_symbols(DemapDev &d, const float2 *fft, int n_sym, const float *clock_err, int16_t *soft, hipStream_t st);
int launch_demap_snr(DemapDev &d, float *snr_db_dev, hipStream_t st);
int launch_demap_lcd(DemapDev &d, float *out3_dev /* [B][3]: SNR dB, MER dB, mMeanValue */, hipStream_t st);

}  // namespace dabx
