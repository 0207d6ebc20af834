// pipeline.h -- device-resident state of the stream-batched receiver (engine.hip / pipeline.hip).
#pragma once
#include "dabx_internal.h"

namespace dabx {

constexpr int MSC_SLOTS = 16;     // ring of decoded logical frames per sub-channel (>= 5 + 4)
constexpr int SF_SLOTS = 4;       // ring of RS-corrected super frames per sub-channel
constexpr int ACQ_NEED = 20 * TU + 50 + (TF + 1) + (TN + 50 + 21) + 64;   // worst-case samples one acquisition pass may read
constexpr int FRAME_NEED = TF + 2 * TU;                                    // worst-case samples one in-lock frame may read

enum StreamState : int32_t { ST_INIT = 0, ST_WAIT_SYNC = 1, ST_EVAL_SYNC = 2 };

// Scalars of SampleReader (sample_reader.h:91-101), DabProcessor (dab_processor.h:129-138) and
// FicDecoder (fic_decoder.h:71-76), one record per stream.
struct StreamCtl {
  unsigned long long rd;          // absolute index of the next unread sample
  unsigned long long sym0_pos;    // absolute index of the T_u part of symbol 0 of the current frame
  long long cif_no;               // CIFs written to the time-deinterleaver ring so far
  long long frames;               // frames demodulated
  int32_t state;
  int32_t nco_phase;              // currentPhase
  float s_level, peak_level;
  float f_sync, f_bb;             // mFreqOffsSyncSymb, mFreqOffsBBHz
  float phase_offs, clock_err;    // mPhaseOffsetCyclPrefRad, mClockErrHz
  float sync_thr;
  int32_t sample_count;
  int32_t start_index;
  int32_t correction;
  int32_t frame_ok;               // this batch step carries a frame for this stream
  int32_t phase_sym1;             // NCO phase before the first sample of symbol 1
  int32_t f_frame;                // round(f_bb) used for symbols 1..75
  int32_t fic_ratio;              // mFicDecodeSuccessRatio 0..10
  int32_t cif_count;              // FibDecoder::get_cif_count
  int32_t fic_errors, fic_bits, fic_block;
  float snr_db;
  // counters (summed across streams / GPUs by dabx_get_counters)
  long long fib_ok, fib_total, sync_lost;
  int32_t pad[2];
};

struct SubchDev {
  int32_t cu_start, cu_size, kbps, prot_level, short_form, dab_plus;
  int32_t nbits;                  // 24 * kbps
  int32_t active;
  const uint16_t *map;            // depuncture map, device
  long long start_cif;            // cif_no when the sub-channel was configured (Backend construction)
  long long cif_out;              // logical frames decoded so far
  // Mp4Processor state (mp4processor.h)
  int32_t blocks_in_buf, sf_sync;
  long long sf_count;
  long long sf_ok, sf_fail, rs_corr, rs_fail, fc_corr, au_ok, au_bad;
};

struct EngineDev {
  int32_t n_streams, max_subch, out_frames;
  int32_t ring_len;               // IQ ring capacity per stream in samples
  float threshold;
  int32_t strongest, fic_only, capture_soft;
  int32_t msc_stride;             // bytes per logical-frame slot (3 * max kbps)
  int32_t sf_stride;              // bytes per super-frame slot (110 * max kbps / 8)
  int32_t vit_stride;             // decision-scratch words per trellis (max over FIC and all sub-channels)
  float2 *iq;                     // [S][ring_len]
  unsigned long long *wr;         // [S] absolute index one past the last committed sample
  StreamCtl *ctl;                 // [S]
  DemapDev demap;
  float2 *spectra;                // [S][76][2048]: symbols 1..75, then the null symbol
  float2 *cp_part;                // [S][75] cyclic-prefix correlation partial sums
  float *abs_part;                // [S][76] sum |x| of the samples read per symbol (level tracking)
  uint8_t *fic_sym;               // [S][9216] Viterbi symbols of OFDM symbols 1..3
  uint8_t *tdi;                   // [S][TDI_SLOTS][55296] time-deinterleaver ring (Viterbi symbols)
  uint32_t *vit_scratch;          // decision words
  SubchDev *subch;                // [S][max_subch]
  uint8_t *fib_out;               // [S][out_frames][12][32]
  uint8_t *fib_crc;               // [S][out_frames][12]
  uint8_t *msc_out;               // [S][max_subch][MSC_SLOTS][msc_stride]
  uint8_t *sf_out;                // [S][max_subch][SF_SLOTS][sf_stride]
  int16_t *soft_cap;              // [S][75][3072] or null
};

}  // namespace dabx
