/*
 * dabx.h -- C ABI of libdabx: MI355X-native OFDM-demodulation + FEC back end for DAB/DAB+ Mode I.
 *
 * Drop-in boundary for the one hot path of tomneda/DABstar (reference tree /root/reference, paths
 * below relative to its src/).  The reference has no FFI seam: the seam is the C++ class surface
 * OfdmDecoder / FicDecoder / MscHandler selected at compile time (base/main/dab_processor.h:53-57
 * picks ofdm_decoder_simd.h or ofdm_decoder.h).  A HIP back end enters as a third alternative; the
 * functions below are what that alternative binds to (see INTEGRATION.md for the C++ stub).
 *
 * Conventions: extern "C", plain pointers and sizes, int return codes (0 = ok, <0 = dabx_err),
 * caller-allocated buffers, no exceptions cross the boundary, one caller thread per handle.
 * Functions ending in _dev take DEVICE pointers (HBM-resident batches); all others take HOST
 * pointers and stage through the device themselves.  There is no CPU fallback: every entry point
 * fails with DABX_E_NODEVICE when no HIP device is usable.
 */
#ifndef DABX_H
#define DABX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DABX_ABI_VERSION 6

typedef enum {
  DABX_OK = 0,
  DABX_E_NODEVICE = -1,   /* no usable HIP device / kernel image not loadable */
  DABX_E_ARG = -2,        /* bad argument */
  DABX_E_PROFILE = -3,    /* illegal (bit rate, protection level) combination */
  DABX_E_HIP = -4,        /* HIP runtime error (see dabx_last_error) */
  DABX_E_STATE = -5,      /* call not valid in this state */
  DABX_E_NOMEM = -6
} dabx_err;

/* Mode-I constants, common/glob_defs.h:40-55 */
enum { DABX_L = 76, DABX_K = 1536, DABX_TN = 2656, DABX_TF = 196608, DABX_TS = 2552, DABX_TU = 2048,
       DABX_TG = 504, DABX_2K = 3072, DABX_FIC_IN = 2304, DABX_FIC_OUT = 768, DABX_CIF_BITS = 55296 };

typedef struct { float re, im; } dabx_cf32;

const char *dabx_last_error(void);
int dabx_abi_version(void);
/* Selects the HIP device for the calling thread's subsequent dabx calls (default 0). */
int dabx_set_device(int device);
int dabx_device_count(void);

/* ===================================================================================== stage level
 * Each entry mirrors one reference function; batch = number of independent problems.           */

/* ViterbiSpiral::deconvolve (base/support/viterbi_spiral/viterbi_spiral.h:20, .cpp:95-126):
 * soft  : batch x 4*(nbits+6) int16 mother-code soft bits (0 at punctured positions)
 * bits  : batch x nbits bytes, one decoded bit per byte.  Canonical scalar tie rule. */
int dabx_viterbi(const int16_t *soft, int nbits, int batch, uint8_t *bits);
/* The same with the decoder arithmetic selected: tie_mode 0 = the canonical scalar body (viterbi_scalar.h: int32 metrics, a
 * tie keeps predecessor i); 1 = the reference's VITERBI_AVX2 build (viterbi_16way.h:9-58: uint16 saturating metrics,
 * renormalisation above 60000, a tie goes to predecessor i + 32); 2 = its VITERBI_SSE2 / NEON builds (viterbi_8way.h:9-53:
 * signed int16 metrics saturating at 32767, renormalisation above 30000, ties as in the scalar body).  Each is bit-identical
 * to the respective object code of the reference. */
int dabx_viterbi_mode(const int16_t *soft, int nbits, int batch, int tie_mode, uint8_t *bits);

/* Protection::deconvolve for EEP/UEP (base/protection/protection.h:44; eep_protection.cpp:43-167,
 * uep_protection.cpp:52-212): in = batch x cu_size*64 punctured soft bits; out = batch x 24*kbps bits
 * (one per byte, NOT yet de-dispersed).  short_form != 0 selects the UEP table. */
int dabx_deconvolve(const int16_t *in, int in_stride, int kbps, int prot_level, int short_form, int batch,
                    uint8_t *bits);
/* Size in soft bits (cu_size*64) of a legal profile, or DABX_E_PROFILE. */
int dabx_profile_input_bits(int kbps, int prot_level, int short_form);
/* Host only: the depuncture index list of a profile (96*kbps + 24 entries: mother-code bit i comes from transmitted bit
 * map[i], 0xFFFF = punctured) -- EepProtection / UepProtection constructors, eep_protection.cpp:43-150,
 * uep_protection.cpp:136-196.  Returns the number of transmitted bits. */
int dabx_profile_map(int kbps, int prot_level, int short_form, uint16_t *map, int max_entries);

/* FicDecoder::process_block x3 (base/decoder/fic_decoder.h:49, .cpp:143-262): soft = batch x 9216
 * int16 (OFDM symbols 1..3); fibs = batch x 12 x 32 bytes (packed, de-dispersed);
 * crc_ok = batch x 12 flags. */
int dabx_fic_decode(const int16_t *soft, int batch, uint8_t *fibs, uint8_t *crc_ok);

/* ---- FicDecoder, per OFDM symbol and stateful (base/decoder/fic_decoder.h:42-58) -------------------------------------
 * The handle is the GPU-side FicDecoder of ONE ensemble: what the `FicDecoder` class of the HIP build of the reference
 * (shim/fic_decoder_hip.h) binds to.  dabx_fic_process_block == FicDecoder::process_block(iOfdmSoftBits, iOfdmSymbIdx)
 * (.cpp:143-167): sym_idx = 1, 2, 3; soft = the 3072 int16 soft bits of that symbol.  A FIC block is decoded as soon as
 * its 2304 soft bits are complete, exactly when the reference calls _process_fic_input: block 0 during symbol 1, block 1
 * during symbol 2, blocks 2 and 3 during symbol 3.  Returns the number of FIC blocks this call completed (0..2; 0 also
 * while stopped); *first_fic (optional) = index of the first of them.  Their FIBs -- what the reference hands to
 * IFibDecoder::process_FIB for every FIB whose CRC holds (.cpp:234-261) -- are read with dabx_fic_get_fibs.          */
typedef struct dabx_fic dabx_fic;
int  dabx_fic_create(dabx_fic **out);
void dabx_fic_destroy(dabx_fic *f);
int  dabx_fic_process_block(dabx_fic *f, const int16_t *soft /* 3072 */, int sym_idx, int *first_fic);
/* FIC block fic_idx (0..3) of the current frame: 3 FIBs x 32 packed bytes (de-dispersed) + 3 CRC flags. */
int  dabx_fic_get_fibs(dabx_fic *f, int fic_idx, uint8_t fibs[96], uint8_t crc_ok[3]);
/* FicDecoder::get_fib_bits(u8 *, bool *) (.cpp:310-321): mFibBitsEntireFrame as 3072 bytes holding one bit each and
 * mFicValid[4] (all three FIBs of the block passed their CRC). */
int  dabx_fic_get_fib_bits(dabx_fic *f, uint8_t *bits /* 3072 */, uint8_t *valid /* 4 */);
int  dabx_fic_get_decode_ratio_percent(dabx_fic *f);        /* get_fic_decode_ratio_percent, .cpp:323-326 */
int  dabx_fic_reset_decode_success_ratio(dabx_fic *f);      /* fic_decoder.h:53 */
int  dabx_fic_stop(dabx_fic *f);                            /* stop(): process_block becomes a no-op, .cpp:182-185, 264-268 */
int  dabx_fic_restart(dabx_fic *f);                         /* restart(): ratio = 0, running, .cpp:270-275 */
/* FibDecoder::get_cif_count as the FIG 0/0 walk of the decoded FIBs left it (fib_decoder_fig0.cpp:89-101): hi * 250 + lo */
int  dabx_fic_get_cif_count(dabx_fic *f);
/* The channel BER of the FIC (ViterbiSpiral::calculate_BER, viterbi_spiral.cpp:128-164, driven from fic_decoder.cpp:199-210): the decoded
 * bits re-encoded and compared with the signs of the 2304 transmitted soft bits of every block; both counters are halved after every
 * 40th block.  status_* = the pair as it stood at the newest 40th block BEFORE halving: errors / bits of it is the value the reference
 * hands to signal_fic_status (fic_decoder.cpp:205); blocks = FIC blocks since the last such report (mFicBlock). */
typedef struct { int32_t bits, errors, status_bits, status_errors, blocks, reserved[3]; } dabx_fic_ber;
int  dabx_fic_get_ber(dabx_fic *f, dabx_fic_ber *out);

/* ---- MscHandler, per OFDM symbol and stateful (base/backend/msc_handler.h:36-47, backend.cpp:60-161) --------------
 * The handle is the GPU-side MscHandler of ONE ensemble with up to max_services back ends: the CIF buffer, every back
 * end's 16-CIF time de-interleaver history, depuncture + Viterbi + energy de-dispersal and (for DAB+ services) the
 * Mp4Processor's super-frame synchronisation + RS(120,110) live on the device.
 * dabx_msc_set_channel == MscHandler::set_channel (msc_handler.cpp:123-135): adds a back end for the sub-channel; its
 * de-interleaver starts filling with the next CIF (Backend ctor, backend.cpp:60-84).  Returns the service slot (>= 0).
 * dabx_msc_process_block == MscHandler::process_block(iSoftBits, iBlockNr) (msc_handler.cpp:140-168): blk_nr = 4..75;
 * the block that completes a CIF ((blk_nr - 4) % 18 == 17) runs every back end and returns 1, the others return 0.
 * dabx_msc_get_frame: the logical frame Backend::_process_segment hands to BackendDriver::add_to_frame for that CIF
 * (backend.cpp:140-160) as 3 * kbps PACKED bytes (the reference's outV holds the same 24 * kbps bits one per byte, MSB of
 * byte 0 first); returns the byte count, or 0 while the service's de-interleaver is still filling (first 16 CIFs).
 * dabx_msc_get_superframe: the RS-corrected DAB+ super frame completed by that CIF, if any (110 * kbps / 8 bytes, else 0);
 * dabx_msc_get_superframe_info: its dabx_superframe_info record (returns 1, else 0). */
typedef struct dabx_msc dabx_msc;
struct dabx_subch_desc_s;
int  dabx_msc_create(int max_services, dabx_msc **out);
void dabx_msc_destroy(dabx_msc *m);
int  dabx_msc_set_channel(dabx_msc *m, const struct dabx_subch_desc_s *d);
int  dabx_msc_stop_service(dabx_msc *m, int slot);          /* stop_service, msc_handler.cpp:77-103 (the shim maps SubChId + flag to the slot) */
int  dabx_msc_stop_all_services(dabx_msc *m);               /* msc_handler.cpp:105-118 */
int  dabx_msc_is_service_running(dabx_msc *m, int slot);
int  dabx_msc_process_block(dabx_msc *m, const int16_t *soft /* 3072 */, int blk_nr);
int  dabx_msc_get_frame(dabx_msc *m, int slot, uint8_t *bytes, int max_bytes);
int  dabx_msc_get_superframe(dabx_msc *m, int slot, uint8_t *bytes, int max_bytes);
struct dabx_superframe_info_s;
int  dabx_msc_get_superframe_info(dabx_msc *m, int slot, struct dabx_superframe_info_s *out);
struct dabx_subch_stats_s;
int  dabx_msc_get_stats(dabx_msc *m, int slot, struct dabx_subch_stats_s *out);

/* ReedSolomon::dec(in, out, 135) with (8,0435,0,1,10) (base/backend/reed_solomon.h:28, mp4processor.cpp:63,203):
 * in = batch x 120, out = batch x 110, ret = batch x int16 (#corrected | 0 | -1). */
int dabx_rs_decode(const uint8_t *in, int batch, uint8_t *out, int16_t *ret);

/* FirecodeChecker::check / check_and_correct_6bits (base/backend/firecode_checker.h:42-43):
 * x = batch x 12 bytes (11 used; byte 11 may be touched exactly as in the reference), ok = batch flags */
int dabx_firecode_check(const uint8_t *x, int batch, uint8_t *ok);
int dabx_firecode_check_and_correct(uint8_t *x, int batch, uint8_t *ok);

/* check_crc_bytes (base/backend/crc.cpp:88-96): msgs = batch x stride bytes, CRC follows len bytes */
int dabx_crc16_check(const uint8_t *msgs, int stride, int len, int batch, uint8_t *ok);

/* fftwf_execute of the 2048-point forward/backward plan (base/main/dab_processor.cpp:63,201;
 * ofdm/phasereference.cpp:51-52): unnormalised DFT, batch x 2048 cf32. */
int dabx_fft2048(const dabx_cf32 *in, int batch, int inverse, dabx_cf32 *out);

/* OfdmDecoder (base/ofdm/ofdm_decoder.h:46-73) -- opaque per-stream demapper state on the device. */
typedef struct dabx_demap dabx_demap;
int dabx_demap_create(int batch, dabx_demap **out);
void dabx_demap_destroy(dabx_demap *d);
int dabx_demap_reset(dabx_demap *d);                                              /* reset()  :90-101  */
int dabx_demap_store_reference_symbol_0(dabx_demap *d, const dabx_cf32 *fft);     /* :132-145 batch x 2048 */
int dabx_demap_store_null_symbol_without_tii(dabx_demap *d, const dabx_cf32 *fft);/* :114-130 */
/* decode_symbol :147-355 for n_sym consecutive symbols: fft = batch x n_sym x 2048,
 * clock_err = batch floats, soft = batch x n_sym x 3072 int16. */
int dabx_demap_decode_symbols(dabx_demap *d, const dabx_cf32 *fft, int n_sym, const float *clock_err, int16_t *soft);
int dabx_demap_set_soft_bit_gen_type(dabx_demap *d, int type /*1..3*/);
/* SLcdData::SNR as decode_symbol computes it for the LCD statistics (:326-343 with _compute_noise_Power :358-371) from the
 * state as it stands: 10 log10((mMeanPowerOvrAll - noise) / noise); snr_db = batch floats. */
int dabx_demap_get_snr_db(dabx_demap *d, float *snr_db);
/* The device-side numbers of the LCD record (OfdmDecoder::SLcdData, ofdm_decoder.h:53-61; filled at ofdm_decoder.cpp:326-345) from the state
 * as it stands, batch floats each, any pointer may be NULL: SNR as above; MER = 10 log10((pi/4)^2 / mean over the carriers of
 * mStdDevSqPhaseVector) (:204-208, :331-340: the per-carrier IIR of the squared phase distance from the diagonal); mean_value = mMeanValue
 * (:294), which the record carries as TestData1. */
int dabx_demap_get_lcd_data(dabx_demap *d, float *snr_db, float *mer_db, float *mean_value);

/* PhaseReference::correlate_with_phase_ref_and_find_max_peak (base/ofdm/phasereference.cpp:87-213):
 * v = batch x 2048 cf32, returns start index per problem (or -1). */
int dabx_prs_correlate(const dabx_cf32 *v, int batch, float threshold, int strongest, int32_t *start_index);
/* PhaseReference::estimate_carrier_offset_from_sync_symbol_0 (:223-280): fft = batch x 2048 */
int dabx_coarse_cfo(const dabx_cf32 *fft_sym0, int batch, int32_t *hz);

/* FIB/FIG subset (host side, no device needed): sub-channel organisation FIG 0/1, service components FIG 0/2,
 * CIF counter FIG 0/0 -- base/decoder/fib_decoder.cpp:59-110, fib_decoder_fig0.cpp:89-101, 142-224, 230-293,
 * getters fib_decoder.cpp:547-557, 673-691.  fibs = n x 32 bytes as delivered by dabx_read_fibs / dabx_fic_decode.
 * Returns the number of sub-channels found (in order of first appearance, as FibDecoder::get_sub_channel_id_list
 * lists them); dab_plus = 1 / 0 / -1 (ASCTy 63 / other / unknown). */
struct dabx_subch_desc_s;
int dabx_parse_fibs(const uint8_t *fibs, const uint8_t *crc_ok, int n_fibs, struct dabx_subch_desc_s *out, int max_out,
                    int32_t *cif_count);
/* The same as a running decoder (FibDecoder::process_FIB, fib_decoder.cpp:59-106, fed FIB by FIB in transmission order): it keeps
 * the CURRENT and the NEXT multiplex configuration -- FIG 0/1 and 0/2 are filed under _get_config_ptr(C/N flag),
 * fib_decoder.h:97, fib_decoder_fig0.cpp:149, 240 -- and swaps them when the change flags of FIG 0/0 go back to 0
 * (fib_decoder_fig0.cpp:102-111: std::swap(curr, next); next->reset(); see dabx_fibdec_set_reference_quirks for "from which value").  A sub-channel that reaches beyond the CIF or overlaps a
 * known one restarts the collection (fib_decoder.cpp:131-141), as in the reference. */
typedef struct dabx_fibdec dabx_fibdec;
typedef struct {
  int64_t fibs_processed;     /* FIBs handed to dabx_fibdec_process so far (those with a failed CRC are counted and skipped) */
  int64_t fig00_fib;          /* index, counted like fibs_processed, of the FIB that carried the newest FIG 0/0; -1: none yet */
  int64_t last_change_fib;    /* ... of the FIB whose FIG 0/0 made the newest swap; -1: none yet */
  int32_t cif_count;          /* newest FIG 0/0: CIFCountHi * 250 + CIFCountLo (mCifCount); -1: none yet */
  int32_t cif_count_hi, cif_count_lo;
  int32_t change_flags;       /* its ChangeFlags (EN 300 401 6.4.1: 0 none, 1 sub-channel, 2 service organisation, 3 both) */
  int32_t occurrence_change;  /* its OccurrenceChange: low byte (0..249) of the CIF count from which the next configuration applies;
                                 meaningful while change_flags != 0 */
  int32_t n_changes;          /* swaps so far */
  int32_t n_restarts;         /* _restart_fib_decoding calls so far */
  int32_t reserved[3];
} dabx_fibdec_info;
int  dabx_fibdec_create(dabx_fibdec **out);
void dabx_fibdec_destroy(dabx_fibdec *d);
int  dabx_fibdec_reset(dabx_fibdec *d);                              /* FibDecoder::connect_channel, fib_decoder.cpp:143-150 */
/* When the two configurations are swapped.  0 (default): whenever an announcement ends -- the change flags of FIG 0/0 go from ANY non-zero
 * value back to 0 (EN 300 401 6.4.1: 1 = sub-channel organisation, 2 = service organisation, 3 = both) -- and a table the announcement
 * never filled (no FIG 0/1 resp. 0/2 with C/N = 1 seen) is carried over from the current configuration.  1: the reference's rule, bit for
 * bit: only after flags 3 (fib_decoder_fig0.cpp:103); after flags 1 or 2 its next table is neither swapped nor reset, and the stale
 * entries take part in the reconfiguration after that. */
int  dabx_fibdec_set_reference_quirks(dabx_fibdec *d, int on);
/* n_fibs x 32 bytes + CRC flags (what dabx_read_fibs / dabx_fic_decode deliver); returns the number of swaps made in this call */
int  dabx_fibdec_process(dabx_fibdec *d, const uint8_t *fibs, const uint8_t *crc_ok, int n_fibs);
int  dabx_fibdec_get_info(const dabx_fibdec *d, dabx_fibdec_info *out);
/* sub-channel table of the current (next = 0) or the next (next = 1) configuration, like dabx_parse_fibs */
int  dabx_fibdec_subchannels(const dabx_fibdec *d, int next, struct dabx_subch_desc_s *out, int max_out);

/* ===================================================================================== engine level
 * Stream-batched receiver: the device-side equivalent of DabProcessor::run
 * (base/main/dab_processor.cpp:110-442) for n_streams independent ensembles.                   */
typedef struct dabx_engine dabx_engine;

typedef struct {
  int32_t n_streams;        /* independent ensembles resident on this GPU */
  int32_t ring_frames;      /* IQ ring capacity per stream in units of T_F samples (>= 2) */
  int32_t max_subch;        /* sub-channels decoded per stream (<= 64) */
  int32_t out_frames;       /* output ring depth in frames (>= 1) */
  float   sync_threshold;   /* ProcessParams::threshold, main/dabradio.cpp:92 (3.0) */
  int32_t sync_strongest;   /* configuration.cpp:65 default 0 */
  int32_t soft_bit_type;    /* glob_enums.h:49-56, default 1 (SOFTDEC1) */
  int32_t fic_only;         /* 1: BASELINE config 2 (FIC Viterbi only) */
  int32_t capture_soft;     /* 1: keep int16 soft bits of the last frame (debug / parity tests) */
  int32_t viterbi_tie_mode; /* 0: canonical scalar Viterbi (CMake default); 1 / 2: arithmetic of the VITERBI_AVX2 / VITERBI_SSE2 builds,
                               see dabx_viterbi_mode */
  int32_t dc_iq_correction; /* SampleReader::set_dc_and_iq_correction (sample_reader.cpp:218-243, 334-346; configuration.cpp:75-76
                               default off): 0 off, 1 DC removal, 2 DC removal + IQ-imbalance correction, applied to every
                               committed sample in place in the ring (dabx_read_iq then returns corrected samples) */
  int32_t schedule;         /* 0 (default): overlapped -- the batched MSC decode and the demapping of a frame's MSC symbols run on
                               their own HIP streams next to the front end of the following frames; 1: serial -- every kernel on
                               ONE HIP stream in program order (debugging aid, deterministic per-kernel profiles; same bytes) */
  int32_t msc_fast_min_jobs;  /* trellises per 7-frame MSC batch (all streams together) from which the MSC runs on the
                                 lane-per-trellis kernels; below it the wave-per-trellis kernel decodes everything.  0 = default
                                 20480 (measured break-even, about 41 streams x 18 sub-channels) */
  int32_t msc_class_min_jobs; /* smallest protection-profile class (trellises per batch, all streams together) that gets its own
                                 lane-per-trellis class; smaller ones stay on the wave-per-trellis kernel.  0 = default 256 */
  int32_t exact_level_tracker;/* SampleReader's signal-level IIR (sample_reader.cpp:246-248).  Out of lock -- where the null-symbol
                                 search compares against it -- it is always the reference's recurrence, sample by sample.  In lock,
                                 where nothing reads it:
                                 0 (default): advanced chunk by chunk (relative error ~1e-5: what dabx_stats.signal_level shows in
                                   lock), and when a stream falls out of lock, walked exactly from the point and value at which the
                                   search had handed it over -- so the search continues from the reference's value, provided the
                                   samples read since then are still in the ring (pushes through dabx_push_iq* and the file
                                   readers are tracked; a zero-copy producer's are if it uses dabx_announce_write; a lock that outlasts
                                   the ring has lost them).  Then two walks are started a little below and above the chunk-wise
                                   value of the oldest frame boundary still in the ring; the recurrence forgets, and once they have
                                   merged into the same float (5 - 9 frames) that float is the exact value (level_healed_events).
                                   Failing that too, the level continues from the chunk-wise value: level_unanchored_events; the
                                   returns walked from the anchor itself are counted in level_rewalk_events;
                                 1: exact in lock too (a second pass over every sample on a HIP stream of its own: -27 % throughput
                                   at 512 streams, docs/history/r01-r04_design_notebook.md 6);
                                 2: chunk-wise only, the search continues from that value (the behaviour before round 4). */
  int32_t acquire_mode;       /* streams OUT of lock (null-symbol search + candidate correlations, k_acquire): 0 (default) -- searched on a HIP
                                 stream of their own next to the steps of the streams in lock, which never wait for them (a stream joins the
                                 first step after its search has finished); dabx_process(sync != 0) waits for the frames it issued, not for a
                                 search pass beside them.  While fewer than half of the streams are in lock -- start-up, a lone stream that
                                 lost its lock -- the search runs in step: every step first gives every such stream a frame's worth of
                                 search, exactly DabProcessor's order of events per stream.  1: always in step; 2: always asynchronous.
                                 While the device's count of streams in lock equals n_streams no search is launched at all (it would
                                 return at once for every stream): a stream that then loses its lock is searched from the first step ISSUED
                                 after the loss -- with sync = 0 the host may have queued steps ahead, which the stream sits out.
                                 Same samples, same decisions, same frames either way -- only WHEN differs.  (With dc_iq_correction the
                                 search always runs in step: the correction of newly committed samples is ordered on the front-end stream.) */
} dabx_config;

/* SDescriptorType subset (common/dab_constants.h:119-135) */
typedef struct dabx_subch_desc_s {
  int32_t subch_id, cu_start, cu_size, kbps, prot_level /* +4 => EEP-B */, short_form /* 1 = UEP */;
  int32_t dab_plus;          /* 1: run super-frame sync + RS(120,110) (Mp4Processor) */
  int32_t reserved;
} dabx_subch_desc;

typedef struct {
  int64_t frames;            /* frames demodulated since open */
  int64_t samples_consumed;
  int32_t state;             /* 0 wait-for-dip, 1 eval-sync, 2 in-frame */
  int32_t fic_ratio_percent; /* FicDecoder::get_fic_decode_ratio_percent */
  float   freq_offs_bb_hz, clock_err_hz;
  float   snr_db_est;        /* OfdmDecoder's LCD SNR (ofdm_decoder.cpp:326-343) after the newest frame's last symbol */
  int32_t last_start_index, cif_count;
  int64_t fib_ok, fib_total, sf_ok, sf_fail, rs_corrected, rs_failed, au_ok, au_bad, cifs_decoded;
  float   signal_level;      /* SampleReader::sLevel (sample_reader.h:70,95; .cpp:245-248) after the newest frame */
  float   peak_level;        /* SampleReader::peakLevel (.cpp:247); tracked out of lock and, with exact_level_tracker, in lock */
  /* -- ABI 4 (everything above is the ABI-3 record) -- */
  int64_t level_margin_events; /* null-symbol search: comparisons level/50 <> 0.55 / 0.75 sLevel (timesyncer.cpp:58, 74) that fell within
                                1e-4 (relative) of their threshold, i.e. that the default (chunk-wise, ~1e-5) in-lock level tracker
                                could conceivably have decided differently from the sample-serial one; 0 = the approximation never mattered */
  int64_t level_rewalk_events;     /* losses of lock after which the level was re-walked exactly from its anchor (exact_level_tracker 0) */
  int64_t level_unanchored_events; /* ... after which it had to continue from the chunk-wise value (samples no longer in the ring) */
  int64_t level_healed_events;     /* ... after which the anchor was gone but two walks started 2^-9 either side of a frame boundary's
                                      chunk-wise value, four or more frames back in the ring, had merged into one float: exact all the same */
  /* -- ABI 5 -- */
  int64_t fic_ber_bits;      /* FicDecoder::mFicBits / mFicErrors (fic_decoder.h:74-75): transmitted FIC bits compared with the re-encoded */
  int64_t fic_ber_errors;    /* decoder output and those that differed (ViterbiSpiral::calculate_BER, viterbi_spiral.cpp:128-164), both halved
                                every 40 FIC blocks (fic_decoder.cpp:201-210); the channel BER the reference displays is errors / bits */
  /* -- ABI 6 (in the place of the first reserved word: the record's size is unchanged) -- */
  float   mer_db_est;        /* OfdmDecoder's LCD MER (ofdm_decoder.cpp:204-208, 331-340) after the newest frame's last symbol; 0 unless
                                dabx_set_lcd_statistics switched its per-carrier IIR on */
  float   reserved_f;
  int64_t reserved[1];       /* zero; later fields go here without changing the record's size */
} dabx_stats;

void dabx_default_config(dabx_config *cfg);
int  dabx_create(const dabx_config *cfg, dabx_engine **out);
void dabx_destroy(dabx_engine *e);
/* MscHandler::set_channel / stop_service equivalent for stream (or all streams when stream < 0): d[j] describes slot j
 * (kbps == 0: empty slot).  A slot whose description is unchanged keeps decoding without interruption, and so does one whose
 * sub-channel only moves to other capacity units (same SubChId, size, bit rate, protection: a multiplex reconfiguration; its
 * de-interleaver reads the CIFs before the change at the old address); new or otherwise changed
 * slots start their 16-CIF de-interleaver fill at the current CIF; a new largest bit rate re-strides the output rings in
 * place, running services are not disturbed.
 * Streams may carry different layouts: the decoder groups the slots of all streams by protection profile (rebuilt by the
 * next dabx_process after a series of calls).  At most DABX_MSC_FAST_CLASSES (16) profiles -- those with the most decoder
 * work, sub-channels x bit rate -- run on the lane-per-trellis kernels; every further profile, and classes smaller than
 * dabx_config.msc_class_min_jobs, are decoded by the wave-per-trellis kernel in the same batch.  That is a throughput
 * distinction only: there is NO limit on the number of different profiles per engine and no error code for exceeding 16
 * (tests/test_gpu_engine.py::test_more_profiles_than_decoder_classes runs 26). */
#define DABX_MSC_FAST_CLASSES 16
int  dabx_set_subchannels(dabx_engine *e, int stream, const dabx_subch_desc *d, int n);
/* Host IQ -> device ring (IDeviceHandler::getSamples contract, common/device_handler_if.h:47-48).
 * fmt: 0 = cf32, 1 = int16 IQ (/32768, wav_reader.cpp:164), 2 = uint8 IQ ((x-127.38)/128, raw_reader.cpp:66-70).
 * Returns when the caller's buffer is free again; copy and conversion run on their own HIP stream next to frames still
 * being decoded (the receiver is drained only if the ring looks full). */
int  dabx_push_iq(dabx_engine *e, int stream, const void *iq, int fmt, size_t n_samples);   /* DABX_E_STATE: would overwrite unread samples */
/* The same without waiting for the copy: the caller keeps `iq` alive and unchanged until dabx_push_wait returns.  Meant
 * for producers that cycle through a few page-locked buffers (hipHostMalloc, or their own memory passed once through
 * dabx_host_register): consecutive pushes then run back to back as DMA at PCIe rate next to the decode.  With pageable
 * memory the call behaves like dabx_push_iq. */
int  dabx_push_iq_async(dabx_engine *e, int stream, const void *iq, int fmt, size_t n_samples);
int  dabx_push_wait(dabx_engine *e);
int  dabx_host_register(void *p, size_t bytes);     /* hipHostRegister: page-lock a producer's buffer */
int  dabx_host_unregister(void *p);
/* Device-resident producers: ring base (cf32, capacity ring_frames*T_F) and commit of n new samples. */
int  dabx_iq_ring_dev(dabx_engine *e, int stream, void **ring, size_t *capacity_samples);
int  dabx_commit_iq(dabx_engine *e, int stream /* <0: all */, size_t n_samples);
/* Optional, for device-resident producers that want the level tracker's anchor kept (dabx_config.exact_level_tracker = 0): the library
 * cannot see their writes, so after a plain dabx_commit_iq it must assume that anything behind the read cursor may have been
 * overwritten.  dabx_announce_write says, BEFORE the producer writes: "everything I have written so far, and everything I am going to
 * write until my next announcement, lies below committed + n_samples".  From its first announcement on the producer is taken at
 * its word (a commit no longer means "unknown writes"): announce before every write -- or once, with the ring's capacity, for a ring that
 * is filled once and only read again (periodic test signals). */
int  dabx_announce_write(dabx_engine *e, int stream /* <0: all */, size_t n_samples);
/* Host copy of ring samples [first, first + n) counted from the first sample ever committed (scopes, tests);
 * they must still be in the ring. */
int  dabx_read_iq(dabx_engine *e, int stream, uint64_t first, size_t n, float *iq_out);
/* Advance every stream by up to max_frames frames (bounded by available samples); returns the number of
 * batch steps executed.  Asynchronous on the engine's HIP stream unless sync != 0. */
int  dabx_process(dabx_engine *e, int max_frames, int sync);
int  dabx_synchronize(dabx_engine *e);
void *dabx_hip_stream(dabx_engine *e);
/* Results of the most recent frames (host copies). fibs: n x 12 x 32, crc: n x 12 */
int  dabx_read_fibs(dabx_engine *e, int stream, int n_frames, uint8_t *fibs, uint8_t *crc_ok);
/* Where the newest n_frames frames (n_frames <= out_frames, oldest first, the frames dabx_read_fibs returns) sit in the stream's
 * sample sequence: sym0_pos = index, counted from the first sample ever committed, of the first sample of the useful part of the
 * frame's phase reference symbol; start_index = the PRS correlation peak that placed it (DabProcessor's startIndex,
 * dab_processor.cpp:394-411).  Either array may be NULL.  Returns the number of frames written.  Time-stamps frames for a host
 * that knows when it committed which samples; lets a test compare the receiver's walk through the samples frame by frame. */
int  dabx_read_frame_info(dabx_engine *e, int stream, int n_frames, int64_t *sym0_pos, int32_t *start_index);
/* Sub-channel table announced in the FIBs of the newest frames of `stream` (dabx_parse_fibs over the FIB ring);
 * feed the result to dabx_set_subchannels to decode "everything found in the FIC" like EtiGenerator does. */
int  dabx_discover_subchannels(dabx_engine *e, int stream, dabx_subch_desc *out, int max_out);
/* Multiplex reconfiguration (EN 300 401 6.4.1, 6.5; FibDecoder's current / next tables).  The engine keeps one dabx_fibdec per
 * stream; dabx_follow_fic feeds it the FIBs of the frames decoded since the last call (in order; frames that have already left
 * the FIB ring of out_frames frames are reported in frames_missed) and says where the stream stands:
 *   pending      the newest FIG 0/0 carries change flags != 0: a reconfiguration is announced;
 *   at_cif       ... and takes effect from this CIF on, counted like the engine counts CIFs (4 per frame from the first frame
 *                decoded: CIF 4 f + k is the k-th of frame f) -- the CIF whose counter's low byte equals OccurrenceChange, at or
 *                after the CIF that carried the announcement;
 *   n_changes / last_change_cif   swaps the decoder has made (change flags 3 -> 0) and the engine CIF of the FIB group that made the
 *                newest one, i.e. the first CIF of the new configuration as the reference sees it.
 * dabx_next_subchannels returns the announced (next) table.  To follow a reconfiguration: call dabx_process up to the frame that
 * holds at_cif (frames = at_cif / 4 - frames decoded), then dabx_set_subchannels_at(..., at_cif) with the table wanted from then
 * on, and go on -- slots whose description does not change keep running, new ones start their 16-CIF de-interleaver fill at at_cif,
 * slots that end stop with the call: their last logical frame is that of the last CIF of the frame decoded before it, also when
 * at_cif lies inside the coming frame (tests/test_gpu_reconfig.py, cif_in_frame = 2). */
typedef struct {
  int32_t pending, n_changes, frames_missed, reserved;
  int64_t at_cif, last_change_cif, frames_fed;
} dabx_reconf;
int  dabx_follow_fic(dabx_engine *e, int stream, dabx_reconf *out);
/* The FIB decoders behind dabx_follow_fic / dabx_next_subchannels / dabx_current_subchannels swap their tables after ANY announcement by
 * default; on != 0 makes them (those that exist and those created later) follow the reference's rule to the bit -- only after change flags 3
 * (fib_decoder_fig0.cpp:103), dabx_fibdec_set_reference_quirks -- so that an engine-level reconfiguration can be compared with the reference's
 * own behaviour for flags 1 and 2 as well. */
int  dabx_set_fig_reference_quirks(dabx_engine *e, int on);
/* OfdmDecoder's LCD record (SLcdData, ofdm_decoder.h:53-61) carries, next to the SNR the engine always estimates (dabx_stats.snr_db_est), the MER:
 * 10 log10((pi/4)^2 / mean_k mStdDevSqPhaseVector[k]), a per-carrier IIR of the squared phase distance from the constellation's diagonal
 * (ofdm_decoder.cpp:204-208, 331-340) that feeds no soft bit.  on != 0 advances that IIR in the demapper too (a few instructions per carrier and
 * symbol; off by default: a display statistic of one receiver, not of 512) and dabx_stats.mer_db_est / dabx_chunk_stream.mer_db_est report it
 * after every frame; switched on in mid-stream the IIR starts from what it last held (zero after a reset / loss of lock, like the reference's).
 * Drains the engine. */
int  dabx_set_lcd_statistics(dabx_engine *e, int on);
int  dabx_next_subchannels(dabx_engine *e, int stream, dabx_subch_desc *out, int max_out);
/* ... and the CURRENT table of the same decoder (after a swap: the former next table plus whatever the new configuration's own FIGs,
 * C/N = 0, have added since -- first description wins). */
int  dabx_current_subchannels(dabx_engine *e, int stream, dabx_subch_desc *out, int max_out);
/* dabx_set_subchannels whose new and changed slots start at CIF at_cif instead of at the next CIF to be demodulated:
 * next CIF <= at_cif <= next CIF + 3 (a reconfiguration inside the coming frame).  stream >= 0. */
int  dabx_set_subchannels_at(dabx_engine *e, int stream, const dabx_subch_desc *d, int n, int64_t at_cif);
/* Decoded logical frames of a sub-channel: n_cifs x 3*kbps bytes, newest last; returns #CIFs valid. */
int  dabx_read_msc(dabx_engine *e, int stream, int subch_idx, int n_cifs, uint8_t *bytes);
/* RS-corrected DAB+ super frames (110*kbps/8 bytes each), newest last; returns count copied. */
int  dabx_read_superframes(dabx_engine *e, int stream, int subch_idx, int n, uint8_t *bytes);
/* What Mp4Processor::_process_super_frame (base/backend/audio/mp4processor.cpp:249-333) knows about a super frame when it hands the
 * access units to the AAC decoder: the stream parameters (:258-262), numAUs and mAuStartArr (:272-304), the verdict of the length check
 * (:311) and of check_crc_bytes (:321) for every access unit, and what the RS decoder / the fire code corrected on the way
 * (:184-241).  One record per super frame in the super-frame ring, written by the device stage that ran all of it (k_dabplus): a host
 * that feeds an AAC decoder slices the super frame at au_start[] and looks at the masks -- it re-parses no header and re-runs no CRC:
 *     for (a = 0; a < r.num_aus; a++)
 *       if (r.au_crc_ok >> a & 1) decoder.decode(sf + r.au_start[a], r.au_start[a + 1] - r.au_start[a] - 2);   // the 2 CRC bytes excluded, :306
 *       else                      decoder.conceal();                                                         // :316, :339
 * 32 bytes, little-endian, no padding holes. */
typedef struct dabx_superframe_info_s {
  uint8_t  num_aus;        /* 2, 3, 4 or 6 */
  uint8_t  au_crc_ok;      /* bit a: access unit a passed its CRC */
  uint8_t  au_len_bad;     /* bit a: access unit a failed the length check (aacFrameLen > 960, < 0, or beyond the super frame); no CRC run */
  uint8_t  stream_parms;   /* mOutVec[2] & 0x7F: dacRate 0x40, sbrFlag 0x20, aacChannelMode 0x10, psFlag 0x08, mpegSurround 0x07 */
  uint16_t au_start[7];    /* mAuStartArr[0 .. num_aus]; au_start[num_aus] = 110 * kbps / 8 */
  uint16_t rs_corrected;   /* byte corrections of the RS decoder in this super frame (sum of its returns >= 0) */
  uint8_t  rs_failed;      /* code words the RS decoder gave up on (the fire code passed all the same) */
  uint8_t  fc_corrected;   /* 1: check_and_correct_6bits changed the 11-byte header (mSumFcCorrections) */
  uint16_t reserved;
  int64_t  first_frame;    /* index, in the slot's sequence of logical frames, of the first of the super frame's five */
} dabx_superframe_info;
/* the records of the newest n super frames, oldest first: row i describes row i of dabx_read_superframes(..., n, ...) */
int  dabx_read_superframe_info(dabx_engine *e, int stream, int subch_idx, int n, dabx_superframe_info *out);
int  dabx_read_soft(dabx_engine *e, int stream, int16_t *soft /* 75*3072 */);
/* dabx_get_stats_sized fills the first `size` bytes of a dabx_stats.  dabx_get_stats is that call with the caller's own
 * sizeof -- as a macro, so that a binary and the library never disagree about how much is written; the exported function of the
 * same name is what binaries built against ABI 3 call and writes the ABI-3 record only.  Hosts that load the library at run time
 * compare dabx_abi_version() with the DABX_ABI_VERSION they were built against (shim/dabx_shim_env.h does). */
int  dabx_get_stats_sized(dabx_engine *e, int stream, void *out, size_t size);
int  dabx_get_stats(dabx_engine *e, int stream, dabx_stats *out);
#define dabx_get_stats(e, stream, out) dabx_get_stats_sized((e), (stream), (out), sizeof(*(out)))
/* Per-slot counters (Backend / Mp4Processor members: backend.cpp:146-150 warm-up, mp4processor.h:106-112 signals). */
typedef struct dabx_subch_stats_s {
  int64_t start_cif;         /* CIF index (since open) at which the slot was configured */
  int64_t cifs_decoded;      /* logical frames produced so far; frame i belongs to CIF start_cif + 16 + i */
  int64_t sf_count;          /* super frames written to the super-frame ring */
  int64_t sf_ok, sf_fail, rs_corrected, rs_failed, fc_corrected, au_ok, au_bad;
  int32_t active, subch_id;
} dabx_subch_stats;
int  dabx_get_subch_stats(dabx_engine *e, int stream, int subch_idx, dabx_subch_stats *out);
/* Sum of the counters over all streams of this engine (the values one RCCL all-reduce combines). */
int  dabx_get_counters(dabx_engine *e, int64_t out[16]);
/* ------------------------------------------------------------------------------------------------------------
 * Bulk delivery of the results to the host.  The reference hands every FIB to IFibDecoder::process_FIB
 * (base/decoder/fib_decoder_if.h:81, called from fic_decoder.cpp:234-261) and every logical frame to
 * FrameProcessor::add_to_frame (base/backend/frame_processor.h:43-46, called from backend.cpp:160) the moment it exists;
 * Mp4Processor hands on the RS-corrected super frame (mp4processor.cpp:149-158).  For n_streams ensembles the engine does the
 * same in bulk: with a delivery open, everything a CHUNK of frames produced -- a chunk is what one MSC batch decodes, at most
 * DABX_CHUNK_FRAMES frames of every stream; a dabx_process call closes its last chunk -- is gathered on the device into ONE
 * contiguous slab and copied with ONE asynchronous DMA (an SDMA engine) into a page-locked host slab, next to the decode of the following chunk.
 * No call of the receiver waits for it, nothing is copied per stream or per CIF.  dabx_read_fibs / _msc / _superframes /
 * _eti remain as the convenience form for single streams (they drain the engine and copy from the device rings).
 *
 * A slab starts with a dabx_chunk_header; all dabx_chunk_* offsets are bytes from the slab's first byte:
 *   stream table  dabx_chunk_stream[n_streams]              which frames of the stream the chunk holds + the stream's scalars
 *   slot table    dabx_chunk_subch[n_streams * max_subch]   which logical / super frames of the sub-channel + its counters
 *   FIBs          [n_streams][max_frames][12][32] bytes, CRC flags [n_streams][max_frames][12], frame records
 *                 dabx_chunk_frame[n_streams][max_frames]   (row f = frame first_frame + f of the stream, f < n_frames)
 *   logical frames of slot (s, j): n_cifs x 3 * kbps bytes from msc_off -- the bytes dabx_read_msc returns, oldest first
 *   super frames   of slot (s, j): n_sf rows of sf_pitch bytes (110 * kbps / 8 used) from sf_off -- dabx_read_superframes' bytes;
 *                  their n_sf dabx_superframe_info records (AU table, per-AU CRC verdicts, corrections) from sfi_off
 * Chunks are numbered from 0 and delivered in order.  dabx_process fails with DABX_E_STATE, before it has started anything,
 * when the call would close more chunks than there are free host slabs: a consumer that falls behind holds the receiver up,
 * it never loses data silently.  dabx_delivery_open refuses an engine whose FIB ring is shorter than a chunk (dabx_config.out_frames
 * < DABX_CHUNK_FRAMES) when FIBs are to be delivered; frames_lost / cifs_lost / sf_lost count what a device ring could not hold until
 * the chunk was packed all the same (always 0 so far).
 * Threads: dabx_delivery_next / dabx_delivery_release may be called from ONE consumer thread next to the thread that drives the
 * engine (dabx_push_iq*, dabx_process, ...); everything else keeps the one-thread-per-handle rule. */
#define DABX_CHUNK_FRAMES 7
#define DABX_CHUNK_MAGIC 0x43584244u       /* "DBXC" */
enum { DABX_DELIVER_FIB = 1, DABX_DELIVER_MSC = 2, DABX_DELIVER_SF = 4,
       /* logical frames only of the slots that are NOT DAB+ (instead of DABX_DELIVER_MSC): for a DAB+ service the logical frames' consumer,
          Mp4Processor, runs on the device and the host's input is the super frame -- FIB | SF | this = what a receiver's host side needs,
          half the bytes of "everything" for a DAB+ multiplex */
       DABX_DELIVER_MSC_NOT_DABPLUS = 8 };
typedef struct {
  int32_t host_slabs;       /* page-locked host slabs, >= 2 (0 = default 4) */
  int32_t what;             /* DABX_DELIVER_* mask, 0 = everything */
  int32_t copy_engine;      /* 0 (default): the slab goes to the host on an SDMA engine (HSA runtime, hsa_amd_memory_async_copy): no CU is
                               involved, the receiver's kernels do not notice it.  1: hipMemcpyAsync -- on ROCm 7.2 a shader copy that holds up
                               every kernel running next to it for its duration (kept for comparison: tools/copy_interference.hip) */
  int32_t reserved[5];
} dabx_delivery_config;
typedef struct {
  uint32_t magic, abi;      /* DABX_CHUNK_MAGIC, DABX_ABI_VERSION */
  uint64_t seq;             /* chunk number */
  int32_t  n_streams, max_subch, max_frames /* DABX_CHUNK_FRAMES */, what;
  uint64_t bytes;           /* size of the slab as copied */
  uint64_t off_stream, off_subch, off_fib, off_crc, off_frame, off_msc, off_sf;
  uint64_t reserved[4];
} dabx_chunk_header;        /* 128 bytes */
typedef struct {
  int64_t first_frame;      /* index, since the stream was opened, of the first frame in the chunk */
  int32_t n_frames;         /* frames of this stream in the chunk, 0..max_frames (a stream out of lock delivers none) */
  int32_t frames_lost;      /* frames decoded since the previous chunk that had left the FIB ring before this one was packed */
  int32_t state, fic_ratio_percent, cif_count;      /* as in dabx_stats, after the chunk's last frame */
  float   snr_db_est, freq_offs_bb_hz, clock_err_hz, signal_level;
  int32_t fic_ber_bits, fic_ber_errors;             /* FicDecoder's channel-BER counters (dabx_stats) */
  float   mer_db_est;                               /* dabx_stats.mer_db_est (0 unless dabx_set_lcd_statistics).  The two LCD statistics
                                                       (snr_db_est, mer_db_est) are what the demapper had last written when the record was
                                                       gathered: with 48 streams and more the chunk's last frame's MSC symbols may still be
                                                       in the demapper then (a HIP stream of its own), and they are that frame's or the
                                                       previous one's; dabx_get_stats after dabx_synchronize shows the last frame's */
  int64_t fib_ok, fib_total;                        /* cumulative */
} dabx_chunk_stream;        /* 72 bytes */
typedef struct {
  int64_t sym0_pos;         /* dabx_read_frame_info */
  int32_t start_index, reserved;
} dabx_chunk_frame;
typedef struct {
  int32_t active, subch_id, kbps, dab_plus;
  int64_t start_cif;        /* dabx_subch_stats.start_cif: logical frame i of the slot belongs to CIF start_cif + 16 + i */
  int64_t first_cif;        /* index of the chunk's first logical frame in the slot's sequence (0 = the slot's first) */
  int32_t n_cifs;           /* logical frames in the chunk */
  int32_t cifs_lost;
  int64_t first_sf;         /* index of the chunk's first super frame in the slot's sequence */
  int32_t n_sf, sf_lost;
  uint64_t msc_off, sf_off;
  int32_t sf_pitch, reserved;
  int64_t sf_ok, sf_fail, rs_corrected, rs_failed, fc_corrected, au_ok, au_bad;      /* cumulative, as dabx_subch_stats */
  uint64_t sfi_off;         /* n_sf dabx_superframe_info records, row i for super-frame row i (ABI 6) */
} dabx_chunk_subch;         /* 144 bytes */
typedef struct {
  uint64_t seq;
  const void *data;         /* the host slab: valid until dabx_delivery_release(seq) */
  uint64_t bytes;
} dabx_chunk;
int  dabx_delivery_open(dabx_engine *e, const dabx_delivery_config *cfg /* NULL = defaults */);
int  dabx_delivery_close(dabx_engine *e);      /* drains the engine; chunks not yet fetched are dropped.  The consumer thread must have
                                                  left dabx_delivery_next before this (and before dabx_destroy) is called */
/* The oldest chunk not yet handed out: returns 1 and fills *out when it is complete in host memory, 0 when no chunk is ready
 * (wait == 0) or none is queued at all (wait != 0 waits for a queued one to land). */
int  dabx_delivery_next(dabx_engine *e, int wait, dabx_chunk *out);
int  dabx_delivery_release(dabx_engine *e, uint64_t seq);
/* Back-pressure for the engine's thread: waits (at most timeout_ms, < 0 = for ever) until n host slabs are free and returns the
 * number that are (>= n: a dabx_process call that closes n chunks will be accepted; < n: timed out).  n above the number of host slabs
 * the delivery was opened with is an error (it would wait for ever). */
int  dabx_delivery_wait_free(dabx_engine *e, int n, int timeout_ms);
/* What the copies themselves took (the copier's own clock around each transfer): link rate = bytes_copied / copy_seconds. */
typedef struct {
  uint64_t chunks_closed, chunks_landed, bytes_copied;
  double   copy_seconds, copy_seconds_max;     /* sum / longest single transfer */
  double   gather_wait_seconds;                /* the copier's waits for the chunks' gather kernels (idle time, not a cost) */
  int32_t  copy_engine;                        /* dabx_delivery_config.copy_engine */
  uint32_t sdma_engine_mask;                   /* hsa_amd_sdma_engine_id_t the transfers are put on (0: the runtime's choice) */
  double   calibration_GBps;                   /* rate of the 16-MiB probe transfer dabx_delivery_open made on that engine (an engine below
                                                  35 GB/s is replaced by the fastest of engines 0..7) */
  uint64_t reserved[3];
} dabx_delivery_info;
int  dabx_delivery_get_info(dabx_engine *e, dabx_delivery_info *out);
/* Bytes one slab takes with the sub-channels configured now (what one chunk moves over the link). */
long long dabx_delivery_slab_bytes(dabx_engine *e);

/* ------------------------------------------------------------------------------------------------------------
 * Bulk ingest: the delivery's mirror image for the samples.  dabx_push_iq* hands over one stream's samples per call -- right for a
 * device front end, 512 calls per chunk for a host that feeds 512 recordings.  With an ingest open the host fills ONE page-locked slab
 * with the next n samples of EVERY stream ([n_streams][n] samples of fmt, stream after stream), ONE SDMA transfer moves it, one kernel
 * converts it into all rings and one commit makes it readable:
 *     fill slab k;  dabx_ingest_submit(e, k, n);            the transfer starts, the call returns
 *     dabx_ingest_commit(e, k');                            waits for the transfer of slab k' (the previous one), converts, commits
 *     dabx_process(e, frames, 0);                           decodes it while slab k is still on the link
 * The raw_reader.cpp:66-70 / wav_reader.cpp:164 sample maps (fmt 2 / 1) run on the device: only 2 / 4 bytes per sample cross the link. */
typedef struct {
  int32_t host_slabs;       /* page-locked input slabs, >= 1 (0 = default 2) */
  int32_t fmt;              /* 0 cf32, 1 int16 IQ, 2 uint8 IQ: as dabx_push_iq */
  int32_t max_frames;       /* samples per stream a slab holds, in frames of T_F (0 = default DABX_CHUNK_FRAMES) */
  int32_t copy_engine;      /* as dabx_delivery_config.copy_engine */
  int32_t reserved[4];
} dabx_ingest_config;
int  dabx_ingest_open(dabx_engine *e, const dabx_ingest_config *cfg);
/* The general form -- what a host with n_streams RECORDINGS has: every stream its own container, byte order, sample rate (the readers'
 * 1-ms linear interpolation of wav_reader.cpp:67-82,190-206 / xml_reader.cpp:237-244 runs on the device, per stream, with its state carried
 * from slab to slab) and its own length.  formats[s] = the dabx_iq_format of stream s as dabx_probe_iq_file returns it (data_offset /
 * data_bytes are ignored; cfg->fmt too).  A slab is n_streams regions of dabx_ingest_pitch() bytes; the host puts the next payload bytes
 * of stream s at byte s * pitch and says how many (n_bytes[s]: whole samples -- a reader keeps an odd tail for its next slab --, 0 = the
 * recording has ended or is not ready).  Still ONE SDMA transfer per slab, TWO kernel launches per commit whatever the number of streams
 * and formats, and the results are byte-identical to n_streams dabx_feed_bytes calls (tests/test_gpu_ingest.py). */
struct dabx_iq_format_s;                                                      /* "Recorded-IQ files" below */
int  dabx_ingest_open_formats(dabx_engine *e, const dabx_ingest_config *cfg, const struct dabx_iq_format_s *formats /* [n_streams] */);
long long dabx_ingest_pitch(dabx_engine *e);                                  /* bytes per stream region of a slab */
int  dabx_ingest_submit_bytes(dabx_engine *e, int k, const size_t *n_bytes /* [n_streams] */);   /* then dabx_ingest_commit(e, k) as above */
int  dabx_ingest_close(dabx_engine *e);
int  dabx_ingest_slab(dabx_engine *e, int k, void **host, size_t *capacity_bytes);
int  dabx_ingest_submit(dabx_engine *e, int k, size_t n_samples);     /* DABX_E_STATE: the slab's previous transfer has not been committed */
int  dabx_ingest_commit(dabx_engine *e, int k);                        /* DABX_E_STATE: a ring cannot take the samples (dabx_process first) */

/* Per-kernel timing with HIP events recorded on the engine's stream around every launch of a batch step
 * (bench.py's roofline leg).  dabx_get_profile drains the events recorded since the last call: for each of
 * the n kernels of a step it returns the accumulated milliseconds and the number of launches; names[i]
 * points to a static string.  Returns n.  dabx_set_profiling: 0 = off, 1 = every kernel as scheduled (kernels of the engine's
 * HIP streams overlap, so a duration includes what the kernel waited for the chip's other tenants; the event pairs
 * serialise neighbouring kernels a little: ~4 % on the 512-stream bench), -1 = every kernel with the host waiting for each
 * one (one kernel on the chip at a time: stand-alone durations; slow), 2 + i = only kernel i, as scheduled. */
#define DABX_MAX_KERNELS 16
int  dabx_set_profiling(dabx_engine *e, int on);
int  dabx_get_profile(dabx_engine *e, double total_ms[DABX_MAX_KERNELS], int64_t launches[DABX_MAX_KERNELS],
                      const char *names[DABX_MAX_KERNELS]);

/* ------------------------------------------------------------------------------------------------------------
 * Recorded-IQ files: the byte formats the reference's file readers accept, decoded and -- for
 * recordings that are not at 2.048 MS/s -- resampled ON THE GPU into a stream's IQ ring.  Only the raw bytes cross
 * PCIe (2..8 B per sample instead of 8).
 *   family RAW  .raw/.iq   uint8 IQ, (x - 127.38)/128                      devices/filereaders/raw_files/raw_reader.cpp:66-70,155-158
 *   family WAV  .sdr/.wav  RIFF/WAVE, 2 channels, 1.536..3.0 MS/s, normalised like libsndfile's sf_readf_float
 *                          (wav_files/wavfiles.cpp:55-92, wav_reader.cpp:164): u8 (x-128)/128, s8 x/128, i16 x/2^15,
 *                          i24 x/2^23, i32 (float)x/2^31, f32 as stored
 *   family UFF  .uff       XML header + payload (xml_filereader/xml_descriptor.cpp:98-240, xml_reader.cpp:254-398):
 *                          int8 x/127, uint8 (x-127.38)/128, int16/24/32 x/2^(Bits-1), float32; MSB|LSB; IQ|QI
 * Resampling follows the reference's non-liquid build: 1-ms blocks of rate/1000 input samples are linearly
 * interpolated to 2048 output samples (wav_reader.cpp:67-82,190-206; xml_reader.cpp:76-81,226-248; the two readers
 * differ in their table arithmetic and in how the first block is primed, selected by `family`).
 * By default four defects of the reference's UFF reader are NOT reproduced (each evidently a typo, not a format rule;
 * docs/history/r01-r04_design_notebook.md 9): QI/float32 is decoded as swapped IQ (the reference does not swap, xml_reader.cpp:530,540), int24/MSB takes
 * the middle byte of Q from its own sample (the reference reads lbuf[4*i+4] of its 1-ms read block, :316 and :462),
 * QI/int24/MSB sign-extends with 0xFF000000 (the reference ORs 0x7F000000, :465,:469), QI/uint8 is decoded from the data
 * (the reference indexes its 256-entry table with the loop counter, :423).  With dabx_iq_format.reference_quirks = 1 the
 * first three are reproduced bit for bit -- the bytes a user of the reference gets from such a file -- and QI/uint8 is
 * refused, because the reference's read runs past its table from the 128th sample of every block (undefined behaviour).
 * Single-channel (I-only / Q-only) UFF files are refused in both modes. */
enum { DABX_FAMILY_RAW = 0, DABX_FAMILY_WAV = 1, DABX_FAMILY_UFF = 2 };
enum { DABX_C_U8 = 0, DABX_C_S8 = 1, DABX_C_I16 = 2, DABX_C_I24 = 3, DABX_C_I32 = 4, DABX_C_F32 = 5 };
typedef struct dabx_iq_format_s {
  int32_t family;        /* DABX_FAMILY_* : normalisation rule + resampler flavour */
  int32_t container;     /* DABX_C_* */
  int32_t big_endian;    /* UFF Ordering="MSB", RIFX */
  int32_t swap_iq;       /* UFF channel order Q,I */
  int32_t bits;          /* UFF Bits attribute: integer scale is 2^(bits-1) */
  int32_t sample_rate;   /* Hz; 2048000 = no resampling */
  int64_t data_offset;   /* first payload byte in the file */
  int64_t data_bytes;    /* payload length (clipped to the file) */
  int32_t reference_quirks; /* 1: reproduce the reference's UFF reader defects bit for bit (see above); dabx_probe_iq_file sets 0 */
  int32_t reserved;
} dabx_iq_format;
typedef struct dabx_feed dabx_feed;

/* Host only: recognise the container by content (RIFF/RIFX magic, "<?xml"/"<SDR" header) else by extension
 * (.raw/.iq) and fill *fmt.  DABX_E_ARG for unreadable / unsupported files (dabx_last_error says why). */
int  dabx_probe_iq_file(const char *path, dabx_iq_format *fmt);
/* Bytes per complex sample of a format (2, 4, 6 or 8), or DABX_E_ARG. */
int  dabx_iq_sample_bytes(const dabx_iq_format *fmt);
/* One-shot conversion (stage level, fresh resampler state): payload bytes -> cf32 at 2.048 MS/s on the host.
 * Returns the number of complex samples written (<= max_out), or < 0. */
long long dabx_convert_iq_bytes(const dabx_iq_format *fmt, const void *bytes, size_t n_bytes, float *iq_out, size_t max_out);
/* Streaming feed into the IQ ring of `stream`: keeps the resampler state between calls; any split of the payload
 * into calls gives the same samples.  dabx_feed_bytes returns the number of 2.048 MS/s samples committed to the
 * ring, DABX_E_STATE when the ring cannot take them (process first), or another error. */
int  dabx_feed_open(dabx_engine *e, int stream, const dabx_iq_format *fmt, dabx_feed **out);
long long dabx_feed_bytes(dabx_feed *f, const void *bytes, size_t n_bytes);
/* Upper bound of the samples dabx_feed_bytes(n_bytes) can commit (ring-space planning). */
long long dabx_feed_bound(const dabx_feed *f, size_t n_bytes);
void dabx_feed_close(dabx_feed *f);

/* ------------------------------------------------------------------------------------------------------------
 * ETI(NI) output: the 6144-byte / 24-ms container EtiGenerator writes
 * (base/eti_handler/eti_generator.cpp:169-199 frame assembly, :207-308 header).
 * dabx_eti_frame is host only: cif_hi/cif_lo are FibDecoder::get_cif_count's values when the frame's FIC had been
 * parsed, minor the CIF's position 0..3 in the transmission frame, sc/msc the sub-channels in FIC order with their
 * 3*kbps logical-frame bytes, fic96 the three FIBs of this CIF.  Returns the bytes used before the 0x55 padding. */
#define DABX_ETI_FRAME_BYTES 6144
int  dabx_eti_frame(int cif_hi, int cif_lo, int minor, const dabx_subch_desc *sc, int n_subch, const uint8_t *fic96,
                    const uint8_t *const *msc, uint8_t *out /* 6144 */);
/* ETI frames of `stream` for the CIFs decoded since the previous call (at most max_frames, oldest first), assembled
 * from the engine's FIB and logical-frame rings for the configured sub-channels.  Like the reference the FIC of a
 * CIF is paired with the (time-de-interleaved) MSC data that completes with that CIF; frames start once every
 * sub-channel's de-interleaver is filled and FIG 0/0 has been seen.  *lost_cifs (optional) counts CIFs that had
 * already left the rings: call at least every min(out_frames, 8) processed frames.  Returns the frame count. */
int  dabx_read_eti(dabx_engine *e, int stream, int max_frames, uint8_t *out, int32_t *lost_cifs);

/* ------------------------------------------------------------------------------------------------------------
 * TII: TiiDetector, base/ofdm/tii_detector.h:26-37, .cpp:149-240.  Host only: the engine
 * accumulates the FFT of every TII null symbol ((CIF count & 7) >= 4, dab_processor.cpp:273-300) on the device;
 * dabx_read_tii hands the sum to the stream's detector once `min_frames` of them are in (DabProcessor's
 * tiiFramesToCount) and returns the transmitters found, strongest first. */
typedef struct {
  uint8_t main_id, sub_id;   /* STiiResult */
  float strength, phase_deg;
  int32_t non_etsi_phase;
} dabx_tii_result;
typedef struct dabx_tii dabx_tii;
int  dabx_tii_create(dabx_tii **out);
void dabx_tii_destroy(dabx_tii *t);
void dabx_tii_reset(dabx_tii *t);
void dabx_tii_set_collisions(dabx_tii *t, int on, int sub_id);       /* set_detect_collisions, set_subid_for_collision_search */
int  dabx_tii_add(dabx_tii *t, const float *null_fft /* 2048 cf32 */);   /* add_to_tii_buffer */
int  dabx_tii_process(dabx_tii *t, int threshold_db, dabx_tii_result *out, int max_out);   /* process_tii_data */
/* Returns the number of results (0 while fewer than min_frames TII null symbols have been accumulated);
 * *frames_accumulated (optional) = the count before the call.  A loss of lock resets the detector like
 * dab_processor.cpp:150-152. */
int  dabx_read_tii(dabx_engine *e, int stream, int min_frames, int threshold_db, int collisions, int collision_sub_id,
                   dabx_tii_result *out, int max_out, int32_t *frames_accumulated);

#ifdef __cplusplus
}
#endif
#endif
