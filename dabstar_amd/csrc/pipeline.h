// pipeline.h -- device-resident state of the stream-batched receiver (engine.hip / pipeline.hip).
#pragma once
#include "dabx_internal.h"

namespace dabx {

// Cache hints of the streaming accesses: a set bit makes the access __builtin_nontemporal_load / _store (the `nt` modifier: the line is not kept
// in L2 beyond its use).  The defaults are the measured ones (same-box A/B over three sessions, profiles/r05_ab/ab10_nontemporal_hints.txt:
// +1.9 % together); the other bits stay selectable for A/B builds (tools/build_variant.sh -DDABX_VIT_NT=0 ...).
#ifndef DABX_VIT_NT
#define DABX_VIT_NT 3      // k_msc_vitT: 1 = survivor-decision stores, 2 = their chain-back loads (3.6 + 3.2 GB per launch, each touched once: +1.1 %);
#endif                     //             4 = the transposed input (slower: a dword row is read up to four times in a row)
#ifndef DABX_SYM_NT
#define DABX_SYM_NT 2      // k_symbols: 2 = spectra stores (read next by the demapper, 478 MB later); 1 = IQ loads (slower: the cyclic prefix is read twice)
#endif
#ifndef DABX_DEMAP_NT
#define DABX_DEMAP_NT 1    // demapper: 1 = spectra loads; 2 = ring stores (no effect)
#endif
#ifndef DABX_PREP_NT
#define DABX_PREP_NT 0     // k_msc_prep: 1 = ring reads (-2.4 %: 64-byte runs, the other half of the line follows), 2 = transposed stores (-0.8 %)
#endif
#ifndef DABX_MSC_BATCH               // experiment builds only (tools/build_variant.sh -DDABX_MSC_BATCH=n, bench.py --chunk n)
#define DABX_MSC_BATCH 7
#endif
constexpr int MSC_SLOTS = DABX_MSC_BATCH > 7 ? 64 : 32;   // ring of decoded logical frames per sub-channel (>= 4 x batch new + 4 of the previous super frame)
constexpr int MSC_BATCH_FRAMES = DABX_MSC_BATCH; // frames whose MSC CIFs are decoded together (the MSC has no feedback into the front end);
                                    // 7 x 4 CIFs x 18 sub-channels x 512 streams / 64 = 4032 waves = 3.94 per SIMD
constexpr int SF_SLOTS = 16;      // ring of RS-corrected super frames per sub-channel: two chunks' worth (a 7-frame MSC batch completes up to 6)
constexpr int ACQ_NEED = 20 * TU + 50 + (TF + 1) + (TN + 50 + 21) + 64;   // worst-case samples one acquisition pass may read
constexpr int FRAME_NEED = TF + 2 * TU;                                    // worst-case samples one in-lock frame may read

enum StreamState : int32_t { ST_INIT = 0, ST_WAIT_SYNC = 1, ST_EVAL_SYNC = 2 };

// Scalars of SampleReader (sample_reader.h:91-101), DabProcessor (dab_processor.h:129-138) and
// FicDecoder (fic_decoder.h:71-76), one record per stream.
struct StreamCtl {
  unsigned long long rd;          // absolute index of the next unread sample
  unsigned long long sym0_pos;    // absolute index of the T_u part of symbol 0 of the current frame
  long long cif_no;               // CIFs written to the time-deinterleaver ring so far
  long long msc_done_cif;         // CIFs already passed through the MSC decoder (<= cif_no, lag <= 16)
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
  int32_t fic_errors, fic_bits, fic_block;   // mFicErrors, mFicBits, mFicBlock (fic_decoder.h:73-75)
  int32_t fic_status_errors, fic_status_bits;  // ... as they stood when the 40th block was reached: what signal_fic_status reports (fic_decoder.cpp:203-205)
  float mer_db;                   // MER of the LCD record after the frame's last symbol (0 unless dabx_set_lcd_statistics)
  float snr_db;
  // counters (summed across streams / GPUs by dabx_get_counters)
  long long fib_ok, fib_total, sync_lost;
  float head_abs_a, head_abs_b;   // sum |x| over the T_u correlation window / the start_index samples read after it (level tracker)
  int32_t np_sel;                 // which noise-power buffer (DemapDev::null_power / null_power2) is current; k_frame_tail flips it
  int32_t pad_;
  long long level_margin;         // null-dip comparisons that fell within 1e-4 (relative) of their threshold (dabx_stats.level_margin_events)
  // The level tracker's ANCHOR (EngineDev::anchor_level, the default): s_level was exact -- lvl_anchor_S, the value the sample-serial
  // recurrence has -- before sample lvl_anchor_pos.  The search (k_acquire) keeps it at rd; the frame chain moves rd on and advances
  // s_level chunk-wise (k_frame_tail: valid before sample lvl_approx_pos).  When the stream comes back to the search, the samples
  // since the anchor are walked exactly, if they are all still in the ring (EngineDev::wr_horizon).
  unsigned long long lvl_anchor_pos, lvl_approx_pos;
  float lvl_anchor_S;
  int32_t pad3_;
  long long lvl_rewalks, lvl_unanchored;   // returns to the search that re-walked from the anchor / that had to start from the approximation
  long long lvl_healed;                    // ... that started two walks around a frame boundary's chunk-wise value and saw them merge (exact all the same)
  unsigned long long lvl_hist_pos[16];     // read position after each of the last 16 frames ...
  float lvl_hist_S[16];                    // ... and the chunk-wise level there (k_frame_tail)
};
constexpr int LVL_HIST = 16;

struct SubchDev {
  int32_t cu_start, cu_size, kbps, prot_level, short_form, dab_plus;
  int32_t nbits;                  // 24 * kbps
  int32_t active;
  int32_t fast_class, pad_;       // 1 + index of the lane-per-trellis class this slot belongs to, 0 = none (vit_t.hip)
  const uint16_t *map;            // depuncture map, device
  long long start_cif;            // cif_no when the sub-channel was configured (Backend construction)
  long long cif_out;              // logical frames decoded so far
  // Mp4Processor state (mp4processor.h)
  int32_t blocks_in_buf, sf_sync;
  long long sf_count;
  long long sf_ok, sf_fail, rs_corr, rs_fail, fc_corr, au_ok, au_bad;
  // A sub-channel that only MOVED to other capacity units (multiplex reconfiguration): it keeps running, and its time de-interleaver
  // reads the CIFs before move_cif at the old address.  Zero = never moved (logical frame r >= 16 reads CIFs >= 0).
  long long move_cif;
  int32_t prev_cu_start, pad2_;
};
#ifdef __HIPCC__
// of the 16 CIFs r - 16 + m (m = 0..15) that logical frame r is de-interleaved from, those with m < msc_move_thr lie before the move
__device__ __forceinline__ int msc_move_thr(const SubchDev &sc, long long r)
{
  const long long t = sc.move_cif - r + 16;
  return t <= 0 ? 0 : (t >= 16 ? 16 : (int)t);
}
#endif

// Per-batch snapshot of the CIF counters: the MSC kernels of a batch run on their own HIP stream while the front
// end already advances cif_no for the next frames.
struct BatchSnap { long long msc_done, cif_no; };
// What the demapper of the MSC symbols (second k_demap_frame launch, own HIP stream) needs from StreamCtl: taken by the first
// launch, because the frame tail and the next frame's head already rewrite those fields while it runs.
struct FrameSnap { long long cif0; float clock_err; int32_t frame_ok, np_sel, pad_; };

struct EngineDev {
  int32_t n_streams, max_subch, out_frames;
  int32_t ring_len;               // IQ ring capacity per stream in samples
  float threshold;
  int32_t strongest, fic_only, capture_soft;
  int32_t exact_level;            // 1: in lock, SampleReader's level IIR is run sample by sample too (cfg.exact_level_tracker = 1)
  int32_t anchor_level;           // 1 (cfg.exact_level_tracker = 0, default): chunk-wise in lock, re-walked exactly from the anchor when the search needs it
  unsigned long long *wr_horizon; // [S] host memory: one past the highest sample index a push has been ISSUED for (written before the copy
                                  //     starts); ~0 after a zero-copy commit (writes the library does not see).  Samples >= horizon - ring_len are intact
  unsigned long long *level_pos;  // [S] exact_level only: index of the first sample the level tracker has not seen yet (<= ctl.rd)
  int32_t *locked_count;          // streams in ST_EVAL_SYNC, in host memory the device updates (system-scope atomics on hand-over): dabx_process
                                  // looks at it without waiting for anything -- while NO stream is in lock there is nobody the search could hold up
  int32_t tie_mode;               // 1: Viterbi arithmetic of the reference's AVX2 / SSE2 builds (viterbi_core.h, vit_step_simd)
  int32_t msc_stride;             // bytes per logical-frame slot (3 * max kbps)
  int32_t sf_stride;              // bytes per super-frame slot (110 * max kbps / 8)
  int32_t vit_stride;             // decision-scratch words per trellis (max over FIC and all sub-channels)
  float2 *iq;                     // [S][ring_len]
  unsigned long long *wr;         // [S] absolute index one past the last committed sample
  StreamCtl *ctl;                 // [S]
  DemapDev demap;
  float2 *spectra;                // [2][S][75][1536]: symbols 1..75 in CARRIER order (frequency de-interleaved by k_symbols); two
                                  // buffers by step parity: k_symbols of step n + 1 runs while step n's MSC symbols are demapped
  FrameSnap *fsnap;               // [S]
  int32_t *demap_busy;            // [S] 1 while the demapper launches of the stream's newest frame have not all finished (they run on
                                  //     their own HIP stream next to the NEXT frame's head: k_frame_head must not reset the demapper under them)
  float *dciq_state;              // [S][8] meanI, meanQ, meanII, meanQQ, meanIQ of SampleReader's DC / IQ correction (sample_reader.h:102-106)
  unsigned long long *dciq_done;  // [S] absolute index of the first sample not yet corrected
  int32_t parity;                 // step parity (host sets it per launch)
  int32_t s0;                     // first stream of a launch that covers a GROUP of streams (k_symbols and the demapper; experiment builds with
                                  // -DDABX_GROUPS=n issue them per group so that a group's spectra stay in the 256-MB Infinity Cache; 0 otherwise)
  // Few streams (EngineStreams::fic_on_d): the two hand-overs between the frame chain (HIP stream a) and the demapper (d) are DEVICE-side
  // sequence numbers instead of HIP events -- an event record or wait between two kernels of a HIP stream is a 6-17 us bubble in that stream
  // (profiles/r06_single_ensemble_timeline_after.txt), and the demapper's loop is what a lone ensemble's frame rate is.  k_sym_publish (stream a,
  // behind k_symbols) stores step_seq into sym_seq[s]; k_demap_fic waits for it, and stores it into fic_seq[s] when the FIC symbols are out;
  // k_fic_frame waits for that.  A waiting kernel only ever waits for one launched BEFORE it whose own waits are satisfied by still earlier
  // launches (no cycle), and with fewer than 48 streams every block of every kernel involved is resident at once (no block waits for a slot
  // held by a spinning one).  0: HIP events (the schedule of 48 and more streams).
  int32_t flag_sync;
  uint32_t step_seq;              // number of this step (1, 2, ...: the host sets it per launch)
  uint32_t *sym_seq, *fic_seq;    // [S]
  int32_t *seq_timeouts;          // host memory: waits that gave up after ~2 s (a launch in front of them must have failed): the next dabx_synchronize /
                                  // dabx_process(sync) reports it instead of the GPU hanging
  double2 *nco_tid;               // [S][256] e^{-j 2 pi f tid / fs} of the current frame (k_frame_head -> k_symbols)
  double2 *nco_sym;               // [S][76]  NCO phasor of the first FFT sample of symbols 1..75 ([75] = rotation per 256 samples)
  int32_t *sym_off;               // [S][76]  ring offset of the first (cyclic-prefix) sample of symbols 1..75 of this step's frame,
                                  //          -1 = the stream has no frame in this step (k_acquire / k_frame_head -> k_symbols)
  float2 *cp_part;                // [S][75] cyclic-prefix correlation partial sums
  float *abs_part;                // [S][76] sum |x| of the samples read per symbol (level tracking)
  uint8_t *fic_sym;               // [S][9216] Viterbi symbols of OFDM symbols 1..3
  uint8_t *tdi;                   // [S][TDI_SLOTS][55296] time-deinterleaver ring (Viterbi symbols)
  uint32_t *vit_scratch;          // decision words
  SubchDev *subch;                // [S][max_subch]
  uint8_t *fib_out;               // [S][out_frames][12][32]
  uint8_t *fib_crc;               // [S][out_frames][12]
  long long *frame_pos;           // [S][out_frames] absolute sample index of the T_u part of symbol 0 of the frame in that output slot
  int32_t *frame_start;           // [S][out_frames] its start index (PRS correlation peak, dab_processor.cpp:394)
  uint8_t *msc_out;               // [S][max_subch][MSC_SLOTS][msc_stride]
  uint8_t *sf_out;                // [S][max_subch][SF_SLOTS][sf_stride]
  dabx_superframe_info *sf_info;  // [S][max_subch][SF_SLOTS] AU table, per-AU CRC verdicts, corrections of the super frame in that slot (k_dabplus)
  int16_t *soft_cap;              // [S][75][3072] or null
  BatchSnap *snap;                // [S] counters of the MSC batch being decoded
  float2 *tii_acc;                // [S][2048] sum of the FFTs of the TII null symbols (TiiDetector::mNullSymbolBufferVec)
  int32_t *tii_cnt;               // [S][2] null symbols in the sum; detector-reset epoch (bumped on loss of lock)
};

// ---- bulk delivery (deliver.hip, include/dabx.h "Bulk delivery"): what the two gather kernels of a chunk are given, by value
struct DeliverDev {
  uint8_t *slab;                          // device slab of this chunk
  const unsigned long long *layout_off;   // [S * max_subch][3] offset of the slot's logical frames / super frames / super-frame records in a slab
  const int32_t *subch_id;                // [S * max_subch] SubChId (host knowledge: the device never needs it otherwise)
  long long *frames_done;                 // [S] frames of the stream delivered so far
  long long *cif_done, *sf_done;          // [S * max_subch] logical / super frames of the slot delivered so far
  dabx_chunk_header hdr;                  // as it goes into the slab
  // host side only: recorded behind k_deliver_lf (the chunk's logical frames are in the slab: their share of the transfer may start while the
  // DAB+ stage still runs); null = one transfer behind everything
  hipEvent_t lf_done;
};

// ---- lane-per-trellis path of the MSC decoder (vit_t.hip) ------------------------------------------------------
// The sub-channels of ALL streams are grouped into classes of equal protection profile (same depuncture map and
// trellis length): the 64 lanes of a decoder wave then share the map and step count whatever ensemble they come from.
constexpr int MSC_MAX_CLASSES = DABX_MSC_FAST_CLASSES;   // include/dabx.h
struct MscClass {
  int n_in, nbits;          // soft bits per job (cu_size*64), decoded bits (24*kbps)
  int n_pairs;              // (stream, slot) pairs in the class
  const uint16_t *map2;     // depuncture map with punctured entries remapped to n_in
  const uint32_t *pairs;    // [n_pairs] (stream << 8) | slot, sorted
  uint32_t *inT[2];         // [groups][n_in/4 + 1][64] transposed de-interleaved symbols, double-buffered per batch
  uint2 *decT;              // [groups][nbits + 6][64] decision words
};
struct MscFast {
  int n_cls;
  int min_jobs;             // below this many trellises per batch the wave-per-trellis kernel is used for everything
  int slots_active;         // active (stream, slot) pairs in total (> sum of n_pairs: the rest goes wave-per-trellis)
  MscClass cls[MSC_MAX_CLASSES];
};
// what one batch launches (by value): classes in launch order with their first decoder group
struct MscLaunchCls { int n_in, nbits, n_pairs, g0; const uint16_t *map2; const uint32_t *pairs; uint32_t *inT; uint2 *decT; };
struct MscLaunch { int n, groups; MscLaunchCls c[MSC_MAX_CLASSES]; };

// HIP streams/events of the engine: front end on `a`; the long lane-per-trellis decode of batch n runs on `b`
// while `a` already demodulates the frames of batch n+1.
struct EngineStreams {
  hipStream_t a = nullptr, b = nullptr, d = nullptr;   // d: demapper of the MSC symbols of the frame in flight; b, d null = serial schedule
  hipStream_t q = nullptr;                             // k_acquire of dabx_process(sync == 0): streams out of lock are searched next to the steps of the others
  hipEvent_t acq_done = nullptr, tail_done = nullptr;  // tail_done: the frame chain of the last step has moved the read cursors (exact level tracker)
  bool tail_recorded = false;
  hipEvent_t acq_a_done = nullptr;                     // the last pass that ran IN STEP on stream a: a pass on q must not start before it has finished
  bool acq_a_pending = false;
  bool acq_in_flight = false;                          // a pass on q may still be running
  hipEvent_t prep_done = nullptr, msc_done = nullptr, fic_go = nullptr, prep_b_done = nullptr, demap_done = nullptr;
  hipEvent_t sym_done = nullptr;
  // Few streams (one ensemble: BASELINE configs[1] / [2]): the frame rate is the length of a dependency loop, not a throughput.  The loop that
  // binds is the DEMAPPER's -- 75 symbols serial in one block per stream (per-carrier IIRs + the block-wide mean of every symbol), k_demap_fic (n + 1)
  // behind k_demap_frame6 (n) -- and with the two launches on two HIP streams every frame paid two event hops of 17-19 us on it
  // (profiles/r06_single_ensemble_timeline_before.txt: 7.5 + 19 + 88 + 17 = 132 us per frame).  fic_on_d puts both demapper launches on stream d,
  // back to back; the frame chain on a (head -> symbols -> | FIC decoder -> tail) forks off behind k_demap_fic and joins in front of the next one:
  // its 68 us + one hop run next to the 88 us of the MSC symbols.
  bool fic_on_d = false;
  bool demap_in_flight = false;   // stream d still demaps the MSC symbols of the previous step
  bool demap_unrecorded = false;  // ... and demap_done has not been recorded behind them yet (fic_on_d: recorded on demand)
  unsigned step_count = 0;
  bool prep_pending = false;      // k_msc_prep of the previous batch may still be reading the TDI ring on stream b
  bool msc_in_flight = false;
  int batch_parity = 0;
};

// ---- profiling hook: HIP events around every kernel launch of a batch step ------------------------------------
constexpr int N_STEP_KERNELS = 11;
struct Marker {
  bool on = false;
  bool serial = false;            // the host waits for every instrumented kernel: one kernel on the chip at a time = stand-alone durations
  int only = -1;                  // >= 0: instrument just this kernel (2 events per launch instead of 2 per kernel)
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Rec { int k; size_t a, b; };
  std::vector<Rec> recs;
  size_t open_ev[N_STEP_KERNELS] = {0};
  size_t take(hipStream_t st)
  {
    // device-scope release: these events only time kernels, nothing on the host reads device memory behind them
    if (used == pool.size()) { hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventReleaseToDevice); pool.push_back(ev); }
    (void)hipEventRecord(pool[used], st);
    return used++;
  }
  void begin(int k, hipStream_t st) { if (on && (only < 0 || only == k)) open_ev[k] = take(st); }
  void end(int k, hipStream_t st)
  {
    if (!on || !(only < 0 || only == k)) return;
    const size_t b = take(st);
    recs.push_back(Rec{k, open_ev[k], b});
    if (serial) (void)hipEventSynchronize(pool[b]);
  }
};

// ---- MSC job decoding shared by the decoder kernels: job J = (stream, pending CIF k, sub-channel j) -------------
struct MscJob {
  int s, k, j;
  long long r;          // CIF to output
  bool valid;
  long long out_idx;    // index of the logical frame in the sub-channel's output ring
};

#ifdef __HIPCC__
__device__ __forceinline__ MscJob msc_job(const EngineDev &e, int J, int cifs)
{
  MscJob q;
  const int per_stream = cifs * e.max_subch;
  q.s = J / per_stream;
  const int rem = J - q.s * per_stream;
  q.k = rem / e.max_subch;
  q.j = rem - q.k * e.max_subch;
  q.valid = false; q.r = 0; q.out_idx = 0;
  if (q.s >= e.n_streams) return q;
  const BatchSnap c = e.snap[q.s];
  const SubchDev &sc = e.subch[(size_t)q.s * e.max_subch + q.j];
  q.r = c.msc_done + q.k;
  const long long valid_from = sc.start_cif + 16;                 // de-interleaver filled, backend.cpp:146-150
  q.valid = sc.active && q.r < c.cif_no && q.r >= valid_from;
  q.out_idx = sc.cif_out + (q.r - (c.msc_done > valid_from ? c.msc_done : valid_from));
  return q;
}
// job J of a class: adjacent jobs = adjacent (stream, slot) pairs of one pending CIF
__device__ __forceinline__ MscJob msc_class_job(const EngineDev &e, const MscLaunchCls &c, int J, int cifs)
{
  MscJob q;
  q.k = J / c.n_pairs;
  const int pi = J - q.k * c.n_pairs;
  q.valid = false; q.r = 0; q.out_idx = 0; q.s = 0; q.j = 0;
  if (q.k >= cifs) return q;
  const uint32_t pr = c.pairs[pi];
  q.s = (int)(pr >> 8); q.j = (int)(pr & 255u);
  const BatchSnap b = e.snap[q.s];
  const SubchDev &sc = e.subch[(size_t)q.s * e.max_subch + q.j];
  q.r = b.msc_done + q.k;
  const long long valid_from = sc.start_cif + 16;
  q.valid = sc.active && q.r < b.cif_no && q.r >= valid_from;
  q.out_idx = sc.cif_out + (q.r - (b.msc_done > valid_from ? b.msc_done : valid_from));
  return q;
}
// TDI ring, planar: within a CIF slot soft bit i lives at plane (i & 15), position (i >> 4)
__device__ __forceinline__ size_t tdi_off(long long cif, int i)
{
  return (size_t)(cif & (TDI_SLOTS - 1)) * CIF_BITS + (size_t)(i & 15) * (CIF_BITS / 16) + (size_t)(i >> 4);
}
#endif

}  // namespace dabx
