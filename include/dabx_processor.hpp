// dabx_processor.hpp -- header-only C++17 adapter over the C ABI of libdabx with the method names of the reference's
// DabProcessor (base/main/dab_processor.h:71-110), so that a front end written against that class can be pointed at
// the MI355X back end (selection pattern: dab_processor.h:53-57 + a CMake option).
//
// No Qt: the reference's signals become std::function callbacks, QString becomes std::string, the QThread body
// (DabProcessor::run, dab_processor.cpp:110-265) becomes run(), which the caller invokes from its own thread
// whenever samples have been handed over.  One Processor = one ensemble = one stream of a one-stream engine; the
// stream-batched engine API (dabx_create with n_streams > 1) is the one to use for many ensembles at once.
#pragma once
#include <algorithm>
#include <complex>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#include "dabx.h"

namespace dabx {

struct ProcessorParams {                    // ProcessParams subset (base/main/dab_processor.h / dabradio.cpp:86-101)
  float threshold = 3.0f;          //   threshold
  int soft_bit_type = 1;           //   ESoftBitType (1..3)
  bool sync_on_strongest_peak = false;
  int ring_frames = 12;            // IQ ring capacity in transmission frames
  int max_services = 64;           // slots for concurrently decoded sub-channels
  int fib_ring_frames = 8;         // frames of FIBs kept for callbacks / ETI
  int tii_frames_to_count = 5;     //   tiiFramesToCount: TII null symbols summed before the detector runs
};

class Processor {
public:
  using Params = ProcessorParams;

  // signals of the reference, as callbacks (all optional; called from run())
  std::function<void(const uint8_t *fib32, bool crc_ok, int fib_no)> on_fib;                      // IFibDecoder::process_FIB
  std::function<void(int subChId, const uint8_t *bytes, int n)> on_logical_frame;                 // FrameProcessor::add_to_frame
  std::function<void(int subChId, const uint8_t *bytes, int n)> on_super_frame;                   // Mp4Processor: RS-corrected super frame
  // Mp4Processor::_process_super_frame's loop over the access units (mp4processor.cpp:306-345), with the verdicts the device stage already
  // reached: au = the AAC frame WITHOUT its two CRC bytes (what _build_aac_stream is given, :324), crc_ok false = the frame the reference
  // conceals (:316 length check, :339 CRC).  No header is parsed and no CRC is run on the host.
  std::function<void(int subChId, const uint8_t *au, int len, bool crc_ok, int au_idx, const dabx_superframe_info &sf)> on_access_unit;
  std::function<void(int ficPercent, float freqOffsBbHz, float clockErrHz, float snrDb)> on_status;   // slot_show_fic_status, ..._freq_corr_bb_Hz, ..._clock_error
  // OfdmDecoder::signal_show_lcd_data's device-side numbers (ofdm_decoder.cpp:326-345): setting it before the first run() switches the MER's
  // per-carrier IIR on in the engine (dabx_set_lcd_statistics)
  std::function<void(float snrDb, float merDb)> on_lcd_data;
  // signal_show_clock_err and signal_linear_peak_and_rms_level, "about each second" (dab_processor.cpp:252-263): the sample clock's error from
  // the samples the last more-than-ten frames took (INPUT_RATE * (samples / (frames * T_F) - 1): an average, not on_status's IIR), and
  // SampleReader's levels (get_linear_peak_level_and_clear; the engine tracks the peak in lock with cfg.exact_level_tracker = 1 only).
  // A loss of lock starts the count over.
  std::function<void(float clockErrHz)> on_clock_error;
  std::function<void(float peakLevel, float meanLevel)> on_signal_level;
  // signal_dip_sync_found (a null symbol's end was found: the stream went from the search into a frame) / signal_no_dip_sync_found (eight
  // frame lengths searched without one, dab_processor.cpp:418-433)
  std::function<void(bool found)> on_time_sync;

  std::function<void(const std::vector<dabx_tii_result> &)> on_tii;                                 // signal_show_tii
  // IFibDecoder::signal_change_in_configuration (fib_decoder_fig0.cpp:109) -- which the reference answers with "not supported yet"
  // (dabradio.cpp:271-274).  Here the running services are carried over to the announced configuration at its first CIF before the
  // callback fires: a service described as before runs through, also at other capacity units; one whose sub-channel changes size or
  // protection restarts its de-interleaver there; one that is no longer announced stops.  first_cif counts the CIFs this processor has demodulated (4 per frame).
  std::function<void(long long first_cif)> on_configuration_change;
  bool follow_reconfigurations = true;

  explicit Processor(const Params &p = Params()) : params_(p)
  {
    dabx_config cfg;
    dabx_default_config(&cfg);
    cfg.n_streams = 1; cfg.ring_frames = p.ring_frames; cfg.max_subch = p.max_services; cfg.out_frames = p.fib_ring_frames;
    cfg.sync_threshold = p.threshold; cfg.soft_bit_type = p.soft_bit_type; cfg.sync_strongest = p.sync_on_strongest_peak ? 1 : 0;
    if (dabx_create(&cfg, &eng_) < 0) throw std::runtime_error(std::string("dabx_create: ") + dabx_last_error());
    slots_.assign((size_t)p.max_services, dabx_subch_desc{});
    delivered_.assign((size_t)p.max_services, 0);
    sf_delivered_.assign((size_t)p.max_services, 0);
  }
  ~Processor()
  {
    stop_eti_generator();
    if (feed_) dabx_feed_close(feed_);
    if (eng_) dabx_destroy(eng_);
  }
  Processor(const Processor &) = delete;
  Processor &operator=(const Processor &) = delete;

  // ---- DabProcessor::start / stop -------------------------------------------------------------------------------
  void start() { running_ = true; }
  void stop() { running_ = false; }
  bool is_sample_reader_running() const { return running_; }

  // ---- sample hand-over: stands for IDeviceHandler::getSamples (common/device_handler_if.h:47-48) ----------------
  void put_samples(const std::complex<float> *iq, size_t n) { check(dabx_push_iq(eng_, 0, iq, 0, n), "dabx_push_iq"); }
  // recorded files (.raw/.iq, .sdr/.wav, .uff): bytes are decoded / resampled on the GPU
  dabx_iq_format open_recording(const std::string &path)
  {
    dabx_iq_format fmt;
    check(dabx_probe_iq_file(path.c_str(), &fmt), "dabx_probe_iq_file");
    if (feed_) { dabx_feed_close(feed_); feed_ = nullptr; }
    check(dabx_feed_open(eng_, 0, &fmt, &feed_), "dabx_feed_open");
    return fmt;
  }
  long long put_file_bytes(const void *bytes, size_t n)
  {
    if (!feed_) throw std::logic_error("open_recording first");
    const long long k = dabx_feed_bytes(feed_, bytes, n);
    if (k < 0) throw std::runtime_error(std::string("dabx_feed_bytes: ") + dabx_last_error());
    return k;
  }

  // ---- the body of DabProcessor::run for the frames that are available ---------------------------------------------
  // returns the number of frames demodulated; fires the callbacks and appends to the ETI file
  int run(int max_frames = 4)
  {
    if (!running_) return 0;
    dabx_stats before, after;
    check(dabx_get_stats(eng_, 0, &before), "dabx_get_stats");
    if (follow_reconfigurations) {
      // FIG 0/0 announces a change seconds ahead (change flags, OccurrenceChange): never demodulate past the frame boundary in front of
      // its first CIF without having switched the tables there
      dabx_reconf rc;
      check(dabx_follow_fic(eng_, 0, &rc), "dabx_follow_fic");
      if (rc.pending && rc.at_cif != applied_cif_) {
        const long long frame = rc.at_cif / 4;
        if (before.frames < frame) max_frames = (int)std::min<long long>(max_frames, frame - before.frames);
        else apply_next_configuration(before.frames == frame ? rc.at_cif : -1);     // (-1: the announcement was seen too late: switch now)
        if (before.frames >= frame) applied_cif_ = rc.at_cif;
      }
      if (verify_frame_ >= 0 && before.frames >= verify_frame_) stop_services_that_ended();
    }
    if (on_lcd_data && !lcd_on_) { check(dabx_set_lcd_statistics(eng_, 1), "dabx_set_lcd_statistics"); lcd_on_ = true; }
    check(dabx_process(eng_, max_frames, 1), "dabx_process");
    check(dabx_get_stats(eng_, 0, &after), "dabx_get_stats");
    const int frames = (int)(after.frames - before.frames);
    if (frames > 0) {
      deliver_fibs(frames);
      deliver_services();
      if (tii_on_ && on_tii) deliver_tii();
      if (eti_ && any_service()) write_eti();     // like EtiGenerator: nothing before the FIC has named the sub-channels
      if (on_status) on_status(after.fic_ratio_percent, after.freq_offs_bb_hz, after.clock_err_hz, after.snr_db_est);
      if (on_lcd_data && lcd_on_) on_lcd_data(after.snr_db_est, after.mer_db_est);
    }
    status_signals(before, after, frames);
    return frames;
  }

private:
  void status_signals(const dabx_stats &before, const dabx_stats &after, int frames)
  {
    const bool was_in = before.state == 2, is_in = after.state == 2;
    if (!was_in && is_in && on_time_sync) on_time_sync(true);
    if (!is_in) {
      // TimeSyncer reports NO_DIP_FOUND after a frame length without a dip (timesyncer.cpp:63-67); eight in a row raise the signal
      searched_ += after.samples_consumed - before.samples_consumed;
      if (searched_ >= 8LL * 196608) { searched_ = 0; if (on_time_sync) on_time_sync(false); }
    } else searched_ = 0;
    if (!(was_in && is_in) || frames <= 0) { clk_frames_ = 0; clk_samples_ = 0; return; }
    clk_frames_ += frames; clk_samples_ += after.samples_consumed - before.samples_consumed;
    if (clk_frames_ > 10) {
      if (on_clock_error) on_clock_error(2048000.0f * ((float)clk_samples_ / ((float)clk_frames_ * 196608.0f) - 1.0f));
      if (on_signal_level) on_signal_level(after.peak_level, after.signal_level);
      clk_frames_ = 0; clk_samples_ = 0;
    }
  }
  long long clk_frames_ = 0, clk_samples_ = 0, searched_ = 0;
public:

  // ---- FIB decoder getters (IFibDecoder subset: fib_decoder.cpp:547-570, 673-691) ------------------------------------
  std::vector<dabx_subch_desc> get_sub_channels()
  {
    std::vector<dabx_subch_desc> v(64);
    const int n = dabx_discover_subchannels(eng_, 0, v.data(), 64);
    check(n, "dabx_discover_subchannels");
    v.resize((size_t)n);
    return v;
  }
  int get_fic_decode_ratio_percent()
  {
    dabx_stats st;
    check(dabx_get_stats(eng_, 0, &st), "dabx_get_stats");
    return st.fic_ratio_percent;
  }

  // ---- MscHandler interface of DabProcessor (dab_processor.h:95-101) -------------------------------------------------
  bool set_audio_channel(int subChId)          // SAudioData: looked up in the FIC like DabRadio does before calling
  {
    for (const auto &d : get_sub_channels())
      if (d.subch_id == subChId) return set_channel(d);
    return false;
  }
  bool set_channel(dabx_subch_desc d)
  {
    // FIG 0/2 has not classified the component yet: treat it as DAB+ only if the bit rate can be one (multiples of
    // 8 kbit/s up to 384); the FIC is over-the-air data and may announce anything
    if (d.dab_plus < 0) d.dab_plus = (d.kbps > 0 && d.kbps <= 384 && d.kbps % 8 == 0) ? 1 : 0;
    int free_slot = -1;
    for (size_t j = 0; j < slots_.size(); j++) {
      if (slots_[j].kbps && slots_[j].subch_id == d.subch_id) return true;        // already running (msc_handler.cpp:100-108)
      if (!slots_[j].kbps && free_slot < 0) free_slot = (int)j;
    }
    if (free_slot < 0) return false;
    slots_[(size_t)free_slot] = d;
    delivered_[(size_t)free_slot] = 0; sf_delivered_[(size_t)free_slot] = 0;
    return apply();
  }
  void stop_service(int subChId)
  {
    for (auto &s : slots_) if (s.kbps && s.subch_id == subChId) s = dabx_subch_desc{};
    apply();
  }
  void stop_all_services()
  {
    for (auto &s : slots_) s = dabx_subch_desc{};
    apply();
  }
  bool is_service_running(int subChId) const
  {
    for (const auto &s : slots_) if (s.kbps && s.subch_id == subChId) return true;
    return false;
  }
  // "decode everything the FIC announces" -- what EtiGenerator does on its own (eti_generator.cpp:132-134)
  int set_all_channels()
  {
    int n = 0;
    for (const auto &d : get_sub_channels()) n += set_channel(d) ? 1 : 0;
    return n;
  }

  // ---- TII (dab_processor.h:106-109) ---------------------------------------------------------------------------------
  void set_tii_processing(bool on) { tii_on_ = on; }
  void set_tii_threshold(uint8_t db) { tii_threshold_ = db; }
  void set_tii_sub_id(uint8_t sub_id) { tii_sub_id_ = sub_id; }
  void set_tii_collisions(bool on) { tii_collisions_ = on; }

  // ---- ETI (dab_processor.h:82-84) ----------------------------------------------------------------------------------
  bool start_eti_generator(const std::string &path)
  {
    stop_eti_generator();
    eti_ = std::fopen(path.c_str(), "wb");
    return eti_ != nullptr;
  }
  void stop_eti_generator()
  {
    if (eti_) std::fclose(eti_);
    eti_ = nullptr;
  }
  void reset_eti_generator() {}                // the engine re-primes by itself after a configuration change
  long long eti_frames_written() const { return eti_frames_; }

  dabx_engine *engine() { return eng_; }

private:
  Params params_;
  dabx_engine *eng_ = nullptr;
  dabx_feed *feed_ = nullptr;
  bool running_ = false;
  std::vector<dabx_subch_desc> slots_;
  std::vector<long long> delivered_, sf_delivered_;
  std::FILE *eti_ = nullptr;
  long long eti_frames_ = 0;
  bool tii_on_ = false, tii_collisions_ = false, lcd_on_ = false;
  int tii_threshold_ = 6, tii_sub_id_ = 0;
  long long applied_cif_ = -1;                  // first CIF of the newest configuration that has been switched to
  long long verify_frame_ = -1;                 // frame at which services kept through a switch without being listed are checked against the new table

  // carries the running services over to the NEXT configuration (FIG 0/1 and 0/2 with C/N = 1) from CIF at_cif on
  void apply_next_configuration(long long at_cif)
  {
    std::vector<dabx_subch_desc> next(64);
    const int n_next = dabx_next_subchannels(eng_, 0, next.data(), 64);
    if (n_next <= 0) return;                     // nothing announced (yet): keep what runs
    int n = 0;
    for (size_t j = 0; j < slots_.size(); j++) {
      if (!slots_[j].kbps) continue;
      const dabx_subch_desc old = slots_[j];
      dabx_subch_desc now{};
      bool listed = false, displaced = false;
      for (int k = 0; k < n_next; k++) {
        const dabx_subch_desc &q = next[(size_t)k];
        if (q.subch_id == old.subch_id) { now = q; listed = true; }
        else if (q.cu_start < old.cu_start + old.cu_size && old.cu_start < q.cu_start + q.cu_size) displaced = true;
      }
      // A running service stops only when the next table says so positively: its capacity units go to another sub-channel.  One that
      // the table simply does not list (FIBs lost before the switch, or a multiplexer that announces only what changes) keeps running.
      if (!listed && !displaced) { now = old; verify_frame_ = std::max<long long>(0, at_cif) / 4 + 3; if (at_cif < 0) verify_frame_ = -2; }
      if (now.kbps && now.dab_plus < 0) now.dab_plus = old.dab_plus;
      // (a sub-channel that only moves to other capacity units keeps running in the engine: same counters)
      const bool same = now.kbps == old.kbps && now.cu_size == old.cu_size &&
                        now.prot_level == old.prot_level && now.short_form == old.short_form && now.dab_plus == old.dab_plus;
      slots_[j] = now;
      if (!same) { delivered_[j] = 0; sf_delivered_[j] = 0; }      // a changed slot counts its logical frames from zero again
      if (now.kbps) n = (int)j + 1;
    }
    for (size_t j = 0; j < slots_.size(); j++) if (slots_[j].kbps) n = (int)j + 1;
    if (at_cif >= 0) check(dabx_set_subchannels_at(eng_, 0, slots_.data(), n, at_cif), "dabx_set_subchannels_at");
    else check(dabx_set_subchannels(eng_, 0, slots_.data(), n), "dabx_set_subchannels");
    if (on_configuration_change) on_configuration_change(at_cif);
    if (verify_frame_ == -2) {                   // (switched late, without a CIF to count from: three frames from now)
      dabx_stats st;
      check(dabx_get_stats(eng_, 0, &st), "dabx_get_stats");
      verify_frame_ = st.frames + 3;
    }
  }
  // Three frames into the new configuration its own FIGs (C/N = 0) have completed whatever the announcement left out of the table
  // (first description wins): a service that was kept running only because the next table did not list it, and that the current
  // table still does not list, has ended with the old configuration.
  void stop_services_that_ended()
  {
    verify_frame_ = -1;
    std::vector<dabx_subch_desc> cur(64);
    const int n_cur = dabx_current_subchannels(eng_, 0, cur.data(), 64);
    if (n_cur <= 0) return;
    bool changed = false;
    for (auto &sl : slots_) {
      if (!sl.kbps) continue;
      bool listed = false;
      for (int k = 0; k < n_cur; k++) listed = listed || cur[(size_t)k].subch_id == sl.subch_id;
      if (!listed) { sl = dabx_subch_desc{}; changed = true; }
    }
    if (changed) apply();
  }

  bool any_service() const
  {
    for (const auto &s : slots_) if (s.kbps) return true;
    return false;
  }
  static void check(long long rc, const char *what)
  {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + dabx_last_error());
  }
  bool apply()
  {
    int n = 0;
    for (size_t j = 0; j < slots_.size(); j++) if (slots_[j].kbps) n = (int)j + 1;
    return dabx_set_subchannels(eng_, 0, slots_.data(), n) == 0;
  }
  void deliver_fibs(int frames)
  {
    if (!on_fib) return;
    frames = frames > params_.fib_ring_frames ? params_.fib_ring_frames : frames;
    std::vector<uint8_t> fibs((size_t)frames * 384), crc((size_t)frames * 12);
    const int have = dabx_read_fibs(eng_, 0, frames, fibs.data(), crc.data());
    for (int i = 0; i < have * 12; i++) on_fib(fibs.data() + 32 * i, crc[(size_t)i] != 0, i % 12);
  }
  void deliver_services()
  {
    if (!on_logical_frame && !on_super_frame && !on_access_unit) return;
    std::vector<uint8_t> buf;
    for (size_t j = 0; j < slots_.size(); j++) {
      if (!slots_[j].kbps) continue;
      dabx_subch_stats st;
      if (dabx_get_subch_stats(eng_, 0, (int)j, &st) < 0 || !st.active) continue;
      if (on_logical_frame && st.cifs_decoded > delivered_[j]) {
        long long fresh = st.cifs_decoded - delivered_[j];
        if (fresh > 32) fresh = 32;                                   // ring depth; older frames are gone
        const int nb = 3 * slots_[j].kbps;
        buf.resize((size_t)fresh * nb);
        const int got = dabx_read_msc(eng_, 0, (int)j, (int)fresh, buf.data());
        for (int i = 0; i < got; i++) on_logical_frame(slots_[j].subch_id, buf.data() + (size_t)i * nb, nb);
        delivered_[j] = st.cifs_decoded;
      }
      if ((on_super_frame || on_access_unit) && st.sf_count > sf_delivered_[j]) {
        long long fresh = st.sf_count - sf_delivered_[j];
        if (fresh > 16) fresh = 16;                                   // ring depth
        const int nb = 110 * slots_[j].kbps / 8;
        buf.resize((size_t)fresh * nb);
        const int got = dabx_read_superframes(eng_, 0, (int)j, (int)fresh, buf.data());
        dabx_superframe_info info[16];
        const int got_i = on_access_unit ? dabx_read_superframe_info(eng_, 0, (int)j, (int)fresh, info) : 0;
        for (int i = 0; i < got; i++) {
          const uint8_t *sf = buf.data() + (size_t)i * nb;
          if (on_super_frame) on_super_frame(slots_[j].subch_id, sf, nb);
          if (on_access_unit && i < got_i) {
            const dabx_superframe_info &r = info[i];
            for (int a = 0; a < r.num_aus; a++) {
              const bool bad_len = (r.au_len_bad >> a) & 1;
              on_access_unit(slots_[j].subch_id, bad_len ? nullptr : sf + r.au_start[a], bad_len ? 0 : r.au_start[a + 1] - r.au_start[a] - 2,
                             ((r.au_crc_ok >> a) & 1) != 0, a, r);
            }
          }
        }
        sf_delivered_[j] = st.sf_count;
      }
    }
  }
  void deliver_tii()
  {
    dabx_tii_result r[64];
    const int n = dabx_read_tii(eng_, 0, params_.tii_frames_to_count, tii_threshold_, tii_collisions_ ? 1 : 0, tii_sub_id_, r, 64, nullptr);
    if (n > 0) on_tii(std::vector<dabx_tii_result>(r, r + n));      // dab_processor.cpp:293-297: only non-empty lists are shown
  }
  void write_eti()
  {
    std::vector<uint8_t> frames((size_t)32 * DABX_ETI_FRAME_BYTES);
    int32_t lost = 0;
    const int n = dabx_read_eti(eng_, 0, 32, frames.data(), &lost);
    if (n > 0) { std::fwrite(frames.data(), DABX_ETI_FRAME_BYTES, (size_t)n, eti_); eti_frames_ += n; }
  }
};

}  // namespace dabx
