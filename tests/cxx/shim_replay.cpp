// shim_replay.cpp -- test program for include/dabx_processor.hpp: replays a recorded-IQ file through the
// DabProcessor-shaped adapter exactly as a front end would (start, feed, discover services, select, ETI out) and prints
// counters as one JSON line.  Built with plain g++ against libdabx.so (C ABI only); exit code 3 = no usable GPU.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
#include "dabx_processor.hpp"

int main(int argc, char **argv)
{
  if (argc < 3) { std::fprintf(stderr, "usage: shim_replay <recording> <out.eti> [subChId ...] [--late subChId]\n"); return 2; }
  int late_id = -1;                       // a service added 8 frames after the others (MscHandler::set_channel on a running receiver)
  for (int i = 3; i + 1 < argc; i++)
    if (std::string(argv[i]) == "--late") { late_id = std::atoi(argv[i + 1]); argc = i; break; }
  try {
    dabx::Processor::Params pp;
    pp.max_services = 24;
    dabx::Processor rx(pp);
    long long fibs_ok = 0, fibs = 0, lf = 0, sf = 0;
    std::map<int, long long> lf_per_service, lf_this_run, stalls;
    long long late_added_at = -1;
    rx.on_fib = [&](const uint8_t *, bool ok, int) { fibs++; fibs_ok += ok; };
    rx.on_logical_frame = [&](int id, const uint8_t *, int) { lf++; lf_per_service[id]++; lf_this_run[id]++; };
    rx.on_super_frame = [&](int, const uint8_t *, int) { sf++; };
    long long lcd_records = 0;
    float lcd_snr = 0.f, lcd_mer = 0.f;
    rx.on_lcd_data = [&](float snr_db, float mer_db) { lcd_records++; lcd_snr = snr_db; lcd_mer = mer_db; };   // signal_show_lcd_data's device-side numbers
    long long clock_reports = 0, sync_found = 0, sync_not_found = 0;
    float clock_err = 0.f, level_mean = 0.f;
    rx.on_clock_error = [&](float hz) { clock_reports++; clock_err = hz; };
    rx.on_signal_level = [&](float, float mean) { level_mean = mean; };
    rx.on_time_sync = [&](bool found) { (found ? sync_found : sync_not_found)++; };
    // the AAC decoder's seat: access units arrive sliced and judged (dabx_superframe_info); this stub only counts -- and, being a TEST, checks
    // the verdicts it was handed with a CRC of its own (crc.cpp:75-86: CCITT, start 0xFFFF, complemented, over the frame without its last two bytes)
    long long aus = 0, aus_ok = 0, au_verdict_mismatch = 0, au_bytes = 0;
    rx.on_access_unit = [&](int, const uint8_t *au, int len, bool crc_ok, int, const dabx_superframe_info &) {
      aus++; aus_ok += crc_ok;
      if (!au) { au_verdict_mismatch += crc_ok; return; }               // failed the length check: never "good"
      au_bytes += len;
      unsigned crc = 0xFFFF;
      for (int i = 0; i < len; i++) {
        crc ^= (unsigned)au[i] << 8;
        for (int b = 0; b < 8; b++) crc = (crc & 0x8000) ? ((crc << 1) ^ 0x1021) & 0xFFFF : (crc << 1) & 0xFFFF;
      }
      const bool mine = ((~crc) & 0xFFFF) == (((unsigned)au[len] << 8) | au[len + 1]);
      au_verdict_mismatch += mine != crc_ok;
    };
    long long config_changes = 0, change_cif = -1;
    rx.on_configuration_change = [&](long long cif) { config_changes++; change_cif = cif; };
    const dabx_iq_format fmt = rx.open_recording(argv[1]);
    rx.start();
    if (!rx.start_eti_generator(argv[2])) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
    std::FILE *fp = std::fopen(argv[1], "rb");
    std::fseek(fp, (long)fmt.data_offset, SEEK_SET);
    const size_t spb = (size_t)dabx_iq_sample_bytes(&fmt);
    std::vector<uint8_t> block((size_t)(fmt.sample_rate / 1000) * 96 * 3 * spb);       // three frames' worth of recording
    long long left = fmt.data_bytes, frames = 0;
    bool configured = false;
    while (left > 0) {
      const size_t want = (size_t)std::min<long long>(left, (long long)block.size());
      const size_t got = std::fread(block.data(), 1, want, fp);
      if (got == 0) break;
      left -= (long long)got;
      rx.put_file_bytes(block.data(), got);
      lf_this_run.clear();
      const int ran = rx.run(4);
      frames += ran;
      // a service that has started delivering must deliver with every processed frame (4 logical frames each)
      for (const auto &kv : lf_per_service)
        if (ran > 0 && lf_this_run[kv.first] != 4 * ran && lf_per_service[kv.first] != lf_this_run[kv.first]) stalls[kv.first]++;
      if (configured && late_id >= 0 && late_added_at < 0 && lf_per_service.size() > 0 && frames >= 14) {
        if (!rx.set_audio_channel(late_id)) { std::fprintf(stderr, "late service %d not found\n", late_id); return 4; }
        late_added_at = frames;
      }
      if (!configured && rx.get_fic_decode_ratio_percent() >= 90) {
        if (argc > 3) { configured = true; for (int i = 3; i < argc; i++) configured = rx.set_audio_channel(std::atoi(argv[i])) && configured; }
        else configured = rx.set_all_channels() > 0;
      }
    }
    std::fclose(fp);
    rx.stop_eti_generator();
    rx.stop();
    long long n_stalls = 0;
    for (const auto &kv : stalls) n_stalls += kv.second;
    std::string per = "{";
    for (const auto &kv : lf_per_service) per += (per.size() > 1 ? ", \"" : "\"") + std::to_string(kv.first) + "\": " + std::to_string(kv.second);
    per += "}";
    std::printf("{\"frames\": %lld, \"fibs\": %lld, \"fibs_ok\": %lld, \"logical_frames\": %lld, \"super_frames\": %lld, \"services\": %zu, "
                "\"eti_frames\": %lld, \"stalls\": %lld, \"late_added_at\": %lld, \"config_changes\": %lld, \"change_cif\": %lld, "
                "\"access_units\": %lld, \"access_units_ok\": %lld, \"au_verdict_mismatch\": %lld, \"au_bytes\": %lld, \"lcd_records\": %lld, \"lcd_snr\": %.3f, "
                "\"lcd_mer\": %.3f, \"clock_reports\": %lld, \"clock_err_hz\": %.3f, \"level_mean\": %.5f, \"sync_found\": %lld, \"sync_not_found\": %lld, "
                "\"lf_per_service\": %s}\n",
                frames, fibs, fibs_ok, lf, sf, lf_per_service.size(), rx.eti_frames_written(), n_stalls, late_added_at, config_changes, change_cif,
                aus, aus_ok, au_verdict_mismatch, au_bytes, lcd_records, (double)lcd_snr, (double)lcd_mer, clock_reports,
                (double)clock_err, (double)level_mean, sync_found, sync_not_found, per.c_str());
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "shim_replay: %s\n", e.what());
    return std::string(e.what()).find("dabx_create") == 0 ? 3 : 1;
  }
}
