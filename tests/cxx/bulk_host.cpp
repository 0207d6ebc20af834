// bulk_host.cpp -- the host loop of INTEGRATION.md section 3 as a program: plain C++ against the C ABI only (no Python, no Qt).
//   bulk_host <input> <output prefix>
// input : int32 n_streams, int32 n_frames, int32 n_subch, n_subch x 8 int32 {SubChId, CuStart, CuSize, kbps, protLevel, shortForm, dabPlus, 0},
//         then per stream n_frames * 196608 uint8 I/Q pairs (the raw_reader.cpp:66-70 format).
// The engine's thread fills page-locked ingest slabs (dabx_ingest_*: one SDMA transfer + one conversion kernel per slab) and steps the
// receiver; ONE consumer thread takes the chunks (dabx_delivery_next / _release) and appends, per stream, every FIB + CRC flag and, per
// sub-channel, every logical frame and super frame to files -- what IFibDecoder::process_FIB (fic_decoder.cpp:234-261),
// FrameProcessor::add_to_frame (backend.cpp:160) and the super-frame consumer of mp4processor.cpp:149-158 would be handed.
// Exit code 3 = no usable GPU.  Prints one JSON line of totals.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "dabx.h"

static void die(const char *what) { std::fprintf(stderr, "bulk_host: %s: %s\n", what, dabx_last_error()); std::exit(1); }

int main(int argc, char **argv)
{
  if (argc < 3) { std::fprintf(stderr, "usage: bulk_host <input> <output prefix>\n"); return 2; }
  if (dabx_device_count() < 1) { std::fprintf(stderr, "no HIP device: libdabx has no CPU fallback\n"); return 3; }
  std::FILE *in = std::fopen(argv[1], "rb");
  if (!in) { std::perror(argv[1]); return 2; }
  int32_t hdr[3];
  if (std::fread(hdr, 4, 3, in) != 3) return 2;
  const int S = hdr[0], n_frames = hdr[1], M = hdr[2], CH = 4;              // four frames per slab
  std::vector<dabx_subch_desc> sc((size_t)M);
  for (auto &d : sc) {
    int32_t v[8];
    if (std::fread(v, 4, 8, in) != 8) return 2;
    d = dabx_subch_desc{v[0], v[1], v[2], v[3], v[4], v[5], v[6], 0};
  }
  const size_t per_frame = (size_t)DABX_TF * 2;
  std::vector<std::vector<uint8_t>> iq((size_t)S, std::vector<uint8_t>((size_t)n_frames * per_frame));
  for (auto &v : iq) if (std::fread(v.data(), 1, v.size(), in) != v.size()) return 2;
  std::fclose(in);

  dabx_config cfg;
  dabx_default_config(&cfg);
  cfg.n_streams = S; cfg.ring_frames = 3 * CH; cfg.max_subch = M; cfg.out_frames = 8;
  dabx_engine *eng = nullptr;
  if (dabx_create(&cfg, &eng) < 0) die("dabx_create");
  if (dabx_set_subchannels(eng, -1, sc.data(), M) < 0) die("dabx_set_subchannels");
  dabx_ingest_config ic{};
  ic.host_slabs = 2; ic.fmt = 2; ic.max_frames = CH;
  if (dabx_ingest_open(eng, &ic) < 0) die("dabx_ingest_open");
  dabx_delivery_config dc{};
  dc.host_slabs = 3;
  if (dabx_delivery_open(eng, &dc) < 0) die("dabx_delivery_open");
  void *slab[2];
  size_t cap = 0;
  for (int k = 0; k < 2; k++) if (dabx_ingest_slab(eng, k, &slab[k], &cap) < 0) die("dabx_ingest_slab");

  // ---- the consumer thread
  const std::string prefix = argv[2];
  std::atomic<bool> producer_done{false};
  long long n_chunks = 0, n_fibs = 0, n_fibs_ok = 0, n_lf = 0, n_sf = 0, n_lost = 0, n_au = 0, n_au_ok = 0, n_au_bytes = 0;
  std::thread consumer([&] {
    std::vector<std::FILE *> ffib((size_t)S), flf((size_t)S * M), fsf((size_t)S * M), fsfi((size_t)S * M);
    for (int s = 0; s < S; s++) {
      ffib[(size_t)s] = std::fopen((prefix + ".s" + std::to_string(s) + ".fibs").c_str(), "wb");
      for (int j = 0; j < M; j++) {
        flf[(size_t)s * M + j] = std::fopen((prefix + ".s" + std::to_string(s) + ".lf" + std::to_string(j)).c_str(), "wb");
        fsf[(size_t)s * M + j] = std::fopen((prefix + ".s" + std::to_string(s) + ".sf" + std::to_string(j)).c_str(), "wb");
        fsfi[(size_t)s * M + j] = std::fopen((prefix + ".s" + std::to_string(s) + ".sfi" + std::to_string(j)).c_str(), "wb");
      }
    }
    for (;;) {
      dabx_chunk ch;
      const int got = dabx_delivery_next(eng, /*wait*/ 1, &ch);
      if (got < 0) die("dabx_delivery_next");
      if (got == 0) { if (producer_done.load()) break; std::this_thread::yield(); continue; }
      const char *b = (const char *)ch.data;
      const dabx_chunk_header *h = (const dabx_chunk_header *)b;
      if (h->magic != DABX_CHUNK_MAGIC || h->seq != (uint64_t)n_chunks || h->bytes != ch.bytes) { std::fprintf(stderr, "bulk_host: bad chunk header\n"); std::exit(1); }
      const dabx_chunk_stream *st = (const dabx_chunk_stream *)(b + h->off_stream);
      const dabx_chunk_subch *q = (const dabx_chunk_subch *)(b + h->off_subch);
      for (int s = 0; s < h->n_streams; s++) {
        n_lost += st[s].frames_lost;
        for (int f = 0; f < st[s].n_frames; f++) {
          const size_t row = (size_t)s * h->max_frames + f;
          std::fwrite(b + h->off_fib + row * 384, 1, 384, ffib[(size_t)s]);          // 12 FIBs, then their 12 CRC flags
          std::fwrite(b + h->off_crc + row * 12, 1, 12, ffib[(size_t)s]);
          for (int i = 0; i < 12; i++) { n_fibs++; n_fibs_ok += b[h->off_crc + row * 12 + i] != 0; }
        }
        for (int j = 0; j < h->max_subch; j++) {
          const dabx_chunk_subch &r = q[(size_t)s * h->max_subch + j];
          n_lost += r.cifs_lost + r.sf_lost;
          if (!r.active) continue;
          std::fwrite(b + r.msc_off, 1, (size_t)r.n_cifs * 3 * r.kbps, flf[(size_t)s * M + j]);
          for (int c = 0; c < r.n_sf; c++) std::fwrite(b + r.sf_off + (size_t)c * r.sf_pitch, 1, (size_t)(110 * r.kbps / 8), fsf[(size_t)s * M + j]);
          // the access units as the AAC decoder's seat gets them (mp4processor.cpp:306-345): sliced by the record, judged by the record
          const dabx_superframe_info *inf = (const dabx_superframe_info *)(b + r.sfi_off);
          if (r.n_sf) std::fwrite(inf, sizeof(dabx_superframe_info), (size_t)r.n_sf, fsfi[(size_t)s * M + j]);
          for (int c = 0; c < r.n_sf; c++)
            for (int a = 0; a < inf[c].num_aus; a++) {
              n_au++;
              if (inf[c].au_crc_ok >> a & 1) { n_au_ok++; n_au_bytes += inf[c].au_start[a + 1] - inf[c].au_start[a] - 2; }
            }
          n_lf += r.n_cifs; n_sf += r.n_sf;
        }
      }
      n_chunks++;
      if (dabx_delivery_release(eng, ch.seq) < 0) die("dabx_delivery_release");
    }
    for (auto f : ffib) std::fclose(f);
    for (auto f : flf) std::fclose(f);
    for (auto f : fsf) std::fclose(f);
    for (auto f : fsfi) std::fclose(f);
  });

  // ---- the engine's thread: slab k + 1 goes on the link while slab k is converted and decoded
  const int n_slabs = n_frames / CH;
  auto fill = [&](int k) {
    for (int s = 0; s < S; s++) std::memcpy((uint8_t *)slab[k & 1] + (size_t)s * CH * per_frame, iq[(size_t)s].data() + (size_t)k * CH * per_frame, (size_t)CH * per_frame);
  };
  fill(0);
  if (dabx_ingest_submit(eng, 0, (size_t)CH * DABX_TF) < 0) die("dabx_ingest_submit");
  for (int k = 0; k < n_slabs; k++) {
    if (k + 1 < n_slabs) { fill(k + 1); if (dabx_ingest_submit(eng, (k + 1) & 1, (size_t)CH * DABX_TF) < 0) die("dabx_ingest_submit"); }
    if (dabx_ingest_commit(eng, k & 1) < 0) die("dabx_ingest_commit");
    if (dabx_delivery_wait_free(eng, 1, -1) < 1) die("dabx_delivery_wait_free");
    if (dabx_process(eng, CH, /*sync*/ 0) < 0) die("dabx_process");
  }
  for (int k = 0; k < 3; k++) {                                    // the frames a stream stayed behind while it searched for its first lock
    if (dabx_delivery_wait_free(eng, 1, -1) < 1) die("dabx_delivery_wait_free");
    if (dabx_process(eng, 2, 0) < 0) die("dabx_process");
  }
  if (dabx_synchronize(eng) < 0) die("dabx_synchronize");         // every chunk has landed
  producer_done.store(true);
  consumer.join();
  dabx_delivery_info info;
  if (dabx_delivery_get_info(eng, &info) < 0) die("dabx_delivery_get_info");
  long long frames = 0;
  for (int s = 0; s < S; s++) { dabx_stats stt; if (dabx_get_stats(eng, s, &stt) < 0) die("dabx_get_stats"); frames += stt.frames; }
  dabx_delivery_close(eng);
  dabx_ingest_close(eng);
  dabx_destroy(eng);
  std::printf("{\"streams\": %d, \"chunks\": %lld, \"frames\": %lld, \"fibs\": %lld, \"fibs_ok\": %lld, \"logical_frames\": %lld, \"super_frames\": %lld, \"lost\": %lld, "
              "\"access_units\": %lld, \"access_units_ok\": %lld, \"au_bytes\": %lld, "
              "\"copies\": %llu, \"link_GBps\": %.2f}\n", S, n_chunks, frames, n_fibs, n_fibs_ok, n_lf, n_sf, n_lost, n_au, n_au_ok, n_au_bytes, (unsigned long long)info.chunks_landed,
              info.copy_seconds > 0 ? (double)info.bytes_copied / info.copy_seconds / 1e9 : 0.0);
  return 0;
}
