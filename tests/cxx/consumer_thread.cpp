// consumer_thread.cpp -- a host CONSUMER of the bulk delivery as a C++ thread on the C ABI alone (include/dabx.h "Bulk delivery"): takes every
// chunk as it lands (dabx_delivery_next), adds up what its records carry, optionally keeps the FIBs + CRC flags of a few sampled streams, and
// gives the slab back (dabx_delivery_release).  bench.py loads it (ctypes) to measure the delivery with this consumer next to the python-thread
// consumer of the same loop (report both); the two library entry points are handed over as function pointers so that the consumer
// talks to whichever libdabx build the bench has loaded (DABX_LIB variants, the hipModule form).
// Not product code: the measuring stick of bench.py's delivery legs, built by tests/cxx/Makefile.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "dabx.h"

namespace {
typedef int (*next_fn)(dabx_engine *, int, dabx_chunk *);
typedef int (*release_fn)(dabx_engine *, uint64_t);
struct Consumer {
  dabx_engine *eng = nullptr;
  next_fn next = nullptr;
  release_fn release = nullptr;
  std::thread th;
  std::atomic<bool> stop{false};
  std::atomic<long long> chunks{0};
  long long frames = 0, cifs = 0, sfs = 0, slab_bytes = 0, payload_bytes = 0, lost = 0, aus = 0, aus_ok = 0;
  std::atomic<int> error{0};
  std::vector<int> sample;                       // streams whose FIBs are kept
  std::vector<uint8_t> fib_log;                  // records: int32 stream, int32 n, int64 first_frame, then n x (384 + 12) bytes
  void run()
  {
    for (;;) {
      dabx_chunk ch;
      const int got = next(eng, /*wait*/ 1, &ch);
      if (got < 0) { error.store(got); return; }
      if (got == 0) {
        if (stop.load()) return;
        std::this_thread::sleep_for(std::chrono::microseconds(100));
        continue;
      }
      const char *b = (const char *)ch.data;
      const dabx_chunk_header *h = (const dabx_chunk_header *)b;
      const dabx_chunk_stream *st = (const dabx_chunk_stream *)(b + h->off_stream);
      const dabx_chunk_subch *sc = (const dabx_chunk_subch *)(b + h->off_subch);
      long long f = 0, c = 0, s_ = 0, l = 0, pay = 0, au = 0, au_ok = 0;
      for (int s = 0; s < h->n_streams; s++) { f += st[s].n_frames; l += st[s].frames_lost; }
      const size_t nsj = (size_t)h->n_streams * (size_t)h->max_subch;
      for (size_t q = 0; q < nsj; q++) {
        c += sc[q].n_cifs; s_ += sc[q].n_sf; l += sc[q].cifs_lost + sc[q].sf_lost;
        pay += (long long)sc[q].n_cifs * 3 * sc[q].kbps + (long long)sc[q].n_sf * 110 * (sc[q].kbps / 8);
        if ((h->what & DABX_DELIVER_SF) && sc[q].n_sf) {            // the access units, judged on the device (dabx_superframe_info): counted, not re-checked
          const dabx_superframe_info *inf = (const dabx_superframe_info *)(b + sc[q].sfi_off);
          for (int k = 0; k < sc[q].n_sf; k++) { au += inf[k].num_aus; au_ok += __builtin_popcount(inf[k].au_crc_ok); }
        }
      }
      if (h->what & DABX_DELIVER_FIB)
        for (int s : sample) {
          const int n = st[s].n_frames;
          if (!n) continue;
          const size_t at = fib_log.size();
          fib_log.resize(at + 16 + (size_t)n * 396);
          const int32_t hd[2] = {s, n};
          std::memcpy(&fib_log[at], hd, 8);
          std::memcpy(&fib_log[at + 8], &st[s].first_frame, 8);
          for (int k = 0; k < n; k++) {
            const size_t row = (size_t)s * h->max_frames + k;
            std::memcpy(&fib_log[at + 16 + (size_t)k * 396], b + h->off_fib + row * 384, 384);
            std::memcpy(&fib_log[at + 16 + (size_t)k * 396 + 384], b + h->off_crc + row * 12, 12);
          }
        }
      frames += f; cifs += c; sfs += s_; lost += l; slab_bytes += (long long)ch.bytes; payload_bytes += f * 396 + pay; aus += au; aus_ok += au_ok;
      if (release(eng, ch.seq) < 0) { error.store(-1); return; }
      chunks.fetch_add(1);                        // last: whoever sees the count sees the chunk's sums too
    }
  }
};
}  // namespace

extern "C" {
void *dbxc_start(void *eng, void *next, void *release, const int *sample, int n_sample)
{
  Consumer *c = new Consumer();
  c->eng = (dabx_engine *)eng; c->next = (next_fn)next; c->release = (release_fn)release;
  c->sample.assign(sample, sample + n_sample);
  c->th = std::thread([c] { c->run(); });
  return c;
}
long long dbxc_chunks(void *h) { return ((Consumer *)h)->chunks.load(); }
int dbxc_error(void *h) { return ((Consumer *)h)->error.load(); }
// chunks, frames, logical frames, super frames, slab bytes, payload bytes, lost, access units, access units ok (valid for the chunks counted)
void dbxc_totals(void *h, long long out[9])
{
  Consumer *c = (Consumer *)h;
  out[0] = c->chunks.load();
  out[1] = c->frames; out[2] = c->cifs; out[3] = c->sfs; out[4] = c->slab_bytes; out[5] = c->payload_bytes; out[6] = c->lost; out[7] = c->aus; out[8] = c->aus_ok;
}
// ends the thread (after the engine was synchronised: everything queued has landed); returns the FIB log's size
long long dbxc_stop(void *h)
{
  Consumer *c = (Consumer *)h;
  c->stop.store(true);
  if (c->th.joinable()) c->th.join();
  return (long long)c->fib_log.size();
}
void dbxc_fib_log(void *h, uint8_t *out) { Consumer *c = (Consumer *)h; if (!c->fib_log.empty()) std::memcpy(out, c->fib_log.data(), c->fib_log.size()); }
void dbxc_free(void *h) { delete (Consumer *)h; }
}
