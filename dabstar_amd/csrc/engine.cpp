// engine.cpp -- host side of the stream-batched receiver and the engine-level C ABI (include/dabx.h).
#include "pipeline.h"
#include "viterbi_core.h"
#include "sdma.h"
#include "iqfile.h"
#ifndef DABX_CU_SPLIT_DEMAP_FRONT
#define DABX_CU_SPLIT_DEMAP_FRONT 0
#endif
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

namespace dabx {
int launch_front_step(const EngineDev &e, EngineStreams &ss, Marker &mk, bool async_acquire, bool all_locked);
int launch_msc_batch(const EngineDev &e, int cifs, const MscFast *fast, EngineStreams &ss, Marker &mk, const DeliverDev *dv = nullptr,
                     hipStream_t *tail = nullptr);
int launch_deliver_front(const EngineDev &e, const DeliverDev &dv, hipStream_t st);
int launch_ingest_convert(const EngineDev &e, const void *src, int fmt, size_t n, hipStream_t st);
void dabx_internal_fibdec_skip(dabx_fibdec *d, long long n_fibs);     // fib.cpp: FIBs the decoder never saw (they had left the ring)
int launch_dciq(const EngineDev &e, int mode, hipStream_t st);
int launch_level_exact(const EngineDev &e, hipStream_t st);
int launch_stage_msc_block(const EngineDev &e, const int16_t *soft_dev, int blk, bool closes_cif, hipStream_t st);
extern const char *const kStepKernelNames[11];
int launch_commit(const EngineDev &e, int stream, unsigned long long n, hipStream_t st);
int launch_convert_iq(const void *src, int fmt, float2 *ring, int ring_len, unsigned long long wr0, size_t n, hipStream_t st);
int launch_fic_only(const EngineDev &e, hipStream_t st, int first, int count);
int launch_i16_to_sym(const int16_t *soft, uint8_t *sym, size_t n, hipStream_t st);
}  // namespace dabx
using namespace dabx;

// Bulk delivery (include/dabx.h): host slabs (page-locked) the chunks land in, device slabs they are packed into, and the copier --
// a thread of the library that waits (polling every 50 us) for a chunk's gather kernels and then moves the slab with ONE SDMA
// transfer (sdma.h: the HIP runtime's own device-to-host copy is a shader copy that stalls the receiver's kernels while it runs).
struct Delivery {
  bool open = false;
  int what = 0;
  int copy_engine = 0;                           // dabx_delivery_config.copy_engine: 0 SDMA through the HSA runtime, 1 hipMemcpyAsync
  static constexpr int NDEV = 3;                 // device slabs: chunk n + 3 is packed into the slab of chunk n once its copy has left
  uint8_t *dev[NDEV] = {nullptr, nullptr, nullptr};
  hipEvent_t packed[NDEV] = {nullptr, nullptr, nullptr};    // the chunk's gather kernels have finished (system-scope release; the copier polls them)
  hipEvent_t packed_lf[NDEV] = {nullptr, nullptr, nullptr}; // ... its logical frames (the slab's tail, [off_msc, bytes)) are in the slab: that share of the
                                                            // transfer starts while the DAB+ stage still runs (round 6)
  bool dev_busy[NDEV] = {false, false, false};   // packed into or being copied from (guarded by mu)
  hipStream_t cs = nullptr;                      // copy_engine 1 only
  Sdma sdma;
  size_t capacity = 0, bytes = 0;                // bytes allocated per slab / bytes the current layout uses (= what is copied)
  enum { FREE = 0, IN_FLIGHT = 1, LANDED = 2, HELD = 3 };
  struct Slot { uint8_t *host = nullptr; uint64_t sig = 0, sig2 = 0; int state = FREE; uint64_t seq = 0; size_t bytes = 0, lf_from = 0; int devslab = 0; };
  std::vector<Slot> slots;
  std::deque<int> queue;                         // slots in flight or landed, oldest first (what dabx_delivery_next hands out)
  std::deque<int> jobs;                          // slots whose copy the copier still has to make
  std::mutex mu;                                 // everything above: the engine's thread, the copier and ONE consumer thread
  std::condition_variable cv;                    // any state change
  std::thread copier;
  bool quit = false;
  int device = 0;
  std::string copier_error;
  uint64_t next_seq = 0, landed = 0, bytes_copied = 0;
  double copy_s = 0, copy_s_max = 0, gather_wait_s = 0, calib_gbps = 0;
  unsigned long long *layout_off = nullptr;      // device tables (DeliverDev)
  int32_t *subch_id = nullptr;
  long long *frames_done = nullptr, *cif_done = nullptr, *sf_done = nullptr;
  dabx_chunk_header hdr{};
};

// Bulk ingest (include/dabx.h "Bulk ingest"): page-locked input slabs, their device twins, one SDMA transfer per slab.
struct Ingest {
  bool open = false;
  int fmt = 0, copy_engine = 0, max_frames = 0;
  size_t capacity = 0;                           // bytes per slab
  struct Slab { uint8_t *host = nullptr, *dev = nullptr; uint64_t sig = 0; size_t n = 0; bool in_flight = false; };
  std::vector<Slab> slabs;
  Sdma sdma;
  hipStream_t cs = nullptr;                      // copy_engine 1 only
  hipEvent_t committed = nullptr;                // the previous ingest commit has run on the front-end stream (the converter reads the committed indices)
  bool committed_recorded = false;
  hipEvent_t front = nullptr;                    // where the front-end stream stood when a conversion was queued (commits of other entry points in between)
  // general form (dabx_ingest_open_formats): every stream its own container, rate and length
  bool general = false;
  size_t pitch = 0;                              // bytes per stream region of a slab
  std::vector<IqDecode> dec;                     // [S]
  std::vector<int> M, tab, carry_n;              // [S] input samples per ms (0 = 2.048 MS/s), table index, samples carried between slabs
  std::vector<std::vector<size_t>> n_bytes;      // [slab][S] payload bytes submitted
  IngestJob *jobs_host = nullptr, *jobs_dev = nullptr;     // [S] page-locked staging / device
  unsigned *counts_host = nullptr, *counts_dev = nullptr;  // [S] samples committed per stream by this commit
  float2 *work = nullptr, *carry = nullptr;
  size_t work_pitch = 0, carry_pitch = 0;
  int16_t *tab_int = nullptr; float *tab_frac = nullptr;
};

struct dabx_engine {
  dabx_config cfg{};
  EngineDev dev{};
  hipStream_t stream = nullptr;                // == ss.a (front end)
  EngineStreams ss;
  BatchSnap *snap_buf[2] = {nullptr, nullptr};
  int device = 0;
  std::vector<unsigned long long> wr_host;     // host mirror of committed samples
  std::vector<SubchDev> subch_host;            // [S][max_subch]
  std::vector<int> subch_id_host;              // [S][max_subch] SubChId (host only: ETI STC field)
  struct EtiCursor { long long next_cif = -1; int hi = -1, lo = -1; long long fib_frames_seen = 0; };
  std::vector<EtiCursor> eti;                  // [S]
  std::vector<dabx_fibdec *> fibdec;           // [S] FIB decoders (current / next configuration), created on first dabx_follow_fic
  std::vector<long long> fib_frames_fed;       // [S] frames whose FIBs the decoder has seen
  bool fig_reference_quirks = false;           // dabx_set_fig_reference_quirks: the engine's own FIB decoders swap like the reference (flags 3 only)
  std::vector<dabx_tii *> tii;                 // [S] detectors, created on first dabx_read_tii
  std::vector<int> tii_epoch;                  // [S] reset epoch seen by the detector
  std::vector<void *> allocs;
  void *stage = nullptr;                       // host -> device staging of dabx_push_iq
  size_t stage_cap = 0;
  // dabx_push_iq_async: a small pool of device staging slots, each guarded by the event of its last conversion kernel, so
  // that consecutive pushes from pinned host memory queue back to back on the ingest stream (DMA at PCIe rate, no host wait)
  static constexpr int ASYNC_SLOTS = 8;
  void *aslot[ASYNC_SLOTS] = {nullptr};
  size_t aslot_cap[ASYNC_SLOTS] = {0};
  hipEvent_t aslot_done[ASYNC_SLOTS] = {nullptr};
  unsigned long long async_pushes = 0;
  hipStream_t ingest = nullptr;                // dabx_push_iq: copy + format conversion, concurrent with the receiver streams
  hipStream_t ingest2 = nullptr;               // dabx_push_iq_async alternates between the two: the DMA of push k + 1 runs under the conversion of push k
  hipEvent_t ingest_done = nullptr;
  std::vector<unsigned long long> rd_seen;     // [S] read index of every stream when last looked at (lower bound)
  std::vector<StreamCtl> ctl_peek;
  int max_kbps = 0;
  bool buffers_ready = false;
  Marker mk;
  double prof_ms[N_STEP_KERNELS] = {0};
  long long prof_n[N_STEP_KERNELS] = {0};
  int pending_frames = 0;                      // front-end steps whose CIFs still await the MSC decoder
  bool have_fast = false;
  MscFast fast{};

  std::vector<void *> fast_allocs;             // buffers of the current MSC classes (replaced on reconfiguration)
  bool classes_dirty = false;
  std::vector<char> announcing;                // per stream: the zero-copy producer uses dabx_announce_write
  unsigned long long *horizon_host = nullptr;  // hipHostMalloc'ed, EngineDev::wr_horizon: what pushes may have overwritten (written BEFORE a copy is issued)
  int32_t *locked_host = nullptr;              // hipHostMalloc'ed: number of streams in lock, kept by the device (EngineDev::locked_count)
  int32_t *seq_timeouts_host = nullptr;        // hipHostMalloc'ed: device-side waits that gave up (EngineDev::seq_timeouts)
  bool level_dirty = false;                    // exact_level_tracker: steps have been issued since k_level_exact last ran behind them
  Delivery dl;
  Ingest ing;
  int build_msc_classes();
  int delivery_layout();                       // offsets of every slot's bytes in a slab for the sub-channels configured now
  int delivery_begin(DeliverDev *dv, int *slot, int *devslab);     // a chunk closes: host + device slab, front gather on stream a
  int delivery_finish(int slot, int devslab, hipStream_t tail);    // ... its slot gather is queued on `tail`: the one copy
  void delivery_abort(int slot, int devslab);                      // ... or it cannot be: both slabs go back

  template <class T> int alloc(T **p, size_t count, bool zero = true)
  {
    void *q = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    DABX_HIP(hipMalloc(&q, bytes));
    if (zero) DABX_HIP(hipMemsetAsync(q, 0, bytes, stream));
    allocs.push_back(q);
    *p = reinterpret_cast<T *>(q);
    return 0;
  }
};

// Groups the active (stream, slot) pairs by protection profile for the lane-per-trellis decoder (vit_t.hip): every
// class gets its depuncture map, pair list and transposed-symbol / decision scratch.  Up to MSC_MAX_CLASSES classes,
// largest first; classes too small to fill a few waves and anything beyond stay with the wave-per-trellis kernel.
int dabx_engine::build_msc_classes()
{
  const EngineDev &d = dev;
  have_fast = false;
  fast = MscFast{};
  for (void *q : fast_allocs) {
    (void)hipFree(q);
    allocs.erase(std::remove(allocs.begin(), allocs.end(), q), allocs.end());
  }
  fast_allocs.clear();
  for (auto &sc : subch_host) sc.fast_class = 0;
  if (d.max_subch <= 0) return 0;
  struct Key { int kbps, prot, shortf, cu; bool operator<(const Key &o) const { return std::tie(kbps, prot, shortf, cu) < std::tie(o.kbps, o.prot, o.shortf, o.cu); } };
  std::map<Key, std::vector<uint32_t>> groups;
  int active = 0;
  for (int s = 0; s < d.n_streams; s++)
    for (int j = 0; j < d.max_subch; j++) {
      const SubchDev &sc = subch_host[(size_t)s * d.max_subch + j];
      if (!sc.active) continue;
      active++;
      groups[Key{sc.kbps, sc.prot_level, sc.short_form, sc.cu_size}].push_back(((uint32_t)s << 8) | (uint32_t)j);
    }
  std::vector<std::pair<Key, std::vector<uint32_t>>> order(groups.begin(), groups.end());
  std::stable_sort(order.begin(), order.end(), [](const auto &a, const auto &b) {
    return (long long)a.second.size() * a.first.kbps > (long long)b.second.size() * b.first.kbps; });
  // 64 trellises per wave: measured break-even against the wave-per-trellis kernel at ~320 waves per batch (40-48 streams
  // of 18 sub-channels); a class with fewer than 4 decoder waves per full batch is not worth its own launch slice.
  // dabx_config.msc_fast_min_jobs / msc_class_min_jobs override both (include/dabx.h).
  const size_t min_jobs = cfg.msc_fast_min_jobs > 0 ? (size_t)cfg.msc_fast_min_jobs : (size_t)64 * 320;
  const size_t class_min_jobs = cfg.msc_class_min_jobs > 0 ? (size_t)cfg.msc_class_min_jobs : 256;
  const int max_cifs = 4 * MSC_BATCH_FRAMES;
  long long jobs_total = 0;
  std::vector<std::vector<uint32_t>> host_pairs;
  int rc;
  auto falloc = [&](auto **p, size_t count) {
    int r = alloc(p, count, false);
    if (!r) fast_allocs.push_back(*p);
    return r;
  };
  for (const auto &kv : order) {
    if (fast.n_cls >= MSC_MAX_CLASSES) break;
    const Key &k = kv.first;
    const std::vector<uint32_t> &pairs = kv.second;
    if (pairs.size() * (size_t)max_cifs < class_min_jobs) continue;
    std::vector<uint16_t> m;
    int n_in = 0;
    if ((rc = host_profile_map(k.kbps, k.prot, k.shortf, m, &n_in))) return rc;
    if (n_in % 64 != 0 || n_in != k.cu * 64) continue;
    for (auto &v : m) if (v == PUNCT) v = (uint16_t)n_in;
    MscClass c{};
    c.n_in = n_in; c.nbits = 24 * k.kbps; c.n_pairs = (int)pairs.size();
    uint16_t *map2 = nullptr;
    uint32_t *pr = nullptr;
    if ((rc = falloc(&map2, m.size()))) return rc;
    if ((rc = falloc(&pr, pairs.size()))) return rc;
    DABX_HIP(hipMemcpy(map2, m.data(), m.size() * 2, hipMemcpyHostToDevice));
    DABX_HIP(hipMemcpy(pr, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    c.map2 = map2; c.pairs = pr;
    const size_t ngroups = (pairs.size() * (size_t)max_cifs + 63) / 64;
    if ((rc = falloc(&c.inT[0], ngroups * (size_t)(n_in / 4 + 1) * 64))) return rc;
    if ((rc = falloc(&c.inT[1], ngroups * (size_t)(n_in / 4 + 1) * 64))) return rc;
    if ((rc = falloc(&c.decT, ngroups * (size_t)(c.nbits + 6) * 64))) return rc;
    fast.cls[fast.n_cls++] = c;
    host_pairs.push_back(pairs);
    jobs_total += (long long)pairs.size() * max_cifs;
  }
  // launch order: longest trellises first, so that the short ones fill in behind them
  std::vector<int> idx(fast.n_cls);
  for (int i = 0; i < fast.n_cls; i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return fast.cls[a].nbits > fast.cls[b].nbits; });
  MscFast sorted = fast;
  for (int i = 0; i < fast.n_cls; i++) {
    sorted.cls[i] = fast.cls[idx[i]];
    for (uint32_t q : host_pairs[idx[i]]) subch_host[(size_t)(q >> 8) * d.max_subch + (q & 255u)].fast_class = i + 1;
  }
  fast = sorted;
  fast.min_jobs = (int)std::min<size_t>(min_jobs, 0x7fffffff);
  fast.slots_active = active;
  have_fast = fast.n_cls > 0 && (size_t)jobs_total >= min_jobs;   // every viterbi_tie_mode has its lane-per-trellis kernel (vit_t.hip)
  DABX_HIP(hipMemcpy(d.subch, subch_host.data(), sizeof(SubchDev) * subch_host.size(), hipMemcpyHostToDevice));
  return 0;
}

// Engine calls may come from any host thread: bind the thread to the engine's device first (allocations, launches and
// copies below otherwise go to whatever device the calling thread happens to have current).
static int use_device(const dabx_engine *e)
{
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == e->device) return 0;
  DABX_HIP(hipSetDevice(e->device));
  return 0;
}

static int delivery_drain(dabx_engine *e);
// chain_only: what dabx_process(sync != 0) waits for -- the frame chain, the MSC batches and the delivery of the frames it issued -- but not
// a search pass that runs next to them on stream q for streams out of lock (its results are picked up by the next step either way)
static int sync_all(dabx_engine *e, bool chain_only = false)
{
  if (int rc = use_device(e)) return rc;
  DABX_HIP(hipStreamSynchronize(e->stream));
  if (e->ss.b) DABX_HIP(hipStreamSynchronize(e->ss.b));
  if (e->ss.d) DABX_HIP(hipStreamSynchronize(e->ss.d));
  if (e->seq_timeouts_host && __atomic_load_n(e->seq_timeouts_host, __ATOMIC_RELAXED) != 0) {
    set_error("%d device-side hand-overs of the few-stream schedule timed out (a kernel launch in front of them failed): the results since are undefined",
              (int)__atomic_load_n(e->seq_timeouts_host, __ATOMIC_RELAXED));
    return DABX_E_HIP;
  }
  if (int rc = delivery_drain(e)) return rc;                       // every chunk closed so far has landed
  if (chain_only && !e->dev.exact_level) return 0;
  if (e->ss.q) DABX_HIP(hipStreamSynchronize(e->ss.q));
  e->ss.acq_in_flight = false;
  // cfg.exact_level_tracker: the level tracker follows the frame chain on its own; behind the last frame it is run once more, so
  // that what the host reads next (dabx_get_stats, the ring's read cursor) includes every sample the receiver has read
  if (e->dev.exact_level && e->dev.level_pos && e->level_dirty) {
    if (int rc = launch_level_exact(e->dev, e->stream)) return rc;
    DABX_HIP(hipStreamSynchronize(e->stream));
    e->level_dirty = false;
  }
  return 0;
}

// ---- bulk delivery (include/dabx.h "Bulk delivery", deliver.hip) -------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// the slab's records are ABI: hosts and the python binding (dabstar_amd/lib.py, CHUNK_*) parse them by these sizes
static_assert(sizeof(dabx_chunk_header) == 128 && sizeof(dabx_chunk_stream) == 72 && sizeof(dabx_chunk_frame) == 16 && sizeof(dabx_chunk_subch) == 144 &&
              sizeof(dabx_superframe_info) == 32,
              "include/dabx.h: chunk record layout");
static constexpr int DL_FRAMES = MSC_BATCH_FRAMES;                 // a chunk = what one MSC batch decodes
static constexpr int DL_SF_CAP = (4 * DL_FRAMES + 4) / 5;          // super frames one chunk can complete (4 CIFs may be waiting from before)
#if DABX_MSC_BATCH == 7
static_assert(DL_FRAMES == DABX_CHUNK_FRAMES, "include/dabx.h: DABX_CHUNK_FRAMES is the library's MSC batch");
#endif

// Where every slot's bytes lie in a slab with the sub-channels configured now: table part (header, stream and slot records, FIBs,
// CRC flags, frame records), then the logical frames of all slots, then the super frames of all slots.  Uploaded to the device;
// called with the engine drained (dabx_delivery_open, dabx_set_subchannels*).
int dabx_engine::delivery_layout()
{
  Delivery &D = dl;
  const EngineDev &d = dev;
  const size_t S = (size_t)d.n_streams, M = (size_t)d.max_subch, F = DL_FRAMES;
  dabx_chunk_header h{};
  h.magic = DABX_CHUNK_MAGIC; h.abi = DABX_ABI_VERSION;
  h.n_streams = d.n_streams; h.max_subch = d.max_subch; h.max_frames = DL_FRAMES; h.what = D.what;
  size_t off = sizeof(dabx_chunk_header);
  h.off_stream = off; off = align_up(off + S * sizeof(dabx_chunk_stream), 16);
  h.off_subch = off; off = align_up(off + S * M * sizeof(dabx_chunk_subch), 16);
  const bool fib = (D.what & DABX_DELIVER_FIB) != 0;
  h.off_fib = off; if (fib) off = align_up(off + S * F * 384, 16);
  h.off_crc = off; if (fib) off = align_up(off + S * F * 12, 16);
  h.off_frame = off; if (fib) off = align_up(off + S * F * sizeof(dabx_chunk_frame), 16);
  std::vector<unsigned long long> lo(3 * S * M + 3, 0);
  // (super frames in front of the logical frames since round 6: the logical frames are the slab's TAIL, [off_msc, bytes), and travel first --
  //  behind the Viterbi decode, next to the DAB+ stage; the offsets in the records are what a host goes by)
  h.off_sf = off;
  if ((D.what & DABX_DELIVER_SF) && !d.fic_only)
    for (size_t sj = 0; sj < S * M; sj++) {
      const SubchDev &sc = subch_host[sj];
      if (!sc.active || !sc.dab_plus) continue;
      lo[3 * sj + 1] = off;
      off = align_up(off + (size_t)DL_SF_CAP * (size_t)((110 * (sc.kbps / 8) + 3) & ~3), 16);
      lo[3 * sj + 2] = off;
      off += (size_t)DL_SF_CAP * sizeof(dabx_superframe_info);
    }
  off = align_up(off, 256);
  h.off_msc = off;
  if ((D.what & (DABX_DELIVER_MSC | DABX_DELIVER_MSC_NOT_DABPLUS)) && !d.fic_only)
    for (size_t sj = 0; sj < S * M; sj++) {
      const SubchDev &sc = subch_host[sj];
      if (!sc.active || (!(D.what & DABX_DELIVER_MSC) && sc.dab_plus)) continue;
      lo[3 * sj] = off;
      off = align_up(off + (size_t)4 * F * 3 * sc.kbps, 16);
    }
  h.bytes = off;
  if (off > D.capacity) {
    set_error("delivery: the configured sub-channels need %zu bytes per chunk, the slabs hold %zu (sub-channels of a stream that together "
              "exceed a CIF's capacity?)", off, D.capacity);
    return DABX_E_NOMEM;
  }
  D.hdr = h;
  D.bytes = off;
  if (S * M) {
    DABX_HIP(hipMemcpy(D.layout_off, lo.data(), sizeof(unsigned long long) * 3 * S * M, hipMemcpyHostToDevice));
    std::vector<int32_t> ids(subch_id_host.begin(), subch_id_host.begin() + S * M);
    DABX_HIP(hipMemcpy(D.subch_id, ids.data(), sizeof(int32_t) * S * M, hipMemcpyHostToDevice));
  }
  return 0;
}

// A chunk closes (dabx_process, before the MSC batch of its frames is launched): take a free host slab and the next device slab, and
// gather the front end's results of the chunk's frames on the front-end stream.
int dabx_engine::delivery_begin(DeliverDev *dv, int *slot, int *devslab)
{
  Delivery &D = dl;
  int h = -1;
  uint64_t seq;
  {
    std::unique_lock<std::mutex> lk(D.mu);
    if (!D.copier_error.empty()) { set_error("delivery: %s", D.copier_error.c_str()); return DABX_E_HIP; }
    for (size_t i = 0; i < D.slots.size() && h < 0; i++) if (D.slots[i].state == Delivery::FREE) h = (int)i;
    if (h < 0) { set_error("delivery: no free host slab (dabx_delivery_release)"); return DABX_E_STATE; }
    seq = D.next_seq++;
    const int k = (int)(seq % Delivery::NDEV);
    // the copy of chunk seq - NDEV has left the device slab (long ago, unless the link is the bottleneck: then the receiver waits here)
    // (bounded: a device slab that never comes back -- a copier that died -- must fail the call, not hang it)
    if (!D.cv.wait_for(lk, std::chrono::seconds(30), [&]() { return !D.dev_busy[k] || !D.copier_error.empty(); })) {
      D.next_seq--;
      set_error("delivery: device slab %d still busy after 30 s (chunk %llu)", k, (unsigned long long)seq);
      return DABX_E_STATE;
    }
    if (!D.copier_error.empty()) { D.next_seq--; set_error("delivery: %s", D.copier_error.c_str()); return DABX_E_HIP; }
    D.dev_busy[k] = true;
    D.slots[(size_t)h].state = Delivery::IN_FLIGHT;
    D.slots[(size_t)h].seq = seq;
    D.slots[(size_t)h].devslab = k;
    *devslab = k;
  }
  dv->slab = D.dev[*devslab]; dv->layout_off = D.layout_off; dv->subch_id = D.subch_id;
  dv->frames_done = D.frames_done; dv->cif_done = D.cif_done; dv->sf_done = D.sf_done;
  dv->hdr = D.hdr; dv->hdr.seq = seq;
  // two transfers when the slab has a tail of logical frames worth a transfer of its own (SDMA path)
  const size_t lf_from = (size_t)D.hdr.off_msc;
  const bool split = D.copy_engine == 0 && D.bytes > lf_from && D.bytes - lf_from >= ((size_t)1 << 20) && !dev.fic_only && dev.max_subch > 0 && dev.msc_out;
  dv->lf_done = split ? D.packed_lf[*devslab] : nullptr;
  {
    std::lock_guard<std::mutex> lk(D.mu);
    D.slots[(size_t)h].lf_from = split ? lf_from : 0;
  }
  *slot = h;
  const int rc = launch_deliver_front(dev, *dv, stream);
  if (rc) {                                                        // nothing was queued: give the slabs back
    std::lock_guard<std::mutex> lk(D.mu);
    D.dev_busy[*devslab] = false;
    D.slots[(size_t)h].state = Delivery::FREE;
    D.next_seq--;
    D.cv.notify_all();
  }
  return rc;
}

// A chunk that was begun cannot be finished (a launch of its MSC batch failed, or the event below): the host slab and the device slab go
// back, so that neither is lost and no later chunk waits for a copy nobody will make.  The chunk number is given back too (nothing was queued
// for the consumer); what the front gather already wrote into the device slab is overwritten by the next chunk that takes it.  The delivery
// is marked failed: every later call reports why.
void dabx_engine::delivery_abort(int slot, int devslab)
{
  Delivery &D = dl;
  std::lock_guard<std::mutex> lk(D.mu);
  D.slots[(size_t)slot].state = Delivery::FREE;
  D.dev_busy[devslab] = false;
  if (D.next_seq > 0) D.next_seq--;
  if (D.copier_error.empty()) D.copier_error = "a chunk was abandoned after a failed launch: " + std::string(dabx::last_error());
  D.cv.notify_all();
}

// ... and once its slot gather is queued behind the DAB+ stage on `tail`: the copier takes over.
int dabx_engine::delivery_finish(int slot, int devslab, hipStream_t tail)
{
  Delivery &D = dl;
  {
    const hipError_t he = hipEventRecord(D.packed[devslab], tail);
    if (he != hipSuccess) {
      set_error("HIP error %d (%s) at %s:%d", (int)he, hipGetErrorString(he), __FILE__, __LINE__);
      delivery_abort(slot, devslab);
      return DABX_E_HIP;
    }
  }
  std::lock_guard<std::mutex> lk(D.mu);
  D.slots[(size_t)slot].bytes = D.bytes;
  D.queue.push_back(slot);
  D.jobs.push_back(slot);
  D.cv.notify_all();
  return 0;
}

// The copier: one chunk at a time, in order -- wait for the gather kernels, ONE transfer of
// the slab, wait for it, hand the slab to the consumer.
static void delivery_copier(Delivery *Dp)
{
  Delivery &D = *Dp;
  (void)hipSetDevice(D.device);
  for (;;) {
    int h;
    {
      std::unique_lock<std::mutex> lk(D.mu);
      D.cv.wait(lk, [&]() { return D.quit || !D.jobs.empty(); });
      if (D.jobs.empty()) return;                // quit, nothing left to copy
      h = D.jobs.front();
    }
    Delivery::Slot &sl = D.slots[(size_t)h];
    std::string err;
    const auto t_a = std::chrono::steady_clock::now();
    // polled every 50 us, not hipEventSynchronize: see sdma_wait
    hipError_t he;
    // first the slab's tail -- the logical frames, gathered behind the Viterbi decode: on the link while the DAB+ stage and the second gather run
    size_t head_bytes = sl.bytes;
    bool lf_started = false;
    auto t_lf = t_a;
#ifndef DABX_DELIVER_NOCOPY
    if (sl.lf_from) {
      while ((he = hipEventQuery(D.packed_lf[sl.devslab])) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (he != hipSuccess) err = std::string("hipEventQuery: ") + hipGetErrorString(he);
      else if (sdma_copy(D.sdma, sl.host + sl.lf_from, D.dev[sl.devslab] + sl.lf_from, sl.bytes - sl.lf_from, true, sl.sig2)) err = dabx::last_error();
      else { lf_started = true; head_bytes = sl.lf_from; t_lf = std::chrono::steady_clock::now(); }
    }
#endif
    while ((he = hipEventQuery(D.packed[sl.devslab])) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (he != hipSuccess && err.empty()) err = std::string("hipEventQuery: ") + hipGetErrorString(he);
    const auto t_b = std::chrono::steady_clock::now();
#ifndef DABX_DELIVER_NOCOPY            // experiment builds only (tools/build_variant.sh): what the gather kernels alone cost
    if (err.empty()) {
      if (D.copy_engine == 0) {
        if (sdma_copy(D.sdma, sl.host, D.dev[sl.devslab], head_bytes, true, sl.sig) || sdma_wait(sl.sig, head_bytes)) err = dabx::last_error();
        if (lf_started && sdma_wait(sl.sig2, 0) && err.empty()) err = dabx::last_error();
      } else {
        he = hipMemcpyAsync(sl.host, D.dev[sl.devslab], sl.bytes, hipMemcpyDeviceToHost, D.cs);
        if (he == hipSuccess) he = hipStreamSynchronize(D.cs);
        if (he != hipSuccess) err = std::string("hipMemcpyAsync: ") + hipGetErrorString(he);
      }
    }
#endif
    const auto t_c = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lk(D.mu);
    {
      // (two-part transfers: from the start of the first part to the end of the second, the wait for the second gather in between included --
      //  the link rate derived from it is a lower bound)
      const double cs_ = std::chrono::duration<double>(t_c - (lf_started ? t_lf : t_b)).count();
      D.gather_wait_s += std::chrono::duration<double>((lf_started ? t_lf : t_b) - t_a).count();
      D.copy_s += cs_; D.copy_s_max = std::max(D.copy_s_max, cs_);
      D.landed++; D.bytes_copied += sl.bytes;
    }
    D.jobs.pop_front();
    D.dev_busy[sl.devslab] = false;
    sl.state = Delivery::LANDED;                 // (after an error too: nobody may wait for ever; the error is reported by the next call)
    if (!err.empty() && D.copier_error.empty()) D.copier_error = err;
    D.cv.notify_all();
  }
}

// every chunk closed so far has landed in its host slab (dabx_synchronize and everything that drains the engine)
static int delivery_drain(dabx_engine *e)
{
  Delivery &D = e->dl;
  if (!D.open) return 0;
  std::unique_lock<std::mutex> lk(D.mu);
  D.cv.wait(lk, [&]() { return D.jobs.empty(); });
  if (!D.copier_error.empty()) { set_error("delivery: %s", D.copier_error.c_str()); return DABX_E_HIP; }
  return 0;
}

static void delivery_free(dabx_engine *e)
{
  Delivery &D = e->dl;
  if (D.copier.joinable()) {
    { std::lock_guard<std::mutex> lk(D.mu); D.quit = true; D.cv.notify_all(); }
    D.copier.join();
  }
  D.quit = false;
  if (D.cs) (void)hipStreamSynchronize(D.cs);
  for (auto &sl : D.slots) { if (sl.host) (void)hipHostFree(sl.host); sdma_signal_destroy(sl.sig); sdma_signal_destroy(sl.sig2); }
  D.slots.clear();
  D.queue.clear();
  D.jobs.clear();
  for (int k = 0; k < Delivery::NDEV; k++) {
    if (D.dev[k]) (void)hipFree(D.dev[k]);
    if (D.packed[k]) (void)hipEventDestroy(D.packed[k]);
    if (D.packed_lf[k]) (void)hipEventDestroy(D.packed_lf[k]);
    D.dev[k] = nullptr; D.packed[k] = nullptr; D.packed_lf[k] = nullptr; D.dev_busy[k] = false;
  }
  for (void *q : {(void *)D.layout_off, (void *)D.subch_id, (void *)D.frames_done, (void *)D.cif_done, (void *)D.sf_done}) if (q) (void)hipFree(q);
  D.layout_off = nullptr; D.subch_id = nullptr; D.frames_done = D.cif_done = D.sf_done = nullptr;
  if (D.cs) (void)hipStreamDestroy(D.cs);
  D.cs = nullptr;
  D.copier_error.clear();
  D.open = false; D.capacity = D.bytes = 0; D.next_seq = 0;
  D.landed = D.bytes_copied = 0; D.copy_s = D.copy_s_max = D.gather_wait_s = 0;
}

static void ingest_free(dabx_engine *e)
{
  Ingest &I = e->ing;
  for (auto &sl : I.slabs) {
    if (sl.in_flight && I.copy_engine == 0) (void)sdma_wait(sl.sig, 0);
    if (sl.host) (void)hipHostFree(sl.host);
    if (sl.dev) (void)hipFree(sl.dev);
    sdma_signal_destroy(sl.sig);
  }
  I.slabs.clear();
  if (I.cs) { (void)hipStreamSynchronize(I.cs); (void)hipStreamDestroy(I.cs); }
  if (I.committed) (void)hipEventDestroy(I.committed);
  if (I.front) (void)hipEventDestroy(I.front);
  for (void *q : {(void *)I.jobs_dev, (void *)I.counts_dev, (void *)I.work, (void *)I.carry, (void *)I.tab_int, (void *)I.tab_frac}) if (q) (void)hipFree(q);
  if (I.jobs_host) (void)hipHostFree(I.jobs_host);
  if (I.counts_host) (void)hipHostFree(I.counts_host);
  I.jobs_host = nullptr; I.jobs_dev = nullptr; I.counts_host = nullptr; I.counts_dev = nullptr; I.work = I.carry = nullptr; I.tab_int = nullptr; I.tab_frac = nullptr;
  I.general = false; I.dec.clear(); I.M.clear(); I.tab.clear(); I.carry_n.clear(); I.n_bytes.clear();
  I.cs = nullptr; I.committed = nullptr; I.front = nullptr; I.committed_recorded = false; I.open = false; I.capacity = 0;
}

static int need_device_e()
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device available (libdabx has no CPU fallback)"); return DABX_E_NODEVICE; }
  return 0;
}

extern "C" {

void dabx_default_config(dabx_config *c)
{
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->n_streams = 1; c->ring_frames = 4; c->max_subch = 18; c->out_frames = 4;
  c->sync_threshold = 3.0f;            // main/dabradio.cpp:92
  c->sync_strongest = 0;               // configuration.cpp:65
  c->soft_bit_type = 1;                // glob_enums.h:49-56 (SOFTDEC1)
}

int dabx_create(const dabx_config *cfg, dabx_engine **out)
{
  if (!cfg || !out || cfg->n_streams <= 0 || cfg->ring_frames < 2 || cfg->max_subch < 0 || cfg->max_subch > MAX_SUBCH ||
      cfg->out_frames < 1 || cfg->soft_bit_type < 1 || cfg->soft_bit_type > 3 || cfg->dc_iq_correction < 0 || cfg->dc_iq_correction > 2 ||
      cfg->viterbi_tie_mode < 0 || cfg->viterbi_tie_mode > 2 || cfg->schedule < 0 || cfg->schedule > 1 ||
      cfg->msc_fast_min_jobs < 0 || cfg->msc_class_min_jobs < 0 || cfg->exact_level_tracker < 0 || cfg->exact_level_tracker > 2 ||
      cfg->acquire_mode < 0 || cfg->acquire_mode > 2) {
    set_error("dabx_create: bad configuration");
    return DABX_E_ARG;
  }
  int rc = need_device_e();
  if (rc) return rc;
  auto *e = new dabx_engine();
  e->cfg = *cfg;
  // from here on every failure goes through dabx_destroy (streams, events and buffers created so far are released)
#define H(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { set_error("HIP error %d (%s) at %s:%d", (int)err__, hipGetErrorString(err__), __FILE__, __LINE__); dabx_destroy(e); return DABX_E_HIP; } } while (0)
  H(hipGetDevice(&e->device));
  // front end (frame-to-frame feedback = critical path) above the batched MSC decode
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);       // lo = least urgent (numerically greatest)
#ifdef DABX_CU_SPLIT
  // Experiment builds only (tools/build_variant.sh -DDABX_CU_SPLIT=n; docs/history/r01-r04_design_notebook.md 6 "spatial partitioning"): the front-end stream gets
  // n of the 256 CUs, the decoder and the MSC symbols' demapper the other 256 - n.  The KFD deals the bits of a queue's CU mask
  // round-robin to the 8 XCDs, so the first n bits are n / 8 CUs on every XCD.
  uint32_t mask_front[8], mask_back[8];
  for (int w = 0; w < 8; w++) {
    mask_front[w] = mask_back[w] = 0;
    for (int b = 0; b < 32; b++) { if (32 * w + b < DABX_CU_SPLIT) mask_front[w] |= 1u << b; else mask_back[w] |= 1u << b; }
  }
  H(hipExtStreamCreateWithCUMask(&e->stream, 8, mask_front));
#else
  H(hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio_hi));
#endif
  e->ss.a = e->stream;
  if (cfg->schedule == 0) {
    // overlapped schedule (default): MSC batches on b, the MSC symbols' demapper on d (pipeline.hip, launch_front_step /
    // launch_msc_batch).  All streams live on this device: a device-scope release is all a dependency needs (the default
    // system-scope release writes the caches back for host visibility on every record)
#ifdef DABX_CU_SPLIT
    H(hipExtStreamCreateWithCUMask(&e->ss.b, 8, mask_back));
    H(hipExtStreamCreateWithCUMask(&e->ss.d, 8, DABX_CU_SPLIT_DEMAP_FRONT ? mask_front : mask_back));
#else
    H(hipStreamCreateWithPriority(&e->ss.b, hipStreamNonBlocking, prio_lo));
    H(hipStreamCreateWithPriority(&e->ss.d, hipStreamNonBlocking, prio_hi));
#endif
    H(hipEventCreateWithFlags(&e->ss.prep_done, hipEventDisableTiming | hipEventReleaseToDevice));
    H(hipEventCreateWithFlags(&e->ss.msc_done, hipEventDisableTiming | hipEventReleaseToDevice));
    H(hipEventCreateWithFlags(&e->ss.fic_go, hipEventDisableTiming | hipEventReleaseToDevice));
    H(hipEventCreateWithFlags(&e->ss.prep_b_done, hipEventDisableTiming | hipEventReleaseToDevice));
    H(hipEventCreateWithFlags(&e->ss.demap_done, hipEventDisableTiming | hipEventReleaseToDevice));
    H(hipEventCreateWithFlags(&e->ss.sym_done, hipEventDisableTiming | hipEventReleaseToDevice));
    // few streams: the demapper of a frame on stream d in one launch, hand-overs by device-side sequence numbers (pipeline.h, fic_on_d /
    // EngineDev::flag_sync).  The threshold is k_symbols' own (sym_blocks_per_stream: below 48 streams a frame's symbols are spread over more
    // blocks because latency, not throughput, is what is left) -- and the bound under which every block of the kernels that wait for each other
    // is resident at once.  (A/B builds: tools/build_variant.sh -DDABX_FEW_STREAMS=n.)
#ifndef DABX_FEW_STREAMS
#define DABX_FEW_STREAMS 48
#endif
    e->ss.fic_on_d = cfg->n_streams < DABX_FEW_STREAMS;
    // the ingest stream BEFORE q: the runtime deals streams to its hardware queues in order of creation, and as the fifth stream the
    // ingest stream shared one (every synchronous push 20 us = 20 % dearer at 512 streams, tools/bench_ingest.py)
    H(hipStreamCreateWithFlags(&e->ingest, hipStreamNonBlocking));
    // streams out of lock are searched on q next to the steps of the others (dabx_process with sync == 0)
    // With the exact level tracker every stream has a block on q in every step (two lone waves for 1.6 ms): at the lowest queue
    // priority those blocks were dispatched only into the gaps the frame chain's kernels left (5 ms per step); at the chain's own
    // priority they are resident from the start of the step.
        H(hipStreamCreateWithPriority(&e->ss.q, hipStreamNonBlocking, cfg->exact_level_tracker == 1 ? prio_hi : prio_lo));
    H(hipEventCreateWithFlags(&e->ss.acq_done, hipEventDisableTiming | hipEventReleaseToDevice));
    H(hipEventCreateWithFlags(&e->ss.tail_done, hipEventDisableTiming | hipEventReleaseToDevice));
    H(hipEventCreateWithFlags(&e->ss.acq_a_done, hipEventDisableTiming | hipEventReleaseToDevice));
  }
  if (!e->ingest) H(hipStreamCreateWithFlags(&e->ingest, hipStreamNonBlocking));
  // (ingest2 is created by the first dabx_push_iq_async: every HIP stream that exists makes each synchronous push -- a pageable
  // hipMemcpyAsync, an event, a stream wait and a stream synchronisation -- about 20 us dearer, 20 % of a push at 512 streams)
  H(hipEventCreateWithFlags(&e->ingest_done, hipEventDisableTiming | hipEventReleaseToDevice));
  e->rd_seen.assign(cfg->n_streams, 0);
  const int S = cfg->n_streams;
  EngineDev &d = e->dev;
  d.n_streams = S; d.max_subch = cfg->max_subch; d.out_frames = cfg->out_frames;
  d.ring_len = cfg->ring_frames * TF;
  d.threshold = cfg->sync_threshold; d.strongest = cfg->sync_strongest;
  d.fic_only = cfg->fic_only; d.capture_soft = cfg->capture_soft; d.tie_mode = cfg->viterbi_tie_mode;
  d.exact_level = cfg->exact_level_tracker == 1;
  d.anchor_level = cfg->exact_level_tracker == 0;
  const DevTables *t;
  if ((rc = get_tables(&t))) { dabx_destroy(e); return rc; }
#define A(x) if ((rc = (x))) { dabx_destroy(e); return rc; }
  A(e->alloc(&d.iq, (size_t)S * d.ring_len, false));
  A(e->alloc(&d.wr, S));
  A(e->alloc(&d.ctl, S));
  A(e->alloc(&d.spectra, (size_t)2 * S * 75 * K, false));
  A(e->alloc(&d.fsnap, S));
  A(e->alloc(&d.sym_seq, S));
  A(e->alloc(&d.fic_seq, S));
  A(e->alloc(&d.demap_busy, S));
  A(e->alloc(&d.dciq_state, (size_t)S * 8));
  A(e->alloc(&d.dciq_done, S));
  if (d.exact_level) A(e->alloc(&d.level_pos, S));
  H(hipHostMalloc((void **)&e->locked_host, sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
  *e->locked_host = 0;
  H(hipHostMalloc((void **)&e->horizon_host, sizeof(unsigned long long) * S, hipHostMallocMapped | hipHostMallocCoherent));
  for (int s = 0; s < S; s++) e->horizon_host[s] = 0;
  e->announcing.assign(S, 0);
  H(hipHostGetDevicePointer((void **)&d.wr_horizon, e->horizon_host, 0));
  H(hipHostGetDevicePointer((void **)&d.locked_count, e->locked_host, 0));
  H(hipHostMalloc((void **)&e->seq_timeouts_host, sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
  *e->seq_timeouts_host = 0;
  H(hipHostGetDevicePointer((void **)&d.seq_timeouts, e->seq_timeouts_host, 0));
  {
    std::vector<float> st8((size_t)S * 8, 0.0f);                  // sample_reader.h:102-106: meanII = meanQQ = 1
    for (int s_ = 0; s_ < S; s_++) { st8[(size_t)s_ * 8 + 2] = 1.0f; st8[(size_t)s_ * 8 + 3] = 1.0f; }
    H(hipMemcpyAsync(d.dciq_state, st8.data(), sizeof(float) * st8.size(), hipMemcpyHostToDevice, e->stream));
    H(hipStreamSynchronize(e->stream));
  }
  A(e->alloc(&d.nco_tid, (size_t)S * 256));
  A(e->alloc(&d.nco_sym, (size_t)S * 76));
  A(e->alloc(&d.sym_off, (size_t)S * 76));
  A(e->alloc(&e->snap_buf[0], S));
  A(e->alloc(&e->snap_buf[1], S));
  d.snap = e->snap_buf[0];
  A(e->alloc(&d.tii_acc, (size_t)S * TU));
  A(e->alloc(&d.tii_cnt, (size_t)S * 2));
  e->tii.assign((size_t)S, nullptr);
  e->fibdec.assign((size_t)S, nullptr);
  e->fib_frames_fed.assign((size_t)S, 0);
  e->tii_epoch.assign((size_t)S, 0);
  A(e->alloc(&d.cp_part, (size_t)S * 75));
  A(e->alloc(&d.abs_part, (size_t)S * 76));
  A(e->alloc(&d.fic_sym, (size_t)S * 3 * K2));
  A(e->alloc(&d.tdi, (size_t)S * TDI_SLOTS * CIF_BITS));
  A(e->alloc(&d.subch, (size_t)S * std::max(1, d.max_subch)));
  A(e->alloc(&d.fib_out, (size_t)S * d.out_frames * 12 * 32));
  A(e->alloc(&d.fib_crc, (size_t)S * d.out_frames * 12));
  A(e->alloc(&d.frame_pos, (size_t)S * d.out_frames));
  A(e->alloc(&d.frame_start, (size_t)S * d.out_frames));
  if (d.capture_soft) A(e->alloc(&d.soft_cap, (size_t)S * 75 * K2));
  A(e->alloc(&d.sf_info, (size_t)S * std::max(1, d.max_subch) * SF_SLOTS));
  // demapper state (constructor defaults: ofdm_decoder.h:101-104)
  A(demap_alloc(d.demap, S));
  d.demap.soft_type = cfg->soft_bit_type;
  A(launch_demap_init(d.demap, e->stream));
  // per-stream scalars: SampleReader / DabProcessor defaults (sample_reader.h:95,101; dab_processor.h:129-138)
  std::vector<StreamCtl> ctl(S);
  for (auto &c : ctl) {
    memset(&c, 0, sizeof(c));
    c.state = ST_INIT; c.s_level = 0.1f; c.peak_level = -1.0e6f; c.sync_thr = cfg->sync_threshold;
    c.lvl_anchor_S = 0.1f;
  }
  H(hipMemcpyAsync(d.ctl, ctl.data(), sizeof(StreamCtl) * S, hipMemcpyHostToDevice, e->stream));
  H(hipStreamSynchronize(e->stream));
  e->wr_host.assign(S, 0);
  e->subch_host.assign((size_t)S * std::max(1, d.max_subch), SubchDev{});
  e->subch_id_host.assign((size_t)S * std::max(1, d.max_subch), -1);
  e->eti.assign((size_t)S, dabx_engine::EtiCursor{});
  // scratch sized for the FIC now; re-sized when sub-channels are configured
  d.vit_stride = (int)vit_scratch_words(FIC_OUT);
  A(e->alloc(&d.vit_scratch, (size_t)S * (4 + 4 * MSC_BATCH_FRAMES * d.max_subch) * d.vit_stride, false));
  d.msc_stride = 0; d.sf_stride = 0;
#undef A
#undef H
  *out = e;
  return 0;
}

void dabx_destroy(dabx_engine *e)
{
  if (!e) return;
  (void)use_device(e);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->ss.b) { (void)hipStreamSynchronize(e->ss.b); (void)hipStreamDestroy(e->ss.b); }
  if (e->ss.fic_go) (void)hipEventDestroy(e->ss.fic_go);
  if (e->ss.prep_b_done) (void)hipEventDestroy(e->ss.prep_b_done);
  if (e->ss.d) { (void)hipStreamSynchronize(e->ss.d); (void)hipStreamDestroy(e->ss.d); }
  if (e->ss.demap_done) (void)hipEventDestroy(e->ss.demap_done);
  if (e->ss.sym_done) (void)hipEventDestroy(e->ss.sym_done);
  if (e->ss.q) { (void)hipStreamSynchronize(e->ss.q); (void)hipStreamDestroy(e->ss.q); }
  if (e->ss.acq_done) (void)hipEventDestroy(e->ss.acq_done);
  if (e->ss.tail_done) (void)hipEventDestroy(e->ss.tail_done);
  if (e->ss.acq_a_done) (void)hipEventDestroy(e->ss.acq_a_done);
  if (e->ss.prep_done) (void)hipEventDestroy(e->ss.prep_done);
  if (e->ss.msc_done) (void)hipEventDestroy(e->ss.msc_done);
  delivery_free(e);
  ingest_free(e);
  for (dabx_tii *t : e->tii) dabx_tii_destroy(t);
  for (dabx_fibdec *f : e->fibdec) dabx_fibdec_destroy(f);
  if (e->ingest) { (void)hipStreamSynchronize(e->ingest); (void)hipStreamDestroy(e->ingest); }
  if (e->ingest2) { (void)hipStreamSynchronize(e->ingest2); (void)hipStreamDestroy(e->ingest2); }
  if (e->ingest_done) (void)hipEventDestroy(e->ingest_done);
  if (e->stage) (void)hipFree(e->stage);
  for (int i = 0; i < dabx_engine::ASYNC_SLOTS; i++) {
    if (e->aslot[i]) (void)hipFree(e->aslot[i]);
    if (e->aslot_done[i]) (void)hipEventDestroy(e->aslot_done[i]);
  }
  for (void *p : e->allocs) (void)hipFree(p);
  if (e->locked_host) (void)hipHostFree(e->locked_host);
  if (e->seq_timeouts_host) (void)hipHostFree(e->seq_timeouts_host);
  if (e->horizon_host) (void)hipHostFree(e->horizon_host);
  for (auto &ev : e->mk.pool) (void)hipEventDestroy(ev);
  demap_free(e->dev.demap);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

static int set_subchannels_impl(dabx_engine *e, int stream, const dabx_subch_desc *desc, int n, long long at_cif /* < 0: the next CIF */)
{
  if (!e || n < 0 || n > e->dev.max_subch || (n > 0 && !desc) || stream >= e->dev.n_streams) {
    set_error("dabx_set_subchannels: bad argument");
    return DABX_E_ARG;
  }
  EngineDev &d = e->dev;
  int rc;
  if ((rc = sync_all(e))) return rc;
  // current CIF counters (Backend construction time, backend.cpp:38-70)
  std::vector<StreamCtl> ctl(d.n_streams);
  DABX_HIP(hipMemcpy(ctl.data(), d.ctl, sizeof(StreamCtl) * d.n_streams, hipMemcpyDeviceToHost));
  if (at_cif >= 0 && (stream < 0 || at_cif < ctl[stream].cif_no || at_cif > ctl[stream].cif_no + 3)) {
    set_error("dabx_set_subchannels_at: CIF %lld is not in the coming frame of stream %d (next CIF %lld)", at_cif, stream,
              stream < 0 ? -1ll : ctl[stream].cif_no);
    return DABX_E_ARG;
  }
  // refresh the host mirror: the device owns the dynamic fields (cif_out, super-frame state, counters)
  DABX_HIP(hipMemcpy(e->subch_host.data(), d.subch, sizeof(SubchDev) * e->subch_host.size(), hipMemcpyDeviceToHost));
  int max_kbps = e->max_kbps;
  std::vector<SubchDev> row(std::max(1, d.max_subch));
  for (int j = 0; j < n; j++) {
    const dabx_subch_desc &q = desc[j];
    SubchDev sc{};
    if (q.kbps == 0) { row[j] = sc; continue; }            // empty slot (keeps the indices of the others stable)
    const uint16_t *map = nullptr;
    int n_in = 0;
    if ((rc = get_profile_map(q.kbps, q.prot_level, q.short_form, &map, &n_in))) return rc;
    if (q.cu_size * 64 < n_in || q.cu_start < 0 || q.cu_start + q.cu_size > 864) {
      set_error("sub-channel %d: %d CU at %d do not hold %d coded bits", j, q.cu_size, q.cu_start, n_in);
      return DABX_E_PROFILE;
    }
    // the DAB+ stage (k_dabplus) holds a super frame of at most 384 kbit/s (RS interleaving depth kbps / 8 <= 48) in LDS;
    // ETSI TS 102 563 defines DAB+ sub-channels in multiples of 8 kbit/s only
    if (q.dab_plus && (q.kbps > 384 || q.kbps % 8 != 0)) {
      set_error("sub-channel %d: %d kbit/s is not a DAB+ rate (multiples of 8 up to 384); configure it with dab_plus = 0", j, q.kbps);
      return DABX_E_PROFILE;
    }
    sc.cu_start = q.cu_start; sc.cu_size = q.cu_size; sc.kbps = q.kbps; sc.prot_level = q.prot_level;
    sc.short_form = q.short_form; sc.dab_plus = q.dab_plus; sc.nbits = 24 * q.kbps; sc.active = 1; sc.map = map;
    row[j] = sc;
    max_kbps = std::max(max_kbps, q.kbps);
  }
  // When the largest bit rate grows the output rings get wider slots.  Running services are not disturbed
  // (MscHandler::set_channel only adds a Backend, msc_handler.cpp:95-131): the rings are re-strided with their contents,
  // every counter and ring index stays valid, the superseded buffers are freed.
  if (max_kbps > e->max_kbps) {
    // new buffers first; pointers, strides and max_kbps change only after all three exist and the contents are moved, so a
    // failed allocation leaves the engine exactly as it was
    const int new_msc = 3 * max_kbps, new_sf = ((110 * max_kbps / 8) + 15) & ~15;
    const int new_vit = (int)std::max(vit_scratch_words(FIC_OUT), vit_scratch_words(24 * max_kbps));
    const size_t msc_rows = (size_t)d.n_streams * d.max_subch * MSC_SLOTS, sf_rows = (size_t)d.n_streams * d.max_subch * SF_SLOTS;
    uint8_t *n_msc = nullptr, *n_sf = nullptr;
    uint32_t *n_scratch = nullptr;
    auto drop = [&](void *q) { if (q) { (void)hipFree(q); e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), q), e->allocs.end()); } };
    if ((rc = e->alloc(&n_msc, msc_rows * new_msc)) || (rc = e->alloc(&n_sf, sf_rows * new_sf)) ||
        (rc = e->alloc(&n_scratch, (size_t)d.n_streams * (4 + 4 * MSC_BATCH_FRAMES * d.max_subch) * new_vit, false))) {
      drop(n_msc); drop(n_sf); drop(n_scratch);
      return rc;
    }
    hipError_t herr = hipSuccess;
    if (d.msc_out && d.msc_stride > 0)
      herr = hipMemcpy2DAsync(n_msc, new_msc, d.msc_out, d.msc_stride, d.msc_stride, msc_rows, hipMemcpyDeviceToDevice, e->stream);
    if (herr == hipSuccess && d.sf_out && d.sf_stride > 0)
      herr = hipMemcpy2DAsync(n_sf, new_sf, d.sf_out, d.sf_stride, d.sf_stride, sf_rows, hipMemcpyDeviceToDevice, e->stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(e->stream);
    if (herr != hipSuccess) {
      drop(n_msc); drop(n_sf); drop(n_scratch);
      set_error("dabx_set_subchannels: HIP error %d (%s) while re-striding the output rings", (int)herr, hipGetErrorString(herr));
      return DABX_E_HIP;
    }
    drop(d.msc_out); drop(d.sf_out); drop(d.vit_scratch);
    d.msc_out = n_msc; d.sf_out = n_sf; d.vit_scratch = n_scratch;
    d.msc_stride = new_msc; d.sf_stride = new_sf; d.vit_stride = new_vit;
    e->max_kbps = max_kbps;
  }
  std::vector<size_t> restarted;
  for (int s = 0; s < d.n_streams; s++) {
    if (stream >= 0 && s != stream) continue;
    for (int j = 0; j < d.max_subch; j++) {
      SubchDev sc = j < n ? row[j] : SubchDev{};
      sc.start_cif = at_cif >= 0 ? at_cif : ctl[s].cif_no;
      // a slot whose description does not change keeps running (MscHandler::set_channel only adds a Backend,
      // msc_handler.cpp:95-131): its de-interleaver history, super-frame state and counters stay
      const SubchDev &old = e->subch_host[(size_t)s * d.max_subch + j];
      const bool same = old.active && sc.active && old.cu_start == sc.cu_start && old.cu_size == sc.cu_size &&
                        old.kbps == sc.kbps && old.prot_level == sc.prot_level && old.short_form == sc.short_form &&
                        old.dab_plus == sc.dab_plus && e->subch_id_host[(size_t)s * d.max_subch + j] == desc[j].subch_id;
      if (same) continue;
      // A sub-channel that only MOVES (same SubChId, size, bit rate, protection; other capacity units -- a multiplex reconfiguration):
      // the slot keeps running, its de-interleaver reads the CIFs before the change at the old address (what a Backend that is handed
      // its slice from another place does, msc_handler.cpp:161-166).  One move per 16 CIFs; anything faster restarts the slot.
      const bool moved = old.active && sc.active && old.cu_start != sc.cu_start && old.cu_size == sc.cu_size && old.kbps == sc.kbps &&
                         old.prot_level == sc.prot_level && old.short_form == sc.short_form && old.dab_plus == sc.dab_plus &&
                         e->subch_id_host[(size_t)s * d.max_subch + j] == desc[j].subch_id && sc.start_cif >= old.move_cif + 16;
      if (moved) {
        SubchDev keep = old;
        keep.prev_cu_start = old.cu_start; keep.cu_start = sc.cu_start; keep.move_cif = sc.start_cif;
        e->subch_host[(size_t)s * d.max_subch + j] = keep;
        continue;
      }
      e->subch_host[(size_t)s * d.max_subch + j] = sc;
      e->subch_id_host[(size_t)s * d.max_subch + j] = (j < n && sc.active) ? desc[j].subch_id : -1;
      e->eti[s] = dabx_engine::EtiCursor{};
      restarted.push_back((size_t)s * d.max_subch + j);
    }
  }
  DABX_HIP(hipMemcpy(d.subch, e->subch_host.data(), sizeof(SubchDev) * e->subch_host.size(), hipMemcpyHostToDevice));
  if (e->dl.open) {
    // slots that start anew count their frames from 0 again; the slab layout follows the new sub-channels (engine drained above)
    const long long zero = 0;
    for (size_t sj : restarted) {
      DABX_HIP(hipMemcpy(e->dl.cif_done + sj, &zero, sizeof(zero), hipMemcpyHostToDevice));
      DABX_HIP(hipMemcpy(e->dl.sf_done + sj, &zero, sizeof(zero), hipMemcpyHostToDevice));
    }
    if ((rc = e->delivery_layout())) {             // the new sub-channels do not fit the slabs: no gather may run with a stale layout
      const std::string why = dabx::last_error();
      delivery_free(e);
      set_error("%s -- the delivery has been closed", why.c_str());
      return rc;
    }
  }
  e->have_fast = false;
  e->classes_dirty = true;            // the decoder classes are rebuilt by the next dabx_process (one rebuild for a series of per-stream calls)
  return 0;
}

int dabx_set_subchannels(dabx_engine *e, int stream, const dabx_subch_desc *desc, int n) { return set_subchannels_impl(e, stream, desc, n, -1); }
int dabx_set_subchannels_at(dabx_engine *e, int stream, const dabx_subch_desc *desc, int n, int64_t at_cif)
{
  if (at_cif < 0 || stream < 0) { set_error("dabx_set_subchannels_at: bad argument"); return DABX_E_ARG; }
  return set_subchannels_impl(e, stream, desc, n, at_cif);
}

int dabx_iq_ring_dev(dabx_engine *e, int stream, void **ring, size_t *cap)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !ring) return DABX_E_ARG;
  *ring = e->dev.iq + (size_t)stream * e->dev.ring_len;
  if (cap) *cap = (size_t)e->dev.ring_len;
  return 0;
}

// What a push may overwrite is announced BEFORE its copy is issued (EngineDev::wr_horizon, host memory the device reads): the level
// tracker's re-walk from its anchor (k_acquire) only trusts samples at or above horizon - ring_len, and looks again after the walk.
static void announce_write(dabx_engine *e, int stream, unsigned long long upto)
{
  if (!e->horizon_host) return;
  for (int s = 0; s < e->dev.n_streams; s++)
    if ((stream < 0 || s == stream) && e->horizon_host[s] < upto) __atomic_store_n(&e->horizon_host[s], upto, __ATOMIC_RELEASE);
}
static int commit_impl(dabx_engine *e, int stream, size_t n);
int dabx_commit_iq(dabx_engine *e, int stream, size_t n)
{
  if (!e || stream >= e->dev.n_streams) return DABX_E_ARG;
  // a zero-copy producer writes into the ring on its own: unless it has said how far (dabx_announce_write), nothing behind the read
  // cursor can be taken for intact from here on
  if (e->horizon_host)
    for (int s = 0; s < e->dev.n_streams; s++)
      if ((stream < 0 || s == stream) && !e->announcing[s]) __atomic_store_n(&e->horizon_host[s], ~0ull, __ATOMIC_RELEASE);
  return commit_impl(e, stream, n);
}
int dabx_announce_write(dabx_engine *e, int stream, size_t n)
{
  if (!e || stream >= e->dev.n_streams) { set_error("dabx_announce_write: bad argument"); return DABX_E_ARG; }
  if (!e->horizon_host) return 0;
  for (int s = 0; s < e->dev.n_streams; s++)
    if (stream < 0 || s == stream) {
      // from its first announcement on the producer is taken at its word: commits no longer mean "unknown writes" (a ring that is
      // filled once and only read again -- periodic test signals -- is announced once)
      const unsigned long long prev = e->announcing[s] ? e->horizon_host[s] : 0ull;
      e->announcing[s] = 1;
      __atomic_store_n(&e->horizon_host[s], std::max(prev, e->wr_host[s] + (unsigned long long)n), __ATOMIC_RELEASE);
    }
  return 0;
}
// iqfile.cpp: samples it converted into the ring itself, while nothing was running (dabx_internal_ring_info drains the engine)
int dabx_internal_commit(dabx_engine *e, int stream, size_t n)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams) return DABX_E_ARG;
  announce_write(e, stream, e->wr_host[stream] + n);
  return commit_impl(e, stream, n);
}
static int commit_impl(dabx_engine *e, int stream, size_t n)
{
  if (!e || stream >= e->dev.n_streams) return DABX_E_ARG;
  if (int rc = use_device(e)) return rc;
  for (int s = 0; s < e->dev.n_streams; s++)
    if (stream < 0 || s == stream) e->wr_host[s] += n;
  if (int rc = launch_commit(e->dev, stream, n, e->stream)) return rc;
  // SampleReader's DC / IQ correction (off by default): the new samples are corrected in place before anything reads them
  if (e->cfg.dc_iq_correction) return launch_dciq(e->dev, e->cfg.dc_iq_correction, e->stream);
  return 0;
}

// never overwrite samples the receiver has not read yet.  rd only grows, so the value seen at the last look is a safe
// bound: the pipeline is drained (and rd read again) only when that bound says the ring is full
static int push_room(dabx_engine *e, int stream, size_t n, const char *who)
{
  for (int attempt = 0; e->wr_host[stream] - e->rd_seen[stream] + n > (unsigned long long)e->dev.ring_len; attempt++) {
    if (attempt == 2) {
      set_error("%s: ring of stream %d has room for %llu samples, %zu offered (call dabx_process first)", who, stream,
                (unsigned long long)e->dev.ring_len - (e->wr_host[stream] - e->rd_seen[stream]), n);
      return DABX_E_STATE;
    }
    // first a look at the counters while the receiver keeps running (any value read is a valid lower bound), then,
    // if that is not enough, with the pipeline drained
    if (attempt == 1) { if (int rc0 = sync_all(e)) return rc0; }
    e->ctl_peek.resize(e->dev.n_streams);
    DABX_HIP(hipMemcpy(e->ctl_peek.data(), e->dev.ctl, sizeof(StreamCtl) * e->dev.n_streams, hipMemcpyDeviceToHost));
    if (e->dev.exact_level && e->dev.level_pos) {          // the exact level tracker still has to read what lies behind ITS cursor
      std::vector<unsigned long long> lp(e->dev.n_streams);
      DABX_HIP(hipMemcpy(lp.data(), e->dev.level_pos, sizeof(unsigned long long) * lp.size(), hipMemcpyDeviceToHost));
      for (int s = 0; s < e->dev.n_streams; s++) e->ctl_peek[s].rd = std::min(e->ctl_peek[s].rd, lp[s]);
    }
    for (int s = 0; s < e->dev.n_streams; s++) e->rd_seen[s] = std::max(e->rd_seen[s], e->ctl_peek[s].rd);
  }
  return 0;
}

int dabx_push_iq(dabx_engine *e, int stream, const void *iq, int fmt, size_t n)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !iq || fmt < 0 || fmt > 2 || n > (size_t)e->dev.ring_len) {
    set_error("dabx_push_iq: bad argument");
    return DABX_E_ARG;
  }
  if (n == 0) return 0;
  if (int rc = use_device(e)) return rc;
  if (int rc = push_room(e, stream, n, "dabx_push_iq")) return rc;
  static const int bps[3] = {8, 4, 2};
  const size_t bytes = n * bps[fmt];
  if (bytes > e->stage_cap) {                   // one staging buffer per engine, grown on demand
    if (e->stage) DABX_HIP(hipFree(e->stage));
    e->stage = nullptr; e->stage_cap = 0;
    DABX_HIP(hipMalloc(&e->stage, bytes));
    e->stage_cap = bytes;
  }
  // copy + conversion on the ingest stream, next to whatever the receiver streams are computing: the samples land
  // beyond the committed write index, which no queued kernel reads; only the commit is ordered into the front-end stream
  announce_write(e, stream, e->wr_host[stream] + n);
  DABX_HIP(hipMemcpyAsync(e->stage, iq, bytes, hipMemcpyHostToDevice, e->ingest));
  int rc = launch_convert_iq(e->stage, fmt, e->dev.iq + (size_t)stream * e->dev.ring_len, e->dev.ring_len, e->wr_host[stream], n, e->ingest);
  if (rc) return rc;
  DABX_HIP(hipEventRecord(e->ingest_done, e->ingest));
  DABX_HIP(hipStreamWaitEvent(e->stream, e->ingest_done, 0));
  rc = commit_impl(e, stream, n);
  DABX_HIP(hipStreamSynchronize(e->ingest));   // the caller's buffer and the staging buffer are free again
  return rc;
}

// The same without waiting for the copy: for producers that keep their buffers alive and unchanged until dabx_push_wait --
// a file reader cycling through a few pinned buffers (hipHostMalloc / dabx_host_register).  From pinned memory the copies
// of consecutive calls run back to back as DMA at PCIe rate while the host already issues the next ones; from pageable
// memory the HIP runtime stages the copy itself and the call degrades gracefully to the synchronous behaviour.
int dabx_push_iq_async(dabx_engine *e, int stream, const void *iq, int fmt, size_t n)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !iq || fmt < 0 || fmt > 2 || n > (size_t)e->dev.ring_len) {
    set_error("dabx_push_iq_async: bad argument");
    return DABX_E_ARG;
  }
  if (n == 0) return 0;
  if (int rc = use_device(e)) return rc;
  if (int rc = push_room(e, stream, n, "dabx_push_iq_async")) return rc;
  static const int bps[3] = {8, 4, 2};
  const size_t bytes = n * bps[fmt];
  const int k = (int)(e->async_pushes++ % dabx_engine::ASYNC_SLOTS);
  if (!e->aslot_done[k]) DABX_HIP(hipEventCreateWithFlags(&e->aslot_done[k], hipEventDisableTiming));
  else DABX_HIP(hipEventSynchronize(e->aslot_done[k]));              // the slot's previous conversion has read it
  if (bytes > e->aslot_cap[k]) {
    if (e->aslot[k]) DABX_HIP(hipFree(e->aslot[k]));
    e->aslot[k] = nullptr; e->aslot_cap[k] = 0;
    DABX_HIP(hipMalloc(&e->aslot[k], bytes));
    e->aslot_cap[k] = bytes;
  }
  if (!e->ingest2) DABX_HIP(hipStreamCreateWithFlags(&e->ingest2, hipStreamNonBlocking));
  hipStream_t ing = (k & 1) ? e->ingest2 : e->ingest;
  announce_write(e, stream, e->wr_host[stream] + n);
  DABX_HIP(hipMemcpyAsync(e->aslot[k], iq, bytes, hipMemcpyHostToDevice, ing));
  int rc = launch_convert_iq(e->aslot[k], fmt, e->dev.iq + (size_t)stream * e->dev.ring_len, e->dev.ring_len, e->wr_host[stream], n, ing);
  if (rc) return rc;
  DABX_HIP(hipEventRecord(e->aslot_done[k], ing));
  DABX_HIP(hipStreamWaitEvent(e->stream, e->aslot_done[k], 0));      // the commit (and every frame after it) sees the samples
  return commit_impl(e, stream, n);
}

int dabx_push_wait(dabx_engine *e)
{
  if (!e) return DABX_E_ARG;
  if (int rc = use_device(e)) return rc;
  DABX_HIP(hipStreamSynchronize(e->ingest));
  if (e->ingest2) DABX_HIP(hipStreamSynchronize(e->ingest2));
  return 0;
}

// hipHostRegister / hipHostUnregister for a producer's own buffers (page-locks them so that pushes are true DMA)
int dabx_host_register(void *p, size_t bytes)
{
  if (!p || !bytes) return DABX_E_ARG;
  if (int rc = need_device_e()) return rc;
  DABX_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return 0;
}
int dabx_host_unregister(void *p)
{
  if (!p) return DABX_E_ARG;
  if (int rc = need_device_e()) return rc;
  DABX_HIP(hipHostUnregister(p));
  return 0;
}

int dabx_read_iq(dabx_engine *e, int stream, uint64_t first, size_t n, float *iq_out)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !iq_out) return DABX_E_ARG;
  const unsigned long long wr = e->wr_host[stream];
  if (first + n > wr || wr - first > (unsigned long long)e->dev.ring_len) { set_error("dabx_read_iq: samples not in the ring"); return DABX_E_STATE; }
  if (int rc = sync_all(e)) return rc;
  const float2 *ring = e->dev.iq + (size_t)stream * e->dev.ring_len;
  size_t done = 0;
  while (done < n) {
    const size_t o = (size_t)((first + done) % (unsigned long long)e->dev.ring_len);
    const size_t take = std::min(n - done, (size_t)e->dev.ring_len - o);
    DABX_HIP(hipMemcpy(iq_out + 2 * done, ring + o, take * sizeof(float2), hipMemcpyDeviceToHost));
    done += take;
  }
  return 0;
}

// internal hook of iqfile.cpp (not part of include/dabx.h)
int dabx_internal_ring_info(dabx_engine *e, int stream, float2 **ring, int *ring_len, unsigned long long *wr, unsigned long long *rd, hipStream_t *st)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams) { set_error("bad engine / stream"); return DABX_E_ARG; }
  if (int rc = sync_all(e)) return rc;
  StreamCtl c;
  DABX_HIP(hipMemcpy(&c, e->dev.ctl + stream, sizeof(StreamCtl), hipMemcpyDeviceToHost));
  *ring = e->dev.iq + (size_t)stream * e->dev.ring_len; *ring_len = e->dev.ring_len;
  *wr = e->wr_host[stream]; *rd = c.rd; *st = e->stream;
  return 0;
}

int dabx_process(dabx_engine *e, int max_frames, int sync)
{
  if (!e || max_frames < 0) return DABX_E_ARG;
  if (int rc = use_device(e)) return rc;
  // The front end (sync, FFT, demap, FIC) has frame-to-frame feedback and runs once per frame; the MSC decoder
  // has none, so its CIFs are decoded MSC_BATCH_FRAMES frames at a time (more trellises per launch) and always
  // before this call returns.
  if (e->classes_dirty) {
    if (int rc = e->build_msc_classes()) return rc;
    e->classes_dirty = false;
  }
  // Streams out of lock are searched next to the steps, on their own HIP stream (k_acquire on q): a step of the streams in lock never waits
  // for a stream in a drop-out -- with sync != 0 too: the call then waits for the frame chain it issued, not for the search pass beside it
  // (a pass costs ~2 ms, two steps of 512 streams).  cfg.acquire_mode 1 / 2 fixes either form.
  // (With cfg.dc_iq_correction the committed samples are corrected in place on the front-end stream before anything reads them:
  // a search running next to that stream could read them uncorrected, so it stays in step.)
  // And while fewer than half of the streams are in lock (start-up of a whole engine; the device keeps the count in host memory, read
  // here without a wait) the search runs in step: there is little to hold up, and streams that start together lock together instead of
  // falling behind their producers while nearly empty steps go by.  (A stream that joins late stays late: a step never advances a
  // stream by more than one frame.)
  if (e->dl.open && max_frames > 0) {
    // every chunk this call closes needs a free host slab; checked before anything is launched, so that a refused call changes nothing
    const int need = (e->pending_frames + max_frames + MSC_BATCH_FRAMES - 1) / MSC_BATCH_FRAMES;
    int free_slots = 0;
    {
      std::lock_guard<std::mutex> lk(e->dl.mu);
      for (const auto &sl : e->dl.slots) free_slots += sl.state == Delivery::FREE;
    }
    if (free_slots < need) {
      set_error("dabx_process: the call closes %d chunks, %d host slabs are free (dabx_delivery_next / dabx_delivery_release)", need, free_slots);
      return DABX_E_STATE;
    }
  }
  const bool some_locked = !e->locked_host || 2 * __atomic_load_n(e->locked_host, __ATOMIC_RELAXED) >= e->dev.n_streams;
  const bool async_acquire = !e->cfg.dc_iq_correction && (e->cfg.acquire_mode == 2 || (e->cfg.acquire_mode == 0 && some_locked));
  for (int i = 0; i < max_frames; i++) {
    // the 5th frame after a batch starts rewriting time-de-interleaver slots the previous batch's k_msc_prep (stream b) reads
    if (e->ss.prep_pending && e->pending_frames >= 4) {
      DABX_HIP(hipStreamWaitEvent(e->stream, e->ss.prep_b_done, 0));
      e->ss.prep_pending = false;
    }
    const bool all_locked = e->locked_host && __atomic_load_n(e->locked_host, __ATOMIC_RELAXED) == e->dev.n_streams;
    int rc = launch_front_step(e->dev, e->ss, e->mk, async_acquire, all_locked);
    if (rc) return rc;
    e->level_dirty = true;
    if (++e->pending_frames == MSC_BATCH_FRAMES || i == max_frames - 1) {
      DeliverDev dv{};
      int dl_slot = -1, dl_dev = -1;
      if (e->dl.open && (rc = e->delivery_begin(&dv, &dl_slot, &dl_dev))) return rc;
      e->dev.snap = e->snap_buf[e->ss.batch_parity];
      hipStream_t tail = e->stream;
      rc = launch_msc_batch(e->dev, 4 * e->pending_frames, e->have_fast ? &e->fast : nullptr, e->ss, e->mk, e->dl.open ? &dv : nullptr, &tail);
      if (rc) {
        if (e->dl.open) e->delivery_abort(dl_slot, dl_dev);          // the slabs of the chunk that was begun: never left IN_FLIGHT without a copy job
        e->pending_frames = 0;
        return rc;
      }
      if (e->dl.open && (rc = e->delivery_finish(dl_slot, dl_dev, tail))) return rc;
      e->pending_frames = 0;
    }
  }
  if (sync && (max_frames = sync_all(e, async_acquire) ? -1 : max_frames) < 0) return DABX_E_HIP;
  return max_frames;
}

int dabx_synchronize(dabx_engine *e)
{
  if (!e) return DABX_E_ARG;
  return sync_all(e);
}
void *dabx_hip_stream(dabx_engine *e) { return e ? (void *)e->stream : nullptr; }

static int fetch_ctl(dabx_engine *e, int stream, StreamCtl *c)
{
  if (int rc = sync_all(e)) return rc;
  DABX_HIP(hipMemcpy(c, e->dev.ctl + stream, sizeof(StreamCtl), hipMemcpyDeviceToHost));
  return 0;
}

int dabx_read_fibs(dabx_engine *e, int stream, int n_frames, uint8_t *fibs, uint8_t *crc)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || n_frames <= 0 || n_frames > e->dev.out_frames || !fibs || !crc) return DABX_E_ARG;
  StreamCtl c;
  int rc = fetch_ctl(e, stream, &c);
  if (rc) return rc;
  const int have = (int)std::min<long long>(c.frames, n_frames);
  for (int i = 0; i < have; i++) {            // oldest first
    const long long fr = c.frames - have + i;
    const int slot = (int)(fr % e->dev.out_frames);
    DABX_HIP(hipMemcpy(fibs + (size_t)i * 384, e->dev.fib_out + ((size_t)stream * e->dev.out_frames + slot) * 384, 384, hipMemcpyDeviceToHost));
    DABX_HIP(hipMemcpy(crc + (size_t)i * 12, e->dev.fib_crc + ((size_t)stream * e->dev.out_frames + slot) * 12, 12, hipMemcpyDeviceToHost));
  }
  return have;
}

int dabx_read_frame_info(dabx_engine *e, int stream, int n_frames, int64_t *sym0_pos, int32_t *start_index)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || n_frames <= 0 || n_frames > e->dev.out_frames || (!sym0_pos && !start_index)) return DABX_E_ARG;
  if (!e->dev.frame_pos) { set_error("dabx_read_frame_info: engine keeps no frame records"); return DABX_E_STATE; }
  StreamCtl c;
  int rc = fetch_ctl(e, stream, &c);
  if (rc) return rc;
  const int have = (int)std::min<long long>(c.frames, n_frames);
  for (int i = 0; i < have; i++) {            // oldest first
    const long long fr = c.frames - have + i;
    const size_t slot = (size_t)stream * e->dev.out_frames + (size_t)(fr % e->dev.out_frames);
    if (sym0_pos) DABX_HIP(hipMemcpy(sym0_pos + i, e->dev.frame_pos + slot, sizeof(int64_t), hipMemcpyDeviceToHost));
    if (start_index) DABX_HIP(hipMemcpy(start_index + i, e->dev.frame_start + slot, sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return have;
}

static int fetch_subch(dabx_engine *e, int stream, int j, SubchDev *sc)
{
  if (int rc = sync_all(e)) return rc;
  DABX_HIP(hipMemcpy(sc, e->dev.subch + (size_t)stream * e->dev.max_subch + j, sizeof(SubchDev), hipMemcpyDeviceToHost));
  return 0;
}

int dabx_read_msc(dabx_engine *e, int stream, int j, int n_cifs, uint8_t *bytes)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || j < 0 || j >= e->dev.max_subch || n_cifs <= 0 || n_cifs > MSC_SLOTS || !bytes) return DABX_E_ARG;
  SubchDev sc;
  int rc = fetch_subch(e, stream, j, &sc);
  if (rc) return rc;
  if (!sc.active) return 0;
  const int have = (int)std::min<long long>(sc.cif_out, n_cifs), nb = 3 * sc.kbps;
  for (int i = 0; i < have; i++) {
    const long long q = sc.cif_out - have + i;
    DABX_HIP(hipMemcpy(bytes + (size_t)i * nb,
                       e->dev.msc_out + (((size_t)stream * e->dev.max_subch + j) * MSC_SLOTS + (size_t)(q % MSC_SLOTS)) * e->dev.msc_stride,
                       nb, hipMemcpyDeviceToHost));
  }
  return have;
}

int dabx_read_superframes(dabx_engine *e, int stream, int j, int n, uint8_t *bytes)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || j < 0 || j >= e->dev.max_subch || n <= 0 || n > SF_SLOTS || !bytes) return DABX_E_ARG;
  SubchDev sc;
  int rc = fetch_subch(e, stream, j, &sc);
  if (rc) return rc;
  if (!sc.active) return 0;
  const int have = (int)std::min<long long>(sc.sf_count, n), nb = 110 * sc.kbps / 8;
  for (int i = 0; i < have; i++) {
    const long long q = sc.sf_count - have + i;
    DABX_HIP(hipMemcpy(bytes + (size_t)i * nb,
                       e->dev.sf_out + (((size_t)stream * e->dev.max_subch + j) * SF_SLOTS + (size_t)(q % SF_SLOTS)) * e->dev.sf_stride,
                       nb, hipMemcpyDeviceToHost));
  }
  return have;
}

int dabx_read_superframe_info(dabx_engine *e, int stream, int j, int n, dabx_superframe_info *out)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || j < 0 || j >= e->dev.max_subch || n <= 0 || n > SF_SLOTS || !out) return DABX_E_ARG;
  SubchDev sc;
  int rc = fetch_subch(e, stream, j, &sc);
  if (rc) return rc;
  if (!sc.active || !e->dev.sf_info) return 0;
  const int have = (int)std::min<long long>(sc.sf_count, n);
  std::vector<dabx_superframe_info> ring(SF_SLOTS);
  DABX_HIP(hipMemcpy(ring.data(), e->dev.sf_info + ((size_t)stream * e->dev.max_subch + j) * SF_SLOTS, sizeof(dabx_superframe_info) * SF_SLOTS,
                     hipMemcpyDeviceToHost));
  for (int i = 0; i < have; i++) out[i] = ring[(size_t)((sc.sf_count - have + i) % SF_SLOTS)];
  return have;
}

int dabx_read_eti(dabx_engine *e, int stream, int max_frames, uint8_t *out, int32_t *lost_cifs)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || max_frames < 0 || (!out && max_frames)) return DABX_E_ARG;
  if (lost_cifs) *lost_cifs = 0;
  if (e->dev.fic_only) { set_error("dabx_read_eti: engine was created FIC-only"); return DABX_E_STATE; }
  StreamCtl c;
  int rc = fetch_ctl(e, stream, &c);
  if (rc) return rc;
  const EngineDev &d = e->dev;
  std::vector<SubchDev> row(std::max(1, d.max_subch));
  DABX_HIP(hipMemcpy(row.data(), d.subch + (size_t)stream * d.max_subch, sizeof(SubchDev) * d.max_subch, hipMemcpyDeviceToHost));
  std::vector<int> act;
  for (int j = 0; j < d.max_subch; j++) if (row[j].active) act.push_back(j);
  // CIFs [lo_cif, hi_cif) are complete in both rings
  long long hi_cif = act.empty() ? c.cif_no : c.msc_done_cif, lo_cif = std::max(0ll, (c.frames - d.out_frames) * 4);
  for (int j : act) lo_cif = std::max(lo_cif, row[j].start_cif + 16 + std::max(0ll, row[j].cif_out - MSC_SLOTS));
  dabx_engine::EtiCursor &cur = e->eti[stream];
  if (cur.next_cif < 0) { cur.next_cif = lo_cif; cur.fib_frames_seen = lo_cif / 4; }
  if (cur.next_cif < lo_cif) {
    if (lost_cifs) *lost_cifs = (int32_t)(lo_cif - cur.next_cif);
    cur.next_cif = lo_cif;
  }
  cur.fib_frames_seen = std::max(cur.fib_frames_seen, std::max(0ll, c.frames - d.out_frames));
  int n = 0;
  std::vector<uint8_t> fibs(384), msc((size_t)act.size() * std::max(1, d.msc_stride));
  std::vector<dabx_subch_desc> desc(act.size());
  std::vector<const uint8_t *> ptr(act.size());
  for (size_t a = 0; a < act.size(); a++) {
    const SubchDev &sc = row[act[a]];
    desc[a] = dabx_subch_desc{e->subch_id_host[(size_t)stream * d.max_subch + act[a]], sc.cu_start, sc.cu_size, sc.kbps, sc.prot_level, sc.short_form, sc.dab_plus, 0};
    ptr[a] = msc.data() + a * (size_t)d.msc_stride;
  }
  long long fib_frame = -1;
  while (n < max_frames && cur.next_cif < hi_cif) {
    const long long r = cur.next_cif, F = r / 4;
    // FibDecoder state at symbol 4 of frame F: every FIB up to and including this frame's 12 has been parsed
    while (cur.fib_frames_seen <= F) {
      const long long G = cur.fib_frames_seen;
      std::vector<uint8_t> fb(384), fc(12);
      const size_t slot = (size_t)stream * d.out_frames + (size_t)(G % d.out_frames);
      DABX_HIP(hipMemcpy(fb.data(), d.fib_out + slot * 384, 384, hipMemcpyDeviceToHost));
      DABX_HIP(hipMemcpy(fc.data(), d.fib_crc + slot * 12, 12, hipMemcpyDeviceToHost));
      for (int i = 0; i < 12; i++) if (fc[i]) fib_cif_count(fb.data() + 32 * i, &cur.hi, &cur.lo);
      if (G == F) { fibs = fb; fib_frame = F; }
      cur.fib_frames_seen++;
    }
    if (fib_frame != F) {
      const size_t slot = (size_t)stream * d.out_frames + (size_t)(F % d.out_frames);
      DABX_HIP(hipMemcpy(fibs.data(), d.fib_out + slot * 384, 384, hipMemcpyDeviceToHost));
      fib_frame = F;
    }
    cur.next_cif++;
    if (cur.hi < 0 || cur.lo < 0) continue;              // eti_generator.cpp:156-160: no FIG 0/0 yet
    for (size_t a = 0; a < act.size(); a++) {
      const SubchDev &sc = row[act[a]];
      desc[a].cu_start = r < sc.move_cif ? sc.prev_cu_start : sc.cu_start;     // a sub-channel that moved: the address the FIC of CIF r gave it
      const long long lf = r - sc.start_cif - 16;
      DABX_HIP(hipMemcpy(msc.data() + a * (size_t)d.msc_stride,
                         d.msc_out + (((size_t)stream * d.max_subch + act[a]) * MSC_SLOTS + (size_t)(lf % MSC_SLOTS)) * d.msc_stride,
                         (size_t)3 * sc.kbps, hipMemcpyDeviceToHost));
    }
    rc = dabx_eti_frame(cur.hi, cur.lo, (int)(r & 3), desc.data(), (int)desc.size(), fibs.data() + 96 * (r & 3), ptr.data(), out + (size_t)n * 6144);
    if (rc < 0) return rc;
    n++;
  }
  return n;
}

int dabx_read_tii(dabx_engine *e, int stream, int min_frames, int threshold_db, int collisions, int collision_sub_id,
                  dabx_tii_result *out, int max_out, int32_t *frames_accumulated)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || (!out && max_out > 0) || max_out < 0) return DABX_E_ARG;
  if (!e->dev.tii_acc) { set_error("dabx_read_tii: engine has no TII accumulator"); return DABX_E_STATE; }
  if (int rc = sync_all(e)) return rc;
  int32_t cnt[2];
  DABX_HIP(hipMemcpy(cnt, e->dev.tii_cnt + 2 * stream, sizeof(cnt), hipMemcpyDeviceToHost));
  if (frames_accumulated) *frames_accumulated = cnt[0];
  dabx_tii *&t = e->tii[(size_t)stream];
  if (!t) { if (int rc = dabx_tii_create(&t)) return rc; e->tii_epoch[(size_t)stream] = cnt[1]; }
  if (cnt[1] != e->tii_epoch[(size_t)stream]) { dabx_tii_reset(t); e->tii_epoch[(size_t)stream] = cnt[1]; }   // lock was lost meanwhile
  if (cnt[0] < std::max(1, min_frames)) return 0;
  std::vector<float> acc(2 * (size_t)TU);
  DABX_HIP(hipMemcpy(acc.data(), e->dev.tii_acc + (size_t)stream * TU, sizeof(float2) * TU, hipMemcpyDeviceToHost));
  DABX_HIP(hipMemset(e->dev.tii_acc + (size_t)stream * TU, 0, sizeof(float2) * TU));
  DABX_HIP(hipMemset(e->dev.tii_cnt + 2 * stream, 0, sizeof(int32_t)));
  dabx_tii_set_collisions(t, collisions, collision_sub_id);
  dabx_tii_add(t, acc.data());
  return dabx_tii_process(t, threshold_db, out, max_out);
}

int dabx_read_soft(dabx_engine *e, int stream, int16_t *soft)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !soft) return DABX_E_ARG;
  if (!e->dev.soft_cap) { set_error("engine was created without capture_soft"); return DABX_E_STATE; }
  if (int rc = sync_all(e)) return rc;
  DABX_HIP(hipMemcpy(soft, e->dev.soft_cap + (size_t)stream * 75 * K2, sizeof(int16_t) * 75 * K2, hipMemcpyDeviceToHost));
  return 0;
}

int dabx_discover_subchannels(dabx_engine *e, int stream, dabx_subch_desc *out, int max_out)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !out || max_out <= 0) return DABX_E_ARG;
  const int nf = e->dev.out_frames;
  std::vector<uint8_t> fibs((size_t)nf * 384), crc((size_t)nf * 12);
  const int have = dabx_read_fibs(e, stream, nf, fibs.data(), crc.data());
  if (have < 0) return have;
  return dabx_parse_fibs(fibs.data(), crc.data(), have * 12, out, max_out, nullptr);
}

int dabx_set_fig_reference_quirks(dabx_engine *e, int on)
{
  if (!e) return DABX_E_ARG;
  e->fig_reference_quirks = on != 0;
  for (dabx_fibdec *fd : e->fibdec) if (fd) (void)dabx_fibdec_set_reference_quirks(fd, on);
  return 0;
}

int dabx_follow_fic(dabx_engine *e, int stream, dabx_reconf *out)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !out) return DABX_E_ARG;
  memset(out, 0, sizeof(*out));
  out->at_cif = out->last_change_cif = -1;
  StreamCtl c;
  int rc = fetch_ctl(e, stream, &c);
  if (rc) return rc;
  dabx_fibdec *&fd = e->fibdec[(size_t)stream];
  if (!fd) {
    if ((rc = dabx_fibdec_create(&fd))) return rc;
    if (e->fig_reference_quirks) (void)dabx_fibdec_set_reference_quirks(fd, 1);
  }
  long long &fed = e->fib_frames_fed[(size_t)stream];
  const EngineDev &d = e->dev;
  if (c.frames - fed > d.out_frames) {                      // frames that have left the FIB ring: their FIGs are lost to the decoder
    out->frames_missed = (int32_t)std::min<long long>(c.frames - fed - d.out_frames, 0x7fffffff);
    fed = c.frames - d.out_frames;
  }
  // FIB k of frame f is FIB 12 f + k of the stream: the decoder counts the FIBs of missed frames as skipped
  dabx_fibdec_info inf;
  dabx_fibdec_get_info(fd, &inf);
  if (inf.fibs_processed < 12 * fed) dabx_internal_fibdec_skip(fd, 12 * fed - inf.fibs_processed);      // counted, not processed
  std::vector<uint8_t> fb(384), fc(12);
  for (; fed < c.frames; fed++) {
    const size_t slot = (size_t)stream * d.out_frames + (size_t)(fed % d.out_frames);
    DABX_HIP(hipMemcpy(fb.data(), d.fib_out + slot * 384, 384, hipMemcpyDeviceToHost));
    DABX_HIP(hipMemcpy(fc.data(), d.fib_crc + slot * 12, 12, hipMemcpyDeviceToHost));
    dabx_fibdec_process(fd, fb.data(), fc.data(), 12);
  }
  dabx_fibdec_get_info(fd, &inf);
  out->frames_fed = fed;
  out->n_changes = inf.n_changes;
  // FIB i of the stream belongs to frame i / 12 and, within its FIC, to the CIF (i % 12) / 3 (three FIBs per CIF in Mode I)
  auto cif_of_fib = [](long long i) { return 4 * (i / 12) + (i % 12) / 3; };
  if (inf.last_change_fib >= 0) out->last_change_cif = cif_of_fib(inf.last_change_fib);
  if (inf.change_flags != 0 && inf.fig00_fib >= 0) {
    out->pending = 1;
    // the announcing FIG 0/0 carried the counter of ITS CIF: the change applies (occurrence - lo) mod 250 CIFs later.  While the flags
    // are set the change has not happened yet: a distance of 0 is a full turn of the low counter (250 CIFs = the 6 s of lead)
    int ahead = ((inf.occurrence_change - inf.cif_count_lo) % 250 + 250) % 250;
    if (ahead == 0) ahead = 250;
    out->at_cif = cif_of_fib(inf.fig00_fib) + ahead;
  }
  return 0;
}

int dabx_next_subchannels(dabx_engine *e, int stream, dabx_subch_desc *out, int max_out)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !out || max_out <= 0) return DABX_E_ARG;
  dabx_fibdec *fd = e->fibdec[(size_t)stream];
  if (!fd) { set_error("dabx_next_subchannels: call dabx_follow_fic first"); return DABX_E_STATE; }
  return dabx_fibdec_subchannels(fd, 1, out, max_out);
}

int dabx_current_subchannels(dabx_engine *e, int stream, dabx_subch_desc *out, int max_out)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !out || max_out <= 0) return DABX_E_ARG;
  dabx_fibdec *fd = e->fibdec[(size_t)stream];
  if (!fd) { set_error("dabx_current_subchannels: call dabx_follow_fic first"); return DABX_E_STATE; }
  return dabx_fibdec_subchannels(fd, 0, out, max_out);
}

#undef dabx_get_stats
static int get_stats_full(dabx_engine *e, int stream, dabx_stats *out);
// The entry point binaries built against ABI 3 call: writes exactly the ABI-3 record (up to and including peak_level), so a
// caller whose dabx_stats is the old, shorter one is not overrun.  Sources compiled against this header reach
// dabx_get_stats_sized through the macro of the same name and get everything their record has room for.
int dabx_get_stats(dabx_engine *e, int stream, dabx_stats *out)
{
  return dabx_get_stats_sized(e, stream, out, offsetof(dabx_stats, peak_level) + sizeof(float));
}
int dabx_get_stats_sized(dabx_engine *e, int stream, void *out, size_t size)
{
  if (!out || size < sizeof(int64_t)) return DABX_E_ARG;
  dabx_stats full;
  if (int rc = get_stats_full(e, stream, &full)) return rc;
  memcpy(out, &full, std::min(size, sizeof(full)));
  if (size > sizeof(full)) memset((char *)out + sizeof(full), 0, size - sizeof(full));
  return 0;
}
static int get_stats_full(dabx_engine *e, int stream, dabx_stats *out)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || !out) return DABX_E_ARG;
  StreamCtl c;
  int rc = fetch_ctl(e, stream, &c);
  if (rc) return rc;
  memset(out, 0, sizeof(*out));
  out->level_margin_events = c.level_margin;
  out->level_rewalk_events = c.lvl_rewalks; out->level_unanchored_events = c.lvl_unanchored; out->level_healed_events = c.lvl_healed;
  out->frames = c.frames; out->samples_consumed = (int64_t)c.rd; out->state = c.state;
  out->fic_ratio_percent = c.fic_ratio * 10; out->freq_offs_bb_hz = c.f_bb; out->clock_err_hz = c.clock_err;
  out->snr_db_est = c.snr_db; out->mer_db_est = c.mer_db; out->last_start_index = c.start_index; out->cif_count = c.cif_count;
  out->fib_ok = c.fib_ok; out->fib_total = c.fib_total;
  out->signal_level = c.s_level; out->peak_level = c.peak_level;
  out->fic_ber_bits = c.fic_bits; out->fic_ber_errors = c.fic_errors;
  std::vector<SubchDev> sc(std::max(1, e->dev.max_subch));
  DABX_HIP(hipMemcpy(sc.data(), e->dev.subch + (size_t)stream * e->dev.max_subch, sizeof(SubchDev) * e->dev.max_subch, hipMemcpyDeviceToHost));
  for (int j = 0; j < e->dev.max_subch; j++) {
    out->sf_ok += sc[j].sf_ok; out->sf_fail += sc[j].sf_fail; out->rs_corrected += sc[j].rs_corr; out->rs_failed += sc[j].rs_fail;
    out->au_ok += sc[j].au_ok; out->au_bad += sc[j].au_bad; out->cifs_decoded += sc[j].cif_out;
  }
  return 0;
}

int dabx_get_subch_stats(dabx_engine *e, int stream, int j, dabx_subch_stats *out)
{
  if (!e || stream < 0 || stream >= e->dev.n_streams || j < 0 || j >= e->dev.max_subch || !out) return DABX_E_ARG;
  SubchDev sc;
  int rc = fetch_subch(e, stream, j, &sc);
  if (rc) return rc;
  *out = dabx_subch_stats{sc.start_cif, sc.cif_out, sc.sf_count, sc.sf_ok, sc.sf_fail, sc.rs_corr, sc.rs_fail, sc.fc_corr, sc.au_ok,
                          sc.au_bad, sc.active, e->subch_id_host[(size_t)stream * e->dev.max_subch + j]};
  return 0;
}

int dabx_get_counters(dabx_engine *e, int64_t out[16])
{
  if (!e || !out) return DABX_E_ARG;
  if (int rc0 = sync_all(e)) return rc0;
  const int S = e->dev.n_streams;
  std::vector<StreamCtl> ctl(S);
  std::vector<SubchDev> sc((size_t)S * std::max(1, e->dev.max_subch));
  DABX_HIP(hipMemcpy(ctl.data(), e->dev.ctl, sizeof(StreamCtl) * S, hipMemcpyDeviceToHost));
  DABX_HIP(hipMemcpy(sc.data(), e->dev.subch, sizeof(SubchDev) * sc.size(), hipMemcpyDeviceToHost));
  memset(out, 0, sizeof(int64_t) * 16);
  for (auto &c : ctl) {
    out[0] += c.frames; out[1] += (int64_t)c.rd; out[2] += c.fib_ok; out[3] += c.fib_total; out[4] += c.sync_lost;
    out[5] += (c.state == ST_EVAL_SYNC);
  }
  for (auto &q : sc) {
    out[6] += q.cif_out; out[7] += q.sf_ok; out[8] += q.sf_fail; out[9] += q.rs_corr; out[10] += q.rs_fail;
    out[11] += q.fc_corr; out[12] += q.au_ok; out[13] += q.au_bad;
    out[14] += (int64_t)q.cif_out * 3 * q.kbps;                 // MSC bytes out
  }
  return 0;
}

static int ingest_open_impl(dabx_engine *e, const dabx_ingest_config *cfg, const dabx_iq_format *formats);
int dabx_ingest_open(dabx_engine *e, const dabx_ingest_config *cfg) { return ingest_open_impl(e, cfg, nullptr); }
int dabx_ingest_open_formats(dabx_engine *e, const dabx_ingest_config *cfg, const dabx_iq_format *formats)
{
  if (!formats) { set_error("dabx_ingest_open_formats: bad argument"); return DABX_E_ARG; }
  return ingest_open_impl(e, cfg, formats);
}
long long dabx_ingest_pitch(dabx_engine *e)
{
  if (!e) return DABX_E_ARG;
  if (!e->ing.open) { set_error("dabx_ingest_pitch: no ingest open"); return DABX_E_STATE; }
  return (long long)(e->ing.general ? e->ing.pitch : e->ing.capacity / (size_t)e->dev.n_streams);
}

static int ingest_open_impl(dabx_engine *e, const dabx_ingest_config *cfg, const dabx_iq_format *formats)
{
  if (!e || (cfg && (cfg->host_slabs < 0 || cfg->host_slabs > 64 || cfg->fmt < 0 || cfg->fmt > 2 || cfg->max_frames < 0 || cfg->copy_engine < 0 || cfg->copy_engine > 1))) {
    set_error("dabx_ingest_open: bad argument");
    return DABX_E_ARG;
  }
  if (e->ing.open) { set_error("dabx_ingest_open: already open"); return DABX_E_STATE; }
  if (int rc = use_device(e)) return rc;
  Ingest &I = e->ing;
  I.fmt = cfg ? cfg->fmt : 0;
  I.copy_engine = cfg ? cfg->copy_engine : 0;
  I.max_frames = cfg && cfg->max_frames ? cfg->max_frames : DL_FRAMES;
  if ((long long)I.max_frames * TF > e->dev.ring_len) { set_error("dabx_ingest_open: a slab of %d frames does not fit the ring (%d frames)", I.max_frames, e->dev.ring_len / TF); return DABX_E_ARG; }
  static const int bps[3] = {8, 4, 2};
  I.capacity = (size_t)e->dev.n_streams * I.max_frames * TF * bps[I.fmt];
  int rc;
  const int S_ = e->dev.n_streams;
  std::vector<int16_t> tabs_i; std::vector<float> tabs_f;
  int m_max = 0;
  if (formats) {
    // every stream's own recording: the region of a slab that holds max_frames frames' worth of ITS payload (+ one read block) sets the pitch
    I.general = true;
    I.dec.assign((size_t)S_, IqDecode{}); I.M.assign((size_t)S_, 0); I.tab.assign((size_t)S_, 0); I.carry_n.assign((size_t)S_, 0);
    std::map<std::pair<int, int>, int> tab_of;
    size_t need = 0;
    for (int s = 0; s < S_; s++) {
      if ((rc = iq_check_format(&formats[s], &I.dec[(size_t)s]))) { ingest_free(e); return rc; }
      const int rate = formats[s].sample_rate;
      if (rate != INPUT_RATE) {
        const auto key = std::make_pair((int)formats[s].family, rate);
        if (!tab_of.count(key)) {
          tab_of[key] = (int)tab_of.size();
          tabs_i.resize(tabs_i.size() + 2048); tabs_f.resize(tabs_f.size() + 2048);
          int m = 0;
          iq_resample_tables(formats[s].family, rate, &m, tabs_i.data() + tabs_i.size() - 2048, tabs_f.data() + tabs_f.size() - 2048);
        }
        I.tab[(size_t)s] = tab_of[key];
        I.M[(size_t)s] = rate / 1000;
        I.carry_n[(size_t)s] = formats[s].family == DABX_FAMILY_UFF ? 1 : 0;     // xml_reader.cpp:84-85,226: convBuffer[0] starts as a zero sample
        m_max = std::max(m_max, rate / 1000);
      }
      const size_t in_per_frame = (size_t)((long long)TF * (rate / 1000) / 2048) + (size_t)(rate / 1000);
      need = std::max(need, ((size_t)I.max_frames * in_per_frame + (size_t)(rate / 1000)) * 2 * (size_t)I.dec[(size_t)s].bytes);
    }
    I.pitch = align_up(need, 256);
    I.capacity = I.pitch * (size_t)S_;
  }
  if (I.copy_engine == 0 && (rc = sdma_open(e->device, &I.sdma))) { ingest_free(e); return rc; }      // (the per-stream tables above go with it)
#define H(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { set_error("HIP error %d (%s) at %s:%d", (int)err__, hipGetErrorString(err__), __FILE__, __LINE__); ingest_free(e); return DABX_E_HIP; } } while (0)
  if (I.copy_engine == 1) H(hipStreamCreateWithFlags(&I.cs, hipStreamNonBlocking));
  H(hipEventCreateWithFlags(&I.committed, hipEventDisableTiming | hipEventReleaseToDevice));
  H(hipEventCreateWithFlags(&I.front, hipEventDisableTiming | hipEventReleaseToDevice));
  I.slabs.resize((size_t)(cfg && cfg->host_slabs ? cfg->host_slabs : 2));
  if (I.general) {
    I.n_bytes.assign(I.slabs.size(), std::vector<size_t>((size_t)S_, 0));
    H(hipHostMalloc((void **)&I.jobs_host, sizeof(IngestJob) * (size_t)S_, hipHostMallocDefault));
    H(hipHostMalloc((void **)&I.counts_host, sizeof(unsigned) * (size_t)S_, hipHostMallocDefault));
    H(hipMalloc((void **)&I.jobs_dev, sizeof(IngestJob) * (size_t)S_));
    H(hipMalloc((void **)&I.counts_dev, sizeof(unsigned) * (size_t)S_));
    if (m_max) {
      // [carry | decoded samples of one slab] per resampling stream, and the carry between slabs (<= M + 1 samples)
      size_t max_in = 0;
      for (int s = 0; s < S_; s++) if (I.M[(size_t)s]) max_in = std::max(max_in, I.pitch / (size_t)(2 * I.dec[(size_t)s].bytes));
      I.work_pitch = align_up(max_in + (size_t)m_max + 2, 64);
      I.carry_pitch = align_up((size_t)m_max + 2, 64);
      H(hipMalloc((void **)&I.work, sizeof(float2) * I.work_pitch * (size_t)S_));
      H(hipMalloc((void **)&I.carry, sizeof(float2) * I.carry_pitch * (size_t)S_));
      H(hipMemset(I.carry, 0, sizeof(float2) * I.carry_pitch * (size_t)S_));
      H(hipMalloc((void **)&I.tab_int, tabs_i.size() * sizeof(int16_t)));
      H(hipMalloc((void **)&I.tab_frac, tabs_f.size() * sizeof(float)));
      H(hipMemcpy(I.tab_int, tabs_i.data(), tabs_i.size() * sizeof(int16_t), hipMemcpyHostToDevice));
      H(hipMemcpy(I.tab_frac, tabs_f.data(), tabs_f.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  for (auto &sl : I.slabs) {
    H(hipHostMalloc((void **)&sl.host, I.capacity, hipHostMallocDefault));
    H(hipMalloc((void **)&sl.dev, I.capacity));
    if (I.copy_engine == 0 && (rc = sdma_signal_create(&sl.sig))) { ingest_free(e); return rc; }
  }
#undef H
  if (I.copy_engine == 0 && I.capacity >= ((size_t)16 << 20) && (rc = sdma_calibrate(I.sdma, I.slabs[0].host, I.slabs[0].dev, false, I.slabs[0].sig, nullptr))) {
    ingest_free(e);
    return rc;
  }
  I.open = true;
  return 0;
}

int dabx_ingest_close(dabx_engine *e)
{
  if (!e) return DABX_E_ARG;
  if (!e->ing.open) return 0;
  const int rc = sync_all(e);
  ingest_free(e);
  return rc;
}

int dabx_ingest_slab(dabx_engine *e, int k, void **host, size_t *capacity_bytes)
{
  if (!e || !host) return DABX_E_ARG;
  if (!e->ing.open || k < 0 || k >= (int)e->ing.slabs.size()) { set_error("dabx_ingest_slab: no such slab"); return DABX_E_STATE; }
  *host = e->ing.slabs[(size_t)k].host;
  if (capacity_bytes) *capacity_bytes = e->ing.capacity;
  return 0;
}

int dabx_ingest_submit(dabx_engine *e, int k, size_t n)
{
  if (!e) return DABX_E_ARG;
  Ingest &I = e->ing;
  if (!I.open || k < 0 || k >= (int)I.slabs.size()) { set_error("dabx_ingest_submit: no such slab"); return DABX_E_STATE; }
  if (I.general) { set_error("dabx_ingest_submit: this ingest was opened with per-stream formats (dabx_ingest_submit_bytes)"); return DABX_E_STATE; }
  if (n == 0 || n > (size_t)I.max_frames * TF) { set_error("dabx_ingest_submit: %zu samples per stream, the slabs hold %d frames", n, I.max_frames); return DABX_E_ARG; }
  if (int rc = use_device(e)) return rc;
  Ingest::Slab &sl = I.slabs[(size_t)k];
  if (sl.in_flight) { set_error("dabx_ingest_submit: slab %d has a transfer that was not committed", k); return DABX_E_STATE; }
  static const int bps[3] = {8, 4, 2};
  const size_t bytes = (size_t)e->dev.n_streams * n * bps[I.fmt];
  // (the device twin is free: its converter ran on the ingest stream before the commit that cleared in_flight was queued, and a slab is
  //  only reused after its commit -- by then, with two slabs, a whole chunk later)
  DABX_HIP(hipStreamSynchronize(e->ingest));
  if (I.copy_engine == 0) { if (int rc = sdma_copy(I.sdma, sl.dev, sl.host, bytes, false, sl.sig)) return rc; }
  else DABX_HIP(hipMemcpyAsync(sl.dev, sl.host, bytes, hipMemcpyHostToDevice, I.cs));
  sl.n = n; sl.in_flight = true;
  return 0;
}

int dabx_ingest_submit_bytes(dabx_engine *e, int k, const size_t *n_bytes)
{
  if (!e || !n_bytes) return DABX_E_ARG;
  Ingest &I = e->ing;
  if (!I.open || !I.general || k < 0 || k >= (int)I.slabs.size()) { set_error("dabx_ingest_submit_bytes: no such slab of an ingest opened with dabx_ingest_open_formats"); return DABX_E_STATE; }
  if (int rc = use_device(e)) return rc;
  Ingest::Slab &sl = I.slabs[(size_t)k];
  if (sl.in_flight) { set_error("dabx_ingest_submit_bytes: slab %d has a transfer that was not committed", k); return DABX_E_STATE; }
  size_t last = 0;
  for (int s = 0; s < e->dev.n_streams; s++) {
    const IqDecode &d = I.dec[(size_t)s];
    const size_t unit = (size_t)(2 * d.bytes) * (d.quirk_block ? (size_t)d.quirk_block : 1);
    if (n_bytes[s] > I.pitch || n_bytes[s] % unit) {
      set_error("dabx_ingest_submit_bytes: stream %d: %zu bytes -- at most %zu, whole samples%s only (a reader keeps the odd tail for its next slab)", s, n_bytes[s], I.pitch,
                d.quirk_block ? " and whole 1-ms read blocks" : "");
      return DABX_E_ARG;
    }
    if (n_bytes[s]) last = (size_t)s * I.pitch + n_bytes[s];
  }
  I.n_bytes[(size_t)k].assign(n_bytes, n_bytes + e->dev.n_streams);
  DABX_HIP(hipStreamSynchronize(e->ingest));
  // ONE transfer, up to the last byte any stream uses (the regions of streams that end early travel as they are)
  if (last) {
    if (I.copy_engine == 0) { if (int rc = sdma_copy(I.sdma, sl.dev, sl.host, last, false, sl.sig)) return rc; }
    else DABX_HIP(hipMemcpyAsync(sl.dev, sl.host, last, hipMemcpyHostToDevice, I.cs));
  }
  sl.n = last; sl.in_flight = true;
  return 0;
}

// general form: per stream its own decode, resampling state and sample count; two launches for all streams together
static int ingest_commit_general(dabx_engine *e, int k)
{
  Ingest &I = e->ing;
  Ingest::Slab &sl = I.slabs[(size_t)k];
  const int S = e->dev.n_streams;
  const std::vector<size_t> &nb = I.n_bytes[(size_t)k];
  std::vector<IngestJob> jobs((size_t)S);
  unsigned max_n = 0, max_out = 0;
  // the page-locked staging records (jobs_host, counts_host) are written below: the previous commit's asynchronous copies of them -- two
  // commits may follow each other without a submit in between -- have to be through first
  DABX_HIP(hipStreamSynchronize(e->ingest));
  for (int s = 0; s < S; s++) {
    IngestJob &j = jobs[(size_t)s];
    j = IngestJob{};
    j.dec = I.dec[(size_t)s];
    j.src_off = (unsigned long long)s * I.pitch;
    j.n = (unsigned)(nb[(size_t)s] / (size_t)(2 * j.dec.bytes));
    j.M = (unsigned)I.M[(size_t)s]; j.tab = (unsigned)I.tab[(size_t)s]; j.carry_n = (unsigned)I.carry_n[(size_t)s];
    unsigned produced = j.n;
    if (j.M && j.n) {                                  // feed_push's bookkeeping (iqfile.cpp): block c needs V[c M .. c M + M]
      const unsigned len = j.carry_n + j.n;
      j.blocks = len >= j.M + 1 ? (len - 1) / j.M : 0;
      j.keep = len - j.blocks * j.M;
      produced = j.blocks * 2048;
      max_out = std::max(max_out, produced);
    }
    I.counts_host[s] = j.n ? produced : 0;
    max_n = std::max(max_n, j.n);
    if (int rc = push_room(e, s, I.counts_host[s], "dabx_ingest_commit")) return rc;   // (the transfer stays pending: process, then commit again)
  }
  if (sl.n) {
    if (I.copy_engine == 0) { if (int rc = sdma_wait(sl.sig, 0)) return rc; }
    else DABX_HIP(hipStreamSynchronize(I.cs));
  }
  for (int s = 0; s < S; s++) {
    jobs[(size_t)s].dst0 = e->wr_host[s];               // the host's own count of committed samples: no device-side index is read
    announce_write(e, s, e->wr_host[s] + I.counts_host[s]);
  }
  memcpy(I.jobs_host, jobs.data(), sizeof(IngestJob) * (size_t)S);
  DABX_HIP(hipMemcpyAsync(I.jobs_dev, I.jobs_host, sizeof(IngestJob) * (size_t)S, hipMemcpyHostToDevice, e->ingest));
  DABX_HIP(hipMemcpyAsync(I.counts_dev, I.counts_host, sizeof(unsigned) * (size_t)S, hipMemcpyHostToDevice, e->ingest));
  IngestMulti m{};
  m.slab = sl.dev; m.jobs = I.jobs_dev; m.iq = e->dev.iq; m.ring_len = e->dev.ring_len; m.work = I.work; m.work_pitch = I.work_pitch;
  m.carry = I.carry; m.carry_pitch = I.carry_pitch; m.tab_int = I.tab_int; m.tab_frac = I.tab_frac;
  if (int rc = launch_ingest_multi(m, S, max_n, max_out, e->ingest)) return rc;
  DABX_HIP(hipEventRecord(e->ingest_done, e->ingest));
  DABX_HIP(hipStreamWaitEvent(e->stream, e->ingest_done, 0));
  for (int s = 0; s < S; s++) {
    e->wr_host[s] += I.counts_host[s];
    if (jobs[(size_t)s].M && jobs[(size_t)s].n) I.carry_n[(size_t)s] = (int)jobs[(size_t)s].keep;
  }
  if (int rc = launch_commit_counts(e->dev.wr, I.counts_dev, S, e->stream)) return rc;
  if (e->cfg.dc_iq_correction) { if (int rc = launch_dciq(e->dev, e->cfg.dc_iq_correction, e->stream)) return rc; }
  sl.in_flight = false;
  return 0;
}

int dabx_ingest_commit(dabx_engine *e, int k)
{
  if (!e) return DABX_E_ARG;
  Ingest &I = e->ing;
  if (!I.open || k < 0 || k >= (int)I.slabs.size()) { set_error("dabx_ingest_commit: no such slab"); return DABX_E_STATE; }
  if (int rc = use_device(e)) return rc;
  Ingest::Slab &sl = I.slabs[(size_t)k];
  if (!sl.in_flight) { set_error("dabx_ingest_commit: slab %d was not submitted", k); return DABX_E_STATE; }
  if (I.general) return ingest_commit_general(e, k);
  for (int s = 0; s < e->dev.n_streams; s++)
    if (int rc = push_room(e, s, sl.n, "dabx_ingest_commit")) return rc;          // (the transfer stays pending: process, then commit again)
  if (I.copy_engine == 0) { if (int rc = sdma_wait(sl.sig, 0)) return rc; }
  else DABX_HIP(hipStreamSynchronize(I.cs));
  for (int s = 0; s < e->dev.n_streams; s++) announce_write(e, s, e->wr_host[s] + sl.n);
  // the converter reads the committed indices on the device: behind EVERYTHING the front-end stream has been given so far -- the previous
  // ingest commit, and a dabx_push_iq / dabx_commit_iq of another entry point in between (their index updates run on that stream too)
  DABX_HIP(hipEventRecord(I.front, e->stream));
  DABX_HIP(hipStreamWaitEvent(e->ingest, I.front, 0));
  if (int rc = launch_ingest_convert(e->dev, sl.dev, I.fmt, sl.n, e->ingest)) return rc;
  DABX_HIP(hipEventRecord(e->ingest_done, e->ingest));
  DABX_HIP(hipStreamWaitEvent(e->stream, e->ingest_done, 0));
  const int rc = commit_impl(e, -1, sl.n);
  DABX_HIP(hipEventRecord(I.committed, e->stream));
  I.committed_recorded = true;
  sl.in_flight = false;
  return rc;
}

int dabx_delivery_open(dabx_engine *e, const dabx_delivery_config *cfg)
{
  if (!e || (cfg && (cfg->host_slabs < 0 || cfg->host_slabs == 1 || cfg->host_slabs > 64 || (cfg->what & ~15) || cfg->copy_engine < 0 || cfg->copy_engine > 1))) {
    set_error("dabx_delivery_open: bad argument");
    return DABX_E_ARG;
  }
  if (e->dl.open) { set_error("dabx_delivery_open: already open"); return DABX_E_STATE; }
  int rc = sync_all(e);
  if (rc) return rc;
  Delivery &D = e->dl;
  const EngineDev &d = e->dev;
  D.what = cfg && cfg->what ? cfg->what : (DABX_DELIVER_FIB | DABX_DELIVER_MSC | DABX_DELIVER_SF);
  if ((D.what & DABX_DELIVER_FIB) && d.out_frames < DL_FRAMES) {
    set_error("dabx_delivery_open: the engine's FIB ring holds %d frames, a chunk up to %d: create it with dabx_config.out_frames >= %d "
              "(the FIBs of a chunk's first frames would have left the ring before they are gathered)", d.out_frames, DL_FRAMES, DL_FRAMES);
    return DABX_E_STATE;
  }
  D.copy_engine = cfg ? cfg->copy_engine : 0;
  D.device = e->device;
  if (D.copy_engine == 0 && (rc = sdma_open(e->device, &D.sdma))) return rc;
  const int n_slots = cfg && cfg->host_slabs ? cfg->host_slabs : 4;
  const size_t S = (size_t)d.n_streams, M = (size_t)d.max_subch, F = DL_FRAMES;
  // capacity: the tables + per stream what a full CIF can carry at the highest code rate of the standard (EEP 4-B, 4/5: 5530 B
  // of logical frames per CIF) for 4 F CIFs, and the same again for the super frames of up to DL_SF_CAP x 5 CIFs
  const size_t per_cif = 5632;
  size_t cap = sizeof(dabx_chunk_header) + S * sizeof(dabx_chunk_stream) + S * M * sizeof(dabx_chunk_subch) + S * F * (384 + 12 + sizeof(dabx_chunk_frame)) + 6 * 16 + 256;
  if (M && !d.fic_only) cap += S * ((size_t)4 * F * per_cif + (size_t)DL_SF_CAP * 5 * per_cif + 2 * 16 * M + M * DL_SF_CAP * sizeof(dabx_superframe_info));
  D.capacity = align_up(cap, 4096);
#define H(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { set_error("HIP error %d (%s) at %s:%d", (int)err__, hipGetErrorString(err__), __FILE__, __LINE__); delivery_free(e); return DABX_E_HIP; } } while (0)
  if (D.copy_engine == 1) H(hipStreamCreateWithFlags(&D.cs, hipStreamNonBlocking));
  for (int k = 0; k < Delivery::NDEV; k++) {
    H(hipMalloc((void **)&D.dev[k], D.capacity));
    H(hipMemset(D.dev[k], 0, D.capacity));
    // system-scope release, explicitly: the SDMA engine (raw HSA, outside HIP's own fences) and the host read what the gather kernels wrote
    H(hipEventCreateWithFlags(&D.packed[k], hipEventDisableTiming | hipEventReleaseToSystem));
    H(hipEventCreateWithFlags(&D.packed_lf[k], hipEventDisableTiming | hipEventReleaseToSystem));
  }
  D.slots.resize((size_t)n_slots);
  for (auto &sl : D.slots) {
    H(hipHostMalloc((void **)&sl.host, D.capacity, hipHostMallocDefault));
    if (D.copy_engine == 0 && ((rc = sdma_signal_create(&sl.sig)) || (rc = sdma_signal_create(&sl.sig2)))) { delivery_free(e); return rc; }
  }
  H(hipMalloc((void **)&D.layout_off, sizeof(unsigned long long) * std::max<size_t>(3 * S * M, 3)));
  H(hipMalloc((void **)&D.subch_id, sizeof(int32_t) * std::max<size_t>(S * M, 1)));
  H(hipMalloc((void **)&D.frames_done, sizeof(long long) * S));
  H(hipMalloc((void **)&D.cif_done, sizeof(long long) * std::max<size_t>(S * M, 1)));
  H(hipMalloc((void **)&D.sf_done, sizeof(long long) * std::max<size_t>(S * M, 1)));
  // delivery starts with what is decoded from now on
  {
    std::vector<StreamCtl> ctl(S);
    H(hipMemcpy(ctl.data(), d.ctl, sizeof(StreamCtl) * S, hipMemcpyDeviceToHost));
    std::vector<long long> fr(S), cd(std::max<size_t>(S * M, 1), 0), sd(std::max<size_t>(S * M, 1), 0);
    for (size_t s_ = 0; s_ < S; s_++) fr[s_] = ctl[s_].frames;
    if (S * M) {
      H(hipMemcpy(e->subch_host.data(), d.subch, sizeof(SubchDev) * S * M, hipMemcpyDeviceToHost));
      for (size_t sj = 0; sj < S * M; sj++) { cd[sj] = e->subch_host[sj].cif_out; sd[sj] = e->subch_host[sj].sf_count; }
    }
    H(hipMemcpy(D.frames_done, fr.data(), sizeof(long long) * S, hipMemcpyHostToDevice));
    H(hipMemcpy(D.cif_done, cd.data(), sizeof(long long) * cd.size(), hipMemcpyHostToDevice));
    H(hipMemcpy(D.sf_done, sd.data(), sizeof(long long) * sd.size(), hipMemcpyHostToDevice));
  }
#undef H
  if ((rc = e->delivery_layout())) { delivery_free(e); return rc; }
  // the engine the slabs will travel on must be one of the fast ones (sdma.h): checked with a 16-MiB transfer, replaced if it is not
  if (D.copy_engine == 0 && D.capacity >= ((size_t)16 << 20) && (rc = sdma_calibrate(D.sdma, D.slots[0].host, D.dev[0], true, D.slots[0].sig, &D.calib_gbps))) {
    delivery_free(e);
    return rc;
  }
  D.quit = false;
  D.copier = std::thread(delivery_copier, &D);
  D.open = true;
  return 0;
}

int dabx_delivery_close(dabx_engine *e)
{
  if (!e) return DABX_E_ARG;
  if (!e->dl.open) return 0;
  const int rc = sync_all(e);
  delivery_free(e);
  return rc;
}

long long dabx_delivery_slab_bytes(dabx_engine *e)
{
  if (!e) return DABX_E_ARG;
  if (!e->dl.open) { set_error("dabx_delivery_slab_bytes: no delivery open"); return DABX_E_STATE; }
  return (long long)e->dl.bytes;
}

// Consumer side (may run on a second thread): chunks in the order they were closed.
int dabx_delivery_next(dabx_engine *e, int wait, dabx_chunk *out)
{
  if (!e || !out) return DABX_E_ARG;
  Delivery &D = e->dl;
  if (!D.open) { set_error("dabx_delivery_next: no delivery open"); return DABX_E_STATE; }
  std::unique_lock<std::mutex> lk(D.mu);
  if (D.queue.empty()) return 0;
  Delivery::Slot &sl = D.slots[(size_t)D.queue.front()];      // only this thread pops: the front stays the front
  if (sl.state != Delivery::LANDED) {
    if (!wait) return 0;
    D.cv.wait(lk, [&]() { return sl.state == Delivery::LANDED; });
  }
  if (!D.copier_error.empty()) { set_error("delivery: %s", D.copier_error.c_str()); return DABX_E_HIP; }
  D.queue.pop_front();
  sl.state = Delivery::HELD;
  out->seq = sl.seq; out->data = sl.host; out->bytes = sl.bytes;
  return 1;
}

int dabx_delivery_release(dabx_engine *e, uint64_t seq)
{
  if (!e) return DABX_E_ARG;
  Delivery &D = e->dl;
  if (!D.open) { set_error("dabx_delivery_release: no delivery open"); return DABX_E_STATE; }
  std::lock_guard<std::mutex> lk(D.mu);
  for (auto &sl : D.slots)
    if (sl.state == Delivery::HELD && sl.seq == seq) { sl.state = Delivery::FREE; D.cv.notify_all(); return 0; }
  set_error("dabx_delivery_release: chunk %llu is not held", (unsigned long long)seq);
  return DABX_E_ARG;
}

int dabx_delivery_get_info(dabx_engine *e, dabx_delivery_info *out)
{
  if (!e || !out) return DABX_E_ARG;
  Delivery &D = e->dl;
  if (!D.open) { set_error("dabx_delivery_get_info: no delivery open"); return DABX_E_STATE; }
  std::lock_guard<std::mutex> lk(D.mu);
  memset(out, 0, sizeof(*out));
  out->chunks_closed = D.next_seq; out->chunks_landed = D.landed; out->bytes_copied = D.bytes_copied;
  out->copy_seconds = D.copy_s; out->copy_seconds_max = D.copy_s_max; out->gather_wait_seconds = D.gather_wait_s;
  out->copy_engine = D.copy_engine; out->sdma_engine_mask = D.copy_engine == 0 ? D.sdma.engine_to_host : 0;
  out->calibration_GBps = D.calib_gbps;
  return 0;
}

int dabx_delivery_wait_free(dabx_engine *e, int n, int timeout_ms)
{
  if (!e || n < 0) return DABX_E_ARG;
  Delivery &D = e->dl;
  if (!D.open) { set_error("dabx_delivery_wait_free: no delivery open"); return DABX_E_STATE; }
  if ((size_t)n > D.slots.size()) { set_error("dabx_delivery_wait_free: %d slabs asked for, the delivery has %zu", n, D.slots.size()); return DABX_E_ARG; }
  std::unique_lock<std::mutex> lk(D.mu);
  auto free_now = [&D]() { int k = 0; for (const auto &sl : D.slots) k += sl.state == Delivery::FREE; return k; };
  if (timeout_ms < 0) D.cv.wait(lk, [&]() { return free_now() >= n; });
  else D.cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&]() { return free_now() >= n; });
  return free_now();
}

int dabx_set_lcd_statistics(dabx_engine *e, int on)
{
  if (!e) return DABX_E_ARG;
  if (int rc = sync_all(e)) return rc;          // no frame may see the switch between its two demapper launches
  e->dev.demap.track_mer = on != 0;
  return 0;
}

int dabx_set_profiling(dabx_engine *e, int on)
{
  if (!e) return DABX_E_ARG;
  if (int rc = sync_all(e)) return rc;
  e->mk.on = on != 0;
  e->mk.serial = on < 0;
  e->mk.only = on >= 2 ? on - 2 : -1;
  e->mk.used = 0;
  e->mk.recs.clear();
  for (int k = 0; k < N_STEP_KERNELS; k++) { e->prof_ms[k] = 0; e->prof_n[k] = 0; }
  return 0;
}

int dabx_get_profile(dabx_engine *e, double total_ms[DABX_MAX_KERNELS], int64_t launches[DABX_MAX_KERNELS],
                     const char *names[DABX_MAX_KERNELS])
{
  if (!e || !total_ms || !launches || !names) return DABX_E_ARG;
  if (int rc = sync_all(e)) return rc;
  for (const auto &r : e->mk.recs) {
    float ms = 0.f;
    DABX_HIP(hipEventElapsedTime(&ms, e->mk.pool[r.a], e->mk.pool[r.b]));
    e->prof_ms[r.k] += ms; e->prof_n[r.k]++;
  }
  e->mk.recs.clear();
  e->mk.used = 0;
  for (int k = 0; k < N_STEP_KERNELS; k++) { total_ms[k] = e->prof_ms[k]; launches[k] = e->prof_n[k]; names[k] = kStepKernelNames[k]; }
  return N_STEP_KERNELS;
}

// ---- stage-level FIC decode through the pipeline kernel (FicDecoder::process_block x 3) ------------
int dabx_fic_decode(const int16_t *soft, int batch, uint8_t *fibs, uint8_t *crc_ok)
{
  if (!soft || !fibs || !crc_ok || batch <= 0) { set_error("dabx_fic_decode: bad argument"); return DABX_E_ARG; }
  int rc = need_device_e();
  if (rc) return rc;
  dabx_config cfg;
  dabx_default_config(&cfg);
  cfg.n_streams = batch; cfg.ring_frames = 2; cfg.max_subch = 0; cfg.out_frames = 1; cfg.fic_only = 1;
  // a minimal engine gives us the buffers; the IQ ring is not touched
  dabx_engine *e = nullptr;
  cfg.ring_frames = 2;
  {
    // avoid the (large) IQ ring for big batches: temporarily shrink via a dedicated light-weight allocation
    e = new dabx_engine();
    e->cfg = cfg;
    (void)hipGetDevice(&e->device);
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return DABX_E_HIP; }
    e->ss.a = e->stream;
    EngineDev &d = e->dev;
    d.n_streams = batch; d.max_subch = 0; d.out_frames = 1; d.fic_only = 1;
    d.vit_stride = (int)vit_scratch_words(FIC_OUT);
#define A(x) if ((rc = (x))) { dabx_destroy(e); return rc; }
    A(e->alloc(&d.ctl, batch));
    A(e->alloc(&d.fic_sym, (size_t)batch * 3 * K2));
    A(e->alloc(&d.fib_out, (size_t)batch * 384));
    A(e->alloc(&d.fib_crc, (size_t)batch * 12));
    A(e->alloc(&d.vit_scratch, (size_t)batch * 4 * d.vit_stride, false));
    std::vector<StreamCtl> ctl(batch);
    for (auto &c : ctl) { memset(&c, 0, sizeof(c)); c.frame_ok = 1; }
    DABX_HIP(hipMemcpyAsync(d.ctl, ctl.data(), sizeof(StreamCtl) * batch, hipMemcpyHostToDevice, e->stream));
    int16_t *dsoft = nullptr;
    A(e->alloc(&dsoft, (size_t)batch * 3 * K2, false));
    DABX_HIP(hipMemcpyAsync(dsoft, soft, sizeof(int16_t) * (size_t)batch * 3 * K2, hipMemcpyHostToDevice, e->stream));
    A(launch_i16_to_sym(dsoft, d.fic_sym, (size_t)batch * 3 * K2, e->stream));
    A(launch_fic_only(d, e->stream, 0, 4));
#undef A
    DABX_HIP(hipStreamSynchronize(e->stream));
    DABX_HIP(hipMemcpy(fibs, d.fib_out, (size_t)batch * 384, hipMemcpyDeviceToHost));
    DABX_HIP(hipMemcpy(crc_ok, d.fib_crc, (size_t)batch * 12, hipMemcpyDeviceToHost));
  }
  dabx_destroy(e);
  return 0;
}

}  // extern "C"

// ====================================================================================================================
// Per-symbol, stateful stage entries: the GPU side of the reference's FicDecoder and MscHandler CLASS surface
// (fic_decoder.h:42-58, msc_handler.h:36-47).  Both reuse the engine's kernels on the state of a one-stream engine: what
// the frame-batched path does for 512 ensembles at once these do for one ensemble, one OFDM symbol per call.
// ====================================================================================================================
struct dabx_fic {
  dabx_engine *eng = nullptr;        // light-weight: control record, FIC symbols, FIB outputs, Viterbi scratch only
  int16_t *soft_dev = nullptr;       // staging of one symbol's soft bits
  bool running = true;               // mIsRunning (the shim calls restart() from DabProcessor::start like the reference)
  int index = 0, fic_idx = 0;        // mIndex / mFicIdx: soft bits collected of the current FIC block, next block
};

extern "C" {

int dabx_fic_create(dabx_fic **out)
{
  if (!out) { set_error("dabx_fic_create: bad argument"); return DABX_E_ARG; }
  int rc = need_device_e();
  if (rc) return rc;
  auto *f = new dabx_fic();
  auto *e = f->eng = new dabx_engine();
  (void)hipGetDevice(&e->device);
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; delete f; return DABX_E_HIP; }
  e->ss.a = e->stream;
  EngineDev &d = e->dev;
  d.n_streams = 1; d.max_subch = 0; d.out_frames = 1; d.fic_only = 1;
  d.vit_stride = (int)vit_scratch_words(FIC_OUT);
#define A(x) if ((rc = (x))) { dabx_fic_destroy(f); return rc; }
  A(e->alloc(&d.ctl, 1));
  A(e->alloc(&d.fic_sym, (size_t)3 * K2));
  A(e->alloc(&d.fib_out, 384));
  A(e->alloc(&d.fib_crc, 12));
  A(e->alloc(&d.vit_scratch, (size_t)4 * d.vit_stride, false));
  A(e->alloc(&f->soft_dev, (size_t)K2, false));
#undef A
  StreamCtl c;
  memset(&c, 0, sizeof(c));
  c.frame_ok = 1;
  DABX_HIP(hipMemcpyAsync(d.ctl, &c, sizeof(c), hipMemcpyHostToDevice, e->stream));
  DABX_HIP(hipStreamSynchronize(e->stream));
  *out = f;
  return 0;
}

void dabx_fic_destroy(dabx_fic *f)
{
  if (!f) return;
  dabx_destroy(f->eng);
  delete f;
}

int dabx_fic_process_block(dabx_fic *f, const int16_t *soft, int sym_idx, int *first_fic)
{
  if (!f || !soft || sym_idx < 1 || sym_idx > 3) { set_error("dabx_fic_process_block: bad argument"); return DABX_E_ARG; }
  dabx_engine *e = f->eng;
  if (int rc = use_device(e)) return rc;
  if (sym_idx == 1) { f->index = 0; f->fic_idx = 0; }            // fic_decoder.cpp:148-152
  // the 3072 soft bits continue the running FIC block; blocks complete at 2304-bit boundaries (:154-165)
  const int pos0 = f->fic_idx * FIC_IN + f->index;               // position in the frame's 9216 FIC soft bits
  if (pos0 + K2 > 3 * K2) { set_error("dabx_fic_process_block: symbols out of order"); return DABX_E_STATE; }
  const int done_before = f->fic_idx;
  const int total = f->index + K2;
  const int completed = total / FIC_IN;
  f->index = total % FIC_IN;
  f->fic_idx += completed;
  if (first_fic) *first_fic = done_before;
  if (!f->running) return 0;                                     // :182-185: _process_fic_input returns at once
  DABX_HIP(hipMemcpyAsync(f->soft_dev, soft, sizeof(int16_t) * K2, hipMemcpyHostToDevice, e->stream));
  int rc = launch_i16_to_sym(f->soft_dev, e->dev.fic_sym + pos0, (size_t)K2, e->stream);
  if (rc) return rc;
  if (completed > 0 && (rc = launch_fic_only(e->dev, e->stream, done_before, completed))) return rc;
  DABX_HIP(hipStreamSynchronize(e->stream));
  return completed;
}

int dabx_fic_get_fibs(dabx_fic *f, int fic_idx, uint8_t fibs[96], uint8_t crc_ok[3])
{
  if (!f || fic_idx < 0 || fic_idx > 3 || !fibs || !crc_ok) return DABX_E_ARG;
  if (int rc = sync_all(f->eng)) return rc;
  DABX_HIP(hipMemcpy(fibs, f->eng->dev.fib_out + 96 * fic_idx, 96, hipMemcpyDeviceToHost));
  DABX_HIP(hipMemcpy(crc_ok, f->eng->dev.fib_crc + 3 * fic_idx, 3, hipMemcpyDeviceToHost));
  return 0;
}

int dabx_fic_get_fib_bits(dabx_fic *f, uint8_t *bits, uint8_t *valid)
{
  if (!f || !bits || !valid) return DABX_E_ARG;
  if (int rc = sync_all(f->eng)) return rc;
  uint8_t packed[384], crc[12];
  DABX_HIP(hipMemcpy(packed, f->eng->dev.fib_out, 384, hipMemcpyDeviceToHost));
  DABX_HIP(hipMemcpy(crc, f->eng->dev.fib_crc, 12, hipMemcpyDeviceToHost));
  for (int i = 0; i < 3072; i++) bits[i] = (uint8_t)((packed[i >> 3] >> (7 - (i & 7))) & 1);
  for (int g = 0; g < 4; g++) valid[g] = (uint8_t)(crc[3 * g] && crc[3 * g + 1] && crc[3 * g + 2]);
  return 0;
}

static int fic_ctl(dabx_fic *f, StreamCtl *c)
{
  if (int rc = sync_all(f->eng)) return rc;
  DABX_HIP(hipMemcpy(c, f->eng->dev.ctl, sizeof(StreamCtl), hipMemcpyDeviceToHost));
  return 0;
}
int dabx_fic_get_decode_ratio_percent(dabx_fic *f)
{
  if (!f) return DABX_E_ARG;
  StreamCtl c;
  if (int rc = fic_ctl(f, &c)) return rc;
  return c.fic_ratio * 10;
}
int dabx_fic_get_cif_count(dabx_fic *f)
{
  if (!f) return DABX_E_ARG;
  StreamCtl c;
  if (int rc = fic_ctl(f, &c)) return rc;
  return c.cif_count;
}
int dabx_fic_get_ber(dabx_fic *f, dabx_fic_ber *out)
{
  if (!f || !out) return DABX_E_ARG;
  StreamCtl c;
  if (int rc = fic_ctl(f, &c)) return rc;
  memset(out, 0, sizeof(*out));
  out->bits = c.fic_bits; out->errors = c.fic_errors; out->status_bits = c.fic_status_bits; out->status_errors = c.fic_status_errors;
  out->blocks = c.fic_block;
  return 0;
}
int dabx_fic_reset_decode_success_ratio(dabx_fic *f)
{
  if (!f) return DABX_E_ARG;
  StreamCtl c;
  if (int rc = fic_ctl(f, &c)) return rc;
  c.fic_ratio = 0;
  DABX_HIP(hipMemcpy(f->eng->dev.ctl, &c, sizeof(StreamCtl), hipMemcpyHostToDevice));
  return 0;
}
int dabx_fic_stop(dabx_fic *f) { if (!f) return DABX_E_ARG; f->running = false; return 0; }
int dabx_fic_restart(dabx_fic *f)
{
  if (!f) return DABX_E_ARG;
  if (int rc = dabx_fic_reset_decode_success_ratio(f)) return rc;
  f->running = true;
  return 0;
}

}  // extern "C"

struct dabx_msc {
  dabx_engine *eng = nullptr;              // one-stream engine: TDI ring, sub-channel slots, output rings, DAB+ stage
  int16_t *soft_dev = nullptr;
  std::vector<dabx_subch_desc> slots;      // kbps == 0: free
  std::vector<long long> frames_seen, sf_seen;   // per slot: logical / super frames that existed before the CIF just closed
  std::vector<long long> frames_now, sf_now;
};

static int msc_apply(dabx_msc *m)
{
  return dabx_set_subchannels(m->eng, 0, m->slots.data(), (int)m->slots.size());
}

extern "C" {

int dabx_msc_create(int max_services, dabx_msc **out)
{
  if (!out || max_services < 1 || max_services > MAX_SUBCH) { set_error("dabx_msc_create: bad argument"); return DABX_E_ARG; }
  dabx_config cfg;
  dabx_default_config(&cfg);
  cfg.n_streams = 1; cfg.ring_frames = 2; cfg.max_subch = max_services; cfg.out_frames = 1;
  auto *m = new dabx_msc();
  int rc = dabx_create(&cfg, &m->eng);
  if (rc) { delete m; return rc; }
  if ((rc = m->eng->alloc(&m->soft_dev, (size_t)K2, false))) { dabx_msc_destroy(m); return rc; }
  m->slots.assign((size_t)max_services, dabx_subch_desc{});
  m->frames_seen.assign((size_t)max_services, 0); m->sf_seen.assign((size_t)max_services, 0);
  m->frames_now.assign((size_t)max_services, 0); m->sf_now.assign((size_t)max_services, 0);
  *out = m;
  return 0;
}

void dabx_msc_destroy(dabx_msc *m)
{
  if (!m) return;
  dabx_destroy(m->eng);
  delete m;
}

int dabx_msc_set_channel(dabx_msc *m, const dabx_subch_desc *d)
{
  if (!m || !d || d->kbps <= 0) { set_error("dabx_msc_set_channel: bad argument"); return DABX_E_ARG; }
  int slot = -1;
  for (size_t j = 0; j < m->slots.size() && slot < 0; j++) if (!m->slots[j].kbps) slot = (int)j;
  if (slot < 0) { set_error("dabx_msc_set_channel: all %zu service slots in use", m->slots.size()); return DABX_E_STATE; }
  m->slots[(size_t)slot] = *d;
  if (m->slots[(size_t)slot].dab_plus < 0) m->slots[(size_t)slot].dab_plus = (d->kbps <= 384 && d->kbps % 8 == 0) ? 1 : 0;
  const int rc = msc_apply(m);
  if (rc) { m->slots[(size_t)slot] = dabx_subch_desc{}; return rc; }
  m->frames_seen[(size_t)slot] = m->sf_seen[(size_t)slot] = m->frames_now[(size_t)slot] = m->sf_now[(size_t)slot] = 0;
  return slot;
}

int dabx_msc_stop_service(dabx_msc *m, int slot)
{
  if (!m || slot < 0 || slot >= (int)m->slots.size()) return DABX_E_ARG;
  m->slots[(size_t)slot] = dabx_subch_desc{};
  return msc_apply(m);
}

int dabx_msc_stop_all_services(dabx_msc *m)
{
  if (!m) return DABX_E_ARG;
  for (auto &s : m->slots) s = dabx_subch_desc{};
  return msc_apply(m);
}

int dabx_msc_is_service_running(dabx_msc *m, int slot)
{
  if (!m || slot < 0 || slot >= (int)m->slots.size()) return DABX_E_ARG;
  return m->slots[(size_t)slot].kbps != 0;
}

int dabx_msc_process_block(dabx_msc *m, const int16_t *soft, int blk_nr)
{
  if (!m || !soft || blk_nr < 4 || blk_nr >= L) { set_error("dabx_msc_process_block: bad argument"); return DABX_E_ARG; }
  dabx_engine *e = m->eng;
  if (int rc = use_device(e)) return rc;
  if (e->classes_dirty) {                       // one stream never reaches the lane-per-trellis path, but the slots' class tags must be current
    if (int rc = e->build_msc_classes()) return rc;
    e->classes_dirty = false;
  }
  const int cur = (blk_nr - 4) % 18;            // msc_handler.cpp:145
  const bool closes = cur == 17;
  DABX_HIP(hipMemcpyAsync(m->soft_dev, soft, sizeof(int16_t) * K2, hipMemcpyHostToDevice, e->stream));
  int rc = launch_stage_msc_block(e->dev, m->soft_dev, cur, closes, e->stream);
  if (rc) return rc;
  if (!closes) { DABX_HIP(hipStreamSynchronize(e->stream)); return 0; }
  // a full CIF: every back end runs (msc_handler.cpp:155-167)
  e->dev.snap = e->snap_buf[e->ss.batch_parity];
  if ((rc = launch_msc_batch(e->dev, 1, nullptr, e->ss, e->mk))) return rc;
  if ((rc = sync_all(e))) return rc;
  std::vector<SubchDev> sc(m->slots.size());
  DABX_HIP(hipMemcpy(sc.data(), e->dev.subch, sizeof(SubchDev) * sc.size(), hipMemcpyDeviceToHost));
  for (size_t j = 0; j < sc.size(); j++) {
    m->frames_seen[j] = m->frames_now[j]; m->sf_seen[j] = m->sf_now[j];
    m->frames_now[j] = sc[j].active ? sc[j].cif_out : 0;
    m->sf_now[j] = sc[j].active ? sc[j].sf_count : 0;
  }
  return 1;
}

int dabx_msc_get_frame(dabx_msc *m, int slot, uint8_t *bytes, int max_bytes)
{
  if (!m || slot < 0 || slot >= (int)m->slots.size() || !bytes) return DABX_E_ARG;
  const size_t j = (size_t)slot;
  if (!m->slots[j].kbps || m->frames_now[j] == m->frames_seen[j]) return 0;       // de-interleaver still filling / no new CIF
  const int nb = 3 * m->slots[j].kbps;
  if (max_bytes < nb) { set_error("dabx_msc_get_frame: %d bytes needed", nb); return DABX_E_ARG; }
  const int got = dabx_read_msc(m->eng, 0, slot, 1, bytes);
  return got < 0 ? got : (got == 1 ? nb : 0);
}

int dabx_msc_get_superframe(dabx_msc *m, int slot, uint8_t *bytes, int max_bytes)
{
  if (!m || slot < 0 || slot >= (int)m->slots.size() || !bytes) return DABX_E_ARG;
  const size_t j = (size_t)slot;
  if (!m->slots[j].kbps || m->sf_now[j] == m->sf_seen[j]) return 0;
  const int nb = 110 * m->slots[j].kbps / 8;
  if (max_bytes < nb) { set_error("dabx_msc_get_superframe: %d bytes needed", nb); return DABX_E_ARG; }
  const int got = dabx_read_superframes(m->eng, 0, slot, 1, bytes);
  return got < 0 ? got : (got == 1 ? nb : 0);
}

int dabx_msc_get_superframe_info(dabx_msc *m, int slot, dabx_superframe_info *out)
{
  if (!m || slot < 0 || slot >= (int)m->slots.size() || !out) return DABX_E_ARG;
  const size_t j = (size_t)slot;
  if (!m->slots[j].kbps || m->sf_now[j] == m->sf_seen[j]) return 0;
  const int got = dabx_read_superframe_info(m->eng, 0, slot, 1, out);
  return got < 0 ? got : (got == 1 ? 1 : 0);
}

int dabx_msc_get_stats(dabx_msc *m, int slot, dabx_subch_stats *out)
{
  if (!m) return DABX_E_ARG;
  return dabx_get_subch_stats(m->eng, 0, slot, out);
}

}  // extern "C"
