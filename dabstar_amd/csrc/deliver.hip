// deliver.hip -- bulk delivery of a chunk's results (include/dabx.h, "Bulk delivery"): two gather kernels pack everything the
// frames of one MSC batch produced -- FIBs + CRC flags + frame records (front-end stream), logical frames + super frames + the
// slots' counters (behind the DAB+ stage, on the stream that ran it) -- into ONE device slab; engine.cpp then moves the slab to a
// page-locked host slab with ONE transfer on an SDMA engine (sdma.h; the copier thread).  The host side of what the reference does per item:
// IFibDecoder::process_FIB from fic_decoder.cpp:234-261, FrameProcessor::add_to_frame from backend.cpp:160, the super frame of
// mp4processor.cpp:149-158.  Pure HBM copies: 14 208 B of results per frame + 13 % (super frames are the logical
// frames' bytes again, RS-corrected) -- 0.7 % of the chain's algorithmic bytes.
#include "pipeline.h"

namespace dabx {

// grid = n_streams, 128 threads, front-end stream, behind the chunk's last frame tail
__global__ __launch_bounds__(128) void k_deliver_front(EngineDev e, DeliverDev dv)
{
  const int s = blockIdx.x, tid = threadIdx.x;
  const StreamCtl &c = e.ctl[s];
  uint8_t *slab = dv.slab;
  if (s == 0 && tid == 0) *reinterpret_cast<dabx_chunk_header *>(slab) = dv.hdr;
  const int F = dv.hdr.max_frames;
  const long long done = dv.frames_done[s], have = c.frames - done;
  int n = (int)(have < F ? have : F);
  if (n > e.out_frames) n = e.out_frames;
  if (!(dv.hdr.what & DABX_DELIVER_FIB)) n = (int)(have < F ? have : F);     // no FIB bytes wanted: the count alone
  const long long first = c.frames - n;
  if (tid == 0) {
    dabx_chunk_stream r;
    r.first_frame = first; r.n_frames = n; r.frames_lost = (int)(have - n);
    r.state = c.state == ST_EVAL_SYNC ? 2 : (c.state == ST_WAIT_SYNC ? 1 : 0);
    r.fic_ratio_percent = c.fic_ratio * 10; r.cif_count = c.cif_count;
    r.snr_db_est = c.snr_db; r.freq_offs_bb_hz = c.f_bb; r.clock_err_hz = c.clock_err; r.signal_level = c.s_level;
    r.fic_ber_bits = c.fic_bits; r.fic_ber_errors = c.fic_errors; r.mer_db_est = c.mer_db;
    r.fib_ok = c.fib_ok; r.fib_total = c.fib_total;
    reinterpret_cast<dabx_chunk_stream *>(slab + dv.hdr.off_stream)[s] = r;
  }
  if (dv.hdr.what & DABX_DELIVER_FIB) {
    uint32_t *fo = reinterpret_cast<uint32_t *>(slab + dv.hdr.off_fib) + (size_t)s * F * 96;
    uint32_t *co = reinterpret_cast<uint32_t *>(slab + dv.hdr.off_crc) + (size_t)s * F * 3;
    dabx_chunk_frame *ro = reinterpret_cast<dabx_chunk_frame *>(slab + dv.hdr.off_frame) + (size_t)s * F;
    for (int i = tid; i < n * 96; i += 128) {
      const int f = i / 96, w = i - 96 * f;
      const size_t slot = (size_t)s * e.out_frames + (size_t)((first + f) % e.out_frames);
      fo[i] = reinterpret_cast<const uint32_t *>(e.fib_out + slot * 384)[w];
    }
    for (int i = tid; i < n * 3; i += 128) {
      const int f = i / 3, w = i - 3 * f;
      const size_t slot = (size_t)s * e.out_frames + (size_t)((first + f) % e.out_frames);
      co[i] = reinterpret_cast<const uint32_t *>(e.fib_crc + slot * 12)[w];
    }
    if (tid < n) {
      const size_t slot = (size_t)s * e.out_frames + (size_t)((first + tid) % e.out_frames);
      dabx_chunk_frame q;
      q.sym0_pos = e.frame_pos ? e.frame_pos[slot] : -1; q.start_index = e.frame_start ? e.frame_start[slot] : -1; q.reserved = 0;
      ro[tid] = q;
    }
  }
  if (tid == 0) dv.frames_done[s] = c.frames;
}

// Which logical frames of slot sj the chunk carries.  Called BEFORE the DAB+ stage (k_deliver_lf: cif_out has not been moved on yet, the
// batch's new frames are counted from its snapshot exactly as k_dabplus counts them) and after it (k_deliver_msc: pre = false).
struct LfRange { long long first; int n, lost; };
__device__ __forceinline__ LfRange deliver_lf_range(const EngineDev &e, const DeliverDev &dv, int sj, const SubchDev &sc, bool pre)
{
  long long cif_out = sc.cif_out;
  if (pre) {
    const BatchSnap bs = e.snap[sj / e.max_subch];
    for (long long r = bs.msc_done; r < bs.cif_no; r++) if (r >= sc.start_cif + 16) cif_out++;
  }
  const int cap_cifs = 4 * dv.hdr.max_frames;
  const long long have = cif_out - dv.cif_done[sj];
  LfRange q;
  q.n = (int)(have < cap_cifs ? have : cap_cifs);
  if (q.n > MSC_SLOTS) q.n = MSC_SLOTS;
  if (q.n < 0) q.n = 0;
  q.lost = (int)(have > q.n ? have - q.n : 0);
  q.first = cif_out - q.n;
  const bool want_lf = (dv.hdr.what & DABX_DELIVER_MSC) || ((dv.hdr.what & DABX_DELIVER_MSC_NOT_DABPLUS) && !sc.dab_plus);
  if (!want_lf) { q.lost = 0; q.first = cif_out; q.n = 0; }      // not wanted: nothing is "lost"
  return q;
}

// grid = n_streams * max_subch, 64 threads, behind the chunk's Viterbi decode and IN FRONT of the DAB+ stage (same stream): the logical frames --
// half of an "everything" slab -- are gathered as soon as they exist, and their share of the slab goes onto the link while k_dabplus still runs
__global__ __launch_bounds__(64) void k_deliver_lf(EngineDev e, DeliverDev dv)
{
  const int sj = blockIdx.x, lane = threadIdx.x;
  const SubchDev &sc = e.subch[sj];
  if (!(sc.active && e.msc_out && !e.fic_only)) return;
  const LfRange q = deliver_lf_range(e, dv, sj, sc, true);
  if (q.n == 0) return;
  const uint8_t *ring = e.msc_out + (size_t)sj * MSC_SLOTS * e.msc_stride;
  uint32_t *o = reinterpret_cast<uint32_t *>(dv.slab + dv.layout_off[3 * (size_t)sj]);
  const int wpf = 3 * sc.kbps / 4;
  for (int f = 0; f < q.n; f++) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(ring + (size_t)((q.first + f) % MSC_SLOTS) * e.msc_stride);
    for (int w = lane; w < wpf; w += 64) o[(size_t)f * wpf + w] = src[w];
  }
}

// grid = n_streams * max_subch, 64 threads, behind k_dabplus of the chunk's MSC batch (same stream)
__global__ __launch_bounds__(64) void k_deliver_msc(EngineDev e, DeliverDev dv, int with_lf)
{
  const int sj = blockIdx.x, lane = threadIdx.x;
  const SubchDev &sc = e.subch[sj];
  uint8_t *slab = dv.slab;
  const unsigned long long msc_off = dv.layout_off[3 * (size_t)sj], sf_off = dv.layout_off[3 * (size_t)sj + 1], sfi_off = dv.layout_off[3 * (size_t)sj + 2];
  const int R = sc.kbps / 8, nb = 3 * sc.kbps, sfb = 110 * R, pitch = (sfb + 3) & ~3;
  const int cap_sf = (4 * dv.hdr.max_frames + 4) / 5;
  long long c_done = dv.cif_done[sj], s_done = dv.sf_done[sj];
  int n_c = 0, n_s = 0, lost_c = 0, lost_s = 0;
  long long first_c = c_done, first_s = s_done;
  const bool live = sc.active && e.msc_out && !e.fic_only;
  if (live) {
    const long long have_s = sc.sf_count - s_done;
    const LfRange q = deliver_lf_range(e, dv, sj, sc, false);          // the very range k_deliver_lf copied (it counted this batch's frames ahead)
    n_c = q.n; lost_c = q.lost; first_c = q.first;
    n_s = (int)(have_s < cap_sf ? have_s : cap_sf);
    if (n_s > SF_SLOTS) n_s = SF_SLOTS;
    if (n_s < 0) n_s = 0;
    lost_s = (int)(have_s > n_s ? have_s - n_s : 0);
    first_s = sc.sf_count - n_s;
    if (with_lf && n_c) {                                              // (no k_deliver_lf ran in front: the logical frames here as well)
      const uint8_t *ring = e.msc_out + (size_t)sj * MSC_SLOTS * e.msc_stride;
      uint32_t *o = reinterpret_cast<uint32_t *>(slab + msc_off);
      const int wpf = nb / 4;
      for (int f = 0; f < n_c; f++) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(ring + (size_t)((first_c + f) % MSC_SLOTS) * e.msc_stride);
        for (int w = lane; w < wpf; w += 64) o[(size_t)f * wpf + w] = src[w];
      }
    }
    if ((dv.hdr.what & DABX_DELIVER_SF) && sc.dab_plus) {
      const uint8_t *ring = e.sf_out + (size_t)sj * SF_SLOTS * e.sf_stride;
      uint32_t *o = reinterpret_cast<uint32_t *>(slab + sf_off);
      const int wpf = pitch / 4;
      for (int f = 0; f < n_s; f++) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(ring + (size_t)((first_s + f) % SF_SLOTS) * e.sf_stride);
        for (int w = lane; w < wpf; w += 64) o[(size_t)f * wpf + w] = src[w];
      }
      // ... and their records (32 bytes = 8 words each)
      if (e.sf_info && lane < 8 * n_s) {
        const uint32_t *inf = reinterpret_cast<const uint32_t *>(e.sf_info + (size_t)sj * SF_SLOTS);
        uint32_t *oi = reinterpret_cast<uint32_t *>(slab + sfi_off);
        for (int w = lane; w < 8 * n_s; w += 64) oi[w] = inf[(size_t)((first_s + (w >> 3)) % SF_SLOTS) * 8 + (w & 7)];
      }
    }
  }
  if (lane == 0) {
    dabx_chunk_subch r;
    r.active = live ? 1 : 0; r.subch_id = dv.subch_id[sj]; r.kbps = sc.kbps; r.dab_plus = sc.dab_plus;
    r.start_cif = sc.start_cif; r.first_cif = first_c; r.n_cifs = n_c; r.cifs_lost = lost_c;
    r.first_sf = first_s; r.n_sf = n_s; r.sf_lost = lost_s; r.msc_off = msc_off; r.sf_off = sf_off; r.sf_pitch = pitch; r.reserved = 0;
    r.sf_ok = sc.sf_ok; r.sf_fail = sc.sf_fail; r.rs_corrected = sc.rs_corr; r.rs_failed = sc.rs_fail; r.fc_corrected = sc.fc_corr;
    r.au_ok = sc.au_ok; r.au_bad = sc.au_bad; r.sfi_off = sfi_off;
    reinterpret_cast<dabx_chunk_subch *>(slab + dv.hdr.off_subch)[sj] = r;
    if (live) { dv.cif_done[sj] = sc.cif_out; dv.sf_done[sj] = sc.sf_count; }
  }
}

int launch_deliver_front(const EngineDev &e, const DeliverDev &dv, hipStream_t st)
{
  hipLaunchKernelGGL(k_deliver_front, dim3(e.n_streams), dim3(128), 0, st, e, dv);
  DABX_HIP(hipGetLastError());
  return 0;
}

// with_lf: no launch_deliver_lf went in front (the logical frames are gathered here too)
int launch_deliver_msc(const EngineDev &e, const DeliverDev &dv, hipStream_t st, bool with_lf)
{
  if (e.max_subch <= 0) return 0;
#ifdef DABX_DELIVER_NOPACK             // experiment builds only: what the copy alone costs (the slot table then says "nothing")
  return 0;
#endif
  hipLaunchKernelGGL(k_deliver_msc, dim3(e.n_streams * e.max_subch), dim3(64), 0, st, e, dv, with_lf ? 1 : 0);
  DABX_HIP(hipGetLastError());
  return 0;
}
// behind the batch's Viterbi decode, in front of its DAB+ stage; records dv.lf_done behind the gather
int launch_deliver_lf(const EngineDev &e, const DeliverDev &dv, hipStream_t st)
{
  if (e.max_subch <= 0 || !dv.lf_done) return 0;
#ifndef DABX_DELIVER_NOPACK
  hipLaunchKernelGGL(k_deliver_lf, dim3(e.n_streams * e.max_subch), dim3(64), 0, st, e, dv);
#endif
  DABX_HIP(hipEventRecord(dv.lf_done, st));
  DABX_HIP(hipGetLastError());
  return 0;
}

}  // namespace dabx
