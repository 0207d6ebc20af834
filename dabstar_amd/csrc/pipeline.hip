// pipeline.hip -- per-frame kernels of the stream-batched receiver.  One batch step advances every
// stream that is in lock (and has a frame of samples) by one DAB frame; this file is the device-side
// equivalent of DabProcessor::run (base/main/dab_processor.cpp:110-442).
//
//   k_acquire      streams out of lock (ST_INIT / ST_WAIT_SYNC): level seeding + null-dip detector (timesyncer.cpp:40-90) + correlation of
//                  every candidate; in step on the frame chain's HIP stream or next to it on its own (launch_front_step)
//   k_level_exact  cfg.exact_level_tracker: SampleReader's level IIR sample by sample in lock too (sample_reader.cpp:245-248)
//   k_frame_head   PRS correlation -> start index, symbol-0 FFT, optional coarse CFO (dab_processor.cpp:389-414, 191-224)
//   k_symbols      symbols 1..75: NCO mix, cyclic-prefix correlation, FFT (dab_processor.cpp:304-341)
//   k_demap_frame  75 x decode_symbol, soft bits -> Viterbi symbols -> FIC buffer / time-deinterleaver ring
//   k_fic_frame    4 x depuncture + Viterbi + PRBS + 12 x CRC + FIG 0/0 walk (fic_decoder.cpp:143-262)
//   k_frame_tail   fine CFO, null symbol, clock error, cursor bookkeeping (dab_processor.cpp:226-302)
//   k_msc_frame    time de-interleave + depuncture + Viterbi + PRBS per (CIF, sub-channel) (backend.cpp:129-161)
//   k_dabplus      super-frame sync, RS(120,110), fire code, AU CRCs (mp4processor.cpp:96-333)
#include <type_traits>
#include "pipeline.h"
#include <algorithm>
#include "ofdm_core.h"
#include "viterbi_core.h"
#include "fec_core.h"
#include "acq_walk.h"
#include "level_par.h"

namespace dabx {

// Issue priority of the frame-serial front end.  Its kernels run with few waves per SIMD and long dependent chains (one
// block per stream); the batched MSC decoder (k_msc_vitT, own HIP stream) keeps every SIMD's VALU busy with four waves of
// independent work.  At equal priority the arbiter shares issue slots evenly and the front end -- the critical path of a
// step -- runs at half speed whenever the decoder is resident.  s_setprio 3 lets front-end waves issue first; the decoder
// fills the slots they leave (+2.5 % same-box, docs/history/r01-r04_design_notebook.md 6).
__device__ __forceinline__ void front_prio() { __builtin_amdgcn_s_setprio(3); }

// IQ ring addressing: one 64-bit modulo per thread and kernel (for the window base), then 32-bit
// add + conditional subtract per sample.
// Block-uniform word that no thread of the running kernel writes: read through the CONSTANT address space, i.e. one
// s_load over the scalar cache instead of a vector load of the same address in every lane.
template <class T> __device__ __forceinline__ T uniform_load(const T *p)
{
  typedef unsigned Raw __attribute__((ext_vector_type(sizeof(T) / 4)));
  const Raw raw = *(const __attribute__((address_space(4))) Raw *)(const void *)p;
  return __builtin_bit_cast(T, raw);
}

struct RingView {
  const float2 *p;
  unsigned len, base;
  __device__ RingView(const float2 *ring, int ring_len, unsigned long long abs_base)
      : p(ring), len((unsigned)ring_len), base((unsigned)(abs_base % (unsigned long long)ring_len)) {}
  __device__ float2 at(unsigned i) const     // i < len
  {
    unsigned o = base + i;
    if (o >= len) o -= len;
    return p[o];
  }
};

// ------------------------------------------------------------------------------------- DC / IQ correction
// SampleReader::get_samples, scalar body (sample_reader.cpp:218-243; off by default, configuration.cpp:75-76): every
// sample read passes five one-pole filters with alpha = 1 / 2 048 000 -- meanI, meanQ; meanII, meanIQ of the DC-free
// sample; meanQQ of the phase-corrected Q -- and leaves as (x_i, x_q_corr * sqrt(meanII / meanQQ)) or, DC only, as
// (v_i - meanI, v_q - meanQ).  Here the newly committed samples of a stream are corrected IN PLACE in the IQ ring before
// any kernel reads them (everything the reference reads goes through get_samples exactly once, in order).  A filter
// y <- y + alpha (x - y) is the affine map y -> (1 - alpha) y + alpha x: per tile of 4096 samples each thread composes the
// maps of its 16 consecutive samples, a block scan gives every thread its start state, and the thread then runs the
// reference's float recurrence over its 16 samples from there -- three dependent levels (means -> second moments -> QQ).
// Not bit-identical to the 196 608-step serial recurrence (start states differ in the last ulps; each filter forgets rounding
// noise with its own 1-s time constant), equal within 1e-5 of full scale (tests/test_gpu_engine.py).
// The maps are kept as y -> y - e y + b with e = 1 - a: a itself (1 - 4.9e-7 per sample) has no exact float, its rounding
// would change the filter's time constant by up to 6 %; e and b are small numbers and compose exactly enough.
struct Affine { float a /* e = 1 - slope */, b; };
__device__ __forceinline__ Affine aff_then(Affine f, Affine g) { return {f.a + g.a - g.a * f.a, f.b - g.a * f.b + g.b}; }   // g after f
__device__ __forceinline__ float aff_apply(Affine f, float y) { return y + (f.b - f.a * y); }
__device__ __forceinline__ Affine aff_block_exclusive(Affine v, Affine *sh /* [4] */, int tid, Affine &total)
{
  const int lane = tid & 63, w = tid >> 6;
  Affine inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    Affine t = {__shfl_up(inc.a, o), __shfl_up(inc.b, o)};
    if (lane >= o) inc = aff_then(t, inc);
  }
  __syncthreads();
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  Affine pre = {0.0f, 0.0f};
  for (int q = 0; q < w; q++) pre = aff_then(pre, sh[q]);
  total = aff_then(aff_then(aff_then(sh[0], sh[1]), sh[2]), sh[3]);
  Affine ex = {__shfl_up(inc.a, 1), __shfl_up(inc.b, 1)};
  if (lane == 0) ex = {0.0f, 0.0f};
  return aff_then(pre, ex);                                  // everything before this thread within the tile
}

__global__ __launch_bounds__(256) void k_dciq(EngineDev e, int mode /* 1 DC, 2 DC + IQ */)
{
  __shared__ Affine sh[4];
  const int s = blockIdx.x, tid = threadIdx.x;
  const unsigned long long from = e.dciq_done[s], to = e.wr[s];
  if (from >= to) return;
  float2 *ring = e.iq + (size_t)s * e.ring_len;
  float *st = e.dciq_state + (size_t)s * 8;
  float meanI = st[0], meanQ = st[1], meanII = st[2], meanQQ = st[3], meanIQ = st[4];     // tile start states (all threads)
  constexpr float ALPHA = 1.0f / (float)INPUT_RATE;
  constexpr int PER = 16, TILE = 256 * PER;
  for (unsigned long long t0 = from; t0 < to; t0 += TILE) {
    const unsigned long long mine = t0 + (unsigned long long)tid * PER;
    const int n = mine >= to ? 0 : (int)(to - mine < PER ? to - mine : PER);
    float vi[PER], vq[PER];
    unsigned o = (unsigned)(mine % (unsigned long long)e.ring_len);
#pragma unroll
    for (int k = 0; k < PER; k++) {
      unsigned idx = o + k; if (idx >= (unsigned)e.ring_len) idx -= e.ring_len;
      const float2 v = k < n ? ring[idx] : make_float2(0.f, 0.f);
      vi[k] = v.x; vq[k] = v.y;
    }
    // ---- level 1: meanI, meanQ
    Affine fI = {0.f, 0.f}, fQ = {0.f, 0.f}, tot;
#pragma unroll
    for (int k = 0; k < PER; k++) if (k < n) { fI = aff_then(fI, {ALPHA, ALPHA * vi[k]}); fQ = aff_then(fQ, {ALPHA, ALPHA * vq[k]}); }
    Affine exI = aff_block_exclusive(fI, sh, tid, tot);
    const float mI_end = aff_apply(tot, meanI);
    float mI = aff_apply(exI, meanI);
    Affine exQ = aff_block_exclusive(fQ, sh, tid, tot);
    const float mQ_end = aff_apply(tot, meanQ);
    float mQ = aff_apply(exQ, meanQ);
#pragma unroll
    for (int k = 0; k < PER; k++) if (k < n) {                // mean_filter + subtraction, sample_reader.cpp:222-225,246
      mI += ALPHA * (vi[k] - mI); mQ += ALPHA * (vq[k] - mQ);
      vi[k] -= mI; vq[k] -= mQ;
    }
    meanI = mI_end; meanQ = mQ_end;
    if (mode == 2) {
      // ---- level 2: meanII, meanIQ -> phi -> x_q_corr
      Affine fII = {0.f, 0.f}, fIQ = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < PER; k++) if (k < n) { fII = aff_then(fII, {ALPHA, ALPHA * (vi[k] * vi[k])}); fIQ = aff_then(fIQ, {ALPHA, ALPHA * (vi[k] * vq[k])}); }
      Affine ex2 = aff_block_exclusive(fII, sh, tid, tot);
      const float mII_end = aff_apply(tot, meanII);
      float mII = aff_apply(ex2, meanII);
      Affine ex3 = aff_block_exclusive(fIQ, sh, tid, tot);
      const float mIQ_end = aff_apply(tot, meanIQ);
      float mIQ = aff_apply(ex3, meanIQ);
      float mIIk[PER];
#pragma unroll
      for (int k = 0; k < PER; k++) if (k < n) {              // :229-233
        mII += ALPHA * (vi[k] * vi[k] - mII); mIQ += ALPHA * (vi[k] * vq[k] - mIQ);
        const float phi = mIQ / mII;
        vq[k] = vq[k] - phi * vi[k];
        mIIk[k] = mII;
      }
      meanII = mII_end; meanIQ = mIQ_end;
      // ---- level 3: meanQQ -> gainQ
      Affine fQQ = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < PER; k++) if (k < n) fQQ = aff_then(fQQ, {ALPHA, ALPHA * (vq[k] * vq[k])});
      Affine ex4 = aff_block_exclusive(fQQ, sh, tid, tot);
      const float mQQ_end = aff_apply(tot, meanQQ);
      float mQQ = aff_apply(ex4, meanQQ);
#pragma unroll
      for (int k = 0; k < PER; k++) if (k < n) {              // :234-236
        mQQ += ALPHA * (vq[k] * vq[k] - mQQ);
        vq[k] *= sqrtf(mIIk[k] / mQQ);
      }
      meanQQ = mQQ_end;
    }
#pragma unroll
    for (int k = 0; k < PER; k++) if (k < n) {
      unsigned idx = o + k; if (idx >= (unsigned)e.ring_len) idx -= e.ring_len;
      ring[idx] = make_float2(vi[k], vq[k]);
    }
  }
  if (tid == 0) { st[0] = meanI; st[1] = meanQ; st[2] = meanII; st[3] = meanQQ; st[4] = meanIQ; e.dciq_done[s] = to; }
}
int launch_dciq(const EngineDev &e, int mode, hipStream_t st)
{
  hipLaunchKernelGGL(k_dciq, dim3(e.n_streams), dim3(256), 0, st, e, mode);
  DABX_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ acquire
// WAIT_FOR_TIME_SYNC_MARKER of one stream (dab_processor.cpp:146-160 + timesyncer.cpp:40-90): decoder reset, then the null-dip
// search.  The reference walks the stream one sample at a time through two float recurrences,
//     sLevel += 0.00001f * (|x| - sLevel)                                  (sample_reader.cpp:245-248, every sample read)
//     level  += |x| - |x 50 samples earlier|                               (timesyncer.cpp:64-66, 78-80: 50-tap moving sum)
// and tests  level / 50 > 0.55 sLevel  (dip begins)  /  level / 50 < 0.75 sLevel  (dip ends)  BEFORE it reads the next sample.
// Only the two recurrences are serial, and sLevel -- three dependent float operations per sample -- depends on nothing but
// the samples.  So the four waves of the block form a pipeline over blocks of 1024 samples, one barrier per block:
//   wave 0:           sLevel over block i (every lane the same walk), operands and results as 16-byte LDS accesses -- it never waits for anything else
//                     and sets the pace (acq_walk_S: ~24 cycles per sample for a lone wave, tools/acq_walk_bench.hip);
//   wave 1:           everything else about block i - 1: the moving sum's increments d[n] = |x[n]| - |x[n - 50]| (64 lanes, the
//                     same float subtraction), level (one lane, one add per sample), then the two comparisons of all 1024
//                     positions from the stored (level, sLevel) pairs -- the IEEE division included -- collected with ballots;
//                     the phase machine (first dip begin, first dip end at or behind it, the two time-outs, the next attempt)
//                     is scalar code on those masks.  The sample at which an attempt gives up is known in advance (T_F + 50
//                     samples after its start without a dip, T_n + 70 after the dip's begin without an end): the block is cut
//                     there and the next attempt's level restarts from zero mid-block;
//   waves 2, 3:       |x| of block i + 1 (coalesced loads, square roots);
//   peakLevel is a maximum: taken in parallel over the samples consumed.
// Same float operations in the same order on the same values as the sample-serial form: sLevel, the sample the search stops at
// and every decision are bit-identical to it (and to the oracle).  A frame of silence (70 attempts) is walked in ~2 ms; round 3
// (one thread, every operand an LDS round trip behind a polled flag, ~1000 cycles per sample) took 80-100.
constexpr int ACQ_CH = 1024;
struct AcqLds {
  __attribute__((aligned(16))) float a[3][64 + ACQ_CH + 32];  // |x| of three consecutive blocks: a[j % 3][64 + i] = sample i of block j, [0..63] = the 64 samples before it
  __attribute__((aligned(16))) float r[3][64 + ACQ_CH + 32];  // |x * osc| of the same blocks: what the time syncer sees (the sample reader hands it the sample times
                                                              // oscillatorTable[currentPhase], sample_reader.cpp:274-281 -- a constant phasor while searching, but the rounding of the product is in the magnitude)
  __attribute__((aligned(16))) float d[2][ACQ_CH + 32];       // increments of the moving sum |x osc|[n] - |x osc|[n - 50] of block j in d[j & 1]
  float bmax[3][4];                                           // maximum of a[j % 3] in four parts
  float Sc[2][ACQ_CH / 16 + 8];                               // sLevel CHECKPOINTS of block j: Sc[j & 1][1 + k] after sample 16 k + 15, [0] before the block
  float Lc[ACQ_CH / 16 + 8];                                  // level checkpoints of the block being evaluated: Lc[1 + k] after sample 16 k + 15
  unsigned long long consumed;                                // results of the pass (wave 1 -> everyone)
  float s_final, pk;
  int done, ok, margin;
  float red[8];
  int flag[4];
};

// sLevel alone over m samples, m a multiple of 16 (the T_u window of a failed correlation; k_frame_head has no room for acq_walk_S's
// registers): sixteen operands are requested before the first is used, so that one LDS latency is paid per 16 samples, not per 4
__device__ __forceinline__ float level_walk(const float *__restrict__ a, int m, float S)
{
  const float4 *a4 = reinterpret_cast<const float4 *>(a);
  for (int i = 0; i < (m >> 2); i += 4) {
    const float4 v0 = a4[i], v1 = a4[i + 1], v2 = a4[i + 2], v3 = a4[i + 3];
    asm volatile("" ::: "memory");
#define DABX_LV4(v) S += 0.00001f * (v.x - S); S += 0.00001f * (v.y - S); S += 0.00001f * (v.z - S); S += 0.00001f * (v.w - S);
    DABX_LV4(v0) DABX_LV4(v1) DABX_LV4(v2) DABX_LV4(v3)
#undef DABX_LV4
  }
  return S;
}
// maximum of non-negative floats over the block (red: >= 8 floats of LDS; every thread gets the result)
__device__ __forceinline__ float block_max_nonneg(float v, float *red, int tid)
{
  const unsigned w = wave_butterfly_u32(__builtin_bit_cast(unsigned, v), [](unsigned x, unsigned y) { return x > y ? x : y; });   // order of non-negative floats = order of their bits
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = __builtin_bit_cast(float, w);
  __syncthreads();
  float r = red[0];
  for (int q = 1; q < (int)(blockDim.x >> 6); q++) r = fmaxf(r, red[q]);
  return r;
}

// One pass of the search for stream s (whole block, 256 threads).  The caller owns the stream (it is not in ST_EVAL_SYNC and
// no demapper launch of it is in flight).  wr = the committed-sample count this kernel works with.
// Returns -1: nothing done (too few samples), 0: searched, no end of a null symbol yet, 1: a null symbol ended at c.rd.
__device__ __forceinline__ int acquire_stream(EngineDev &e, int s, int tid, int st, unsigned long long wr,
                                              unsigned long long budget_samples, AcqLds &w)
{
  constexpr int T = 256;
  StreamCtl &c = e.ctl[s];
  const unsigned long long rd0 = c.rd, avail = wr - rd0;
  if (avail < (unsigned long long)ACQ_NEED) return -1;
  // WAIT_FOR_TIME_SYNC_MARKER entry (dab_processor.cpp:146-153): decoder reset
  for (int i = tid; i < K; i += T) {
    e.demap.integ[(size_t)s * K + i] = 0.f; e.demap.mean_power[(size_t)s * K + i] = 0.f; e.demap.mean_sigma[(size_t)s * K + i] = 0.f;
    e.demap.std_dev[(size_t)s * K + i] = 0.f;
  }
  for (int i = tid; i < TU; i += T) { e.demap.null_power[(size_t)s * TU + i] = 0.f; e.demap.null_power2[(size_t)s * TU + i] = 0.f; }
  if (tid == 0) e.demap.mean_power_all[s] = 1.0f;
  if (e.tii_acc) {                     // mTiiDetector.reset(); mTiiCounter = 0 (dab_processor.cpp:150-152)
    for (int i = tid; i < TU; i += T) e.tii_acc[(size_t)s * TU + i] = make_float2(0.f, 0.f);
    if (tid == 0) { e.tii_cnt[2 * s] = 0; e.tii_cnt[2 * s + 1]++; }
  }
  const float2 *ring = e.iq + (size_t)s * e.ring_len;
  const unsigned len = (unsigned)e.ring_len, base = (unsigned)(rd0 % (unsigned long long)e.ring_len);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave-uniform: the roles below branch on the scalar unit
  // The search reads with frequency offset 0: currentPhase stays where the last frame left it and every sample is multiplied by the
  // same table entry (sample_reader.cpp:274-281).  sLevel and peakLevel are taken before that product (:245-248), the time syncer's
  // envelope after it (timesyncer.cpp:52, 64, 78): two magnitudes per sample, equal only while the phase is still 0.
  float2 osc;
  {
    double si, co;
    sincospi(2.0 * (double)c.nco_phase / (double)INPUT_RATE, &si, &co);
    osc = make_float2((float)co, (float)si);
  }
  // |x| of block j of this pass (and, in front of it, the last 64 magnitudes of block j - 1) by threads t0, t0 + nt, ...; nothing
  // beyond the committed samples is touched (a line fetched before its samples were committed would stay in this CU's cache
  // for the rest of the kernel)
  // ... in two halves for the waves that fetch while the others walk: request(j) puts the samples of block j into registers (8 per
  // thread of waves 2-3), publish(j) turns them into the two magnitudes -- a whole block later, so that the HBM latency of the loads
  // (several microseconds next to the frame chain's traffic) lies behind a block's walk instead of in front of every eight samples
  float2 raw[ACQ_CH / 128];
  auto request = [&](int j) {
    const unsigned long long p0 = (unsigned long long)j * ACQ_CH;
    const unsigned o0 = (unsigned)((base + p0) % len);     // one 64-bit modulo per block, then add + conditional subtract
#pragma unroll
    for (int u = 0; u < ACQ_CH / 128; u++) {
      const int q = (tid - 128) + 128 * u;
      unsigned o = o0 + (unsigned)q;
      if (o >= len) o -= len;
      raw[u] = p0 + (unsigned)q < avail ? ring[o] : make_float2(0.f, 0.f);
    }
  };
  auto publish = [&](int j) {
    float *dst = w.a[j % 3], *dsr = w.r[j % 3];
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < ACQ_CH / 128; u++) {
      const int q = (tid - 128) + 128 * u;
      const float2 v = raw[u];
      const float a = sqrtf(v.x * v.x + v.y * v.y);
      const float2 m = cmul(v, osc);
      dst[64 + q] = a; dsr[64 + q] = sqrtf(m.x * m.x + m.y * m.y);
      mx = fmaxf(mx, a);
    }
    if (tid - 128 < 64) dsr[tid - 128] = w.r[(j - 1) % 3][ACQ_CH + (tid - 128)];       // (j >= 1)
    mx = __builtin_bit_cast(float, wave_butterfly_u32(__builtin_bit_cast(unsigned, mx), [](unsigned x, unsigned y) { return x > y ? x : y; }));
    if (lane == 0) w.bmax[j % 3][wave] = mx;
    if (tid < 128 + 2) w.bmax[j % 3][tid - 128] = 0.f;     // two waves fill the block: the other two parts are empty
  };
  auto mags = [&](int j, int t0, int nt) {                 // block 0, by everyone, before the pipeline runs
    float *dst = w.a[j % 3], *dsr = w.r[j % 3];
    const unsigned long long p0 = (unsigned long long)j * ACQ_CH;
    const unsigned o0 = (unsigned)((base + p0) % len);
    float mx = 0.f;
    for (int q = t0; q < ACQ_CH; q += nt) {
      float a = 0.f, ar = 0.f;
      if (p0 + (unsigned)q < avail) {
        unsigned o = o0 + (unsigned)q;
        if (o >= len) o -= len;
        const float2 v = ring[o];
        a = sqrtf(v.x * v.x + v.y * v.y);
        const float2 m = cmul(v, osc);
        ar = sqrtf(m.x * m.x + m.y * m.y);
      }
      dst[64 + q] = a; dsr[64 + q] = ar;
      mx = fmaxf(mx, a);
    }
    mx = __builtin_bit_cast(float, wave_butterfly_u32(__builtin_bit_cast(unsigned, mx), [](unsigned x, unsigned y) { return x > y ? x : y; }));
    if (lane == 0) w.bmax[j % 3][wave] = mx;               // (order of non-negative floats = order of their bits)
  };
  // the increments of block j as they are in the middle of an attempt (its first 50 samples are patched by the search, below)
  auto incs = [&](int j, int t0, int nt) {
    const float *rb = w.r[j % 3] + 64;
    float *dst = w.d[j & 1];
    for (int q = t0; q < ACQ_CH; q += nt) dst[q] = rb[q] - rb[q - 50];       // timesyncer.cpp:78-80
  };
  if (tid == 0) { w.done = 0; w.ok = 0; }
  if (wave >= 2) request(1);
  mags(0, tid, T);
  __syncthreads();
  incs(0, tid, T);                       // (block 0 has no samples in front of it: its first 50 increments are never used as they are)
  __syncthreads();
  // wave 0: sLevel
  float S = c.s_level;
  LevelPar lp;
  if (wave == 0) lp.init(lane);
  // wave 1: the search proper (all of it wave-uniform)
  int phase = (st == ST_INIT) ? 0 : 1;   // 0: seeding the level (20 T_u samples, dab_processor.cpp:130-139); 1: looking for the begin of a dip; 3: for its end
  int nb = 0;                            // samples of the seed / of the attempt evaluated so far
  int n2 = 0;                            // attempt-relative index at which the dip began
  float L = 0.f, pk = 0.f;
  int margin = 0;                        // comparisons within 1e-4 of their threshold
#ifdef DABX_ACQ_TIMING                   // experiment builds only (tools/build_variant.sh): where a block's time goes, per wave role
  long long tm[6] = {0, 0, 0, 0, 0, 0}, t_loop = clock64();
#define ACQ_T0 const long long t0_ = clock64();
#define ACQ_T(k) tm[k] += clock64() - t0_;
#else
#define ACQ_T0
#define ACQ_T(k)
#endif
  int i = 0;
  for (;; i++) {
    if (wave == 0) {
      // the block's 64 groups of 16 samples in parallel, one per lane (level_par.h): Sc[g] = sLevel before group g, bit for bit what
      // the sample-serial walk gives (round 4 began with that walk here: 17.6 cycles per sample, 20 000 per block, the pace of the search)
      ACQ_T0
      S = lp.block(w.a[i % 3] + 64, ACQ_CH / 16, S, w.Sc[i & 1], lane);
      ACQ_T(0)
    } else if (wave == 1) {
      if (i > 0) {
        const int jb = i - 1;
        const float *ab = w.a[jb % 3] + 64, *rb = w.r[jb % 3] + 64, *Scb = w.Sc[jb & 1];
        const unsigned long long P = (unsigned long long)jb * ACQ_CH;
        // sLevel BEFORE sample pos of this block (pos <= ACQ_CH), from the checkpoint of its group: a 16-step walk by the lane that owns it
        auto s_before = [&](int pos) {
          if (pos == ACQ_CH) return Scb[ACQ_CH / 16];
          float Sx = Scb[lane];                                             // before sample 16 * lane
          float cap = 0.f;
#pragma unroll
          for (int k = 0; k < 16; k++) {
            if (16 * lane + k == pos) cap = Sx;
            Sx += 0.00001f * (ab[16 * lane + k] - Sx);
          }
          return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cap), pos >> 4));
        };
        int q = 0, stop = 0, ok = 0;
        while (q < ACQ_CH && !stop) {
          // the segment ends where the attempt would time out: NO_DIP_FOUND after reading sample T_F + 50 of the attempt (counter > T_F,
          // timesyncer.cpp:68-71), NO_END_OF_DIP_FOUND after sample n2 + T_n + 70 (counter > T_n + 50 + 20, :82-85)
          const int last = phase == 0 ? 20 * TU : (phase == 3 ? n2 + TN + 71 : TF + 51);
          const int m = min(ACQ_CH - q, last - nb);
          int lim = m;                                                       // samples of the segment that are consumed
          if (phase != 0) {
            // An attempt that starts mid-block starts its level at zero: the walk begins at the 16-sample boundary below q with
            // zero increments up to q (what it writes there belongs to positions already evaluated)
            const int q16 = q & ~15, g0 = q16 >> 4;
            float *db = w.d[jb & 1];
            { ACQ_T0
            if (nb < 50 || q16 < q)                                          // the attempt's first 50 samples: level += |x| (:64-66)
              for (int p = q16 + lane; p < q16 + 128; p += 64) {
                if (p < q) db[p] = 0.f;
                else if (p < q + m && nb + (p - q) < 50) db[p] = rb[p];
              }
            ACQ_T(1) }
            { ACQ_T0
            __builtin_amdgcn_wave_barrier();
            // level before sample q16 (= before q: zero increments in between) -> w.Lc[g0], then checkpoint by checkpoint
            level_sum_block(db + q16, __builtin_amdgcn_readfirstlane((q + m - q16 + 15) >> 4), L, w.Lc + g0, lane);
            __builtin_amdgcn_wave_barrier();
            ACQ_T(2) }
            ACQ_T0
            // Lane t owns samples 16 t .. 16 t + 15: from the two checkpoints in front of its group it re-walks both recurrences in
            // registers and evaluates, BEFORE each sample, the two comparisons of timesyncer.cpp:58, 74 -- 16 bits per lane each
            unsigned bm = 0, em = 0, nbm = 0, nem = 0;                       // dip begins / ends here; the comparison came within 1e-4 of its threshold
            const int p_last = q + m - 1;
            float L_end = 0.f;
            {
              float Sx = Scb[lane], Lx = w.Lc[lane];
              // the group's 16 samples and 16 increments as eight 16-byte LDS reads, four samples at a time (one read per use, each waited
              // for, and a branch per comparison, was 7 500 - 9 500 of the block's cycles), and the comparisons straight-line: bits for all 16
              // positions, masked afterwards with the positions that count -- inside the segment, 50 samples or more into the attempt
              const float4 *pa = reinterpret_cast<const float4 *>(ab + 16 * lane), *pd = reinterpret_cast<const float4 *>(db + 16 * lane);
              const int k_last = p_last - 16 * lane;
#pragma unroll
              for (int v = 0; v < 4; v++) {
                const float4 fa = pa[v], fd = pd[v];
                const float xa[4] = {fa.x, fa.y, fa.z, fa.w}, xd[4] = {fd.x, fd.y, fd.z, fd.w};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                  const int k = 4 * v + u;
                  const float mean = Lx / 50.f, tb = 0.55f * Sx, te = 0.75f * Sx;
                  bm |= (unsigned)(!(mean > tb)) << k;
                  em |= (unsigned)(!(mean < te)) << k;
                  nbm |= (unsigned)(fabsf(mean - tb) <= 1e-4f * tb) << k;
                  nem |= (unsigned)(fabsf(mean - te) <= 1e-4f * te) << k;
                  Sx += 0.00001f * (xa[u] - Sx);
                  Lx += xd[u];
                  L_end = k == k_last ? Lx : L_end;
                }
              }
              // positions p = 16 lane + k with q' <= p < q + m, q' = q + (what is missing to 50 samples of the attempt), in groups g0 and up
              const int lo = q + (nb < 50 ? 50 - nb : 0) - 16 * lane, hi = q + m - 16 * lane;
              const unsigned below_hi = hi >= 16 ? 0xFFFFu : (hi <= 0 ? 0u : (1u << hi) - 1u);
              const unsigned from_lo = lo <= 0 ? 0xFFFFu : (lo >= 16 ? 0u : (0xFFFFu << lo) & 0xFFFFu);
              const unsigned vm = lane >= g0 ? (below_hi & from_lo) : 0u;
              bm &= vm; em &= vm; nbm &= vm; nem &= vm;
            }
            int p2 = -1, p3 = -1;                                            // block positions of the dip's begin / end in this segment
            if (phase != 3) {
              const unsigned long long any = __ballot(bm != 0);
              if (any) {
                const int t2 = __builtin_ctzll(any);
                p2 = 16 * t2 + __builtin_ctz((unsigned)__builtin_amdgcn_readlane((int)bm, t2));
                phase = 3; n2 = nb + (p2 - q);
              }
            }
            if (phase == 3) {
              // the dip's end is looked for from the same sample on (:74)
              const unsigned em2 = p2 < 0 ? em : (lane < (p2 >> 4) ? 0u : (lane == (p2 >> 4) ? em & (~0u << (p2 & 15)) : em));
              const unsigned long long any = __ballot(em2 != 0);
              if (any) {
                const int t3 = __builtin_ctzll(any);
                p3 = 16 * t3 + __builtin_ctz((unsigned)__builtin_amdgcn_readlane((int)em2, t3));
              }
              // comparisons a relative error of 1e-4 in sLevel could have turned: begin comparisons up to the begin, end comparisons from it to the end
              const int lo = p2 < 0 ? 0 : p2, hi = p3 < 0 ? ACQ_CH : p3;
              unsigned ne = nem, nbq = (p2 < 0) ? 0u : nbm;                    // (p2 < 0: the segment began in phase 3: no begin comparisons at all)
              if (p2 >= 0) { if (lane > (p2 >> 4)) nbq = 0; else if (lane == (p2 >> 4)) nbq &= ~(~1u << (p2 & 15)); }
              if (lane < (lo >> 4)) ne = 0; else if (lane == (lo >> 4)) ne &= ~0u << (lo & 15);
              if (hi != ACQ_CH) { if (lane > (hi >> 4)) ne = 0; else if (lane == (hi >> 4)) ne &= ~(~1u << (hi & 15)); }
              margin += wave_sum_int(__builtin_popcount(nbq) + __builtin_popcount(ne));
              if (p3 >= 0) { lim = p3 - q; ok = 1; stop = 1; }
            } else margin += wave_sum_int(__builtin_popcount(nbm));
            ACQ_T(3)
            L = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, L_end), p_last >> 4));   // level after the segment's last sample
          }
          q += lim;
          if (ok) break;
          nb += m;
          if (nb == last) {
            if (phase == 0) { phase = 1; nb = 0; L = 0.f; }
            else {
              // NO_DIP_FOUND / NO_END_OF_DIP_FOUND: dab_processor.cpp:154-160 tries again at once.  The same here while the next
              // attempt's worst case is still in the ring and the pass's sample budget is not used up (one attempt is only
              // T_n + 121 samples long in silence): a stream in a drop-out walks through it a frame of samples per pass.
              const unsigned long long consumed = P + (unsigned)q;
              if (avail - consumed >= (unsigned long long)ACQ_NEED && consumed < budget_samples) { phase = 1; nb = 0; L = 0.f; }
              else stop = 1;
            }
          }
        }
        // sample_reader.cpp:247: peakLevel over the samples consumed -- the whole block's maximum, or its first q samples at the end
        { ACQ_T0
        if (!stop) pk = fmaxf(pk, lane < 4 ? w.bmax[jb % 3][lane] : 0.f);
        else for (int p = lane; p < q; p += 64) pk = fmaxf(pk, ab[p]);
        ACQ_T(4) }
        if (stop) {
          const float pkw = __builtin_bit_cast(float, wave_butterfly_u32(__builtin_bit_cast(unsigned, pk), [](unsigned x, unsigned y) { return x > y ? x : y; }));
          const float s_fin = s_before(q);
          if (lane == 0) { w.consumed = P + (unsigned)q; w.s_final = s_fin; w.pk = pkw; w.ok = ok; w.margin = margin; w.done = 1; }
        }
      }
    } else {
      ACQ_T0
      publish(i + 1);                                      // requested a block ago
      request(i + 2);
      if (i > 0) incs(i, tid - 128, T - 128);             // block i's magnitudes are complete since the last barrier
      ACQ_T(5)
    }
    __syncthreads();
    if (w.done) break;
  }
#ifdef DABX_ACQ_TIMING
  if (s == 0 && lane == 0 && wave < 3)
    printf("acq wave %d: %d blocks, %lld cycles in all; S walk %lld, increments %lld, level walk %lld, comparisons %lld, peak %lld, magnitudes %lld (cycles per block)\n",
           wave, i + 1, clock64() - t_loop, tm[0] / (i + 1), tm[1] / (i + 1), tm[2] / (i + 1), tm[3] / (i + 1), tm[4] / (i + 1), tm[5] / (i + 1));
#endif
  if (tid == 0) {
    c.rd = rd0 + w.consumed;           // frequency offset is 0 while searching: NCO phase unchanged
    c.s_level = w.s_final; c.peak_level = fmaxf(c.peak_level, w.pk);
    c.level_margin += w.margin;
    c.sample_count = 0;
    c.sync_thr = e.threshold;
    c.clock_err = 0.0f;
  }
  const int ok = w.ok;
  __syncthreads();
  return ok;
}

// A phase-reference correlation failed at rd (dab_processor.cpp:396-400 -> WAIT_FOR_TIME_SYNC_MARKER): the T_u samples just read
// went through SampleReader's level tracker (sample_reader.cpp:245-248) -- run it exactly, sample by sample: right after
// start-up the level is still far from settled (it starts at 0.1) and the null-dip detector of the next attempt compares
// against it.  buf: >= T_u floats of LDS (16-byte aligned), red: >= 8.  The caller changes c.state.
__device__ __forceinline__ void sync_failed(EngineDev &e, int s, int tid, const RingView &rv, unsigned long long rd, int phase0, int f,
                                            float *buf, float *red, bool track_level)
{
  StreamCtl &c = e.ctl[s];
  float pk = 0.f;
  if (track_level) {                                       // (not the frame chain with cfg.exact_level_tracker: k_level_exact walks everything it reads)
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const float2 x = rv.at(tid + 256 * u);
      const float a = sqrtf(x.x * x.x + x.y * x.y);
      buf[tid + 256 * u] = a;
      pk = fmaxf(pk, a);
    }
    pk = block_max_nonneg(pk, red, tid);                   // (its barriers also publish buf)
  }
  if (tid == 0) {
    if (track_level) {
      c.s_level = level_walk(buf, TU, c.s_level);
      c.peak_level = fmaxf(c.peak_level, pk);
    }
    c.rd = rd + TU;
    c.nco_phase = nco_advance(phase0, f, TU);
    c.sync_lost++;
  }
  __syncthreads();
}

// cfg.exact_level_tracker = 0 (EngineDev::anchor_level, the default).  In lock nobody reads sLevel, so the frame chain only advances it
// chunk-wise (k_frame_tail: ~1e-5 relative).  The search reads it -- and decides where a null symbol begins and ends by comparing
// against it -- so before a stream that the frame chain had goes back into the search, the level is brought to where the sample-serial
// recurrence has it: walked (level_par.h, 1024 samples at a time) from the ANCHOR -- the position and value at which the search last
// handed the stream over -- over everything the receiver has read since.  That needs those samples to be in the ring still: a push
// announces what it may overwrite before it starts (EngineDev::wr_horizon, host memory), the walk looks before and after.  If they are
// not (a long time in lock with a short ring, or a zero-copy producer whose writes the library does not see), the level continues from
// the chunk-wise value over the samples read since (the T_u window of the failed correlation), as it did before round 4, and the
// event is counted (dabx_stats.level_unanchored_events; level_rewalk_events counts the exact ones).
__device__ __forceinline__ void level_from_anchor(EngineDev &e, int s, int tid, AcqLds &w)
{
  constexpr int T = 256;
  StreamCtl &c = e.ctl[s];
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned long long p1 = c.rd, len64 = (unsigned long long)e.ring_len;
  const float2 *ring = e.iq + (size_t)s * e.ring_len;
  const unsigned len = (unsigned)e.ring_len;
  LevelPar lp;
  if (wave <= 1) lp.init(lane);
  // How the level gets to rd, best first:
  //   1  from the anchor, exactly (everything read since the hand-over is still in the ring);
  //   2  from the oldest frame boundary still in the ring whose chunk-wise value the frame tail has kept (lvl_hist_*): TWO walks, from
  //      that value -/+ 2^-9 (two hundred times the chunk-wise tracker's observed error, six times what one symbol with a 100 % level step in it can contribute).  The recurrence forgets: two trajectories close in
  //      on each other by 1e-5 of their distance per sample and, one float apart, merge for good within ~1e5 samples -- after four to
  //      eight frames they are the SAME float, and then so is every trajectory that started in between (the step is monotone in the
  //      level while no sample is thousands of times larger than it: checked): the exact value, with a certificate;
  //   0  from the chunk-wise value over the samples read since (the T_u window of the failed correlation) -- as before round 4; counted.
  int mode = 0;
  float S = 0.f, pk = 0.f;
  for (int attempt = 0; attempt < 2; attempt++) {
    if (tid == 0) {                                          // one thread looks, everyone follows
      const unsigned long long hz = __hip_atomic_load(&e.wr_horizon[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      int m = 0, pick = -1;
      if (attempt == 0) {
        // (1 only over at most 20 frames: a producer that announced its ring once, a periodic test signal, passes the horizon check for ever,
        //  and a stream that falls out of lock after a long time in it would walk every sample since the hand-over -- seconds of kernel time on
        //  the step's critical path.  Beyond that the two-walk merge from the frame history gives the same float with a certificate.)
        if (hz <= c.lvl_anchor_pos + len64 && p1 - c.lvl_anchor_pos <= 20ull * TF) m = 1;
        else {
          unsigned long long best = ~0ull;
          for (int h = 0; h < LVL_HIST; h++) {
            const unsigned long long hp = c.lvl_hist_pos[h];
            if (hp > c.lvl_anchor_pos && hp <= c.lvl_approx_pos && hz <= hp + len64 && p1 - hp >= 4ull * TF && hp < best) { best = hp; pick = h; }
          }
          if (pick >= 0) m = 2;
        }
      }
      w.flag[2] = m; w.flag[3] = pick;
    }
    __syncthreads();
    mode = w.flag[2];
    const int pick = w.flag[3];
    const unsigned long long p0 = mode == 1 ? c.lvl_anchor_pos : (mode == 2 ? c.lvl_hist_pos[pick] : c.lvl_approx_pos);
    const float Sh = mode == 2 ? c.lvl_hist_S[pick] : 0.f;
    S = mode == 1 ? c.lvl_anchor_S : (mode == 2 ? (wave == 0 ? Sh - Sh * 0x1p-9f : Sh + Sh * 0x1p-9f) : c.s_level);
    float s_min = S;
    const unsigned long long n = p1 - p0;
    const int nblk = (int)((n + ACQ_CH - 1) / ACQ_CH);
    float mx = 0.f;
    auto mags = [&](int j, int t0, int nt) {                 // |x| of block j into w.a[j & 1] (zeros beyond the last sample)
      float *dst = w.a[j & 1] + 64;
      const unsigned long long b0 = (unsigned long long)j * ACQ_CH;
      const unsigned o0 = (unsigned)((p0 + b0) % len64);
      for (int q = t0; q < ACQ_CH; q += nt) {
        float a = 0.f;
        if (b0 + (unsigned)q < n) {
          unsigned o = o0 + (unsigned)q;
          if (o >= len) o -= len;
          const float2 v = ring[o];
          a = sqrtf(v.x * v.x + v.y * v.y);
        }
        dst[q] = a;
        mx = fmaxf(mx, a);
      }
    };
    if (nblk > 0) mags(0, tid, T);
    __syncthreads();
    for (int i = 0; i < nblk; i++) {
      if (wave == 0 || (wave == 1 && mode == 2)) {           // wave 0: the walk (mode 2: the lower one); wave 1: the upper one
        const float *ab = w.a[i & 1] + 64;
        const unsigned long long left = n - (unsigned long long)i * ACQ_CH;
        const int m = left < (unsigned long long)ACQ_CH ? (int)left : ACQ_CH;
        const int n16 = __builtin_amdgcn_readfirstlane(m >> 4);
        if (n16 > 0) S = lp.block(ab, n16, S, nullptr, lane);
        for (int r = 16 * n16; r < m; r++) S += 0.00001f * (ab[r] - S);
        s_min = fminf(s_min, S);
      } else if (wave >= 2 && i + 1 < nblk) mags(i + 1, tid - 128, T - 128);
      __syncthreads();
    }
    pk = block_max_nonneg(mx, w.red, tid);
    if (mode == 0) break;
    if (mode == 2) {                                         // merged?  (and monotone all the way: no sample beyond 512 x the lowest level seen)
      if (lane == 0 && wave <= 1) { w.Lc[wave] = S; w.Lc[2 + wave] = s_min; }
      __syncthreads();
      const bool same = __builtin_bit_cast(unsigned, w.Lc[0]) == __builtin_bit_cast(unsigned, w.Lc[1]) && pk <= 512.f * fminf(w.Lc[2], w.Lc[3]) && w.Lc[2] > 0.f;
      S = w.Lc[0];
      __syncthreads();
      if (!same) continue;                                   // not yet (a short window, a level that rose a thousandfold): mode 0
    }
    if (tid == 0) {                                          // nothing of it was overwritten while it was read?
      const unsigned long long hz = __hip_atomic_load(&e.wr_horizon[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      w.flag[2] = hz <= p0 + len64 ? 1 : 0;
    }
    __syncthreads();
    const bool still = w.flag[2] != 0;
    __syncthreads();
    if (still) break;
  }
  // (wave 0 holds the level; S is the same in all its lanes)
  if (tid == 0) {
    if (c.lvl_approx_pos != c.lvl_anchor_pos) { if (mode == 1) c.lvl_rewalks++; else if (mode == 2) c.lvl_healed++; else c.lvl_unanchored++; }
    c.s_level = S; c.peak_level = fmaxf(c.peak_level, pk);
    c.lvl_anchor_pos = c.lvl_approx_pos = p1; c.lvl_anchor_S = S;
  }
  __syncthreads();
}

// Acquisition kernel: one block per stream, does something only for streams that are NOT in lock.  Ownership of a stream's
// control record follows c.state: ST_EVAL_SYNC = the frame chain (k_frame_head ... k_frame_tail on HIP stream a), anything
// else = this kernel.  It therefore runs either in step (on stream a, before k_frame_head: every step waits for the streams
// that are searching -- the contract of dabx_process(sync != 0)) or on its own HIP stream next to the steps of the streams in
// lock (dabx_process(sync == 0): a step never waits for it; k_frame_head passes over a stream that is still being searched and
// picks it up in the first step after the search has handed it back).  Hand-over in both directions is one release store of
// c.state behind everything else the owner wrote, read with acquire by the other side.
// A pass = up to budget_frames frames of samples: null-dip search, phase-reference correlation of the candidate, and after a
// failed correlation straight back to the search like the reference (one candidate per step starved streams in deep fades:
// false dips every few thousand samples -- fuzz seed 5001).  A candidate that correlates is left at ST_EVAL_SYNC with the read
// cursor in front of it: k_frame_head repeats the correlation (same samples, same threshold, same result) and decodes the frame.
__global__ __launch_bounds__(256, 2) void k_acquire(EngineDev e, DevTables t, int budget_frames)
{
  __shared__ AcqLds w;
  __shared__ float2 lds[FFT_LDS_FLOAT2];
  __shared__ __attribute__((aligned(16))) float peak[TU];
  __shared__ float red[8];
  __shared__ unsigned long long s_wr;
  const int s = blockIdx.x, tid = threadIdx.x;
  StreamCtl &c = e.ctl[s];
  front_prio();                                            // four lone waves per stream: almost no issue slots, but the ones they need, at once
  if (tid == 0) {                                          // one thread looks, everyone follows (a flag that flips meanwhile must not split the block)
    const int st0 = __hip_atomic_load(&c.state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    // The search resets the demapper (dab_processor.cpp:146-153).  The MSC symbols of the stream's last frame may still be going
    // through it on their own HIP stream: then this pass is skipped (no new launch for the stream can start while it is out of lock)
    const int busy = e.demap_busy ? __hip_atomic_load(&e.demap_busy[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : 0;
    // cfg.exact_level_tracker: the search continues the level where k_level_exact (in front of this kernel on the same HIP stream)
    // left it -- which must be where the stream stands (a correlation that failed in between has moved rd: next pass)
    const int behind = e.exact_level ? (e.level_pos[s] != c.rd) : 0;
    w.flag[0] = (st0 != ST_EVAL_SYNC && !busy && !behind) ? 1 : 0;
    w.flag[1] = st0;
    s_wr = e.wr[s];
  }
  __syncthreads();
  if (!w.flag[0]) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  int st = w.flag[1], new_state = -1;
  const unsigned long long wr = s_wr, rd_start = c.rd, budget = (unsigned long long)budget_frames * TF;
  const float2 *ring = e.iq + (size_t)s * e.ring_len;
  if (e.anchor_level && c.lvl_anchor_pos != c.rd) level_from_anchor(e, s, tid, w);
  for (;;) {
    const unsigned long long used = c.rd - rd_start;
    if (used >= budget) break;
    const int r = acquire_stream(e, s, tid, st, wr, budget - used, w);
    if (r < 0) break;
    st = new_state = ST_WAIT_SYNC;
    if (r == 0) break;
    new_state = ST_EVAL_SYNC;
    if (wr - c.rd < (unsigned long long)FRAME_NEED) break;   // k_frame_head correlates when the frame's samples are there
    const unsigned long long rd = c.rd;
    const int phase0 = c.nco_phase, f = (int)roundf(c.f_bb);
    Nco nco;
    nco.init(phase0, f, tid);
    const RingView rv(ring, e.ring_len, rd);
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { v[u] = nco.mix(rv.at(tid + 256 * u)); nco.step(); }
    const int start = prs_correlate_block(v, c.sync_thr, e.strongest, t, lds, peak, red, tid);   // dab_processor.cpp:394
    __syncthreads();
    if (start >= 0) break;
    sync_failed(e, s, tid, rv, rd, phase0, f, peak, red, true);
    new_state = ST_WAIT_SYNC;
    if (wr - c.rd < (unsigned long long)(ACQ_NEED + FRAME_NEED)) break;
  }
  if (new_state < 0) return;
  if (tid == 0 && e.exact_level) e.level_pos[s] = c.rd;    // every sample read here went through the level tracker
  if (tid == 0 && e.anchor_level) { c.lvl_anchor_pos = c.lvl_approx_pos = c.rd; c.lvl_anchor_S = c.s_level; }   // ... sample by sample: exact up to here
  if (tid == 0 && new_state == ST_EVAL_SYNC && e.locked_count) __hip_atomic_fetch_add(e.locked_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&c.state, new_state, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// --------------------------------------------------------------------------------------------- frame head
#ifndef DABX_HEAD_OCC
#define DABX_HEAD_OCC 3      // 166 VGPRs, no scratch
#endif
__global__ __launch_bounds__(256, DABX_HEAD_OCC) void k_frame_head(EngineDev e, DevTables t)
{
  front_prio();
  __shared__ float2 lds[FFT_LDS_FLOAT2];
  __shared__ __attribute__((aligned(16))) float peak[TU];
  __shared__ float red[8];
  __shared__ float mag[160];
  __shared__ int s_state;
  const int s = blockIdx.x, tid = threadIdx.x;
  StreamCtl &c = e.ctl[s];
  const float2 *ring = e.iq + (size_t)s * e.ring_len;
  // first kernel of a step: "no frame yet"
  if (tid < 76) e.sym_off[(size_t)s * 76 + tid] = -1;
  if (tid == 0) {
    c.frame_ok = 0;
    // a stream out of lock belongs to k_acquire (which may be running right now on its own HIP stream): pass over it
    s_state = __hip_atomic_load(&c.state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_state != ST_EVAL_SYNC) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (e.wr[s] - c.rd < (unsigned long long)FRAME_NEED) return;
  const unsigned long long rd = c.rd;
  const int phase0 = c.nco_phase;
  const int f = (int)roundf(c.f_bb);                       // sample_reader.cpp:211
  float2 v[8];
  Nco nco;
  nco.init(phase0, f, tid);
  const RingView rv(ring, e.ring_len, rd);
  float abs_a = 0.f, abs_b = 0.f;                          // level tracker: sum |x| of what this frame head reads
#pragma unroll
  for (int u = 0; u < 8; u++) { const float2 x = rv.at(tid + 256 * u); abs_a += cabsf_level(x); v[u] = nco.mix(x); nco.step(); }
  const int start = prs_correlate_block(v, c.sync_thr, e.strongest, t, lds, peak, red, tid);   // :394
  __syncthreads();
  if (start < 0) {
    // :396-400 -> WAIT_FOR_TIME_SYNC_MARKER: the stream goes over to k_acquire
    // (the level over the T_u samples just read: here from the chunk-wise value in the round-3 mode only; k_level_exact or k_acquire's
    //  re-walk from the anchor see to it otherwise)
    sync_failed(e, s, tid, rv, rd, phase0, f, peak, red, !e.exact_level && !e.anchor_level);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(&c.state, (int)ST_WAIT_SYNC, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (e.locked_count) __hip_atomic_fetch_add(e.locked_count, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  // symbol 0 = samples [start, start + Tu) of the same (mixed) stream, :402-411
  nco.init(phase0, f, (long long)start + tid);
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const float2 x = rv.at(start + tid + 256 * u);
    if (start + tid + 256 * u >= TU) abs_b += cabsf_level(x);   // the start_index samples read beyond the correlation window
    v[u] = nco.mix(x); nco.step();
  }
  block_sum2(abs_a, abs_b, red, tid);
  fft2048<false>(v, lds, t.twiddle, tid);                 // dab_processor.cpp:199-201
#pragma unroll
  for (int u = 0; u < 8; u++) e.demap.phase_ref[(size_t)s * TU + tid + 256 * u] = v[u];   // store_reference_symbol_0

  int correction = 0;
  float f_sync = c.f_sync, f_bb = c.f_bb, clock_err = c.clock_err;
  if (c.fic_ratio * 10 < 30) {                            // :205-224
    correction = coarse_cfo_block(v, t, lds, mag, tid);
    if (correction != 100000) {
      f_sync += (float)correction;
      if (fabsf(f_sync) > 35000.0f) f_sync = 0.0f;
    }
    if (correction != 0) clock_err = 0.0f;
    f_bb = f_sync;
  }
  e.nco_tid[(size_t)s * 256 + tid] = Nco::tid_factor((int)roundf(f_bb), tid);
  // NCO phasor of the first FFT-window sample of every symbol 1..75 (two double sincospi each): 75 lanes in parallel here
  // instead of one lane of every k_symbols block while its other 255 threads wait (12 % of that kernel's VALU issue)
  if (tid < 75) {
    double2 b, st;
    Nco::block_consts(nco_advance(phase0, f, (long long)start + TU), (int)roundf(f_bb), (long long)tid * TS + TG, b, st);
    e.nco_sym[(size_t)s * 76 + tid] = b;
    if (tid == 0) e.nco_sym[(size_t)s * 76 + 75] = st;
    // where the symbol starts in the IQ ring (the 64-bit modulo once per symbol here, not once per k_symbols block)
    e.sym_off[(size_t)s * 76 + tid] = (int32_t)((rd + (unsigned long long)start + TU + (unsigned long long)tid * TS) % (unsigned long long)e.ring_len);
  }
  if (tid == 0) {
    c.start_index = start;
    c.head_abs_a = abs_a; c.head_abs_b = abs_b;
    c.sample_count = start + TU;
    c.sym0_pos = rd + start;
    c.phase_sym1 = nco_advance(phase0, f, (long long)start + TU);
    c.correction = correction;
    c.f_sync = f_sync; c.f_bb = f_bb; c.clock_err = clock_err;
    c.f_frame = (int)roundf(f_bb);
    c.frame_ok = 1;
  }
}

// ------------------------------------------------------------------------------------------------ symbols
// Symbols 1..75 of a frame: NCO mix, cyclic-prefix correlation, FFT, frequency de-interleave (dab_processor.cpp:304-341).
// (The null symbol is handled by k_frame_tail: it needs the fine-CFO update that depends on all 75 correlations.)
// PERSISTENT blocks, grid (G, S), G = sym_blocks_per_stream(S): a block walks symbols l = g, g + G, ... of its stream, requests the NEXT symbol's
// twelve samples per thread (4 for the cyclic-prefix correlation sum x[Tu+i] conj(x[i]), i < 504, taken on the RAW samples --
// the NCO contributes the constant factor e^{-j 2 pi f Tu / fs}, which k_frame_tail applies once -- and 8 for the transform;
// coalesced 8-byte loads) before it transforms the current one, and keeps everything that depends on the thread index only --
// twenty twiddles, the de-interleaver slots, the NCO factor -- in registers across symbols.  The block-uniform words (ring
// offset of the symbol, written per symbol by k_frame_head; NCO base) come over the scalar cache, so the only vector loads in
// the loop are the prefetch (vmcnt is in order: a table look-up behind it would wait for it).  tools/sym_mem_bound.hip: the
// memory side of this kernel runs at 0.23-0.25 ms per step in that shape, with or without the transform next to it.
// The three block sums ride on the FFT's barriers: per-wave partials go to LDS, thread 0 adds them in wave order.
// Frequency de-interleaving rides on the way out (freq_interleaver.cpp:40-76): each bin goes to its carrier's slot in LDS
// (the transform's exchange buffer is free again), the 1536 used carriers are then stored contiguously; the demapper reads
// carrier k of every symbol with coalesced loads, the 512 unused bins are never written.
// blocks per stream (a divisor of 75): 15 blocks of 5 symbols each fill the chip from ~50 streams up (3 resident per CU); with
// fewer streams the symbols of a frame are spread over more blocks so that the kernel's latency, not its throughput, shrinks
// (one stream: 75 blocks of one symbol -- the single-ensemble configurations are latency-bound on the frame's serial chain)
#ifndef DABX_SYM_G
#define DABX_SYM_G 15
#endif
__host__ __device__ constexpr int sym_blocks_per_stream(int n_streams) { return n_streams >= 48 ? DABX_SYM_G : (n_streams >= 16 ? 25 : 75); }
// 3 waves per SIMD: 170 VGPRs without spills (bounded to 4 it spills 8 registers and runs 25 % slower)
__global__ __launch_bounds__(256, 3) void k_symbols_persistent(EngineDev e, DevTables t)
{
  front_prio();
  __shared__ float2 lds[FFT_LDS_FLOAT2];
  __shared__ float red3[3][4];
  const int s = blockIdx.y + e.s0, tid = threadIdx.x;
  int l = blockIdx.x;
  int off = uniform_load(e.sym_off + (size_t)s * 76 + l);
  double2 nco_base = uniform_load(e.nco_sym + (size_t)s * 76 + l);
  const double2 nco_step = uniform_load(e.nco_sym + (size_t)s * 76 + 75);
  if (off < 0) return;                                       // no frame for this stream in this step (all 75 entries are -1 then)
  const float2 *ring = e.iq + (size_t)s * e.ring_len;
  const unsigned len = (unsigned)e.ring_len;
  const bool two = tid + 256 < TG;
  float2 nx[12];
  auto request = [&](int o) {
#if DABX_SYM_NT & 1          // (A/B builds: the samples past the caches, pipeline.h)
    typedef float sym_f2 __attribute__((ext_vector_type(2)));
    auto at = [&](unsigned i) { unsigned a = (unsigned)o + i; if (a >= len) a -= len; const sym_f2 q = __builtin_nontemporal_load(reinterpret_cast<const sym_f2 *>(ring + a)); return make_float2(q.x, q.y); };
#else
    auto at = [&](unsigned i) { unsigned a = (unsigned)o + i; if (a >= len) a -= len; return ring[a]; };
#endif
    nx[0] = at(tid); nx[1] = at(TU + tid);
    nx[2] = at(two ? tid + 256 : tid); nx[3] = at(two ? TU + tid + 256 : TU + tid);
#pragma unroll
    for (int u = 0; u < 8; u++) nx[4 + u] = at(TG + tid + 256 * u);
  };
  request(off);
  const double2 nco_t = e.nco_tid[(size_t)s * 256 + tid];
  FftTwiddles tw;
  fft_load_twiddles(tw, t.twiddle, tid);
  // LDS slot of the carrier each of this thread's eight bins goes to (-1: unused bin), and the low four slot bits of the six
  // carriers it stores afterwards (tables.cpp, carrier_slots: a permutation within each run of 16 carriers, chosen so that
  // neither the scatter nor the read-back has a bank conflict)
  const uint4 kk4 = reinterpret_cast<const uint4 *>(t.bin_to_slot8)[tid];
  const unsigned kkw[4] = {kk4.x, kk4.y, kk4.z, kk4.w};
  const unsigned rd_lo = t.carrier_slot_rd[tid];
  for (;;) {
    const float2 cb0 = nx[0], ca0 = nx[1], cb1 = nx[2], ca1 = nx[3];
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = nx[4 + u];
    const double2 base_now = nco_base;
    const int l_next = l + (int)gridDim.x;
    if (l_next < 75) {                                       // block-uniform
      off = uniform_load(e.sym_off + (size_t)s * 76 + l_next);
      nco_base = uniform_load(e.nco_sym + (size_t)s * 76 + l_next);
      request(off);
    }
    asm volatile("" ::: "memory");
    float cre = 0.f, cim = 0.f, asum = 0.f;
    cre += ca0.x * cb0.x + ca0.y * cb0.y;
    cim += ca0.y * cb0.x - ca0.x * cb0.y;
    asum += cabsf_level(cb0);
    if (two) {
      cre += ca1.x * cb1.x + ca1.y * cb1.y;
      cim += ca1.y * cb1.x - ca1.x * cb1.y;
      asum += cabsf_level(cb1);
    }
    Nco nco;
    nco.init_from(base_now, nco_step, nco_t);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const float2 x = v[u];
      asum += cabsf_level(x);
      v[u] = nco.mix(x);
      nco.step();
    }
    cre = wave_sum(cre); cim = wave_sum(cim); asum = wave_sum(asum);
    if ((tid & 63) == 0) { red3[0][tid >> 6] = cre; red3[1][tid >> 6] = cim; red3[2][tid >> 6] = asum; }
    fft2048_regs<false>(v, lds, tw, tid);
    if (tid == 0) {
      float r[3];
#pragma unroll
      for (int q = 0; q < 3; q++) { float a = 0.f; for (int w = 0; w < 4; w++) a += red3[q][w]; r[q] = a; }
      e.cp_part[(size_t)s * 75 + l] = make_float2(r[0], r[1]); e.abs_part[(size_t)s * 76 + l] = r[2];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int kk = (int)(int16_t)(kkw[u >> 1] >> (16 * (u & 1)));
      if (kk >= 0) lds[kk] = v[u];
    }
    __syncthreads();
    float2 *dst = e.spectra + (((size_t)e.parity * e.n_streams + s) * 75 + l) * K;
#ifdef DABX_SYM_ST_AUX       // experiment builds: the spectra through raw-buffer stores with the cache-policy bits of choice (16 = sc1: write-through)
    {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, K * 8, 0x00020000);
      typedef float sym_f2b __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int u = 0; u < K / 256; u++) {
        const float2 q = lds[((tid + 256 * u) & ~15) | ((rd_lo >> (4 * u)) & 15u)];
        sym_f2b qv; qv.x = q.x; qv.y = q.y;
        __builtin_amdgcn_raw_buffer_store_b64(qv, rs, (tid + 256 * u) * 8, 0, DABX_SYM_ST_AUX);
      }
    }
#else
#pragma unroll
#if DABX_SYM_NT & 2          // the spectra, read next by the demapper after 478 MB more have been written: past the caches (pipeline.h)
    for (int u = 0; u < K / 256; u++) {
      typedef float sym_f2 __attribute__((ext_vector_type(2)));
      const float2 q = lds[((tid + 256 * u) & ~15) | ((rd_lo >> (4 * u)) & 15u)];
      sym_f2 qv; qv.x = q.x; qv.y = q.y;
      __builtin_nontemporal_store(qv, reinterpret_cast<sym_f2 *>(dst + tid + 256 * u));
    }
#else
    for (int u = 0; u < K / 256; u++) dst[tid + 256 * u] = lds[((tid + 256 * u) & ~15) | ((rd_lo >> (4 * u)) & 15u)];
#endif
#endif
    if (l_next >= 75) break;
    l = l_next;
    __syncthreads();                                          // the exchange buffer and red3 are free again
  }
}

// ------------------------------------------------------------------------- device-side hand-overs (few streams)
// EngineDev::flag_sync.  The sequence numbers only grow; a block publishes after ALL its threads' stores (block barrier, then a device-scope
// release by one thread) and a waiting block lets one thread poll (s_sleep between polls: no issue slots taken from the kernel it waits for),
// then every thread passes a device-scope acquire.
__device__ __forceinline__ void seq_publish(uint32_t *p, uint32_t seq)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_store(p, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void seq_wait(const uint32_t *p, uint32_t seq, int32_t *timeouts)
{
  if (threadIdx.x == 0) {
    // bounded: what is waited for was launched BEFORE this kernel; if it never arrives a launch failed on the host, and a kernel that spins for
    // ever would take the GPU with it -- after ~2 s (8 M polls of ~0.25 us) the wait gives up, counts itself (host memory) and lets the block go on
    int spins = 0;
    while ((int32_t)(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (8 << 20)) {
        if (timeouts) __hip_atomic_fetch_add(timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// behind k_symbols on HIP stream a (k_symbols itself, 166 VGPRs at three waves per SIMD, stays as it is)
__global__ void k_sym_publish(EngineDev e)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < e.n_streams) __hip_atomic_store(e.sym_seq + s, e.step_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// -------------------------------------------------------------------------------------------------- demap
#ifndef DABX_DEMAP_Q                 // experiment builds (tools/build_variant.sh -DDABX_DEMAP_Q=4 -DDABX_DEMAP_OCC=3): carriers per thread
#define DABX_DEMAP_Q 2
#define DABX_DEMAP_OCC 6
#endif
constexpr int DEMAP_Q = DABX_DEMAP_Q, DEMAP_THREADS = K / DEMAP_Q, DEMAP_NP = DEMAP_Q / 2;   // carriers per thread (in pairs); 12 waves per stream
constexpr int DEMAP_NOUT = (K2 / 4) / DEMAP_THREADS;              // output dwords per thread and symbol
constexpr int TILE_PLANE = 196;                                   // LDS bytes per plane of the output tile (192 used)
// cache hints of the demapper's streams (pipeline.h, DABX_DEMAP_NT): 1 = the spectra, read once, past the caches; 2 = the ring stores
__device__ __forceinline__ float2 demap_ld_spec(const float2 *p)
{
#if DABX_DEMAP_NT & 1
  typedef float dm_f2 __attribute__((ext_vector_type(2)));
  const dm_f2 q = __builtin_nontemporal_load(reinterpret_cast<const dm_f2 *>(p));
  return make_float2(q.x, q.y);
#else
  return *p;
#endif
}
__device__ __forceinline__ void demap_st_ring(uint32_t *p, uint32_t v)
{
#if DABX_DEMAP_NT & 2
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
// ESoftBitType 1..3 and the symbol conversion (SAT: the SIMD builds' saturating one, cfg.viterbi_tie_mode != 0) as compile-time
// constants: no per-carrier branches on either
// MER: the LCD record's phase-deviation IIR is advanced too (dabx_set_lcd_statistics) -- its own instances: even as a block-uniform branch it
// cost k_demap_frame6 48 bytes of scratch at six waves per SIMD
template <int SOFT_TYPE, bool SAT, bool WHOLE = false, bool MER = false>
__device__ __forceinline__ void demap_frame_body(EngineDev &e, const DevTables &t, const int l0, const int l1)
{
  // symbols [l0, l1) of the frame (0-based: l = symbol index - 1).  The engine runs [0, 3) -- the FIC symbols -- first so
  // that k_fic_frame can start on its own HIP stream while [3, 75) is demapped; the per-carrier state passes through
  // HBM between the two launches exactly as it does from frame to frame.
  // Wave priority: the FIC symbols are on the frame's feedback chain (FIC decoder -> frame tail -> next head); the MSC symbols
  // are not once they are demapped on their own HIP stream, and then they yield to the FIC decoder running next to them but
  // stay above the batched MSC decoder (+1.7 % same-box against priority 3, profiles/r02_ab/ab11.json).
  if (l0 == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
  __shared__ __attribute__((aligned(16))) float red[32];
  __shared__ __attribute__((aligned(16))) uint8_t tile[2][16 * TILE_PLANE];
  __shared__ v2f sd_lds[MER ? DEMAP_NP * DEMAP_THREADS : 1];        // mStdDevSqPhaseVector of the thread's carriers (LCD statistics on: ofdm_core.h, demap_pair)
  const int s = blockIdx.x + e.s0, tid = threadIdx.x;
  StreamCtl &c = e.ctl[s];
  if (l0 == 0 && e.flag_sync) seq_wait(e.sym_seq + s, e.step_seq, e.seq_timeouts);     // few streams: k_symbols of this step (HIP stream a) is through
  // the first launch of a frame (l0 == 0) reads the stream's scalars and leaves a snapshot; a later launch of the same frame
  // uses the snapshot only (the frame tail / next head may already have moved the originals on)
  FrameSnap fs;
  if (l0 == 0) {
    fs.cif0 = c.cif_no; fs.clock_err = c.clock_err; fs.frame_ok = c.frame_ok; fs.np_sel = c.np_sel; fs.pad_ = 0;
    if (tid == 0) {
      e.fsnap[s] = fs;
      if (fs.frame_ok && e.demap_busy) __hip_atomic_store(&e.demap_busy[s], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else fs = e.fsnap[s];
  if (!fs.frame_ok) {
    if (l0 == 0 && e.flag_sync) seq_publish(e.fic_seq + s, e.step_seq);
    return;
  }
  DemapDev &d = e.demap;
  const float *null_power = fs.np_sel ? d.null_power2 : d.null_power;
  const float2 *spectra = e.spectra + (size_t)e.parity * e.n_streams * 75 * K;
  DemapPair cr[DEMAP_NP];                                  // the thread's carriers in pairs, component-wise (demap_pair)
  v2f rel_f[DEMAP_NP], wk[DEMAP_NP], pacc[DEMAP_NP];
#pragma unroll
  for (int q = 0; q < DEMAP_Q; q++) {
    const int k = tid + DEMAP_THREADS * q, p = q >> 1, h = q & 1;
    const int bin = t.perm_bin[k];
    rel_f[p][h] = (float)(K / 2 - t.perm_rel[k]);
    // X_(l-1): the phase reference (symbol 0, FFT bin order) or the previous symbol's spectrum (carrier order)
    const float2 pr = l0 == 0 ? d.phase_ref[(size_t)s * TU + bin] : spectra[((size_t)s * 75 + (l0 - 1)) * K + k];
    cr[p].prev_re[h] = pr.x; cr[p].prev_im[h] = pr.y;
    cr[p].integ[h] = d.integ[(size_t)s * K + k];
    cr[p].mean_power[h] = d.mean_power[(size_t)s * K + k];
    cr[p].mean_sigma_sq[h] = d.mean_sigma[(size_t)s * K + k];
    cr[p].null_power[h] = null_power[(size_t)s * TU + bin];
    wk[p][h] = mpa_weight(k);
    pacc[p][h] = 0.0f;
    if constexpr (MER) sd_lds[p * DEMAP_THREADS + tid][h] = d.std_dev[(size_t)s * K + k];
  }
  float mean_value = d.mean_value[s], mpa = d.mean_power_all[s];
  const float ce = fs.clock_err;                          // mClockErrHz of the previous frame, dab_processor.cpp:342
  // block-uniform (every thread loaded the same word): in SGPRs, so that the CIF slot / block offset of the ring address below
  // is scalar arithmetic and the store address is one 64-bit add per symbol
  const long long cif0 = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)fs.cif0 >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)fs.cif0));
  uint8_t *fic = e.fic_sym + (size_t)s * 3 * K2;
  uint8_t *tdi = e.tdi + (size_t)s * TDI_SLOTS * CIF_BITS;
  int16_t *cap = e.capture_soft ? e.soft_cap + (size_t)s * 75 * K2 : nullptr;
  float2 xn[DEMAP_Q];                                           // spectrum values of the next symbol (gather latency off the chain)
  // one pointer into the middle of the thread's carriers, stepped by a symbol per iteration: the DEMAP_Q loads of a symbol are
  // immediate offsets of +- (DEMAP_Q / 2 - 1/2) * DEMAP_THREADS elements from it, no 64-bit address arithmetic in the loop
  const float2 *xp = spectra + ((size_t)s * 75 + l0) * K + tid + (DEMAP_Q / 2) * DEMAP_THREADS;
#pragma unroll
  for (int q = 0; q < DEMAP_Q; q++) xn[q] = demap_ld_spec(xp + (q - DEMAP_Q / 2) * DEMAP_THREADS);
  // The 3072 Viterbi symbols of an OFDM symbol leave through LDS: every thread drops its bytes into a tile that is
  // already laid out like the planar ring (plane = i & 15, 192 positions per plane and symbol), and after the barrier of
  // the mean-value reduction the block stores aligned dwords -- one per thread with 768 threads, 48 consecutive dwords per
  // plane -- instead of scattered byte stores with their address arithmetic.  Two tiles: a fast thread may already fill the
  // next symbol's tile while a slow one still drains this one.  A plane takes TILE_PLANE = 196 bytes of LDS (49 dwords,
  // odd): the 16 planes a wave's byte stores touch then fall on 16 different banks (with 192 B = 48 dwords they fell on two,
  // an 8-way conflict on each of the four stores per thread and symbol; tools/lds_conflicts.py).
  static_assert(DEMAP_Q % 2 == 0 && K % DEMAP_Q == 0 && (K2 / 4) % DEMAP_THREADS == 0 && DEMAP_THREADS % 64 == 0, "tile <-> thread mapping below");
  int tpos[2 * DEMAP_Q];                                   // tile byte offsets of (re, im) of this thread's carriers
#pragma unroll
  for (int q = 0; q < DEMAP_Q; q++) {
    const int k = tid + DEMAP_THREADS * q;
    tpos[2 * q] = (k & 15) * TILE_PLANE + (k >> 4);
    tpos[2 * q + 1] = ((K + k) & 15) * TILE_PLANE + ((K + k) >> 4);
  }
  // one symbol; PAR = (l - l0) & 1 selects the tile and the partial-sum buffer at compile time (LDS immediate offsets)
  auto symbol = [&](const int l, auto par) {
    constexpr int PAR = decltype(par)::value;
    if (l < 74) xp += K;                                   // the symbol after the last one: fetched again, never used
    const int m = l - 3, cif = m / 18, blk = m % 18;       // msc_handler.cpp:148-168 : 18 symbols per CIF
    const float w2 = demap_w2(mean_value, SOFT_TYPE);
    uint8_t *tl = tile[PAR];
    float2 xc[DEMAP_Q];
#pragma unroll
    for (int q = 0; q < DEMAP_Q; q++) { xc[q] = xn[q]; xn[q] = demap_ld_spec(xp + (q - DEMAP_Q / 2) * DEMAP_THREADS); }
    float part = 0.f;
#pragma unroll
    for (int p = 0; p < DEMAP_NP; p++) {
      int16_t sr[2], si[2];
      v2f pw;
      const v2f mag = demap_pair<SOFT_TYPE>(cr[p], (v2f){xc[2 * p].x, xc[2 * p + 1].x}, (v2f){xc[2 * p].y, xc[2 * p + 1].y}, rel_f[p], ce, w2, sr, si, pw,
                                            MER ? &sd_lds[p * DEMAP_THREADS + tid] : nullptr);
      part = p == 0 ? mag.x + mag.y : part + (mag.x + mag.y);
      pacc[p] = pacc[p] * mpa_decay() + pw;                 // per carrier: sum_l d^(74-l) p_l, reduced once per frame below
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int q = 2 * p + h;
        tl[tpos[2 * q]] = SAT ? soft_to_sym_sat(sr[h]) : soft_to_sym(sr[h]);
        tl[tpos[2 * q + 1]] = SAT ? soft_to_sym_sat(si[h]) : soft_to_sym(si[h]);
        if (cap) { const int k = tid + DEMAP_THREADS * q; cap[(size_t)l * K2 + k] = sr[h]; cap[(size_t)l * K2 + K + k] = si[h]; }
      }
    }
    // mMeanValue (ofdm_decoder.cpp:256,294): block sum -- per-wave butterflies (wave_sum), then the wave partials added by every
    // thread, pairwise in packed form.  ONE barrier per symbol: the partials (and the tile) are double-buffered by
    // symbol parity, so what a thread still reads of symbol l cannot be overwritten before the barrier of symbol l + 1.
    {
      float *rp = red + 16 * PAR;
      const float pw_sum = wave_sum(part);
      if ((tid & 63) == 0) rp[tid >> 6] = pw_sum;
      __syncthreads();                                      // the tile is complete behind it, too.  (The barrier is not what the kernel waits
                                                            // for: a timing build without it is no faster, profiles/r03_ab/ab12.)
      float sum = 0.f;
      if constexpr (DEMAP_THREADS / 64 == 12) {               // twelve wave partials as six packed pairs: 5 v_pk_add_f32 + 1 add
        const v2f *rp2 = reinterpret_cast<const v2f *>(rp);
        const v2f a = ((rp2[0] + rp2[1]) + (rp2[2] + rp2[3])) + (rp2[4] + rp2[5]);
        sum = a.x + a.y;
      } else {
#pragma unroll
        for (int w = 0; w < DEMAP_THREADS / 64; w++) sum += rp[w];
      }
      mean_value = sum * (1.0f / (float)K);
    }
#pragma unroll
    for (int j = 0; j < DEMAP_NOUT; j++) {
      const int dwi = tid + DEMAP_THREADS * j;              // output dword of the symbol
      if (l < 3) {                                          // symbols 1..3 -> FIC, linear: bytes 4 dwi .. 4 dwi + 3
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) { const int i = 4 * dwi + b; v |= (uint32_t)tl[(i & 15) * TILE_PLANE + (i >> 4)] << (8 * b); }
        reinterpret_cast<uint32_t *>(fic + l * K2)[dwi] = v;
      } else {                                              // MSC -> planar time-de-interleaver ring
        const int out_plane = dwi / 48, out_dw = dwi - out_plane * 48;
        const uint32_t v = *reinterpret_cast<const uint32_t *>(tl + out_plane * TILE_PLANE + 4 * out_dw);
        // tdi_off(cif0 + cif, blk * K2 + out_plane) + 4 out_dw, split into the thread's constant part (plane, dword) and the
        // block-uniform part (CIF slot, position of the symbol's 192 bytes within the plane): one 64-bit add per store
        static_assert(K2 % 16 == 0, "a symbol is a whole number of positions in every plane");
        uint8_t *const tdi_thr = tdi + (size_t)out_plane * (CIF_BITS / 16) + 4 * out_dw;
        const size_t uoff = (size_t)((cif0 + cif) & (TDI_SLOTS - 1)) * CIF_BITS + (size_t)blk * (K2 / 16);
        demap_st_ring(reinterpret_cast<uint32_t *>(tdi_thr + uoff), v);
      }
    }
  };
  // the demapper state advances on all 75 symbols in every mode
  auto run = [&](int a, int b) {
    int l = a;
    for (; l + 1 < b; l += 2) { symbol(l, std::integral_constant<int, 0>{}); symbol(l + 1, std::integral_constant<int, 1>{}); }
    if (l < b) symbol(l, std::integral_constant<int, 0>{});
  };
  constexpr bool whole_frame_flagged = WHOLE;
  if constexpr (WHOLE) {
    // few streams (k_demap_whole): ONE launch for the frame's 75 symbols -- no kernel boundary on the demapper's loop; the FIC decoder (HIP
    // stream a) is told as soon as symbols 1..3 are out.  (The publish holds a block barrier: the tile / partial-sum parity may start over
    // behind it.)  Its own kernel: the second copy of the symbol loop costs k_demap_frame6 its six waves per SIMD (56-84 bytes of scratch).
    run(0, 3);
    seq_publish(e.fic_seq + s, e.step_seq);
    run(3, l1);
  } else run(l0, l1);
#pragma unroll
  for (int q = 0; q < DEMAP_Q; q++) {
    const int k = tid + DEMAP_THREADS * q, p = q >> 1, h = q & 1;
    d.integ[(size_t)s * K + k] = cr[p].integ[h];
    d.mean_power[(size_t)s * K + k] = cr[p].mean_power[h];
    d.mean_sigma[(size_t)s * K + k] = cr[p].mean_sigma_sq[h];
  }
  if constexpr (MER) {
    float sd = 0.f;
#pragma unroll
    for (int q = 0; q < DEMAP_Q; q++) {
      const float v = sd_lds[(q >> 1) * DEMAP_THREADS + tid][q & 1];
      d.std_dev[(size_t)s * K + tid + DEMAP_THREADS * q] = v;
      sd += v;
    }
    if (l1 == 75) {                                          // :331-340 after the frame's last symbol, like the SNR below
      sd = block_sum(sd, red, tid);
      if (tid == 0) c.mer_db = mer_db_from(sd);
      __syncthreads();                                       // red is used again below
    }
  }
  // SNR estimate as the LCD statistics compute it (ofdm_decoder.cpp:326-343) after the last symbol of the frame
  // mMeanPowerOvrAll (ofdm_decoder.cpp:214) over the 75 symbols in closed form: x d^75 + sum_k w_k sum_l d^(74-l) p_(k,l)
  float ns = cr[0].null_power.x + cr[0].null_power.y, wsum = wk[0].x * pacc[0].x + wk[0].y * pacc[0].y;
#pragma unroll
  for (int p = 1; p < DEMAP_NP; p++) { ns += cr[p].null_power.x + cr[p].null_power.y; wsum += wk[p].x * pacc[p].x + wk[p].y * pacc[p].y; }
  block_sum2w(ns, wsum, red, tid);
  if (tid == 0) {
    mpa = mpa * mpa_decay_n(l1 - l0) + wsum;
    d.mean_value[s] = mean_value; d.mean_power_all[s] = mpa;
    if (l1 == 75) c.snr_db = snr_db_from(mpa, ns);
  }
  if (l0 == 0 && e.flag_sync && !whole_frame_flagged) seq_publish(e.fic_seq + s, e.step_seq);  // the FIC symbols (and the snapshot) are out: k_fic_frame may go
  if (l1 == 75 && e.demap_busy) {
    // the frame's last demapper launch is done with the stream's state: every thread's stores are ordered before the flag
    // (block barrier, then a device-scope release by the thread that clears it)
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&e.demap_busy[s], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Built for six waves per SIMD (<= 80 VGPRs, 7 values spilled outside the loop): two 12-wave blocks share a CU instead of
// taking turns (at 94 VGPRs only one fitted: 0.32 -> 0.24 ms per step).
template <int SOFT_TYPE, bool SAT, bool MER>
__global__ __launch_bounds__(DEMAP_THREADS, DABX_DEMAP_OCC) void k_demap_frame6(EngineDev e, DevTables t, int l0, int l1) { demap_frame_body<SOFT_TYPE, SAT, false, MER>(e, t, l0, l1); }
// the FIC symbols alone (first launch of a frame in the overlapped schedule): its own kernel symbol so that rocprofv3's
// per-kernel statistics keep the 3-symbol and the 72-symbol launches apart, as bench.py's event pairs do
template <int SOFT_TYPE, bool SAT, bool MER>
__global__ __launch_bounds__(DEMAP_THREADS) void k_demap_fic(EngineDev e, DevTables t) { demap_frame_body<SOFT_TYPE, SAT, false, MER>(e, t, 0, 3); }
// few streams with device-side hand-overs (EngineDev::flag_sync): the whole frame in one launch
template <int SOFT_TYPE, bool SAT, bool MER>
__global__ __launch_bounds__(DEMAP_THREADS) void k_demap_whole(EngineDev e, DevTables t) { demap_frame_body<SOFT_TYPE, SAT, true, MER>(e, t, 0, 75); }

// ---------------------------------------------------------------------------------------------------- FIC
struct SrcFic {                       // 2304 Viterbi symbols of one FIC + depuncture map (viterbi_core.h: key / raw / syms)
  const uint8_t *sym;
  const uint16_t *map;
  typedef ushort4 Key;
  struct Raw { uint8_t a, b, c, d; };
  __device__ Key key(int t) const { return *reinterpret_cast<const ushort4 *>(map + 4 * t); }
  __device__ uint8_t ld(uint16_t idx) const { return sym[idx == PUNCT ? 0 : idx]; }
  __device__ Raw raw(Key m) const { return {ld(m.x), ld(m.y), ld(m.z), ld(m.w)}; }
  __device__ static int cv(uint8_t v, uint16_t idx) { return vit_sym_from_u8(idx == PUNCT ? (uint8_t)127 : v); }
  __device__ VitSyms syms(Raw r, Key m) const { return {cv(r.a, m.x), cv(r.b, m.y), cv(r.c, m.z), cv(r.d, m.w)}; }
};

// FIC blocks [first, first + count) of the frame: the engine decodes all four at once (first = 0, count = 4); the
// per-symbol stage entry dabx_fic_process_block decodes each block as soon as its 2304 soft bits are complete, like
// FicDecoder::process_block does (fic_decoder.cpp:155-165): block 0 with OFDM symbol 1, block 1 with symbol 2, blocks 2
// and 3 with symbol 3.
__global__ __launch_bounds__(256) void k_fic_frame(EngineDev e, DevTables t, int first, int count)
{
  front_prio();
  __shared__ __attribute__((aligned(16))) char wtab[4][VIT_BLK * 16];
  __shared__ uint32_t fibw[4][24];          // 4 x 768 decoded + de-dispersed bits, packed
  __shared__ uint32_t raw[4][32];           // chain-back output, 30 bits per word (26 words + padding)
  __shared__ uint8_t crc_ok[12];
  __shared__ uint32_t encw[4][26];          // the decoded bits before de-dispersal (+ zero tail), for the BER re-encoder
  __shared__ int ber_err[4];
  __shared__ uint16_t s_crc[256];           // CCITT table: the 30-step look-up chain of a FIB's CRC stays in LDS
  const int s = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  StreamCtl &c = e.ctl[s];
  // few streams: the FIC symbols come from HIP stream d.  Waited for in EVERY case -- also when this step has no frame for the stream: the
  // demapper's first launch reads the stream's scalars, which the kernels behind this one (tail, next head) rewrite
  if (e.flag_sync && first == 0 && count == 4) seq_wait(e.fic_seq + s, e.step_seq, e.seq_timeouts);
  if (!c.frame_ok) return;
  s_crc[threadIdx.x] = t.crc_ccitt[threadIdx.x];
  const int fic = first + wave;             // this wave's FIC block
  if (wave < count) {
    SrcFic src{e.fic_sym + (size_t)s * 3 * K2 + fic * FIC_IN, t.fic_map};
    uint32_t *dec = e.vit_scratch + ((size_t)s * 4 + fic) * (size_t)e.vit_stride;
    const VitLaneConst k = vit_lane_const(lane);
    if (e.tie_mode == 2) vit_forward<2>(src, FIC_OUT + 6, wtab[wave], dec, lane, k);
    else if (e.tie_mode) vit_forward<1>(src, FIC_OUT + 6, wtab[wave], dec, lane, k);
    else vit_forward<0>(src, FIC_OUT + 6, wtab[wave], dec, lane, k);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    vit_traceback(dec, FIC_OUT, lane, raw[wave]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t dw = 0;
    if (lane < 24) { dw = vit_output_word(raw[wave], lane); fibw[fic][lane] = dw ^ t.prbs_words[lane]; }   // fic_decoder.cpp:219-222
    // ViterbiSpiral::calculate_BER (viterbi_spiral.cpp:128-164, called fic_decoder.cpp:199 on the bits BEFORE the PRBS): the 768 + 6
    // decoded bits re-encoded (polys 109, 79, 83, 109) and compared, at the 2304 transmitted positions, with the sign of the received
    // soft bit.  Lane l re-encodes steps 13 l .. 13 l + 12; the hard decision is (Viterbi symbol > 127) == (soft > 0): exact for every
    // int16 soft bit in the SIMD builds' conversion, and in the canonical one for all but soft >= 32641 (where the reference's own
    // `soft + 127` wraps, viterbi_scalar.h:34-40).
    if (lane < 24) encw[wave][lane] = dw;
    if (lane >= 24 && lane < 26) encw[wave][lane] = 0;      // the six tail steps shift in zeros
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const uint8_t *eb = reinterpret_cast<const uint8_t *>(encw[wave]);
      auto dbit = [&](int i) -> unsigned { return i < 0 ? 0u : (unsigned)((eb[i >> 3] >> (7 - (i & 7))) & 1); };
      const int t0 = 13 * lane;
      unsigned sr = 0;
#pragma unroll
      for (int q = 6; q >= 1; q--) sr = (sr << 1) | dbit(t0 - q);
      int err = 0;
      const uint8_t *sym = e.fic_sym + (size_t)s * 3 * K2 + fic * FIC_IN;
      for (int i = t0; i < t0 + 13 && i < FIC_OUT + 6; i++) {
        sr = ((sr << 1) | dbit(i)) & 0xFFu;
        const ushort4 m = *reinterpret_cast<const ushort4 *>(t.fic_map + 4 * i);
        const unsigned p0 = __builtin_popcount(sr & 109u) & 1u, p1 = __builtin_popcount(sr & 79u) & 1u, p2 = __builtin_popcount(sr & 83u) & 1u;
        if (m.x != PUNCT) err += (unsigned)(sym[m.x] > 127) != p0;
        if (m.y != PUNCT) err += (unsigned)(sym[m.y] > 127) != p1;
        if (m.z != PUNCT) err += (unsigned)(sym[m.z] > 127) != p2;
        if (m.w != PUNCT) err += (unsigned)(sym[m.w] > 127) != p0;
      }
      err = wave_sum_int(err);
      if (lane == 0) ber_err[fic] = err;
    }
  }
  __syncthreads();
  const int slot = (int)(c.frames % e.out_frames);
  uint8_t *fo = e.fib_out + ((size_t)s * e.out_frames + slot) * 12 * 32;
  const int fib0 = 3 * first, nfib = 3 * count;
  __shared__ int fib_cif[12];                 // CIF counter a FIB's FIG 0/0 carries (-1: none, or the FIB failed its CRC)
  if ((int)threadIdx.x < nfib) {              // one lane per FIB: CRC (crc.cpp:98-132 == CCITT over 30 bytes vs the last 2) and its FIG walk
    const int fibi = fib0 + threadIdx.x;
    const uint8_t *b = reinterpret_cast<const uint8_t *>(&fibw[fibi / 3][0]) + (fibi % 3) * 32;
    const uint8_t good = crc16_check_bytes(b, 30, s_crc);
    crc_ok[fibi] = good;
    e.fib_crc[((size_t)s * e.out_frames + slot) * 12 + fibi] = good;
    int cif = -1;
    if (good) {
      int p = 0;                            // fib_decoder.cpp:59-110
      while (p < 30) {
        const int type = b[p] >> 5, len = b[p] & 0x1F;
        if (type == 7 && len == 0x1F) break;
        if (type == 0 && p + 5 < 32 && (b[p + 1] & 0x1F) == 0)                       // FIG 0/0, fib_decoder_fig0.cpp:89-101
          cif = (b[p + 4] & 0x1F) * 250 + b[p + 5];
        p += len + 1;
      }
    }
    fib_cif[fibi] = cif;
  }
  for (int i = threadIdx.x; i < 24 * count; i += 256) reinterpret_cast<uint32_t *>(fo)[24 * first + i] = fibw[first + i / 24][i % 24];
  __syncthreads();
  if (threadIdx.x == 0) {
    // per-FIB bookkeeping in FIB order (fic_decoder.cpp:234-261): the CIF counter is that of the last good FIB that carries one
    int ratio = c.fic_ratio, cif_count = c.cif_count;
    long long ok = 0;
    for (int fibi = fib0; fibi < fib0 + nfib; fibi++) {
      if (crc_ok[fibi]) {
        ok++;
        if (fib_cif[fibi] >= 0) cif_count = fib_cif[fibi];
        if (ratio < 10) ratio++;
      } else if (ratio > 0) ratio--;
    }
    c.fic_ratio = ratio; c.cif_count = cif_count;
    c.fib_ok += ok; c.fib_total += nfib;
    // mFicBits / mFicErrors / mFicBlock, block by block (fic_decoder.cpp:199-210): every block adds its 2304 transmitted bits
    int bits = c.fic_bits, errs = c.fic_errors, blk = c.fic_block;
    for (int f = first; f < first + count; f++) {
      bits += FIC_IN; errs += ber_err[f];
      if (++blk == 40) { c.fic_status_errors = errs; c.fic_status_bits = bits; blk = 0; errs /= 2; bits /= 2; }
    }
    c.fic_bits = bits; c.fic_errors = errs; c.fic_block = blk;
  }
}

// --------------------------------------------------------------------------------------------- frame tail
__global__ __launch_bounds__(256) void k_frame_tail(EngineDev e, DevTables t)
{
  front_prio();
  __shared__ float2 lds[FFT_LDS_FLOAT2];
  __shared__ float red[8];
  __shared__ float s_fbb;
  const int s = blockIdx.x, tid = threadIdx.x;
  StreamCtl &c = e.ctl[s];
  if (!c.frame_ok) return;
  // fine CFO from the 75 cyclic-prefix correlations (dab_processor.cpp:366, 236-242): their loads go first (vmcnt counts in order)
  const float LNQ = -1.00000500003333e-5f;                 // ln(1 - 1e-5): decay of the level tracker per sample
  float2 cpp = make_float2(0.f, 0.f);
  float absp = 0.f;
  if (tid < 75) { cpp = e.cp_part[(size_t)s * 75 + tid]; absp = e.abs_part[(size_t)s * 76 + tid]; }
  // The null symbol's samples do not depend on the fine-CFO update below, only their mixing does: requested here, behind the
  // two loads above, so that their HBM latency runs behind the reductions (they used to be the first thing after the barrier).
  const float2 *ring = e.iq + (size_t)s * e.ring_len;
  const unsigned long long base = c.sym0_pos + TU + 75ull * TS;
  const RingView rv(ring, e.ring_len, base);
  float2 xr[8];
#pragma unroll
  for (int u = 0; u < 8; u++) xr[u] = rv.at(TG + tid + 256 * u);
  constexpr int N_REST = (TN - TU + 255) / 256;            // the rest of the T_n samples read (guard interval and tail of the null symbol)
  float2 xq[N_REST];
#pragma unroll
  for (int k = 0; k < N_REST; k++) {                       // unconditional (index clamped; the sum below skips the surplus): no branch, no wait between the requests
    const int i = min(tid + 256 * k, TN - TU - 1);
    xq[k] = rv.at(i < TG ? i : i + TU);
  }
  asm volatile("" ::: "memory");                           // keep the order of the requests
  float cre = cpp.x, cim = cpp.y, sym_w = 0.f;
  // level tracker (see the end of this kernel): chunk mean of symbol tid + 1, weighted by the decay over the symbols after it
  if (tid < 75) sym_w = absp * (1.0f / (float)TS) * __expf((float)((74 - tid) * TS) * LNQ);
  block_sum2(cre, cim, red, tid);
  sym_w = block_sum(sym_w, red, tid);
  const int f = c.f_frame;
  if (tid == 0) {
    // the reference correlates NCO-mixed samples: x'[i] conj(x'[i-Tu]) = x[i] conj(x[i-Tu]) e^{-j 2 pi f Tu / fs}
    double sr, cr;
    sincospi(-2.0 * (double)(((long long)f * TU) % INPUT_RATE) / (double)INPUT_RATE, &sr, &cr);
    const float rr = cre * (float)cr - cim * (float)sr, ri = cre * (float)sr + cim * (float)cr;
    float ph = atan2f(ri, rr);
    const float lim = 20.0f * 0.01745329251994329577f;
    if (ph > lim) ph = lim; else if (ph < -lim) ph = -lim;
    c.phase_offs = ph;
    c.f_sync += ph / 6.28318530717958647692f * 1000.0f;
    c.f_bb = c.f_sync;
    s_fbb = c.f_bb;
  }
  __syncthreads();
  // null symbol (dab_processor.cpp:267-302): T_n samples with the updated frequency, FFT of [Tg, Tg+Tu)
  const int phase_null = nco_advance(c.phase_sym1, f, 75LL * TS);
  const int f2 = (int)roundf(s_fbb);
  float2 v[8];
  Nco nco;
  nco.init(phase_null, f2, TG + tid);
  float an = 0.f;                                          // sum of this thread's |x|: transform samples first, then the rest
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const float2 x = xr[u];
    an += cabsf_level(x);
    v[u] = nco.mix(x);
    nco.step();
  }
#pragma unroll
  for (int k = 0; k < N_REST; k++) if (tid + 256 * k < TN - TU) an += cabsf_level(xq[k]);
  an = block_sum(an, red, tid);
  fft2048<false>(v, lds, t.twiddle, tid);
  const bool is_tii = (c.cif_count & 7) >= 4;              // :274
  if (is_tii && e.tii_acc) {                               // add_to_tii_buffer, dab_processor.cpp:287
#pragma unroll
    for (int u = 0; u < 8; u++) {
      float2 *a = &e.tii_acc[(size_t)s * TU + tid + 256 * u];
      *a = make_float2(a->x + v[u].x, a->y + v[u].y);
    }
    if (tid == 0) e.tii_cnt[2 * s]++;
  }
  // store_null_symbol_without_tii (ofdm_decoder.cpp:114-130).  The updated noise power goes into the buffer the demapper of
  // THIS frame does not read (its MSC symbols may still be in flight on another HIP stream); a TII frame carries the values
  // over unchanged.  np_sel is flipped below.
  {
    const float kMinNoisePower = (1.0f / 32767.0f) * (1.0f / 32767.0f);
    const float *np_cur = c.np_sel ? e.demap.null_power2 : e.demap.null_power;
    float *np_new = c.np_sel ? e.demap.null_power : e.demap.null_power2;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int bin = tid + 256 * u;
      if ((bin >= 1 && bin <= K / 2) || bin >= TU - K / 2) {
        float np = np_cur[(size_t)s * TU + bin];
        if (!is_tii) {
          const float power = v[u].x * v[u].x + v[u].y * v[u].y + kMinNoisePower;
          np += 0.05f * (power - np);
        }
        np_new[(size_t)s * TU + bin] = np;
      }
    }
  }
  if (tid == 0) {
    const int sample_count = c.sample_count + 75 * TS + TN;
    if (c.correction == 0) {                               // :246-251
      float ce = (float)INPUT_RATE * ((float)sample_count / (float)TF - 1.0f);
      if (ce > 307.2f) ce = 307.2f; else if (ce < -307.2f) ce = -307.2f;
      c.clock_err += 0.1f * (ce - c.clock_err);
    }
    // Level IIR of SampleReader (sample_reader.cpp:246-248: s += 1e-5 (|x| - s) for every sample read) over the samples of
    // this frame, chunk by chunk in the order they were read: the T_u correlation window, the start_index samples after
    // it, symbols 1..75, the null symbol.  Within a chunk the samples are weighted equally (chunk mean); across chunks the
    // decay q^n is exact -- the null symbol, read last, keeps its full weight.  Only the out-of-lock dip detector reads it.
    {
      float lv = c.s_level;
      auto upd = [&lv](float mean, float qn) { lv += (1.0f - qn) * (mean - lv); };
      const int start = c.start_index;
      upd(c.head_abs_a * (1.0f / (float)TU), __expf((float)TU * LNQ));
      if (start > 0) upd(c.head_abs_b / (float)start, __expf((float)start * LNQ));
      // symbols 1..75: lv <- lv q^75 + (1 - q) sum_l q^(74-l) mean_l, the weighted sum taken in parallel above
      lv = lv * __expf((float)(75 * TS) * LNQ) + (1.0f - __expf((float)TS * LNQ)) * sym_w;
      upd(an * (1.0f / (float)TN), __expf((float)TN * LNQ));
      if (!e.exact_level) c.s_level = lv;                  // cfg.exact_level_tracker = 1: k_level_exact walks the frame's samples instead
    }
    if (e.frame_pos) {                                      // per-frame record next to the FIBs (dabx_read_frame_info)
      const size_t slot = (size_t)s * e.out_frames + (size_t)(c.frames % e.out_frames);
      e.frame_pos[slot] = (long long)c.sym0_pos; e.frame_start[slot] = c.start_index;
    }
    c.sample_count = sample_count;
    c.rd = base + TN;
    c.lvl_approx_pos = c.rd;                               // (anchor_level: s_level is the chunk-wise value up to here; the anchor stays where the search left it)
    { const int h = (int)(c.frames % LVL_HIST); c.lvl_hist_pos[h] = c.rd; c.lvl_hist_S[h] = c.s_level; }   // ... and kept per frame for level_from_anchor's second resort
    c.nco_phase = nco_advance(phase_null, f2, TN);
    c.cif_no += 4;
    c.frames += 1;
    c.np_sel ^= 1;
    c.sync_thr = 2.0f * e.threshold;                       // :178
    c.state = ST_EVAL_SYNC;
  }
}

// ------------------------------------------------------------------------------------- exact level tracker
// cfg.exact_level_tracker = 1: SampleReader's level IIR (sample_reader.cpp:245-248: a = |x|; peak = max(peak, a);
// sLevel += 0.00001f * (a - sLevel) for EVERY sample get_samples hands out) over the samples of the frame just demodulated,
// one by one in the order they were read -- the correlation window, the start_index samples behind it, symbols 1..75 and the
// null symbol are one contiguous run [sym0_pos - start_index, rd) of the ring.  A dependent chain of three float operations
// per sample, 196 104 + start_index samples per frame: one wave per stream computes |x| for 1024 samples at a time into LDS
// (all lanes) and then walks them (every lane redundantly: LDS broadcast reads, no divergence).  About 1.5 ms per frame,
// more than the rest of the receiver together -- which is why the default advances the tracker chunk-wise (k_frame_tail).
__global__ __launch_bounds__(128) void k_level_exact(EngineDev e)
{
  front_prio();                          // (wave priority 0 / 1 / 3 and the HIP stream's priority make no difference to what the tracker costs
                                         //  the frame chain: profiles/r04_ab/ab27_exact_level_priorities.txt)
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  StreamCtl &c = e.ctl[s];
  constexpr int CH = 1024;
  __shared__ __attribute__((aligned(16))) float chunk[2][CH + 32];   // (+ what acq_walk_S reads ahead)
  // Everything the receiver has read that the tracker has not seen: [level_pos, rd) -- in lock the frame chain moves rd (k_frame_tail;
  // k_frame_head by T_u on a failed correlation), out of lock k_acquire tracks the level itself and moves level_pos along.  The
  // kernel runs next to the frame chain (its own HIP stream, or in front of k_acquire in step): whatever value of rd it sees is a
  // point the receiver has reached; dabx_synchronize and every read-out run it once more behind the last frame.
  // Wave 0 walks chunk k out of LDS; wave 1 meanwhile turns the samples of chunk k + 1 -- requested a whole chunk
  // earlier, so that their HBM latency next to the frame chain's traffic (several microseconds) stays behind the walk -- into
  // magnitudes and requests chunk k + 2.  (With one wave doing both, every chunk began by waiting for its own loads: 5.5 ms per
  // frame next to the frame chain instead of 1.6.)
  const unsigned long long rd0 = e.level_pos[s];
  const unsigned long long rd1 = __hip_atomic_load(&c.rd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (rd1 <= rd0) return;
  const unsigned long long n = rd1 - rd0;
  const float2 *ring = e.iq + (size_t)s * e.ring_len;
  const unsigned len = (unsigned)e.ring_len;
  float2 raw[CH / 64];
  __shared__ float s_pk;
  auto request = [&](unsigned long long p0) {               // samples [p0, p0 + CH) into registers (zeros beyond n)
    const unsigned o0 = (unsigned)((rd0 + p0) % len);
#pragma unroll
    for (int q = 0; q < CH / 64; q++) {
      const unsigned i = lane + 64 * q;
      unsigned o = o0 + i; if (o >= len) o -= len;
      raw[q] = p0 + i < n ? ring[o] : make_float2(0.f, 0.f);
    }
  };
  float pk = 0.f;                                            // peakLevel is a maximum: taken by the wave that makes the magnitudes
  auto publish = [&](float *dst) {
#pragma unroll
    for (int q = 0; q < CH / 64; q++) {
      const float a = sqrtf(raw[q].x * raw[q].x + raw[q].y * raw[q].y);
      dst[lane + 64 * q] = a;
      pk = fmaxf(pk, a);
    }
  };
  float lv = c.s_level;
  LevelPar lp;
  if (wave == 0) lp.init(lane);
  if (wave == 1) { request(0); publish(chunk[0]); request(CH); }
  __syncthreads();
  unsigned b = 0;
  for (unsigned long long p0 = 0; p0 < n; p0 += CH, b ^= 1) {
    if (wave == 1) {
      publish(chunk[b ^ 1]);                               // chunk k + 1: requested one walk ago
      request(p0 + 2 * CH);
    } else {                                                 // every lane of the wave walks: see acquire_stream
      const unsigned m = n - p0 < CH ? (unsigned)(n - p0) : CH;
      const int n16 = __builtin_amdgcn_readfirstlane((int)(m >> 4));
      if (n16 > 0) lv = lp.block(chunk[b], n16, lv, nullptr, lane);
      for (unsigned i = 16u * (unsigned)n16; i < m; i++) lv += 0.00001f * (chunk[b][i] - lv);
    }
    __syncthreads();
  }
  if (wave == 1) {
    pk = __builtin_bit_cast(float, wave_butterfly_u32(__builtin_bit_cast(unsigned, pk), [](unsigned x, unsigned y) { return x > y ? x : y; }));
    if (lane == 0) s_pk = pk;
  }
  __syncthreads();
  if (tid == 0) { c.s_level = lv; c.peak_level = fmaxf(c.peak_level, s_pk); e.level_pos[s] = rd1; }
}
int launch_level_exact(const EngineDev &e, hipStream_t st)
{
  hipLaunchKernelGGL(k_level_exact, dim3(e.n_streams), dim3(128), 0, st, e);
  DABX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------- MSC
struct SrcMsc {                        // time de-interleaver read + depuncture (backend.cpp:131-139, protection.cpp:46-59); key / raw / syms
  const uint8_t *tdi;                  // this stream's ring
  const uint16_t *map;
  long long r;                         // CIF being output
  int base;                            // cu_start * 64
  int base_prev, thr;                  // a sub-channel that moved: CIFs r - 16 + m with m < thr are read at its old address (msc_move_thr)
  typedef ushort4 Key;
  struct Raw { uint8_t a, b, c, d; };
  __device__ Key key(int t) const { return *reinterpret_cast<const ushort4 *>(map + 4 * t); }
  __device__ uint8_t ld(uint16_t idx_in) const
  {
    const int idx = idx_in == PUNCT ? 0 : idx_in;
    // out_r[i] = in_{r-16+map[i&15]}[i], map = {0,8,4,12,2,10,6,14,1,9,5,13,3,11,7,15} (bit reversal of 4 bits)
    const int i4 = idx & 15;
    const int m = ((i4 & 1) << 3) | ((i4 & 2) << 1) | ((i4 & 4) >> 1) | ((i4 & 8) >> 3);
    const long long q = r - 16 + m;
    return tdi[tdi_off(q, (m < thr ? base_prev : base) + idx)];
  }
  __device__ Raw raw(Key m) const { return {ld(m.x), ld(m.y), ld(m.z), ld(m.w)}; }
  __device__ static int cv(uint8_t v, uint16_t idx) { return vit_sym_from_u8(idx == PUNCT ? (uint8_t)127 : v); }
  __device__ VitSyms syms(Raw q, Key m) const { return {cv(q.a, m.x), cv(q.b, m.y), cv(q.c, m.z), cv(q.d, m.w)}; }
};

// fast_mask: bit c set = class c + 1 is decoded by the lane-per-trellis kernels in this batch (vit_t.hip)
__global__ __launch_bounds__(256, 8) void k_msc_frame(EngineDev e, DevTables t, int cifs, unsigned fast_mask)
{
  __shared__ __attribute__((aligned(16))) char wtab[4][VIT_BLK * 16];
  __shared__ uint32_t raw[4][VIT_RAW_WORDS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int job = blockIdx.x * 4 + wave;
  const MscJob q = msc_job(e, job, cifs);
  if (!q.valid) return;
  const int s = q.s, j = q.j;
  const SubchDev &sc = e.subch[(size_t)s * e.max_subch + j];
  if (sc.fast_class && ((fast_mask >> (sc.fast_class - 1)) & 1u)) return;
  const long long r = q.r, out_idx = q.out_idx;
  SrcMsc src{e.tdi + (size_t)s * TDI_SLOTS * CIF_BITS, sc.map, r, sc.cu_start * 64, sc.prev_cu_start * 64, msc_move_thr(sc, r)};
  uint32_t *dec = e.vit_scratch + ((size_t)e.n_streams * 4 + (size_t)job) * (size_t)e.vit_stride;
  const VitLaneConst k = vit_lane_const(lane);
  if (e.tie_mode == 2) vit_forward<2>(src, sc.nbits + 6, wtab[wave], dec, lane, k);
  else if (e.tie_mode) vit_forward<1>(src, sc.nbits + 6, wtab[wave], dec, lane, k);
  else vit_forward<0>(src, sc.nbits + 6, wtab[wave], dec, lane, k);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  uint32_t *out = reinterpret_cast<uint32_t *>(e.msc_out + (((size_t)s * e.max_subch + j) * MSC_SLOTS + (size_t)(out_idx % MSC_SLOTS)) * e.msc_stride);
  vit_traceback(dec, sc.nbits, lane, raw[wave]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int w = lane; w < sc.nbits / 32; w += 64) out[w] = vit_output_word(raw[wave], w) ^ t.prbs_words[w];   // backend.cpp:155-158
}

// ------------------------------------------------------------------------------------------------- DAB+
// One wave per (stream, sub-channel); walks the logical frames produced in this batch step.
// The 5-frame window is staged in LDS once per super frame.  RS code word j is the byte sequence
// window[j + k R], k < 120 (mp4processor.cpp:193-201) and its corrected data bytes go back to the same
// positions of mOutVec (:225-228), so the super frame is simply window[0 .. 110 R) corrected in place.
// Syndromes of all R code words x 10 roots are evaluated lane-parallel (Horner over LDS); the full
// Berlekamp-Massey / Chien / Forney decoder runs only for code words whose syndromes are not all zero.
__global__ __launch_bounds__(64, 4) void k_dabplus(EngineDev e, DevTables t)   // <= 128 VGPRs: four waves per SIMD (it took 129)
{
  const int job = blockIdx.x, lane = threadIdx.x;
  const int s = job / e.max_subch, j = job % e.max_subch;
  SubchDev &sc = e.subch[(size_t)s * e.max_subch + j];
  if (!sc.active) return;
  const BatchSnap bs = e.snap[s];
  long long n_new = 0;                        // logical frames the decoder just produced for this sub-channel
  for (long long r = bs.msc_done; r < bs.cif_no; r++) if (r >= sc.start_cif + 16) n_new++;
  if (n_new == 0) return;
  const int R = sc.kbps / 8, nbytes = 3 * sc.kbps;         // nbytes = 24 R
  const uint8_t *ring = e.msc_out + ((size_t)s * e.max_subch + j) * MSC_SLOTS * e.msc_stride;
  long long cif_out = sc.cif_out;
  int blocks_in_buf = sc.blocks_in_buf, sf_sync = sc.sf_sync;
  long long sf_count = sc.sf_count, sf_ok = 0, sf_fail = 0, rs_corr = 0, rs_fail = 0, fc_corr = 0, au_ok = 0, au_bad = 0;
  __shared__ __attribute__((aligned(16))) uint8_t win[120 * 48 + 16];   // 5 logical frames (<= 384 kbit/s)
  __shared__ uint8_t gexp[512], glog[256];
  __shared__ uint16_t s_crc[256], s_fc[256];                // CCITT and fire-code CRC tables (serial look-up chains: keep them in LDS)
  __shared__ __attribute__((aligned(16))) uint16_t s_xpow[1024];   // x^(8 m) mod P, m <= 960: two look-ups per access unit sat behind an L2 round trip each
  __shared__ unsigned syn_or[48];                           // != 0: some syndrome of the code word is non-zero
  __shared__ __attribute__((aligned(4))) unsigned syn_w[48][3];   // the code word's ten syndromes (bytes 0..9 of the three words): the full decoder starts from them
  __shared__ uint8_t s_lam[48][12], s_deg[48], s_root[48][RS_NR];  // per dirty code word: locator (index form) + degree -> roots found by the wave-wide Chien search
  __shared__ int s_rootn[48];
  __shared__ uint8_t hdr0[12];
  __shared__ int s_flag;
  __shared__ int s_au[8];
  if (sc.dab_plus) {
    for (int i = lane; i < 512; i += 64) gexp[i] = t.gf_exp[i];
    for (int i = lane; i < 256; i += 64) { glog[i] = t.gf_log[i]; s_crc[i] = t.crc_ccitt[i]; s_fc[i] = t.fc_crctab[i]; }
    for (int i = lane; i < 128; i += 64) reinterpret_cast<uint4 *>(s_xpow)[i] = reinterpret_cast<const uint4 *>(t.crc_xpow)[i];
  }
  unsigned syn_ex[2][3];                      // (r (119 - k)) mod 255 for r = 0..9 as bytes, k = lane + 64 h: the syndrome sums' exponents
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int m = 119 - (lane + 64 * h);
    int ex = 0;
    syn_ex[h][0] = syn_ex[h][1] = syn_ex[h][2] = 0;
#pragma unroll
    for (int r = 0; r < 10; r++) {
      syn_ex[h][r >> 2] |= (unsigned)ex << (8 * (r & 3));
      ex += m;
      if (ex >= 255) ex -= 255;
    }
  }
  __syncthreads();
  for (long long n = 0; n < n_new; n++) {
    const long long newest = cif_out;        // index of the logical frame just added
    cif_out++;
    if (!sc.dab_plus) continue;
    blocks_in_buf++;                         // mp4processor.cpp:113
    if (blocks_in_buf < 5) continue;
    const long long oldest = newest - 4;
    if (sf_sync == 0) {                      // :132-142: fire code over the first 11 bytes of the oldest frame
      const uint8_t *f0 = ring + (size_t)(oldest % MSC_SLOTS) * e.msc_stride;
      const bool ok = firecode_syndrome([&](int i) { return f0[i]; }, s_fc) == 0;
      if (ok) sf_sync = 4; else { blocks_in_buf = 4; continue; }
    }
    blocks_in_buf = 0;                       // :147
    // ---- stage the window (coalesced 4-byte loads) and the GF tables
    __syncthreads();
    {                                         // the five frames' loads of a chunk in flight together (they were five memory latencies in a row)
      const uint32_t *src[5];
#pragma unroll
      for (int f = 0; f < 5; f++) src[f] = reinterpret_cast<const uint32_t *>(ring + (size_t)((oldest + f) % MSC_SLOTS) * e.msc_stride);
      for (int i = lane; i < nbytes / 4; i += 64) {
        uint32_t w[5];
#pragma unroll
        for (int f = 0; f < 5; f++) w[f] = src[f][i];
#pragma unroll
        for (int f = 0; f < 5; f++) reinterpret_cast<uint32_t *>(win + f * nbytes)[i] = w[f];
      }
    }
    if (lane < 12) hdr0[lane] = 0;
    __syncthreads();
    if (lane < 11) hdr0[lane] = win[lane];
    // ---- syndromes S_r(j) = XOR_k c_k alpha^(r (119 - k)), r < 10: the Horner recursion of reed_solomon.cpp:254-290 written
    //      as a sum, lanes over the byte index k (no 120-step dependent look-up chain); only "all ten are zero" is needed
    //      here, the full decoder below recomputes them for the code words that are not clean
    // one code word's partial sums of this lane: ten 8-bit sums packed into three words
    auto syn_partial = [&](int j, unsigned &a0, unsigned &a1, unsigned &a2) {
      a0 = a1 = a2 = 0;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int k = lane + 64 * h;
        const int b = k < 120 ? win[j + k * R] : 0;
        if (b) {
          const int lg = glog[b];
#pragma unroll
          for (int r = 0; r < 10; r++) {                      // exponents r (119 - k) mod 255: per-lane constants (syn_ex), no running sum on the look-up chain
            const unsigned v = gexp[lg + (int)((syn_ex[h][r >> 2] >> (8 * (r & 3))) & 0xFFu)];
            if (r < 4) a0 ^= v << (8 * r); else if (r < 8) a1 ^= v << (8 * (r - 4)); else a2 ^= v << (8 * (r - 8));
          }
        }
      }
    };
    int cw = 0;
    for (; cw + 1 < R; cw += 2) {                               // two code words at a time: their look-up and reduction chains interleave
      unsigned a0, a1, a2, b0, b1, b2;
      syn_partial(cw, a0, a1, a2);
      syn_partial(cw + 1, b0, b1, b2);
      a0 = wave_xor(a0); b0 = wave_xor(b0); a1 = wave_xor(a1); b1 = wave_xor(b1); a2 = wave_xor(a2); b2 = wave_xor(b2);
      if (lane == 0) {
        syn_or[cw] = a0 | a1 | a2; syn_or[cw + 1] = b0 | b1 | b2;
        syn_w[cw][0] = a0; syn_w[cw][1] = a1; syn_w[cw][2] = a2; syn_w[cw + 1][0] = b0; syn_w[cw + 1][1] = b1; syn_w[cw + 1][2] = b2;
      }
    }
    if (cw < R) {
      unsigned a0, a1, a2;
      syn_partial(cw, a0, a1, a2);
      a0 = wave_xor(a0); a1 = wave_xor(a1); a2 = wave_xor(a2);
      if (lane == 0) { syn_or[cw] = a0 | a1 | a2; syn_w[cw][0] = a0; syn_w[cw][1] = a1; syn_w[cw][2] = a2; }
    }
    __syncthreads();
    // ---- full decoder only where needed.  Berlekamp-Massey and Forney: one lane per dirty code word, from the syndromes summed above (byte r of
    //      syn_w[code word]: no second, 1200-step pass over the 120 bytes on one lane).  The Chien search over all 255 positions, the longest part
    //      (255 x deg dependent table look-ups per lane), is made by the WHOLE wave for one dirty code word after the other: 4 positions per lane, the
    //      roots collected in the order the serial loop finds them (ballot + prefix count).  At 5 dB, where most super frames have dirty code words,
    //      the kernel went 0.178 -> 0.117 (syndromes) -> see docs/history/r06.md ms per step; at 8 dB and above nothing of this runs.
    int my_ret = 0;
    const bool dirty = lane < R && syn_or[lane];
    const Gf gf{gexp, glog};
    uint8_t lam[RS_NR + 1];
    int deg_lambda = 0;
    if (__builtin_amdgcn_ballot_w64(dirty)) {                  // wave-uniform: clean super frames skip all of it
      if (dirty) {
        rs_berlekamp_massey(reinterpret_cast<const uint8_t *>(syn_w[lane]), gf, lam, deg_lambda);
#pragma unroll
        for (int i = 0; i <= RS_NR; i++) s_lam[lane][i] = lam[i];
        s_deg[lane] = (uint8_t)deg_lambda;
      }
      __syncthreads();
      unsigned long long dm = __builtin_amdgcn_ballot_w64(dirty);
      while (dm) {
        const int c = __builtin_ctzll(dm);
        dm &= dm - 1;
        const int dg = s_deg[c];
        int count = 0;
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
          const int i = 64 * pass + lane + 1;
          const bool root = i <= RS_NN && rs_chien_at(s_lam[c], dg, gexp, i) == 0;
          const unsigned long long b = __builtin_amdgcn_ballot_w64(root);
          const int idx = count + __builtin_popcountll(b & ((1ull << lane) - 1ull));
          if (root && idx < RS_NR) s_root[c][idx] = (uint8_t)i;
          count += __builtin_popcountll(b);
        }
        if (lane == 0) s_rootn[c] = count;
      }
      __syncthreads();
      if (dirty)
        my_ret = s_rootn[lane] != deg_lambda ? -1
                                             : rs_forney(CwStrided{win + lane, R}, gf, reinterpret_cast<const uint8_t *>(syn_w[lane]), lam, deg_lambda, s_root[lane], s_rootn[lane]);
    }
    int corr = my_ret > 0 ? my_ret : 0, fail = my_ret < 0 ? 1 : 0;
    corr = wave_sum_int(corr); fail = wave_sum_int(fail);
    rs_corr += corr; rs_fail += fail;
    __syncthreads();
    if (lane == 0) {
      uint8_t hdr[12];
      for (int i = 0; i < 12; i++) hdr[i] = win[i];
      const bool ok = firecode_check_and_correct(hdr, s_fc, t.fc_syndrome);   // :230-240
      int flag = ok ? 1 : 0;
      if (ok) {
        bool changed = false;
        for (int i = 0; i < 11; i++) changed = changed || (hdr[i] != hdr0[i]);
        if (changed) flag |= 2;
        for (int i = 0; i < 12; i++) win[i] = hdr[i];
        // AU table, mp4processor.cpp:256-306
        const int dac = (hdr[2] >> 6) & 1, sbr = (hdr[2] >> 5) & 1, end = 110 * R;
        int n_au;
        switch (2 * dac + sbr) {
        case 0: n_au = 4; s_au[0] = 8; s_au[1] = hdr[3] * 16 + (hdr[4] >> 4); s_au[2] = (hdr[4] & 0xf) * 256 + hdr[5];
                s_au[3] = hdr[6] * 16 + (hdr[7] >> 4); s_au[4] = end; break;
        case 1: n_au = 2; s_au[0] = 5; s_au[1] = hdr[3] * 16 + (hdr[4] >> 4); s_au[2] = end; break;
        case 2: n_au = 6; s_au[0] = 11; s_au[1] = hdr[3] * 16 + (hdr[4] >> 4); s_au[2] = (hdr[4] & 0xf) * 256 + hdr[5];
                s_au[3] = hdr[6] * 16 + (hdr[7] >> 4); s_au[4] = (hdr[7] & 0xf) * 256 + hdr[8];
                s_au[5] = hdr[9] * 16 + (hdr[10] >> 4); s_au[6] = end; break;
        default: n_au = 3; s_au[0] = 6; s_au[1] = hdr[3] * 16 + (hdr[4] >> 4); s_au[2] = (hdr[4] & 0xf) * 256 + hdr[5];
                s_au[3] = end; break;
        }
        s_au[7] = n_au;
      }
      s_flag = flag;
    }
    __syncthreads();
    const int flag = s_flag;
    if (flag & 1) {                          // :149-158
      if (flag & 2) fc_corr++;
      sf_sync = 4; sf_ok++;
      const int n_au = s_au[7];
      // :318-333 AU CRCs.  The CRC register is linear in the message: every lane runs the table recursion over its own
      // slice from state 0, the slice results are moved to the end of the AU by multiplying with x^(8 n) mod P
      // (crc_xpow, in LDS) and XOR-ed together; the 0xFFFF start value rides on the first two bytes -- calc_crc (crc.cpp:75-86)
      // without a several-hundred-step look-up chain on one lane.
      int good = 0, bad = 0;
      unsigned crc_mask = 0, len_mask = 0;     // per access unit: passed its CRC / failed the length check (dabx_superframe_info)
      for (int a = 0; a < n_au; a++) {
        const int st = s_au[a], len = s_au[a + 1] - st - 2;
        if (len > 960 || len < 0 || st + len + 2 > 110 * R) { bad++; len_mask |= 1u << a; continue; }
        const int per = (len + 63) >> 6, from = lane * per, to = min(len, from + per);
        const unsigned xp_slice = s_xpow[from < to ? len - to : 0];
        // The 0xFFFF start value of a 16-bit CRC is the same as complementing the first two message bytes and starting from 0
        // (the register only ever shifts the start value through those two steps): the lanes that own bytes 0 and 1 do that, and
        // the second crc_mulmod that lane 0 ran for the start value's contribution -- with the other 63 lanes waiting -- is gone.
        const unsigned first2 = len >= 2 ? 0xFFu : 0u;
        unsigned crc = 0;
        for (int i = from; i < to; i++) crc = (s_crc[(win[st + i] ^ (i < 2 ? first2 : 0u) ^ (crc >> 8)) & 0xFF] ^ (crc << 8)) & 0xFFFFu;
        unsigned acc = from < to ? crc_mulmod(crc, xp_slice) : 0u;
        if (len < 2 && lane == 0) acc ^= crc_mulmod(0xFFFFu, s_xpow[len]);      // a message shorter than the register: the start value's contribution as it was
        acc = wave_xor(acc);
        const unsigned want = ((unsigned)win[st + len] << 8) | win[st + len + 1];
        if (((~acc) & 0xFFFFu) == want) { good++; crc_mask |= 1u << a; } else bad++;
      }
      au_ok += good; au_bad += bad;
      if (lane == 0 && e.sf_info) {            // what _process_super_frame knows when it hands the access units on (mp4processor.cpp:256-333)
        dabx_superframe_info r;
        r.num_aus = (uint8_t)n_au; r.au_crc_ok = (uint8_t)crc_mask; r.au_len_bad = (uint8_t)len_mask; r.stream_parms = (uint8_t)(win[2] & 0x7F);
#pragma unroll
        for (int a = 0; a < 7; a++) r.au_start[a] = a <= n_au ? (uint16_t)s_au[a] : (uint16_t)0;
        r.rs_corrected = (uint16_t)corr; r.rs_failed = (uint8_t)fail; r.fc_corrected = (flag & 2) ? 1 : 0; r.reserved = 0;
        r.first_frame = oldest;
        e.sf_info[((size_t)s * e.max_subch + j) * SF_SLOTS + (size_t)(sf_count % SF_SLOTS)] = r;
      }
      uint8_t *sfo = e.sf_out + (((size_t)s * e.max_subch + j) * SF_SLOTS + (size_t)(sf_count % SF_SLOTS)) * e.sf_stride;
      for (int i = lane; i < (110 * R + 3) / 4; i += 64)
        reinterpret_cast<uint32_t *>(sfo)[i] = reinterpret_cast<const uint32_t *>(win)[i];
      sf_count++;
    } else {                                 // :159-169
      sf_sync--;
      if (sf_sync == 0) { blocks_in_buf = 4; sf_fail++; }
    }
  }
  if (lane == 0) {
    sc.cif_out = cif_out; sc.blocks_in_buf = blocks_in_buf; sc.sf_sync = sf_sync; sc.sf_count = sf_count;
    sc.sf_ok += sf_ok; sc.sf_fail += sf_fail; sc.rs_corr += rs_corr; sc.rs_fail += rs_fail;
    sc.fc_corr += fc_corr; sc.au_ok += au_ok; sc.au_bad += au_bad;
  }
}

// ---------------------------------------------------------------------------------------------- launchers
__global__ void k_msc_snap(EngineDev e, int cifs)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= e.n_streams) return;
  // CIFs older than this batch were produced while no sub-channel was configured (no MSC batches ran): skip them
  const long long cif_no = e.ctl[s].cif_no, done = e.ctl[s].msc_done_cif;
  e.snap[s] = BatchSnap{done > cif_no - cifs ? done : cif_no - cifs, cif_no};
}
__global__ void k_msc_done(EngineDev e)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < e.n_streams) e.ctl[s].msc_done_cif = e.snap[s].cif_no;
}

extern const char *const kStepKernelNames[11];
// "k_demap_fic": the first k_demap_frame launch of a frame (symbols 1..3) when the FIC is decoded on its own stream
const char *const kStepKernelNames[11] = {"k_acquire", "k_frame_head", "k_symbols", "k_demap_frame", "k_fic_frame",
                                          "k_frame_tail", "k_msc_prep", "k_msc_vitT", "k_msc_frame", "k_dabplus", "k_demap_fic"};

// Front end of one batch step (everything with frame-to-frame feedback).
// Overlapped schedule (ss.d set): the FIC lives in symbols 1..3 -- those are demapped first on the front-end stream a, then
// the FIC decoder (four 774-step trellises per stream, a latency-bound kernel and the next link of the frame's feedback
// chain) and the frame tail follow on a, while the 72 MSC symbols are demapped on stream d: nothing on the chain of the NEXT
// frame needs them, so a goes on to frame n + 1 while d is busy; the next frame's first demapper launch (and the MSC batch)
// wait for d.  Serial schedule (cfg.schedule = 1, ss.d null): every kernel on a in program order.
int launch_front_step(const EngineDev &e_in, EngineStreams &ss, Marker &mk, bool async_acquire, bool all_locked)
{
  const DevTables *t;
  int rc = get_tables(&t);
  if (rc) return rc;
  EngineDev e = e_in;
  e.parity = (int)(ss.step_count++ & 1u);               // spectra buffer of this step
  e.step_seq = ss.step_count;                           // 1, 2, ...
  e.flag_sync = (ss.d && ss.fic_on_d && e.sym_seq) ? 1 : 0;
  hipStream_t st = ss.a;
  // Streams out of lock (k_acquire).  In step: on the front-end stream, before the frame head -- every step then offers every
  // stream a frame's worth of search.  Asynchronous: on HIP stream q; a pass is launched when the previous one has finished
  // (hipEventQuery, no wait); a pass is a frame's worth of samples (< 2 ms: what a dabx_synchronize may have to wait for).
  // cfg.exact_level_tracker: k_level_exact in front of every pass (same HIP stream).  The frame chain never waits for it: the level is
  // read nowhere in lock, the tracker walks whatever the receiver has read since its last pass ([level_pos, rd): 0.2 ms per frame
  // and stream since level_par.h, a fifth of a step), and the samples it still has to see stay in the ring (push_room, engine.cpp).
  // all_locked: the device's count of streams in lock (host memory, read by dabx_process without a wait) equals n_streams: nobody is
  // searching, k_acquire would return from every block at once -- its launch, event record and query are left out (round 4 had put them on
  // every step; the one-stream chain, 0.15 ms per frame, paid 8-11 % for them: profiles/r05_ab/ab3_single_ensemble_r3_r4_head.txt).  A
  // stream that loses its lock decrements the count in k_frame_head: the first step issued after that launches the search again.
  if (all_locked && !e.exact_level) {
  } else if (async_acquire && ss.q) {
    // a search pass only when q has nothing left to do (a host that queues twenty steps in a millisecond starts one, not twenty: a pass can
    // take 2 ms); the level tracker with EVERY step -- it is short, and what it leaves undone the next one has to do
    const bool go = !ss.acq_in_flight || hipEventQuery(ss.acq_done) == hipSuccess;
    if (go || e.exact_level) {
      // two passes must never work on a stream at the same time: one that ran in step (an earlier call, or the start-up rule of
      // dabx_process) has to be through before the first one on q starts
      if (ss.acq_a_pending) { DABX_HIP(hipStreamWaitEvent(ss.q, ss.acq_a_done, 0)); ss.acq_a_pending = false; }
      if (e.exact_level) {
        // the tracker of step n walks the frame of step n - 1 while the chain demodulates frame n: behind that chain's tail, not before
        if (ss.tail_recorded) DABX_HIP(hipStreamWaitEvent(ss.q, ss.tail_done, 0));
        hipLaunchKernelGGL(k_level_exact, dim3(e.n_streams), dim3(128), 0, ss.q, e);
      }
      if (go) { mk.begin(0, ss.q); hipLaunchKernelGGL(k_acquire, dim3(e.n_streams), dim3(256), 0, ss.q, e, *t, 1); mk.end(0, ss.q); }
      DABX_HIP(hipEventRecord(ss.acq_done, ss.q));
      ss.acq_in_flight = true;
    }
  } else {
    if (ss.acq_in_flight) { DABX_HIP(hipStreamWaitEvent(st, ss.acq_done, 0)); ss.acq_in_flight = false; }   // a pass of an earlier, asynchronous call
    if (e.exact_level) hipLaunchKernelGGL(k_level_exact, dim3(e.n_streams), dim3(128), 0, st, e);
    mk.begin(0, st); hipLaunchKernelGGL(k_acquire, dim3(e.n_streams), dim3(256), 0, st, e, *t, 1); mk.end(0, st);
    if (ss.q && ss.acq_a_done) { DABX_HIP(hipEventRecord(ss.acq_a_done, st)); ss.acq_a_pending = true; }
  }
  mk.begin(1, st); hipLaunchKernelGGL(k_frame_head, dim3(e.n_streams), dim3(256), 0, st, e, *t); mk.end(1, st);
#ifndef DABX_GROUPS
#define DABX_GROUPS 1
#endif
  // Experiment builds (-DDABX_GROUPS=n): k_symbols -> k_demap_fic -> k_demap_frame6 issued per group of n_streams / n streams,
  // back to back, all streams resident: does a group's spectra (128 streams = 118 MB) survive in the Infinity Cache between its writer and its reader?
  const bool grouped = DABX_GROUPS > 1 && ss.d && !ss.fic_on_d && e.n_streams % DABX_GROUPS == 0 && e.n_streams / DABX_GROUPS >= 48;
  if (!grouped) { mk.begin(2, st); hipLaunchKernelGGL(k_symbols_persistent, dim3(sym_blocks_per_stream(e.n_streams), e.n_streams), dim3(256), 0, st, e, *t); mk.end(2, st); }
  // kernel instance by (ESoftBitType, symbol conversion of the canonical / SIMD builds)
#define DABX_DEMAP_DISPATCH(KERNEL, ...)                                                                                         \
  do {                                                                                                                          \
    const int st_ = e.demap.soft_type == 3 ? 3 : e.demap.soft_type == 2 ? 2 : 1;                                                \
    if (e.demap.track_mer) {                                   /* LCD statistics on: the instances that advance the MER's IIR too */ \
      if (e.tie_mode) {                                                                                                         \
        if (st_ == 3) hipLaunchKernelGGL((KERNEL<3, true, true>), __VA_ARGS__);                                                 \
        else if (st_ == 2) hipLaunchKernelGGL((KERNEL<2, true, true>), __VA_ARGS__);                                            \
        else hipLaunchKernelGGL((KERNEL<1, true, true>), __VA_ARGS__);                                                          \
      } else {                                                                                                                  \
        if (st_ == 3) hipLaunchKernelGGL((KERNEL<3, false, true>), __VA_ARGS__);                                                \
        else if (st_ == 2) hipLaunchKernelGGL((KERNEL<2, false, true>), __VA_ARGS__);                                           \
        else hipLaunchKernelGGL((KERNEL<1, false, true>), __VA_ARGS__);                                                         \
      }                                                                                                                         \
    } else if (e.tie_mode) {                                                                                                    \
      if (st_ == 3) hipLaunchKernelGGL((KERNEL<3, true, false>), __VA_ARGS__);                                                  \
      else if (st_ == 2) hipLaunchKernelGGL((KERNEL<2, true, false>), __VA_ARGS__);                                             \
      else hipLaunchKernelGGL((KERNEL<1, true, false>), __VA_ARGS__);                                                           \
    } else {                                                                                                                    \
      if (st_ == 3) hipLaunchKernelGGL((KERNEL<3, false, false>), __VA_ARGS__);                                                 \
      else if (st_ == 2) hipLaunchKernelGGL((KERNEL<2, false, false>), __VA_ARGS__);                                            \
      else hipLaunchKernelGGL((KERNEL<1, false, false>), __VA_ARGS__);                                                          \
    }                                                                                                                           \
  } while (0)
  auto demap = [&](hipStream_t q, int l0, int l1) {
    DABX_DEMAP_DISPATCH(k_demap_frame6, dim3(e.n_streams), dim3(DEMAP_THREADS), 0, q, e, *t, l0, l1);
  };
  if (ss.d && ss.fic_on_d) {
    // few streams: both demapper launches on d (in order: no event between the per-carrier state's writer and its reader), the FIC decoder and the
    // tail on a behind the FIC symbols (pipeline.h, EngineStreams::fic_on_d)
    // the two hand-overs (k_symbols -> k_demap_fic, k_demap_fic -> k_fic_frame) are device-side sequence numbers: no packet between the two
    // demapper launches on d, none in front of them (EngineDev::flag_sync)
    hipLaunchKernelGGL(k_sym_publish, dim3((e.n_streams + 63) / 64), dim3(64), 0, st, e);
    mk.begin(3, ss.d);
    DABX_DEMAP_DISPATCH(k_demap_whole, dim3(e.n_streams), dim3(DEMAP_THREADS), 0, ss.d, e, *t);
    mk.end(3, ss.d);
    // (demap_done is recorded when somebody needs it -- the MSC batch, once per 7 frames: every packet between two kernels of a HIP stream is a
    //  bubble of 6-12 us on this loop, profiles/r06_single_ensemble_timeline_after.txt)
    ss.demap_in_flight = true; ss.demap_unrecorded = true;
#if DABX_GROUPS > 1          // experiment builds only (profiles/r06_ab/ab7_grouped_issue_negative.txt)
  } else if (grouped) {
    static hipEvent_t ev[16] = {nullptr};      // (one engine per process in the A/B runs)
    const int sg = e.n_streams / DABX_GROUPS;
    for (int g = 0; g < DABX_GROUPS; g++) {
      EngineDev eg = e;
      eg.s0 = g * sg;
      if (!ev[g]) DABX_HIP(hipEventCreateWithFlags(&ev[g], hipEventDisableTiming | hipEventReleaseToDevice));
      mk.begin(2, st); hipLaunchKernelGGL(k_symbols_persistent, dim3(sym_blocks_per_stream(e.n_streams), sg), dim3(256), 0, st, eg, *t); mk.end(2, st);
      if (g == 0 && ss.demap_in_flight) { DABX_HIP(hipStreamWaitEvent(st, ss.demap_done, 0)); ss.demap_in_flight = false; }
      mk.begin(10, st);
      DABX_DEMAP_DISPATCH(k_demap_fic, dim3(sg), dim3(DEMAP_THREADS), 0, st, eg, *t);
      mk.end(10, st);
      DABX_HIP(hipEventRecord(ev[g], st));
      DABX_HIP(hipStreamWaitEvent(ss.d, ev[g], 0));
      mk.begin(3, ss.d);
      DABX_DEMAP_DISPATCH(k_demap_frame6, dim3(sg), dim3(DEMAP_THREADS), 0, ss.d, eg, *t, 3, 75);
      mk.end(3, ss.d);
    }
    DABX_HIP(hipEventRecord(ss.demap_done, ss.d));
    ss.demap_in_flight = true;
#endif
  } else if (ss.d) {
    if (ss.demap_in_flight) { DABX_HIP(hipStreamWaitEvent(st, ss.demap_done, 0)); ss.demap_in_flight = false; }
    mk.begin(10, st);
    DABX_DEMAP_DISPATCH(k_demap_fic, dim3(e.n_streams), dim3(DEMAP_THREADS), 0, st, e, *t);
    mk.end(10, st);
    DABX_HIP(hipEventRecord(ss.fic_go, st));
    DABX_HIP(hipStreamWaitEvent(ss.d, ss.fic_go, 0));
    mk.begin(3, ss.d); demap(ss.d, 3, 75); mk.end(3, ss.d);
    DABX_HIP(hipEventRecord(ss.demap_done, ss.d));
    ss.demap_in_flight = true;
  } else {
    mk.begin(3, st); demap(st, 0, 75); mk.end(3, st);
  }
  mk.begin(4, st); hipLaunchKernelGGL(k_fic_frame, dim3(e.n_streams), dim3(256), 0, st, e, *t, 0, 4); mk.end(4, st);
  mk.begin(5, st); hipLaunchKernelGGL(k_frame_tail, dim3(e.n_streams), dim3(256), 0, st, e, *t); mk.end(5, st);
  if (e.exact_level && ss.tail_done) { DABX_HIP(hipEventRecord(ss.tail_done, st)); ss.tail_recorded = true; }
  DABX_HIP(hipGetLastError());
  return 0;
}

int launch_msc_prep(const EngineDev &e, int cifs, const MscLaunch &L, hipStream_t st, Marker &mk);
int launch_msc_vitT(const EngineDev &e, int cifs, const MscLaunch &L, hipStream_t st, Marker &mk);

// MSC decode of the newest `cifs` CIFs (4 per front-end step, <= 4 * MSC_BATCH_FRAMES; 1 for the per-symbol stage entry) + DAB+ stage.
// `e.snap` must point at the snapshot buffer of this batch.
int launch_deliver_msc(const EngineDev &e, const DeliverDev &dv, hipStream_t st, bool with_lf);
int launch_deliver_lf(const EngineDev &e, const DeliverDev &dv, hipStream_t st);
// `dv` (optional): the chunk's slot gather (deliver.hip) goes behind the DAB+ stage on the stream that ran it, before the batch's
// completion event; *tail (optional) = the stream whose work completes the batch.
int launch_msc_batch(const EngineDev &e, int cifs, const MscFast *fast, EngineStreams &ss, Marker &mk, const DeliverDev *dv, hipStream_t *tail)
{
  const DevTables *t;
  int rc = get_tables(&t);
  if (rc) return rc;
  if (tail) *tail = ss.a;
  if (e.fic_only || e.max_subch <= 0 || !e.msc_out) {
    if (dv && (rc = launch_deliver_msc(e, *dv, ss.a, true))) return rc;    // the slot table says "nothing" for every slot
    return 0;
  }
  const int jobs = e.n_streams * cifs * e.max_subch;
  // The MSC symbols of the newest frame may still be on their way into the time-de-interleaver ring (stream d): whoever READS the ring waits
  // for them -- the batch's own stream, not the frame chain (up to round 5 stream a waited here: 0.19 ms of every 7-step period during which the
  // next frame's head and symbols could have run; one ensemble: 90 us of every 7 frames)
  auto wait_for_demapper = [&](hipStream_t who) -> int {
    if (!ss.demap_in_flight) return 0;
    if (ss.demap_unrecorded) { DABX_HIP(hipEventRecord(ss.demap_done, ss.d)); ss.demap_unrecorded = false; }
    DABX_HIP(hipStreamWaitEvent(who, ss.demap_done, 0));
    if (who == ss.a) ss.demap_in_flight = false;          // (a has it behind it for good; otherwise the next k_demap_fic on a still waits)
    return 0;
  };
  // the previous batch (stream b) owns SubchDev / msc_done_cif until it has finished
  if (ss.msc_in_flight) { DABX_HIP(hipStreamWaitEvent(ss.a, ss.msc_done, 0)); ss.msc_in_flight = false; }
  hipLaunchKernelGGL(k_msc_snap, dim3((e.n_streams + 255) / 256), dim3(256), 0, ss.a, e, cifs);
  // classes worth a lane-per-trellis launch in this batch (pipeline.h, MscClass)
  MscLaunch L{};
  unsigned fast_mask = 0;
  long long fast_jobs = 0, fast_pairs = 0;
  if (fast) {
    for (int c = 0; c < fast->n_cls; c++) fast_jobs += (long long)fast->cls[c].n_pairs * cifs;
    if (fast_jobs >= fast->min_jobs)
      for (int c = 0; c < fast->n_cls; c++) {
        const MscClass &k = fast->cls[c];
        MscLaunchCls &o = L.c[L.n++];
        o.n_in = k.n_in; o.nbits = k.nbits; o.n_pairs = k.n_pairs; o.g0 = L.groups;
        o.map2 = k.map2; o.pairs = k.pairs; o.inT = k.inT[ss.batch_parity]; o.decT = k.decT;
        L.groups += (int)(((long long)k.n_pairs * cifs + 63) / 64);
        fast_mask |= 1u << c;
        fast_pairs += k.n_pairs;
      }
  }
  if (L.n > 0) {
    // Overlapped schedule: time de-interleave, lane-per-trellis decode and DAB+ stage of the batch run on stream b while the
    // front end (a) goes straight on to the next frames.  k_msc_prep reads ring slots the front end only rewrites 20 CIFs
    // (5 frames) later; dabx_process makes the front-end stream wait for `prep_b_done` before it gets there.  Serial
    // schedule (ss.b null): the same kernels on a.
    hipStream_t sb = ss.b ? ss.b : ss.a;
    if (ss.b) {
      DABX_HIP(hipEventRecord(ss.prep_done, ss.a));
      DABX_HIP(hipStreamWaitEvent(ss.b, ss.prep_done, 0));
    }
    if (fast_pairs < fast->slots_active && (rc = wait_for_demapper(ss.a))) return rc;   // the wave-per-trellis leftovers below read the ring on a
    if ((rc = wait_for_demapper(sb))) return rc;
    if ((rc = launch_msc_prep(e, cifs, L, sb, mk))) return rc;
    if (ss.b) {
      DABX_HIP(hipEventRecord(ss.prep_b_done, ss.b));
      ss.prep_pending = true;
    }
    if ((rc = launch_msc_vitT(e, cifs, L, sb, mk))) return rc;
    if (fast_pairs < fast->slots_active) {
      // the remaining sub-channels: wave per trellis, on the front-end stream (it reads the TDI ring in place)
      mk.begin(8, ss.a);
      hipLaunchKernelGGL(k_msc_frame, dim3((jobs + 3) / 4), dim3(256), 0, ss.a, e, *t, cifs, fast_mask);
      mk.end(8, ss.a);
      if (ss.b) {
        DABX_HIP(hipEventRecord(ss.prep_done, ss.a));
        DABX_HIP(hipStreamWaitEvent(ss.b, ss.prep_done, 0));
      }
    }
    // the chunk's logical frames exist: into the slab with them, their share of the transfer starts while the DAB+ stage runs (deliver.hip)
    if (dv && (rc = launch_deliver_lf(e, *dv, sb))) return rc;
    mk.begin(9, sb);
    hipLaunchKernelGGL(k_dabplus, dim3(e.n_streams * e.max_subch), dim3(64), 0, sb, e, *t);
    hipLaunchKernelGGL(k_msc_done, dim3((e.n_streams + 255) / 256), dim3(256), 0, sb, e);
    mk.end(9, sb);
    if (dv && (rc = launch_deliver_msc(e, *dv, sb, !dv->lf_done))) return rc;
    if (tail) *tail = sb;
    if (ss.b) {
      DABX_HIP(hipEventRecord(ss.msc_done, ss.b));
      ss.msc_in_flight = true;
#ifdef DABX_EXCLUSIVE_MSC            // experiment builds only: the next frames' front end waits for the batch (no overlap of the decoder with the frame chain)
      DABX_HIP(hipStreamWaitEvent(ss.a, ss.msc_done, 0));
      ss.msc_in_flight = false;
#endif
    }
  } else {
    // small batches (few streams / small profile classes only): the wave-per-trellis decoder.  With the overlapped schedule on stream b as
    // well (round 6): for ONE ensemble its 0.1 ms + the DAB+ stage's 0.05 ms per 7-frame batch stood on the frame chain (23 us per frame of
    // configs[2]'s 157).  It reads the time-de-interleaver ring in place: prep_b_done says when the front end may rewrite those slots.
    hipStream_t sb = ss.b ? ss.b : ss.a;
    if (ss.b) {
      DABX_HIP(hipEventRecord(ss.prep_done, ss.a));
      DABX_HIP(hipStreamWaitEvent(ss.b, ss.prep_done, 0));
    }
    if ((rc = wait_for_demapper(sb))) return rc;
    mk.begin(8, sb);
    hipLaunchKernelGGL(k_msc_frame, dim3((jobs + 3) / 4), dim3(256), 0, sb, e, *t, cifs, 0u);
    mk.end(8, sb);
    if (ss.b) {
      DABX_HIP(hipEventRecord(ss.prep_b_done, ss.b));
      ss.prep_pending = true;
    }
    if (dv && (rc = launch_deliver_lf(e, *dv, sb))) return rc;
    mk.begin(9, sb);
    hipLaunchKernelGGL(k_dabplus, dim3(e.n_streams * e.max_subch), dim3(64), 0, sb, e, *t);
    hipLaunchKernelGGL(k_msc_done, dim3((e.n_streams + 255) / 256), dim3(256), 0, sb, e);
    mk.end(9, sb);
    if (dv && (rc = launch_deliver_msc(e, *dv, sb, !dv->lf_done))) return rc;
    if (tail) *tail = sb;
    if (ss.b) {
      DABX_HIP(hipEventRecord(ss.msc_done, ss.b));
      ss.msc_in_flight = true;
    }
  }
  ss.batch_parity ^= 1;
  DABX_HIP(hipGetLastError());
  return 0;
}

int launch_fic_only(const EngineDev &e, hipStream_t st, int first, int count)
{
  const DevTables *t;
  int rc = get_tables(&t);
  if (rc) return rc;
  hipLaunchKernelGGL(k_fic_frame, dim3(e.n_streams), dim3(256), 0, st, e, *t, first, count);
  DABX_HIP(hipGetLastError());
  return 0;
}

__global__ void k_i16_to_sym(const int16_t *soft, uint8_t *sym, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sym[i] = soft_to_sym(soft[i]);
}
int launch_i16_to_sym(const int16_t *soft, uint8_t *sym, size_t n, hipStream_t st)
{
  hipLaunchKernelGGL(k_i16_to_sym, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, soft, sym, n);
  DABX_HIP(hipGetLastError());
  return 0;
}

// Per-symbol stage entry (dabx_msc_process_block == MscHandler::process_block, msc_handler.cpp:148-168): the 3072 soft bits
// of OFDM symbol blk (0..17 within the CIF) go into the planar time-de-interleaver ring as Viterbi symbols; closing the
// CIF advances the CIF counter, exactly what k_demap_frame / k_frame_tail do for a whole frame.
__global__ void k_stage_msc_block(EngineDev e, const int16_t *soft, int blk)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K2) return;
  uint8_t *tdi = e.tdi;                                     // stream 0
  tdi[tdi_off(e.ctl[0].cif_no, blk * K2 + i)] = soft_to_sym_mode(soft[i], e.tie_mode);
}
__global__ void k_stage_cif_done(EngineDev e) { e.ctl[0].cif_no += 1; }
int launch_stage_msc_block(const EngineDev &e, const int16_t *soft_dev, int blk, bool closes_cif, hipStream_t st)
{
  hipLaunchKernelGGL(k_stage_msc_block, dim3(K2 / 256), dim3(256), 0, st, e, soft_dev, blk);
  if (closes_cif) hipLaunchKernelGGL(k_stage_cif_done, dim3(1), dim3(1), 0, st, e);
  DABX_HIP(hipGetLastError());
  return 0;
}

__global__ void k_commit(unsigned long long *wr, int n_streams, int stream, unsigned long long n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_streams && (stream < 0 || i == stream)) wr[i] += n;
}
int launch_commit(const EngineDev &e, int stream, unsigned long long n, hipStream_t st)
{
  hipLaunchKernelGGL(k_commit, dim3((e.n_streams + 255) / 256), dim3(256), 0, st, e.wr, e.n_streams, stream, n);
  DABX_HIP(hipGetLastError());
  return 0;
}

// host IQ formats -> cf32 ring (raw_reader.cpp:66-70, wav_reader.cpp:164)
__global__ void k_convert_iq(const void *src, int fmt, float2 *ring, int ring_len, unsigned long long wr0, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2 v;
  if (fmt == 0) v = reinterpret_cast<const float2 *>(src)[i];
  else if (fmt == 1) { const short2 q = reinterpret_cast<const short2 *>(src)[i]; v = make_float2(q.x / 32768.0f, q.y / 32768.0f); }
  else { const uchar2 q = reinterpret_cast<const uchar2 *>(src)[i]; v = make_float2((q.x - 127.38f) / 128.0f, (q.y - 127.38f) / 128.0f); }
  ring[(size_t)((wr0 + i) % (unsigned long long)ring_len)] = v;
}
int launch_convert_iq(const void *src, int fmt, float2 *ring, int ring_len, unsigned long long wr0, size_t n, hipStream_t st)
{
  hipLaunchKernelGGL(k_convert_iq, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, fmt, ring, ring_len, wr0, n);
  DABX_HIP(hipGetLastError());
  return 0;
}

// bulk ingest (engine.cpp, dabx_ingest_commit): [S][n] samples of fmt in `src` -> every stream's ring behind its committed index
__global__ __launch_bounds__(256) void k_ingest_convert(const void *src, int fmt, float2 *iq, int ring_len, const unsigned long long *wr, size_t n)
{
  const int s = blockIdx.y;
  const unsigned long long wr0 = wr[s];
  float2 *ring = iq + (size_t)s * ring_len;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t j = (size_t)s * n + i;
    float2 v;
    if (fmt == 0) v = reinterpret_cast<const float2 *>(src)[j];
    else if (fmt == 1) { const short2 q = reinterpret_cast<const short2 *>(src)[j]; v = make_float2(q.x / 32768.0f, q.y / 32768.0f); }
    else { const uchar2 q = reinterpret_cast<const uchar2 *>(src)[j]; v = make_float2((q.x - 127.38f) / 128.0f, (q.y - 127.38f) / 128.0f); }
    ring[(size_t)((wr0 + i) % (unsigned long long)ring_len)] = v;
  }
}
int launch_ingest_convert(const EngineDev &e, const void *src, int fmt, size_t n, hipStream_t st)
{
  const unsigned bx = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_ingest_convert, dim3(bx, e.n_streams), dim3(256), 0, st, src, fmt, e.iq, e.ring_len, e.wr, n);
  DABX_HIP(hipGetLastError());
  return 0;
}

}  // namespace dabx
