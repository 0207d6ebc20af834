// iqfile.h -- internal: recorded-IQ decode/resample launchers (iqfile.hip) and the engine hooks they need.
#pragma once
#include "dabx_internal.h"

namespace dabx {

struct IqDecode {      // by-value kernel argument
  int family, container, big_endian, swap_iq;
  int bytes;           // per channel
  float int_scale;     // 1 / 2^(bits-1) for the integer containers
  // reference_quirks (UFF int24 / MSB only): samples per read block of the reference's reader (rate / 1000), 0 = off --
  // Q's middle byte comes from byte 4 i + 4 of the block (xml_reader.cpp:316,462); sign7f: QI ORs 0x7F000000 (:465,:469)
  int quirk_block, quirk_i24, quirk_sign7f;   // quirk_block != 0: the feed hands over whole read blocks only
};
int launch_decode_iq(const uint8_t *src, const IqDecode &d, float2 *dst, unsigned long long dst0, int dst_len, size_t n, hipStream_t st);
int launch_resample_1ms(const float2 *V, int M, const int16_t *tab_int, const float *tab_frac, float2 *dst, unsigned long long dst0,
                        int dst_len, size_t n_out, hipStream_t st);

// iqfile.cpp: format check and the readers' interpolation tables (wav_reader.cpp:67-82, xml_reader.cpp:237-244), shared with the bulk ingest
int iq_check_format(const dabx_iq_format *f, IqDecode *d);
void iq_resample_tables(int family, int rate, int *M, int16_t *tab_int /* [2048] */, float *tab_frac /* [2048] */);

// Bulk ingest, general form (engine.cpp, dabx_ingest_open_formats): what one stream's share of a slab is and where it goes.  One record per
// stream and commit, uploaded in front of the two kernels below.
struct IngestJob {
  unsigned long long src_off;     // byte offset of the stream's payload in the device slab
  unsigned long long dst0;        // absolute ring index of the first sample written (the committed index, host mirror)
  unsigned n;                     // complete input samples in the slab (0: the stream takes no part)
  unsigned carry_n;               // samples carried over from the previous slab (resampling streams: <= M + 1)
  unsigned M;                     // input samples per millisecond (rate / 1000); 0: the recording is at 2.048 MS/s, no resampling
  unsigned blocks;                // 1-ms blocks this commit resamples: 2048 output samples each
  unsigned keep;                  // samples of [carry | decoded] kept for the next slab
  unsigned tab;                   // index of the stream's interpolation tables
  IqDecode dec;
};
struct IngestMulti {              // by-value kernel argument
  const uint8_t *slab;
  const IngestJob *jobs;          // [S] device
  float2 *iq; int ring_len;       // EngineDev::iq
  float2 *work; size_t work_pitch;    // [S][work_pitch] carry + decoded samples of the resampling streams
  float2 *carry; size_t carry_pitch;  // [S][carry_pitch]
  const int16_t *tab_int; const float *tab_frac;   // [n_tabs][2048]
};
int launch_ingest_multi(const IngestMulti &m, int n_streams, unsigned max_n, unsigned max_out, hipStream_t st);
int launch_commit_counts(unsigned long long *wr, const unsigned *counts_dev, int n_streams, hipStream_t st);
}  // namespace dabx

extern "C" int dabx_internal_commit(dabx_engine *e, int stream, size_t n);   // commit of samples iqfile.cpp wrote itself (announces them first)
extern "C" int dabx_internal_ring_info(dabx_engine *e, int stream, float2 **ring, int *ring_len, unsigned long long *wr,
                                        unsigned long long *rd, hipStream_t *st);
