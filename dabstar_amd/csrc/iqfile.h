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

// engine.cpp: ring of a stream, its write position (host mirror), the read position (device, synchronises) and stream
}  // namespace dabx

extern "C" int dabx_internal_commit(dabx_engine *e, int stream, size_t n);   // commit of samples iqfile.cpp wrote itself (announces them first)
extern "C" int dabx_internal_ring_info(dabx_engine *e, int stream, float2 **ring, int *ring_len, unsigned long long *wr,
                                        unsigned long long *rd, hipStream_t *st);
