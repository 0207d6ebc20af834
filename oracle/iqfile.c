/* iqfile.c -- recorded-IQ payload decoding + 1-ms linear resampler (oracle; test infrastructure only).
 * PARITY UNPINNED against the running reference: the readers are QThread/libsndfile/QtXml classes that cannot be
 * built here.  Restates devices/filereaders/raw_files/raw_reader.cpp:66-70,155-158, wav_files/wav_reader.cpp:67-82,
 * 164,190-206 (with libsndfile's documented sf_readf_float normalisation) and xml_filereader/xml_reader.cpp:43-51,
 * 76-81,226-248,254-398, sample by sample with the reader's own buffer walk (conv buffer, block of rate/1000). */
#include "dab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { FAM_RAW = 0, FAM_WAV = 1, FAM_UFF = 2 };
enum { C_U8 = 0, C_S8 = 1, C_I16 = 2, C_I24 = 3, C_I32 = 4, C_F32 = 5 };

static int shift_bits(int a) { unsigned r = 1; while (--a > 0) r <<= 1; return (int)r; }   /* xml_reader.cpp:43-51 */

static float one_channel(const uint8_t *p, int family, int container, int be, int bits)
{
  const int nb = container == C_I16 ? 2 : container == C_I24 ? 3 : (container == C_I32 || container == C_F32) ? 4 : 1;
  uint32_t raw = 0;
  for (int i = 0; i < nb; i++) raw = (raw << 8) | p[be ? i : nb - 1 - i];
  switch (container) {
  case C_U8:
    if (family == FAM_WAV) return (float)((int)raw - 128) / 128.0f;
    return ((float)raw - 127.38f) / 128.0f;
  case C_S8:
    if (family == FAM_UFF) return (float)(int8_t)raw / 127.0f;
    return (float)(int8_t)raw / 128.0f;
  case C_I16: {
    const float sc = family == FAM_UFF ? (float)shift_bits(bits) : 32768.0f;
    return (float)(int16_t)raw / sc;
  }
  case C_I24: {
    int32_t v = (int32_t)raw;
    if (v & 0x800000) v |= (int32_t)0xFF000000;
    const float sc = family == FAM_UFF ? (float)shift_bits(bits) : 8388608.0f;
    return (float)v / sc;
  }
  case C_I32: {
    const float sc = family == FAM_UFF ? (float)shift_bits(bits) : 2147483648.0f;
    return (float)(int32_t)raw / sc;
  }
  default: { float f; memcpy(&f, &raw, 4); return f; }
  }
}

/* The reference's UFF loops that differ from the format's evident meaning, literally (test infrastructure for
 * dabx_iq_format.reference_quirks): one read block of `amount` samples, lbuf = the block's bytes.
 *   int24 / MSB, IQ  xml_reader.cpp:310-326 : Q's middle byte is lbuf[4 * i + 4]
 *   int24 / MSB, QI  xml_reader.cpp:456-473 : the same index, sign extension ORs 0x7F000000, result swapped
 *   float32,     QI  xml_reader.cpp:522-545 : stored order kept (no swap)                                              */
static void uff_quirk_block(int container, int swap_iq, int be, int bits, const uint8_t *lbuf, int amount, float *out)
{
  const float scaler = (float)shift_bits(bits);
  for (int i = 0; i < amount; i++) {
    if (container == C_I24) {
      int32_t t1 = (lbuf[6 * i] << 16) | (lbuf[6 * i + 1] << 8) | lbuf[6 * i + 2];
      int32_t t2 = (lbuf[6 * i + 3] << 16) | (lbuf[4 * i + 4] << 8) | lbuf[6 * i + 5];
      const int32_t ext = swap_iq ? 0x7F000000 : (int32_t)0xFF000000;
      if (t1 & 0x800000) t1 |= ext;
      if (t2 & 0x800000) t2 |= ext;
      out[2 * i] = swap_iq ? (float)t2 / scaler : (float)t1 / scaler;
      out[2 * i + 1] = swap_iq ? (float)t1 / scaler : (float)t2 / scaler;
    } else {                                          /* float32 QI: c1, c2 in stored order */
      uint32_t w[2];
      for (int k = 0; k < 2; k++) {
        const uint8_t *p = lbuf + 8 * i + 4 * k;
        w[k] = be ? ((uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]) : ((uint32_t)p[3] << 24 | p[2] << 16 | p[1] << 8 | p[0]);
      }
      memcpy(&out[2 * i], &w[0], 4); memcpy(&out[2 * i + 1], &w[1], 4);
    }
  }
}
static int uff_quirk_applies(int family, int container, int big_endian, int swap_iq)
{
  return family == FAM_UFF && ((container == C_I24 && big_endian) || (container == C_F32 && swap_iq));
}

long long ora_iq_convert_q(int family, int container, int big_endian, int swap_iq, int bits, int rate, const uint8_t *bytes,
                           long long n_bytes, float *out, long long max_out, int quirks);
long long ora_iq_convert(int family, int container, int big_endian, int swap_iq, int bits, int rate, const uint8_t *bytes,
                         long long n_bytes, float *out, long long max_out)
{
  return ora_iq_convert_q(family, container, big_endian, swap_iq, bits, rate, bytes, n_bytes, out, max_out, 0);
}

long long ora_iq_convert_q(int family, int container, int big_endian, int swap_iq, int bits, int rate, const uint8_t *bytes,
                           long long n_bytes, float *out, long long max_out, int quirks)
{
  if (quirks && uff_quirk_applies(family, container, big_endian, swap_iq)) {
    /* decode read block by read block (readSamples, xml_reader.cpp:224-227: convBufferSize = rate / 1000 samples) into a
     * cf32 stream in "as read" order, then let the normal path below resample it as already-decoded float32 IQ */
    const int nb0 = container == C_I24 ? 3 : 4, M0 = (int16_t)(rate / 1000);
    const long long blocks = n_bytes / (2LL * nb0 * M0);
    float *tmp = (float *)malloc(sizeof(float) * 2 * (size_t)(blocks * M0 + 1));
    for (long long c = 0; c < blocks; c++) uff_quirk_block(container, swap_iq, big_endian, bits, bytes + c * 2LL * nb0 * M0, M0, tmp + 2 * c * M0);
    /* host-order float32, no swap: exactly the decoded values */
    const union { uint32_t u; uint8_t b[4]; } probe = {1u};
    const long long got = ora_iq_convert_q(FAM_UFF, C_F32, probe.b[0] ? 0 : 1, 0, 32, rate, (const uint8_t *)tmp, blocks * M0 * 8LL, out, max_out, 0);
    free(tmp);
    return got;
  }
  const int nb = container == C_I16 ? 2 : container == C_I24 ? 3 : (container == C_I32 || container == C_F32) ? 4 : 1;
  const long long n = n_bytes / (2 * nb);
  long long produced = 0;
  if (rate == 2048000) {
    for (long long i = 0; i < n && produced < max_out; i++, produced++) {
      const float a = one_channel(bytes + i * 2 * nb, family, container, big_endian, bits);
      const float b = one_channel(bytes + i * 2 * nb + nb, family, container, big_endian, bits);
      out[2 * produced] = swap_iq ? b : a; out[2 * produced + 1] = swap_iq ? a : b;
    }
    return produced;
  }
  int16_t tab_int[2048];
  float tab_frac[2048];
  const int M = (int16_t)(rate / 1000);
  for (int i = 0; i < 2048; i++) {
    if (family == FAM_WAV) {                                  /* wav_reader.cpp:76-82 */
      const float in_val = (float)rate / 1000.0f;
      tab_int[i] = (int16_t)floorf((float)i * (in_val / 2048.0f));
      tab_frac[i] = (float)i * (in_val / 2048.0f) - (float)tab_int[i];
    } else {                                                  /* xml_reader.cpp:76-81 */
      const float in_val = (float)(rate / 1000);
      tab_int[i] = (int16_t)floor(i * (in_val / 2048.0));
      tab_frac[i] = i * (in_val / 2048.0f) - tab_int[i];
    }
  }
  float *conv = (float *)calloc((size_t)(M + 1) * 2, sizeof(float));
  int idx = family == FAM_WAV ? 0 : 1;                        /* wav_reader.cpp:83 mConvIndex = 0; xml_reader.cpp:226 &convBuffer[1] */
  for (long long i = 0; i < n; i++) {
    const float a = one_channel(bytes + i * 2 * nb, family, container, big_endian, bits);
    const float b = one_channel(bytes + i * 2 * nb + nb, family, container, big_endian, bits);
    conv[2 * idx] = swap_iq ? b : a; conv[2 * idx + 1] = swap_iq ? a : b;
    idx++;
    if (idx > M) {
      if (produced + 2048 > max_out) break;
      for (int j = 0; j < 2048; j++) {
        const int base = tab_int[j];
        const float r = tab_frac[j];
        out[2 * (produced + j)] = conv[2 * (base + 1)] * r + conv[2 * base] * (1.0f - r);
        out[2 * (produced + j) + 1] = conv[2 * (base + 1) + 1] * r + conv[2 * base + 1] * (1.0f - r);
      }
      produced += 2048;
      conv[0] = conv[2 * M]; conv[1] = conv[2 * M + 1];
      idx = 1;
    }
  }
  free(conv);
  return produced;
}
