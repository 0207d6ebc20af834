/* msc.c -- MSC back end: time de-interleaver, deconvolution, energy de-dispersal,
 * DAB+ super-frame sync + RS (oracle; test infrastructure only).
 * Restates backend/backend.cpp:38-161 and backend/audio/mp4processor.cpp:96-333. */
#include "dab_oracle.h"
#include <stdlib.h>
#include <string.h>

static const int16_t kInterleaveMap[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15}; /* backend.cpp:129 */
const int16_t *ora_interleave_map(void) { return kInterleaveMap; }      /* (tests: compared with the reference object's own table, eti_generator.cpp:22) */

static void sink_append(uint8_t **buf, size_t *len, size_t *cap, const uint8_t *src, size_t n)
{
  if (*len + n > *cap) {
    size_t nc = *cap ? *cap * 2 : 4096;
    while (nc < *len + n) nc *= 2;
    *buf = (uint8_t *)realloc(*buf, nc);
    *cap = nc;
  }
  memcpy(*buf + *len, src, n);
  *len += n;
}

int ora_backend_init(ora_backend *b, const ora_subch_desc *d)
{
  memset(b, 0, sizeof(*b));
  b->d = *d;
  b->frag = d->cu_size * 64;                               /* backend.cpp:50 */
  b->hist = (int16_t *)calloc((size_t)16 * b->frag, sizeof(int16_t));
  b->tmp = (int16_t *)calloc((size_t)b->frag, sizeof(int16_t));
  b->map = (int32_t *)malloc(sizeof(int32_t) * (size_t)(96 * d->kbps + 24));
  b->prbs = (uint8_t *)malloc((size_t)24 * d->kbps);
  b->outv = (uint8_t *)calloc((size_t)24 * d->kbps, 1);
  const int n_in = d->short_form ? ora_uep_map(d->kbps, d->prot_level, b->map)   /* backend_deconvolver.cpp:36-46 */
                                 : ora_eep_map(d->kbps, d->prot_level, b->map);
  if (n_in < 0 || n_in > b->frag) return -1;
  ora_prbs(b->prbs, 24 * d->kbps);                         /* backend.cpp:72-84 */
  b->rs_dims = d->kbps / 8;                                /* mp4processor.cpp:62 */
  b->frame_bytes = (uint8_t *)calloc((size_t)b->rs_dims * 120, 1);
  b->out_vec = (uint8_t *)calloc((size_t)b->rs_dims * 110 + 16, 1);
  return 0;
}

void ora_backend_free(ora_backend *b)
{
  free(b->hist); free(b->tmp); free(b->map); free(b->prbs); free(b->outv);
  free(b->frame_bytes); free(b->out_vec); free(b->msc_bytes); free(b->sf_bytes); free(b->sfi_bytes);
  memset(b, 0, sizeof(*b));
}

/* mp4processor.cpp:184-241 */
static int process_rs_frame(ora_backend *b, int base)
{
  const int R = b->rs_dims;
  for (int j = 0; j < R; j++) {
    uint8_t in[120], out[110];
    for (int k = 0; k < 120; k++) in[k] = b->frame_bytes[(base + j + k * R) % (R * 120)];
    const int ler = ora_rs_dec(in, out);
    if (ler < 0) b->n_rs_fail++; else b->n_rs_corr += ler;
    for (int k = 0; k < 110; k++) b->out_vec[j + k * R] = out[k];
  }
  if (ora_firecode_check_and_correct(b->out_vec)) {
    if (memcmp(b->out_vec, &b->frame_bytes[base], 11)) b->n_fc_corr++;
    return 1;
  }
  return 0;
}

/* mp4processor.cpp:249-333 : AU table + AU CRCs (the AAC decode after it is out of scope) */
static int process_super_frame(ora_backend *b, int base)
{
  const long corr0 = b->n_rs_corr, fail0 = b->n_rs_fail, fc0 = b->n_fc_corr;
  if (!process_rs_frame(b, base)) return 0;
  const uint8_t *o = b->out_vec;
  const int dac = (o[2] >> 6) & 1, sbr = (o[2] >> 5) & 1;
  int au[7], n_au;
  const int end = 110 * (b->d.kbps / 8);
  switch (2 * dac + sbr) {
  case 0: n_au = 4; au[0] = 8; au[1] = o[3] * 16 + (o[4] >> 4); au[2] = (o[4] & 0xf) * 256 + o[5];
          au[3] = o[6] * 16 + (o[7] >> 4); au[4] = end; break;
  case 1: n_au = 2; au[0] = 5; au[1] = o[3] * 16 + (o[4] >> 4); au[2] = end; break;
  case 2: n_au = 6; au[0] = 11; au[1] = o[3] * 16 + (o[4] >> 4); au[2] = (o[4] & 0xf) * 256 + o[5];
          au[3] = o[6] * 16 + (o[7] >> 4); au[4] = (o[7] & 0xf) * 256 + o[8];
          au[5] = o[9] * 16 + (o[10] >> 4); au[6] = end; break;
  default: n_au = 3; au[0] = 6; au[1] = o[3] * 16 + (o[4] >> 4); au[2] = (o[4] & 0xf) * 256 + o[5];
          au[3] = end; break;
  }
  /* the super frame's record as include/dabx.h (dabx_superframe_info, 32 bytes little-endian) lays it out: what
   * _process_super_frame knows about the super frame when it hands the access units on (:256-333) */
  uint8_t rec[32];
  memset(rec, 0, sizeof rec);
  rec[0] = (uint8_t)n_au;
  rec[3] = (uint8_t)(o[2] & 0x7F);                         /* dacRate, sbrFlag, aacChannelMode, psFlag, mpegSurround (:258-262) */
  for (int i = 0; i <= n_au; i++) { rec[4 + 2 * i] = (uint8_t)(au[i] & 0xFF); rec[5 + 2 * i] = (uint8_t)(au[i] >> 8); }
  for (int i = 0; i < n_au; i++) {
    const int len = au[i + 1] - au[i] - 2;
    if (len > 960 || len < 0 || au[i] + len + 2 > end) { b->n_au_bad++; rec[2] |= (uint8_t)(1u << i); continue; }
    if (ora_check_crc_bytes(&o[au[i]], len)) { b->n_au_ok++; rec[1] |= (uint8_t)(1u << i); } else b->n_au_bad++;
  }
  {
    const long corr = b->n_rs_corr - corr0, fail = b->n_rs_fail - fail0;
    const long long first = (long long)b->n_cif_out - 5;   /* the oldest of the five logical frames (0 = the slot's first) */
    rec[18] = (uint8_t)(corr & 0xFF); rec[19] = (uint8_t)((corr >> 8) & 0xFF);
    rec[20] = (uint8_t)fail;
    rec[21] = (uint8_t)(b->n_fc_corr - fc0);
    for (int i = 0; i < 8; i++) rec[24 + i] = (uint8_t)(((unsigned long long)first >> (8 * i)) & 0xFF);
  }
  sink_append(&b->sfi_bytes, &b->sfi_len, &b->sfi_cap, rec, sizeof rec);
  sink_append(&b->sf_bytes, &b->sf_len, &b->sf_cap, b->out_vec, (size_t)end);
  return 1;
}

/* mp4processor.cpp:96-182 */
static void mp4_add_to_frame(ora_backend *b, const uint8_t *bits)
{
  const int nbytes = 24 * b->d.kbps / 8;
  for (int i = 0; i < nbytes; i++) {
    uint8_t t = 0;
    for (int j = 0; j < 8; j++) t = (uint8_t)((t << 1) | (bits[i * 8 + j] & 1));
    b->frame_bytes[b->block_fill * nbytes + i] = t;
  }
  sink_append(&b->msc_bytes, &b->msc_len, &b->msc_cap, &b->frame_bytes[b->block_fill * nbytes], (size_t)nbytes);
  b->blocks_in_buf++;
  b->block_fill = (b->block_fill + 1) % 5;
  if (b->blocks_in_buf >= 5) {
    if (b->sf_sync == 0) {
      if (ora_firecode_check(&b->frame_bytes[b->block_fill * nbytes])) b->sf_sync = 4;
      else b->blocks_in_buf = 4;
    }
    if (b->sf_sync) {
      b->blocks_in_buf = 0;
      if (process_super_frame(b, b->block_fill * nbytes)) { b->sf_sync = 4; b->n_sf_ok++; }
      else {
        b->sf_sync--;
        if (b->sf_sync == 0) { b->blocks_in_buf = 4; b->n_sf_fail++; }
      }
    }
  }
}

/* backend/backend.cpp:131-161 */
void ora_backend_process(ora_backend *b, const int16_t *in)
{
  const int F = b->frag;
  for (int i = 0; i < F; i++) {
    b->tmp[i] = b->hist[(size_t)((b->idx + kInterleaveMap[i & 15]) & 15) * F + i];
    b->hist[(size_t)b->idx * F + i] = in[i];
  }
  b->idx = (b->idx + 1) & 15;
  if (b->cnt <= 15) { b->cnt++; return; }
  ora_deconvolve(b->tmp, b->map, b->d.kbps, b->outv);
  for (int i = 0; i < 24 * b->d.kbps; i++) b->outv[i] ^= b->prbs[i];
  b->n_cif_out++;
  mp4_add_to_frame(b, b->outv);
}

/* accessors for ctypes-based tests */
const uint8_t *ora_backend_msc_bytes(const ora_backend *b, size_t *len) { *len = b->msc_len; return b->msc_bytes; }
const uint8_t *ora_backend_sf_bytes(const ora_backend *b, size_t *len) { *len = b->sf_len; return b->sf_bytes; }
const uint8_t *ora_backend_sfi_bytes(const ora_backend *b, size_t *len) { *len = b->sfi_len; return b->sfi_bytes; }
void ora_backend_stats(const ora_backend *b, long out[8])
{
  out[0] = b->n_cif_out; out[1] = b->n_sf_ok; out[2] = b->n_sf_fail; out[3] = b->n_rs_corr;
  out[4] = b->n_rs_fail; out[5] = b->n_fc_corr; out[6] = b->n_au_ok; out[7] = b->n_au_bad;
}
