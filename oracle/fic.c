/* fic.c -- FIC decoder (oracle; test infrastructure only).
 * Restates decoder/fic_decoder.cpp:55-262 plus the part of the FIB walk that feeds
 * back into the OFDM path (CIF counter, decoder/fib_decoder.cpp:59-110,
 * fib_decoder_fig0.cpp:36-101). */
#include "dab_oracle.h"
#include <string.h>

void ora_fic_init(ora_fic *f)
{
  memset(f, 0, sizeof(*f));
  ora_fic_map(f->map);
  for (int i = 0; i < 3096; i++) f->punct[i] = f->map[i] >= 0;
  ora_prbs(f->prbs, 768);
}

static unsigned get_bits(const uint8_t *d, int off, int n)
{
  unsigned v = 0;
  for (int i = 0; i < n; i++) v = (v << 1) | (d[off + i] & 1);
  return v;
}

/* decoder/fib_decoder.cpp:59-110 : walk the FIGs of one FIB (bits, one per byte); only
 * FIG 0/0 (fib_decoder_fig0.cpp:89-101) has an effect on the hot path. */
static void fib_walk(ora_fic *f, const uint8_t *fib_bits)
{
  int processed = 0;
  while (processed < 30) {
    const uint8_t *d = fib_bits + processed * 8;
    const unsigned type = get_bits(d, 0, 3), len = get_bits(d, 3, 5);
    if (type == 7 && len == 0x1F) break;
    if (type == 0) {
      const unsigned ext = get_bits(d, 8 + 3, 5);
      if (ext == 0) {
        f->cif_hi = (int)get_bits(d, 16 + 19, 5);
        f->cif_lo = (int)get_bits(d, 16 + 24, 8);
        f->cif_count = f->cif_hi * 250 + f->cif_lo;
      }
    }
    processed += (int)len + 1;
  }
}

/* decoder/fic_decoder.cpp:178-262 */
static void fic_process_input(ora_fic *f, int fic_idx)
{
  /* :188-192 depuncture; punctured positions stay 0 (array zeroed once, :76 of the header) */
  for (int i = 0; i < 3096; i++)
    if (f->map[i] >= 0) f->vit_in[i] = f->soft[f->map[i]];
  uint8_t *bits = &f->fib_bits[fic_idx * 768];
  ora_viterbi_build(f->vit_in, 768, bits);                                         /* :197 */
  ora_viterbi_ber(f->vit_in, f->punct, bits, 768, &f->fic_bits, &f->fic_errors); /* :199 */
  if (++f->fic_block == 40) { f->fic_block = 0; f->fic_errors /= 2; f->fic_bits /= 2; } /* :201-210 */
  for (int i = 0; i < 768; i++) bits[i] ^= f->prbs[i];                       /* :219-222 */
  f->fic_valid[fic_idx] = 1;
  for (int k = 0; k < 3; k++) {                                              /* :234-261 */
    const uint8_t *fib = bits + 256 * k;
    const int ok = ora_check_crc_bits(fib, 256);
    f->fib_crc[fic_idx * 3 + k] = (uint8_t)ok;
    if (ok) {
      fib_walk(f, fib);
      if (f->success_ratio < 10) f->success_ratio++;
    } else {
      f->fic_valid[fic_idx] = 0;
      if (f->success_ratio > 0) f->success_ratio--;
    }
  }
}

/* decoder/fic_decoder.cpp:143-167 */
void ora_fic_process_block(ora_fic *f, const int16_t soft[ORA_2K], int sym_idx)
{
  if (sym_idx == 1) { f->index = 0; f->fic_idx = 0; }
  for (int i = 0; i < ORA_2K; i++) {
    f->soft[f->index++] = soft[i];
    if (f->index >= ORA_FIC_IN) {
      fic_process_input(f, f->fic_idx);
      f->index = 0;
      f->fic_idx++;
    }
  }
}
