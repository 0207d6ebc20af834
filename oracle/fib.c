/* fib.c -- FIB/FIG subset (oracle; test infrastructure only; PARITY UNPINNED: FibDecoder is Qt-entangled).
 * Restates, on the reference's own one-bit-per-byte representation, decoder/fib_decoder.cpp:59-110 (FIG walk),
 * fib_decoder_fig0.cpp:89-101 (FIG 0/0), :142-224 (FIG 0/1), :230-293 (FIG 0/2), fib_table.h:44-117. */
#include "dab_oracle.h"
#include <string.h>

static unsigned gb(const uint8_t *d, int off, int n)      /* bit_extractors.h getBits */
{
  unsigned v = 0;
  for (int i = 0; i < n; i++) v = (v << 1) | (d[off + i] & 1);
  return v;
}

/* fib_table.h:44-117 : {CU size, protection level, bit rate} per short-form table index */
static const short prot_tab[64][3] = {
  {16,5,32},{21,4,32},{24,3,32},{29,2,32},{35,1,32},{24,5,48},{29,4,48},{35,3,48},{42,2,48},{52,1,48},{29,5,56},{35,4,56},
  {42,3,56},{52,2,56},{32,5,64},{42,4,64},{48,3,64},{58,2,64},{70,1,64},{40,5,80},{52,4,80},{58,3,80},{70,2,80},{84,1,80},
  {48,5,96},{58,4,96},{70,3,96},{84,2,96},{104,1,96},{58,5,112},{70,4,112},{84,3,112},{104,2,112},{64,5,128},{84,4,128},
  {96,3,128},{116,2,128},{140,1,128},{80,5,160},{104,4,160},{116,3,160},{140,2,160},{168,1,160},{96,5,192},{116,4,192},
  {140,3,192},{168,2,192},{208,1,192},{116,5,224},{140,4,224},{168,3,224},{208,2,224},{232,1,224},{128,5,256},{168,4,256},
  {192,3,256},{232,2,256},{280,1,256},{160,5,320},{208,4,320},{280,2,320},{192,5,384},{280,3,384},{416,1,384}};

typedef struct { int used; ora_subch_desc d; int ascty; } slot_t;

/* returns the number of sub-channels (in order of first appearance); dab_plus[i] = 1/0/-1 */
int ora_parse_fibs(const uint8_t *fib_bytes, const uint8_t *crc_ok, int n_fibs, ora_subch_desc *out, int *dab_plus, int max_out,
                   int *cif_count)
{
  slot_t tab[64];
  int order[64], n_order = 0;                              /* first-appearance order, fib_decoder.cpp:547-557 */
  memset(tab, 0, sizeof(tab));
  for (int i = 0; i < 64; i++) tab[i].ascty = -1;
  int cif = -1;
  for (int f = 0; f < n_fibs; f++) {
    if (!crc_ok[f]) continue;
    uint8_t b[256];
    for (int i = 0; i < 256; i++) b[i] = (fib_bytes[f * 32 + i / 8] >> (7 - (i & 7))) & 1;
    int processed = 0, restart = 0;
    while (processed < 30 && !restart) {
      const uint8_t *d = b + processed * 8;
      const unsigned type = gb(d, 0, 3), len = gb(d, 3, 5);
      if (type == 7 && len == 0x1F) break;
      if (processed + 1 + (int)len > 30) break;
      if (type == 0 && len >= 1) {
        const unsigned cn = gb(d, 8, 1), pd = gb(d, 10, 1), ext = gb(d, 11, 5);
        if (ext == 0 && len >= 5) cif = (int)(gb(d, 16 + 19, 5) * 250 + gb(d, 16 + 24, 8));
        else if (ext == 1 && cn == 0) {
          int used = 2;
          while (used <= (int)len) {
            int o = used * 8;
            if (used + 3 > (int)len + 1) break;
            ora_subch_desc q;
            memset(&q, 0, sizeof(q));
            q.subch_id = (int)gb(d, o, 6);
            q.cu_start = (int)gb(d, o + 6, 10);
            if (gb(d, o + 16, 1) == 0) {
              const unsigned idx = gb(d, o + 18, 6);
              q.short_form = 1; q.cu_size = prot_tab[idx][0]; q.prot_level = prot_tab[idx][1]; q.kbps = prot_tab[idx][2];
              used += 3;
            } else {
              if (used + 4 > (int)len + 1) break;
              const unsigned option = gb(d, o + 17, 3), lvl = gb(d, o + 20, 2);
              q.cu_size = (int)gb(d, o + 22, 10);
              q.prot_level = (int)lvl;
              if (option == 0) { static const int t[4] = {12, 8, 6, 4}; q.kbps = q.cu_size / t[lvl] * 8; }
              else if (option == 1) { static const int t[4] = {27, 21, 18, 15}; q.kbps = q.cu_size / t[lvl] * 32; q.prot_level += 4; }
              used += 4;
            }
            if (q.cu_start + q.cu_size > 864) { restart = 1; break; }
            if (!tab[q.subch_id].used) {
              for (int k = 0; k < 64 && !restart; k++)
                if (tab[k].used && q.cu_start < tab[k].d.cu_start + tab[k].d.cu_size && tab[k].d.cu_start < q.cu_start + q.cu_size) restart = 1;
              if (restart) break;
              tab[q.subch_id].used = 1; tab[q.subch_id].d = q; order[n_order++] = q.subch_id;
            }
          }
        } else if (ext == 2 && cn == 0) {
          int used = 2;
          while (used <= (int)len) {
            int o = used * 8 + (pd ? 32 : 16);
            if (o / 8 + 1 > (int)len + 1) break;            /* service header runs past the FIG */
            const int ncomp = (int)gb(d, o + 4, 4);
            o += 8;
            for (int c = 0; c < ncomp; c++, o += 16) {
              if ((o + 16) / 8 > (int)len + 1) break;
              if (gb(d, o, 2) == 0) tab[gb(d, o + 8, 6)].ascty = (int)gb(d, o + 2, 6);
            }
            used = o / 8;
          }
        }
      }
      processed += (int)len + 1;
    }
    if (restart) { memset(tab, 0, sizeof(tab)); for (int i = 0; i < 64; i++) tab[i].ascty = -1; cif = -1; n_order = 0; }
  }
  if (cif_count) *cif_count = cif;
  int n = 0;
  for (int i = 0; i < n_order && n < max_out; i++) {
    const int k = order[i];
    out[n] = tab[k].d; dab_plus[n] = tab[k].ascty < 0 ? -1 : (tab[k].ascty == 63); n++;
  }
  return n;
}
