/* fib.c -- FIB/FIG subset (oracle; test infrastructure only; PARITY UNPINNED: FibDecoder is Qt-entangled).
 * Restates, on the reference's own one-bit-per-byte representation, decoder/fib_decoder.cpp:59-110 (FIG walk),
 * fib_decoder_fig0.cpp:89-112 (FIG 0/0 incl. the change-flag swap of the current and the next configuration), :142-224
 * (FIG 0/1), :230-293 (FIG 0/2), both filed under _get_config_ptr(C/N) (fib_decoder.h:97), fib_table.h:44-117. */
#include "dab_oracle.h"
#include <stdlib.h>
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

/* FibConfigFig0 subset (fib_config_fig0.h): FIG 0/1 and FIG 0/2 vectors of one multiplex configuration */
typedef struct { unsigned sid; int idx, tmid, ascty, subch; } comp_t;
typedef struct {
  ora_subch_desc sc[64]; int n_sc;          /* Fig0s1_BasicSubChannelOrganizationVec (first description of a SubChId wins) */
  comp_t comp[256]; int n_comp;             /* Fig0s2_BasicService_ServiceCompDefVec (first definition of (SId, index) wins) */
} cfg_t;

/* FibDecoder subset (fib_decoder.h:60-100): mpFibConfigFig0Curr / Next, mCifCount*, mPrevChangeFlag, mRestartFibDecoding */
struct ora_fibdec {
  cfg_t cfg[2]; int cur;
  int cif_count, cif_hi, cif_lo, change_flags, occurrence, prev_change_flag;
  long long fibs, fig00_fib, last_change_fib;
  int n_changes, n_restarts, restart;
};

static void fd_reset_scalars(ora_fibdec *t)   /* FibDecoder::_reset, fib_decoder.cpp:108-126 (counters: -1 = none yet instead of 0) */
{
  t->cif_count = t->cif_hi = t->cif_lo = -1; t->change_flags = t->occurrence = t->prev_change_flag = 0; t->fig00_fib = -1;
}
static void fd_restart(ora_fibdec *t)         /* _restart_fib_decoding, fib_decoder.cpp:131-141 */
{
  t->cfg[0].n_sc = t->cfg[0].n_comp = t->cfg[1].n_sc = t->cfg[1].n_comp = 0;
  fd_reset_scalars(t);
  t->n_restarts++; t->restart = 1;
}
ora_fibdec *ora_fibdec_new(void)
{
  ora_fibdec *t = (ora_fibdec *)calloc(1, sizeof(*t));
  fd_reset_scalars(t);
  t->last_change_fib = -1;
  return t;
}
void ora_fibdec_free(ora_fibdec *t) { free(t); }

/* FibDecoder::process_FIB (fib_decoder.cpp:59-106) for one FIB that passed its CRC, on the reference's one-bit-per-byte form */
static void fd_process_fib(ora_fibdec *t, const uint8_t *fib32)
{
  uint8_t b[256 + 64];
  memset(b, 0, sizeof(b));
  for (int i = 0; i < 256; i++) b[i] = (fib32[i / 8] >> (7 - (i & 7))) & 1;
  int processed = 0;
  t->restart = 0;                                              /* :72 */
  while (processed < 30 && !t->restart) {
    const uint8_t *d = b + processed * 8;
    const unsigned type = gb(d, 0, 3), len = gb(d, 3, 5);
    if (type == 7 && len == 0x1F) break;                       /* :83-86 */
    if (processed + 1 + (int)len > 30) break;
    if (type == 0 && len >= 1) {
      const unsigned cn = gb(d, 8, 1), pd = gb(d, 10, 1), ext = gb(d, 11, 5);      /* _get_fig_header, :1023-1040 */
      cfg_t *cfg = &t->cfg[cn == 0 ? t->cur : t->cur ^ 1];     /* _get_config_ptr, fib_decoder.h:97 */
      if (ext == 0 && len >= 5) {                              /* _process_Fig0s0, fib_decoder_fig0.cpp:89-112 */
        const int flags = (int)gb(d, 16 + 16, 2);
        t->cif_hi = (int)gb(d, 16 + 19, 5); t->cif_lo = (int)gb(d, 16 + 24, 8);
        t->cif_count = t->cif_hi * 250 + t->cif_lo;
        t->occurrence = len >= 6 ? (int)gb(d, 16 + 32, 8) : 0;
        t->change_flags = flags;
        t->fig00_fib = t->fibs;
        if (flags == 0 && t->prev_change_flag == 3) {          /* :103-110 std::swap(curr, next); next->reset() */
          t->cur ^= 1;
          t->cfg[t->cur ^ 1].n_sc = t->cfg[t->cur ^ 1].n_comp = 0;
          t->n_changes++;
          t->last_change_fib = t->fibs;
        }
        t->prev_change_flag = flags;                           /* :112 */
      } else if (ext == 1) {                                   /* _subprocess_Fig0s1, :142-224 */
        int used = 2;
        while (used <= (int)len && !t->restart) {
          int o = used * 8;
          if (used + 3 > (int)len + 1) break;
          ora_subch_desc q;
          memset(&q, 0, sizeof(q));
          q.subch_id = (int)gb(d, o, 6);
          const ora_subch_desc *known = NULL;
          for (int k = 0; k < cfg->n_sc; k++) if (cfg->sc[k].subch_id == q.subch_id) known = &cfg->sc[k];
          if (known) { used += known->short_form ? 3 : 4; continue; }              /* :219-223 */
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
            if (option == 0) { static const int tt[4] = {12, 8, 6, 4}; q.kbps = q.cu_size / tt[lvl] * 8; }
            else if (option == 1) { static const int tt[4] = {27, 21, 18, 15}; q.kbps = q.cu_size / tt[lvl] * 32; q.prot_level += 4; }
            used += 4;
          }
          if (q.cu_start + q.cu_size > 864) { fd_restart(t); break; }               /* :198-202 */
          int collide = 0;
          for (int k = 0; k < cfg->n_sc; k++)                                       /* :204-209 */
            if (q.cu_start < cfg->sc[k].cu_start + cfg->sc[k].cu_size && cfg->sc[k].cu_start < q.cu_start + q.cu_size) collide = 1;
          if (collide) { fd_restart(t); break; }
          if (cfg->n_sc < 64) cfg->sc[cfg->n_sc++] = q;
        }
      } else if (ext == 2) {                                   /* _subprocess_Fig0s2, :230-293 */
        int used = 2;
        while (used <= (int)len) {
          int o = used * 8;
          if ((o + (pd ? 32 : 16)) / 8 + 1 > (int)len + 1) break;                   /* service header runs past the FIG */
          const unsigned sid = gb(d, o, pd ? 32 : 16);
          o += pd ? 32 : 16;
          const int ncomp = (int)gb(d, o + 4, 4);
          o += 8;
          for (int c = 0; c < ncomp; c++, o += 16) {
            if ((o + 16) / 8 > (int)len + 1) break;
            int seen = 0;
            for (int k = 0; k < cfg->n_comp; k++) if (cfg->comp[k].sid == sid && cfg->comp[k].idx == c) seen = 1;
            if (seen) continue;
            comp_t k = {sid, c, (int)gb(d, o, 2), -1, -1};
            if (k.tmid == 0) { k.ascty = (int)gb(d, o + 2, 6); k.subch = (int)gb(d, o + 8, 6); }
            else if (k.tmid == 1) k.subch = (int)gb(d, o + 8, 6);
            if (cfg->n_comp < 256) cfg->comp[cfg->n_comp++] = k;
          }
          used = o / 8;
        }
      }
    }
    processed += (int)len + 1;
  }
  t->fibs++;
}

int ora_fibdec_process(ora_fibdec *t, const uint8_t *fib_bytes, const uint8_t *crc_ok, int n_fibs)
{
  const int before = t->n_changes;
  for (int f = 0; f < n_fibs; f++) {
    if (crc_ok[f]) fd_process_fib(t, fib_bytes + (size_t)f * 32);
    else t->fibs++;
  }
  return t->n_changes - before;
}

/* info[0..9] = fibs, fig00_fib, last_change_fib, cif_count, cif_hi, cif_lo, change_flags, occurrence, n_changes, n_restarts */
void ora_fibdec_info(const ora_fibdec *t, long long info[10])
{
  info[0] = t->fibs; info[1] = t->fig00_fib; info[2] = t->last_change_fib; info[3] = t->cif_count; info[4] = t->cif_hi;
  info[5] = t->cif_lo; info[6] = t->change_flags; info[7] = t->occurrence; info[8] = t->n_changes; info[9] = t->n_restarts;
}

int ora_fibdec_subchannels(const ora_fibdec *t, int next, ora_subch_desc *out, int *dab_plus, int max_out)
{
  const cfg_t *cfg = &t->cfg[next ? t->cur ^ 1 : t->cur];
  int n = 0;
  for (int i = 0; i < cfg->n_sc && n < max_out; i++) {
    out[n] = cfg->sc[i];
    dab_plus[n] = -1;
    for (int k = 0; k < cfg->n_comp; k++)
      if (cfg->comp[k].tmid == 0 && cfg->comp[k].subch == cfg->sc[i].subch_id) { dab_plus[n] = cfg->comp[k].ascty == 63; break; }
    n++;
  }
  return n;
}

/* one-shot form: returns the number of sub-channels of the current configuration (in order of first appearance); dab_plus[i] = 1/0/-1 */
int ora_parse_fibs(const uint8_t *fib_bytes, const uint8_t *crc_ok, int n_fibs, ora_subch_desc *out, int *dab_plus, int max_out,
                   int *cif_count)
{
  ora_fibdec *t = ora_fibdec_new();
  ora_fibdec_process(t, fib_bytes, crc_ok, n_fibs);
  if (cif_count) *cif_count = t->cif_count;
  const int n = ora_fibdec_subchannels(t, 0, out, dab_plus, max_out);
  ora_fibdec_free(t);
  return n;
}
