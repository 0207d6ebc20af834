// fib.cpp -- host-side subset of the FIB/FIG parser: sub-channel organisation (FIG 0/1), service components
// (FIG 0/2) and ensemble information (FIG 0/0: CIF counter, change flags), for the CURRENT and the NEXT multiplex
// configuration, with the swap of the two that fib_decoder_fig0.cpp:102-111 makes when the change flags go 3 -> 0.  SURVEY.md 8(f) rank 1: makes the engine self-configuring the way
// EtiGenerator is (eti_handler/eti_generator.cpp:132-134, 335-380): decode every sub-channel announced in the FIC.
// Follows decoder/fib_decoder.cpp:59-110 (FIG walk), fib_decoder_fig0.cpp:142-224 (FIG 0/1), :230-293 (FIG 0/2),
// fib_table.h:44-117 (short-form table) and fib_decoder.cpp:547-557, 673-691 (getters).  Tiny, branchy, per-FIB:
// stays on the host ("OUT OF SCOPE for GPU").
#include "dabx_internal.h"
#include <cstring>
#include <vector>

namespace dabx {

// EN 300 401 table 8 short-form index -> (bit rate, level): rows of the UEP profile table in order (fib_table.h:44-117
// lists the same 64 rows with their size in CU; the size is derived here from the profile itself).
static const int16_t kUepIndex[64][3] = {   // {kbit/s, protection level, CUs}
    {32, 5, 16}, {32, 4, 21}, {32, 3, 24}, {32, 2, 29}, {32, 1, 35}, {48, 5, 24}, {48, 4, 29}, {48, 3, 35},
    {48, 2, 42}, {48, 1, 52}, {56, 5, 29}, {56, 4, 35}, {56, 3, 42}, {56, 2, 52}, {64, 5, 32}, {64, 4, 42},
    {64, 3, 48}, {64, 2, 58}, {64, 1, 70}, {80, 5, 40}, {80, 4, 52}, {80, 3, 58}, {80, 2, 70}, {80, 1, 84},
    {96, 5, 48}, {96, 4, 58}, {96, 3, 70}, {96, 2, 84}, {96, 1, 104}, {112, 5, 58}, {112, 4, 70}, {112, 3, 84},
    {112, 2, 104}, {128, 5, 64}, {128, 4, 84}, {128, 3, 96}, {128, 2, 116}, {128, 1, 140}, {160, 5, 80}, {160, 4, 104},
    {160, 3, 116}, {160, 2, 140}, {160, 1, 168}, {192, 5, 96}, {192, 4, 116}, {192, 3, 140}, {192, 2, 168}, {192, 1, 208},
    {224, 5, 116}, {224, 4, 140}, {224, 3, 168}, {224, 2, 208}, {224, 1, 232}, {256, 5, 128}, {256, 4, 168}, {256, 3, 192},
    {256, 2, 232}, {256, 1, 280}, {320, 5, 160}, {320, 4, 208}, {320, 2, 280}, {384, 5, 192}, {384, 3, 280}, {384, 1, 416}};

static unsigned bits(const uint8_t *b, int off, int n)   // MSB-first bit field of a packed byte array
{
  unsigned v = 0;
  for (int i = 0; i < n; i++) v = (v << 1) | ((b[(off + i) >> 3] >> (7 - ((off + i) & 7))) & 1u);
  return v;
}

// FibConfigFig0 (fib_config_fig0.h) as far as the path needs it: the FIG 0/1 and FIG 0/2 vectors of ONE multiplex configuration.
struct FibConfig {
  std::vector<dabx_subch_desc> subch;       // Fig0s1_BasicSubChannelOrganizationVec: in order of first appearance, first description of a SubChId wins
  struct Comp { uint32_t sid; int idx, tmid, ascty, subch_id; };
  std::vector<Comp> comps;                  // Fig0s2_BasicService_ServiceCompDefVec: first definition of (SId, component index) wins
  void reset() { subch.clear(); comps.clear(); }
  const dabx_subch_desc *find(int id) const { for (auto &q : subch) if (q.subch_id == id) return &q; return nullptr; }
};

}  // namespace dabx

// FibDecoder (fib_decoder.h) as far as the path needs it: the current and the next configuration, the FIG 0/0 scalars and the
// change-flag memory that swaps the two (fib_decoder_fig0.cpp:89-112).
struct dabx_fibdec {
  dabx::FibConfig cfg[2];
  int cur = 0;                              // cfg[cur] = mpFibConfigFig0Curr, cfg[cur ^ 1] = mpFibConfigFig0Next
  int cif_count = -1, cif_hi = -1, cif_lo = -1;
  int change_flags = 0, occurrence = 0, prev_change_flag = 0;
  long long fibs = 0, fig00_fib = -1, last_change_fib = -1;
  int n_changes = 0, n_restarts = 0;
  bool restart = false;                     // mRestartFibDecoding
  bool strict = false;                      // dabx_fibdec_set_reference_quirks: swap only after change flags 3, like fib_decoder_fig0.cpp:103
  void reset_scalars() { cif_count = cif_hi = cif_lo = -1; change_flags = occurrence = prev_change_flag = 0; fig00_fib = -1; }   // FibDecoder::_reset
  void reset_all() { cfg[0].reset(); cfg[1].reset(); reset_scalars(); }
  void restart_decoding() { reset_all(); n_restarts++; restart = true; }            // _restart_fib_decoding, fib_decoder.cpp:131-141
};

namespace dabx {

// one FIB (30 data bytes; the CRC has been checked by the caller, fic_decoder.cpp:234-243): FibDecoder::process_FIB, fib_decoder.cpp:59-106
static void walk_fib(const uint8_t *fib, dabx_fibdec &t)
{
  t.restart = false;
  int p = 0;
  while (p < 30 && !t.restart) {                                     // fib_decoder.cpp:74-103
    const int type = fib[p] >> 5, len = fib[p] & 0x1F;
    if (type == 7 && len == 0x1F) break;
    if (p + 1 + len > 30) break;                                       // FIG runs past the FIB data field
    if (type == 0 && len >= 1) {
      const uint8_t *d = fib + p;
      const int ext = d[1] & 0x1F, pd = (d[1] >> 5) & 1, cn = (d[1] >> 7) & 1;
      FibConfig &cfg = t.cfg[cn == 0 ? t.cur : t.cur ^ 1];             // _get_config_ptr(CN_Flag), fib_decoder.h:97
      if (ext == 0 && len >= 5) {                                      // _process_Fig0s0, fib_decoder_fig0.cpp:89-112
        const int flags = (int)bits(d, 32, 2);
        t.cif_hi = (int)bits(d, 35, 5); t.cif_lo = (int)bits(d, 40, 8);
        t.cif_count = t.cif_hi * 250 + t.cif_lo;
        t.occurrence = len >= 6 ? (int)bits(d, 48, 8) : 0;             // OccurrenceChange: present while the change flags are set
        t.change_flags = flags;
        t.fig00_fib = t.fibs;
        // :103-110: the next configuration becomes the current one.  The reference swaps only after flags 3 (sub-channel AND service
        // organisation): after an announcement with flags 1 or 2 (EN 300 401 6.4.1) its next table is never swapped or reset, and its
        // stale entries ("first description wins") then shape the reconfiguration after that.  By default every announcement that ends
        // (non-zero -> 0) switches; what the announcement did not cover -- a table the next configuration never received an entry for --
        // is carried over from the current one.  dabx_fibdec_set_reference_quirks(1) restores the reference's rule.
        if (flags == 0 && t.prev_change_flag != 0 && (t.prev_change_flag == 3 || !t.strict)) {
          if (!t.strict) {
            FibConfig &nx = t.cfg[t.cur ^ 1];
            if (nx.subch.empty()) nx.subch = t.cfg[t.cur].subch;
            if (nx.comps.empty()) nx.comps = t.cfg[t.cur].comps;
          }
          t.cur ^= 1;
          t.cfg[t.cur ^ 1].reset();
          t.n_changes++;
          t.last_change_fib = t.fibs;
        }
        t.prev_change_flag = flags;
      } else if (ext == 1) {                                           // _subprocess_Fig0s1, :142-224
        int used = 2;
        while (used <= len) {
          const int o = used * 8;
          if (used + 3 > len + 1) break;
          dabx_subch_desc q{};
          q.subch_id = (int)bits(d, o, 6);
          if (const dabx_subch_desc *known = cfg.find(q.subch_id)) {                  // :151-153, 219-223: described already, step over it
            used += known->short_form ? 3 : 4;                                        //   (by the STORED entry's form, like the reference)
            continue;
          }
          q.cu_start = (int)bits(d, o + 6, 10);
          const bool short_form = bits(d, o + 16, 1) == 0;
          if (short_form) {
            const int idx = (int)bits(d, o + 18, 6);
            q.short_form = 1; q.kbps = kUepIndex[idx][0]; q.prot_level = kUepIndex[idx][1];
            q.cu_size = kUepIndex[idx][2];   // from the table, as the reference does (its 80 kbit/s level-1 puncturing row disagrees)
            used += 3;
          } else {
            if (used + 4 > len + 1) break;
            const int option = (int)bits(d, o + 17, 3), lvl = (int)bits(d, o + 20, 2);
            q.cu_size = (int)bits(d, o + 22, 10);
            q.short_form = 0; q.prot_level = lvl;
            if (option == 0) { static const int tab[4] = {12, 8, 6, 4}; q.kbps = q.cu_size / tab[lvl] * 8; }
            else if (option == 1) { static const int tab[4] = {27, 21, 18, 15}; q.kbps = q.cu_size / tab[lvl] * 32; q.prot_level += 4; }
            else q.kbps = 0;
            used += 4;
          }
          if (q.cu_start + q.cu_size > 864) { t.restart_decoding(); break; }           // :198-202
          bool collide = false;
          for (auto &k : cfg.subch)                                                  // :204-209 overlap -> restart
            if (q.cu_start < k.cu_start + k.cu_size && k.cu_start < q.cu_start + q.cu_size) collide = true;
          if (collide) { t.restart_decoding(); break; }
          cfg.subch.push_back(q);
        }
      } else if (ext == 2) {                                           // _subprocess_Fig0s2, :230-293
        int used = 2;
        while (used <= len) {
          int o = used * 8;
          if ((o + (pd ? 32 : 16)) / 8 + 1 > len + 1) break;  // service header (SId + component count) runs past the FIG
          const uint32_t sid = bits(d, o, pd ? 32 : 16);
          o += pd ? 32 : 16;
          const int ncomp = (int)bits(d, o + 4, 4);
          o += 8;
          for (int c = 0; c < ncomp; c++, o += 16) {
            if ((o + 16) / 8 > len + 1) break;
            bool known = false;
            for (auto &k : cfg.comps) if (k.sid == sid && k.idx == c) known = true;
            if (known) continue;
            FibConfig::Comp k{sid, c, (int)bits(d, o, 2), -1, -1};
            if (k.tmid == 0) { k.ascty = (int)bits(d, o + 2, 6); k.subch_id = (int)bits(d, o + 8, 6); }
            else if (k.tmid == 1) k.subch_id = (int)bits(d, o + 8, 6);
            cfg.comps.push_back(k);
          }
          used = o / 8;
        }
      }
    }
    p += len + 1;
  }
  t.fibs++;
}

static int table_out(const FibConfig &cfg, dabx_subch_desc *out, int max_out)
{
  int n = 0;
  for (const auto &s : cfg.subch) {
    if (n >= max_out) break;
    dabx_subch_desc q = s;
    q.dab_plus = -1;                                                  // ASCTy 63 = DAB+ (backend_driver.cpp:41-50); -1 = no FIG 0/2 for it yet
    for (const auto &k : cfg.comps) if (k.tmid == 0 && k.subch_id == q.subch_id) { q.dab_plus = k.ascty == 63 ? 1 : 0; break; }
    out[n++] = q;
  }
  return n;
}

// FIG 0/0 of one FIB -> CIF counter halves (mCifCount_hi / _lo, fib_decoder_fig0.cpp:89-101); false if the FIB has none
bool fib_cif_count(const uint8_t *fib, int *hi, int *lo)
{
  bool found = false;
  int p = 0;
  while (p < 30) {
    const int type = fib[p] >> 5, len = fib[p] & 0x1F;
    if (type == 7 && len == 0x1F) break;
    if (p + 1 + len > 30) break;
    if (type == 0 && len >= 5 && (fib[p + 1] & 0x1F) == 0) { *hi = fib[p + 4] & 0x1F; *lo = fib[p + 5]; found = true; }
    p += len + 1;
  }
  return found;
}

}  // namespace dabx

using namespace dabx;

namespace dabx {
void dabx_internal_fibdec_skip(dabx_fibdec *d, long long n_fibs) { if (d && n_fibs > 0) d->fibs += n_fibs; }
}

extern "C" {

int dabx_fibdec_create(dabx_fibdec **out)
{
  if (!out) { set_error("dabx_fibdec_create: bad argument"); return DABX_E_ARG; }
  *out = new dabx_fibdec();
  return 0;
}
void dabx_fibdec_destroy(dabx_fibdec *d) { delete d; }
int dabx_fibdec_reset(dabx_fibdec *d)            // FibDecoder::connect_channel, fib_decoder.cpp:143-150
{
  if (!d) return DABX_E_ARG;
  const bool strict = d->strict;
  *d = dabx_fibdec();
  d->strict = strict;
  return 0;
}
int dabx_fibdec_set_reference_quirks(dabx_fibdec *d, int on)
{
  if (!d) return DABX_E_ARG;
  d->strict = on != 0;
  return 0;
}
int dabx_fibdec_process(dabx_fibdec *d, const uint8_t *fibs, const uint8_t *crc_ok, int n_fibs)
{
  if (!d || !fibs || !crc_ok || n_fibs < 0) { set_error("dabx_fibdec_process: bad argument"); return DABX_E_ARG; }
  const int before = d->n_changes;
  for (int i = 0; i < n_fibs; i++) {
    if (crc_ok[i]) walk_fib(fibs + (size_t)i * 32, *d);                // fic_decoder.cpp:234-243: only FIBs that pass their CRC reach process_FIB
    else d->fibs++;
  }
  return d->n_changes - before;
}
int dabx_fibdec_get_info(const dabx_fibdec *d, dabx_fibdec_info *out)
{
  if (!d || !out) return DABX_E_ARG;
  memset(out, 0, sizeof(*out));
  out->fibs_processed = d->fibs; out->cif_count = d->cif_count; out->change_flags = d->change_flags;
  out->occurrence_change = d->occurrence; out->fig00_fib = d->fig00_fib; out->n_changes = d->n_changes;
  out->last_change_fib = d->last_change_fib; out->n_restarts = d->n_restarts;
  out->cif_count_hi = d->cif_hi; out->cif_count_lo = d->cif_lo;
  return 0;
}
int dabx_fibdec_subchannels(const dabx_fibdec *d, int next, dabx_subch_desc *out, int max_out)
{
  if (!d || (!out && max_out > 0) || max_out < 0) return DABX_E_ARG;
  return table_out(d->cfg[next ? d->cur ^ 1 : d->cur], out, max_out);
}

// fibs: n x 32 bytes, crc_ok: n flags.  Returns the number of sub-channels of the CURRENT configuration written to out (in order of
// first appearance, like FibDecoder::get_sub_channel_id_list, fib_decoder.cpp:547-557),
// dab_plus = 1 when FIG 0/2 announces ASCTy 63 for it (backend_driver.cpp:41-50), -1 when unknown yet.  One-shot form of
// dabx_fibdec_process on a fresh decoder.
int dabx_parse_fibs(const uint8_t *fibs, const uint8_t *crc_ok, int n_fibs, dabx_subch_desc *out, int max_out, int32_t *cif_count)
{
  if (!fibs || !crc_ok || n_fibs < 0 || (!out && max_out > 0)) { set_error("dabx_parse_fibs: bad argument"); return DABX_E_ARG; }
  dabx_fibdec t;
  for (int i = 0; i < n_fibs; i++) if (crc_ok[i]) walk_fib(fibs + (size_t)i * 32, t); else t.fibs++;
  if (cif_count) *cif_count = t.cif_count;
  return table_out(t.cfg[t.cur], out, max_out);
}

}  // extern "C"
