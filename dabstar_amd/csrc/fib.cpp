// fib.cpp -- host-side subset of the FIB/FIG parser: sub-channel organisation (FIG 0/1), service components
// (FIG 0/2) and the CIF counter (FIG 0/0).  SURVEY.md 8(f) rank 1: makes the engine self-configuring the way
// EtiGenerator is (eti_handler/eti_generator.cpp:132-134, 335-380): decode every sub-channel announced in the FIC.
// Follows decoder/fib_decoder.cpp:59-110 (FIG walk), fib_decoder_fig0.cpp:142-224 (FIG 0/1), :230-293 (FIG 0/2),
// fib_table.h:44-117 (short-form table) and fib_decoder.cpp:547-557, 673-691 (getters).  Tiny, branchy, per-FIB:
// stays on the host ("OUT OF SCOPE for GPU").
#include "dabx_internal.h"
#include <cstring>
#include <map>

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

struct FibTable {
  std::map<int, dabx_subch_desc> subch;     // by SubChId, first description wins (fib_decoder_fig0.cpp:151-153)
  std::vector<int> order;                   // SubChIds in order of first appearance (the reference's vector order)
  std::map<int, int> ascty;                 // SubChId -> ASCTy of its audio component (FIG 0/2, TMId 0)
  int cif_count = -1;
  bool restart = false;
};

// one FIB (30 data bytes; the CRC has been checked by the caller, fic_decoder.cpp:234-243)
static void walk_fib(const uint8_t *fib, FibTable &t)
{
  int p = 0;
  while (p < 30) {                                                   // fib_decoder.cpp:74-103
    const int type = fib[p] >> 5, len = fib[p] & 0x1F;
    if (type == 7 && len == 0x1F) break;
    if (p + 1 + len > 30) break;                                       // FIG runs past the FIB data field
    if (type == 0 && len >= 1) {
      const uint8_t *d = fib + p;
      const int ext = d[1] & 0x1F, pd = (d[1] >> 5) & 1, cn = (d[1] >> 7) & 1;
      if (ext == 0 && len >= 5) t.cif_count = (d[4] & 0x1F) * 250 + d[5];          // fib_decoder_fig0.cpp:89-101
      else if (ext == 1 && cn == 0) {                                                // :142-224 (current configuration)
        int used = 2;
        while (used <= len) {
          const int o = used * 8;
          if (used + 3 > len + 1) break;
          dabx_subch_desc q{};
          q.subch_id = (int)bits(d, o, 6);
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
          if (q.cu_start + q.cu_size > 864) { t.restart = true; return; }              // :198-202
          if (!t.subch.count(q.subch_id)) {
            for (auto &kv : t.subch)                                                  // :204-209 overlap -> restart
              if (q.cu_start < kv.second.cu_start + kv.second.cu_size && kv.second.cu_start < q.cu_start + q.cu_size) { t.restart = true; return; }
            t.subch[q.subch_id] = q;
            t.order.push_back(q.subch_id);
          }
        }
      } else if (ext == 2 && cn == 0) {                                              // :230-293
        int used = 2;
        while (used <= len) {
          int o = used * 8 + (pd ? 32 : 16);
          if (o / 8 + 1 > len + 1) break;                   // service header (SId + component count) runs past the FIG
          const int ncomp = (int)bits(d, o + 4, 4);
          o += 8;
          for (int c = 0; c < ncomp; c++, o += 16) {
            if ((o + 16) / 8 > len + 1) break;
            const int tmid = (int)bits(d, o, 2);
            if (tmid == 0) t.ascty[(int)bits(d, o + 8, 6)] = (int)bits(d, o + 2, 6);
          }
          used = o / 8;
        }
      }
    }
    p += len + 1;
  }
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

extern "C" {

// fibs: n x 32 bytes, crc_ok: n flags.  Returns the number of sub-channels written to out (in order of first appearance, like
// FibDecoder::get_sub_channel_id_list, fib_decoder.cpp:547-557),
// dab_plus = 1 when FIG 0/2 announces ASCTy 63 for it (backend_driver.cpp:41-50), -1 when unknown yet.
int dabx_parse_fibs(const uint8_t *fibs, const uint8_t *crc_ok, int n_fibs, dabx_subch_desc *out, int max_out, int32_t *cif_count)
{
  if (!fibs || !crc_ok || n_fibs < 0 || (!out && max_out > 0)) { set_error("dabx_parse_fibs: bad argument"); return DABX_E_ARG; }
  FibTable t;
  for (int i = 0; i < n_fibs; i++) {
    if (!crc_ok[i]) continue;
    walk_fib(fibs + (size_t)i * 32, t);
    if (t.restart) { t = FibTable{}; }                       // fib_decoder.cpp:131-141: throw everything away
  }
  if (cif_count) *cif_count = t.cif_count;
  int n = 0;
  for (int id : t.order) {
    if (n >= max_out) break;
    dabx_subch_desc q = t.subch[id];
    const auto it = t.ascty.find(q.subch_id);
    q.dab_plus = it == t.ascty.end() ? -1 : (it->second == 63 ? 1 : 0);
    out[n++] = q;
  }
  return n;
}

}  // extern "C"
