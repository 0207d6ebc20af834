// eti.cpp -- ETI(NI) frame assembly, host side.  One 6144-byte frame per CIF:
// SYNC | FC | STC x NST | EOH | MST = FIC (96 B, the 3 FIBs of this CIF) + sub-channel streams | EOF | TIST | 0x55 padding.
// Follows base/eti_handler/eti_generator.cpp:169-199 (assembly) and :207-308 (_init_eti), byte for byte.
#include "dabx_internal.h"
#include <string.h>

namespace dabx {

// calc_crc, base/backend/crc.cpp:75-86 (CRC-16-CCITT, init 0xFFFF, complemented)
static uint16_t crc_ccitt(const uint8_t *d, int n)
{
  uint16_t crc = 0xFFFF;
  for (int i = 0; i < n; i++) {
    crc ^= (uint16_t)(d[i] << 8);
    for (int k = 0; k < 8; k++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
  }
  return (uint16_t)~crc;
}

}  // namespace dabx

using namespace dabx;

extern "C" int dabx_eti_frame(int cif_hi, int cif_lo, int minor, const dabx_subch_desc *sc, int n_subch, const uint8_t *fic96,
                              const uint8_t *const *msc, uint8_t *out)
{
  if (cif_hi < 0 || cif_lo < 0 || minor < 0 || minor > 3 || n_subch < 0 || n_subch > 64 || (n_subch && (!sc || !msc)) || !fic96 || !out) {
    set_error("dabx_eti_frame: bad argument");
    return DABX_E_ARG;
  }
  for (int i = 0; i < n_subch; i++) {
    const dabx_subch_desc &q = sc[i];
    const bool lvl_ok = q.short_form ? (q.prot_level >= 1 && q.prot_level <= 5) : (q.prot_level >= 0 && q.prot_level <= 7);
    if (!lvl_ok || q.subch_id < 0 || q.subch_id > 63 || q.cu_start < 0 || q.cu_start > 863 || q.kbps <= 0 || q.kbps > 1023 || !msc[i]) {
      set_error("dabx_eti_frame: sub-channel %d is not a legal description", i);
      return DABX_E_ARG;
    }
  }
  int fl = 0, mst = 96;
  for (int i = 0; i < n_subch; i++) { fl += sc[i].kbps * 3 / 4; mst += sc[i].kbps * 3; }
  if (4 + 4 + 4 * n_subch + 4 + mst + 8 > 6144) { set_error("dabx_eti_frame: %d bytes of sub-channel data do not fit an ETI frame", mst); return DABX_E_ARG; }
  int p = 0;
  cif_lo += minor;                                       // :212-220
  if (cif_lo >= 250) { cif_lo %= 250; cif_hi++; }
  if (cif_hi >= 20) cif_hi = 20;
  out[p++] = 0xFF;                                       // ERR: level 0
  if (cif_lo & 1) { out[p++] = 0xF8; out[p++] = 0xC5; out[p++] = 0x49; }     // FSYNC alternates
  else { out[p++] = 0x07; out[p++] = 0x3A; out[p++] = 0xB6; }
  out[p++] = (uint8_t)cif_lo;                            // FCT
  fl += n_subch + 1 + 24;                                // STC + EOH + FIC words (mode I)
  out[p++] = (uint8_t)((1 << 7) | n_subch);              // FICF | NST
  const int fp = (cif_hi * 250 + cif_lo) % 8, mid = 1;
  out[p++] = (uint8_t)((fp << 5) | (mid << 3) | ((fl & 0x700) >> 8));
  out[p++] = (uint8_t)(fl & 0xFF);
  for (int i = 0; i < n_subch; i++) {                    // STC, :271-294
    const int tpl = sc[i].short_form ? (0x10 | (sc[i].prot_level - 1)) : (0x20 | sc[i].prot_level);
    const int stl = sc[i].kbps * 3 / 8;
    out[p++] = (uint8_t)((sc[i].subch_id << 2) | ((sc[i].cu_start & 0x300) >> 8));
    out[p++] = (uint8_t)(sc[i].cu_start & 0xFF);
    out[p++] = (uint8_t)((tpl << 2) | ((stl & 0x300) >> 8));
    out[p++] = (uint8_t)(stl & 0xFF);
  }
  out[p++] = 0xFF; out[p++] = 0xFF;                      // EOH: MNSC
  const uint16_t hcrc = crc_ccitt(out + 4, p - 4);
  out[p++] = (uint8_t)(hcrc >> 8); out[p++] = (uint8_t)(hcrc & 0xFF);
  const int base = p;
  memcpy(out + p, fic96, 96); p += 96;                   // :171-172
  for (int i = 0; i < n_subch; i++) { memcpy(out + p, msc[i], (size_t)sc[i].kbps * 3); p += sc[i].kbps * 3; }
  const uint16_t crc = crc_ccitt(out + base, p - base);  // EOF, :180-192
  out[p++] = (uint8_t)(crc >> 8); out[p++] = (uint8_t)(crc & 0xFF);
  out[p++] = 0xFF; out[p++] = 0xFF;                      // RFU
  out[p++] = 0xFF; out[p++] = 0xFF; out[p++] = 0xFF; out[p++] = 0xFF;   // TIST unused
  memset(out + p, 0x55, (size_t)(6144 - p));
  return p;
}
