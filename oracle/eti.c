/* eti.c -- ETI(NI) frame header + assembly (oracle; test infrastructure only; PARITY UNPINNED: EtiGenerator pulls in
 * dabradio.h / FibDecoder and cannot be built here).  Restates base/eti_handler/eti_generator.cpp:169-199, :207-308
 * with the table-driven calc_crc of base/backend/crc.cpp:75-86. */
#include "dab_oracle.h"
#include <string.h>

int ora_eti_frame(int hi, int lo, int minor, const ora_subch_desc *sc, int nst, const uint8_t *fic96, const uint8_t *const *msc, uint8_t *eti)
{
  int fill = 0, FL = 0;
  lo += minor;
  if (lo >= 250) { lo = lo % 250; hi++; }
  if (hi >= 20) hi = 20;
  eti[fill++] = 0xFF;
  if (lo & 1) { eti[fill++] = 0xf8; eti[fill++] = 0xc5; eti[fill++] = 0x49; }
  else { eti[fill++] = 0x07; eti[fill++] = 0x3a; eti[fill++] = 0xb6; }
  eti[fill++] = (uint8_t)lo;
  for (int i = 0; i < nst; i++) FL += (sc[i].kbps * 3) / 4;
  FL += nst + 1 + 24;
  eti[fill++] = (uint8_t)((1 << 7) | nst);
  {
    const uint8_t FP = (uint8_t)(((hi * 250) + lo) % 8);
    eti[fill++] = (uint8_t)((FP << 5) | (0x01 << 3) | ((FL & 0x700) >> 8));
    eti[fill++] = (uint8_t)(FL & 0xff);
  }
  for (int i = 0; i < nst; i++) {
    const int SCID = sc[i].subch_id, SAD = sc[i].cu_start;
    const int TPL = sc[i].short_form ? (0x10 | (sc[i].prot_level - 1)) : (0x20 | sc[i].prot_level);
    const int STL = sc[i].kbps * 3 / 8;
    eti[fill++] = (uint8_t)((SCID << 2) | ((SAD & 0x300) >> 8));
    eti[fill++] = (uint8_t)(SAD & 0xFF);
    eti[fill++] = (uint8_t)((TPL << 2) | ((STL & 0x300) >> 8));
    eti[fill++] = (uint8_t)(STL & 0xFF);
  }
  eti[fill++] = 0xFF; eti[fill++] = 0xFF;
  {
    const uint16_t h = ora_calc_crc(eti + 4, fill - 4);
    eti[fill++] = (uint8_t)((h & 0xff00) >> 8); eti[fill++] = (uint8_t)(h & 0xff);
  }
  {
    const int base = fill;
    memcpy(eti + fill, fic96, 96); fill += 96;
    for (int i = 0; i < nst; i++) { memcpy(eti + fill, msc[i], (size_t)(sc[i].kbps * 24 / 8)); fill += sc[i].kbps * 24 / 8; }
    const uint16_t c = ora_calc_crc(eti + base, fill - base);
    eti[fill++] = (uint8_t)((c & 0xFF00) >> 8); eti[fill++] = (uint8_t)(c & 0xFF);
  }
  eti[fill++] = 0xFF; eti[fill++] = 0xFF;
  eti[fill++] = 0xFF; eti[fill++] = 0xFF; eti[fill++] = 0xFF; eti[fill++] = 0xFF;
  memset(eti + fill, 0x55, (size_t)(6144 - fill));
  return fill;
}
