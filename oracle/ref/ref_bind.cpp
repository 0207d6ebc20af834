// ref_bind.cpp -- extern "C" access to the GENUINE reference leaf classes (test infrastructure).
// Compiled together with the unmodified sources under /root/reference/src by oracle/ref/Makefile
// into oracle/_ref/libdabref.so.  No reference source is copied; this file only calls the
// reference's public (or protected, via a derived class) interfaces.
#include "viterbi_spiral.h"
#include "protTables.h"
#include "eep_protection.h"
#include "uep_protection.h"
#include "backend_deconvolver.h"
#include "reed_solomon.h"
#include "firecode_checker.h"
#include "crc.h"
#include "freq_interleaver.h"
#include "phasetable.h"
#include "tii_detector.h"
#include "xml_descriptor.h"     // .uff header parser (QtXml)
#include "fib_table.h"          // cProtLevelTable: the short-form (UEP) sub-channel table of FIG 0/1
#include <chrono>
#include <cstring>
#include <vector>

namespace {
template <class P> struct Open : P {
  using P::P;
  // protection.h:50-53 (protected): depuncture address list -> index map
  int map(int32_t * out, int n) {
    for (int i = 0; i < n; i++) out[i] = -1;
    int k = 0;
    for (i16 * a : this->viterbiBlockAddresses) out[a - this->viterbiBlock.data()] = k++;
    return k;
  }
};
struct OpenPhase : PhaseTable { using PhaseTable::mRefTable; };
}

extern "C" {

int ref_viterbi(const int16_t * soft, int nbits, uint8_t * out)   // viterbi_spiral.h:20
{
  ViterbiSpiral v((short)nbits, true);
  v.deconvolve(soft, out);
  return 0;
}

// one decoder object, `reps` decodes of the same block: seconds per decode (CPU baseline of bench.py)
double ref_viterbi_seconds(const int16_t * soft, int nbits, uint8_t * out, int reps)
{
  ViterbiSpiral v((short)nbits, true);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) v.deconvolve(soft, out);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (reps > 0 ? reps : 1);
}

int ref_viterbi_ber(const int16_t * soft, uint8_t * punct, const uint8_t * bits, int nbits, int * io_bits, int * io_err)
{
  ViterbiSpiral v((short)nbits, true);
  v.calculate_BER(soft, punct, bits, *io_bits, *io_err);
  return 0;
}

void ref_pi_codes(int pi, int8_t * out32) { memcpy(out32, get_PI_codes((i16)pi), 32); }   // protTables.h

int ref_eep_map(int kbps, int prot, int32_t * out)
{
  Open<EepProtection> p((i16)kbps, (i16)prot);
  return p.map(out, 96 * kbps + 24);
}
int ref_uep_map(int kbps, int prot, int32_t * out)
{
  Open<UepProtection> p((i16)kbps, (i16)prot);
  return p.map(out, 96 * kbps + 24);
}
int ref_eep_deconvolve(int kbps, int prot, const int16_t * in, int n_in, uint8_t * out)   // protection.h:44
{
  EepProtection p((i16)kbps, (i16)prot);
  return p.deconvolve(in, n_in, out) ? 0 : -1;
}
int ref_uep_deconvolve(int kbps, int prot, const int16_t * in, int n_in, uint8_t * out)
{
  UepProtection p((i16)kbps, (i16)prot);
  return p.deconvolve(in, n_in, out) ? 0 : -1;
}
void ref_backend_deconvolve(int short_form, int kbps, int prot, const int16_t * in, int n_in, uint8_t * out)   // backend_deconvolver.h:29-31
{
  SDescriptorType d{};
  d.shortForm = short_form != 0;
  d.bitRate = (i16)kbps;
  d.protLevel = (i16)prot;
  BackendDeconvolver(&d).deconvolve(in, n_in, out);
}

int ref_rs_dec(const uint8_t * in120, uint8_t * out110)   // reed_solomon.h:28, params mp4processor.cpp:63,203
{
  static ReedSolomon rs(8, 0435, 0, 1, 10);
  return rs.dec(in120, out110, 135);
}
void ref_rs_enc(const uint8_t * in110, uint8_t * out120)
{
  static ReedSolomon rs(8, 0435, 0, 1, 10);
  rs.enc(in110, out120, 135);
}

int ref_firecode_check(const uint8_t * x11)
{
  static FirecodeChecker fc;
  return fc.check(x11) ? 1 : 0;
}
int ref_firecode_check_and_correct(uint8_t * x12)
{
  static FirecodeChecker fc;
  return fc.check_and_correct_6bits(x12) ? 1 : 0;
}

int ref_check_crc_bits(const uint8_t * bits, int n) { return check_CRC_bits(bits, n) ? 1 : 0; }
int ref_calc_crc(const uint8_t * d, int n) { return calc_crc(d, n); }
int ref_check_crc_bytes(const uint8_t * d, int n) { return check_crc_bytes(d, n) ? 1 : 0; }

void ref_freq_interleaver(int16_t * out1536)
{
  FreqInterleaver f;
  for (int k = 0; k < 1536; k++) out1536[k] = f.map_k_to_fft_bin((i16)k);
}
void ref_phase_table(float * out4096)
{
  OpenPhase p;
  memcpy(out4096, p.mRefTable.data(), sizeof(float) * 4096);
}

void ref_uep_table(int16_t * out192)     // 64 x {CU size, protection level, bit rate}, fib_table.h:51-117
{
  for (int i = 0; i < 64; i++) {
    out192[3 * i] = cProtLevelTable[i].CUSize; out192[3 * i + 1] = cProtLevelTable[i].ProtLevel; out192[3 * i + 2] = cProtLevelTable[i].BitRate;
  }
}

// ---- TiiDetector (base/ofdm/tii_detector.cpp, compiled unmodified) ----------------------------------------
void * ref_tii_new() { return new TiiDetector(); }
void ref_tii_free(void * p) { delete static_cast<TiiDetector *>(p); }
void ref_tii_reset(void * p) { static_cast<TiiDetector *>(p)->reset(); }
void ref_tii_set(void * p, int collisions, int sub_id)
{
  static_cast<TiiDetector *>(p)->set_detect_collisions(collisions != 0);
  static_cast<TiiDetector *>(p)->set_subid_for_collision_search((u8)sub_id);
}
void ref_tii_add(void * p, const float * null_fft4096)
{
  TArrayTu v;
  memcpy(v.data(), null_fft4096, sizeof(float) * 4096);
  static_cast<TiiDetector *>(p)->add_to_tii_buffer(v);
}
// out: n x {mainId, subId, strength, phaseDeg, isNonEtsi} as 5 floats
int ref_tii_process(void * p, int threshold_db, float * out, int max_out)
{
  const std::vector<STiiResult> r = static_cast<TiiDetector *>(p)->process_tii_data((i16)threshold_db);
  int n = 0;
  for (const auto & e : r) {
    if (n >= max_out) break;
    out[5 * n] = e.mainId; out[5 * n + 1] = e.subId; out[5 * n + 2] = e.strength; out[5 * n + 3] = e.phaseDeg; out[5 * n + 4] = e.isNonEtsiPhase ? 1.f : 0.f;
    n++;
  }
  return n;
}

// ---- XmlDescriptor (devices/filereaders/xml_filereader/xml_descriptor.cpp, compiled unmodified) ----------------
// ints: sampleRate, nrChannels, bitsperChannel, nrBlocks, ok ; strings (<= 15 chars + NUL each): container, byteOrder, iqOrder
int ref_uff_describe(const char * path, int32_t * ints, char * strings48, long long * nr_elements)
{
  FILE * f = fopen(path, "rb");
  if (!f) return -1;
  bool ok = false;
  XmlDescriptor d(f, &ok);
  fclose(f);
  ints[0] = d.sampleRate; ints[1] = d.nrChannels; ints[2] = d.bitsperChannel; ints[3] = (int)d.blockList.size();      // nrBlocks itself is uninitialised when the header has no <Datablocks> element
  ints[4] = ok ? 1 : 0;
  memset(strings48, 0, 48);
  strncpy(strings48, d.container.toUtf8().constData(), 15);
  strncpy(strings48 + 16, d.byteOrder.toUtf8().constData(), 15);
  strncpy(strings48 + 32, d.iqOrder.toUtf8().constData(), 15);
  *nr_elements = 0;
  for (const auto & b : d.blockList) *nr_elements += b.nrElements;
  return 0;
}

}  // extern "C"
