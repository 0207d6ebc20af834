// tii.cpp -- transmitter identification (TII) from accumulated null-symbol spectra, host side.
// Behaviour of TiiDetector (base/ofdm/tii_detector.cpp:163-530): carrier-pair products with a slow IIR, removal of
// lone carriers, collapse of the 4 x 192 pairs onto 8 groups x 24 sub-ids (plain and with the "non-ETSI" phase turn),
// threshold against the weakest sub-id's mean, main-id from the 4-of-8 group pattern, optional collision listing.
// Both tables are derived, not stored: the 70 patterns are the bytes with four bits set in ascending order
// (EN 300 401 table 43) and the phase turn of pair (k, k+1) is the phase-reference difference phi_k - phi_k+1.
// Pinned against the reference's object code: tests/test_tii.py (oracle/_ref, golden fixture).
#include "dabx_internal.h"
#include <algorithm>
#include <math.h>

namespace dabx {
int prs_quarter_turns(int k);   // tables.cpp: (h + n) mod 4 of carrier k, phasetable.cpp:122-135
}
using namespace dabx;

// Complex arithmetic spelled out in float so that the result does not depend on how a compiler lowers std::complex:
// the reference (g++, no -ffast-math in the pinned build) rounds every product and sum separately, |z| is libm's
// hypotf (std::abs -> cabsf) and arg is atan2f.
namespace {
struct cf {
  float re, im;
  cf() : re(0), im(0) {}
  cf(float r, float i) : re(r), im(i) {}
  float real() const { return re; }
  float imag() const { return im; }
  cf &operator+=(cf o) { re += o.re; im += o.im; return *this; }
  cf &operator*=(float k) { re *= k; im *= k; return *this; }
  cf operator-() const { return cf(-re, -im); }
};
inline cf operator-(cf a, cf b) { return cf(a.re - b.re, a.im - b.im); }
inline cf operator*(float k, cf a) { return cf(a.re * k, a.im * k); }
inline cf mul_conj(cf a, cf b) { return cf(a.re * b.re - a.im * (-b.im), a.re * (-b.im) + a.im * b.re); }   // a * conj(b)
inline float cabs_(cf a) { return hypotf(a.re, a.im); }
inline float carg_(cf a) { return atan2f(a.im, a.re); }
}  // namespace

struct dabx_tii {
  cf null_acc[TU];        // mNullSymbolBufferVec
  cf pairs[K / 2];        // mDecodedBufferArr
  uint8_t turn[K / 2];    // phase turn per carrier pair, in quarter turns
  uint8_t pattern[70];    // group pattern of each main id, MSB = group 0
  bool collisions = false;
  int coll_sub_id = 0;
};

static inline int pair_bin(int i) { const int k = -K / 2 + 2 * i; return k < 0 ? k + TU : k + 1; }   // fft_shift_skip_dc

static cf quarter_turn(cf v, int q)      // tii_detector.cpp:282-297
{
  switch (q) {
  case 1: return cf(v.imag(), -v.real());
  case 2: return -v;
  case 3: return cf(-v.imag(), v.real());
  default: return v;
  }
}

extern "C" {

int dabx_tii_create(dabx_tii **out)
{
  if (!out) return DABX_E_ARG;
  dabx_tii *t = new dabx_tii();
  int n = 0;
  for (int v = 0; v < 256 && n < 70; v++) if (__builtin_popcount(v) == 4) t->pattern[n++] = (uint8_t)v;
  for (int i = 0; i < K / 2; i++) {
    const int k = -K / 2 + 2 * i, c0 = k < 0 ? k : k + 1;
    t->turn[i] = (uint8_t)((prs_quarter_turns(c0) - prs_quarter_turns(c0 + 1)) & 3);
  }
  *out = t;
  return 0;
}
void dabx_tii_destroy(dabx_tii *t) { delete t; }
void dabx_tii_reset(dabx_tii *t)          // TiiDetector::reset, :149-153
{
  if (!t) return;
  for (auto &v : t->null_acc) v = cf(0, 0);
  for (auto &v : t->pairs) v = cf(0, 0);
}
void dabx_tii_set_collisions(dabx_tii *t, int on, int sub_id)
{
  if (t) { t->collisions = on != 0; t->coll_sub_id = sub_id; }
}
int dabx_tii_add(dabx_tii *t, const float *null_fft)     // add_to_tii_buffer, :156-162
{
  if (!t || !null_fft) return DABX_E_ARG;
  for (int i = 0; i < TU; i++) t->null_acc[i] += cf(null_fft[2 * i], null_fft[2 * i + 1]);
  return 0;
}

int dabx_tii_process(dabx_tii *t, int threshold_db, dabx_tii_result *out, int max_out)     // process_tii_data, :164-240
{
  if (!t || (!out && max_out > 0) || max_out < 0) return DABX_E_ARG;
  constexpr int NB = 4, NG = 8, GS = 24, BS = NG * GS;
  for (int i = 0; i < K / 2; i++) {                       // :247-266
    const int b = pair_bin(i);
    const cf prod = mul_conj(t->null_acc[b], t->null_acc[b + 1]);
    t->pairs[i] += 0.01f * (prod - t->pairs[i]);
  }
  cf work[K / 2];
  std::copy(t->pairs, t->pairs + K / 2, work);
  for (int i = 0; i < BS; i++) {                          // a carrier alone in its four blocks is no TII, :268-296
    float mx = 0, sum = 0;
    int at = 0;
    for (int j = 0; j < NB; j++) {
      const float x = cabs_(work[i + j * BS]);
      sum += x;
      if (x > mx) { mx = x; at = j; }
    }
    const float mn = (sum - mx) / (NB - 1);
    if (sum < mx * 1.5 && mx > 0.0) work[i + at * BS] *= mn / mx;
  }
  cf etsi[BS], turned[BS];
  float etsi_abs[BS], turned_abs[BS], top = 0;
  for (int i = 0; i < BS; i++) {                          // :320-344
    etsi[i] = turned[i] = cf(0, 0);
    for (int j = 0; j < NB; j++) {
      const cf x = work[i + j * BS];
      etsi[i] += x;
      turned[i] += quarter_turn(x, t->turn[i + j * BS]);
    }
  }
  for (int i = 0; i < BS; i++) { etsi_abs[i] = cabs_(etsi[i]); if (etsi_abs[i] > top) top = etsi_abs[i]; }
  for (int i = 0; i < BS; i++) { turned_abs[i] = cabs_(turned[i]); if (turned_abs[i] > top) top = turned_abs[i]; }
  float noise = 1e9;                                      // weakest sub-id, :513-530
  for (int sub = 0; sub < GS; sub++) {
    float avg = 0;
    for (int g = 0; g < NG; g++) avg += etsi_abs[sub + g * GS];
    avg /= NG;
    if (avg < noise) noise = avg;
  }
  // F_DEG_PER_RAD = (f32)(180.0 / M_PI) under the reference's -fsingle-precision-constant: a float division
  const float kDegPerRad = 180.0f / 3.14159265358979323846f;
  volatile float ten = 10.0f;                               // keeps powf from being rewritten as exp10f
  std::vector<dabx_tii_result> res;
  auto push = [&](int main_id, int sub, float strength, cf sum, bool non_etsi) {
    res.push_back(dabx_tii_result{(uint8_t)main_id, (uint8_t)sub, strength, carg_(sum) * kDegPerRad, non_etsi ? 1 : 0});
  };
  for (int sub = 0; sub < GS; sub++) {
    const float level = noise * powf(ten, (float)threshold_db / 10.0f);
    cf s_e(0, 0), s_t(0, 0);
    int n_e = 0, n_t = 0;
    unsigned p_e = 0, p_t = 0;
    for (int g = 0; g < NG; g++) {                        // :389-441
      const int ix = sub + g * GS;
      if (etsi_abs[ix] > level) { n_e++; p_e |= 0x80u >> g; s_e += etsi[ix]; }
      if (turned_abs[ix] > level) { n_t++; p_t |= 0x80u >> g; s_t += turned[ix]; }
    }
    const bool non_etsi = (n_e >= 4 || n_t >= 4) && cabs_(s_t) > cabs_(s_e);
    cf sum = non_etsi ? s_t : s_e;
    const int count = non_etsi ? n_t : n_e;
    const unsigned pat = non_etsi ? p_t : p_e;
    const cf *tab = non_etsi ? turned : etsi;
    const float *tab_abs = non_etsi ? turned_abs : etsi_abs;
    int main_id = 0;
    if (count == 4) {                                     // :346-357
      main_id = -1;
      for (int m = 0; m < 70; m++) if (t->pattern[m] == pat) { main_id = m; break; }
    } else if (count > 4) {                               // best four of the groups, :359-387
      float best = 0;
      main_id = -1;
      sum = cf(0, 0);
      for (int m = 0; m < 70; m++) {
        cf v(0, 0);
        for (int g = 0; g < NG; g++) if (t->pattern[m] & (0x80u >> g)) v += tab[sub + GS * g];
        if (cabs_(v) > best) { best = cabs_(v); sum = v; main_id = m; }
      }
    }
    if (count >= 4) push(main_id, sub, cabs_(sum) / top / 4, sum, non_etsi);
    if (count > 4 && t->collisions) {                     // :443-500
      cf rest(0, 0);
      for (int g = 0; g < NG; g++)
        if (!(t->pattern[main_id] & (0x80u >> g)) && tab_abs[sub + GS * g] > level) rest += tab[sub + GS * g];
      const float strength = cabs_(rest) / top / (float)(count - 4);
      if (sub == t->coll_sub_id) {
        for (int m = 0; m < 70; m++)
          if (__builtin_popcount(t->pattern[m] & pat) == 4 && m != main_id) push(m, sub, strength, rest, non_etsi);
      } else push(99, sub, strength, rest, non_etsi);
    }
  }
  for (auto &v : t->null_acc) v = cf(0, 0);
  std::sort(res.begin(), res.end(), [](const dabx_tii_result &a, const dabx_tii_result &b) { return a.strength > b.strength; });
  const int n = (int)std::min<size_t>(res.size(), (size_t)max_out);
  for (int i = 0; i < n; i++) out[i] = res[(size_t)i];
  return n;
}

}  // extern "C"
