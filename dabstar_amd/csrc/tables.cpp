// tables.cpp -- host-side construction of the constant tables of the Mode-I path and their upload.
// (Product code; independent of oracle/.)  Reference anchors are quoted per table.
#include "dabx_internal.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

namespace dabx {

static thread_local std::string g_err;
void set_error(const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
  set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu) ? DABX_E_NODEVICE
                                                                                                 : DABX_E_HIP;
}
const char *last_error() { return g_err.c_str(); }

// ---- puncturing (EN 300 401 11.1.2 table 13; protection/protTables.cpp:36-68) ---------------------
static void pi_vector(int pi, uint8_t v[32])
{
  static const int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
  const int base = (pi - 1) / 8 + 1, extra = (pi - 1) % 8 + 1;
  int ones[8];
  for (int g = 0; g < 8; g++) ones[g] = base;
  for (int e = 0; e < extra; e++) ones[order[e]] = base + 1;
  for (int g = 0; g < 8; g++)
    for (int j = 0; j < 4; j++) v[4 * g + j] = j < ones[g];
}

struct MapBuilder {
  std::vector<uint16_t> &map;
  int in_idx = 0;
  explicit MapBuilder(std::vector<uint16_t> &m) : map(m) { map.clear(); }
  void blocks(int n128, int pi)   // eep_protection.cpp:153-167
  {
    uint8_t v[32];
    if (pi > 0) pi_vector(pi, v);
    for (int i = 0; i < n128 * 128; i++) map.push_back(pi > 0 && v[i % 32] ? (uint16_t)in_idx++ : PUNCT);
  }
  void tail()                     // eep_protection.cpp:137-150 : 24 bits with PI_8
  {
    uint8_t v[32];
    pi_vector(8, v);
    for (int i = 0; i < 24; i++) map.push_back(v[i] ? (uint16_t)in_idx++ : PUNCT);
  }
};

void host_fic_map(std::vector<uint16_t> &map)   // fic_decoder.cpp:79-124
{
  MapBuilder b(map);
  b.blocks(21, 16);
  b.blocks(3, 15);
  b.tail();
}

// EN 300 401 table 8 (UEP profiles), protection/uep_protection.cpp:52-134: kbps, level, L1..L4, PI1..PI4 (0 = none)
static const int16_t kUep[][10] = {
    {32, 5, 3, 4, 17, 0, 5, 3, 2, 0},       {32, 4, 3, 3, 18, 0, 11, 6, 5, 0},      {32, 3, 3, 4, 14, 3, 15, 9, 6, 8},
    {32, 2, 3, 4, 14, 3, 22, 13, 8, 13},    {32, 1, 3, 5, 13, 3, 24, 17, 12, 17},   {48, 5, 4, 3, 26, 3, 5, 4, 2, 3},
    {48, 4, 3, 4, 26, 3, 9, 6, 4, 6},       {48, 3, 3, 4, 26, 3, 15, 10, 6, 9},     {48, 2, 3, 4, 26, 3, 24, 14, 8, 15},
    {48, 1, 3, 5, 25, 3, 24, 18, 13, 18},   {56, 5, 6, 10, 23, 3, 5, 4, 2, 3},      {56, 4, 6, 10, 23, 3, 9, 6, 4, 5},
    {56, 3, 6, 12, 21, 3, 16, 7, 6, 9},     {56, 2, 6, 10, 23, 3, 23, 13, 8, 13},   {64, 5, 6, 9, 31, 2, 5, 3, 2, 3},
    {64, 4, 6, 9, 33, 0, 11, 6, 5, 0},      {64, 3, 6, 12, 27, 3, 16, 8, 6, 9},     {64, 2, 6, 10, 29, 3, 23, 13, 8, 13},
    {64, 1, 6, 11, 28, 3, 24, 18, 12, 18},  {80, 5, 6, 10, 41, 3, 6, 3, 2, 3},      {80, 4, 6, 10, 41, 3, 11, 6, 5, 6},
    {80, 3, 6, 11, 40, 3, 16, 8, 6, 7},     {80, 2, 6, 10, 41, 3, 23, 13, 8, 13},   {80, 1, 6, 10, 41, 3, 24, 7, 12, 18},
    {96, 5, 7, 9, 53, 3, 5, 4, 2, 4},       {96, 4, 7, 10, 52, 3, 9, 6, 4, 6},      {96, 3, 6, 12, 51, 3, 16, 9, 6, 10},
    {96, 2, 6, 10, 53, 3, 22, 12, 9, 12},   {96, 1, 6, 13, 50, 3, 24, 18, 13, 19},  {112, 5, 14, 17, 50, 3, 5, 4, 2, 5},
    {112, 4, 11, 21, 49, 3, 9, 6, 4, 8},    {112, 3, 11, 23, 47, 3, 16, 8, 6, 9},   {112, 2, 11, 21, 49, 3, 23, 12, 9, 14},
    {128, 5, 12, 19, 62, 3, 5, 3, 2, 4},    {128, 4, 11, 21, 61, 3, 11, 6, 5, 7},   {128, 3, 11, 22, 60, 3, 16, 9, 6, 10},
    {128, 2, 11, 21, 61, 3, 22, 12, 9, 14}, {128, 1, 11, 20, 62, 3, 24, 17, 13, 19}, {160, 5, 11, 19, 87, 3, 5, 4, 2, 4},
    {160, 4, 11, 23, 83, 3, 11, 6, 5, 9},   {160, 3, 11, 24, 82, 3, 16, 8, 6, 11},  {160, 2, 11, 21, 85, 3, 22, 11, 9, 13},
    {160, 1, 11, 22, 84, 3, 24, 18, 12, 19}, {192, 5, 11, 20, 110, 3, 6, 4, 2, 5},  {192, 4, 11, 22, 108, 3, 10, 6, 4, 9},
    {192, 3, 11, 24, 106, 3, 16, 10, 6, 11}, {192, 2, 11, 20, 110, 3, 22, 13, 9, 13}, {192, 1, 11, 21, 109, 3, 24, 20, 13, 24},
    {224, 5, 12, 22, 131, 3, 8, 6, 2, 6},   {224, 4, 12, 26, 127, 3, 12, 8, 4, 11}, {224, 3, 11, 20, 134, 3, 16, 10, 7, 9},
    {224, 2, 11, 22, 132, 3, 24, 16, 10, 15}, {224, 1, 11, 24, 130, 3, 24, 20, 12, 20}, {256, 5, 11, 24, 154, 3, 6, 5, 2, 5},
    {256, 4, 11, 24, 154, 3, 12, 9, 5, 10}, {256, 3, 11, 27, 151, 3, 16, 10, 7, 10}, {256, 2, 11, 22, 156, 3, 24, 14, 10, 13},
    {256, 1, 11, 26, 152, 3, 24, 19, 14, 18}, {320, 5, 11, 26, 200, 3, 8, 5, 2, 6}, {320, 4, 11, 25, 201, 3, 13, 9, 5, 10},
    {320, 2, 11, 26, 200, 3, 24, 17, 9, 17}, {384, 5, 11, 27, 247, 3, 8, 6, 2, 7},  {384, 3, 11, 24, 250, 3, 16, 9, 7, 10},
    {384, 1, 12, 28, 245, 3, 24, 20, 14, 23}};

int host_profile_map(int kbps, int prot, int short_form, std::vector<uint16_t> &map, int *n_in)
{
  MapBuilder b(map);
  if (kbps <= 0 || 96 * kbps + 24 > 65535 - 24) return DABX_E_PROFILE;   // 16-bit indices; the reference's own i16 counters
                                                                         // overflow above 341 kbit/s (protection.cpp:48)
  if (short_form) {                        // uep_protection.cpp:136-196
    const int16_t *row = nullptr;
    for (auto &r : kUep)
      if (r[0] == kbps && r[1] == prot) row = r;
    if (!row) return DABX_E_PROFILE;
    for (int k = 0; k < 4; k++) b.blocks(row[2 + k], row[6 + k]);
  } else {                                 // eep_protection.cpp:43-135
    const int lvl = prot & 3, opt = (prot >> 2) & 1;
    if (prot < 0 || prot > 7) return DABX_E_PROFILE;
    int L1, L2, p1, p2;
    if (opt == 0) {
      if (kbps % 8) return DABX_E_PROFILE;
      const int n = kbps / 8;
      switch (lvl) {
      case 0: L1 = 6 * n - 3; L2 = 3; p1 = 24; p2 = 23; break;
      case 1:
        if (n == 1) { L1 = 5; L2 = 1; p1 = 13; p2 = 12; }
        else { L1 = 2 * n - 3; L2 = 4 * n + 3; p1 = 14; p2 = 13; }
        break;
      case 2: L1 = 6 * n - 3; L2 = 3; p1 = 8; p2 = 7; break;
      default: L1 = 4 * n - 3; L2 = 2 * n + 3; p1 = 3; p2 = 2; break;
      }
    } else {
      if (kbps % 32) return DABX_E_PROFILE;
      const int n = kbps / 32;
      static const int pib[4] = {10, 6, 4, 2};
      L1 = 24 * n - 3; L2 = 3; p1 = pib[lvl]; p2 = pib[lvl] - 1;
    }
    b.blocks(L1, p1);
    b.blocks(L2, p2);
  }
  b.tail();
  if ((int)map.size() != 96 * kbps + 24) return DABX_E_PROFILE;
  *n_in = b.in_idx;
  return 0;
}

// ---- per-device table set ---------------------------------------------------------------------------
static std::mutex g_mu;
static std::map<int, DevTables> g_tables;
static std::map<std::tuple<int, int, int, int>, std::pair<uint16_t *, int>> g_maps;

template <class T> static int upload(T **dst, const std::vector<T> &src)
{
  DABX_HIP(hipMalloc((void **)dst, src.size() * sizeof(T)));
  DABX_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// ofdm/phasetable.cpp:35-135 (EN 300 401 14.3.2, table 39/44)
static const uint8_t kPrsNLo[24] = {1, 2, 0, 1, 3, 2, 2, 3, 2, 1, 2, 3, 1, 2, 3, 3, 2, 2, 2, 1, 1, 3, 1, 2};
static const uint8_t kPrsNHi[24] = {3, 1, 1, 1, 2, 2, 1, 0, 2, 2, 3, 3, 0, 2, 1, 3, 3, 3, 3, 0, 3, 0, 1, 1};
static const uint8_t kPrsH[4][16] = {{0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1},
                                     {0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0},
                                     {0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3},
                                     {0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2}};
static float prs_phi(int k)
{
  int b, kp, i, n;
  if (k < 0) { b = (k + 768) / 32; kp = -768 + 32 * b; i = b & 3; n = kPrsNLo[b]; }
  else { b = (k - 1) / 32; kp = 1 + 32 * b; i = (4 - (b & 3)) & 3; n = kPrsNHi[b]; }
  return (float)(M_PI / 2.0) * (float)(kPrsH[i][(k - kp) & 15] + n);   // phasetable.cpp:122-135
}

// (h + n) mod 4 of carrier k (quarter turns of the phase reference), for the TII pair table (tii.cpp)
int prs_quarter_turns(int k)
{
  if (k == 0 || k < -768 || k > 768) return 0;
  return (int)lroundf(prs_phi(k) / (float)(M_PI / 2.0)) & 3;
}

static void host_dft(std::vector<double> &re, std::vector<double> &im, bool inverse)
{
  const int N = TU;
  std::vector<double> ore(N), oim(N);
  // radix-2 in double; only used at init for the coarse-CFO reference
  for (int i = 0; i < N; i++) {
    unsigned r = 0;
    for (int b = 0; b < 11; b++) r |= ((i >> b) & 1u) << (10 - b);
    ore[r] = re[i]; oim[r] = im[i];
  }
  for (int len = 2; len <= N; len <<= 1)
    for (int s = 0; s < N; s += len)
      for (int k = 0; k < len / 2; k++) {
        const double a = (inverse ? 2.0 : -2.0) * M_PI * k / len, wr = cos(a), wi = sin(a);
        const double xr = ore[s + k + len / 2], xi = oim[s + k + len / 2];
        const double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
        ore[s + k + len / 2] = ore[s + k] - tr; oim[s + k + len / 2] = oim[s + k] - ti;
        ore[s + k] += tr; oim[s + k] += ti;
      }
  re.swap(ore); im.swap(oim);
}

// frequency interleaver, freq_interleaver.cpp:40-76 + the index fold of ofdm_decoder.cpp:171-179: carrier k -> FFT bin, relative index
static void host_freq_interleaver(std::vector<uint16_t> &bin, std::vector<int16_t> &rel)
{
  bin.clear(); rel.clear();
  int v = 0;
  for (int i = 0; i < TU; i++) {
    if (i > 0) v = (13 * v + 511) % TU;
    if (v == TU / 2 || v < 256 || v > 256 + K) continue;
    const int k = v - TU / 2;
    bin.push_back((uint16_t)(k < 0 ? k + TU : k));
    rel.push_back((int16_t)(k < 0 ? k + K / 2 : k + K / 2 - 1));
  }
}

// LDS slots of k_symbols' frequency de-interleave.  Thread tid of the 256-thread transform holds bins tid + 256 u and scatters
// them (ds_write_b64: groups of 16 contiguous lanes, bank = slot mod 16), then reads carriers tid + 256 u back in order
// (ds_read_b64: groups of 32 lanes, bank = slot mod 32).  slot(k) = (k & ~15) | sigma(k) keeps every aligned run of 16
// carriers in place, so the read-back is conflict-free for ANY set of permutations sigma; the scatter is conflict-free
// iff the carriers written by one 16-lane group get 16 different sigma.  That is an edge colouring of the bipartite
// multigraph (write group) -- carrier k -- (run k >> 4) with 16 colours, which exists because no vertex has more
// than 16 edges (Koenig); built with alternating-path recolouring.  Deterministic.
static void host_carrier_slots(const std::vector<uint16_t> &bin, std::vector<int> &sigma)
{
  sigma.assign(K, -1);
  const int NG = 8 * 16, NB = K / 16, NC = 16;
  std::vector<int> at_g((size_t)NG * NC, -1), at_b((size_t)NB * NC, -1);      // edge (carrier) using colour c at the vertex
  auto group_of = [&](int k) { const int b = bin[k]; return (b >> 8) * 16 + ((b & 255) >> 4); };   // (u, tid >> 4)
  for (int k = 0; k < K; k++) {
    const int g = group_of(k), b = k >> 4;
    int cg = 0, cb = 0;
    while (at_g[(size_t)g * NC + cg] >= 0) cg++;
    while (at_b[(size_t)b * NC + cb] >= 0) cb++;
    if (at_b[(size_t)b * NC + cg] >= 0) {
      // cg is taken at b: swap cg <-> cb along the alternating path that starts at b with colour cg (it cannot reach g)
      std::vector<int> path;
      int v_is_b = 1, v = b, want = cg;
      for (;;) {
        const int e = v_is_b ? at_b[(size_t)v * NC + want] : at_g[(size_t)v * NC + want];
        if (e < 0) break;
        path.push_back(e);
        v = v_is_b ? group_of(e) : (e >> 4);
        v_is_b ^= 1;
        want = want == cg ? cb : cg;
      }
      for (int e : path) { at_g[(size_t)group_of(e) * NC + sigma[e]] = -1; at_b[(size_t)(e >> 4) * NC + sigma[e]] = -1; }
      for (int e : path) {
        sigma[e] = sigma[e] == cg ? cb : cg;
        at_g[(size_t)group_of(e) * NC + sigma[e]] = e; at_b[(size_t)(e >> 4) * NC + sigma[e]] = e;
      }
    }
    sigma[k] = cg;
    at_g[(size_t)g * NC + cg] = k; at_b[(size_t)b * NC + cg] = k;
  }
}

static int build_tables(DevTables &t)
{
  std::vector<uint16_t> bin; std::vector<int16_t> rel;
  host_freq_interleaver(bin, rel);
  if ((int)bin.size() != K) { set_error("frequency interleaver table has %zu entries", bin.size()); return DABX_E_ARG; }
  // PRS and coarse-CFO reference, phasetable.cpp:87-101, phasereference.cpp:58-66
  std::vector<float2> prs(TU, make_float2(0.f, 0.f)), argc(TU);
  for (int i = 1; i <= K / 2; i++) {
    const float p = prs_phi(i), m = prs_phi(-i);
    prs[i] = make_float2(cosf(p), sinf(p));
    prs[TU - i] = make_float2(cosf(m), sinf(m));
  }
  {
    std::vector<double> re(TU, 0.0), im(TU, 0.0);
    for (int i = 0; i < TU - 1; i++) {   // conj(f[i]) * f[i+1] in float like the reference, phasereference.cpp:293-297
      re[i] = prs[i].x * prs[i + 1].x + prs[i].y * prs[i + 1].y;
      im[i] = prs[i].x * prs[i + 1].y - prs[i].y * prs[i + 1].x;
    }
    host_dft(re, im, true);
    for (int i = 0; i < TU; i++) argc[i] = make_float2((float)re[i], -(float)im[i]);
  }
  // Twiddles e^{-j 2 pi i / 2048} (double, rounded once), stored in the order the passes of fft_core.h read them so that
  // the 64 lanes of a wave always load consecutive entries: [7][8] for the second pass (i = 32 t k), [7][64] for the third
  // (i = 4 t k), [3][512] for the last (i = m j).
  std::vector<float2> tw;
  tw.reserve(TU);
  auto W = [](int i) { i &= TU - 1; return make_float2((float)cos(2.0 * M_PI * i / TU), (float)-sin(2.0 * M_PI * i / TU)); };
  for (int t = 1; t < 8; t++) for (int k = 0; k < 8; k++) tw.push_back(W(32 * t * k));
  for (int t = 1; t < 8; t++) for (int k = 0; k < 64; k++) tw.push_back(W(4 * t * k));
  for (int m = 1; m < 4; m++) for (int j = 0; j < 512; j++) tw.push_back(W(m * j));
  tw.resize(TU, make_float2(0.f, 0.f));
  std::vector<uint16_t> ficm;
  host_fic_map(ficm);
  // PRBS x^9+x^5+1, all ones (fic_decoder.cpp:59-73, backend.cpp:72-84), packed MSB-first
  std::vector<uint32_t> prbs(288, 0);
  {
    uint8_t sr[9];
    memset(sr, 1, 9);
    uint8_t *pb = reinterpret_cast<uint8_t *>(prbs.data());
    for (int i = 0; i < 288 * 32; i++) {
      const uint8_t b = sr[8] ^ sr[4];
      memmove(sr + 1, sr, 8);
      sr[0] = b;
      pb[i >> 3] |= (uint8_t)(b << (7 - (i & 7)));
    }
  }
  // CRC tables
  std::vector<uint16_t> fctab(256), cctab(256);
  for (int i = 0; i < 256; i++) {
    uint16_t a = (uint16_t)(i << 8), c = (uint16_t)(i << 8);
    for (int j = 0; j < 8; j++) {
      a = (a & 0x8000) ? (uint16_t)((a << 1) ^ 0x782F) : (uint16_t)(a << 1);   // firecode_checker.h:53
      c = (c & 0x8000) ? (uint16_t)((c << 1) ^ 0x1021) : (uint16_t)(c << 1);   // crc.cpp:38
    }
    fctab[i] = a; cctab[i] = c;
  }
  // fire-code burst table, firecode_checker.cpp:61-144 (pattern list = data of firecode_checker.h:55-69)
  static const uint8_t pat[124] = {
      17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 30, 31, 34, 36, 38, 40, 42, 44, 46, 50, 52, 54, 56, 60, 62, 68, 72, 76, 84,
      88, 92, 100, 104, 108, 120, 124, 136, 152, 168, 184, 200, 216, 248, 33, 35, 37, 39, 41, 43, 45, 49, 51, 53, 55, 57, 59, 61,
      63, 66, 70, 74, 78, 82, 86, 90, 98, 102, 106, 110, 114, 118, 122, 126, 132, 140, 148, 156, 164, 172, 180, 196, 204, 212, 220,
      228, 236, 244, 252, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 29, 32, 48, 58, 64, 80, 96, 112, 116, 128, 144,
      160, 176, 192, 208, 224, 232, 240};
  std::vector<uint16_t> syn(65536, 0);
  {
    auto crc = [&](const uint8_t *x) {
      uint16_t c = 0;
      static const int ord[11] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 0, 1};
      for (int k = 0; k < 11; k++) c = (uint16_t)((c << 8) ^ fctab[(c >> 8) ^ x[ord[k]]]);
      return c;
    };
    uint8_t e[12] = {0};
    auto tryit = [&](int bit, int p) { const uint16_t s = crc(e); if (!syn[s]) syn[s] = (uint16_t)((bit << 8) + p); };
    for (int i = 0; i < 11; i++)
      for (int j = 0; j < 124; j++) { e[i] = pat[j]; tryit(i * 8, pat[j]); e[i] = 0; }
    static const int pass[3][3] = {{4, 0, 45}, {2, 45, 75}, {6, 60, 90}};
    for (auto &ps : pass)
      for (int i = 0; i < 10; i++)
        for (int j = ps[1]; j < ps[2]; j++) {
          e[i] = (uint8_t)(pat[j] >> ps[0]); e[i + 1] = (uint8_t)(pat[j] << (8 - ps[0]));
          tryit(i * 8 + ps[0], pat[j]);
          e[i] = e[i + 1] = 0;
        }
  }
  // GF(2^8), poly 0x11D (galois.cpp:37-66)
  std::vector<uint8_t> gexp(512, 0), glog(256, 0);
  {
    unsigned sr = 1;
    for (int i = 0; i < 255; i++) {
      glog[sr] = (uint8_t)i; gexp[i] = (uint8_t)sr;
      sr <<= 1;
      if (sr & 0x100) sr ^= 0x11D;
    }
    for (int i = 255; i < 510; i++) gexp[i] = gexp[i - 255];
    glog[0] = 255;
  }
  int rc;
  if ((rc = upload(&t.perm_bin, bin))) return rc;
  {
    std::vector<int16_t> inv(TU, (int16_t)-1);
    for (int k = 0; k < K; k++) inv[bin[k]] = (int16_t)k;
    if ((rc = upload(&t.bin_to_k, inv))) return rc;
    std::vector<int> sigma;
    host_carrier_slots(bin, sigma);
    std::vector<int16_t> slot8((size_t)TU);
    for (int tid = 0; tid < 256; tid++)
      for (int u = 0; u < 8; u++) {
        const int k = inv[tid + 256 * u];
        slot8[(size_t)tid * 8 + u] = (int16_t)(k < 0 ? -1 : ((k & ~15) | sigma[k]));
      }
    if ((rc = upload(&t.bin_to_slot8, slot8))) return rc;
    std::vector<uint32_t> rd(256, 0);
    for (int tid = 0; tid < 256; tid++)
      for (int u = 0; u < K / 256; u++) rd[tid] |= (uint32_t)sigma[tid + 256 * u] << (4 * u);
    if ((rc = upload(&t.carrier_slot_rd, rd))) return rc;
  }
  if ((rc = upload(&t.perm_rel, rel))) return rc;
  if ((rc = upload(&t.prs_ref, prs))) return rc;
  if ((rc = upload(&t.prs_arg_conj, argc))) return rc;
  if ((rc = upload(&t.twiddle, tw))) return rc;
  if ((rc = upload(&t.fic_map, ficm))) return rc;
  if ((rc = upload(&t.prbs_words, prbs))) return rc;
  if ((rc = upload(&t.fc_syndrome, syn))) return rc;
  if ((rc = upload(&t.fc_crctab, fctab))) return rc;
  if ((rc = upload(&t.crc_ccitt, cctab))) return rc;
  {
    std::vector<uint16_t> xp(1024);
    uint16_t st = 1;
    for (int m = 0; m < 1024; m++) { xp[m] = st; st = (uint16_t)(cctab[st >> 8] ^ (uint16_t)(st << 8)); }
    if ((rc = upload(&t.crc_xpow, xp))) return rc;
  }
  if ((rc = upload(&t.gf_exp, gexp))) return rc;
  if ((rc = upload(&t.gf_log, glog))) return rc;
  return 0;
}

int get_tables(const DevTables **out)
{
  int dev = 0;
  DABX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_tables.find(dev);
  if (it == g_tables.end()) {
    DevTables t{};
    const int rc = build_tables(t);
    if (rc) return rc;
    it = g_tables.emplace(dev, t).first;
  }
  *out = &it->second;
  return 0;
}

int get_profile_map(int kbps, int prot, int short_form, const uint16_t **dev_map, int *n_in)
{
  int dev = 0;
  DABX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  const auto key = std::make_tuple(dev, kbps, prot, short_form);
  auto it = g_maps.find(key);
  if (it == g_maps.end()) {
    std::vector<uint16_t> m;
    int n = 0;
    const int rc = host_profile_map(kbps, prot, short_form, m, &n);
    if (rc) { set_error("illegal profile kbps=%d prot=%d short=%d", kbps, prot, short_form); return rc; }
    m.resize(m.size() + 256, PUNCT);    // padded trellis steps of the last LDS block read past the end
    uint16_t *d = nullptr;
    const int rc2 = upload(&d, m);
    if (rc2) return rc2;
    it = g_maps.emplace(key, std::make_pair(d, n)).first;
  }
  *dev_map = it->second.first;
  *n_in = it->second.second;
  return 0;
}

}  // namespace dabx

// Host only, not part of include/dabx.h (tests/test_lds_layouts.py): the tables k_symbols uses for its frequency de-interleave
// through LDS -- bin_to_slot8[256][8] (LDS slot of the carrier of bin tid + 256 u, -1 for unused bins), carrier_slot_rd[256]
// (six 4-bit sigma per thread) and perm_bin[1536] (carrier -> FFT bin) -- exactly as build_tables uploads them.
extern "C" int dabx_internal_carrier_slots(int16_t *slot8, uint32_t *rd, uint16_t *perm_bin)
{
  using namespace dabx;
  if (!slot8 || !rd || !perm_bin) return DABX_E_ARG;
  std::vector<uint16_t> bin; std::vector<int16_t> rel;
  host_freq_interleaver(bin, rel);
  std::vector<int> sigma;
  host_carrier_slots(bin, sigma);
  std::vector<int16_t> inv(TU, (int16_t)-1);
  for (int k = 0; k < K; k++) { inv[bin[k]] = (int16_t)k; perm_bin[k] = bin[k]; }
  for (int tid = 0; tid < 256; tid++) {
    rd[tid] = 0;
    for (int u = 0; u < 8; u++) {
      const int k = inv[tid + 256 * u];
      slot8[tid * 8 + u] = (int16_t)(k < 0 ? -1 : ((k & ~15) | sigma[k]));
    }
    for (int u = 0; u < K / 256; u++) rd[tid] |= (uint32_t)sigma[tid + 256 * u] << (4 * u);
  }
  return 0;
}
