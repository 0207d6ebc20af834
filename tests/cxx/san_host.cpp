// san_host.cpp -- AddressSanitizer + UBSan run of libdabx's host-side parsers (the code that reads untrusted bytes:
// FIBs off the air, file headers).  Built by tests/cxx/Makefile with plain g++ from the host-only sources of
// dabstar_amd/csrc (fib.cpp, eti.cpp, tii.cpp, iqfile.cpp) plus the stubs below for the GPU launchers, which are never
// reached here.  GPU ASan is not available on the pool; this is the CPU-side sanitizer leg.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <unistd.h>
#include "dabx.h"
#include "../../dabstar_amd/csrc/iqfile.h"

// ---- stubs: device side (never called: no engine, no conversion is started) ------------------------------------------
namespace dabx {
static std::string g_err;
void set_error(const char *fmt, ...) { g_err = fmt; }
int hip_fail(hipError_t, const char *, const char *, int) { return DABX_E_HIP; }
int launch_decode_iq(const uint8_t *, const IqDecode &, float2 *, unsigned long long, int, size_t, hipStream_t) { abort(); }
int launch_resample_1ms(const float2 *, int, const int16_t *, const float *, float2 *, unsigned long long, int, size_t, hipStream_t) { abort(); }
int host_profile_map(int, int, int, std::vector<uint16_t> &, int *) { abort(); }
int prs_quarter_turns(int k) { return (k * 7 + 3) & 3; }      // any table will do for memory-safety purposes
}  // namespace dabx
extern "C" int dabx_internal_ring_info(dabx_engine *, int, float2 **, int *, unsigned long long *, unsigned long long *, hipStream_t *) { abort(); }
extern "C" int dabx_commit_iq(dabx_engine *, int, size_t) { abort(); }
extern "C" int dabx_internal_commit(dabx_engine *, int, size_t) { abort(); }
extern "C" const char *dabx_last_error(void) { return dabx::g_err.c_str(); }

static std::mt19937 rng(12345);
static int rnd(int n) { return (int)(rng() % (unsigned)n); }

static void write_file(const std::string &p, const std::vector<uint8_t> &b)
{
  FILE *f = fopen(p.c_str(), "wb");
  fwrite(b.data(), 1, b.size(), f);
  fclose(f);
}

int main()
{
  // 1. FIB walk on random and structured-random FIBs
  long found = 0;
  for (int it = 0; it < 30000; it++) {
    uint8_t fibs[12 * 32], crc[12];
    for (auto &b : fibs) b = (uint8_t)rng();
    for (int i = 0; i < 12; i++) {
      crc[i] = rnd(8) != 0;
      if (rnd(2)) fibs[32 * i] = (uint8_t)rnd(32);               // FIG type 0 with a random length
      if (rnd(2)) fibs[32 * i + 1] = (uint8_t)rnd(3);            // extension 0..2
    }
    dabx_subch_desc out[64];
    int32_t cif = 0;
    const int n = dabx_parse_fibs(fibs, crc, 12, out, rnd(65), &cif);
    if (n < 0) return 10;
    found += n;
  }
  // 2. ETI assembly
  for (int it = 0; it < 3000; it++) {
    const int n = rnd(19);
    std::vector<dabx_subch_desc> sc((size_t)n);
    std::vector<std::vector<uint8_t>> data((size_t)n);
    std::vector<const uint8_t *> ptr((size_t)n);
    for (int i = 0; i < n; i++) {
      sc[(size_t)i] = dabx_subch_desc{rnd(64), rnd(864), rnd(200), 8 * (1 + rnd(48)), rnd(8), rnd(2), 1, 0};
      data[(size_t)i].assign((size_t)sc[(size_t)i].kbps * 3, (uint8_t)it);
      ptr[(size_t)i] = data[(size_t)i].data();
    }
    uint8_t fic[96] = {0}, out[6144];
    (void)dabx_eti_frame(rnd(25), rnd(256), rnd(4), sc.data(), n, fic, ptr.data(), out);
  }
  // 3. container probing on mutated headers
  char tmpl[] = "/tmp/dabx_san_XXXXXX";
  const std::string dir = mkdtemp(tmpl);
  const char *uff = "<?xml version=\"1.0\"?>\n<SDR><Sample><Samplerate Unit=\"Hz\" Value=\"2048000\"/><Channels Bits=\"16\" Container=\"int16\" "
                    "Ordering=\"LSB\" Amount=\"2\"><Channel Value=\"I\"/><Channel Value=\"Q\"/></Channels></Sample><Datablocks>"
                    "<Datablock Count=\"1000\" Number=\"1\" Channel=\"Channel\"><Frequency Value=\"1\" Unit=\"KHz\"/></Datablock></Datablocks></SDR>\n";
  std::vector<uint8_t> wav = {'R', 'I', 'F', 'F', 0x24, 0x10, 0, 0, 'W', 'A', 'V', 'E', 'f', 'm', 't', ' ', 16, 0, 0, 0, 1, 0, 2, 0,
                              0x00, 0x40, 0x1F, 0x00, 0x00, 0x00, 0x7D, 0x00, 4, 0, 16, 0, 'd', 'a', 't', 'a', 0x00, 0x10, 0, 0};
  wav.resize(wav.size() + 4096, 7);
  int accepted = 0;
  for (int it = 0; it < 4000; it++) {
    std::vector<uint8_t> b;
    const bool is_wav = it & 1;
    if (is_wav) b = wav;
    else { b.assign(uff, uff + strlen(uff)); b.resize(b.size() + 600 + (size_t)rnd(3000), 0); b.resize(b.size() + (size_t)rnd(5000), 0x55); }
    const int muts = rnd(6);
    for (int m = 0; m < muts; m++) {
      const size_t lim = std::min<size_t>(is_wav ? 48 : strlen(uff), b.size());
      const int kind = rnd(4);
      if (kind == 1) b.resize((size_t)rnd((int)b.size() + 1));
      else if (lim == 0) continue;
      else if (kind == 0) b[(size_t)rnd((int)lim)] = (uint8_t)rng();
      else if (kind == 2) b.erase(b.begin() + rnd((int)lim));
      else b.insert(b.begin() + rnd((int)lim + 1), (uint8_t)("<>\"'=/ &\0x"[rnd(10)]));
    }
    const std::string p = dir + (is_wav ? "/t.wav" : "/t.uff");
    write_file(p, b);
    dabx_iq_format fmt;
    if (dabx_probe_iq_file(p.c_str(), &fmt) == 0) {
      accepted++;
      if (fmt.data_bytes < 0 || fmt.data_offset < 0) return 11;
      if (fmt.data_bytes > 0 && fmt.data_offset + fmt.data_bytes > (long long)b.size()) return 12;
    }
    unlink(p.c_str());
  }
  rmdir(dir.c_str());
  // 4. TII detector on random spectra
  dabx_tii *t = nullptr;
  if (dabx_tii_create(&t)) return 13;
  dabx_tii_set_collisions(t, 1, 3);
  long tii_hits = 0;
  for (int it = 0; it < 300; it++) {
    std::vector<float> z(4096);
    for (auto &v : z) v = (float)((int)(rng() % 2001) - 1000) * (rnd(50) == 0 ? 50.0f : 0.01f);
    dabx_tii_add(t, z.data());
    dabx_tii_result r[8];
    const int n = dabx_tii_process(t, rnd(12), r, rnd(9));
    if (n < 0) return 14;
    tii_hits += n;
  }
  dabx_tii_destroy(t);
  std::printf("san_host ok: %ld sub-channels parsed, %d containers accepted, %ld tii results\n", found, accepted, tii_hits);
  return 0;
}
