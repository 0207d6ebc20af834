// iqfile.cpp -- recorded-IQ containers (host side): probing of .raw/.iq, RIFF/WAVE (.sdr/.wav) and .uff headers,
// resampler tables, and the streaming feed that moves payload bytes through iqfile.hip into a stream's IQ ring.
// Reference behaviour: devices/filereaders/{raw_files,wav_files,xml_filereader}; see include/dabx.h for the rules.
#include "iqfile.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>

using namespace dabx;

// ------------------------------------------------------------------------------------------------ format rules
static int channel_bytes(int container)
{
  static const int b[6] = {1, 1, 2, 3, 4, 4};
  return container >= 0 && container < 6 ? b[container] : 0;
}

static int check_format(const dabx_iq_format *f, IqDecode *d);
int dabx::iq_check_format(const dabx_iq_format *f, IqDecode *d) { return check_format(f, d); }
static int check_format(const dabx_iq_format *f, IqDecode *d)
{
  if (!f || channel_bytes(f->container) == 0 || f->family < DABX_FAMILY_RAW || f->family > DABX_FAMILY_UFF) {
    set_error("iq format: bad family/container");
    return DABX_E_ARG;
  }
  if (f->family == DABX_FAMILY_RAW && f->container != DABX_C_U8) { set_error("iq format: RAW files are uint8"); return DABX_E_ARG; }
  if (f->sample_rate != INPUT_RATE && (f->sample_rate < 1536000 || f->sample_rate > 3000000 || f->family == DABX_FAMILY_RAW)) {
    set_error("iq format: sample rate %d not supported (wavfiles.cpp:70)", f->sample_rate);
    return DABX_E_ARG;
  }
  d->family = f->family; d->container = f->container; d->big_endian = f->big_endian != 0; d->swap_iq = f->swap_iq != 0;
  d->bytes = channel_bytes(f->container);
  d->int_scale = 1.0f;
  if (f->container >= DABX_C_I16 && f->container <= DABX_C_I32) {
    int bits = f->family == DABX_FAMILY_UFF ? f->bits : 8 * d->bytes;
    if (bits < 1 || bits > 32) { set_error("iq format: Bits=%d", bits); return DABX_E_ARG; }
    // xml_reader.cpp:43-51 shift(): 1 << (bits - 1) in 32-bit int arithmetic (Bits=32 wraps to -2^31, i.e. the sign flips)
    const int32_t sc = (int32_t)(1u << (bits - 1));
    d->int_scale = f->family == DABX_FAMILY_UFF ? 1.0f / (float)sc : ldexpf(1.0f, 1 - bits);
  }
  d->quirk_block = 0; d->quirk_i24 = 0; d->quirk_sign7f = 0;
  if (f->reference_quirks && f->family == DABX_FAMILY_UFF) {
    // like the reference's reader, deliver whole 1-ms read blocks only (readSamples, xml_reader.cpp:224-227)
    d->quirk_block = (int16_t)(f->sample_rate / 1000);
    if (f->container == DABX_C_F32 && f->swap_iq) d->swap_iq = 0;                 // xml_reader.cpp:530,540: no swap
    if (f->container == DABX_C_I24 && f->big_endian) {                            // :316 / :462, :465-469
      d->quirk_i24 = 1;
      d->quirk_sign7f = f->swap_iq ? 1 : 0;
    }
    if (f->container == DABX_C_U8 && f->swap_iq) {                                // :423
      set_error("iq format: UFF QI/uint8 -- the reference indexes its 256-entry table with the loop counter "
                "(xml_reader.cpp:423) and reads past it from the 128th sample of every block: undefined, not reproducible");
      return DABX_E_ARG;
    }
  }
  return 0;
}

int dabx_iq_sample_bytes(const dabx_iq_format *fmt)
{
  const int b = fmt ? channel_bytes(fmt->container) : 0;
  return b ? 2 * b : DABX_E_ARG;
}

// Resampler tables.  WAV flavour wav_reader.cpp:67-82, UFF flavour xml_reader.cpp:76-81.
static void resample_tables(int family, int rate, int *M, std::vector<int16_t> &ti, std::vector<float> &tf);
void dabx::iq_resample_tables(int family, int rate, int *M, int16_t *tab_int, float *tab_frac)
{
  std::vector<int16_t> ti; std::vector<float> tf;
  resample_tables(family, rate, M, ti, tf);
  memcpy(tab_int, ti.data(), 2048 * sizeof(int16_t)); memcpy(tab_frac, tf.data(), 2048 * sizeof(float));
}
static void resample_tables(int family, int rate, int *M, std::vector<int16_t> &ti, std::vector<float> &tf)
{
  ti.resize(2048); tf.resize(2048);
  *M = (int16_t)(rate / 1000);
  for (int i = 0; i < 2048; i++) {
    if (family == DABX_FAMILY_WAV) {
      const float in_val = (float)rate / 1000.0f;
      const float pos = (float)i * (in_val / 2048.0f);
      ti[i] = (int16_t)floorf(pos);
      tf[i] = pos - (float)ti[i];
    } else {
      const float in_val = (float)(rate / 1000);
      ti[i] = (int16_t)floor(i * ((double)in_val / 2048.0));
      tf[i] = (float)i * (in_val / 2048.0f) - (float)ti[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------ probing
static uint32_t rd_le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t rd_be32(const uint8_t *p) { return p[3] | (p[2] << 8) | (p[1] << 16) | ((uint32_t)p[0] << 24); }

static int probe_wav(FILE *fp, long long file_len, dabx_iq_format *out)
{
  uint8_t h[12];
  fseek(fp, 0, SEEK_SET);
  if (fread(h, 1, 12, fp) != 12 || memcmp(h + 8, "WAVE", 4)) { set_error("not a WAVE file"); return DABX_E_ARG; }
  const bool be = !memcmp(h, "RIFX", 4);
  auto r32 = [&](const uint8_t *p) { return be ? rd_be32(p) : rd_le32(p); };
  auto r16 = [&](const uint8_t *p) { return be ? (p[0] << 8 | p[1]) : (p[1] << 8 | p[0]); };
  long long pos = 12;
  int tag = -1, channels = 0, bits = 0, rate = 0;
  for (;;) {
    uint8_t ch[8];
    fseek(fp, (long)pos, SEEK_SET);
    if (fread(ch, 1, 8, fp) != 8) { set_error("WAVE: no data chunk"); return DABX_E_ARG; }
    const long long len = r32(ch + 4);
    if (!memcmp(ch, "fmt ", 4)) {
      uint8_t f[40] = {0};
      const size_t want = (size_t)std::min<long long>(len, 40);
      if (fread(f, 1, want, fp) != want || want < 16) { set_error("WAVE: short fmt chunk"); return DABX_E_ARG; }
      tag = r16(f); channels = r16(f + 2); rate = (int)r32(f + 4); bits = r16(f + 14);
      if (tag == 0xFFFE && want >= 26) tag = r16(f + 24);       // WAVE_FORMAT_EXTENSIBLE: sub-format GUID starts with the tag
    } else if (!memcmp(ch, "data", 4)) {
      if (tag < 0) { set_error("WAVE: data before fmt"); return DABX_E_ARG; }
      memset(out, 0, sizeof(*out));
      out->family = DABX_FAMILY_WAV; out->big_endian = be; out->sample_rate = rate; out->bits = bits;
      if (tag == 1 && bits == 8) out->container = DABX_C_U8;
      else if (tag == 1 && bits == 16) out->container = DABX_C_I16;
      else if (tag == 1 && bits == 24) out->container = DABX_C_I24;
      else if (tag == 1 && bits == 32) out->container = DABX_C_I32;
      else if (tag == 3 && bits == 32) out->container = DABX_C_F32;
      else { set_error("WAVE: format tag %d with %d bits is not supported (wavfiles.cpp:77-95)", tag, bits); return DABX_E_ARG; }
      if (rate < 1536000 || rate > 3000000 || channels != 2) {   // wavfiles.cpp:70-76
        set_error("WAVE: sample rate %d (1536..3000 kS/s) or channel count %d (2) is not supported", rate, channels);
        return DABX_E_ARG;
      }
      out->data_offset = pos + 8;
      long long avail = file_len - out->data_offset;
      out->data_bytes = (len == 0xFFFFFFFFll || len > avail) ? avail : len;   // streamed writers leave the size open
      out->data_bytes -= out->data_bytes % (2 * channel_bytes(out->container));
      return 0;
    }
    pos += 8 + len + (len & 1);
  }
}

// ---- a small XML reader for the .uff header: element tree with attributes (what QDomDocument gives the reference).
// Well-formedness is enforced the way QDomDocument::setContent does for these headers: a document that does not parse
// yields no elements at all, i.e. the defaults and no data block (xml_descriptor.cpp:127-129, 240).
namespace {
struct XmlNode {
  std::string name;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<XmlNode> kids;
  std::string attr(const char *key, const char *dflt) const
  {
    for (const auto &a : attrs) if (a.first == key) return a.second;
    return dflt;
  }
};
struct XmlParser {
  const std::string &d;
  size_t p = 0;
  explicit XmlParser(const std::string &doc) : d(doc) {}
  static bool name_char(char c) { return isalnum((unsigned char)c) || c == '_' || c == '-' || c == '.' || c == ':'; }
  void skip_ws() { while (p < d.size() && isspace((unsigned char)d[p])) p++; }
  bool skip_misc()      // whitespace, comments, processing instructions, DOCTYPE
  {
    for (;;) {
      skip_ws();
      if (d.compare(p, 4, "<!--") == 0) { const size_t e = d.find("-->", p + 4); if (e == std::string::npos) return false; p = e + 3; }
      else if (d.compare(p, 2, "<?") == 0) { const size_t e = d.find("?>", p + 2); if (e == std::string::npos) return false; p = e + 2; }
      else if (d.compare(p, 9, "<!DOCTYPE") == 0) { const size_t e = d.find('>', p); if (e == std::string::npos) return false; p = e + 1; }
      else return true;
    }
  }
  static std::string unescape(const std::string &v)
  {
    std::string o;
    for (size_t i = 0; i < v.size(); i++) {
      if (v[i] != '&') { o.push_back(v[i]); continue; }
      static const char *ent[5][2] = {{"&amp;", "&"}, {"&lt;", "<"}, {"&gt;", ">"}, {"&quot;", "\""}, {"&apos;", "'"}};
      bool hit = false;
      for (auto &e : ent) if (v.compare(i, strlen(e[0]), e[0]) == 0) { o += e[1]; i += strlen(e[0]) - 1; hit = true; break; }
      if (!hit) o.push_back('&');
    }
    return o;
  }
  // Parses one element into parent.kids.  Like Qt's SAX-driven QDomDocument builder, a node is attached as soon as its
  // start tag is complete, and a later syntax error stops the parse but leaves everything attached so far in place.
  bool element(XmlNode &parent, int depth)
  {
    if (depth > 32 || p >= d.size() || d[p] != '<') return false;
    p++;
    const size_t n0 = p;
    while (p < d.size() && name_char(d[p])) p++;
    if (p == n0) return false;
    XmlNode n;
    n.name = d.substr(n0, p - n0);
    bool empty = false;
    for (;;) {                                            // attributes
      const size_t before = p;
      skip_ws();
      if (p >= d.size()) return false;
      if (d[p] == '/') { if (p + 1 < d.size() && d[p + 1] == '>') { p += 2; empty = true; break; } return false; }
      if (d[p] == '>') { p++; break; }
      if (p == before) return false;                      // attributes must be separated by white space
      const size_t a0 = p;
      while (p < d.size() && name_char(d[p])) p++;
      if (p == a0) return false;
      const std::string key = d.substr(a0, p - a0);
      skip_ws();
      if (p >= d.size() || d[p] != '=') return false;
      p++;
      skip_ws();
      if (p >= d.size() || (d[p] != '"' && d[p] != '\'')) return false;
      const char q = d[p++];
      const size_t e = d.find(q, p);
      if (e == std::string::npos) return false;
      const std::string val = d.substr(p, e - p);
      if (val.find('<') != std::string::npos) return false;
      for (const auto &a : n.attrs) if (a.first == key) return false;      // duplicate attribute
      n.attrs.emplace_back(key, unescape(val));
      p = e + 1;
    }
    parent.kids.push_back(std::move(n));
    if (empty) return true;
    const size_t me = parent.kids.size() - 1;
    for (;;) {                                            // content
      const size_t lt = d.find('<', p);
      if (lt == std::string::npos) return false;
      p = lt;
      if (d.compare(p, 4, "<!--") == 0) { const size_t e = d.find("-->", p + 4); if (e == std::string::npos) return false; p = e + 3; continue; }
      if (d.compare(p, 9, "<![CDATA[") == 0) { const size_t e = d.find("]]>", p); if (e == std::string::npos) return false; p = e + 3; continue; }
      if (d.compare(p, 2, "<?") == 0) { const size_t e = d.find("?>", p + 2); if (e == std::string::npos) return false; p = e + 2; continue; }
      if (d.compare(p, 2, "</") == 0) {
        const std::string &name = parent.kids[me].name;
        p += 2;
        if (d.compare(p, name.size(), name) != 0 || (p + name.size() < d.size() && name_char(d[p + name.size()]))) return false;
        p += name.size();
        skip_ws();
        if (p >= d.size() || d[p] != '>') return false;
        p++;
        return true;
      }
      if (!element(parent.kids[me], depth + 1)) return false;
    }
  }
  // Fills doc.kids[0] with the root element (as far as the text is well formed); false if there is no root at all.
  bool document(XmlNode &doc)
  {
    if (!skip_misc()) return false;
    (void)element(doc, 0);
    return !doc.kids.empty();
  }
};
}  // namespace

static int probe_uff(FILE *fp, long long file_len, dabx_iq_format *out)
{
  // xml_descriptor.cpp:111-125: the header text ends where 500 consecutive zero bytes have been seen
  std::string doc;
  fseek(fp, 0, SEEK_SET);
  int zeros = 0, c;
  while (zeros < 500 && (c = fgetc(fp)) != EOF) {
    zeros = c == 0 ? zeros + 1 : 0;
    if (c) doc.push_back((char)c);
    if (doc.size() > (1u << 20)) break;
  }
  // defaults of xml_descriptor.cpp:103-110, attribute defaults of :160-190
  int rate = 2048000, bits = 16, channels = 2, n_blocks = 0;
  long long n_elements = 0;
  std::string container = "i16", ordering = "MSB", iq_order = "IQ";
  XmlNode tree;
  XmlParser px(doc);
  if (px.document(tree)) {
    for (const XmlNode &comp : tree.kids[0].kids) {
      if (comp.name == "Sample") {
        for (const XmlNode &ch : comp.kids) {
          if (ch.name == "Samplerate") {
            const std::string unit = ch.attr("Unit", "Hz");
            const int factor = unit == "Hz" ? 1 : (unit == "KHz" || unit == "Khz") ? 1000 : 1000000;
            const long long r = (long long)atoi(ch.attr("Value", "2048000").c_str()) * factor;   // the reference multiplies in int
            rate = (r < 0 || r > 2000000000ll) ? 0 : (int)r;                                        // 0: refused at feed time
          } else if (ch.name == "Channels") {
            channels = atoi(ch.attr("Amount", "2").c_str());
            bits = atoi(ch.attr("Bits", "8").c_str());
            container = ch.attr("Container", "u8");
            ordering = ch.attr("Ordering", "N/A");
            int k = 0;
            for (const XmlNode &cc : ch.kids) {             // xml_descriptor.cpp:58-75
              if (cc.name != "Channel") continue;
              const std::string v = cc.attr("Value", "I");
              if (k == 0) iq_order = v == "I" ? "I_ONLY" : "Q_ONLY";
              else if (k == 1 && iq_order == "I_ONLY" && v == "Q") iq_order = "IQ";
              else if (k == 1 && iq_order == "Q_ONLY" && v == "I") iq_order = "QI";
              k++;
            }
          }
        }
      } else if (comp.name == "Datablocks") {
        n_blocks = 0; n_elements = 0;
        for (const XmlNode &b : comp.kids)
          if (b.name == "Datablock") { n_elements += atoll(b.attr("Count", "100").c_str()); n_blocks++; }
      }
    }
  }
  if (n_blocks == 0) { set_error("uff: header does not parse or names no Datablock"); return DABX_E_ARG; }   // xml_descriptor.cpp:240
  if (channels != 2 || (iq_order != "IQ" && iq_order != "QI")) {
    set_error("uff: %d channel(s), order %s: only I+Q recordings are supported", channels, iq_order.c_str());
    return DABX_E_ARG;
  }
  memset(out, 0, sizeof(*out));
  out->family = DABX_FAMILY_UFF; out->sample_rate = rate; out->bits = bits;
  out->big_endian = ordering == "MSB"; out->swap_iq = iq_order == "QI";
  if (container == "int8") out->container = DABX_C_S8;
  else if (container == "uint8") out->container = DABX_C_U8;
  else if (container == "int16") out->container = DABX_C_I16;
  else if (container == "int24") out->container = DABX_C_I24;
  else if (container == "int32") out->container = DABX_C_I32;
  else if (container == "float32") out->container = DABX_C_F32;
  else { set_error("uff: container '%s' is not supported", container.c_str()); return DABX_E_ARG; }
  // xml_filereader.cpp:113-127: the payload is the tail of the file
  const int bc = channel_bytes(out->container);
  long long start = file_len - n_elements * bc;
  if (start < 2048 || start > 1000000) start = 2048;
  out->data_offset = start;
  out->data_bytes = std::max(0ll, file_len - start);
  out->data_bytes -= out->data_bytes % (2 * bc);
  return 0;
}

int dabx_probe_iq_file(const char *path, dabx_iq_format *fmt)
{
  if (!path || !fmt) { set_error("dabx_probe_iq_file: bad argument"); return DABX_E_ARG; }
  FILE *fp = fopen(path, "rb");
  if (!fp) { set_error("cannot open '%s'", path); return DABX_E_ARG; }
  fseek(fp, 0, SEEK_END);
  const long long len = ftell(fp);
  uint8_t magic[16] = {0};
  fseek(fp, 0, SEEK_SET);
  const size_t got = fread(magic, 1, sizeof(magic), fp);
  int rc;
  if (got >= 12 && (!memcmp(magic, "RIFF", 4) || !memcmp(magic, "RIFX", 4))) rc = probe_wav(fp, len, fmt);
  else if (got >= 5 && (!memcmp(magic, "<?xml", 5) || !memcmp(magic, "<SDR", 4))) rc = probe_uff(fp, len, fmt);
  else {
    const char *dot = strrchr(path, '.');
    if (dot && (!strcasecmp(dot, ".raw") || !strcasecmp(dot, ".iq"))) {
      memset(fmt, 0, sizeof(*fmt));
      fmt->family = DABX_FAMILY_RAW; fmt->container = DABX_C_U8; fmt->bits = 8; fmt->sample_rate = INPUT_RATE;
      fmt->data_offset = 0; fmt->data_bytes = len & ~1ll;
      rc = 0;
    } else {
      set_error("'%s': neither RIFF/WAVE, .uff XML nor a .raw/.iq file", path);
      rc = DABX_E_ARG;
    }
  }
  fclose(fp);
  return rc;
}

// ------------------------------------------------------------------------------------------------ feed
struct dabx_feed {
  dabx_engine *eng = nullptr;      // nullptr: one-shot conversion into `lin`
  int stream = 0;
  dabx_iq_format fmt{};
  IqDecode dec{};
  bool resample = false;
  int M = 0;
  int16_t *tab_int = nullptr;
  float *tab_frac = nullptr;
  uint8_t *stage = nullptr; size_t stage_cap = 0;
  float2 *work = nullptr; size_t work_cap = 0;       // [carry | decoded block]
  float2 *carry = nullptr; int carry_n = 0;          // <= M + 1 samples kept between calls
  std::vector<uint8_t> odd;                          // bytes of an incomplete sample
  hipStream_t st = nullptr;
  float2 *lin = nullptr; size_t lin_cap = 0, lin_n = 0;

  ~dabx_feed()
  {
    for (void *p : {(void *)tab_int, (void *)tab_frac, (void *)stage, (void *)work, (void *)carry}) if (p) (void)hipFree(p);
  }
};

static int feed_setup(dabx_feed *f, const dabx_iq_format *fmt)
{
  int rc = check_format(fmt, &f->dec);
  if (rc) return rc;
  f->fmt = *fmt;
  f->resample = fmt->sample_rate != INPUT_RATE;
  if (f->resample) {
    std::vector<int16_t> ti; std::vector<float> tf;
    resample_tables(fmt->family, fmt->sample_rate, &f->M, ti, tf);
    DABX_HIP(hipMalloc((void **)&f->tab_int, 2048 * sizeof(int16_t)));
    DABX_HIP(hipMalloc((void **)&f->tab_frac, 2048 * sizeof(float)));
    DABX_HIP(hipMalloc((void **)&f->carry, (size_t)(f->M + 1) * sizeof(float2)));
    DABX_HIP(hipMemcpy(f->tab_int, ti.data(), 2048 * sizeof(int16_t), hipMemcpyHostToDevice));
    DABX_HIP(hipMemcpy(f->tab_frac, tf.data(), 2048 * sizeof(float), hipMemcpyHostToDevice));
    // xml_reader.cpp:84-85,226: convBuffer[0] starts as a zero sample; wav_reader.cpp:192-206 primes with the first sample
    f->carry_n = fmt->family == DABX_FAMILY_UFF ? 1 : 0;
    DABX_HIP(hipMemset(f->carry, 0, (size_t)(f->M + 1) * sizeof(float2)));
  }
  return 0;
}

static long long bound_samples(const dabx_feed *f, size_t n_bytes)
{
  size_t n = (n_bytes + f->odd.size()) / (size_t)(2 * f->dec.bytes);
  if (f->dec.quirk_block) n -= n % (size_t)f->dec.quirk_block;
  if (!f->resample) return (long long)n;
  return (long long)(((size_t)f->carry_n + n) / (size_t)f->M) * 2048;
}
long long dabx_feed_bound(const dabx_feed *f, size_t n_bytes) { return f ? bound_samples(f, n_bytes) : DABX_E_ARG; }

static long long feed_push(dabx_feed *f, const uint8_t *bytes, size_t n_bytes)
{
  const size_t sb = (size_t)(2 * f->dec.bytes);
  // complete samples only; the tail bytes wait for the next call
  std::vector<uint8_t> joined;
  const uint8_t *src = bytes;
  size_t total = n_bytes;
  if (!f->odd.empty()) {
    joined = f->odd;
    joined.insert(joined.end(), bytes, bytes + n_bytes);
    src = joined.data(); total = joined.size();
  }
  size_t n = total / sb;
  if (f->dec.quirk_block) n -= n % (size_t)f->dec.quirk_block;      // whole read blocks of the reference's reader only
  f->odd.assign(src + n * sb, src + total);
  if (n == 0) return 0;

  float2 *dst; int dst_len; unsigned long long dst0;
  if (f->eng) {
    float2 *ring; int ring_len; unsigned long long wr, rd;
    int rc = dabx_internal_ring_info(f->eng, f->stream, &ring, &ring_len, &wr, &rd, &f->st);
    if (rc) return rc;
    const long long out_n = f->resample ? (long long)(((size_t)f->carry_n + n) / (size_t)f->M) * 2048 : (long long)n;
    if ((long long)(wr - rd) + out_n > ring_len) {
      f->odd.clear();
      if (!joined.empty()) f->odd.assign(joined.begin(), joined.begin() + (joined.size() - n_bytes));   // as before the call
      set_error("dabx_feed_bytes: ring of stream %d has room for %lld samples, %lld offered", f->stream, (long long)ring_len - (long long)(wr - rd), out_n);
      return DABX_E_STATE;
    }
    dst = ring; dst_len = ring_len; dst0 = wr;
  } else {
    dst = f->lin; dst_len = 0; dst0 = f->lin_n;
  }
  if (n * sb > f->stage_cap) {
    if (f->stage) DABX_HIP(hipFree(f->stage));
    f->stage = nullptr; f->stage_cap = 0;
    DABX_HIP(hipMalloc((void **)&f->stage, n * sb));
    f->stage_cap = n * sb;
  }
  DABX_HIP(hipMemcpyAsync(f->stage, src, n * sb, hipMemcpyHostToDevice, f->st));
  long long produced;
  int rc;
  if (!f->resample) {
    if (!f->eng && f->lin_n + n > f->lin_cap) { set_error("dabx_convert_iq_bytes: output buffer too small"); return DABX_E_ARG; }
    if ((rc = launch_decode_iq(f->stage, f->dec, dst, dst0, dst_len, n, f->st))) return rc;
    produced = (long long)n;
  } else {
    const size_t len = (size_t)f->carry_n + n;
    if (len > f->work_cap) {
      if (f->work) DABX_HIP(hipFree(f->work));
      f->work = nullptr; f->work_cap = 0;
      DABX_HIP(hipMalloc((void **)&f->work, len * sizeof(float2)));
      f->work_cap = len;
    }
    if (f->carry_n) DABX_HIP(hipMemcpyAsync(f->work, f->carry, (size_t)f->carry_n * sizeof(float2), hipMemcpyDeviceToDevice, f->st));
    if ((rc = launch_decode_iq(f->stage, f->dec, f->work, (unsigned long long)f->carry_n, 0, n, f->st))) return rc;
    // block c needs V[c M .. c M + M]
    const size_t blocks = len >= (size_t)f->M + 1 ? (len - 1) / (size_t)f->M : 0;
    produced = (long long)blocks * 2048;
    if (!f->eng && f->lin_n + (size_t)produced > f->lin_cap) { set_error("dabx_convert_iq_bytes: output buffer too small"); return DABX_E_ARG; }
    if ((rc = launch_resample_1ms(f->work, f->M, f->tab_int, f->tab_frac, dst, dst0, dst_len, (size_t)produced, f->st))) return rc;
    const size_t keep = len - blocks * (size_t)f->M;          // 1 .. M samples (0 only before the first WAV sample)
    if (keep) DABX_HIP(hipMemcpyAsync(f->carry, f->work + blocks * (size_t)f->M, keep * sizeof(float2), hipMemcpyDeviceToDevice, f->st));
    f->carry_n = (int)keep;
  }
  if (f->eng) {
    if (produced && (rc = dabx_internal_commit(f->eng, f->stream, (size_t)produced))) return rc;
  } else f->lin_n += (size_t)produced;
  DABX_HIP(hipStreamSynchronize(f->st));     // `src` (caller memory / joined) and the stage buffer are free again
  return produced;
}

int dabx_feed_open(dabx_engine *e, int stream, const dabx_iq_format *fmt, dabx_feed **out)
{
  if (!e || !fmt || !out) { set_error("dabx_feed_open: bad argument"); return DABX_E_ARG; }
  float2 *ring; int ring_len; unsigned long long wr, rd; hipStream_t st;
  int rc = dabx_internal_ring_info(e, stream, &ring, &ring_len, &wr, &rd, &st);
  if (rc) return rc;
  dabx_feed *f = new dabx_feed();
  f->eng = e; f->stream = stream; f->st = st;
  if ((rc = feed_setup(f, fmt))) { delete f; return rc; }
  *out = f;
  return 0;
}

long long dabx_feed_bytes(dabx_feed *f, const void *bytes, size_t n_bytes)
{
  if (!f || (!bytes && n_bytes)) { set_error("dabx_feed_bytes: bad argument"); return DABX_E_ARG; }
  return feed_push(f, static_cast<const uint8_t *>(bytes), n_bytes);
}

void dabx_feed_close(dabx_feed *f) { delete f; }

long long dabx_convert_iq_bytes(const dabx_iq_format *fmt, const void *bytes, size_t n_bytes, float *iq_out, size_t max_out)
{
  if (!fmt || (!bytes && n_bytes) || (!iq_out && max_out)) { set_error("dabx_convert_iq_bytes: bad argument"); return DABX_E_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device"); return DABX_E_NODEVICE; }
  dabx_feed f;
  int rc = feed_setup(&f, fmt);
  if (rc) return rc;
  const long long bound = bound_samples(&f, n_bytes);
  if (bound == 0) return 0;
  DABX_HIP(hipMalloc((void **)&f.lin, (size_t)bound * sizeof(float2)));
  f.lin_cap = std::min<size_t>((size_t)bound, max_out);
  long long n = feed_push(&f, static_cast<const uint8_t *>(bytes), n_bytes);
  if (n > 0 && hipMemcpy(iq_out, f.lin, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess) n = DABX_E_HIP;
  (void)hipFree(f.lin);
  return n;
}
