// shim_symbols.cpp -- drives ONE ensemble symbol by symbol through the three HIP class shims exactly as
// DabProcessor::_state_process_rest_of_frame does on the reference classes (dab_processor.cpp:199-202, :336-360, :267-286):
//   OfdmDecoder::store_reference_symbol_0 -> 75 x decode_symbol -> FicDecoder::process_block (symbols 1..3) /
//   MscHandler::process_block (symbols 4..75) -> store_null_symbol_with(out)_tii.
// Input (argv[1]): int32 n_frames, n_services; per service int32 {SubChId, CuStartAddr, CuSize, bitRate, protLevel,
// shortForm, set_at_frame, stop_at_frame}; per frame: 76 x 2048 cf32 FFT outputs, 2048 cf32 null-symbol FFT, f32 clock error.
// Output (argv[2] prefix): <p>.fibs (u16 fic + 32 bytes per process_FIB call), <p>.fibbits (per frame 3072 bits + 4 flags
// from get_fib_bits), <p>.svc<i> (add_to_frame vectors packed to bytes), one JSON line on stdout.
#define DABX_SHIM_STANDALONE "shim_env_standalone.h"
#include "ofdm_decoder_hip.h"
#include "fic_decoder_hip.h"
#include "msc_handler_hip.h"
#include <cstdio>
#include <cstring>
#include <string>

struct Svc { int32_t subch, cu_start, cu_size, kbps, prot, short_form, set_at, stop_at; };

int main(int argc, char ** argv)
{
  if (argc < 3) { std::fprintf(stderr, "usage: shim_symbols <input> <output prefix>\n"); return 2; }
  if (dabx_device_count() < 1) { std::fprintf(stderr, "no HIP device: the HIP classes have no CPU fallback\n"); return 3; }
  FILE * in = std::fopen(argv[1], "rb");
  if (!in) { std::perror(argv[1]); return 2; }
  int32_t hdr[2];
  if (std::fread(hdr, 4, 2, in) != 2) return 2;
  const int n_frames = hdr[0], n_svc = hdr[1];
  std::vector<Svc> svc((size_t)n_svc);
  if (std::fread(svc.data(), sizeof(Svc), (size_t)n_svc, in) != (size_t)n_svc) return 2;

  RingBuffer<cf32> iqBuf; RingBuffer<f32> carrBuf; RingBuffer<u8> frameBuf; RingBuffer<i16> audioBuf; RingBuffer<u8> dataBuf;
  OfdmDecoder ofdm(nullptr, &iqBuf, &carrBuf);
  FicDecoder fic(nullptr);
  MscHandler msc(nullptr, &frameBuf);
  ofdm.set_soft_bit_gen_type(ESoftBitType::SOFTDEC1);
  fic.restart();                                             // DabProcessor::start, dab_processor.cpp:472

  const std::string prefix = argv[2];
  FILE * fbits = std::fopen((prefix + ".fibbits").c_str(), "wb");
  std::vector<TArrayTu> spec(76);
  TArrayTu null_fft;
  std::vector<i16> bits((size_t)c2K);
  long ratio_sum = 0;
  int cif_count = 0;
  for (int f = 0; f < n_frames; f++)
  {
    for (int i = 0; i < n_svc; i++)
    {
      if (svc[(size_t)i].set_at == f)
      {
        SDescriptorType d;
        d.isDefined = true; d.SubChId = (i16)svc[(size_t)i].subch; d.CuStartAddr = (i16)svc[(size_t)i].cu_start; d.CuSize = (i16)svc[(size_t)i].cu_size;
        d.bitRate = (i16)svc[(size_t)i].kbps; d.protLevel = (i16)svc[(size_t)i].prot; d.shortForm = svc[(size_t)i].short_form != 0;
        if (!msc.set_channel(&d, &audioBuf, &dataBuf, EProcessFlag::Primary)) { std::fprintf(stderr, "set_channel failed\n"); return 4; }
        if (!msc.is_service_running(d.SubChId, EProcessFlag::Primary) || msc.is_service_running(d.SubChId, EProcessFlag::Secondary)) return 5;
      }
      if (svc[(size_t)i].stop_at == f) msc.stop_service(svc[(size_t)i].subch, EProcessFlag::Primary);
    }
    float clock_err = 0;
    for (int s = 0; s < 76; s++) if (std::fread(spec[(size_t)s].data(), sizeof(cf32), cTu, in) != (size_t)cTu) return 2;
    if (std::fread(null_fft.data(), sizeof(cf32), cTu, in) != (size_t)cTu || std::fread(&clock_err, 4, 1, in) != 1) return 2;

    ofdm.store_reference_symbol_0(spec[0]);                                    // dab_processor.cpp:202
    for (int sym = 1; sym < cL; sym++)                                         // :304-367
    {
      ofdm.decode_symbol(spec[(size_t)sym], (u16)sym, 0.0f, clock_err, bits);
      if (sym <= 3) fic.process_block(bits, sym);
      else msc.process_block(bits, sym);
    }
    // :267-286: TII frames are (cif_count & 7) >= 4; cif_count is the IFibDecoder's (here: what the FIC stage itself walked)
    u8 v[3072]; bool b[4];
    fic.get_fib_bits(v, b);
    std::fwrite(v, 1, 3072, fbits);
    for (int i = 0; i < 4; i++) { const u8 q = b[i]; std::fwrite(&q, 1, 1, fbits); }
    // The CIF counter DabProcessor asks its IFibDecoder for (FibDecoder::get_cif_count): in the synthetic ensemble FIG 0/0
    // leads the first FIB of every FIC block, the last one parsed wins (fib_decoder_fig0.cpp:89-101)
    for (int g = 0; g < 4; g++)
    {
      if (!b[g]) continue;
      const u8 * fb = v + g * 768;
      int hi = 0, lo = 0;
      for (int i = 0; i < 5; i++) hi = (hi << 1) | fb[4 * 8 + 3 + i];
      for (int i = 0; i < 8; i++) lo = (lo << 1) | fb[5 * 8 + i];
      cif_count = hi * 250 + lo;
    }
    if ((cif_count & 7) >= 4) ofdm.store_null_symbol_with_tii(null_fft);
    else ofdm.store_null_symbol_without_tii(null_fft);
    ratio_sum += fic.get_fic_decode_ratio_percent();
  }
  std::fclose(fbits);
  std::fclose(in);

  auto & rec = ShimRecorder::get();
  FILE * ff = std::fopen((prefix + ".fibs").c_str(), "wb");
  for (const auto & q : rec.fibs) { std::fwrite(&q.fic, 2, 1, ff); std::fwrite(q.bytes.data(), 1, 32, ff); }
  std::fclose(ff);
  size_t total_frames = 0;
  for (size_t i = 0; i < rec.frames.size(); i++)
  {
    FILE * fs = std::fopen((prefix + ".svc" + std::to_string(i)).c_str(), "wb");
    for (const auto & fr : rec.frames[i])
    {
      std::vector<u8> packed(fr.size() / 8);
      for (size_t k = 0; k < fr.size(); k++) packed[k >> 3] = (u8)((packed[k >> 3] << 1) | (fr[k] & 1));
      std::fwrite(packed.data(), 1, packed.size(), fs);
    }
    std::fclose(fs);
    total_frames += rec.frames[i].size();
  }
  // stop / restart semantics (fic_decoder.cpp:182-185, 264-275)
  fic.stop();
  const size_t before = rec.fibs.size();
  std::fill(bits.begin(), bits.end(), (i16)50);
  fic.process_block(bits, 1);
  const bool stopped_ok = rec.fibs.size() == before && rec.disconnects == 1;
  fic.restart();
  const bool ratio_reset = fic.get_fic_decode_ratio_percent() == 0 && rec.connects == 2;
  msc.stop_all_services();
  // the status signals of the replaced classes: signal_fic_status's BER (every 40th FIC block) and signal_show_lcd_data's record
  i32 lcd_count = 0;
  const OfdmDecoder::SLcdData & lcd = ofdm.dabx_last_lcd_data(&lcd_count);
  std::printf("{\"frames\": %d, \"fibs_delivered\": %zu, \"drivers\": %zu, \"logical_frames\": %zu, \"mean_fic_ratio\": %.1f, "
              "\"stopped_ok\": %s, \"ratio_reset\": %s, \"fic_status_ber\": %.9g, \"lcd_count\": %d, \"lcd_snr\": %.4f, \"lcd_symbol\": %d, "
              "\"lcd_mer\": %.4f, \"lcd_mean_value\": %.6g}\n",
              n_frames, rec.fibs.size(), rec.frames.size(), total_frames, (double)ratio_sum / n_frames, stopped_ok ? "true" : "false",
              ratio_reset ? "true" : "false", (double)fic.dabx_last_fic_ber(), (int)lcd_count, (double)lcd.SNR, (int)lcd.CurOfdmSymbolNo,
              (double)lcd.MER, (double)lcd.TestData1);
  return 0;
}
