/*
 * fic_decoder_hip.h -- class FicDecoder of the HIP build: public members of base/decoder/fic_decoder.h:42-58, bound to
 * libdabx's per-symbol FIC stage (dabx_fic_*).  Depuncturing, the four K = 7 Viterbi blocks per frame, energy
 * de-dispersal and the FIB CRCs run on the MI355X; the FIB/FIG database stays the reference's own IFibDecoder, fed
 * through process_FIB exactly as fic_decoder.cpp:234-261 feeds it (bit-per-byte FIBs, in FIB order, CRC-clean ones only).
 */
#pragma once
#include "dabx_shim_env.h"

#ifdef DABX_SHIM_STANDALONE
class FicDecoder
{
#else
class FicDecoder : public QObject
{
  Q_OBJECT
#endif
public:
  explicit FicDecoder(DabRadio * iMr) : mpFibDecoder(FibDecoderFactory::create(iMr))
  {
    dabx_shim_check_abi();
    dabx_shim_check(dabx_fic_create(&mpFic), "dabx_fic_create");
    dabx_shim_check(dabx_fic_stop(mpFic), "dabx_fic_stop");        // mIsRunning{false} until restart(), fic_decoder.h:77
    dabx_shim_connect_gui(iMr);                                    // fic_decoder.cpp:126
  }
#ifdef DABX_SHIM_STANDALONE
  ~FicDecoder() { dabx_fic_destroy(mpFic); }
#else
  ~FicDecoder() override { dabx_fic_destroy(mpFic); }
#endif
  FicDecoder(const FicDecoder &) = delete;
  FicDecoder & operator=(const FicDecoder &) = delete;

  void process_block(const std::vector<i16> & iOfdmSoftBits, const i32 iOfdmSymbIdx)     // fic_decoder.cpp:143-167
  {
    if (iOfdmSoftBits.size() != (size_t)c2K || iOfdmSymbIdx < 1 || iOfdmSymbIdx > 3) dabx_shim_check(DABX_E_ARG, "FicDecoder::process_block (size / symbol index)");
    int first = 0;
    const int n = dabx_fic_process_block(mpFic, iOfdmSoftBits.data(), iOfdmSymbIdx, &first);
    dabx_shim_check(n, "dabx_fic_process_block");
    for (int fic = first; fic < first + n; fic++)                  // _process_fic_input(fic), :178-262
    {
      uint8_t fibs[96], crc[3];
      dabx_shim_check(dabx_fic_get_fibs(mpFic, fic, fibs, crc), "dabx_fic_get_fibs");
      for (int k = 0; k < cFibPerFic; k++)
      {
        if (!crc[k]) continue;
        std::array<std::byte, cFibSizeVitOut> oneFib;              // one bit per byte, like mFibBitsEntireFrame
        for (int i = 0; i < cFibSizeVitOut; i++) oneFib[(size_t)i] = static_cast<std::byte>((fibs[32 * k + (i >> 3)] >> (7 - (i & 7))) & 1);
        FILE * const dump = mpFicDump.load();
        if (dump != nullptr) fwrite(&fibs[32 * k], 1, 32, dump);   // _dump_fib_to_file, :294-308
        mpFibDecoder->process_FIB(oneFib, (u16)fic);
      }
      if (++mFicBlock == 40)                                       // :201-210
      {
        mFicBlock = 0;
        dabx_fic_ber ber;
        dabx_shim_check(dabx_fic_get_ber(mpFic, &ber), "dabx_fic_get_ber");
        mLastBer = ber.status_bits > 0 ? (f32)ber.status_errors / (f32)ber.status_bits : 0.0f;
#ifndef DABX_SHIM_STANDALONE
        emit signal_fic_status(get_fic_decode_ratio_percent(), mLastBer);      // (mFicDecodeSuccessRatio * 10, mFicErrors / mFicBits), :205
#endif
      }
    }
  }
  void stop()                                                      // :264-268
  {
    mpFibDecoder->disconnect_channel();
    dabx_shim_check(dabx_fic_stop(mpFic), "dabx_fic_stop");
  }
  void restart()                                                   // :270-275
  {
    dabx_shim_check(dabx_fic_restart(mpFic), "dabx_fic_restart");
    mpFibDecoder->connect_channel();
  }
  void get_fib_bits(u8 * v, bool * b)                              // :310-321
  {
    uint8_t valid[4];
    dabx_shim_check(dabx_fic_get_fib_bits(mpFic, v, valid), "dabx_fic_get_fib_bits");
    for (int i = 0; i < 4; i++) b[i] = valid[i] != 0;
  }
  i32 get_fic_decode_ratio_percent() const                         // :323-326
  {
    const int r = dabx_fic_get_decode_ratio_percent(mpFic);
    dabx_shim_check(r, "dabx_fic_get_decode_ratio_percent");
    return r;
  }
  void reset_fic_decode_success_ratio() { dabx_shim_check(dabx_fic_reset_decode_success_ratio(mpFic), "dabx_fic_reset_decode_success_ratio"); }
  void start_fic_dump(FILE * f) { FILE * expected = nullptr; mpFicDump.compare_exchange_strong(expected, f); }   // :277-284
  void stop_fic_dump() { mpFicDump = nullptr; }

  IFibDecoder * get_fib_decoder() { return mpFibDecoder.get(); }
  f32 dabx_last_fic_ber() const { return mLastBer; }               // (not a member of the reference's class: lets a test see the signal's value)

private:
  std::unique_ptr<IFibDecoder> mpFibDecoder;
  dabx_fic * mpFic = nullptr;
  i32 mFicBlock = 0;
  f32 mLastBer = 0.0f;                                             // what signal_fic_status last carried
  std::atomic<FILE *> mpFicDump{ nullptr };

  void dabx_shim_connect_gui(DabRadio *);

#ifndef DABX_SHIM_STANDALONE
signals:
  void signal_fic_status(i32, f32);
#endif
};

#ifdef DABX_SHIM_STANDALONE
inline void FicDecoder::dabx_shim_connect_gui(DabRadio *) {}
#endif
