/*
 * ofdm_decoder_hip.h -- class OfdmDecoder of the HIP build (-DDAB_HIP): same name, same public members as
 * base/ofdm/ofdm_decoder.h:46-73 (and ofdm_decoder_simd.h, the precedent: dab_processor.h:53-57 picks one of them at
 * compile time), bound to libdabx's demapper stage (include/dabx.h, dabx_demap_*).  The demapper state of
 * ofdm_decoder.h:88-104 (phase reference, per-carrier integrators and IIRs, null-symbol noise power, mMeanValue) lives
 * on the MI355X.
 */
#pragma once
#include "dabx_shim_env.h"
#include <cmath>

#ifdef DABX_SHIM_STANDALONE
class OfdmDecoder
{
#else
class OfdmDecoder : public QObject
{
  Q_OBJECT
#endif
public:
  OfdmDecoder(DabRadio * ipRadio, RingBuffer<cf32> * ipIqBuffer, RingBuffer<f32> * ipCarrBuffer)
    : mpRadioInterface(ipRadio), mpIqBuffer(ipIqBuffer), mpCarrBuffer(ipCarrBuffer)
  {
    dabx_shim_check_abi();
    dabx_shim_check(dabx_demap_create(1, &mpDemap), "dabx_demap_create");
    dabx_shim_connect_gui();      // ofdm_decoder.cpp:60-61 (defined in shim/dab_hip_gui.cpp in the reference tree)
  }
#ifdef DABX_SHIM_STANDALONE
  ~OfdmDecoder() { dabx_demap_destroy(mpDemap); }
#else
  ~OfdmDecoder() override { dabx_demap_destroy(mpDemap); }
#endif
  OfdmDecoder(const OfdmDecoder &) = delete;
  OfdmDecoder & operator=(const OfdmDecoder &) = delete;

  struct SLcdData      // ofdm_decoder.h:52-60 (DabRadio::slot_show_lcd_data takes it)
  {
    i32 CurOfdmSymbolNo;
    f32 MeanSigmaSqFreqCorr;
    f32 SNR;
    f32 MER;
    f32 TestData1;
    f32 TestData2;
  };

  void reset()                                                     // ofdm_decoder.cpp:90-101
  {
    dabx_shim_check(dabx_demap_reset(mpDemap), "dabx_demap_reset");
  }
  // ofdm_decoder.cpp:103-112: with TII the symbol only feeds the NULL_TII_LIN / _LOG carrier plots (a GUI scope, out of
  // this back end's scope); the decoder state is not touched
  void store_null_symbol_with_tii(const TArrayTu &) {}
  void store_null_symbol_without_tii(const TArrayTu & iV)          // :114-130
  {
    dabx_shim_check(dabx_demap_store_null_symbol_without_tii(mpDemap, reinterpret_cast<const dabx_cf32 *>(iV.data())), "dabx_demap_store_null_symbol_without_tii");
  }
  void store_reference_symbol_0(const TArrayTu & iV)               // :132-145
  {
    dabx_shim_check(dabx_demap_store_reference_symbol_0(mpDemap, reinterpret_cast<const dabx_cf32 *>(iV.data())), "dabx_demap_store_reference_symbol_0");
  }
  // :147-355.  iPhaseCorr only enters the LCD statistics (:296-300), never the soft bits.
  void decode_symbol(const TArrayTu & iV, const u16 iCurOfdmSymbIdx, const f32 iPhaseCorr, const f32 iClockErr, std::vector<i16> & oBits)
  {
    if (oBits.size() != (size_t)c2K) oBits.resize(c2K);
    const float ce = iClockErr;
    ++mShowCntStatistics;                                          // :155-157: one statistics record about every 5 frames, on a rotating symbol
    const bool showStatisticData = (mShowCntStatistics > 5 * 76 && iCurOfdmSymbIdx == mNextShownOfdmSymbIdx);
    dabx_shim_check(dabx_demap_decode_symbols(mpDemap, reinterpret_cast<const dabx_cf32 *>(iV.data()), 1, &ce, oBits.data()), "dabx_demap_decode_symbols");
    const f32 freqCorr = iPhaseCorr / 6.28318530717958647692f * 1000.0f;      // iPhaseCorr / F_2_M_PI * cCarrDiff
    if (iCurOfdmSymbIdx == 1) mMeanSigmaSqFreqCorr += 0.2f * (freqCorr * freqCorr - mMeanSigmaSqFreqCorr);   // :296-300 mean_filter(.., 0.2f)
    if (showStatisticData)                                         // :326-352
    {
      float snr = 0.0f, mer = 0.0f, meanValue = 0.0f;
      dabx_shim_check(dabx_demap_get_lcd_data(mpDemap, &snr, &mer, &meanValue), "dabx_demap_get_lcd_data");
      mLcdData.CurOfdmSymbolNo = iCurOfdmSymbIdx + 1;
      mLcdData.SNR = snr;                                          // 10 log10((mMeanPowerOvrAll - noise) / noise), computed on the device
      mLcdData.MeanSigmaSqFreqCorr = std::sqrt(mMeanSigmaSqFreqCorr);
      mLcdData.TestData2 = freqCorr;
      mLcdData.MER = mer;                                          // 10 log10((pi/4)^2 / mean mStdDevSqPhaseVector), :204-208, 331-340: on the device too
      mLcdData.TestData1 = meanValue;                              // mMeanValue, :344
#ifndef DABX_SHIM_STANDALONE
      emit signal_show_lcd_data(mLcdData);
#endif
      mLcdEmitted++;
      mShowCntStatistics = 0;
      mNextShownOfdmSymbIdx = (mNextShownOfdmSymbIdx + 1) % 76;
      if (mNextShownOfdmSymbIdx == 0) mNextShownOfdmSymbIdx = 1;
    }
  }
  const SLcdData & dabx_last_lcd_data(i32 * oCount = nullptr) const { if (oCount) *oCount = mLcdEmitted; return mLcdData; }   // (for tests: what the signal last carried)

  void set_select_carrier_plot_type(ECarrierPlotType iPlotType) { mCarrierPlotType = iPlotType; }   // scopes: not fed by this back end
  void set_select_iq_plot_type(EIqPlotType iPlotType) { mIqPlotType = iPlotType; }
  void set_soft_bit_gen_type(ESoftBitType iSoftBitType)            // glob_enums.h:49-56 -> 1..3
  {
    mSoftBitType = iSoftBitType;
    dabx_shim_check(dabx_demap_set_soft_bit_gen_type(mpDemap, 1 + (int)iSoftBitType), "dabx_demap_set_soft_bit_gen_type");
  }
  inline void set_dc_offset(cf32 iDcOffset) { mDcAdc = iDcOffset; }

private:
  DabRadio * const mpRadioInterface;
  RingBuffer<cf32> * const mpIqBuffer;
  RingBuffer<f32> * const mpCarrBuffer;
  dabx_demap * mpDemap = nullptr;
  std::atomic<ECarrierPlotType> mCarrierPlotType{ ECarrierPlotType::DEFAULT };
  std::atomic<EIqPlotType> mIqPlotType{ EIqPlotType::DEFAULT };
  std::atomic<ESoftBitType> mSoftBitType{ ESoftBitType::DEFAULT };
  cf32 mDcAdc{ 0.0f, 0.0f };
  i32 mShowCntStatistics = 0, mNextShownOfdmSymbIdx = 1, mLcdEmitted = 0;      // ofdm_decoder.h:86-88
  f32 mMeanSigmaSqFreqCorr = 0.0f;                                             // :103
  SLcdData mLcdData{};

  void dabx_shim_connect_gui();

#ifndef DABX_SHIM_STANDALONE
signals:
  void signal_slot_show_iq(i32, f32);
  void signal_show_lcd_data(const SLcdData &);
#endif
};

#ifdef DABX_SHIM_STANDALONE
inline void OfdmDecoder::dabx_shim_connect_gui() {}
#else
Q_DECLARE_METATYPE(OfdmDecoder::SLcdData)
#endif
