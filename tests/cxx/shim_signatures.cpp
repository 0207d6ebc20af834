// shim_signatures.cpp -- compile-time check that the HIP class shims expose the reference's class surface:
// every public member SURVEY.md 8(b) lists for OfdmDecoder (base/ofdm/ofdm_decoder.h:46-73), FicDecoder
// (base/decoder/fic_decoder.h:42-58) and MscHandler (base/backend/msc_handler.h:36-47), with the reference's exact
// parameter and return types.  Compiles the same shim text the reference tree would (shim/*.h) against the standalone
// vocabulary; running it only prints a line (no GPU needed: nothing is constructed).
// With -DSHIM_IN_TREE the same assertions are made against the reference's OWN headers (glob_defs.h, dab_constants.h,
// glob_enums.h, ringbuffer.h, fib_decoder_if.h, backend_driver.h; syntax / type check only: the object is not linked).
#ifndef SHIM_IN_TREE
#define DABX_SHIM_STANDALONE "shim_env_standalone.h"
#endif
#include "ofdm_decoder_hip.h"
#include "fic_decoder_hip.h"
#include "msc_handler_hip.h"
#include <cstdio>
#include <type_traits>

template <class T, class U> constexpr bool same = std::is_same_v<T, U>;

// ---- OfdmDecoder
static_assert(std::is_constructible_v<OfdmDecoder, DabRadio *, RingBuffer<cf32> *, RingBuffer<f32> *>);
static_assert(same<decltype(&OfdmDecoder::reset), void (OfdmDecoder::*)()>);
static_assert(same<decltype(&OfdmDecoder::store_null_symbol_with_tii), void (OfdmDecoder::*)(const TArrayTu &)>);
static_assert(same<decltype(&OfdmDecoder::store_null_symbol_without_tii), void (OfdmDecoder::*)(const TArrayTu &)>);
static_assert(same<decltype(&OfdmDecoder::store_reference_symbol_0), void (OfdmDecoder::*)(const TArrayTu &)>);
static_assert(same<decltype(&OfdmDecoder::decode_symbol), void (OfdmDecoder::*)(const TArrayTu &, u16, f32, f32, std::vector<i16> &)>);
static_assert(same<decltype(&OfdmDecoder::set_select_carrier_plot_type), void (OfdmDecoder::*)(ECarrierPlotType)>);
static_assert(same<decltype(&OfdmDecoder::set_select_iq_plot_type), void (OfdmDecoder::*)(EIqPlotType)>);
static_assert(same<decltype(&OfdmDecoder::set_soft_bit_gen_type), void (OfdmDecoder::*)(ESoftBitType)>);
static_assert(same<decltype(&OfdmDecoder::set_dc_offset), void (OfdmDecoder::*)(cf32)>);
static_assert(same<decltype(OfdmDecoder::SLcdData::SNR), f32> && same<decltype(OfdmDecoder::SLcdData::CurOfdmSymbolNo), i32>);
// ---- FicDecoder
static_assert(std::is_constructible_v<FicDecoder, DabRadio *> && !std::is_convertible_v<DabRadio *, FicDecoder>);   // explicit
static_assert(same<decltype(&FicDecoder::process_block), void (FicDecoder::*)(const std::vector<i16> &, i32)>);
static_assert(same<decltype(&FicDecoder::stop), void (FicDecoder::*)()>);
static_assert(same<decltype(&FicDecoder::restart), void (FicDecoder::*)()>);
static_assert(same<decltype(&FicDecoder::get_fib_bits), void (FicDecoder::*)(u8 *, bool *)>);
static_assert(same<decltype(&FicDecoder::get_fic_decode_ratio_percent), i32 (FicDecoder::*)() const>);
static_assert(same<decltype(&FicDecoder::reset_fic_decode_success_ratio), void (FicDecoder::*)()>);
static_assert(same<decltype(&FicDecoder::start_fic_dump), void (FicDecoder::*)(FILE *)>);
static_assert(same<decltype(&FicDecoder::stop_fic_dump), void (FicDecoder::*)()>);
static_assert(same<decltype(&FicDecoder::get_fib_decoder), IFibDecoder * (FicDecoder::*)()>);
// ---- MscHandler
static_assert(std::is_constructible_v<MscHandler, DabRadio *, RingBuffer<u8> *>);
static_assert(same<decltype(&MscHandler::process_block), void (MscHandler::*)(const std::vector<i16> &, i32)>);
static_assert(same<decltype(&MscHandler::set_channel), bool (MscHandler::*)(const SDescriptorType *, RingBuffer<i16> *, RingBuffer<u8> *, EProcessFlag)>);
static_assert(same<decltype(&MscHandler::reset_channel), void (MscHandler::*)()>);
static_assert(same<decltype(&MscHandler::stop_service), void (MscHandler::*)(i32, EProcessFlag)>);
static_assert(same<decltype(&MscHandler::stop_all_services), void (MscHandler::*)()>);
static_assert(same<decltype(&MscHandler::is_service_running), bool (MscHandler::*)(i32, EProcessFlag) const>);

int main()
{
  std::printf("shim signatures ok: OfdmDecoder 10, FicDecoder 9, MscHandler 6 members\n");
  return 0;
}
