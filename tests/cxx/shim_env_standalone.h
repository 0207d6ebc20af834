/*
 * shim_env_standalone.h -- the handful of vocabulary types the HIP class shims (the headers under shim/) take from the reference tree,
 * restated for this repository's tests so that the very same shim text compiles without Qt.  Test infrastructure: names,
 * field names and signatures follow the reference (common/glob_defs.h:25-59, common/dab_constants.h:72-135,
 * base/main/glob_enums.h:18-56, base/support/ringbuffer.h, base/decoder/fib_decoder_if.h:81-125,
 * base/backend/backend_driver.h:37-48); bodies are recorders the tests read back.
 */
#pragma once
#include <array>
#include <complex>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

using i8 = int8_t; using i16 = int16_t; using i32 = int32_t; using i64 = int64_t;
using u8 = uint8_t; using u16 = uint16_t; using u32 = uint32_t; using u64 = uint64_t;
using f32 = float; using f64 = double;
constexpr auto cL = 76; constexpr auto cK = 1536; constexpr auto cTn = 2656; constexpr auto cTF = 196608;
constexpr auto cTs = 2552; constexpr auto cTu = 2048; constexpr auto cTg = 504;
constexpr auto cBitsPerSymb = 2; constexpr auto cFicPerFrame = 4; constexpr auto cFibPerFic = 3;
constexpr auto c2K = cK * cBitsPerSymb; constexpr auto cFicSizeVitIn = c2K * 3 / 4; constexpr auto cFicSizeVitOut = 768;
constexpr auto cFibSizeVitOut = cFicSizeVitOut / cFibPerFic;
using cf32 = std::complex<f32>;
using TArrayTu = std::array<cf32, cTu>;

enum class EIqPlotType { PHASE_CORR_CARR_NORMED, PHASE_CORR_MEAN_NORMED, RAW_MEAN_NORMED, DC_OFFSET_FFT_100, DC_OFFSET_ADC_100, DEFAULT = PHASE_CORR_CARR_NORMED };
enum class ECarrierPlotType { SB_WEIGHT, EVM_PER, EVM_DB, STD_DEV, PHASE_ERROR, PRS_PHASE, PRS_PHASE_UNWRAP, FOUR_QUAD_PHASE, REL_POWER, SNR,
                              NULL_TII_LIN, NULL_TII_LOG, NULL_NO_TII, NULL_OVR_POW, DEFAULT = SB_WEIGHT };
enum class ESoftBitType { SOFTDEC1, SOFTDEC2, SOFTDEC3, DEFAULT = SOFTDEC1 };
enum class EProcessFlag { Primary, Secondary };
enum class ETMId { StreamModeAudio = 0, StreamModeData = 1, PacketModeData = 3 };

struct SDescriptorType       // dab_constants.h:119-135 without the QString labels
{
  bool isDefined = false;
  ETMId TMId = ETMId::StreamModeAudio;
  u32 SId = 0;
  i16 SubChId = 0;
  i16 CuStartAddr = 0;
  i16 CuSize = 0;
  bool shortForm = false;
  i16 protLevel = 0;
  i16 bitRate = 0;
};

class DabRadio;
template <class T> class RingBuffer { public: explicit RingBuffer(u32 = 0) {} };

class IFibDecoder            // fib_decoder_if.h:81-84 (the members the FicDecoder shim calls)
{
public:
  virtual ~IFibDecoder() = default;
  virtual void process_FIB(const std::array<std::byte, cFibSizeVitOut> &, u16) = 0;
  virtual void connect_channel() = 0;
  virtual void disconnect_channel() = 0;
};

// ---- recorders -----------------------------------------------------------------------------------------------------
struct ShimRecorder
{
  struct Fib { std::array<u8, 32> bytes; u16 fic; };
  std::vector<Fib> fibs;                                  // every process_FIB call, packed
  int connects = 0, disconnects = 0;
  std::vector<std::vector<std::vector<u8>>> frames;       // per BackendDriver (creation order): add_to_frame vectors
  std::vector<i16> driver_subch;
  static ShimRecorder & get() { static ShimRecorder r; return r; }
};

class RecordingFibDecoder : public IFibDecoder
{
public:
  void process_FIB(const std::array<std::byte, cFibSizeVitOut> & b, u16 fic) override
  {
    ShimRecorder::Fib f{};
    for (int i = 0; i < cFibSizeVitOut; i++) f.bytes[(size_t)(i >> 3)] = (u8)((f.bytes[(size_t)(i >> 3)] << 1) | (static_cast<u8>(b[(size_t)i]) & 1));
    f.fic = fic;
    ShimRecorder::get().fibs.push_back(f);
  }
  void connect_channel() override { ShimRecorder::get().connects++; }
  void disconnect_channel() override { ShimRecorder::get().disconnects++; }
};
class FibDecoderFactory { public: static std::unique_ptr<IFibDecoder> create(DabRadio *) { return std::make_unique<RecordingFibDecoder>(); } };

class BackendDriver          // backend_driver.h:37-48
{
public:
  BackendDriver(DabRadio *, const SDescriptorType * d, RingBuffer<i16> *, RingBuffer<u8> *, RingBuffer<u8> *)
  {
    auto & r = ShimRecorder::get();
    mIdx = r.frames.size();
    r.frames.emplace_back();
    r.driver_subch.push_back(d->SubChId);
  }
  void add_to_frame(const std::vector<u8> & outData) const { ShimRecorder::get().frames[mIdx].push_back(outData); }
private:
  size_t mIdx;
};
