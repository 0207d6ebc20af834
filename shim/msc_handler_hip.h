/*
 * msc_handler_hip.h -- class MscHandler of the HIP build: public members of base/backend/msc_handler.h:36-47, bound to
 * libdabx's per-symbol MSC stage (dabx_msc_*).  The CIF buffer, every service's 16-CIF time de-interleaver, depuncturing,
 * the K = 7 Viterbi and the energy de-dispersal (Backend::_process_segment, backend.cpp:129-161, Protection::deconvolve)
 * run on the MI355X; each decoded logical frame is handed to the reference's own BackendDriver::add_to_frame as the
 * 24 * bitRate one-bit-per-byte vector Backend::outV would hold, so the audio / data frame processors behind it are untouched.
 */
#pragma once
#include "dabx_shim_env.h"

class MscHandler
{
public:
  MscHandler(DabRadio * ipRadio, RingBuffer<u8> * ipFrameBuffer) : mpRadioInterface(ipRadio), mpFrameBuffer(ipFrameBuffer)
  {
    dabx_shim_check_abi();
    dabx_shim_check(dabx_msc_create(cMaxServices, &mpMsc), "dabx_msc_create");
  }
  ~MscHandler() { dabx_msc_destroy(mpMsc); }
  MscHandler(const MscHandler &) = delete;
  MscHandler & operator=(const MscHandler &) = delete;

  // msc_handler.cpp:140-168.  Called from the DabProcessor thread; the set_ / stop_ members come from the GUI thread.
  void process_block(const std::vector<i16> & iSoftBits, i32 iBlockNr)
  {
    if (iSoftBits.size() != (size_t)c2K || iBlockNr < 4) dabx_shim_check(DABX_E_ARG, "MscHandler::process_block (size / block number)");
    std::lock_guard<std::mutex> lock(mMutex);
    const int closed = dabx_msc_process_block(mpMsc, iSoftBits.data(), iBlockNr);
    dabx_shim_check(closed, "dabx_msc_process_block");
    if (closed == 0) return;
    for (auto & s : mServices)                                      // b->process(...) for every back end, :163-167
    {
      if (!s.driver) continue;
      const int nb = dabx_msc_get_frame(mpMsc, s.slot, mPacked.data(), (int)mPacked.size());
      dabx_shim_check(nb, "dabx_msc_get_frame");
      if (nb == 0) continue;                                        // de-interleaver still filling, backend.cpp:146-150
      s.outV.resize((size_t)nb * 8);                                // outV: bitRate * 24 bits, one per byte, backend.cpp:40
      for (int i = 0; i < nb * 8; i++) s.outV[(size_t)i] = (u8)((mPacked[(size_t)(i >> 3)] >> (7 - (i & 7))) & 1);
      s.driver->add_to_frame(s.outV);                               // backend.cpp:160
    }
  }
  // msc_handler.cpp:123-135: a new Backend (here: a device-side service slot) + its BackendDriver
  bool set_channel(const SDescriptorType * d, RingBuffer<i16> * ipoAudioBuffer, RingBuffer<u8> * ipoDataBuffer, EProcessFlag iProcessFlag)
  {
    std::lock_guard<std::mutex> lock(mMutex);
    dabx_subch_desc q{};
    q.subch_id = d->SubChId; q.cu_start = d->CuStartAddr; q.cu_size = d->CuSize; q.kbps = d->bitRate;
    q.prot_level = d->protLevel; q.short_form = d->shortForm ? 1 : 0;
    q.dab_plus = 0;              // super-frame sync + RS(120,110) stay in the reference's Mp4Processor behind add_to_frame
    if (q.kbps > 0 && mPacked.size() < (size_t)3 * (size_t)q.kbps) mPacked.resize((size_t)3 * (size_t)q.kbps);   // EEP 4-A reaches 2304 kbit/s
    const int slot = dabx_msc_set_channel(mpMsc, &q);
    if (slot < 0) { std::fprintf(stderr, "dabx shim: MscHandler::set_channel: %s\n", dabx_last_error()); return false; }
    SService s;
    s.slot = slot; s.subChId = d->SubChId; s.flag = iProcessFlag;
    s.driver = std::make_unique<BackendDriver>(mpRadioInterface, d, ipoAudioBuffer, ipoDataBuffer, mpFrameBuffer);
    mServices.push_back(std::move(s));
    return true;
  }
  void reset_channel() { stop_all_services(); }                    // msc_handler.h:42 (declared there, unused and undefined in the reference)
  void stop_service(i32 iSubChId, EProcessFlag iProcessFlag)       // :77-103
  {
    std::lock_guard<std::mutex> lock(mMutex);
    for (size_t i = 0; i < mServices.size(); )
    {
      if (mServices[i].subChId == iSubChId && mServices[i].flag == iProcessFlag)
      {
        dabx_shim_check(dabx_msc_stop_service(mpMsc, mServices[i].slot), "dabx_msc_stop_service");
        mServices.erase(mServices.begin() + (std::ptrdiff_t)i);
      }
      else ++i;
    }
  }
  void stop_all_services()                                         // :105-118
  {
    std::lock_guard<std::mutex> lock(mMutex);
    dabx_shim_check(dabx_msc_stop_all_services(mpMsc), "dabx_msc_stop_all_services");
    mServices.clear();
  }
  bool is_service_running(i32 iSubChId, EProcessFlag iProcessFlag) const   // :110-121
  {
    std::lock_guard<std::mutex> lock(mMutex);
    for (const auto & s : mServices) if (s.subChId == iSubChId && s.flag == iProcessFlag) return true;
    return false;
  }

private:
  static constexpr int cMaxServices = 16;
  struct SService
  {
    int slot = -1;
    i32 subChId = -1;
    EProcessFlag flag = EProcessFlag::Primary;
    std::unique_ptr<BackendDriver> driver;
    std::vector<u8> outV;
  };
  DabRadio * const mpRadioInterface;
  RingBuffer<u8> * const mpFrameBuffer;
  dabx_msc * mpMsc = nullptr;
  mutable std::mutex mMutex;
  std::vector<SService> mServices;
  std::vector<u8> mPacked = std::vector<u8>(3 * 384);              // one logical frame, 3 * bitRate bytes: grown by set_channel to the largest configured rate
};
