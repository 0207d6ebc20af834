/*
 * dab_hip_gui.cpp -- the only part of the HIP class shims that needs the GUI class: the connect() calls the replaced
 * classes make in their constructors (ofdm_decoder.cpp:60-61, fic_decoder.cpp:126).  Compiled in the reference tree only
 * (-DDAB_HIP, see dab_hip.cmake); this repository's tests build the shims with DABX_SHIM_STANDALONE, where these members
 * are empty inline functions.
 */
#ifndef DABX_SHIM_STANDALONE
#include "ofdm_decoder_hip.h"
#include "fic_decoder_hip.h"
#include "dabradio.h"

void OfdmDecoder::dabx_shim_connect_gui()
{
  qRegisterMetaType<SLcdData>("SLcdData");
  connect(this, &OfdmDecoder::signal_slot_show_iq, mpRadioInterface, &DabRadio::slot_show_iq);
  connect(this, &OfdmDecoder::signal_show_lcd_data, mpRadioInterface, &DabRadio::slot_show_lcd_data);
}

void FicDecoder::dabx_shim_connect_gui(DabRadio * iMr)
{
  connect(this, &FicDecoder::signal_fic_status, iMr, &DabRadio::slot_show_fic_status);
}
#endif
