/*
 * dabx_shim_env.h -- vocabulary of the HIP class shims (ofdm_decoder_hip.h, fic_decoder_hip.h, msc_handler_hip.h).
 *
 * In the reference tree (tomneda/DABstar, paths relative to its src/) the shims are compiled with -DDAB_HIP and see the
 * reference's own headers: common/glob_defs.h (i16, f32, cf32, TArrayTu, cK ...), common/dab_constants.h
 * (SDescriptorType, EProcessFlag), base/main/glob_enums.h (ESoftBitType ...), base/support/ringbuffer.h,
 * base/decoder/fib_decoder_if.h (IFibDecoder, FibDecoderFactory), base/backend/backend_driver.h (BackendDriver).
 * Outside it (this repository's tests) DABX_SHIM_STANDALONE names a header that supplies those few types, so the very
 * same shim text is compiled, signature-checked and run on the GPU without Qt.
 */
#pragma once
#ifdef DABX_SHIM_STANDALONE
  #include DABX_SHIM_STANDALONE
#else
  #include "glob_defs.h"
  #include "dab_constants.h"
  #include "glob_enums.h"
  #include "ringbuffer.h"
  #include "fib_decoder_if.h"
  #include "backend_driver.h"
  #include <QObject>
#endif
#include <array>
#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>
#include "dabx.h"

/* The reference's class surface has no error channel (void / bool, SURVEY 8b "Errors"): a failing GPU call is fatal and
 * says why -- there is no CPU fallback behind these classes. */
inline void dabx_shim_check(int rc, const char * what)
{
  if (rc >= 0) return;
  std::fprintf(stderr, "dabx shim: %s failed (%d): %s\n", what, rc, dabx_last_error());
  std::abort();
}

/* The shims are compiled against include/dabx.h; the library is found at run time.  Records such as dabx_stats and dabx_config
 * grow with the ABI version: a library of another version than the header is refused before the first object is created. */
inline void dabx_shim_check_abi()
{
  static const bool ok = [] {
    if (dabx_abi_version() == DABX_ABI_VERSION) return true;
    std::fprintf(stderr, "dabx shim: built against libdabx ABI %d, the library loaded is ABI %d\n", DABX_ABI_VERSION, dabx_abi_version());
    std::abort();
    return false;
  }();
  (void)ok;
}
