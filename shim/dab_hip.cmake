# dab_hip.cmake -- CMake fragment for building tomneda/DABstar with the MI355X back end (libdabx).
#
# The reference selects its OfdmDecoder implementation at compile time (CMakeLists.txt:149 option SSE_OR_AVX,
# :167-169 add_compile_definitions(HAVE_SSE_OR_AVX); src/base/CMakeLists.txt:23-44 picks ofdm_decoder_simd.{h,cpp} or
# ofdm_decoder.{h,cpp}; src/base/main/dab_processor.h:53-57 includes one of the two headers -- both define the same
# class name).  The HIP back end enters the same way, as a third alternative, and extends the switch to FicDecoder and
# MscHandler:
#
#   1. top-level CMakeLists.txt, next to option(SSE_OR_AVX ...):
#        option(DAB_HIP "OFDM demapper, FIC and MSC decoding on an AMD MI355X through libdabx" OFF)
#        set(DABX_ROOT "" CACHE PATH "checkout of the libdabx repository (include/, shim/, dabstar_amd/libdabx.so)")
#        if (DAB_HIP)
#          add_compile_definitions(DAB_HIP)        # before add_executable()/add_subdirectory(), like HAVE_SSE_OR_AVX (ODR)
#        endif ()
#   2. src/base/CMakeLists.txt: include(${DABX_ROOT}/shim/dab_hip.cmake) in place of the SSE_OR_AVX if/else block.
#   3. src/base/main/dab_processor.h:53-57 and :39-40 (see INTEGRATION.md section 2 for the three-line patch):
#        #if defined(DAB_HIP)
#          #include "ofdm_decoder_hip.h"
#          #include "fic_decoder_hip.h"
#          #include "msc_handler_hip.h"
#        #elif defined(HAVE_SSE_OR_AVX) ...
#
# ${baseLibName}_HDRS / _SRCS / baseExtraLibs are the variables src/base/CMakeLists.txt already uses.

if (DAB_HIP)
    if (NOT EXISTS ${DABX_ROOT}/include/dabx.h)
        message(FATAL_ERROR "DAB_HIP needs -DDABX_ROOT=<libdabx checkout> (include/dabx.h not found in '${DABX_ROOT}')")
    endif ()
    find_library(DABX_LIBRARY NAMES dabx PATHS ${DABX_ROOT}/dabstar_amd NO_DEFAULT_PATH REQUIRED)
    include_directories(${DABX_ROOT}/include ${DABX_ROOT}/shim)
    # the shim classes are QObjects with the signals of the classes they replace: AUTOMOC must see the headers
    set(${baseLibName}_HDRS
            ${${baseLibName}_HDRS}
            ${DABX_ROOT}/shim/dabx_shim_env.h
            ${DABX_ROOT}/shim/ofdm_decoder_hip.h
            ${DABX_ROOT}/shim/fic_decoder_hip.h
            ${DABX_ROOT}/shim/msc_handler_hip.h
    )
    # the one translation unit that needs the GUI class: connect() of the shim signals to DabRadio's slots
    set(${baseLibName}_SRCS
            ${${baseLibName}_SRCS}
            ${DABX_ROOT}/shim/dab_hip_gui.cpp
    )
    # replaced by the shims: not compiled in this configuration
    list(REMOVE_ITEM ${baseLibName}_SRCS
            ofdm/ofdm_decoder.cpp ofdm/ofdm_decoder_simd.cpp decoder/fic_decoder.cpp backend/msc_handler.cpp
            backend/backend.cpp backend/backend_deconvolver.cpp)
    list(REMOVE_ITEM ${baseLibName}_HDRS
            ofdm/ofdm_decoder.h ofdm/ofdm_decoder_simd.h decoder/fic_decoder.h backend/msc_handler.h backend/backend.h)
    list(APPEND baseExtraLibs ${DABX_LIBRARY})
elseif (SSE_OR_AVX)
    set(${baseLibName}_HDRS ${${baseLibName}_HDRS} support/simd_extensions.h ofdm/ofdm_decoder_simd.h)
    set(${baseLibName}_SRCS ${${baseLibName}_SRCS} ofdm/ofdm_decoder_simd.cpp)
    find_package(Volk REQUIRED)
    list(APPEND baseExtraLibs ${VOLK_LIBRARIES})
else ()
    set(${baseLibName}_HDRS ${${baseLibName}_HDRS} ofdm/ofdm_decoder.h)
    set(${baseLibName}_SRCS ${${baseLibName}_SRCS} ofdm/ofdm_decoder.cpp)
endif ()
