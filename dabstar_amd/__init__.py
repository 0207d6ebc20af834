"""dabstar_amd -- MI355X-native OFDM-demodulation + FEC back end for DAB/DAB+ Mode I.

The product is libdabx.so (hand-written HIP kernels for gfx950 behind the C ABI of include/dabx.h).
This package only loads it (ctypes) and offers thin numpy/torch-pointer wrappers; there is no CPU
fallback -- every compute entry point raises when no HIP device is usable."""
from .lib import DabxError, load, lib_path  # noqa: F401
