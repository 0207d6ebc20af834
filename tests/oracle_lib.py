"""ctypes access to the CPU oracle (oracle/_build/liboracle.so) and, when built, to the genuine
reference leaf objects (oracle/_ref/libdabref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORA_DIR = os.path.join(ROOT, "oracle")
ORA_SO = os.path.join(ORA_DIR, "_build", "liboracle.so")
REF_SO = os.path.join(ORA_DIR, "_ref", "libdabref.so")

_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_c64p = np.ctypeslib.ndpointer(np.complex64, flags="C_CONTIGUOUS")


class SubchDesc(C.Structure):
    _fields_ = [("subch_id", C.c_int), ("cu_start", C.c_int), ("cu_size", C.c_int),
                ("kbps", C.c_int), ("prot_level", C.c_int), ("short_form", C.c_int)]


class RxCapture(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("fibs", C.POINTER(C.c_uint8)), ("fib_crc", C.POINTER(C.c_uint8)),
                ("soft", C.POINTER(C.c_int16)), ("start_idx", C.POINTER(C.c_int32)),
                ("fbb", C.POINTER(C.c_float)), ("sym0_pos", C.POINTER(C.c_int32)),
                ("fbb_end", C.POINTER(C.c_float)), ("clock_err", C.POINTER(C.c_float)), ("fic_ratio", C.POINTER(C.c_int32)),
                ("snr_db", C.POINTER(C.c_float)), ("fic_overflow", C.POINTER(C.c_int32)), ("msc_overflow", C.POINTER(C.c_int32)),
                ("s_level", C.POINTER(C.c_float)), ("peak_level", C.POINTER(C.c_float)),
                ("fic_ber_bits", C.POINTER(C.c_int32)), ("fic_ber_errors", C.POINTER(C.c_int32)),
                ("mer_db", C.POINTER(C.c_float))]


def build_oracle():
    if not os.path.exists(ORA_SO) or any(
            os.path.getmtime(os.path.join(ORA_DIR, f)) > os.path.getmtime(ORA_SO)
            for f in os.listdir(ORA_DIR) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", ORA_DIR], stdout=subprocess.DEVNULL)
    return ORA_SO


_ora = None
_ref = None


def oracle():
    global _ora
    if _ora is not None:
        return _ora
    L = C.CDLL(build_oracle())
    L.ora_pi_codes.restype = C.POINTER(C.c_int8)
    L.ora_pi_codes.argtypes = [C.c_int]
    L.ora_freq_interleaver.argtypes = [_i16p]
    L.ora_phase_table.argtypes = [_c64p]
    L.ora_prbs.argtypes = [_u8p, C.c_int]
    L.ora_viterbi.argtypes = [_i16p, C.c_int, _u8p]
    L.ora_viterbi_simd.argtypes = [_i16p, C.c_int, _u8p]
    L.ora_viterbi_sse2.argtypes = [_i16p, C.c_int, _u8p]
    L.ora_set_viterbi_mode.argtypes = [C.c_int]
    L.ora_viterbi_ber.argtypes = [_i16p, _u8p, _u8p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for f in (L.ora_eep_map, L.ora_uep_map):
        f.argtypes = [C.c_int, C.c_int, _i32p]
        f.restype = C.c_int
    L.ora_fic_map.argtypes = [_i32p]
    L.ora_fic_map.restype = C.c_int
    L.ora_deconvolve.argtypes = [_i16p, _i32p, C.c_int, _u8p]
    L.ora_check_crc_bits.argtypes = [_u8p, C.c_int]
    L.ora_calc_crc.argtypes = [_u8p, C.c_int]
    L.ora_calc_crc.restype = C.c_uint16
    L.ora_check_crc_bytes.argtypes = [_u8p, C.c_int]
    L.ora_firecode_check.argtypes = [_u8p]
    L.ora_firecode_check_and_correct.argtypes = [_u8p]
    L.ora_firecode_syndrome_table.restype = C.POINTER(C.c_uint16)
    L.ora_rs_dec.argtypes = [_u8p, _u8p]
    L.ora_rs_enc.argtypes = [_u8p, _u8p]
    L.ora_fft2048.argtypes = [_c64p, _c64p, C.c_int]
    L.ora_rx_create.restype = C.c_void_p
    L.ora_rx_create.argtypes = [C.POINTER(SubchDesc), C.c_int]
    L.ora_rx_destroy.argtypes = [C.c_void_p]
    L.ora_rx_configure.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
    L.ora_rx_run.argtypes = [C.c_void_p, _c64p, C.c_size_t, C.c_int]
    L.ora_rx_enable_soft_capture.argtypes = [C.c_void_p, C.c_int]
    L.ora_rx_set_dc_iq.argtypes = [C.c_void_p, C.c_int]
    L.ora_dciq_buffer.argtypes = [_c64p, C.c_size_t, C.c_int, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")]
    L.ora_dciq_buffer_f64.argtypes = [_c64p, C.c_size_t, C.c_int, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
    L.ora_rx_run_spectra.argtypes = [C.c_void_p, _c64p, _c64p, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), C.c_int]
    L.ora_rx_get_capture.restype = C.POINTER(RxCapture)
    L.ora_rx_take_tii.argtypes = [C.c_void_p, _c64p]
    L.ora_rx_get_capture.argtypes = [C.c_void_p]
    L.ora_rx_backend.restype = C.c_void_p
    L.ora_rx_backend.argtypes = [C.c_void_p, C.c_int]
    for f in (L.ora_backend_msc_bytes, L.ora_backend_sf_bytes, L.ora_backend_sfi_bytes):
        f.restype = C.POINTER(C.c_uint8)
        f.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.ora_backend_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
    L.ora_parse_fibs.argtypes = [_u8p, _u8p, C.c_int, C.POINTER(SubchDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    L.ora_rx_move_subch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long]
    L.ora_level_walk.argtypes = [_c64p, C.c_size_t, C.c_float]
    L.ora_level_walk.restype = C.c_float
    L.ora_fibdec_new.restype = C.c_void_p
    L.ora_fibdec_free.argtypes = [C.c_void_p]
    L.ora_fibdec_process.argtypes = [C.c_void_p, _u8p, _u8p, C.c_int]
    L.ora_fibdec_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.ora_fibdec_subchannels.argtypes = [C.c_void_p, C.c_int, C.POINTER(SubchDesc), C.POINTER(C.c_int), C.c_int]
    L.ora_eti_frame.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(SubchDesc), C.c_int, _u8p, C.POINTER(C.c_void_p), _u8p]
    L.ora_iq_convert.restype = C.c_longlong
    L.ora_iq_convert.argtypes = [C.c_int] * 6 + [_u8p, C.c_longlong, C.c_void_p, C.c_longlong]
    L.ora_iq_convert_q.argtypes = [C.c_int] * 6 + [_u8p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int]
    L.ora_iq_convert_q.restype = C.c_longlong
    L.ora_fic_init.argtypes = [C.c_void_p]
    L.ora_fic_process_block.argtypes = [C.c_void_p, _i16p, C.c_int]
    L.ora_demap_new.restype = C.c_void_p
    L.ora_demap_free.argtypes = [C.c_void_p]
    L.ora_demap_snr_db.argtypes = [C.c_void_p]
    L.ora_demap_snr_db.restype = C.c_float
    L.ora_demap_mer_db.argtypes = [C.c_void_p]
    L.ora_demap_mer_db.restype = C.c_float
    L.ora_interleave_map.argtypes = []
    L.ora_interleave_map.restype = C.POINTER(C.c_int16)
    L.ora_demap_mean_value.argtypes = [C.c_void_p]
    L.ora_demap_mean_value.restype = C.c_float
    L.ora_demap_std_dev_sq.argtypes = [C.c_void_p]
    L.ora_demap_std_dev_sq.restype = C.POINTER(C.c_float)
    L.ora_demap_set_type.argtypes = [C.c_void_p, C.c_int]
    L.ora_demap_reset.argtypes = [C.c_void_p]
    L.ora_demap_store_ref.argtypes = [C.c_void_p, _c64p]
    L.ora_demap_store_null.argtypes = [C.c_void_p, _c64p]
    L.ora_demap_symbol.argtypes = [C.c_void_p, _c64p, C.c_float, _i16p]
    L.ora_phaseref_new.restype = C.c_void_p
    L.ora_phaseref_free.argtypes = [C.c_void_p]
    L.ora_phaseref_set_strongest.argtypes = [C.c_void_p, C.c_int]
    L.ora_phaseref_correlate.argtypes = [C.c_void_p, _c64p, C.c_float]
    L.ora_phaseref_coarse_cfo.argtypes = [C.c_void_p, _c64p]
    _ora = L
    return L


def ref_viterbi_variant(name):
    """The reference's Viterbi built with its SIMD option (name = "sse2" | "avx2"), or None (not built / CPU lacks it)."""
    path = os.path.join(os.path.dirname(REF_SO), "libdabref_vit_%s.so" % name)
    if not os.path.exists(path):
        return None
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    if (name == "avx2" and " avx2" not in flags) or (name == "sse2" and " sse4_1" not in flags):
        return None
    L = C.CDLL(path)
    L.ref_viterbi.argtypes = [_i16p, C.c_int, _u8p]
    L.ref_viterbi_seconds.argtypes = [_i16p, C.c_int, _u8p, C.c_int]
    L.ref_viterbi_seconds.restype = C.c_double
    return L


def have_ref():
    return os.path.exists(REF_SO)


REF_VARIANTS = {"ieee": "libdabref.so",                 # CMakeLists.txt:76 minus -ffast-math
                "fastmath": "libdabref_fastmath.so"}    # the complete flag set of CMakeLists.txt:76
_ref_variant = "ieee"
_ref_cache = {}


def use_ref_variant(name):
    """Selects which build of the reference's leaf objects ref() returns; False when that build is absent."""
    global _ref_variant, _ref
    path = os.path.join(os.path.dirname(REF_SO), REF_VARIANTS[name])
    if not os.path.exists(path):
        return False
    _ref_variant = name
    _ref = _ref_cache.get(name)
    return True


def ref():
    global _ref
    if _ref is not None:
        return _ref
    L = C.CDLL(os.path.join(os.path.dirname(REF_SO), REF_VARIANTS[_ref_variant]))
    L.ref_viterbi.argtypes = [_i16p, C.c_int, _u8p]
    L.ref_viterbi_seconds.argtypes = [_i16p, C.c_int, _u8p, C.c_int]
    L.ref_viterbi_seconds.restype = C.c_double
    L.ref_viterbi_ber.argtypes = [_i16p, _u8p, _u8p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ref_pi_codes.argtypes = [C.c_int, _i8p]
    for f in (L.ref_eep_map, L.ref_uep_map):
        f.argtypes = [C.c_int, C.c_int, _i32p]
    for f in (L.ref_eep_deconvolve, L.ref_uep_deconvolve):
        f.argtypes = [C.c_int, C.c_int, _i16p, C.c_int, _u8p]
    L.ref_backend_deconvolve.argtypes = [C.c_int, C.c_int, C.c_int, _i16p, C.c_int, _u8p]
    L.ref_backend_deconvolve.restype = None
    L.ref_rs_dec.argtypes = [_u8p, _u8p]
    L.ref_rs_enc.argtypes = [_u8p, _u8p]
    L.ref_firecode_check.argtypes = [_u8p]
    L.ref_firecode_check_and_correct.argtypes = [_u8p]
    L.ref_check_crc_bits.argtypes = [_u8p, C.c_int]
    L.ref_calc_crc.argtypes = [_u8p, C.c_int]
    L.ref_check_crc_bytes.argtypes = [_u8p, C.c_int]
    L.ref_freq_interleaver.argtypes = [_i16p]
    L.ref_phase_table.argtypes = [_f32p]
    L.ref_uep_table.argtypes = [_i16p]
    L.ref_uff_describe.argtypes = [C.c_char_p, _i32p, C.c_char_p, C.POINTER(C.c_longlong)]
    L.ref_tii_new.restype = C.c_void_p
    L.ref_tii_free.argtypes = [C.c_void_p]
    L.ref_tii_reset.argtypes = [C.c_void_p]
    L.ref_tii_set.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ref_tii_add.argtypes = [C.c_void_p, _c64p]
    L.ref_tii_process.argtypes = [C.c_void_p, C.c_int, _f32p, C.c_int]
    _ref = L
    _ref_cache[_ref_variant] = L
    return L


# ---- small numpy-facing wrappers around the oracle -------------------------------------------

def ora_viterbi(soft, nbits):
    out = np.zeros(nbits, np.uint8)
    oracle().ora_viterbi(np.ascontiguousarray(soft, np.int16), nbits, out)
    return out


def ora_viterbi_sse2(soft, nbits):
    out = np.zeros(nbits, np.uint8)
    oracle().ora_viterbi_sse2(np.ascontiguousarray(soft, np.int16), nbits, out)
    return out


def ora_viterbi_simd(soft, nbits):
    out = np.zeros(nbits, np.uint8)
    oracle().ora_viterbi_simd(np.ascontiguousarray(soft, np.int16), nbits, out)
    return out


def ora_eep_map(kbps, prot):
    m = np.zeros(96 * kbps + 24, np.int32)
    n = oracle().ora_eep_map(kbps, prot, m)
    return n, m


def ora_uep_map(kbps, prot):
    m = np.zeros(96 * kbps + 24, np.int32)
    n = oracle().ora_uep_map(kbps, prot, m)
    return n, m


def ora_fic_map():
    m = np.zeros(3096, np.int32)
    n = oracle().ora_fic_map(m)
    return n, m


def ora_fft(x, inverse=False):
    out = np.zeros(2048, np.complex64)
    oracle().ora_fft2048(np.ascontiguousarray(x, np.complex64), out, 1 if inverse else 0)
    return out


def backend_bytes(rx, i, which="msc"):
    L = oracle()
    b = L.ora_rx_backend(rx, i)
    n = C.c_size_t(0)
    p = {"msc": L.ora_backend_msc_bytes, "sf": L.ora_backend_sf_bytes, "sfi": L.ora_backend_sfi_bytes}[which](b, C.byref(n))
    return np.ctypeslib.as_array(p, (n.value,)).copy() if n.value else np.zeros(0, np.uint8)


def backend_stats(rx, i):
    out = (C.c_long * 8)()
    oracle().ora_backend_stats(oracle().ora_rx_backend(rx, i), out)
    return dict(zip(["cif_out", "sf_ok", "sf_fail", "rs_corr", "rs_fail", "fc_corr", "au_ok", "au_bad"], list(out)))


def make_descs(subch):
    return (SubchDesc * len(subch))(*[SubchDesc(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form)
                                      for c in subch])


class OraFibDecoder:
    """oracle/fib.c: the stateful restatement (current / next configuration, change-flag swap)."""
    INFO = ("fibs_processed", "fig00_fib", "last_change_fib", "cif_count", "cif_count_hi", "cif_count_lo", "change_flags",
            "occurrence_change", "n_changes", "n_restarts")

    def __init__(self):
        self._h = oracle().ora_fibdec_new()

    def process(self, fibs, crc_ok):
        fibs = np.ascontiguousarray(fibs, np.uint8).reshape(-1, 32)
        crc_ok = np.ascontiguousarray(crc_ok, np.uint8).reshape(-1)
        return oracle().ora_fibdec_process(self._h, fibs, crc_ok, fibs.shape[0])

    def info(self):
        v = (C.c_longlong * 10)()
        oracle().ora_fibdec_info(self._h, v)
        return dict(zip(self.INFO, list(v)))

    def subchannels(self, next=False, max_out=64):
        out = (SubchDesc * max_out)()
        dp = (C.c_int * max_out)()
        n = oracle().ora_fibdec_subchannels(self._h, int(next), out, dp, max_out)
        return [(out[i].subch_id, out[i].cu_start, out[i].cu_size, out[i].kbps, out[i].prot_level, out[i].short_form, dp[i]) for i in range(n)]

    def close(self):
        if self._h:
            oracle().ora_fibdec_free(self._h)
        self._h = None
