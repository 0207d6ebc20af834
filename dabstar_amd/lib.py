"""ctypes binding of libdabx.so (C ABI: include/dabx.h)."""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DabxError(RuntimeError):
    pass


def lib_path():
    """The product library.  DABX_LIB (read by this test/bench binding only -- the library itself reads no environment variable) names
    an experiment build instead (tools/build_variant.sh -> tools/_build/ab/*.so, same-box A/B runs of tools/ab.sh): nothing ever
    overwrites libdabx.so."""
    return os.environ.get("DABX_LIB") or os.path.join(HERE, "libdabx.so")


def load(build_if_missing=True):
    """Loads libdabx.so, building it in-tree with hipcc when absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if build_if_missing and not os.environ.get("DABX_LIB"):
        from . import build as _b
        if _b.needs_build():
            _b.build()
    if not os.path.exists(lib_path()):
        raise DabxError("libdabx.so is missing: run `python -m dabstar_amd.build` (needs hipcc)")
    L = C.CDLL(lib_path())
    L.dabx_last_error.restype = C.c_char_p
    for name in ("dabx_tii_destroy", "dabx_tii_reset"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = None
    L.dabx_tii_set_collisions.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.dabx_tii_add.argtypes = [C.c_void_p, C.c_void_p]
    L.dabx_tii_process.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    for name in ("dabx_convert_iq_bytes", "dabx_feed_bytes", "dabx_feed_bound"):
        getattr(L, name).restype = C.c_longlong
    L.dabx_feed_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.dabx_feed_bound.argtypes = [C.c_void_p, C.c_size_t]
    L.dabx_feed_close.argtypes = [C.c_void_p]
    L.dabx_convert_iq_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    # size_t arguments must not travel as C int (buffers of 2 GiB and more)
    L.dabx_push_iq.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t]
    L.dabx_push_iq_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t]
    L.dabx_push_wait.argtypes = [C.c_void_p]
    L.dabx_host_register.argtypes = [C.c_void_p, C.c_size_t]
    L.dabx_host_unregister.argtypes = [C.c_void_p]
    L.dabx_commit_iq.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    if hasattr(L, "dabx_delivery_next"):
        L.dabx_delivery_open.argtypes = [C.c_void_p, C.c_void_p]
        L.dabx_delivery_close.argtypes = [C.c_void_p]
        L.dabx_delivery_next.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.dabx_delivery_release.argtypes = [C.c_void_p, C.c_uint64]
        L.dabx_delivery_slab_bytes.argtypes = [C.c_void_p]
        L.dabx_delivery_wait_free.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dabx_delivery_slab_bytes.restype = C.c_longlong
    if hasattr(L, "dabx_announce_write"):            # (absent from libraries older than the level anchor: tools/ab.sh runs those through this binding too)
        L.dabx_announce_write.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    _LIB = L
    return L


def declared_symbols():
    """Every function name declared in include/dabx.h."""
    hdr = open(os.path.join(HERE, "..", "include", "dabx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dabx_[a-z0-9_]+)\s*\(", hdr)))


def check(rc):
    if rc < 0:
        raise DabxError("libdabx error %d: %s" % (rc, load().dabx_last_error().decode()))
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------- stage level
def viterbi(soft, nbits, tie_mode=0):
    """soft: [batch, 4*(nbits+6)] int16 -> [batch, nbits] uint8 (ViterbiSpiral::deconvolve); tie_mode 1 = the arithmetic of
    the reference's AVX2 / SSE2 builds."""
    soft = np.ascontiguousarray(soft, np.int16).reshape(-1, 4 * (nbits + 6))
    out = np.zeros((soft.shape[0], nbits), np.uint8)
    check(load().dabx_viterbi_mode(_p(soft), nbits, soft.shape[0], int(tie_mode), _p(out)))
    return out


def viterbi_lane_per_trellis(soft, nbits, tie_mode=0, always_clamp=False):
    """The same decode on the lane-per-trellis kernel of the MSC path (vit_t.hip) through the library's internal test entry
    (not part of include/dabx.h): unpunctured trellises, 64 per wavefront.  always_clamp forces the saturating step bodies of
    the tie modes in every cycle (must not change a bit)."""
    soft = np.ascontiguousarray(soft, np.int16).reshape(-1, 4 * (nbits + 6))
    out = np.zeros((soft.shape[0], nbits), np.uint8)
    check(load().dabx_internal_vitT(_p(soft), nbits, soft.shape[0], int(tie_mode), int(bool(always_clamp)), _p(out)))
    return out


def profile_input_bits(kbps, prot_level, short_form=0):
    return check(load().dabx_profile_input_bits(kbps, prot_level, short_form))


def profile_map(kbps, prot_level, short_form=0):
    """Host only: (transmitted bits, index list as int32 with -1 for punctured positions)."""
    m = np.zeros(96 * kbps + 24, np.uint16)
    n = check(load().dabx_profile_map(kbps, prot_level, short_form, _p(m), m.size))
    out = m.astype(np.int32)
    out[m == 0xFFFF] = -1
    return n, out


def deconvolve(soft, kbps, prot_level, short_form=0):
    """soft: [batch, cu_size*64] int16 -> [batch, 24*kbps] uint8 (Protection::deconvolve)."""
    n_in = check(load().dabx_profile_input_bits(kbps, prot_level, short_form))
    soft = np.ascontiguousarray(soft, np.int16).reshape(-1, soft.shape[-1])
    assert soft.shape[1] >= n_in
    out = np.zeros((soft.shape[0], 24 * kbps), np.uint8)
    check(load().dabx_deconvolve(_p(soft), soft.shape[1], kbps, prot_level, short_form, soft.shape[0], _p(out)))
    return out


def rs_decode(cw):
    """cw: [batch,120] uint8 -> (out [batch,110], ret [batch] int16) (ReedSolomon::dec)."""
    cw = np.ascontiguousarray(cw, np.uint8).reshape(-1, 120)
    out = np.zeros((cw.shape[0], 110), np.uint8)
    ret = np.zeros(cw.shape[0], np.int16)
    check(load().dabx_rs_decode(_p(cw), cw.shape[0], _p(out), _p(ret)))
    return out, ret


def firecode_check(x):
    x = np.ascontiguousarray(x, np.uint8).reshape(-1, 12)
    ok = np.zeros(x.shape[0], np.uint8)
    check(load().dabx_firecode_check(_p(x), x.shape[0], _p(ok)))
    return ok


def firecode_check_and_correct(x):
    x = np.ascontiguousarray(x, np.uint8).reshape(-1, 12).copy()
    ok = np.zeros(x.shape[0], np.uint8)
    check(load().dabx_firecode_check_and_correct(_p(x), x.shape[0], _p(ok)))
    return x, ok


def crc16_check(msgs, length):
    msgs = np.ascontiguousarray(msgs, np.uint8)
    ok = np.zeros(msgs.shape[0], np.uint8)
    check(load().dabx_crc16_check(_p(msgs), msgs.shape[1], length, msgs.shape[0], _p(ok)))
    return ok


def fft2048(x, inverse=False):
    x = np.ascontiguousarray(x, np.complex64).reshape(-1, 2048)
    out = np.zeros_like(x)
    check(load().dabx_fft2048(_p(x), x.shape[0], 1 if inverse else 0, _p(out)))
    return out


def prs_correlate(v, threshold, strongest=False):
    v = np.ascontiguousarray(v, np.complex64).reshape(-1, 2048)
    out = np.zeros(v.shape[0], np.int32)
    check(load().dabx_prs_correlate(_p(v), v.shape[0], C.c_float(threshold), int(strongest), _p(out)))
    return out


def coarse_cfo(fft0):
    fft0 = np.ascontiguousarray(fft0, np.complex64).reshape(-1, 2048)
    out = np.zeros(fft0.shape[0], np.int32)
    check(load().dabx_coarse_cfo(_p(fft0), fft0.shape[0], _p(out)))
    return out


class Demap:
    """Mirror of the reference's OfdmDecoder class surface (base/ofdm/ofdm_decoder.h:46-73), batched."""

    def __init__(self, batch=1):
        self.batch = batch
        self._h = C.c_void_p()
        check(load().dabx_demap_create(batch, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.dabx_demap_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(load().dabx_demap_reset(self._h))

    def set_soft_bit_gen_type(self, t):
        check(load().dabx_demap_set_soft_bit_gen_type(self._h, t))

    def store_reference_symbol_0(self, fft):
        fft = np.ascontiguousarray(fft, np.complex64).reshape(self.batch, 2048)
        check(load().dabx_demap_store_reference_symbol_0(self._h, _p(fft)))

    def store_null_symbol_without_tii(self, fft):
        fft = np.ascontiguousarray(fft, np.complex64).reshape(self.batch, 2048)
        check(load().dabx_demap_store_null_symbol_without_tii(self._h, _p(fft)))

    def snr_db(self):
        out = np.zeros(self.batch, np.float32)
        check(load().dabx_demap_get_snr_db(self._h, _p(out)))
        return out

    def lcd_data(self):
        """(snr_db, mer_db, mean_value) of the LCD record, batch floats each (dabx_demap_get_lcd_data)."""
        out = [np.zeros(self.batch, np.float32) for _ in range(3)]
        load().dabx_demap_get_lcd_data.argtypes = [C.c_void_p] * 4
        check(load().dabx_demap_get_lcd_data(self._h, *[_p(o) for o in out]))
        return tuple(out)

    def decode_symbols(self, fft, clock_err):
        fft = np.ascontiguousarray(fft, np.complex64).reshape(self.batch, -1, 2048)
        ce = np.ascontiguousarray(np.broadcast_to(np.asarray(clock_err, np.float32), (self.batch,)))
        out = np.zeros((self.batch, fft.shape[1], 3072), np.int16)
        check(load().dabx_demap_decode_symbols(self._h, _p(fft), fft.shape[1], _p(ce), _p(out)))
        return out


# ---------------------------------------------------------------------------------- engine level
class Config(C.Structure):
    _fields_ = [("n_streams", C.c_int32), ("ring_frames", C.c_int32), ("max_subch", C.c_int32), ("out_frames", C.c_int32),
                ("sync_threshold", C.c_float), ("sync_strongest", C.c_int32), ("soft_bit_type", C.c_int32),
                ("fic_only", C.c_int32), ("capture_soft", C.c_int32), ("viterbi_tie_mode", C.c_int32), ("dc_iq_correction", C.c_int32),
                ("schedule", C.c_int32), ("msc_fast_min_jobs", C.c_int32), ("msc_class_min_jobs", C.c_int32),
                ("exact_level_tracker", C.c_int32), ("acquire_mode", C.c_int32)]


class SubchDesc(C.Structure):
    _fields_ = [("subch_id", C.c_int32), ("cu_start", C.c_int32), ("cu_size", C.c_int32), ("kbps", C.c_int32),
                ("prot_level", C.c_int32), ("short_form", C.c_int32), ("dab_plus", C.c_int32), ("reserved", C.c_int32)]


class SubchStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("start_cif", "cifs_decoded", "sf_count", "sf_ok", "sf_fail", "rs_corrected", "rs_failed",
                                         "fc_corrected", "au_ok", "au_bad")] + [("active", C.c_int32), ("subch_id", C.c_int32)]


class TiiResult(C.Structure):
    _fields_ = [("main_id", C.c_uint8), ("sub_id", C.c_uint8), ("strength", C.c_float), ("phase_deg", C.c_float),
                ("non_etsi_phase", C.c_int32)]

    def as_tuple(self):
        return (self.main_id, self.sub_id, self.strength, self.phase_deg, self.non_etsi_phase)


class Tii:
    """Host-side TII detector (dabx_tii_*): TiiDetector of the reference."""

    def __init__(self):
        self._h = C.c_void_p()
        check(load().dabx_tii_create(C.byref(self._h)))

    def reset(self):
        load().dabx_tii_reset(self._h)

    def set_collisions(self, on, sub_id=0):
        load().dabx_tii_set_collisions(self._h, int(on), int(sub_id))

    def add(self, null_fft):
        v = np.ascontiguousarray(null_fft, np.complex64)
        assert v.size == 2048
        check(load().dabx_tii_add(self._h, _p(v)))

    def process(self, threshold_db, max_out=64):
        out = (TiiResult * max_out)()
        n = check(load().dabx_tii_process(self._h, int(threshold_db), out, max_out))
        return [out[i].as_tuple() for i in range(n)]

    def close(self):
        if self._h and _LIB is not None:
            _LIB.dabx_tii_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IqFormat(C.Structure):
    """dabx_iq_format: family 0 raw / 1 wav / 2 uff; container 0 u8, 1 s8, 2 i16, 3 i24, 4 i32, 5 f32."""
    _fields_ = [("family", C.c_int32), ("container", C.c_int32), ("big_endian", C.c_int32), ("swap_iq", C.c_int32),
                ("bits", C.c_int32), ("sample_rate", C.c_int32), ("data_offset", C.c_int64), ("data_bytes", C.c_int64),
                ("reference_quirks", C.c_int32), ("reserved", C.c_int32)]

    def sample_bytes(self):
        return 2 * (1, 1, 2, 3, 4, 4)[self.container]

    def as_tuple(self):
        """The probed description (family .. data_bytes); reference_quirks is a caller's switch, not part of it."""
        return tuple(getattr(self, k) for k, _ in self._fields_[:8])


class Stats(C.Structure):
    _fields_ = [("frames", C.c_int64), ("samples_consumed", C.c_int64), ("state", C.c_int32), ("fic_ratio_percent", C.c_int32),
                ("freq_offs_bb_hz", C.c_float), ("clock_err_hz", C.c_float), ("snr_db_est", C.c_float),
                ("last_start_index", C.c_int32), ("cif_count", C.c_int32),
                ("fib_ok", C.c_int64), ("fib_total", C.c_int64), ("sf_ok", C.c_int64), ("sf_fail", C.c_int64),
                ("rs_corrected", C.c_int64), ("rs_failed", C.c_int64), ("au_ok", C.c_int64), ("au_bad", C.c_int64),
                ("cifs_decoded", C.c_int64), ("signal_level", C.c_float), ("peak_level", C.c_float),
                ("level_margin_events", C.c_int64), ("level_rewalk_events", C.c_int64), ("level_unanchored_events", C.c_int64),
                ("level_healed_events", C.c_int64), ("fic_ber_bits", C.c_int64), ("fic_ber_errors", C.c_int64),
                ("mer_db_est", C.c_float), ("reserved_f", C.c_float), ("reserved", C.c_int64 * 1)]


# ---- bulk delivery (include/dabx.h "Bulk delivery"): the slab's records as numpy dtypes -----------------------------
CHUNK_FRAMES = 7
CHUNK_MAGIC = 0x43584244
DELIVER_FIB, DELIVER_MSC, DELIVER_SF, DELIVER_MSC_NOT_DABPLUS = 1, 2, 4, 8
CHUNK_HEADER = np.dtype([("magic", "<u4"), ("abi", "<u4"), ("seq", "<u8"), ("n_streams", "<i4"), ("max_subch", "<i4"),
                         ("max_frames", "<i4"), ("what", "<i4"), ("bytes", "<u8"), ("off_stream", "<u8"), ("off_subch", "<u8"),
                         ("off_fib", "<u8"), ("off_crc", "<u8"), ("off_frame", "<u8"), ("off_msc", "<u8"), ("off_sf", "<u8"),
                         ("reserved", "<u8", 4)])
CHUNK_STREAM = np.dtype([("first_frame", "<i8"), ("n_frames", "<i4"), ("frames_lost", "<i4"), ("state", "<i4"),
                         ("fic_ratio_percent", "<i4"), ("cif_count", "<i4"), ("snr_db_est", "<f4"), ("freq_offs_bb_hz", "<f4"),
                         ("clock_err_hz", "<f4"), ("signal_level", "<f4"), ("fic_ber_bits", "<i4"), ("fic_ber_errors", "<i4"),
                         ("mer_db_est", "<f4"), ("fib_ok", "<i8"), ("fib_total", "<i8")])
CHUNK_FRAME = np.dtype([("sym0_pos", "<i8"), ("start_index", "<i4"), ("reserved", "<i4")])
CHUNK_SUBCH = np.dtype([("active", "<i4"), ("subch_id", "<i4"), ("kbps", "<i4"), ("dab_plus", "<i4"), ("start_cif", "<i8"),
                        ("first_cif", "<i8"), ("n_cifs", "<i4"), ("cifs_lost", "<i4"), ("first_sf", "<i8"), ("n_sf", "<i4"),
                        ("sf_lost", "<i4"), ("msc_off", "<u8"), ("sf_off", "<u8"), ("sf_pitch", "<i4"), ("reserved", "<i4"),
                        ("sf_ok", "<i8"), ("sf_fail", "<i8"), ("rs_corrected", "<i8"), ("rs_failed", "<i8"),
                        ("fc_corrected", "<i8"), ("au_ok", "<i8"), ("au_bad", "<i8"), ("sfi_off", "<u8")])
# dabx_superframe_info: what Mp4Processor::_process_super_frame knows when it hands the access units on (mp4processor.cpp:249-333)
SUPERFRAME_INFO = np.dtype([("num_aus", "u1"), ("au_crc_ok", "u1"), ("au_len_bad", "u1"), ("stream_parms", "u1"), ("au_start", "<u2", 7),
                            ("rs_corrected", "<u2"), ("rs_failed", "u1"), ("fc_corrected", "u1"), ("reserved", "<u2"), ("first_frame", "<i8")])
assert CHUNK_HEADER.itemsize == 128 and CHUNK_STREAM.itemsize == 72 and CHUNK_FRAME.itemsize == 16 and CHUNK_SUBCH.itemsize == 144
assert SUPERFRAME_INFO.itemsize == 32


class DeliveryConfig(C.Structure):
    _fields_ = [("host_slabs", C.c_int32), ("what", C.c_int32), ("copy_engine", C.c_int32), ("reserved", C.c_int32 * 5)]


class IngestConfig(C.Structure):
    _fields_ = [("host_slabs", C.c_int32), ("fmt", C.c_int32), ("max_frames", C.c_int32), ("copy_engine", C.c_int32), ("reserved", C.c_int32 * 4)]


class DeliveryInfo(C.Structure):
    _fields_ = [("chunks_closed", C.c_uint64), ("chunks_landed", C.c_uint64), ("bytes_copied", C.c_uint64), ("copy_seconds", C.c_double),
                ("copy_seconds_max", C.c_double), ("gather_wait_seconds", C.c_double), ("copy_engine", C.c_int32),
                ("sdma_engine_mask", C.c_uint32), ("calibration_GBps", C.c_double), ("reserved", C.c_uint64 * 3)]


class ChunkRef(C.Structure):
    _fields_ = [("seq", C.c_uint64), ("data", C.c_void_p), ("bytes", C.c_uint64)]


class Chunk:
    """One delivered slab, viewed in place (no copy): valid until release()."""

    def __init__(self, engine, ref):
        self._eng, self.seq, self.nbytes = engine, ref.seq, ref.bytes
        self.raw = np.ctypeslib.as_array(C.cast(ref.data, C.POINTER(C.c_uint8)), shape=(ref.bytes,))
        self.header = self.raw[:128].view(CHUNK_HEADER)[0]
        h = self.header
        assert h["magic"] == CHUNK_MAGIC and h["bytes"] == ref.bytes, (hex(int(h["magic"])), int(h["bytes"]), ref.bytes)
        S, M, F = int(h["n_streams"]), int(h["max_subch"]), int(h["max_frames"])
        self.S, self.M, self.F = S, M, F
        self.streams = self.raw[int(h["off_stream"]):int(h["off_stream"]) + S * 72].view(CHUNK_STREAM)
        self.subch = self.raw[int(h["off_subch"]):int(h["off_subch"]) + S * M * 144].view(CHUNK_SUBCH).reshape(S, M)
        if h["what"] & DELIVER_FIB:
            self.fibs = self.raw[int(h["off_fib"]):int(h["off_fib"]) + S * F * 384].reshape(S, F, 12, 32)
            self.crc = self.raw[int(h["off_crc"]):int(h["off_crc"]) + S * F * 12].reshape(S, F, 12)
            self.frames = self.raw[int(h["off_frame"]):int(h["off_frame"]) + S * F * 16].view(CHUNK_FRAME).reshape(S, F)

    def msc(self, s, j):
        """Logical frames of slot (s, j) in this chunk: [n_cifs, 3 * kbps] uint8 (a view)."""
        r = self.subch[s, j]
        nb = 3 * int(r["kbps"])
        o = int(r["msc_off"])
        return self.raw[o:o + int(r["n_cifs"]) * nb].reshape(int(r["n_cifs"]), nb)

    def superframes(self, s, j):
        """Super frames of slot (s, j) in this chunk: [n_sf, 110 * kbps / 8] uint8 (a view)."""
        r = self.subch[s, j]
        nb, pitch, o = 110 * int(r["kbps"]) // 8, int(r["sf_pitch"]), int(r["sf_off"])
        return self.raw[o:o + int(r["n_sf"]) * pitch].reshape(int(r["n_sf"]), pitch)[:, :nb]

    def superframe_info(self, s, j):
        """Their records (AU table, per-AU CRC verdicts, corrections): [n_sf] SUPERFRAME_INFO (a view)."""
        r = self.subch[s, j]
        o = int(r["sfi_off"])
        return self.raw[o:o + int(r["n_sf"]) * 32].view(SUPERFRAME_INFO)

    def release(self):
        if self._eng is not None:
            check(load().dabx_delivery_release(self._eng._h, C.c_uint64(self.seq)))
            self._eng = None


COUNTER_NAMES = ["frames", "samples", "fib_ok", "fib_total", "sync_lost", "streams_locked", "cifs_decoded", "sf_ok", "sf_fail",
                 "rs_corrected", "rs_failed", "fc_corrected", "au_ok", "au_bad", "msc_bytes", "reserved"]
TF = 196608


class Engine:
    """Stream-batched receiver (device-side DabProcessor::run for n_streams ensembles)."""

    def __init__(self, n_streams=1, ring_frames=4, max_subch=18, out_frames=4, fic_only=False, capture_soft=False,
                 sync_threshold=3.0, soft_bit_type=1, sync_strongest=False, viterbi_tie_mode=0, dc_iq_correction=0,
                 schedule=0, msc_fast_min_jobs=0, msc_class_min_jobs=0, exact_level_tracker=False, acquire_mode=0):
        L = load()
        cfg = Config()
        L.dabx_default_config(C.byref(cfg))
        cfg.n_streams, cfg.ring_frames, cfg.max_subch, cfg.out_frames = n_streams, ring_frames, max_subch, out_frames
        cfg.fic_only, cfg.capture_soft, cfg.sync_threshold = int(fic_only), int(capture_soft), sync_threshold
        cfg.soft_bit_type, cfg.sync_strongest = soft_bit_type, int(sync_strongest)
        cfg.viterbi_tie_mode = int(viterbi_tie_mode)
        cfg.dc_iq_correction = int(dc_iq_correction)
        cfg.schedule, cfg.msc_fast_min_jobs, cfg.msc_class_min_jobs = int(schedule), int(msc_fast_min_jobs), int(msc_class_min_jobs)
        cfg.exact_level_tracker = int(exact_level_tracker)
        cfg.acquire_mode = int(acquire_mode)
        self.cfg = cfg
        self.n_streams = n_streams
        self._h = C.c_void_p()
        check(L.dabx_create(C.byref(cfg), C.byref(self._h)))
        self.subch = []

    def close(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.dabx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_subchannels(self, subch, stream=-1, dab_plus=True):
        arr = (SubchDesc * max(1, len(subch)))()
        for i, c in enumerate(subch):
            dp = getattr(c, "dab_plus", -1)      # discovered descriptors carry FIG 0/2's answer; -1 = not yet known
            arr[i] = SubchDesc(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form, int(dab_plus) if dp < 0 else dp, 0)
        check(load().dabx_set_subchannels(self._h, stream, arr, len(subch)))
        self.subch = list(subch)

    def push_iq(self, stream, iq):
        iq = np.ascontiguousarray(iq)
        fmt = {np.dtype(np.complex64): 0, np.dtype(np.int16): 1, np.dtype(np.uint8): 2}[iq.dtype]
        n = iq.size if fmt == 0 else iq.size // 2
        check(load().dabx_push_iq(self._h, stream, _p(iq), fmt, n))

    def push_iq_async(self, stream, iq):
        """iq must stay alive and unchanged until push_wait() (see dabx_push_iq_async); it is pushed in place, so it has to be
        C-contiguous (a strided view cannot be copied here: the copy would not outlive the call)."""
        if not iq.flags.c_contiguous:
            raise ValueError("push_iq_async needs a C-contiguous array")
        fmt = {np.dtype(np.complex64): 0, np.dtype(np.int16): 1, np.dtype(np.uint8): 2}[iq.dtype]
        n = iq.size if fmt == 0 else iq.size // 2
        check(load().dabx_push_iq_async(self._h, stream, _p(iq), fmt, n))

    def push_wait(self):
        check(load().dabx_push_wait(self._h))

    def read_iq(self, stream, first, n):
        out = np.zeros(n, np.complex64)
        check(load().dabx_read_iq(self._h, stream, C.c_uint64(first), C.c_size_t(n), _p(out)))
        return out

    def ring_ptr(self, stream):
        p, cap = C.c_void_p(), C.c_size_t()
        check(load().dabx_iq_ring_dev(self._h, stream, C.byref(p), C.byref(cap)))
        return p.value, cap.value

    def commit(self, n_samples, stream=-1):
        check(load().dabx_commit_iq(self._h, stream, n_samples))

    def announce_write(self, n_samples, stream=-1):
        if hasattr(load(), "dabx_announce_write"):
            check(load().dabx_announce_write(self._h, stream, n_samples))

    def process(self, max_frames, sync=True):
        return check(load().dabx_process(self._h, max_frames, int(sync)))

    def synchronize(self):
        check(load().dabx_synchronize(self._h))

    def hip_stream(self):
        load().dabx_hip_stream.restype = C.c_void_p
        return load().dabx_hip_stream(self._h)

    def read_fibs(self, stream, n_frames=1):
        fibs = np.zeros((n_frames, 12, 32), np.uint8)
        crc = np.zeros((n_frames, 12), np.uint8)
        n = check(load().dabx_read_fibs(self._h, stream, n_frames, _p(fibs), _p(crc)))
        return fibs[:n], crc[:n]

    def read_frame_info(self, stream, n_frames=1):
        """(sym0_pos int64[n], start_index int32[n]) of the newest frames, oldest first (dabx_read_frame_info)."""
        pos = np.zeros(n_frames, np.int64)
        st = np.zeros(n_frames, np.int32)
        n = check(load().dabx_read_frame_info(self._h, stream, n_frames, _p(pos), _p(st)))
        return pos[:n], st[:n]

    def discover_subchannels(self, stream, max_out=64):
        out = (SubchDesc * max_out)()
        n = check(load().dabx_discover_subchannels(self._h, stream, out, max_out))
        return [out[i] for i in range(n)]

    def set_fig_reference_quirks(self, on):
        check(load().dabx_set_fig_reference_quirks(self._h, int(on)))

    def set_lcd_statistics(self, on):
        check(load().dabx_set_lcd_statistics(self._h, int(on)))

    def follow_fic(self, stream):
        out = Reconf()
        check(load().dabx_follow_fic(self._h, stream, C.byref(out)))
        return {k: getattr(out, k) for k, _ in Reconf._fields_ if not k.startswith("reserved")}

    def next_subchannels(self, stream, max_out=64):
        out = (SubchDesc * max_out)()
        n = check(load().dabx_next_subchannels(self._h, stream, out, max_out))
        return [out[i] for i in range(n)]

    def set_subchannels_at(self, subch, stream, at_cif, dab_plus=True):
        arr = (SubchDesc * max(1, len(subch)))()
        for i, c in enumerate(subch):
            dp = getattr(c, "dab_plus", -1)
            arr[i] = SubchDesc(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form, int(dab_plus) if dp < 0 else dp, 0)
        load().dabx_set_subchannels_at.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64]
        check(load().dabx_set_subchannels_at(self._h, stream, arr, len(subch), at_cif))
        self.subch = list(subch)

    def subch_stats(self, stream, j):
        st = SubchStats()
        check(load().dabx_get_subch_stats(self._h, stream, j, C.byref(st)))
        return {k: getattr(st, k) for k, _ in SubchStats._fields_}

    def read_tii(self, stream, min_frames=1, threshold_db=6, collisions=False, collision_sub_id=0, max_out=64):
        out = (TiiResult * max_out)()
        acc = C.c_int32(0)
        n = check(load().dabx_read_tii(self._h, stream, min_frames, threshold_db, int(collisions), collision_sub_id, out, max_out,
                                       C.byref(acc)))
        return [out[i].as_tuple() for i in range(n)], acc.value

    def read_eti(self, stream, max_frames=32):
        out = np.zeros((max_frames, 6144), np.uint8)
        lost = C.c_int32(0)
        n = check(load().dabx_read_eti(self._h, stream, max_frames, _p(out), C.byref(lost)))
        return out[:n], lost.value

    def read_msc(self, stream, j, n_cifs=4):
        nb = 3 * self.subch[j].kbps
        out = np.zeros((n_cifs, nb), np.uint8)
        n = check(load().dabx_read_msc(self._h, stream, j, n_cifs, _p(out)))
        return out[:n]

    def read_superframes(self, stream, j, n=1):
        nb = 110 * self.subch[j].kbps // 8
        out = np.zeros((n, nb), np.uint8)
        k = check(load().dabx_read_superframes(self._h, stream, j, n, _p(out)))
        return out[:k]

    def read_superframe_info(self, stream, j, n=1):
        """Records of the newest n super frames of slot j, oldest first (row i belongs to row i of read_superframes(stream, j, n))."""
        out = np.zeros(n, SUPERFRAME_INFO)
        k = check(load().dabx_read_superframe_info(self._h, stream, j, n, _p(out)))
        return out[:k]

    def read_soft(self, stream):
        out = np.zeros((75, 3072), np.int16)
        check(load().dabx_read_soft(self._h, stream, _p(out)))
        return out

    def ingest_open(self, fmt, slabs=2, max_frames=0, copy_engine=0):
        """fmt: numpy dtype (complex64 / int16 / uint8) or 0..2.  Returns the page-locked slabs as numpy arrays [n_streams * max_frames * TF (* 2)]."""
        code = fmt if isinstance(fmt, int) else {np.dtype(np.complex64): 0, np.dtype(np.int16): 1, np.dtype(np.uint8): 2}[np.dtype(fmt)]
        cfg = IngestConfig(host_slabs=slabs, fmt=code, max_frames=max_frames, copy_engine=copy_engine)
        L = load()
        L.dabx_ingest_submit.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        check(L.dabx_ingest_open(self._h, C.byref(cfg)))
        out = []
        for k in range(slabs):
            p, cap = C.c_void_p(), C.c_size_t()
            check(L.dabx_ingest_slab(self._h, k, C.byref(p), C.byref(cap)))
            dt = (np.complex64, np.int16, np.uint8)[code]
            out.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cap.value,)).view(dt))
        return out

    def ingest_open_formats(self, formats, slabs=2, max_frames=0, copy_engine=0):
        """The general form: formats[s] = IqFormat of stream s.  Returns (slabs as uint8 arrays [n_streams, pitch], pitch)."""
        assert len(formats) == self.n_streams
        cfg = IngestConfig(host_slabs=slabs, fmt=0, max_frames=max_frames, copy_engine=copy_engine)
        arr = (IqFormat * len(formats))(*formats)
        L = load()
        L.dabx_ingest_pitch.restype = C.c_longlong
        check(L.dabx_ingest_open_formats(self._h, C.byref(cfg), arr))
        pitch = check(L.dabx_ingest_pitch(self._h))
        out = []
        for k in range(slabs):
            p, cap = C.c_void_p(), C.c_size_t()
            check(L.dabx_ingest_slab(self._h, k, C.byref(p), C.byref(cap)))
            assert cap.value == pitch * self.n_streams
            out.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cap.value,)).reshape(self.n_streams, pitch))
        return out, pitch

    def ingest_submit_bytes(self, k, n_bytes):
        arr = (C.c_size_t * self.n_streams)(*[int(v) for v in n_bytes])
        check(load().dabx_ingest_submit_bytes(self._h, k, arr))

    def ingest_submit(self, k, n_samples):
        check(load().dabx_ingest_submit(self._h, k, n_samples))

    def ingest_commit(self, k):
        check(load().dabx_ingest_commit(self._h, k))

    def ingest_close(self):
        check(load().dabx_ingest_close(self._h))

    def delivery_open(self, slots=4, what=0, copy_engine=0):
        cfg = DeliveryConfig(host_slabs=slots, what=what, copy_engine=copy_engine)
        check(load().dabx_delivery_open(self._h, C.byref(cfg)))

    def delivery_close(self):
        check(load().dabx_delivery_close(self._h))

    def delivery_slab_bytes(self):
        load().dabx_delivery_slab_bytes.restype = C.c_longlong
        return check(load().dabx_delivery_slab_bytes(self._h))

    def delivery_info(self):
        out = DeliveryInfo()
        load().dabx_delivery_get_info.argtypes = [C.c_void_p, C.c_void_p]
        check(load().dabx_delivery_get_info(self._h, C.byref(out)))
        return {k: getattr(out, k) for k, _ in DeliveryInfo._fields_ if not k.startswith("reserved")}

    def delivery_wait_free(self, n=1, timeout_ms=-1):
        return check(load().dabx_delivery_wait_free(self._h, int(n), int(timeout_ms)))

    def delivery_next(self, wait=False):
        """The oldest chunk not yet fetched as a Chunk (views into the page-locked host slab), or None."""
        ref = ChunkRef()
        if check(load().dabx_delivery_next(self._h, int(wait), C.byref(ref))) == 0:
            return None
        return Chunk(self, ref)

    def stats(self, stream):
        st = Stats()
        check(load().dabx_get_stats_sized(self._h, stream, C.byref(st), C.c_size_t(C.sizeof(st))))
        return {k: getattr(st, k) for k, _ in Stats._fields_ if not k.startswith("reserved")}

    def counters(self):
        out = (C.c_int64 * 16)()
        check(load().dabx_get_counters(self._h, out))
        return dict(zip(COUNTER_NAMES, list(out)))


def eti_frame(cif_hi, cif_lo, minor, subch, fic96, msc):
    """Host only: one 6144-byte ETI(NI) frame; subch = SubchDesc list (FIC order), msc = list of byte arrays."""
    arr = (SubchDesc * max(1, len(subch)))(*subch)
    bufs = [np.ascontiguousarray(m, np.uint8) for m in msc]
    ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
    fic96 = np.ascontiguousarray(fic96, np.uint8)
    out = np.zeros(6144, np.uint8)
    used = check(load().dabx_eti_frame(cif_hi, cif_lo, minor, arr, len(subch), _p(fic96), ptrs, _p(out)))
    return out, used


def host_register(a):
    """Page-locks a numpy array's memory (hipHostRegister) so that pushes from it are DMA; returns the array."""
    check(load().dabx_host_register(_p(a), a.nbytes))
    return a


def host_unregister(a):
    check(load().dabx_host_unregister(_p(a)))


def probe_iq_file(path):
    """Host only: container / sample format of a recorded-IQ file (.raw/.iq, .sdr/.wav, .uff)."""
    fmt = IqFormat()
    check(load().dabx_probe_iq_file(os.fsencode(path), C.byref(fmt)))
    return fmt


def convert_iq_bytes(fmt, payload):
    """One-shot GPU decode (+ resample) of payload bytes -> complex64 at 2.048 MS/s."""
    payload = np.frombuffer(payload, np.uint8) if not isinstance(payload, np.ndarray) else np.ascontiguousarray(payload, np.uint8)
    n_in = payload.size // fmt.sample_bytes()
    cap = n_in if fmt.sample_rate == 2048000 else (n_in // (fmt.sample_rate // 1000) + 1) * 2048
    out = np.zeros(max(cap, 1), np.complex64)
    n = check(load().dabx_convert_iq_bytes(C.byref(fmt), _p(payload), payload.size, _p(out), out.size))
    return out[:n]


class Feed:
    """Streaming file feed into one stream's IQ ring (dabx_feed_*)."""

    def __init__(self, engine, stream, fmt):
        self._h = C.c_void_p()
        self.fmt = fmt
        check(load().dabx_feed_open(engine._h, stream, C.byref(fmt), C.byref(self._h)))

    def push(self, payload):
        payload = np.frombuffer(payload, np.uint8) if not isinstance(payload, np.ndarray) else np.ascontiguousarray(payload, np.uint8)
        return check(load().dabx_feed_bytes(self._h, _p(payload), payload.size))

    def bound(self, n_bytes):
        return check(load().dabx_feed_bound(self._h, n_bytes))

    def close(self):
        if self._h and _LIB is not None:
            _LIB.dabx_feed_close(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def play_file(engine, stream, path, block_frames=4, on_block=None):
    """Un-paced replay of a recorded file through one stream: probe, feed in blocks, process as frames become
    available.  Like the reference's readers (raw_reader.cpp:140-150, wav_reader.cpp:164-177) the final partial
    32768-unit read block is dropped.  Returns the number of frames processed."""
    fmt = probe_iq_file(path)
    unit = 32768 if fmt.family == 0 else 32768 * fmt.sample_bytes()      # raw: bytes, wav: frames
    if fmt.family == 2:
        unit = (fmt.sample_rate // 1000) * fmt.sample_bytes()            # uff: 1-ms reads, xml_reader.cpp:224-226
    n_units = fmt.data_bytes // unit
    feed = Feed(engine, stream, fmt)
    frames = 0
    per_block = max(1, (block_frames * 196608 * fmt.sample_bytes() * (fmt.sample_rate // 1000) // 2048) // unit)
    with open(path, "rb") as fh:
        fh.seek(fmt.data_offset)
        done = 0
        while done < n_units:
            take = min(per_block, n_units - done)
            feed.push(fh.read(take * unit))
            done += take
            before = engine.stats(stream)["frames"]
            engine.process(block_frames + 1)
            frames += engine.stats(stream)["frames"] - before
            if on_block:
                on_block(engine)
    feed.close()
    return frames


class FibdecInfo(C.Structure):
    _fields_ = [("fibs_processed", C.c_int64), ("fig00_fib", C.c_int64), ("last_change_fib", C.c_int64), ("cif_count", C.c_int32),
                ("cif_count_hi", C.c_int32), ("cif_count_lo", C.c_int32), ("change_flags", C.c_int32), ("occurrence_change", C.c_int32),
                ("n_changes", C.c_int32), ("n_restarts", C.c_int32), ("reserved", C.c_int32 * 3)]


class Reconf(C.Structure):
    _fields_ = [("pending", C.c_int32), ("n_changes", C.c_int32), ("frames_missed", C.c_int32), ("reserved", C.c_int32),
                ("at_cif", C.c_int64), ("last_change_cif", C.c_int64), ("frames_fed", C.c_int64)]


class FibDecoder:
    """dabx_fibdec_*: FibDecoder's FIG 0/0-0/2 walk with a current and a next configuration (host only)."""

    def __init__(self, reference_quirks=False):
        self._h = C.c_void_p()
        check(load().dabx_fibdec_create(C.byref(self._h)))
        if reference_quirks:
            check(load().dabx_fibdec_set_reference_quirks(self._h, 1))

    def process(self, fibs, crc_ok):
        fibs = np.ascontiguousarray(fibs, np.uint8).reshape(-1, 32)
        crc_ok = np.ascontiguousarray(crc_ok, np.uint8).reshape(-1)
        return check(load().dabx_fibdec_process(self._h, _p(fibs), _p(crc_ok), fibs.shape[0]))

    def info(self):
        out = FibdecInfo()
        check(load().dabx_fibdec_get_info(self._h, C.byref(out)))
        return {k: getattr(out, k) for k, _ in FibdecInfo._fields_ if not k.startswith("reserved")}

    def subchannels(self, next=False, max_out=64):
        out = (SubchDesc * max_out)()
        n = check(load().dabx_fibdec_subchannels(self._h, int(next), out, max_out))
        return [out[i] for i in range(n)]

    def reset(self):
        check(load().dabx_fibdec_reset(self._h))

    def close(self):
        if self._h and _LIB is not None:
            _LIB.dabx_fibdec_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_fibs(fibs, crc_ok, max_out=64):
    """FIB/FIG subset (host only): -> (list of SubchDesc, cif_count)."""
    fibs = np.ascontiguousarray(fibs, np.uint8).reshape(-1, 32)
    crc_ok = np.ascontiguousarray(crc_ok, np.uint8).reshape(-1)
    out = (SubchDesc * max_out)()
    cif = C.c_int32(-1)
    n = check(load().dabx_parse_fibs(_p(fibs), _p(crc_ok), fibs.shape[0], out, max_out, C.byref(cif)))
    return [out[i] for i in range(n)], cif.value


def fic_decode(soft):
    """soft: [batch, 3*3072] int16 (OFDM symbols 1..3) -> (fibs [batch,12,32], crc_ok [batch,12])."""
    soft = np.ascontiguousarray(soft, np.int16).reshape(-1, 9216)
    fibs = np.zeros((soft.shape[0], 12, 32), np.uint8)
    crc = np.zeros((soft.shape[0], 12), np.uint8)
    check(load().dabx_fic_decode(_p(soft), soft.shape[0], _p(fibs), _p(crc)))
    return fibs, crc
