"""GPU parity of the recorded-IQ path: payload bytes -> cf32 at 2.048 MS/s (iqfile.hip) vs the oracle restatement of
the reference's readers, bit for bit (all operations are single IEEE mul/add/div; the only reference-side freedom is
-ffast-math turning x/127.0f into a reciprocal multiply for UFF int8, which is why that case allows 1 ulp), then
whole files replayed through the engine."""
import os
import sys

import numpy as np
import pytest
from scipy.signal import resample_poly

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402
from tools import iq_files as iqf  # noqa: E402

pytestmark = pytest.mark.gpu


def _ora(fmt, payload):
    payload = np.ascontiguousarray(payload, np.uint8)
    cap = payload.size + 4096
    out = np.zeros(cap, np.complex64)
    n = ol.oracle().ora_iq_convert(fmt.family, fmt.container, fmt.big_endian, fmt.swap_iq, fmt.bits, fmt.sample_rate,
                                   payload, payload.size, out.ctypes.data, cap)
    return out[:n]


def _fmt(family, container, be=0, swap=0, bits=None, rate=2048000):
    bits = bits or (8, 8, 16, 24, 32, 32)[container]
    return dx.IqFormat(family, container, be, swap, bits, rate, 0, 0)


CASES = [(0, 0, 0, 0, None), (1, 0, 0, 0, None), (1, 2, 0, 0, None), (1, 2, 1, 0, None), (1, 3, 0, 0, None), (1, 4, 0, 0, None),
         (1, 5, 0, 0, None), (2, 1, 0, 0, None), (2, 0, 0, 1, None), (2, 2, 1, 0, 16), (2, 2, 0, 1, 12), (2, 3, 1, 1, 24),
         (2, 3, 0, 0, 20), (2, 4, 1, 0, 32), (2, 4, 0, 0, 28), (2, 5, 1, 1, None), (2, 5, 0, 0, None)]


@pytest.mark.parametrize("family,container,be,swap,bits", CASES)
def test_decode_is_bit_exact(family, container, be, swap, bits):
    rng = np.random.default_rng(container * 7 + family)
    fmt = _fmt(family, container, be, swap, bits)
    n = 70001
    if container == 5:
        vals = rng.standard_normal(2 * n).astype(np.float32)
        payload = vals.view(np.uint8).reshape(-1, 4)[:, ::-1].reshape(-1).copy() if be else vals.view(np.uint8)
    else:
        payload = rng.integers(0, 256, n * fmt.sample_bytes()).astype(np.uint8)
    got, want = dx.convert_iq_bytes(fmt, payload), _ora(fmt, payload)
    assert len(got) == len(want) == n
    if (family, container) == (2, 1):
        assert np.max(np.abs(got.view(np.float32) - want.view(np.float32))) <= 2.0 ** -23
    else:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("family,rate", [(1, 2500000), (2, 2500000), (1, 2000000), (2, 1792000), (1, 1536000), (2, 2999000)])
def test_resampled_stream_is_bit_exact_and_split_invariant(family, rate):
    rng = np.random.default_rng(rate // 1000 + family)
    fmt = _fmt(family, 2, 0, 0, 16, rate)
    n = (rate // 1000) * 37 + 123
    payload = rng.integers(0, 256, 4 * n).astype(np.uint8)
    want = _ora(fmt, payload)
    got = dx.convert_iq_bytes(fmt, payload)
    assert len(want) == 37 * 2048 if family == 2 else len(want) in (36 * 2048, 37 * 2048)
    assert len(got) == len(want) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the streaming feed: arbitrary (odd) splits give the same ring contents
    eng = dx.Engine(n_streams=2, ring_frames=2, max_subch=0, fic_only=1)
    feed = dx.Feed(eng, 1, fmt)
    pos, total = 0, 0
    for step in (1, 4097, 3, 20000, 1 << 20):
        total += feed.push(payload[pos:pos + step])
        pos += step
    assert total == len(want)
    ring = eng.read_iq(1, 0, total)
    assert np.array_equal(ring.view(np.uint32), want.view(np.uint32))
    with pytest.raises(dx.DabxError):                            # ring full: refused, state untouched
        feed.push(rng.integers(0, 256, 4 * (rate // 1000) * 200).astype(np.uint8))
    feed.close()
    eng.close()


@pytest.mark.parametrize("kind", ["raw", "sdr", "uff_i16_msb", "uff_f32"])
def test_file_replay_matches_pushing_the_same_samples(tmp_path, kind):
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(5, subch, seed=5)
    x = ds.channel(ens.iq, snr_db=25.0, cfo_hz=310.0, timing_offset=1000, seed=2, n_out=9 * ds.TF)
    g = 0.25 / np.sqrt(np.mean(np.abs(x) ** 2))
    path = str(tmp_path / ("rec." + {"raw": "iq", "sdr": "sdr"}.get(kind, "uff")))
    if kind == "raw":
        iqf.write_raw(path, x, g)
        fmt_expect = (0, 0)
    elif kind == "sdr":
        iqf.write_sdr(path, x, 2048000, g)
        fmt_expect = (1, 2)
    elif kind == "uff_i16_msb":
        iqf.write_uff(path, iqf.pack_int(iqf.to_int(x, 16, g), 2, True), 2048000, 16, "int16", "MSB")
        fmt_expect = (2, 2)
    else:
        iqf.write_uff(path, (x * g).astype(np.complex64).view(np.uint8), 2048000, 32, "float32", "LSB")
        fmt_expect = (2, 5)
    fmt = dx.probe_iq_file(path)
    assert (fmt.family, fmt.container) == fmt_expect
    with open(path, "rb") as fh:
        fh.seek(fmt.data_offset)
        payload = np.frombuffer(fh.read(fmt.data_bytes), np.uint8)
    samples = _ora(fmt, payload)                                  # what the reference's reader would hand the receiver
    eng = dx.Engine(n_streams=1, ring_frames=10, max_subch=18)
    eng.set_subchannels(subch)
    n_play = dx.play_file(eng, 0, path, block_frames=3)
    ref = dx.Engine(n_streams=1, ring_frames=10, max_subch=18)
    ref.set_subchannels(subch)
    unit = {0: 16384, 1: 32768, 2: 2048}[fmt.family]              # the readers drop the last partial block
    ref.push_iq(0, samples[:len(samples) // unit * unit])
    ref.process(9)
    a, b = eng.stats(0), ref.stats(0)
    assert n_play == a["frames"] == b["frames"] and a["frames"] >= 7
    for key in ("fib_ok", "fib_total", "sf_ok", "sf_fail", "au_ok", "last_start_index", "cifs_decoded"):
        assert a[key] == b[key], key
    assert a["fib_ok"] >= a["fib_total"] - 24 and a["sf_fail"] == 0
    fa, fb = eng.read_fibs(0, 4), ref.read_fibs(0, 4)
    assert np.array_equal(fa[0], fb[0]) and np.array_equal(fa[1], fb[1])
    assert np.array_equal(eng.read_msc(0, 3, 8), ref.read_msc(0, 3, 8))
    eng.close(); ref.close()


@pytest.mark.parametrize("kind,n_sub", [("sdr", 1), ("raw", 1), ("sdr", 18), ("raw", 18)])
def test_recorded_file_replay_equals_the_oracle_receiver_on_the_quantised_samples(tmp_path, kind, n_sub):
    """BASELINE configs[0] as worded -- a single recorded .sdr (and .raw) file, Mode I, ONE 64 kbit/s audio sub-channel, the
    CPU reference path in file-player mode -- and the same with all 18: the file is replayed through the engine
    (dabx_probe_iq_file + dabx_feed_bytes: bytes over PCIe, conversion on the GPU) and, independently, the oracle's reader
    (oracle/iqfile.c: the samples the reference's WavReader / RawReader would hand to DabProcessor, int16 / uint8 quantisation
    included) feeds the oracle RECEIVER.  FIBs and CRC flags of every frame, start indices, every logical frame and every
    RS-corrected super frame of the service(s) must be identical."""
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=55)
    x = ds.channel(ens.iq, snr_db=14.0, cfo_hz=-640.0, timing_offset=41000, seed=12, n_out=16 * ds.TF)
    g = 0.25 / np.sqrt(np.mean(np.abs(x) ** 2))
    path = str(tmp_path / ("rec." + {"raw": "iq", "sdr": "sdr"}[kind]))
    (iqf.write_raw(path, x, g) if kind == "raw" else iqf.write_sdr(path, x, 2048000, g))
    fmt = dx.probe_iq_file(path)
    with open(path, "rb") as fh:
        fh.seek(fmt.data_offset)
        payload = np.frombuffer(fh.read(fmt.data_bytes), np.uint8)
    samples = _ora(fmt, payload)
    unit = {0: 16384, 1: 32768}[fmt.family]                        # the readers drop the last partial block
    samples = np.ascontiguousarray(samples[:len(samples) // unit * unit])
    service = subch[:1] if n_sub == 1 else subch                   # one audio service (SubChId 1, CU 0..47), or the whole multiplex
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs(service), len(service))
    n = L.ora_rx_run(rx, samples, len(samples), 10000)
    cap = L.ora_rx_get_capture(rx).contents
    o_fibs = np.ctypeslib.as_array(cap.fibs, (n, 12, 32)).copy()
    o_crc = np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy()
    o_start = np.ctypeslib.as_array(cap.start_idx, (n,)).copy()
    o_msc = [ol.backend_bytes(rx, j, "msc").reshape(-1, 192) for j in range(len(service))]
    o_sf = [ol.backend_bytes(rx, j, "sf").reshape(-1, 880) for j in range(len(service))]
    L.ora_rx_destroy(rx)
    eng = dx.Engine(n_streams=1, ring_frames=10, max_subch=len(service), out_frames=4)
    eng.set_subchannels(service)
    fibs, crcs, starts = [], [], []

    def collect(e):
        st = e.stats(0)
        new = st["frames"] - len(fibs)
        if new:
            f, c = e.read_fibs(0, new)
            assert len(f) == new
            fibs.extend(f); crcs.extend(c)
            starts.append(st["last_start_index"])
    frames = dx.play_file(eng, 0, path, block_frames=3, on_block=collect)
    assert frames == len(fibs) and n - 1 <= frames <= n and frames >= 13       # the oracle also counts a last, partially read frame
    k = frames
    assert np.array_equal(np.array(crcs), o_crc[:k]) and np.array_equal(np.array(fibs), o_fibs[:k])
    assert o_crc[6:k].all() and starts[-1] == o_start[k - 1]
    n_lf = 4 * k - 16
    for j in range(len(service)):
        sub = eng.subch_stats(0, j)
        assert sub["cifs_decoded"] == n_lf
        m = min(n_lf, 32)
        assert np.array_equal(eng.read_msc(0, j, m), o_msc[j][n_lf - m:n_lf]), j
        q = min(4, sub["sf_ok"])
        assert sub["sf_ok"] >= 5 and np.array_equal(eng.read_superframes(0, j, q), o_sf[j][sub["sf_ok"] - q:sub["sf_ok"]]), j
        assert any(np.array_equal(eng.read_superframes(0, j, 1)[0], t) for t in ens.superframes[j])      # and it is what was transmitted
    eng.close()


@pytest.mark.parametrize("rate,up,down", [(2000000, 125, 128), (2500000, 625, 512)])
def test_recording_at_another_rate_decodes_after_gpu_resampling(tmp_path, rate, up, down):
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(5, subch, seed=8)
    x = ds.channel(ens.iq, snr_db=30.0, cfo_hz=-150.0, timing_offset=300, seed=3, n_out=10 * ds.TF)
    y = resample_poly(x.astype(np.complex128), up, down).astype(np.complex64)       # the recorder's view at `rate`
    g = 0.25 / np.sqrt(np.mean(np.abs(y) ** 2))
    path = str(tmp_path / "rec.sdr")
    iqf.write_sdr(path, y, rate, g)
    eng = dx.Engine(n_streams=1, ring_frames=10, max_subch=18)
    eng.set_subchannels(subch)
    frames = dx.play_file(eng, 0, path, block_frames=3)
    st = eng.stats(0)
    assert frames >= 8 and st["fib_ok"] >= st["fib_total"] - 24 and st["sf_ok"] >= 18 and st["sf_fail"] == 0
    sf = eng.read_superframes(0, 7, 1)
    assert any(np.array_equal(sf[0], ens.superframes[7][q]) for q in range(len(ens.superframes[7])))
    eng.close()


@pytest.mark.parametrize("rate", [2048000, 2500000])
def test_long_replay_through_a_small_ring(tmp_path, rate):
    """32 frames through a 6-frame IQ ring: the feed wraps the ring many times (also with the GPU resampler in the path),
    refuses blocks that do not fit yet, and nothing is lost: every sub-channel ends with the expected number of
    super frames and not one failure."""
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=77)
    x = ds.channel(ens.iq, snr_db=24.0, cfo_hz=95.0, timing_offset=900, seed=5, n_out=32 * ds.TF)
    if rate != 2048000:
        x = resample_poly(x.astype(np.complex128), 625, 512).astype(np.complex64)
    path = str(tmp_path / "long.sdr")
    iqf.write_sdr(path, x, rate, 0.25 / np.sqrt(np.mean(np.abs(x) ** 2)))
    eng = dx.Engine(n_streams=1, ring_frames=6, max_subch=18)
    eng.set_subchannels(subch)
    fmt = dx.probe_iq_file(path)
    feed = dx.Feed(eng, 0, fmt)
    refused = 0
    with open(path, "rb") as fh:
        fh.seek(fmt.data_offset)
        block = 2 * ds.TF * fmt.sample_bytes() * rate // 2048000 + 4 * 1237          # ~2 frames, deliberately odd-sized
        while True:
            data = fh.read(block)
            if not data:
                break
            while True:
                try:
                    feed.push(data)
                    break
                except dx.DabxError:                                                # ring full: decode, then offer it again
                    refused += 1
                    eng.process(2)
            eng.process(1)
    eng.process(6)
    st = eng.stats(0)
    assert st["frames"] >= 30 and st["fib_ok"] >= st["fib_total"] - 24 and st["sf_fail"] == 0
    assert st["sf_ok"] >= 18 * ((st["frames"] * 4 - 16) // 5 - 1)
    assert refused >= 1 or st["samples_consumed"] > 6 * ds.TF                       # the ring did wrap
    feed.close()
    eng.close()


@pytest.mark.parametrize("container,be,swap,bits,rate", [(3, 1, 0, 24, 2048000), (3, 1, 1, 24, 2048000), (3, 1, 0, 20, 2500000),
                                                         (3, 1, 1, 24, 1792000), (5, 1, 1, 32, 2048000), (5, 0, 1, 32, 2000000)])
def test_reference_quirks_mode_reproduces_the_uff_reader_defects_bit_for_bit(container, be, swap, bits, rate):
    """dabx_iq_format.reference_quirks = 1: what a user of the reference gets from such a file -- int24/MSB with Q's middle
    byte taken from lbuf[4*i+4] of the 1-ms read block (xml_reader.cpp:316,462), QI/int24/MSB sign-extended with
    0x7F000000 (:465,:469), QI/float32 not swapped (:530,:540) -- against the oracle's literal restatement of those loops;
    the default mode decodes the format's evident meaning, and the two differ."""
    rng = np.random.default_rng(container * 100 + be * 10 + swap + rate // 1000)
    M = rate // 1000
    n = M * 23 + 77                                             # whole read blocks + a tail the reference never delivers
    if container == 5:
        vals = rng.standard_normal(2 * n).astype(np.float32)
        payload = vals.view(np.uint8).reshape(-1, 4)[:, ::-1].reshape(-1).copy() if be else vals.view(np.uint8).copy()
    else:
        payload = rng.integers(0, 256, n * 6).astype(np.uint8)
    fq = dx.IqFormat(2, container, be, swap, bits, rate, 0, 0, 1, 0)
    f0 = dx.IqFormat(2, container, be, swap, bits, rate, 0, 0, 0, 0)
    cap = payload.size + 4096
    want = np.zeros(cap, np.complex64)
    nw = ol.oracle().ora_iq_convert_q(2, container, be, swap, bits, rate, payload, payload.size, want.ctypes.data, cap, 1)
    want = want[:nw]
    got = dx.convert_iq_bytes(fq, payload)
    assert len(got) == len(want) == 23 * 2048 if rate != 2048000 else len(got) == len(want) == 23 * M
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    plain = dx.convert_iq_bytes(f0, payload)
    assert not np.array_equal(plain[:len(got)].view(np.uint32), got.view(np.uint32))        # the defects are visible in the samples
    # streaming feed: any split of the payload gives the same samples (read blocks are re-assembled inside the feed)
    eng = dx.Engine(n_streams=1, ring_frames=2, max_subch=0, fic_only=1)
    feed = dx.Feed(eng, 0, fq)
    pos, total = 0, 0
    for step in (5, 6143, 1, 40000, 1 << 22):
        total += feed.push(payload[pos:pos + step])
        pos += step
    assert total == len(want)
    assert np.array_equal(eng.read_iq(0, 0, total).view(np.uint32), want.view(np.uint32))
    feed.close()
    eng.close()


def test_reference_quirks_mode_refuses_what_the_reference_leaves_undefined():
    payload = np.zeros(4096, np.uint8)
    with pytest.raises(dx.DabxError, match="423"):
        dx.convert_iq_bytes(dx.IqFormat(2, 0, 0, 1, 8, 2048000, 0, 0, 1, 0), payload)       # UFF QI/uint8: out-of-bounds table walk
    assert len(dx.convert_iq_bytes(dx.IqFormat(2, 0, 0, 1, 8, 2048000, 0, 0, 0, 0), payload)) == 2048   # default mode decodes it
