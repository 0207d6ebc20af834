"""Bulk ingest (include/dabx.h "Bulk ingest"): one page-locked slab with the next samples of EVERY stream, one SDMA transfer, one
conversion kernel, one commit -- decodes to exactly what the per-stream pushes (dabx_push_iq) decode to, for all three sample formats of
IDeviceHandler::getSamples' callers (cf32; int16 /32768, wav_reader.cpp:164; uint8 (x - 127.38)/128, raw_reader.cpp:66-70)."""
import os
import sys

import numpy as np
import pytest

from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402
from test_gpu_engine import _oracle_run  # noqa: E402

pytestmark = pytest.mark.gpu
TF = ds.TF


def _quantise(x, dtype):
    pairs = np.ascontiguousarray(x).view(np.float32)
    if dtype == np.int16:
        return np.clip(np.round(pairs * 32768.0), -32768, 32767).astype(np.int16)
    if dtype == np.uint8:
        return np.clip(np.round(pairs * 128.0 + 127.38), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(x)


def _dequantise(q, dtype):
    if dtype == np.int16:
        return (q.astype(np.float32) / np.float32(32768.0)).view(np.complex64)
    if dtype == np.uint8:
        return ((q.astype(np.float32) - np.float32(127.38)) / np.float32(128.0)).view(np.complex64)
    return q


@pytest.mark.parametrize("dtype,copy_engine", [(np.uint8, 0), (np.int16, 0), (np.complex64, 0), (np.uint8, 1)])
def test_slab_ingest_decodes_like_per_stream_pushes_and_like_the_oracle(dtype, copy_engine):
    subch = ds.default_subchannels(18, 64)
    S, n_frames, chunk = 3, 27, 4
    xs = []
    for s in range(S):
        ens = ds.build_ensemble(10, subch, seed=60 + s)
        xs.append(_quantise(ds.channel(ens.iq, snr_db=19.0, cfo_hz=400.0 * (s - 1), timing_offset=9000 * s + 77, seed=60 + s, n_out=n_frames * TF), dtype))
    per = chunk * TF * (1 if dtype == np.complex64 else 2)                # array elements per stream and slab
    a = dx.Engine(n_streams=S, ring_frames=3 * chunk, max_subch=18, out_frames=8)      # bulk ingest
    b = dx.Engine(n_streams=S, ring_frames=3 * chunk, max_subch=18, out_frames=8)      # one push per stream and chunk
    a.set_subchannels(subch); b.set_subchannels(subch)
    slabs = a.ingest_open(dtype, slabs=2, max_frames=chunk, copy_engine=copy_engine)
    assert len(slabs) == 2 and slabs[0].size == S * per
    n_chunks = n_frames // chunk
    with pytest.raises(dx.DabxError, match="was not submitted"):
        a.ingest_commit(0)

    def fill(k):
        for s in range(S):
            slabs[k % 2][s * per:(s + 1) * per] = xs[s][k * per:(k + 1) * per]
    fill(0)
    a.ingest_submit(0, chunk * TF)
    with pytest.raises(dx.DabxError, match="not committed"):
        a.ingest_submit(0, chunk * TF)
    for k in range(n_chunks):
        if k + 1 < n_chunks:                       # the next slab goes on the link while this one is decoded
            fill(k + 1)
            a.ingest_submit((k + 1) % 2, chunk * TF)
        a.ingest_commit(k % 2)
        a.process(chunk, sync=False)
        for s in range(S):
            b.push_iq(s, xs[s][k * per:(k + 1) * per])
        b.process(chunk, sync=False)
    a.synchronize(); b.synchronize()
    for s in range(S):
        sa, sb = a.stats(s), b.stats(s)
        assert sa["frames"] == sb["frames"] >= n_chunks * chunk - 4 and sa == sb, s
        fa, ca = a.read_fibs(s, 8); fb, cb = b.read_fibs(s, 8)
        assert np.array_equal(fa, fb) and np.array_equal(ca, cb) and ca.all()
        for j in range(18):
            assert np.array_equal(a.read_msc(s, j, 16), b.read_msc(s, j, 16)), (s, j)
            assert np.array_equal(a.read_superframes(s, j, 4), b.read_superframes(s, j, 4)), (s, j)
    # ... and stream 1 against the oracle receiver on the samples as the device sees them
    ora = _oracle_run(np.ascontiguousarray(_dequantise(xs[1][: n_chunks * per], dtype)), subch)
    f = a.stats(1)["frames"]
    fa, ca = a.read_fibs(1, 8)
    assert f <= ora["n"] and np.array_equal(fa, ora["fibs"][f - 8:f]) and np.array_equal(ca, ora["crc"][f - 8:f])
    k = 4 * f - 16
    for j in range(18):
        assert np.array_equal(a.read_msc(1, j, 16), ora["msc"][j].reshape(-1, 192)[k - 16:k]), j
    a.ingest_close()
    a.close(); b.close()


def test_slab_ingest_of_recordings_at_three_rates_and_unequal_lengths():
    """what recordings really look like.  Eight streams, every one its own container / byte order / sample rate (2.048, 2.5
    and 1.792 MS/s: the readers' 1-ms linear interpolation, wav_reader.cpp:67-82,190-206 / xml_reader.cpp:237-244, runs on the device with
    its state carried from slab to slab) and its own LENGTH (they end at different slabs; odd payload tails wait for the next slab on the
    host).  ONE slab, one SDMA transfer and two kernel launches per commit -- byte-identical to eight per-stream dabx_feed_bytes feeds, and
    to the oracle's reader + receiver."""
    from scipy.signal import resample_poly
    from tools import iq_files as iqf
    import oracle_lib as ol
    subch = ds.default_subchannels(6, 64)
    #        family, container, big-endian, swap, bits, rate, (up, down)
    kinds = [(1, 2, 0, 0, 16, 2048000, None), (2, 2, 1, 0, 16, 2500000, (625, 512)), (0, 0, 0, 0, 8, 2048000, None), (1, 2, 0, 0, 16, 2500000, (625, 512)),
             (2, 5, 0, 1, 32, 1792000, (7, 8)), (2, 3, 1, 0, 24, 2048000, None), (1, 5, 0, 0, 32, 2500000, (625, 512)), (2, 1, 0, 0, 8, 1792000, (7, 8))]
    S = len(kinds)
    lengths = [21, 17, 24, 13, 19, 24, 15, 22]                 # frames of signal each recording holds
    fmts, payloads = [], []
    for s, (fam, cont, be, swap, bits, rate, ud) in enumerate(kinds):
        ens = ds.build_ensemble(10, subch, seed=90 + s)
        x = ds.channel(ens.iq, snr_db=20.0 + s, cfo_hz=150.0 * (s - 3), timing_offset=5000 * s + 33, seed=90 + s, n_out=lengths[s] * TF)
        if ud:
            x = resample_poly(x.astype(np.complex128), ud[0], ud[1]).astype(np.complex64)
        g = 0.25 / np.sqrt(np.mean(np.abs(x) ** 2))
        pairs = np.ascontiguousarray(x * g).view(np.float32)
        if swap:
            pairs = pairs.reshape(-1, 2)[:, ::-1].reshape(-1)
        if cont == 0:
            raw = np.clip(np.round(pairs * 128.0 + (127.38 if fam != 1 else 128.0)), 0, 255).astype(np.uint8).tobytes()
        elif cont == 1:
            raw = np.clip(np.round(pairs * 127.0), -127, 127).astype(np.int8).tobytes()
        elif cont == 5:
            raw = pairs.astype(">f4" if be else "<f4").tobytes()
        else:
            nb = {2: 2, 3: 3, 4: 4}[cont]
            v = np.clip(np.round(pairs.astype(np.float64) * (1 << (bits - 1))), -(1 << (bits - 1)), (1 << (bits - 1)) - 1).astype(np.int64)
            raw = iqf.pack_int(v, nb, bool(be)).tobytes()
        fmts.append(dx.IqFormat(fam, cont, be, swap, bits, rate, 0, len(raw)))
        payloads.append(np.frombuffer(raw, np.uint8))
    chunk = 4
    a = dx.Engine(n_streams=S, ring_frames=3 * chunk + 2, max_subch=6, out_frames=8)      # bulk ingest, general form
    b = dx.Engine(n_streams=S, ring_frames=3 * chunk + 2, max_subch=6, out_frames=8)      # one feed per stream
    a.set_subchannels(subch); b.set_subchannels(subch)
    slabs, pitch = a.ingest_open_formats(fmts, slabs=2, max_frames=chunk)
    feeds = [dx.Feed(b, s, fmts[s]) for s in range(S)]
    # per stream and slab: about `chunk` frames' worth of ITS payload, deliberately not a whole number of samples (the host keeps the tail)
    per = [int(chunk * TF * fmts[s].sample_bytes() * (fmts[s].sample_rate / 2048000.0)) // 7 * 7 + 3 for s in range(S)]
    assert max(per) <= pitch
    pos = [0] * S
    with pytest.raises(dx.DabxError, match="whole samples"):
        a.ingest_submit_bytes(0, [3] + [0] * (S - 1))

    def fill(k):
        nb = []
        for s in range(S):
            sb = fmts[s].sample_bytes()
            take = min(per[s], len(payloads[s]) - pos[s]) // sb * sb        # whole samples; an odd tail stays with the "reader"
            slabs[k % 2][s, :take] = payloads[s][pos[s]:pos[s] + take]
            nb.append(take)
        return nb
    k = 0
    nb = fill(0)
    a.ingest_submit_bytes(0, nb)
    while True:
        cur = nb
        for s in range(S):
            pos[s] += cur[s]
        more = any(pos[s] + fmts[s].sample_bytes() <= len(payloads[s]) for s in range(S))
        if more:
            nb = fill(k + 1)
            a.ingest_submit_bytes((k + 1) % 2, nb)
        a.ingest_commit(k % 2)
        a.process(chunk + 1, sync=False)
        for s in range(S):
            if cur[s]:
                feeds[s].push(payloads[s][pos[s] - cur[s]:pos[s]])
        b.process(chunk + 1, sync=False)
        k += 1
        if not more:
            break
    assert k >= 5
    for e in (a, b):
        e.process(3)
    for s in range(S):
        sa, sb_ = a.stats(s), b.stats(s)
        assert sa["frames"] == sb_["frames"] >= lengths[s] - 3 and sa == sb_, (s, sa["frames"], sb_["frames"], lengths[s])
        fa, ca = a.read_fibs(s, 8); fb, cb = b.read_fibs(s, 8)
        assert np.array_equal(fa, fb) and np.array_equal(ca, cb) and ca.all(), s
        for j in range(6):
            assert np.array_equal(a.read_msc(s, j, 16), b.read_msc(s, j, 16)), (s, j)
            assert np.array_equal(a.read_superframes(s, j, 3), b.read_superframes(s, j, 3)), (s, j)
    # ... and against the oracle's own reader + receiver: a resampled big-endian UFF (1), a swapped float UFF at 1.792 MS/s (4), raw u8 (2)
    for s in (1, 4, 2):
        f = fmts[s]
        used = payloads[s][:pos[s]]
        cap = used.size + 4096
        x = np.zeros(cap, np.complex64)
        n = ol.oracle().ora_iq_convert(f.family, f.container, f.big_endian, f.swap_iq, f.bits, f.sample_rate, used, used.size, x.ctypes.data, cap)
        ora = _oracle_run(np.ascontiguousarray(x[:n]), subch)
        fr = a.stats(s)["frames"]
        fa, ca = a.read_fibs(s, 8)
        assert fr <= ora["n"] and np.array_equal(fa, ora["fibs"][fr - 8:fr]) and np.array_equal(ca, ora["crc"][fr - 8:fr]), s
        kk = 4 * fr - 16
        for j in range(6):
            assert np.array_equal(a.read_msc(s, j, 16), ora["msc"][j].reshape(-1, 192)[kk - 16:kk]), (s, j)
    for fd in feeds:
        fd.close()
    a.ingest_close()
    a.close(); b.close()
