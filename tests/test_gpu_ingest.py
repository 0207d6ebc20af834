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
