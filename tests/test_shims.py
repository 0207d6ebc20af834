"""The HIP class shims (shim/ofdm_decoder_hip.h, fic_decoder_hip.h, msc_handler_hip.h): the reference's OfdmDecoder /
FicDecoder / MscHandler class surface bound to libdabx's per-symbol stage entries.

CPU: the shims compile with plain g++, every member signature SURVEY 8(b) lists is asserted at compile time -- against a
standalone vocabulary and, where the reference tree is present, against the reference's OWN headers -- and the driver
refuses to run without a GPU.  GPU: one ensemble goes symbol by symbol through the three classes, exactly the calls
DabProcessor makes, and the FIBs handed to IFibDecoder::process_FIB and the logical frames handed to
BackendDriver::add_to_frame equal the oracle's class-level run on the same FFT outputs, the engine's bytes and what was
transmitted."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)
CXX = os.path.join(ROOT, "tests", "cxx")
BUILD = os.path.join(CXX, "_build")


def _make(target):
    from dabstar_amd import lib as dx
    dx.load()
    subprocess.run(["make", "-s", "-C", CXX, target], check=True)


def test_shims_compile_and_have_the_reference_signatures():
    _make("shims")
    p = subprocess.run([os.path.join(BUILD, "shim_signatures")], capture_output=True, text=True)
    assert p.returncode == 0 and "OfdmDecoder 10, FicDecoder 9, MscHandler 6" in p.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/src") or not os.path.isdir("/opt/conda/include/qt"),
                    reason="needs the reference tree and Qt headers (present in the build container only)")
def test_shims_type_check_against_the_references_own_headers():
    """-DSHIM_IN_TREE: shim/*.h include glob_defs.h, dab_constants.h, glob_enums.h, ringbuffer.h, fib_decoder_if.h and
    backend_driver.h where they lie under /root/reference and the same static_asserts hold on the reference's types."""
    p = subprocess.run(["make", "-s", "-C", CXX, "shims-in-tree"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


def test_cmake_fragment_and_integration_notes_name_every_shim_file():
    frag = open(os.path.join(ROOT, "shim", "dab_hip.cmake")).read()
    for f in ("ofdm_decoder_hip.h", "fic_decoder_hip.h", "msc_handler_hip.h", "dab_hip_gui.cpp", "DAB_HIP", "dabx"):
        assert f in frag, f


def test_symbol_driver_refuses_to_run_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    _make("shims")
    p = subprocess.run([os.path.join(BUILD, "shim_symbols"), str(tmp_path / "x"), str(tmp_path / "o")], capture_output=True, text=True)
    assert p.returncode == 3 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_one_ensemble_symbol_by_symbol_through_the_three_shims(tmp_path):
    import oracle_lib as ol
    from dabstar_amd import lib as dx
    from tools import dab_synth as ds
    _make("shims")
    subch = [ds.SubCh(3, 0, 48, 64, 2, 0), ds.SubCh(9, 100, 96, 128, 2, 0), ds.SubCh(17, 300, 112, 112, 1, 0), ds.SubCh(21, 500, 24, 32, 2, 0)]
    n_frames = 12
    ens = ds.build_ensemble(10, subch, seed=33, cif_start=240)
    x = ds.channel(ens.iq, snr_db=19.0, cfo_hz=0.0, timing_offset=0, seed=33, n_out=n_frames * ds.TF)
    # FFT outputs of every symbol (timing known: frame = null + 76 symbols) through the stage FFT of the library
    fr = x.reshape(n_frames, ds.TF)
    win = np.stack([fr[:, 2656 + l * 2552 + 504: 2656 + l * 2552 + 2552] for l in range(76)], axis=1)        # [F][76][2048]
    spectra = dx.fft2048(np.ascontiguousarray(win.reshape(-1, 2048))).reshape(n_frames, 76, 2048)
    nulls = dx.fft2048(np.ascontiguousarray(fr[:, 504:2552]))
    clock = np.zeros(n_frames, np.float32)
    set_at, stop_at = [0, 0, 0, 3], [-1, -1, 8, -1]          # service 3 joins at frame 3, service 2 stops at frame 8
    inp = tmp_path / "spectra.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<ii", n_frames, len(subch)))
        for c, a, b in zip(subch, set_at, stop_at):
            f.write(struct.pack("<8i", c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form, a, b))
        for k in range(n_frames):
            f.write(spectra[k].astype(np.complex64).tobytes())
            f.write(nulls[k].astype(np.complex64).tobytes())
            f.write(struct.pack("<f", 0.0))
    out = str(tmp_path / "shim")
    p = subprocess.run([os.path.join(BUILD, "shim_symbols"), str(inp), out], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["frames"] == n_frames and res["drivers"] == 4 and res["stopped_ok"] and res["ratio_reset"]
    # the status signals: one signal_fic_status (48 FIC blocks: the 40th reports) with the channel BER of a 19-dB channel, one LCD record
    # (after 5 frames of symbols, on symbol 1 of the frame that follows: ofdm_decoder.cpp:155-157; the next would fall into a 13th frame)
    assert 0.0 <= res["fic_status_ber"] < 5e-3 and res["lcd_count"] == 1 and 15.0 < res["lcd_snr"] < 30.0 and res["lcd_symbol"] == 2
    # ... whose MER (:204-208, 331-340: the phase-deviation IIR has seen 5 frames and a symbol, not settled yet) and TestData1 = mMeanValue come from the device too
    assert 10.0 < res["lcd_mer"] < 30.0 and res["lcd_mean_value"] > 0.0

    # ---- oracle: the same per-symbol class calls on the same FFT outputs (all four services from frame 0)
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
    assert L.ora_rx_run_spectra(rx, np.ascontiguousarray(spectra.astype(np.complex64)), np.ascontiguousarray(nulls.astype(np.complex64)), clock, n_frames) == n_frames
    cap = L.ora_rx_get_capture(rx).contents
    o_fibs = np.ctypeslib.as_array(cap.fibs, (n_frames, 12, 32)).copy()
    o_crc = np.ctypeslib.as_array(cap.fib_crc, (n_frames, 12)).copy()
    o_msc = [ol.backend_bytes(rx, i, "msc").reshape(-1, 3 * c.kbps) for i, c in enumerate(subch)]
    L.ora_rx_destroy(rx)
    assert o_crc.all()

    # FIBs handed to IFibDecoder::process_FIB: CRC-clean ones, in FIB order, tagged with their FIC block
    rec = np.fromfile(out + ".fibs", np.uint8).reshape(-1, 34)
    assert len(rec) == int(o_crc.sum()) == res["fibs_delivered"]
    assert np.array_equal(rec[:, 2:], o_fibs.reshape(-1, 32)[o_crc.reshape(-1) != 0])
    assert np.array_equal(rec[:, 0].astype(int) + 256 * rec[:, 1].astype(int), np.tile(np.repeat(np.arange(4), 3), n_frames))
    assert np.array_equal(o_fibs, ens.fibs[np.arange(n_frames) % 10])                      # ... and they are the transmitted FIBs
    # get_fib_bits after every frame: 3072 bits one per byte + the four FIC-valid flags
    fb = np.fromfile(out + ".fibbits", np.uint8).reshape(n_frames, 3076)
    assert np.array_equal(np.packbits(fb[:, :3072], axis=1).reshape(n_frames, 12, 32), o_fibs) and (fb[:, 3072:] == 1).all()
    assert res["mean_fic_ratio"] > 95.0

    # logical frames handed to BackendDriver::add_to_frame: start 16 CIFs after set_channel, stop with stop_service
    for i, c in enumerate(subch):
        got = np.fromfile(out + ".svc%d" % i, np.uint8).reshape(-1, 3 * c.kbps)
        first = 4 * set_at[i] + 16
        last = 4 * (stop_at[i] if stop_at[i] >= 0 else n_frames)
        assert len(got) == last - first, (i, len(got))
        assert np.array_equal(got, o_msc[i][first - 16:last - 16]), i                     # the oracle's back ends ran from CIF 0
        assert np.array_equal(got, ens.msc_bytes[i][(np.arange(first, last) - 16) % 40]), i   # == transmitted (cyclic, 40 CIFs)

    # ---- the frame-batched engine on the same IQ decodes the same bytes (it finds its own frame start: its frame numbering
    #      lags the buffer's by the frames its acquisition consumed, so sequences are matched cyclically: 10 frames / 40 CIFs)
    eng = dx.Engine(n_streams=1, ring_frames=n_frames + 1, max_subch=4, out_frames=8)
    eng.set_subchannels(subch, dab_plus=False)
    eng.push_iq(0, x)
    eng.process(n_frames)
    f = eng.stats(0)["frames"]
    assert f >= n_frames - 3
    e_fibs, e_crc = eng.read_fibs(0, 8)
    assert e_crc.all()
    off = [q for q in range(10) if np.array_equal(e_fibs[0], ens.fibs[q])]
    assert len(off) == 1
    assert np.array_equal(e_fibs, ens.fibs[(off[0] + np.arange(8)) % 10])          # == the FIBs the shims delivered (o_fibs == ens.fibs above)
    for i, c in enumerate(subch):
        got = eng.read_msc(0, i, 16)
        q0 = [q for q in range(40) if np.array_equal(got[0], ens.msc_bytes[i][q])]
        assert len(q0) == 1, i
        assert np.array_equal(got, ens.msc_bytes[i][(q0[0] + np.arange(16)) % 40]), i   # the logical frames the shims handed to add_to_frame
    eng.close()
