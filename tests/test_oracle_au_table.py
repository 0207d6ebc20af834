"""The oracle's side of SURVEY 8 a21 (no GPU): the 32-byte super-frame records oracle/msc.c writes -- the layout of include/dabx.h's
dabx_superframe_info -- are the header's AU table and the CRCs' verdicts (mp4processor.cpp:249-333), checked against a CRC and a table walk
written here independently, on a clean signal and at 4 dB where access units fail."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402
from dabstar_amd import lib as dx  # noqa: E402
from test_gpu_engine import _oracle_run  # noqa: E402
from test_gpu_au_table import _check_record_against_its_super_frame  # noqa: E402


@pytest.mark.parametrize("snr", [20.0, 4.0])
def test_oracle_records_describe_their_super_frames(snr):
    subch = ds.default_subchannels(2, 64)
    ens = ds.build_ensemble(10, subch, seed=1)
    x = ds.channel(ens.iq, snr_db=snr, cfo_hz=-730.0, timing_offset=123456, seed=1, n_out=16 * ds.TF)
    ora = _oracle_run(x, subch)
    bad = 0
    for j in range(2):
        r = ora["sfi"][j].view(dx.SUPERFRAME_INFO)
        sf = ora["sf"][j].reshape(-1, 880)
        st = ora["stats"][j]
        assert len(r) == len(sf) == st["sf_ok"] >= 5
        for i in range(len(r)):
            _check_record_against_its_super_frame(r[i], sf[i], 64)
            assert r[i]["first_frame"] % 5 == r[0]["first_frame"] % 5
        ok = sum(bin(int(v)).count("1") for v in r["au_crc_ok"])
        assert ok == st["au_ok"] and int(r["num_aus"].sum()) - ok == st["au_bad"]
        bad += st["au_bad"]
        if snr >= 20:
            assert np.array_equal(sf[-1], ens.superframes[j][np.argmax([np.array_equal(sf[-1], q) for q in ens.superframes[j]])])
    assert (bad > 0) == (snr < 5)
