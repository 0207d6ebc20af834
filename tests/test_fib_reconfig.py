"""FIG 0/0 change flags, current / next configuration and their swap (fib_decoder_fig0.cpp:89-112, :149, :240; fib_decoder.h:97):
libdabx's running FIB decoder (dabx_fibdec_*) against the oracle restatement (oracle/fib.c, ora_fibdec_*) and against what the
synthetic transmitter announced.  CPU only."""
import os
import sys

import numpy as np

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402


def _tab(descs):
    return [(g.subch_id, g.cu_start, g.cu_size, g.kbps, g.prot_level, g.short_form, g.dab_plus) for g in descs]


def _layouts():
    a = [ds.SubCh(i, 48 * i, 48, 64, 2, 0) for i in range(6)]
    b = a[:3] + [ds.SubCh(3, 400, 48, 64, 2, 0),            # moves
                 ds.SubCh(4, 500, 72, 96, 2, 0),            # grows: 64 -> 96 kbit/s
                 ds.SubCh(6, 192, 24, 32, 2, 0, dab_plus=0)]   # 5 ends, 6 begins
    return a, b


def _fib_stream(a, b, n_cif, switch, announce_from, cif_start=0):
    out = []
    for q in range(n_cif):
        ann = announce_from <= q < switch
        out.append(ds.build_fibs_reconf(a if q < switch else b, b if ann else None, cif_start + q, 3 if ann else 0,
                                        (cif_start + switch) % 250).reshape(3, 32))
    return np.concatenate(out)


def _both(fibs, crc, step=3):
    """feeds both decoders `step` FIBs at a time; yields (index of the first FIB of the chunk, libdabx decoder, oracle decoder)"""
    d, o = dx.FibDecoder(reference_quirks=True), ol.OraFibDecoder()        # the oracle restates the reference: its swap rule
    for i in range(0, len(fibs), step):
        nd, no = d.process(fibs[i:i + step], crc[i:i + step]), o.process(fibs[i:i + step], crc[i:i + step])
        assert nd == no
        yield i, d, o
    d.close(); o.close()


def test_announced_reconfiguration_swaps_at_the_first_fig00_with_cleared_flags():
    a, b = _layouts()
    switch, ann = 60, 28
    fibs = _fib_stream(a, b, 100, switch, ann, cif_start=4990)          # the CIF counter wraps 4999 -> 0 on the way
    crc = np.ones(len(fibs), np.uint8)
    want_a = [(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, 0, 1) for c in a]
    want_b = [(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, 0, int(c.dab_plus)) for c in b]
    for i, d, o in _both(fibs, crc):
        q = i // 3                                                       # the CIF whose FIC group was just fed
        inf = d.info()
        assert inf == o.info(), (q, inf, o.info())
        cur, nxt = _tab(d.subchannels()), _tab(d.subchannels(next=True))
        assert cur == o.subchannels() and nxt == o.subchannels(next=True), q
        assert inf["cif_count"] == (4990 + q) % 5000 and inf["fig00_fib"] == 3 * q
        if q >= 1:
            assert sorted(cur) == sorted(want_a if q < switch else want_b), q      # complete after two CIFs, swapped AT the switch
        assert inf["change_flags"] == (3 if ann <= q < switch else 0)
        if ann <= q < switch:
            assert inf["occurrence_change"] == (4990 + switch) % 250
            assert ((inf["occurrence_change"] - inf["cif_count_lo"]) % 250) == switch - q    # CIFs to go
        if ann + 3 <= q < switch:
            assert sorted(nxt) == sorted(want_b), q                      # FIG 0/1 and 0/2 with C/N = 1 both seen
        assert inf["n_changes"] == (1 if q >= switch else 0)
        assert inf["last_change_fib"] == (3 * switch if q >= switch else -1)
        if q >= switch:
            assert nxt == []                                             # next->reset(), fib_decoder_fig0.cpp:107
        assert inf["n_restarts"] == 0


def test_without_the_swap_the_new_table_would_collide_with_the_old_one():
    """What round 3 did (current configuration only, change flags ignored): after the switch the new FIG 0/1 entries overlap the
    remembered ones and the whole collection restarts (fib_decoder_fig0.cpp:204-209) -- the table is found again, but late, and
    the announced switch CIF is never seen.  With change flags stuck at 1 (not 3) the reference does not swap either."""
    a, b = _layouts()
    fibs = _fib_stream(a, b, 80, 40, 40)                                 # no announcement at all: flags stay 0
    crc = np.ones(len(fibs), np.uint8)
    restarts = 0
    for i, d, o in _both(fibs, crc):
        assert d.info() == o.info() and _tab(d.subchannels()) == o.subchannels()
        restarts = d.info()["n_restarts"]
        assert d.info()["n_changes"] == 0
    assert restarts >= 1


def test_crc_failures_damaged_figs_and_garbage_follow_the_oracle():
    a, b = _layouts()
    rng = np.random.default_rng(3)
    fibs = _fib_stream(a, b, 120, 70, 40).copy()
    crc = (rng.random(len(fibs)) > 0.25).astype(np.uint8)                # a quarter of the FIBs fail their CRC
    for k in rng.choice(len(fibs), 40, replace=False):                   # ... and some that "pass" are damaged (as after a false CRC match)
        fibs[k, rng.integers(0, 30)] ^= 1 << rng.integers(0, 8)
    junk = rng.integers(0, 256, (30, 32)).astype(np.uint8)               # pure noise that passed its CRC
    fibs = np.concatenate([fibs[:150], junk, fibs[150:]])
    crc = np.concatenate([crc[:150], np.ones(30, np.uint8), crc[150:]])
    for step in (1, 3, 12):
        last = None
        for i, d, o in _both(fibs, crc, step):
            assert d.info() == o.info(), (step, i)
            assert _tab(d.subchannels()) == o.subchannels() and _tab(d.subchannels(next=True)) == o.subchannels(next=True), (step, i)
            last = (d.info(), _tab(d.subchannels()))
        if step == 1:
            first = last
        assert last == first                                             # chunking of the calls changes nothing


def test_one_shot_parse_is_the_running_decoder_on_a_fresh_state():
    a, b = _layouts()
    fibs = _fib_stream(a, b, 30, 20, 8)
    crc = np.ones(len(fibs), np.uint8)
    got, cif = dx.parse_fibs(fibs, crc)
    d = dx.FibDecoder()
    d.process(fibs, crc)
    assert _tab(got) == _tab(d.subchannels()) and cif == d.info()["cif_count"] == 29
    assert sorted(g.subch_id for g in got) == [0, 1, 2, 3, 4, 6]          # the configuration after the switch
    d.reset()
    assert d.info()["fibs_processed"] == 0 and d.subchannels() == []
    d.close()


def test_random_fig_streams_with_both_configurations_follow_the_oracle():
    """Differential run on random FIG streams: FIG 0/0 with random change flags (every transition, not only 3 -> 0), FIG 0/1 and 0/2 with
    random C/N flags, random sub-channel ids / start addresses / sizes (collisions and out-of-range entries restart the collection),
    short and long form, P/D = 1 services, truncated FIGs, unknown extensions and types.  After every FIB both decoders agree on
    every scalar and on both tables."""
    rng = np.random.default_rng(11)

    def rand_fig():
        kind = rng.integers(0, 10)
        cn = int(rng.integers(0, 2)) << 7
        if kind == 0:
            flags = int(rng.choice([0, 0, 3, 3, 1, 2]))
            body = bytes([0x00, 0x10, 0xF2, (flags << 6) | int(rng.integers(0, 20)), int(rng.integers(0, 250))]) + \
                (bytes([int(rng.integers(0, 250))]) if flags and rng.random() < 0.9 else b"")
        elif kind <= 4:
            body = bytearray([cn | 0x01])
            for _ in range(int(rng.integers(1, 5))):
                sid, start = int(rng.integers(0, 12)), int(rng.integers(0, 900))
                if rng.random() < 0.3:
                    body += bytes([(sid << 2) | (start >> 8), start & 0xFF, int(rng.integers(0, 64))])
                else:
                    w = (sid << 26) | (start << 16) | (1 << 15) | (int(rng.integers(0, 3)) << 12) | (int(rng.integers(0, 4)) << 10) | int(rng.integers(1, 120))
                    body += w.to_bytes(4, "big")
            body = bytes(body[: len(body) - (1 if rng.random() < 0.1 else 0)])       # sometimes truncated
        elif kind <= 7:
            pd = int(rng.random() < 0.2)
            body = bytearray([cn | (pd << 5) | 0x02])
            for _ in range(int(rng.integers(1, 3))):
                body += bytes(rng.integers(0, 256, 4 if pd else 2).tolist())
                ncomp = int(rng.integers(0, 3))
                body += bytes([ncomp])
                for _c in range(ncomp):
                    tmid = int(rng.choice([0, 0, 0, 1, 3]))
                    body += bytes([(tmid << 6) | int(rng.choice([63, 0, 5])), (int(rng.integers(0, 12)) << 2) | 2])
            body = bytes(body)
        elif kind == 8:
            body = bytes([cn | int(rng.integers(3, 32))]) + bytes(rng.integers(0, 256, int(rng.integers(0, 8))).tolist())
        else:
            return bytes([(int(rng.integers(1, 7)) << 5) | 3]) + bytes(rng.integers(0, 256, 3).tolist())   # another FIG type
        return bytes([len(body) & 0x1F]) + body

    fibs = []
    for _ in range(1500):
        data = b""
        while True:
            g = rand_fig()
            if len(data) + len(g) > 30:
                break
            data += g
        if len(data) < 30:
            data += b"\xFF" + b"\x00" * (29 - len(data))
        c = ds.crc16(data)
        fibs.append(np.frombuffer(data + bytes([c >> 8, c & 0xFF]), np.uint8))
    fibs = np.stack(fibs)
    crc = (rng.random(len(fibs)) > 0.1).astype(np.uint8)
    swaps = restarts = 0
    for i, d, o in _both(fibs, crc, 1):
        a, b = d.info(), o.info()
        assert a == b, (i, a, b)
        assert _tab(d.subchannels()) == o.subchannels() and _tab(d.subchannels(next=True)) == o.subchannels(next=True), i
        swaps, restarts = a["n_changes"], a["n_restarts"]
    assert swaps >= 5 and restarts >= 5                                   # the stream did exercise both mechanisms


def _fib_stream_flags(cur, nxt, n_cif, switch, announce_from, flags, cif_start=0, only01=False):
    """like _fib_stream with the announcement's change flags given; only01: the next configuration is announced with FIG 0/1 only
    (flags 1 = sub-channel organisation: a multiplexer need not repeat the unchanged service organisation with C/N = 1)"""
    out = []
    for q in range(n_cif):
        ann = announce_from <= q < switch
        fb = ds.build_fibs_reconf(cur if q < switch else nxt, nxt if ann else None, cif_start + q, flags if ann else 0, (cif_start + switch) % 250)
        if ann and only01 and (cif_start + q) % 4 == 3:        # the CIF that would carry the next FIG 0/2: the current one instead
            fb = ds.build_fibs_reconf(cur, None, cif_start + q, flags, (cif_start + switch) % 250)
        out.append(fb.reshape(3, 32))
    return np.concatenate(out)


def test_flags_1_reconfiguration_switches_too_and_the_one_after_it_starts_from_a_clean_next_table():
    """the reference swaps only after change flags 3; after flags 1 (sub-channel organisation only) its next table is never
    swapped or reset, so 'first description wins' keeps the stale entries for the reconfiguration after that.  libdabx's default:
    every announcement that ends switches; a table the announcement never filled is carried over; the next table is clean afterwards."""
    a, b = _layouts()
    c = b[:4] + [ds.SubCh(4, 600, 72, 96, 2, 0), ds.SubCh(6, 192, 24, 32, 2, 0, dab_plus=0)]         # second change: sub-channel 4 moves again
    s1 = _fib_stream_flags(a, b, 60, 40, 12, flags=1, only01=True)
    s2 = _fib_stream_flags(b, c, 60, 40, 12, flags=3, cif_start=60)
    fibs = np.concatenate([s1, s2]); crc = np.ones(len(fibs), np.uint8)
    want = lambda L: sorted((g.subch_id, g.cu_start, g.cu_size, g.kbps, g.prot_level) for g in L)
    got = lambda t: sorted(x[:5] for x in t)
    d = dx.FibDecoder()
    for q in range(120):
        d.process(fibs[3 * q:3 * q + 3], crc[3 * q:3 * q + 3])
        cur, nxt, inf = _tab(d.subchannels()), _tab(d.subchannels(next=True)), d.info()
        if 2 <= q < 40:
            assert got(cur) == want(a), q
        if 40 <= q < 100:
            assert got(cur) == want(b), q                      # switched after flags 1 ...
            # ... and the service components (never re-announced with C/N = 1) were carried over; sub-channel 6 is new: its FIG 0/2
            # comes with the current configuration's own FIGs a few CIFs later
            assert all(x[6] in (0, 1) for x in cur if x[0] != 6), q
            assert q < 44 or all(x[6] in (0, 1) for x in cur), q
        if 40 <= q < 72:
            assert nxt == [], q                                # next->reset()
        if 76 <= q < 100:
            assert got(nxt) == want(c), q                      # the second announcement fills a clean table: sub-channel 4 at its NEW address
        if q >= 100:
            assert got(cur) == want(c), q
        assert inf["n_changes"] == (0 if q < 40 else 1 if q < 100 else 2) and inf["n_restarts"] == 0, q
    d.close()
    # the reference's rule on the same stream: no switch after flags 1 -- the new FIG 0/1 entries then collide with the remembered
    # ones and the whole collection restarts (here that also wipes the stale next table; without a collision it would stay)
    r, o = dx.FibDecoder(reference_quirks=True), ol.OraFibDecoder()
    for q in range(120):
        r.process(fibs[3 * q:3 * q + 3], crc[3 * q:3 * q + 3]); o.process(fibs[3 * q:3 * q + 3], crc[3 * q:3 * q + 3])
        assert r.info() == o.info() and _tab(r.subchannels()) == o.subchannels() and _tab(r.subchannels(next=True)) == o.subchannels(next=True), q
        if q == 39:
            assert r.info()["n_changes"] == 0
    assert r.info()["n_restarts"] >= 1
    r.close(); o.close()
