"""GPU differential test: randomly drawn channels and layouts, engine vs the CPU oracle receiver, stream by stream.

24 streams in ONE engine, each with its own SNR (3.5 .. 28 dB), carrier offset (up to +-36 kHz, i.e. also beyond the
+-35 kHz the reference follows), timing, level (-60 .. +30 dB), an echo, a drop-out, a sample-clock offset (+-90 ppm) and one
of three sub-channel layouts; every seventh stream goes through a time-variant multipath channel (channel_mobile).  DABX_FUZZ_SEED / DABX_FUZZ_CFG (threshold, strongest-peak sync, soft-bit generator) and
DABX_FUZZ_FAST (dabx_config.msc_fast_min_jobs / msc_class_min_jobs: the lane-per-trellis MSC path) -- knobs of THIS TEST, read
here, not by the library -- select other draws, receiver options and decoder kernels for hunting runs (tools/fuzz_hunt.py).
Whatever the reference's state machine does with such an input -- late lock, loss of lock, no lock at all -- the engine must
do the same: FIBs and CRC flags of every frame, the logical frames and the super frames of every sub-channel."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402

pytestmark = pytest.mark.gpu

N_CASES, N_FRAMES = 24, 22


def _layouts():
    uep = lambda k, l: (ol.ora_uep_map(k, l)[1] >= 0).astype(np.uint8)                   # noqa: E731
    mixed = [ds.SubCh(3, 0, 24, 32, 3, 1, mask=uep(32, 3), dab_plus=0), ds.SubCh(7, 30, 48, 32, 0, 0),
             ds.SubCh(12, 80, 128, 128, 1, 0), ds.SubCh(20, 210, 54, 96, 6, 0), ds.SubCh(21, 270, 24, 48, 3, 0),
             ds.SubCh(33, 300, 48, 64, 2, 0), ds.SubCh(40, 350, 54, 64, 4, 0), ds.SubCh(63, 410, 116, 128, 2, 1, mask=uep(128, 2))]
    full = ds.default_subchannels(18, 64)
    return [full, mixed, [full[1], full[8], full[17]]]


def _oracle(x, subch, cfg, tie=0, set_mode=True):
    """tie = 1 / 2: the oracle receiver decodes with its restatement of the VITERBI_AVX2 / VITERBI_SSE2 build's arithmetic
    (a process-wide switch: set_mode = False when the caller has set it for a pool of threads, one receiver each)"""
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
    L.ora_rx_configure(rx, *cfg)
    if set_mode:
        L.ora_set_viterbi_mode(tie)
    try:
        n = L.ora_rx_run(rx, x, len(x), 10000)
    finally:
        if set_mode:
            L.ora_set_viterbi_mode(0)
    cap = L.ora_rx_get_capture(rx).contents
    res = dict(n=n, fibs=np.ctypeslib.as_array(cap.fibs, (n, 12, 32)).copy() if n else np.zeros((0, 12, 32), np.uint8),
               crc=np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy() if n else np.zeros((0, 12), np.uint8),
               start=np.ctypeslib.as_array(cap.start_idx, (n,)).copy() if n else np.zeros(0, np.int32),
               sym0=np.ctypeslib.as_array(cap.sym0_pos, (n,)).copy() if n else np.zeros(0, np.int32),
               fic_ovf=np.ctypeslib.as_array(cap.fic_overflow, (n,)).copy() if n else np.zeros(0, np.int32),
               msc_ovf=np.ctypeslib.as_array(cap.msc_overflow, (n,)).copy() if n else np.zeros(0, np.int32),
               msc=[ol.backend_bytes(rx, i, "msc") for i in range(len(subch))],
               sf=[ol.backend_bytes(rx, i, "sf") for i in range(len(subch))],
               stats=[ol.backend_stats(rx, i) for i in range(len(subch))])
    L.ora_rx_destroy(rx)
    return res


def draw_streams(seed, only=None):
    """The N_CASES random streams of one seed: (layouts, cases, xs).  only = i: the draw stops after stream i (tools/debug_fuzz_case.py)."""
    layouts = _layouts()
    base = [ds.build_ensemble(10, lay, seed=500 + i) for i, lay in enumerate(layouts)]
    # what was transmitted, per layout and sub-channel: the logical frames and the super frames as byte strings
    draw_streams.tx_frames = [[{r.tobytes() for r in b.msc_bytes[j]} for j in range(len(b.subch))] for b in base]
    draw_streams.tx_superframes = [[{r.tobytes() for r in b.superframes[j]} if b.subch[j].dab_plus else set() for j in range(len(b.subch))] for b in base]
    rng = np.random.default_rng(seed)
    cases, xs = [], []
    for i in range(N_CASES):
        li = int(rng.integers(0, 3))
        snr = float(rng.uniform(3.5, 28.0))
        cfo = float(rng.uniform(-36000.0, 36000.0)) if i % 3 == 0 else float(rng.uniform(-2500.0, 2500.0))
        toff = int(rng.integers(0, ds.TF))
        gain = float(10 ** rng.uniform(-3.0, 1.5)) * 0.25
        if i % 7 == 4:                                    # a moving receiver: Rayleigh taps with Jakes Doppler, 5 .. 80 Hz, drifting sample clock
            snr = max(snr, 8.0)
            prof = ["TU6", "RA4", "SFN2", "HT6"][int(rng.integers(0, 4))]
            x = ds.channel_mobile(base[li].iq, prof, doppler_hz=float(rng.uniform(5.0, 80.0)), snr_db=snr, cfo_hz=cfo, timing_offset=toff,
                                  gain=gain, seed=700 + i, n_out=(N_FRAMES + 2) * ds.TF, clock_ppm=float(rng.uniform(-30.0, 30.0)),
                                  clock_drift_ppm_per_s=float(rng.uniform(-5.0, 5.0)))
        else:
            x = ds.channel(base[li].iq, snr_db=snr, cfo_hz=cfo, timing_offset=toff, gain=gain, seed=700 + i, n_out=(N_FRAMES + 2) * ds.TF)
        if i % 4 == 1:                                    # an echo inside the guard interval
            d = int(rng.integers(5, 400))
            x[d:] += np.complex64(rng.uniform(0.2, 0.8) * np.exp(1j * rng.uniform(0, 6.28))) * x[:-d].copy()
        if i % 5 == 2:                                    # a drop-out of 0.3 .. 2.5 frames somewhere after lock
            a = int(rng.uniform(7, 12) * ds.TF)
            x[a:a + int(rng.uniform(0.3, 2.5) * ds.TF)] *= np.float32(1e-3)
        if i % 6 == 3:                                    # sample-clock offset up to +-90 ppm (linear interpolation)
            ppm = float(os.environ.get("DABX_FUZZ_PPM", "90")) * 1e-6                            # 90 ppm of 4.7 M = 425 samples
            t = np.arange(len(x) - 1000, dtype=np.float64) * (1.0 + rng.uniform(-ppm, ppm))
            i0 = np.floor(t).astype(np.int64)
            fr = (t - i0).astype(np.float32)
            x = np.concatenate([(x[i0] * (1 - fr) + x[i0 + 1] * fr).astype(np.complex64), x[-1000:]])
        xs.append(np.ascontiguousarray(x, np.complex64))
        cases.append((li, snr, cfo, toff, gain))
        if only is not None and i == only:
            break
    return layouts, cases, xs, rng


def _rerun_exact_level(x, subch, cfg, tie=0, level_mode=1):
    """One stream alone on an engine with cfg.exact_level_tracker = level_mode: FIBs, CRC flags and the walk of all its frames."""
    thr, strongest, soft_type = cfg
    eng = dx.Engine(n_streams=1, ring_frames=N_FRAMES + 3, max_subch=18, out_frames=12, sync_threshold=thr, sync_strongest=bool(strongest),
                    soft_bit_type=soft_type, exact_level_tracker=level_mode, viterbi_tie_mode=tie)
    eng.set_subchannels(subch)
    eng.push_iq(0, x)
    fibs, crcs, walk = [], [], []
    seen = 0
    for _ in range(N_FRAMES + 40):
        eng.process(1)
        f = eng.stats(0)["frames"]
        if f > seen:
            a, b = eng.read_fibs(0, f - seen)
            pos, sti = eng.read_frame_info(0, f - seen)
            fibs.extend(a); crcs.extend(b); walk.extend(zip(pos.tolist(), sti.tolist()))
            seen = f
    eng.close()
    return fibs, crcs, walk


def test_the_search_sees_the_samples_behind_the_oscillator():
    """The time syncer gets its samples from the sample reader, i.e. multiplied by oscillatorTable[currentPhase] also while the frequency
    offset handed in is 0 (sample_reader.cpp:274-281): after a frame that was read with a carrier offset the phase stands somewhere, and
    the magnitude of the product differs from the magnitude of the sample in its last bit now and then.  Hunt 11000-11059 found the stream
    on which that decides where a null symbol ends (seed 11033, stream 20: lock on a false peak, loss, second search one sample late,
    start index 458 instead of 459): with the exact level tracker the walk must be the oracle's, frame by frame."""
    layouts, cases, xs, _rng = draw_streams(11033, only=20)
    li = cases[20][0]
    cfg = (3.0, 1, 1)
    ora = _oracle(xs[20], layouts[li], cfg)
    fibs, crcs, walk = _rerun_exact_level(xs[20], layouts[li], cfg)
    n = min(len(fibs), ora["n"])
    assert n >= 20 and abs(len(fibs) - ora["n"]) <= 1
    assert ora["start"][:3].tolist() == [295, 459, 504]                    # the case is the one described above
    assert [w[0] for w in walk[:n]] == ora["sym0"][:n].tolist() and [w[1] for w in walk[:n]] == ora["start"][:n].tolist()
    assert np.array_equal(np.array(crcs)[:n], ora["crc"][:n])
    ok = ora["crc"][:n].astype(bool)
    assert ok.sum() >= 12 * 18 and np.array_equal(np.array(fibs)[:n][ok], ora["fibs"][:n][ok])


@pytest.mark.parametrize("seed,stream", [(11036, 18), (11054, 18)])
def test_streams_that_needed_the_exact_tracker_follow_the_oracle_by_default(seed, stream):
    """Hunt 11000-11059 (round 4, before the level anchor): two streams of 1 440 re-locked on a different sample than the oracle because the
    level the search resumed with was the chunk-wise one, and followed it only with cfg.exact_level_tracker = 1.  With the default mode
    re-walking the level from its anchor when a lock is lost, they follow it as they are; mode 2 (chunk-wise only) still shows the difference."""
    layouts, cases, xs, _rng = draw_streams(seed, only=stream)
    subch, cfg = layouts[cases[stream][0]], (3.0, 0, 1)
    ora = _oracle(xs[stream], subch, cfg)
    fibs, crcs, walk = _rerun_exact_level(xs[stream], subch, cfg, level_mode=0)
    n = min(len(fibs), ora["n"])
    assert n >= 3 and abs(len(fibs) - ora["n"]) <= 1
    assert [w[0] for w in walk[:n]] == ora["sym0"][:n].tolist() and [w[1] for w in walk[:n]] == ora["start"][:n].tolist()
    _f2, _c2, walk2 = _rerun_exact_level(xs[stream], subch, cfg, level_mode=2)
    m = min(len(walk2), ora["n"])
    assert [w[1] for w in walk2[:m]] != ora["start"][:m].tolist() or [w[0] for w in walk2[:m]] != ora["sym0"][:m].tolist() or len(walk2) != len(walk)


# The committed draws: three seeds, each with its own receiver options, all under the strict rules (no stream may need the exact
# level tracker, no logical frame may differ where the oracle delivers the transmitted one).  DABX_FUZZ_SEED (tools/fuzz_hunt.py)
# replaces them by one hunting draw under the tolerant rules.
# Round 4: three more strict draws cover what only hunting runs covered before -- the lane-per-trellis decoder classes
# (msc_fast_min_jobs = 64, msc_class_min_jobs = 1: every profile class of the draw goes through k_msc_vitT) and the arithmetic of the
# reference's VITERBI_AVX2 / VITERBI_SSE2 builds (viterbi_tie_mode 1 / 2; the oracle receiver decodes with ora_viterbi_simd / _sse2).
# (seed, receiver options, lane-per-trellis classes, viterbi_tie_mode)
# The seventh draw runs the search for the null symbol on its own HIP stream next to the steps (acquire_mode = 2; 24 streams that lose
# and find their lock at different times, the frame chain and k_acquire handing them to each other): same walk, same bytes.
COMMITTED = [(20260101, "3.0,0,1", 0, 0), (7003, "4.0,1,2", 0, 0), (9003, "2.5,0,3", 0, 0),
             (9107, "3.0,0,1", 1, 0), (9211, "3.0,0,2", 1, 1), (9313, "3.5,1,1", 1, 2), (9419, "3.0,0,1", 2, 0)]
OVF_FRAMES_SEEN = {20260101: 4, 7003: 25, 9003: 8, 9107: 12, 9211: 32, 9313: 12, 9419: 30}   # frames with FIC soft-bit overflow in each committed draw (excluded from the FIB comparison)


@pytest.mark.parametrize("seed,cfg,fast,tie", COMMITTED if "DABX_FUZZ_SEED" not in os.environ else [
    (int(os.environ["DABX_FUZZ_SEED"]), os.environ.get("DABX_FUZZ_CFG", "3.0,0,1"), int(os.environ.get("DABX_FUZZ_FAST", "0")), int(os.environ.get("DABX_FUZZ_TIE", "0")))])
def test_random_channels_and_layouts_follow_the_oracle(seed, cfg, fast, tie):
    # receiver options (sync threshold, strongest-peak sync, soft-bit generator 1..3): DABX_FUZZ_CFG="4.0,1,2"
    thr, strongest, soft_type = [t(v) for t, v in zip((float, int, int), cfg.split(","))]
    layouts, cases, xs, rng = draw_streams(seed)

    fast = dict(acquire_mode=2) if fast == 2 else (dict(msc_fast_min_jobs=64, msc_class_min_jobs=1) if fast else {})
    # DABX_FUZZ_LEVEL (hunting runs): cfg.exact_level_tracker of the engine under test -- 1: the exact tracker in lock (k_level_exact), 2: chunk-wise only
    eng = dx.Engine(n_streams=N_CASES, ring_frames=N_FRAMES + 3, max_subch=18, out_frames=12, sync_threshold=thr,
                    sync_strongest=bool(strongest), soft_bit_type=soft_type, viterbi_tie_mode=tie,
                    exact_level_tracker=int(os.environ.get("DABX_FUZZ_LEVEL", "0")), **fast)
    for s, (li, *_rest) in enumerate(cases):
        eng.set_subchannels(layouts[li], stream=s)
        eng.push_iq(s, xs[s])
    fibs = [[] for _ in range(N_CASES)]
    crcs = [[] for _ in range(N_CASES)]
    walk = [[] for _ in range(N_CASES)]                  # (position of symbol 0, start index) of every frame: the receiver's walk
    frames_seen = [0] * N_CASES
    steps = 0
    while steps < N_FRAMES + 40:                          # failed acquisition attempts cost steps too
        m = int(rng.integers(1, 10)) if steps >= 3 else 1 # frames per dabx_process call: 1..9 (MSC batches of 7 + remainder)
        eng.process(m)
        steps += m
        for s in range(N_CASES):
            f = eng.stats(s)["frames"]
            new = f - frames_seen[s]
            if new:
                frames_seen[s] = f
                a, b = eng.read_fibs(s, new)
                assert len(a) == new
                fibs[s].extend(a); crcs[s].extend(b)
                pos, sti = eng.read_frame_info(s, new)
                walk[s].extend(zip(pos.tolist(), sti.tolist()))

    if fast.get("acquire_mode", 0) != 1:            # (0, the default, searches next to the steps too once half of the streams are in lock)
        # The search runs next to the steps and a pass is launched only when the previous one has finished: of a call that queues
        # several steps at once only the first starts one.  A stream that flaps on false peaks (lock, one junk frame, loss, search, ...)
        # therefore gets through one such cycle per CALL, not per step -- it holds nobody up, which is the point, but needs more calls
        # to get through its samples: drain with single steps until nothing moves any more.
        idle, last = 0, None
        for _ in range(400):
            eng.process(1)
            steps += 1
            now = [(eng.stats(s)["frames"], eng.stats(s)["samples_consumed"]) for s in range(N_CASES)]
            idle = idle + 1 if now == last else 0
            last = now
            for s in range(N_CASES):
                new = now[s][0] - frames_seen[s]
                if new:
                    frames_seen[s] = now[s][0]
                    a, b = eng.read_fibs(s, new)
                    fibs[s].extend(a); crcs[s].extend(b)
                    pos, sti = eng.read_frame_info(s, new)
                    walk[s].extend(zip(pos.tolist(), sti.tolist()))
            if idle >= 4:
                break
    locked = n_bad = n_bad_diff = compared = eti_checked = n_ovf_frames = 0
    msc_frames = msc_oracle_wrong = msc_wrong_differ = msc_engine_wrong = 0
    msc_events = []
    level_approx_streams = []
    # the 24 oracle receivers, each on a thread of its own (ctypes releases the GIL; the oracle keeps no state outside a receiver except
    # the Viterbi arithmetic switch, set once here)
    from concurrent.futures import ThreadPoolExecutor
    ol.oracle().ora_set_viterbi_mode(tie)
    try:
        run = lambda q: _oracle(xs[q], layouts[cases[q][0]], (thr, strongest, soft_type), tie, set_mode=False)   # noqa: E731
        first = run(0)                                     # (its lazily built tables are complete before the pool starts)
        with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as pool:
            oras = [first] + list(pool.map(run, range(1, N_CASES)))
    finally:
        ol.oracle().ora_set_viterbi_mode(0)
    for s, (li, snr, cfo, toff, gain) in enumerate(cases):
        tag = (s, li, round(snr, 1), round(cfo), toff, gain)
        subch = layouts[li]
        ora = oras[s]
        n = len(fibs[s])
        assert abs(n - ora["n"]) <= 1, (tag, n, ora["n"])           # the oracle also counts a last, partially read frame
        n = min(n, ora["n"])
        if n == 0:
            continue
        # Frames in which soft values of the FIC symbols left the int16 range (the level returning after a drop-out before the
        # demapper's means have followed) are outside the comparison: the reference's `(i16)` cast is undefined behaviour
        # there, the x86 wrap-around both sides reproduce turns a last-ulp float difference into a full-scale one, and the
        # decoder's answer to such symbols is arbitrary (fuzz seed 3003: one FIB of one frame, docs/history/r01-r04_design_notebook.md 4).
        # the walk through the samples: every frame found at the same sample with the same start index, whatever the state
        # machine did in between (failed correlations, false dips, losses of lock) -- also for streams that never decode a FIB
        same_walk = [w[0] for w in walk[s][:n]] == ora["sym0"][:n].tolist() and [w[1] for w in walk[s][:n]] == ora["start"][:n].tolist()
        if not same_walk:
            # The one deliberate approximation of the state machine (docs/history/r01-r04_design_notebook.md 4): in lock the level tracker advances chunk by
            # chunk, not sample by sample, which moves s_level by ~1e-5 relative -- the size of the float noise of the
            # reference's own 196 608-step recurrence.  On a stream that only ever locks on false peaks the null-dip detector
            # of a later attempt can then fall on the other side of its threshold (hunt 5000-5047: seed 5030, stream 18 of
            # 1 152, a carrier offset outside the +-35 kHz range, no FIB ever decoded).  Such a stream must follow the oracle
            # with cfg.exact_level_tracker -- the approximation is then the proven cause -- and stays the exception.
            # Since the end of round 4 the default re-walks the level from its anchor when a lock is lost (every sample of a stream is
            # pushed at once here, so the anchor is always in the ring): no stream of 156 hunting draws has come this way since, and
            # none may -- the re-run only serves to say WHY, should one ever do.
            level_approx_streams.append(tag)
            fibs[s], crcs[s], walk[s] = _rerun_exact_level(xs[s], subch, (thr, strongest, soft_type), tie)
            n = min(len(fibs[s]), ora["n"])
            assert abs(len(fibs[s]) - ora["n"]) <= 1, (tag, len(fibs[s]), ora["n"])
            assert [w[0] for w in walk[s][:n]] == ora["sym0"][:n].tolist() and [w[1] for w in walk[s][:n]] == ora["start"][:n].tolist(), tag
        clean = ora["fic_ovf"][:n] == 0
        n_ovf_frames += int((~clean).sum())
        assert np.array_equal(np.array(crcs[s])[:n][clean], ora["crc"][:n][clean]), tag
        ef, of, okm = np.array(fibs[s])[:n], ora["fibs"][:n], ora["crc"][:n].astype(bool)
        okm &= clean[:, None]
        assert np.array_equal(ef[okm], of[okm]), tag                  # every FIB that passes its CRC: identical bytes
        # FIBs that fail the CRC are the decoder's answer to noise; with the float demapper equal only within tolerance
        # (docs/history/r01-r04_design_notebook.md 4) a few of them may differ -- they must stay rare and confined to frames without a good FIB
        bad_diff = (ef != of).any(axis=2) & ~okm & clean[:, None]
        n_bad_diff += int(bad_diff.sum()); n_bad += int((~okm & clean[:, None]).sum())
        if os.environ.get("DABX_FUZZ_VERBOSE") and bad_diff.any():
            print("garbage FIBs differ:", tag, "frames", np.nonzero(bad_diff.any(axis=1))[0].tolist(), "of", n, "crc ok per frame", okm.sum(axis=1).tolist())
        locked += int(ora["crc"][:n].sum() > 12 * n // 2)
        # MSC bytes are compared where the signal is decodable.  Below ~6 dB the EEP 3-A sub-channels decode with residual
        # errors, and a soft bit that differs by one LSB (float demapper, docs/history/r01-r04_design_notebook.md 4) can tip a survivor path: seen once in
        # 2 600 streams, at 3.7 dB, identically on both MSC decoder kernels.  Above that the same happens inside the fades of
        # the mobile channels and under echoes (hunt 8000-8095: a stream at 8 dB in a fading channel, soft-bit generator 3, where
        # the oracle itself got 271 of its logical frames wrong and the engine's wrong frames differ from the oracle's wrong frames
        # by a few bytes; and one frame of a 7-dB stream under an echo that the oracle got right and the engine did not -- it can
        # as well be the other way round).  So the rule is per logical frame: a frame / super frame in which the ORACLE delivers
        # what was transmitted (its decoder was inside its capability) must be byte-identical; a frame the oracle got wrong is
        # the decoder's answer to noise and may differ (counted).  The exception -- frames the oracle got right and the engine
        # did not, in ONE stream below 10 dB -- is tolerated once per hunting draw and not at all with the committed seed.
        if n < 7 or not okm[n - 7:].all() or snr < 6.0 or ora["msc_ovf"][max(0, n - 9):n].any() or not same_walk:
            continue                                       # the newest 16 logical
        eng.subch = list(subch)                            # frames reach back 32 CIFs = 8 frames, of which the last 7 are clean here
        compared += 1
        msc_stream_differs = False
        for j, c in enumerate(subch):
            st = eng.subch_stats(s, j)
            k, nb = st["cifs_decoded"], 3 * c.kbps
            o = ora["msc"][j].reshape(-1, nb)
            assert k <= len(o), (tag, j)
            m = min(16, k)
            if m:
                got, want = eng.read_msc(s, j, m), o[k - m:k]
                sent = np.array([r.tobytes() in draw_streams.tx_frames[li][j] for r in want])
                differ = (got != want).any(axis=1)
                msc_frames += m; msc_oracle_wrong += int((~sent).sum()); msc_wrong_differ += int((differ & ~sent).sum())
                msc_engine_wrong += sum(r.tobytes() not in draw_streams.tx_frames[li][j] for r in got)
                if differ.any():
                    msc_stream_differs = True
                if (differ & sent).any():
                    msc_events.append(dict(stream=tag, subch=j, frames=np.nonzero(differ & sent)[0].tolist(), bytes_differing=(got != want).sum(axis=1)[differ & sent].tolist()))
            if c.dab_plus and st["sf_ok"]:
                o_sf = ora["sf"][j].reshape(-1, 110 * c.kbps // 8)
                q = min(4, st["sf_ok"])
                got, want = eng.read_superframes(s, j, q), o_sf[st["sf_ok"] - q:st["sf_ok"]]       # reach back 20 logical frames
                sent = np.array([r.tobytes() in draw_streams.tx_superframes[li][j] for r in want])
                differ = (got != want).any(axis=1)
                if differ.any():
                    msc_stream_differs = True
                if (differ & sent).any() and not (msc_events and msc_events[-1]["stream"] == tag):
                    msc_events.append(dict(stream=tag, subch=j, superframes=np.nonzero(differ & sent)[0].tolist()))
        # ETI-NI frames of the newest CIFs (eti_generator.cpp:169-308): header with the FIG 0/0 counter, stream characterisation of
        # the layout (UEP / EEP-A / EEP-B TPL fields), FIC, MST, CRCs -- against the oracle's assembly of the oracle's bytes
        kmin = min(eng.subch_stats(s, j)["cifs_decoded"] for j in range(len(subch)))
        frames_eti, _lost = eng.read_eti(s, 32)                    # everything the rings still hold: the last one is the newest CIF
        if same_walk and not msc_stream_differs and kmin >= 16 and len(frames_eti) >= 4 and all(eng.subch_stats(s, j)["cifs_decoded"] == kmin for j in range(len(subch))):
            import test_eti as te
            descs = [dx.SubchDesc(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form, int(c.dab_plus), 0) for c in subch]
            Lf = len(frames_eti)
            for i in range(Lf - 4, Lf):
                r = 16 + (kmin - Lf + i)
                F, q = divmod(r, 4)
                fib = of[F].reshape(-1)
                hi, lo = int(fib[4] & 0x1F), int(fib[5])              # FIG 0/0 leads FIB 0 of every CIF of the synthetic ensembles;
                for g in range(4):                                    # the counter state after a frame is that of its last good group
                    if okm[F][3 * g]:
                        hi, lo = int(of[F][3 * g][4] & 0x1F), int(of[F][3 * g][5])
                msc_r = [ora["msc"][j].reshape(-1, 3 * c.kbps)[r - 16] for j, c in enumerate(subch)]
                want, _ = te._ora_frame(hi, lo, q, descs, fib[96 * q:96 * q + 96], msc_r)
                assert np.array_equal(frames_eti[i], want), (tag, i, r)
            eti_checked += 1
    if level_approx_streams and os.environ.get("DABX_FUZZ_VERBOSE"):
        print("walk differs with the chunk-wise level tracker, equal with the exact one:", level_approx_streams)
    assert not level_approx_streams, level_approx_streams
    # logical frames the oracle got right and the engine did not (see above): none with the committed seed, one stream below 10 dB in a hunting draw
    if os.environ.get("DABX_FUZZ_VERBOSE"):
        print("logical frames compared:", msc_frames, "of which the oracle got wrong:", msc_oracle_wrong, "(the engine:", msc_engine_wrong, ") of which differ:", msc_wrong_differ, "events:", msc_events)
    ev_streams = {ev["stream"] for ev in msc_events}
    assert len(ev_streams) <= (0 if "DABX_FUZZ_SEED" not in os.environ else 1) and all(t[2] < 10.0 for t in ev_streams), msc_events
    assert msc_oracle_wrong <= 0.15 * msc_frames, (msc_oracle_wrong, msc_frames)         # the comparison must not become vacuous
    # ... and the frames that MAY differ (the oracle got them wrong: the decoder's answer to noise) must not hide a systematic difference:
    # engine and oracle get about the same number of frames wrong (hunts: 2 008 vs 2 009 of 251 616; 1 369 vs 1 369 of 252 528), and
    # only a small part of the wrong ones differ at all (65 of 2 008; 1 of 1 369)
    assert abs(msc_engine_wrong - msc_oracle_wrong) <= max(2, 0.005 * msc_frames), (msc_engine_wrong, msc_oracle_wrong, msc_frames)
    assert msc_wrong_differ <= max(3, 0.25 * msc_oracle_wrong), (msc_wrong_differ, msc_oracle_wrong)
    assert eti_checked >= N_CASES // 4
    assert compared >= N_CASES // 3 and locked >= N_CASES // 2                                    # most of the draws do lock and decode
    # FIBs that fail their CRC on both sides: a soft bit that differs by one LSB (2-4 in 10^5, docs/history/r01-r04_design_notebook.md 4) anywhere in a FIC block
    # that is noise anyway changes its junk.  A few per thousand failing FIBs (the fading streams produce hundreds of them);
    # generator 3 (no normalisation) is the touchiest
    assert n_bad_diff <= max(2 if soft_type != 3 else 6, 0.03 * n_bad), (n_bad_diff, n_bad)
    # overflow frames (excluded above) stay the exception: 4 of 528 frames with the first committed seed (25 with the second, whose
    # drop-outs are longer); a regression that produced
    # spurious overflows, or excluded frames wholesale, trips this bound (hunting seeds draw other drop-outs: looser there)
    assert n_ovf_frames <= (OVF_FRAMES_SEEN[seed] + 4 if "DABX_FUZZ_SEED" not in os.environ else 2 * N_CASES), n_ovf_frames
    eng.close()


@pytest.mark.parametrize("fast", [0, 1])
def test_random_service_start_stop_schedules(fast):
    """MscHandler::set_channel / stop_service at random times (msc_handler.cpp:95-146): 5 streams of one engine, each with
    its own channel; between dabx_process calls a random stream gets a new random subset of the 18 services.  A service that
    keeps running is never disturbed, one that (re)starts at CIF c delivers exactly the oracle's logical frames c, c+1, ...
    (its 16-CIF de-interleaver fill starts at c).  fast = 1 routes the MSC through the lane-per-trellis classes, which are
    rebuilt after every change."""
    kw = dict(msc_fast_min_jobs=64, msc_class_min_jobs=1) if fast else {}
    rng = np.random.default_rng(int(os.environ.get("DABX_FUZZ_SEED", "77")) + fast)
    subch = ds.default_subchannels(18, 64)
    n_streams, n_frames = 5, 30
    ens = ds.build_ensemble(10, subch, seed=90)
    xs = [ds.channel(ens.iq, snr_db=16.0 + 2 * s, cfo_hz=300.0 * (s - 2), timing_offset=30011 * s + 9, seed=900 + s,
                     n_out=(n_frames + 3) * ds.TF) for s in range(n_streams)]
    oras = [_oracle(x, subch, (3.0, 0, 1)) for x in xs]
    mk = lambda c: dx.SubchDesc(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form, 1, 0)   # noqa: E731
    empty = dx.SubchDesc(0, 0, 0, 0, 0, 0, 0, 0)
    eng = dx.Engine(n_streams=n_streams, ring_frames=n_frames + 4, max_subch=18, out_frames=4, **kw)
    active = [set(int(j) for j in rng.choice(18, 6, replace=False)) for _ in range(n_streams)]
    for s in range(n_streams):
        eng.set_subchannels([mk(subch[j]) if j in active[s] else empty for j in range(18)], stream=s)
        eng.push_iq(s, xs[s])
    done = 0
    while done < n_frames:
        m = int(rng.integers(1, 6))
        eng.process(m)
        done += m
        for _ in range(int(rng.integers(0, 3))):                      # 0..2 changes between calls
            s = int(rng.integers(0, n_streams))
            keep = {j for j in active[s] if rng.random() < 0.7}
            new = set(int(j) for j in rng.choice(18, int(rng.integers(0, 5)), replace=False))
            active[s] = keep | new
            eng.set_subchannels([mk(subch[j]) if j in active[s] else empty for j in range(18)], stream=s)
    checked = 0
    for s in range(n_streams):
        st = eng.stats(s)
        assert st["frames"] >= n_frames - 2 and st["frames"] * 4 == oras[s]["crc"][:st["frames"]].shape[0] * 4
        eng.subch = list(subch)
        for j in range(18):
            sub = eng.subch_stats(s, j)
            assert bool(sub["active"]) == (j in active[s]), (s, j)
            if not sub["active"] or sub["cifs_decoded"] == 0:
                continue
            c0, k = sub["start_cif"], sub["cifs_decoded"]
            assert k == st["frames"] * 4 - c0 - 16, (s, j, c0, k)
            o = oras[s]["msc"][j].reshape(-1, 192)
            m = min(16, k)
            assert np.array_equal(eng.read_msc(s, j, m), o[c0 + k - m:c0 + k]), (s, j, c0, k)
            checked += 1
    assert checked >= 10
    eng.close()
