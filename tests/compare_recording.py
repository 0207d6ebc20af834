#!/usr/bin/env python3
"""Side-by-side parity check of a RECORDED IQ file (.raw / .iq, .sdr / .wav, .uff):   python tests/compare_recording.py <file> [--subch all]

The file is replayed twice, independently:
  * through the GPU engine (dabstar_amd: dabx_probe_iq_file + dabx_feed_bytes -- the bytes cross PCIe, the sample map of
    raw_reader.cpp:66-70 / wav_reader.cpp:164 / xml_reader.cpp:254-398 runs on the device -- dabx_process, results through the bulk delivery);
  * through the oracle receiver (oracle/: the CPU restatement of the reference, with its own file-payload converter oracle/iqfile.c), on
    the same quantised samples the reference's readers would hand to DabProcessor (the last partial read block dropped like they drop it).
It prints what a maintainer with a real recording wants to know: FIB match %, per sub-channel the logical-frame, super-frame and AU-record
match, the first frame that differs with both sides' scalars -- and exits with 1 on ANY difference (0: bit-identical).

Why under tests/: the oracle is test infrastructure.  Nothing under dabstar_amd/ imports it; this command is a checker and lives with the
other checkers.  The sub-channels are discovered from the recording's own FIC (FIG 0/1 + 0/2) in a first pass of the engine; the oracle's
own FIG walk (oracle/fib.c) must find the same list."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
TF = 196608


class Sub:
    def __init__(self, d):
        self.subch_id, self.cu_start, self.cu_size, self.kbps = d.subch_id, d.cu_start, d.cu_size, d.kbps
        self.prot_level, self.short_form, self.dab_plus = d.prot_level, d.short_form, d.dab_plus


def payload_units(fmt):
    """Bytes per read block of the reference's reader for this family, and how many whole blocks the file holds (dabstar_amd.lib.play_file)."""
    unit = 32768 if fmt.family == 0 else 32768 * fmt.sample_bytes()
    if fmt.family == 2:
        unit = (fmt.sample_rate // 1000) * fmt.sample_bytes()
    return unit, fmt.data_bytes // unit


def engine_pass(dx, path, subch, max_frames, corrupt=False):
    """Replays the file; returns FIBs, CRC flags, frame records and, per sub-channel, every logical frame / super frame / AU record."""
    eng = dx.Engine(n_streams=1, ring_frames=12, max_subch=max(1, len(subch)), out_frames=8, fic_only=not subch)
    if subch:
        eng.set_subchannels(subch)
    eng.delivery_open(slots=4)
    fibs, crc, frm = [], [], []
    msc = [[] for _ in subch]
    sf = [[] for _ in subch]
    sfi = [[] for _ in subch]
    scal = []

    def drain(e):
        while True:
            ch = e.delivery_next(wait=False)
            if ch is None:
                break
            n = int(ch.streams[0]["n_frames"])
            assert ch.streams[0]["frames_lost"] == 0
            if n:
                fibs.append(ch.fibs[0, :n].copy()); crc.append(ch.crc[0, :n].copy()); frm.append(ch.frames[0, :n].copy())
            for j in range(len(subch)):
                q = ch.subch[0, j]
                assert q["cifs_lost"] == 0 and q["sf_lost"] == 0
                if q["n_cifs"]:
                    msc[j].append(ch.msc(0, j).copy())
                if q["n_sf"]:
                    sf[j].append(ch.superframes(0, j).copy()); sfi[j].append(ch.superframe_info(0, j).copy())
            ch.release()
        st = e.stats(0)
        scal.append((st["frames"], st["freq_offs_bb_hz"], st["clock_err_hz"], st["snr_db_est"], st["fic_ratio_percent"]))
        return st["frames"]
    # un-paced replay in the reference readers' read blocks (dabstar_amd.lib.play_file, with an early end for --max-frames)
    fmt = dx.probe_iq_file(path)
    unit, n_units = payload_units(fmt)
    feed = dx.Feed(eng, 0, fmt)
    per_block = max(1, (3 * TF * fmt.sample_bytes() * (fmt.sample_rate // 1000) // 2048) // unit)
    with open(path, "rb") as fh:
        fh.seek(fmt.data_offset)
        done = 0
        while done < n_units:
            take = min(per_block, n_units - done)
            feed.push(fh.read(take * unit))
            done += take
            left = 4 if not max_frames else max(0, min(4, max_frames - eng.stats(0)["frames"]))
            if left:
                eng.process(left)
            if drain(eng) >= max_frames > 0:
                break
    feed.close()
    found = eng.discover_subchannels(0) if not subch else None
    stats = [eng.subch_stats(0, j) for j in range(len(subch))]
    eng.delivery_close()
    eng.close()
    cat = lambda parts, shape: np.concatenate(parts) if parts else np.zeros(shape, np.uint8)   # noqa: E731
    out = {"fibs": cat(fibs, (0, 12, 32)), "crc": cat(crc, (0, 12)), "frames": np.concatenate(frm) if frm else np.zeros(0, dx.CHUNK_FRAME),
           "msc": [cat(msc[j], (0, 3 * c.kbps)) for j, c in enumerate(subch)],
           "sf": [cat(sf[j], (0, 110 * c.kbps // 8)) for j, c in enumerate(subch)],
           "sfi": [np.concatenate(sfi[j]) if sfi[j] else np.zeros(0, dx.SUPERFRAME_INFO) for j in range(len(subch))],
           "scal": scal, "stats": stats, "found": found}
    if corrupt and len(out["fibs"]):            # test hook (--self-test-corrupt): one bit of one delivered FIB flipped -> the tool must say so and exit 1
        out["fibs"][len(out["fibs"]) // 2, 5, 7] ^= 0x10
    return out


def oracle_pass(ol, fmt, path, subch, max_frames):
    unit, n_units = payload_units(fmt)
    with open(path, "rb") as fh:
        fh.seek(fmt.data_offset)
        payload = np.frombuffer(fh.read(n_units * unit), np.uint8)
    cap = payload.size + 4096
    x = np.zeros(cap, np.complex64)
    n = ol.oracle().ora_iq_convert(fmt.family, fmt.container, fmt.big_endian, fmt.swap_iq, fmt.bits, fmt.sample_rate, payload, payload.size,
                                   x.ctypes.data, cap)
    x = np.ascontiguousarray(x[:n])
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
    nf = L.ora_rx_run(rx, x, len(x), max_frames or 1000000)
    cap_ = L.ora_rx_get_capture(rx).contents
    g = lambda p, shape: np.ctypeslib.as_array(p, shape).copy() if nf else np.zeros(shape)   # noqa: E731
    out = {"n": nf, "samples": len(x), "fibs": g(cap_.fibs, (nf, 12, 32)), "crc": g(cap_.fib_crc, (nf, 12)), "start": g(cap_.start_idx, (nf,)),
           "sym0": g(cap_.sym0_pos, (nf,)), "fbb": g(cap_.fbb_end, (nf,)), "clock_err": g(cap_.clock_err, (nf,)), "snr": g(cap_.snr_db, (nf,)),
           "ratio": g(cap_.fic_ratio, (nf,)),
           "msc": [ol.backend_bytes(rx, j, "msc").reshape(-1, 3 * c.kbps) for j, c in enumerate(subch)],
           "sf": [ol.backend_bytes(rx, j, "sf").reshape(-1, 110 * c.kbps // 8) for j, c in enumerate(subch)],
           "sfi": [ol.backend_bytes(rx, j, "sfi") for j in range(len(subch))],
           "stats": [ol.backend_stats(rx, j) for j in range(len(subch))]}
    L.ora_rx_destroy(rx)
    return out


def oracle_discover(ol, fibs, crc):
    """The oracle's own FIG 0/1 + 0/2 walk over the FIBs of the first pass (oracle/fib.c)."""
    import ctypes as C
    L = ol.oracle()
    out = (ol.SubchDesc * 64)()
    dp = (C.c_int * 64)()
    cif = C.c_int(0)
    f = np.ascontiguousarray(fibs.reshape(-1, 32))
    c = np.ascontiguousarray(crc.reshape(-1))
    n = L.ora_parse_fibs(f, c, len(f), out, dp, 64, C.byref(cif))
    return [(out[i].subch_id, out[i].cu_start, out[i].cu_size, out[i].kbps, out[i].prot_level, out[i].short_form, dp[i]) for i in range(n)]


def first_diff(a, b):
    n = min(len(a), len(b))
    if n == 0:
        return None if len(a) == len(b) else 0
    ne = np.nonzero((a[:n].reshape(n, -1) != b[:n].reshape(n, -1)).any(axis=1))[0]
    if len(ne):
        return int(ne[0])
    return None if len(a) == len(b) else n


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("file")
    ap.add_argument("--subch", default="all", help="all (default: every sub-channel the FIC announces), none (FIC only), or SubChIds: 1,5,9")
    ap.add_argument("--max-frames", type=int, default=0, help="stop after this many frames (0 = the whole file)")
    ap.add_argument("--json", action="store_true", help="one JSON object instead of the table")
    ap.add_argument("--self-test-corrupt", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    from dabstar_amd import lib as dx
    import oracle_lib as ol
    if dx.load().dabx_device_count() <= 0:
        print("compare_recording: no MI355X (the engine has no CPU fallback)", file=sys.stderr)
        return 3
    fmt = dx.probe_iq_file(args.file)
    unit, n_units = payload_units(fmt)
    fam = ("raw (.raw/.iq)", "wav (.sdr/.wav)", "uff")[fmt.family]
    cont = ("u8", "s8", "i16", "i24", "i32", "f32")[fmt.container]
    head = {"file": args.file, "family": fam, "container": cont, "bits": fmt.bits, "sample_rate": fmt.sample_rate, "payload_bytes": int(fmt.data_bytes),
            "read_blocks": int(n_units), "dropped_tail_bytes": int(fmt.data_bytes - n_units * unit)}
    # ---- pass 1: FIC only -> the sub-channels the recording announces
    subch = []
    disc = {"engine": [], "oracle": [], "equal": True}
    if args.subch != "none":
        p1 = engine_pass(dx, args.file, [], 30)
        found = [Sub(d) for d in p1["found"]]
        disc["engine"] = [(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form, c.dab_plus) for c in found]
        disc["oracle"] = oracle_discover(ol, p1["fibs"], p1["crc"])
        disc["equal"] = sorted(disc["engine"]) == sorted(disc["oracle"])
        want = None if args.subch == "all" else {int(v) for v in args.subch.split(",")}
        subch = [c for c in found if want is None or c.subch_id in want]
        if want is not None and {c.subch_id for c in subch} != want:
            print("compare_recording: SubChIds %s not announced in the first 30 frames (found %s)" % (sorted(want - {c.subch_id for c in subch}), [c.subch_id for c in found]), file=sys.stderr)
            return 2
    # ---- pass 2: both receivers from the first sample, the sub-channels configured from the start
    e = engine_pass(dx, args.file, subch, args.max_frames, corrupt=args.self_test_corrupt)
    o = oracle_pass(ol, fmt, args.file, subch, args.max_frames)
    ne, no = len(e["fibs"]), o["n"]
    k = min(ne, no)
    ok = True
    same_fib = int(((e["fibs"][:k] == o["fibs"][:k]).all(axis=2) & (e["crc"][:k] == o["crc"][:k])).sum()) if k else 0
    # (the oracle also counts a last, partially read frame: one frame more on its side is not a difference)
    frames_ok = no - 1 <= ne <= no if not args.max_frames else k > 0
    total_fibs = 12 * k
    fib_pct = 100.0 * same_fib / max(1, total_fibs)
    start_same = bool(np.array_equal(e["frames"]["start_index"][:k], o["start"][:k])) if k else False
    bad_frame = first_diff(np.concatenate([e["fibs"][:k].reshape(k, -1), e["crc"][:k]], axis=1), np.concatenate([o["fibs"][:k].reshape(k, -1), o["crc"][:k]], axis=1)) if k else None
    reasons = []
    if not frames_ok:
        reasons.append("frame counts: engine %d, oracle %d" % (ne, no))
    if same_fib != total_fibs:
        reasons.append("%d of %d FIBs differ" % (total_fibs - same_fib, total_fibs))
    if not start_same:
        reasons.append("start indices differ")
    if not disc["equal"]:
        reasons.append("the two FIG walks announce different sub-channels")
    if k == 0:
        reasons.append("no frame decoded")
    ok = not reasons
    rows = []
    for j, c in enumerate(subch):
        # the engine's sequences must be the oracle's, from the first item on, and complete for the frames the engine demodulated: 4 logical
        # frames per frame after the 16-CIF de-interleaver fill (backend.cpp:146-150); the oracle may be one (partial last) frame ahead
        lf_e, lf_o = e["msc"][j], o["msc"][j]
        d_lf = first_diff(lf_e, lf_o[:len(lf_e)])
        n_sf = len(e["sf"][j])
        d_sf = first_diff(e["sf"][j], o["sf"][j][:n_sf])
        ri_e, ri_o = e["sfi"][j].view(np.uint8).reshape(-1, 32), o["sfi"][j].reshape(-1, 32)
        d_ri = first_diff(ri_e, ri_o[:len(ri_e)])
        # (the oracle decodes what it can of a last, partially read frame: up to one frame's worth of logical frames and one super frame ahead)
        enough = len(lf_e) == max(0, 4 * ne - 16) and len(ri_e) == n_sf and 0 <= len(lf_o) - len(lf_e) <= 4 and 0 <= len(o["sf"][j]) - n_sf <= 1
        same_extent = len(lf_o) == len(lf_e) and len(o["sf"][j]) == n_sf
        row = {"subch_id": c.subch_id, "kbps": c.kbps, "dab_plus": c.dab_plus, "logical_frames": len(lf_e), "logical_frames_oracle": len(o["msc"][j]),
               "first_different_logical_frame": d_lf, "super_frames": n_sf, "super_frames_oracle": len(o["sf"][j]), "first_different_super_frame": d_sf,
               "first_different_au_record": d_ri, "au_ok": e["stats"][j]["au_ok"], "au_bad": e["stats"][j]["au_bad"],
               "rs_corrected": e["stats"][j]["rs_corrected"], "sf_fail": e["stats"][j]["sf_fail"],
               "counters_equal": all(e["stats"][j][a] == o["stats"][j][b] for a, b in (("sf_ok", "sf_ok"), ("sf_fail", "sf_fail"), ("rs_corrected", "rs_corr"),
                                                                                       ("rs_failed", "rs_fail"), ("au_ok", "au_ok"), ("au_bad", "au_bad")))
               if same_extent else None}   # (counters are comparable only where both sides decoded exactly the same CIFs)
        row["match"] = d_lf is None and d_sf is None and d_ri is None and enough and row["counters_equal"] is not False
        if not row["match"]:
            reasons.append("sub-channel %d" % c.subch_id)
        ok = ok and row["match"]
        rows.append(row)
    res = {"recording": head, "subchannels_discovered": disc, "frames_engine": ne, "frames_oracle": no, "fib_match_pct": round(fib_pct, 4),
           "fibs_compared": total_fibs, "fib_crc_pass_pct": round(100.0 * float(e["crc"][:k].mean()) if k else 0.0, 3), "start_indices_equal": start_same,
           "first_different_frame": bad_frame, "subchannels": rows, "differences": reasons, "identical": bool(ok)}
    if bad_frame is not None and bad_frame < k:
        i = bad_frame
        sc = next((s for s in e["scal"] if s[0] > i), e["scal"][-1] if e["scal"] else None)
        res["first_different_frame_detail"] = {
            "frame": i, "fibs_different": [int(q) for q in np.nonzero((e["fibs"][i] != o["fibs"][i]).any(axis=1) | (e["crc"][i] != o["crc"][i]))[0]],
            "engine": {"start_index": int(e["frames"]["start_index"][i]), "sym0_pos": int(e["frames"]["sym0_pos"][i]), "crc": e["crc"][i].tolist(),
                       "scalars_after_block": None if sc is None else {"frames": sc[0], "f_bb_hz": round(sc[1], 3), "clock_err_hz": round(sc[2], 3), "snr_db": round(sc[3], 2), "fic_ratio_pct": sc[4]}},
            "oracle": {"start_index": int(o["start"][i]), "sym0_pos": int(o["sym0"][i]), "crc": o["crc"][i].tolist(), "f_bb_hz": round(float(o["fbb"][i]), 3),
                       "clock_err_hz": round(float(o["clock_err"][i]), 3), "snr_db": round(float(o["snr"][i]), 2), "fic_ratio_pct": int(o["ratio"][i]) * 10}}
    if args.json:
        print(json.dumps(res))
    else:
        h = res["recording"]
        print("%s: %s, %s %d bit, %d S/s, %d payload bytes = %d read blocks (%d tail bytes dropped like the reference's reader drops them)" % (
            h["file"], h["family"], h["container"], h["bits"], h["sample_rate"], h["payload_bytes"], h["read_blocks"], h["dropped_tail_bytes"]))
        print("sub-channels announced by the FIC: %d (engine) / %d (oracle's FIG walk)%s" % (len(disc["engine"]), len(disc["oracle"]), "" if disc["equal"] else "   ** DIFFERENT **"))
        print("frames: engine %d, oracle %d;  FIB match %.4f %% of %d (CRC pass %.3f %%);  start indices %s" % (
            ne, no, fib_pct, total_fibs, res["fib_crc_pass_pct"], "equal" if start_same else "** DIFFERENT **"))
        if rows:
            print("%-6s %-5s %-5s %-22s %-20s %-10s %-8s %-8s %-8s %s" % ("SubCh", "kbps", "DAB+", "logical frames (e/o)", "super frames (e/o)", "AU recs", "au_bad", "rs_corr", "sf_fail", "match"))
            for r in rows:
                print("%-6d %-5d %-5d %-22s %-20s %-10s %-8d %-8d %-8d %s" % (
                    r["subch_id"], r["kbps"], r["dab_plus"], "%d / %d%s" % (r["logical_frames"], r["logical_frames_oracle"], "" if r["first_different_logical_frame"] is None else "  diff@%d" % r["first_different_logical_frame"]),
                    "%d / %d%s" % (r["super_frames"], r["super_frames_oracle"], "" if r["first_different_super_frame"] is None else "  diff@%d" % r["first_different_super_frame"]),
                    "equal" if r["first_different_au_record"] is None else "diff@%d" % r["first_different_au_record"], r["au_bad"], r["rs_corrected"], r["sf_fail"], "yes" if r["match"] else "** NO **"))
        if "first_different_frame_detail" in res:
            print("first different frame:", json.dumps(res["first_different_frame_detail"]))
        print("RESULT: %s" % ("bit-identical" if ok else "DIFFERENT (%s)" % "; ".join(reasons)))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
