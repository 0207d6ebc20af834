"""Generates tests/golden/*.npz from the GENUINE reference objects (oracle/_ref/libdabref.so, built by
oracle/ref/Makefile from /root/reference).  Run in the build container only; the fixtures are data (inputs and the
reference's outputs) and travel with the repo, the reference does not.

    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def viterbi_big_inputs(n):
    """Seeded soft-bit rows for the long trellises: random, small (many ties), saturating mix, hard +-127."""
    rng = np.random.default_rng(7000 + n)
    m = 4 * (n + 6)
    return np.stack([rng.integers(-200, 201, m), rng.integers(-3, 4, m), rng.choice([-32768, 32767, 300, -300, 0, 127, -127], m),
                     rng.choice([-127, 127], m)]).astype(np.int16)


def main():
    R = ol.ref()
    rng = np.random.default_rng(20261002)
    out = {}
    # G1 tables
    pi = np.zeros((24, 32), np.int8)
    for k in range(24):
        R.ref_pi_codes(k + 1, pi[k])
    perm = np.zeros(1536, np.int16)
    R.ref_freq_interleaver(perm)
    prs = np.zeros(4096, np.float32)
    R.ref_phase_table(prs)
    uep = np.zeros(192, np.int16)
    R.ref_uep_table(uep)                                   # FIG 0/1 short-form table, fib_table.h
    out.update(pi_codes=pi, freq_perm=perm, prs_table=prs, uep_table=uep.reshape(64, 3))
    maps = {}
    for kbps, prot in [(8, 0), (8, 1), (8, 2), (8, 3), (16, 1), (32, 2), (32, 4), (32, 7), (64, 0), (64, 2), (64, 3), (64, 5), (128, 1), (192, 3), (256, 4), (320, 2)]:
        m = np.zeros(96 * kbps + 24, np.int32)
        n = R.ref_eep_map(kbps, prot, m)
        maps["eep_%d_%d" % (kbps, prot)] = (n, hashlib.sha256(m.tobytes()).hexdigest())
    for kbps, prot in [(32, 5), (32, 1), (48, 3), (56, 2), (64, 5), (64, 4), (80, 1), (96, 3), (112, 4), (128, 1), (160, 2), (192, 5), (224, 3), (256, 4), (320, 2)]:
        m = np.zeros(96 * kbps + 24, np.int32)
        n = R.ref_uep_map(kbps, prot, m)
        maps["uep_%d_%d" % (kbps, prot)] = (n, hashlib.sha256(m.tobytes()).hexdigest())
    out["map_names"] = np.array(sorted(maps))
    out["map_n_in"] = np.array([maps[k][0] for k in sorted(maps)], np.int32)
    out["map_sha256"] = np.array([maps[k][1] for k in sorted(maps)])
    # G2 Viterbi: FIC + a 64 kbit/s block, random / ties / saturating inputs (canonical scalar build)
    for n in (768, 1536, 192):
        m = 4 * (n + 6)
        soft = np.stack([rng.integers(-200, 201, m), rng.integers(-40, 41, m), np.zeros(m, np.int64), np.full(m, 127), np.full(m, -127),
                         rng.choice([-32768, -32767, 32767, 32640, 32641, -200, 200], m), rng.choice([-127, 0, 127, 128, -128, 1, -1], m)]).astype(np.int16)
        bits = np.zeros((len(soft), n), np.uint8)
        for i in range(len(soft)):
            R.ref_viterbi(soft[i], n, bits[i])
        out["vit%d_soft" % n] = soft
        out["vit%d_bits" % n] = np.packbits(bits, axis=1)
    # Protection::deconvolve
    for name, fn, kbps, prot in (("eep64_2", R.ref_eep_deconvolve, 64, 2), ("eep32_4", R.ref_eep_deconvolve, 32, 4), ("uep64_3", R.ref_uep_deconvolve, 64, 3)):
        m = np.zeros(96 * kbps + 24, np.int32)
        n_in = (R.ref_uep_map if name.startswith("uep") else R.ref_eep_map)(kbps, prot, m)
        soft = rng.integers(-150, 151, (3, n_in)).astype(np.int16)
        bits = np.zeros((3, 24 * kbps), np.uint8)
        for i in range(3):
            fn(kbps, prot, soft[i], n_in, bits[i])
        out["dec_%s_soft" % name] = soft
        out["dec_%s_bits" % name] = np.packbits(bits, axis=1)
    # G5 RS (0..8 byte errors, garbage) and fire code (bursts)
    cws, outs, rets = [], [], []
    for trial in range(120):
        data = rng.integers(0, 256, 110).astype(np.uint8)
        cw = np.zeros(120, np.uint8)
        R.ref_rs_enc(data, cw)
        if trial % 20 == 19:
            cw = rng.integers(0, 256, 120).astype(np.uint8)
        else:
            pos = rng.choice(120, trial % 9, replace=False)
            cw[pos] ^= rng.integers(1, 256, trial % 9).astype(np.uint8)
        o = np.zeros(110, np.uint8)
        rets.append(R.ref_rs_dec(cw, o)); cws.append(cw); outs.append(o)
    out.update(rs_in=np.array(cws), rs_out=np.array(outs), rs_ret=np.array(rets, np.int16))
    xs = np.zeros((400, 12), np.uint8)
    for i in range(400):
        xs[i, :11] = rng.integers(0, 256, 11)
        if i % 4:
            crc = 0
            for b in xs[i, 2:11]:
                crc ^= int(b) << 8
                for _ in range(8):
                    crc = ((crc << 1) ^ 0x782F) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
            xs[i, 0], xs[i, 1] = crc >> 8, crc & 0xFF
        if i % 4 >= 2:
            blen = int(rng.integers(1, 9)); start = int(rng.integers(0, 88 - blen + 1))
            for k in range(blen):
                if k in (0, blen - 1) or rng.random() < 0.5:
                    xs[i, (start + k) // 8] ^= 0x80 >> ((start + k) % 8)
    chk = np.array([R.ref_firecode_check(x.copy()) for x in xs], np.uint8)
    fixed = xs.copy()
    okc = np.array([R.ref_firecode_check_and_correct(fixed[i]) for i in range(400)], np.uint8)
    out.update(fc_in=xs, fc_check=chk, fc_fixed=fixed, fc_ok=okc)
    # CRC
    msgs = rng.integers(0, 256, (64, 34)).astype(np.uint8)
    for i in range(0, 64, 2):
        c = R.ref_calc_crc(msgs[i], 30)
        msgs[i, 30], msgs[i, 31] = (c >> 8) & 0xFF, c & 0xFF
    out.update(crc_msgs=msgs, crc_bytes_ok=np.array([R.ref_check_crc_bytes(m_, 30) for m_ in msgs], np.uint8),
               crc_bits_ok=np.array([R.ref_check_crc_bits(np.unpackbits(m_[:32]), 256) for m_ in msgs], np.uint8))
    # G1 extended: the depuncture map of EVERY profile the reference can hold (EN 300 401 11.3: EEP-A in steps of 8 kbit/s,
    # EEP-B in steps of 32, the UEP rows of table 8) as (transmitted bits, SHA-256 of the index list).  Above 341 kbit/s the
    # reference's i16 indices wrap (96 * kbps + 24 > 32767) and its constructors write out of bounds: those are left out.
    allmaps = {}
    for kbps in range(8, 337, 8):
        for prot in range(8):
            if prot >= 4 and kbps % 32:
                continue
            m = np.zeros(96 * kbps + 24, np.int32)
            n = R.ref_eep_map(kbps, prot, m)
            allmaps["eep_%d_%d" % (kbps, prot)] = (n, hashlib.sha256(m.tobytes()).hexdigest())
    for cu, lvl, kbps in uep.reshape(64, 3).tolist():
        if kbps > 336:
            continue
        m = np.zeros(96 * kbps + 24, np.int32)
        n = R.ref_uep_map(kbps, lvl, m)
        allmaps["uep_%d_%d" % (kbps, lvl)] = (n, hashlib.sha256(m.tobytes()).hexdigest())
    out["allmap_names"] = np.array(sorted(allmaps))
    out["allmap_n_in"] = np.array([allmaps[k][0] for k in sorted(allmaps)], np.int32)
    out["allmap_sha256"] = np.array([allmaps[k][1] for k in sorted(allmaps)])
    # G2 extended: 128 and 384 kbit/s trellises; the inputs are regenerated from the seed by the tests (viterbi_big_inputs)
    for n in (3072, 9216):
        soft = viterbi_big_inputs(n)
        bits = np.zeros((len(soft), n), np.uint8)
        for i in range(len(soft)):
            R.ref_viterbi(soft[i], n, bits[i])
        out["vitbig%d_bits" % n] = np.packbits(bits, axis=1)
    np.savez_compressed(os.path.join(HERE, "ref_leaf_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_leaf_vectors.npz"), os.path.getsize(os.path.join(HERE, "ref_leaf_vectors.npz")), "bytes")


if __name__ == "__main__":
    main()
