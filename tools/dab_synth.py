"""Synthetic DAB Mode-I ensemble modulator (test/bench infrastructure, numpy only).

Builds what SURVEY.md 8(d) calls the canonical synthetic ensemble: FIC with FIG 0/0 (CIF counter)
+ FIG 0/1 (sub-channel organisation), N sub-channels carrying DAB+ super frames (fire code, AU
table, AU CRCs, RS(120,110) parity), energy dispersal, K=7 r=1/4 convolutional code with EEP/UEP/FIC
puncturing, 16-CIF time interleaving, QPSK + frequency interleaving + pi/4-D-QPSK against the phase
reference symbol, 2048-IFFT + 504 cyclic prefix, 2656-sample null symbol, then a channel (gain, CFO,
timing offset, AWGN).  Everything follows ETSI EN 300 401 as mirrored by the receiver side of the
reference (see oracle/ for file:line); the reference itself contains no modulator.

The generated ensemble can be made *cyclic* (time interleaver, super frames and CIF counter wrap
around) so that a ring buffer of n_frames frames can be replayed forever by the bench.
"""
from __future__ import annotations

import dataclasses

import numpy as np

L, K, TN, TF, TS, TU, TG = 76, 1536, 2656, 196608, 2552, 2048, 504
FS = 2048000
INTERLEAVE_MAP = np.array([0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15])
POLYS = (109, 79, 83, 109)   # Karn bit order == octal 133,171,145,133


# ----------------------------------------------------------------------------- tables
def pi_code(pi: int) -> np.ndarray:
    order = [0, 4, 2, 6, 1, 5, 3, 7]
    base, extra = (pi - 1) // 8 + 1, (pi - 1) % 8 + 1
    ones = [base] * 8
    for e in range(extra):
        ones[order[e]] = base + 1
    return np.array([[1 if j < ones[g] else 0 for j in range(4)] for g in range(8)], np.uint8).reshape(32)


def _blocks(L_, pi):
    return np.tile(np.tile(pi_code(pi), 4), L_) if L_ > 0 else np.zeros(0, np.uint8)


def eep_mask(kbps: int, prot: int) -> np.ndarray:
    """1 = transmitted, over the 96*kbps+24 mother-code bits (EN 300 401 11.3.2)."""
    lvl, opt = prot & 3, (prot >> 2) & 1
    if opt == 0:
        n = kbps // 8
        L1, L2, p1, p2 = [(6 * n - 3, 3, 24, 23),
                          (5, 1, 13, 12) if n == 1 else (2 * n - 3, 4 * n + 3, 14, 13),
                          (6 * n - 3, 3, 8, 7),
                          (4 * n - 3, 2 * n + 3, 3, 2)][lvl]
    else:
        n = kbps // 32
        p1 = [10, 6, 4, 2][lvl]
        L1, L2, p2 = 24 * n - 3, 3, p1 - 1
    return np.concatenate([_blocks(L1, p1), _blocks(L2, p2), pi_code(8)[:24]])


def fic_mask() -> np.ndarray:
    return np.concatenate([_blocks(21, 16), _blocks(3, 15), pi_code(8)[:24]])


def prbs(n: int) -> np.ndarray:
    sr = [1] * 9
    out = np.zeros(n, np.uint8)
    for i in range(n):
        b = sr[8] ^ sr[4]
        sr = [b] + sr[:8]
        out[i] = b
    return out


def freq_perm() -> np.ndarray:
    t = np.zeros(TU, np.int64)
    for i in range(1, TU):
        t[i] = (13 * t[i - 1] + 511) % TU
    keep = [v - TU // 2 for v in t if v != TU // 2 and 256 <= v <= 256 + K]
    return np.array(keep, np.int64)


_PRS_N_LO = [1, 2, 0, 1, 3, 2, 2, 3, 2, 1, 2, 3, 1, 2, 3, 3, 2, 2, 2, 1, 1, 3, 1, 2]
_PRS_N_HI = [3, 1, 1, 1, 2, 2, 1, 0, 2, 2, 3, 3, 0, 2, 1, 3, 3, 3, 3, 0, 3, 0, 1, 1]
_PRS_H = [[0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1], [0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0],
          [0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3], [0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2]]


def prs_spectrum() -> np.ndarray:
    """Phase reference symbol in FFT bin order (EN 300 401 14.3.2)."""
    z = np.zeros(TU, np.complex128)
    for k in list(range(-768, 0)) + list(range(1, 769)):
        if k < 0:
            b = (k + 768) // 32
            kp, i, n = -768 + 32 * b, b & 3, _PRS_N_LO[b]
        else:
            b = (k - 1) // 32
            kp, i, n = 1 + 32 * b, (4 - (b & 3)) & 3, _PRS_N_HI[b]
        z[k % TU] = np.exp(1j * np.pi / 2 * (_PRS_H[i][(k - kp) & 15] + n))
    return z


def crc16(data: bytes) -> int:
    crc = 0xFFFF
    for b in data:
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ 0xFFFF


def firecode_parity(nine: bytes) -> int:
    crc = 0
    for b in nine:
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x782F) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


# GF(256) / RS(255,245) shortened to (120,110): poly 0x11D, first root alpha^0 (EN 102 563)
_GF_EXP = np.zeros(512, np.int64)
_GF_LOG = np.zeros(256, np.int64)
_x = 1
for _i in range(255):
    _GF_EXP[_i] = _x
    _GF_LOG[_x] = _i
    _x <<= 1
    if _x & 0x100:
        _x ^= 0x11D
_GF_EXP[255:510] = _GF_EXP[:255]


def _gf_mul(a, b):
    return 0 if a == 0 or b == 0 else int(_GF_EXP[_GF_LOG[a] + _GF_LOG[b]])


def _rs_gen():
    g = [1]
    for r in range(10):
        g = [(_gf_mul(g[i], int(_GF_EXP[r])) if i < len(g) else 0) ^ (g[i - 1] if i > 0 else 0) for i in range(len(g) + 1)]
    return g   # g[0] + g[1] x + ... + x^10


_RS_G = _rs_gen()


def rs_parity(data110: np.ndarray) -> np.ndarray:
    """Systematic RS parity: remainder of data(x) * x^10 by g(x), highest power first."""
    rem = [0] * 10
    for d in data110:
        fb = int(d) ^ rem[0]
        rem = rem[1:] + [0]
        if fb:
            for j in range(10):
                rem[j] ^= _gf_mul(fb, _RS_G[9 - j])
    return np.array(rem, np.uint8)


# ----------------------------------------------------------------------------- coding
def conv_encode(bits: np.ndarray) -> np.ndarray:
    """K=7 rate 1/4 mother code incl. 6 tail bits -> 4*(n+6) bits, order x0,x1,x2,x3 per input bit."""
    n = len(bits)
    pad = np.concatenate([np.zeros(6, np.uint8), bits.astype(np.uint8), np.zeros(6, np.uint8)])
    out = np.zeros((n + 6, 4), np.uint8)
    for p, poly in enumerate(POLYS):
        acc = np.zeros(n + 6, np.uint8)
        for d in range(7):
            if (poly >> d) & 1:
                acc ^= pad[6 - d: 6 - d + n + 6]
        out[:, p] = acc
    return out.reshape(-1)


@dataclasses.dataclass
class SubCh:
    subch_id: int
    cu_start: int
    cu_size: int
    kbps: int
    prot_level: int = 2      # EEP 3-A
    short_form: int = 0
    mask: object = None      # optional puncturing mask over the 96*kbps+24 mother-code bits (UEP: from the caller)
    dab_plus: int = 1        # 0: payload is an opaque byte stream (MP2 / data), no super-frame structure


def default_subchannels(n: int = 18, kbps: int = 64) -> list[SubCh]:
    cu = {8: 6, 16: 12, 32: 24, 48: 36, 64: 48, 96: 72, 128: 96}[kbps]   # EEP 3-A: 6 CU per 8 kbps
    return [SubCh(i, i * cu, cu, kbps, 2, 0) for i in range(n)]


def build_superframe(kbps: int, rng: np.random.Generator) -> np.ndarray:
    """One DAB+ audio super frame (ETSI TS 102 563): 110*s payload bytes + RS parity, s = kbps/8,
    returned as the 120*s byte sequence that is cut into 5 logical frames."""
    s = kbps // 8
    n = 110 * s
    sf = rng.integers(0, 256, n).astype(np.uint8)
    if n // 3 <= 900:
        # header: dac_rate=1, sbr=1 -> 3 AUs, first starts at 6
        sf[2] = (1 << 6) | (1 << 5) | (1 << 4)
        a0, a1, a2 = 6, 6 + (n - 6) // 3, 6 + 2 * ((n - 6) // 3)
        sf[3], sf[4], sf[5] = a1 >> 4, ((a1 & 0xF) << 4) | (a2 >> 8), a2 & 0xFF
        bounds = (a0, a1, a2, n)
    else:
        # high bit rates: dac_rate=1, sbr=0 -> 6 AUs (each must stay <= 960 bytes), first starts at 11
        sf[2] = (1 << 6) | (1 << 4)
        step = min((n - 11) // 6, (4095 - 11) // 5)        # AU starts are 12-bit fields (beyond 352 kbit/s the last AU takes the rest)
        a = [11 + i * step for i in range(6)]
        sf[3], sf[4], sf[5] = a[1] >> 4, ((a[1] & 0xF) << 4) | (a[2] >> 8), a[2] & 0xFF
        sf[6], sf[7], sf[8] = a[3] >> 4, ((a[3] & 0xF) << 4) | (a[4] >> 8), a[4] & 0xFF
        sf[9], sf[10] = a[5] >> 4, (a[5] & 0xF) << 4
        bounds = tuple(a) + (n,)
    for st, en in zip(bounds[:-1], bounds[1:]):
        c = crc16(bytes(sf[st:en - 2]))
        sf[en - 2], sf[en - 1] = c >> 8, c & 0xFF
    fc = firecode_parity(bytes(sf[2:11]))
    sf[0], sf[1] = fc >> 8, fc & 0xFF
    full = np.zeros(120 * s, np.uint8)
    full[:n] = sf
    for j in range(s):
        full[n + j::s][:10] = rs_parity(sf[j::s])
    return full


UEP_ROWS = [(k, l) for k, ls in [(32, (5, 4, 3, 2, 1)), (48, (5, 4, 3, 2, 1)), (56, (5, 4, 3, 2)), (64, (5, 4, 3, 2, 1)),
                                 (80, (5, 4, 3, 2, 1)), (96, (5, 4, 3, 2, 1)), (112, (5, 4, 3, 2)), (128, (5, 4, 3, 2, 1)),
                                 (160, (5, 4, 3, 2, 1)), (192, (5, 4, 3, 2, 1)), (224, (5, 4, 3, 2, 1)), (256, (5, 4, 3, 2, 1)),
                                 (320, (5, 4, 2)), (384, (5, 3, 1))] for l in ls]


TII_PATTERNS = [v for v in range(256) if bin(v).count("1") == 4]      # EN 300 401 table 43: 70 combinations, ascending


def tii_null_spectrum(main_id: int, sub_id: int, amp: float = 1.0, etsi: bool = True) -> np.ndarray:
    """FFT-order spectrum (2048 bins) of the TII signal of transmitter (main_id, sub_id), EN 300 401 14.8.1: in each of
    the four 384-carrier blocks, carrier pairs (k, k+1) at pair index 24 g + sub_id for the four groups g of the pattern;
    both carriers carry the phase-reference phase of the FIRST one (etsi) or each its own (the deviation some
    transmitters show, which the reference detects as 'non-ETSI phase')."""
    prs = prs_spectrum()
    z = np.zeros(TU, np.complex128)
    pat = TII_PATTERNS[main_id]
    for b in range(4):
        for g in range(8):
            if pat & (0x80 >> g):
                i = b * 192 + g * 24 + sub_id
                k = -768 + 2 * i
                f0 = k + TU if k < 0 else k + 1
                z[f0] = amp * prs[f0]
                z[f0 + 1] = amp * (prs[f0] if etsi else prs[f0 + 1])
    return z


def build_fibs(subch: list[SubCh], cif_count: int, eid: int = 0x10F2) -> np.ndarray:
    """3 FIBs (one FIC group, 96 bytes) for the CIF with the given counter."""
    hi, lo = (cif_count // 250) % 20, cif_count % 250

    def fig01(chs):
        body = bytearray([0x01])          # C/N=0 OE=0 P/D=0 ext=1
        for c in chs:
            if c.short_form:              # UEP: table index of (bit rate, protection level), EN 300 401 table 8
                idx = UEP_ROWS.index((c.kbps, c.prot_level))
                body += bytes([(c.subch_id << 2) | (c.cu_start >> 8), c.cu_start & 0xFF, idx])
                continue
            opt, lvl = (c.prot_level >> 2) & 1, c.prot_level & 3
            w = (c.subch_id << 26) | (c.cu_start << 16) | (1 << 15) | (opt << 12) | (lvl << 10) | c.cu_size
            body += w.to_bytes(4, "big")
        return bytes([len(body)]) + bytes(body)   # type 0 -> header = length

    def fig02(chs):
        body = bytearray([0x02])          # C/N=0 OE=0 P/D=0 ext=2
        for c in chs:                     # one programme service per sub-channel, one DAB+ audio component (ASCTy 63)
            sid = 0x1000 + c.subch_id
            comp = (0 << 14) | ((63 if c.dab_plus else 0) << 8) | (c.subch_id << 2) | (1 << 1)
            body += bytes([sid >> 8, sid & 0xFF, 0x01, comp >> 8, comp & 0xFF])
        return bytes([len(body)]) + bytes(body)

    fig00 = bytes([0x05, 0x00, eid >> 8, eid & 0xFF, hi & 0x1F, lo])
    phase = cif_count % 4                 # FIG 0/1 on even CIFs, FIG 0/2 on odd ones (15 + 3 services)
    if len(subch) > 18:                   # large ensembles: the description rotates through the CIFs (19 / 14 entries each)
        n, turn = len(subch), cif_count // 2
        take = lambda o, k: [subch[(o + i) % n] for i in range(k)]      # noqa: E731
        if cif_count % 2 == 0:
            o = (19 * turn) % n
            payloads = [fig01(take(o, 5)), fig01(take(o + 5, 7)), fig01(take(o + 12, 7))]
        else:
            o = (14 * turn) % n
            payloads = [fig02(take(o, 4)), fig02(take(o + 4, 5)), fig02(take(o + 9, 5))]
    elif phase in (0, 2):
        payloads = [fig01(subch[0:5]), fig01(subch[5:12]), fig01(subch[12:18])]
    elif phase == 1:
        payloads = [fig02(subch[0:4]), fig02(subch[4:9]), fig02(subch[9:14])]
    else:
        payloads = [fig02(subch[14:18]), b"", b""]
    fibs = []
    for g, pl in enumerate(payloads):
        data = (fig00 if g == 0 else b"") + (pl if len(pl) > 2 else b"")
        assert len(data) <= 30
        if len(data) < 30:
            data += b"\xFF" + b"\x00" * (29 - len(data))
        c = crc16(data)
        fibs.append(data + bytes([c >> 8, c & 0xFF]))
    return np.frombuffer(b"".join(fibs), np.uint8).copy()


@dataclasses.dataclass
class Ensemble:
    n_frames: int
    subch: list
    iq: np.ndarray             # clean, unit-scaled complex64, n_frames*TF samples, frame = null + 76 symbols
    fibs: np.ndarray           # [n_frames, 12, 32] uint8
    msc_bytes: list            # per sub-channel: [n_cif, 3*kbps] uint8 (logical frames, de-dispersed payload)
    superframes: list          # per sub-channel: [n_cif/5, 110*kbps/8] uint8
    tx_bits: np.ndarray        # [n_frames, 75, 3072] uint8 transmitted (interleaved) bits per OFDM symbol


def _ofdm_frames(tx_bits: np.ndarray, n_frames: int, cyclic: bool, cif_start: int, tii: list | None) -> np.ndarray:
    """[n_frames, 75, 3072] transmitted bits -> [n_frames, TF] complex64 frames (null symbol + PRS + 75 D-QPSK symbols)."""
    # ---- QPSK, frequency interleaving, D-QPSK, OFDM
    perm = freq_perm() % TU
    prs = prs_spectrum()
    q = ((1 - 2.0 * tx_bits[:, :, :K]) + 1j * (1 - 2.0 * tx_bits[:, :, K:])) / np.sqrt(2)
    iq = np.zeros((n_frames, TF), np.complex64)
    scale = 1.0 / np.sqrt(K)          # unit mean power in the useful part
    tii_z = None
    if tii:                               # [(main_id, sub_id, amplitude, etsi)]: sum of the transmitters' TII signals
        tii_z = sum(tii_null_spectrum(m, c, a, e) for (m, c, a, e) in tii)
    for f in range(n_frames):
        # the null symbol that opens frame f closes frame f-1: it carries TII when the CIF counter of the last FIG 0/0
        # of that frame has (count & 7) >= 4 (the receiver's rule, dab_processor.cpp:274)
        if tii_z is not None and ((cif_start + 4 * ((f - 1) % n_frames if cyclic else f - 1) + 3) & 7) >= 4 and (cyclic or f > 0):
            t = np.fft.ifft(tii_z) * TU * scale
            iq[f, 0:TG] = t[-TG:]
            iq[f, TG:TG + TU] = t
            iq[f, TG + TU:TN] = t[:TN - TG - TU]
        z = prs.copy()
        pos = TN
        for l in range(L):
            if l > 0:
                y = np.zeros(TU, np.complex128)
                y[perm] = q[f, l - 1]
                z = np.where(np.abs(prs) > 0, z * np.where(y == 0, 1, y), 0)
            t = np.fft.ifft(z) * TU * scale
            iq[f, pos: pos + TG] = t[-TG:]
            iq[f, pos + TG: pos + TS] = t
            pos += TS
    return iq


def build_ensemble(n_frames: int = 10, subch: list | None = None, seed: int = 0, cyclic: bool = True,
                   cif_start: int = 0, tii: list | None = None) -> Ensemble:
    subch = default_subchannels() if subch is None else subch
    rng = np.random.default_rng(seed)
    n_cif = 4 * n_frames
    if cyclic:
        assert n_cif % 5 == 0, "cyclic ensembles need a whole number of super frames"
    # ---- MSC logical frames -> coded CIFs
    coded = np.zeros((n_cif, 55296), np.uint8)
    coded[:] = rng.integers(0, 2, coded.shape, dtype=np.uint8)   # padding in unused CUs
    msc_bytes, superframes = [], []
    for c in subch:
        nb = 3 * c.kbps
        n_sf = (n_cif + 4) // 5
        sfs = [build_superframe(c.kbps, rng) if c.dab_plus else rng.integers(0, 256, 15 * c.kbps).astype(np.uint8) for _ in range(n_sf)]
        stream = np.concatenate(sfs)[: n_cif * nb].reshape(n_cif, nb)
        msc_bytes.append(stream.copy())
        superframes.append(np.stack([s[: 110 * (c.kbps // 8)] for s in sfs[: n_cif // 5]]) if n_cif >= 5 else np.zeros((0, 0), np.uint8))
        mask = (eep_mask(c.kbps, c.prot_level) if c.mask is None else np.asarray(c.mask)).astype(bool)
        disp = prbs(24 * c.kbps)
        for q in range(n_cif):
            bits = np.unpackbits(stream[q]) ^ disp
            cw = conv_encode(bits)[mask]
            assert len(cw) <= c.cu_size * 64, (len(cw), c.cu_size)      # some UEP profiles leave padding bits
            coded[q, c.cu_start * 64: c.cu_start * 64 + len(cw)] = cw
    # ---- time interleaving: CIF t carries bit i of coded CIF t - map[i%16]
    delay = INTERLEAVE_MAP[np.arange(55296) & 15]
    tx_cif = np.zeros_like(coded)
    for t in range(n_cif):
        src = t - delay
        if cyclic:
            src %= n_cif
            tx_cif[t] = coded[src, np.arange(55296)]
        else:
            ok = src >= 0
            tx_cif[t, ok] = coded[src[ok], np.arange(55296)[ok]]
    # ---- FIC
    fmask = fic_mask().astype(bool)
    fdisp = prbs(768)
    fibs = np.zeros((n_frames, 12, 32), np.uint8)
    tx_bits = np.zeros((n_frames, 75, 3072), np.uint8)
    for f in range(n_frames):
        fic = []
        for g in range(4):
            fb = build_fibs(subch, cif_start + 4 * f + g)
            fibs[f, 3 * g: 3 * g + 3] = fb.reshape(3, 32)
            fic.append(conv_encode(np.unpackbits(fb) ^ fdisp)[fmask])
        tx_bits[f, :3] = np.concatenate(fic).reshape(3, 3072)
        tx_bits[f, 3:] = tx_cif[4 * f: 4 * f + 4].reshape(72, 3072)
    iq = _ofdm_frames(tx_bits, n_frames, cyclic, cif_start, tii)
    return Ensemble(n_frames, subch, iq.reshape(-1), fibs, msc_bytes, superframes, tx_bits)


# ---------------------------------------------------------------------------------------------- multiplex reconfiguration
def fig00_bytes(cif_count: int, eid: int = 0x10F2, change_flags: int = 0, occurrence: int = 0) -> bytes:
    """FIG 0/0 (EN 300 401 6.4.1): EId, change flags, CIF count and -- while a change is announced -- OccurrenceChange."""
    hi, lo = (cif_count // 250) % 20, cif_count % 250
    body = bytes([0x00, eid >> 8, eid & 0xFF, ((change_flags & 3) << 6) | (hi & 0x1F), lo]) + (bytes([occurrence]) if change_flags else b"")
    return bytes([len(body)]) + body


def fig01_bytes(chs, cn: int = 0) -> bytes:
    body = bytearray([(cn << 7) | 0x01])
    for c in chs:
        if c.short_form:
            idx = UEP_ROWS.index((c.kbps, c.prot_level))
            body += bytes([(c.subch_id << 2) | (c.cu_start >> 8), c.cu_start & 0xFF, idx])
            continue
        opt, lvl = (c.prot_level >> 2) & 1, c.prot_level & 3
        w = (c.subch_id << 26) | (c.cu_start << 16) | (1 << 15) | (opt << 12) | (lvl << 10) | c.cu_size
        body += w.to_bytes(4, "big")
    return bytes([len(body)]) + bytes(body)


def fig02_bytes(chs, cn: int = 0) -> bytes:
    body = bytearray([(cn << 7) | 0x02])
    for c in chs:
        sid = 0x1000 + c.subch_id
        comp = (0 << 14) | ((63 if c.dab_plus else 0) << 8) | (c.subch_id << 2) | (1 << 1)
        body += bytes([sid >> 8, sid & 0xFF, 0x01, comp >> 8, comp & 0xFF])
    return bytes([len(body)]) + bytes(body)


def pack_fibs(figs: list) -> np.ndarray:
    """FIGs (header included, in order) into the three FIBs of one CIF: end marker, padding, CRC.  96 bytes."""
    fibs, cur = [], b""
    for g in figs:
        assert len(g) <= 30
        if len(cur) + len(g) > 30:
            fibs.append(cur); cur = b""
        cur += g
    fibs.append(cur)
    assert len(fibs) <= 3, "the FIGs of this CIF do not fit into three FIBs"
    out = []
    for data in fibs + [b""] * (3 - len(fibs)):
        if len(data) < 30:
            data += b"\xFF" + b"\x00" * (29 - len(data))
        c = crc16(data)
        out.append(data + bytes([c >> 8, c & 0xFF]))
    return np.frombuffer(b"".join(out), np.uint8).copy()


def build_fibs_reconf(cur: list, nxt: list | None, cif_count: int, change_flags: int = 0, occurrence: int = 0) -> np.ndarray:
    """One CIF's FIC group while a reconfiguration may be announced: FIG 0/0 in every CIF; the current configuration's FIG 0/1 and
    0/2 (C/N = 0) on CIFs 0 and 1 of four, the next one's (C/N = 1) on CIFs 2 and 3 while it is announced (else the current again)."""
    def chunks(chs, first, rest):
        out, k = [], first
        while chs:
            out.append(chs[:k]); chs = chs[k:]; k = rest
        return out
    figs = [fig00_bytes(cif_count, change_flags=change_flags, occurrence=occurrence)]
    phase = cif_count % 4
    which, cn = (nxt, 1) if (nxt is not None and phase >= 2) else (cur, 0)
    if phase % 2 == 0:
        figs += [fig01_bytes(ch, cn) for ch in chunks(list(which), 5, 7)]
    else:
        figs += [fig02_bytes(ch, cn) for ch in chunks(list(which), 4, 5)]
    return pack_fibs(figs)


@dataclasses.dataclass
class ReconfEnsemble:
    n_frames: int
    subch_a: list              # the configuration before the switch ...
    subch_b: list              # ... and from CIF switch_cif on
    switch_cif: int
    iq: np.ndarray
    fibs: np.ndarray           # [n_frames, 12, 32]
    payload: dict              # (layout "a" / "b", subch_id) -> (first CIF, [n, 3*kbps] logical frames from there on); a sub-channel
                               # with the same description in both layouts runs through and appears under "a" only


def _same_desc(x, y):
    return (x.subch_id, x.cu_start, x.cu_size, x.kbps, x.prot_level, x.short_form, x.dab_plus) == \
           (y.subch_id, y.cu_start, y.cu_size, y.kbps, y.prot_level, y.short_form, y.dab_plus)


def build_reconfigured_ensemble(n_frames: int, subch_a: list, subch_b: list, switch_frame: int, announce_frames: int = 8,
                                seed: int = 0, cif_start: int = 0, switch_cif_in_frame: int = 0) -> ReconfEnsemble:
    """A multiplex reconfiguration (EN 300 401 6.5) at CIF switch_cif_in_frame (0..3) of frame switch_frame: layout A before, layout B from then on,
    announced for announce_frames frames in advance (FIG 0/0 change flags 3 + OccurrenceChange, the next configuration's FIG 0/1
    and 0/2 with C/N = 1); afterwards the flags are 0 and B is the current configuration.  Sub-channels described identically in
    A and B run through (same convolutional interleaver, no gap), and so does one that only moves to other capacity units (its bits are
    sent at the new address from the switch on); one that ends has its last 15 logical frames cut off in
    the air (their later interleaver branches fall on CUs that belong to B); one that begins starts its interleaver at the switch."""
    rng = np.random.default_rng(seed)
    n_cif, N = 4 * n_frames, 4 * switch_frame + switch_cif_in_frame
    through = [c for c in subch_a if any(_same_desc(c, d) for d in subch_b)]
    # a sub-channel that only MOVES (same id, size, bit rate, protection; other capacity units) keeps its interleaver running: from the
    # switch on its bits are sent at the new address
    def _moved_to(c):
        for d in subch_b:
            if (d.subch_id, d.cu_size, d.kbps, d.prot_level, d.short_form, d.dab_plus) == (c.subch_id, c.cu_size, c.kbps, c.prot_level, c.short_form, c.dab_plus) \
                    and d.cu_start != c.cu_start:
                return d
        return None
    moved = {c.subch_id: _moved_to(c) for c in subch_a if _moved_to(c) is not None}
    only_b = [d for d in subch_b if not any(_same_desc(c, d) for c in subch_a) and d.subch_id not in moved]
    pos = np.arange(55296)
    delay = INTERLEAVE_MAP[pos & 15]

    def code(c, first, last):
        """coded bits of the sub-channel's logical frames first..last-1: [n_cif, 64 * cu_size] (zero outside), payload"""
        nb = 3 * c.kbps
        n = last - first
        n_sf = (n + 4) // 5
        sfs = [build_superframe(c.kbps, rng) if c.dab_plus else rng.integers(0, 256, 15 * c.kbps).astype(np.uint8) for _ in range(n_sf)]
        stream = np.concatenate(sfs)[: n * nb].reshape(n, nb)
        mask = (eep_mask(c.kbps, c.prot_level) if c.mask is None else np.asarray(c.mask)).astype(bool)
        disp = prbs(24 * c.kbps)
        out = rng.integers(0, 2, (n_cif, 64 * c.cu_size), dtype=np.uint8)          # what is not payload is random
        for q in range(first, last):
            cw = conv_encode(np.unpackbits(stream[q - first]) ^ disp)[mask]
            out[q, : len(cw)] = cw
        return out, stream

    tx_cif = rng.integers(0, 2, (n_cif, 55296), dtype=np.uint8)                  # padding / the cut-off branches: random
    payload = {}
    for layout, chs, first, last, t0, t1 in (("a", subch_a, 0, None, 0, None), ("b", only_b, N, n_cif, N, n_cif)):
        for c in chs:
            runs_through = layout == "a" and (c in through or c.subch_id in moved)
            lf_last = n_cif if (layout == "b" or runs_through) else N                # logical frames it carries
            cif_last = n_cif if (layout == "b" or runs_through) else N               # CIFs in which its CUs are its own
            coded, stream = code(c, first, lf_last)
            payload[(layout, c.subch_id)] = (first, stream)
            cols = np.arange(64 * c.cu_size)
            for t in range(t0, cif_last):
                start = moved[c.subch_id].cu_start if (layout == "a" and c.subch_id in moved and t >= N) else c.cu_start
                sl = slice(start * 64, (start + c.cu_size) * 64)
                d = delay[sl]                                                         # (the branch a bit takes depends on its position in the CIF)
                src = t - d
                ok = (src >= first) & (src < lf_last)
                row = tx_cif[t, sl]
                row[ok] = coded[src[ok], cols[ok]]
    # ---- FIC
    fmask = fic_mask().astype(bool)
    fdisp = prbs(768)
    fibs = np.zeros((n_frames, 12, 32), np.uint8)
    tx_bits = np.zeros((n_frames, 75, 3072), np.uint8)
    occurrence = (cif_start + N) % 250
    for f in range(n_frames):
        fic = []
        for g in range(4):
            q = 4 * f + g
            announced = N - 4 * announce_frames <= q < N
            fb = build_fibs_reconf(subch_a if q < N else subch_b, subch_b if announced else None, cif_start + q,
                                   change_flags=3 if announced else 0, occurrence=occurrence)
            fibs[f, 3 * g: 3 * g + 3] = fb.reshape(3, 32)
            fic.append(conv_encode(np.unpackbits(fb) ^ fdisp)[fmask])
        tx_bits[f, :3] = np.concatenate(fic).reshape(3, 3072)
        tx_bits[f, 3:] = tx_cif[4 * f: 4 * f + 4].reshape(72, 3072)
    iq = _ofdm_frames(tx_bits, n_frames, False, cif_start, None)
    return ReconfEnsemble(n_frames, list(subch_a), list(subch_b), N, iq.reshape(-1), fibs, payload)


def channel(iq: np.ndarray, snr_db: float = 20.0, cfo_hz: float = 0.0, timing_offset: int = 0, gain: float = 0.25,
            seed: int = 0, cyclic: bool = True, n_out: int | None = None) -> np.ndarray:
    """Gain, CFO, timing offset (cyclic roll, or zero prefix when not cyclic) and AWGN. complex64."""
    rng = np.random.default_rng(seed)
    x = np.roll(iq, timing_offset) if cyclic else np.concatenate([np.zeros(timing_offset, iq.dtype), iq])
    if n_out is not None:
        reps = -(-n_out // len(x))
        x = np.tile(x, reps)[:n_out]
    n = np.arange(len(x), dtype=np.float64)
    x = x * np.exp(2j * np.pi * cfo_hz * n / FS)
    sigma = np.sqrt(10 ** (-snr_db / 10) / 2)
    x = x + sigma * (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x)))
    return (gain * x).astype(np.complex64)


# Tap sets after the COST 207 profiles the DAB standard's own simulations use (EN 300 401 informative annexes / TR 101 496):
# delays in microseconds, mean powers in dB.  2.048 samples per microsecond.
MOBILE_PROFILES = {
    "TU6": [(0.0, -3.0), (0.2, 0.0), (0.5, -2.0), (1.6, -6.0), (2.3, -8.0), (5.0, -10.0)],      # typical urban
    "RA4": [(0.0, 0.0), (0.2, -2.0), (0.4, -10.0), (0.6, -20.0)],                               # rural area (Rayleigh variant)
    "SFN2": [(0.0, 0.0), (73.2, -3.0)],                                                          # two transmitters of a single-frequency network
    "HT6": [(0.0, 0.0), (0.2, -2.0), (0.4, -4.0), (0.6, -7.0), (15.0, -6.0), (17.2, -12.0)],    # hilly terrain
}


def channel_mobile(iq: np.ndarray, profile="TU6", doppler_hz: float = 50.0, snr_db: float = 20.0, cfo_hz: float = 0.0,
                   timing_offset: int = 0, gain: float = 0.25, seed: int = 0, n_out: int | None = None,
                   clock_ppm: float = 0.0, clock_drift_ppm_per_s: float = 0.0) -> np.ndarray:
    """Time-variant multipath channel: every tap of `profile` (name in MOBILE_PROFILES or a list of (delay_us, power_dB)) is
    an independent Rayleigh process with the Jakes Doppler spectrum (sum of 16 sinusoids with random arrival angles and
    phases, maximum Doppler shift `doppler_hz`; 100 Hz is 480 km/h in Band III), normalised to unit mean total power; then
    a sample clock that is off by `clock_ppm` and drifts by `clock_drift_ppm_per_s` (linear interpolation), CFO, AWGN and
    gain as in channel().  The tap gains are evaluated every 256 samples (8 kHz, far above the Doppler rate) and interpolated
    linearly.  complex64."""
    rng = np.random.default_rng(seed)
    x = np.roll(iq, timing_offset)
    if n_out is not None:
        x = np.tile(x, -(-(n_out + 4096) // len(x)))[: n_out + 4096]
    x = x.astype(np.complex128)
    n = len(x)
    taps = MOBILE_PROFILES[profile] if isinstance(profile, str) else list(profile)
    pw = np.array([10.0 ** (p / 10.0) for _, p in taps])
    pw /= pw.sum()
    step = 256
    tc = np.arange(0, n + step, step, dtype=np.float64) / FS               # coarse time axis
    y = np.zeros(n, np.complex128)
    ti = np.arange(n, dtype=np.float64) / step
    i0 = np.floor(ti).astype(np.int64)
    fr = ti - i0
    for (delay_us, _), p in zip(taps, pw):
        M = 16
        alpha = rng.uniform(0.0, 2.0 * np.pi, M)
        phi = rng.uniform(0.0, 2.0 * np.pi, M)
        g = np.exp(1j * (2.0 * np.pi * doppler_hz * np.cos(alpha)[:, None] * tc[None, :] + phi[:, None])).sum(axis=0) * np.sqrt(p / M)
        gi = g[i0] * (1.0 - fr) + g[i0 + 1] * fr
        d = int(round(delay_us * FS / 1e6))
        y[d:] += gi[d:] * x[: n - d]
    if clock_ppm or clock_drift_ppm_per_s:
        m = n - 4096
        t = np.arange(m, dtype=np.float64)
        t = t * (1.0 + clock_ppm * 1e-6) + 0.5 * clock_drift_ppm_per_s * 1e-6 * t * t / FS
        j0 = np.floor(t).astype(np.int64)
        f2 = t - j0
        y = y[j0] * (1.0 - f2) + y[j0 + 1] * f2
    if n_out is not None:
        y = y[:n_out]
    k = np.arange(len(y), dtype=np.float64)
    y = y * np.exp(2j * np.pi * cfo_hz * k / FS)
    sigma = np.sqrt(10 ** (-snr_db / 10) / 2)
    y = y + sigma * (rng.standard_normal(len(y)) + 1j * rng.standard_normal(len(y)))
    return (gain * y).astype(np.complex64)


def to_raw_u8(iq: np.ndarray) -> np.ndarray:
    """.raw/.iq file format read by the reference as (u8 - 127.38)/128 (raw_reader.cpp:66-70)."""
    v = np.empty(2 * len(iq), np.float64)
    v[0::2], v[1::2] = iq.real, iq.imag
    return np.clip(np.round(v * 128.0 + 127.38), 0, 255).astype(np.uint8)


def from_raw_u8(raw: np.ndarray) -> np.ndarray:
    v = (raw.astype(np.float32) - np.float32(127.38)) / np.float32(128.0)
    return (v[0::2] + 1j * v[1::2]).astype(np.complex64)
