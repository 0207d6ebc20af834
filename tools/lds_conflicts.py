#!/usr/bin/env python3
"""LDS bank-conflict model of the transform (fft_core.h, k_symbols_persistent) and of the demapper's output tile
(pipeline.hip, demap_frame_body), after the rules of MI355X_MICROARCH.md "LDS [CDNA4]":

  ds_read_b64   two groups of 32 lanes, bank = (a / 4) mod 64
  ds_write_b64  four groups of 16 contiguous lanes, bank = (a / 4) mod 32
  ds_write_b8 / ds_write_b32 / ds_read_b32 / ds_read_u8   two groups of 32 lanes, bank = (a / 4) mod 32

Within a group identical dword addresses broadcast (reads) / merge; each extra distinct dword address on a busy bank
costs one more LDS cycle.  The script prints, per access pattern, conflict-free cycles and extra cycles, so that a
candidate layout (fft_pad, tile stride, carrier slot permutation) can be judged before it is built and measured
(SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS on the GPU: profiles/r03_*).

    python tools/lds_conflicts.py                 # current layouts
"""
import sys
from collections import defaultdict


def extra_cycles(addrs_bytes, width, kind):
    """addrs_bytes: 64 byte addresses (None = lane inactive). width: bytes per lane. kind: 'r' / 'w'."""
    if width == 8 and kind == "r":
        groups, mod = [range(0, 32), range(32, 64)], 64
    elif width == 8 and kind == "w":
        groups, mod = [range(16 * g, 16 * g + 16) for g in range(4)], 32
    else:
        groups, mod = [range(0, 32), range(32, 64)], 32
    extra = 0
    for g in groups:
        per_bank = defaultdict(set)
        for lane in g:
            a = addrs_bytes[lane]
            if a is None:
                continue
            for d in range(max(1, width // 4)):
                dw = a // 4 + d
                per_bank[dw % mod].add(dw)
        if per_bank:
            extra += max(len(v) for v in per_bank.values()) - 1
    return extra, len(groups)


def interleaver():
    """freq_interleaver.cpp:40-76 -> bin_to_k[2048] (carrier index per FFT bin, -1 unused)."""
    TU, K = 2048, 1536
    perm, v = [], 0
    tmp = [0] * TU
    for i in range(1, TU):
        tmp[i] = (13 * tmp[i - 1] + 511) % TU
    for i in range(TU):
        if tmp[i] == TU // 2 or tmp[i] < 256 or tmp[i] > 256 + K:
            continue
        perm.append(tmp[i] - TU // 2)
    assert len(perm) == K
    inv = [-1] * TU
    for k, p in enumerate(perm):
        inv[(p + TU) % TU] = k
    return inv


def fft_patterns(pad):
    """(name, kind, per-wave list of 64 float2 indices) for every LDS instruction of one fft2048_regs call."""
    out = []
    for w in range(4):                                   # 4 waves of the 256-thread block
        js = [64 * w + l for l in range(64)]
        for t in range(8):
            out.append(("pass1 write", "w", [pad(8 * j + t) for j in js]))
        for u in range(8):
            out.append(("read strided", "r", [pad(j + 256 * u) for j in js]))
        for t in range(8):
            out.append(("pass2 write", "w", [pad((j - (j & 7)) * 8 + (j & 7) + 8 * t) for j in js]))
        for u in range(8):
            out.append(("read strided", "r", [pad(j + 256 * u) for j in js]))
        for t in range(8):
            out.append(("pass3 write", "w", [pad((j - (j & 63)) * 8 + (j & 63) + 64 * t) for j in js]))
        for h in range(2):
            for q in range(4):
                out.append(("pass4 read", "r", [pad(j + 256 * h + 512 * q) for j in js]))
    return out


def report(title, pats, width):
    tot = defaultdict(lambda: [0, 0, 0])
    for name, kind, idx in pats:
        ex, base = extra_cycles([None if i is None else i * width for i in idx], width, kind)
        tot[name][0] += 1; tot[name][1] += base; tot[name][2] += ex
    print(title)
    a = b = 0
    for name, (n, base, ex) in tot.items():
        print("  %-28s %4d wave-instr  %5d conflict-free LDS cycles  +%5d conflict cycles (%.2fx)" % (name, n, base, ex, (base + ex) / base))
        a += base; b += ex
    print("  %-28s %4s             %5d                          +%5d (%.2fx)" % ("total", "", a, b, (a + b) / a))
    return a, b


def main():
    inv = interleaver()
    pad_old = lambda i: i + (i >> 4)
    pad_new = lambda i: i ^ ((i >> 4) & 7) ^ (((i >> 6) & 1) << 3)
    for nm, pad in (("fft_pad(i) = i + (i >> 4)   [round 2]", pad_old), ("fft_pad(i) = i ^ ((i>>4)&7) ^ (((i>>6)&1)<<3)   [round 3]", pad_new)):
        report("FFT-2048, one block, " + nm, fft_patterns(pad), 8)
    # carrier scatter after the transform + contiguous read-back (k_symbols_persistent)
    for nm, slot in (("lds[kk]", lambda k: k),):
        pats = []
        for w in range(4):
            for u in range(8):
                pats.append(("de-interleave scatter", "w", [None if inv[64 * w + l + 256 * u] < 0 else slot(inv[64 * w + l + 256 * u]) for l in range(64)]))
            for u in range(6):
                pats.append(("carrier read-back", "r", [slot(64 * w + l + 256 * u) for l in range(64)]))
        report("k_symbols: frequency de-interleave through LDS, " + nm, pats, 8)
    # demapper tile: byte writes tpos = plane * stride + (k >> 4), k = tid + 768 q; re at k, im at K + k
    for stride in (192, 196):
        pats = []
        for w in range(12):
            for q in range(2):
                for part in range(2):
                    idx = []
                    for l in range(64):
                        k = 64 * w + l + 768 * q + 1536 * part
                        idx.append((k & 15) * stride + (k >> 4))
                    pats.append(("tile byte writes", "w", idx))
            idx = []
            for l in range(64):
                tid = 64 * w + l
                idx.append((tid // 48) * stride + 4 * (tid % 48))
            pats.append(("tile dword read", "r", idx))
        # width 1: addresses are bytes already
        tot = defaultdict(lambda: [0, 0, 0])
        for name, kind, idx in pats:
            ex, base = extra_cycles(idx, 1 if "byte" in name else 4, kind)
            tot[name][0] += 1; tot[name][1] += base; tot[name][2] += ex
        print("demapper tile, plane stride %d B" % stride)
        for name, (n, base, ex) in tot.items():
            print("  %-28s %4d wave-instr  %5d conflict-free LDS cycles  +%5d conflict cycles (%.2fx)" % (name, n, base, ex, (base + ex) / base))


if __name__ == "__main__":
    sys.exit(main())
