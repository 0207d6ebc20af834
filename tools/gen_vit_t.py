#!/usr/bin/env python3
"""Generator for dabstar_amd/csrc/vit_t_gen.h: the lane-per-trellis K=7 r=1/4 Viterbi forward steps.

Scheme (see docs/history/r01-r04_design_notebook.md "Viterbi, lane = trellis"): one lane decodes one trellis; the 64 path metrics live in 32
VGPRs as packed int16 pairs.  Label L (6 bits) = register index (bits 0..4) + half (bit 5); at step t of class
c = t mod 6 label L holds state rotl6(L, c), so the butterfly partner is always L ^ (1 << p), p = (5 - c) mod 6:
for c != 0 a different REGISTER (same half) -> pure packed add/sub/min on register pairs, nothing moves; for
c == 0 the other HALF of the same register (op_sel swap).  Metrics are the reference's doubled and centred
(w = sum +-(2 sym - 255)), int16 with a re-centring every 12 steps (spread <= 14240 + 12*1020 < 32767), decisions
are the sign bits of (upper-path - lower-path), i.e. exactly viterbi_scalar.h's `m0 > m1` with ties -> 0.

This file also contains the pure-Python model of the scheme used to validate it against the oracle
(`python tools/gen_vit_t.py --selftest`)."""
import sys

H = 5


def rotl6(x, c):
    c %= 6
    return ((x << c) | (x >> (6 - c))) & 63


def pat_q(i):
    c0 = ((i >> 1) ^ (i >> 2) ^ (i >> 4)) & 1
    c1 = (i ^ (i >> 1) ^ (i >> 2)) & 1
    c2 = (i ^ (i >> 3)) & 1
    return c0 * 4 + c1 * 2 + c2


def class_plan(c):
    """Returns the static plan of step class c: list of (ra, rb, q_lo, q_hi) register pairs (c != 0) or
    per-register q (c == 0), and the decision bit position of every label."""
    p = (5 - c) % 6
    plan = {"c": c, "p": p}
    if c == 0:
        plan["regs"] = [(r, pat_q(r)) for r in range(32)]          # state = label; butterfly i = r
        dlist = [("m", r) for r in range(32)]                       # merged (dd.lo, ee.hi): labels r (lo), r|32 (hi)
        labels = {("m", r): (r, r | 32) for r in range(32)}
    else:
        pairs = []
        for ra in range(32):
            if (ra >> p) & 1:
                continue
            rb = ra | (1 << p)
            pairs.append((ra, rb, pat_q(rotl6(ra, c) & 31), pat_q(rotl6(ra | 32, c) & 31)))
        plan["pairs"] = pairs
        dlist, labels = [], {}
        for k, (ra, rb, _, _) in enumerate(pairs):
            dlist += [("a", k), ("b", k)]
            labels[("a", k)] = (ra, ra | 32)
            labels[("b", k)] = (rb, rb | 32)
    # c != 0: perm j takes S0 = a-difference, S1 = b-difference of pair j; result bytes [S1.lo, S1.hi, S0.lo, S0.hi] as
    # 0x00 / 0xFF, folded with mask 0x01010101 << (j % 8) into word j // 8.  c == 0: register r goes to bytes (0, 1)
    # (r even) or (2, 3) (r odd) = labels (r, r | 32) of perm j = r // 2.
    pos = [None] * 64
    if c == 0:
        for r in range(32):
            j, base = r // 2, 2 * (r % 2)
            pos[r] = 32 * (j // 8) + 8 * base + j % 8
            pos[r | 32] = 32 * (j // 8) + 8 * (base + 1) + j % 8
    else:
        for j in range(16):
            a, k = j // 8, j % 8
            s0, s1 = dlist[2 * j], dlist[2 * j + 1]
            for b, lab in enumerate((labels[s1][0], labels[s1][1], labels[s0][0], labels[s0][1])):
                pos[lab] = 32 * a + 8 * b + k
    plan["dlist"] = dlist
    plan["pos"] = pos
    return plan


PLANS = [class_plan(c) for c in range(6)]


# ------------------------------------------------------------------------------------------- python model
def model_decode(soft, nbits):
    """Bit-level model of the generated kernel for ONE trellis (numpy int16 semantics emulated with ints)."""
    import numpy as np
    nst = nbits + 6
    sym = np.clip((soft.astype(np.int64) + 127).astype(np.int16).astype(np.int64), 0, 255).reshape(nst, 4)
    xs = 2 * sym - 255

    def wrap(v):
        return ((v + 32768) & 0xFFFF) - 32768

    R = [[2000, 2000] for _ in range(32)]
    R[0][0] = 0
    dec = []
    for t in range(nst):
        c = t % 6
        if c == 0 and (t // 6) % 2 == 0:
            ref = R[0][0]
            R = [[wrap(a - ref), wrap(b - ref)] for a, b in R]
        y0, x1, x2 = xs[t, 0] + xs[t, 3], xs[t, 1], xs[t, 2]
        W = [(1 - 2 * ((q >> 2) & 1)) * y0 + (1 - 2 * ((q >> 1) & 1)) * x1 + (1 - 2 * (q & 1)) * x2 for q in range(8)]
        pl = PLANS[c]
        word = 0
        if c == 0:
            for r, q in pl["regs"]:
                lo, hi = R[r]
                w = W[q]
                t1 = (wrap(lo + w), wrap(hi + w)); t2 = (wrap(lo - w), wrap(hi - w))
                new = (min(t1[0], t2[1]), min(t1[1], t2[0]))
                d_lo, d_hi = wrap(t2[1] - t1[0]), wrap(t1[1] - t2[0])
                assert abs(lo + w) < 32768 and abs(hi + w) < 32768
                if d_lo < 0: word |= 1 << pl["pos"][r]
                if d_hi < 0: word |= 1 << pl["pos"][r | 32]
                R[r] = list(new)
        else:
            for ra, rb, ql, qh in pl["pairs"]:
                A, B = R[ra], R[rb]
                M = (W[ql], W[qh])
                a0 = [wrap(A[h] + M[h]) for h in (0, 1)]; b0 = [wrap(B[h] - M[h]) for h in (0, 1)]
                a1 = [wrap(A[h] - M[h]) for h in (0, 1)]; b1 = [wrap(B[h] + M[h]) for h in (0, 1)]
                for h in (0, 1):
                    if wrap(b0[h] - a0[h]) < 0: word |= 1 << pl["pos"][ra | (32 * h)]
                    if wrap(b1[h] - a1[h]) < 0: word |= 1 << pl["pos"][rb | (32 * h)]
                R[ra] = [min(a0[h], b0[h]) for h in (0, 1)]
                R[rb] = [min(a1[h], b1[h]) for h in (0, 1)]
        dec.append(word)
    L = 0
    out = np.zeros(nbits, np.uint8)
    for t in range(nst - 1, 5, -1):
        c = t % 6
        k = (dec[t] >> PLANS[c]["pos"][L]) & 1
        out[t - 6] = k
        p = PLANS[c]["p"]
        L = (L & ~(1 << p)) | (k << p)
    return out


# ---- the arithmetic of the reference's SIMD builds on the same register scheme (k_msc_vitT_tie, vit_t.hip) -------------
# VITERBI_AVX2 (tie = 1, viterbi_16way.h): uint16 metrics, saturating adds (65535), a tie goes to predecessor i + 32,
# `renormalize` after every second step: if state 0's metric of the step BEFORE exceeds 60000 the minimum of the new metrics
# is subtracted.  VITERBI_SSE2 / NEON (tie = 2, viterbi_8way.h): int16 metrics saturating at 32767, threshold 30000, ties as
# in the scalar body.  The lane keeps its relative, doubled and centred int16 metrics R (2 M_ref = R + C) and follows the
# reference's ABSOLUTE level with one int per lane: Coff = C before the first step of the current 6-step cycle (C grows by
# 1020 per step, by `ref` at a re-centring, and becomes -min(R) at a renormalisation).  A candidate saturates when it
# exceeds lim = LIMTOP - C; that can only happen when max(R) + 12240 > LIMTOP - Coff at the start of the cycle, so the
# clamped step body (four more v_pk_min per butterfly pair) runs only then.
TIE_CONST = {1: (2 * 65535, 2 * 60000), 2: (2 * 32767, 2 * 30000)}


def model_decode_tie(soft, nbits, tie, always_clamp=False, stats=None):
    import numpy as np
    LIMTOP, REN2 = TIE_CONST[tie]
    nst = nbits + 6
    sym = np.clip(soft.astype(np.int64) + 127, 0, 255).reshape(nst, 4)          # saturating +127 (viterbi_16way.h:73-76)
    xs = 2 * sym - 255

    def wrap(v):
        return ((v + 32768) & 0xFFFF) - 32768

    R = [[2000, 2000] for _ in range(32)]
    R[0][0] = 0
    Coff = 0
    dec = []
    n_slow = n_ren = 0
    for t0 in range(0, nst, 6):
        if (t0 // 6) % 2 == 0:
            ref = R[0][0]
            R = [[wrap(a - ref), wrap(b - ref)] for a, b in R]
            Coff += ref
        mx = max(max(a, b) for a, b in R)
        clamp_cycle = always_clamp or (mx + 12240 > LIMTOP - Coff)
        n_slow += clamp_cycle
        for j in range(6):
            t = t0 + j
            pre = (j & 1) and (R[0][0] + Coff + 1020 * j > REN2)
            lim = min(LIMTOP - Coff - 1020 * (j + 1), 32767)
            y0, x1, x2 = xs[t, 0] + xs[t, 3], xs[t, 1], xs[t, 2]
            W = [(1 - 2 * ((q >> 2) & 1)) * y0 + (1 - 2 * ((q >> 1) & 1)) * x1 + (1 - 2 * (q & 1)) * x2 for q in range(8)]
            pl = PLANS[j]
            cl = (lambda v: min(v, lim)) if clamp_cycle else (lambda v: v)
            word = 0

            def decide(a, b, lab):                       # a: path through predecessor i, b: through i + 32
                nonlocal word
                d = (wrap(a - b) >= 0) if tie == 1 else (wrap(b - a) < 0)       # tie 1: NOT (a < b); tie 2: b < a
                if d:
                    word |= 1 << pl["pos"][lab]
            if j == 0:
                for r, q in pl["regs"]:
                    lo, hi = R[r]
                    w = W[q]
                    t1 = (cl(wrap(lo + w)), cl(wrap(hi + w))); t2 = (cl(wrap(lo - w)), cl(wrap(hi - w)))
                    decide(t1[0], t2[1], r)
                    decide(t2[0], t1[1], r | 32)
                    R[r] = [min(t1[0], t2[1]), min(t1[1], t2[0])]
            else:
                for ra, rb, ql, qh in pl["pairs"]:
                    A, B = R[ra], R[rb]
                    M = (W[ql], W[qh])
                    a0 = [cl(wrap(A[h] + M[h])) for h in (0, 1)]; b0 = [cl(wrap(B[h] - M[h])) for h in (0, 1)]
                    a1 = [cl(wrap(A[h] - M[h])) for h in (0, 1)]; b1 = [cl(wrap(B[h] + M[h])) for h in (0, 1)]
                    for h in (0, 1):
                        decide(a0[h], b0[h], ra | (32 * h))
                        decide(a1[h], b1[h], rb | (32 * h))
                    R[ra] = [min(a0[h], b0[h]) for h in (0, 1)]
                    R[rb] = [min(a1[h], b1[h]) for h in (0, 1)]
            dec.append(word)
            if pre:
                mn = min(min(a, b) for a, b in R)
                Coff = -mn - 1020 * (j + 1)
                n_ren += 1
        Coff += 6120
    if stats is not None:
        stats["slow_cycles"] = n_slow; stats["cycles"] = nst // 6; stats["renorms"] = n_ren
    L = 0
    out = np.zeros(nbits, np.uint8)
    for t in range(nst - 1, 5, -1):
        c = t % 6
        k = (dec[t] >> PLANS[c]["pos"][L]) & 1
        out[t - 6] = k
        p = PLANS[c]["p"]
        L = (L & ~(1 << p)) | (k << p)
    return out


def selftest_tie():
    import os
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import oracle_lib as ol
    rng = np.random.default_rng(1)
    ora = {1: ol.ora_viterbi_simd, 2: ol.ora_viterbi_sse2}
    for tie in (1, 2):
        slow = tot = ren = 0
        for n in (768, 1536, 192):
            for case in range(8):
                m = 4 * (n + 6)
                soft = [rng.integers(-200, 201, m), rng.integers(-40, 41, m), rng.choice([-127, 0, 127, 128, -128, 1, -1], m),
                        rng.choice([-32768, 32767, 32640, -200, 200, 0], m), np.zeros(m),
                        rng.choice([-127, 127], m), np.where(rng.random(m) < 0.5, 0, rng.choice([-127, 127], m)),     # full-scale: saturation
                        np.where(np.arange(m) % 4 < 2, rng.choice([-120, 120], m), 0)][case].astype(np.int16)       # rate 1/2 puncturing pattern
                st = {}
                a = model_decode_tie(soft, n, tie, stats=st)
                b = model_decode_tie(soft, n, tie, always_clamp=True)
                want = ora[tie](soft, n)
                assert np.array_equal(a, want), (tie, n, case, "conditional clamp")
                assert np.array_equal(b, want), (tie, n, case, "always clamp")
                slow += st["slow_cycles"]; tot += st["cycles"]; ren += st["renorms"]
        print("tie mode %d: lane-per-trellis model == oracle (%d of %d cycles clamped, %d renormalisations)" % (tie, slow, tot, ren))


def selftest():
    import os
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import oracle_lib as ol
    rng = np.random.default_rng(0)
    for n in (768, 1536, 192):
        for case in range(5):
            m = 4 * (n + 6)
            soft = [rng.integers(-200, 201, m), rng.integers(-40, 41, m), rng.choice([-127, 0, 127, 128, -128, 1, -1], m),
                    rng.choice([-32768, 32767, 32640, -200, 200, 0], m), np.zeros(m)][case].astype(np.int16)
            assert np.array_equal(model_decode(soft, n), ol.ora_viterbi(soft, n)), (n, case)
    print("lane-per-trellis model == oracle")
    assert bm_check()
    print("packed branch-metric formulas == W[q]")



# ------------------------------------------------------------------------------------------- packed branch metrics
# The four packed branch-metric registers of a step, M[i] = (W[ql], W[qh]), are formed from the step's four symbol bytes
# with packed 16-bit arithmetic only (W[q] = (1-2c0) y + (1-2c1) x1 + (1-2c2) x2, y = x0 + x3, x = 2 sym - 255).  Atoms:
#   A = (s0, s3), B = (s1, s2), U1 = (s1, s1), U2 = (s2, s2)                 one v_perm_b32 each (wave-uniform selector)
#   TS = A + A.swap = (s0 + s3) twice;  Y2 = 2 TS - 510 = (y, y);  YN = (y, -y)             v_pk_add / v_pk_mad_i16
#   X1 = (x1, x1), X1N = (x1, -x1), X2, X2N likewise, XB = (x1, x2)                          v_pk_mad_i16
# and per class two sums of those plus the four final adds / subs -- 11 to 13 instructions per step where the scalar
# formulation (four byte extractions, eight 32-bit sums, four packs) took 27.  BM_PLAN[c] = (atoms, intermediates, {(ql, qh): expr}).
BM_PLAN = {
    0: (["A", "B", "TS", "Y2", "XB"], [("S", "swap_add(XB)"), ("D", "swap_sub(XB)")],
        {(0, 0): "Y2 + S", (3, 3): "Y2 - S", (1, 1): "add_lolo(Y2, D)", (2, 2): "add_hihi(Y2, D)"}),
    1: (["A", "U1", "U2", "TS", "Y2", "X1N", "X2N"], [("SN", "X1N + X2N"), ("DN", "X1N - X2N")],
        {(0, 3): "Y2 + SN", (3, 0): "Y2 - SN", (1, 2): "Y2 + DN", (2, 1): "Y2 - DN"}),
    2: (["A", "U1", "U2", "TS", "YN", "X1N", "X2"], [("PN", "YN + X1N"), ("QN", "YN - X1N")],
        {(0, 6): "PN + X2", (1, 7): "PN - X2", (3, 5): "QN - X2", (2, 4): "QN + X2"}),
    4: (["A", "U1", "U2", "TS", "Y2", "X1", "X2N"], [("PP", "Y2 + X1"), ("QQ", "Y2 - X1")],
        {(0, 1): "PP + X2N", (1, 0): "PP - X2N", (3, 2): "QQ - X2N", (2, 3): "QQ + X2N"}),
    5: (["A", "U1", "U2", "TS", "YN", "X1", "X2"], [("SS", "X1 + X2"), ("DD", "X1 - X2")],
        {(0, 4): "YN + SS", (3, 7): "YN - SS", (1, 5): "YN + DD", (2, 6): "YN - DD"}),
}
BM_PLAN[3] = BM_PLAN[2]


def bm_check():
    """Evaluates every formula of BM_PLAN on random symbols with int16 wrap-around and compares with W[q]."""
    import random

    class P:                                                     # a packed pair of wrapped int16
        def __init__(self, lo, hi):
            w = lambda v: ((v + 32768) & 0xFFFF) - 32768
            self.lo, self.hi = w(lo), w(hi)
        def __add__(self, o): return P(self.lo + o.lo, self.hi + o.hi)
        def __sub__(self, o): return P(self.lo - o.lo, self.hi - o.hi)
    env_fn = {"swap_add": lambda a: P(a.lo + a.hi, a.hi + a.lo), "swap_sub": lambda a: P(a.lo - a.hi, a.hi - a.lo),
              "add_lolo": lambda a, b: P(a.lo + b.lo, a.hi + b.lo), "add_hihi": lambda a, b: P(a.lo + b.hi, a.hi + b.hi)}
    rnd = random.Random(1)
    for _ in range(200):
        s0, s1, s2, s3 = (rnd.choice([0, 255, 127, rnd.randrange(256)]) for _ in range(4))
        x = [2 * v - 255 for v in (s0, s1, s2, s3)]
        y = x[0] + x[3]
        W = [(1 - 2 * ((q >> 2) & 1)) * y + (1 - 2 * ((q >> 1) & 1)) * x[1] + (1 - 2 * (q & 1)) * x[2] for q in range(8)]
        ts = s0 + s3
        env = dict(env_fn, A=P(s0, s3), B=P(s1, s2), U1=P(s1, s1), U2=P(s2, s2), TS=P(ts, ts), Y2=P(2 * ts - 510, 2 * ts - 510),
                   YN=P(2 * ts - 510, 510 - 2 * ts), X1=P(x[1], x[1]), X1N=P(x[1], -x[1]), X2=P(x[2], x[2]), X2N=P(x[2], -x[2]),
                   XB=P(x[1], x[2]))
        for c, (atoms, inter, finals) in BM_PLAN.items():
            e = dict(env)
            for name, expr in inter:
                e[name] = eval(expr, {}, e)
            for (ql, qh), expr in finals.items():
                r = eval(expr, {}, e)
                assert (r.lo, r.hi) == (W[ql], W[qh]), (c, ql, qh, expr)
    return True

# ------------------------------------------------------------------------------------------- code emitter
def emit():
    o = []
    A = o.append
    A("// vit_t_gen.h -- GENERATED by tools/gen_vit_t.py; do not edit.  Lane-per-trellis Viterbi forward steps.")
    A("// One lane = one trellis; R[32] = 64 path metrics as packed int16 pairs (label = register | half << 5).")
    A("#pragma once")
    A("#include <hip/hip_runtime.h>")
    A("namespace dabx { namespace vt {")
    A("typedef short s2 __attribute__((ext_vector_type(2)));")
    A("__device__ __forceinline__ unsigned u(s2 v) { return __builtin_bit_cast(unsigned, v); }")
    A("__device__ __forceinline__ s2 s(unsigned v) { return __builtin_bit_cast(s2, v); }")
    A("__device__ __forceinline__ s2 pk(int lo, int hi) { return s(__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u)); }")
    A("__device__ __forceinline__ s2 mn(s2 a, s2 b) { return __builtin_elementwise_min(a, b); }")
    A("__device__ __forceinline__ s2 mx(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }")
    A("// smallest / largest of the 64 path metrics of the lane's trellis (tie modes: renormalisation, saturation test)")
    A("__device__ __forceinline__ int min64(const s2 (&R)[32])")
    A("{ s2 a[16];")
    A("  for (int i = 0; i < 16; i++) a[i] = mn(R[i], R[i + 16]);")
    A("  for (int w = 8; w > 0; w >>= 1) for (int i = 0; i < w; i++) a[i] = mn(a[i], a[i + w]);")
    A("  return a[0].x < a[0].y ? (int)a[0].x : (int)a[0].y; }")
    A("__device__ __forceinline__ int max64(const s2 (&R)[32])")
    A("{ s2 a[16];")
    A("  for (int i = 0; i < 16; i++) a[i] = mx(R[i], R[i + 16]);")
    A("  for (int w = 8; w > 0; w >>= 1) for (int i = 0; i < w; i++) a[i] = mx(a[i], a[i + w]);")
    A("  return a[0].x > a[0].y ? (int)a[0].x : (int)a[0].y; }")
    A("// VOP3P op_sel forms (class 0: the butterfly partner is the other half of the same register)")
    A("__device__ __forceinline__ s2 mn_x(s2 a, s2 b)      // (min(a.lo, b.hi), min(a.hi, b.lo))")
    A("{ s2 r; asm(\"v_pk_min_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\" : \"=v\"(r) : \"v\"(a), \"v\"(b)); return r; }")
    A("__device__ __forceinline__ s2 sub_hl(s2 a, s2 b)    // (a.hi - b.lo, a.lo - b.hi)")
    A("{ s2 r; asm(\"v_pk_sub_i16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]\" : \"=v\"(r) : \"v\"(a), \"v\"(b)); return r; }")
    A("__device__ __forceinline__ s2 sub_lh(s2 a, s2 b)    // (a.lo - b.hi, a.hi - b.lo)")
    A("{ s2 r; asm(\"v_pk_sub_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\" : \"=v\"(r) : \"v\"(a), \"v\"(b)); return r; }")
    A("// v_perm_b32 selectors 8..11 = sign of S1.lo, S1.hi, S0.lo, S0.hi replicated over the byte (0x00 / 0xFF); 12 = 0x00")
    A("__device__ __forceinline__ unsigned sg(s2 s0, s2 s1) { return __builtin_amdgcn_perm(u(s0), u(s1), 0x0B0A0908u); }")
    A("__device__ __forceinline__ unsigned sg_even(s2 dd, s2 ee) { return __builtin_amdgcn_perm(u(dd), u(ee), 0x0C0C090Au); }   // [dd.lo, ee.hi, 0, 0]")
    A("__device__ __forceinline__ unsigned sg_odd(s2 dd, s2 ee) { return __builtin_amdgcn_perm(u(dd), u(ee), 0x090A0C0Cu); }    // [0, 0, dd.lo, ee.hi]")
    A("template <int K> __device__ __forceinline__ void fold(unsigned &acc, unsigned p)")
    A("{ constexpr unsigned M = 0x01010101u << K;   // one v_and_or_b32 per gather (hipcc splits the and/or otherwise)")
    A("  if (K == 0) acc = p & M;")
    A("  else asm(\"v_and_or_b32 %0, %1, %2, %0\" : \"+v\"(acc) : \"v\"(p), \"s\"(M)); }")
    A("// ---- packed branch metrics (tools/gen_vit_t.py, BM_PLAN): VOP3P forms the compiler does not pick by itself")
    A("__device__ __forceinline__ s2 swap_add(s2 a)        // (a.lo + a.hi) in both halves")
    A("{ s2 r; asm(\"v_pk_add_u16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]\" : \"=v\"(r) : \"v\"(a)); return r; }")
    A("__device__ __forceinline__ s2 swap_sub(s2 a) { return sub_lh(a, a); }          // (a.lo - a.hi, a.hi - a.lo)")
    A("__device__ __forceinline__ s2 add_lolo(s2 a, s2 b)  // (a.lo + b.lo, a.hi + b.lo)")
    A("{ s2 r; asm(\"v_pk_add_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]\" : \"=v\"(r) : \"v\"(a), \"v\"(b)); return r; }")
    A("__device__ __forceinline__ s2 add_hihi(s2 a, s2 b)  // (a.lo + b.hi, a.hi + b.hi)")
    A("{ s2 r; asm(\"v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]\" : \"=v\"(r) : \"v\"(a), \"v\"(b)); return r; }")
    A("__device__ __forceinline__ s2 mad2(s2 a, unsigned k)             // 2 a + k (k: packed constant in an SGPR)")
    A("{ s2 r; asm(\"v_pk_mad_i16 %0, %1, 2, %2 op_sel_hi:[1,0,1]\" : \"=v\"(r) : \"v\"(a), \"s\"(k)); return r; }")
    A("__device__ __forceinline__ s2 madv(s2 a, s2 m, unsigned k)       // a * m + k per half (m = (2, -2) kept in a VGPR)")
    A("{ s2 r; asm(\"v_pk_mad_i16 %0, %1, %2, %3\" : \"=v\"(r) : \"v\"(a), \"v\"(m), \"s\"(k)); return r; }")
    A("// symbol bytes b_lo of dword w_lo and b_hi of dword w_hi as a zero-extended pair (wave-uniform byte lanes -> SGPR selector)")
    A("__device__ __forceinline__ s2 pick(unsigned w_lo, unsigned b_lo, unsigned w_hi, unsigned b_hi)")
    A("{ return s(__builtin_amdgcn_perm(w_hi, w_lo, 0x0C000C00u | b_lo | ((4u + b_hi) << 16))); }")
    A("constexpr unsigned K_M255 = 0xFF01FF01u, K_M255_P255 = 0x00FFFF01u, K_M510 = 0xFE02FE02u, K_M510_P510 = 0x01FEFE02u;")
    A("")
    A("// exchange bit of each step class and decision bit position of every label (chain-back tables)")
    A("__device__ constexpr unsigned char VT_P[6] = {%s};" % ", ".join(str(pl["p"]) for pl in PLANS))
    A("__device__ constexpr unsigned char VT_POS[6][64] = {")
    for pl in PLANS:
        A("  {%s}," % ", ".join(str(x) for x in pl["pos"]))
    A("};")
    A("")
    A("// M[i]: packed branch metrics (W[ql], W[qh]) of the step's four register-pair groups; W[q], q = c0*4 + c1*2 + c2, is")
    A("// (1-2c0) y0 + (1-2c1) x1 + (1-2c2) x2 with x = 2 sym - 255, y0 = x0 + x3.  w[p] / b[p]: the dword holding symbol p and")
    A("// its byte lane; v2n = (2, -2).")
    atom_code = {
        "A": "const s2 A = pick(w[0], b[0], w[3], b[3]);", "B": "const s2 B = pick(w[1], b[1], w[2], b[2]);",
        "U1": "const s2 U1 = pick(w[1], b[1], w[1], b[1]);", "U2": "const s2 U2 = pick(w[2], b[2], w[2], b[2]);",
        "TS": "const s2 TS = swap_add(A);", "Y2": "const s2 Y2 = mad2(TS, K_M510);", "YN": "const s2 YN = madv(TS, v2n, K_M510_P510);",
        "X1": "const s2 X1 = mad2(U1, K_M255);", "X1N": "const s2 X1N = madv(U1, v2n, K_M255_P255);",
        "X2": "const s2 X2 = mad2(U2, K_M255);", "X2N": "const s2 X2N = madv(U2, v2n, K_M255_P255);",
        "XB": "const s2 XB = mad2(B, K_M255);"}
    for pl in PLANS:
        c = pl["c"]
        atoms, inter, finals = BM_PLAN[c]
        if c == 0:
            order = [(q, q) for q in sorted(set(min(q, 7 - q) for _, q in pl["regs"]))]
        else:
            combos = {}
            for ra, rb, ql, qh in pl["pairs"]:
                key = (ql, qh) if ql < 4 else (7 - ql, 7 - qh)
                combos.setdefault(key, len(combos))
            order = [k for k, _ in sorted(combos.items(), key=lambda kv: kv[1])]
        A("__device__ __forceinline__ void bm%d(const unsigned (&w)[4], const unsigned (&b)[4], s2 v2n, s2 (&M)[4])" % c)
        A("{")
        for a in atoms:
            A("  " + atom_code[a])
        for name, expr in inter:
            A("  const s2 %s = %s;" % (name, expr))
        for i, key in enumerate(order):
            A("  M[%d] = %s;   // (W[%d], W[%d])" % (i, finals[key], key[0], key[1]))
        A("}")
        # TIE: 0 = scalar body (decision = b < a, a tie keeps predecessor i); 1 = VITERBI_AVX2 (decision = NOT (a < b): a tie goes to
        # i + 32; the words are inverted at the end of the step); 2 = VITERBI_SSE2 (scalar tie rule).  CLAMP: every candidate is
        # limited to `lim` first = the saturating adds of the SIMD builds at the lane's current absolute level (vit_t.hip).
        A("template <int TIE = 0, bool CLAMP = false>")
        A("__device__ __forceinline__ void step%d(s2 (&R)[32], const s2 (&M)[4], unsigned &acc0, unsigned &acc1, s2 lim = s2{0, 0})" % c)
        A("{")
        if c == 0:
            qs = [q for q, _ in order]
            for i, q in enumerate(qs):
                A("  const s2 M%d = M[%d];" % (q, i))
            for j in range(16):
                names = []
                for r in (2 * j, 2 * j + 1):
                    q = pl["regs"][r][1]
                    cq, flip = (q, False) if q < 4 else (7 - q, True)
                    # t1 = (a0, b1), t2 = (a1, b0): new = (min(a0, b0), min(b1, a1)), d0 = b0 - a0, d1 = b1 - a1
                    A("  s2 t1_%d = R[%d] %s M%d, t2_%d = R[%d] %s M%d;" % (r, r, "-" if flip else "+", cq, r, r, "+" if flip else "-", cq))
                    A("  if constexpr (CLAMP) { t1_%d = mn(t1_%d, lim); t2_%d = mn(t2_%d, lim); }" % (r, r, r, r))
                    A("  R[%d] = mn_x(t1_%d, t2_%d);" % (r, r, r))
                    names.append((r, "sub_hl(t2_%d, t1_%d)" % (r, r), "sub_lh(t1_%d, t2_%d)" % (r, r)))
                A("  if constexpr (TIE == 1) fold<%d>(acc%d, sg_even(%s, %s) | sg_odd(%s, %s));" %
                  (j % 8, j // 8, names[0][2], names[0][1], names[1][2], names[1][1]))
                A("  else fold<%d>(acc%d, sg_even(%s, %s) | sg_odd(%s, %s));" % (j % 8, j // 8, names[0][1], names[0][2], names[1][1], names[1][2]))
        else:
            for (ql, qh), idx in combos.items():
                A("  const s2 M%d = M[%d];" % (idx, idx))
            for k, (ra, rb, ql, qh) in enumerate(pl["pairs"]):
                flip = ql >= 4
                key = (ql, qh) if not flip else (7 - ql, 7 - qh)
                m = "M%d" % combos[key]
                pa, pb = ("-", "+") if flip else ("+", "-")
                A("  { s2 a0 = R[%d] %s %s, b0 = R[%d] %s %s, a1 = R[%d] %s %s, b1 = R[%d] %s %s;" %
                  (ra, pa, m, rb, pb, m, ra, pb, m, rb, pa, m))
                A("    if constexpr (CLAMP) { a0 = mn(a0, lim); b0 = mn(b0, lim); a1 = mn(a1, lim); b1 = mn(b1, lim); }")
                A("    R[%d] = mn(a0, b0); R[%d] = mn(a1, b1);" % (ra, rb))
                A("    if constexpr (TIE == 1) fold<%d>(acc%d, sg(a0 - b0, a1 - b1)); else fold<%d>(acc%d, sg(b0 - a0, b1 - a1)); }" % (k % 8, k // 8, k % 8, k // 8))
        A("  if constexpr (TIE == 1) { acc0 = ~acc0; acc1 = ~acc1; }")
        A("}")
        A("")
    A("}}  // namespace dabx::vt")
    return "\n".join(o) + "\n"


if __name__ == "__main__" and "--emit" in sys.argv:
    import os
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dabstar_amd", "csrc", "vit_t_gen.h")
    open(out, "w").write(emit())
    print("wrote", os.path.normpath(out))
    for pl in PLANS[1:]:
        combos = set((ql, qh) if ql < 4 else (7 - ql, 7 - qh) for _, _, ql, qh in pl["pairs"])
        print("class", pl["c"], "p", pl["p"], "distinct packed M:", len(combos))

if __name__ == "__main__" and "--selftest" in sys.argv:
    selftest()
    selftest_tie()
