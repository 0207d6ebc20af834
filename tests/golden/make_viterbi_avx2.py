"""Generates tests/golden/ref_viterbi_avx2.npz and ref_viterbi_sse2.npz from the GENUINE reference objects
oracle/_ref/libdabref_vit_{avx2,sse2}.so (the unmodified viterbi_spiral.cpp compiled with -DHAVE_VITERBI_AVX2 -mavx2 /
-DHAVE_VITERBI_SSE2 by oracle/ref/Makefile): seeded soft-bit rows and the bits ViterbiSpiral::deconvolve returns for them.
Build container only; the fixtures are data.

    python tests/golden/make_viterbi_avx2.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def rows_for(n):
    """Soft-bit rows that exercise what differs from the scalar body: ties, saturating symbols, metric saturation and the
    renormalisation (long trellises with large random metrics), next to ordinary noisy code words."""
    rng = np.random.default_rng(16000 + n)
    m = 4 * (n + 6)
    return np.stack([rng.integers(-200, 201, m), rng.integers(-3, 4, m), np.zeros(m), np.full(m, 127), np.full(m, -127),
                     rng.choice([-32768, -32767, 32767, 32640, 32641, -200, 200], m), rng.choice([-127, 0, 127, 128, -128, 1, -1], m),
                     rng.integers(-32768, 32768, m), rng.integers(-60, 61, m)]).astype(np.int16)


def main():
    for variant in ("avx2", "sse2"):
        R = ol.ref_viterbi_variant(variant)
        assert R is not None, "needs oracle/_ref/libdabref_vit_%s.so and a CPU with that instruction set" % variant
        out = {}
        for n in (768, 192, 1536, 9216):
            soft = rows_for(n)
            bits = np.zeros((len(soft), n), np.uint8)
            for i in range(len(soft)):
                R.ref_viterbi(np.ascontiguousarray(soft[i]), n, bits[i])
            out["bits_%d" % n] = np.packbits(bits, axis=1)
        np.savez_compressed(os.path.join(HERE, "ref_viterbi_%s.npz" % variant), **out)
        print("wrote ref_viterbi_%s.npz:" % variant, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
