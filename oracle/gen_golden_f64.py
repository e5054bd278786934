#!/usr/bin/env python3
"""gen_golden_f64.py -- writes tests/golden/f64_2.json from the REAL reference (oracle/_ref/liblfref.so): FFT<Fp2<Fp<1>>>
over p = 2^64 - 2^32 + 1, the second field of the reference's FFT tests and benchmarks (lib/algebra/fft_test.cc:205-229,
BM_FFT_F64_2).  Inputs are the reference's own Bogorng stream (seed in the file, first elements kept to pin it); outputs are
kept whole for small n and by SHA-256 otherwise.  Run in the build container only:
    make -C oracle ref && python oracle/gen_golden_f64.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from oracle_lib import P  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "f64_2.json")


def hx(a):
    return np.ascontiguousarray(a).tobytes().hex()


def main():
    r = ol.ref()
    g = {"about": "reference outputs of FFT<Fp2<Fp<1>>> (p = 2^64 - 2^32 + 1); elements are {re, im} u64 little-endian, Montgomery form",
         "generator": "oracle/gen_golden_f64.py", "fft": [], "binop": []}
    w = np.zeros(2, dtype=np.uint64)
    r.ref_f64_2_omega32(P(w))
    g["omega32"] = hx(w)
    i_elt = np.zeros(2, dtype=np.uint64)
    r.ref_f64_2_of_scalar(0, 1, P(i_elt))
    wc = np.zeros(2, dtype=np.uint64)
    r.ref_f64_2_binop(2, P(w), P(i_elt), P(wc))  # omega * i: a root of order 2^32 outside the base field
    g["omega32_times_i"] = hx(wc)
    a = np.zeros((8, 2), dtype=np.uint64)
    r.ref_f64_2_bogorng_fill(99, 1, 8, P(a))
    out = np.zeros(2, dtype=np.uint64)
    for i in range(4):
        for op in range(4):
            r.ref_f64_2_binop(op, P(a[2 * i]), P(a[2 * i + 1]), P(out))
            g["binop"].append({"op": op, "a": hx(a[2 * i]), "b": hx(a[2 * i + 1]), "out": hx(out)})
    for n in (2, 8, 64, 1024, 1 << 13, 1 << 15, 1 << 17):
        for imag in (0, 1):
            for d in (0, 1):
                for root, wv in (("real", w), ("times_i", wc)):
                    if root == "times_i" and (imag == 0 or n in (2, 64, 1 << 13)):
                        continue
                    x = np.zeros((n, 2), dtype=np.uint64)
                    r.ref_f64_2_bogorng_fill(1000 + n, imag, n, P(x))
                    first = hx(x[:2])
                    r.ref_f64_2_fft(d, n, P(wv), P(x))
                    g["fft"].append({"n": n, "dir": d, "imag": imag, "root": root, "bogorng_seed": 1000 + n, "in_first": first,
                                     "out": hx(x) if n <= 64 else None, "out_sha256": hashlib.sha256(x.tobytes()).hexdigest()})
    with open(OUT, "w") as f:
        json.dump(g, f, indent=0)
    print("wrote", OUT, len(g["fft"]), "fft vectors", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
