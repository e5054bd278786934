#!/usr/bin/env python3
"""gen_golden.py -- writes tests/golden/ref_vectors.json from the REAL reference
(oracle/_ref/liblfref.so, compiled from /root/reference by oracle/Makefile).

Run in the build container only (`make -C oracle ref && python oracle/gen_golden.py`).
The vectors are data (inputs + expected outputs as hex); large outputs are pinned by
their SHA-256.  The GPU box has no reference: there these vectors pin the oracle.
Also copies the reference's own binary test fixtures (data files its Rust tests hold)."""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from oracle_lib import FP, GF, P  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def hx(a):
    return np.ascontiguousarray(a).tobytes().hex()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    r = ol.ref()
    assert r is not None, "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(20261004)
    g = {"_generator": "oracle/gen_golden.py over oracle/_ref (reference @ /root/reference)"}

    # field ops
    for name, field, fns in (("gf", GF, {"mul": r.ref_gf_mul}),
                             ("fp", FP, {"mul": r.ref_fp_mul, "add": r.ref_fp_add, "sub": r.ref_fp_sub})):
        xs, ys = ol.rand_elts(rng, 48, field), ol.rand_elts(rng, 48, field)
        g[name + "_x"], g[name + "_y"] = hx(xs), hx(ys)
        for op, fn in fns.items():
            out = np.zeros_like(xs)
            for i in range(48):
                fn(P(xs[i]), P(ys[i]), P(out[i]))
            g["%s_%s" % (name, op)] = hx(out)
    out = np.zeros(2, dtype=np.uint64)
    for k in (4, 5):
        b = np.zeros((1 << k, 2), dtype=np.uint64)
        for i in range(1 << k):
            r.ref_gf_beta(k, i, P(b[i]))
        g["gf_beta_k%d" % k] = hx(b)
        pts = np.zeros((6, 2), dtype=np.uint64)
        for i in range(6):
            r.ref_gf_poly_evaluation_point(k, i, P(pts[i]))
        g["gf_eval_points_k%d" % k] = hx(pts)
    r.ref_fp_omega32(P(out))
    g["fp_omega32_mont"] = hx(out)

    # LCH14 FFT / IFFT / bidirectional / RS
    lch = []
    for k, l, coset, d in [(4, 6, 64, 0), (4, 6, 64, 1), (4, 10, 3 << 10, 0), (5, 12, 1 << 12, 0), (4, 7, 50, 2), (4, 7, 128, 2),
                           (4, 7, 0, 2), (4, 9, 300, 2)]:
        a = ol.rand_elts(rng, 1 << l)
        y = a.copy()
        r.ref_lch14_fft(k, d, l, coset, P(y))
        lch.append({"k": k, "l": l, "coset_or_k": coset, "dir": d, "in": hx(a) if l <= 7 else None,
                    "in_seed": None if l <= 7 else int(l * 1000 + coset), "out_sha256": sha(y),
                    "out": hx(y) if l <= 7 else None})
        if l > 7:  # regenerate deterministically from the recorded seed
            a2 = ol.rand_elts(np.random.default_rng(lch[-1]["in_seed"]), 1 << l)
            y2 = a2.copy()
            r.ref_lch14_fft(k, d, l, coset, P(y2))
            lch[-1]["out_sha256"] = sha(y2)
    g["lch14"] = lch
    rs = []
    for k, n, m in [(4, 21, 128), (4, 455, 4096), (4, 909, 4096), (4, 910, 8192), (4, 1819, 8192), (4, 682, 4096), (5, 1000, 5000)]:
        seed = n * 10000 + m
        a = ol.rand_elts(np.random.default_rng(seed), m)
        y = a.copy()
        r.ref_lch14_rs_interpolate(k, n, m, P(y))
        rs.append({"k": k, "n": n, "m": m, "seed": seed, "out_sha256": sha(y)})
    g["lch14_rs"] = rs

    # Fp128 FFT (Bogorng input, true 2^32-order root) incl. the recursive path n > 16384
    fft = []
    for n, d in [(8, 0), (8, 1), (1024, 0), (1 << 15, 0), (1 << 15, 1), (1 << 17, 0)]:
        a = np.zeros((n, 2), dtype=np.uint64)
        r.ref_fp_bogorng_fill(1234569 + n, n, P(a))
        y = a.copy()
        r.ref_fp_fft(d, n, P(y))
        fft.append({"n": n, "dir": d, "bogorng_seed": 1234569 + n, "out_sha256": sha(y), "out": hx(y) if n <= 8 else None,
                    "in_first": hx(a[:2])})
    g["fp_fft"] = fft
    frs = []
    for n, m in [(3, 8), (21, 128), (455, 4096)]:
        a = np.zeros((m, 2), dtype=np.uint64)
        r.ref_fp_bogorng_fill(77 + n, m, P(a))
        y = a.copy()
        r.ref_fp_rs_interpolate(n, m, P(y))
        frs.append({"n": n, "m": m, "bogorng_seed": 77 + n, "out_sha256": sha(y)})
    g["fp_rs"] = frs

    # Merkle + column commit
    mk = []
    for n in (1, 2, 5, 1000):
        leaves = np.random.default_rng(n).integers(0, 256, size=(n, 32), dtype=np.uint8)
        lay = np.zeros((2 * n, 32), dtype=np.uint8)
        r.ref_merkle_build_tree(n, P(leaves), P(lay))
        mk.append({"n": n, "seed": n, "root": hx(lay[1]), "layers_sha256": sha(lay[1:])})
    g["merkle"] = mk
    cc = []
    for field, nrow, ld, col0, ncols in [(GF, 20, 4096, 909, 3187), (GF, 7, 64, 13, 51), (FP, 5, 40, 9, 31)]:
        seed = nrow * ld + field
        rg = np.random.default_rng(seed)
        T = ol.rand_elts(rg, nrow * ld, field)
        nonces = rg.integers(0, 256, size=(ncols, 32), dtype=np.uint8)
        root = np.zeros(32, dtype=np.uint8)
        r.ref_column_commit(field, nrow, ld, col0, ncols, P(T), P(nonces), P(root))
        cc.append({"field": field, "nrow": nrow, "ld": ld, "col0": col0, "ncols": ncols, "seed": seed, "root": hx(root)})
    g["column_commit"] = cc

    # sumcheck round pieces
    sc = []
    for field in (GF, FP):
        for n in (2, 17, 1000):
            seed = n + field
            rg = np.random.default_rng(seed)
            QW, W = ol.rand_elts(rg, n, field), ol.rand_elts(rg, n, field)
            eq0, s, rr = (ol.rand_elts(rg, 1, field)[0] for _ in range(3))
            ev = np.zeros((3, 2), dtype=np.uint64)
            r.ref_sumcheck_evaluations(field, n, P(eq0), P(QW), P(W), P(s), P(ev))
            wb = W.copy()
            n2 = r.ref_dense_bind(field, n, P(rr), P(wb))
            sc.append({"field": field, "n": n, "seed": seed, "evals": hx(ev), "bind_sha256": sha(wb[:n2])})
    g["sumcheck"] = sc

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "ref_vectors.json"), "w") as f:
        json.dump(g, f, indent=0)
    # the reference's own data fixtures (binary test vectors its Rust tests hold)
    for src in ("rust/runtime/ligero/tests/ligero_test_vector.bin", "rust/runtime/merkle/tests/merkle_test_vector.bin",
                "rust/runtime/merkle/tests/commitment_test_vector.bin"):
        shutil.copyfile(os.path.join(REF, src), os.path.join(OUT, os.path.basename(src)))
    print("wrote", os.path.join(OUT, "ref_vectors.json"), os.path.getsize(os.path.join(OUT, "ref_vectors.json")), "bytes")


if __name__ == "__main__":
    main()
