#!/usr/bin/env python3
"""gen_golden_p256.py -- writes tests/golden/ref_vectors_p256.json from the REAL reference (oracle/_ref/liblfref.so):
the P-256 base-field leg of BASELINE config 5 (the mdoc signature circuit's Ligero tableau: 19 rows, block 455,
dblock 909, block_enc 4096, 32-byte elements; ZkProver<Fp256Base, ReedSolomonFactory<Fp256Base, FFTExtConvolutionFactory>>,
lib/circuits/mdoc/mdoc_zk.cc:485-500 with kZkSpecs[0].block_enc_sig = 4096, rate 7, 132 queries).

Inputs are regenerated from seeds (lfo_p256_fill, an LCG for the nonces); the file holds the reference's outputs: small
vectors as hex, the encoded tableau by SHA-256, the commitment root.  Run in the build container only:
    make -C oracle ref && python oracle/gen_golden_p256.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from oracle_lib import P  # noqa: E402
import ligero_fixture as lf  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_vectors_p256.json")


def hx(a):
    return np.ascontiguousarray(a).tobytes().hex()


def fill(seed, n):
    a = np.zeros((n, 4), dtype=np.uint64)
    ol.oracle().lfo_p256_fill(seed, n, P(a))
    return a


def sig_tableau(nrow=19, block=455, dblock=909, be=4096):
    """un-encoded rows: rows 1, 2 (IDOT, IQUAD) hold dblock values, the others block; the rest zero"""
    T = np.zeros((nrow, be, 4), dtype=np.uint64)
    for r in range(nrow):
        n = dblock if r in (1, 2) else block
        T[r, :n] = fill(5000 + r, n)
    return T


def main():
    r = ol.ref()
    assert r is not None, "build oracle/_ref first (make -C oracle ref)"
    g = {"_generator": "oracle/gen_golden_p256.py over oracle/_ref (reference @ /root/reference)"}
    ops = []
    a, b = fill(1, 40), fill(2, 40)
    pm1 = [0xFFFFFFFFFFFFFFFE, 0x00000000FFFFFFFF, 0, 0xFFFFFFFF00000001]
    a[0], b[0] = pm1, pm1
    a[1], b[1] = [0, 0, 0, 0], pm1
    a[2], b[2] = pm1, [1, 0, 0, 0]
    for i in range(40):
        o = {}
        for name in ("mul", "add", "sub"):
            out = np.zeros(4, dtype=np.uint64)
            getattr(r, "ref_p256_" + name)(P(a[i]), P(b[i]), P(out))
            o[name] = hx(out)
        by = np.zeros(32, dtype=np.uint8)
        r.ref_p256_to_bytes(P(a[i]), P(by))
        o["bytes"] = hx(by)
        ops.append(o)
    g["field_ops"] = {"seed_a": 1, "seed_b": 2, "n": 40, "edge": "a[0]=b[0]=p-1, a[1]=0 b[1]=p-1, a[2]=p-1 b[2]=1 (raw limb images)", "out": ops}
    rs = []
    for n, m in ((1, 4), (3, 8), (5, 16), (21, 128), (100, 257)):
        y = np.zeros((m, 4), dtype=np.uint64)
        y[:n] = fill(100 + n, n)
        r.ref_p256_rs_interpolate(n, m, P(y))
        rs.append({"n": n, "m": m, "seed": 100 + n, "out_sha256": hashlib.sha256(y.tobytes()).hexdigest(), "out_tail": hx(y[-1])})
    g["rs"] = rs
    for n in (8, 64):
        x = fill(300 + n, n)
        r.ref_p256_rfft(0, n, P(x))
        g["r2hc_%d" % n] = {"seed": 300 + n, "out": hx(x)}
    # config 5: the signature tableau
    nrow, block, dblock, be = 19, 455, 909, 4096
    T = sig_tableau(nrow, block, dblock, be)
    r.ref_p256_rs_encode_rows(1, block, be, P(T[0]), be)
    r.ref_p256_rs_encode_rows(2, dblock, be, P(T[1:]), be)
    r.ref_p256_rs_encode_rows(nrow - 3, block, be, P(T[3:]), be)
    ext = be - dblock
    nonces = np.frombuffer(lf.LcgRng(7).bytes(32 * ext), dtype=np.uint8).reshape(ext, 32).copy()
    root = np.zeros(32, dtype=np.uint8)
    r.ref_p256_column_commit(nrow, be, dblock, ext, P(T), P(nonces), P(root))
    g["config5_sig_tableau"] = {"nrow": nrow, "block": block, "dblock": dblock, "block_enc": be, "block_ext": ext, "row_seed0": 5000,
                                "nonce_lcg_seed": 7, "encoded_sha256": hashlib.sha256(T.tobytes()).hexdigest(),
                                "row0_last": hx(T[0, -1]), "row2_col_dblock": hx(T[2, dblock]), "root": hx(root)}
    with open(OUT, "w") as f:
        json.dump(g, f, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
