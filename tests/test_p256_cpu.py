"""P-256 base-field leg (BASELINE config 5) on the CPU: the oracle restatement (oracle/lf_oracle_p256.c) against the
compiled reference (oracle/_ref, when present) and against tests/golden/ref_vectors_p256.json (always)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fill(seed, n):
    a = np.zeros((n, 4), dtype=np.uint64)
    ol.oracle().lfo_p256_fill(seed, n, P(a))
    return a


def un(h, shape=(4,), dtype=np.uint64):
    return np.frombuffer(bytes.fromhex(h), dtype=dtype).reshape(shape).copy()


def sig_tableau(g):
    c = g["config5_sig_tableau"]
    T = np.zeros((c["nrow"], c["block_enc"], 4), dtype=np.uint64)
    for r in range(c["nrow"]):
        n = c["dblock"] if r in (1, 2) else c["block"]
        T[r, :n] = fill(c["row_seed0"] + r, n)
    return c, T


@pytest.fixture(scope="module")
def g():
    with open(os.path.join(GOLD, "ref_vectors_p256.json")) as f:
        return json.load(f)


def test_oracle_matches_golden_field_ops_and_small_transforms(g):
    o = ol.oracle()
    a, b = fill(1, 40), fill(2, 40)
    pm1 = [0xFFFFFFFFFFFFFFFE, 0x00000000FFFFFFFF, 0, 0xFFFFFFFF00000001]
    a[0], b[0] = pm1, pm1
    a[1], b[1] = [0, 0, 0, 0], pm1
    a[2], b[2] = pm1, [1, 0, 0, 0]
    for i, want in enumerate(g["field_ops"]["out"]):
        for name in ("mul", "add", "sub"):
            got = ol.arr32(getattr(o, "lfo_p256_" + name)(ol.e32(a[i]), ol.e32(b[i])))
            assert (got == un(want[name])).all(), (name, i)
        by = np.zeros(32, dtype=np.uint8)
        o.lfo_p256_to_bytes(P(by), ol.e32(a[i]))
        assert by.tobytes().hex() == want["bytes"]
    for n in (8, 64):
        x = fill(300 + n, n)
        o.lfo_p256_r2hc(P(x), n)
        assert x.tobytes().hex() == g["r2hc_%d" % n]["out"]
        o.lfo_p256_hc2r(P(x), n)  # hc2r(r2hc(x)) = n x
        y = fill(300 + n, n)
        nn = o.lfo_p256_of_scalar(n)
        for i in range(n):
            assert (x[i] == ol.arr32(o.lfo_p256_mul(ol.e32(y[i]), nn))).all()
    for v in g["rs"]:
        y = np.zeros((v["m"], 4), dtype=np.uint64)
        y[:v["n"]] = fill(v["seed"], v["n"])
        o.lfo_p256_rs_interpolate(v["n"], v["m"], P(y))
        assert hashlib.sha256(y.tobytes()).hexdigest() == v["out_sha256"] and y[-1].tobytes().hex() == v["out_tail"]


def test_oracle_reproduces_config5_signature_tableau(g):
    """RS rows (455 -> 4096, 909 -> 4096) + column commit of the mdoc signature tableau shape: encoded bytes and root
    equal the reference's"""
    import ligero_fixture as lf
    o = ol.oracle()
    c, T = sig_tableau(g)
    for r in range(c["nrow"]):
        o.lfo_p256_rs_interpolate(c["dblock"] if r in (1, 2) else c["block"], c["block_enc"], P(T[r]))
    assert hashlib.sha256(T.tobytes()).hexdigest() == c["encoded_sha256"]
    nonces = np.frombuffer(lf.LcgRng(c["nonce_lcg_seed"]).bytes(32 * c["block_ext"]), dtype=np.uint8).reshape(-1, 32).copy()
    root = np.zeros(32, dtype=np.uint8)
    o.lfo_column_commit32(c["nrow"], c["block_enc"], c["dblock"], c["block_ext"], P(T), P(nonces), P(root), None)
    assert root.tobytes().hex() == c["root"]


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (reference absent)")
def test_oracle_vs_compiled_reference():
    o, r = ol.oracle(), ol.ref()
    rng = np.random.default_rng(1)
    a, b = fill(11, 300), fill(12, 300)
    pm1 = [0xFFFFFFFFFFFFFFFE, 0x00000000FFFFFFFF, 0, 0xFFFFFFFF00000001]
    a[0] = [0, 0, 0, 0]
    a[1], b[1] = pm1, pm1
    a[2], b[2] = pm1, [1, 0, 0, 0]
    a[3] = [0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0, 0xFFFFFFFF00000000]
    b[3] = a[3]
    for i in range(300):
        for name in ("mul", "add", "sub"):
            out = np.zeros(4, dtype=np.uint64)
            getattr(r, "ref_p256_" + name)(P(a[i]), P(b[i]), P(out))
            assert (ol.arr32(getattr(o, "lfo_p256_" + name)(ol.e32(a[i]), ol.e32(b[i]))) == out).all(), (name, i)
    wr, wi, xr, xi = (np.zeros(4, dtype=np.uint64) for _ in range(4))
    r.ref_p256_omega(P(wr), P(wi))
    o.lfo_p256_omega(P(xr), P(xi))
    assert (wr == xr).all() and (wi == xi).all()
    for n in (2, 4, 8, 16, 32, 64, 128, 1024, 2048):
        x = fill(20 + n, n)
        y = x.copy()
        r.ref_p256_rfft(0, n, P(x))
        o.lfo_p256_r2hc(P(y), n)
        assert (x == y).all(), ("r2hc", n)
        r.ref_p256_rfft(1, n, P(x))
        o.lfo_p256_hc2r(P(y), n)
        assert (x == y).all(), ("hc2r", n)
    for n, m in ((1, 2), (1, 4), (2, 3), (3, 8), (5, 16), (21, 128), (100, 257), (455, 4096), (909, 4096)):
        x = fill(40 + n, m)
        y = x.copy()
        r.ref_p256_rs_interpolate(n, m, P(x))
        o.lfo_p256_rs_interpolate(n, m, P(y))
        assert (x == y).all(), ("rs", n, m)
    T = fill(77, 7 * 40).reshape(7, 40, 4)
    nz = rng.integers(0, 256, size=(31, 32), dtype=np.uint8)
    r1, r2 = np.zeros(32, dtype=np.uint8), np.zeros(32, dtype=np.uint8)
    r.ref_p256_column_commit(7, 40, 9, 31, P(T), P(nz), P(r1))
    o.lfo_column_commit32(7, 40, 9, 31, P(T), P(nz), P(r2), None)
    assert (r1 == r2).all()


def test_small_p256_fixture_is_consistent():
    """tests/golden/small_p256.json (the reference's zk_test example circuit over Fp256Base, proved by the reference): the stored
    SHA-256 is that of the stored wire bytes, the LFC1 header names field id 1, and the witness satisfies the circuit's
    relation 2 n = (s - 2) m^2 - (s - 4) m over the P-256 base field (values out of Montgomery form)."""
    import hashlib
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_p256.json")))
    wire, lfc1, w = bytes.fromhex(fx["zk_wire"]), bytes.fromhex(fx["lfc1"]), bytes.fromhex(fx["witness"])
    assert hashlib.sha256(wire).hexdigest() == fx["zk_wire_sha256"] and fx["reference_verifier_accepts"] is True
    assert lfc1[0] == 1 and int.from_bytes(lfc1[1:4], "little") == 1  # version, FieldID P256_ID
    p = ol.P256_P
    Rinv = pow(1 << 256, -1, p)
    one, n, m, s_ = (int.from_bytes(w[32 * i:32 * i + 32], "little") * Rinv % p for i in range(4))
    assert one == 1 and (2 * n - ((s_ - 2) * m * m - (s_ - 4) * m)) % p == 0


def test_device_montgomery_limb_algorithm():
    """The device's Fp256Base product (csrc/fp256.h, __HIP_DEVICE_COMPILE__ path) restated limb by limb: 8 x 8 product scanning,
    then the Montgomery reduction in four 64-bit steps -- quotient digit = the two low limbs (p = -1 mod 2^64), one add chain
    over limbs i+3 .. i+9 and one subtract chain over i+7 .. i+9 per step, the four carry and four borrow bits that would ripple
    further applied together at the end, one conditional subtraction -- against Python integers, edge values included.  (The
    instructions themselves are checked on the GPU: tests/test_p256_gpu.py.)"""
    import random
    p = 2**256 - 2**224 + 2**192 + 2**96 - 1
    M32 = 2**32 - 1
    rinv = pow(2**256, -1, p)

    def mont(a, b):
        x = [(a >> (32 * i)) & M32 for i in range(8)]
        y = [(b >> (32 * i)) & M32 for i in range(8)]
        T, acc = [0] * 17, 0
        for k in range(15):
            for i in range(max(0, k - 7), min(7, k) + 1):
                acc += x[i] * y[k - i]
            T[k] = acc & M32
            acc >>= 32
        T[15] = acc & M32
        assert acc >> 32 == 0
        cs, bs = [], []
        for i in (0, 2, 4, 6):
            m0, m1 = T[i], T[i + 1]
            c = 0
            for k, add in enumerate((m0, m1, 0, m0, m1, m0, m1)):
                s = T[i + 3 + k] + add + c
                T[i + 3 + k], c = s & M32, s >> 32
            cs.append(c)
            bo = 0
            for k, sub in enumerate((m0, m1, 0)):
                d = T[i + 7 + k] - sub - bo
                T[i + 7 + k], bo = d & M32, 1 if d < 0 else 0
            bs.append(bo)
        c = 0
        for k, add in enumerate((cs[0], 0, cs[1], 0, cs[2], 0, cs[3])):
            s = T[10 + k] + add + c
            T[10 + k], c = s & M32, s >> 32
        bo = 0
        for k, sub in enumerate((bs[0], 0, bs[1], 0, bs[2], 0, bs[3])):
            d = T[10 + k] - sub - bo
            T[10 + k], bo = d & M32, 1 if d < 0 else 0
        v = sum(T[8 + k] << (32 * k) for k in range(8)) + (T[16] << 256)
        assert T[16] in (0, 1) and v < 2 * p
        return v - p if v >= p else v

    rng = random.Random(7)
    edge = [0, 1, 2, p - 1, p - 2, 2**255, 2**224, 2**192 - 1, 2**96, 2**96 - 1, (p - 1) // 2, M32, 2**64 - 1, 2**256 - 2**224]
    for a in edge:
        for b in edge:
            assert mont(a, b) == (a * b * rinv) % p
    for _ in range(20000):
        a, b = rng.randrange(p), rng.randrange(p)
        if rng.random() < 0.1:
            a = rng.choice(edge)
        if rng.random() < 0.1:
            b = p - rng.randrange(1, 2**70)
        assert mont(a, b) == (a * b * rinv) % p
