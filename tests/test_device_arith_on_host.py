"""The device arithmetic in longfellow-zk_amd/csrc/fields.h is LF_HD (host+device); here its
HOST compilation is checked against the oracle, so carry/borrow logic errors surface in the
CPU-only container before any GPU run.  (The GPU parity tests exercise the device build.)"""
import ctypes as C
import hashlib
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import FP, GF, P, elt, arr

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libhostfields.so")
SRC = os.path.join(HERE, "host_fields.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _lib():
    hdrs = [os.path.join(ol.ROOT, "longfellow-zk_amd", "csrc", h) for h in ("fields.h", "fp256.h", "bitslice.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max([os.path.getmtime(SRC)] + [os.path.getmtime(h) for h in hdrs]):
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available")
        subprocess.check_call([HIPCC, "--cuda-host-only", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    return C.CDLL(SO)


def _bin(fn, a, b):
    out = np.zeros(2, dtype=np.uint64)
    fn(P(a), P(b), P(out))
    return out


def test_fp_ops_match_oracle():
    L, o = _lib(), ol.oracle()
    rng = np.random.default_rng(5)
    xs, ys = ol.rand_elts(rng, 5000, FP), ol.rand_elts(rng, 5000, FP)
    pm1 = np.array([0, 0xFFFFF00000000000], dtype=np.uint64)
    edge = [np.array(v, dtype=np.uint64) for v in ([0, 0], [1, 0], pm1, [0xFFFFFFFFFFFFFFFF, 0xFFFFEFFFFFFFFFFF],
                                                   [0xFFFFFFFFFFFFFFFF, 0], [0, 1], [0, 0xFFFFF00000000000 - 1])]
    k = 0
    for a in edge:
        for b in edge:
            xs[k], ys[k] = a, b
            k += 1
    for x, y in zip(xs, ys):
        assert (_bin(L.hf_fp_mul, x, y) == arr(o.lfo_fp_mul(elt(x), elt(y)))).all()
        assert (_bin(L.hf_fp_add, x, y) == arr(o.lfo_fp_add(elt(x), elt(y)))).all()
        assert (_bin(L.hf_fp_sub, x, y) == arr(o.lfo_fp_sub(elt(x), elt(y)))).all()


def test_f64_ops_match_big_integers_and_oracle():
    """F64 = Fp<1> over p = 2^64 - 2^32 + 1: the shift-only Montgomery reduction of fields.h (f64_mul) against Python
    integers on edge values and 200 000 random pairs, then F64_2 products against the oracle"""
    L, o = _lib(), ol.oracle()
    p = 2**64 - 2**32 + 1
    rinv = pow(1 << 64, -1, p)
    rng = np.random.default_rng(11)
    edge = [0, 1, 2, 0xFFFFFFFF, 1 << 32, (1 << 32) + 1, p - 1, p - 2, (p - 1) // 2, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001,
            0x8000000000000000, 0x7FFFFFFFFFFFFFFF, 0x00000000FFFFFFFE]
    pairs = [(a, b) for a in edge for b in edge]
    n = 200000
    ra = (rng.integers(0, 1 << 63, size=n, dtype=np.uint64).astype(object) * 2 + rng.integers(0, 2, size=n).astype(object)) % p
    rb = (rng.integers(0, 1 << 63, size=n, dtype=np.uint64).astype(object) * 2 + rng.integers(0, 2, size=n).astype(object)) % p
    A = np.array([x for x, _ in pairs] + list(ra), dtype=np.uint64)
    B = np.array([y for _, y in pairs] + list(rb), dtype=np.uint64)
    out = np.zeros_like(A)
    ai, bi = [int(x) for x in A], [int(x) for x in B]
    L.hf_f64_mul_many(C.c_size_t(len(A)), P(A), P(B), P(out))
    assert [int(x) for x in out] == [(x * y * rinv) % p for x, y in zip(ai, bi)]
    L.hf_f64_add_many(C.c_size_t(len(A)), P(A), P(B), P(out))
    assert [int(x) for x in out] == [(x + y) % p for x, y in zip(ai, bi)]
    L.hf_f64_sub_many(C.c_size_t(len(A)), P(A), P(B), P(out))
    assert [int(x) for x in out] == [(x - y) % p for x, y in zip(ai, bi)]
    for i in range(0, 2000, 2):
        a, b = A[i:i + 2].copy(), B[i:i + 2].copy()
        assert (_bin(L.hf_f64x2_mul, a, b) == arr(o.lfo_f64_2_mul(elt(a), elt(b)))).all()
        b[1] = 0
        assert (_bin(L.hf_f64x2_mul_real, a, b) == arr(o.lfo_f64_2_mul(elt(a), elt(b)))).all()


def test_fp_reduce_limbs_matches_big_integers():
    """fp_reduce_limbs: sum_k a_k 2^(32k) mod p for u64 limb accumulators, incl. the extreme words"""
    L = _lib()
    p = 2**128 - 2**108 + 1
    rng = np.random.default_rng(9)
    cases = [rng.integers(0, 2**64, size=4, dtype=np.uint64) for _ in range(4000)]
    m = 2**64 - 1
    ext = [0, 1, m, 2**32, 2**32 - 1, 2**63, 0xFFFFF00000000000, 0xFFFFF, 1 << 20, (1 << 20) - 1]
    for a in ext:
        for b in ext:
            cases.append(np.array([a, b, ext[(a + b) % len(ext)], ext[(a ^ b) % len(ext)]], dtype=np.uint64))
            cases.append(np.array([b, 0, 0, a], dtype=np.uint64))
            cases.append(np.array([m, m, a, b], dtype=np.uint64))
    for a in cases:
        out = np.zeros(2, dtype=np.uint64)
        L.hf_fp_reduce_limbs(P(a), P(out))
        want = sum(int(a[k]) << (32 * k) for k in range(4)) % p
        assert int(out[0]) | (int(out[1]) << 64) == want, [hex(int(x)) for x in a]


def test_gf_mul_matches_oracle():
    L, o = _lib(), ol.oracle()
    rng = np.random.default_rng(6)
    xs, ys = ol.rand_elts(rng, 5000), ol.rand_elts(rng, 5000)
    xs[0] = 0
    ys[1] = 0
    xs[2] = [0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF]
    ys[2] = [0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF]
    for x, y in zip(xs, ys):
        assert (_bin(L.hf_gf_mul, x, y) == arr(o.lfo_gf_mul(elt(x), elt(y)))).all()


def test_sha_compress_matches_hashlib():
    L = _lib()
    rng = np.random.default_rng(7)
    for nblk in (1, 2, 5):
        # message of 64*nblk - 9 bytes so the padded message is exactly nblk blocks
        msg = bytes(rng.integers(0, 256, size=64 * nblk - 9, dtype=np.uint8))
        padded = msg + b"\x80" + (len(msg) * 8).to_bytes(8, "big")
        assert len(padded) == 64 * nblk
        h = np.zeros(8, dtype=np.uint32)
        buf = np.frombuffer(padded, dtype=np.uint8).copy()
        L.hf_sha_blocks(P(buf), nblk, P(h))
        got = b"".join(int(x).to_bytes(4, "big") for x in h)
        assert got == hashlib.sha256(msg).digest()


def test_bitslice_transpose():
    L = _lib()
    rng = np.random.default_rng(8)
    x = rng.integers(0, 2**32, size=32, dtype=np.uint32)
    y = x.copy()
    L.hf_transpose32(P(y))
    for r in range(32):
        for b in range(32):
            assert (int(y[b]) >> r) & 1 == (int(x[r]) >> b) & 1


@pytest.mark.parametrize("k", [4, 5])
@pytest.mark.parametrize("mode", [0, 1])
def test_bitsliced_tower_multiply_matches_gf_mul(k, mode):
    """rows -> bit-sliced tower basis -> multiply by a subfield twiddle -> back == gf_mul per row
    (pins tools/gen_tower.py's basis/programs and csrc/bitslice.h on the CPU)"""
    import ctypes as C
    L, o = _lib(), ol.oracle()
    c = ol.gf_ctx(k)
    rng = np.random.default_rng(9 + k)
    for trial in range(6):
        x = ol.rand_elts(rng, 32)
        if trial == 0:
            t = np.array([1, 0], dtype=np.uint64)
        else:
            t = arr(o.lfo_lch14_twiddle(C.byref(c), int(rng.integers(0, 1 << k)), int(rng.integers(0, 2**((1 << k) - 1)))))
        out = np.zeros((32, 2), dtype=np.uint64)
        L.hf_bs_mul(k, P(t), P(x), P(out), mode)
        for r in range(32):
            assert (out[r] == arr(o.lfo_gf_mul(elt(t), elt(x[r])))).all(), (trial, r)


def test_p256_ops_match_oracle_and_big_integers():
    """fp256.h (Fp256Base Montgomery arithmetic with the multiplication-free reduction step, fp_p256.h:43-62) on the host
    against the oracle's generic CIOS and against Python integers, edge values included"""
    L, o = _lib(), ol.oracle()
    p, R = ol.P256_P, 1 << 256
    rng = np.random.default_rng(256)

    def to_arr(v):
        return np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)

    def to_int(a):
        return sum(int(a[i]) << (64 * i) for i in range(4))

    vals = [0, 1, 2, p - 1, p - 2, (1 << 96) - 1, 1 << 96, (1 << 192), (1 << 224) - 1, p >> 1, (p >> 1) + 1, 2**255, 2**32 - 1, 2**64 - 1, 2**64]
    vals += [int.from_bytes(rng.bytes(32), "little") % p for _ in range(600)]
    pairs = [(a, b) for a in vals[:15] for b in vals[:15]] + list(zip(vals[15:315], vals[315:615]))
    Rinv = pow(R, -1, p)
    for a, b in pairs:
        xa, xb = to_arr(a), to_arr(b)
        for name, want in (("mul", a * b * Rinv % p), ("add", (a + b) % p), ("sub", (a - b) % p)):
            out = np.zeros(4, dtype=np.uint64)
            getattr(L, "hf_p256_" + name)(P(xa), P(xb), P(out))
            assert to_int(out) == want, (name, hex(a), hex(b))
            assert (out == ol.arr32(getattr(o, "lfo_p256_" + name)(ol.e32(xa), ol.e32(xb)))).all()
        out = np.zeros(4, dtype=np.uint64)
        L.hf_p256_canon(P(xa), P(out))
        assert to_int(out) == a * Rinv % p
    # Fp2 product: (a + bi)(c + di) in Montgomery form
    for _ in range(100):
        a, b, c, d = (int.from_bytes(rng.bytes(32), "little") % p for _ in range(4))
        x = np.concatenate([to_arr(a), to_arr(b)])
        y = np.concatenate([to_arr(c), to_arr(d)])
        out = np.zeros(8, dtype=np.uint64)
        L.hf_p256_c2mul(P(x), P(y), P(out))
        assert to_int(out[:4]) == (a * c - b * d) * Rinv % p and to_int(out[4:]) == (a * d + b * c) * Rinv % p


def test_p256_limb_accumulators_and_host_helpers():
    """fp256_reduce_limbs (the reduction of the eight 64-bit limb accumulators the P-256 sumcheck kernels sum canonical
    residues into, csrc/zk256.hip) against Python integers -- up to 2^32 - 1 addends per limb; h256_of_scalar / inv / of_bytes"""
    import ctypes as C
    L = _lib()
    p, R = ol.P256_P, 1 << 256
    rng = np.random.default_rng(7)

    def to_int(a):
        return sum(int(a[i]) << (64 * i) for i in range(4))

    cases = [[0] * 8, [2**64 - 1] * 8, [2**32 - 1] * 8, [1] + [0] * 7, [0] * 7 + [2**64 - 1]]
    cases += [[int(x) for x in rng.integers(0, 2**64, size=8, dtype=np.uint64)] for _ in range(300)]
    # sums of real residues: n copies of limbs of (p - 1)
    w = [((p - 1) >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
    cases += [[n * x for x in w] for n in (1, 2, 1000, 2**32 - 1)]
    # integers around the fold boundaries (S = lo + 2^256 hi: hi up to 2^40 - 1, results next to 0, p and 2p), written as limbs
    D = (1 << 224) - (1 << 192) - (1 << 96) + 1
    for S in (p - 1, p, p + 1, 2 * p - 1, 2 * p, (1 << 256) - 1, 1 << 256, (1 << 256) + D, ((1 << 32) - 1) << 256, (((1 << 32) - 1) << 256) + (1 << 256) - 1,
              (511 << 256) + p - 1, (1 << 264) - 1, 3 * p + 7, (1 << 256) + (1 << 233) - 1, (1 << 288) - 1):  # S < 2^288: what eight 64-bit limbs can hold
        limbs = [(S >> (32 * k)) & 0xFFFFFFFF for k in range(7)] + [S >> 224]  # the top accumulator takes everything above 2^224
        assert limbs[7] < (1 << 64)
        cases.append(limbs)
    for acc in cases:
        a = np.array(acc, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        L.hf_p256_reduce_limbs(P(a), P(out))
        assert to_int(out) == sum(v << (32 * k) for k, v in enumerate(acc)) % p, acc
    for u in (0, 1, 2, 10, 2**64 - 1):
        out = np.zeros(4, dtype=np.uint64)
        L.hf_p256_of_scalar(C.c_uint64(u), P(out))
        assert to_int(out) == u * R % p
    for _ in range(5):
        x = int.from_bytes(rng.bytes(32), "little") % p or 1
        a = np.array([(x * R % p >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        L.hf_p256_inv(P(a), P(out))
        assert to_int(out) == pow(x, -1, p) * R % p
    for v, fits in ((0, 1), (p - 1, 1), (p, 0), (2**256 - 1, 0), (12345, 1)):
        b = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8).copy()
        out = np.zeros(4, dtype=np.uint64)
        assert L.hf_p256_of_bytes(b.ctypes.data_as(C.c_void_p), P(out)) == fits
        if fits:
            assert to_int(out) == v * R % p


def test_p256_bulk_sampling_is_the_reference_stream():
    """h256_sample_many (one RandomEngine call for many elements; csrc/zk256.hip draws its pads and blinding rows with it)
    returns the same elements and consumes the same bytes as one 32-byte attempt at a time, INCLUDING rejected attempts
    (values >= p: probability 2^-32 per draw in practice, forced here)."""
    import ctypes as C
    L = _lib()
    L.hf_p256_sample_both.restype = C.c_size_t
    p = ol.P256_P
    rng = np.random.default_rng(11)
    chunks = [rng.bytes(32) for _ in range(40)]
    for pos in (0, 3, 4, 17, 18, 19):  # rejected attempts: all ones, p itself, p + 5 -- also two in a row and at the start
        chunks[pos] = [(2**256 - 1).to_bytes(32, "little"), p.to_bytes(32, "little"), (p + 5).to_bytes(32, "little")][pos % 3]
    stream = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()
    for n in (1, 2, 5, 16, 30):
        bulk, single = np.zeros((n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)
        used = (C.c_size_t * 2)()
        calls = L.hf_p256_sample_both(stream.ctypes.data_as(C.c_void_p), C.c_size_t(stream.size), C.c_size_t(n), P(bulk), P(single), used)
        assert (bulk == single).all(), n
        assert used[0] == used[1], (n, list(used))
        assert calls < n + 6  # a handful of calls, not one per element
        ok = [c for c in chunks if int.from_bytes(c, "little") < p][:n]
        want = [int.from_bytes(c, "little") * (1 << 256) % p for c in ok]
        got = [sum(int(bulk[i, k]) << (64 * k) for k in range(4)) for i in range(n)]
        assert got == want
