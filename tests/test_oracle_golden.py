"""Pin the oracle WITHOUT the reference: (1) tests/golden/ref_vectors.json, generated from the
compiled reference by oracle/gen_golden.py; (2) the reference's own known-answer vectors:
GF2_128 beta(1) (lib/gf2k/gf2_128_test.cc:233-249), the Merkle vector of
docs/specs/testvectors.md:7-22, and the C++-generated binary fixtures its Rust tests check
(rust/runtime/merkle/tests/{merkle,commitment}_test_vector.bin, copied as data)."""
import ctypes as C
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import FP, GF, P, elt, arr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    with open(os.path.join(GOLD, "ref_vectors.json")) as f:
        return json.load(f)


def un(h, shape=(-1, 2), dtype=np.uint64):
    return np.frombuffer(bytes.fromhex(h), dtype=dtype).reshape(shape).copy()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_beta1_kat():
    c = ol.gf_ctx(4)
    assert (arr(c.beta[0]) == [1, 0]).all()
    assert (arr(c.beta[1]) == np.array([0xF1871E01B64FDA4C, 0x5C5971877501D4B8], dtype=np.uint64)).all()


def test_field_ops(g):
    o = ol.oracle()
    for name, ops in (("gf", {"mul": o.lfo_gf_mul}), ("fp", {"mul": o.lfo_fp_mul, "add": o.lfo_fp_add, "sub": o.lfo_fp_sub})):
        xs, ys = un(g[name + "_x"]), un(g[name + "_y"])
        for op, fn in ops.items():
            want = un(g["%s_%s" % (name, op)])
            for i in range(len(xs)):
                assert (arr(fn(elt(xs[i]), elt(ys[i]))) == want[i]).all()
    for k in (4, 5):
        c = ol.gf_ctx(k)
        b = un(g["gf_beta_k%d" % k])
        for i in range(1 << k):
            assert (arr(c.beta[i]) == b[i]).all()
        pts = un(g["gf_eval_points_k%d" % k])
        for i in range(6):
            assert (arr(o.lfo_gf_poly_evaluation_point(C.byref(c), i)) == pts[i]).all()
    assert (arr(o.lfo_fp_omega32()) == un(g["fp_omega32_mont"])[0]).all()


def test_lch14(g):
    o = ol.oracle()
    for v in g["lch14"]:
        c = ol.gf_ctx(v["k"])
        a = un(v["in"]) if v["in"] else ol.rand_elts(np.random.default_rng(v["in_seed"]), 1 << v["l"])
        fn = (o.lfo_lch14_fft, o.lfo_lch14_ifft, o.lfo_lch14_bidirectional_fft)[v["dir"]]
        fn(C.byref(c), v["l"], v["coset_or_k"], P(a))
        assert sha(a) == v["out_sha256"]
        if v["out"]:
            assert (a == un(v["out"])).all()
    for v in g["lch14_rs"]:
        a = ol.rand_elts(np.random.default_rng(v["seed"]), v["m"])
        o.lfo_lch14_rs_interpolate(C.byref(ol.gf_ctx(v["k"])), v["n"], v["m"], P(a))
        assert sha(a) == v["out_sha256"]


def test_fp_fft_and_rs(g):
    o = ol.oracle()
    for v in g["fp_fft"]:
        a = np.zeros((v["n"], 2), dtype=np.uint64)
        o.lfo_fp_bogorng_fill(v["bogorng_seed"], v["n"], P(a))
        assert (a[:2] == un(v["in_first"])).all()
        (o.lfo_fp_fftf if v["dir"] else o.lfo_fp_fftb)(P(a), v["n"], o.lfo_fp_omega32(), 1 << 32)
        assert sha(a) == v["out_sha256"]
    for v in g["fp_rs"]:
        a = np.zeros((v["m"], 2), dtype=np.uint64)
        o.lfo_fp_bogorng_fill(v["bogorng_seed"], v["m"], P(a))
        o.lfo_fp_rs_interpolate(v["n"], v["m"], P(a))
        assert sha(a) == v["out_sha256"]


def test_merkle_and_column_commit(g):
    o = ol.oracle()
    for v in g["merkle"]:
        n = v["n"]
        leaves = np.random.default_rng(v["seed"]).integers(0, 256, size=(n, 32), dtype=np.uint8)
        lay = np.zeros((2 * n, 32), dtype=np.uint8)
        o.lfo_merkle_build_tree(n, P(leaves), P(lay))
        assert lay[1].tobytes().hex() == v["root"] and sha(lay[1:]) == v["layers_sha256"]
    for v in g["column_commit"]:
        rg = np.random.default_rng(v["seed"])
        T = ol.rand_elts(rg, v["nrow"] * v["ld"], v["field"])
        nonces = rg.integers(0, 256, size=(v["ncols"], 32), dtype=np.uint8)
        root = np.zeros(32, dtype=np.uint8)
        o.lfo_column_commit(v["field"], v["nrow"], v["ld"], v["col0"], v["ncols"], P(T), P(nonces), P(root), None)
        assert root.tobytes().hex() == v["root"]


def test_sumcheck_pieces(g):
    o = ol.oracle()
    for v in g["sumcheck"]:
        field, n = v["field"], v["n"]
        rg = np.random.default_rng(v["seed"])
        QW, W = ol.rand_elts(rg, n, field), ol.rand_elts(rg, n, field)
        eq0, s, rr = (ol.rand_elts(rg, 1, field)[0] for _ in range(3))
        ev = np.zeros((3, 2), dtype=np.uint64)
        o.lfo_sumcheck_evaluations(field, C.byref(ol.gf_ctx(4)), n, elt(eq0), P(QW), P(W), elt(s), P(ev))
        assert (ev == un(v["evals"])).all()
        out = np.zeros(((n + 1) // 2, 2), dtype=np.uint64)
        o.lfo_dense_bind(field, n, elt(rr), P(W), P(out))
        assert sha(out) == v["bind_sha256"]


# ---- the reference's own KATs -------------------------------------------------------------
SPEC_LEAVES = ["4bf5122f344554c53bde2ebb8cd2b7e3d1600ad631c385a5d7cce23c7785459a",
               "dbc1b4c900ffe48d575b5da5c638040125f65db0fe3e24494b76ea986457d986",
               "084fed08b978af4d7d196a7446a86b58009e636b611db16211b65a9aadff29c5",
               "e52d9c508c502347344d8c07ad91cbd6068afc75ff6292f062a09ca381c89e71",
               "e77b9a9ae9e30b0dbdb6f510a264ef9de781501d7b6b92ae89eb059c5ab743db"]
SPEC_ROOT = "f22f4501ffd3bdffcecc9e4cd6828a4479aeedd6aa484eb7c1f808ccf71c6e76"


def host_open(layers, n, pos):
    """MerkleTree::generate_compressed_proof replayed over a layers array (merkle_tree.h:122-143)"""
    tree = [False] * (2 * n)
    for p_ in pos:
        tree[p_ + n] = True
    for i in range(n - 1, 0, -1):
        tree[i] = tree[2 * i] or tree[2 * i + 1]
    out = []
    for i in range(n - 1, 0, -1):
        if tree[i]:
            ch = 2 * i
            if tree[ch]:
                ch = 2 * i + 1
            if not tree[ch]:
                out.append(bytes(layers[ch]))
    return out


def test_spec_merkle_vector():
    """docs/specs/testvectors.md:7-22"""
    o = ol.oracle()
    leaves = np.frombuffer(bytes.fromhex("".join(SPEC_LEAVES)), dtype=np.uint8).reshape(5, 32).copy()
    lay = np.zeros((10, 32), dtype=np.uint8)
    o.lfo_merkle_build_tree(5, P(leaves), P(lay))
    assert lay[1].tobytes().hex() == SPEC_ROOT
    assert [d.hex() for d in host_open(lay, 5, [0, 1])] == [SPEC_LEAVES[2], "f03808f5b8088c61286d505e8e93aa378991d9889ae2d874433ca06acabcd493"]
    assert [d.hex() for d in host_open(lay, 5, [1, 3])] == [SPEC_LEAVES[4], SPEC_LEAVES[2], SPEC_LEAVES[0]]


def read_merkle_fixture():
    """layout decoded by rust/runtime/merkle/tests/merkle.rs:218-268"""
    d = open(os.path.join(GOLD, "merkle_test_vector.bin"), "rb").read()
    off = 0
    (n,) = struct.unpack_from("<Q", d, off); off += 8
    leaves = np.frombuffer(d, dtype=np.uint8, count=32 * n, offset=off).reshape(n, 32).copy(); off += 32 * n
    (nq,) = struct.unpack_from("<Q", d, off); off += 8
    idx = list(struct.unpack_from("<%dQ" % nq, d, off)); off += 8 * nq
    root = d[off:off + 32]; off += 32
    (plen,) = struct.unpack_from("<Q", d, off); off += 8
    proof = [d[off + 32 * i:off + 32 * i + 32] for i in range(plen)]; off += 32 * plen
    assert off == len(d)
    return n, leaves, idx, root, proof


def test_cpp_merkle_fixture():
    o = ol.oracle()
    n, leaves, idx, root, proof = read_merkle_fixture()
    lay = np.zeros((2 * n, 32), dtype=np.uint8)
    o.lfo_merkle_build_tree(n, P(leaves), P(lay))
    assert lay[1].tobytes() == root
    assert host_open(lay, n, idx) == proof


def read_commitment_fixture():
    """layout decoded by rust/runtime/merkle/tests/merkle.rs:318-372; nonce bytes come from a
    counter RNG (0,1,2,...), leaf data is the 8-byte LE column index (see merkle.rs test body)"""
    d = open(os.path.join(GOLD, "commitment_test_vector.bin"), "rb").read()
    off = 0
    n, nq = struct.unpack_from("<QQ", d, off); off += 16
    idx = list(struct.unpack_from("<%dQ" % nq, d, off)); off += 8 * nq
    root = d[off:off + 32]; off += 32
    nonces = [d[off + 32 * i:off + 32 * i + 32] for i in range(nq)]; off += 32 * nq
    (plen,) = struct.unpack_from("<Q", d, off); off += 8
    path = [d[off + 32 * i:off + 32 * i + 32] for i in range(plen)]; off += 32 * plen
    assert off == len(d)
    return n, idx, root, nonces, path


def test_cpp_commitment_fixture():
    """MerkleCommitment::commit replayed (merkle_commitment.h:50-64): leaf i draws the i-th 32-byte
    nonce from the fixture's counter RNG (starting at 42, merkle.rs:374-375) and hashes
    nonce || (3i, 5i, 7i, 11i) mod 256; root, opened nonces and compressed path must match."""
    o = ol.oracle()
    n, idx, root, nonces, path = read_commitment_fixture()
    ctr = 42
    all_nonces, leaves = [], np.zeros((n, 32), dtype=np.uint8)
    for i in range(n):
        nb = bytes((ctr + j) % 256 for j in range(32))
        ctr = (ctr + 32) % 256
        all_nonces.append(nb)
        col = bytes([(i * 3) % 256, (i * 5) % 256, (i * 7) % 256, (i * 11) % 256])
        leaves[i] = np.frombuffer(hashlib.sha256(nb + col).digest(), dtype=np.uint8)
    lay = np.zeros((2 * n, 32), dtype=np.uint8)
    o.lfo_merkle_build_tree(n, P(leaves), P(lay))
    assert lay[1].tobytes() == root
    assert [all_nonces[i] for i in idx] == nonces
    assert host_open(lay, n, idx) == path


def test_f64_2_fft_golden():
    """FFT<Fp2<Fp<1>>> (lib/algebra/fft_test.cc:205-229): the oracle against tests/golden/f64_2.json, written from the
    reference itself by oracle/gen_golden_f64.py"""
    with open(os.path.join(ol.ROOT, "tests", "golden", "f64_2.json")) as f:
        g64 = json.load(f)
    o = ol.oracle()
    fns = (o.lfo_f64_2_add, o.lfo_f64_2_sub, o.lfo_f64_2_mul)
    for v in g64["binop"]:
        a, b = np.frombuffer(bytes.fromhex(v["a"]), dtype=np.uint64), np.frombuffer(bytes.fromhex(v["b"]), dtype=np.uint64)
        got = o.lfo_f64_2_inv(ol.elt(a)) if v["op"] == 3 else fns[v["op"]](ol.elt(a), ol.elt(b))
        assert ol.arr(got).tobytes().hex() == v["out"]
    roots = {"real": np.frombuffer(bytes.fromhex(g64["omega32"]), dtype=np.uint64),
             "times_i": np.frombuffer(bytes.fromhex(g64["omega32_times_i"]), dtype=np.uint64)}
    assert int(roots["real"][0]) == o.lfo_f64_omega32() and int(roots["real"][1]) == 0
    for v in g64["fft"]:
        a = np.zeros((v["n"], 2), dtype=np.uint64)
        o.lfo_f64_2_bogorng_fill(v["bogorng_seed"], v["imag"], v["n"], P(a))
        assert a[:2].tobytes().hex() == v["in_first"]
        (o.lfo_f64_2_fftf if v["dir"] else o.lfo_f64_2_fftb)(P(a), v["n"], ol.elt(roots[v["root"]]), 1 << 32)
        assert sha(a) == v["out_sha256"]
        if v["out"]:
            assert a.tobytes().hex() == v["out"]
