"""Pin the C restatement (oracle/lf_oracle.c) against the REAL reference compiled
into oracle/_ref (only possible in the build container; skipped on the GPU box,
where the committed golden vectors in tests/golden pin the oracle instead)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import FP, GF, P, elt, arr

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")]


def _ref_binop(fn, a, b):
    out = np.zeros(2, dtype=np.uint64)
    fn(P(a), P(b), P(out))
    return out


def test_gf_mul_and_inv():
    o, r = ol.oracle(), ol.ref()
    rng = np.random.default_rng(1)
    xs = ol.rand_elts(rng, 300)
    ys = ol.rand_elts(rng, 300)
    xs[0] = 0
    ys[1] = 0
    xs[2] = [1, 0]
    for x, y in zip(xs, ys):
        want = _ref_binop(r.ref_gf_mul, x, y)
        assert (arr(o.lfo_gf_mul(elt(x), elt(y))) == want).all()
        assert (arr(o.lfo_gf_mul_bitserial(elt(x), elt(y))) == want).all()
    for x in xs[2:40]:
        out = np.zeros(2, dtype=np.uint64)
        r.ref_gf_inv(P(x), P(out))
        assert (arr(o.lfo_gf_inv(elt(x))) == out).all()


@pytest.mark.parametrize("k", [4, 5])
def test_gf_basis_and_twiddles(k):
    o, r = ol.oracle(), ol.ref()
    c = ol.gf_ctx(k)
    out = np.zeros(2, dtype=np.uint64)
    for i in range(1 << k):
        r.ref_gf_beta(k, i, P(out))
        assert (arr(c.beta[i]) == out).all()
    for u in [0, 1, 2, 3, 0x1234, 0xFFFF] + ([0xDEADBEEF] if k == 5 else []):
        r.ref_gf_of_scalar(k, u, P(out))
        assert (arr(o.lfo_gf_of_scalar(C.byref(c), u)) == out).all()
    for i in range(6):
        r.ref_gf_poly_evaluation_point(k, i, P(out))
        assert (arr(o.lfo_gf_poly_evaluation_point(C.byref(c), i)) == out).all()
    for i in range(0, 1 << k, 3):
        for u in (0, 1, 1 << i, 0x5a5a & ((1 << (1 << k)) - 1), (1 << (1 << k)) - 1):
            r.ref_lch14_twiddle(k, i, u, P(out))
            assert (arr(o.lfo_lch14_twiddle(C.byref(c), i, u)) == out).all()


@pytest.mark.parametrize("k,l,coset", [(4, 0, 0), (4, 1, 0), (4, 3, 8), (4, 7, 0), (4, 10, 3 << 10), (4, 12, 0),
                                       (5, 10, 1 << 10), (5, 14, 5 << 14)])
def test_lch14_fft_ifft(k, l, coset):
    o, r = ol.oracle(), ol.ref()
    c = ol.gf_ctx(k)
    rng = np.random.default_rng(l * 31 + k)
    a = ol.rand_elts(rng, 1 << l)
    for d, fn in ((0, o.lfo_lch14_fft), (1, o.lfo_lch14_ifft)):
        x, y = a.copy(), a.copy()
        fn(C.byref(c), l, coset, P(x))
        r.ref_lch14_fft(k, d, l, coset, P(y))
        assert (x == y).all()


@pytest.mark.parametrize("l", [1, 2, 5, 8])
def test_lch14_bidirectional_all_k(l):
    o, r = ol.oracle(), ol.ref()
    c = ol.gf_ctx(4)
    rng = np.random.default_rng(l)
    for kk in range(0, (1 << l) + 1, max(1, (1 << l) // 37)):
        a = ol.rand_elts(rng, 1 << l)
        x, y = a.copy(), a.copy()
        o.lfo_lch14_bidirectional_fft(C.byref(c), l, kk, P(x))
        r.ref_lch14_fft(4, 2, l, kk, P(y))
        assert (x == y).all(), kk


@pytest.mark.parametrize("k,n,m", [(4, 1, 7), (4, 5, 5), (4, 21, 128), (4, 100, 128), (4, 455, 4096), (4, 909, 4096),
                                   (4, 910, 8192), (4, 1819, 8192), (4, 682, 4096), (4, 1363, 4096),
                                   (4, 461, 4151), (4, 921, 4151), (5, 1000, 5000), (4, 64, 64), (4, 64, 300)])
def test_lch14_rs_interpolate(k, n, m):
    o, r = ol.oracle(), ol.ref()
    c = ol.gf_ctx(k)
    rng = np.random.default_rng(n + m)
    a = ol.rand_elts(rng, m)
    x, y = a.copy(), a.copy()
    o.lfo_lch14_rs_interpolate(C.byref(c), n, m, P(x))
    r.ref_lch14_rs_interpolate(k, n, m, P(y))
    assert (x == y).all()


def test_fp_ops():
    o, r = ol.oracle(), ol.ref()
    rng = np.random.default_rng(2)
    xs = ol.rand_elts(rng, 300, FP)
    ys = ol.rand_elts(rng, 300, FP)
    # edge values: 0, 1, p-1, near-p
    pm1 = np.array([0, 0xFFFFF00000000000], dtype=np.uint64)
    xs[0], ys[0] = 0, 0
    xs[1], ys[1] = pm1, pm1
    xs[2], ys[2] = pm1, [1, 0]
    xs[3], ys[3] = [0xFFFFFFFFFFFFFFFF, 0xFFFFEFFFFFFFFFFF], [0xFFFFFFFFFFFFFFFF, 0xFFFFEFFFFFFFFFFF]
    for x, y in zip(xs, ys):
        for of, rf in ((o.lfo_fp_mul, r.ref_fp_mul), (o.lfo_fp_add, r.ref_fp_add), (o.lfo_fp_sub, r.ref_fp_sub)):
            assert (arr(of(elt(x), elt(y))) == _ref_binop(rf, x, y)).all()
    out = np.zeros(2, dtype=np.uint64)
    for x in xs[4:24]:
        r.ref_fp_inv(P(x), P(out))
        assert (arr(o.lfo_fp_inv(elt(x))) == out).all()
        r.ref_fp_from_mont(P(x), P(out))
        assert (arr(o.lfo_fp_from_mont(elt(x))) == out).all()
    for u in (0, 1, 2, 12345, 2**64 - 1):
        r.ref_fp_of_scalar(u, P(out))
        assert (arr(o.lfo_fp_of_scalar(u)) == out).all()
    r.ref_fp_omega32(P(out))
    assert (arr(o.lfo_fp_omega32()) == out).all()
    a = np.zeros((50, 2), dtype=np.uint64)
    b = np.zeros((50, 2), dtype=np.uint64)
    o.lfo_fp_bogorng_fill(1234569, 50, P(a))
    r.ref_fp_bogorng_fill(1234569, 50, P(b))
    assert (a == b).all()


@pytest.mark.parametrize("n", [1, 2, 4, 64, 1024, 1 << 14, 1 << 15, 1 << 17])
def test_fp_fft(n):
    """covers the reference's basecase (n <= 16384) and its recursive six-step path (n > 16384)"""
    o, r = ol.oracle(), ol.ref()
    a = np.zeros((n, 2), dtype=np.uint64)
    o.lfo_fp_bogorng_fill(1234569 + n, n, P(a))
    for d, fn in ((0, o.lfo_fp_fftb), (1, o.lfo_fp_fftf)):
        x, y = a.copy(), a.copy()
        fn(P(x), n, o.lfo_fp_omega32(), 1 << 32)
        r.ref_fp_fft(d, n, P(y))
        assert (x == y).all()


@pytest.mark.parametrize("n,m", [(1, 4), (3, 8), (21, 128), (100, 257), (455, 4096)])
def test_fp_rs_interpolate(n, m):
    o, r = ol.oracle(), ol.ref()
    a = np.zeros((m, 2), dtype=np.uint64)
    o.lfo_fp_bogorng_fill(77 + n, m, P(a))
    x, y = a.copy(), a.copy()
    o.lfo_fp_rs_interpolate(n, m, P(x))
    r.ref_fp_rs_interpolate(n, m, P(y))
    assert (x[:m] == y[:m]).all()


@pytest.mark.parametrize("n", [1, 2, 3, 7, 256, 1000, 3187])
def test_merkle_tree(n):
    o, r = ol.oracle(), ol.ref()
    rng = np.random.default_rng(n)
    leaves = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a = np.zeros((2 * n, 32), dtype=np.uint8)
    b = np.zeros((2 * n, 32), dtype=np.uint8)
    o.lfo_merkle_build_tree(n, P(leaves), P(a))
    r.ref_merkle_build_tree(n, P(leaves), P(b))
    assert (a[1:] == b[1:]).all()


@pytest.mark.parametrize("field,nrow,ld,col0,ncols", [(GF, 1, 8, 0, 8), (GF, 20, 4096, 909, 3187), (GF, 7, 64, 13, 51),
                                                      (FP, 5, 40, 9, 31), (FP, 1, 3, 1, 2)])
def test_column_commit(field, nrow, ld, col0, ncols):
    o, r = ol.oracle(), ol.ref()
    rng = np.random.default_rng(nrow * ld)
    T = ol.rand_elts(rng, nrow * ld, field)
    nonces = rng.integers(0, 256, size=(ncols, 32), dtype=np.uint8)
    ra = np.zeros(32, dtype=np.uint8)
    rb = np.zeros(32, dtype=np.uint8)
    o.lfo_column_commit(field, nrow, ld, col0, ncols, P(T), P(nonces), P(ra), None)
    r.ref_column_commit(field, nrow, ld, col0, ncols, P(T), P(nonces), P(rb))
    assert (ra == rb).all()


@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("n", [1, 2, 3, 8, 17, 1000, 1001])
def test_sumcheck_evaluations_and_bind(field, n):
    o, r = ol.oracle(), ol.ref()
    rng = np.random.default_rng(n + field)
    QW, W = ol.rand_elts(rng, n, field), ol.rand_elts(rng, n, field)
    eq0, s, rr = (ol.rand_elts(rng, 1, field)[0] for _ in range(3))
    ea = np.zeros((3, 2), dtype=np.uint64)
    eb = np.zeros((3, 2), dtype=np.uint64)
    o.lfo_sumcheck_evaluations(field, C.byref(ol.gf_ctx(4)), n, elt(eq0), P(QW), P(W), elt(s), P(ea))
    r.ref_sumcheck_evaluations(field, n, P(eq0), P(QW), P(W), P(s), P(eb))
    assert (ea == eb).all()
    out = np.zeros(((n + 1) // 2, 2), dtype=np.uint64)
    wb = W.copy()
    n2 = o.lfo_dense_bind(field, n, elt(rr), P(W), P(out))
    n3 = r.ref_dense_bind(field, n, P(rr), P(wb))
    assert n2 == n3 == (n + 1) // 2
    assert (out == wb[:n2]).all()


@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("hand", [0, 1])
def test_hquad_bind_h(field, hand):
    o, r = ol.oracle(), ol.ref()
    rng = np.random.default_rng(7 + hand)
    # random sparse corner set, Morton-sorted like EQuad::canonicalize (lib/sumcheck/equad.h:79-106)
    pts = sorted({(int(a), int(b)) for a, b in rng.integers(0, 64, size=(900, 2))},
                 key=lambda p: _morton(p[0], p[1]))
    hc = np.array(pts, dtype=np.uint32)
    n = len(pts)
    vc = ol.rand_elts(rng, n, field)
    for _ in range(6):
        rr = ol.rand_elts(rng, 1, field)[0]
        ha, va = hc.copy(), vc.copy()
        hb, vb = hc.copy(), vc.copy()
        na = o.lfo_hquad_bind_h(field, n, P(ha), P(va), elt(rr), hand)
        nb = r.ref_hquad_bind_h(field, n, P(hb), P(vb), P(rr), hand)
        assert na == nb
        assert (ha[:na] == hb[:nb]).all() and (va[:na] == vb[:nb]).all()
        hc, vc, n = ha[:na].copy(), va[:na].copy(), na
        hand = 1 - hand


def _morton(a, b):
    m = 0
    for i in range(16):
        m |= ((a >> i) & 1) << (2 * i) | ((b >> i) & 1) << (2 * i + 1)
    return m


@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("logv,logw,nterms,n_assert", [(0, 1, 1, 0), (3, 4, 40, 5), (8, 10, 5000, 300), (10, 9, 9000, 0)])
def test_eval_quad_bind_g_raw_eq2(field, logv, logw, nterms, n_assert):
    import quad_util as qu
    o, r = ol.oracle(), ol.ref()
    rng = np.random.default_rng(logv * 100 + logw + field)
    nterms = min(nterms, (1 << logv) * (1 << logw) // 2 + 1)
    L = qu.make_layer(rng, field, logv, logw, nterms, n_assert=min(n_assert, nterms // 2))
    # eval_quad (assertions satisfied)
    Va = np.zeros((L["nv"], 2), dtype=np.uint64)
    Vb = np.zeros((L["nv"], 2), dtype=np.uint64)
    oka = o.lfo_eval_quad(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), P(L["kvec"]), L["nv"], P(L["W"]), P(Va))
    okb = r.ref_eval_quad(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), len(L["kvec"]), P(L["kvec"]),
                          L["nv"], L["nw"], P(L["W"]), P(Vb))
    assert oka == okb == 1 and (Va == Vb).all()
    if n_assert:  # violated assertion: both must report failure
        W2 = ol.rand_elts(rng, L["nw"], field)
        W2[W2[:, 0] == 0, 0] = 1
        oka = o.lfo_eval_quad(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), P(L["kvec"]), L["nv"], P(W2), P(Va))
        okb = r.ref_eval_quad(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), len(L["kvec"]), P(L["kvec"]),
                              L["nv"], L["nw"], P(W2), P(Vb))
        assert oka == okb == 0
    # raw_eq2 + bind_g
    G0, G1 = ol.rand_elts(rng, max(1, logv), field), ol.rand_elts(rng, max(1, logv), field)
    alpha, beta = ol.rand_elts(rng, 1, field)[0], ol.rand_elts(rng, 1, field)[0]
    for n in (L["nv"], max(1, L["nv"] - 3)):
        ea = np.zeros((n, 2), dtype=np.uint64)
        eb = np.zeros((n, 2), dtype=np.uint64)
        o.lfo_raw_eq2(field, logv, n, P(G0), P(G1), elt(alpha), P(ea))
        r.ref_raw_eq2(field, logv, n, P(G0), P(G1), P(alpha), P(eb))
        assert (ea == eb).all()
    ha, va = np.zeros((L["n"], 2), dtype=np.uint32), np.zeros((L["n"], 2), dtype=np.uint64)
    hb, vb = np.zeros((L["n"], 2), dtype=np.uint32), np.zeros((L["n"], 2), dtype=np.uint64)
    na = o.lfo_quad_bind_g(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), P(L["kvec"]), logv, P(G0), P(G1),
                           elt(alpha), elt(beta), P(ha), P(va))
    nb = r.ref_quad_bind_g(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), len(L["kvec"]), P(L["kvec"]), logv,
                           P(G0), P(G1), P(alpha), P(beta), P(hb), P(vb))
    assert na == nb and (ha[:na] == hb[:nb]).all() and (va[:na] == vb[:nb]).all()


# ---------------------------------------------------------------- F64_2 = Fp2<Fp<1>>, p = 2^64 - 2^32 + 1
F64_P = 2**64 - 2**32 + 1


def _f64_2_edge_and_random(n, seed):
    rng = np.random.default_rng(seed)
    edge = [0, 1, 2, 0xFFFFFFFF, 1 << 32, F64_P - 1, F64_P - 2, (F64_P - 1) // 2, 0xFFFFFFFF00000000]
    vals = edge + [int(x) % F64_P for x in rng.integers(0, 2**63, size=n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=n, dtype=np.uint64)]
    return vals


def test_f64_2_arithmetic():
    """the oracle's one-limb Montgomery arithmetic and Fp2 against Fp2<Fp<1>> of the reference, edge values included"""
    o, r = ol.oracle(), ol.ref()
    vals = _f64_2_edge_and_random(40, 5)
    out = np.zeros(2, dtype=np.uint64)
    for i, re in enumerate(vals):
        a = np.array([re, vals[(i * 7 + 3) % len(vals)]], dtype=np.uint64)
        b = np.array([vals[(i * 5 + 1) % len(vals)], vals[(i * 11 + 2) % len(vals)]], dtype=np.uint64)
        for op, fn in ((0, o.lfo_f64_2_add), (1, o.lfo_f64_2_sub), (2, o.lfo_f64_2_mul)):
            r.ref_f64_2_binop(op, P(a), P(b), P(out))
            assert (arr(fn(elt(a), elt(b))) == out).all(), (op, a, b)
        if a.any():
            r.ref_f64_2_binop(3, P(a), P(b), P(out))
            assert (arr(o.lfo_f64_2_inv(elt(a))) == out).all()
    for u in (0, 1, 2, 12345, F64_P - 1):
        r.ref_f64_2_of_scalar(u, (u * 3) % F64_P, P(out))
        assert int(out[0]) == o.lfo_f64_of_scalar(u) and int(out[1]) == o.lfo_f64_of_scalar((u * 3) % F64_P)
        assert o.lfo_f64_from_mont(o.lfo_f64_of_scalar(u)) == u
    r.ref_f64_2_omega32(P(out))
    assert int(out[0]) == o.lfo_f64_omega32() and int(out[1]) == 0
    a = np.zeros((50, 2), dtype=np.uint64)
    b = np.zeros((50, 2), dtype=np.uint64)
    for imag in (0, 1):
        o.lfo_f64_2_bogorng_fill(1234569, imag, 50, P(a))
        r.ref_f64_2_bogorng_fill(1234569, imag, 50, P(b))
        assert (a == b).all()


@pytest.mark.parametrize("n", [1, 2, 4, 64, 1024, 1 << 14, 1 << 15, 1 << 17])
def test_f64_2_fft(n):
    """FFT<Fp2<Fp<1>>>::fftb / fftf (lib/algebra/fft_test.cc:205-229) -- basecase and the recursive path -- with the
    real root of the reference's test and with a root that has an imaginary part (omega_real * i has order 2^32 too)"""
    o, r = ol.oracle(), ol.ref()
    a = np.zeros((n, 2), dtype=np.uint64)
    o.lfo_f64_2_bogorng_fill(1234569 + n, 1, n, P(a))
    w_real = np.array([o.lfo_f64_omega32(), 0], dtype=np.uint64)
    w_cplx = arr(o.lfo_f64_2_mul(elt(w_real), elt(np.array([0, o.lfo_f64_of_scalar(1)], dtype=np.uint64))))
    for w in (w_real, w_cplx):
        for d, fn in ((0, o.lfo_f64_2_fftb), (1, o.lfo_f64_2_fftf)):
            x, y = a.copy(), a.copy()
            fn(P(x), n, elt(w), 1 << 32)
            r.ref_f64_2_fft(d, n, P(w), P(y))
            assert (x == y).all()
