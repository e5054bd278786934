"""GPU parity: every kernel, called through the C ABI, against the CPU oracle on the same
seeded inputs -- bit-exact (integer / GF(2) arithmetic, no tolerance)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import FP, GF, P, elt, arr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


# ---------------------------------------------------------------- a1 / a2 field arithmetic
@pytest.mark.parametrize("field", [GF, FP])
def test_field_binops(G, field):
    """device add/sub/mul (hand-written carry chains / Kronecker clmul) incl. edge values"""
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(3 + field)
    n = 20000
    x, y = ol.rand_elts(rng, n, field), ol.rand_elts(rng, n, field)
    pm1 = [0, 0xFFFFF00000000000]
    edge = [[0, 0], [1, 0], pm1, [0xFFFFFFFFFFFFFFFF, 0xFFFFEFFFFFFFFFFF], [0xFFFFFFFFFFFFFFFF, 0], [0, 1],
            [0, 0xFFFFF00000000000 - 1], [0xFFFFFFFF, 0], [0x100000000, 0], [0, 0xFFFFEFFF00000000],
            # the one-step REDC folds t0 << 12 into limb 3 and takes a carry out of T_lo + m: low 20 bits zero / all ones,
            # limb-3 overflow, T_lo = 0
            [1 << 20, 0], [(1 << 20) - 1, 0], [0xFFFFF, 0xFFFFFFFF00000000], [0, 1 << 44], [0xFFF00000, 0xFFF0000000000000]]
    if field == GF:
        edge += [[0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF], [0, 0x8000000000000000]]
    k = 0
    for a in edge:
        for b in edge:
            x[k], y[k] = a, b
            k += 1
    dx, dy = G.to_dev(x), G.to_dev(y)
    dout = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    for op, fn in ((0, o.lfo_add), (1, o.lfo_sub), (2, o.lfo_mul)):
        G.gpu().field_binop(field, op, n, dx.data_ptr(), dy.data_ptr(), dout.data_ptr())
        got = G.from_dev(dout, np.uint64, (n, 2))
        for i in list(range(k)) + list(range(k, n, 37)):
            assert (got[i] == arr(fn(field, elt(x[i]), elt(y[i])))).all(), (op, i)


# ---------------------------------------------------------------- K1 Fp128 FFT
@pytest.mark.parametrize("n,rows", [(2, 1), (4, 3), (64, 5), (1024, 2), (8192, 3), (1 << 14, 2), (1 << 16, 1), (1 << 17, 2)])
@pytest.mark.parametrize("forward", [False, True])
def test_fp128_fft(G, n, rows, forward):
    o = ol.oracle()
    a = np.zeros((rows, n, 2), dtype=np.uint64)
    for r in range(rows):
        o.lfo_fp_bogorng_fill(1234569 + r, n, P(a[r]))
    want = a.copy()
    for r in range(rows):
        (o.lfo_fp_fftf if forward else o.lfo_fp_fftb)(P(want[r]), n, o.lfo_fp_omega32(), 1 << 32)
    d = G.to_dev(a)
    G.gpu().fp128_fft(d.data_ptr(), rows, n, forward=forward)
    got = G.from_dev(d, np.uint64, (rows, n, 2))
    assert (got == want).all()


def test_fp128_fft_strided_rows_and_noop(G):
    o = ol.oracle()
    n, ld, rows = 256, 300, 4
    a = np.zeros((rows, ld, 2), dtype=np.uint64)
    o.lfo_fp_bogorng_fill(99, rows * ld, P(a))
    want = a.copy()
    for r in range(rows):
        o.lfo_fp_fftb(P(want[r]), n, o.lfo_fp_omega32(), 1 << 32)
    d = G.to_dev(a)
    G.gpu().fp128_fft(d.data_ptr(), rows, n, ld=ld)
    assert (G.from_dev(d, np.uint64, (rows, ld, 2)) == want).all()
    # n = 1 and rows = 0 are no-ops (fft.h:187 `if (n > 1)`)
    G.gpu().fp128_fft(d.data_ptr(), rows, 1, ld=ld)
    G.gpu().fp128_fft(d.data_ptr(), 0, n, ld=ld)
    assert (G.from_dev(d, np.uint64, (rows, ld, 2)) == want).all()
    with pytest.raises(G.pkg.LfGpuError):
        G.gpu().fp128_fft(d.data_ptr(), 1, 96, ld=ld)  # not a power of two


def test_fp128_fft_roundtrip_full_size(G):
    """size-independent property at a BASELINE-sized row: fftb(fftf(x)) = n*x (fft_test.cc:56-72)"""
    import torch
    o = ol.oracle()
    n = 1 << 20
    a = np.zeros((n, 2), dtype=np.uint64)
    o.lfo_fp_bogorng_fill(1234569, n, P(a))
    d = G.to_dev(a)
    G.gpu().fp128_fft(d.data_ptr(), 1, n, forward=True)
    G.gpu().fp128_fft(d.data_ptr(), 1, n, forward=False)
    got = G.from_dev(d, np.uint64, (n, 2))
    nn = o.lfo_fp_of_scalar(n)
    for i in list(range(0, n, 65521))[:64] + [n - 1]:
        assert (got[i] == arr(o.lfo_fp_mul(elt(a[i]), nn))).all()


def test_fp128_fft_full_row_2pow20_vs_oracle(G):
    """one whole BASELINE-sized row (n = 2^20, the two-pass plan) compared element by element with the oracle's fftb"""
    o = ol.oracle()
    n = 1 << 20
    a = np.zeros((2, n, 2), dtype=np.uint64)
    for r in range(2):
        o.lfo_fp_bogorng_fill(1234569 + r, n, P(a[r]))
    want = a.copy()
    for r in range(2):
        o.lfo_fp_fftb(P(want[r]), n, o.lfo_fp_omega32(), 1 << 32)
    d = G.to_dev(a)
    G.gpu().fp128_fft(d.data_ptr(), 2, n)
    assert (G.from_dev(d, np.uint64, (2, n, 2)) == want).all()


@pytest.mark.parametrize("logn,rows", [(21, 2), (22, 1), (23, 1)])
def test_fp128_fft_beyond_2pow20(G, logn, rows):
    """n > 2^20 (the reference benchmarks BM_FFT_Fp128/4194304; fft.h:185-201 takes any power of two under the root's
    order): one more tile pass in front of the two-pass plan.  Whole rows against the oracle, then fftf(fftb(x)) = n x."""
    o = ol.oracle()
    n = 1 << logn
    a = np.zeros((rows, n, 2), dtype=np.uint64)
    for r in range(rows):
        o.lfo_fp_bogorng_fill(777 + r + logn, n, P(a[r]))
    want = a.copy()
    for r in range(rows):
        o.lfo_fp_fftb(P(want[r]), n, o.lfo_fp_omega32(), 1 << 32)
    d = G.to_dev(a)
    G.gpu().fp128_fft(d.data_ptr(), rows, n)
    assert (G.from_dev(d, np.uint64, a.shape) == want).all()
    G.gpu().fp128_fft(d.data_ptr(), rows, n, forward=True)
    got = G.from_dev(d, np.uint64, a.shape)
    nn = o.lfo_fp_of_scalar(n)
    for i in list(range(0, n, 262139))[:40] + [n - 1]:
        assert (got[0, i] == arr(o.lfo_fp_mul(elt(a[0, i]), nn))).all()


# ---------------------------------------------------------------- K1 over F64_2 = Fp2<Fp<1>> (fft_test.cc:205-229)
def _f64_2_roots(o):
    w_real = np.array([o.lfo_f64_omega32(), 0], dtype=np.uint64)
    w_cplx = arr(o.lfo_f64_2_mul(elt(w_real), elt(np.array([0, o.lfo_f64_of_scalar(1)], dtype=np.uint64))))  # omega * i: order 2^32 too
    return {"real": w_real, "times_i": w_cplx}


@pytest.mark.parametrize("n,rows", [(2, 1), (4, 3), (64, 5), (1024, 2), (4096, 9), (8192, 3), (1 << 14, 2), (1 << 16, 1), (1 << 17, 2)])
@pytest.mark.parametrize("forward", [False, True])
@pytest.mark.parametrize("root", ["real", "times_i"])
def test_f64_2_fft(G, n, rows, forward, root):
    """FFT<Fp2<Fp<1>>>::fftb / fftf against the oracle: the real root of the reference's test (two base-field products
    per twiddle) and a root outside the base field (the general three-product Fp2 multiplication)"""
    o = ol.oracle()
    w = _f64_2_roots(o)[root]
    a = np.zeros((rows, n, 2), dtype=np.uint64)
    for r in range(rows):
        o.lfo_f64_2_bogorng_fill(1234569 + r, 1, n, P(a[r]))
    want = a.copy()
    for r in range(rows):
        (o.lfo_f64_2_fftf if forward else o.lfo_f64_2_fftb)(P(want[r]), n, elt(w), 1 << 32)
    d = G.to_dev(a)
    G.gpu().f64_2_fft(d.data_ptr(), rows, n, forward=forward, omega=(int(w[0]), int(w[1])))
    assert (G.from_dev(d, np.uint64, (rows, n, 2)) == want).all()


def test_f64_2_fft_golden_through_the_device(G):
    """tests/golden/f64_2.json (outputs of the reference itself, oracle/gen_golden_f64.py) through the HIP path"""
    import hashlib
    import json
    o = ol.oracle()
    with open(os.path.join(ol.ROOT, "tests", "golden", "f64_2.json")) as f:
        g64 = json.load(f)
    roots = {k: np.frombuffer(bytes.fromhex(g64[k2]), dtype=np.uint64) for k, k2 in (("real", "omega32"), ("times_i", "omega32_times_i"))}
    for v in g64["fft"]:
        a = np.zeros((v["n"], 2), dtype=np.uint64)
        o.lfo_f64_2_bogorng_fill(v["bogorng_seed"], v["imag"], v["n"], P(a))
        w = roots[v["root"]]
        G.gpu().f64_2_fft_host(a, forward=bool(v["dir"]), omega=(int(w[0]), int(w[1])))
        assert hashlib.sha256(a.tobytes()).hexdigest() == v["out_sha256"], v["n"]


def test_f64_2_fft_strided_rows_noop_and_errors(G):
    o = ol.oracle()
    n, ld, rows = 256, 300, 4
    a = np.zeros((rows, ld, 2), dtype=np.uint64)
    o.lfo_f64_2_bogorng_fill(99, 1, rows * ld, P(a))
    w = _f64_2_roots(o)["real"]
    want = a.copy()
    for r in range(rows):
        o.lfo_f64_2_fftb(P(want[r]), n, elt(w), 1 << 32)
    d = G.to_dev(a)
    G.gpu().f64_2_fft(d.data_ptr(), rows, n, ld=ld)
    assert (G.from_dev(d, np.uint64, (rows, ld, 2)) == want).all()
    G.gpu().f64_2_fft(d.data_ptr(), rows, 1, ld=ld)
    G.gpu().f64_2_fft(d.data_ptr(), 0, n, ld=ld)
    assert (G.from_dev(d, np.uint64, (rows, ld, 2)) == want).all()
    with pytest.raises(G.pkg.LfGpuError):
        G.gpu().f64_2_fft(d.data_ptr(), 1, 96, ld=ld)  # not a power of two
    with pytest.raises(G.pkg.LfGpuError):
        G.gpu().f64_2_fft(d.data_ptr(), 1, n, ld=ld, omega=(2**64 - 1, 0))  # not a reduced field element


@pytest.mark.parametrize("logn,rows,root", [(20, 2, "real"), (20, 1, "times_i"), (22, 1, "real")])
def test_f64_2_fft_benchmark_sizes_vs_oracle(G, logn, rows, root):
    """the sizes of BM_FFT_F64_2 the two- and three-pass plans serve (2^20, 2^22 = the benchmark's largest), whole rows
    against the oracle, then fftf(fftb(x)) = n x"""
    o = ol.oracle()
    w = _f64_2_roots(o)[root]
    n = 1 << logn
    a = np.zeros((rows, n, 2), dtype=np.uint64)
    for r in range(rows):
        o.lfo_f64_2_bogorng_fill(4242 + r + logn, 1, n, P(a[r]))
    want = a.copy()
    for r in range(rows):
        o.lfo_f64_2_fftb(P(want[r]), n, elt(w), 1 << 32)
    d = G.to_dev(a)
    om = (int(w[0]), int(w[1]))
    G.gpu().f64_2_fft(d.data_ptr(), rows, n, omega=om)
    assert (G.from_dev(d, np.uint64, a.shape) == want).all()
    G.gpu().f64_2_fft(d.data_ptr(), rows, n, forward=True, omega=om)
    got = G.from_dev(d, np.uint64, a.shape)
    nn = o.lfo_f64_of_scalar(n)
    for i in list(range(0, n, 65521))[:64] + [n - 1]:
        assert int(got[0, i, 0]) == o.lfo_f64_mul(int(a[0, i, 0]), nn) and int(got[0, i, 1]) == o.lfo_f64_mul(int(a[0, i, 1]), nn)


@pytest.mark.parametrize("l,rows", [(21, 1), (22, 2), (21, 32), (22, 32)])
def test_lch14_fft_beyond_2pow20(G, l, rows):
    """l > 20 (GF2_128<5>): the tile plan for small batches, the bit-sliced passes for >= 32 rows; rows 0 and last
    against the oracle and the IFFT round trip"""
    o = ol.oracle()
    c = ol.gf_ctx(5)
    rng = np.random.default_rng(l * 31 + rows)
    a = ol.rand_elts(rng, rows << l).reshape(rows, 1 << l, 2)
    d = G.to_dev(a)
    G.gpu().gf2128_lch14_fft(d.data_ptr(), rows, l, subfield_log_bits=5)
    got = G.from_dev(d, np.uint64, a.shape).copy()
    for r in sorted({0, rows - 1}):
        w = a[r].copy()
        o.lfo_lch14_fft(C.byref(c), l, 0, P(w))
        assert (got[r] == w).all()
    G.gpu().gf2128_lch14_fft(d.data_ptr(), rows, l, inverse=True, subfield_log_bits=5)
    assert (G.from_dev(d, np.uint64, a.shape) == a).all()


# ---------------------------------------------------------------- K2 LCH14 FFT
@pytest.mark.parametrize("k,l,coset,rows", [(4, 1, 0, 3), (4, 4, 16, 2), (4, 10, 0, 3), (4, 10, 5 << 10, 9), (4, 13, 0, 2),
                                            (4, 14, 0, 2), (4, 16, 0, 1), (5, 17, 3 << 17, 1), (5, 11, 1 << 11, 4)])
@pytest.mark.parametrize("inverse", [False, True])
def test_lch14_fft(G, k, l, coset, rows, inverse):
    o = ol.oracle()
    c = ol.gf_ctx(k)
    rng = np.random.default_rng(l * 7 + k)
    a = ol.rand_elts(rng, rows << l).reshape(rows, 1 << l, 2)
    want = a.copy()
    for r in range(rows):
        (o.lfo_lch14_ifft if inverse else o.lfo_lch14_fft)(C.byref(c), l, coset, P(want[r]))
    d = G.to_dev(a)
    G.gpu().gf2128_lch14_fft(d.data_ptr(), rows, l, coset=coset, inverse=inverse, subfield_log_bits=k)
    assert (G.from_dev(d, np.uint64, a.shape) == want).all()


@pytest.mark.parametrize("k,l,coset,rows", [(4, 7, 0, 32), (4, 7, 128, 40), (4, 8, 0, 64), (4, 10, 3 << 10, 150), (4, 11, 0, 33),
                                            (4, 13, 1 << 13, 32), (5, 14, 0, 64), (5, 9, 7 << 9, 96)])
@pytest.mark.parametrize("inverse", [False, True])
def test_lch14_fft_bitsliced_batches(G, k, l, coset, rows, inverse):
    _bitsliced_batch(G, k, l, coset, rows, inverse)


@pytest.mark.parametrize("k,l,coset,rows", [(5, 8, 0, 512), (5, 13, 5 << 13, 544), (4, 10, 0, 288), (5, 7, 1 << 7, 1024), (4, 9, 3 << 9, 256),
                                            (5, 11, 0, 2048), (5, 17, 0, 512), (4, 14, 1 << 14, 300)])
@pytest.mark.parametrize("inverse", [False, True])
def test_lch14_fft_bitsliced_register_resident(G, k, l, coset, rows, inverse):
    """batches that fill whole 64-slot waves (>= 64 (row-group, coordinate) combos: 512 rows for GF2_128<5>, 256 for <4>)
    take the register-resident butterfly kernel (bs_bfly2_kernel): full and short last groups (l mod 4 = 0..3), ragged row
    groups, cosets, both directions; rows sampled against the oracle"""
    _bitsliced_batch(G, k, l, coset, rows, inverse, sample=True)


def _bitsliced_batch(G, k, l, coset, rows, inverse, sample=False):
    """>= 32 rows take the bit-sliced tower path (lch_bs.hip), incl. ragged row groups"""
    o = ol.oracle()
    c = ol.gf_ctx(k)
    rng = np.random.default_rng(l * 13 + k + rows)
    ld = (1 << l) + 5
    a = ol.rand_elts(rng, rows * ld).reshape(rows, ld, 2)
    d = G.to_dev(a)
    G.gpu().gf2128_lch14_fft(d.data_ptr(), rows, l, coset=coset, ld=ld, inverse=inverse, subfield_log_bits=k)
    got = G.from_dev(d, np.uint64, a.shape)
    check = sorted({0, 1, 31, 32, 33, rows // 2, rows - 33, rows - 2, rows - 1} & set(range(rows))) if sample else range(rows)
    for r in check:
        want = a[r].copy()
        (o.lfo_lch14_ifft if inverse else o.lfo_lch14_fft)(C.byref(c), l, coset, P(want))
        assert (got[r] == want).all(), r
    if sample:  # the untouched tail of every row and a checksum over all rows against a second run through the LDS-tile kernel
        assert (got[:, 1 << l:] == a[:, 1 << l:]).all()


@pytest.mark.parametrize("l,coset", [(20, 0), (17, 1 << 17), (18, 0), (19, 5 << 19), (15, 0), (16, 3 << 16)])
def test_lch14_fft_bitsliced_full_size_rows(G, l, coset):
    """l = 15 .. 20 (GF2_128<5>), 32 rows: row 0 and row 31 against the oracle, then IFFT round trip.  l = 17, 18, 19 end
    with a SHORT last butterfly group (1, 2, 3 index bits): the sizes the truncated transform of big Reed-Solomon rows
    calls (a stage-table read past the table's end faulted there at l = 17, 18)."""
    o = ol.oracle()
    c = ol.gf_ctx(5)
    rng = np.random.default_rng(2000 + l)
    rows = 32
    a = ol.rand_elts(rng, rows << l).reshape(rows, 1 << l, 2)
    d = G.to_dev(a)
    G.gpu().gf2128_lch14_fft(d.data_ptr(), rows, l, coset=coset, subfield_log_bits=5)
    got = G.from_dev(d, np.uint64, a.shape).copy()
    for r in (0, 31):
        w = a[r].copy()
        o.lfo_lch14_fft(C.byref(c), l, coset, P(w))
        assert (got[r] == w).all()
    G.gpu().gf2128_lch14_fft(d.data_ptr(), rows, l, coset=coset, inverse=True, subfield_log_bits=5)
    assert (G.from_dev(d, np.uint64, a.shape) == a).all()


def test_lch14_fft_roundtrip_full_size(G):
    """IFFT(FFT(x)) = x at l = 20 (GF2_128<5>), the BASELINE row length"""
    rng = np.random.default_rng(20)
    a = ol.rand_elts(rng, 1 << 20)
    d = G.to_dev(a)
    G.gpu().gf2128_lch14_fft(d.data_ptr(), 1, 20, subfield_log_bits=5)
    mid = G.from_dev(d, np.uint64, a.shape).copy()
    assert not (mid == a).all()
    G.gpu().gf2128_lch14_fft(d.data_ptr(), 1, 20, inverse=True, subfield_log_bits=5)
    assert (G.from_dev(d, np.uint64, a.shape) == a).all()
    with pytest.raises(G.pkg.LfGpuError):
        G.gpu().gf2128_lch14_fft(d.data_ptr(), 1, 17, subfield_log_bits=4)  # l <= kSubFieldBits (lch14.h:107)


# ---------------------------------------------------------------- K3 / K4 RS rows
@pytest.mark.parametrize("k,n,m,nrow", [(4, 1, 7, 2), (4, 5, 5, 2), (4, 21, 128, 8), (4, 100, 128, 3), (4, 64, 64, 1),
                                        (4, 64, 300, 2), (4, 455, 4096, 20), (4, 909, 4096, 3), (4, 910, 8192, 5),
                                        (4, 1819, 8192, 2), (4, 682, 4096, 8), (4, 461, 4151, 3), (5, 1000, 5000, 2),
                                        # rows larger than LDS (2^l > 4096): global-memory op sweeps + batched coset FFTs
                                        (4, 4097, 8192, 2), (4, 5000, 20000, 3), (4, 7279, 32768, 40), (5, 9000, 70000, 2),
                                        (4, 8192, 8192 + 5, 1), (4, 16384, 65536, 33),
                                        # the S-lig shape (LigeroParam(nw, 0, 4, 132, 2^20) over GF2_128<5>) scaled by 1/4: block 43690, all cosets fit
                                        (5, 43690, 262144, 36),
                                        # block transforms of 2^17 points and short last butterfly groups inside the truncated transform
                                        (5, 174762, 1 << 20, 32),
                                        # >= 32 rows: the whole encoder in the tower representation (one conversion in, one out per coset);
                                        # a partial last coset, rows that are not a multiple of 32, both subfields, n = 2^l
                                        (4, 5000, 20000, 33), (5, 5000, 20000, 70), (4, 8192, 3 * 8192 + 100, 64), (5, 4100, 8192, 129),
                                        # >= 64 (row group, coordinate) combos: the register-resident butterfly kernel on sub-blocks, out of
                                        # place for the further cosets, workgroups of 1 / 2 / 4 / 8 waves for the short groups
                                        (5, 5000, 20000, 512), (4, 4100, 8192 + 100, 256), (5, 2731, 16384, 544)])
def test_gf2128_rs_encode_rows(G, k, n, m, nrow):
    o = ol.oracle()
    rng = np.random.default_rng(n * 3 + m)
    ld = m + 3
    T = ol.rand_elts(rng, nrow * ld).reshape(nrow, ld, 2)
    want = T.copy()
    for r in range(nrow):
        o.lfo_lch14_rs_interpolate(C.byref(ol.gf_ctx(k)), n, m, P(want[r]))
    d = G.to_dev(T)
    G.gpu().gf2128_rs_encode_rows(d.data_ptr(), nrow, n, m, ld=ld, subfield_log_bits=k)
    assert (G.from_dev(d, np.uint64, T.shape) == want).all()


@pytest.mark.parametrize("k,nrow,n1,n2,lo2,hi2,m", [(4, 20, 455, 909, 1, 3, 4096), (4, 150, 910, 1819, 1, 3, 8192), (4, 7, 21, 64, 0, 7, 128),
                                                     (4, 5, 100, 33, 2, 2, 300), (5, 9, 300, 500, 4, 9, 70000 // 16)])
def test_gf2128_rs_encode_tableau_equals_row_groups(G, k, nrow, n1, n2, lo2, hi2, m):
    """the single-launch tableau encode (rows [lo2, hi2) are n2 long) == the oracle row by row"""
    o = ol.oracle()
    rng = np.random.default_rng(nrow * 7 + m)
    ld = m + 1
    T = ol.rand_elts(rng, nrow * ld).reshape(nrow, ld, 2)
    want = T.copy()
    for r in range(nrow):
        o.lfo_lch14_rs_interpolate(C.byref(ol.gf_ctx(k)), n2 if lo2 <= r < hi2 else n1, m, P(want[r]))
    d = G.to_dev(T)
    G.gpu().gf2128_rs_encode_tableau(d.data_ptr(), nrow, n1, n2, lo2, hi2, m, ld=ld, subfield_log_bits=k)
    assert (G.from_dev(d, np.uint64, T.shape) == want).all()


@pytest.mark.parametrize("n,m,nrow", [(1, 4, 2), (3, 8, 2), (21, 128, 4), (100, 257, 3), (455, 4096, 2), (910, 8192, 3),
                                      (10922, 1 << 16, 2), (5000, 1 << 16, 1), (174762, 1 << 20, 1)])
def test_fp128_rs_encode_rows(G, n, m, nrow):
    o = ol.oracle()
    T = np.zeros((nrow, m, 2), dtype=np.uint64)
    o.lfo_fp_bogorng_fill(5 + n, nrow * m, P(T))
    want = T.copy()
    for r in range(nrow):
        o.lfo_fp_rs_interpolate(n, m, P(want[r]))
    d = G.to_dev(T)
    G.gpu().fp128_rs_encode_rows(d.data_ptr(), nrow, n, m)
    assert (G.from_dev(d, np.uint64, T.shape) == want).all()


# ---------------------------------------------------------------- K10 Eqs::raw_eq2
@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("logn,n", [(0, 1), (1, 2), (3, 5), (7, 128), (11, 1500), (17, 111000), (20, 1 << 20)])
def test_raw_eq2(G, field, logn, n):
    """lfgpu_raw_eq2 == Eqs::raw_eq2 (lib/arrays/eqs.h:46-80): called directly (the ZK driver and bind_g reach it
    through other entry points), ragged n < 2^logn included"""
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(100 * logn + field)
    G0, G1 = ol.rand_elts(rng, max(logn, 1), field), ol.rand_elts(rng, max(logn, 1), field)
    alpha = ol.rand_elts(rng, 1, field)[0]
    want = np.zeros((n, 2), dtype=np.uint64)
    o.lfo_raw_eq2(field, logn, n, P(G0), P(G1), elt(alpha), P(want))
    d = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    G.gpu().raw_eq2(field, logn, n, G0, G1, alpha, d.data_ptr())
    assert (G.from_dev(d, np.uint64, (n, 2)) == want).all()


_EQ_FUSED_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import gpu_util as G, oracle_lib as ol
o = ol.oracle()
for field in (ol.GF, ol.FP):
    for logn, n in ((6, 64), (6, 33), (7, 128), (11, 1500), (13, 8192), (16, 40001), (16, 1 << 16)):
        rng = np.random.default_rng(100 * logn + field + n)
        G0, G1 = ol.rand_elts(rng, logn, field), ol.rand_elts(rng, logn, field)
        alpha = ol.rand_elts(rng, 1, field)[0]
        want = np.zeros((n, 2), dtype=np.uint64)
        o.lfo_raw_eq2(field, logn, n, ol.P(G0), ol.P(G1), ol.elt(alpha), ol.P(want))
        d = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
        G.gpu().raw_eq2(field, logn, n, G0, G1, alpha, d.data_ptr())
        assert (G.from_dev(d, np.uint64, (n, 2)) == want).all(), (field, logn, n)
print('OK')
"""


def test_raw_eq2_fused_launch():
    """raw_eq2_fused_kernel (factor tables + combination in one launch; the default only in throughput mode)
    == Eqs::raw_eq2, both fields, ragged n, up to its limit of 2^16 entries.  The switch is read once per process: a child."""
    import os, subprocess, sys
    e = dict(os.environ, LFGPU_EQ_FUSED="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", _EQ_FUSED_CHILD, here], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---------------------------------------------------------------- K5 / K6 Merkle
@pytest.mark.parametrize("field,nrow,ld,col0,ncols", [(GF, 1, 8, 0, 8), (GF, 2, 8, 1, 1), (GF, 20, 4096, 909, 3187),
                                                      (GF, 150, 8192, 1819, 6373), (GF, 7, 64, 13, 51),
                                                      (FP, 5, 40, 9, 31), (FP, 19, 4096, 909, 3187), (FP, 3, 3, 1, 2)])
def test_column_commit(G, field, nrow, ld, col0, ncols):
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(nrow * ld)
    T = ol.rand_elts(rng, nrow * ld, field)
    nonces = rng.integers(0, 256, size=(ncols, 32), dtype=np.uint8)
    layers = np.zeros((2 * ncols, 32), dtype=np.uint8)
    root = np.zeros(32, dtype=np.uint8)
    o.lfo_column_commit(field, nrow, ld, col0, ncols, P(T), P(nonces), P(root), P(layers))
    dT, dN = G.to_dev(T), G.to_dev(nonces)
    dL = torch.zeros(2 * ncols * 32, dtype=torch.uint8, device="cuda")
    got = G.gpu().column_commit(field, nrow, ld, col0, ncols, dT.data_ptr(), dN.data_ptr(), dL.data_ptr())
    assert got == bytes(root)
    gl = G.from_dev(dL, np.uint8, (2 * ncols, 32))
    assert (gl[1:] == layers[1:]).all()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 1000, 2049, 70000])
def test_merkle_build_tree_and_open(G, n):
    o = ol.oracle()
    rng = np.random.default_rng(n)
    leaves = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    layers = np.zeros((2 * n, 32), dtype=np.uint8)
    o.lfo_merkle_build_tree(n, P(leaves), P(layers))
    dev = np.zeros((2 * n, 32), dtype=np.uint8)
    dev[n:] = leaves
    d = G.to_dev(dev)
    root = G.gpu().merkle_build_tree(n, d.data_ptr())
    assert root == bytes(layers[1])
    # compressed opening (merkle_tree.h:122-143) replayed on the host from the oracle's layers
    pos = sorted(set(int(x) for x in rng.integers(0, n, size=min(n, 7))))
    tree = [False] * (2 * n)
    for p_ in pos:
        tree[p_ + n] = True
    for i in range(n - 1, 0, -1):
        tree[i] = tree[2 * i] or tree[2 * i + 1]
    want = []
    for i in range(n - 1, 0, -1):
        if tree[i]:
            ch = 2 * i
            if tree[ch]:
                ch = 2 * i + 1
            if not tree[ch]:
                want.append(bytes(layers[ch]))
    assert G.gpu().merkle_open(n, d.data_ptr(), pos) == want


# ---------------------------------------------------------------- K7 / K8 / K9 sumcheck round
@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("n", [1, 2, 3, 8, 17, 1000, 1001, 1 << 16, (1 << 18) + 1])
def test_sumcheck_partials_and_dense_bind(G, field, n):
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(n + field)
    QW, W = ol.rand_elts(rng, n, field), ol.rand_elts(rng, n, field)
    a0, a2 = ol.Elt(), ol.Elt()
    o.lfo_sumcheck_partials(field, n, P(QW), P(W), C.byref(a0), C.byref(a2))
    dQ, dW = G.to_dev(QW), G.to_dev(W)
    g0, g2 = G.gpu().sumcheck_partials(field, n, dQ.data_ptr(), dW.data_ptr())
    assert g0 == (a0.l[0], a0.l[1]) and g2 == (a2.l[0], a2.l[1])
    r = ol.rand_elts(rng, 1, field)[0]
    want = np.zeros(((n + 1) // 2, 2), dtype=np.uint64)
    o.lfo_dense_bind(field, n, elt(r), P(W), P(want))
    dO = torch.zeros(((n + 1) // 2) * 16, dtype=torch.uint8, device="cuda")
    G.gpu().dense_bind(field, n, (int(r[0]), int(r[1])), dW.data_ptr(), dO.data_ptr())
    assert (G.from_dev(dO, np.uint64, want.shape) == want).all()
    # in place (Dense::bind(r, F) binds *this)
    G.gpu().dense_bind(field, n, (int(r[0]), int(r[1])), dW.data_ptr(), dW.data_ptr())
    assert (G.from_dev(dW, np.uint64, (n, 2))[:(n + 1) // 2] == want).all()


def _morton(a, b):
    m = 0
    for i in range(16):
        m |= ((a >> i) & 1) << (2 * i) | ((b >> i) & 1) << (2 * i + 1)
    return m


@pytest.mark.parametrize("field", [GF, FP])
def test_hquad_bind_h_rounds(G, field):
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(11)
    pts = sorted({(int(a), int(b)) for a, b in rng.integers(0, 512, size=(60000, 2))}, key=lambda p: _morton(*p))
    hc = np.array(pts, dtype=np.uint32)
    n = len(pts)
    vc = ol.rand_elts(rng, n, field)
    hand = 0
    for _ in range(8):
        r = ol.rand_elts(rng, 1, field)[0]
        ha, va = hc.copy(), vc.copy()
        na = o.lfo_hquad_bind_h(field, n, P(ha), P(va), elt(r), hand)
        dh, dv = G.to_dev(hc), G.to_dev(vc)
        dho = torch.zeros(n * 8, dtype=torch.uint8, device="cuda")
        dvo = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
        nb = G.gpu().hquad_bind_h(field, n, dh.data_ptr(), dv.data_ptr(), (int(r[0]), int(r[1])), hand,
                                  dho.data_ptr(), dvo.data_ptr())
        assert nb == na
        assert (G.from_dev(dho, np.uint32, (n, 2))[:nb] == ha[:na]).all()
        assert (G.from_dev(dvo, np.uint64, (n, 2))[:nb] == va[:na]).all()
        hc, vc, n = ha[:na].copy(), va[:na].copy(), na
        hand = 1 - hand


@pytest.mark.parametrize("field", [GF, FP])
def test_qw_scatter(G, field):
    """GF2_128: wave-folded atomic XOR; Fp128: 32-bit limb integer accumulators + one reduction.  A hot target
    (index 0 hit by a third of the terms, partly in contiguous runs) exercises the contended paths."""
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(12 + field)
    nw = 4096
    n = 50000
    hc = rng.integers(0, nw, size=(n, 2)).astype(np.uint32)
    hot = rng.random(n) < 0.33
    hc[hot, 0] = 0
    hc[1000:3000, 0] = 0
    hc[5000:5100, 1] = 7
    vc = ol.rand_elts(rng, n, field)
    W = ol.rand_elts(rng, nw, field)
    for hand in (0, 1):
        want = np.zeros((nw, 2), dtype=np.uint64)
        o.lfo_qw_scatter(field, n, P(hc), P(vc), hand, P(W), nw, P(want))
        dh, dv, dw = G.to_dev(hc), G.to_dev(vc), G.to_dev(W)
        dq = torch.ones(nw * 16, dtype=torch.uint8, device="cuda")
        G.gpu().qw_scatter(field, n, dh.data_ptr(), dv.data_ptr(), hand, dw.data_ptr(), nw, dq.data_ptr())
        assert (G.from_dev(dq, np.uint64, (nw, 2)) == want).all()


# ---------------------------------------------------------------- K12 row combos
@pytest.mark.parametrize("field", [GF, FP])
def test_rows_axpy_and_gather(G, field):
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(13)
    nrows, n, ld = 17, 910, 1000
    T = ol.rand_elts(rng, nrows * ld, field).reshape(nrows, ld, 2)
    u = ol.rand_elts(rng, nrows, field)
    y = ol.rand_elts(rng, n, field)
    want = y.copy()
    for i in range(nrows):
        o.lfo_axpy(field, n, P(want), elt(u[i]), P(T[i]))
    dT, dy = G.to_dev(T), G.to_dev(y)
    G.gpu().rows_axpy(field, nrows, n, dy.data_ptr(), u, dT.data_ptr(), ld)
    assert (G.from_dev(dy, np.uint64, (n, 2)) == want).all()
    idx = [int(x) for x in rng.choice(ld - 50, size=36, replace=False)]
    dreq = torch.zeros(nrows * 36 * 16, dtype=torch.uint8, device="cuda")
    G.gpu().gather_columns(nrows, ld, 50, dT.data_ptr(), idx, dreq.data_ptr())
    got = G.from_dev(dreq, np.uint64, (nrows, 36, 2))
    assert (got == T[:, [50 + i for i in idx], :]).all()


def test_host_buffer_wrappers(G):
    o = ol.oracle()
    rng = np.random.default_rng(14)
    a = np.zeros((512, 2), dtype=np.uint64)
    o.lfo_fp_bogorng_fill(3, 512, P(a))
    want = a.copy()
    o.lfo_fp_fftb(P(want), 512, o.lfo_fp_omega32(), 1 << 32)
    G.gpu().fp128_fft_host(a)
    assert (a == want).all()
    b = ol.rand_elts(rng, 256)
    wb = b.copy()
    o.lfo_lch14_fft(C.byref(ol.gf_ctx(4)), 8, 256, P(wb))
    G.gpu().gf2128_lch14_fft_host(b, 8, coset=256)
    assert (b == wb).all()
    nrow, n, m = 6, 21, 128
    T = ol.rand_elts(rng, nrow * m).reshape(nrow, m, 2)
    wT = T.copy()
    for r in range(nrow):
        o.lfo_lch14_rs_interpolate(C.byref(ol.gf_ctx(4)), n, m, P(wT[r]))
    G.gpu().gf2128_rs_encode_rows_host(T, nrow, n, m, m)
    assert (T == wT).all()
    nonces = rng.integers(0, 256, size=(m - 41, 32), dtype=np.uint8)
    root = np.zeros(32, dtype=np.uint8)
    o.lfo_column_commit(GF, nrow, m, 41, m - 41, P(wT), P(nonces), P(root), None)
    assert G.gpu().column_commit_host(GF, T, nrow, m, 41, m - 41, nonces) == bytes(root)


# ---------------------------------------------------------------- reference fixtures through the GPU
def test_gpu_reference_merkle_fixtures(G):
    """docs/specs/testvectors.md:7-22 and rust/runtime/merkle/tests/merkle_test_vector.bin
    (C++-generated) reproduced by the HIP Merkle path: root and compressed proof bytes."""
    import test_oracle_golden as tg
    leaves = np.frombuffer(bytes.fromhex("".join(tg.SPEC_LEAVES)), dtype=np.uint8).reshape(5, 32)
    dev = np.zeros((10, 32), dtype=np.uint8)
    dev[5:] = leaves
    d = G.to_dev(dev)
    assert G.gpu().merkle_build_tree(5, d.data_ptr()).hex() == tg.SPEC_ROOT
    assert [x.hex() for x in G.gpu().merkle_open(5, d.data_ptr(), [1, 3])] == [tg.SPEC_LEAVES[4], tg.SPEC_LEAVES[2], tg.SPEC_LEAVES[0]]
    n, lv, idx, root, proof = tg.read_merkle_fixture()
    dev = np.zeros((2 * n, 32), dtype=np.uint8)
    dev[n:] = lv
    d = G.to_dev(dev)
    assert G.gpu().merkle_build_tree(n, d.data_ptr()) == root
    assert G.gpu().merkle_open(n, d.data_ptr(), idx) == proof


# ---------------------------------------------------------------- Ligero commit / prove pieces
def test_ligero_commit_reference_fixture(G):
    """LigeroProver::commit through the GPU path reproduces the C++ commitment root of
    rust/runtime/ligero/tests/ligero_test_vector.bin: pins RNG draw order, row layout,
    LCH14 RS encode (K3), column hashing (K5) and the Merkle tree (K6) end to end."""
    import ligero_fixture as lf
    v = lf.load()
    pkg = G.pkg
    p = pkg.ligero_param(pkg.FIELD_GF2_128, v["nw"], v["nq"], 4, v["nreq"], 4096)
    assert (p.block, p.dblock, p.nrow, p.r, p.w, p.block_ext) == (682, 1363, 8, 36, 646, 2733)
    rng = lf.LcgRng(100)
    pr = pkg.LigeroProver(G.gpu(), pkg.FIELD_GF2_128, p)
    root = pr.commit(v["W"], v["subfield_boundary"], v["lqc"], rng.bytes)
    assert root == v["root"]
    pr.close()


@pytest.mark.parametrize("field", [GF, FP])
def test_ligero_prove_pieces_vs_oracle(G, field):
    """low_degree_proof / dot_proof / quadratic_proof / compute_req+open against the oracle's
    Blas restatement on the tableau the GPU committed (challenges are random test inputs)."""
    pkg = G.pkg
    o = ol.oracle()
    rng = np.random.default_rng(77 + field)
    nw, nq, nreq, be = 700, 40, 12, 512
    p = pkg.ligero_param(field, nw, nq, 4, nreq, be)
    W = ol.rand_elts(rng, nw, field)
    lqc = []
    zs = rng.choice(np.arange(nw // 2, nw), size=nq, replace=False)
    for i in range(nq):  # W[z] = W[x] * W[y], x, y in the first half, z distinct in the second half
        x, y, z = int(rng.integers(0, nw // 2)), int(rng.integers(0, nw // 2)), int(zs[i])
        W[z] = arr(o.lfo_mul(field, elt(W[x]), elt(W[y])))
        lqc.append((x, y, z))
    seed_rng = np.random.default_rng(5)
    pr = pkg.LigeroProver(G.gpu(), field, p)
    root = pr.commit(W, 0, lqc, lambda n: bytes(seed_rng.integers(0, 256, size=n, dtype=np.uint8)))
    T = G.from_dev(_wrap(G, pr.tableau_ptr(), p.nrow * p.block_enc * 16), np.uint64, (p.nrow, p.block_enc, 2)).copy()
    # every row is a codeword: re-encode the first n columns with the oracle
    for i in range(p.nrow):
        n = p.dblock if i in (p.idot, p.iquad) else p.block
        row = T[i].copy()
        row[n:] = 0
        if field == GF:
            o.lfo_lch14_rs_interpolate(C.byref(ol.gf_ctx(4)), n, p.block_enc, P(row))
        else:
            o.lfo_fp_rs_interpolate(n, p.block_enc, P(row))
        assert (row == T[i]).all(), i
    # low degree
    u = ol.rand_elts(rng, p.nwqrow, field)
    want = T[p.ildt, :p.block].copy()
    for i in range(p.nwqrow):
        o.lfo_axpy(field, p.block, P(want), elt(u[i]), P(np.ascontiguousarray(T[p.iw + i, :p.block])))
    assert (pr.low_degree_proof(u) == want).all()
    # dot
    A = ol.rand_elts(rng, p.nwqrow * p.w, field)
    want = T[p.idot, :p.dblock].copy()
    for i in range(p.nwqrow):
        ext = np.zeros((p.dblock, 2), dtype=np.uint64)
        ext[p.r:p.r + p.w] = A[i * p.w:(i + 1) * p.w]
        if field == GF:
            o.lfo_lch14_rs_interpolate(C.byref(ol.gf_ctx(4)), p.block, p.dblock, P(ext))
        else:
            o.lfo_fp_rs_interpolate(p.block, p.dblock, P(ext))
        o.lfo_vaxpy(field, p.dblock, P(want), P(ext), P(np.ascontiguousarray(T[p.iw + i, :p.dblock])))
    assert (pr.dot_proof(A) == want).all()
    # the same A handed over as dense block (on the device) x scale + sorted sparse terms (inner_product_vector on the device)
    nd = (p.nwqrow * p.w) // 2 + 3
    dense = ol.rand_elts(rng, nd, field)
    scale = ol.rand_elts(rng, 1, field)[0]
    sp_idx = np.sort(rng.choice(p.nwqrow * p.w, size=57, replace=False)).astype(np.uint64)
    sp_idx[0], sp_idx[-1] = 0, p.nwqrow * p.w - 1
    sp_idx = np.unique(sp_idx)
    sp_val = ol.rand_elts(rng, len(sp_idx), field)
    A2 = np.zeros((p.nwqrow * p.w, 2), dtype=np.uint64)
    for t in range(nd):
        A2[t] = arr(o.lfo_mul(field, elt(scale), elt(dense[t])))
    for t, i in enumerate(sp_idx):
        A2[int(i)] = arr(o.lfo_add(field, elt(A2[int(i)]), elt(sp_val[t])))
    d_dense = G.to_dev(dense)
    got = pr.dot_proof_sparse(d_dense.data_ptr(), nd, scale, sp_idx, sp_val)
    assert (got == pr.dot_proof(A2)).all()
    with pytest.raises(pkg.LfGpuError):  # unsorted sparse indices are refused
        pr.dot_proof_sparse(d_dense.data_ptr(), nd, scale, sp_idx[::-1].copy(), sp_val)
    # quadratic
    uq = ol.rand_elts(rng, p.nqtriples, field)
    y = T[p.iquad, :p.dblock].copy()
    iqx, iqy, iqz = p.iq, p.iq + p.nqtriples, p.iq + 2 * p.nqtriples
    for i in range(p.nqtriples):
        for j in range(p.dblock):
            t = o.lfo_sub(field, elt(T[iqz + i, j]), o.lfo_mul(field, elt(T[iqx + i, j]), elt(T[iqy + i, j])))
            y[j] = arr(o.lfo_add(field, elt(y[j]), o.lfo_mul(field, elt(uq[i]), t)))
    y0, y2 = pr.quadratic_proof(uq)
    assert (y[p.r:p.r + p.w] == 0).all()
    assert (y0 == y[:p.r]).all() and (y2 == y[p.block:p.dblock]).all()
    # open
    idx = [int(t) for t in rng.choice(p.block_ext, size=p.nreq, replace=False)]
    req, nonces, path = pr.open(idx)
    assert (req == T[:, [p.dblock + i for i in idx], :]).all()
    pr.close()


def _wrap(G, ptr, nbytes):
    """copy `nbytes` of device memory at raw pointer `ptr` into a torch byte tensor"""
    import torch
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    import ctypes
    G.gpu()._ck(G.gpu().L.lfgpu_memcpy_d2h(G.gpu().h, 0, 0, 0))  # sync the lfgpu stream
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(t.data_ptr(), ptr, nbytes, 3) == 0  # hipMemcpyDeviceToDevice
    return t


# ---------------------------------------------------------------- K10 / K11 quad
@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("logv,logw,nterms,n_assert", [(0, 1, 1, 0), (3, 4, 40, 5), (8, 10, 5000, 300), (12, 13, 60000, 2000)])
def test_eval_quad_and_bind_g(G, field, logv, logw, nterms, n_assert):
    import torch
    import quad_util as qu
    o = ol.oracle()
    rng = np.random.default_rng(logv * 100 + logw + field)
    nterms = min(nterms, (1 << logv) * (1 << logw) // 2 + 1)
    L = qu.make_layer(rng, field, logv, logw, nterms, n_assert=min(n_assert, nterms // 2))
    q = G.pkg.Quad(G.gpu(), field, L["g"], L["h0"], L["h1"], L["vi"], L["kvec"], L["nv"])
    want = np.zeros((L["nv"], 2), dtype=np.uint64)
    assert o.lfo_eval_quad(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), P(L["kvec"]), L["nv"], P(L["W"]), P(want)) == 1
    dW = G.to_dev(L["W"])
    dV = torch.ones(L["nv"] * 16, dtype=torch.uint8, device="cuda")
    assert q.eval(L["nw"], dW.data_ptr(), dV.data_ptr()) is True
    assert (G.from_dev(dV, np.uint64, (L["nv"], 2)) == want).all()
    if L["nw"] > 1:  # a caller that passes fewer wires than the corners index gets an argument error, not an out-of-bounds read
        hmax = int(max(L["h0"].max(), L["h1"].max()))
        with pytest.raises(G.pkg.LfGpuError):
            q.eval(hmax, dW.data_ptr(), dV.data_ptr())
    if n_assert:  # a violated assert-zero term is reported, as eval_quad returns false
        W2 = ol.rand_elts(rng, L["nw"], field)
        W2[W2[:, 0] == 0, 0] = 1
        assert q.eval(L["nw"], G.to_dev(W2).data_ptr(), dV.data_ptr()) is False
    G0, G1 = ol.rand_elts(rng, max(1, logv), field), ol.rand_elts(rng, max(1, logv), field)
    alpha, beta = ol.rand_elts(rng, 1, field)[0], ol.rand_elts(rng, 1, field)[0]
    ha, va = np.zeros((L["n"], 2), dtype=np.uint32), np.zeros((L["n"], 2), dtype=np.uint64)
    na = o.lfo_quad_bind_g(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), P(L["kvec"]), logv, P(G0), P(G1),
                           elt(alpha), elt(beta), P(ha), P(va))
    dh = torch.zeros(L["n"] * 8, dtype=torch.uint8, device="cuda")
    dv = torch.zeros(L["n"] * 16, dtype=torch.uint8, device="cuda")
    nb = q.bind_g(logv, G0, G1, (int(alpha[0]), int(alpha[1])), (int(beta[0]), int(beta[1])), dh.data_ptr(), dv.data_ptr())
    assert nb == na
    assert (G.from_dev(dh, np.uint32, (L["n"], 2))[:nb] == ha[:na]).all()
    assert (G.from_dev(dv, np.uint64, (L["n"], 2))[:nb] == va[:na]).all()
    q.close()


@pytest.mark.gpu
@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("w,r,ld,nrows,ndense,nsparse", [
    (7, 3, 16, 5, 0, 0),        # nothing but the clear
    (7, 3, 16, 5, 35, 0),       # dense block fills every row
    (7, 0, 7, 5, 11, 9),        # no randomness columns, rows back to back, dense ends mid-row
    (64, 13, 200, 9, 300, 40),
    (910, 455, 8192, 3, 2000, 500),  # the flatsha-32 row shape; more sparse terms than one staging slot holds
])
def test_ligero_inner_product_rows(G, field, w, r, ld, nrows, ndense, nsparse):
    """lfgpu_ligero_inner_product_rows against inner_product_vector + layout_Aext restated with the oracle's field
    arithmetic (ligero_param.h:382-430): columns < r + w are overwritten, the columns beyond stay as they were."""
    pkg, o = G.pkg, ol.oracle()
    rng = np.random.default_rng(1000 * w + nsparse + field)
    dense = ol.rand_elts(rng, max(ndense, 1), field)
    scale = ol.rand_elts(rng, 1, field)[0]
    idx = np.sort(rng.choice(nrows * w, size=nsparse, replace=False)).astype(np.uint64) if nsparse else np.zeros(0, np.uint64)
    val = ol.rand_elts(rng, max(nsparse, 1), field)[:nsparse]
    before = ol.rand_elts(rng, nrows * ld, field).reshape(nrows, ld, 2)
    want = before.copy()
    want[:, :r + w] = 0
    for t in range(ndense):
        want[t // w, r + t % w] = arr(o.lfo_mul(field, elt(scale), elt(dense[t])))
    for t in range(nsparse):
        i, j = int(idx[t]) // w, int(idx[t]) % w
        want[i, r + j] = arr(o.lfo_add(field, elt(want[i, r + j]), elt(val[t])))
    d_dense, d_rows = G.to_dev(dense), G.to_dev(before)
    G.gpu().ligero_inner_product_rows(field, w, r, ld, nrows, d_dense.data_ptr(), ndense, scale, idx, val, d_rows.data_ptr())
    got = G.from_dev(d_rows, np.uint64, (nrows, ld, 2))
    assert (got == want).all()
    if nsparse >= 2:  # refused: duplicate / decreasing indices, index out of range
        bad = idx.copy()
        bad[1] = bad[0]
        with pytest.raises(pkg.LfGpuError):
            G.gpu().ligero_inner_product_rows(field, w, r, ld, nrows, d_dense.data_ptr(), ndense, scale, bad, val, d_rows.data_ptr())
        bad = idx.copy()
        bad[-1] = nrows * w
        with pytest.raises(pkg.LfGpuError):
            G.gpu().ligero_inner_product_rows(field, w, r, ld, nrows, d_dense.data_ptr(), ndense, scale, bad, val, d_rows.data_ptr())
    with pytest.raises(pkg.LfGpuError):  # r + w must fit the row
        G.gpu().ligero_inner_product_rows(field, w, ld, ld, nrows, d_dense.data_ptr(), ndense, scale, idx, val, d_rows.data_ptr())


def test_two_contexts_in_one_process(G):
    """launch configuration (dynamic-LDS attributes, tile geometry, plan caches) is per context, not per process: a second
    context created after the first has run everything produces the same results (round 1 kept this state in process-global
    statics, which is wrong for one process driving two devices)"""
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(99)
    g2 = G.pkg.LfGpu(0)
    g2.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        n = 1 << 14
        a = np.zeros((3, n, 2), dtype=np.uint64)
        o.lfo_fp_bogorng_fill(5, 3 * n, P(a))
        b = ol.rand_elts(rng, 40 << 9).reshape(40, 1 << 9, 2)
        t = ol.rand_elts(rng, 3 * 4096).reshape(3, 4096, 2)
        res = []
        for gpu in (G.gpu(), g2):
            da, db, dt = G.to_dev(a), G.to_dev(b), G.to_dev(t)
            gpu.fp128_fft(da.data_ptr(), 3, n)
            gpu.gf2128_lch14_fft(db.data_ptr(), 40, 9)
            gpu.gf2128_rs_encode_rows(dt.data_ptr(), 3, 455, 4096)
            res.append([G.from_dev(x, np.uint64, y.shape).copy() for x, y in ((da, a), (db, b), (dt, t))])
        for x, y in zip(*res):
            assert (x == y).all()
        want = a[0].copy()
        o.lfo_fp_fftb(P(want), n, o.lfo_fp_omega32(), 1 << 32)
        assert (res[1][0][0] == want).all()
    finally:
        g2.close()
