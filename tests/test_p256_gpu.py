"""GPU parity of the P-256 base-field leg (field id 1, 32-byte elements; BASELINE config 5's signature tableau) through
the C ABI: element-wise ops, lfgpu_fp256_rs_encode_rows, lfgpu_column_commit -- bit-exact against the oracle and against
the reference's outputs in tests/golden/ref_vectors_p256.json."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import P
from test_p256_cpu import GOLD, fill, sig_tableau

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


@pytest.fixture(scope="module")
def g():
    with open(os.path.join(GOLD, "ref_vectors_p256.json")) as f:
        return json.load(f)


def test_p256_binops(G):
    import torch
    o = ol.oracle()
    n = 6000
    a, b = fill(1, n), fill(2, n)
    pm1 = [0xFFFFFFFFFFFFFFFE, 0x00000000FFFFFFFF, 0, 0xFFFFFFFF00000001]
    edge = [[0, 0, 0, 0], [1, 0, 0, 0], pm1, [0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0, 0xFFFFFFFF00000000], [0, 0, 0, 1], [0xFFFFFFFFFFFFFFFF, 0, 0, 0],
            [0, 1 << 32, 0, 0], [0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0x7FFFFFFFFFFFFFFF]]
    k = 0
    for x in edge:
        for y in edge:
            a[k], b[k] = x, y
            k += 1
    da, db = G.to_dev(a), G.to_dev(b)
    dout = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    for op, name in ((0, "add"), (1, "sub"), (2, "mul")):
        G.gpu().field_binop(G.pkg.FIELD_P256, op, n, da.data_ptr(), db.data_ptr(), dout.data_ptr())
        got = G.from_dev(dout, np.uint64, (n, 4))
        fn = getattr(o, "lfo_p256_" + name)
        for i in list(range(k)) + list(range(k, n, 23)):
            assert (got[i] == ol.arr32(fn(ol.e32(a[i]), ol.e32(b[i])))).all(), (name, i)


@pytest.mark.parametrize("n,m,nrow,ld", [(1, 2, 1, 2), (1, 4, 2, 4), (2, 3, 3, 5), (3, 8, 2, 8), (5, 16, 5, 16), (21, 128, 4, 130), (100, 257, 3, 257),
                                         (455, 4096, 3, 4096), (909, 4096, 2, 4096), (600, 1500, 7, 1500), (3000, 16384, 2, 16384)])
def test_fp256_rs_encode_rows(G, n, m, nrow, ld):
    """odd and even row counts (rows travel in pairs through one complex transform), ragged ld, one-tile and multi-stage sizes"""
    o = ol.oracle()
    T = fill(7 * n + m, nrow * ld).reshape(nrow, ld, 4)
    want = T.copy()
    for r in range(nrow):
        row = np.ascontiguousarray(want[r, :m])
        o.lfo_p256_rs_interpolate(n, m, P(row))
        want[r, :m] = row
    d = G.to_dev(T)
    G.gpu().fp256_rs_encode_rows(d.data_ptr(), nrow, n, m, ld=ld)
    assert (G.from_dev(d, np.uint64, T.shape) == want).all()
    with pytest.raises(G.pkg.LfGpuError):
        G.gpu().fp256_rs_encode_rows(d.data_ptr(), nrow, m + 1, m, ld=ld)


@pytest.mark.parametrize("nrow,ld,col0,ncols", [(1, 8, 0, 8), (2, 8, 1, 1), (19, 600, 131, 469), (5, 40, 9, 31), (3, 3, 1, 2), (150, 300, 7, 260)])
def test_column_commit_p256(G, nrow, ld, col0, ncols):
    import torch
    o = ol.oracle()
    rng = np.random.default_rng(nrow * 1000 + ncols)
    T = fill(nrow + ld, nrow * ld).reshape(nrow, ld, 4)
    nonces = rng.integers(0, 256, size=(ncols, 32), dtype=np.uint8)
    want_root = np.zeros(32, dtype=np.uint8)
    lay = np.zeros((2 * ncols, 32), dtype=np.uint8)
    o.lfo_column_commit32(nrow, ld, col0, ncols, P(T), P(nonces), P(want_root), P(lay))
    dT, dN = G.to_dev(T), G.to_dev(nonces)
    dL = torch.zeros(2 * ncols * 32, dtype=torch.uint8, device="cuda")
    root = G.gpu().column_commit(G.pkg.FIELD_P256, nrow, ld, col0, ncols, dT.data_ptr(), dN.data_ptr(), dL.data_ptr())
    assert root == want_root.tobytes()
    assert (G.from_dev(dL, np.uint8, (2 * ncols, 32))[1:] == lay[1:]).all()


def test_config5_signature_tableau_matches_reference(G, g):
    """the mdoc signature circuit's Ligero tableau shape (19 rows x 4096 x 32 B, block 455, dblock 909): RS-extend every
    row and commit the columns on the GPU; encoded bytes and Merkle root equal the REFERENCE's (golden) and the oracle's"""
    import ligero_fixture as lf
    import torch
    c, T = sig_tableau(g)
    nrow, be, block, dblock, ext = c["nrow"], c["block_enc"], c["block"], c["dblock"], c["block_ext"]
    d = G.to_dev(T)
    base = d.data_ptr()
    gpu = G.gpu()
    gpu.fp256_rs_encode_rows(base, 1, block, be)
    gpu.fp256_rs_encode_rows(base + be * 32, 2, dblock, be)
    gpu.fp256_rs_encode_rows(base + 3 * be * 32, nrow - 3, block, be)
    enc = G.from_dev(d, np.uint64, T.shape)
    assert enc[0, -1].tobytes().hex() == c["row0_last"] and enc[2, dblock].tobytes().hex() == c["row2_col_dblock"]
    assert hashlib.sha256(enc.tobytes()).hexdigest() == c["encoded_sha256"]
    nonces = np.frombuffer(lf.LcgRng(c["nonce_lcg_seed"]).bytes(32 * ext), dtype=np.uint8).reshape(-1, 32).copy()
    dN = G.to_dev(nonces)
    dL = torch.zeros(2 * ext * 32, dtype=torch.uint8, device="cuda")
    root = gpu.column_commit(G.pkg.FIELD_P256, nrow, be, dblock, ext, base, dN.data_ptr(), dL.data_ptr())
    assert root.hex() == c["root"]
