"""Shared pieces of the multi-rank tests: an engine backed by the ORACLE (CPU, test infrastructure) with the same
interface as longfellow-zk_amd/parallel.py's GpuEngine, the rank worker, and the statements."""
import ctypes as C
import os

import numpy as np
import torch

import oracle_lib as ol
from oracle_lib import FP, GF, P, elt, arr


class OracleSlabProver:
    """the prove entry points of a row slab, restated with the oracle's Blas (partial sums, like lfgpu_ligero_prover_from_slab)"""

    def __init__(self, field, p, row_lo, row_hi, slab, layers, nonces):
        self.f, self.p, self.lo, self.hi = field, p, row_lo, row_hi
        self.T = slab.numpy().view(np.uint64).reshape(row_hi - row_lo, p.block_enc, 2)
        self.layers, self.nonces = layers, nonces

    def _row(self, i, n):
        return np.ascontiguousarray(self.T[i - self.lo, :n])

    def _has(self, i):
        return self.lo <= i < self.hi

    def _wq(self):
        p = self.p
        lo, hi = min(max(self.lo, p.iw), p.iw + p.nwqrow), min(max(self.hi, p.iw), p.iw + p.nwqrow)
        return range(lo - p.iw, max(hi, lo) - p.iw)

    def low_degree_proof(self, u):
        o, p = ol.oracle(), self.p
        y = self._row(p.ildt, p.block).copy() if self._has(p.ildt) else np.zeros((p.block, 2), dtype=np.uint64)
        for i in self._wq():
            o.lfo_axpy(self.f, p.block, P(y), elt(u[i]), P(self._row(p.iw + i, p.block)))
        return y

    def dot_proof(self, A):
        o, p = ol.oracle(), self.p
        y = self._row(p.idot, p.dblock).copy() if self._has(p.idot) else np.zeros((p.dblock, 2), dtype=np.uint64)
        for i in self._wq():
            ext = np.zeros((p.dblock, 2), dtype=np.uint64)
            ext[p.r:p.r + p.w] = A[i * p.w:(i + 1) * p.w]
            _rs(self.f, p.block, p.dblock, ext)
            o.lfo_vaxpy(self.f, p.dblock, P(y), P(ext), P(self._row(p.iw + i, p.dblock)))
        return y

    def quadratic_proof(self, uq):
        o, p, f = ol.oracle(), self.p, self.f
        y = self._row(p.iquad, p.dblock).copy() if self._has(p.iquad) else np.zeros((p.dblock, 2), dtype=np.uint64)
        if p.nqtriples and self._has(p.iq):
            assert self._has(p.iq + 3 * p.nqtriples - 1)
            for i in range(p.nqtriples):
                X, Y, Z = (self._row(p.iq + k * p.nqtriples + i, p.dblock) for k in range(3))
                for j in range(p.dblock):
                    t = o.lfo_sub(f, elt(Z[j]), o.lfo_mul(f, elt(X[j]), elt(Y[j])))
                    y[j] = arr(o.lfo_add(f, elt(y[j]), o.lfo_mul(f, elt(uq[i]), t)))
        return y[:p.r].copy(), y[p.block:p.dblock].copy()

    def open(self, idx, rows=None):
        p = self.p
        req = np.ascontiguousarray(self.T[:, [p.dblock + i for i in idx], :])
        nz = np.frombuffer(self.nonces, dtype=np.uint8).reshape(p.block_ext, 32)[list(idx)].copy()
        return req, nz, merkle_open_host(p.block_ext, self.layers.numpy(), idx)

    def close(self):
        pass


def merkle_open_host(n, layers, pos):
    """MerkleTree::generate_compressed_proof (lib/merkle/merkle_tree.h:122-143) on a host heap"""
    tree = [False] * (2 * n)
    for q in pos:
        tree[q + n] = True
    for i in range(n - 1, 0, -1):
        tree[i] = tree[2 * i] or tree[2 * i + 1]
    out = []
    for i in range(n - 1, 0, -1):
        if tree[i]:
            child = 2 * i
            if tree[child]:
                child = 2 * i + 1
            if not tree[child]:
                out.append(bytes(layers[child]))
    return out


def _rs(field, n, m, row):
    o = ol.oracle()
    if field == GF:
        o.lfo_lch14_rs_interpolate(C.byref(ol.gf_ctx(4)), n, m, P(row))
    else:
        o.lfo_fp_rs_interpolate(n, m, P(row))


class OracleEngine:
    """CPU stand-in for GpuEngine in the gloo tests: the same methods, computed by the oracle"""

    def __init__(self, field):
        self.field = field

    def encode_rows(self, p, row_lo, row_hi, h_rows):
        nr = row_hi - row_lo
        T = np.zeros((nr, p.block_enc, 2), dtype=np.uint64)
        for i in range(nr):
            T[i, :p.dblock] = h_rows[i]
            _rs(self.field, p.dblock if (row_lo + i) in (p.idot, p.iquad) else p.block, p.block_enc, T[i])
        return torch.from_numpy(T.view(np.uint8).reshape(nr, p.block_enc * 16))

    def column_leaves(self, nrow, cols, nonces):
        o = ol.oracle()
        a = np.ascontiguousarray(cols.numpy()).view(np.uint64).reshape(nrow, -1, 2)
        n = a.shape[1]
        out = np.zeros((n, 32), dtype=np.uint8)
        if n:
            o.lfo_column_leaves(self.field, nrow, n, 0, n, P(a), P(np.ascontiguousarray(nonces.numpy())), P(out))
        return torch.from_numpy(out)

    def build_tree(self, leaves):
        o = ol.oracle()
        lv = np.ascontiguousarray(leaves.numpy())
        lay = np.zeros((2 * len(lv), 32), dtype=np.uint8)
        o.lfo_merkle_build_tree(len(lv), P(lv), P(lay))
        return lay[1].tobytes(), torch.from_numpy(lay)

    def slab_prover(self, p, row_lo, row_hi, slab, layers, nonces):
        return OracleSlabProver(self.field, p, row_lo, row_hi, slab, layers, nonces)


def statement(pkg, field):
    """(param, W, subfield_boundary, lqc, seed, expected root or None).  GF2_128: the reference's own C++ Ligero vector
    (tests/golden/ligero_test_vector.bin, root pinned); Fp128: a synthetic statement with valid quadratic constraints."""
    import ligero_fixture as lf
    if field == GF:
        v = lf.load()
        p = pkg.ligero_param(GF, v["nw"], v["nq"], 4, v["nreq"], 4096)
        return p, v["W"], v["subfield_boundary"], v["lqc"], 100, v["root"]
    o = ol.oracle()
    rng = np.random.default_rng(77)
    nw, nq = 700, 40
    p = pkg.ligero_param(FP, nw, nq, 4, 12, 512)
    W = ol.rand_elts(rng, nw, FP)
    lqc = []
    zs = rng.choice(np.arange(nw // 2, nw), size=nq, replace=False)
    for i in range(nq):
        x, y, z = int(rng.integers(0, nw // 2)), int(rng.integers(0, nw // 2)), int(zs[i])
        W[z] = arr(o.lfo_mul(FP, elt(W[x]), elt(W[y])))
        lqc.append((x, y, z))
    return p, W, 0, lqc, 5, None


def run_rank_c(pkg, par, gpu, field, group_world):
    """the PRODUCT path: lfgpu_ligero_commit_sharded + the prove entry points through parallel.ShardedLigeroProver (hooks bound
    to torch.distributed by parallel.TorchComm) on the same statement / vectors as run_rank"""
    import ligero_fixture as lf
    p, W, sfb, lqc, seed, want_root = statement(pkg, field)
    rng = np.random.default_rng(9)
    u = ol.rand_elts(rng, p.nwqrow, field)
    A = ol.rand_elts(rng, p.nwqrow * p.w, field)
    uq = ol.rand_elts(rng, max(1, p.nqtriples), field)
    idx = [int(t) for t in rng.choice(p.block_ext, size=p.nreq, replace=False)]
    pr = par.ShardedLigeroProver(gpu, field, p, 4, group_world)
    pr.comm.selftest()
    out = {"root": pr.commit(W, sfb, lqc, lf.LcgRng(seed).bytes)}
    out["y_ldt"] = pr.low_degree_proof(u)
    out["y_dot"] = pr.dot_proof(A)
    out["y_q0"], out["y_q2"] = pr.quadratic_proof(uq)
    out["req"], out["nonces"], out["path"] = pr.open(idx)
    out["spans"] = pr.spans
    pr.close()
    if want_root is not None:
        assert out["root"] == want_root, "sharded root (C entry point) differs from the reference's C++ fixture root"
    return out


def run_rank(pkg, par, engine, field, group_world, group_solo):
    """every rank: sharded commit + prove over `group_world`, the same over a one-rank group, compare.  Returns a dict
    of the sharded results (root, y vectors, req) for further checks by the caller."""
    import ligero_fixture as lf
    p, W, sfb, lqc, seed, want_root = statement(pkg, field)
    lib = pkg.load_library()
    rng = np.random.default_rng(9)
    u = ol.rand_elts(rng, p.nwqrow, field)
    A = ol.rand_elts(rng, p.nwqrow * p.w, field)
    uq = ol.rand_elts(rng, max(1, p.nqtriples), field)
    idx = [int(t) for t in rng.choice(p.block_ext, size=p.nreq, replace=False)]
    res = []
    for grp in (group_world, group_solo):
        import sharded_reference as sref
        pr = sref.ShardedLigeroProver(engine, lib, field, p, 4, grp)
        out = {"root": pr.commit(W, sfb, lqc, lf.LcgRng(seed).bytes)}
        out["y_ldt"] = pr.low_degree_proof(u)
        out["y_dot"] = pr.dot_proof(A)
        out["y_q0"], out["y_q2"] = pr.quadratic_proof(uq)
        out["req"], out["nonces"], out["path"] = pr.open(idx)
        out["spans"] = pr.spans
        pr.close()
        res.append(out)
    multi, solo = res
    if want_root is not None:
        assert multi["root"] == want_root, "sharded root differs from the reference's C++ fixture root"
    for key in ("root", "path"):
        assert multi[key] == solo[key], key
    for key in ("y_ldt", "y_dot", "y_q0", "y_q2", "req", "nonces"):
        assert (multi[key] == solo[key]).all(), key
    multi.update(p=p, W=W, sfb=sfb, lqc=lqc, seed=seed, u=u, A=A, uq=uq, idx=idx)
    return multi
