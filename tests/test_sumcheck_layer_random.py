"""lfgpu_sumcheck_layer (the library's C++ round loop + the fused / resident / shrinking-grid kernels) against a
step-by-step replay on the oracle, for BOTH fields and for sizes that exercise every driver:
  small  (a few hundred entries)   single workgroup from the first round
  mid    (thousands)               several workgroups of the cooperative grid, shrinking to one
  large  (> 64K entries)           multi-kernel path for the first rounds, then the grid
The reference semantics are ProverLayers::layer with logc = 0 (lib/sumcheck/prover_layers.h:185-271)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import FP, GF, P, arr, elt


def _morton(a, b):
    a = a.astype(np.uint64)
    b = b.astype(np.uint64)
    m = np.zeros_like(a)
    for i in range(24):
        m |= ((a >> np.uint64(i)) & np.uint64(1)) << np.uint64(2 * i)
        m |= ((b >> np.uint64(i)) & np.uint64(1)) << np.uint64(2 * i + 1)
    return m


def make_layer_vec(rng, field, logv, logw, nterms, nk=9):
    """canonical-order synthetic layer (EQuad::canonicalize, lib/sumcheck/equad.h:79-106), vectorised"""
    nv, nw = 1 << logv, 1 << logw
    g = rng.integers(0, nv, size=2 * nterms, dtype=np.uint32)
    a = rng.integers(0, nw, size=2 * nterms, dtype=np.uint32)
    b = rng.integers(0, nw, size=2 * nterms, dtype=np.uint32)
    # cluster hand pairs so that many gates share one (exercises bind_g's merge and the scatter's run folding)
    rep = rng.random(2 * nterms) < 0.5
    src = rng.integers(0, 2 * nterms, size=2 * nterms)
    a = np.where(rep, a[src], a)
    b = np.where(rep, b[src], b)
    h0, h1 = np.minimum(a, b), np.maximum(a, b)
    key = np.stack([_morton(h0, h1), g.astype(np.uint64)], axis=1)
    _, idx = np.unique(key, axis=0, return_index=True)
    idx = idx[rng.permutation(len(idx))[:nterms]]
    order = np.lexsort((g[idx], _morton(h0[idx], h1[idx])))
    idx = idx[order]
    kvec = ol.rand_elts(rng, nk, field)
    kvec[0] = 0
    return dict(g=np.ascontiguousarray(g[idx]), h0=np.ascontiguousarray(h0[idx]), h1=np.ascontiguousarray(h1[idx]),
                vi=rng.integers(1, nk, size=len(idx), dtype=np.uint32), kvec=kvec, W=ol.rand_elts(rng, nw, field),
                nv=nv, nw=nw, logv=logv, logw=logw, n=len(idx))


class HostField:
    def __init__(self, field):
        self.f, self.o = field, ol.oracle()

    def add(self, a, b):
        return (a[0] ^ b[0], a[1] ^ b[1]) if self.f == GF else tuple(int(x) for x in arr(self.o.lfo_fp_add(elt(a), elt(b))))

    def sub(self, a, b):
        return self.add(a, b) if self.f == GF else tuple(int(x) for x in arr(self.o.lfo_fp_sub(elt(a), elt(b))))

    def mul(self, a, b):
        fn = self.o.lfo_gf_mul if self.f == GF else self.o.lfo_fp_mul
        return tuple(int(x) for x in arr(fn(elt(a), elt(b))))


def oracle_layer(field, L, logv, G0, G1, alpha, beta, wc_in, chal):
    """-> (evals per round-hand [(e0,e1,e2)], wc_out, bound_quad)"""
    o, F = ol.oracle(), HostField(field)
    ctx = ol.gf_ctx(4)
    n = L["n"]
    hc = np.zeros((n, 2), dtype=np.uint32)
    vc = np.zeros((n, 2), dtype=np.uint64)
    nh = o.lfo_quad_bind_g(field, n, P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), P(L["kvec"]), logv, P(G0), P(G1),
                           elt(alpha), elt(beta), P(hc), P(vc))
    WH = [L["W"].copy(), L["W"].copy()]
    nW = [L["nw"], L["nw"]]
    s = F.add(wc_in[0], F.mul(alpha, wc_in[1]))
    one = (1, 0) if field == GF else tuple(int(x) for x in arr(o.lfo_fp_of_scalar(1)))
    evs, k = [], 0
    for rnd in range(L["logw"]):
        for hand in (0, 1):
            qw = np.zeros((nW[hand], 2), dtype=np.uint64)
            o.lfo_qw_scatter(field, nh, P(hc), P(vc), hand, P(WH[1 - hand]), nW[hand], P(qw))
            ev = (ol.Elt * 3)()
            o.lfo_sumcheck_evaluations(field, C.byref(ctx), nW[hand], elt(one), P(qw), P(WH[hand]), elt(s), ev)
            evs.append(tuple((e.l[0], e.l[1]) for e in ev))
            a0, a2 = ol.Elt(), ol.Elt()
            o.lfo_sumcheck_partials(field, nW[hand], P(qw), P(WH[hand]), C.byref(a0), C.byref(a2))
            c0, c2 = (a0.l[0], a0.l[1]), (a2.l[0], a2.l[1])
            c1 = F.sub(F.sub(F.sub(s, c0), c0), c2)
            r = chal[k]
            k += 1
            s = F.add(c0, F.mul(r, F.add(c1, F.mul(r, c2))))  # the round polynomial at the challenge
            out = np.zeros(((nW[hand] + 1) // 2, 2), dtype=np.uint64)
            o.lfo_dense_bind(field, nW[hand], elt(r), P(WH[hand]), P(out))
            WH[hand], nW[hand] = out, (nW[hand] + 1) // 2
            nh = o.lfo_hquad_bind_h(field, nh, P(hc), P(vc), elt(r), hand)
    wc = [tuple(int(x) for x in WH[0][0]), tuple(int(x) for x in WH[1][0])]
    return evs, wc, tuple(int(x) for x in vc[0])


CASES = [("small", 5, 6, 300), ("mid", 9, 12, 7000), ("large", 12, 17, 150000)]


@pytest.mark.gpu
@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sumcheck_layer_matches_oracle_replay(field, case):
    import gpu_util as G
    _, logv, logw, nterms = case
    rng = np.random.default_rng(1000 * field + logw)
    L = make_layer_vec(rng, field, logv, logw, nterms)
    G0, G1 = ol.rand_elts(rng, max(1, logv), field), ol.rand_elts(rng, max(1, logv), field)
    alpha, beta = (tuple(int(x) for x in ol.rand_elts(rng, 1, field)[0]) for _ in range(2))
    wc_in = [tuple(int(x) for x in ol.rand_elts(rng, 1, field)[0]) for _ in range(2)]
    chal = [tuple(int(x) for x in e) for e in ol.rand_elts(rng, 2 * logw, field)]
    q = G.pkg.Quad(G.gpu(), field, L["g"], L["h0"], L["h1"], L["vi"], L["kvec"], L["nv"])
    dW = G.to_dev(L["W"])
    got_ev, order = [], []

    def round_cb(hand, rnd, ev):
        got_ev.append(tuple((int(e[0]), int(e[1])) for e in ev))
        order.append((rnd, hand))
        return chal[len(got_ev) - 1]

    wc, ch, bq = q.sumcheck_layer(logv, G0, G1, alpha, beta, logw, L["nw"], dW.data_ptr(), wc_in, round_cb)
    q.close()
    assert order == [(r, h) for r in range(logw) for h in (0, 1)]
    want_ev, want_wc, want_bq = oracle_layer(field, L, logv, G0, G1, alpha, beta, wc_in, chal)
    for i, (a, b) in enumerate(zip(got_ev, want_ev)):
        assert a == b, "round-hand %d" % i
    assert [tuple(int(x) for x in w) for w in wc] == want_wc
    assert tuple(int(x) for x in bq) == want_bq
    assert [[tuple(int(x) for x in c) for c in ch[h]] for h in (0, 1)] == [[chal[2 * r + h] for r in range(logw)] for h in (0, 1)]


def _bind_gh_all_oracle(field, L, logv, G0, G1, alpha, beta, H0, H1):
    r = ol.oracle().lfo_quad_bind_gh_all(field, L["n"], P(L["g"]), P(L["h0"]), P(L["h1"]), P(L["vi"]), P(L["kvec"]), logv, L["nv"],
                                         P(G0), P(G1), elt(alpha), elt(beta), L["logw"], L["nw"], P(H0), P(H1))
    return (r.l[0], r.l[1])


@pytest.mark.parametrize("field", [GF, FP])
def test_oracle_bind_gh_all_equals_bind_g_then_bind_h(field):
    """Quad::bind_gh_all (lib/sumcheck/quad.h:188-210) is by definition the scalar left after bind_g and binding every
    hand variable (what the prover reports as ProofAux::bound_quad): pins the oracle's restatement of the verifier's
    shortcut against the already pinned bind_g / bind_h path, no reference needed."""
    rng = np.random.default_rng(31 + field)
    logv, logw = 6, 7
    L = make_layer_vec(rng, field, logv, logw, 900)
    G0, G1 = ol.rand_elts(rng, logv, field), ol.rand_elts(rng, logv, field)
    alpha, beta = (tuple(int(x) for x in ol.rand_elts(rng, 1, field)[0]) for _ in range(2))
    chal = [tuple(int(x) for x in e) for e in ol.rand_elts(rng, 2 * logw, field)]
    _, _, bq = oracle_layer(field, L, logv, G0, G1, alpha, beta, [(0, 0), (0, 0)], chal)
    H0 = np.array([chal[2 * r] for r in range(logw)], dtype=np.uint64)
    H1 = np.array([chal[2 * r + 1] for r in range(logw)], dtype=np.uint64)
    assert _bind_gh_all_oracle(field, L, logv, G0, G1, alpha, beta, H0, H1) == bq


@pytest.mark.gpu
@pytest.mark.parametrize("field", [GF, FP])
@pytest.mark.parametrize("case", CASES[:2] + [("big", 12, 16, 400000)], ids=["small", "mid", "big"])
def test_quad_bind_gh_all_matches_oracle(field, case):
    import gpu_util as G
    _, logv, logw, nterms = case
    rng = np.random.default_rng(7000 + field + logw)
    L = make_layer_vec(rng, field, logv, logw, nterms)
    G0, G1 = ol.rand_elts(rng, logv, field), ol.rand_elts(rng, logv, field)
    H0, H1 = ol.rand_elts(rng, logw, field), ol.rand_elts(rng, logw, field)
    alpha, beta = (tuple(int(x) for x in ol.rand_elts(rng, 1, field)[0]) for _ in range(2))
    q = G.pkg.Quad(G.gpu(), field, L["g"], L["h0"], L["h1"], L["vi"], L["kvec"], L["nv"])
    got = q.bind_gh_all(logv, G0, G1, alpha, beta, logw, L["nw"], H0, H1)
    q.close()
    assert got == _bind_gh_all_oracle(field, L, logv, G0, G1, alpha, beta, H0, H1)
