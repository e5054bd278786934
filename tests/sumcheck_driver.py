"""Test-harness driver: the reference's layered sumcheck prover (ProverLayers::eval_circuit / prove /
layer, lib/sumcheck/prover_layers.h:52-271; Prover::prove, lib/sumcheck/prover.h:45-54) with every
data-parallel step on the GPU through the C ABI (K11 eval_quad, K10 bind_g, K8 scatter, K7 partial sums,
K9 binds) and the sequential pieces -- Fiat-Shamir transcript, 3-point round polynomial -- on the host,
exactly where an integration leaves them.  Host scalar field arithmetic uses the oracle (this module is
test infrastructure); GF2_128 only.

Also: a vectorised reader for the reference's LFC1 circuit wire format (lib/proto/circuit_reader.h:55-233).
"""
import ctypes as C
import lzma
import time

import numpy as np
import torch

import oracle_lib as ol
from fs_transcript import Transcript
from oracle_lib import GF, arr, elt

KMAX = 40  # Proof::kMaxBindings (lib/sumcheck/circuit.h:84)


def read_lfc1(raw):
    """-> dict(nv, nc, npub_in, subfield_boundary, ninputs, nl, logv, kvec[nk,2], layers=[{logw,nw,g,h0,h1,vi}], id)"""
    b = np.frombuffer(raw, dtype=np.uint8)
    assert b[0] == 1, "LFC1 version"
    pos = 1

    def num():
        nonlocal pos
        v = int(b[pos]) | int(b[pos + 1]) << 8 | int(b[pos + 2]) << 16
        pos += 3
        return v

    fid, nv, nc, npub, sfb, nin, nl, nk = (num() for _ in range(8))
    assert fid == 4, "GF2_128 circuits only"
    kvec = np.frombuffer(raw, dtype=np.uint64, count=2 * nk, offset=pos).reshape(nk, 2).copy()
    pos += 16 * nk
    layers = []
    for _ in range(nl):
        logw, nw, nq = num(), num(), num()
        t = b[pos:pos + 12 * nq].reshape(nq, 4, 3).astype(np.int64)
        pos += 12 * nq
        v = t[:, :, 0] | (t[:, :, 1] << 8) | (t[:, :, 2] << 16)
        d = (v[:, :3] >> 1) * (1 - 2 * (v[:, :3] & 1))  # LSB = sign (circuit_writer.h:103-114)
        idx = np.cumsum(d, axis=0)
        layers.append(dict(logw=logw, nw=nw, g=idx[:, 0].astype(np.uint32), h0=idx[:, 1].astype(np.uint32),
                           h1=idx[:, 2].astype(np.uint32), vi=v[:, 3].astype(np.uint32)))
    cid = bytes(b[pos:pos + 32])
    assert pos + 32 == len(b)
    logv = max(0, (nv - 1).bit_length())
    return dict(nv=nv, nc=nc, npub_in=npub, subfield_boundary=sfb, ninputs=nin, nl=nl, logv=logv, kvec=kvec,
                layers=layers, id=cid)


def load_fixture(golden_dir, nb):
    import json
    import os
    raw = lzma.decompress(open(os.path.join(golden_dir, "flatsha_nb%d.lfc1.xz" % nb), "rb").read())
    W = np.frombuffer(lzma.decompress(open(os.path.join(golden_dir, "flatsha_nb%d.w.xz" % nb), "rb").read()),
                      dtype=np.uint64).reshape(-1, 2).copy()
    proof = open(os.path.join(golden_dir, "flatsha_nb%d.scproof" % nb), "rb").read()
    info = json.load(open(os.path.join(golden_dir, "flatsha_nb%d.json" % nb)))
    return read_lfc1(raw), W, proof, info


# ---------------------------------------------------------------- host scalar field helpers (GF2_128)
class HostGF:
    def __init__(self):
        self.o = ol.oracle()
        c = ol.gf_ctx(4)
        self.one = (1, 0)
        self.pts = [tuple(int(x) for x in arr(self.o.lfo_gf_poly_evaluation_point(C.byref(c), i))) for i in range(3)]

    def mul(self, a, b):
        r = self.o.lfo_gf_mul(elt(a), elt(b))
        return (r.l[0], r.l[1])

    def inv(self, a):
        r = self.o.lfo_gf_inv(elt(a))
        return (r.l[0], r.l[1])

    @staticmethod
    def add(a, b):
        return (a[0] ^ b[0], a[1] ^ b[1])

    def eval_monomial(self, coef, x):  # Poly::eval_monomial (lib/algebra/poly.h:100-108)
        e = coef[-1]
        for c_ in reversed(coef[:-1]):
            e = self.add(self.mul(e, x), c_)
        return e

    def eval_lagrange3(self, ev, x):
        """value at x of the degree-2 polynomial through (P0,ev0),(P1,ev1),(P2,ev2)
        (= Poly<3>::eval_lagrange, lib/algebra/poly.h:72-98; exact arithmetic => same element)"""
        acc = (0, 0)
        for i in range(3):
            num, den = self.one, self.one
            for j in range(3):
                if j != i:
                    num = self.mul(num, self.add(x, self.pts[j]))
                    den = self.mul(den, self.add(self.pts[i], self.pts[j]))
            acc = self.add(acc, self.mul(ev[i], self.mul(num, self.inv(den))))
        return acc


def _b16(e):
    return int(e[0]).to_bytes(8, "little") + int(e[1]).to_bytes(8, "little")


def _e(bs):
    return (int.from_bytes(bs[:8], "little"), int.from_bytes(bs[8:16], "little"))


class SumcheckBase:
    """ProverLayers control flow; subclasses provide the data-parallel steps"""

    def __init__(self, circ):
        self.c = circ
        self.F = HostGF()

    # --- steps (subclass): buffers are opaque handles
    def eval_circuit(self, W):
        raise NotImplementedError

    def prove(self, ins, W_host, seed=b"testing"):
        """Prover::prove + ProverLayers::prove/layer with pad = nullptr.  Returns the transmitted proof bytes in
        the fixture's order: per layer, per round, per hand p(0), p(2); then wc[0], wc[1]."""
        c, F = self.c, self.F
        ts = Transcript(seed)
        ts.write_array([bytes(W_host[i].tobytes()) for i in range(len(W_host))])  # write_input, nc = 1
        for _ in range(KMAX):  # begin_circuit: Q then G (transcript_sumcheck.h:49-52)
            ts.elt_gf2128()
        g0 = [_e(ts.elt_gf2128()) for _ in range(KMAX)]
        G = [list(g0), list(g0)]
        logv = c["logv"]
        WC = [(0, 0), (0, 0)]
        out = bytearray()
        for ly, layer in enumerate(c["layers"]):
            alpha, beta = _e(ts.elt_gf2128()), _e(ts.elt_gf2128())
            logw, n = layer["logw"], layer["nw"]
            G0 = np.array(G[0][:max(1, logv)], dtype=np.uint64)
            G1 = np.array(G[1][:max(1, logv)], dtype=np.uint64)
            self.begin_layer(ly, ins[ly], logv, G0, G1, alpha, beta)
            s = F.add(WC[0], F.mul(alpha, WC[1]))
            eq0 = F.one  # logc = 0: Eqs(0, 1, q) = [1]
            hands = [[], []]
            for rnd in range(logw):
                for hand in (0, 1):
                    a0, a2 = self.round_partials(hand)
                    c0, c2 = F.mul(eq0, a0), F.mul(eq0, a2)
                    c1 = F.add(F.add(F.add(s, c0), c0), c2)
                    ev = [F.eval_monomial([c0, c1, c2], F.pts[k]) for k in range(3)]
                    out += _b16(ev[0]) + _b16(ev[2])
                    ts.write_elt(_b16(ev[0]))
                    ts.write_elt(_b16(ev[2]))
                    r = _e(ts.elt_gf2128())
                    hands[hand].append(r)
                    s = F.eval_lagrange3(ev, r)
                    self.round_bind(hand, r, first=(rnd == 0 and hand == 0))
            WC = self.end_layer()
            out += _b16(WC[0]) + _b16(WC[1])
            ts.write_array([_b16(WC[0]), _b16(WC[1])])
            G = [hands[0] + [(0, 0)] * (KMAX - logw), hands[1] + [(0, 0)] * (KMAX - logw)]
            logv = logw
        return bytes(out)


def prove_padded(sc, ins, tst, pads):
    """ProverLayers::prove as ZkProver::prove runs it (lib/zk/zk_prover.h:117-127): every transmitted value is poly - pad
    (round_h / end_layer of the padded prover, prover_layers.h:320-344), on the transcript `tst` -- the clone taken after
    initialize_sumcheck_fiat_shamir.  pads[ly] = dict(hp={(hand, round): (t0, t2)}, wc=(wc0, wc1)).  Returns the bytes of
    ZkProof::write_sc_proof (lib/zk/zk_proof.h:115-131): per layer, per round, p(0) of both hands then p(2) of both hands; wc."""
    c, F = sc.c, sc.F
    for _ in range(KMAX):
        tst.elt_gf2128()
    g0 = [_e(tst.elt_gf2128()) for _ in range(KMAX)]
    G = [list(g0), list(g0)]
    logv = c["logv"]
    WC = [(0, 0), (0, 0)]
    out = bytearray()
    for ly, layer in enumerate(c["layers"]):
        alpha, beta = _e(tst.elt_gf2128()), _e(tst.elt_gf2128())
        logw = layer["logw"]
        G0 = np.array(G[0][:max(1, logv)], dtype=np.uint64)
        G1 = np.array(G[1][:max(1, logv)], dtype=np.uint64)
        sc.begin_layer(ly, ins[ly], logv, G0, G1, alpha, beta)
        s = F.add(WC[0], F.mul(alpha, WC[1]))
        hands = [[], []]
        sent = {}
        for rnd in range(logw):
            for hand in (0, 1):
                a0, a2 = sc.round_partials(hand)
                c1 = F.add(F.add(F.add(s, a0), a0), a2)
                ev = [F.eval_monomial([a0, c1, a2], F.pts[k]) for k in range(3)]
                t0 = F.add(ev[0], pads[ly]["hp"][(hand, rnd)][0])
                t2 = F.add(ev[2], pads[ly]["hp"][(hand, rnd)][1])
                sent[(hand, rnd)] = (t0, t2)
                tst.write_elt(_b16(t0))
                tst.write_elt(_b16(t2))
                r = _e(tst.elt_gf2128())
                hands[hand].append(r)
                s = F.eval_lagrange3(ev, r)
                sc.round_bind(hand, r, first=(rnd == 0 and hand == 0))
        WC = sc.end_layer()
        wcp = (F.add(WC[0], pads[ly]["wc"][0]), F.add(WC[1], pads[ly]["wc"][1]))
        tst.write_array([_b16(wcp[0]), _b16(wcp[1])])
        for rnd in range(logw):
            out += _b16(sent[(0, rnd)][0]) + _b16(sent[(1, rnd)][0]) + _b16(sent[(0, rnd)][1]) + _b16(sent[(1, rnd)][1])
        out += _b16(wcp[0]) + _b16(wcp[1])
        G = [hands[0] + [(0, 0)] * (KMAX - logw), hands[1] + [(0, 0)] * (KMAX - logw)]
        logv = logw
    return bytes(out)


class OracleSumcheck(SumcheckBase):
    """CPU: every step through the oracle (pins driver + transcript against the reference fixture without a GPU)"""

    def eval_circuit(self, W):
        c, o = self.c, ol.oracle()
        nl = c["nl"]
        ins = [None] * nl
        ins[nl - 1] = W.copy()
        cur = ins[nl - 1]
        for l in range(nl - 1, -1, -1):
            L = c["layers"][l]
            nout = c["layers"][l - 1]["nw"] if l > 0 else c["nv"]
            V = np.zeros((nout, 2), dtype=np.uint64)
            ok = o.lfo_eval_quad(GF, len(L["g"]), ol.P(L["g"]), ol.P(L["h0"]), ol.P(L["h1"]), ol.P(L["vi"]), ol.P(c["kvec"]),
                                 nout, ol.P(cur), ol.P(V))
            if not ok:
                return None, None
            if l > 0:
                ins[l - 1] = V
            cur = V
        return ins, cur

    def begin_layer(self, ly, Win, logv, G0, G1, alpha, beta):
        c, o = self.c, ol.oracle()
        L = c["layers"][ly]
        n = len(L["g"])
        self.hc = np.zeros((n, 2), dtype=np.uint32)
        self.vc = np.zeros((n, 2), dtype=np.uint64)
        self.nh = o.lfo_quad_bind_g(GF, n, ol.P(L["g"]), ol.P(L["h0"]), ol.P(L["h1"]), ol.P(L["vi"]), ol.P(c["kvec"]), logv,
                                    ol.P(G0), ol.P(G1), elt(alpha), elt(beta), ol.P(self.hc), ol.P(self.vc))
        self.WH = [Win, Win]
        self.nW = [L["nw"], L["nw"]]

    def round_partials(self, hand):
        o = ol.oracle()
        qw = np.zeros((self.nW[hand], 2), dtype=np.uint64)
        o.lfo_qw_scatter(GF, self.nh, ol.P(self.hc), ol.P(self.vc), hand, ol.P(self.WH[1 - hand]), self.nW[hand], ol.P(qw))
        a0, a2 = ol.Elt(), ol.Elt()
        o.lfo_sumcheck_partials(GF, self.nW[hand], ol.P(qw), ol.P(self.WH[hand]), C.byref(a0), C.byref(a2))
        return (a0.l[0], a0.l[1]), (a2.l[0], a2.l[1])

    def round_bind(self, hand, r, first):
        o = ol.oracle()
        n = self.nW[hand]
        out = np.zeros(((n + 1) // 2, 2), dtype=np.uint64)
        o.lfo_dense_bind(GF, n, elt(r), ol.P(self.WH[hand]), ol.P(out))
        self.WH[hand] = out
        self.nW[hand] = (n + 1) // 2
        self.nh = o.lfo_hquad_bind_h(GF, self.nh, ol.P(self.hc), ol.P(self.vc), elt(r), hand)

    def end_layer(self):
        return [(int(self.WH[0][0, 0]), int(self.WH[0][0, 1])), (int(self.WH[1][0, 0]), int(self.WH[1][0, 1]))]


class GpuSumcheck(SumcheckBase):
    """one circuit resident on the GPU (quads uploaded once), prove() per witness"""

    def __init__(self, pkg, gpu, circ):
        super().__init__(circ)
        self.pkg, self.gpu = pkg, gpu
        self.quads = []
        nv = circ["nv"]
        for ly in circ["layers"]:
            self.quads.append(pkg.Quad(gpu, GF, ly["g"], ly["h0"], ly["h1"], ly["vi"], circ["kvec"], nv))
            nv = ly["nw"]
        self.maxterms = max(len(ly["g"]) for ly in circ["layers"])
        self.maxw = max(ly["nw"] for ly in circ["layers"])
        dev = "cuda"
        self.hc = [torch.empty(self.maxterms * 8, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.vc = [torch.empty(self.maxterms * 16, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.qw = torch.empty(self.maxw * 16, dtype=torch.uint8, device=dev)
        self.wtmp = torch.empty(self.maxw * 16, dtype=torch.uint8, device=dev)

    def eval_circuit(self, W):
        c = self.c
        nl = c["nl"]
        ins = [None] * nl
        ins[nl - 1] = torch.from_numpy(W.view(np.uint8).reshape(-1).copy()).cuda()
        cur = ins[nl - 1]
        for l in range(nl - 1, -1, -1):
            nout = c["layers"][l - 1]["nw"] if l > 0 else c["nv"]
            V = torch.empty(nout * 16, dtype=torch.uint8, device="cuda")
            ok = self.quads[l].eval(c["layers"][l]["nw"], cur.data_ptr(), V.data_ptr())
            if not ok:
                return None, None
            if l > 0:
                ins[l - 1] = V
            cur = V
        torch.cuda.synchronize()
        return ins, cur.cpu().numpy().view(np.uint64).reshape(-1, 2)

    def begin_layer(self, ly, Win, logv, G0, G1, alpha, beta):
        self.cur = 0
        self.nh = self.quads[ly].bind_g(logv, G0, G1, alpha, beta, self.hc[0].data_ptr(), self.vc[0].data_ptr())
        self.WH = [Win, Win]  # hand 1 is bound in place on the layer input (prover_layers.h:222-226)
        n = self.c["layers"][ly]["nw"]
        self.nW = [n, n]

    def round_partials(self, hand):
        gpu, cur = self.gpu, self.cur
        gpu.qw_scatter(GF, self.nh, self.hc[cur].data_ptr(), self.vc[cur].data_ptr(), hand, self.WH[1 - hand].data_ptr(),
                       self.nW[hand], self.qw.data_ptr())
        return gpu.sumcheck_partials(GF, self.nW[hand], self.qw.data_ptr(), self.WH[hand].data_ptr())

    def round_bind(self, hand, r, first):
        gpu, cur = self.gpu, self.cur
        if first:
            gpu.dense_bind(GF, self.nW[0], r, self.WH[0].data_ptr(), self.wtmp.data_ptr())
            self.WH[0] = self.wtmp
        else:
            gpu.dense_bind(GF, self.nW[hand], r, self.WH[hand].data_ptr(), self.WH[hand].data_ptr())
        self.nW[hand] = (self.nW[hand] + 1) // 2
        self.nh = gpu.hquad_bind_h(GF, self.nh, self.hc[cur].data_ptr(), self.vc[cur].data_ptr(), r, hand,
                                   self.hc[1 - cur].data_ptr(), self.vc[1 - cur].data_ptr())
        self.cur = 1 - cur

    def end_layer(self):
        torch.cuda.synchronize()
        w0 = self.WH[0][:16].cpu().numpy().view(np.uint64)
        w1 = self.WH[1][:16].cpu().numpy().view(np.uint64)
        return [(int(w0[0]), int(w0[1])), (int(w1[0]), int(w1[1]))]

    def close(self):
        for q in self.quads:
            q.close()


class GpuSumcheckLayerApi(GpuSumcheck):
    """same proof through lfgpu_sumcheck_layer: the whole layer loop runs in the library's C++ host code and
    only the transcript round (the caller's round_h + ts.round) comes back through a callback"""

    def prove(self, ins, W_host, seed=b"testing"):
        c = self.c
        ts = Transcript(seed)
        ts.write_array([bytes(W_host[i].tobytes()) for i in range(len(W_host))])
        for _ in range(KMAX):
            ts.elt_gf2128()
        g0 = [_e(ts.elt_gf2128()) for _ in range(KMAX)]
        G = [list(g0), list(g0)]
        logv = c["logv"]
        WC = [(0, 0), (0, 0)]
        out = bytearray()

        def round_cb(hand, rnd, ev):
            nonlocal out
            out += _b16(ev[0]) + _b16(ev[2])
            ts.write_elt(_b16(ev[0]))
            ts.write_elt(_b16(ev[2]))
            return _e(ts.elt_gf2128())

        for ly, layer in enumerate(c["layers"]):
            alpha, beta = _e(ts.elt_gf2128()), _e(ts.elt_gf2128())
            logw = layer["logw"]
            G0 = np.array(G[0][:max(1, logv)], dtype=np.uint64)
            G1 = np.array(G[1][:max(1, logv)], dtype=np.uint64)
            WC, ch, _bq = self.quads[ly].sumcheck_layer(logv, G0, G1, alpha, beta, logw, layer["nw"], ins[ly].data_ptr(), WC, round_cb)
            out += _b16(WC[0]) + _b16(WC[1])
            ts.write_array([_b16(WC[0]), _b16(WC[1])])
            G = [ch[0] + [(0, 0)] * (KMAX - logw), ch[1] + [(0, 0)] * (KMAX - logw)]
            logv = logw
        return bytes(out)
