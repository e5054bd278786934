"""Test-harness mirror of ZkProver::commit / prove (lib/zk/zk_prover.h:72-149), ZkCommon::verifier_constraints
(lib/zk/zk_common.h:49-136,406-439) and LigeroProver::prove (lib/ligero/ligero_prover.h:84-146) for GF2_128:
every data-parallel step runs on the GPU through the C ABI (Ligero commit: K3/K5/K6; sumcheck: K10/K11/K7-K9 via
lfgpu_sumcheck_layer; Ligero prove: K12 + K3), the Fiat-Shamir transcript, the symbolic constraint bookkeeping
and the RandomEngine stay on the host as they do in the reference.  Host scalar field arithmetic uses the
oracle: this module is test infrastructure that pins BASELINE's headline path (BM_ShaZK_fp2_128) end to end."""
import ctypes as C
import lzma
import os

import numpy as np
import torch

import oracle_lib as ol
import sumcheck_driver as sd
from fs_transcript import Transcript
from oracle_lib import GF, P, arr, elt
from sumcheck_driver import KMAX, _b16, _e


def layer_size(logw):  # PadLayout::layer_size (zk_common.h:210-222)
    return 4 * logw + 3


class ZkProverGpu:
    def __init__(self, pkg, gpu, circ, rate=7, nreq=132):
        self.pkg, self.gpu, self.c = pkg, gpu, circ
        self.F = sd.HostGF()
        self.sc = sd.GpuSumcheckLayerApi(pkg, gpu, circ)
        self.npub = circ["npub_in"]
        self.n_witness = circ["ninputs"] - self.npub
        self.pad_size = sum(layer_size(l["logw"]) for l in circ["layers"])
        self.param = pkg.ligero_param(GF, self.n_witness + self.pad_size, circ["nl"], rate, nreq, 0)
        self.lp = None

    # ---- ZkProver::commit
    def commit(self, W, rng, ts):
        c, F = self.c, self.F
        wit = []  # pad elements only; the private inputs are concatenated as an array below
        self.pad = []  # per layer: dict(hp[(hand, round)] = (t0, t2), wc = (wc0, wc1))
        lqc = []
        pi = self.n_witness
        for layer in c["layers"]:  # fill_pad (zk_prover.h:152-188), logc = 0
            hp = {}
            for j in range(layer["logw"]):
                for h in (0, 1):
                    t0 = _e(rng.bytes(16))
                    t2 = _e(rng.bytes(16))
                    hp[(h, j)] = (t0, t2)
                    wit += [t0, t2]
            wc = (_e(rng.bytes(16)), _e(rng.bytes(16)))
            wit += [wc[0], wc[1], F.mul(wc[0], wc[1])]
            self.pad.append(dict(hp=hp, wc=wc))
            cp = pi + 4 * layer["logw"]  # claim_pad(0)
            lqc.append((cp, cp + 1, cp + 2))  # setup_lqc (zk_common.h:149-160)
            pi += layer_size(layer["logw"])
        self.lqc = lqc
        Wv = np.concatenate([np.ascontiguousarray(W[self.npub:]), np.array(wit, dtype=np.uint64).reshape(-1, 2)])
        assert len(Wv) == self.param.nw
        sfb = c["subfield_boundary"] - self.npub if c["subfield_boundary"] >= self.npub else 0
        self.lp = self.pkg.LigeroProver(self.gpu, GF, self.param)
        root = self.lp.commit(Wv, sfb, lqc, rng.bytes)
        ts.write_bytes(root)  # LigeroTranscript::write_commitment
        return root

    # ---- ZkProver::prove
    def prove(self, W, ts):
        c, F = self.c, self.F
        # initialize_sumcheck_fiat_shamir (zk_common.h:163-180)
        ts.write_bytes(c["id"])
        for i in range(self.npub):
            ts.write_elt(W[i].tobytes())
        ts.write_elt(b"\x00" * 16)
        ts.write_bytes(b"\x00" * sum(len(l["g"]) for l in c["layers"]))
        tst = ts.clone()
        ins, V = self.sc.eval_circuit(W)
        if ins is None or not (V == 0).all():
            return None
        proof, aux = self._padded_sumcheck(ins, tst)
        a_small, dense, b, ci = self._verifier_constraints(W, proof, aux, ts)
        com = self._ligero_prove(ts, ci, a_small, dense)
        return dict(sumcheck=proof, **com)

    def _padded_sumcheck(self, ins, tst):
        c, F = self.c, self.F
        for _ in range(KMAX):
            tst.elt_gf2128()
        g0 = [_e(tst.elt_gf2128()) for _ in range(KMAX)]
        G = [list(g0), list(g0)]
        logv = c["logv"]
        WC = [(0, 0), (0, 0)]
        proof, aux = [], []
        for ly, layer in enumerate(c["layers"]):
            alpha, beta = _e(tst.elt_gf2128()), _e(tst.elt_gf2128())
            logw = layer["logw"]
            pad = self.pad[ly]
            hp = {}

            def round_cb(hand, rnd, ev, pad=pad, hp=hp):  # round_h (prover_layers.h:320-329): poly - pad
                t0 = F.add(ev[0], pad["hp"][(hand, rnd)][0])
                t2 = F.add(ev[2], pad["hp"][(hand, rnd)][1])
                hp[(hand, rnd)] = (t0, t2)
                tst.write_elt(_b16(t0))
                tst.write_elt(_b16(t2))
                return _e(tst.elt_gf2128())

            G0 = np.array(G[0][:max(1, logv)], dtype=np.uint64)
            G1 = np.array(G[1][:max(1, logv)], dtype=np.uint64)
            WC, ch, bq = self.sc.quads[ly].sumcheck_layer(logv, G0, G1, alpha, beta, logw, layer["nw"], ins[ly].data_ptr(), WC, round_cb)
            wcp = (F.add(WC[0], pad["wc"][0]), F.add(WC[1], pad["wc"][1]))  # end_layer (:331-344)
            tst.write_array([_b16(wcp[0]), _b16(wcp[1])])
            proof.append(dict(hp=hp, wc=wcp))
            aux.append(bq)
            G = [ch[0] + [(0, 0)] * (KMAX - logw), ch[1] + [(0, 0)] * (KMAX - logw)]
            logv = logw
        return proof, aux

    def _lagrange_coef(self, r):  # WPoly::dot_interpolation::coef: p(r) = sum_i lag[i] p(P_i)
        F = self.F
        lag = []
        for i in range(3):
            num, den = F.one, F.one
            for j in range(3):
                if j != i:
                    num = F.mul(num, F.add(r, F.pts[j]))
                    den = F.mul(den, F.add(F.pts[i], F.pts[j]))
            lag.append(F.mul(num, F.inv(den)))
        return lag

    def _verifier_constraints(self, W, proof, aux, tsv):
        """ZkCommon::verifier_constraints with aux (zk_common.h:49-136): returns (sparse terms [(c, w, k)],
        dense block (c, vector over the private inputs), b, number of constraints)"""
        c, F = self.c, self.F
        for _ in range(KMAX):
            tsv.elt_gf2128()
        g0 = [_e(tsv.elt_gf2128()) for _ in range(KMAX)]
        claims = [(0, 0), (0, 0)]
        gh = [list(g0), list(g0)]
        logv = c["logv"]
        a, b = [], []
        ci, pi = 0, self.n_witness
        for ly, layer in enumerate(c["layers"]):
            alpha, _beta = _e(tsv.elt_gf2128()), _e(tsv.elt_gf2128())
            logw = layer["logw"]
            n = 3 + layer_size(logw)  # ovp_layer_size
            known, sym = (0, 0), [(0, 0)] * n

            def axpy(var, kv, k, sign_known=True):
                nonlocal known, sym
                known = F.add(known, F.mul(k, kv))
                sym[var] = F.add(sym[var], k)

            axpy(0, claims[0], F.one)  # ConstraintBuilder::first
            axpy(1, claims[1], alpha)
            hb = [[], []]
            for rnd in range(logw):
                for hand in (0, 1):
                    r = 2 * rnd + hand
                    t0, t2 = proof[ly]["hp"][(hand, rnd)]
                    tsv.write_elt(_b16(t0))
                    tsv.write_elt(_b16(t2))
                    chal = _e(tsv.elt_gf2128())
                    hb[hand].append(chal)
                    lag = self._lagrange_coef(chal)
                    axpy(3 + 2 * r, t0, F.one)  # axmy == axpy in characteristic 2: p(1) = claim - p(0)
                    known = F.mul(known, lag[1])  # scale
                    sym = [F.mul(s_, lag[1]) if s_ != (0, 0) else s_ for s_ in sym]
                    axpy(3 + 2 * r, t0, lag[0])
                    axpy(3 + 2 * r + 1, t2, lag[2])
            eqq = aux[ly]  # Eq::eval(logc = 0) = 1
            wc = proof[ly]["wc"]
            rhs = F.add(F.mul(eqq, F.mul(wc[0], wc[1])), known)  # finalize
            cp = 3 + 4 * logw
            sym[cp] = F.add(sym[cp], F.mul(eqq, wc[1]))
            sym[cp + 1] = F.add(sym[cp + 1], F.mul(eqq, wc[0]))
            sym[cp + 2] = F.add(sym[cp + 2], eqq)
            b.append(rhs)
            for i in range(3 if ly == 0 else 0, n):
                a.append((ci, pi + i - 3, sym[i]))
            ci += 1
            tsv.write_array([_b16(wc[0]), _b16(wc[1])])
            claims = [wc[0], wc[1]]
            gh = [hb[0], hb[1]]
            logv = logw
            pi += layer_size(logw)
        alpha = _e(tsv.elt_gf2128())
        wc = proof[-1]["wc"]
        got = F.add(wc[0], F.mul(alpha, wc[1]))
        # input_constraint (zk_common.h:406-439): b_i = EQ(g0, i) + alpha EQ(g1, i)
        nin = c["ninputs"]
        G0 = np.array(gh[0][:max(1, logv)], dtype=np.uint64)
        G1 = np.array(gh[1][:max(1, logv)], dtype=np.uint64)
        bi = np.zeros((nin, 2), dtype=np.uint64)
        ol.oracle().lfo_raw_eq2(GF, logv, nin, P(G0), P(G1), elt(alpha), P(bi))
        pub_binding = (0, 0)
        for i in range(self.npub):
            pub_binding = F.add(pub_binding, F.mul((int(bi[i, 0]), int(bi[i, 1])), (int(W[i, 0]), int(W[i, 1]))))
        dense = (ci, np.ascontiguousarray(bi[self.npub:]))
        a.append((ci, pi - 3 + 0, F.one))
        a.append((ci, pi - 3 + 1, alpha))
        b.append(F.add(got, pub_binding))
        return a, dense, b, ci + 1

    def _ligero_prove(self, ts, nl, a_small, dense):
        """LigeroProver::prove (ligero_prover.h:84-146)"""
        F, p, o = self.F, self.param, ol.oracle()
        ts.write_bytes(bytes([0xDE, 0xAD, 0xBE, 0xEF]) + b"\x00" * 28)  # hash_of_A (zk_prover.h:143)
        u_ldt = np.array([_e(ts.elt_gf2128()) for _ in range(p.nwqrow)], dtype=np.uint64)
        y_ldt = self.lp.low_degree_proof(u_ldt)
        alphal = [_e(ts.elt_gf2128()) for _ in range(nl)]
        alphaq = [[_e(ts.elt_gf2128()) for _ in range(3)] for _ in range(p.nq)]
        # inner_product_vector (ligero_param.h:382-421)
        A = np.zeros((p.nwqrow * p.w, 2), dtype=np.uint64)
        dc, dvec = dense
        o.lfo_axpy(GF, len(dvec), P(A), elt(alphal[dc]), P(dvec))  # A[w] += k * alphal[c] over the private inputs
        for (cc, w, k) in a_small:
            v = F.mul(k, alphal[cc])
            A[w] = (int(A[w, 0]) ^ v[0], int(A[w, 1]) ^ v[1])
        base = p.nwrow * p.w
        Ax, Ay, Az = base, base + p.nqtriples * p.w, base + 2 * p.nqtriples * p.w
        for i in range(p.nqtriples):
            for j in range(p.w):
                iw = j + i * p.w
                if iw >= p.nq:
                    break
                x, y, z = self.lqc[iw]
                for off, tgt, aq in ((Ax, x, alphaq[iw][0]), (Ay, y, alphaq[iw][1]), (Az, z, alphaq[iw][2])):
                    A[off + iw] = (int(A[off + iw, 0]) ^ aq[0], int(A[off + iw, 1]) ^ aq[1])
                    A[tgt] = (int(A[tgt, 0]) ^ aq[0], int(A[tgt, 1]) ^ aq[1])
        y_dot = self.lp.dot_proof(A)
        u_quad = np.array([_e(ts.elt_gf2128()) for _ in range(p.nqtriples)], dtype=np.uint64).reshape(-1, 2)
        y_q0, y_q2 = self.lp.quadratic_proof(u_quad)
        for y in (y_ldt, y_dot, y_q0, y_q2):
            ts.write_array([y[i].tobytes() for i in range(len(y))])
        idx = ts.choose(p.block_ext, p.nreq)
        req, nonces, path = self.lp.open(idx)
        return dict(y_ldt=y_ldt, y_dot=y_dot, y_quad_0=y_q0, y_quad_2=y_q2, req=req, nonces=nonces, path=path, idx=idx)

    def close(self):
        self.sc.close()
        if self.lp:
            self.lp.close()


def serialize(circ, root, pr):
    """same component order as oracle/ref_flatsha.cc's .zkproof dump"""
    out = bytearray(root)
    for ly, layer in enumerate(circ["layers"]):
        for r in range(layer["logw"]):
            for h in (0, 1):
                t0, t2 = pr["sumcheck"][ly]["hp"][(h, r)]
                out += _b16(t0) + _b16(t2)
        out += _b16(pr["sumcheck"][ly]["wc"][0]) + _b16(pr["sumcheck"][ly]["wc"][1])
    for k in ("y_ldt", "y_dot", "y_quad_0", "y_quad_2", "req"):
        out += np.ascontiguousarray(pr[k]).tobytes()
    out += np.ascontiguousarray(pr["nonces"]).tobytes()
    out += len(pr["path"]).to_bytes(8, "little")
    for d in pr["path"]:
        out += d
    return bytes(out)


def load_zk_fixture(golden_dir, nb):
    return lzma.decompress(open(os.path.join(golden_dir, "flatsha_nb%d.zkproof.xz" % nb), "rb").read())
