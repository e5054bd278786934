#!/usr/bin/env python3
"""gen_tower.py -- build-time generator for the bit-sliced tower representation of GF(2^128)
used by the LCH14 kernels (longfellow-zk_amd/csrc/tower_k{4,5}.h).

Why: CDNA4 has no carry-less multiply.  All LCH14 twiddles lie in the subfield S = GF(2^m)
(m = 16 for GF2_128<4>, 32 for GF2_128<5>; reference lib/gf2k/lch14.h:45-100), and GF(2^128) is a
d = 128/m dimensional vector space over S.  In the basis  { h^j * X^i : i < d, j < m }  (h a root in
S of a sparse irreducible mu of degree m, X the generator of the reference's polynomial basis) a
twiddle multiply is d independent multiplies by the SAME constant in GF(2)[h]/mu, which in a
bit-sliced layout (one 32-bit word = one coordinate bit of 32 batch rows) is ~m/2 conditional
plane-XORs plus 3 XORs per "times h" step -- 45..85 VALU ops per element instead of ~510.

This script derives, with its own GF(2^128) arithmetic (no dependency on oracle/):
  * the subfield generator g (reference lib/gf2k/gf2_128.h:369-391) and a root h in S of mu,
  * the 128x128 GF(2) matrices poly-basis <-> tower-basis,
  * straight-line XOR programs for both directions (greedy common-subexpression elimination),
and writes them as C headers.  Run: python tools/gen_tower.py   (outputs are committed).
"""
import os
import random
import sys

P_LOW = 0x87  # x^128 = x^7 + x^2 + x + 1
MASK128 = (1 << 128) - 1
MU = {4: 0x1002B, 5: 0x10000008D}  # x^16+x^5+x^3+x+1 ; x^32+x^7+x^3+x^2+1


def gmul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        b >>= 1
        a <<= 1
    # reduce 256 -> 128
    hi = r >> 128
    while hi:
        r = (r & MASK128) ^ hi ^ (hi << 1) ^ (hi << 2) ^ (hi << 7)
        hi = r >> 128
    return r


def gpow(a, e):
    r = 1
    while e:
        if e & 1:
            r = gmul(r, a)
        a = gmul(a, a)
        e >>= 1
    return r


def subfield_generator(k):
    r = 2
    for i in range(k, 7):
        s = r
        for _ in range(1 << i):
            s = gmul(s, s)
        r = gmul(r, s)
    return r


def minpoly(x, m):
    pw = [1]
    for _ in range(m):
        pw.append(gmul(pw[-1], x))
    basis = {}
    for idx, v in enumerate(pw):
        combo = 1 << idx
        while v:
            p = v.bit_length() - 1
            if p in basis:
                bv, bc = basis[p]
                v ^= bv
                combo ^= bc
            else:
                basis[p] = (v, combo)
                break
        if v == 0:
            return combo
    raise RuntimeError("no dependency")


# ---- abstract K = GF(2)[z]/mu and polynomials over K (for root finding)
class K:
    def __init__(self, mu, m):
        self.mu, self.m = mu, m

    def mul(self, a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            b >>= 1
            a <<= 1
            if (a >> self.m) & 1:
                a ^= self.mu
        return r

    def inv(self, a):
        return self.pow(a, (1 << self.m) - 2)

    def pow(self, a, e):
        r = 1
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r


def ptrim(p):
    while p and p[-1] == 0:
        p.pop()
    return p


def pmod(F, a, b):
    a = list(a)
    ib = F.inv(b[-1])
    while len(a) >= len(b):
        c = F.mul(a[-1], ib)
        if c:
            sh = len(a) - len(b)
            for i, bc in enumerate(b):
                a[sh + i] ^= F.mul(c, bc)
        a.pop()
    return ptrim(a)


def pmulmod(F, a, b, mod):
    r = [0] * (len(a) + len(b) - 1) if a and b else []
    for i, ac in enumerate(a):
        if ac:
            for j, bc in enumerate(b):
                r[i + j] ^= F.mul(ac, bc)
    return pmod(F, ptrim(r), mod)


def pgcd(F, a, b):
    a, b = list(a), list(b)
    while b:
        a, b = b, pmod(F, a, b)
    ia = F.inv(a[-1])
    return [F.mul(c, ia) for c in a]


def find_root(F, poly_gf2, m, rng):
    """one root in K of a polynomial with GF(2) coefficients that splits into m distinct linear factors over K
    (Berlekamp trace splitting)."""
    f = [(poly_gf2 >> i) & 1 for i in range(m + 1)]
    while len(f) > 2:
        beta = rng.randrange(1, 1 << m)
        # T(y) = sum_{i<m} (beta*y)^(2^i) mod f
        t = pmod(F, [0, beta], f)
        acc = list(t)
        for _ in range(m - 1):
            t = pmulmod(F, t, t, f)
            acc = [x ^ y for x, y in zip(acc + [0] * (len(t) - len(acc)), t + [0] * (len(acc) - len(t)))]
        acc = ptrim(acc)
        if not acc:
            continue
        g1 = pgcd(F, f, acc)
        if 1 < len(g1) < len(f):
            # keep the smaller factor
            other = None
            if len(g1) - 1 > (len(f) - 1) // 2:
                # f / g1
                q, rem = [], list(f)
                ib = F.inv(g1[-1])
                while len(rem) >= len(g1):
                    c = F.mul(rem[-1], ib)
                    q.append(c)
                    sh = len(rem) - len(g1)
                    for i, bc in enumerate(g1):
                        rem[sh + i] ^= F.mul(c, bc)
                    rem.pop()
                other = list(reversed(q))
            f = other if other else g1
    # f = c1*y + c0  -> root c0/c1
    return F.mul(f[0], F.inv(f[1]))


def solve_gf2(cols, target, n):
    """find x with sum_j x_j cols[j] = target over GF(2) (cols: n ints of n bits)"""
    rows = []
    for j, c in enumerate(cols):
        rows.append((c, 1 << j))
    basis = {}
    for v, combo in rows:
        while v:
            p = v.bit_length() - 1
            if p in basis:
                bv, bc = basis[p]
                v ^= bv
                combo ^= bc
            else:
                basis[p] = (v, combo)
                break
    x, t = 0, target
    while t:
        p = t.bit_length() - 1
        bv, bc = basis[p]
        t ^= bv
        x ^= bc
    return x


def invert_bitmatrix(cols, n):
    rows = [0] * n
    for j, cv in enumerate(cols):
        for i in range(n):
            if (cv >> i) & 1:
                rows[i] |= 1 << j
    aug = [(rows[i], 1 << i) for i in range(n)]
    for col in range(n):
        piv = next(r for r in range(col, n) if (aug[r][0] >> col) & 1)
        aug[col], aug[piv] = aug[piv], aug[col]
        for r in range(n):
            if r != col and (aug[r][0] >> col) & 1:
                aug[r] = (aug[r][0] ^ aug[col][0], aug[r][1] ^ aug[col][1])
    return [a[1] for a in aug]


def paar(rows, nin):
    """greedy pairwise CSE.  returns (gates, outs): gates = list of (new, a, b); outs[i] = list of vars to XOR"""
    rows = list(rows)
    nv = nin
    gates = []
    while True:
        cnt = {}
        for r in rows:
            bits = [i for i in range(nv) if (r >> i) & 1]
            for x in range(len(bits)):
                for y in range(x + 1, len(bits)):
                    key = (bits[x], bits[y])
                    cnt[key] = cnt.get(key, 0) + 1
        if not cnt:
            break
        bp = max(cnt, key=lambda kk: (cnt[kk], -kk[0], -kk[1]))
        if cnt[bp] < 2:
            break
        a, b = bp
        gates.append((nv, a, b))
        msk = (1 << a) | (1 << b)
        rows = [(r ^ msk) | (1 << nv) if (r & msk) == msk else r for r in rows]
        nv += 1
    outs = [[i for i in range(nv) if (r >> i) & 1] for r in rows]
    return gates, outs


def emit_program(name, rows, nin, in_expr, out_expr, doc):
    """straight-line program:  out[o] = XOR of in[i] for bits of rows[o]"""
    gates, outs = paar(rows, nin)
    nx = len(gates) + sum(max(0, len(o) - 1) for o in outs)
    lines = ["// %s  (%d inputs -> %d outputs, %d XORs after CSE)" % (doc, nin, len(rows), nx),
             "#define %s(IN, OUT) do { \\" % name]

    def var(i):
        return in_expr % i if i < nin else "t%d_" % i

    # emission order: output by output, each temporary right before its first use (depth first), and the outputs
    # in an order that keeps temporaries shared by few outputs short-lived -- the straight-line program is
    # register-bound on the GPU (128 inputs in LDS, 32 outputs and the live temporaries in VGPRs)
    gate = {n: (a, b) for (n, a, b) in gates}
    need = []
    for vs in outs:
        seen, stack = set(), [v for v in vs if v >= nin]
        while stack:
            v = stack.pop()
            if v in seen:
                continue
            seen.add(v)
            stack.extend(x for x in gate[v] if x >= nin)
        need.append(seen)
    order, done_t, left = [], set(), set(range(len(outs)))
    while left:
        o = min(left, key=lambda i: (len(need[i] - done_t), i))  # cheapest next output given what is already live
        left.remove(o)
        order.append(o)
        done_t |= need[o]
    emitted = set()

    def emit_temp(v):
        if v < nin or v in emitted:
            return
        a, b = gate[v]
        emit_temp(a)
        emit_temp(b)
        emitted.add(v)
        lines.append("  const u32 t%d_ = %s ^ %s; \\" % (v, var(a), var(b)))

    for o in order:
        vs = outs[o]
        for v in vs:
            emit_temp(v)
        expr = " ^ ".join(var(v) for v in vs) if vs else "0u"
        lines.append("  %s = %s; \\" % (out_expr % o, expr))
    lines.append("} while (0)")
    return "\n".join(lines), nx


def main():
    rng = random.Random(20261004)
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "longfellow-zk_amd", "csrc")
    for k in (4, 5):
        m, d = 1 << k, 128 >> k
        mu = MU[k]
        g = subfield_generator(k)
        nu = minpoly(g, m)
        F = K(mu, m)
        r = find_root(F, nu, m, rng)  # root of nu in K: psi(g) = r
        # check nu(r) = 0 in K
        acc = 0
        for j in range(m, -1, -1):
            acc = F.mul(acc, r)
            if (nu >> j) & 1:
                acc ^= 1
        assert acc == 0
        # psi(g^j) = r^j ; find c with sum c_j r^j = z (=2)  ->  h = sum c_j g^j
        rp, gp = [1], [1]
        for _ in range(m - 1):
            rp.append(F.mul(rp[-1], r))
            gp.append(gmul(gp[-1], g))
        c = solve_gf2(rp, 2, m)
        h = 0
        for j in range(m):
            if (c >> j) & 1:
                h ^= gp[j]
        # check mu(h) = 0 in GF(2^128)
        acc = 0
        for j in range(m, -1, -1):
            acc = gmul(acc, h)
            if (mu >> j) & 1:
                acc ^= 1
        assert acc == 0, "h is not a root of mu"
        hp = [1]
        for _ in range(m - 1):
            hp.append(gmul(hp[-1], h))
        cols = [gmul(hp[j], 1 << i) for i in range(d) for j in range(m)]  # tower bit p=i*m+j -> poly image
        to_tower = invert_bitmatrix(cols, 128)  # row p: mask over poly bits
        to_poly = [0] * 128
        for p, cv in enumerate(cols):
            for i in range(128):
                if (cv >> i) & 1:
                    to_poly[i] |= 1 << p
        # sanity: round trip on random values
        for _ in range(20):
            x = rng.getrandbits(128)
            t = 0
            for p in range(128):
                if bin(to_tower[p] & x).count("1") & 1:
                    t |= 1 << p
            y = 0
            for p in range(128):
                if (t >> p) & 1:
                    y ^= cols[p]
            assert y == x
        # subfield multiplication check: tower(s * x) coordinate-wise = s *_mu coord
        out = ["// GENERATED by tools/gen_tower.py -- do not edit.",
               "// Tower basis of GF(2^128) over GF(2^%d): element = sum_{i<%d} sum_{j<%d} c[i*%d+j] h^j X^i," % (m, d, m, m),
               "// h = root in the LCH14 subfield of mu = %#x.  Plane index p = i*%d + j." % (mu, m),
               "#pragma once",
               "#define TOWER_K%d_M %d" % (k, m),
               "#define TOWER_K%d_D %d" % (k, d),
               "#define TOWER_K%d_MU_LOW %#xu  // mu minus the leading term" % (k, mu ^ (1 << m)),
               "static const unsigned long long kTowerK%dH[2] = {%#xull, %#xull};  // h (polynomial basis)" % (k, h & (2**64 - 1), h >> 64),
               "// rows of the poly->tower map for the first %d tower bits (twiddle coordinates): bit j of t = parity(mask & poly)" % m,
               "static const unsigned long long kTowerK%dTwMask[%d][2] = {" % (k, m)]
        for j in range(m):
            out.append("  {%#xull, %#xull}," % (to_tower[j] & (2**64 - 1), to_tower[j] >> 64))
        out.append("};")
        total = 0
        # poly -> tower, one program per output coordinate (32 or 16 planes), inputs = 128 poly planes
        for q in range(d):
            prog, nx = emit_program("TOWER_K%d_P2T_Q%d" % (k, q), to_tower[q * m:(q + 1) * m], 128, "IN(%d)", "OUT(%d)",
                                    "poly planes -> tower coordinate %d" % q)
            out.append(prog)
            total += nx
        if m == 32:  # half programs (16 outputs): lower register pressure, 2x the tasks
            for q in range(d):
                for hh in range(2):
                    prog, nx = emit_program("TOWER_K%d_P2T_Q%dH%d" % (k, q, hh), to_tower[q * m + 16 * hh:q * m + 16 * hh + 16], 128,
                                            "IN(%d)", "OUT(%d)", "poly planes -> tower coordinate %d, planes %d..%d" % (q, 16 * hh, 16 * hh + 15))
                    out.append(prog)
                    halves_total = nx
        if m == 32:  # quarter programs (8 outputs): <= 128 VGPRs for the conversion kernels -> 4 waves per SIMD
            for q in range(d):
                for o in range(4):
                    prog, nx = emit_program("TOWER_K%d_P2T_Q%dO%d" % (k, q, o), to_tower[q * m + 8 * o:q * m + 8 * o + 8], 128,
                                            "IN(%d)", "OUT(%d)", "poly planes -> tower coordinate %d, planes %d..%d" % (q, 8 * o, 8 * o + 7))
                    out.append(prog)
            for w in range(4):
                for o in range(4):
                    prog, nx = emit_program("TOWER_K%d_T2P_W%dO%d" % (k, w, o), to_poly[w * 32 + 8 * o:w * 32 + 8 * o + 8], 128,
                                            "IN(%d)", "OUT(%d)", "tower planes -> poly dword %d, bits %d..%d" % (w, 8 * o, 8 * o + 7))
                    out.append(prog)
        # tower -> poly, output chunks of 32 poly planes (one dword of the element), inputs = 128 tower planes
        for w in range(4):
            prog, nx = emit_program("TOWER_K%d_T2P_W%d" % (k, w), to_poly[w * 32:(w + 1) * 32], 128, "IN(%d)", "OUT(%d)",
                                    "tower planes -> poly dword %d" % w)
            out.append(prog)
            total += nx
        path = os.path.join(outdir, "tower_k%d.h" % k)
        with open(path, "w") as f:
            f.write("\n".join(out) + "\n")
        print("k=%d: h=%#x, wrote %s (%d XORs total for both directions)" % (k, h, path, total))


if __name__ == "__main__":
    main()
