"""synthetic sumcheck layers in the reference's canonical order: corners sorted by
Morton(h0,h1) then g (EQuad::canonicalize, lib/sumcheck/equad.h:79-106), h0 <= h1, unique
(g,h0,h1); kvec[0] = 0 marks assert-zero terms (quad.h:213-220)."""
import numpy as np

import oracle_lib as ol


def morton(a, b):
    m = 0
    for i in range(24):
        m |= ((a >> i) & 1) << (2 * i) | ((b >> i) & 1) << (2 * i + 1)
    return m


def make_layer(rng, field, logv, logw, nterms, nk=9, n_assert=0, satisfy=True):
    nv, nw = 1 << logv, 1 << logw
    kvec = ol.rand_elts(rng, nk, field)
    kvec[0] = 0
    W = ol.rand_elts(rng, nw, field)
    zero_wires = set(int(x) for x in rng.choice(nw, size=max(1, nw // 8), replace=False))
    for z in zero_wires:
        W[z] = 0
    zl = sorted(zero_wires)
    seen, terms = set(), []
    while len(terms) < nterms:
        g = int(rng.integers(0, nv))
        a, b = int(rng.integers(0, nw)), int(rng.integers(0, nw))
        # cluster hands so that several g share a hand pair (exercises the merge in bind_g)
        if rng.random() < 0.5 and terms:
            _, a, b, _ = terms[int(rng.integers(0, len(terms)))]
        vi = int(rng.integers(1, nk))
        if len(terms) < n_assert:
            vi = 0
            a = zl[int(rng.integers(0, len(zl)))] if satisfy else a
        h0, h1 = min(a, b), max(a, b)
        if (g, h0, h1) in seen:
            continue
        seen.add((g, h0, h1))
        terms.append((g, h0, h1, vi))
    terms.sort(key=lambda t: (morton(t[1], t[2]), t[0]))
    t = np.array(terms, dtype=np.uint32)
    return dict(g=np.ascontiguousarray(t[:, 0]), h0=np.ascontiguousarray(t[:, 1]), h1=np.ascontiguousarray(t[:, 2]),
                vi=np.ascontiguousarray(t[:, 3]), kvec=kvec, W=W, nv=nv, nw=nw, logv=logv, logw=logw, n=len(terms))
