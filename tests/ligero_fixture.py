"""Parser for tests/golden/ligero_test_vector.bin -- the C++-generated GF2_128 Ligero vector the
reference's Rust port checks (layout: rust/runtime/ligero/tests/ligero.rs:594-681; statement
LigeroParam(1000, 50, rateinv 4, nreq 36, block_enc 4096), transcript "test", LCG seed 100)."""
import os
import struct

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class LcgRng:
    """SimpleRng of rust/runtime/ligero/tests/ligero.rs:28-43 (= the C++ generator's engine)"""

    def __init__(self, seed):
        self.state = seed

    A, Cc = 6364136223846793005, 1442695040888963407

    def bytes(self, n):
        if n > 32:  # vectorised: s_i = A^i s_0 + C (1 + A + ... + A^(i-1)) in wrapping 64-bit arithmetic
            with np.errstate(over="ignore"):
                ap = np.cumprod(np.full(n, self.A, dtype=np.uint64))
                g = np.empty(n, dtype=np.uint64)
                g[0] = 1
                if n > 1:
                    g[1:] = np.cumsum(ap[:-1], dtype=np.uint64) + np.uint64(1)
                st = ap * np.uint64(self.state) + g * np.uint64(self.Cc)
            self.state = int(st[-1])
            return ((st >> np.uint64(32)) & np.uint64(0xFF)).astype(np.uint8).tobytes()
        out = bytearray(n)
        s = self.state
        for i in range(n):
            s = (s * self.A + self.Cc) & 0xFFFFFFFFFFFFFFFF
            out[i] = (s >> 32) & 0xFF
        self.state = s
        return bytes(out)


def load():
    d = open(os.path.join(GOLD, "ligero_test_vector.bin"), "rb").read()
    off = 0

    def u64():
        nonlocal off
        (v,) = struct.unpack_from("<Q", d, off)
        off += 8
        return v

    def elts(n):
        nonlocal off
        a = np.frombuffer(d, dtype=np.uint64, count=2 * n, offset=off).reshape(n, 2).copy()
        off += 16 * n
        return a

    v = {}
    v["nw"], v["nq"], v["nreq"], v["nl"], v["subfield_boundary"] = u64(), u64(), u64(), u64(), u64()
    v["W"] = elts(v["nw"])
    v["A"] = elts(v["nw"])
    v["lqc"] = [(u64(), u64(), u64()) for _ in range(v["nq"])]
    nll = u64()
    ll = []
    for _ in range(nll):
        c, w = u64(), u64()
        ll.append((c, w, elts(1)[0]))
    v["llterm"] = ll
    v["b"] = elts(v["nl"])
    v["hash_of_statement"] = d[off:off + 32]
    off += 32
    v["root"] = d[off:off + 32]
    off += 32
    plen = u64()
    v["proof"] = d[off:off + plen]
    off += plen
    assert off == len(d)
    return v
