"""stress: many commit + prove + verify cycles (fresh randomness each time) through the resident sumcheck kernels;
every proof must verify.  Looks for rare races in the host/device handshake and the device-wide barriers."""
import ctypes as C, json, lzma, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpu_util as G

stem, iters = sys.argv[1], int(sys.argv[2])
gold = os.path.join(ROOT, "tests", "golden")
raw = lzma.decompress(open(os.path.join(gold, stem + ".lfc1.xz"), "rb").read())
W = np.frombuffer(lzma.decompress(open(os.path.join(gold, stem + ".w.xz"), "rb").read()), dtype=np.uint64).reshape(-1, 2).copy()
pkg, gpu = G.pkg, G.gpu()
circ = pkg.Circuit(gpu, raw)
zk = pkg.ZkProver(gpu, circ, 7, 132)
L = gpu.L
rng_t = pkg.FsTranscript(b"stress rng " + stem.encode())
rng_fn = C.cast(L.lfgpu_transcript_bytes, pkg.RNG_FN)
Wp, root, ok = C.c_void_p(W.ctypes.data), (C.c_uint8 * 32)(), C.c_int()
t0 = time.time()
sizes = set()
for it in range(iters):
    seed = b"seed %d" % it
    ts = pkg.FsTranscript(seed)
    ops = ts.ops()
    gpu._ck(L.lfgpu_zk_commit(zk.h, Wp, rng_fn, rng_t.h, C.byref(ops), root))
    gpu._ck(L.lfgpu_zk_prove(zk.h, Wp, C.byref(ops), C.byref(ok)))
    assert ok.value == 1, it
    wire = zk.wire()
    ts.close()
    tv = pkg.FsTranscript(seed)
    acc, why = pkg.zk_verify(gpu, circ, wire, W[:circ.info.npub_in], tv)
    tv.close()
    assert acc, (it, why)
    sizes.add(len(wire))
print(json.dumps({"stem": stem, "iterations": iters, "all_verified": True, "seconds": round(time.time() - t0, 1), "proof_sizes": [min(sizes), max(sizes)]}))
