"""repeat the Fp256Base proof of the mdoc signature circuit (fixture randomness) and compare the wire bytes every time: the
limb-atomic sums and the posted round results must be exact and order-independent, run after run"""
import hashlib, json, lzma, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpu_util as G
import ligero_fixture as lf
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
gold = os.path.join(ROOT, "tests", "golden")
meta = json.load(open(os.path.join(gold, "mdoc.json")))
info = meta["sig"]
raw = lzma.decompress(open(os.path.join(gold, "mdoc_sig.lfc1.xz"), "rb").read())
W = np.frombuffer(lzma.decompress(open(os.path.join(gold, "mdoc_sig.w.xz"), "rb").read()), dtype=np.uint64).reshape(-1, 4).copy()
pkg, gpu = G.pkg, G.gpu()
circ = pkg.Circuit(gpu, raw)
zk = pkg.ZkProver(gpu, circ, meta["hash"]["rate"], meta["hash"]["nreq"], info["block_enc"])
bad = 0
for i in range(reps):
    ts = pkg.FsTranscript(b"test")
    zk.commit(W, lf.LcgRng(100).bytes, ts)
    ok = zk.prove(W, ts)
    wire = zk.wire() if ok else b""
    ts.close()
    if hashlib.sha256(wire).hexdigest() != info["zk_wire_sha256"]:
        bad += 1
        print("MISMATCH at repetition", i, flush=True)
    if i % 20 == 0:
        tv = pkg.FsTranscript(b"test")
        okv, why = pkg.zk_verify(gpu, circ, wire, W[:circ.info.npub_in], tv, meta["hash"]["rate"], meta["hash"]["nreq"], info["block_enc"])
        tv.close()
        if not okv:
            bad += 1
            print("VERIFY FAILED at", i, why, flush=True)
print(json.dumps({"repetitions": reps, "mismatches": bad}))
sys.exit(1 if bad else 0)
