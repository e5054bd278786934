"""BASELINE headline path, BM_ShaZK_fp2_128 (ZkProver commit + prove on the flatsha256 GF2_128 circuit): phase
timings of the GPU-kernel path driven by the Python test harness (tests/zk_driver.py) next to the reference CPU
prover (oracle/_ref/gen_flatsha on this host).  The proof is asserted byte-identical before timing.  The host
loop here is Python (transcript, constraint bookkeeping, RNG), so the wall time is an upper bound on what a C++
integration pays; the per-phase split shows where the kernels stand."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import gpu_util as G
import sumcheck_driver as sd
import zk_driver as zd
from fs_transcript import Transcript


class FastLcg:
    """vectorised SimpleRng (rust/runtime/ligero/tests/ligero.rs:28-43): state_n = a^n s0 + c * sum_{k<n} a^k mod 2^64"""
    A, Cc = np.uint64(6364136223846793005), np.uint64(1442695040888963407)

    def __init__(self, seed, total):
        with np.errstate(over="ignore"):
            pw = np.cumprod(np.full(total, self.A, dtype=np.uint64))
            S = np.cumsum(np.concatenate([np.ones(1, dtype=np.uint64), pw[:-1]]))
            st = pw * np.uint64(seed) + self.Cc * S
        self.buf = ((st >> np.uint64(32)) & np.uint64(0xFF)).astype(np.uint8).tobytes()
        self.pos = 0

    def bytes(self, n):
        b = self.buf[self.pos:self.pos + n]
        assert len(b) == n
        self.pos += n
        return b


nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
gold = os.path.join(ROOT, "tests", "golden")
circ, W, _, info = sd.load_fixture(gold, nb)
want = zd.load_zk_fixture(gold, nb)
zp = zd.ZkProverGpu(G.pkg, G.gpu(), circ)
res = {"nb": nb, "shape": {k: info[k] for k in ("zk_nw", "zk_block_enc", "zk_nrow", "nterms", "round_hands")}}
for rep in range(2):  # second repetition is the timed one (tables, plans and code are warm)
    ts = Transcript(b"test")
    rng = FastLcg(100, 4_000_000)
    t0 = time.perf_counter(); root = zp.commit(W, rng, ts); torch.cuda.synchronize(); t1 = time.perf_counter()
    c = zp.c
    ts.write_bytes(c["id"]); ts.write_elt(b"\x00" * 16); ts.write_bytes(b"\x00" * info["nterms"])
    tst = ts.clone()
    ta = time.perf_counter(); ins, V = zp.sc.eval_circuit(W); tb = time.perf_counter()
    proof, aux = zp._padded_sumcheck(ins, tst); tc = time.perf_counter()
    a_small, dense, b, ci = zp._verifier_constraints(W, proof, aux, ts); td = time.perf_counter()
    com = zp._ligero_prove(ts, ci, a_small, dense); te = time.perf_counter()
    got = zd.serialize(circ, root, dict(sumcheck=proof, **com))
    assert got == want, "proof differs from the reference"
    zp.lp.close()
res.update({"bit_exact_vs_reference": True, "gpu_path_ms": {
    "commit (host RNG draw + row layout in C++, RS encode, column hash, tree)": (t1 - t0) * 1e3,
    "eval_circuit": (tb - ta) * 1e3, "sumcheck (lfgpu_sumcheck_layer + Python transcript callback)": (tc - tb) * 1e3,
    "verifier_constraints (host, Python)": (td - tc) * 1e3, "ligero prove (host challenges + K12/K3 + open)": (te - td) * 1e3,
    "total": (t1 - t0 + te - ta) * 1e3}})
gen = os.path.join(ROOT, "oracle", "_ref", "gen_flatsha")
if os.path.exists(gen):
    with tempfile.TemporaryDirectory() as td_:
        r = json.loads(subprocess.check_output([gen, str(nb), os.path.join(td_, "x")]).decode())
    res["cpu_reference_ms"] = {"commit": r["ref_zk_commit_ms"], "prove": r["ref_zk_prove_ms"], "total": r["ref_zk_commit_ms"] + r["ref_zk_prove_ms"]}
print(json.dumps(res))
