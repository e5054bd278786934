"""BASELINE configs[3]: sumcheck on the flatsha256 GF2_128 circuit -- GPU-stepped prover (Python host loop +
C-ABI kernels) vs the reference CPU prover (oracle/_ref/gen_flatsha, when present).  Bit-exactness is
asserted against the committed fixture before timing."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_util as G
import sumcheck_driver as sd
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
circ, W, proof, info = sd.load_fixture(os.path.join(ROOT, "tests", "golden"), nb)
t0 = time.perf_counter(); sc = sd.GpuSumcheck(G.pkg, G.gpu(), circ); torch.cuda.synchronize(); t_up = time.perf_counter() - t0
ins, V = sc.eval_circuit(W)
assert (V == 0).all() and sc.prove(ins, W) == proof, "not bit-exact"
t0 = time.perf_counter(); ins, V = sc.eval_circuit(W); torch.cuda.synchronize(); t_ev = time.perf_counter() - t0
t0 = time.perf_counter(); got = sc.prove(ins, W); t_pr = time.perf_counter() - t0
out = {"nb": nb, "nterms": info["nterms"], "round_hands": info["round_hands"], "bit_exact_vs_reference_fixture": got == proof,
       "gpu_upload_circuit_ms": t_up * 1e3, "gpu_eval_circuit_ms": t_ev * 1e3, "gpu_sumcheck_prove_ms": t_pr * 1e3,
       "note": "prove = Python host loop (transcript, 3-point polynomial via ctypes) + 5 kernel groups per round-hand"}
sc2 = sd.GpuSumcheckLayerApi(G.pkg, G.gpu(), circ)
ins, _ = sc2.eval_circuit(W)
assert sc2.prove(ins, W) == proof
ins, _ = sc2.eval_circuit(W); torch.cuda.synchronize()
t0 = time.perf_counter(); got2 = sc2.prove(ins, W); out["gpu_sumcheck_prove_layer_api_ms"] = (time.perf_counter() - t0) * 1e3
out["layer_api_bit_exact"] = got2 == proof
gen = os.path.join(ROOT, "oracle", "_ref", "gen_flatsha")
if os.path.exists(gen):
    with tempfile.TemporaryDirectory() as td:
        r = json.loads(subprocess.check_output([gen, str(nb), os.path.join(td, "x")]).decode())
    out["cpu_reference_eval_circuit_ms"] = r["ref_eval_circuit_ms"]; out["cpu_reference_sumcheck_ms"] = r["ref_sumcheck_ms"]
print(json.dumps(out))
