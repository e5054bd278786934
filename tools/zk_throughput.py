"""Throughput mode: K concurrent provers on ONE device (K host threads, each with its own lfgpu context + stream, prover and
transcripts; ONE copy of every circuit in HBM through lfgpu_circuit_share).

The reference's benchmark proves one statement after the other on one core (BM_ShaZK_fp2_128,
lib/circuits/sha/flatsha256_circuit_test.cc:510-536); a single proof is latency-bound on the GPU (a chain of a few hundred
Fiat-Shamir round trips) and leaves the device mostly idle, so independent proofs -- SURVEY.md section 8(e): "replicas" --
are what fills it.  A job is one statement: `flatsha32` = commit + prove of the 32-block flatsha256 circuit; `mdoc` = what
run_mdoc_prover does (lib/circuits/mdoc/mdoc_zk.cc:494-522): both commits (hash circuit over GF2_128, signature circuit over
Fp256Base), then both proofs.  Every worker first proves each of its circuits once with the fixtures' RandomEngine and
transcript seed and REQUIRES the reference's wire bytes (untimed); the timed jobs then draw from a C-speed engine, the way the
benchmark uses SecureRandomEngine.

  python tools/zk_throughput.py [--jobs flatsha32,mdoc] [--k 1,2,4,8,16] [--seconds 2.0] [--json]

Prints one JSON object: {job: {"k": {K: {"proofs_per_s", "ms_per_proof_latency", "proofs"}}, ...}}."""
import ctypes as C
import hashlib
import json
import lzma
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
# more hardware queues than HIP's default of 4: streams beyond that share a queue and their kernels serialise (read by the
# runtime when it initialises, so it must be in the environment before the first HIP call of the process)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
JOBS = {"flatsha1": ["flatsha_nb1"], "flatsha32": ["flatsha_nb32"], "mdoc": ["mdoc_hash", "mdoc_sig"], "mdoc_hash": ["mdoc_hash"], "mdoc_sig": ["mdoc_sig"]}


def load_stem(stem):
    raw = lzma.decompress(open(os.path.join(GOLD, stem + ".lfc1.xz"), "rb").read())
    sig = stem == "mdoc_sig"
    W = np.frombuffer(lzma.decompress(open(os.path.join(GOLD, stem + ".w.xz"), "rb").read()), dtype=np.uint64).reshape(-1, 4 if sig else 2).copy()
    if stem.startswith("mdoc"):
        info = json.load(open(os.path.join(GOLD, "mdoc.json")))["sig" if sig else "hash"]
        be = info["block_enc"]
    else:
        info = json.load(open(os.path.join(GOLD, stem + ".json")))
        be = 0
    return dict(stem=stem, raw=raw, W=W, be=be, wire_bytes=info["zk_wire_bytes"], wire_sha=info["zk_wire_sha256"])


class Worker:
    """one prover thread: context with its own stream, shared circuits, one ZkProver per circuit"""

    def __init__(self, pkg, base_circuits, stems, device=0):
        import ligero_fixture as lf
        self.pkg, self.lf = pkg, lf
        self.gpu = pkg.LfGpu(device).own_stream()
        self.L = self.gpu.L
        self.items = []
        for st in stems:
            circ = base_circuits[st["stem"]].share(self.gpu)
            zk = pkg.ZkProver(self.gpu, circ, 7, 132, st["be"])
            self.items.append((st, circ, zk, C.c_void_p(st["W"].ctypes.data)))
        self.rng_t = pkg.FsTranscript(b"rng-%d" % id(self))
        self.rng_fn = C.cast(self.L.lfgpu_transcript_bytes, pkg.RNG_FN)
        self.done = 0
        self.lat = []
        self.err = None
        self.phases = None  # {stem.phase: summed ms} when the caller wants the provers' own phase timers

    def check_parity(self):
        """the fixtures' engine and seed: the wire bytes must be the reference's"""
        tss = []
        for st, circ, zk, _ in self.items:
            ts = self.pkg.FsTranscript(b"test")
            zk.commit(st["W"], self.lf.LcgRng(100).bytes, ts)
            tss.append(ts)
        for (st, circ, zk, _), ts in zip(self.items, tss):
            assert zk.prove(st["W"], ts), "prove failed: " + st["stem"]
            wire = zk.wire()
            ts.close()
            assert len(wire) == st["wire_bytes"] and hashlib.sha256(wire).hexdigest() == st["wire_sha"], "wire bytes differ from the reference: " + st["stem"]

    def one_job(self):
        """run_mdoc_prover's order: every commit, then every proof"""
        L, gpu = self.L, self.gpu
        root = (C.c_uint8 * 32)()
        ok = C.c_int()
        opss = []
        for st, circ, zk, Wp in self.items:
            ts = self.pkg.FsTranscript(b"test")
            ops = ts.ops()
            gpu._ck(L.lfgpu_zk_commit(zk.h, Wp, self.rng_fn, self.rng_t.h, C.byref(ops), root))
            opss.append((ts, ops))
        for (st, circ, zk, Wp), (ts, ops) in zip(self.items, opss):
            gpu._ck(L.lfgpu_zk_prove(zk.h, Wp, C.byref(ops), C.byref(ok)))
            assert ok.value == 1
            ts.close()
            if self.phases is not None:
                for k, v in zk.timings().items():
                    self.phases[st["stem"] + "." + k] = self.phases.get(st["stem"] + "." + k, 0.0) + v

    def run(self, start_evt, stop_at):
        try:
            start_evt.wait()
            while time.perf_counter() < stop_at[0]:
                t0 = time.perf_counter()
                self.one_job()
                self.lat.append(time.perf_counter() - t0)
                self.done += 1
        except Exception as e:  # surfaces in the main thread
            self.err = e

    def close(self):
        for st, circ, zk, _ in self.items:
            zk.close()
            circ.close()
        self.rng_t.close()
        self.gpu.close()


class SmiSampler:
    """mean GPU busy % over a run (rocm-smi --showuse, ~4 samples per second); None when rocm-smi is not usable"""

    def __init__(self, device):
        self.device, self.vals, self._stop = device, [], threading.Event()
        self.th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", str(self.device), "--showuse"], capture_output=True, text=True, timeout=5).stdout
                m = re.search(r"GPU use \(%\):\s*(\d+)", out)
                if m:
                    self.vals.append(int(m.group(1)))
            except Exception:  # noqa: BLE001
                return
            self._stop.wait(0.2)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self.th.join(timeout=6)

    def mean(self):
        return round(sum(self.vals) / len(self.vals), 1) if self.vals else None


def measure(pkg, base_gpu, job, ks, seconds, warm_jobs=2, log=None, device=0, smi=False, phases=False):
    stems = [load_stem(s) for s in JOBS[job]]
    base = {st["stem"]: pkg.Circuit(base_gpu, st["raw"]) for st in stems}
    out = {}
    workers = []
    try:
        for K in ks:
            while len(workers) < K:
                w = Worker(pkg, base, stems, device)
                w.check_parity()  # also the first proof of the handle: fills its per-circuit caches
                for _ in range(warm_jobs):
                    w.one_job()
                workers.append(w)
            act = workers[:K]
            for w in act:
                w.done, w.lat, w.phases = 0, [], ({} if phases else None)
            start, stop_at = threading.Event(), [0.0]
            th = [threading.Thread(target=w.run, args=(start, stop_at)) for w in act]
            for t in th:
                t.start()
            sampler = SmiSampler(device) if smi else None
            if sampler:
                sampler.__enter__()
            t0 = time.perf_counter()
            stop_at[0] = t0 + seconds
            start.set()
            for t in th:
                t.join()
            wall = time.perf_counter() - t0
            if sampler:
                sampler.__exit__()
            for w in act:
                if w.err:
                    raise w.err
            n = sum(w.done for w in act)
            lat = sorted(x for w in act for x in w.lat)
            out[str(K)] = {"proofs_per_s": round(n / wall, 2), "proofs": n, "wall_s": round(wall, 3),
                           "ms_per_proof_latency_median": round(1e3 * lat[len(lat) // 2], 3) if lat else None}
            if sampler:
                out[str(K)]["gpu_busy_pct_mean_rocm_smi"] = sampler.mean()
            if phases and n:
                keys = sorted({k for w in act for k in w.phases})
                out[str(K)]["phase_ms_mean"] = {k: round(sum(w.phases.get(k, 0.0) for w in act) / n, 3) for k in keys}
            if log:
                log("zk_throughput %s K=%d: %.1f proofs/s (%d proofs, median latency %.2f ms)" % (job, K, n / wall, n, 1e3 * lat[len(lat) // 2]))
    finally:
        for w in workers:
            w.close()
        for c in base.values():
            c.close()
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", default="flatsha32,mdoc")
    ap.add_argument("--k", default="1,2,4,8,16")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--smi", action="store_true", help="sample rocm-smi --showuse during every run")
    ap.add_argument("--phases", action="store_true", help="mean of the provers' own phase timers (lfgpu_zk_timings) per proof")
    a = ap.parse_args()
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    pkg = ge.load_package()
    base_gpu = pkg.LfGpu(a.device)  # uploads the circuits; proves nothing (no stream of its own: it does not count as a sharer of the device)
    res = {"hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES")}
    log = (lambda s: print(s, file=sys.stderr, flush=True))
    for job in a.jobs.split(","):
        res[job] = {"circuits": JOBS[job], "k": measure(pkg, base_gpu, job, [int(k) for k in a.k.split(",")], a.seconds, log=log, device=a.device, smi=a.smi, phases=a.phases)}
    base_gpu.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
