#!/usr/bin/env python3
"""bench.py -- headline measurement for the MI355X prover hot path.

Workload (BASELINE.json configs[1]): batched FFT, 2^20 points x 1024 rows of 16-byte
field elements (16 GiB resident in HBM), one "step" = one pass of the batch through
lfgpu_fp128_fft (FFT<Fp128>::fftb, reference lib/algebra/fft.h:185-195).  The same batch
through the GF(2^128) LCH14 additive FFT (lib/gf2k/lch14.h:106-124) is reported as
secondary keys.  N > 1: rows are independent, each rank owns its own 1024 rows (weak
scaling, no data-path collective); value = all ranks' elements / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(logn, budget_s=10.0):
    """Reference CPU path (oracle/_ref = the real reference compiled in the build container) on a bounded sample of the
    same workload; falls back to the C port.  `value` is ONE thread (the reference's FFT is single-threaded); SURVEY 8(d)
    also asks for independent rows on all host cores: `value_all_cores` (rows are independent, one thread per core)."""
    import numpy as np
    import oracle_lib as ol
    from concurrent.futures import ThreadPoolExecutor

    n = 1 << logn
    o = ol.oracle()
    r = ol.ref()
    a = np.zeros((n, 2), dtype=np.uint64)
    o.lfo_fp_bogorng_fill(1234569, n, ol.P(a))

    def one_row(_=None):
        x = a.copy()
        if r is not None:
            r.ref_fp_fft(0, n, ol.P(x))  # ctypes releases the GIL for the duration of the call
        else:
            o.lfo_fp_fftb(ol.P(x), n, o.lfo_fp_omega32(), 1 << 32)

    # one thread, pinned to one core of this job's share for the duration (round 2 saw this figure move 2.5x between runs with
    # the thread free to migrate and the measurement at the end of the run, behind every GPU leg's spinning host threads): it
    # now runs FIRST, pinned, and reports the best row next to the mean
    aff = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    if aff:
        os.sched_setaffinity(0, {aff[len(aff) // 2]})
    rows, t0, best_row = 0, time.perf_counter(), None
    try:
        while True:
            tr = time.perf_counter()
            one_row()
            tr = time.perf_counter() - tr
            best_row = tr if best_row is None or tr < best_row else best_row
            rows += 1
            dt = time.perf_counter() - t0
            if dt > budget_s * 0.5 or rows >= 64:
                break
    finally:
        if aff:
            os.sched_setaffinity(0, set(aff))
    one = n / best_row
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # all cores: one thread per core of this job's CPU share (a 1-GPU box gives 16), every thread transforms rows until the
    # deadline -- bounded wall time whatever the core count
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))
    deadline = time.perf_counter() + budget_s * 0.5
    done = [0] * cores

    def worker(i):
        while time.perf_counter() < deadline:
            one_row()
            done[i] += 1

    t1 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(worker, range(cores)))
    dt2 = time.perf_counter() - t1
    nrows_all = sum(done)
    return {"value": one, "unit": "field-elems/s", "cores": 1,
            "kind": "reference" if r is not None else "port",
            "sample": "%d rows of 2^%d Fp128 points through FFT<Fp128>::fftb, 1 pinned thread, %.1f s, best row (mean %.3g elem/s); host CPU: %s" % (rows, logn, dt, rows * n / dt, cpu_model),
            "value_mean": rows * n / dt, "host_cpu": cpu_model,
            "value_all_cores": nrows_all * n / dt2, "cores_all": cores,
            "sample_all_cores": "%d independent rows on %d threads, %.1f s" % (nrows_all, cores, dt2)}


def ligero_commit_shape(gpu, torch, np, stream, with_cpu):
    """BASELINE configs[2]: Ligero RS encode + Merkle column commit on the flatsha256 32-block tableau
    shape (GF2_128<4>: 150 rows, block 910, dblock 1819, block_enc 8192, block_ext 6373 -- SURVEY 6b),
    synthetic witness rows resident in HBM.  GPU: K3 (one launch for all rows) + K5 + K6."""
    import oracle_lib as ol
    nrow, block, dblock, be = 150, 910, 1819, 8192
    ext = be - dblock
    rng = np.random.default_rng(32)
    T = ol.rand_elts(rng, nrow * be).reshape(nrow, be, 2)
    nonces = rng.integers(0, 256, size=(ext, 32), dtype=np.uint8)
    dT0 = torch.from_numpy(T.view(np.int64).reshape(-1)).cuda()
    dT = dT0.clone()
    dN = torch.from_numpy(nonces).cuda()
    dL = torch.zeros(2 * ext * 32, dtype=torch.uint8, device="cuda")

    def run():
        p = dT.data_ptr()
        gpu.gf2128_rs_encode_tableau(p, nrow, block, dblock, 1, 3, be, ld=be)  # rows 1, 2 (IDOT, IQUAD) are dblock long
        return gpu.column_commit(4, nrow, be, dblock, ext, p, dN.data_ptr(), dL.data_ptr())

    root = run()
    reps = 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dT.copy_(dT0)
        run()
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    res = {"shape": "150 rows x 8192 (block 910, dblock 1819, 6373 leaves), GF2_128<4>", "gpu_ms": gpu_ms,
           "includes": "device copy of the 19.7 MB tableau + RS encode of all rows (one launch) + leaf hash + tree + 32-byte root readback"}
    if with_cpu:
        r = ol.ref()
        if r is not None:
            Tc = T.copy()
            t0 = time.perf_counter()
            r.ref_lch14_rs_encode_rows(4, 1, block, be, ol.P(Tc), be)
            r.ref_lch14_rs_encode_rows(4, 2, dblock, be, ol.P(Tc[1:]), be)
            r.ref_lch14_rs_encode_rows(4, nrow - 3, block, be, ol.P(Tc[3:]), be)
            rootc = np.zeros(32, dtype=np.uint8)
            r.ref_column_commit(4, nrow, be, dblock, ext, ol.P(Tc), ol.P(nonces), ol.P(rootc))
            res["cpu_reference_ms"] = (time.perf_counter() - t0) * 1e3
            res["root_matches_reference"] = bool(rootc.tobytes() == root)
    return res


def ligero_commit_slig(gpu, torch, np, A, rows, logn, pmc_all=None):
    """SURVEY 8(d) row 3, the synthetic Ligero shape large enough for an HBM roofline: LigeroParam(nw, nq = 0,
    rateinv = 4, nreq = 132, block_enc = 2^20) over GF2_128<5> (lib/ligero/ligero_param.h:185-243) => block = 174 762,
    dblock = 349 523, block_ext = 699 053 leaves; nrow = 1024 rows resident in HBM (the batch buffer A: its first
    `block` columns are the message).  Timed: K3 (RS-extend every row 174 762 -> 2^20) and K5 + K6 (column hash of the
    699 053 x 1024 x 16 B columns + tree).  Checked against the oracle on a sample: two whole rows of the RS extension,
    64 leaves, and the whole tree rebuilt from the device's leaves."""
    import ctypes as C
    import oracle_lib as ol
    be = 1 << logn
    block = (be + 1) // 6
    dblock = 2 * block - 1
    ext = be - dblock
    o = ol.oracle()
    ctx5 = ol.gf_ctx(5)
    msg = [A[r * be:r * be + block].cpu().numpy().view(np.uint64).copy() for r in (0, rows - 1)]
    stream = torch.cuda.current_stream()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    nonces = torch.randint(0, 256, (ext, 32), dtype=torch.uint8, device="cuda")
    layers = torch.zeros(2 * ext * 32, dtype=torch.uint8, device="cuda")
    # warm (tables, plans), then timed; the encode is idempotent on the message columns
    gpu.gf2128_rs_encode_rows(A.data_ptr(), rows, block, be, subfield_log_bits=5)
    torch.cuda.synchronize()
    e0.record(stream)
    gpu.gf2128_rs_encode_rows(A.data_ptr(), rows, block, be, subfield_log_bits=5)
    e1.record(stream)
    root = gpu.column_commit(4, rows, be, dblock, ext, A.data_ptr(), nonces.data_ptr(), layers.data_ptr())
    e2.record(stream)
    torch.cuda.synchronize()
    rs_ms, hash_ms = e0.elapsed_time(e1), e1.elapsed_time(e2)
    ok_rs = True
    for r, m0 in zip((0, rows - 1), msg):
        want = np.zeros((be, 2), dtype=np.uint64)
        want[:block] = m0
        o.lfo_lch14_rs_interpolate(C.byref(ctx5), block, be, ol.P(want))
        ok_rs = ok_rs and bool((A[r * be:(r + 1) * be].cpu().numpy().view(np.uint64) == want).all())
    # leaves of 64 sampled columns from a host copy of those columns; the tree from the device's leaves
    c0 = 12345
    cols = A.view(rows, be, 2)[:, dblock + c0:dblock + c0 + 64, :].contiguous().cpu().numpy().view(np.uint64)
    nz = nonces[c0:c0 + 64].cpu().numpy()
    want_leaves = np.zeros((64, 32), dtype=np.uint8)
    o.lfo_column_leaves(4, rows, 64, 0, 64, ol.P(np.ascontiguousarray(cols)), ol.P(np.ascontiguousarray(nz)), ol.P(want_leaves))
    lay = layers.view(2 * ext, 32).cpu().numpy()
    ok_leaves = bool((lay[ext + c0:ext + c0 + 64] == want_leaves).all())
    hl = np.zeros((2 * ext, 32), dtype=np.uint8)
    o.lfo_merkle_build_tree(ext, ol.P(np.ascontiguousarray(lay[ext:])), ol.P(hl))
    ok_tree = bool(hl[1].tobytes() == root)
    rs_bytes = rows * (block + be) * 16.0          # read the message, write the codeword
    hash_bytes = rows * ext * 16.0 + ext * 32.0    # read the columns, write the leaves
    pmc_all = pmc_all or {}
    rs_traffic = pmc_all.get("slig_rs_encode", {}).get("hbm_bytes_per_encode")      # sum over the encode's launches (PMC)
    hash_traffic = pmc_all.get("column_leaves_kernel", {}).get("hbm_bytes_per_launch")
    return {"shape": "LigeroParam(nw, 0, 4, 132, 2^%d), GF2_128<5>: %d rows, block %d, dblock %d, %d leaves" % (logn, rows, block, dblock, ext),
            "rs_roofline": {"bound": "hbm", "achieved": rs_bytes / rs_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rs_bytes / rs_ms / 1e6 / HBM_PEAK_GBS,
                            "traffic": rs_traffic, "traffic_over_algorithmic": (rs_traffic / rs_bytes) if rs_traffic else None,
                            "kernel": "bs_cin + bs_bfly2 / bs_range passes + bs_cout (tower representation)"},
            "hash_roofline": {"bound": "hbm", "achieved": hash_bytes / hash_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hash_bytes / hash_ms / 1e6 / HBM_PEAK_GBS,
                              "traffic": hash_traffic, "traffic_over_algorithmic": (hash_traffic / hash_bytes) if hash_traffic else None,
                              "kernel": "column_leaves_kernel (+ merkle levels)"},
            "rs_encode_ms": rs_ms, "rs_algo_GBps": rs_bytes / rs_ms / 1e6, "rs_frac_of_hbm": rs_bytes / rs_ms / 1e6 / HBM_PEAK_GBS,
            "column_commit_ms": hash_ms, "hash_algo_GBps": hash_bytes / hash_ms / 1e6, "hash_frac_of_hbm": hash_bytes / hash_ms / 1e6 / HBM_PEAK_GBS,
            "sha256_compressions": ext * ((32 + 16 * rows + 9 + 63) // 64) + ext - 1,
            "checked_vs_oracle": {"rs_rows_0_and_last": ok_rs, "leaves_64_sampled_columns": ok_leaves, "tree_from_device_leaves": ok_tree}}


def ligero_commit_sharded(pkg, gpu, torch, np, dist, rank, world):
    """LigeroProver::commit with the tableau rows sharded over the ranks (longfellow-zk_amd/parallel.py:
    host layout replayed from one RandomEngine stream, row-slab RS encode, all_to_all column re-partition over RCCL,
    local column hash, all_gather of the leaf digests, tree on every rank).  Statement: GF2_128<4>,
    LigeroParam(nw, 0, 4, 132, 2^15) with nrow = 1024 (block 5461, block_ext 21 847; 512 MiB tableau).  Reports the wall
    time of commit and whether every rank derived the same root."""
    import importlib
    import time as _t
    par = importlib.import_module("longfellow_zk_amd.parallel")
    be = 1 << 15
    p0 = pkg.ligero_param(pkg.FIELD_GF2_128, 1, 0, 4, 132, be)
    nw = p0.w * 1021
    p = pkg.ligero_param(pkg.FIELD_GF2_128, nw, 0, 4, 132, be)
    W = np.random.default_rng(7).integers(0, 2**63, size=(nw, 2), dtype=np.int64).view(np.uint64)
    comm = par.TorchComm(None, torch.device("cuda", torch.cuda.current_device()))
    rng_t = pkg.FsTranscript(b"bench-sharded-commit")
    times, roots = [], []
    for rep in range(3):
        pr = par.ShardedLigeroProver(gpu, pkg.FIELD_GF2_128, p, 4, comm=comm)  # lfgpu_ligero_commit_sharded behind the C ABI
        torch.cuda.synchronize()
        dist.barrier()
        t0 = _t.perf_counter()
        root = pr.commit(W, 0, [], rng_t.bytes)
        torch.cuda.synchronize()
        dist.barrier()
        times.append((_t.perf_counter() - t0) * 1e3)
        roots.append(root)
        pr.close()
    rng_t.close()
    mine = torch.tensor(list(roots[-1]), dtype=torch.uint8, device="cuda")
    if dist.get_backend() == "gloo":
        mine = mine.cpu()
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    same = all(bool((t == allr[0]).all()) for t in allr)
    return {"shape": "GF2_128<4>, %d rows x 2^15, block %d, %d leaves, rows sharded x%d" % (p.nrow, p.block, p.block_ext, world),
            "commit_wall_ms": min(times), "all_ranks_same_root": same,
            "includes": "host layout replay + upload of the slab + RS encode + all_to_all + column hash + all_gather + tree"}


PUBLISHED_M4_MS = {1: 5.30, 2: 9.60, 4: 18.73, 8: 35.39, 16: 65.62, 32: 125.23, 33: 132.71}  # docs/content/en/docs/benchmarks.md:55-61


def zk_throughput(device, jobs, ks, seconds, timeout=420):
    """Throughput mode (tools/zk_throughput.py): K concurrent provers on this rank's device -- K host threads, each with its own
    lfgpu context + stream, one copy of the circuit in HBM -- in a CHILD process (the HIP runtime reads GPU_MAX_HW_QUEUES when it
    initialises, and this process initialised it long ago).  Every worker first REQUIRES the reference's wire bytes."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "zk_throughput.py")
    r = subprocess.run([sys.executable, tool, "--device", str(device), "--jobs", ",".join(jobs), "--k", ",".join(str(k) for k in ks),
                        "--seconds", str(seconds)], capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-400:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def zk_prove_flatsha(pkg, gpu, np, nb, with_cpu, reps=3):
    """BASELINE's first metric, "flatsha256 fp2_128 prove ms" (BM_ShaZK_fp2_128, lib/circuits/sha/
    flatsha256_circuit_test.cc:510-536: ZkProver commit + prove, rate 7, 132 queries): the library's C++ ZK driver
    (include/lfgpu_zk.h) on the committed fixture circuit + witness (tests/golden, made by the real reference).  The wire
    bytes are first checked against the reference's (LCG RandomEngine of the fixture); the timed repetitions use a
    C-speed engine, as the reference benchmark uses SecureRandomEngine.  CPU figure: the reference prover itself
    (oracle/_ref/gen_flatsha, single thread) on this host."""
    import hashlib
    import lzma
    import subprocess
    import tempfile
    import ligero_fixture as lf
    gold = os.path.join(ROOT, "tests", "golden")
    raw = lzma.decompress(open(os.path.join(gold, "flatsha_nb%d.lfc1.xz" % nb), "rb").read())
    W = np.frombuffer(lzma.decompress(open(os.path.join(gold, "flatsha_nb%d.w.xz" % nb), "rb").read()), dtype=np.uint64).reshape(-1, 2).copy()
    info = json.load(open(os.path.join(gold, "flatsha_nb%d.json" % nb)))
    circ = pkg.Circuit(gpu, raw)
    zk = pkg.ZkProver(gpu, circ, 7, 132)
    ts = pkg.FsTranscript(b"test")
    zk.commit(W, lf.LcgRng(100).bytes, ts)
    ok = zk.prove(W, ts)
    wire = zk.wire() if ok else b""
    ts.close()
    identical = ok and len(wire) == info["zk_wire_bytes"] and hashlib.sha256(wire).hexdigest() == info["zk_wire_sha256"]
    L = gpu.L
    rng_t = pkg.FsTranscript(b"rng")
    rng_fn = C.cast(L.lfgpu_transcript_bytes, pkg.RNG_FN)
    Wp, root, okc = C.c_void_p(W.ctypes.data), (C.c_uint8 * 32)(), C.c_int()
    best = None
    for _ in range(reps):
        ts = pkg.FsTranscript(b"test")
        ops = ts.ops()
        t0 = time.perf_counter()
        gpu._ck(L.lfgpu_zk_commit(zk.h, Wp, rng_fn, rng_t.h, C.byref(ops), root))
        t1 = time.perf_counter()
        gpu._ck(L.lfgpu_zk_prove(zk.h, Wp, C.byref(ops), C.byref(okc)))
        t2 = time.perf_counter()
        ts.close()
        cur = {"commit_ms": (t1 - t0) * 1e3, "prove_ms": (t2 - t1) * 1e3, "total_ms": (t2 - t0) * 1e3}
        if best is None or cur["total_ms"] < best["total_ms"]:
            best = dict(cur, phases_ms=zk.timings())
    res = {"sha_blocks": nb, "nterms": info["nterms"], "wire_bytes_identical_to_reference": bool(identical)}
    res.update(best)
    # the verifier (lfgpu_zk_verify = ZkVerifier::recv_commitment + verify) on that proof
    wire2, vbest = zk.wire(), None
    for _ in range(reps):
        ts = pkg.FsTranscript(b"test")
        t0 = time.perf_counter()
        okv, _why = pkg.zk_verify(gpu, circ, wire2, W[:circ.info.npub_in], ts)
        dt = (time.perf_counter() - t0) * 1e3
        ts.close()
        vbest = dt if vbest is None or dt < vbest else vbest
    res["verify_ms"] = vbest
    res["verify_accepts"] = bool(okv)
    res["published_mac_m4_total_ms"] = PUBLISHED_M4_MS.get(nb)  # reference docs/content/en/docs/benchmarks.md:55-61 (BM_ShaZK_fp2_128/nb)
    gen = os.path.join(ROOT, "oracle", "_ref", "gen_flatsha")
    if with_cpu and os.path.exists(gen):
        with tempfile.TemporaryDirectory() as td:
            r = json.loads(subprocess.check_output([gen, str(nb), os.path.join(td, "x")]).decode())
        res["cpu_reference"] = {"commit_ms": r["ref_zk_commit_ms"], "prove_ms": r["ref_zk_prove_ms"],
                                "total_ms": r["ref_zk_commit_ms"] + r["ref_zk_prove_ms"], "verify_ms": r.get("ref_zk_verify_ms"),
                                "cores": 1, "kind": "reference"}
    rng_t.close()
    zk.close()
    circ.close()
    return res


def zk_prove_mdoc(pkg, gpu, np, reps=3, end_to_end=True):
    """BASELINE config 5 ("End-to-end MDOC/ECDSA prove"): the two real mdoc circuits (kZkSpecs[0]) with the witnesses of a real
    proof (tests/golden/mdoc_*, made by the reference: oracle/ref_mdoc.cc) -- hash circuit over GF2_128 (7.76 M terms), signature
    circuit over Fp256Base (32-byte elements) -- each committed, proved and verified stand-alone by the library; wire bytes
    checked against the reference's first.  The CPU figures are the reference's own provers, measured when the fixture was made
    (one thread of the build container).  The whole run_mdoc_prover / run_mdoc_verifier bodies with the library's provers and
    verifiers in the reference's place: oracle/ref_mdoc_gpu.cc (tests/test_reference_integration.py)."""
    import hashlib
    import lzma
    import ligero_fixture as lf
    gold = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gold, "mdoc.json")))
    rate, nreq = meta["hash"]["rate"], meta["hash"]["nreq"]
    L = gpu.L
    rng_t = pkg.FsTranscript(b"rng")
    rng_fn = C.cast(L.lfgpu_transcript_bytes, pkg.RNG_FN)
    res = {}
    for half, stem, words in (("hash", "mdoc_hash", 2), ("sig", "mdoc_sig", 4)):
        info = meta[half]
        raw = lzma.decompress(open(os.path.join(gold, stem + ".lfc1.xz"), "rb").read())
        W = np.frombuffer(lzma.decompress(open(os.path.join(gold, stem + ".w.xz"), "rb").read()), dtype=np.uint64).reshape(-1, words).copy()
        t0 = time.perf_counter()
        circ = pkg.Circuit(gpu, raw)
        t_up = (time.perf_counter() - t0) * 1e3
        del raw
        zk = pkg.ZkProver(gpu, circ, rate, nreq, info["block_enc"])
        ts = pkg.FsTranscript(b"test")
        zk.commit(W, lf.LcgRng(100).bytes, ts)
        ok = zk.prove(W, ts)
        wire = zk.wire() if ok else b""
        ts.close()
        identical = ok and len(wire) == info["zk_wire_bytes"] and hashlib.sha256(wire).hexdigest() == info["zk_wire_sha256"]
        Wp, root, okc = C.c_void_p(W.ctypes.data), (C.c_uint8 * 32)(), C.c_int()
        best = None
        for _ in range(reps):
            ts = pkg.FsTranscript(b"test")
            ops = ts.ops()
            t0 = time.perf_counter()
            gpu._ck(L.lfgpu_zk_commit(zk.h, Wp, rng_fn, rng_t.h, C.byref(ops), root))
            t1 = time.perf_counter()
            gpu._ck(L.lfgpu_zk_prove(zk.h, Wp, C.byref(ops), C.byref(okc)))
            t2 = time.perf_counter()
            ts.close()
            cur = {"commit_ms": (t1 - t0) * 1e3, "prove_ms": (t2 - t1) * 1e3, "total_ms": (t2 - t0) * 1e3}
            if best is None or cur["total_ms"] < best["total_ms"]:
                best = cur
        wire2, vbest, okv = zk.wire(), None, False
        for _ in range(reps):
            ts = pkg.FsTranscript(b"test")
            t0 = time.perf_counter()
            okv, _why = pkg.zk_verify(gpu, circ, wire2, W[:circ.info.npub_in], ts, rate, nreq, info["block_enc"])
            dt = (time.perf_counter() - t0) * 1e3
            ts.close()
            vbest = dt if vbest is None or dt < vbest else vbest
        res[half] = dict(best, field="GF2_128" if half == "hash" else "Fp256Base", nterms=info["nterms"], layers=info["nl"],
                         wire_bytes_identical_to_reference=bool(identical), verify_ms=vbest, verify_accepts=bool(okv), circuit_parse_upload_ms=t_up,
                         cpu_reference={"commit_ms": info["ref_commit_ms"], "prove_ms": info["ref_prove_ms"],
                                        "total_ms": info["ref_commit_ms"] + info["ref_prove_ms"], "cores": 1, "kind": "reference",
                                        "host": "build container, NOT this host",
                                        "note": "cross-host: measured in the build container when the fixture was made; the same-host comparison is end_to_end below"})
        zk.close()
        circ.close()
    rng_t.close()
    # the whole run_mdoc_prover / run_mdoc_verifier bodies with the library's provers / verifiers in the reference's place, next
    # to the reference's own on this host (oracle/_ref/mdoc_gpu: built in the build container, travels with the snapshot)
    exe = os.path.join(ROOT, "oracle", "_ref", "mdoc_gpu")
    if end_to_end and os.path.exists(exe):
        import subprocess
        try:
            r = subprocess.run([exe, "3"], capture_output=True, timeout=240)
            e2e = json.loads(r.stdout.decode().strip().splitlines()[-1])
            res["end_to_end"] = {"proof_bytes_identical_to_reference": e2e["identical"], "reference_verifier_accepts": e2e["reference_verifier_accepts_gpu_proof"],
                                 "prove_ms": e2e["gpu_ms"], "cpu_reference_prove_ms": dict(e2e["ref_ms"], cores=1, kind="reference"),
                                 "verify_ms": e2e["verify"]["gpu_ms"], "cpu_reference_verify_ms": e2e["verify"]["ref_ms"],
                                 "circuit_parse_upload_ms": e2e["host_ms"]["gpu_parse_upload"], "cpu_reference_parse_ms": e2e["host_ms"]["reference_parse"]}
        except Exception as e:  # noqa: BLE001
            res["end_to_end"] = {"error": repr(e)[:200]}
    res["total_ms"] = res["hash"]["total_ms"] + res["sig"]["total_ms"]
    res["cpu_reference_total_ms"] = res["hash"]["cpu_reference"]["total_ms"] + res["sig"]["cpu_reference"]["total_ms"]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: rehearses the N > 1 code path on a 1-GPU box")
    ap.add_argument("--rehearse-gloo", action="store_true", help="rehearsal of the N > 1 control flow on a ONE-GPU box: backend gloo (host-staged), every rank on device 0 (RCCL cannot put two ranks on one device); not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare with --gpus N: become the launcher.  Nothing in this process has touched the GPU (torch is not even
        # imported yet); the N ranks are children of torch.distributed.run, one per GPU, and their single JSON line
        # (printed by rank 0) passes through.
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    # libraries chat on stdout (RCCL prints a version banner at init): keep fd 1 for the ONE JSON line, send the rest to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("bench: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d (the ranks that exist)" % (args.gpus, world, world), file=sys.stderr)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            local_rank = 0
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    def allreduce(t, op):  # gloo rehearsal: through the host
        if args.rehearse_gloo:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    from __graft_entry__ import load_package
    pkg = load_package()
    gpu = pkg.LfGpu(local_rank)  # raises if liblfgpu.so is missing: no CPU fallback
    stream = torch.cuda.current_stream()
    gpu.set_stream(stream.cuda_stream)

    rows, logn = args.rows, args.logn
    n = 1 << logn
    nelem = rows * n
    # synthetic witness rows: uniform 128-bit values < p are valid Montgomery images
    g = torch.Generator(device="cuda")
    g.manual_seed(1234569 + rank)
    A = torch.randint(-2**63, 2**63 - 1, (nelem, 2), dtype=torch.int64, device="cuda", generator=g)
    A[:, 1] &= 0x7FFFFFFFFFFFFFFF  # hi limb < 2^63 < p_hi

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the CPU baseline first (rank 0; bounded to ~10 s), before any GPU leg has host threads spinning; for N > 1 as well
    cpu_base = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(logn)
    barrier()

    # ---- correctness guard (untimed): row 0 of one fftb against the oracle
    import oracle_lib as ol
    o = ol.oracle()
    row0 = A[:n].cpu().numpy().view(np.uint64).copy()
    gpu.fp128_fft(A.data_ptr(), rows, n)
    torch.cuda.synchronize()
    got0 = A[:n].cpu().numpy().view(np.uint64)
    o.lfo_fp_fftb(ol.P(row0), n, o.lfo_fp_omega32(), 1 << 32)
    if not (got0 == row0).all():
        raise SystemExit("bench: GPU fftb row 0 differs from the oracle -- refusing to report a number")

    for _ in range(args.warmup):
        gpu.fp128_fft(A.data_ptr(), rows, n)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        gpu.fp128_fft(A.data_ptr(), rows, n)
    ev1.record(stream)
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        allreduce(tt, dist.ReduceOp.MAX)
        dt = float(tt.item())

    launches_per_step = 1 if logn <= 12 else 2  # fp_fft_tile passes (fft.hip: tiles of 2^12 elements)
    # HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE with the gfx950 correction + WRITE_SIZE), collected
    # with the same command and committed under profiles/ -- counters cannot be read in-process
    traffic = None
    pmc_all = {}
    pmc = os.path.join(ROOT, "profiles", "r03", "pmc_traffic.json")  # this round's counters (tools/pmc_traffic.sh -> tools/pmc_summary.py)
    if os.path.exists(pmc):
        with open(pmc) as f:
            pmc_all = json.load(f)
    if rows == 1024 and logn == 20 and "fp_fft_tile" in pmc_all:
        traffic = pmc_all["fp_fft_tile"]["hbm_bytes_per_launch"]
    kern_ms = dev_ms / (args.steps * launches_per_step)
    algo_bytes_per_launch = 2.0 * nelem * 16 / launches_per_step
    achieved = algo_bytes_per_launch / (kern_ms * 1e-3) / 1e9

    out = {
        "metric": "FFT field-elems/s (batched FFT 2^%d x %d rows, Fp128)" % (logn, rows),
        "value": world * nelem * args.steps / dt,
        "unit": "field-elems/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u128 (Fp128 Montgomery, 4 x u32 limbs)",
        "data": "synthetic",
        "config": {"workload": "batched FFT 2^%d points x %d rows per GPU, Fp128 fftb, in place in HBM" % (logn, rows),
                   "field": "Fp128 p=2^128-2^108+1", "rows_per_gpu": rows, "n": n, "parallelism": "rows sharded x%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "fp_fft_tile", "launches_per_step": launches_per_step, "avg_launch_ms": kern_ms,
                     "note": "integer-ALU-bound (no 64-bit multiplier on CDNA4): see DESIGN.md"},
    }
    if logn == 20:
        # the roofline that actually binds this kernel: VALU instruction issue.  Static instruction counts of the
        # hand-written arithmetic (csrc/fields.h): 71 per Montgomery product (one-step REDC), 27 per add+sub pair; per element and
        # transform: 9 products (4 + 4 butterfly products of the two 1024-point passes, skipping w^0, + 1 inter-pass
        # twiddle) and 10 butterfly add/sub pairs.  Peak: 36 T lane-instr/s, the measured rate of the slow instruction class
        # (carry adds, v_mad_u64_u32) that this kernel consists of (tools/ubench.hip, profiles/r02/ubench_int_rates.txt).  The
        # counter-based figure (SQ_INSTS_VALU x 4 cycles / SIMD cycles = 96 %) is in profiles/r02/pmc_fp_fft_tile_redc1.json.
        instr = nelem * (9 * 71 + 10 * 27)
        rate = instr / (dev_ms / args.steps * 1e-3) / 1e12
        out["alu_roofline"] = {"bound": "valu", "achieved": rate, "peak": 36.0, "unit": "T lane-instr/s", "frac": rate / 36.0,
                               "basis": "static instruction counts x live kernel time"}

    if rank == 0 and not args.no_secondary:
        # secondary: same batch through the GF(2^128) LCH14 additive FFT (GF2_128<5>, l = logn)
        k = 5 if logn > 16 else 4
        for _ in range(1):
            gpu.gf2128_lch14_fft(A.data_ptr(), rows, logn, subfield_log_bits=k)
        torch.cuda.synchronize()
        s2 = max(1, args.steps // 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(s2):
            gpu.gf2128_lch14_fft(A.data_ptr(), rows, logn, subfield_log_bits=k)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / s2
        agb = 2.0 * nelem * 16 / (ms * 1e-3) / 1e9
        out["gf2128_lch14_fft"] = {"field_elems_per_s": nelem / (ms * 1e-3), "ms_per_step": ms, "algo_GBps": agb,
                                   "roofline": {"bound": "hbm", "achieved": agb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agb / HBM_PEAK_GBS,
                                                "note": "whole transform = bs_cin + bs_bfly passes + bs_cout; per-kernel HBM bytes (PMC) in profiles/r02/pmc_bs_kernels.json"}}
    if rank == 0 and not args.no_secondary:
        # secondary: the same plan and kernel over F64_2 = Fp2<Fp<1>>, p = 2^64 - 2^32 + 1 (BM_FFT_F64_2, lib/algebra/
        # fft_test.cc:205-229; docs/content/en/docs/benchmarks.md:37 publishes 66.65 ms for ONE 2^20-point row)
        v = A.view(torch.int64)
        v[(v < 0) & (v > -(1 << 32))] = 0  # words >= p are not field elements
        del v
        gpu.f64_2_fft(A.data_ptr(), rows, 1 << logn)
        torch.cuda.synchronize()
        s2 = max(1, args.steps // 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(s2):
            gpu.f64_2_fft(A.data_ptr(), rows, 1 << logn)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / s2
        agb = 2.0 * nelem * 16 / (ms * 1e-3) / 1e9
        out["f64_2_fft"] = {"field_elems_per_s": nelem / (ms * 1e-3), "ms_per_step": ms, "ms_per_row": ms / rows,
                            "published_cpu_ms_per_row_2pow20": 66.65,
                            "roofline": {"bound": "hbm", "achieved": agb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agb / HBM_PEAK_GBS,
                                         "note": "same two-pass plan and kernel (fp_fft_tile) as the headline, cheaper field"}}
    if rank == 0 and not args.no_secondary and logn == 20 and rows == 1024:
        out["ligero_commit_slig"] = ligero_commit_slig(gpu, torch, np, A, rows, logn, pmc_all)
    if rank == 0 and not args.no_secondary:
        out["ligero_commit_flatsha32"] = ligero_commit_shape(gpu, torch, np, stream, not args.no_cpu_baseline)
    del A  # the ZK path allocates its own buffers
    torch.cuda.empty_cache()
    if not args.no_secondary:
        # ---- BASELINE's first metric on EVERY rank (independent proofs are the N-GPU mode of the real circuits: their tableaux are
        # 2.5 - 20 MB): latency of one proof per rank, then the throughput mode, summed over the ranks below
        mine = {}
        try:
            if rank == 0:  # every block count the reference publishes (benchmarks.md:55-61); the other ranks: the 32-block one
                by_nb = {}
                for nb in (1, 2, 4, 8, 16, 32, 33):
                    by_nb[str(nb)] = zk_prove_flatsha(pkg, gpu, np, nb, not args.no_cpu_baseline)
                out["zk_prove_flatsha256"] = dict(by_nb["32"], by_sha_blocks={k: {kk: v.get(kk) for kk in ("total_ms", "commit_ms", "prove_ms", "verify_ms", "wire_bytes_identical_to_reference", "published_mac_m4_total_ms", "cpu_reference")} for k, v in by_nb.items()})
                mine["flatsha32_ms"] = by_nb["32"]["total_ms"]
            else:
                mine["flatsha32_ms"] = zk_prove_flatsha(pkg, gpu, np, 32, False)["total_ms"]
        except Exception as e:  # noqa: BLE001 -- a failure here must not cost the headline line
            if rank == 0:
                out["zk_prove_flatsha256"] = {"error": repr(e)[:300]}
        try:  # BASELINE config 5
            md = zk_prove_mdoc(pkg, gpu, np, end_to_end=(rank == 0))
            mine["mdoc_ms"] = md["total_ms"]
            if rank == 0:
                out["zk_prove_mdoc"] = md
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                out["zk_prove_mdoc"] = {"error": repr(e)[:300]}
        barrier()
        # throughput: rank 0 sweeps K on one GPU at N = 1; with N > 1 every rank runs K = 16 (or 8) at the same time (replicas)
        # (K per GPU at N > 1: 16 when this rank's share of the host cores has room for 16 polling threads beside the rank itself -- a
        # prover is one host thread answering its resident kernel, DESIGN.md 4.9 -- else 8)
        try:
            cores_here = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cores_here = os.cpu_count() or 8
        ks = [1, 2, 4, 8, 16] if world == 1 else ([16] if cores_here // world >= 18 else [8])
        try:  # whatever happens here, every rank reaches the collectives below
            thr = zk_throughput(local_rank, ["flatsha32", "mdoc"], ks, 2.0)
        except Exception as e:  # noqa: BLE001
            thr = {"error": repr(e)[:300]}
        agg = {}
        for job in ("flatsha32", "mdoc"):
            best = max((v["proofs_per_s"] for v in thr.get(job, {}).get("k", {}).values()), default=0.0)
            agg[job] = float(best)
        if dist is not None:
            tt = torch.tensor([agg["flatsha32"], agg["mdoc"], mine.get("flatsha32_ms", 0.0), mine.get("mdoc_ms", 0.0)], dtype=torch.float64, device="cuda")
            mx = tt.clone()
            allreduce(tt, dist.ReduceOp.SUM)
            allreduce(mx, dist.ReduceOp.MAX)
            agg = {"flatsha32": float(tt[0]), "mdoc": float(tt[1])}
            lat = {"flatsha32_ms_max_over_ranks": float(mx[2]), "mdoc_ms_max_over_ranks": float(mx[3])}
        else:
            lat = {"flatsha32_ms_max_over_ranks": mine.get("flatsha32_ms"), "mdoc_ms_max_over_ranks": mine.get("mdoc_ms")}
        if rank == 0:
            out["zk_throughput"] = {
                "what": "independent proofs (replicas): K concurrent provers per GPU (host threads, own context + stream each, one copy of the circuit; K = %s), summed over %d GPU(s); every worker first reproduces the reference's wire bytes" % ("/".join(str(k) for k in ks), world),
                "proofs_per_s": {"flatsha256_32_blocks": agg["flatsha32"], "mdoc_hash_plus_signature": agg["mdoc"]},
                "single_proof_latency": lat,
                "rank0_sweep": thr,
                "cpu_reference_proofs_per_s_1_thread": {"flatsha256_32_blocks": (1e3 / out["zk_prove_flatsha256"]["cpu_reference"]["total_ms"]) if isinstance(out.get("zk_prove_flatsha256"), dict) and out["zk_prove_flatsha256"].get("cpu_reference") else None}}
    watchdog = None
    if dist is not None:
        # the legs below are collectives that have never run on more than one GPU in the build container: should one hang, the
        # line measured so far still goes out (rank 0) and every rank leaves, instead of the whole run being killed without a line
        import threading

        def _bail():
            if rank == 0:
                out["multi_rank_legs"] = {"error": "a collective leg did not finish within 240 s; the line above it is complete"}
                if cpu_base is not None:
                    out["cpu_baseline"] = cpu_base
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(0)

        watchdog = threading.Timer(240.0, _bail)
        watchdog.daemon = True
        watchdog.start()
    if dist is not None:
        # RCCL small-message latency (what a sharded sumcheck round-hand would pay per collective: all_gather of one (a0, a2)
        # pair = 32 bytes per rank; DESIGN.md section 6 weighs it against the 15 - 35 us round-hand)
        try:
            if args.rehearse_gloo:
                raise RuntimeError("gloo rehearsal: no RCCL latency to measure")
            t32 = torch.zeros(4, dtype=torch.int64, device="cuda")
            g32 = torch.zeros(4 * world, dtype=torch.int64, device="cuda")
            for _ in range(20):
                dist.all_gather_into_tensor(g32, t32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                dist.all_gather_into_tensor(g32, t32)
                torch.cuda.synchronize()  # a round-hand needs the result on the host before it can go on
            lat_us = (time.perf_counter() - t0) / 200 * 1e6
            if rank == 0:
                out["rccl_small_message"] = {"all_gather_32B_per_rank_us": lat_us, "ranks": world, "includes": "launch + collective + stream synchronisation, as a host-driven round-hand would see it"}
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                out["rccl_small_message"] = {"error": repr(e)[:200]}
    if dist is not None and not args.no_secondary:
        try:  # the sharded Ligero commit over RCCL (every rank takes part); a failure here must not cost the headline line
            sh = ligero_commit_sharded(pkg, gpu, torch, np, dist, rank, world)
        except Exception as e:  # noqa: BLE001
            sh = {"error": repr(e)[:300]}
        if rank == 0:
            out["ligero_commit_sharded"] = sh
    if dist is not None:
        dist.barrier()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    gpu.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
