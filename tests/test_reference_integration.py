"""The integration of INTEGRATION.md section 2a, executed: the reference's OWN ZkProver / LigeroProver / sumcheck prover /
transcript / ZkProof::write, compiled from /root/reference in the build container with one template argument swapped --
InterpolatorFactory = lfgpu::GpuReedSolomonFactory<Field> (include/lfgpu_adapters.h) -- and linked against liblfgpu.so
(oracle/ref_zk_adapters.cc -> oracle/_ref/zk_adapters[_fp|_p256], built by `make -C oracle ref`).  On the GPU box the binary
proves the fixture circuits with every Reed-Solomon row extension running in the HIP kernels; its wire bytes must hash to
what the unmodified reference produced (tests/golden/flatsha_*.json)."""
import json
import lzma
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.gpu
@pytest.mark.parametrize("stem,binary", [("flatsha_nb1", "zk_adapters"), ("flatsha_nb32", "zk_adapters"), ("flatsha_nb33", "zk_adapters"),
                                         ("flatsha_fp_nb1", "zk_adapters_fp")])
def test_reference_zkprover_with_gpu_interpolator_emits_reference_wire_bytes(stem, binary):
    exe = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s not built (needs the reference sources: make -C oracle ref in the build container)" % binary)
    info = json.load(open(os.path.join(GOLD, stem + ".json")))
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for ext in (".lfc1", ".w"):
            p = os.path.join(td, "x" + ext)
            with open(p, "wb") as f:
                f.write(lzma.decompress(open(os.path.join(GOLD, stem + ext + ".xz"), "rb").read()))
            paths.append(p)
        out = subprocess.run([exe] + paths, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    res = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert res["wire_bytes"] == info["zk_wire_bytes"]
    assert res["wire_sha256"] == info["zk_wire_sha256"]
    assert (res["block_enc"], res["nrow"]) == (info["zk_block_enc"], info["zk_nrow"])


@pytest.mark.gpu
def test_reference_zkprover_p256_signature_circuit_with_gpu_interpolator():
    """BASELINE config 5, the signature half: the reference's ZkProver<Fp256Base, .> on the REAL mdoc signature circuit
    (kZkSpecs[0]: 21 layers, 481 833 terms, 32-byte elements, block_enc 4096, 19 rows) and the witness of a real proof
    (oracle/ref_mdoc.cc), with lfgpu::GpuReedSolomonFactory<Fp256Base> in place of
    ReedSolomonFactory<Fp256Base, FFTExtConvolutionFactory> (mdoc_zk.cc:75-76): every row extension of commit, dot_proof and
    quadratic_proof runs in csrc/p256.hip.  The wire bytes must hash to what the unmodified reference produced; the timings
    the binary prints go beside the reference's own (tests/golden/mdoc.json: 206 ms commit + 507 ms prove on one CPU
    thread, most of it in the RFFT twiddle rebuilds of SURVEY section 8 f2)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "zk_adapters_p256")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/zk_adapters_p256 not built (needs the reference sources: make -C oracle ref in the build container)")
    info = json.load(open(os.path.join(GOLD, "mdoc.json")))["sig"]
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for ext in (".lfc1", ".w"):
            p = os.path.join(td, "x" + ext)
            with open(p, "wb") as f:
                f.write(lzma.decompress(open(os.path.join(GOLD, "mdoc_sig" + ext + ".xz"), "rb").read()))
            paths.append(p)
        out = subprocess.run([exe] + paths + [str(info["block_enc"])], capture_output=True, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    res = json.loads(out.stdout.decode().strip().splitlines()[-1])
    print("mdoc signature circuit, reference prover + GPU Reed-Solomon:", res, "reference alone:", info["ref_commit_ms"], info["ref_prove_ms"])
    assert res["wire_bytes"] == info["zk_wire_bytes"]
    assert res["wire_sha256"] == info["zk_wire_sha256"]
    assert (res["block_enc"], res["nrow"]) == (info["block_enc"], info["nrow"])


@pytest.mark.gpu
@pytest.mark.parametrize("which,reps", [(0, 3), (1, 1), (2, 1)], ids=["age_over_18", "other_document_text_attribute", "two_attribute_circuits"])
def test_mdoc_end_to_end_with_gpu_provers_in_run_mdoc_prover(which, reps):
    """BASELINE config 5 end to end: the body of the reference's run_mdoc_prover (lib/circuits/mdoc/mdoc_zk.cc:398-546:
    circuit generation and parsing, CBOR witness filling, the shared transcript, MAC key and update_macs between commit and
    prove, ZkProof::write of both proofs) with lfgpu::GpuZkProver (include/lfgpu_zk_adapters.h) in the place of
    ZkProver<f_128, .> AND ZkProver<Fp256Base, .> -- every commit and prove on the device -- against the same body with the
    reference's own provers, same witness, same deterministic RandomEngine (oracle/ref_mdoc_gpu.cc).  The two mdoc proof
    strings must be byte-identical and the reference's run_mdoc_verifier must accept the library's; then the verifier's body
    with lfgpu::GpuZkVerifier for both circuits must accept that proof and reject it with one bit flipped.

    Cases are the reference's own examples (lib/circuits/mdoc/mdoc_zk_test.cc:118-240): kZkSpecs[0] on mdoc_tests[0] with
    age_over_18 (the BASELINE configuration), kZkSpecs[0] on mdoc_tests[3] with the text attribute family_name, and
    kZkSpecs[1] -- the two-attribute pair of circuits, different circuits from the first two cases -- on mdoc_tests[3]."""
    exe = os.path.join(ROOT, "oracle", "_ref", "mdoc_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/mdoc_gpu not built (needs the reference sources and zstd.h: make -C oracle ref in the build container)")
    out = subprocess.run([exe, str(reps), "--ref", str(which)], capture_output=True, timeout=900)
    assert out.returncode == 0, (out.returncode, out.stdout.decode()[-1000:], out.stderr.decode()[-2000:])
    res = json.loads(out.stdout.decode().strip().splitlines()[-1])
    print("mdoc end to end:", res)
    assert res["case"] == which and res["attributes"] == (2 if which == 2 else 1)
    assert res["identical"] is True and res["gpu_sha256"] == res["ref_sha256"]
    assert res["reference_verifier_accepts_gpu_proof"] is True
    # the body of run_mdoc_verifier (mdoc_zk.cc:548-712) with lfgpu::GpuZkVerifier in the place of both ZkVerifiers
    v = res["verify"]
    assert v["gpu_verifiers_accept"] is True and v["gpu_verifiers_reject_flipped_bit"] is True and v["reference_verifiers_accept"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [0, 1, 2, 3], ids=["v7_one_attribute", "v7_two_attributes", "v7_four_attributes", "v6_one_attribute_shipped_circuit"])
def test_reference_stored_mdoc_proofs_reproduced_and_verified(tmp_path, idx):
    """The reference's own STORED mdoc proofs (rust/applications/mdoc_zk/artifacts/proofs/<circuit hash>.bin with both witness files,
    the fixtures of its prior_zk.rs test; committed under tests/golden/ by oracle/gen_mdoc_artifact_fixtures.py): complete proofs of
    BASELINE config 5 -- MACs, hash circuit over GF2_128, signature circuit over Fp256Base, one shared transcript.  From the stored
    input vectors and the reference's DeterministicRng(42), lfgpu::GpuZkProver inside run_mdoc_prover's body must write exactly the
    stored bytes (the reference's C++ provers do: checked in the build container, oracle/ref_mdoc_gpu.cc `stored ... --with-ref`),
    and lfgpu::GpuZkVerifier inside run_mdoc_verifier's body must accept the stored bytes and reject them with one bit flipped."""
    exe = os.path.join(ROOT, "oracle", "_ref", "mdoc_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/mdoc_gpu not built (needs the reference sources and zstd.h: make -C oracle ref in the build container)")
    spec = json.load(open(os.path.join(GOLD, "mdoc_artifacts.json")))["specs"][idx]
    stem = os.path.join(GOLD, spec["stem"])
    hw, sw = tmp_path / "hash_w.bin", tmp_path / "sig_w.bin"
    hw.write_bytes(lzma.decompress(open(stem + ".hash_witness.xz", "rb").read()))
    sw.write_bytes(lzma.decompress(open(stem + ".sig_witness.xz", "rb").read()))
    extra = ["--circuit", stem + ".circuit.zst"] if spec["circuit_file_bytes"] else []  # version 6: the circuit pair as the reference ships it
    out = subprocess.run([exe, "stored", str(spec["zk_spec_index"]), stem + ".proof.bin", str(hw), str(sw)] + extra, capture_output=True, timeout=900)
    assert out.returncode == 0, (out.returncode, out.stdout.decode()[-1000:], out.stderr.decode()[-2000:])
    res = json.loads(out.stdout.decode().strip().splitlines()[-1])
    print("stored mdoc artifact:", res)
    assert res["stored_artifact"] == spec["circuit_hash"] and res["stored_sha256"] == spec["proof_sha256"] and res["stored_bytes"] == spec["proof_bytes"]
    assert res["gpu_proof_identical_to_stored"] is True
    assert res["gpu_verifiers_accept_stored"] is True and res["gpu_verifiers_reject_flipped_bit"] is True


def test_reference_stored_mdoc_fixtures_are_the_reference_files():
    """the committed fixtures are byte copies of the reference's artifact files (manifest hashes; the witnesses decompress to them)"""
    import hashlib
    man = json.load(open(os.path.join(GOLD, "mdoc_artifacts.json")))
    for spec in man["specs"]:
        stem = os.path.join(GOLD, spec["stem"])
        proof = open(stem + ".proof.bin", "rb").read()
        assert len(proof) == spec["proof_bytes"] and hashlib.sha256(proof).hexdigest() == spec["proof_sha256"]
        hw = lzma.decompress(open(stem + ".hash_witness.xz", "rb").read())
        sw = lzma.decompress(open(stem + ".sig_witness.xz", "rb").read())
        assert len(hw) == 16 * spec["hash_witness_elements"] and hashlib.sha256(hw).hexdigest() == spec["hash_witness_sha256"]
        assert len(sw) == 32 * spec["sig_witness_elements"] and hashlib.sha256(sw).hexdigest() == spec["sig_witness_sha256"]
        src = os.path.join("/root/reference/rust/applications/mdoc_zk/artifacts/proofs", spec["circuit_hash"])
        if spec["circuit_file_bytes"]:
            circ = open(stem + ".circuit.zst", "rb").read()
            assert len(circ) == spec["circuit_file_bytes"] and hashlib.sha256(circ).hexdigest() == spec["circuit_file_sha256"]
        if os.path.exists(src + ".bin"):  # the build container: compare with the reference's files where they lie
            assert open(src + ".bin", "rb").read() == proof
            assert open(src + "_hash_witness.bin", "rb").read() == hw and open(src + "_sig_witness.bin", "rb").read() == sw


def test_reference_cpp_provers_reproduce_a_stored_mdoc_proof():
    """CPU, build container only (reads the reference's artifact files where they lie): the reference's own C++ provers inside
    run_mdoc_prover's flow write exactly the stored proof string -- which is what makes the stored files known-answer vectors
    (oracle/ref_mdoc_stored_cpu.cc; all twelve specs reproduce, one is run here: version 6, circuit from the artifact file)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "mdoc_stored_cpu")
    if not os.path.exists(exe) or not os.path.isdir("/root/reference/rust/applications/mdoc_zk/artifacts/proofs"):
        pytest.skip("needs oracle/_ref/mdoc_stored_cpu and the reference tree (build container)")
    out = subprocess.run([exe, "4"], capture_output=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout.decode()[-500:], out.stderr.decode()[-1000:])
    res = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert res["identical"] is True and res["public_inputs_differing_from_stored"] == 0 and res["stored_bytes"] == 323932


@pytest.mark.gpu
def test_remaining_adapters_executed_against_the_reference_classes():
    """lfgpu::GpuFFT, GpuLCH14, GpuMerkleCommitment (commit + open) and GpuSumcheckRound (partials, Dense::bind, HQuad::bind_h)
    run next to FFT<Fp128>, LCH14<GF2_128<4>>, MerkleCommitment with LigeroCommon::column_hash, and ProverLayers::evaluations /
    Dense::bind / HQuad::bind_h of the reference, in one program (oracle/ref_adapters_exec.cc), on Bogorng inputs: bit-exact."""
    exe = os.path.join(ROOT, "oracle", "_ref", "adapters_exec")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/adapters_exec not built (needs the reference sources: make -C oracle ref in the build container)")
    out = subprocess.run([exe], capture_output=True, timeout=600)
    res = json.loads(out.stdout.decode().strip().splitlines()[-1]) if out.stdout.strip() else {}
    assert out.returncode == 0 and res.get("all_ok") is True, (res, out.stderr.decode()[-2000:])
    for k in ("fft", "lch14", "merkle_commitment", "sumcheck_round_gf2128", "sumcheck_round_fp128"):
        assert res[k][0] == res[k][1] and res[k][1] > 0, (k, res)
