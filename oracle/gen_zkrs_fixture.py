"""Extracts the one complete ZK known-answer vector the reference itself holds -- test_zk_rfc_testvector1 of
rust/runtime/zk/tests/zk.rs:230-345 (bytes produced by the C++ ZkProver: circuit, expected sumcheck proof, commitment and Ligero
proof) -- into a data fixture, tests/golden/zk_rs_testvector1.json.  Only the byte arrays and the test's parameters are taken
(data, as with the .bin fixtures); nothing of the Rust code.  Also copies rust/runtime/random/tests/transcript_test_vector.bin
(25 600 bytes of C++ Transcript output; layout = rust/runtime/random/tests/transcript.rs:19-67).  Build container only."""
import json
import os
import re
import shutil
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(REF, "rust/runtime/zk/tests/zk.rs")).read()
body = src[src.index("fn test_zk_rfc_testvector1"):]


def array(name):
    m = re.search(r"let\s+%s\s*:\s*&\[u8\]\s*=\s*&\[(.*?)\];" % name, body, re.S)
    assert m, name
    return bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", m.group(1)))


out = {
    "source": "rust/runtime/zk/tests/zk.rs:230-345 (test_zk_rfc_testvector1; C++-generated expectations)",
    "field": "GF2_128<4>", "transcript_seed": "test", "rateinv": 4, "nreq": 6, "block_enc": 128,
    "rng": "TestRng: every RandomEngine::bytes(len) call returns {2, 0, 0, ...} (zk.rs:283-292)",
    "witness": "W = [1, of_scalar(5), of_scalar(6), (of_scalar(5) + of_scalar(6)) * x]  (x = the element with integer image 2)",
    "circuit_lfc1": array("circuit_bytes").hex(),
    "expected_sc_proof": array("expected_sc_proof").hex(),
    "expected_com": array("expected_com").hex(),
    "expected_com_proof": array("expected_com_proof").hex(),
}
assert re.search(r"rateinv:\s*4", body) and re.search(r"nreq:\s*6", body) and re.search(r"block_enc:\s*128", body)
dst = os.path.join(ROOT, "tests", "golden", "zk_rs_testvector1.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, {k: len(v) // 2 for k, v in out.items() if k.startswith(("circuit", "expected"))})
shutil.copyfile(os.path.join(REF, "rust/runtime/random/tests/transcript_test_vector.bin"), os.path.join(ROOT, "tests", "golden", "transcript_test_vector.bin"))
os.chmod(os.path.join(ROOT, "tests", "golden", "transcript_test_vector.bin"), 0o644)
