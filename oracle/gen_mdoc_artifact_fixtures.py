"""Copies the reference's STORED mdoc proofs (data its own tests hold: rust/applications/mdoc_zk/artifacts/proofs/, loaded by
rust/applications/mdoc_zk/runtime/tests/all/prior_zk.rs:131-145) into tests/golden/ for the specs the C++ reference can still
generate circuits for (kZkSpecs[0], [1] of lib/circuits/mdoc/zk_spec.cc:45-51: version 7, one and two attributes), and for one
version-6 spec (kZkSpecs[4]) together with its circuit file.

Per spec: <hash>.bin (the whole mdoc proof string: 6 MACs, hash-circuit proof, signature-circuit proof), <hash>_hash_witness.bin
and <hash>_sig_witness.bin (the complete input vectors W of both circuits, public inputs included, to_bytes_field order).
Data only -- no source text.  The witnesses are xz-compressed (they are mostly bits); the proof is stored as it is.

  python oracle/gen_mdoc_artifact_fixtures.py [/root/reference]
"""
import hashlib
import json
import lzma
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(REF, "rust", "applications", "mdoc_zk", "artifacts", "proofs")
GOLD = os.path.join(ROOT, "tests", "golden")
SPECS = [  # (index in kZkSpecs, circuit hash, version, attributes, ship the circuit file too)
    (0, "8d079211715200ff06c5109639245502bfe94aa869908d31176aae4016182121", 7, 1, False),
    (1, "6a5810683e62b6d7766ebd0d7ca72518a2b8325418142adcadb10d51dbbcd5ad", 7, 2, False),
    (3, "5aebdaaafe17296a3ef3ca6c80c6e7505e09291897c39700410a365fb278e460", 7, 4, False),  # the largest pair of circuits
    # version 6: other Ligero parameters (rate 4, 189 opened columns, block_enc_sig 2945); the C++ generate_circuit no longer builds
    # these circuits, so the compressed circuit pair the reference ships (artifacts/circuits/<hash>) travels with the fixture
    (4, "137e5a75ce72735a37c8a72da1a8a0a5df8d13365c2ae3d2c2bd6a0e7197c7c6", 6, 1, True),
]
ATTRS = ["family_name = Mustermann", "birth_date = 1971-09-01", "issue_date = 2024-03-15", "height = 175"]


def main():
    out = {"source": "rust/applications/mdoc_zk/artifacts/proofs/ of the reference (fixtures of its prior_zk.rs test)",
           "document": "mdoc_tests[3] of lib/circuits/mdoc/mdoc_examples.h == BIRTHDATE_1971_09_01_MDOC_3 of the Rust test vectors",
           "random_engine": "DeterministicRng(42): state = state * 6364136223846793005 + 1, byte = state >> 56; the first 96 bytes are generate_mac_ap's",
           "specs": []}
    for idx, h, ver, na, with_circuit in SPECS:
        proof = open(os.path.join(SRC, h + ".bin"), "rb").read()
        hw = open(os.path.join(SRC, h + "_hash_witness.bin"), "rb").read()
        sw = open(os.path.join(SRC, h + "_sig_witness.bin"), "rb").read()
        stem = "mdoc_artifact_v%d_%dattr" % (ver, na)
        open(os.path.join(GOLD, stem + ".proof.bin"), "wb").write(proof)
        open(os.path.join(GOLD, stem + ".hash_witness.xz"), "wb").write(lzma.compress(hw, preset=9))
        open(os.path.join(GOLD, stem + ".sig_witness.xz"), "wb").write(lzma.compress(sw, preset=9))
        circ = None
        if with_circuit:
            circ = open(os.path.join(REF, "rust", "applications", "mdoc_zk", "artifacts", "circuits", h), "rb").read()
            open(os.path.join(GOLD, stem + ".circuit.zst"), "wb").write(circ)
        out["specs"].append({"circuit_file_bytes": len(circ) if circ else 0, "circuit_file_sha256": hashlib.sha256(circ).hexdigest() if circ else None,
                             "zk_spec_index": idx, "circuit_hash": h, "version": ver, "attributes": ATTRS[:na], "stem": stem,
                             "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(),
                             "hash_witness_elements": len(hw) // 16, "hash_witness_sha256": hashlib.sha256(hw).hexdigest(),
                             "sig_witness_elements": len(sw) // 32, "sig_witness_sha256": hashlib.sha256(sw).hexdigest()})
    json.dump(out, open(os.path.join(GOLD, "mdoc_artifacts.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
