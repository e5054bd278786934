#!/usr/bin/env python3
"""BASELINE config 5 ("End-to-end MDOC/ECDSA prove"): fixtures from the REAL reference run (oracle/_ref/gen_mdoc, built by
`make -C oracle _ref/gen_mdoc` from /root/reference; build container only).

  tests/golden/mdoc_hash.lfc1.xz   the mdoc HASH circuit (GF2_128, kZkSpecs[0]: 17 layers, 7.76 M terms, 952 public inputs,
                                   subfield boundary 85112, block_enc 4151) in the reference's LFC1 wire format
  tests/golden/mdoc_hash.w.xz      its witness as the reference's prover holds it at prove time (mdoc_tests[0], age_over_18,
                                   MACs and MAC key filled in: lib/circuits/mdoc/mdoc_zk.cc:466-515)
  tests/golden/mdoc_sig.lfc1.xz    the mdoc SIGNATURE circuit (Fp256Base, 32-byte elements, block_enc 4096) in LFC1
  tests/golden/mdoc_sig.w.xz       its witness at prove time (in-memory Elt images, 32 bytes each)
  tests/golden/mdoc.json           sizes of both circuits (hash: GF2_128, signature: Fp256Base), the Ligero parameters, and
                                   length + SHA-256 of ZkProof::write for the hash circuit proved stand-alone (transcript
                                   "test", LCG RandomEngine seed 100, rate 7, 132 queries), with the reference's timings

The witness contains MAC key shares drawn from the reference's SecureRandomEngine (fill_witness takes that concrete type), so
every run of this generator makes a different -- equally valid -- witness; circuit, witness and expected hash are stored together."""
import hashlib
import json
import lzma
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "oracle", "_ref", "gen_mdoc")
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/gen_mdoc"])
    with tempfile.TemporaryDirectory() as td:
        pre = os.path.join(td, "x")
        info = json.loads(subprocess.check_output([GEN, pre]).decode())
        for ext in (".hash.lfc1", ".hash.w"):
            data = open(pre + ext, "rb").read()
            dst = os.path.join(OUT, "mdoc_hash" + ext[5:] + ".xz")
            with open(dst, "wb") as f:
                f.write(lzma.compress(data, preset=9 | lzma.PRESET_EXTREME))
            print(dst, os.path.getsize(dst))
        wire = open(pre + ".hash.zkwire", "rb").read()
        info["hash"].update(zk_wire_bytes=len(wire), zk_wire_sha256=hashlib.sha256(wire).hexdigest(), zk_root=wire[:32].hex())
        for ext in (".sig.lfc1", ".sig.w"):
            data = open(pre + ext, "rb").read()
            dst = os.path.join(OUT, "mdoc_sig" + ext[4:] + ".xz")
            with open(dst, "wb") as f:
                f.write(lzma.compress(data, preset=9 | lzma.PRESET_EXTREME))
            print(dst, os.path.getsize(dst))
        wire = open(pre + ".sig.zkwire", "rb").read()
        info["sig"].update(zk_wire_bytes=len(wire), zk_wire_sha256=hashlib.sha256(wire).hexdigest(), zk_root=wire[:32].hex())
    with open(os.path.join(OUT, "mdoc.json"), "w") as f:
        json.dump(info, f)
    print(info)


if __name__ == "__main__":
    main()
