#!/usr/bin/env python3
"""Runs oracle/_ref/gen_flatsha (the REAL reference's circuit builder, witness generator and
sumcheck prover) and stores xz-compressed fixtures under tests/golden/:
  flatsha_nb<N>.lfc1.xz  circuit in the reference's LFC1 wire format (CircuitWriter output)
  flatsha_nb<N>.w.xz     witness, ninputs x 16-byte GF2_128 elements
  flatsha_nb<N>.scproof  transmitted sumcheck evaluations of run_prover (transcript "testing")
  flatsha_nb<N>.zkproof.xz  every component of the full ZK proof (ZkProver commit+prove, rate 7, 132 queries,
                         transcript "test", LCG RandomEngine seed 100): root, padded sumcheck proof, y_ldt, y_dot,
                         y_quad_0, y_quad_2, req, opened nonces, Merkle path
                         (+ length and SHA-256 of the reference's own wire serialization, ZkProof::write, in the json)
  flatsha_nb<N>.json     sizes + the reference's single-thread timings in this container
Build container only (needs /root/reference)."""
import hashlib
import json
import lzma
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "oracle", "_ref", "gen_flatsha")
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    blocks = [int(a) for a in sys.argv[1:]] or [1, 32]
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/gen_flatsha"])
    for nb in blocks:
        with tempfile.TemporaryDirectory() as td:
            pre = os.path.join(td, "x")
            info = json.loads(subprocess.check_output([GEN, str(nb), pre]).decode())
            for ext, comp in ((".lfc1", True), (".w", True), (".scproof", False), (".zkproof", True)):
                data = open(pre + ext, "rb").read()
                dst = os.path.join(OUT, "flatsha_nb%d%s" % (nb, ext + (".xz" if comp else "")))
                with open(dst, "wb") as f:
                    f.write(lzma.compress(data, preset=9 | lzma.PRESET_EXTREME) if comp else data)
                print(dst, os.path.getsize(dst))
            wire = open(pre + ".zkwire", "rb").read()  # ZkProof::write bytes: kept as length + SHA-256 only
            info["zk_wire_bytes"] = len(wire)
            info["zk_wire_sha256"] = hashlib.sha256(wire).hexdigest()
            with open(os.path.join(OUT, "flatsha_nb%d.json" % nb), "w") as f:
                json.dump(info, f)
            print(info)


def light(blocks):
    """the other block counts of BM_ShaZK_fp2_128 (docs/content/en/docs/benchmarks.md:55-61; 33 = ragged, non-power-of-two
    witness): circuit + witness + the reference's commitment root and the length / SHA-256 of its wire bytes only"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/gen_flatsha"])
    for nb in blocks:
        with tempfile.TemporaryDirectory() as td:
            pre = os.path.join(td, "x")
            info = json.loads(subprocess.check_output([GEN, str(nb), pre]).decode())
            for ext in (".lfc1", ".w"):
                data = open(pre + ext, "rb").read()
                dst = os.path.join(OUT, "flatsha_nb%d%s.xz" % (nb, ext))
                with open(dst, "wb") as f:
                    f.write(lzma.compress(data, preset=9 | lzma.PRESET_EXTREME))
                print(dst, os.path.getsize(dst))
            wire = open(pre + ".zkwire", "rb").read()
            info.update(zk_wire_bytes=len(wire), zk_wire_sha256=hashlib.sha256(wire).hexdigest(),
                        zk_root=open(pre + ".zkproof", "rb").read()[:32].hex())
            with open(os.path.join(OUT, "flatsha_nb%d.json" % nb), "w") as f:
                json.dump(info, f)
            print(info)


def variant(nb=1, npub=9, sfb=777):
    """same circuit with the first `npub` inputs declared public and inputs below `sfb` declared subfield (what the
    mdoc hash circuit does): only sizes and hashes are stored -- the LFC1 bytes are the nb fixture with two header
    fields patched (pinned by lfc1_sha256), the witness is the nb fixture's"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/gen_flatsha"])
    with tempfile.TemporaryDirectory() as td:
        pre = os.path.join(td, "x")
        info = json.loads(subprocess.check_output([GEN, str(nb), pre, str(npub), str(sfb)]).decode())
        wire = open(pre + ".zkwire", "rb").read()
        info.update(npub_in=npub, subfield_boundary=sfb, zk_wire_bytes=len(wire), zk_wire_sha256=hashlib.sha256(wire).hexdigest(),
                    lfc1_sha256=hashlib.sha256(open(pre + ".lfc1", "rb").read()).hexdigest())
        dst = os.path.join(OUT, "flatsha_nb%d_pub%d_sfb%d.json" % (nb, npub, sfb))
        with open(dst, "w") as f:
            json.dump(info, f)
        print(dst, info)


def fp128(nb=1):
    """the same generator compiled over Fp128 (oracle/_ref/gen_flatsha_fp): ZK over the prime field"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/gen_flatsha_fp"])
    with tempfile.TemporaryDirectory() as td:
        pre = os.path.join(td, "x")
        info = json.loads(subprocess.check_output([GEN + "_fp", str(nb), pre]).decode())
        for ext in (".lfc1", ".w", ".zkproof"):
            data = open(pre + ext, "rb").read()
            dst = os.path.join(OUT, "flatsha_fp_nb%d%s.xz" % (nb, ext))
            with open(dst, "wb") as f:
                f.write(lzma.compress(data, preset=9 | lzma.PRESET_EXTREME))
            print(dst, os.path.getsize(dst))
        wire = open(pre + ".zkwire", "rb").read()
        info.update(field="Fp128", zk_wire_bytes=len(wire), zk_wire_sha256=hashlib.sha256(wire).hexdigest())
        with open(os.path.join(OUT, "flatsha_fp_nb%d.json" % nb), "w") as f:
            json.dump(info, f)
        print(info)


if __name__ == "__main__":
    if sys.argv[1:2] == ["fp128"]:
        fp128()
    elif sys.argv[1:2] == ["variant"]:
        variant()
    elif sys.argv[1:2] == ["light"]:
        light([int(a) for a in sys.argv[2:]] or [2, 4, 8, 16, 33])
    else:
        main()
