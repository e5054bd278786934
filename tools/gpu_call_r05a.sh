#!/bin/bash
# three mdoc end-to-end cases through the reference's run_mdoc_prover / run_mdoc_verifier bodies
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_reference_integration.py -m gpu -x -q -s -k mdoc_end_to_end > gpurun_out/mdoc_cases.log 2>&1
tail -5 gpurun_out/mdoc_cases.log
