#!/bin/bash
# Whole GPU suite after the integer dropout counter, the fused moving-average rule, the seeded
# MNIST example and the capture-time collection guard test; then, LAST, the guard's with/without
# evidence run (its unguarded child may abort).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/suite_tests.log 2>&1
rc=$?; tail -8 gpurun_out/suite_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/gc_during_capture.py > gpurun_out/gc_during_capture.jsonl 2> gpurun_out/gc_during_capture.err
rc=$?; cat gpurun_out/gc_during_capture.jsonl; tail -3 gpurun_out/gc_during_capture.err
exit $rc
