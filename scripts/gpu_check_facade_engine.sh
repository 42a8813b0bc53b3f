#!/bin/bash
# GPU check of the facade-on-engine work: new tests, then the headline bench next to the same
# product through sess.run.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_peer_mem.py tests/test_gpu_rccl.py tests/test_gpu_kernels.py -m gpu -x -q --timeout 300 --timeout-method thread -k "rehearsed or pack_into or native or watchdog or identical" > gpurun_out/r3a_tests.log 2>&1
rc=$?; tail -25 gpurun_out/r3a_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err || { tail -20 gpurun_out/r3a_bench.err; exit 1; }
cat gpurun_out/r3a_bench.json
timeout -k 10 300 python bench.py --through-session > gpurun_out/r3a_bench_session.json 2> gpurun_out/r3a_bench_session.err || { tail -30 gpurun_out/r3a_bench_session.err; exit 1; }
cat gpurun_out/r3a_bench_session.json
timeout -k 10 300 python bench.py --through-session --dtype fp8 > gpurun_out/r3a_bench_session_fp8.json 2> gpurun_out/r3a_bench_session_fp8.err || { tail -30 gpurun_out/r3a_bench_session_fp8.err; exit 1; }
cat gpurun_out/r3a_bench_session_fp8.json
