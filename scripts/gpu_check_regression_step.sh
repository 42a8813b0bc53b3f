#!/bin/bash
# GPU check of the fused regression step + counters in the owner update: tests, kernel traces
# (MNIST collective step, linear-regression step on a local session), step-time benches.
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_rccl.py tests/test_gpu_kernels.py tests/test_gpu_examples.py -m gpu -x -q --timeout 300 --timeout-method thread -k "mse_step or regression or gradient_step or examples or linreg or linear" > gpurun_out/regstep_tests.log 2>&1
rc=$?; tail -15 gpurun_out/regstep_tests.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
prof() {   # tag, program args...
  tag=$1; shift
  rm -rf $R/gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o run -- python3 "$@" > $R/gpurun_out/regstep_$tag.log 2>&1 || { tail -20 $R/gpurun_out/regstep_$tag.log; exit 1; }
  grep -v "^W2026\|^E2026\|simple_timer" $R/gpurun_out/regstep_$tag.log | tail -1
  f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/regstep_${tag}_kernel_stats.csv
  head -12 "$f" | cut -c1-140
  rm -rf $R/gpurun_out/prof_$tag
}
prof mnist_sgd_rs $R/bench/run_collective_steps.py --steps 300
prof linreg_fused $R/bench/bench_linreg.py --only fused --steps 2000
cd $R
timeout -k 10 300 python bench/bench_linreg.py --steps 3000 > gpurun_out/regstep_linreg_bench.jsonl 2>&1 || { tail gpurun_out/regstep_linreg_bench.jsonl; exit 1; }
cat gpurun_out/regstep_linreg_bench.jsonl
