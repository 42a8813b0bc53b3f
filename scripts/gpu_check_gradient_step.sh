#!/bin/bash
# GPU check of the gradient-mode step: tests, then the kernels of 300 captured steps under the
# collective runtime (RCCL, world of one), fused against the chain, SGD and Adam.
set -o pipefail
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -u -m pytest tests/test_gpu_rccl.py tests/test_gpu_examples.py tests/test_optimizers.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3b_tests.log 2>&1
rc=$?; tail -25 gpurun_out/r3b_tests.log
[ $rc -ne 0 ] && exit $rc
fi
cd /tmp && export TMPDIR=/tmp
for tag in sgd_fused sgd_chain adam_fused; do
  case $tag in
    sgd_fused) extra="";;
    sgd_chain) extra="--chain";;
    adam_fused) extra="--optimizer adam";;
  esac
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o run -- python3 $GRAFT_REPO_ROOT/bench/run_collective_steps.py --steps 300 $extra > $GRAFT_REPO_ROOT/gpurun_out/r3b_$tag.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/r3b_$tag.log; exit 1; }
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/r3b_$tag.log
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r3b_${tag}_kernel_stats.csv
  head -14 "$f" | cut -c1-150
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
done
