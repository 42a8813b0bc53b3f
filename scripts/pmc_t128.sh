#!/bin/bash
# counters of the 128x128-tile bf16 kernel on a one-workgroup-per-CU grid (2048 x 2048 x 8192)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof/t128
mkdir -p "$out"
cat > "$out/run.py" <<'PY'
import sys, torch
sys.path.insert(0, ".")
from distributed_tensorflow_amd.ops import hip
a = torch.randn(2048, 8192, device="cuda").bfloat16(); bt = torch.randn(2048, 8192, device="cuda").bfloat16()
out = torch.empty(2048, 2048, device="cuda", dtype=torch.bfloat16)
for _ in range(5): hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out)
for _ in range(5): torch.matmul(a, bt.t(), out=out)
torch.cuda.synchronize()
PY
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$out/sq1" -- python3 "$out/run.py" > "$out/sq1.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/sq2" -- python3 "$out/run.py" > "$out/sq2.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$out/run.py" > "$out/trace.log" 2>&1 || exit 1
python3 scripts/pmc_summary.py "$out" t128
python3 scripts/pmc_summary.py "$out" Cijk
grep -h "t128\|Cijk" "$out"/trace/*/*kernel_stats.csv | cut -c1-60,400-520
