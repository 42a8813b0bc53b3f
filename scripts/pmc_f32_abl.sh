#!/bin/bash
# MFMA-busy against elapsed cycles for the ablated fp32 GEMM loops and for the bare MFMA probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof/f32_abl
mkdir -p "$out"
for abl in 0 15 7; do
  DTF_F32_ABL=$abl timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$out/abl$abl" -- \
    python3 bench/prof_gemm.py --n 8192 --dtype f32 --iters 3 > "$out/abl$abl.log" 2>&1 || exit 1
done
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$out/probe" -- \
  scripts/probes/mfma_f32_data 5000 512 > "$out/probe.log" 2>&1 || exit 1
for d in abl0 abl15 abl7; do echo "== $d"; python3 scripts/pmc_summary.py "$out/$d" gemm_f32; done
echo "== probe"; python3 scripts/pmc_summary.py "$out/probe" k32
