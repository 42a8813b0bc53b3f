#!/bin/bash
# usage: f32_split_trace.sh <MxKxN,...>  -- kernel-trace of scripts/f32_split_trace.py, one line per (shape, plan):
# microseconds of GPU time per product, kernels per product
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof/f32split
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/f32split -- python3 scripts/f32_split_trace.py "$1" > /dev/null 2>&1
python3 - "$1" <<'PY'
import csv, glob, sys
shapes = sys.argv[1].split(",")
f = glob.glob("gpurun_out/prof/f32split/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
group, out = [], []
for r in rows:
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "gemm_f32" in n or "splitk" in n.lower():
        group.append(d)
    elif group:
        out.append((round(sum(group) / 20.0, 2), len(group) // 20)); group = []
plans = ["default", "unsplit", "split2"]
for i, o in enumerate(out):
    print(shapes[i // 3], plans[i % 3], o)
PY
