#!/bin/bash
# usage: bf16_ws_plan_trace.sh <MxNxK,...>  -- microseconds of GPU time per product (sum over the launches of one call)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof/wsplan
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/wsplan -- python3 scripts/bf16_ws_plan_trace.py "$1" > /dev/null 2> gpurun_out/ws_plan_err.txt
python3 - "$1" <<'PY'
import csv, glob, sys
shapes = sys.argv[1].split(",")
f = glob.glob("gpurun_out/prof/wsplan/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = ["unsplit", "t64", "s2", "s4", "s8", "s16", "default", "torch"]
total, count, i, line, started = 0.0, 0, 0, {}, False
for r in rows:
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "FillFunctor<double>" in n:          # the script's own markers (torch launches fills of other types itself)
        if not started:
            started = True          # the marker in front of a shape's first case
        else:
            if count:
                line[names[i % 8]] = round(total / 20, 2)
            i += 1
            if i % 8 == 0:
                print(shapes[i // 8 - 1], line); line = {}; started = False
        total, count = 0.0, 0
    elif started and ("gemm_bf16" in n or "Cijk" in n or "gemm_nt" in n or "planes_reduce" in n):
        total += d; count += 1
PY
