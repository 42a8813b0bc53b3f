#!/bin/bash
# usage: bf16_tw_trace.sh <MxNxK,...>  -- microseconds of GPU time per product and TFLOP/s: default plan, variant 13, torch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof/tw
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/tw -- python3 scripts/bf16_tw_trace.py "$1" > /dev/null 2> gpurun_out/tw_err.txt
python3 - "$1" <<'PY'
import csv, glob, sys
shapes = sys.argv[1].split(",")
f = glob.glob("gpurun_out/prof/tw/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = ["default", "t128x256", "torch"]
total, i, line, started = 0.0, 0, {}, False
for r in rows:
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "FillFunctor<double>" in n:
        if not started:
            started = True
        else:
            m, nn, k = (int(v) for v in shapes[i // 3].split("x"))
            us = total / 20
            line[names[i % 3]] = "%.1f us (%.0f TF)" % (us, 2.0 * m * nn * k / us / 1e6); i += 1
            if i % 3 == 0:
                print(shapes[i // 3 - 1], line); line = {}; started = False
        total = 0.0
    elif started and "elementwise" not in n and "Fill" not in n:
        total += d
PY
