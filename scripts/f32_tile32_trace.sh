#!/bin/bash
# usage: f32_tile32_trace.sh <MxKxN,...>  -- microseconds of GPU time per product: default plan, 32x32-tile kernel, torch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof/f32t32
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/f32t32 -- python3 scripts/f32_tile32_trace.py "$1" > /dev/null 2>&1
python3 - "$1" <<'PY'
import csv, glob, sys
shapes = sys.argv[1].split(",")
f = glob.glob("gpurun_out/prof/f32t32/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
group, i, line = [], 0, {}
for r in rows:
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "gemm_f32" in n or "splitk" in n or "Cijk" in n:
        group.append(d)
    elif group:
        line[["default", "tile32", "torch"][i % 3]] = round(sum(group) / 20, 2); group = []; i += 1
        if i % 3 == 0:
            print(shapes[i // 3 - 1], line); line = {}
PY
