#!/bin/bash
# usage: t128_spread_trace.sh <MxNxK,...>  -- microseconds of GPU time per product: requests in one run, spread, torch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof/t128s
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/t128s -- python3 scripts/t128_spread_trace.py "$1" > /dev/null 2> gpurun_out/t128_spread_check.txt
grep "spread ==" gpurun_out/t128_spread_check.txt
python3 - "$1" <<'PY'
import csv, glob, sys
shapes = sys.argv[1].split(",")
f = glob.glob("gpurun_out/prof/t128s/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
group, i, line = [], 0, {}
for r in rows:
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "gemm_bf16" in n or "Cijk" in n or "gemm_nt" in n:
        group.append(d)
    elif group and "fill" in n.lower():
        if len(group) >= 20:          # (the single fp32 product of the numerics check is not a case)
            line[["four waves", "four + four request waves", "torch"][i % 3]] = round(sum(group[-15:]) / 15, 2); i += 1
            if i % 3 == 0:
                print(shapes[i // 3 - 1], line); line = {}
        group = []
PY
