#!/bin/bash
# usage: fp8_small_trace.sh <MxNxK,...>  -- microseconds of GPU time per product (sum over the launches of one call)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof/fp8small
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/fp8small -- python3 scripts/fp8_small_trace.py "$1" > /dev/null 2> gpurun_out/fp8_small_err.txt
python3 - "$1" <<'PY'
import csv, glob, sys
shapes = sys.argv[1].split(",")
f = glob.glob("gpurun_out/prof/fp8small/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = ["ours", "torch"]
total, i, line, started, kn = 0.0, 0, {}, False, set()
for r in rows:
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "FillFunctor<double>" in n:
        if not started:
            started = True
        else:
            line[names[i % 2]] = round(total / 20, 2); line[names[i % 2] + "_kernels"] = sorted(kn); i += 1
            if i % 2 == 0:
                print(shapes[i // 2 - 1], line); line = {}; started = False
        total, kn = 0.0, set()
    elif started and "elementwise" not in n and "Fill" not in n:
        total += d; kn.add(n[:48])
PY
