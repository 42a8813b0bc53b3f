#!/usr/bin/env python3
"""Per-kernel instruction census of a hipcc -S listing: counts of MFMA / LDS / LDS-DMA /
barrier / wait instructions overall and inside the hottest loop (the back-edge with the most
MFMAs), so pipeline-draining waits (vmcnt(0) inside the K-loop) are easy to spot."""
import re
import sys

KEYS = ["v_mfma", "ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_write", "global_load_lds",
        "buffer_load", "global_load_dword", "global_store", "s_barrier", "vmcnt(0)", "vmcnt(",
        "lgkmcnt(0)", "lgkmcnt(", "s_setprio", "scratch_", "s_nop", "v_readfirstlane"]


def main(path, pattern=""):
    text = open(path).read()
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n([\s\S]*?)s_endpgm", text, re.M):
        name, body = m.group(1), m.group(2)
        if pattern and pattern not in name:
            continue
        lines = body.split("\n")
        labels = {}
        for i, ln in enumerate(lines):
            lm = re.match(r"^(\.LBB\d+_\d+):", ln)
            if lm:
                labels[lm.group(1)] = i
        best = None
        for i, ln in enumerate(lines):
            bm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
            if bm and bm.group(1) in labels and labels[bm.group(1)] < i:
                seg = lines[labels[bm.group(1)]:i + 1]
                n = sum("v_mfma" in x for x in seg)
                if best is None or n > best[0]:
                    best = (n, seg)
        print(name[:70])
        loop = "\n".join(best[1]) if best else ""
        for k in KEYS:
            print("   %-18s total %5d   hot-loop %5d" % (k, body.count(k), loop.count(k)))
        print("   hot-loop instructions: %d" % (len(best[1]) if best else 0))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
