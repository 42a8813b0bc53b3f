#!/usr/bin/env python3
"""Build the native libraries in-tree.

* ``distributed_tensorflow_amd/lib/libdtf_hip.so``  -- every gfx950 kernel (``csrc/kernels/*.hip``)
  compiled with ``hipcc --offload-arch=gfx950``; plain C ABI, no torch headers, so a file
  compiles in seconds and the library loads with ``ctypes``.
* ``distributed_tensorflow_amd/lib/libdtf_host.so`` -- the host runtime (``csrc/io``,
  ``csrc/runtime``: CRC-32C, event-file writer, checkpoint SSTable, parameter-server data
  plane) compiled with ``g++``.

hipcc cross-compiles without a GPU, so this runs on a CPU-only build box; the resulting
``.so`` files are git-ignored build artefacts that travel with the source tree to the GPU
machine. Object files are cached under ``build/`` and only rebuilt when a source or header
is newer.
"""
from __future__ import annotations

import argparse
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "distributed_tensorflow_amd")
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(ROOT, "build", "obj")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd) -> None:
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write("FAILED: %s\n%s\n" % (" ".join(cmd), proc.stdout))
        raise SystemExit(1)
    if proc.stdout.strip() and os.environ.get("DTF_BUILD_VERBOSE"):
        print(proc.stdout)


def build_hip(force: bool = False, jobs: int = 8) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    sources = sorted(glob.glob(os.path.join(CSRC, "kernels", "*.hip")))
    headers = glob.glob(os.path.join(CSRC, "kernels", "*.h"))
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden",
             "-Wall", "-Wno-unused-function", "-I" + os.path.join(CSRC, "kernels")]
    if os.environ.get("DTF_GEMM_EXPERIMENTS"):      # timing-ablation variants of the GEMM kernels
        flags.append("-DDTF_GEMM_EXPERIMENTS")
    # a change of flags (e.g. DTF_GEMM_EXPERIMENTS toggled) rebuilds everything
    stamp = os.path.join(OBJDIR, "hip_flags.txt")
    flag_text = " ".join(flags)
    if not os.path.exists(stamp) or open(stamp).read() != flag_text:
        force = True
        with open(stamp, "w") as fh:
            fh.write(flag_text)
    objs, todo = [], []
    for src in sources:
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            todo.append([hipcc] + flags + ["-c", src, "-o", obj])
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(_run, todo))
    out = os.path.join(LIBDIR, "libdtf_hip.so")
    if todo or force or not os.path.exists(out):
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_host(force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    sources = sorted(glob.glob(os.path.join(CSRC, "io", "*.cc"))
                     + glob.glob(os.path.join(CSRC, "runtime", "*.cc")))
    headers = glob.glob(os.path.join(CSRC, "io", "*.h")) + \
        glob.glob(os.path.join(CSRC, "runtime", "*.h"))
    out = os.path.join(LIBDIR, "libdtf_host.so")
    if not sources:
        return out
    if force or _newer(out, sources + headers):
        _run([cxx, "-O2", "-fPIC", "-shared", "-std=c++17", "-msse4.2", "-pthread",
              "-fvisibility=hidden", "-Wall", "-I" + os.path.join(CSRC, "io"),
              "-I" + os.path.join(CSRC, "runtime"), "-o", out] + sources + ["-ldl"])
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--only", choices=["hip", "host"])
    args = ap.parse_args()
    if args.only in (None, "host"):
        print("built", build_host(args.force))
    if args.only in (None, "hip"):
        print("built", build_hip(args.force, args.jobs))


if __name__ == "__main__":
    main()
