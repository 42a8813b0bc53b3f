"""Static checks on the compiled gfx950 kernels (hipcc cross-compiles without a GPU).

The GEMM kernels wait on LDS-DMA with COUNTED ``s_waitcnt vmcnt(N)``: a register spill inside
them adds scratch loads and stores to the same counter and silently turns those waits into
waits for the wrong transfers. The hot kernels must therefore compile without scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = os.path.join(ROOT, "distributed_tensorflow_amd", "csrc", "kernels")


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc")
                                     else None)


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not installed")
@pytest.mark.parametrize("source,pattern", [("gemm_bf16.hip", "gemm_nt_pp_kernel"),
                                            ("gemm_f32.hip", "gemm_f32_kernel"),
                                            ("gemm_f32.hip", "gemm_f32_slot_kernel"),
                                            ("mlp_head.hip", "mlp_head")])
def test_hot_kernels_use_no_scratch(source, pattern, tmp_path):
    asm = str(tmp_path / "k.s")
    proc = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                           "--cuda-device-only", "-I" + KERNELS, "-o", asm,
                           os.path.join(KERNELS, source)],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:]
    text = open(asm).read()
    # each kernel's resource block: "; Kernel Name ... ; ScratchSize: N"
    found = 0
    for m in re.finditer(r"^(_Z\w+):.*?; ScratchSize: (\d+)", text, re.S | re.M):
        name, scratch = m.group(1), int(m.group(2))
        if pattern in name:
            found += 1
            assert scratch == 0, "%s spills %d bytes of scratch" % (name, scratch)
    assert found > 0


def test_every_declared_entry_point_is_exported_by_the_built_library():
    """The ctypes signatures in ops/hip.py and the symbols of the in-tree libdtf_hip.so must
    agree (a launcher lost from a .hip file otherwise shows up only on the GPU box)."""
    import ctypes
    import re
    from distributed_tensorflow_amd.ops import hip
    if not hip.library_present():
        pytest.skip("native library not built")
    lib = ctypes.CDLL(hip.LIB_PATH)
    src = open(os.path.join(ROOT, "distributed_tensorflow_amd", "ops", "hip.py")).read()
    names = sorted(set(re.findall(r'"(dtf_\w+)":', src)))
    assert len(names) > 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
