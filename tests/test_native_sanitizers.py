"""The C++ runtime under sanitizers (host code only; CPU): ``csrc/io/host_io.cc`` with its
self-test under AddressSanitizer + UBSan, and the multi-threaded parameter-server task
``csrc/runtime/ps_server.cc`` with concurrent socket clients under ThreadSanitizer, and the TCP
rendezvous of the native RCCL communicator (``csrc/runtime/rendezvous.cc``) with its ranks as
threads under both."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distributed_tensorflow_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_io_selftest_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_io_selftest")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-msse4.2",
           os.path.join(CSRC, "io", "host_io.cc"),
           os.path.join(CSRC, "tests", "host_io_selftest.cc"), "-o", exe]
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout
    work = tmp_path / "work"
    work.mkdir()
    # the environment may preload libraries ahead of the sanitizer runtime; ASan is told not to
    # insist on being first (its interceptors still cover this binary's own allocations)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0")
    run = subprocess.run([exe, str(work)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, env=env, timeout=120)
    assert run.returncode == 0, run.stdout
    assert "host_io selftest ok" in run.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_ps_server_selftest_under_asan_ubsan(tmp_path):
    """Same self-test (concurrent clients, storage replaced under readers, malformed frames) with
    AddressSanitizer + UBSan: a reader left holding freed variable storage, or a frame whose sizes
    wrap, shows up here as a heap error."""
    exe = str(tmp_path / "ps_server_selftest_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
           os.path.join(CSRC, "runtime", "ps_server.cc"),
           os.path.join(CSRC, "tests", "ps_server_selftest.cc"), "-o", exe]
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0")
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         env=env, timeout=180)
    assert "AddressSanitizer" not in run.stdout and "runtime error" not in run.stdout, run.stdout
    assert run.returncode == 0, run.stdout
    assert "ps_server selftest ok" in run.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_ps_server_selftest_under_tsan(tmp_path):
    exe = str(tmp_path / "ps_server_selftest")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=thread",
           "-pthread", os.path.join(CSRC, "runtime", "ps_server.cc"),
           os.path.join(CSRC, "tests", "ps_server_selftest.cc"), "-o", exe]
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if build.returncode != 0 and "tsan" in build.stdout.lower():
        pytest.skip("toolchain has no ThreadSanitizer runtime")
    assert build.returncode == 0, build.stdout
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         env=env, timeout=120)
    assert "ThreadSanitizer" not in run.stdout, run.stdout
    assert run.returncode == 0, run.stdout
    assert "ps_server selftest ok" in run.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_rendezvous_selftest_under_sanitizers(tmp_path, sanitizer):
    """Five ranks as threads (all_gather up to 1 MiB, broadcasts, barriers), a job one rank
    short (bounded failure, no hang), a peer that leaves before a collective, argument errors."""
    exe = str(tmp_path / "rendezvous_selftest")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + sanitizer,
           "-pthread", os.path.join(CSRC, "runtime", "rendezvous.cc"),
           os.path.join(CSRC, "tests", "rendezvous_selftest.cc"), "-o", exe]
    if sanitizer != "thread":
        cmd.insert(7, "-fno-sanitize-recover=undefined")
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if build.returncode != 0 and sanitizer == "thread" and "tsan" in build.stdout.lower():
        pytest.skip("toolchain has no ThreadSanitizer runtime")
    assert build.returncode == 0, build.stdout
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66",
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0")
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         env=env, timeout=180)
    for word in ("ThreadSanitizer", "AddressSanitizer", "runtime error"):
        assert word not in run.stdout, run.stdout
    assert run.returncode == 0, run.stdout
    assert "rendezvous selftest ok" in run.stdout
