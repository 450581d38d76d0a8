"""The library's multi-threaded HOST code under ThreadSanitizer / AddressSanitizer on the CPU: sift_amd/csrc/group.cpp
(one thread per shard + the gather thread, two batches in flight), phase_gate.h and launch_guard.h's locks are compiled exactly as
they ship against a HIP runtime and a context layer made of plain host code (tests/host_tsan/fake_hip.cpp) and driven by
tests/host_tsan/group_main.cpp / gate_main.cpp.  A data race, a leak of the fake device memory, a heap error or a wrong gathered
list fails the test; a deadlock runs into the timeout.  No GPU involved."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "host_tsan")
HIP_INCLUDE = "/opt/rocm/include"

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists(os.path.join(HIP_INCLUDE, "hip", "hip_runtime.h")),
                                reason="needs g++ and the HIP headers")


def _build(tmp_path, name, sanitize, sources, defines=()):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitize}", "-fno-sanitize-recover=all", "-pthread", "-D__HIP_PLATFORM_AMD__",
           "-I" + HIP_INCLUDE] + ["-D" + d for d in defines] + sources + ["-ldl", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and ("cannot find" in r.stderr or "unrecognized" in r.stderr):
        pytest.skip("this toolchain has no " + sanitize + " runtime: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(exe, args, timeout):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-2000:] + r.stderr[-6000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-6000:]
    return r.stdout


GROUP_SOURCES = [os.path.join(SRC, "group_main.cpp"), os.path.join(SRC, "fake_hip.cpp"), os.path.join(ROOT, "sift_amd", "csrc", "group.cpp")]


def test_group_threads_under_thread_sanitizer(tmp_path):
    """sift_hip_group: 1 - 4 shards (one of them on another fake device), the three wire settings, calculate and
    submit / submit / collect / collect with a larger second batch, failing frames, destruction with a batch in flight."""
    exe = _build(tmp_path, "group_tsan", "thread", GROUP_SOURCES)
    assert "group ok" in _run(exe, ["60"], 900)


def test_group_threads_under_address_sanitizer(tmp_path):
    exe = _build(tmp_path, "group_asan", "address,undefined", GROUP_SOURCES)
    assert "group ok" in _run(exe, ["40"], 900)


def test_phase_gate_under_thread_sanitizer(tmp_path):
    """PhaseGate: 2 - 4 host threads taking tickets from one gate, both schedules, batches that end early."""
    exe = _build(tmp_path, "gate_tsan", "thread", [os.path.join(SRC, "gate_main.cpp"), os.path.join(SRC, "fake_hip.cpp")])
    assert "gate ok" in _run(exe, [], 600)


# include/sift/sift.hpp's collect() worker pool and sift_amd/csrc/launch_guard.cpp as it ships
COLLECT_SOURCES = [os.path.join(SRC, "collect_main.cpp"), os.path.join(SRC, "fake_hip.cpp"), os.path.join(ROOT, "sift_amd", "csrc", "launch_guard.cpp")]


def test_collect_pool_and_launch_cache_under_thread_sanitizer(tmp_path):
    """sift::Sift::collect(): small and large results through one object (the pool starts at the first large one and parks between
    calls and at destruction), two objects on two threads; the launch locks and the table of cached function objects: threads of two
    devices resolving the same kernels, the per-thread launch error."""
    exe = _build(tmp_path, "collect_tsan", "thread", COLLECT_SOURCES, defines=("SIFT_FAKE_WITH_REAL_LAUNCH_GUARD",))
    assert "collect ok" in _run(exe, ["22"], 900)


def test_collect_pool_and_launch_cache_under_address_sanitizer(tmp_path):
    exe = _build(tmp_path, "collect_asan", "address,undefined", COLLECT_SOURCES, defines=("SIFT_FAKE_WITH_REAL_LAUNCH_GUARD",))
    assert "collect ok" in _run(exe, ["22"], 900)
