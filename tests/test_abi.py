"""The C-ABI libraries load and export every symbol include/*.h declares; without a
device the product refuses to run instead of falling back."""
import ctypes as C
import os
import re

import pytest

from frog_amd import _abi

INC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")


def declared(header):
    text = open(os.path.join(INC, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frog_[a-z0-9_]+)\s*\(", text)) - {"frog_options_default", "frog_volume_voxel_bytes"})    # static inline helpers


def test_device_library_exports_every_declared_symbol():
    lib = _abi.hip_lib()
    names = sorted(declared("frog_hip.h") + declared("frog_match.h") + declared("frog_chain.h"))   # one library, three headers
    assert len(names) >= 46
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_abi.HIP_SYMBOLS) == names        # the ctypes table is complete, nothing undeclared


def test_host_library_exports_every_declared_symbol():
    lib = _abi.host_lib()
    names = declared("frog_host.h")
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_abi.HOST_SYMBOLS) == names


def test_comm_library_exports_every_declared_symbol():
    # libfrog_comm.so links RCCL: loaded in a child process, so that this one never maps it next to torch's own copy
    import subprocess
    import sys
    names = declared("frog_comm.h")
    assert len(names) >= 8
    assert sorted(_abi.COMM_SYMBOLS) == names       # the ctypes table of the hosts that shard from C is complete
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); "
            "missing = [n for n in sys.argv[2:] if not hasattr(L, n)]; print(missing); sys.exit(1 if missing else 0)")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(_abi.LIB_DIR, "libfrog_comm.so")] + names,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_struct_layouts_match_the_headers():
    assert C.sizeof(_abi.FrogOptions) == 4 * 10 + 4 * 6
    assert C.sizeof(_abi.FrogGridInfo) == 16 + 8 * 12
    assert C.sizeof(_abi.FrogCounts) == 8 * 4 + 4 * 4
    assert C.sizeof(_abi.FrogKernelTime) == 16
    assert C.sizeof(_abi.FrogModel) == 8 + 5 * 8
    assert C.sizeof(_abi.FrogKeypoints) == 8 + 5 * 8
    assert C.sizeof(_abi.FrogMatchOptions) == 4 * 4 + 4 * 4
    # frog_schedule_plan / frog_schedule_result (frog_host.h): the library itself refuses a caller whose sizes differ
    # (plan_bytes / result_bytes); the sizes a C compiler gives the header, checked where a compiler is at hand
    import shutil
    import subprocess
    import tempfile
    if shutil.which("g++"):
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "s.cpp")
            open(src, "w").write('#include "frog_host.h"\n#include <cstdio>\nint main(){printf("%zu %zu", sizeof(frog_schedule_plan), sizeof(frog_schedule_result));}')
            subprocess.run(["g++", "-I", INC, src, "-o", os.path.join(d, "s")], check=True)
            a, b = subprocess.run([os.path.join(d, "s")], capture_output=True, text=True).stdout.split()
        assert (int(a), int(b)) == (C.sizeof(_abi.FrogSchedulePlan), C.sizeof(_abi.FrogScheduleResult))


def test_no_cpu_fallback(tiny_pairs):
    lib = _abi.hip_lib()
    if lib.frog_device_count() > 0:
        pytest.skip("a HIP device is present")
    ctx = C.c_void_p()
    o = _abi.FrogOptions.default()
    rc = lib.frog_create(C.byref(tiny_pairs.model), C.byref(o), 0, 0, tiny_pairs.n_images, C.byref(ctx))
    assert rc == _abi.FROG_E_NODEVICE and not ctx.value
    assert b"no CPU fallback" in lib.frog_last_error()


def test_argument_validation_comes_before_device_use(tiny_pairs):
    lib = _abi.hip_lib()
    ctx = C.c_void_p()
    o = _abi.FrogOptions.default()
    assert lib.frog_create(C.byref(tiny_pairs.model), C.byref(o), 0, 3, 2, C.byref(ctx)) == _abi.FROG_E_INVALID
    o.reserved[0] = 1
    assert lib.frog_create(C.byref(tiny_pairs.model), C.byref(o), 0, 0, 4, C.byref(ctx)) == _abi.FROG_E_INVALID


def test_product_does_not_reference_the_oracle():
    root = os.path.dirname(INC)
    for base, _, files in os.walk(os.path.join(root, "frog_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.replace("(no CPU engine", ""), os.path.join(base, f)


def test_host_threads_follow_the_usable_cpus(monkeypatch):
    """The C++ rule (csrc/common/usable_cpus.h) and the Python one (_abi.usable_cpus) read the same affinity mask and cgroup
    files: same answer, within the machine's CPU count, and OMP_NUM_THREADS still wins when it asks for fewer."""
    import os, subprocess, sys
    from frog_amd import _abi
    n = _abi.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    code = "import sys; sys.path.insert(0, %r); from frog_amd import _abi; print(_abi.host_lib().frog_host_threads())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "OMP_NUM_THREADS"}
    assert int(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout) == min(n, 64)
    env["OMP_NUM_THREADS"] = "1"
    assert int(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout) == 1
