"""Host-side checks that need no GPU: the shared library loads, exports every
symbol the header declares, and refuses to work without a device (there is no
CPU fallback behind the C ABI)."""
import ctypes as C
import os
import re

import pytest

from krylov_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "krylov_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kry_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_function_is_exported():
    L = capi.lib()
    names = header_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), n
    assert sorted(capi.SYMBOLS) == names


def test_struct_layouts_match_the_header():
    # ints and doubles only; sizes as the C compiler lays them out on x86-64
    assert C.sizeof(capi.KryDesc) == 3 * 4 + 4 + 3 * 8 + 6 * 4 + 4 + 4 + 8 + 4 * 5 + 16 * 4 + 2 * 3 * 4 + 4
    assert capi.KryStats.kernel_ms.offset % 8 == 0


def test_status_strings():
    L = capi.lib()
    assert L.kry_status_string(0) == b"ok"
    assert L.kry_status_string(1) == b"alpha is nan"
    assert L.kry_status_string(3) == b"norm is nan"
    assert L.kry_status_string(-2) == b"HIP error"


def test_null_descriptor_is_rejected():
    L = capi.lib()
    st = C.c_int(0)
    assert not L.kry_create(None, C.byref(st))
    assert st.value == -1


def test_bad_arguments_are_rejected_before_any_device_work():
    with pytest.raises(capi.KryError, match="L=0"):
        capi.Solver((16, 16, 16), (1 / 16,) * 3, L=0)
    with pytest.raises(capi.KryError, match="bad slab range"):
        capi.Solver((16, 16, 16), (1 / 16,) * 3, nslabs=2, slab_begin=1, slab_count=2)
    with pytest.raises(capi.KryError, match="bad grid"):
        capi.Solver((16, 2, 16), (1 / 16,) * 3)


def _device_count():
    try:
        hip = C.CDLL("libamdhip64.so")
        n = C.c_int(0)
        return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.skipif(_device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback_without_a_device():
    with pytest.raises(capi.KryError, match="no HIP device"):
        capi.Solver((16, 16, 16), (1 / 16,) * 3)


def test_header_and_library_serve_a_plain_c99_caller(tmp_path):
    """The C ABI from C, not C++: examples/solve_poisson.c compiles with -std=c99 -pedantic -Werror
    against include/krylov_amd.h and links against the library (run on the GPU by test_gpu_compat)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "solve_poisson")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "solve_poisson.c"), "-L", os.path.join(root, "krylov_amd"),
                    "-lkrylov_amd", "-lm", "-o", out], check=True, capture_output=True, text=True)
    assert os.path.getsize(out) > 0
