"""The boundary is a C ABI, not a Python extension: examples/c_abi_demo.c is compiled with plain gcc against
include/ibgs_rast.h + libibgs_rast.so (+ the HIP runtime for device memory) and runs forward and backward on the MI355X."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_c_program_drives_forward_and_backward(tmp_path):
    from ibgs_amd import _lib
    _lib.load()                                         # makes sure the library is built
    exe = str(tmp_path / "c_abi_demo")
    lib_dir = os.path.join(ROOT, "ibgs_amd")
    cmd = ["gcc", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", lib_dir, "-libgs_rast", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libibgs_rast.so" in ldd and "torch" not in ldd and "python" not in ldd
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0 and "c_abi_demo OK" in out.stdout, out.stdout + out.stderr
