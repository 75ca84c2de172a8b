"""Builds tests/fake_rccl/libfake_rccl.so - the transport double of the native halo exchange's tests (fake_rccl.c;
TEST INFRASTRUCTURE: never part of the product, loaded only where a test names it in SEIGEN_RCCL_LIB)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.c")
OUT = os.path.join(HERE, "libfake_rccl.so")


def build(force=False):
    """gcc -> libfake_rccl.so beside the source (rebuilt when the source is newer); returns its path."""
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        tmp = OUT + ".%d.tmp" % os.getpid()
        subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-g", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared",
                               "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), SRC, "-o", tmp,
                               "-ldl", "-lrt", "-pthread"])
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
