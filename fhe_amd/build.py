"""Builds liblfhe.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree next to this file."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblfhe.so")
SOURCES = ["kernels.hip", "engine.cpp", "host_tfhe.cpp", "api.cpp", "radix.cpp", "radix_program.cpp"]
HEADERS = ["pbs_device.h", "pbs_phases.h", "tables.h", "engine.h", "host_tfhe.h", "ctx.h", "radix_program.h",
           os.path.join("..", "..", "include", "lfhe.h")]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, out: str = OUT, defines: tuple = ()) -> str:
    if out == OUT and not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread",
           "-Wall", "-Wno-unused-function", "-x", "hip"]
    cmd += [f"-D{d}" for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
