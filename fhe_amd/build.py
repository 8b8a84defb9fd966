"""Builds liblfhe.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree next to this file."""
from __future__ import annotations

import fcntl
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblfhe.so")
SOURCES = ["kernels.hip", "ks_mfma.hip", "noalloc.hip", "engine.cpp", "host_tfhe.cpp", "api.cpp", "radix.cpp", "radix_program.cpp"]
HEADERS = ["pbs_device.h", "pbs_phases.h", "ms_noise.h", "tables.h", "engine.h", "host_tfhe.h", "ctx.h", "radix_program.h",
           os.path.join("..", "..", "include", "lfhe.h")]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


OBJ_DIR = os.path.join(HERE, "_build")


def build(force: bool = False, verbose: bool = False, out: str = OUT, defines: tuple = ()) -> str:
    """One object per source under fhe_amd/_build (compiled side by side, kept between builds: a
    change to one source recompiles that source, a change to a header everything), then one link."""
    if out == OUT and not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function"]
    flags += [f"-D{d}" for d in defines]
    # objects of a non-default build (other defines, other output) get a tag that is the same in every process
    tag = "" if out == OUT else "." + hashlib.sha1("\0".join((out,) + tuple(defines)).encode()).hexdigest()[:10]
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    # one builder at a time (ranks, a test run beside a benchmark): the others wait and then find fresh objects
    lock = open(os.path.join(OBJ_DIR, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        return _build_locked(force, verbose, out, flags, tag, newest_header, hipcc)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(force, verbose, out, flags, tag, newest_header, hipcc) -> str:
    from concurrent.futures import ThreadPoolExecutor
    if out == OUT and not force and not _stale():
        return OUT      # built by whoever held the lock before us

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, src + tag + ".o")
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj
        tmp = f"{obj}.{os.getpid()}.tmp"
        cmd = [hipcc] + flags + ["-x", "hip", "-c", path, "-o", tmp]
        report = src.endswith(".hip")
        if report:   # registers, scratch and occupancy per kernel, kept next to the object (tests read them)
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        if report:
            done = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            remarks = [l for l in done.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l]
            rest = [l for l in done.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l]
            if rest:
                print("\n".join(rest), file=sys.stderr)
            if done.returncode:
                raise subprocess.CalledProcessError(done.returncode, cmd)
            with open(f"{obj}.resources.{os.getpid()}.tmp", "w") as f:
                f.write("\n".join(remarks) + "\n")
            os.replace(f"{obj}.resources.{os.getpid()}.tmp", obj + ".resources.txt")
        else:
            subprocess.check_call(cmd)
        os.replace(tmp, obj)      # a reader never sees a half-written object
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), int(os.environ.get("MAX_JOBS", "8")))) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    tmp = f"{out}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-pthread"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, out)
    return out


NODE_DIR = os.path.join(os.path.dirname(HERE), "bindings", "node")


def build_node_addon(verbose: bool = False) -> str | None:
    """The N-API addon over the C ABI (bindings/node/lfhe_napi.node), plain C against the Node
    headers of the image.  Returns None when the headers are absent."""
    inc = os.environ.get("NODE_INCLUDE", "/usr/include/node")
    if not os.path.exists(os.path.join(inc, "node_api.h")):
        return None
    out = os.path.join(NODE_DIR, "lfhe_napi.node")
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", f"-I{inc}",
           "-DNODE_GYP_MODULE_NAME=lfhe_napi", "-o", out, os.path.join(NODE_DIR, "lfhe_napi.c"), "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_node_addon(verbose=True))
