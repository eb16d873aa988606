"""Builds libgmvae_hip.so in-tree with hipcc for gfx950 (no hipify, no JIT cache)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "gmvae_hip.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "gemm.hpp"), os.path.join(HERE, "csrc", "kernels.hpp"),
        os.path.join(os.path.dirname(HERE), "include", "gmvae_hip.h")]
OUT = os.path.join(HERE, "lib", "libgmvae_hip.so")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
