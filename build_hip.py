"""Builds gmvae_amd/lib/libgmvae_hip.so in-tree with hipcc for gfx950 (no hipify, no JIT cache).

Standalone on purpose (`python build_hip.py [--force]`): importing the gmvae_amd package loads the
shared library, so the builder must not live inside it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
HERE = os.path.join(ROOT, "gmvae_amd")
SRC = os.path.join(HERE, "csrc", "gmvae_hip.hip")
import glob
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*"))) + [os.path.join(ROOT, "include", "gmvae_hip.h")]
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
    # PyTorch-ROCm ships its own libamdhip64.so (soname without the .7).  Link against THAT
    # copy so that the process holds one HIP runtime: streams and device pointers handed over
    # by torch are then valid inside this library whatever the import order.
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.submodule_search_locations:
        tl = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        if os.path.exists(os.path.join(tl, "libamdhip64.so")):
            cmd += [f"-L{tl}", f"-Wl,-rpath,{tl}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
