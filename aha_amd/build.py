"""Builds aha_amd/csrc/libaha_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree, no JIT cache."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libaha_hip.so")
SOURCES = ["kernels_elem.hip", "kernels_gemv.hip", "kernels_attn.hip", "kernels_attn64.hip", "kernels_gemm.hip", "kernels_gemm_sk.hip", "kernels_vit.hip", "kernels_audio.hip", "kernels_sample.hip", "sampler_rng.hip", "audio_tower.hip", "audio_pre.hip", "image_pre.hip", "tp_rccl.hip", "loader.hip", "model.hip",
           "vision.hip", "vision_tower.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-Wno-unused-result"]
# Debug builds (measurement scripts only; never the shipped library): AHA_BUILD_DEFINES="-DAHA_DEBUG_KERNELS" compiles the ablation / trace
# instantiations and their environment dispatch in (README "Debug kernels"); the flags are part of the build digest.
FLAGS += [f for f in os.environ.get("AHA_BUILD_DEFINES", "").split() if f.startswith("-D")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())  # not the absolute path: the tree is copied to other roots (gpurun)
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(HERE, "..", "include", "aha_hip.h")]
    stamp = os.path.join(CSRC, ".build_stamp")
    dig = _digest(srcs + hdrs)
    def fresh() -> bool:
        return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig

    if not force and fresh():
        return LIB
    # one builder at a time: several ranks of one node (bench.py --gpus N) may all find the library missing
    import fcntl
    lock = open(os.path.join(CSRC, ".build_lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and fresh():  # another process built it while this one waited
            return LIB
        return _build_locked(srcs, stamp, dig, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(srcs, stamp: str, dig: str, verbose: bool) -> str:
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs, "-L/opt/rocm/lib", "-lrccl",
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(tmp, LIB)  # atomic: a process that dlopens concurrently sees the old or the new file, never a partial one
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
