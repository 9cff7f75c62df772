"""Builds libagents_amd.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

`python -m agents_amd._build` or `__graft_entry__.build()` calls `build()`.  hipcc
cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
BUILD_DIR = os.path.join(PKG_DIR, "_build_obj")
LIB_PATH = os.path.join(PKG_DIR, "libagents_amd.so")
ARCH = "gfx950"

# (source, extra flags).  Everything except the MFMA GEMM is compiled without FMA contraction so
# that elementwise arithmetic is bit-identical to the numpy oracle.
SOURCES = [
    ("replay.hip", []),
    ("prio.hip", []),
    ("gemm.hip", []),
    ("conv_pair.hip", []),
    ("conv_pair_x6.hip", []),
    ("conv_dx_frame.hip", []),
    ("conv_dx_frame_x6.hip", []),
    ("conv_dw_frame_x6.hip", []),
    ("nn.hip", ["-ffp-contract=off"]),
    ("dense_small.hip", []),
    ("mlp_small.hip", []),
    ("mlp_wide.hip", ["-ffp-contract=off"]),
    ("dqn.hip", ["-ffp-contract=off"]),
    ("optim.hip", ["-ffp-contract=off"]),
    ("rollout.hip", ["-ffp-contract=off"]),
    ("value_ops.hip", ["-ffp-contract=off"]),
    ("ppo.hip", ["-ffp-contract=off"]),
    ("ppo_fused.hip", ["-ffp-contract=off"]),
    ("sac.hip", ["-ffp-contract=off"]),
    ("normalizer.hip", ["-ffp-contract=off"]),
    ("runtime_guard.hip", []),
]
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
    [os.path.join(INCLUDE, "agents_amd.h")]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build libagents_amd.so")


def _digest(paths, flags):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _compile_one(hipcc, src, extra, verbose):
    src_path = os.path.join(CSRC, src)
    obj = os.path.join(BUILD_DIR, src.replace(".hip", ".o"))
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}",
             "-Wall", "-Wno-unused-function"] + extra
    stamp = obj + ".sha"
    dig = _digest([src_path] + HEADERS, flags)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [hipcc] + flags + ["-c", src_path, "-o", obj]
    if verbose:
        print("[agents_amd build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(verbose=True, force=False):
    """Compile every HIP source for gfx950 and link libagents_amd.so.  Returns the .so path."""
    hipcc = _hipcc()
    os.makedirs(BUILD_DIR, exist_ok=True)
    if force:
        for f in os.listdir(BUILD_DIR):
            os.remove(os.path.join(BUILD_DIR, f))
    srcs = [(s, e) for s, e in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda se: _compile_one(hipcc, se[0], se[1], verbose), srcs))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or not os.path.exists(LIB_PATH):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print("[agents_amd build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
