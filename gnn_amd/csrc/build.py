"""Builds libmgn_hip.so (gfx950) in-tree with hipcc.  Used by ``__graft_entry__.build()``
and runnable by hand: ``python gnn_amd/csrc/build.py``."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["mgn_kernels.hip", "mgn_train.hip", "mgn_graph.hip", "mgn_halo.hip", "mgn_coarse.hip", "mgn_host.cpp"]
OUT = os.path.join(PKG, "libmgn_hip.so")
# Per-file compiler flags (none at present).
EXTRA_FLAGS = {}


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; set HIPCC")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    newest = max(os.path.getmtime(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith((".hip", ".cpp", ".h")))
    newest = max(newest, os.path.getmtime(os.path.join(PKG, "..", "include", "mgn_hip.h")))
    return newest > os.path.getmtime(OUT)


def _compile(src: str, obj: str, verbose: bool) -> None:
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stderr)


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per translation unit under ``_obj/`` (recompiled when the source or any header is newer), the
    units compiled side by side, then one link."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    headers.append(os.path.join(PKG, "..", "include", "mgn_hip.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(HERE, s), os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or verbose or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as pool:
        for fut in [pool.submit(_compile, src, obj, verbose) for src, obj in jobs]:
            fut.result()
    res = subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", OUT] + objs,
                         cwd=HERE, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return OUT


def build_examples() -> str:
    """The C++ host example (no Python, no torch): links against the library through its C ABI alone."""
    root = os.path.abspath(os.path.join(PKG, ".."))
    src, out = os.path.join(root, "examples", "partitioned_step.cpp"), os.path.join(root, "examples", "partitioned_step")
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(OUT)):
        return out
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(root, "include"), src,
           "-L", PKG, "-lmgn_hip", "-Wl,-rpath,$ORIGIN/../gnn_amd", "-o", out]
    res = subprocess.run(cmd, cwd=root, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (example) failed:\n" + res.stdout + res.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_examples())
