"""Build libasrk.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container; the resulting
``csrc/libasrk.so`` is git-ignored but travels to the GPU box with the repo snapshot.
Only re-compiles translation units whose sources are newer than their object files.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libasrk.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
          "-ffp-contract=off"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(ROOT, "include", "asrk.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(spath)
            and os.path.getmtime(obj) >= _deps_mtime()):
        return obj, False
    cmd = [HIPCC] + CFLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[asrk] built", LIB)
    elif verbose:
        print("[asrk] up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
