"""Build libdsvgp_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build().

Incremental per translation unit: every object carries a stamp (``<name>.o.stamp``) with the SHA-256 of its source, the
shared headers and the compile flags; a source is recompiled when its stamp does not match, the library is relinked when
any object changed or is newer than it.  ``build()`` reports what it compiled and what it found up to date, so a log of
the call shows whether the compiler ran.  ``force=True`` (``--force``) recompiles everything.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
LIB = os.path.join(HERE, "libdsvgp_hip.so")
SOURCES = ["gemm.hip", "gemm64.hip", "gemm32.hip", "gemm3b.hip", "assemble.hip", "assemble64.hip", "elbo.hip", "potrf.hip", "ciq.hip", "step.hip", "api.hip"]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _headers():
    hs = [os.path.join(INCLUDE, "dsvgp.h")]
    hs += sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return hs


def _stamp(src):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for path in [os.path.join(CSRC, src)] + _headers():
        h.update(path.encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(src):
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    stamp = obj + ".stamp"
    if not os.path.exists(obj) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != _stamp(src)


def needs_build():
    return not os.path.exists(LIB) or any(_stale(s) for s in SOURCES)


def build(force=False, verbose=True):
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    todo = [s for s in SOURCES if force or _stale(s)]
    procs = []
    for src in todo:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd)))
    for src, obj, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
        with open(obj + ".stamp", "w") as f:
            f.write(_stamp(src))
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    relink = bool(todo) or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if relink:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [
            "-L", os.path.join(ROCM, "lib"), "-lrocsolver", "-lrocblas", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if verbose:
        print("build_ext: compiled %d of %d HIP sources for gfx950 (%s)%s; %s"
              % (len(todo), len(SOURCES), ", ".join(todo) if todo else "all objects match their source stamps",
                 "" if todo else " -- nothing to do", "linked " + LIB if relink else "library up to date"), flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
