"""Builds orc_format_amd/liborcd.so (HIP kernels for gfx950 + C ABI + C++ host reader) in-tree."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liborcd.so")
SOURCES = ["rlev2.hip", "byte_rle.hip", "strings.hip", "scatter.hip", "inflate.hip", "orcd_api.hip", "orcd_host.cpp"]
HEADERS = ["orcd_internal.h", "orcd_column_internal.h", "seg_scan.h", os.path.join("..", "..", "include", "orcd.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compiles every translation unit for gfx950 and links the shared library."""
    if not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter"]
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".", "_") + ".o")
        cmd = [hipcc] + common + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".hip"):
            cmd[1:1] = ["--offload-arch=gfx950"]
        else:
            cmd[1:1] = ["-x", "c++"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("compiling %s failed:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [
        "-pthread", "-lz", "-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
