"""In-tree build of the native pieces (no JIT cache: the .so files travel with the tree).

  librtr_hip.so   HIP kernels + C ABI, hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
  librtr_host.so  C++ host layer: the reference's scene-description API + flattening (g++)
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
INC = os.path.join(ROOT, "include")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d, exts):
    out = []
    for base, _, files in os.walk(d):
        out += [os.path.join(base, f) for f in files if f.endswith(exts)]
    return out + [os.path.join(INC, f) for f in os.listdir(INC)]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_hip(force=False, verbose=False):
    """Exact-arithmetic build: -ffp-contract=off keeps the reference's mul/add sequence."""
    out = os.path.join(HERE, "librtr_hip.so")
    if not force and not _newer(out, _sources(CSRC, (".hip", ".h"))):
        return out
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wno-unused-value", "-I" + INC, "-I" + CSRC, os.path.join(CSRC, "rtr_capi.hip"), "-o", out]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return out


def build_host(force=False):
    src = os.path.join(HOST, "rtr_host.cpp")
    if not os.path.exists(src):
        return None
    out = os.path.join(HERE, "librtr_host.so")
    if not force and not _newer(out, _sources(HOST, (".cpp", ".h"))):
        return out
    cmd = ["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-I" + INC, "-I" + HOST, src, "-o", out, "-lz"]
    subprocess.run(cmd, check=True)
    return out


def build_cli(force=False):
    """Headless C++ driver (host/rtr_cli.cpp) linked against librtr_hip.so: the C++ face of the boundary."""
    src = os.path.join(HOST, "rtr_cli.cpp")
    out = os.path.join(HERE, "rtr_cli")
    lib = os.path.join(HERE, "librtr_hip.so")
    if not os.path.exists(src) or not os.path.exists(lib):
        return None
    if not force and not _newer(out, _sources(HOST, (".cpp", ".h")) + [lib]):
        return out
    cmd = ["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-I" + INC, "-I" + HOST, src,
           os.path.join(HOST, "rtr_host.cpp"), "-L" + HERE, "-lrtr_hip", "-Wl,-rpath,$ORIGIN",
           "-Wl,--allow-shlib-undefined", "-lz", "-o", out]
    subprocess.run(cmd, check=True)
    return out


def build_all(force=False):
    return {"hip": build_hip(force), "host": build_host(force), "cli": build_cli(force)}
