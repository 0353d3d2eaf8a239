"""In-tree build of the native pieces (no JIT cache: the .so files travel with the tree).

  librtr_hip.so   HIP kernels + C ABI, hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
  librtr_hip_test.so  device unit kernels of the parity tests (include/rtr_hip_test.h); links against librtr_hip.so
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


def source_hash():
    """sha256 over what librtr_hip.so is built from (csrc/, include/, the compile flags): ties a committed counter file
    (profiles/rNN_counts.json) to the library it was measured on -- bench.py marks its roofline stale when they differ."""
    import hashlib
    h = hashlib.sha256()
    for d in (CSRC, INC):
        for name in sorted(os.listdir(d)):
            if name.endswith((".h", ".hip")) and "test" not in name:  # the test library's sources are not the product's
                h.update(name.encode())
                with open(os.path.join(d, name), "rb") as f:
                    h.update(f.read())
    h.update(" ".join(HIP_FLAGS).encode())
    return h.hexdigest()[:16]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]
# translation units of librtr_hip.so: (object name, source, extra defines); compiled in parallel
HIP_UNITS = [("capi", "rtr_capi.hip", []), ("mega_mis", "rtr_mega.hip", ["-DRTR_MEGA_GROUP=0"]),
             ("mega_rr_path", "rtr_mega.hip", ["-DRTR_MEGA_GROUP=1"]), ("mega_pbr_nee", "rtr_mega.hip", ["-DRTR_MEGA_GROUP=2"]),
             ("wavefront", "rtr_wavefront.hip", []), ("test", "rtr_test.hip", [])]


def build_hip(force=False, verbose=False, extra_flags=()):
    """Exact-arithmetic build: -ffp-contract=off keeps the reference's mul/add sequence.  The units are
    compiled side by side (hipcc cross-compiles gfx950 without a GPU) and linked into one library."""
    from concurrent.futures import ThreadPoolExecutor
    out = os.path.join(HERE, "librtr_hip.so")
    sources = _sources(CSRC, (".hip", ".h"))
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    # the flags the objects on disk were compiled with: a library left behind by an experiment
    # (tools/build_hip.sh -DSOMETHING) is never mistaken for the product build
    stamp = os.path.join(HERE, "librtr_hip.flags")  # next to the library: it travels with it
    flags_now = " ".join(HIP_FLAGS + list(extra_flags))
    try:
        flags_then = open(stamp).read()
    except OSError:
        flags_then = None
    if flags_then != flags_now:
        force = True
    if not force and not _newer(out, sources):
        return out

    def compile_unit(unit):
        name, src, defs = unit
        obj = os.path.join(objdir, name + ".o")
        if not force and not _newer(obj, sources):
            return obj, ""
        cmd = [hipcc()] + HIP_FLAGS + list(extra_flags) + defs + ["-I" + INC, "-I" + CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr.decode()[-4000:]))
        return obj, r.stderr.decode()

    with ThreadPoolExecutor(max_workers=min(len(HIP_UNITS), os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_unit, HIP_UNITS))
    product = [o for (name, _, _), (o, _) in zip(HIP_UNITS, results) if name != "test"]
    test_objs = [o for (name, _, _), (o, _) in zip(HIP_UNITS, results) if name == "test"]
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + product + ["-o", out], check=True)
    # the device unit kernels of the parity tests: a library of their own next to the product (include/rtr_hip_test.h)
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + test_objs +
                   ["-L" + HERE, "-lrtr_hip", "-Wl,-rpath,$ORIGIN", "-o", os.path.join(HERE, "librtr_hip_test.so")], check=True)
    with open(stamp, "w") as f:
        f.write(flags_now)
    if verbose:
        return out, "".join(log for _, log in results)
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
