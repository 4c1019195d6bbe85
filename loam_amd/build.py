"""Builds libloamx.so (the HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m loam_amd.build            # build if sources are newer than the library
    python -m loam_amd.build --force
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libloamx.so")
SOURCES = ["loamx_api.hip", "extract_kernels.hip", "register_kernels.hip", "synth_kernels.hip"]
HEADERS = ["loamx_internal.h", "extract_math.h", "select_rows.h", "reg_math.h", "synth.h", os.path.join("..", "..", "include", "loamx.h")]
# -ffp-contract=off: the reference's x86-64 build has no FMA; curvature bits must match
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libloamx.so cannot be built (there is no CPU fallback)")


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources + the C ABI header: profiles/ files are stamped with it and
    bench.py only quotes PMC numbers that were measured on the sources it runs."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(f).encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _obj_path(src):
    return os.path.join(LIB_DIR, "obj", os.path.splitext(src)[0] + ".o")


def build(force=False, verbose=False):
    """One object per .hip source (compiled in parallel, only when the source or a header is newer), then the link.
    Safe to call from several processes at once (the ranks of a multi-GPU launch on a fresh checkout): one holds the
    lock and builds, the others wait and find the library up to date; the library appears by an atomic rename."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    os.makedirs(os.path.join(LIB_DIR, "obj"), exist_ok=True)
    with open(os.path.join(LIB_DIR, "obj", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # somebody else built it while this process waited
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    from concurrent.futures import ThreadPoolExecutor
    extra = os.environ.get("LOAMX_EXTRA_FLAGS", "").split()  # experiments only, e.g. -DLOAMX_ASSOC_WAVES=6
    flags_tag = os.path.join(LIB_DIR, "obj", "flags.txt")
    flags_now = " ".join(FLAGS + extra)
    same_flags = os.path.exists(flags_tag) and open(flags_tag).read() == flags_now
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_time = max(hdr_time, os.path.getmtime(os.path.abspath(__file__)))

    def compile_one(src):
        obj, path = _obj_path(src), os.path.join(CSRC, src)
        if not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_time):
            return
        cmd = [_hipcc()] + [f for f in FLAGS if f != "-shared"] + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        list(ex.map(compile_one, SOURCES))
    open(flags_tag, "w").write(flags_now)
    # (librccl — the gather of the result records in multi-GPU batch mode — is opened at run time by the loamx_comm_*
    # entry points; the rpath lets that dlopen find ROCm's copy when the process has none mapped yet)
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = ([_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + [_obj_path(s) for s in SOURCES] +
           ["-ldl", "-Wl,-rpath," + rocm_lib])
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


PY_DIR = os.path.join(HERE, "python")
PY_SRC = os.path.join(PY_DIR, "loam_bindings.cpp")


def pybind_path():
    import sysconfig
    return os.path.join(PY_DIR, "loam", "loam_python" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force=False, verbose=False):
    """Builds the pybind11 module loam_amd/python/loam/loam_python*.so (host code only; it calls the C ABI)."""
    import sysconfig
    import pybind11
    out = pybind_path()
    inc = os.path.join(os.path.dirname(HERE), "include")
    deps = [PY_SRC] + [os.path.join(inc, "loam", f) for f in os.listdir(os.path.join(inc, "loam"))] + [os.path.join(inc, "loamx.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    build(force=False, verbose=verbose)
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-I", inc, "-I", pybind11.get_include(),
           "-I", sysconfig.get_paths()["include"], PY_SRC, "-o", out, "-L", LIB_DIR, "-lloamx",
           "-Wl,-rpath,$ORIGIN/../../lib", f"-Wl,-rpath,{LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_pybind(force="--force" in sys.argv, verbose=True))
