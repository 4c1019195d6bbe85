"""CPU-only: the host builds of the kernels' math headers (tests/hostcheck) and of the oracle under AddressSanitizer +
UndefinedBehaviorSanitizer (SURVEY section 5 "race detection / sanitizers"). `make -C tests/hostcheck san`
and `make -C oracle san` build the instrumented libraries; the CPU tests that exercise them run once more in a child
process with libasan preloaded, and any "runtime error" (UBSan) or "AddressSanitizer" line fails this test.
Sanitizers run on the CPU build only: GPU ASan / XNACK are not available on the MI355X pool."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SUITE = ["test_hostcheck.py", "test_oracle_kats.py", "test_oracle_crosschecks.py", "test_golden_fixtures.py"]


def test_host_math_and_oracle_are_clean_under_asan_and_ubsan():
    if os.environ.get("LOAMX_IN_SANITIZER_CHILD"):
        return  # (this file is not part of SUITE; belt and braces against recursion)
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostcheck"), "san"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    # the runtime of the SAME compiler the `san` targets use ($(CXX), make's default g++): a preloaded runtime of another
    # compiler or version does not match the instrumented libraries. UBSan findings abort the child
    # (-fno-sanitize-recover=undefined in SANFLAGS), so they fail it by exit code as well as by the grep below.
    cxx = os.environ.get("CXX", "g++")
    runtime = "libclang_rt.asan-x86_64.so" if "clang" in os.path.basename(cxx) else "libasan.so"
    asan = subprocess.check_output([cxx, "-print-file-name=" + runtime], text=True).strip()
    assert os.path.isabs(asan) and os.path.exists(asan), runtime + " not found next to " + cxx
    env = dict(os.environ, LD_PRELOAD=asan, LOAMX_IN_SANITIZER_CHILD="1",
               HOSTCHECK_LIB=os.path.join(HERE, "hostcheck", "libhostcheck_san.so"),
               ORACLE_LIB=os.path.join(ROOT, "oracle", "libloam_oracle_san.so"),
               # Python itself is not instrumented: its arena allocations are not leaks of ours
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + [os.path.join(HERE, f) for f in SUITE],
                         env=env, capture_output=True, text=True, timeout=1500)
    text = out.stdout + out.stderr
    bad = [ln for ln in text.splitlines() if "runtime error" in ln or "AddressSanitizer" in ln]
    assert not bad, "\n".join(bad[:40])
    assert out.returncode == 0, text[-4000:]
    assert " passed" in out.stdout
