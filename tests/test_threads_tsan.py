"""The threading contract of the C ABI (include/mpshuffle.h) under ThreadSanitizer, and the absence of hidden inputs
(item 3).  The reference's trait members are associated functions without `self` or global state
[REF barnett-smart-card-protocol/src/lib.rs:74-197]: any number of host threads may call them at once.  CPU tests: the engine's kernel
bodies run as plain loops (tools/hostemu, a development aid that is never shipped), compiled WITHOUT OpenMP for this test so that the
sanitizer sees every access; the same scenarios run on the HIP library in tests/test_gpu_round6.py."""
import os
import re
import subprocess

from conftest import ROOT

CSRC = os.path.join(ROOT, "mental-poker_amd", "csrc")
EMU = os.path.join(ROOT, "tools", "hostemu")


def test_no_environment_variable_steers_the_library():
    """a drop-in for [REF src/lib.rs:41-198] has no hidden inputs: no getenv / environ in the engine's sources or in the symbols the
    built library imports"""
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hpp", ".hip")):
            text = open(os.path.join(CSRC, f)).read()
            assert not re.search(r"\bgetenv\b|\benviron\b|secure_getenv", text), f
    assert "getenv" not in open(os.path.join(ROOT, "include", "mpshuffle.h")).read().replace("no environment", "")
    lib = os.path.join(ROOT, "mental-poker_amd", "libmpshuffle.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", "--undefined-only", lib], stdout=subprocess.PIPE, check=True).stdout.decode()
        assert not re.search(r"\b(secure_)?getenv\b|\benviron\b", syms)


def test_host_threads_under_thread_sanitizer(tmp_path):
    """four host threads with a context each, then two threads (and a third calling setters) on ONE table: same bytes and status words as
    the single-threaded run, and ThreadSanitizer reports nothing"""
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU])          # the other curves' objects (not instrumented, not executed here)
    objs = []
    flags = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=thread", "-x", "c++", "-include", os.path.join(EMU, "rt.hpp"), "-I", EMU,
             "-Wno-unknown-pragmas"]
    procs = []
    for unit in ("capi", "curve_stark", "curve_stark_msm"):
        obj = str(tmp_path / (unit + ".o"))
        procs.append(subprocess.Popen(flags + ["-c", os.path.join(CSRC, unit + ".hip"), "-o", obj]))
        objs.append(obj)
    for p in procs:
        assert p.wait() == 0
    others = [os.path.join(EMU, "_obj", u + ".o") for u in ("curve_bn254", "curve_secp256k1", "curve_bls12_377", "curve_bn254_msm",
                                                           "curve_secp256k1_msm", "curve_bls12_377_msm")]
    exe = str(tmp_path / "threads_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "threads_tsan.cpp")] + objs + others + ["-fopenmp", "-pthread", "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1", OMP_NUM_THREADS="1"))
    err = out.stderr.decode()
    assert out.returncode == 0, err[-4000:]
    assert "ThreadSanitizer" not in err, err[:6000]
    assert "threads ok" in out.stdout.decode()
