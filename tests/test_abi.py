"""The C-ABI library loads on a CPU-only box and exports every symbol include/raiko_hip.h declares
(no compute calls: there is no GPU here), and refuses to pretend otherwise."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "raiko_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    from raiko_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "library does not export " + n
        assert n in _lib.SYMBOLS, "no ctypes prototype for " + n
    assert sorted(_lib.SYMBOLS) == names
    assert lib.rk_abi_version() == 4
    assert lib.rk_strerror(-1) == b"invalid argument"


def test_no_gpu_means_loud_failure_not_fallback():
    from raiko_amd import _lib
    from raiko_amd.hal import HipHal
    lib = _lib.load()
    n = C.c_int(-1)
    st = lib.rk_device_count(C.byref(n))
    if st == 0 and n.value > 0:
        pytest.skip("a GPU is visible: this check is for CPU-only boxes")
    with pytest.raises(_lib.HipLibraryError):
        HipHal(0)
    ctx = C.c_void_p()
    assert lib.rk_ctx_create(0, None, C.byref(ctx)) == -4  # RK_ERR_NODEVICE
    assert not ctx.value
    # the session entry point and the Prover mirror fail the same way: no seal comes out of a CPU
    from raiko_amd.hal import prove_session
    from raiko_amd.segment import synthetic_segment
    segs = [synthetic_segment(5, (2, 2, 3), seed=s) for s in (1, 2)]
    with pytest.raises(_lib.RkError) as ei:
        prove_session(segs, inflight=2)
    assert ei.value.status == -4 and ei.value.segment == -1
    assert lib.rk_strerror(-7) == b"a produced seal failed verification"
    opts = _lib.RkSessionOpts(device=0, inflight=0, upload_ahead=0, verify=1)
    assert lib.rk_prove_session(C.byref(opts), None, 0, None, None, None, None) == -1  # inflight out of range
    # device-list argument checks happen before any GPU is touched
    two = (C.c_int * 2)(0, 0)
    opts = _lib.RkSessionOpts(device=0, inflight=1, upload_ahead=0, verify=0, devices=two, n_devices=2)
    assert lib.rk_prove_session(C.byref(opts), None, 0, None, None, None, None) == -1  # duplicate GPU
    opts = _lib.RkSessionOpts(device=0, inflight=1, upload_ahead=0, verify=0, devices=None, n_devices=2)
    assert lib.rk_prove_session(C.byref(opts), None, 0, None, None, None, None) == -1  # count without a list
    opts = _lib.RkSessionOpts(device=0, inflight=1, upload_ahead=0, verify=0, devices=two, n_devices=65)
    assert lib.rk_prove_session(C.byref(opts), None, 0, None, None, None, None) == -1
    with pytest.raises(_lib.RkError) as ei:
        prove_session(segs, inflight=2, devices=[0, 1])
    assert ei.value.status == -4
    assert lib.rk_session_release() == 0


def test_product_never_imports_the_oracle():
    """the product package must not reference oracle/ or the tests' oracle binding"""
    pkg = os.path.join(ROOT, "raiko_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("(oracle/", "(x/"), f


def _build_demo(tmp_path, name="session_demo"):
    import subprocess
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-o", exe,
                           "-L", os.path.join(ROOT, "raiko_amd"), "-lraiko_hip"])
    return exe


def test_plain_c_caller_compiles_against_the_header(tmp_path):
    """include/raiko_hip.h is C (not C++) and the library links from a C program: the boundary a
    Rust / cgo binding sees.  Without a GPU the demo must stop with the library's error, not a seal."""
    import subprocess
    exe = _build_demo(tmp_path)
    from raiko_amd import _lib
    n = C.c_int(-1)
    if _lib.load().rk_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is visible: the run is covered by the gpu-marked test")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raiko_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, "2", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no usable GPU" in r.stderr
    # the constraint-list demo: the program is compiled on the host before any GPU is needed
    exe2 = _build_demo(tmp_path, "program_demo")
    r = subprocess.run([exe2, "6"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no usable GPU" in r.stderr and "19 steps, 11 ops" in r.stdout
    exe3 = _build_demo(tmp_path, "pcs_demo")
    r = subprocess.run([exe3], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no usable GPU" in r.stderr
    exe4 = _build_demo(tmp_path, "p3_demo")          # the AIR front end is host code: it runs before the GPU is looked for
    r = subprocess.run([exe4], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no usable GPU" in r.stderr and "5 constraints, degree 2, 1 quotient chunk(s)" in r.stdout


@pytest.mark.gpu
def test_plain_c_caller_proves_a_session(tmp_path):
    import subprocess
    exe = _build_demo(tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raiko_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, "5", "12"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "5 segments of 2^12 cycles proven and verified" in r.stdout
    # the same through the device-list form of rk_session_opts (a one-entry list on this box)
    r = subprocess.run([exe, "4", "10", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "4 segments of 2^10 cycles proven and verified" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.gpu
def test_plain_c_caller_proves_from_a_constraint_list(tmp_path):
    """examples/program_demo.c: a circuit given as data (rk_program) proven and verified from C; a wrong
    witness cell is caught by the constraint identity only"""
    import subprocess
    exe = _build_demo(tmp_path, "program_demo")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raiko_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for po2 in ("4", "11"):
        r = subprocess.run([exe, po2], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert "verifier says 0" in r.stdout and "with it 70" in r.stdout


@pytest.mark.gpu
def test_plain_c_caller_runs_the_pcs_steps(tmp_path):
    """examples/pcs_demo.c: commit -> open -> reduce rows -> folds from C under SP1's parameter set; the folded
    reduced opening is a constant (Plonky3's assertion at the end of the commit phase)"""
    import subprocess
    exe = _build_demo(tmp_path, "pcs_demo")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raiko_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for args in (["5", "3"], ["14", "20"], ["18", "8"]):
        r = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert "constant" in r.stdout and "NOT constant" not in r.stdout and "grind(12 bits) -> witness" in r.stdout


@pytest.mark.gpu
def test_plain_c_caller_proves_an_air(tmp_path):
    """examples/p3_demo.c: the Fibonacci AIR as a step list through rk_air_create / rk_p3_prove / rk_p3_verify from C under
    SP1's parameter set; a wrong public value fails the constraint identity, a changed proof word the openings"""
    import subprocess
    exe = _build_demo(tmp_path, "p3_demo")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raiko_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for k in ("3", "12", "17"):
        r = subprocess.run([exe, k], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert "verifier: 0" in r.stdout and "wrong public value: verifier 3" in r.stdout and "fibonacci proven and verified" in r.stdout
        # two tables tied by a lookup, the permutation constraints written by the library (rk_air_create_lookup with ext_w)
        assert "squares, proof of" in r.stdout and "one multiplicity off by one: verifier 8" in r.stdout and "lookup proven and verified" in r.stdout
