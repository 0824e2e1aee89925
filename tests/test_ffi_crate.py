"""The Rust crate of the drop-in (provers/hip/driver) cannot be compiled in this image (no
rustc / cargo): what can be checked is that its FFI declarations are exactly the C header's
(tools/check_ffi.py), that the crate files are complete (no elisions) and that the registration
patch targets the lines of the reference it claims to."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "provers", "hip", "driver")


def test_ffi_rs_matches_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ffi.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "ffi.rs matches include/raiko_hip.h" in r.stdout


def test_checker_sees_a_drift(tmp_path):
    """the check is not vacuous: a changed argument type in the header is reported"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_ffi
    src = open(check_ffi.HEADER).read()
    gen = check_ffi.emit(src)
    drift = check_ffi.emit(src.replace("int rk_sync(rk_ctx* ctx);", "int rk_sync(rk_ctx* ctx, int flags);"))
    assert check_ffi.normalise(gen) != check_ffi.normalise(drift)
    assert "pub fn rk_prove_session(opts: *const rk_session_opts, segs: *const rk_segment, n: usize, h_seals: *const *mut u32," in gen
    assert "pub hooks: *const rk_circuit_hooks," in gen and "pub poly_ext: Option<rk_poly_ext_fn>," in gen


def test_crate_files_are_whole():
    for rel in ("Cargo.toml", "build.rs", "src/ffi.rs", "src/lib.rs", "src/hal.rs"):
        txt = open(os.path.join(CRATE, rel)).read()
        assert "same pattern" not in txt and "// ..." not in txt and "todo!()" not in txt and "unimplemented!()" not in txt, rel
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    assert "impl Prover for HipProver" in lib and "rk_circuit_hooks" in lib
    # the session entry in its streaming form, with back-pressure (a block's witnesses do not fit in memory at once)
    assert "rk_stream_open(" in lib and "rk_stream_submit(" in lib and "rk_stream_wait(" in lib and "rk_stream_close(" in lib
    assert "enable_profiler(" in lib                     # `profile: true` of the request (bonsai.rs:252-255)
    hal = open(os.path.join(CRATE, "src", "hal.rs")).read()
    assert "impl Hal for HipHal" in hal
    # every Hal operator the header offers is bound in hal.rs
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    hal_ops = [n for n in re.findall(r"pub fn (rk_[a-z0-9_]+)\(", ffi)
               if n.split("rk_")[1] in ("batch_interpolate_ntt", "zk_shift", "batch_expand_into_evaluate_ntt", "batch_bit_reverse",
                                         "hash_rows", "hash_fold", "batch_evaluate_any", "mix_poly_coeffs", "eltwise_add_elem",
                                         "eltwise_sum_extelem", "eltwise_copy_elem", "eltwise_zeroize_elem", "fri_fold",
                                         "gather_sample", "prefix_products", "scatter")]
    assert len(hal_ops) == 16
    for n in hal_ops:
        assert n + "(" in hal, n


def test_registration_patch_matches_the_reference_lines():
    ref = "/root/reference/core/src/interfaces.rs"
    if not os.path.exists(ref):
        import pytest
        pytest.skip("reference tree not present (GPU box)")
    src = open(ref).read()
    patch = open(os.path.join(ROOT, "provers", "hip", "patches", "core-interfaces.patch")).read()
    removed = [l[1:].strip() for l in patch.splitlines() if l.startswith("-") and not l.startswith("---")]
    for l in removed:
        assert l in src, l
    context = [l[1:].strip() for l in patch.splitlines() if l.startswith(" ") and l.strip()]
    for l in context:
        assert l in src, l
