"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every
symbol include/rusty_compression_amd.h declares, and the host mirror fails
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os

import pytest
import torch

import rusty_compression_amd as rc
from rusty_compression_amd import _lib


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    assert os.path.dirname(_lib.LIB_PATH) == os.path.dirname(os.path.abspath(rc.__file__))


def test_every_declared_symbol_is_exported():
    lib = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) >= 60
    missing = [s for s in declared if not hasattr(lib, s)]
    assert missing == []
    assert lib.rc_abi_version() == 1


def test_both_precisions_are_declared_for_every_typed_entry_point():
    declared = set(_lib.declared_symbols())
    for s in declared:
        if s.endswith("_f64"):
            assert s[:-4] + "_f32" in declared, s


def test_struct_layout_matches_the_header():
    assert ctypes.sizeof(_lib.rc_matrix) == 40
    assert ctypes.sizeof(_lib.rc_rsvd_id_out) == 7 * 40 + 2 * 8
    from rusty_compression_amd.operator import rc_operator

    assert ctypes.sizeof(rc_operator) == 5 * 8 and rc_operator.matmat.offset == 16 and rc_operator.user.offset == 32


def test_operator_entry_points_reject_bad_tables_without_a_gpu():
    """rc_*_op_*: the operator-generic samplers of the reference (src/random_sampling.rs:102, :130, :222) at the C ABI."""
    lib = _lib.lib()
    for name in ("rc_sample_range_by_rank_op", "rc_sample_range_power_iteration_op", "rc_sample_range_adaptive_op", "rc_qr_from_range_estimate_op",
                 "rc_svd_from_range_estimate_op", "rc_rsvd_id_op"):
        assert hasattr(lib, name + "_f64") and hasattr(lib, name + "_f32"), name
    none = _lib.mat(None)
    assert lib.rc_sample_range_by_rank_op_f64(ctypes.c_void_p(None), None, ctypes.c_int64(1), ctypes.c_int64(1), none, ctypes.c_uint64(0), none) == _lib.RC_INVALID_ARGUMENT
    assert lib.rc_get_stream(ctypes.c_void_p(None), None) == _lib.RC_INVALID_ARGUMENT


def test_null_context_is_rejected_without_touching_a_gpu():
    lib = _lib.lib()
    assert lib.rc_synchronize(ctypes.c_void_p(None)) == _lib.RC_INVALID_ARGUMENT
    assert lib.rc_destroy(ctypes.c_void_p(None)) == _lib.RC_OK


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_a_loud_failure_not_a_cpu_fallback():
    import numpy as np

    with pytest.raises(rc.HipRuntimeError):
        rc.QR.compute_from(np.eye(4))
    with pytest.raises(rc.HipRuntimeError):
        rc.sample_range_by_rank(np.eye(8), 2, 1, np.ones((8, 3)))
    h = ctypes.c_void_p()
    assert _lib.lib().rc_create(ctypes.byref(h), 0, None) == _lib.RC_RUNTIME_ERROR


def test_mirror_exposes_the_reference_surface():
    # reference src/lib.rs:90-102 re-exports
    for name in ("QR", "LQ", "SVD", "ColumnID", "RowID", "TwoSidedID", "CompressionType", "random_gaussian",
                 "apply_permutation", "invert_permutation_vector", "rel_diff_fro", "rel_diff_l2",
                 "sample_range_by_rank", "sample_range_power_iteration", "sample_range_adaptive", "max_col_norm"):
        assert hasattr(rc, name), name
    for cls, methods in ((rc.QR, ("compute_from", "compress", "to_mat", "column_id", "compute_from_range_estimate", "nrows", "ncols", "rank")),
                         (rc.LQ, ("compute_from", "compress", "to_mat", "row_id")),
                         (rc.SVD, ("compute_from", "compress", "to_mat", "to_qr", "compute_from_range_estimate")),
                         (rc.ColumnID, ("two_sided_id", "to_mat", "dot", "new")),
                         (rc.RowID, ("two_sided_id", "to_mat", "dot", "new")),
                         (rc.TwoSidedID, ("to_mat", "dot", "new"))):
        for m in methods:
            assert hasattr(cls, m), (cls.__name__, m)


def build_cpp_mirror_examples(out_dir, source="mirror_examples.cpp"):
    """g++ build of tests/cpp/<source> against include/rusty_compression.hpp and the in-tree library."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(out_dir), os.path.splitext(source)[0])
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-pthread", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
           os.path.join(root, "tests", "cpp", source), "-o", exe, "-L" + libdir, "-lrusty_compression_amd",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    return exe


def test_cpp_mirror_header_compiles_and_links_against_the_c_abi(tmp_path):
    exe = build_cpp_mirror_examples(tmp_path)
    assert os.path.exists(exe)
    # the compiled twin of the Rust crate's test ports (bindings/rust/tests/reference_tests.rs)
    assert os.path.exists(build_cpp_mirror_examples(tmp_path, "reference_tests.cpp"))


def test_rust_binding_is_generated_from_the_header_and_covers_the_reference_surface():
    """bindings/rust cannot be compiled here (no Rust toolchain): what CAN be checked is that src/ffi.rs is the generator's
    output for the current header (every exported function, every scalar type), that every trait / type the reference
    re-exports (src/lib.rs:90-102) exists in the crate, and that all 89 reference tests are ported by name."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    ffi = open(os.path.join(root, "bindings", "rust", "src", "ffi.rs")).read()
    assert set(re.findall(r"pub fn (rc_\w+)", ffi)) == set(_lib.declared_symbols())
    src = "".join(open(os.path.join(root, "bindings", "rust", "src", f)).read() for f in sorted(os.listdir(os.path.join(root, "bindings", "rust", "src"))))
    for item in ("pub trait QRTraits", "pub trait LQTraits", "pub trait ColumnIDTraits", "pub trait RowIDTraits", "pub trait TwoSidedIDTraits",
                 "pub trait SVDTraits", "pub trait RandomMatrix", "pub trait RelDiff", "pub trait SampleRange<", "pub trait SampleRangePowerIteration<",
                 "pub trait AdaptiveSampling<", "pub trait MaxColNorm<", "pub trait Apply<", "pub trait MatVec", "pub trait MatMat", "pub trait ConjMatVec",
                 "pub trait ConjMatMat", "pub trait ApplyPermutationToMatrix", "pub trait ApplyPermutationToVector", "pub fn invert_permutation_vector",
                 "pub enum MatrixPermutationMode", "pub enum VectorPermutationMode", "pub enum CompressionType", "pub enum RustyCompressionError",
                 "pub struct QR<", "pub struct LQ<", "pub struct SVD<", "pub struct ColumnID<", "pub struct RowID<", "pub struct TwoSidedID<"):
        assert item in src, item
    tests = open(os.path.join(root, "bindings", "rust", "tests", "reference_tests.rs")).read()
    names = set(re.findall(r"^\s+((?:test_|pivoted_)\w+): ", tests, flags=re.M)) | set(re.findall(r"fn (test_\w+)\(\)", tests))
    # the reference's 89 unit tests + one test of the operator-generic range finders (the reference has none of those)
    assert len(names) == 90 and "test_range_finders_over_a_matvec_only_operator" in names, len(names)
    for item in ("pub trait DeviceOperator", "pub trait OpScalar", "pub fn with_table", "pub struct HostMatMat", "pub struct HostConjMatMat"):
        assert item in src, item
    # the C++ twin runs the real-scalar half of them (the mirror header is instantiated for f32 / f64)
    cpp = open(os.path.join(root, "tests", "cpp", "reference_tests.cpp")).read()
    for stem in ("pivoted_qr_test_", "pivoted_lq_test_", "test_qr_compression_by_rank_", "test_qr_compression_by_tol_", "test_col_id_compression_by_tol_",
                 "test_row_id_compression_by_tol_", "test_svd_to_qr_", "test_svd_compression_by_rank_", "test_svd_compression_by_tol_",
                 "test_two_sided_from_col_id_compression_by_tol_", "test_two_sided_from_row_id_compression_by_tol_", "test_matrix_permutation",
                 "test_vector_permutaiton"):
        assert stem in cpp, stem
