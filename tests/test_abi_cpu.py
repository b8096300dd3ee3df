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


def build_cpp_mirror_examples(out_dir):
    """g++ build of tests/cpp/mirror_examples.cpp against include/rusty_compression.hpp and the in-tree library."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(out_dir), "mirror_examples")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
           os.path.join(root, "tests", "cpp", "mirror_examples.cpp"), "-o", exe, "-L" + libdir, "-lrusty_compression_amd",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    return exe


def test_cpp_mirror_header_compiles_and_links_against_the_c_abi(tmp_path):
    exe = build_cpp_mirror_examples(tmp_path)
    assert os.path.exists(exe)
