//! Raw declarations of include/rusty_compression_amd.h (f64 shown in full; the `_f32`
//! functions have identical signatures with `f32` scalars).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub type rc_status = i32;
pub const RC_OK: rc_status = 0;
pub const RC_LINALG_ERROR: rc_status = 1;
pub const RC_COMPRESSION_ERROR: rc_status = 2;
pub const RC_LAYOUT_ERROR: rc_status = 3;
pub const RC_PIVOTED_QR_ERROR: rc_status = 4;
pub const RC_INVALID_ARGUMENT: rc_status = 5;
pub const RC_RUNTIME_ERROR: rc_status = 6;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rc_matrix {
    pub data: *mut c_void,
    pub rows: i64,
    pub cols: i64,
    pub row_stride: i64,
    pub col_stride: i64,
}

#[repr(C)]
pub struct rc_context {
    _private: [u8; 0],
}

extern "C" {
    pub fn rc_abi_version() -> i32;
    pub fn rc_create(ctx: *mut *mut rc_context, device: i32, hip_stream: *mut c_void) -> rc_status;
    pub fn rc_destroy(ctx: *mut rc_context) -> rc_status;
    pub fn rc_synchronize(ctx: *mut rc_context) -> rc_status;
    pub fn rc_set_stream(ctx: *mut rc_context, hip_stream: *mut c_void) -> rc_status;
    pub fn rc_stream_create(device: i32, hip_stream: *mut *mut c_void) -> rc_status;
    pub fn rc_stream_destroy(device: i32, hip_stream: *mut c_void) -> rc_status;
    /// options: 1 tall-skinny fast path, 2 lazy wide QRCP, 3 cooperative wide QRCP, 4 power iteration that really iterates
    pub fn rc_set_option(ctx: *mut rc_context, option: i32, value: i64) -> rc_status;
    pub fn rc_get_health(ctx: *mut rc_context, word: *mut i32) -> rc_status;
    pub fn rc_graph_begin_capture(ctx: *mut rc_context) -> rc_status;
    pub fn rc_graph_end_capture(ctx: *mut rc_context, graph_exec: *mut *mut c_void) -> rc_status;
    pub fn rc_graph_launch(ctx: *mut rc_context, graph_exec: *mut c_void) -> rc_status;
    pub fn rc_graph_destroy(ctx: *mut rc_context, graph_exec: *mut c_void) -> rc_status;
    pub fn rc_last_error_message(ctx: *const rc_context) -> *const c_char;
    pub fn rc_device_malloc(ctx: *mut rc_context, bytes: usize, ptr: *mut *mut c_void) -> rc_status;
    pub fn rc_device_free(ctx: *mut rc_context, ptr: *mut c_void) -> rc_status;
    pub fn rc_memcpy_h2d(ctx: *mut rc_context, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> rc_status;
    pub fn rc_memcpy_d2h(ctx: *mut rc_context, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> rc_status;

    pub fn rc_random_gaussian_f64(ctx: *mut rc_context, out: rc_matrix, seed: u64, offset: u64) -> rc_status;
    pub fn rc_matmat_f64(ctx: *mut rc_context, a: rc_matrix, x: rc_matrix, y: rc_matrix) -> rc_status;
    pub fn rc_conj_matmat_f64(ctx: *mut rc_context, a: rc_matrix, x: rc_matrix, y: rc_matrix) -> rc_status;
    pub fn rc_gemm_f64(ctx: *mut rc_context, trans_a: i32, trans_b: i32, alpha: f64, a: rc_matrix, b: rc_matrix, beta: f64, c: rc_matrix) -> rc_status;
    pub fn rc_rel_diff_fro_f64(ctx: *mut rc_context, first: rc_matrix, second: rc_matrix, out: *mut f64) -> rc_status;

    pub fn rc_invert_permutation(ctx: *mut rc_context, perm: *const i64, n: i64, inverse: *mut i64) -> rc_status;
    pub fn rc_apply_permutation_matrix_f64(ctx: *mut rc_context, mode: i32, input: rc_matrix, perm: *const i64, perm_len: i64, out: rc_matrix) -> rc_status;

    pub fn rc_pivoted_qr_f64(ctx: *mut rc_context, a: rc_matrix, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status;
    pub fn rc_pivoted_lq_f64(ctx: *mut rc_context, a: rc_matrix, l: rc_matrix, q: rc_matrix, ind: *mut i64) -> rc_status;
    pub fn rc_compute_svd_f64(ctx: *mut rc_context, a: rc_matrix, u: rc_matrix, s: *mut f64, vt: rc_matrix) -> rc_status;

    pub fn rc_rank_by_tolerance_f64(ctx: *mut rc_context, tri: rc_matrix, tol: f64, rank: *mut i64) -> rc_status;
    pub fn rc_qr_to_mat_f64(ctx: *mut rc_context, q: rc_matrix, r: rc_matrix, ind: *const i64, out: rc_matrix) -> rc_status;
    pub fn rc_qr_column_id_f64(ctx: *mut rc_context, q: rc_matrix, r: rc_matrix, ind: *const i64, c: rc_matrix, z: rc_matrix) -> rc_status;
    pub fn rc_lq_row_id_f64(ctx: *mut rc_context, l: rc_matrix, q: rc_matrix, ind: *const i64, x: rc_matrix, r_rows: rc_matrix) -> rc_status;
    pub fn rc_qr_from_range_estimate_f64(ctx: *mut rc_context, range: rc_matrix, a: rc_matrix, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status;
    pub fn rc_svd_from_range_estimate_f64(ctx: *mut rc_context, range: rc_matrix, a: rc_matrix, u: rc_matrix, s: *mut f64, vt: rc_matrix) -> rc_status;
    pub fn rc_column_id_two_sided_f64(ctx: *mut rc_context, c: rc_matrix, c_out: rc_matrix, x: rc_matrix, row_ind: *mut i64) -> rc_status;
    pub fn rc_row_id_two_sided_f64(ctx: *mut rc_context, r: rc_matrix, x: rc_matrix, r_out: rc_matrix, col_ind: *mut i64) -> rc_status;

    pub fn rc_max_col_norm_f64(ctx: *mut rc_context, y: rc_matrix, out: *mut f64) -> rc_status;
    pub fn rc_sample_range_by_rank_f64(ctx: *mut rc_context, a: rc_matrix, k: i64, p: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status;
    pub fn rc_sample_range_power_iteration_f64(ctx: *mut rc_context, a: rc_matrix, k: i64, p: i64, it_count: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status;
    pub fn rc_sample_range_adaptive_f64(ctx: *mut rc_context, a: rc_matrix, rel_tol: f64, sample_size: i64, omegas: rc_matrix, seed: u64,
                                        q_cap: rc_matrix, rank: *mut i64, hist_rank: *mut i64, hist_res: *mut f64, hist_cap: i64, hist_len: *mut i64) -> rc_status;
}
