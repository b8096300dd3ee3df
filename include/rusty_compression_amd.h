/* rusty_compression_amd.h -- C ABI of the MI355X-native randomized low-rank
 * compression engine (librusty_compression_amd.so).
 *
 * This is the drop-in boundary for the hot path of rusty-compression v0.1.1
 * (reference at /root/reference).  The reference has no FFI of its own: its
 * arithmetic crosses into Fortran LAPACK inside src/pivoted_qr.rs:138-173 and
 * through ndarray-linalg.  A maintainer swapping the hot path to the GPU binds
 * ONE level higher, at the crate's own trait methods, so every entry point
 * below names the reference interface it replaces (file:line).  The reference
 * side binding (Rust `extern "C"` block + trait impls) is in INTEGRATION.md
 * and bindings/rust/.
 *
 * Conventions
 *  - All matrix arguments are DEVICE memory described by `rc_matrix` (an
 *    ndarray-style strided view: element (i, j) lives at
 *    data[i*row_stride + j*col_stride], strides in elements).  Any layout is
 *    accepted (C order, Fortran order, transposed views); inputs are never
 *    modified (reference: every trait method borrows views and returns owned
 *    arrays).  Outputs are caller-allocated.
 *  - `_f64` / `_f32` select the scalar type (reference macros instantiate
 *    f32/f64/c32/c64; complex is out of scope).
 *  - Index vectors are int64, 0-based, in device memory (reference: usize,
 *    src/qr.rs:36-39).
 *  - Calls are asynchronous on the context's HIP stream unless they return a
 *    host scalar (documented per function); rc_synchronize() waits.
 *  - Every function returns rc_status; rc_last_error_message() explains it.
 *    Status values mirror RustyCompressionError (src/types.rs:11-21).
 *  - One context per host thread per GPU; contexts share nothing.
 */
#ifndef RUSTY_COMPRESSION_AMD_H
#define RUSTY_COMPRESSION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_ABI_VERSION 1

typedef int32_t rc_status;
enum {
    RC_OK = 0,
    RC_LINALG_ERROR = 1,      /* RustyCompressionError::LinalgError      src/types.rs:13-14 */
    RC_COMPRESSION_ERROR = 2, /* RustyCompressionError::CompressionError src/types.rs:15-16 */
    RC_LAYOUT_ERROR = 3,      /* RustyCompressionError::LayoutError      src/types.rs:17-18 */
    RC_PIVOTED_QR_ERROR = 4,  /* RustyCompressionError::PivotedQRError   src/types.rs:19-20 */
    RC_INVALID_ARGUMENT = 5,  /* the reference panics (assert!) on these: src/qr.rs:99, src/permutation.rs:96-99 */
    RC_RUNTIME_ERROR = 6      /* HIP runtime failure (no reference counterpart) */
};

/* ndarray-style strided device view; strides in ELEMENTS. */
typedef struct rc_matrix {
    void *data;
    int64_t rows;
    int64_t cols;
    int64_t row_stride;
    int64_t col_stride;
} rc_matrix;

typedef struct rc_context rc_context;

/* CompressionType (src/lib.rs:82-87) */
enum { RC_COMPRESS_ADAPTIVE = 0, RC_COMPRESS_RANK = 1 };
/* MatrixPermutationMode (src/permutation.rs:7-16) */
enum { RC_PERM_COL = 0, RC_PERM_ROW = 1, RC_PERM_COLINV = 2, RC_PERM_ROWINV = 3 };
/* VectorPermutationMode (src/permutation.rs:19-24) */
enum { RC_VPERM_INV = 0, RC_VPERM_NOINV = 1 };

/* ---------------------------------------------------------------- context -- */
int32_t rc_abi_version(void);
/* stream: a hipStream_t (may be NULL = the device's default stream). */
rc_status rc_create(rc_context **ctx, int32_t device, void *hip_stream);
rc_status rc_destroy(rc_context *ctx);
rc_status rc_set_stream(rc_context *ctx, void *hip_stream);
/* Stream helpers for hosts without HIP bindings: a non-blocking hipStream_t on `device`, to be passed to
 * rc_create / rc_set_stream (one context + stream per independent compression in flight). */
rc_status rc_stream_create(int32_t device, void **hip_stream);
rc_status rc_stream_destroy(int32_t device, void *hip_stream);
rc_status rc_synchronize(rc_context *ctx);
/* Wait for n contexts at once: a completion event is recorded on EVERY context's stream before the first wait.  A host
 * that keeps dozens of independent compressions in flight (one context + stream each) should use this instead of
 * rc_synchronize context by context: stream-by-stream waits returned after max(259 ms, work) on ROCm 7.2 when more
 * than ~28 streams held work ("Completion waits").  No reference counterpart (the reference is synchronous). */
rc_status rc_synchronize_all(rc_context *const *ctxs, int32_t n);
/* Pre-size the internal workspace arena (bytes); optional, it grows on demand. */
rc_status rc_reserve_workspace(rc_context *ctx, size_t bytes);
const char *rc_last_error_message(const rc_context *ctx);

/* Plain device-memory helpers so a host without HIP bindings (the Rust crate)
 * can stage ndarray data: upload -> call -> download. */
rc_status rc_device_malloc(rc_context *ctx, size_t bytes, void **ptr);
rc_status rc_device_free(rc_context *ctx, void *ptr);
rc_status rc_memcpy_h2d(rc_context *ctx, void *dst_dev, const void *src_host, size_t bytes);
rc_status rc_memcpy_d2h(rc_context *ctx, void *dst_host, const void *src_dev, size_t bytes); /* synchronous */

/* hipGraph capture: everything issued on the context's stream between begin and
 * end becomes one replayable graph (launch-bound chains such as the ~130
 * dependent pivot steps of a pivoted QR replay without host launch overhead).
 * Only calls without host synchronisation may be captured, and the context must
 * have run the same call once eagerly before (workspace sizing).  A graph has workspace
 * addresses baked in: while any graph of a context is alive, a workspace the context
 * outgrows is kept allocated instead of freed, so eager calls of any size may be mixed with
 * replays (stream order keeps them apart).  No reference counterpart: the reference is
 * synchronous host code. */
rc_status rc_graph_begin_capture(rc_context *ctx);
rc_status rc_graph_end_capture(rc_context *ctx, void **graph_exec);
rc_status rc_graph_launch(rc_context *ctx, void *graph_exec);
rc_status rc_graph_destroy(rc_context *ctx, void *graph_exec);

/* Options.  RC_OPT_TALL_SKINNY_FAST_PATH (default 1): tall-skinny pivoted QR / QR inside
 * the SVD run as CholeskyQR2 + an LDS-resident pivoted QR of the small factor + a sign fix
 * that restores LAPACK's Householder sign convention; the path certifies itself
 * (positive Cholesky pivots, ||Q1^T Q1 - I|| small) and the call falls back to the plain
 * Householder chain (exact ?geqp3/?orgqr operation order) when the certificate fails.
 * 0 forces the Householder chain.  While a hipGraph is being captured no fallback is
 * possible: failures are OR-ed into a health word instead, which rc_get_health returns
 * and clears (0 = every captured fast path certified itself). */
/* RC_OPT_WIDE_LAZY_QRCP (default 1): pivoted QR of short-wide matrices (m <= 256 << n) keeps the
 * m x m orthogonal factor explicitly and never rewrites the trailing matrix (same ?laqp2
 * pivoting semantics); 0 selects the eager Householder chain. */
/* RC_OPT_WIDE_COOP_QRCP (default 1): the same short-wide pivoted QR as ONE cooperative launch that
 * keeps the whole matrix in registers (one grid barrier per Householder step instead of two kernel
 * launches); certifies itself like the tall-skinny path (bit 4 of the health word = its workgroups
 * could not all become resident in time) and falls back to the lazy scheme; 0 disables it. */
/* Reproducibility.  Every entry point is deterministic: the same call with the same options on the same context state gives the
 * same bits (no atomics on results, fixed split-K reduction order).  Two settings choose between implementations of the same
 * factorization, and bits are promised PER SETTING, not across them: (i) RC_OPT_CONCURRENCY_HINT decides whether wide products
 * split K (a lone launch) or not (many compressions in flight): results differ by summation order; (ii) a call recorded into a
 * hipGraph cannot read scalars back, so general-shape pivoted QRs run the per-step chain there and the blocked panels eagerly:
 * same pivots on the data-determined prefix, factors equal to rounding (tests: test_captured_pivoted_qr_of_a_blocked_eligible_shape…).
 * The cfg3 pipeline (rc_rsvd_id_*) takes the same path eagerly and captured: its replays equal the eager result bit for bit.
 * Environment switches that select between implementations (measurement aids, read once per process) change rounding the same
 * way and are NOT part of the promise: RC_TSQR_FOLD (order of the small factors of the tall-skinny QR), RC_QRCP_CAND_MB (which
 * columns of a blocked panel are updated reflector by reflector and which through the block update: last bits of R12 / Z),
 * RC_GEMM_LANES_TARGET, RC_GEMM_SMALL_TARGET (split-K counts), RC_GEMM_PIPE_* / RC_GEMM_RING (which loop runs a product).
 * Schedules that only change WHEN or WHERE the same arithmetic is issued -- RC_WQ_STAGES, RC_QRCP_OPTIMISTIC, RC_BATCH_OPTIMISTIC,
 * RC_QRCP_KEEP_DIRECT, RC_ID_FUSED -- give identical bits (tested for the staged k_wq_coop and the optimistic blocked QRCP). */
/* RC_OPT_POWER_ITERATION_FIXED (default 0): rc_sample_range_power_iteration_* performs it_count power steps
 * (Y <- A orth(A^H orth(Y))) as the reference documents; 0 reproduces the reference's behaviour, where a shadowed
 * loop variable leaves exactly one step (src/random_sampling.rs:145-153). */
/* RC_OPT_FORK_BRANCHES (default 0): 1 makes rc_rsvd_id_* run its two independent consumers of B = Q^H A (the SVD and
 * the pivoted-QR / ID branch) side by side on a second stream owned by the context (fork / join with events; captured
 * into the same hipGraph).  Latency of one compression 6.8 -> 5.6 ms; with dozens of such graphs in flight the
 * throughput measured lower (680 vs 913 compressions/s), hence opt-in. */
/* RC_OPT_BLOCKED_QRCP (default 1): pivoted QR of general shapes (neither tall-skinny nor short-wide) runs ?geqp3 the way
 * LAPACK does -- ?laqps panels of 32 steps with the delayed F-matrix update and one MFMA GEMM block update per panel -- with
 * the panel restricted to the candidate columns whose norm can still be the maximum (kernels_qrblk.hip): identical pivoting
 * rule and down-dating formulas, about three passes over the trailing matrix per panel instead of two per step.  Reads one
 * small struct back per panel, so it is not used while a hipGraph is being captured.  0 selects the per-step chain. */
/* RC_OPT_CONCURRENCY_HINT (default 1): how many independent compressions the host keeps in flight on this device (one
 * context + stream each).  A lone GEMM splits its reduction dimension until every CU has a workgroup; with many
 * compressions in flight the other streams fill the chip, and un-split products are cheaper (no partial slabs, no
 * reduction kernel): with a hint >= 8 a product with >= 32 output tiles is not split.  Results are deterministic for a
 * given hint; different hints differ by summation order only. */
/* RC_OPT_COOP_PANEL (default 1): the steps of a blocked-QRCP panel run as ONE cooperative launch with the candidate columns
 * resident in registers (one grid barrier per step instead of two kernel boundaries; the candidates are read and written once
 * per panel) whenever the active rows fit (m - j0 <= 4096 in f32, 3072 in f64) and the candidates fit its 512 waves; a launch
 * that cannot run (too many tied candidates, workgroups not co-resident in time) leaves the panel untouched and the step
 * kernels take it.  Same pivot rule; the candidates' norms are down-dated with ?laqp2's formula (their columns are kept up to
 * date, so a norm that loses its accuracy is recomputed on the spot instead of ending the panel).  0 = step kernels only. */
enum { RC_OPT_TALL_SKINNY_FAST_PATH = 1, RC_OPT_WIDE_LAZY_QRCP = 2, RC_OPT_WIDE_COOP_QRCP = 3, RC_OPT_POWER_ITERATION_FIXED = 4, RC_OPT_FORK_BRANCHES = 5,
       RC_OPT_BLOCKED_QRCP = 6, RC_OPT_CONCURRENCY_HINT = 7, RC_OPT_COOP_PANEL = 8 };
rc_status rc_set_option(rc_context *ctx, int32_t option, int64_t value);
/* Health word (read and cleared), OR of: 1 non-positive Cholesky pivot, 2 first CholeskyQR pass too far from orthonormal
 * (both: tall-skinny fast path inside a graph, where no fallback is possible), 4 cooperative short-wide QR could not get
 * its workgroups resident, 8 the right-vector workgroup of the Jacobi SVD never saw its producer within the spin bound,
 * 16 a Jacobi SVD used up its sweep budget before converging (eager calls as well: the factors are then accurate to the
 * last sweep's rotation angles only), 32 an index handed to a gather (a permutation entry, a column index) was outside the
 * source: the affected outputs are zero instead of whatever a wild address held.  0 = every result stands. */
rc_status rc_get_health(rc_context *ctx, int32_t *word);

/* Stage / kernel timers: HIP events recorded on the context's stream around the
 * dominant kernels and pipeline stages (used by bench.py for the roofline line).
 * rc_profile_count / rc_profile_get synchronise the stream. */
rc_status rc_profile_enable(rc_context *ctx, int32_t on);
rc_status rc_profile_reset(rc_context *ctx);
rc_status rc_profile_count(rc_context *ctx, int32_t *n);
rc_status rc_profile_get(rc_context *ctx, int32_t i, char *name, int32_t name_cap, double *total_ms, int64_t *calls);

/* Instantiation of the most recent GEMM launch of this context, spelled as rocprofv3 prints it
 * (e.g. "k_gemm_f64q<1,1,136,256,16,2,4,2,0,0>"): lets bench.py tell whether a committed PMC traffic figure was taken on
 * the kernel that still runs.  Diagnostic, no reference counterpart. */
const char *rc_last_gemm_kernel_name(const rc_context *ctx);

/* ------------------------------------------------------- random_matrix.rs -- */
/* RandomMatrix::random_gaussian (src/random_matrix.rs:21, :120-125): i.i.d. N(0,1), drawn in f64 and cast to T,
 * filled in row-major order.  The reference's rand 0.8 / rand_distr 0.4 ziggurat stream is sequential host code with no
 * pinned version, so it cannot be reproduced sample for sample; parity tests pass Omega explicitly.  The stream of THIS
 * ABI is defined here (and restated in oracle/philox.py, which the GPU kernel is tested against bit for bit):
 *   Philox4x32-10 (Salmon et al., SC'11), key = (seed & 0xffffffff, seed >> 32), counter = (b & 0xffffffff, b >> 32, 0, 0)
 *   for block b -> words w0..w3;  a = w0 << 32 | w1,  b' = w2 << 32 | w3;
 *   u1 = ((a >> 11) + 1) * 2^-53 in (0, 1],  u2 = (b' >> 11) * 2^-53 in [0, 1);
 *   z0 = sqrt(-2 ln u1) cos(2 pi u2),  z1 = sqrt(-2 ln u1) sin(2 pi u2)   (Box-Muller);
 *   number e of the stream (seed, offset) is z_{(offset + e) & 1} of block (offset + e) >> 1;
 *   element (i, j) of out is number i * cols + j;  the f32 variant is the cast of the f64 value. */
rc_status rc_random_gaussian_f64(rc_context *ctx, rc_matrix out, uint64_t seed, uint64_t offset);
rc_status rc_random_gaussian_f32(rc_context *ctx, rc_matrix out, uint64_t seed, uint64_t offset);
/* The raw uint32 words of the same Philox stream (word w = w_{w & 3} of block w >> 2), n words from word_offset into
 * device memory `out`: the integer generator is checkable bit for bit against oracle/philox.py and Random123's
 * known-answer vectors.  No reference counterpart. */
rc_status rc_random_bits_u32(rc_context *ctx, uint32_t *out, int64_t n, uint64_t seed, uint64_t word_offset);

/* --------------------------------------------------------------- types.rs -- */
/* MatMat::matmat for dense matrices (src/types.rs:58-71, :103-121): Y = A X.
 * One GEMM instead of the reference's per-column gemv loop (blanket impl,
 * src/types.rs:145); results differ by summation order only. */
rc_status rc_matmat_f64(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y);
rc_status rc_matmat_f32(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y);
/* ConjMatMat::conj_matmat (src/types.rs:88-101, :123-133): Y = A^H X (ncols(A) x ncols(X)). */
rc_status rc_conj_matmat_f64(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y);
rc_status rc_conj_matmat_f32(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y);
/* ndarray `.dot` on two matrices (all call sites listed in SURVEY.md 2b N9):
 * C = alpha * op(A) op(B) + beta * C, op = transpose when the flag is non-zero. */
rc_status rc_gemm_f64(rc_context *ctx, int32_t trans_a, int32_t trans_b, double alpha, rc_matrix a, rc_matrix b, double beta, rc_matrix c);
rc_status rc_gemm_f32(rc_context *ctx, int32_t trans_a, int32_t trans_b, float alpha, rc_matrix a, rc_matrix b, float beta, rc_matrix c);
/* RelDiff (src/types.rs:162-196): host scalar out, synchronous. */
rc_status rc_rel_diff_fro_f64(rc_context *ctx, rc_matrix first, rc_matrix second, double *out);
rc_status rc_rel_diff_fro_f32(rc_context *ctx, rc_matrix first, rc_matrix second, float *out);

/* --------------------------------------------------------- permutation.rs -- */
/* invert_permutation_vector (src/permutation.rs:28-38) */
rc_status rc_invert_permutation(rc_context *ctx, const int64_t *perm, int64_t n, int64_t *inverse);
/* ApplyPermutationToMatrix::apply_permutation (src/permutation.rs:84-144).
 * RC_INVALID_ARGUMENT when perm_len mismatches (the reference asserts). */
rc_status rc_apply_permutation_matrix_f64(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t perm_len, rc_matrix out);
rc_status rc_apply_permutation_matrix_f32(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t perm_len, rc_matrix out);
/* ApplyPermutationToVector::apply_permutation (src/permutation.rs:153-183); vectors are n x 1 views. */
rc_status rc_apply_permutation_vector_f64(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t perm_len, rc_matrix out);
rc_status rc_apply_permutation_vector_f32(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t perm_len, rc_matrix out);

/* ---------------------------------------------------------- pivoted_qr.rs -- */
/* PivotedQR::pivoted_qr (src/pivoted_qr.rs:25-31, :81-183) = ?geqp3 + ?orgqr:
 *   A P = Q R,  q: m x k,  r: k x n upper trapezoidal,  ind[j] = column of A at position j.
 * k = q.cols = r.rows.  k == min(m, n) reproduces the reference exactly.
 * k <  min(m, n) is the TRUNCATED factorization (stops after k Householder
 * steps): q, r[:k] and ind[:k] equal the full factorization's, ind[k:] lists
 * the remaining columns in the order the swaps left them. */
rc_status rc_pivoted_qr_f64(rc_context *ctx, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_pivoted_qr_f32(rc_context *ctx, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind);
/* PivotedQR::pivoted_lq (src/pivoted_qr.rs:32-41), LQ::compute_from (src/qr.rs:354-362):
 *   P A = L Q,  l: m x k,  q: k x n,  ind: m. */
rc_status rc_pivoted_lq_f64(rc_context *ctx, rc_matrix a, rc_matrix l, rc_matrix q, int64_t *ind);
rc_status rc_pivoted_lq_f32(rc_context *ctx, rc_matrix a, rc_matrix l, rc_matrix q, int64_t *ind);

/* LAPACK granularity, for a maintainer who swaps only the `$qrf` call (src/pivoted_qr.rs:139-172, ?geqp3), `lax::Lapack::q`
 * (src/pivoted_qr.rs:104-108, ?orgqr / ?ungqr) or the triangular solves of the IDs (src/qr.rs:298, :392, ?trtrs) and keeps the
 * reference's own code around them.  All pointers are DEVICE pointers.
 * rc_geqp3: a (m x n, any strides) is overwritten with LAPACK's output format -- the factorization of A P, columns in pivoted order:
 *   R on and above the diagonal of the first kmax rows, Householder vectors below it in the first kmax columns; jpvt (n, 0-based:
 *   LAPACK's minus one), tau (kmax).  kmax == min(m, n) is ?geqp3; kmax < min(m, n) stops after kmax steps (rows >= kmax of the
 *   columns >= kmax are then unspecified).
 * rc_orgqr: q (m x k) = H_0 ... H_{k-1} [I; 0] from the first k columns of such an a and tau.
 * rc_trsm_upper: T X = B in place (t: k x k upper triangular, b: k x nrhs); no singularity check (?trtrs' INFO is not reproduced). */
rc_status rc_geqp3_f64(rc_context *ctx, rc_matrix a, int64_t kmax, int64_t *jpvt, double *tau);
rc_status rc_geqp3_f32(rc_context *ctx, rc_matrix a, int64_t kmax, int64_t *jpvt, float *tau);
rc_status rc_orgqr_f64(rc_context *ctx, rc_matrix a, const double *tau, int64_t k, rc_matrix q);
rc_status rc_orgqr_f32(rc_context *ctx, rc_matrix a, const float *tau, int64_t k, rc_matrix q);
rc_status rc_trsm_upper_f64(rc_context *ctx, rc_matrix t, rc_matrix b);
rc_status rc_trsm_upper_f32(rc_context *ctx, rc_matrix t, rc_matrix b);

/* ---------------------------------------------------------- compute_svd.rs -- */
/* ComputeSVD::compute_svd (src/compute_svd.rs:18-27) = thin ?gesdd:
 *   u: m x r, s: r (device, descending), vt: r x n, r = min(m, n).
 * Singular vectors are unique up to a sign per pair; S and U diag(S) Vt match gesdd. */
rc_status rc_compute_svd_f64(rc_context *ctx, rc_matrix a, rc_matrix u, double *s, rc_matrix vt);
rc_status rc_compute_svd_f32(rc_context *ctx, rc_matrix a, rc_matrix u, float *s, rc_matrix vt);

/* ------------------------------------------------------------------ qr.rs -- */
/* compress_qr_tolerance / compress_lq_tolerance (src/qr.rs:187-200, :99-112):
 * first i with |d_ii / d_00| < tol on the diagonal of `tri` (R or L).
 * RC_COMPRESSION_ERROR if none, RC_INVALID_ARGUMENT unless 0 <= tol < 1. Synchronous. */
rc_status rc_rank_by_tolerance_f64(rc_context *ctx, rc_matrix tri, double tol, int64_t *rank);
rc_status rc_rank_by_tolerance_f32(rc_context *ctx, rc_matrix tri, double tol, int64_t *rank);
/* QRTraits::to_mat (src/qr.rs:160-166): out = Q (R with COLINV permutation). */
rc_status rc_qr_to_mat_f64(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix out);
rc_status rc_qr_to_mat_f32(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix out);
/* LQTraits::to_mat (src/qr.rs:73-77): out = (L with ROWINV permutation) Q. */
rc_status rc_lq_to_mat_f64(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix out);
rc_status rc_lq_to_mat_f32(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix out);
/* QRTraits::column_id (src/qr.rs:270-309): c: m x k, z: k x n (both branches). */
rc_status rc_qr_column_id_f64(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix c, rc_matrix z);
rc_status rc_qr_column_id_f32(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix c, rc_matrix z);
/* LQTraits::row_id (src/qr.rs:363-403): x: m x k, r_rows: k x n. */
rc_status rc_lq_row_id_f64(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix x, rc_matrix r_rows);
rc_status rc_lq_row_id_f32(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix x, rc_matrix r_rows);
/* QRTraits::compute_from_range_estimate (src/qr.rs:311-323): range m x r', A m x n ->
 * q: m x k, r: k x n, ind: n, with k = min(r', n). */
rc_status rc_qr_from_range_estimate_f64(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_qr_from_range_estimate_f32(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind);

/* ----------------------------------------------------------------- svd.rs -- */
/* compress_svd_tolerance (src/svd.rs:87-101) on a device vector of singular values. Synchronous. */
rc_status rc_svd_rank_by_tolerance_f64(rc_context *ctx, const double *s, int64_t len, double tol, int64_t *rank);
rc_status rc_svd_rank_by_tolerance_f32(rc_context *ctx, const float *s, int64_t len, double tol, int64_t *rank);
/* SVDTraits::to_mat (src/svd.rs:42-54): out = U diag(S) Vt. */
rc_status rc_svd_to_mat_f64(rc_context *ctx, rc_matrix u, const double *s, rc_matrix vt, rc_matrix out);
rc_status rc_svd_to_mat_f32(rc_context *ctx, rc_matrix u, const float *s, rc_matrix vt, rc_matrix out);
/* SVDTraits::to_qr (src/svd.rs:150-163): pivoted QR of diag(S) Vt, Q = U Q_b. */
rc_status rc_svd_to_qr_f64(rc_context *ctx, rc_matrix u, const double *s, rc_matrix vt, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_svd_to_qr_f32(rc_context *ctx, rc_matrix u, const float *s, rc_matrix vt, rc_matrix q, rc_matrix r, int64_t *ind);
/* SVDTraits::compute_from_range_estimate (src/svd.rs:171-183): u: m x r', s: r', vt: r' x n. */
rc_status rc_svd_from_range_estimate_f64(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix u, double *s, rc_matrix vt);
rc_status rc_svd_from_range_estimate_f32(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix u, float *s, rc_matrix vt);

/* ------------------------------- col_interp_decomp.rs / row_interp_decomp.rs -- */
/* ColumnIDTraits::two_sided_id (src/col_interp_decomp.rs:116-125): from C (m x k)
 * computes the row ID of C: c_out: m x k (= row_id.x), x: k x k (= row_id.r), row_ind: m.
 * (r = Z and col_ind are carried over unchanged by the caller.) */
rc_status rc_column_id_two_sided_f64(rc_context *ctx, rc_matrix c, rc_matrix c_out, rc_matrix x, int64_t *row_ind);
rc_status rc_column_id_two_sided_f32(rc_context *ctx, rc_matrix c, rc_matrix c_out, rc_matrix x, int64_t *row_ind);
/* RowIDTraits::two_sided_id (src/row_interp_decomp.rs:120-130): from R (k x n)
 * computes the column ID of R: x: k x k (= col_id.c), r_out: k x n (= col_id.z), col_ind: n. */
rc_status rc_row_id_two_sided_f64(rc_context *ctx, rc_matrix r, rc_matrix x, rc_matrix r_out, int64_t *col_ind);
rc_status rc_row_id_two_sided_f32(rc_context *ctx, rc_matrix r, rc_matrix x, rc_matrix r_out, int64_t *col_ind);

/* ------------------------------------------------------ random_sampling.rs -- */
/* MaxColNorm::max_col_norm (src/random_sampling.rs:184-191). Synchronous, host scalar out. */
rc_status rc_max_col_norm_f64(rc_context *ctx, rc_matrix y, double *out);
rc_status rc_max_col_norm_f32(rc_context *ctx, rc_matrix y, float *out);
/* SampleRange::sample_range_by_rank (src/random_sampling.rs:103-118):
 * q: m x k = first k columns of Q in the pivoted QR of A Omega, Omega: n x (k+p).
 * omega.data == NULL => Omega is generated on the device from (seed, offset 0). */
rc_status rc_sample_range_by_rank_f64(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_by_rank_f32(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q);
/* SampleRangePowerIteration::sample_range_power_iteration (src/random_sampling.rs:131-160),
 * INCLUDING its variable-shadowing quirk: for it_count >= 1 exactly one power
 * step survives. */
rc_status rc_sample_range_power_iteration_f64(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, int64_t it_count, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_power_iteration_f32(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, int64_t it_count, rc_matrix omega, uint64_t seed, rc_matrix q);
/* AdaptiveSampling::sample_range_adaptive (src/random_sampling.rs:223-274).
 *   q_cap: m x cap output buffer; on return the basis is its first *rank columns.
 *   omegas: n x (sample_size * blocks) explicit Gaussian blocks consumed left to
 *           right (data == NULL => generated on device from seed).
 *   hist_rank / hist_res (host, hist_cap entries): the residual history
 *           Vec<(usize, f64)>; *hist_len entries are written.
 * Returns RC_COMPRESSION_ERROR if cap columns (or the explicit Omega blocks) are
 * exhausted before the tolerance is met.  Synchronous (the loop condition is a host scalar). */
rc_status rc_sample_range_adaptive_f64(rc_context *ctx, rc_matrix a, double rel_tol, int64_t sample_size, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len);
rc_status rc_sample_range_adaptive_f32(rc_context *ctx, rc_matrix a, double rel_tol, int64_t sample_size, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len);

/* ------------------------------------------------ operators behind callbacks -- */
/* The reference's range finders are implemented for ANY operator, not only for dense arrays:
 *   impl<Op: MatMat> SampleRange for Op            src/random_sampling.rs:102
 *   impl<Op: MatMat + ConjMatMat> SampleRangePowerIteration for Op   :130
 *   impl<Op: MatMat + ConjMatMat> AdaptiveSampling for Op            :222
 *   QRTraits / SVDTraits::compute_from_range_estimate<Op: ConjMatMat>   src/qr.rs:311-323, src/svd.rs:171-183
 * (trait contract: src/types.rs:40-51 nrows / ncols / matvec, :77-81 conj_matvec, :58-71 / :88-101 the derived products).
 * rc_operator is that contract at the C ABI: the extents and the two PRODUCTS (the per-column matvec loop of the blanket impl
 * src/types.rs:145-146 is the host's business -- a host with only a matvec loops over the columns of x inside its callback).
 *   matmat(user, ctx, x, y):       y (rows x s) = A x,    x: cols x s
 *   conj_matmat(user, ctx, x, y):  y (cols x s) = A^H x,  x: rows x s      (may be NULL where only MatMat is required)
 * x and y are strided DEVICE views (any layout; y may be the transposed view of a row-major buffer) owned by the library for the
 * duration of the call.  A callback enqueues its work on the context's stream (rc_get_stream) -- it may call this library's own
 * entry points with the SAME ctx (they nest: the outer call's workspace stays intact) or launch kernels of its own -- and returns
 * RC_OK or a status that the calling entry point then returns (rc_last_error_message names the product).  It must not synchronise
 * unless it has to; it cannot be recorded into a hipGraph (RC_INVALID_ARGUMENT while capturing).
 * The *_op_* entry points below run the same internal steps as their dense twins with the two products replaced by the callbacks:
 * a dense matrix behind callbacks that call rc_matmat / rc_conj_matmat reproduces the dense entry point bit for bit (f64; the f32
 * products pick their tiles by operand layout, so there the agreement is to rounding).  The complex scalar types have the same
 * entry points (rc_*_op_c64 / _c32, declared with the complex instantiations below; rc_rsvd_id_op_* is real-only). */
typedef struct rc_operator rc_operator;
typedef int32_t (*rc_operator_product_fn)(void *user, rc_context *ctx, rc_matrix x, rc_matrix y); /* returns an rc_status */
struct rc_operator {
    int64_t rows;                       /* MatVec::nrows  src/types.rs:44-45 */
    int64_t cols;                       /* MatVec::ncols  src/types.rs:47-48 */
    rc_operator_product_fn matmat;      /* MatMat::matmat          src/types.rs:58-71  */
    rc_operator_product_fn conj_matmat; /* ConjMatMat::conj_matmat src/types.rs:88-101 */
    void *user;
};
/* the hipStream_t the context orders its work on (for a callback's own kernels) */
rc_status rc_get_stream(rc_context *ctx, void **hip_stream);
/* SampleRange for Op (src/random_sampling.rs:102-121) */
rc_status rc_sample_range_by_rank_op_f64(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_by_rank_op_f32(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q);
/* SampleRangePowerIteration for Op (src/random_sampling.rs:130-163) */
rc_status rc_sample_range_power_iteration_op_f64(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, int64_t it_count, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_power_iteration_op_f32(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, int64_t it_count, rc_matrix omega, uint64_t seed, rc_matrix q);
/* AdaptiveSampling for Op (src/random_sampling.rs:222-277) */
rc_status rc_sample_range_adaptive_op_f64(rc_context *ctx, const rc_operator *op, double rel_tol, int64_t sample_size, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len);
rc_status rc_sample_range_adaptive_op_f32(rc_context *ctx, const rc_operator *op, double rel_tol, int64_t sample_size, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len);
/* QRTraits::compute_from_range_estimate<Op: ConjMatMat> (src/qr.rs:311-323) */
rc_status rc_qr_from_range_estimate_op_f64(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_qr_from_range_estimate_op_f32(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix q, rc_matrix r, int64_t *ind);
/* SVDTraits::compute_from_range_estimate<Op: ConjMatMat> (src/svd.rs:171-183) */
rc_status rc_svd_from_range_estimate_op_f64(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix u, double *s, rc_matrix vt);
rc_status rc_svd_from_range_estimate_op_f32(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix u, float *s, rc_matrix vt);
/* the fused pipeline below (rc_rsvd_id_*) over an operator; not capturable */
struct rc_rsvd_id_out;
rc_status rc_rsvd_id_op_f64(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, const struct rc_rsvd_id_out *out);
rc_status rc_rsvd_id_op_f32(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, const struct rc_rsvd_id_out *out);

/* ------------------------------------------------ fused pipeline (bench) -- */
/* cfg3 "rSVD + ID" in one call, no host synchronisation inside (capturable in
 * a hipGraph): sample_range_by_rank -> SVD::compute_from_range_estimate ->
 * QR::compute_from_range_estimate -> column_id, sharing B = Q^H A between the
 * two range-estimate consumers (identical results to calling them one by one).
 * Any output with data == NULL is skipped; id outputs all NULL => rSVD only. */
typedef struct rc_rsvd_id_out {
    rc_matrix range_q; /* m x k   */
    rc_matrix u;       /* m x k   */
    void *s;           /* k       */
    rc_matrix vt;      /* k x n   */
    rc_matrix qr_q;    /* m x k   */
    rc_matrix qr_r;    /* k x n   */
    int64_t *qr_ind;   /* n       */
    rc_matrix id_c;    /* m x k   */
    rc_matrix id_z;    /* k x n   */
} rc_rsvd_id_out;
rc_status rc_rsvd_id_f64(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, const rc_rsvd_id_out *out);
rc_status rc_rsvd_id_f32(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, const rc_rsvd_id_out *out);

/* cfg5 unit of work: rank-k column ID of one dense matrix,
 * QR::compute_from -> compress(RANK(k)) -> column_id
 * (examples/interpolative_decomposition.rs:25-32) with the truncated
 * factorization (see rc_pivoted_qr).  c: m x k, z: k x n, col_ind: n. */
rc_status rc_column_id_rank_f64(rc_context *ctx, rc_matrix a, int64_t k, rc_matrix c, rc_matrix z, int64_t *col_ind);
rc_status rc_column_id_rank_f32(rc_context *ctx, rc_matrix a, int64_t k, rc_matrix c, rc_matrix z, int64_t *col_ind);

/* ------------------------------------------- batches of independent matrices -- */
/* BASELINE.json configs[4]: many same-shaped matrices, rank-k column ID each (the call sequence of
 * examples/interpolative_decomposition.rs:25-32 per matrix), sharded by MATRIX over the GPUs of a node, then ONE exchange
 * step: the gather of the finished factor blocks.  The reference has neither a batch nor a communication layer (it is
 * single-process host code), so these entry points have no file:line counterpart beyond the per-matrix sequence.
 *
 * Packed layout, per matrix i at byte offset i * rc_batch_packed_bytes(m, n, k, sizeof(T)):
 *   C (m x k, C order) | Z (k x n, C order) | pad to 8 bytes | col_ind (n x int64). */
size_t rc_batch_packed_bytes(int64_t m, int64_t n, int64_t k, int32_t elem_size);
/* Contiguous block partition of n_items over `world` ranks (the first n_items % world ranks take one extra item):
 * matrix i of 64 goes to rank i / 8 on 8 GPUs.  Pure host arithmetic. */
rc_status rc_batch_shard_range(int64_t n_items, int32_t world, int32_t rank, int64_t *start, int64_t *count);
/* Rank-k column ID of `count` same-shaped device matrices into the packed device buffer `packed`
 * (count * rc_batch_packed_bytes).  The matrices are spread over the nctx contexts (one HIP stream each, same device) and
 * pipelined (each lane is waited for on its own event, first in, first out; idle lanes take the next matrix); results are identical to count calls of
 * rc_column_id_rank_*.  Blocking: returns when every factor is in `packed`.  Errors are reported on ctxs[0]. */
rc_status rc_batch_column_id_f64(rc_context *const *ctxs, int32_t nctx, const rc_matrix *mats, int32_t count, int64_t k, void *packed);
rc_status rc_batch_column_id_f32(rc_context *const *ctxs, int32_t nctx, const rc_matrix *mats, int32_t count, int64_t k, void *packed);

/* The gather over RCCL (xGMI inside a node).  One process per GPU: rank 0 calls rc_comm_unique_id and hands the 128
 * bytes to the other ranks by whatever means the host has (MPI, a file, torch.distributed), every rank calls
 * rc_comm_init, then rc_comm_gather moves bytes_per_rank bytes from every rank's `send` into
 * recv[rank * bytes_per_rank ...] on `root` (grouped ncclSend / ncclRecv on the context's stream, asynchronous:
 * rc_synchronize(ctx) completes it).  librccl is opened at run time; RC_RUNTIME_ERROR if it is not available. */
typedef struct rc_comm rc_comm;
rc_status rc_comm_unique_id(void *id128);
rc_status rc_comm_init(rc_comm **comm, int32_t world, int32_t rank, const void *id128, int32_t device);
rc_status rc_comm_gather(rc_comm *comm, rc_context *ctx, const void *send, void *recv, size_t bytes_per_rank, int32_t root);
rc_status rc_comm_destroy(rc_comm *comm);
const char *rc_comm_last_error_message(const rc_comm *comm);
/* The two collectives of the row-sharded single-matrix path below (all ranks end with the same bits):
 *   rc_comm_all_gather      recv[r * bytes_per_rank ...] = rank r's send (send may be the rank's own block of recv)
 *   rc_comm_all_reduce_sum  buf[i] = sum over the ranks, in place; elem_size 8 = double, 4 = float
 * ordered on the context's stream (ncclAllGather / ncclAllReduce: asynchronous).  rc_comm_world reports the extent. */
rc_status rc_comm_all_gather(rc_comm *comm, rc_context *ctx, const void *send, void *recv, size_t bytes_per_rank);
rc_status rc_comm_all_reduce_sum(rc_comm *comm, rc_context *ctx, void *buf, size_t count, int32_t elem_size);
rc_status rc_comm_world(const rc_comm *comm, int32_t *world, int32_t *rank);
/* A communicator over the HOST's own communication layer (MPI, gloo, a test harness) instead of RCCL: the library stages
 * the (small) buffers of the two collectives above through host memory, waits for the stream and calls back.  Both
 * callbacks work on host pointers, return 0 on success, and must leave the same bytes on every rank.
 * rc_comm_gather is not available on such a communicator. */
typedef int32_t (*rc_host_all_gather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
typedef int32_t (*rc_host_all_reduce_sum_fn)(void *user, void *buf, size_t count, int32_t elem_size);
rc_status rc_comm_init_host(rc_comm **comm, int32_t world, int32_t rank, int32_t device, rc_host_all_gather_fn all_gather,
                            rc_host_all_reduce_sum_fn all_reduce_sum, void *user);

/* ----------------------------------------- one matrix sharded by rows over the GPUs -- */
/* The cfg3 pipeline (sample_range_by_rank -> SVD / QR::compute_from_range_estimate -> column_id, src/random_sampling.rs:103-118,
 * src/svd.rs:171-183, src/qr.rs:311-323, :270-309) for ONE matrix A = [A_0; ...; A_{W-1}] whose row block a_local
 * (m_r x n, m_r >= k + p) lives on this rank; the reference is single-process, so there is no file:line for the split.
 * Local: the sketch Y_r = A_r Omega (Omega = the Philox stream of `seed`, identical on every rank), its pivoted QR, every
 * product with A.  Exchanged: the l x l factor of every rank (all-gather; the pivoted QR of the stack gives THE pivoted
 * QR of Y, TSQR) and the k x n projection B (all-reduce).  out: range_q, u, qr_q, id_c have m_r rows (this rank's rows
 * of the global factors); s, vt, qr_r, qr_ind, id_z are replicated bit-identically.  "NULL = skipped" as in rc_rsvd_id.
 * comm == NULL: one rank.  Blocking where the transport is (host communicators); not capturable. */
rc_status rc_rsvd_id_row_sharded_f64(rc_comm *comm, rc_context *ctx, rc_matrix a_local, int64_t k, int64_t p, uint64_t seed, const rc_rsvd_id_out *out);
rc_status rc_rsvd_id_row_sharded_f32(rc_comm *comm, rc_context *ctx, rc_matrix a_local, int64_t k, int64_t p, uint64_t seed, const rc_rsvd_id_out *out);

/* ------------------------------------------------------------- complex scalars (c32 / c64) -- */
/* The reference instantiates every trait for f32, f64, c32 and c64 (macros at src/qr.rs:408-416, src/pivoted_qr.rs:187-190,
 * src/svd.rs:188-191, src/random_sampling.rs:123-126, :165-168, :277-280, src/types.rs:198-204).  Complex matrices are
 * interleaved (re, im) pairs -- the layout of ndarray's Complex<T> and of numpy complex arrays -- described by the same
 * rc_matrix (strides in complex elements).  Same semantics as the real entry points with A^H wherever they have A^T:
 * conj_matmat is A^H X (src/types.rs:128-132), Q has orthonormal columns in the complex inner product, singular values
 * and norms are real (float / double arguments below), R has a real diagonal (?geqp3), permutation indices as before.
 * rc_gemm_*: trans = 0 none, 1 transpose, 2 conjugate transpose.  rc_random_gaussian_*: element (i, j) takes normals
 * 2 (offset + i cols + j) (real part) and the next one (imaginary part) of the Philox stream, the order the reference
 * draws them in (src/random_matrix.rs:136-143).  rc_rsvd_id_c* / rc_batch_column_id_c* are compositions of the calls below with
 * the real entry points' members, layout and "null = skipped" rule (B = Q^H A formed once; not the tuned real hot path);
 * rc_svd_rank_by_tolerance_c* forwards to the real call (singular values are real). */
typedef struct rc_complex32 { float re, im; } rc_complex32;
typedef struct rc_complex64 { double re, im; } rc_complex64;
rc_status rc_random_gaussian_c64(rc_context *ctx, rc_matrix out, uint64_t seed, uint64_t offset);
rc_status rc_matmat_c64(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y);
rc_status rc_conj_matmat_c64(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y);
rc_status rc_gemm_c64(rc_context *ctx, int32_t trans_a, int32_t trans_b, rc_complex64 alpha, rc_matrix a, rc_matrix b, rc_complex64 beta, rc_matrix c);
rc_status rc_rel_diff_fro_c64(rc_context *ctx, rc_matrix first, rc_matrix second, double *out);
rc_status rc_apply_permutation_matrix_c64(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t perm_len, rc_matrix out);
rc_status rc_apply_permutation_vector_c64(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t perm_len, rc_matrix out);
rc_status rc_pivoted_qr_c64(rc_context *ctx, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_pivoted_lq_c64(rc_context *ctx, rc_matrix a, rc_matrix l, rc_matrix q, int64_t *ind);
rc_status rc_geqp3_c64(rc_context *ctx, rc_matrix a, int64_t kmax, int64_t *jpvt, rc_complex64 *tau);   /* ?geqp3, LAPACK format (see rc_geqp3_f64) */
rc_status rc_orgqr_c64(rc_context *ctx, rc_matrix a, const rc_complex64 *tau, int64_t k, rc_matrix q);   /* ?ungqr */
rc_status rc_trsm_upper_c64(rc_context *ctx, rc_matrix t, rc_matrix b);
rc_status rc_compute_svd_c64(rc_context *ctx, rc_matrix a, rc_matrix u, double *s, rc_matrix vt);
rc_status rc_rank_by_tolerance_c64(rc_context *ctx, rc_matrix tri, double tol, int64_t *rank);
rc_status rc_qr_to_mat_c64(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix out);
rc_status rc_lq_to_mat_c64(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix out);
rc_status rc_qr_column_id_c64(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix c, rc_matrix z);
rc_status rc_lq_row_id_c64(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix x, rc_matrix r_rows);
rc_status rc_qr_from_range_estimate_c64(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_svd_to_mat_c64(rc_context *ctx, rc_matrix u, const double *s, rc_matrix vt, rc_matrix out);
rc_status rc_svd_to_qr_c64(rc_context *ctx, rc_matrix u, const double *s, rc_matrix vt, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_svd_from_range_estimate_c64(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix u, double *s, rc_matrix vt);
rc_status rc_column_id_two_sided_c64(rc_context *ctx, rc_matrix c, rc_matrix c_out, rc_matrix x, int64_t *row_ind);
rc_status rc_row_id_two_sided_c64(rc_context *ctx, rc_matrix r, rc_matrix x, rc_matrix r_out, int64_t *col_ind);
rc_status rc_max_col_norm_c64(rc_context *ctx, rc_matrix y, double *out);
rc_status rc_sample_range_by_rank_c64(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q);
/* the operator-callback entry points (rc_operator, above) for c64: views are interleaved complex, strides in complex elements */
rc_status rc_sample_range_by_rank_op_c64(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_power_iteration_op_c64(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, int64_t it_count, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_adaptive_op_c64(rc_context *ctx, const rc_operator *op, double rel_tol, int64_t sample_size, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len);
rc_status rc_qr_from_range_estimate_op_c64(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_svd_from_range_estimate_op_c64(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix u, double *s, rc_matrix vt);
rc_status rc_sample_range_power_iteration_c64(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, int64_t it_count, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_adaptive_c64(rc_context *ctx, rc_matrix a, double rel_tol, int64_t sample_size, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len);
rc_status rc_column_id_rank_c64(rc_context *ctx, rc_matrix a, int64_t k, rc_matrix c, rc_matrix z, int64_t *col_ind);
rc_status rc_svd_rank_by_tolerance_c64(rc_context *ctx, const double *s, int64_t len, double tol, int64_t *rank);
rc_status rc_rsvd_id_c64(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, const rc_rsvd_id_out *out);
rc_status rc_batch_column_id_c64(rc_context *const *ctxs, int32_t nctx, const rc_matrix *mats, int32_t count, int64_t k, void *packed);
rc_status rc_random_gaussian_c32(rc_context *ctx, rc_matrix out, uint64_t seed, uint64_t offset);
rc_status rc_matmat_c32(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y);
rc_status rc_conj_matmat_c32(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y);
rc_status rc_gemm_c32(rc_context *ctx, int32_t trans_a, int32_t trans_b, rc_complex32 alpha, rc_matrix a, rc_matrix b, rc_complex32 beta, rc_matrix c);
rc_status rc_rel_diff_fro_c32(rc_context *ctx, rc_matrix first, rc_matrix second, float *out);
rc_status rc_apply_permutation_matrix_c32(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t perm_len, rc_matrix out);
rc_status rc_apply_permutation_vector_c32(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t perm_len, rc_matrix out);
rc_status rc_pivoted_qr_c32(rc_context *ctx, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_pivoted_lq_c32(rc_context *ctx, rc_matrix a, rc_matrix l, rc_matrix q, int64_t *ind);
rc_status rc_geqp3_c32(rc_context *ctx, rc_matrix a, int64_t kmax, int64_t *jpvt, rc_complex32 *tau);   /* ?geqp3, LAPACK format (see rc_geqp3_f64) */
rc_status rc_orgqr_c32(rc_context *ctx, rc_matrix a, const rc_complex32 *tau, int64_t k, rc_matrix q);   /* ?ungqr */
rc_status rc_trsm_upper_c32(rc_context *ctx, rc_matrix t, rc_matrix b);
rc_status rc_compute_svd_c32(rc_context *ctx, rc_matrix a, rc_matrix u, float *s, rc_matrix vt);
rc_status rc_rank_by_tolerance_c32(rc_context *ctx, rc_matrix tri, double tol, int64_t *rank);
rc_status rc_qr_to_mat_c32(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix out);
rc_status rc_lq_to_mat_c32(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix out);
rc_status rc_qr_column_id_c32(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix c, rc_matrix z);
rc_status rc_lq_row_id_c32(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix x, rc_matrix r_rows);
rc_status rc_qr_from_range_estimate_c32(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_svd_to_mat_c32(rc_context *ctx, rc_matrix u, const float *s, rc_matrix vt, rc_matrix out);
rc_status rc_svd_to_qr_c32(rc_context *ctx, rc_matrix u, const float *s, rc_matrix vt, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_svd_from_range_estimate_c32(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix u, float *s, rc_matrix vt);
rc_status rc_column_id_two_sided_c32(rc_context *ctx, rc_matrix c, rc_matrix c_out, rc_matrix x, int64_t *row_ind);
rc_status rc_row_id_two_sided_c32(rc_context *ctx, rc_matrix r, rc_matrix x, rc_matrix r_out, int64_t *col_ind);
rc_status rc_max_col_norm_c32(rc_context *ctx, rc_matrix y, float *out);
rc_status rc_sample_range_by_rank_c32(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_by_rank_op_c32(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_power_iteration_op_c32(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, int64_t it_count, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_adaptive_op_c32(rc_context *ctx, const rc_operator *op, double rel_tol, int64_t sample_size, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len);
rc_status rc_qr_from_range_estimate_op_c32(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix q, rc_matrix r, int64_t *ind);
rc_status rc_svd_from_range_estimate_op_c32(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix u, float *s, rc_matrix vt);
rc_status rc_sample_range_power_iteration_c32(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, int64_t it_count, rc_matrix omega, uint64_t seed, rc_matrix q);
rc_status rc_sample_range_adaptive_c32(rc_context *ctx, rc_matrix a, double rel_tol, int64_t sample_size, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len);
rc_status rc_column_id_rank_c32(rc_context *ctx, rc_matrix a, int64_t k, rc_matrix c, rc_matrix z, int64_t *col_ind);
rc_status rc_svd_rank_by_tolerance_c32(rc_context *ctx, const float *s, int64_t len, double tol, int64_t *rank);
rc_status rc_rsvd_id_c32(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, const rc_rsvd_id_out *out);
rc_status rc_batch_column_id_c32(rc_context *const *ctxs, int32_t nctx, const rc_matrix *mats, int32_t count, int64_t k, void *packed);

#ifdef __cplusplus
}
#endif
#endif /* RUSTY_COMPRESSION_AMD_H */
