// Internal shared declarations of librusty_compression_amd (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/rusty_compression_amd.h"

namespace rc {

// ---------------------------------------------------------------------------
// strided device view (mirror of rc_matrix with a typed pointer)
// ---------------------------------------------------------------------------
template <typename T>
struct Mat {
    T *p = nullptr;
    int64_t rows = 0, cols = 0, rs = 0, cs = 0;

    __host__ __device__ Mat() {}
    __host__ __device__ Mat(T *p_, int64_t r, int64_t c, int64_t rs_, int64_t cs_) : p(p_), rows(r), cols(c), rs(rs_), cs(cs_) {}
    __host__ __device__ inline T &at(int64_t i, int64_t j) const { return p[i * rs + j * cs]; }
    // sub-view [r0, r0+nr) x [c0, c0+nc)
    __host__ __device__ Mat<T> sub(int64_t r0, int64_t nr, int64_t c0, int64_t nc) const {
        return Mat<T>(p + r0 * rs + c0 * cs, nr, nc, rs, cs);
    }
    __host__ __device__ Mat<T> t() const { return Mat<T>(p, cols, rows, cs, rs); }
    __host__ __device__ bool empty() const { return rows == 0 || cols == 0; }
};

template <typename T>
inline Mat<T> from_c(const rc_matrix &m) {
    return Mat<T>(static_cast<T *>(m.data), m.rows, m.cols, m.row_stride, m.col_stride);
}
// leading dimensions of internal temporaries are kept even so the GEMM can stage them with
// 2-element vector loads
inline int64_t even_ld(int64_t x) { return (x + 1) & ~int64_t(1); }
// freshly allocated column-major / row-major views
template <typename T>
inline Mat<T> colmajor(T *p, int64_t rows, int64_t cols, int64_t ld) { return Mat<T>(p, rows, cols, 1, ld); }
template <typename T>
inline Mat<T> rowmajor(T *p, int64_t rows, int64_t cols, int64_t ld) { return Mat<T>(p, rows, cols, ld, 1); }

// ---------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------
struct Error {
    rc_status code;
    std::string msg;
};

[[noreturn]] inline void fail(rc_status code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error{code, std::string(buf)};
}

#define RC_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) ::rc::fail(RC_RUNTIME_ERROR, "%s failed: %s (%s:%d)", #expr,             \
                                         hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)

#define RC_REQUIRE(cond, code, ...)                  \
    do {                                             \
        if (!(cond)) ::rc::fail(code, __VA_ARGS__);  \
    } while (0)

}  // namespace rc

// ---------------------------------------------------------------------------
// context: device, stream, workspace arena
// ---------------------------------------------------------------------------
struct rc_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    std::string last_gemm_kernel;  // instantiation of the most recent GEMM launch, spelled as rocprofv3 prints it (rc_last_gemm_kernel_name)

    // Workspace arena.  Every C-ABI call resets the bump pointer on entry; all
    // work is ordered on `stream`, so a later call may reuse the bytes of an
    // earlier one.  Growth (rare: first calls only) synchronises the stream.
    int call_depth = 0;                 // > 0 inside a C-ABI call: a nested call (from an operator callback) must not reset the arena
    char *arena = nullptr;
    size_t arena_size = 0;
    size_t arena_off = 0;
    size_t arena_high = 0;              // high-water mark of the current call
    std::vector<void *> overflow;       // extra blocks taken while the arena was too small
    // hipGraphs captured on this context have arena addresses baked in: while any is alive a
    // superseded arena is retired (kept allocated) instead of freed
    int live_graphs = 0;
    std::vector<void *> retired;
    void retire_arena();
    // Second stream + second arena: the two consumers of the projection B (SVD branch, pivoted-QR / ID branch of
    // rc_rsvd_id) are independent and run side by side (fork / join with events; capturable in a hipGraph).  While
    // the side branch is being issued `stream` and the arena fields above are swapped with these.
    hipStream_t aux_stream = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    struct ArenaState { char *base = nullptr; size_t size = 0, off = 0, high = 0; std::vector<void *> overflow; } aux_arena;
    void swap_arena();
    int opt_fork = 0;  // 1: rc_rsvd_id runs its two branches side by side (lower latency; measured LOWER throughput with many graphs in flight)
    hipEvent_t sync_ev = nullptr;       // rc_synchronize_all: one completion event per context
    // blocked QRCP: the ~65 launches of one panel replayed from a cached hipGraph (keyed by every baked-in pointer / size)
    std::map<std::vector<uint64_t>, hipGraphExec_t> qrb_graphs;
    void *pinned = nullptr;             // small pinned host buffer for scalar read-backs
    size_t pinned_size = 0;
    int pinned_cursor = 0;              // next free QrbState slot of `pinned` for optimistic blocked-QRCP jobs (kernels_qrblk.hip)

    // Tall-skinny factorizations: 1 = CholeskyQR2 fast path with certificate + Householder
    // fallback (default), 0 = always the Householder chain.  `health` is a device word the
    // fast path ORs its failure bits into; outside graph capture it is read back (one small
    // synchronisation) and the call falls back, during capture it is left for rc_get_health.
    int opt_tsqr = 1;
    int opt_power_fixed = 0;  // 1: sample_range_power_iteration really iterates it_count times (opt-in; the reference does one)
    int opt_wide_coop = 1;  // short-wide pivoted QR as ONE cooperative register-resident kernel (0: multi-kernel paths)
    int opt_wide_lazy = 1;  // short-wide pivoted QR through the read-only lazy scheme (0: eager Householder chain)
    int opt_blocked = 1;    // general shapes: blocked ?laqps panels + GEMM block update (0: per-step Householder chain)
    int opt_coop_panel = 1; // RC_OPT_COOP_PANEL: cooperative register-resident panels of the blocked QRCP
    int opt_lanes = 1;      // RC_OPT_CONCURRENCY_HINT: independent compressions the host keeps in flight on this device
    int *health = nullptr;
    int *health_word();
    unsigned *epoch = nullptr;  // launch counter of the fused Jacobi (keys its producer -> consumer records)
    unsigned *epoch_word();

    // hipGraph capture state and the event-based stage/kernel timers (rc_profile_*)
    bool capturing = false;
    bool prof_on = false;
    struct ProfPending { std::string name; hipEvent_t beg, end; };
    struct ProfAcc { double ms = 0; int64_t calls = 0; };
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> prof_free;
    std::map<std::string, ProfAcc> prof_acc;
    hipEvent_t prof_event();
    void prof_resolve();

    void reset_arena();
    void *alloc_bytes(size_t bytes);
    template <typename T>
    T *alloc(size_t n) { return static_cast<T *>(alloc_bytes(n * sizeof(T))); }
    void reserve(size_t bytes);
    void release_all();
};

namespace rc {

// Times everything issued on the context's stream during its lifetime with a pair
// of HIP events recorded ON THAT STREAM (no-op unless rc_profile_enable(ctx, 1),
// and never while a hipGraph is being captured).
struct ProfScope {
    rc_context *c;
    hipEvent_t beg = nullptr;
    std::string name;
    ProfScope(rc_context *ctx, const char *fmt, ...) : c(ctx) {
        if (!c->prof_on || c->capturing) return;
        char buf[160];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        name = buf;
        beg = c->prof_event();
        (void)hipEventRecord(beg, c->stream);
    }
    ~ProfScope() {
        if (!beg) return;
        hipEvent_t end = c->prof_event();
        (void)hipEventRecord(end, c->stream);
        c->prof_pending.push_back({name, beg, end});
    }
};

// marks/restores the arena inside a call (stack discipline for temporaries)
struct ArenaMark {
    rc_context *c;
    size_t off;
    explicit ArenaMark(rc_context *ctx) : c(ctx), off(ctx->arena_off) {}
    ~ArenaMark() { c->arena_off = off; }
};

// ---------------------------------------------------------------------------
// kernel launchers (defined in the kernels_*.hip translation units)
// ---------------------------------------------------------------------------
template <typename T> void fill_gaussian(rc_context *c, Mat<T> out, uint64_t seed, uint64_t offset);
void philox_words(rc_context *c, uint32_t *out, int64_t n, uint64_t seed, uint64_t word_offset);
template <typename T> void copy_mat(rc_context *c, Mat<T> src, Mat<T> dst);                 // dst = src (any strides)
template <typename T> void fill_identity(rc_context *c, Mat<T> dst);                        // dst = [I | 0] / [I ; 0]
template <typename T> void fill_zero(rc_context *c, Mat<T> dst);
template <typename T> void scale_rows(rc_context *c, const T *s, Mat<T> src, Mat<T> dst);   // dst[i,:] = s[i] * src[i,:]
template <typename T> void gather_cols(rc_context *c, Mat<T> src, const int64_t *idx, Mat<T> dst);  // dst[:, j] = src[:, idx[j]]
// C, Z of the rank-k column ID from the ?geqp3-format factorization w of a (kernels_qr.hip: two launches, no Q)
template <typename T> void column_id_from_qrcp(rc_context *c, Mat<T> a, Mat<T> w, int64_t k, const int64_t *jpvt, Mat<T> cm, Mat<T> z);
template <typename T> void id_z_from_r(rc_context *c, Mat<T> r, int64_t k, const int64_t *ind, Mat<T> z);
void invert_perm(rc_context *c, const int64_t *perm, int64_t n, int64_t *inv);
void fill_words(rc_context *c, void *p, size_t bytes, unsigned v);  // every 32-bit word of [p, p + bytes) = v, by a kernel on c->stream (no hipMemset*)
void iota_i64(rc_context *c, int64_t *p, int64_t n);
// column 2-norms squared; out[j] = sum_i a(i,j)^2
template <typename T> void col_sumsq(rc_context *c, Mat<T> a, T *out);
// device scalar reductions; results land in device memory `out`
template <typename T> void max_sqrt(rc_context *c, const T *v, int64_t n, T *out);           // out = sqrt(max v)
template <typename T> void fro_diff(rc_context *c, Mat<T> a, Mat<T> b, T *out2);             // out2[0]=|a-b|_F^2, out2[1]=|b|_F^2
template <typename T> void adaptive_residual_update(rc_context *c, Mat<T> y, Mat<T> corr);   // y -= corr

// C = alpha * A * B + beta * C on views (transposes are expressed through strides)
template <typename T> void gemm(rc_context *c, T alpha, Mat<T> a, Mat<T> b, T beta, Mat<T> cmat);
template <typename T> void complete_left_basis(rc_context *c, Mat<T> uc, const T *s);  // kernels_svd.hip: orthonormal vectors for zero singular values

// Householder QR with optional column pivoting (LAPACK ?geqp3 / ?laqp2 semantics).
//   w      : m x n COLUMN-MAJOR working matrix (cs = ld, rs = 1), overwritten
//   jpvt   : n (position -> original column of w).  Columns are NOT moved physically.
//   tau    : kmax
//   vn     : 2n scratch (partial norms)
template <typename T> void geqp3_inplace(rc_context *c, Mat<T> w, int64_t kmax, bool pivot, int64_t *jpvt, T *tau, T *vn);
// general shapes (kernels_qrblk.hip): ?laqps panels on a candidate set + one MFMA GEMM block update per panel; same output
// format as geqp3_inplace; reads one small struct back per panel (not capturable in a hipGraph)
template <typename T> bool geqp3_blocked_supported(int64_t m, int64_t n, int64_t kmax);
// q_out (m x kq column-major, may be empty): Q formed panel by panel with the T factors the panels built
// restore_from (optional, only without q_out): the matrix w is a copy of -- enables the optimistic issue (all panels enqueued on the
// usual outcome, one wait, a check; on a failed check w is restored from it and the per-panel path runs)
template <typename T> void geqp3_blocked(rc_context *c, Mat<T> w, int64_t kmax, int64_t *jpvt, T *tau, Mat<T> q_out, Mat<T> restore_from = Mat<T>());
// the same factorization as a resumable job (one host wait per panel): begin -> { issue, <stream synchronised>, finish } ... -> end
template <typename T> struct BlockedQrcpJob;
template <typename T> BlockedQrcpJob<T> *qrb_begin(rc_context *c, Mat<T> w, int64_t kmax, int64_t *jpvt, T *tau);
template <typename T> void qrb_issue(BlockedQrcpJob<T> *job);
template <typename T> bool qrb_finish(BlockedQrcpJob<T> *job);
template <typename T> void qrb_form_q(BlockedQrcpJob<T> *job, Mat<T> q);  // after completion
template <typename T> void qrb_end(BlockedQrcpJob<T> *job);
struct QrbState;
template <typename T> bool qrb_optimistic_possible(BlockedQrcpJob<T> *job);
template <typename T> bool qrb_issue_all_optimistic(BlockedQrcpJob<T> *job);   // false: no pinned slots left on the context (wait, check what is pending, reset c->pinned_cursor)
template <typename T> bool qrb_verify_optimistic(BlockedQrcpJob<T> *job);
template <typename T> bool qrb_verify_optimistic(const QrbState *host_log, const std::vector<int> &nbp_log, bool broken);
template <typename T> void qrb_optimistic_log(BlockedQrcpJob<T> *job, const QrbState **log, std::vector<int> *nbp_log, bool *broken);
template <typename T> void qrb_keep_t(BlockedQrcpJob<T> *job, bool keep);  // false: no Q will be formed, the panels' T factors need not be kept
// short-wide matrices (m <= 256 << n): read-only "lazy" pivoted QR with the explicit m x m factor
template <typename T> bool wide_lazy_supported(int64_t m, int64_t n);
template <typename T> void geqp3_wide_lazy(rc_context *c, Mat<T> b, int64_t kmax, int64_t *jpvt, Mat<T> q, Mat<T> r);
// short-wide matrices, one cooperative launch (kernels_wqcoop.hip): wf gets the ?geqp3 output format;
// flag gets bit 4 (w is never written) when the workgroups could not all become resident in time
template <typename T> bool wide_coop_supported(int64_t m, int64_t n, int device);
template <typename T> void geqp3_wide_coop(rc_context *c, Mat<T> w, Mat<T> wf, int64_t kmax, int64_t *jpvt, T *tau, int *flag);
void coop_prepare(int device);
// device-wide budget of the cooperative kernels, in half compute units (kernels_wqcoop.hip)
unsigned *coop_semaphore_of(int device);
unsigned coop_budget_units(int device);
void coop_gate_launch(rc_context *c, unsigned need, unsigned *sync, unsigned long long *hdr, int hdr_words, const int *proceed = nullptr);
// r(i, p) = (i <= p) ? w(i, jpvt[p]) : 0  for i < r.rows
template <typename T> void extract_r(rc_context *c, Mat<T> w, const int64_t *jpvt, Mat<T> r);
// qw (m x kq column-major) = H_0 ... H_{k-1} [I ; 0], reflector j stored in column jpvt[j] of w
template <typename T> void form_q(rc_context *c, Mat<T> w, const int64_t *jpvt, const T *tau, int64_t k, Mat<T> qw);
// solve T X = B in place; t: k x k upper triangular view (any strides), b: k x nrhs view
template <typename T> void trsm_upper(rc_context *c, Mat<T> t, Mat<T> b);
// tall-skinny fast path (kernels_tsqr.hip): CholeskyQR2 + LDS-resident QRCP + Householder sign fix
template <typename T> bool tsqr_supported(int64_t m, int64_t n);
template <typename T> void tsqr_cholqr2(rc_context *c, Mat<T> y, Mat<T> q, Mat<T> r, int *flag);
template <typename T> void tsqr_cholqr2_factored(rc_context *c, Mat<T> y, Mat<T> q1, Mat<T> r2i, Mat<T> r, int *flag);
template <typename T> void qrcp_small(rc_context *c, Mat<T> rin, int64_t kmax, bool pivot, int64_t *jpvt, Mat<T> rout, Mat<T> q2);
template <typename T> void householder_sign_fix(rc_context *c, Mat<T> q, Mat<T> r);
template <typename T> void qrcp_tall_fast(rc_context *c, Mat<T> y, int64_t k, bool pivot, Mat<T> q, Mat<T> r, int64_t *ind, int *flag);
// one-sided Jacobi SVD of the square column-major n x n matrix g (destroyed):
//   uc (n x n col-major) = left vectors, s (n) descending, vc (n x n col-major) = right vectors
template <typename T> void jacobi_svd(rc_context *c, Mat<T> g, Mat<T> vwork, Mat<T> uc, T *s, Mat<T> vc);

}  // namespace rc
