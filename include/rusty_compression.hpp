// rusty_compression.hpp -- header-only C++ mirror of the reference crate's operator / trait
// surface (rusty-compression v0.1.1, /root/reference/src/lib.rs:90-102) over the C ABI of
// librusty_compression_amd.so (rusty_compression_amd.h).  The reference is compiled (Rust) code
// and no Rust toolchain exists in the build image, so this is the compiled-language host side:
// same names, same argument meaning, same error behaviour -- and no arithmetic: every method is
// one call into the C ABI.  The Rust rendition of the same surface is in bindings/rust/.
//
//   using namespace rusty_compression;
//   Context ctx(0);
//   auto a   = DeviceMatrix<double>::from_host(ctx, host_ptr, m, n);        // C-order host data
//   auto qr  = QR<double>::compute_from(a).compress(CompressionType::RANK(20));
//   auto tid = qr.column_id().two_sided_id();                               // examples/interpolative_decomposition.rs
//   double err = rel_diff_fro(tid.to_mat(), a);
#pragma once

#include <cmath>
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <type_traits>
#include <vector>

#include "rusty_compression_amd.h"

namespace rusty_compression {

// ---- errors: RustyCompressionError (reference src/types.rs:11-21) ------------------------------
struct RustyCompressionError : std::runtime_error { using std::runtime_error::runtime_error; };
struct LinalgError : RustyCompressionError { using RustyCompressionError::RustyCompressionError; };
struct CompressionError : RustyCompressionError { using RustyCompressionError::RustyCompressionError; };
struct LayoutError : RustyCompressionError { using RustyCompressionError::RustyCompressionError; };
struct PivotedQRError : RustyCompressionError { using RustyCompressionError::RustyCompressionError; };
struct AssertionFailed : std::logic_error { using std::logic_error::logic_error; };  // the reference panics
struct HipRuntimeError : RustyCompressionError { using RustyCompressionError::RustyCompressionError; };

// ---- CompressionType (reference src/lib.rs:82-87) ----------------------------------------------
struct CompressionType {
    enum Kind { ADAPTIVE_, RANK_ } kind;
    double value;
    static CompressionType ADAPTIVE(double tol) { return {ADAPTIVE_, tol}; }
    static CompressionType RANK(std::size_t rank) { return {RANK_, (double)rank}; }
};
enum class MatrixPermutationMode { COL = RC_PERM_COL, ROW = RC_PERM_ROW, COLINV = RC_PERM_COLINV, ROWINV = RC_PERM_ROWINV };

class Context {
  public:
    explicit Context(int device = 0, void *hip_stream = nullptr) {
        if (rc_create(&raw_, device, hip_stream) != RC_OK) throw HipRuntimeError("rc_create failed (no MI355X visible?)");
    }
    ~Context() { rc_destroy(raw_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    rc_context *raw() const { return raw_; }
    void synchronize() const { check(rc_synchronize(raw_)); }
    void set_option(int32_t option, int64_t value) const { check(rc_set_option(raw_, option, value)); }
    void check(rc_status st) const {
        if (st == RC_OK) return;
        const std::string msg = rc_last_error_message(raw_);
        switch (st) {
            case RC_LINALG_ERROR: throw LinalgError(msg);
            case RC_COMPRESSION_ERROR: throw CompressionError(msg.empty() ? "Could not compress to desired tolerance" : msg);
            case RC_LAYOUT_ERROR: throw LayoutError(msg);
            case RC_PIVOTED_QR_ERROR: throw PivotedQRError(msg);
            case RC_INVALID_ARGUMENT: throw AssertionFailed(msg);
            default: throw HipRuntimeError(msg);
        }
    }

  private:
    rc_context *raw_ = nullptr;
};

// ---- scalar dispatch --------------------------------------------------------------------------
// the reference's `Scalar` (f32, f64, c32, c64; src/lib.rs): Real = the type of norms, tolerances and singular values, wire = how
// a scalar crosses the C ABI (complex numbers as the (re, im) pairs rc_complex32 / rc_complex64)
template <typename T> struct Scalar {
    using real = T;
    static constexpr bool is_complex = false;
    static T wire(T v) { return v; }
};
template <> struct Scalar<std::complex<float>> {
    using real = float;
    static constexpr bool is_complex = true;
    static rc_complex32 wire(std::complex<float> v) { return rc_complex32{v.real(), v.imag()}; }
};
template <> struct Scalar<std::complex<double>> {
    using real = double;
    static constexpr bool is_complex = true;
    static rc_complex64 wire(std::complex<double> v) { return rc_complex64{v.real(), v.imag()}; }
};
using c32 = std::complex<float>;
using c64 = std::complex<double>;

template <typename T> struct Api;
#define RC_API(T, SUF, RSUF)                                                                                            \
    template <> struct Api<T> {                                                                                     \
        static constexpr auto random_gaussian = rc_random_gaussian_##SUF;                                           \
        static constexpr auto matmat = rc_matmat_##SUF;                                                             \
        static constexpr auto conj_matmat = rc_conj_matmat_##SUF;                                                   \
        static constexpr auto gemm = rc_gemm_##SUF;                                                                 \
        static constexpr auto rel_diff_fro = rc_rel_diff_fro_##SUF;                                                 \
        static constexpr auto apply_permutation_matrix = rc_apply_permutation_matrix_##SUF;                         \
        static constexpr auto pivoted_qr = rc_pivoted_qr_##SUF;                                                     \
        static constexpr auto pivoted_lq = rc_pivoted_lq_##SUF;                                                     \
        static constexpr auto geqp3 = rc_geqp3_##SUF;                                                               \
        static constexpr auto orgqr = rc_orgqr_##SUF;                                                               \
        static constexpr auto trsm_upper = rc_trsm_upper_##SUF;                                                     \
        static constexpr auto compute_svd = rc_compute_svd_##SUF;                                                   \
        static constexpr auto rank_by_tolerance = rc_rank_by_tolerance_##SUF;                                       \
        static constexpr auto qr_to_mat = rc_qr_to_mat_##SUF;                                                       \
        static constexpr auto lq_to_mat = rc_lq_to_mat_##SUF;                                                       \
        static constexpr auto qr_column_id = rc_qr_column_id_##SUF;                                                 \
        static constexpr auto lq_row_id = rc_lq_row_id_##SUF;                                                       \
        static constexpr auto qr_from_range_estimate = rc_qr_from_range_estimate_##SUF;                             \
        static constexpr auto svd_rank_by_tolerance = rc_svd_rank_by_tolerance_##RSUF; /* singular values are real */                           \
        static constexpr auto svd_to_mat = rc_svd_to_mat_##SUF;                                                     \
        static constexpr auto svd_to_qr = rc_svd_to_qr_##SUF;                                                       \
        static constexpr auto svd_from_range_estimate = rc_svd_from_range_estimate_##SUF;                           \
        static constexpr auto column_id_two_sided = rc_column_id_two_sided_##SUF;                                   \
        static constexpr auto row_id_two_sided = rc_row_id_two_sided_##SUF;                                         \
        static constexpr auto max_col_norm = rc_max_col_norm_##SUF;                                                 \
        static constexpr auto sample_range_by_rank = rc_sample_range_by_rank_##SUF;                                 \
        static constexpr auto sample_range_power_iteration = rc_sample_range_power_iteration_##SUF;                 \
        static constexpr auto sample_range_adaptive = rc_sample_range_adaptive_##SUF;                               \
        static constexpr auto column_id_rank = rc_column_id_rank_##SUF;                                             \
    };
RC_API(double, f64, f64)
RC_API(float, f32, f32)
RC_API(c64, c64, f64)
RC_API(c32, c32, f32)
#undef RC_API

// ---- device-resident C-order arrays (the reference's Array2 / Array1<usize>) ----------------------
template <typename T>
class DeviceBuffer {
  public:
    DeviceBuffer() = default;
    DeviceBuffer(const Context &ctx, std::size_t count) : ctx_(&ctx), count_(count) {
        ctx.check(rc_device_malloc(ctx.raw(), count * sizeof(T), &ptr_));
    }
    DeviceBuffer(DeviceBuffer &&o) noexcept { *this = std::move(o); }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
        release();
        ctx_ = o.ctx_; ptr_ = o.ptr_; count_ = o.count_;
        o.ptr_ = nullptr; o.count_ = 0;
        return *this;
    }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    T *data() const { return static_cast<T *>(ptr_); }
    std::size_t size() const { return count_; }
    const Context &ctx() const { return *ctx_; }
    std::vector<T> to_host() const {
        std::vector<T> h(count_);
        if (count_) ctx_->check(rc_memcpy_d2h(ctx_->raw(), h.data(), ptr_, count_ * sizeof(T)));
        return h;
    }
    void from_host(const T *src) { if (count_) ctx_->check(rc_memcpy_h2d(ctx_->raw(), ptr_, src, count_ * sizeof(T))); }

  private:
    void release() { if (ptr_ && ctx_) rc_device_free(ctx_->raw(), ptr_); ptr_ = nullptr; }
    const Context *ctx_ = nullptr;
    void *ptr_ = nullptr;
    std::size_t count_ = 0;
};

template <typename T>
class DeviceMatrix {
  public:
    DeviceMatrix() = default;
    DeviceMatrix(const Context &ctx, int64_t rows, int64_t cols) : buf_(ctx, (std::size_t)(rows * cols)), rows_(rows), cols_(cols) {}
    static DeviceMatrix from_host(const Context &ctx, const T *c_order, int64_t rows, int64_t cols) {
        DeviceMatrix m(ctx, rows, cols);
        m.buf_.from_host(c_order);
        return m;
    }
    std::vector<T> to_host() const { return buf_.to_host(); }
    int64_t nrows() const { return rows_; }
    int64_t ncols() const { return cols_; }
    const Context &ctx() const { return buf_.ctx(); }
    rc_matrix view() const { return rc_matrix{buf_.data(), rows_, cols_, cols_, 1}; }
    // owned copy of the leading rows x cols block (the reference's slice(..).to_owned()):
    // one strided gather through the permutation entry point with the identity index
    DeviceMatrix leading(int64_t rows, int64_t cols) const {
        DeviceMatrix out(ctx(), rows, cols);
        if (rows == 0 || cols == 0) return out;
        std::vector<int64_t> h((std::size_t)cols);
        for (int64_t i = 0; i < cols; ++i) h[(std::size_t)i] = i;
        DeviceBuffer<int64_t> idx(ctx(), (std::size_t)cols);
        idx.from_host(h.data());
        rc_matrix src{buf_.data(), rows, cols, cols_, 1};
        ctx().check(Api<T>::apply_permutation_matrix(ctx().raw(), RC_PERM_COL, src, idx.data(), cols, out.view()));
        ctx().synchronize();  // idx is freed on return
        return out;
    }

  private:
    DeviceBuffer<T> buf_;
    int64_t rows_ = 0, cols_ = 0;
};

using DeviceIndex = DeviceBuffer<int64_t>;

inline DeviceIndex clone_index(const DeviceIndex &src) {
    DeviceIndex out(src.ctx(), src.size());
    auto h = src.to_host();
    out.from_host(h.data());
    return out;
}
template <typename T>
inline DeviceMatrix<T> clone_matrix(const DeviceMatrix<T> &src) { return src.leading(src.nrows(), src.ncols()); }

// ---- types.rs ----------------------------------------------------------------------------------
template <typename T>
DeviceMatrix<T> dot(const DeviceMatrix<T> &a, const DeviceMatrix<T> &b) {  // ndarray .dot
    DeviceMatrix<T> c(a.ctx(), a.nrows(), b.ncols());
    a.ctx().check(Api<T>::gemm(a.ctx().raw(), 0, 0, Scalar<T>::wire((T)1), a.view(), b.view(), Scalar<T>::wire((T)0), c.view()));
    return c;
}
// op(a) op(b) with op = 0: as is, 1: transpose, 2: conjugate transpose (the reference writes `.t().map(|x| x.conj())`)
template <typename T>
DeviceMatrix<T> dot_op(int op_a, const DeviceMatrix<T> &a, int op_b, const DeviceMatrix<T> &b) {
    DeviceMatrix<T> c(a.ctx(), op_a ? a.ncols() : a.nrows(), op_b ? b.nrows() : b.ncols());
    a.ctx().check(Api<T>::gemm(a.ctx().raw(), op_a, op_b, Scalar<T>::wire((T)1), a.view(), b.view(), Scalar<T>::wire((T)0), c.view()));
    return c;
}
template <typename T>
DeviceMatrix<T> matmat(const DeviceMatrix<T> &op, const DeviceMatrix<T> &x) {  // MatMat::matmat, src/types.rs:58-71
    DeviceMatrix<T> y(op.ctx(), op.nrows(), x.ncols());
    op.ctx().check(Api<T>::matmat(op.ctx().raw(), op.view(), x.view(), y.view()));
    return y;
}
template <typename T>
DeviceMatrix<T> conj_matmat(const DeviceMatrix<T> &op, const DeviceMatrix<T> &x) {  // ConjMatMat, src/types.rs:88-101
    DeviceMatrix<T> y(op.ctx(), op.ncols(), x.ncols());
    op.ctx().check(Api<T>::conj_matmat(op.ctx().raw(), op.view(), x.view(), y.view()));
    return y;
}
template <typename T>
typename Scalar<T>::real rel_diff_fro(const DeviceMatrix<T> &first, const DeviceMatrix<T> &second) {  // RelDiff, src/types.rs:182-188
    typename Scalar<T>::real out = 0;
    first.ctx().check(Api<T>::rel_diff_fro(first.ctx().raw(), first.view(), second.view(), &out));
    return out;
}
template <typename T>
DeviceMatrix<T> random_gaussian(const Context &ctx, int64_t rows, int64_t cols, uint64_t seed, uint64_t offset = 0) {
    DeviceMatrix<T> out(ctx, rows, cols);  // RandomMatrix::random_gaussian, src/random_matrix.rs:21
    ctx.check(Api<T>::random_gaussian(ctx.raw(), out.view(), seed, offset));
    return out;
}

template <typename T> struct ColumnID;
template <typename T> struct RowID;
template <typename T> struct TwoSidedID;

// ---- the LAPACK seam: what the reference calls per factorization ($qrf = ?geqp3 at src/pivoted_qr.rs:139-172, lax::Lapack::q =
// ?orgqr / ?ungqr at :104-108, solve_triangular = ?trtrs at src/qr.rs:298, :392), in LAPACK's own formats ----------------------
template <typename T>
struct Geqp3 {
    DeviceMatrix<T> a;        // the factored A P: R on / above the diagonal, Householder vectors below it (columns in pivoted order)
    DeviceIndex jpvt;         // 0-based
    DeviceBuffer<T> tau;
};
template <typename T>
Geqp3<T> geqp3(const DeviceMatrix<T> &mat, int64_t kmax = -1) {
    const int64_t m = mat.nrows(), n = mat.ncols();
    if (kmax < 0) kmax = m < n ? m : n;
    Geqp3<T> out{clone_matrix(mat), DeviceIndex(mat.ctx(), (std::size_t)n), DeviceBuffer<T>(mat.ctx(), (std::size_t)(kmax > 0 ? kmax : 1))};
    using wire_t = decltype(Scalar<T>::wire(T()));
    mat.ctx().check(Api<T>::geqp3(mat.ctx().raw(), out.a.view(), kmax, out.jpvt.data(), reinterpret_cast<wire_t *>(out.tau.data())));
    return out;
}
template <typename T>
DeviceMatrix<T> orgqr(const Geqp3<T> &f, int64_t k) {
    using wire_t = decltype(Scalar<T>::wire(T()));
    DeviceMatrix<T> q(f.a.ctx(), f.a.nrows(), k);
    f.a.ctx().check(Api<T>::orgqr(f.a.ctx().raw(), f.a.view(), reinterpret_cast<const wire_t *>(f.tau.data()), k, q.view()));
    return q;
}
template <typename T>
DeviceMatrix<T> trsm_upper(const DeviceMatrix<T> &t, const DeviceMatrix<T> &b) {  // returns X with T X = B
    DeviceMatrix<T> x = clone_matrix(b);
    t.ctx().check(Api<T>::trsm_upper(t.ctx().raw(), t.view(), x.view()));
    return x;
}

// ---- qr.rs ---------------------------------------------------------------------------------------
template <typename T>
struct QR {  // src/qr.rs:31-40
    DeviceMatrix<T> q, r;
    DeviceIndex ind;
    int64_t nrows() const { return q.nrows(); }
    int64_t ncols() const { return r.ncols(); }
    int64_t rank() const { return q.ncols(); }

    static QR compute_from(const DeviceMatrix<T> &a) {  // src/qr.rs:251-253
        const int64_t m = a.nrows(), n = a.ncols(), k = m < n ? m : n;
        QR out{DeviceMatrix<T>(a.ctx(), m, k), DeviceMatrix<T>(a.ctx(), k, n), DeviceIndex(a.ctx(), (std::size_t)n)};
        a.ctx().check(Api<T>::pivoted_qr(a.ctx().raw(), a.view(), out.q.view(), out.r.view(), out.ind.data()));
        return out;
    }
    static QR compute_from_range_estimate(const DeviceMatrix<T> &range, const DeviceMatrix<T> &op) {  // src/qr.rs:311-323
        const int64_t m = op.nrows(), n = op.ncols(), k = range.ncols() < n ? range.ncols() : n;
        QR out{DeviceMatrix<T>(op.ctx(), m, k), DeviceMatrix<T>(op.ctx(), k, n), DeviceIndex(op.ctx(), (std::size_t)n)};
        op.ctx().check(Api<T>::qr_from_range_estimate(op.ctx().raw(), range.view(), op.view(), out.q.view(), out.r.view(), out.ind.data()));
        return out;
    }
    DeviceMatrix<T> to_mat() const {  // src/qr.rs:160-166
        DeviceMatrix<T> out(q.ctx(), nrows(), ncols());
        q.ctx().check(Api<T>::qr_to_mat(q.ctx().raw(), q.view(), r.view(), ind.data(), out.view()));
        return out;
    }
    QR compress_qr_rank(int64_t max_rank) const {  // src/qr.rs:169-184
        if (max_rank > q.ncols()) max_rank = q.ncols();
        return QR{q.leading(q.nrows(), max_rank), r.leading(max_rank, r.ncols()), clone_index(ind)};
    }
    QR compress_qr_tolerance(double tol) const {  // src/qr.rs:187-200
        int64_t rank = -1;
        q.ctx().check(Api<T>::rank_by_tolerance(q.ctx().raw(), r.view(), tol, &rank));
        return compress_qr_rank(rank);
    }
    QR compress(CompressionType ct) const {  // src/qr.rs:203-208
        return ct.kind == CompressionType::ADAPTIVE_ ? compress_qr_tolerance(ct.value) : compress_qr_rank((int64_t)ct.value);
    }
    ColumnID<T> column_id() const;  // src/qr.rs:270-309
};

template <typename T>
struct LQ {  // src/qr.rs:42-51
    DeviceMatrix<T> l, q;
    DeviceIndex ind;
    int64_t nrows() const { return l.nrows(); }
    int64_t ncols() const { return q.ncols(); }
    int64_t rank() const { return q.nrows(); }
    static LQ compute_from(const DeviceMatrix<T> &a) {  // src/qr.rs:354-362
        const int64_t m = a.nrows(), n = a.ncols(), k = m < n ? m : n;
        LQ out{DeviceMatrix<T>(a.ctx(), m, k), DeviceMatrix<T>(a.ctx(), k, n), DeviceIndex(a.ctx(), (std::size_t)m)};
        a.ctx().check(Api<T>::pivoted_lq(a.ctx().raw(), a.view(), out.l.view(), out.q.view(), out.ind.data()));
        return out;
    }
    DeviceMatrix<T> to_mat() const {  // src/qr.rs:73-77
        DeviceMatrix<T> out(q.ctx(), nrows(), ncols());
        q.ctx().check(Api<T>::lq_to_mat(q.ctx().raw(), l.view(), q.view(), ind.data(), out.view()));
        return out;
    }
    LQ compress_lq_rank(int64_t max_rank) const {  // src/qr.rs:80-96
        if (max_rank > q.nrows()) max_rank = q.nrows();
        return LQ{l.leading(l.nrows(), max_rank), q.leading(max_rank, q.ncols()), clone_index(ind)};
    }
    LQ compress(CompressionType ct) const {  // src/qr.rs:99-119
        if (ct.kind == CompressionType::RANK_) return compress_lq_rank((int64_t)ct.value);
        int64_t rank = -1;
        q.ctx().check(Api<T>::rank_by_tolerance(q.ctx().raw(), l.view(), ct.value, &rank));
        return compress_lq_rank(rank);
    }
    RowID<T> row_id() const;  // src/qr.rs:363-403
};

// ---- col / row / two-sided interpolative decompositions ----------------------------------------------
template <typename T>
struct TwoSidedID {  // src/two_sided_interp_decomp.rs:19-30
    DeviceMatrix<T> c, x, r;
    DeviceIndex row_ind, col_ind;
    int64_t rank() const { return c.ncols(); }
    DeviceMatrix<T> to_mat() const { return rusty_compression::dot(c, rusty_compression::dot(x, r)); }             // :62-64
    DeviceMatrix<T> dot(const DeviceMatrix<T> &rhs) const {                                                        // Apply, :154-171
        return rusty_compression::dot(c, rusty_compression::dot(x, rusty_compression::dot(r, rhs)));
    }
};
template <typename T>
struct ColumnID {  // src/col_interp_decomp.rs:23-31
    DeviceMatrix<T> c, z;
    DeviceIndex col_ind;
    int64_t rank() const { return c.ncols(); }
    DeviceMatrix<T> to_mat() const { return rusty_compression::dot(c, z); }                                        // :63-65
    DeviceMatrix<T> dot(const DeviceMatrix<T> &rhs) const { return rusty_compression::dot(c, rusty_compression::dot(z, rhs)); }  // Apply
    TwoSidedID<T> two_sided_id() const {  // src/col_interp_decomp.rs:116-125
        const int64_t m = c.nrows(), k = c.ncols(), kk = m < k ? m : k;
        TwoSidedID<T> out{DeviceMatrix<T>(c.ctx(), m, kk), DeviceMatrix<T>(c.ctx(), kk, k), clone_matrix(z), DeviceIndex(c.ctx(), (std::size_t)m), clone_index(col_ind)};
        c.ctx().check(Api<T>::column_id_two_sided(c.ctx().raw(), c.view(), out.c.view(), out.x.view(), out.row_ind.data()));
        return out;
    }
};
template <typename T>
struct RowID {  // src/row_interp_decomp.rs:25-33
    DeviceMatrix<T> x, r;
    DeviceIndex row_ind;
    int64_t rank() const { return r.nrows(); }
    DeviceMatrix<T> to_mat() const { return rusty_compression::dot(x, r); }
    DeviceMatrix<T> dot(const DeviceMatrix<T> &rhs) const { return rusty_compression::dot(x, rusty_compression::dot(r, rhs)); }
    TwoSidedID<T> two_sided_id() const {  // src/row_interp_decomp.rs:120-130
        const int64_t k = r.nrows(), n = r.ncols(), kk = k < n ? k : n;
        TwoSidedID<T> out{clone_matrix(x), DeviceMatrix<T>(r.ctx(), k, kk), DeviceMatrix<T>(r.ctx(), kk, n), clone_index(row_ind), DeviceIndex(r.ctx(), (std::size_t)n)};
        r.ctx().check(Api<T>::row_id_two_sided(r.ctx().raw(), r.view(), out.x.view(), out.r.view(), out.col_ind.data()));
        return out;
    }
};
template <typename T>
ColumnID<T> QR<T>::column_id() const {
    ColumnID<T> out{DeviceMatrix<T>(q.ctx(), nrows(), rank()), DeviceMatrix<T>(q.ctx(), rank(), ncols()), clone_index(ind)};
    q.ctx().check(Api<T>::qr_column_id(q.ctx().raw(), q.view(), r.view(), ind.data(), out.c.view(), out.z.view()));
    return out;
}
template <typename T>
RowID<T> LQ<T>::row_id() const {
    RowID<T> out{DeviceMatrix<T>(q.ctx(), nrows(), rank()), DeviceMatrix<T>(q.ctx(), rank(), ncols()), clone_index(ind)};
    q.ctx().check(Api<T>::lq_row_id(q.ctx().raw(), l.view(), q.view(), ind.data(), out.x.view(), out.r.view()));
    return out;
}

// ---- svd.rs --------------------------------------------------------------------------------------
template <typename T>
struct SVD {  // src/svd.rs:13-20
    using Real = typename Scalar<T>::real;
    DeviceMatrix<T> u;
    DeviceBuffer<Real> s;  // singular values are real for every scalar type
    DeviceMatrix<T> vt;
    int64_t rank() const { return u.ncols(); }
    static SVD compute_from(const DeviceMatrix<T> &a) {  // src/svd.rs:165-169 -> src/compute_svd.rs:18-27
        const int64_t m = a.nrows(), n = a.ncols(), r = m < n ? m : n;
        SVD out{DeviceMatrix<T>(a.ctx(), m, r), DeviceBuffer<Real>(a.ctx(), (std::size_t)r), DeviceMatrix<T>(a.ctx(), r, n)};
        a.ctx().check(Api<T>::compute_svd(a.ctx().raw(), a.view(), out.u.view(), out.s.data(), out.vt.view()));
        return out;
    }
    static SVD compute_from_range_estimate(const DeviceMatrix<T> &range, const DeviceMatrix<T> &op) {  // src/svd.rs:171-183
        const int64_t m = op.nrows(), n = op.ncols(), r = range.ncols() < n ? range.ncols() : n;
        SVD out{DeviceMatrix<T>(op.ctx(), m, r), DeviceBuffer<Real>(op.ctx(), (std::size_t)r), DeviceMatrix<T>(op.ctx(), r, n)};
        op.ctx().check(Api<T>::svd_from_range_estimate(op.ctx().raw(), range.view(), op.view(), out.u.view(), out.s.data(), out.vt.view()));
        return out;
    }
    DeviceMatrix<T> to_mat() const {  // src/svd.rs:42-54
        DeviceMatrix<T> out(u.ctx(), u.nrows(), vt.ncols());
        u.ctx().check(Api<T>::svd_to_mat(u.ctx().raw(), u.view(), s.data(), vt.view(), out.view()));
        return out;
    }
    QR<T> to_qr() const {  // src/svd.rs:150-163
        const int64_t r = vt.nrows(), n = vt.ncols(), k = r < n ? r : n;
        QR<T> out{DeviceMatrix<T>(u.ctx(), u.nrows(), k), DeviceMatrix<T>(u.ctx(), k, n), DeviceIndex(u.ctx(), (std::size_t)n)};
        u.ctx().check(Api<T>::svd_to_qr(u.ctx().raw(), u.view(), s.data(), vt.view(), out.q.view(), out.r.view(), out.ind.data()));
        return out;
    }
    SVD compress_svd_rank(int64_t max_rank) const {  // src/svd.rs:68-84
        if (max_rank > (int64_t)s.size()) max_rank = (int64_t)s.size();
        auto hs = s.to_host();
        DeviceBuffer<Real> s2(u.ctx(), (std::size_t)max_rank);
        s2.from_host(hs.data());
        return SVD{u.leading(u.nrows(), max_rank), std::move(s2), vt.leading(max_rank, vt.ncols())};
    }
    SVD compress(CompressionType ct) const {  // src/svd.rs:60-101
        if (ct.kind == CompressionType::RANK_) return compress_svd_rank((int64_t)ct.value);
        int64_t rank = -1;
        u.ctx().check(Api<T>::svd_rank_by_tolerance(u.ctx().raw(), s.data(), (int64_t)s.size(), ct.value, &rank));
        return compress_svd_rank(rank);
    }
};

// ---- random_matrix.rs (orthogonal / approximately low-rank test matrices) ---------------------------
template <typename T>
DeviceMatrix<T> transpose(const DeviceMatrix<T> &a) {  // owned C-order copy of a^T
    DeviceMatrix<T> out(a.ctx(), a.ncols(), a.nrows());
    if (a.nrows() == 0 || a.ncols() == 0) return out;
    std::vector<int64_t> h((std::size_t)a.nrows());
    for (int64_t i = 0; i < a.nrows(); ++i) h[(std::size_t)i] = i;
    DeviceIndex idx(a.ctx(), h.size());
    idx.from_host(h.data());
    rc_matrix at{a.view().data, a.ncols(), a.nrows(), 1, a.ncols()};
    a.ctx().check(Api<T>::apply_permutation_matrix(a.ctx().raw(), RC_PERM_COL, at, idx.data(), a.nrows(), out.view()));
    a.ctx().synchronize();
    return out;
}
template <typename T>
DeviceMatrix<T> conj_transpose(const DeviceMatrix<T> &a) {  // owned C-order copy of a^H (= a^T for real scalars)
    if (!Scalar<T>::is_complex) return transpose(a);
    std::vector<T> eye((std::size_t)(a.nrows() * a.nrows()), (T)0);
    for (int64_t i = 0; i < a.nrows(); ++i) eye[(std::size_t)(i * a.nrows() + i)] = (T)1;
    return dot_op(2, a, 0, DeviceMatrix<T>::from_host(a.ctx(), eye.data(), a.nrows(), a.nrows()));
}
template <typename T>
DeviceMatrix<T> random_orthogonal_matrix(const Context &ctx, int64_t rows, int64_t cols, uint64_t seed, uint64_t offset = 0) {  // src/random_matrix.rs:35-56
    const bool swap = cols > rows;
    auto g = random_gaussian<T>(ctx, swap ? cols : rows, swap ? rows : cols, seed, offset);
    auto u = std::move(SVD<T>::compute_from(g).u);
    return swap ? conj_transpose(u) : std::move(u);  // src/random_matrix.rs:51-53
}
template <typename T>
DeviceMatrix<T> random_approximate_low_rank_matrix(const Context &ctx, int64_t rows, int64_t cols, double sigma_max, double sigma_min,
                                                   uint64_t seed) {  // src/random_matrix.rs:70-93
    if (!(sigma_min < sigma_max)) throw AssertionFailed("`sigma_min` must be smaller than `sigma_max`");
    if (!(sigma_min > 0.0)) throw AssertionFailed("`sigma_min` must be positive.");
    using Real = typename Scalar<T>::real;
    const int64_t r = rows < cols ? rows : cols;
    std::vector<Real> hs((std::size_t)r);
    const double l0 = std::log10(sigma_min), l1 = std::log10(sigma_max);
    for (int64_t i = 0; i < r; ++i) hs[(std::size_t)i] = (Real)std::pow(10.0, r > 1 ? l0 + (l1 - l0) * (double)i / (double)(r - 1) : l0);
    SVD<T> f{random_orthogonal_matrix<T>(ctx, rows, r, seed, 0), DeviceBuffer<Real>(ctx, (std::size_t)r),
             random_orthogonal_matrix<T>(ctx, r, cols, seed, (uint64_t)(rows * r))};
    f.s.from_host(hs.data());
    return f.to_mat();
}

// ---- random_sampling.rs -----------------------------------------------------------------------------
template <typename T>
DeviceMatrix<T> sample_range_by_rank(const DeviceMatrix<T> &op, int64_t k, int64_t p, uint64_t seed) {  // :103-118
    int64_t kk = k < op.nrows() ? k : op.nrows();
    if (k + p < kk) kk = k + p;
    DeviceMatrix<T> q(op.ctx(), op.nrows(), kk);
    op.ctx().check(Api<T>::sample_range_by_rank(op.ctx().raw(), op.view(), k, p, rc_matrix{nullptr, 0, 0, 0, 0}, seed, q.view()));
    return q;
}
template <typename T>
DeviceMatrix<T> sample_range_power_iteration(const DeviceMatrix<T> &op, int64_t k, int64_t p, int64_t it_count, uint64_t seed) {  // :131-160
    int64_t kk = k < op.nrows() ? k : op.nrows();
    if (k + p < kk) kk = k + p;
    if (op.ncols() < kk) kk = op.ncols();
    DeviceMatrix<T> q(op.ctx(), op.nrows(), kk);
    op.ctx().check(Api<T>::sample_range_power_iteration(op.ctx().raw(), op.view(), k, p, it_count, rc_matrix{nullptr, 0, 0, 0, 0}, seed, q.view()));
    return q;
}
// the unit of work of batches (examples/interpolative_decomposition.rs:25-32 in one call, truncated factorization)
template <typename T>
ColumnID<T> column_id_rank(const DeviceMatrix<T> &a, int64_t k) {
    const int64_t kk = k < a.nrows() ? (k < a.ncols() ? k : a.ncols()) : (a.nrows() < a.ncols() ? a.nrows() : a.ncols());
    ColumnID<T> out{DeviceMatrix<T>(a.ctx(), a.nrows(), kk), DeviceMatrix<T>(a.ctx(), kk, a.ncols()), DeviceIndex(a.ctx(), (std::size_t)a.ncols())};
    a.ctx().check(Api<T>::column_id_rank(a.ctx().raw(), a.view(), kk, out.c.view(), out.z.view(), out.col_ind.data()));
    return out;
}
template <typename T>
typename Scalar<T>::real max_col_norm(const DeviceMatrix<T> &y) {  // :184-191
    typename Scalar<T>::real out = 0;
    y.ctx().check(Api<T>::max_col_norm(y.ctx().raw(), y.view(), &out));
    return out;
}
template <typename T>
struct AdaptiveResult {
    DeviceMatrix<T> q;
    std::vector<std::pair<std::size_t, double>> residuals;  // Vec<(usize, f64)>
};
template <typename T>
AdaptiveResult<T> sample_range_adaptive(const DeviceMatrix<T> &op, double rel_tol, int64_t sample_size, uint64_t seed, int64_t max_rank = -1) {  // :223-274
    const int64_t m = op.nrows(), n = op.ncols();
    if (max_rank < 0) max_rank = (((m < n ? m : n) + sample_size - 1) / sample_size) * sample_size;
    DeviceMatrix<T> qcap(op.ctx(), m, max_rank);
    const int64_t hist_cap = max_rank / (sample_size < 1 ? 1 : sample_size) + 2;
    std::vector<int64_t> hrank((std::size_t)hist_cap);
    std::vector<double> hres((std::size_t)hist_cap);
    int64_t rank = 0, hlen = 0;
    op.ctx().check(Api<T>::sample_range_adaptive(op.ctx().raw(), op.view(), rel_tol, sample_size, rc_matrix{nullptr, 0, 0, 0, 0}, seed, qcap.view(),
                                                 &rank, hrank.data(), hres.data(), hist_cap, &hlen));
    AdaptiveResult<T> out{qcap.leading(m, rank), {}};
    for (int64_t i = 0; i < hlen; ++i) out.residuals.emplace_back((std::size_t)hrank[(std::size_t)i], hres[(std::size_t)i]);
    return out;
}

// ---- operators: MatVec / ConjMatVec / MatMat / ConjMatMat (src/types.rs:40-101) -------------------------------------------
// The reference implements its range finders for ANY operator (`impl<Op: MatMat<A = $scalar>> SampleRange for Op`,
// src/random_sampling.rs:102, :130, :222; compute_from_range_estimate<Op: ConjMatMat>, src/qr.rs:311-323, src/svd.rs:171-183).
// Here an operator is any class with
//     int64_t nrows() const;  int64_t ncols() const;                                   MatVec::nrows / ncols
//     void matmat(const Context &ctx, rc_matrix x, rc_matrix y) const;                 y (nrows x s) = A x     (MatMat)
//     void conj_matmat(const Context &ctx, rc_matrix x, rc_matrix y) const;  [optional] y (ncols x s) = A^H x  (ConjMatMat)
// x, y are strided DEVICE views; the products are enqueued on ctx (through this library or with the host's own kernels on
// rc_get_stream).  The overloads below hand the library an rc_operator whose callbacks forward to those members; everything
// else (Omega, pivoted QR, SVD, the adaptive loop) runs inside the library exactly as for a dense matrix.
template <typename T> struct ApiOp;
#define RC_API_OP(T, SUF)                                                                          \
    template <> struct ApiOp<T> {                                                                  \
        static constexpr auto sample_range_by_rank = rc_sample_range_by_rank_op_##SUF;             \
        static constexpr auto sample_range_power_iteration = rc_sample_range_power_iteration_op_##SUF; \
        static constexpr auto sample_range_adaptive = rc_sample_range_adaptive_op_##SUF;           \
        static constexpr auto qr_from_range_estimate = rc_qr_from_range_estimate_op_##SUF;         \
        static constexpr auto svd_from_range_estimate = rc_svd_from_range_estimate_op_##SUF;       \
    };
RC_API_OP(double, f64)
RC_API_OP(float, f32)
RC_API_OP(c64, c64)
RC_API_OP(c32, c32)
#undef RC_API_OP

namespace detail {
template <class Op, class = void> struct has_conj_matmat : std::false_type {};
template <class Op>
struct has_conj_matmat<Op, std::void_t<decltype(std::declval<const Op &>().conj_matmat(std::declval<const Context &>(), rc_matrix{}, rc_matrix{}))>> : std::true_type {};

inline int32_t status_of_current_exception() {  // a callback must not unwind into C: exceptions become the status the entry point returns
    try { throw; }
    catch (const CompressionError &) { return RC_COMPRESSION_ERROR; }
    catch (const LayoutError &) { return RC_LAYOUT_ERROR; }
    catch (const PivotedQRError &) { return RC_PIVOTED_QR_ERROR; }
    catch (const LinalgError &) { return RC_LINALG_ERROR; }
    catch (const AssertionFailed &) { return RC_INVALID_ARGUMENT; }
    catch (...) { return RC_RUNTIME_ERROR; }
}
}  // namespace detail

// rc_operator of a C++ operator object (borrowed: `op` and `ctx` must outlive the call it is passed to)
template <class Op>
class OperatorTable {
  public:
    OperatorTable(const Context &ctx, const Op &op) : ctx_(&ctx), op_(&op) {
        table_.rows = op.nrows();
        table_.cols = op.ncols();
        table_.matmat = &OperatorTable::matmat_cb;
        table_.conj_matmat = detail::has_conj_matmat<Op>::value ? &OperatorTable::conj_matmat_cb : nullptr;
        table_.user = this;
    }
    const rc_operator *get() const { return &table_; }

  private:
    static int32_t matmat_cb(void *user, rc_context *, rc_matrix x, rc_matrix y) {
        auto *self = static_cast<OperatorTable *>(user);
        try { self->op_->matmat(*self->ctx_, x, y); return RC_OK; } catch (...) { return detail::status_of_current_exception(); }
    }
    static int32_t conj_matmat_cb(void *user, rc_context *, rc_matrix x, rc_matrix y) {
        auto *self = static_cast<OperatorTable *>(user);
        try {
            if constexpr (detail::has_conj_matmat<Op>::value) self->op_->conj_matmat(*self->ctx_, x, y);
            return RC_OK;
        } catch (...) { return detail::status_of_current_exception(); }
    }
    const Context *ctx_;
    const Op *op_;
    rc_operator table_{};
};

// a dense device matrix behind the operator interface (each product is the library's own GEMM on the views it is handed)
template <typename T>
struct DenseOperator {
    const DeviceMatrix<T> *a;
    int64_t nrows() const { return a->nrows(); }
    int64_t ncols() const { return a->ncols(); }
    void matmat(const Context &ctx, rc_matrix x, rc_matrix y) const { ctx.check(Api<T>::matmat(ctx.raw(), a->view(), x, y)); }
    void conj_matmat(const Context &ctx, rc_matrix x, rc_matrix y) const { ctx.check(Api<T>::conj_matmat(ctx.raw(), a->view(), x, y)); }
};
// A = U V^H given by its factors (U: m x r, V: n x r) and never formed: two skinny GEMMs per product
template <typename T>
struct LowRankOperator {
    const DeviceMatrix<T> *u, *v;
    int64_t nrows() const { return u->nrows(); }
    int64_t ncols() const { return v->nrows(); }
    void matmat(const Context &ctx, rc_matrix x, rc_matrix y) const {
        DeviceMatrix<T> t(ctx, v->ncols(), x.cols);
        ctx.check(Api<T>::gemm(ctx.raw(), 2, 0, Scalar<T>::wire((T)1), v->view(), x, Scalar<T>::wire((T)0), t.view()));
        ctx.check(Api<T>::gemm(ctx.raw(), 0, 0, Scalar<T>::wire((T)1), u->view(), t.view(), Scalar<T>::wire((T)0), y));
        ctx.synchronize();  // t is freed on return
    }
    void conj_matmat(const Context &ctx, rc_matrix x, rc_matrix y) const {
        DeviceMatrix<T> t(ctx, u->ncols(), x.cols);
        ctx.check(Api<T>::gemm(ctx.raw(), 2, 0, Scalar<T>::wire((T)1), u->view(), x, Scalar<T>::wire((T)0), t.view()));
        ctx.check(Api<T>::gemm(ctx.raw(), 0, 0, Scalar<T>::wire((T)1), v->view(), t.view(), Scalar<T>::wire((T)0), y));
        ctx.synchronize();
    }
};

// impl<Op: MatMat> SampleRange for Op (src/random_sampling.rs:102-121)
template <typename T, class Op>
DeviceMatrix<T> sample_range_by_rank(const Context &ctx, const Op &op, int64_t k, int64_t p, uint64_t seed) {
    int64_t kk = k < op.nrows() ? k : op.nrows();
    if (k + p < kk) kk = k + p;
    DeviceMatrix<T> q(ctx, op.nrows(), kk);
    OperatorTable<Op> tab(ctx, op);
    ctx.check(ApiOp<T>::sample_range_by_rank(ctx.raw(), tab.get(), k, p, rc_matrix{nullptr, 0, 0, 0, 0}, seed, q.view()));
    return q;
}
// impl<Op: MatMat + ConjMatMat> SampleRangePowerIteration for Op (src/random_sampling.rs:130-163)
template <typename T, class Op>
DeviceMatrix<T> sample_range_power_iteration(const Context &ctx, const Op &op, int64_t k, int64_t p, int64_t it_count, uint64_t seed) {
    int64_t kk = k < op.nrows() ? k : op.nrows();
    if (k + p < kk) kk = k + p;
    if (op.ncols() < kk) kk = op.ncols();
    DeviceMatrix<T> q(ctx, op.nrows(), kk);
    OperatorTable<Op> tab(ctx, op);
    ctx.check(ApiOp<T>::sample_range_power_iteration(ctx.raw(), tab.get(), k, p, it_count, rc_matrix{nullptr, 0, 0, 0, 0}, seed, q.view()));
    return q;
}
// impl<Op: MatMat + ConjMatMat> AdaptiveSampling for Op (src/random_sampling.rs:222-277)
template <typename T, class Op>
AdaptiveResult<T> sample_range_adaptive(const Context &ctx, const Op &op, double rel_tol, int64_t sample_size, uint64_t seed, int64_t max_rank = -1) {
    const int64_t m = op.nrows(), n = op.ncols();
    if (max_rank < 0) max_rank = (((m < n ? m : n) + sample_size - 1) / sample_size) * sample_size;
    DeviceMatrix<T> qcap(ctx, m, max_rank);
    const int64_t hist_cap = max_rank / (sample_size < 1 ? 1 : sample_size) + 2;
    std::vector<int64_t> hrank((std::size_t)hist_cap);
    std::vector<double> hres((std::size_t)hist_cap);
    int64_t rank = 0, hlen = 0;
    OperatorTable<Op> tab(ctx, op);
    ctx.check(ApiOp<T>::sample_range_adaptive(ctx.raw(), tab.get(), rel_tol, sample_size, rc_matrix{nullptr, 0, 0, 0, 0}, seed, qcap.view(), &rank, hrank.data(),
                                              hres.data(), hist_cap, &hlen));
    AdaptiveResult<T> out{qcap.leading(m, rank), {}};
    for (int64_t i = 0; i < hlen; ++i) out.residuals.emplace_back((std::size_t)hrank[(std::size_t)i], hres[(std::size_t)i]);
    return out;
}
// QRTraits / SVDTraits::compute_from_range_estimate<Op: ConjMatMat> (src/qr.rs:311-323, src/svd.rs:171-183)
template <typename T, class Op>
QR<T> qr_from_range_estimate(const Context &ctx, const DeviceMatrix<T> &range, const Op &op) {
    const int64_t m = op.nrows(), n = op.ncols(), k = range.ncols() < n ? range.ncols() : n;
    QR<T> out{DeviceMatrix<T>(ctx, m, k), DeviceMatrix<T>(ctx, k, n), DeviceIndex(ctx, (std::size_t)n)};
    OperatorTable<Op> tab(ctx, op);
    ctx.check(ApiOp<T>::qr_from_range_estimate(ctx.raw(), range.view(), tab.get(), out.q.view(), out.r.view(), out.ind.data()));
    return out;
}
template <typename T, class Op>
SVD<T> svd_from_range_estimate(const Context &ctx, const DeviceMatrix<T> &range, const Op &op) {
    using Real = typename Scalar<T>::real;
    const int64_t m = op.nrows(), n = op.ncols(), r = range.ncols() < n ? range.ncols() : n;
    SVD<T> out{DeviceMatrix<T>(ctx, m, r), DeviceBuffer<Real>(ctx, (std::size_t)r), DeviceMatrix<T>(ctx, r, n)};
    OperatorTable<Op> tab(ctx, op);
    ctx.check(ApiOp<T>::svd_from_range_estimate(ctx.raw(), range.view(), tab.get(), out.u.view(), out.s.data(), out.vt.view()));
    return out;
}

// ---- one matrix sharded by rows over several ranks (rc_rsvd_id_row_sharded_*; the reference is single-process) ----------
// A communicator over RCCL (one process per GPU) or over the host's own communication layer (callbacks on host pointers).
class Comm {
  public:
    static Comm rccl(int32_t world, int32_t rank, const void *id128, int32_t device) {
        Comm c;
        if (rc_comm_init(&c.raw_, world, rank, id128, device) != RC_OK) throw HipRuntimeError(rc_comm_last_error_message(nullptr));
        return c;
    }
    static Comm host(int32_t world, int32_t rank, int32_t device, rc_host_all_gather_fn all_gather, rc_host_all_reduce_sum_fn all_reduce_sum, void *user) {
        Comm c;
        if (rc_comm_init_host(&c.raw_, world, rank, device, all_gather, all_reduce_sum, user) != RC_OK) throw AssertionFailed("rc_comm_init_host: bad arguments");
        return c;
    }
    Comm(Comm &&o) noexcept : raw_(o.raw_) { o.raw_ = nullptr; }
    Comm &operator=(Comm &&o) noexcept { if (this != &o) { rc_comm_destroy(raw_); raw_ = o.raw_; o.raw_ = nullptr; } return *this; }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    ~Comm() { rc_comm_destroy(raw_); }
    rc_comm *raw() const { return raw_; }

  private:
    Comm() = default;
    rc_comm *raw_ = nullptr;
};
// this rank's rows of the global factors (range_q, u, qr_q, c) and the replicated ones (s, vt, r, ind, z)
template <typename T>
struct ShardedRsvdId {
    DeviceMatrix<T> range_q, u;
    DeviceBuffer<T> s;
    DeviceMatrix<T> vt, qr_q, r;
    DeviceIndex ind;
    DeviceMatrix<T> c, z;
};
inline rc_status rsvd_id_row_sharded_raw(rc_comm *cm, rc_context *cx, rc_matrix a, int64_t k, int64_t p, uint64_t seed, const rc_rsvd_id_out *o, double) {
    return rc_rsvd_id_row_sharded_f64(cm, cx, a, k, p, seed, o);
}
inline rc_status rsvd_id_row_sharded_raw(rc_comm *cm, rc_context *cx, rc_matrix a, int64_t k, int64_t p, uint64_t seed, const rc_rsvd_id_out *o, float) {
    return rc_rsvd_id_row_sharded_f32(cm, cx, a, k, p, seed, o);
}
// a_local: the m_r x n row block of this rank (m_r >= k + p); comm == nullptr: one rank.  Real scalar types.
template <typename T>
ShardedRsvdId<T> rsvd_id_row_sharded(const Comm *comm, const DeviceMatrix<T> &a_local, int64_t k, int64_t p, uint64_t seed) {
    const Context &cx = a_local.ctx();
    const int64_t mr = a_local.nrows(), n = a_local.ncols();
    ShardedRsvdId<T> out{DeviceMatrix<T>(cx, mr, k), DeviceMatrix<T>(cx, mr, k), DeviceBuffer<T>(cx, (std::size_t)k), DeviceMatrix<T>(cx, k, n), DeviceMatrix<T>(cx, mr, k),
                         DeviceMatrix<T>(cx, k, n), DeviceIndex(cx, (std::size_t)n), DeviceMatrix<T>(cx, mr, k), DeviceMatrix<T>(cx, k, n)};
    rc_rsvd_id_out o{out.range_q.view(), out.u.view(), out.s.data(), out.vt.view(), out.qr_q.view(), out.r.view(), out.ind.data(), out.c.view(), out.z.view()};
    cx.check(rsvd_id_row_sharded_raw(comm ? comm->raw() : nullptr, cx.raw(), a_local.view(), k, p, seed, &o, T()));
    return out;
}

}  // namespace rusty_compression
