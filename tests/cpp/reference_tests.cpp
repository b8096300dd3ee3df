// The reference crate's unit tests (rusty-compression v0.1.1) written against the C++ mirror include/rusty_compression.hpp:
//   src/pivoted_qr.rs:193-317, src/qr.rs:418-616, src/svd.rs:193-321, src/col_interp_decomp.rs:163-242,
//   src/row_interp_decomp.rs:163-236, src/permutation.rs:187-240.
// Same test names, same matrices (the reference's generator recipe, seeded), same assertions and tolerances, for all four
// scalar types of the reference (f32, f64, c32, c64): 11 tests x 4 types x 2 shapes + the 2 permutation tests = 90, plus one test of the row-sharded call (two ranks = two threads, host communicator).  This is the COMPILED twin of bindings/rust/tests/reference_tests.rs (the
// Rust crate cannot be built in this repository's container): tests/test_gpu_parity.py builds it and runs it on the GPU box,
// the CPU suite compiles and links it.  Prints one line per test and exits non-zero if any assertion failed.
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rusty_compression.hpp"

using namespace rusty_compression;

static int failures = 0, tests_run = 0;
static bool current_ok = true;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) { current_ok = false; std::printf("    assertion failed: %s (line %d)\n", #cond, __LINE__); } \
    } while (0)

template <typename F>
static void run_test(const std::string &name, F &&body) {
    current_ok = true;
    try {
        body();
    } catch (const std::exception &e) {
        current_ok = false;
        std::printf("    exception: %s\n", e.what());
    }
    ++tests_run;
    if (!current_ok) ++failures;
    std::printf("test %s ... %s\n", name.c_str(), current_ok ? "ok" : "FAILED");
}

template <typename T> struct Name;
template <> struct Name<float> { static const char *get() { return "f32"; } };
template <> struct Name<double> { static const char *get() { return "f64"; } };
template <> struct Name<c32> { static const char *get() { return "c32"; } };
template <> struct Name<c64> { static const char *get() { return "c64"; } };
// |x| as a double for real and complex entries
static double mag(float x) { return std::fabs((double)x); }
static double mag(double x) { return std::fabs(x); }
static double mag(c32 x) { return (double)std::abs(x); }
static double mag(c64 x) { return std::abs(x); }

static uint64_t seed_of(const std::string &s) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char c : s) h = (h ^ c) * 0x100000001b3ull;
    return h;
}

// host copies for the element-wise assertions of the reference's tests
template <typename T>
struct Host {
    std::vector<T> d;
    int64_t rows, cols;
    T at(int64_t i, int64_t j) const { return d[(size_t)(i * cols + j)]; }
};
template <typename T>
static Host<T> host(const DeviceMatrix<T> &m) { return Host<T>{m.to_host(), m.nrows(), m.ncols()}; }
static std::vector<int64_t> host_index(const DeviceIndex &ix) { return ix.to_host(); }

template <typename T>
static double rel_diff_col(const Host<T> &a, int64_t ca, const Host<T> &b, int64_t cb) {  // RelDiff::rel_diff_l2 on columns
    double d2 = 0, n2 = 0;
    for (int64_t i = 0; i < a.rows; ++i) {
        const double d = mag((T)(a.at(i, ca) - b.at(i, cb))), y = mag(b.at(i, cb));
        d2 += d * d;
        n2 += y * y;
    }
    return std::sqrt(d2) / std::sqrt(n2);
}
template <typename T>
static double rel_diff_row(const Host<T> &a, int64_t ra, const Host<T> &b, int64_t rb) {
    double d2 = 0, n2 = 0;
    for (int64_t j = 0; j < a.cols; ++j) {
        const double d = mag((T)(a.at(ra, j) - b.at(rb, j))), y = mag(b.at(rb, j));
        d2 += d * d;
        n2 += y * y;
    }
    return std::sqrt(d2) / std::sqrt(n2);
}

template <typename T>
static void group(const Context &ctx, int64_t m, int64_t n, const char *shape) {
    const std::string suf = std::string(Name<T>::get()) + "_" + shape;
    const bool f32 = sizeof(typename Scalar<T>::real) == 4;
    auto mat_for = [&](const std::string &test, double smin) { return random_approximate_low_rank_matrix<T>(ctx, m, n, 1.0, smin, seed_of(test)); };

    // ---- src/pivoted_qr.rs:198-246
    run_test("pivoted_qr_test_" + std::string(shape) + "_" + Name<T>::get(), [&] {
        auto mat = mat_for("pivoted_qr_test_" + suf, 1E-5);
        auto qr = QR<T>::compute_from(mat);
        auto qtq = host(dot_op(2, qr.q, 0, qr.q));  // Q^H Q
        for (int64_t i = 0; i < qtq.rows; ++i)
            for (int64_t j = 0; j < qtq.cols; ++j) CHECK(mag((T)(qtq.at(i, j) - (T)(i == j ? 1.0 : 0.0))) < 1E-6);
        auto prod = host(dot(qr.q, qr.r)), hm = host(mat);
        auto ind = host_index(qr.ind);
        for (int64_t c = 0; c < prod.cols; ++c) CHECK(rel_diff_col(prod, c, hm, ind[(size_t)c]) < 1E-6);
    });
    // ---- src/pivoted_qr.rs:248-294
    run_test("pivoted_lq_test_" + std::string(shape) + "_" + Name<T>::get(), [&] {
        auto mat = mat_for("pivoted_lq_test_" + suf, 1E-5);
        auto lq = LQ<T>::compute_from(mat);
        auto qqt = host(dot_op(0, lq.q, 2, lq.q));  // Q Q^H
        for (int64_t i = 0; i < qqt.rows; ++i)
            for (int64_t j = 0; j < qqt.cols; ++j) CHECK(mag((T)(qqt.at(i, j) - (T)(i == j ? 1.0 : 0.0))) < 1E-6);
        auto prod = host(dot(lq.l, lq.q)), hm = host(mat);
        auto ind = host_index(lq.ind);
        for (int64_t r = 0; r < prod.rows; ++r) CHECK(rel_diff_row(prod, r, hm, ind[(size_t)r]) < 1E-6);
    });
    // ---- src/qr.rs:427-457
    run_test("test_qr_compression_by_rank_" + suf, [&] {
        const int64_t rank = 30;
        auto mat = mat_for("test_qr_compression_by_rank_" + suf, 1E-10);
        auto qr = QR<T>::compute_from(mat).compress(CompressionType::RANK(rank));
        CHECK(qr.q.ncols() == rank);
        CHECK(qr.r.nrows() == rank);
        CHECK((double)rel_diff_fro(qr.to_mat(), mat) < 1E-4);
    });
    // ---- src/qr.rs:459-489
    run_test("test_qr_compression_by_tol_" + suf, [&] {
        const double tol = 1E-4;
        auto mat = mat_for("test_qr_compression_by_tol_" + suf, 1E-10);
        auto qr = QR<T>::compute_from(mat).compress(CompressionType::ADAPTIVE(tol));
        CHECK((double)rel_diff_fro(qr.to_mat(), mat) < 5.0 * tol);
        CHECK(qr.q.ncols() < std::min(m, n));
    });
    // ---- src/qr.rs:491-530
    run_test("test_col_id_compression_by_tol_" + suf, [&] {
        const double tol = 1E-4;
        auto mat = mat_for("test_col_id_compression_by_tol_" + suf, 1E-10);
        auto qr = QR<T>::compute_from(mat).compress(CompressionType::ADAPTIVE(tol));
        const int64_t rank = qr.rank();
        auto cid = qr.column_id();
        CHECK((double)rel_diff_fro(cid.to_mat(), mat) < 5.0 * tol);
        auto hm = host(mat), hc = host(cid.c);
        auto ind = host_index(cid.col_ind);
        for (int64_t i = 0; i < rank; ++i) CHECK(rel_diff_col(hm, ind[(size_t)i], hc, i) < tol);  // mat.apply_permutation(col_ind, COL)[:, i] vs C[:, i]
    });
    // ---- src/qr.rs:532-571
    run_test("test_row_id_compression_by_tol_" + suf, [&] {
        const double tol = 1E-4;
        auto mat = mat_for("test_row_id_compression_by_tol_" + suf, 1E-10);
        auto lq = LQ<T>::compute_from(mat).compress(CompressionType::ADAPTIVE(tol));
        const int64_t rank = lq.rank();
        auto rid = lq.row_id();
        CHECK((double)rel_diff_fro(rid.to_mat(), mat) < 5.0 * tol);
        auto hm = host(mat), hr = host(rid.r);
        auto ind = host_index(rid.row_ind);
        for (int64_t i = 0; i < rank; ++i) CHECK(rel_diff_row(hm, ind[(size_t)i], hr, i) < tol);
    });
    // ---- src/svd.rs:203-227
    run_test("test_svd_to_qr_" + suf, [&] {
        auto mat = mat_for("test_svd_to_qr_" + suf, 1E-10);
        auto svd = SVD<T>::compute_from(mat);
        auto actual = svd.to_qr().to_mat();
        CHECK((double)rel_diff_fro(actual, mat) < (f32 ? 1E-5 : 1E-12));
    });
    // ---- src/svd.rs:229-259
    run_test("test_svd_compression_by_rank_" + suf, [&] {
        const int64_t rank = 20;
        auto mat = mat_for("test_svd_compression_by_rank_" + suf, 1E-10);
        auto svd = SVD<T>::compute_from(mat).compress(CompressionType::RANK(rank));
        CHECK(svd.u.ncols() == rank);
        CHECK(svd.vt.nrows() == rank);
        CHECK((double)rel_diff_fro(svd.to_mat(), mat) < 1E-4);
    });
    // ---- src/svd.rs:261-287
    run_test("test_svd_compression_by_tol_" + suf, [&] {
        const double tol = 1E-4;
        auto mat = mat_for("test_svd_compression_by_tol_" + suf, 1E-10);
        auto svd = SVD<T>::compute_from(mat).compress(CompressionType::ADAPTIVE(tol));
        CHECK((double)rel_diff_fro(svd.to_mat(), mat) < tol);
    });
    // ---- src/col_interp_decomp.rs:176-230 and src/row_interp_decomp.rs:176-224
    auto two_sided_check = [&](const TwoSidedID<T> &ts, const DeviceMatrix<T> &mat, int64_t rank, double tol) {
        CHECK((double)rel_diff_fro(ts.to_mat(), mat) < 5.0 * tol);
        CHECK(ts.x.nrows() == ts.x.ncols());
        CHECK(ts.x.nrows() == rank);
        auto hm = host(mat), hx = host(ts.x);
        auto ri = host_index(ts.row_ind), ci = host_index(ts.col_ind);
        for (int64_t i = 0; i < rank; ++i)
            for (int64_t j = 0; j < rank; ++j) {
                const T ref = hm.at(ri[(size_t)i], ci[(size_t)j]);  // mat.apply_permutation(row_ind, ROW).apply_permutation(col_ind, COL)[i, j]
                CHECK(mag((T)(hx.at(i, j) - ref)) < 10.0 * tol * mag(ref));
            }
    };
    run_test("test_two_sided_from_col_id_compression_by_tol_" + suf, [&] {
        const double tol = 1E-4;
        auto mat = mat_for("test_two_sided_from_col_id_compression_by_tol_" + suf, 1E-10);
        auto qr = QR<T>::compute_from(mat).compress(CompressionType::ADAPTIVE(tol));
        two_sided_check(qr.column_id().two_sided_id(), mat, qr.rank(), tol);
    });
    run_test("test_two_sided_from_row_id_compression_by_tol_" + suf, [&] {
        const double tol = (f32 && m < n) ? 5E-4 : 1E-4;  // the reference relaxes f32 thick (src/row_interp_decomp.rs:231)
        auto mat = mat_for("test_two_sided_from_row_id_compression_by_tol_" + suf, 1E-10);
        auto lq = LQ<T>::compute_from(mat).compress(CompressionType::ADAPTIVE(tol));
        two_sided_check(lq.row_id().two_sided_id(), mat, lq.rank(), tol);
    });
}

// ---- src/permutation.rs:192-239 (known answers)
static void permutation_tests(const Context &ctx) {
    run_test("test_matrix_permutation", [&] {
        const double m[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9};
        const int64_t perm[3] = {2, 0, 1};
        auto mat = DeviceMatrix<double>::from_host(ctx, m, 3, 3);
        DeviceIndex idx(ctx, 3);
        idx.from_host(perm);
        const double want[4][9] = {{3, 1, 2, 6, 4, 5, 9, 7, 8}, {7, 8, 9, 1, 2, 3, 4, 5, 6}, {2, 3, 1, 5, 6, 4, 8, 9, 7}, {4, 5, 6, 7, 8, 9, 1, 2, 3}};
        const int modes[4] = {RC_PERM_COL, RC_PERM_ROW, RC_PERM_COLINV, RC_PERM_ROWINV};
        for (int k = 0; k < 4; ++k) {
            DeviceMatrix<double> out(ctx, 3, 3);
            ctx.check(rc_apply_permutation_matrix_f64(ctx.raw(), modes[k], mat.view(), idx.data(), 3, out.view()));
            auto h = out.to_host();
            for (int e = 0; e < 9; ++e) CHECK(h[(size_t)e] == want[k][e]);
        }
    });
    run_test("test_vector_permutaiton", [&] {
        const double v[3] = {1, 2, 3};
        const int64_t perm[3] = {2, 0, 1};
        auto vec = DeviceMatrix<double>::from_host(ctx, v, 3, 1);
        DeviceIndex idx(ctx, 3);
        idx.from_host(perm);
        const double want[2][3] = {{3, 1, 2}, {2, 3, 1}};
        const int modes[2] = {RC_VPERM_NOINV, RC_VPERM_INV};
        for (int k = 0; k < 2; ++k) {
            DeviceMatrix<double> out(ctx, 3, 1);
            ctx.check(rc_apply_permutation_vector_f64(ctx.raw(), modes[k], vec.view(), idx.data(), 3, out.view()));
            auto h = out.to_host();
            for (int e = 0; e < 3; ++e) CHECK(h[(size_t)e] == want[k][e]);
        }
    });
}

// ---- not a reference test: the row-sharded call driven from C++ by two ranks = two threads of this process, each with its own
// context on GPU 0 and a HOST communicator whose callbacks exchange through shared memory (what an MPI host would plug in)
struct TwoRankBus {
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, generation = 0;
    std::vector<char> gather;
    std::vector<double> sum;
    void barrier(std::unique_lock<std::mutex> &lk) {
        const int gen = generation;
        if (++arrived == 2) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
struct RankView { TwoRankBus *bus; int rank; };
static int32_t bus_all_gather(void *user, const void *send, void *recv, size_t bytes) {
    auto *rv = static_cast<RankView *>(user);
    std::unique_lock<std::mutex> lk(rv->bus->mu);
    if (rv->bus->gather.size() != 2 * bytes) rv->bus->gather.assign(2 * bytes, 0);
    std::memcpy(rv->bus->gather.data() + (size_t)rv->rank * bytes, send, bytes);
    rv->bus->barrier(lk);
    std::memcpy(recv, rv->bus->gather.data(), 2 * bytes);
    rv->bus->barrier(lk);  // nobody overwrites the buffer before both have read it
    return 0;
}
static int32_t bus_all_reduce(void *user, void *buf, size_t count, int32_t elem_size) {
    auto *rv = static_cast<RankView *>(user);
    if (elem_size != 8) return 1;
    double *x = static_cast<double *>(buf);
    std::unique_lock<std::mutex> lk(rv->bus->mu);
    if (rv->rank == 0) rv->bus->sum.assign(x, x + count);
    rv->bus->barrier(lk);
    if (rv->rank == 1) for (size_t i = 0; i < count; ++i) rv->bus->sum[i] += x[i];  // one fixed order: the same bits on both ranks
    rv->bus->barrier(lk);
    std::memcpy(x, rv->bus->sum.data(), count * sizeof(double));
    rv->bus->barrier(lk);
    return 0;
}
static void sharded_test(const Context &ctx) {
    run_test("row_sharded_rsvd_id_two_ranks_over_a_host_communicator", [&] {
        const int64_t m = 640, n = 300, k = 24, p = 6;
        const uint64_t seed = 11;
        auto a = random_approximate_low_rank_matrix<double>(ctx, m, n, 1.0, 1e-9, seed_of("sharded"));
        auto ah = a.to_host();
        auto range1 = sample_range_by_rank(a, k, p, seed);
        auto svd1 = SVD<double>::compute_from_range_estimate(range1, a);
        auto s1 = svd1.s.to_host();
        TwoRankBus bus;
        std::vector<double> s[2], z[2];
        std::vector<int64_t> ind[2];
        std::string err[2];
        auto rank_body = [&](int rank) {
            try {
                Context cx(0);
                RankView rv{&bus, rank};
                Comm comm = Comm::host(2, rank, 0, bus_all_gather, bus_all_reduce, &rv);
                const int64_t r0 = rank * (m / 2), rows = m / 2;
                auto al = DeviceMatrix<double>::from_host(cx, ah.data() + (size_t)r0 * n, rows, n);
                auto res = rsvd_id_row_sharded(&comm, al, k, p, seed);
                cx.synchronize();
                s[rank] = res.s.to_host();
                z[rank] = res.z.to_host();
                ind[rank] = res.ind.to_host();
            } catch (const std::exception &e) { err[rank] = e.what(); }
        };
        std::thread t1(rank_body, 1);
        rank_body(0);
        t1.join();
        CHECK(err[0].empty() && err[1].empty());
        if (!err[0].empty() || !err[1].empty()) { std::printf("    rank errors: '%s' '%s'\n", err[0].c_str(), err[1].c_str()); return; }
        CHECK(s[0] == s[1] && z[0] == z[1] && ind[0] == ind[1]);  // replicated outputs: the same bits
        double worst = 0;
        for (int64_t i = 0; i < k; ++i) worst = std::fmax(worst, std::fabs(s[0][(size_t)i] - s1[(size_t)i]) / s1[0]);
        CHECK(worst <= 1e-10);  // singular values of the single-context pipeline with the same Omega stream
    });
}

// The LAPACK seam (rc_geqp3 / rc_orgqr / rc_trsm_upper where the reference calls $qrf, lax::Lapack::q and solve_triangular:
// src/pivoted_qr.rs:139-172, :104-108, src/qr.rs:298): the two-call route must give the fused call's factorization.
template <typename T>
static void lapack_seam_tests(const Context &ctx, const std::string &suf, double tol) {
    run_test("lapack_seam_geqp3_orgqr_equal_the_fused_pivoted_qr_" + suf, [&] {
        const int64_t m = 150, n = 90, k = 90;
        auto a = random_approximate_low_rank_matrix<T>(ctx, m, n, 1.0, 1e-3, seed_of("seam-" + suf));
        auto fused = QR<T>::compute_from(a);
        auto f = geqp3(a);
        auto q = orgqr(f, k);
        auto ji = f.jpvt.to_host(), fi = fused.ind.to_host();
        int agree = 0;
        while (agree < (int)k && ji[(size_t)agree] == fi[(size_t)agree]) ++agree;
        CHECK(agree >= 20);   // the data-determined prefix (the tail of a 1e-3 spectrum ties in f32)
        // R = upper triangle of the first k rows of the factored matrix (src/pivoted_qr.rs:100-102), compared on the agreed prefix
        auto fa = f.a.to_host(), fr = fused.r.to_host(), qh = q.to_host(), fq = fused.q.to_host();
        double num = 0, den = 0, qn = 0, qd = 0;
        for (int64_t i = 0; i < agree; ++i)
            for (int64_t j = i; j < agree; ++j) {
                num += std::norm(fa[(size_t)(i * n + j)] - fr[(size_t)(i * n + j)]);
                den += std::norm(fr[(size_t)(i * n + j)]);
            }
        for (int64_t i = 0; i < m; ++i)
            for (int64_t j = 0; j < agree; ++j) {
                qn += std::norm(qh[(size_t)(i * k + j)] - fq[(size_t)(i * k + j)]);
                qd += std::norm(fq[(size_t)(i * k + j)]);
            }
        CHECK(std::sqrt(num / den) < tol && std::sqrt(qn / qd) < tol);
        // T X = B
        auto r11 = fused.r.leading(k, k);
        auto b = random_gaussian<T>(ctx, k, 37, 5);
        auto x = trsm_upper(r11, b);
        CHECK((double)rel_diff_fro(dot(r11, x), b) < 1e3 * tol);   // forward residual through a triangle of condition ~1e3
    });
}

// The range finders over an OPERATOR instead of a dense array (impl<Op: MatMat> SampleRange for Op, src/random_sampling.rs:102, :130, :222;
// compute_from_range_estimate<Op: ConjMatMat>, src/qr.rs:311-323, src/svd.rs:171-183): the reference has no tests of these, the two below
// pin the callback path of the C ABI (rc_operator) from compiled host code.
static void operator_tests(const Context &ctx) {
    run_test("operator_dense_matrix_behind_callbacks_equals_the_dense_entry_points_f64", [&] {
        const int64_t m = 600, n = 400, k = 30, p = 6;
        auto a = random_approximate_low_rank_matrix<double>(ctx, m, n, 1.0, 1e-8, seed_of("op-dense"));
        DenseOperator<double> op{&a};
        auto q1 = sample_range_by_rank(a, k, p, 7);
        auto q2 = sample_range_by_rank<double>(ctx, op, k, p, 7);
        CHECK(q1.to_host() == q2.to_host());  // bit for bit: the callbacks run the same products on the same views
        auto qp1 = sample_range_power_iteration(a, k, p, 2, 7);
        auto qp2 = sample_range_power_iteration<double>(ctx, op, k, p, 2, 7);
        CHECK(qp1.to_host() == qp2.to_host());
        auto s1 = SVD<double>::compute_from_range_estimate(q1, a);
        auto s2 = svd_from_range_estimate<double>(ctx, q1, op);
        CHECK(s1.s.to_host() == s2.s.to_host() && s1.vt.to_host() == s2.vt.to_host() && s1.u.to_host() == s2.u.to_host());
        auto r1 = QR<double>::compute_from_range_estimate(q1, a);
        auto r2 = qr_from_range_estimate<double>(ctx, q1, op);
        CHECK(r1.ind.to_host() == r2.ind.to_host() && r1.r.to_host() == r2.r.to_host() && r1.q.to_host() == r2.q.to_host());
        auto ad1 = sample_range_adaptive(a, 1e-5, 8, 3);
        auto ad2 = sample_range_adaptive<double>(ctx, op, 1e-5, 8, 3);
        CHECK(ad1.residuals == ad2.residuals && ad1.q.to_host() == ad2.q.to_host());
    });
    run_test("operator_factored_low_rank_never_materialised_f64", [&] {
        const int64_t m = 900, n = 700, r = 40, k = 20, p = 8;
        auto u = random_orthogonal_matrix<double>(ctx, m, r, seed_of("op-u"));
        auto v = random_orthogonal_matrix<double>(ctx, n, r, seed_of("op-v"));
        // U <- U diag(sigma), sigma from 1 down to 1e-6
        auto uh = u.to_host();
        for (int64_t i = 0; i < m; ++i)
            for (int64_t j = 0; j < r; ++j) uh[(size_t)(i * r + j)] *= std::pow(10.0, -6.0 * (double)j / (double)(r - 1));
        auto us = DeviceMatrix<double>::from_host(ctx, uh.data(), m, r);
        LowRankOperator<double> op{&us, &v};
        auto dense = dot_op(0, us, 1, v);  // the same operator, materialised (for comparison only)
        auto q_op = sample_range_by_rank<double>(ctx, op, k, p, 5);
        auto q_d = sample_range_by_rank(dense, k, p, 5);
        CHECK((double)rel_diff_fro(q_op, q_d) <= 1e-9);  // same Omega stream, products equal to rounding
        auto s_op = svd_from_range_estimate<double>(ctx, q_op, op).s.to_host();
        auto s_d = SVD<double>::compute_from_range_estimate(q_d, dense).s.to_host();
        double worst = 0;
        for (size_t i = 0; i < s_op.size(); ++i) worst = std::fmax(worst, std::fabs(s_op[i] - s_d[i]) / s_d[0]);
        CHECK(worst <= 1e-11);
        for (int64_t j = 0; j < 3; ++j) {  // the leading sigma themselves, to the accuracy a rank-20 sketch of this spectrum gives
            const double sj = std::pow(10.0, -6.0 * (double)j / (double)(r - 1));
            CHECK(std::fabs(s_op[(size_t)j] - sj) <= 1e-2 * sj);
        }
        auto ad = sample_range_adaptive<double>(ctx, op, 1e-4, 10, 9);
        CHECK(!ad.residuals.empty() && ad.residuals.back().second < 1e-4 && ad.q.ncols() <= 60);
    });
}

int main() {
    Context ctx(0);
    group<double>(ctx, 100, 50, "thin");
    group<float>(ctx, 100, 50, "thin");
    group<double>(ctx, 50, 100, "thick");
    group<float>(ctx, 50, 100, "thick");
    group<c64>(ctx, 100, 50, "thin");
    group<c32>(ctx, 100, 50, "thin");
    group<c64>(ctx, 50, 100, "thick");
    group<c32>(ctx, 50, 100, "thick");
    permutation_tests(ctx);
    sharded_test(ctx);
    operator_tests(ctx);
    lapack_seam_tests<double>(ctx, "f64", 1e-10);
    lapack_seam_tests<c32>(ctx, "c32", 1e-4);
    std::printf("%d tests, %d failed\n", tests_run, failures);
    return failures ? 1 : 0;
}
