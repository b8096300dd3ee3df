// Batches of independent matrices (BASELINE.json configs[4]: 64 x (4096 x 4096 f32), rank-64 column ID, 8 per GPU)
// and the one exchange step of the path: the gather of the finished factor blocks over RCCL.
//
// Reference call sequence per matrix (examples/interpolative_decomposition.rs:25-32):
//     QR::compute_from(a) -> compress(RANK(k)) -> column_id()
// (src/qr.rs:354-362 via pivoted_qr, :169-184, :270-309).  The reference has no batch or communication layer: it is
// single-process host code; SURVEY.md section 8(b) / 8(e) define these entry points.
//
// rc_batch_column_id_*: the matrices of one GPU are spread over the caller's contexts (one stream each) and pipelined: every
// lane carries an event behind its ?laqps panel (the panel length is data dependent, so the host has to look at its state),
// the host serves the lanes first in, first out -- wait for that lane only, issue its panel-end kernels and its next panel,
// or its C, Z, ind and the set-up of the lane's next matrix -- so the latency-bound pivot steps of some matrices overlap the
// streaming kernels and the host-side launches of the others.  Results land in one packed device buffer, which is exactly
// what rc_comm_gather moves.
//
// RCCL is opened at run time (dlopen) so that the library has no link-time dependency on it: hosts that never gather,
// and the CPU-side symbol tests, do not need it.  Inside a PyTorch process the soname resolves to the copy torch loaded.
#include "rc_common.hpp"
#include <chrono>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

using namespace rc;

namespace {

struct DeviceGuardB {
    int prev = -1;
    explicit DeviceGuardB(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuardB() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline size_t align8(size_t x) { return (x + 7) & ~(size_t)7; }

template <typename T>
Mat<T> tmp_cm(rc_context *c, int64_t rows, int64_t cols) {
    int64_t ld = even_ld(std::max<int64_t>(rows, 1));
    if (ld % 4) ld += 4 - ld % 4;  // 16-byte aligned columns: the blocked QRCP's vector loads
    return colmajor(c->alloc<T>((size_t)ld * std::max<int64_t>(cols, 1)), rows, cols, ld);
}

// column ID from the factorization in the ?geqp3 output format (same steps as column_id_rank in rc_api.hip)
template <typename T>
void finish_column_id(rc_context *c, Mat<T> a, Mat<T> w, int64_t k, const T *tau, int64_t *ind, Mat<T> cm, Mat<T> z, BlockedQrcpJob<T> *job) {
    const int64_t m = w.rows, n = w.cols;
    // RC_COLUMN_ID_FORM_Q=1: C as the product Q R11 with Q formed from the reflectors (the round-2 path, kept for comparison)
    static const bool long_way = [] { const char *e = getenv("RC_COLUMN_ID_FORM_Q"); return e && atoi(e) != 0; }();
    if (!long_way && k < n) { column_id_from_qrcp(c, a, w, k, ind, cm, z); return; }
    Mat<T> r = rowmajor(c->alloc<T>((size_t)k * even_ld(n)), k, n, even_ld(n));
    extract_r(c, w, ind, r);
    Mat<T> q = tmp_cm<T>(c, m, k);
    if (job) qrb_form_q(job, q);
    else form_q(c, w, ind, tau, k, q);
    // QRTraits::column_id (src/qr.rs:270-309): Z = [I | R11^-1 R12] P^T, C = Q R11
    int64_t *inv = c->alloc<int64_t>((size_t)n);
    invert_perm(c, ind, n, inv);
    Mat<T> zt = rowmajor(c->alloc<T>((size_t)k * even_ld(n)), k, n, even_ld(n));
    if (k == n) {
        gemm<T>(c, 1, q, r, 0, cm);
        fill_identity(c, zt);
    } else {
        fill_identity(c, zt.sub(0, k, 0, k));
        copy_mat(c, r.sub(0, k, k, n - k), zt.sub(0, k, k, n - k));
        trsm_upper(c, r.sub(0, k, 0, k), zt.sub(0, k, k, n - k));
        gemm<T>(c, 1, q, r.sub(0, k, 0, k), 0, cm);
    }
    gather_cols(c, zt, inv, z);
}

template <typename T>
void batch_column_id(rc_context *const *ctxs, int nctx, const rc_matrix *mats, int count, int64_t k, void *packed) {
    RC_REQUIRE(nctx >= 1 && count >= 0 && k >= 1, RC_INVALID_ARGUMENT, "batch_column_id: need >= 1 context, k >= 1");
    if (count == 0) return;
    RC_REQUIRE(mats != nullptr && packed != nullptr, RC_INVALID_ARGUMENT, "batch_column_id: null argument");
    const int64_t m = mats[0].rows, n = mats[0].cols;
    for (int i = 0; i < count; ++i)
        RC_REQUIRE(mats[i].rows == m && mats[i].cols == n && mats[i].data, RC_INVALID_ARGUMENT, "batch_column_id: all matrices must have the shape of the first");
    RC_REQUIRE(k <= std::min(m, n), RC_INVALID_ARGUMENT, "batch_column_id: rank %lld exceeds min(m, n)", (long long)k);
    for (int l = 0; l < nctx; ++l) RC_REQUIRE(ctxs[l] && !ctxs[l]->capturing, RC_INVALID_ARGUMENT, "batch_column_id: bad context");
    const size_t per = rc_batch_packed_bytes(m, n, k, (int32_t)sizeof(T));
    struct Lane {
        rc_context *c = nullptr;
        int idx = -1;
        Mat<T> w;
        T *tau = nullptr;
        int64_t *ind = nullptr;
        BlockedQrcpJob<T> *job = nullptr;
        bool active = false;
    };
    std::vector<Lane> lanes((size_t)nctx);
    struct Cleanup {
        std::vector<Lane> &l;
        rc_context *const *ctxs;
        int nctx;
        // every lane is quiescent before the jobs go and before the call returns -- also when it unwinds on an error: with several
        // issuing threads ANY lane may hold the last matrices' work, and the caller frees `packed` and reuses the arenas afterwards
        ~Cleanup() {
            (void)rc_synchronize_all(ctxs, (int32_t)nctx);
            for (auto &x : l) if (x.job) { qrb_end(x.job); x.job = nullptr; }
        }
    } cleanup{lanes, ctxs, nctx};
    auto slot = [&](int idx) { return static_cast<char *>(packed) + (size_t)idx * per; };
    // RC_BATCH_LOCKSTEP_THREADS=1 (lock-step schedule only): one host thread per lane for the issue phases (measured slower than inline issue once the panels
    // replay from hipGraphs: thread start-up + runtime locks cost more than the launches they overlap)
    static const bool threaded = [] { const char *e = getenv("RC_BATCH_LOCKSTEP_THREADS"); return e && atoi(e) != 0; }();
    auto for_lanes = [&](auto &&pred, auto &&body) {
        std::vector<std::thread> th;
        std::vector<Error> errs((size_t)nctx, Error{RC_OK, ""});
        for (int l = 0; l < nctx; ++l) {
            if (!pred(lanes[(size_t)l])) continue;
            auto run = [&, l] {
                try {
                    DeviceGuardB dg(lanes[(size_t)l].c->device);
                    body(lanes[(size_t)l]);
                } catch (const Error &e) { errs[(size_t)l] = e; }
                catch (const std::exception &e) { errs[(size_t)l] = Error{RC_RUNTIME_ERROR, e.what()}; }
            };
            if (threaded) th.emplace_back(run);
            else run();
        }
        for (auto &t : th) t.join();
        for (auto &e : errs)
            if (e.code != RC_OK) throw e;
    };
    auto post = [&](Lane &ln) {
        char *base = slot(ln.idx);
        Mat<T> cm = rowmajor(reinterpret_cast<T *>(base), m, k, k);
        Mat<T> z = rowmajor(reinterpret_cast<T *>(base) + (size_t)m * k, k, n, n);
        finish_column_id<T>(ln.c, from_c<T>(mats[ln.idx]), ln.w, k, ln.tau, ln.ind, cm, z, ln.job);
        RC_HIP(hipMemcpyAsync(base + align8(((size_t)m * k + (size_t)k * n) * sizeof(T)), ln.ind, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, ln.c->stream));
    };
    // set-up of the next matrix on a lane: working copy, then either the first panel of the blocked factorization (true: the
    // lane has a panel in flight) or, for small shapes, the whole per-step chain (false: nothing to wait for)
    auto start = [&](Lane &ln, int idx) -> bool {
        ln.idx = idx;
        DeviceGuardB dg(ln.c->device);
        ln.c->reset_arena();
        ln.w = tmp_cm<T>(ln.c, m, n);
        ln.tau = ln.c->template alloc<T>((size_t)k);
        // (the permutation is built in a lane-owned buffer whose address repeats from matrix to matrix, so that the
        // cached panel graphs of the blocked QRCP replay; it is copied into the packed slot at the end)
        ln.ind = ln.c->template alloc<int64_t>((size_t)n);
        copy_mat(ln.c, from_c<T>(mats[ln.idx]), ln.w);  // the reference's F-order working copy (pivoted_qr.rs:28-29)
        if (ln.c->opt_blocked && geqp3_blocked_supported<T>(m, n, k)) {
            ln.job = qrb_begin<T>(ln.c, ln.w, k, ln.ind, ln.tau);
            static const bool long_way = [] { const char *e = getenv("RC_COLUMN_ID_FORM_Q"); return e && atoi(e) != 0; }();
            qrb_keep_t(ln.job, long_way || k >= n);  // (the ID comes straight from the factored matrix: no Q, no T factors to keep)
            ln.active = true;
            return true;
        }
        T *vn = ln.c->template alloc<T>((size_t)(2 * n));
        geqp3_inplace(ln.c, ln.w, k, true, ln.ind, ln.tau, vn);
        post(ln);
        ln.active = false;
        return false;
    };
    for (int l = 0; l < nctx; ++l) lanes[(size_t)l].c = ctxs[l];
    // Optimistic schedule (default where it applies: blocked factorization on cooperative panels, no Q wanted): EVERY matrix is
    // enqueued in full on its lane -- all panels on their usual outcome (qrb_issue_all_optimistic), then C, Z, ind -- with no host
    // wait in between; one wait for all lanes at the end, then the per-panel states are checked and the (rare) matrices whose
    // assumptions failed are redone through the per-panel pipeline below.  The lanes' chains then overlap on the GPU without the
    // bubbles of a host round trip per panel.  Measured (8 x 4096^2 f32, k = 64): the host issues the 8 matrices in 0.6 ms and waits 2.05 ms --
    // the batch is bound on the GPU, so this schedule by itself is neutral against the pipeline below (2560 matrices/s either way); with 256
    // instead of 512 candidates per panel both reach 2810-2860.
    std::vector<int> redo;
    bool optimistic_done = false;
    {
        static const bool long_way = [] { const char *e = getenv("RC_COLUMN_ID_FORM_Q"); return e && atoi(e) != 0; }();
        static const bool opt_on = [] { const char *e = getenv("RC_BATCH_OPTIMISTIC"); return !(e && atoi(e) == 0); }();
        if (opt_on && !long_way && k < n && count > 0 && lanes[0].c->opt_blocked && geqp3_blocked_supported<T>(m, n, k)) {
            struct Pending { int idx; const QrbState *log; std::vector<int> nbp; bool broken; };
            std::vector<Pending> pending;
            auto flush = [&] {
                std::vector<rc_context *> all;
                for (auto &ln : lanes) all.push_back(ln.c);
                RC_REQUIRE(rc_synchronize_all(all.data(), (int32_t)all.size()) == RC_OK, RC_RUNTIME_ERROR, "batch_column_id: wait failed");
                for (auto &pd : pending)
                    if (!qrb_verify_optimistic<T>(pd.log, pd.nbp, pd.broken)) redo.push_back(pd.idx);
                pending.clear();
                for (auto &ln : lanes) ln.c->pinned_cursor = 0;
            };
            bool possible = true;
            static const bool dbg = [] { const char *e = getenv("RC_BATCH_DEBUG"); return e && atoi(e) != 0; }();
            const auto t_begin = std::chrono::steady_clock::now();
            for (auto &ln : lanes) ln.c->pinned_cursor = 0;
            for (int idx = 0; idx < count && possible; ++idx) {
                Lane &ln = lanes[(size_t)(idx % nctx)];
                DeviceGuardB dg(ln.c->device);
                ln.idx = idx;
                ln.c->reset_arena();
                ln.w = tmp_cm<T>(ln.c, m, n);
                ln.tau = ln.c->template alloc<T>((size_t)k);
                ln.ind = ln.c->template alloc<int64_t>((size_t)n);
                copy_mat(ln.c, from_c<T>(mats[idx]), ln.w);
                ln.job = qrb_begin<T>(ln.c, ln.w, k, ln.ind, ln.tau);
                qrb_keep_t(ln.job, false);
                if (!qrb_optimistic_possible(ln.job)) {
                    possible = false;
                } else {
                    if (!qrb_issue_all_optimistic(ln.job)) {  // no pinned slots left on this lane: wait, check, go on
                        flush();
                        RC_REQUIRE(qrb_issue_all_optimistic(ln.job), RC_RUNTIME_ERROR, "batch_column_id: no pinned state slots");
                    }
                    Pending pd;
                    pd.idx = idx;
                    qrb_optimistic_log(ln.job, &pd.log, &pd.nbp, &pd.broken);
                    pending.push_back(std::move(pd));
                    post(ln);
                }
                qrb_end(ln.job);
                ln.job = nullptr;
                ln.active = false;
                if (!possible) {  // (first matrix only in practice: all matrices share one shape) -- everything through the pipeline below
                    flush();
                    redo.clear();
                    for (int i2 = 0; i2 < count; ++i2) redo.push_back(i2);
                }
            }
            const auto t_issued = std::chrono::steady_clock::now();
            if (possible) flush();
            if (dbg) {
                const auto t_end = std::chrono::steady_clock::now();
                fprintf(stderr, "rc_batch optimistic: %d matrices issued in %.3f ms, waited %.3f ms, %zu to redo\n", count,
                        std::chrono::duration<double, std::milli>(t_issued - t_begin).count(), std::chrono::duration<double, std::milli>(t_end - t_issued).count(), redo.size());
            }
            optimistic_done = true;
        }
    }
    // what is left to do through the per-panel schedules: everything, or the matrices the optimistic pass has to redo
    std::vector<int> todo;
    if (optimistic_done) todo = redo;
    else for (int i2 = 0; i2 < count; ++i2) todo.push_back(i2);
    const int count_all = count;
    (void)count_all;
    count = (int)todo.size();
    auto mat_of = [&](int t) { return todo[(size_t)t]; };
    // RC_BATCH_LOCKSTEP=1: the round-1 schedule (all lanes issue, ONE wait for all, all lanes finish), kept for comparison.
    static const bool lockstep = [] { const char *e = getenv("RC_BATCH_LOCKSTEP"); return e && atoi(e) != 0; }();
    if (lockstep) {
        for (int base = 0; base < count; base += nctx) {
            int nact = 0;
            for (int l = 0; l < nctx && base + l < count; ++l) nact += start(lanes[(size_t)l], mat_of(base + l)) ? 1 : 0;
            while (nact > 0) {
                for_lanes([](Lane &ln) { return ln.active; }, [](Lane &ln) { qrb_issue(ln.job); });
                // one wait for all lanes: an event on every stream first (see rc_synchronize_all)
                std::vector<rc_context *> act;
                for (auto &ln : lanes)
                    if (ln.active) act.push_back(ln.c);
                RC_REQUIRE(rc_synchronize_all(act.data(), (int32_t)act.size()) == RC_OK, RC_RUNTIME_ERROR, "batch_column_id: wait failed");
                for_lanes([](Lane &ln) { return ln.active; }, [&](Lane &ln) {
                    if (qrb_finish(ln.job)) {
                        ln.active = false;
                        post(ln);
                        qrb_end(ln.job);
                        ln.job = nullptr;
                    }
                });
                nact = 0;
                for (auto &ln : lanes) nact += ln.active ? 1 : 0;
            }
        }
    } else {
        // Pipeline: every lane carries an event behind its panel; the host serves the lanes in the order their panels were
        // issued (first in, first out): wait for that lane only, enqueue its panel-end kernels and its next panel (or its C, Z,
        // ind and the set-up of the lane's next matrix), go on to the next lane.  The lanes drift apart by the host time of one
        // lane's launches, so the host-bound phases of some matrices (dozens of small launches) run while the latency-bound
        // cooperative panels of the others occupy the GPU, instead of everything waiting for the slowest panel of a round.
        auto mark = [&](Lane &ln) {
            rc_context *c = ln.c;
            if (!c->sync_ev) RC_HIP(hipEventCreateWithFlags(&c->sync_ev, hipEventDisableTiming));
            RC_HIP(hipEventRecord(c->sync_ev, c->stream));
        };
        // RC_BATCH_THREADS issuing threads own disjoint sets of lanes and draw matrices from one counter.  Default 1: the batch
        // is bound by the host's launch rate (~46 kernels + copies per matrix), but the runtime serialises launches of different
        // threads -- 2 / 4 / 8 threads measured 2091 / 2056 / 2075 matrices/s against 2173 with one (8 x 4096 x 4096 f32, k = 64).
        static const int want_threads = [] { const char *e = getenv("RC_BATCH_THREADS"); return e ? std::max(1, atoi(e)) : 1; }();
        const int nthr = std::max(1, std::min({want_threads, nctx, count}));
        std::atomic<int> next{0};
        auto worker = [&](int l0, int l1) {
            std::deque<int> fifo;
            for (;;) {
                for (int l = l0; l < l1; ++l) {  // idle lanes take the next matrices
                    Lane &ln = lanes[(size_t)l];
                    if (ln.active) continue;
                    const int idx = next.fetch_add(1);
                    if (idx >= count) break;
                    if (start(ln, mat_of(idx))) {
                        DeviceGuardB dg(ln.c->device);
                        qrb_issue(ln.job);
                        mark(ln);
                        fifo.push_back(l);
                    }
                }
                if (fifo.empty()) {
                    if (next.load() >= count) break;
                    continue;
                }
                const int l = fifo.front();
                Lane &ln = lanes[(size_t)l];
                fifo.pop_front();
                DeviceGuardB dg(ln.c->device);
                RC_HIP(hipEventSynchronize(ln.c->sync_ev));
                if (qrb_finish(ln.job)) {
                    ln.active = false;  // (the lane is handed its next matrix at the top of the loop)
                    post(ln);
                    qrb_end(ln.job);
                    ln.job = nullptr;
                } else {
                    qrb_issue(ln.job);
                    mark(ln);
                    fifo.push_back(l);
                }
            }
        };
        if (nthr == 1) {
            worker(0, nctx);
        } else {
            std::vector<std::thread> th;
            std::vector<Error> errs((size_t)nthr, Error{RC_OK, ""});
            for (int t = 0; t < nthr; ++t) {
                const int l0 = (int)((int64_t)nctx * t / nthr), l1 = (int)((int64_t)nctx * (t + 1) / nthr);
                th.emplace_back([&, t, l0, l1] {
                    try { worker(l0, l1); }
                    catch (const Error &e) { errs[(size_t)t] = e; next.store(count); }
                    catch (const std::exception &e) { errs[(size_t)t] = Error{RC_RUNTIME_ERROR, e.what()}; next.store(count); }
                });
            }
            for (auto &t : th) t.join();
            for (auto &e : errs)
                if (e.code != RC_OK) throw e;
        }
    }
    std::vector<rc_context *> all(ctxs, ctxs + nctx);  // all of them: which lanes were handed matrices depends on the issuing threads
    RC_REQUIRE(rc_synchronize_all(all.data(), (int32_t)all.size()) == RC_OK, RC_RUNTIME_ERROR, "batch_column_id: wait failed");
    for (rc_context *c : all) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) fail(RC_RUNTIME_ERROR, "kernel launch failed: %s", hipGetErrorString(e));
        (void)c;
    }
}

template <typename F>
rc_status guarded_b(rc_context *ctx, F &&f) {
    if (!ctx) return RC_INVALID_ARGUMENT;
    try {
        f();
        return RC_OK;
    } catch (const Error &e) {
        ctx->last_error = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        ctx->last_error = e.what();
        return RC_RUNTIME_ERROR;
    }
}

// ---------------------------------------------------------------- RCCL, opened at run time
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("librccl could not be opened: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

}  // namespace

struct rc_comm {
    ncclComm_t comm = nullptr;  // RCCL transport (rc_comm_init), or
    rc_host_all_gather_fn host_gather = nullptr;  // the host's own collectives (rc_comm_init_host): buffers staged through the host
    rc_host_all_reduce_sum_fn host_reduce = nullptr;
    void *user = nullptr;
    std::vector<char> stage_in, stage_out;
    int world = 1, rank = 0, device = 0;
    std::string last_error;
};

extern "C" {

size_t rc_batch_packed_bytes(int64_t m, int64_t n, int64_t k, int32_t elem_size) {
    if (m < 0 || n < 0 || k < 0 || elem_size <= 0) return 0;
    return align8(((size_t)m * (size_t)k + (size_t)k * (size_t)n) * (size_t)elem_size) + (size_t)n * 8;
}

rc_status rc_batch_shard_range(int64_t n_items, int32_t world, int32_t rank, int64_t *start, int64_t *count) {
    if (n_items < 0 || world < 1 || rank < 0 || rank >= world || !start || !count) return RC_INVALID_ARGUMENT;
    const int64_t base = n_items / world, extra = n_items % world;
    *start = rank * base + std::min<int64_t>(rank, extra);
    *count = base + (rank < extra ? 1 : 0);
    return RC_OK;
}

rc_status rc_batch_column_id_f64(rc_context *const *ctxs, int32_t nctx, const rc_matrix *mats, int32_t count, int64_t k, void *packed) {
    if (!ctxs || nctx < 1 || !ctxs[0]) return RC_INVALID_ARGUMENT;
    return guarded_b(ctxs[0], [&] { batch_column_id<double>(ctxs, nctx, mats, count, k, packed); });
}
rc_status rc_batch_column_id_f32(rc_context *const *ctxs, int32_t nctx, const rc_matrix *mats, int32_t count, int64_t k, void *packed) {
    if (!ctxs || nctx < 1 || !ctxs[0]) return RC_INVALID_ARGUMENT;
    return guarded_b(ctxs[0], [&] { batch_column_id<float>(ctxs, nctx, mats, count, k, packed); });
}

rc_status rc_comm_unique_id(void *id128) {
    if (!id128) return RC_INVALID_ARGUMENT;
    Rccl *r = rccl();
    if (!r->error.empty()) return RC_RUNTIME_ERROR;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return RC_RUNTIME_ERROR;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return RC_OK;
}

rc_status rc_comm_init(rc_comm **comm, int32_t world, int32_t rank, const void *id128, int32_t device) {
    if (!comm) return RC_INVALID_ARGUMENT;
    *comm = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !id128) return RC_INVALID_ARGUMENT;
    Rccl *r = rccl();
    if (!r->error.empty()) return RC_RUNTIME_ERROR;
    DeviceGuardB dg(device);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    rc_comm *c = new rc_comm();
    c->world = world; c->rank = rank; c->device = device;
    if (r->CommInitRank(&c->comm, world, id, rank) != ncclSuccess) { delete c; return RC_RUNTIME_ERROR; }
    *comm = c;
    return RC_OK;
}

const char *rc_comm_last_error_message(const rc_comm *comm) {
    if (comm) return comm->last_error.c_str();
    return rccl()->error.c_str();
}

// Gather `bytes_per_rank` bytes from every rank into recv[rank * bytes_per_rank ...] on `root`, on the context's stream
// (asynchronous; grouped ncclSend / ncclRecv: every peer uses its own direct xGMI link to the root).
rc_status rc_comm_gather(rc_comm *comm, rc_context *ctx, const void *send, void *recv, size_t bytes_per_rank, int32_t root) {
    if (!comm || !ctx || root < 0 || root >= comm->world) return RC_INVALID_ARGUMENT;
    if (bytes_per_rank == 0) return RC_OK;
    if (!send || (comm->rank == root && !recv)) return RC_INVALID_ARGUMENT;
    if (!comm->comm) {  // host transport: the gather is the all-gather's block on the root
        comm->last_error = "rc_comm_gather needs the RCCL transport (use rc_comm_all_gather with a host communicator)";
        ctx->last_error = comm->last_error;
        return RC_INVALID_ARGUMENT;
    }
    Rccl *r = rccl();
    DeviceGuardB dg(comm->device);
    auto chk = [&](ncclResult_t e) {
        if (e == ncclSuccess) return true;
        comm->last_error = r->GetErrorString ? r->GetErrorString(e) : "RCCL error";
        ctx->last_error = comm->last_error;
        return false;
    };
    bool ok = chk(r->GroupStart());
    if (ok && comm->rank == root)
        for (int p = 0; ok && p < comm->world; ++p)
            ok = chk(r->Recv(static_cast<char *>(recv) + (size_t)p * bytes_per_rank, bytes_per_rank, ncclUint8, p, comm->comm, ctx->stream));
    if (ok) ok = chk(r->Send(send, bytes_per_rank, ncclUint8, root, comm->comm, ctx->stream));
    const bool ended = chk(r->GroupEnd());
    return ok && ended ? RC_OK : RC_RUNTIME_ERROR;
}

rc_status rc_comm_init_host(rc_comm **comm, int32_t world, int32_t rank, int32_t device, rc_host_all_gather_fn all_gather,
                            rc_host_all_reduce_sum_fn all_reduce_sum, void *user) {
    if (!comm) return RC_INVALID_ARGUMENT;
    *comm = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !all_gather || !all_reduce_sum) return RC_INVALID_ARGUMENT;
    rc_comm *c = new rc_comm();
    c->world = world; c->rank = rank; c->device = device;
    c->host_gather = all_gather; c->host_reduce = all_reduce_sum; c->user = user;
    *comm = c;
    return RC_OK;
}

rc_status rc_comm_world(const rc_comm *comm, int32_t *world, int32_t *rank) {
    if (!comm) return RC_INVALID_ARGUMENT;
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    return RC_OK;
}

namespace {
bool comm_fail(rc_comm *comm, rc_context *ctx, const std::string &msg) {
    comm->last_error = msg;
    ctx->last_error = msg;
    return false;
}
bool comm_hip(rc_comm *comm, rc_context *ctx, hipError_t e, const char *what) {
    return e == hipSuccess ? true : comm_fail(comm, ctx, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace

// recv[r * bytes_per_rank ...] = rank r's `send`, on every rank, ordered on the context's stream.  `send` may be the
// rank's own block inside `recv` (in place).  RCCL transport: asynchronous (ncclAllGather); host transport: the stream is
// waited for, the host's callback runs on host copies, the result is copied back before the call returns.
rc_status rc_comm_all_gather(rc_comm *comm, rc_context *ctx, const void *send, void *recv, size_t bytes_per_rank) {
    if (!comm || !ctx) return RC_INVALID_ARGUMENT;
    if (bytes_per_rank == 0) return RC_OK;
    if (!send || !recv) return RC_INVALID_ARGUMENT;
    DeviceGuardB dg(comm->device);
    char *own = static_cast<char *>(recv) + (size_t)comm->rank * bytes_per_rank;
    if (comm->world == 1 && !comm->comm) {  // (an RCCL communicator of one rank still goes through ncclAllGather)
        if (own != send && !comm_hip(comm, ctx, hipMemcpyAsync(own, send, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream), "all_gather copy")) return RC_RUNTIME_ERROR;
        return RC_OK;
    }
    if (comm->host_gather) {
        comm->stage_in.resize(bytes_per_rank);
        comm->stage_out.resize(bytes_per_rank * (size_t)comm->world);
        if (!comm_hip(comm, ctx, hipMemcpyAsync(comm->stage_in.data(), send, bytes_per_rank, hipMemcpyDeviceToHost, ctx->stream), "all_gather D2H")) return RC_RUNTIME_ERROR;
        if (!comm_hip(comm, ctx, hipStreamSynchronize(ctx->stream), "all_gather wait")) return RC_RUNTIME_ERROR;
        if (comm->host_gather(comm->user, comm->stage_in.data(), comm->stage_out.data(), bytes_per_rank) != 0) {
            comm_fail(comm, ctx, "the host's all_gather callback reported an error");
            return RC_RUNTIME_ERROR;
        }
        if (!comm_hip(comm, ctx, hipMemcpyAsync(recv, comm->stage_out.data(), comm->stage_out.size(), hipMemcpyHostToDevice, ctx->stream), "all_gather H2D")) return RC_RUNTIME_ERROR;
        if (!comm_hip(comm, ctx, hipStreamSynchronize(ctx->stream), "all_gather wait")) return RC_RUNTIME_ERROR;
        return RC_OK;
    }
    Rccl *r = rccl();
    if (!r->error.empty() || !r->AllGather || !comm->comm) { comm_fail(comm, ctx, r->error.empty() ? "communicator has no transport" : r->error); return RC_RUNTIME_ERROR; }
    ncclResult_t e = r->AllGather(send, recv, bytes_per_rank, ncclUint8, comm->comm, ctx->stream);
    if (e != ncclSuccess) { comm_fail(comm, ctx, r->GetErrorString ? r->GetErrorString(e) : "RCCL error"); return RC_RUNTIME_ERROR; }
    return RC_OK;
}

// buf[i] = sum over the ranks of buf[i], in place, the same bits on every rank; elem_size 8: double, 4: float
rc_status rc_comm_all_reduce_sum(rc_comm *comm, rc_context *ctx, void *buf, size_t count, int32_t elem_size) {
    if (!comm || !ctx || (elem_size != 4 && elem_size != 8)) return RC_INVALID_ARGUMENT;
    if (count == 0 || (comm->world == 1 && !comm->comm)) return RC_OK;
    if (!buf) return RC_INVALID_ARGUMENT;
    DeviceGuardB dg(comm->device);
    const size_t bytes = count * (size_t)elem_size;
    if (comm->host_reduce) {
        comm->stage_in.resize(bytes);
        if (!comm_hip(comm, ctx, hipMemcpyAsync(comm->stage_in.data(), buf, bytes, hipMemcpyDeviceToHost, ctx->stream), "all_reduce D2H")) return RC_RUNTIME_ERROR;
        if (!comm_hip(comm, ctx, hipStreamSynchronize(ctx->stream), "all_reduce wait")) return RC_RUNTIME_ERROR;
        if (comm->host_reduce(comm->user, comm->stage_in.data(), count, elem_size) != 0) {
            comm_fail(comm, ctx, "the host's all_reduce callback reported an error");
            return RC_RUNTIME_ERROR;
        }
        if (!comm_hip(comm, ctx, hipMemcpyAsync(buf, comm->stage_in.data(), bytes, hipMemcpyHostToDevice, ctx->stream), "all_reduce H2D")) return RC_RUNTIME_ERROR;
        if (!comm_hip(comm, ctx, hipStreamSynchronize(ctx->stream), "all_reduce wait")) return RC_RUNTIME_ERROR;
        return RC_OK;
    }
    Rccl *r = rccl();
    if (!r->error.empty() || !r->AllReduce || !comm->comm) { comm_fail(comm, ctx, r->error.empty() ? "communicator has no transport" : r->error); return RC_RUNTIME_ERROR; }
    ncclResult_t e = r->AllReduce(buf, buf, count, elem_size == 8 ? ncclDouble : ncclFloat, ncclSum, comm->comm, ctx->stream);
    if (e != ncclSuccess) { comm_fail(comm, ctx, r->GetErrorString ? r->GetErrorString(e) : "RCCL error"); return RC_RUNTIME_ERROR; }
    return RC_OK;
}

rc_status rc_comm_destroy(rc_comm *comm) {
    if (!comm) return RC_OK;
    Rccl *r = rccl();
    DeviceGuardB dg(comm->device);
    if (comm->comm && r->CommDestroy) (void)r->CommDestroy(comm->comm);
    delete comm;
    return RC_OK;
}

}  // extern "C"
