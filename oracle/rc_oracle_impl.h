/* CPU ORACLE (test infrastructure only) -- LAPACK-free restatement of the
 * numerical kernels the reference's hot path bottoms out in.  Included twice
 * by rc_oracle.c with T = double / float and SUF = d / s.
 *
 * Algorithms restated from the public LAPACK specification (the reference
 * calls them through the `lapack`, `lax` and `ndarray-linalg` crates, which are
 * NOT under /root/reference; versions unpinned, no Cargo.lock):
 *   ?geqp3 / ?laqp2 / ?larfg : /root/reference/src/pivoted_qr.rs:139-150,:161-172
 *   ?orgqr (?org2r)          : /root/reference/src/pivoted_qr.rs:104-108
 *   ?gesdd (thin SVD)        : /root/reference/src/compute_svd.rs:19
 *                              (restated as Householder QR/LQ + one-sided Jacobi;
 *                              S, U S Vt agree with gesdd, column signs differ)
 *   ?trtrs                   : /root/reference/src/qr.rs:290-301, :384-395
 * All matrices here are COLUMN-MAJOR with leading dimension lda, like LAPACK.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(CAT(rco_, name), CAT(_, SUF))

/* ---- ?larfg: elementary reflector H = I - tau v v^T, v[0] = 1 -------------- */
static void FN(larfg)(int64_t n, T *alpha, T *x, T *tau) {
    if (n <= 1) { *tau = 0; return; }
    /* xnorm = nrm2(x[0..n-1)) with scaling (dnrm2-style robust accumulate) */
    T scale = 0, ssq = 1;
    for (int64_t i = 0; i < n - 1; ++i) {
        if (x[i] != 0) {
            T a = FABS(x[i]);
            if (scale < a) { ssq = 1 + ssq * (scale / a) * (scale / a); scale = a; }
            else ssq += (a / scale) * (a / scale);
        }
    }
    T xnorm = scale * SQRT(ssq);
    if (xnorm == 0) { *tau = 0; return; }
    T a = *alpha;
    T beta = -COPYSIGN(HYPOT(a, xnorm), a);
    *tau = (beta - a) / beta;
    T sc = 1 / (a - beta);
    for (int64_t i = 0; i < n - 1; ++i) x[i] *= sc;
    *alpha = beta;
}

static T FN(nrm2)(int64_t n, const T *x) {
    T scale = 0, ssq = 1;
    for (int64_t i = 0; i < n; ++i) {
        if (x[i] != 0) {
            T a = FABS(x[i]);
            if (scale < a) { ssq = 1 + ssq * (scale / a) * (scale / a); scale = a; }
            else ssq += (a / scale) * (a / scale);
        }
    }
    return scale * SQRT(ssq);
}

/* ---- ?geqp3 with ?laqp2 semantics (unblocked; physical column swaps) -------
 * a: m x n column-major, overwritten by R (upper) and reflectors (below).
 * jpvt: n, 0-based on exit (the reference subtracts 1, pivoted_qr.rs:177).
 * kmax: number of Householder steps (min(m,n) = full factorization).
 * pivot = 0 turns pivoting off (plain Householder QR, ?geqr2). */
int FN(geqp3)(int64_t m, int64_t n, T *a, int64_t lda, int64_t *jpvt, T *tau,
              int64_t kmax, int pivot) {
    int64_t mn = m < n ? m : n;
    if (kmax > mn) kmax = mn;
    T *vn1 = (T *)malloc(sizeof(T) * (size_t)(n > 0 ? n : 1));
    T *vn2 = (T *)malloc(sizeof(T) * (size_t)(n > 0 ? n : 1));
    T *w = (T *)malloc(sizeof(T) * (size_t)(n > 0 ? n : 1));
    if (!vn1 || !vn2 || !w) return -1;
    const T tol3z = SQRT(EPS_HALF);
    for (int64_t j = 0; j < n; ++j) {
        jpvt[j] = j;
        vn1[j] = FN(nrm2)(m, a + j * lda);
        vn2[j] = vn1[j];
    }
    for (int64_t i = 0; i < kmax; ++i) {
        /* pivot: first maximum of vn1[i..n) (idamax) */
        if (pivot) {
            int64_t pvt = i;
            T best = FABS(vn1[i]);
            for (int64_t j = i + 1; j < n; ++j)
                if (FABS(vn1[j]) > best) { best = FABS(vn1[j]); pvt = j; }
            if (pvt != i) {
                for (int64_t r = 0; r < m; ++r) {
                    T t = a[r + pvt * lda]; a[r + pvt * lda] = a[r + i * lda]; a[r + i * lda] = t;
                }
                int64_t it = jpvt[pvt]; jpvt[pvt] = jpvt[i]; jpvt[i] = it;
                vn1[pvt] = vn1[i];
                vn2[pvt] = vn2[i];
            }
        }
        /* reflector */
        T *col = a + i * lda;
        if (i < m - 1) FN(larfg)(m - i, &col[i], &col[i + 1], &tau[i]);
        else FN(larfg)(1, &col[m - 1], &col[m - 1], &tau[i]);
        /* apply H_i to A[i:m, i+1:n] from the left */
        if (i + 1 < n) {
            T aii = col[i];
            col[i] = 1;
            for (int64_t j = i + 1; j < n; ++j) {
                T s = 0;
                const T *cj = a + j * lda;
                for (int64_t r = i; r < m; ++r) s += col[r] * cj[r];
                w[j] = s;
            }
            for (int64_t j = i + 1; j < n; ++j) {
                T *cj = a + j * lda;
                T f = tau[i] * w[j];
                for (int64_t r = i; r < m; ++r) cj[r] -= f * col[r];
            }
            col[i] = aii;
        }
        /* partial norm downdate (dlaqp2) */
        for (int64_t j = i + 1; j < n; ++j) {
            if (vn1[j] != 0) {
                T t = FABS(a[i + j * lda]) / vn1[j];
                T temp = 1 - t * t;
                if (temp < 0) temp = 0;
                T r = vn1[j] / vn2[j];
                T temp2 = temp * r * r;
                if (temp2 <= tol3z) {
                    if (i < m - 1) {
                        vn1[j] = FN(nrm2)(m - i - 1, a + (i + 1) + j * lda);
                        vn2[j] = vn1[j];
                    } else { vn1[j] = 0; vn2[j] = 0; }
                } else vn1[j] *= SQRT(temp);
            }
        }
    }
    free(vn1); free(vn2); free(w);
    return 0;
}

/* ---- ?org2r: Q (m x nq) from k reflectors stored in a (columns 0..k) -------- */
int FN(orgqr)(int64_t m, int64_t nq, int64_t k, T *a, int64_t lda, const T *tau) {
    if (nq > m || k > nq) return -1;
    T *w = (T *)malloc(sizeof(T) * (size_t)(nq > 0 ? nq : 1));
    if (!w) return -1;
    for (int64_t j = k; j < nq; ++j) {
        for (int64_t l = 0; l < m; ++l) a[l + j * lda] = 0;
        a[j + j * lda] = 1;
    }
    for (int64_t i = k - 1; i >= 0; --i) {
        T *v = a + i * lda;
        if (i < nq - 1) {
            v[i] = 1;
            for (int64_t j = i + 1; j < nq; ++j) {
                T s = 0;
                const T *cj = a + j * lda;
                for (int64_t r = i; r < m; ++r) s += v[r] * cj[r];
                w[j] = s;
            }
            for (int64_t j = i + 1; j < nq; ++j) {
                T *cj = a + j * lda;
                T f = tau[i] * w[j];
                for (int64_t r = i; r < m; ++r) cj[r] -= f * v[r];
            }
        }
        if (i < m - 1)
            for (int64_t r = i + 1; r < m; ++r) v[r] = -tau[i] * v[r];
        v[i] = 1 - tau[i];
        for (int64_t l = 0; l < i; ++l) v[l] = 0;
    }
    free(w);
    return 0;
}

/* ---- ?trtrs, upper, non-unit, no transpose, nrhs columns -------------------- */
int FN(trtrs_upper)(int64_t k, int64_t nrhs, const T *r, int64_t ldr, T *b, int64_t ldb) {
    for (int64_t i = 0; i < k; ++i)
        if (r[i + i * ldr] == 0) return (int)(i + 1);
    for (int64_t c = 0; c < nrhs; ++c) {
        T *x = b + c * ldb;
        for (int64_t i = k - 1; i >= 0; --i) {
            T s = x[i];
            for (int64_t j = i + 1; j < k; ++j) s -= r[i + j * ldr] * x[j];
            x[i] = s / r[i + i * ldr];
        }
    }
    return 0;
}

/* ---- one-sided (Hestenes) Jacobi SVD of a square n x n matrix g -------------
 * On exit g = U diag(s) (columns), v = V, s descending. */
int FN(jacobi_svd)(int64_t n, T *g, int64_t ldg, T *v, int64_t ldv, T *s) {
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < n; ++i) v[i + j * ldv] = (i == j) ? 1 : 0;
    const T tol = SQRT((T)n) * EPS_HALF * 2;
    int sweep, rotated = 1;
    for (sweep = 0; sweep < 60 && rotated; ++sweep) {
        rotated = 0;
        for (int64_t p = 0; p < n - 1; ++p)
            for (int64_t q = p + 1; q < n; ++q) {
                T app = 0, aqq = 0, apq = 0;
                T *gp = g + p * ldg, *gq = g + q * ldg;
                for (int64_t i = 0; i < n; ++i) { app += gp[i] * gp[i]; aqq += gq[i] * gq[i]; apq += gp[i] * gq[i]; }
                if (apq == 0 || FABS(apq) <= tol * SQRT(app) * SQRT(aqq)) continue;
                rotated = 1;
                T zeta = (aqq - app) / (2 * apq);
                T t = COPYSIGN((T)1, zeta) / (FABS(zeta) + SQRT(1 + zeta * zeta));
                T c = 1 / SQRT(1 + t * t), sn = c * t;
                for (int64_t i = 0; i < n; ++i) {
                    T a = gp[i], b = gq[i];
                    gp[i] = c * a - sn * b; gq[i] = sn * a + c * b;
                }
                T *vp = v + p * ldv, *vq = v + q * ldv;
                for (int64_t i = 0; i < n; ++i) {
                    T a = vp[i], b = vq[i];
                    vp[i] = c * a - sn * b; vq[i] = sn * a + c * b;
                }
            }
    }
    for (int64_t j = 0; j < n; ++j) s[j] = FN(nrm2)(n, g + j * ldg);
    /* selection sort, descending; permute columns of g and v */
    for (int64_t j = 0; j < n; ++j) {
        int64_t best = j;
        for (int64_t l = j + 1; l < n; ++l) if (s[l] > s[best]) best = l;
        if (best != j) {
            T t = s[j]; s[j] = s[best]; s[best] = t;
            for (int64_t i = 0; i < n; ++i) {
                t = g[i + j * ldg]; g[i + j * ldg] = g[i + best * ldg]; g[i + best * ldg] = t;
                t = v[i + j * ldv]; v[i + j * ldv] = v[i + best * ldv]; v[i + best * ldv] = t;
            }
        }
    }
    for (int64_t j = 0; j < n; ++j)
        if (s[j] > 0) for (int64_t i = 0; i < n; ++i) g[i + j * ldg] /= s[j];
    return sweep;
}

/* ---- thin SVD of m x n (column-major): u (m x r), s (r), vt (r x n), r=min --
 * Householder QR (m >= n) or LQ (m < n) to a square core, Jacobi on the core. */
int FN(svd_thin)(int64_t m, int64_t n, const T *a, int64_t lda, T *u, int64_t ldu, T *s, T *vt, int64_t ldvt) {
    int64_t r = m < n ? m : n;
    int tall = m >= n;
    int64_t M = tall ? m : n; /* QR is done on the tall orientation (a or a^T) */
    T *w = (T *)malloc(sizeof(T) * (size_t)(M * r));
    T *tau = (T *)malloc(sizeof(T) * (size_t)r);
    int64_t *jp = (int64_t *)malloc(sizeof(int64_t) * (size_t)r);
    T *core = (T *)malloc(sizeof(T) * (size_t)(r * r));
    T *vv = (T *)malloc(sizeof(T) * (size_t)(r * r));
    if (!w || !tau || !jp || !core || !vv) return -1;
    for (int64_t j = 0; j < r; ++j)
        for (int64_t i = 0; i < M; ++i) w[i + j * M] = tall ? a[i + j * lda] : a[j + i * lda];
    FN(geqp3)(M, r, w, M, jp, tau, r, 0);
    /* core = R (tall) or L = R^T (wide) */
    for (int64_t j = 0; j < r; ++j)
        for (int64_t i = 0; i < r; ++i) {
            T rij = (i <= j) ? w[i + j * M] : 0;
            if (tall) core[i + j * r] = rij; else core[j + i * r] = rij;
        }
    FN(orgqr)(M, r, r, w, M, tau);
    int sweeps = FN(jacobi_svd)(r, core, r, vv, r, s); /* core = Uc, vv = Vc */
    if (tall) {
        /* a = Q R = Q Uc S Vc^T : u = Q Uc, vt = Vc^T */
        for (int64_t j = 0; j < r; ++j)
            for (int64_t i = 0; i < m; ++i) {
                T acc = 0;
                for (int64_t l = 0; l < r; ++l) acc += w[i + l * M] * core[l + j * r];
                u[i + j * ldu] = acc;
            }
        for (int64_t j = 0; j < n; ++j)
            for (int64_t i = 0; i < r; ++i) vt[i + j * ldvt] = vv[j + i * r];
    } else {
        /* a = L Qt, L = Uc S Vc^T : u = Uc, vt = Vc^T Q^T */
        for (int64_t j = 0; j < r; ++j)
            for (int64_t i = 0; i < m; ++i) u[i + j * ldu] = core[i + j * r];
        for (int64_t j = 0; j < n; ++j)
            for (int64_t i = 0; i < r; ++i) {
                T acc = 0;
                for (int64_t l = 0; l < r; ++l) acc += vv[l + i * r] * w[j + l * M];
                vt[i + j * ldvt] = acc;
            }
    }
    free(w); free(tau); free(jp); free(core); free(vv);
    return sweeps;
}

/* ---- C = op(A) op(B), column-major, plain triple loop ----------------------- */
void FN(gemm)(int ta, int tb, int64_t m, int64_t n, int64_t k, const T *a, int64_t lda,
              const T *b, int64_t ldb, T *c, int64_t ldc) {
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < m; ++i) {
            T acc = 0;
            for (int64_t l = 0; l < k; ++l) {
                T av = ta ? a[l + i * lda] : a[i + l * lda];
                T bv = tb ? b[j + l * ldb] : b[l + j * ldb];
                acc += av * bv;
            }
            c[i + j * ldc] = acc;
        }
}

#undef FN
#undef CAT
#undef CAT_
