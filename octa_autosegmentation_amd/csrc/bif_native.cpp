// bif_native.cpp -- native host service for the leaf-bifurcation geometry (greenhouse.py:205-233).
//
// The reference computes it with numpy: np.mean, np.cov (centre, dot, scale) and np.linalg.eig (LAPACK dgeev;
// the sign it gives the dominant eigenvector decides child order). To return the very same bits without a
// Python call per request, this file calls the SAME BLAS/LAPACK numpy is linked against -- the
// libscipy_openblas64_ shipped inside the numpy wheel, located by the Python host and opened with dlopen
// (it is already mapped in the process) -- with the argument shapes numpy uses: sequential column sums for
// the means, one row-major dgemm(Trans, NoTrans) for X X^T, dgeev('N','V') on the column-major copy, ddot for
// the vector norms. The Python host checks this path bit-for-bit against the numpy formula on random
// requests before enabling it and keeps the numpy callback as the fallback (complex eigenpairs, or no
// library). Host code only (no HIP).
#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

typedef void (*dgemm_fn)(int order, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double *a, int64_t lda,
                         const double *b, int64_t ldb, double beta, double *c, int64_t ldc);
typedef double (*ddot_fn)(int64_t n, const double *x, int64_t incx, const double *y, int64_t incy);
typedef void (*dgeev_fn)(const char *jobvl, const char *jobvr, const int64_t *n, double *a, const int64_t *lda, double *wr, double *wi,
                         double *vl, const int64_t *ldvl, double *vr, const int64_t *ldvr, double *work, const int64_t *lwork,
                         int64_t *info);

struct Native {
    void *handle = nullptr;
    dgemm_fn dgemm = nullptr;
    ddot_fn ddot = nullptr;
    dgeev_fn dgeev = nullptr;
    std::vector<double> kappa, cs, sn;
    octa_bif_fn fallback = nullptr;
    void *fallback_user = nullptr;
    long n_native = 0, n_fallback = 0;
    std::atomic<int64_t> lwork3{0};     // dgeev's workspace size for a 3 x 3 matrix with right eigenvectors (0: not asked yet)
} g;

// returns false when the request must go to the numpy fallback
bool eval_one(const octa_bif_request &q, double *out6) {
    const int n = q.n;
    if (n < 2 || n > OCTA_BIF_MAX_ATTS) return false;
    double cs = 0, sn = 0;
    bool have = false;
    for (size_t k = 0; k < g.kappa.size(); k++)
        if (g.kappa[k] == q.kappa) { cs = g.cs[k]; sn = g.sn[k]; have = true; }
    if (!have) return false;
    const double *A = q.atts;
    // c = np.mean(atts, axis=0): sequential sums down the rows, then / n
    double c[3];
    for (int k = 0; k < 3; k++) {
        double s = A[k];
        for (int i = 1; i < n; i++) s += A[3 * i + k];
        c[k] = s / (double)n;
    }
    // axis towards the centroid (normalize_vector: divide only when the norm is not zero)
    double ac[3] = {c[0] - q.pos[0], c[1] - q.pos[1], c[2] - q.pos[2]};
    double nrm = std::sqrt(g.ddot(3, ac, 1, ac, 1));
    if (nrm != 0.0) for (int k = 0; k < 3; k++) ac[k] = ac[k] / nrm;
    // np.cov of X = (atts - c)^T: row means (again sequential), in-place centring, dot, scale
    double Z[3 * OCTA_BIF_MAX_ATTS], Y[3 * OCTA_BIF_MAX_ATTS];   // Z[i][k] = X[k][i] (on the stack: a request is answered while a workgroup waits)
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) Z[3 * i + k] = A[3 * i + k] - c[k];
    double avg[3];
    for (int k = 0; k < 3; k++) {
        double s = Z[k];
        for (int i = 1; i < n; i++) s += Z[3 * i + k];
        avg[k] = s / (double)n;
    }
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) Z[3 * i + k] -= avg[k];
    memcpy(Y, Z, sizeof(double) * 3 * (size_t)n);  // X.T.conj(): a C-contiguous copy
    double cm[9];
    // X (3 x n, F-contiguous = Z transposed) @ Y (n x 3, C-contiguous): row-major gemm with A transposed
    g.dgemm(101 /*RowMajor*/, 112 /*Trans*/, 111 /*NoTrans*/, 3, 3, n, 1.0, Z, 3, Y, 3, 0.0, cm, 3);
    const double fact = 1.0 / (double)(n - 1);
    for (int k = 0; k < 9; k++) cm[k] *= fact;
    // np.linalg.eig: dgeev on the column-major copy, right eigenvectors only
    double af[9], wr[3], wi[3], vr[9], vl[1], wq;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) af[j * 3 + i] = cm[i * 3 + j];
    const int64_t N = 3, one = 1;
    int64_t info = 0;
    // the workspace size LAPACK asks for depends on (jobvl, jobvr, n) only: queried once (numpy queries per call, with the same answer)
    int64_t lwork = g.lwork3.load(std::memory_order_relaxed);
    if (lwork == 0) {
        int64_t query = -1;
        double af2[9];
        memcpy(af2, af, sizeof(af));
        g.dgeev("N", "V", &N, af2, &N, wr, wi, vl, &one, vr, &N, &wq, &query, &info);
        if (info != 0) return false;
        lwork = (int64_t)wq;
        if (lwork < 1 || lwork > 4096) return false;
        g.lwork3.store(lwork, std::memory_order_relaxed);
    }
    double work[4096];
    g.dgeev("N", "V", &N, af, &N, wr, wi, vl, &one, vr, &N, work, &lwork, &info);
    if (info != 0) return false;
    for (int k = 0; k < 3; k++) {
        if (wi[k] != 0.0) return false;                 // complex pair: numpy switches dtype, let it handle that
        if (!(wr[k] == wr[k])) return false;
    }
    int jm = 0;                                          // np.argmax: first maximum
    for (int k = 1; k < 3; k++) if (wr[k] > wr[jm]) jm = k;
    const double dl[3] = {vr[jm * 3 + 0], vr[jm * 3 + 1], vr[jm * 3 + 2]};
    for (int s = 0; s < 2; s++) {
        double a[3];
        for (int k = 0; k < 3; k++) a[k] = (s == 0) ? cs * ac[k] + sn * dl[k] : cs * ac[k] - sn * dl[k];
        double na = std::sqrt(g.ddot(3, a, 1, a, 1));
        for (int k = 0; k < 3; k++) out6[3 * s + k] = q.pos[k] + a[k] / na * q.d;
    }
    return true;
}

}  // namespace

extern "C" int octa_bif_native_init(const char *blas_path, int n_kappa, const double *kappas, const double *cs, const double *sn,
                                    octa_bif_fn fallback, void *fallback_user) {
    if (!blas_path || !fallback) { octa::set_error("octa_bif_native_init: null argument"); return -2; }
    void *h = dlopen(blas_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { octa::set_error("octa_bif_native_init: dlopen(%s) failed: %s", blas_path, dlerror()); return -1; }
    g.dgemm = (dgemm_fn)dlsym(h, "scipy_cblas_dgemm64_");
    g.ddot = (ddot_fn)dlsym(h, "scipy_cblas_ddot64_");
    g.dgeev = (dgeev_fn)dlsym(h, "scipy_dgeev_64_");
    if (!g.dgemm || !g.ddot || !g.dgeev) { octa::set_error("octa_bif_native_init: BLAS/LAPACK symbols not found in %s", blas_path); g.dgemm = nullptr; return -1; }
    g.handle = h;
    g.lwork3.store(0);
    g.kappa.assign(kappas, kappas + n_kappa);
    g.cs.assign(cs, cs + n_kappa);
    g.sn.assign(sn, sn + n_kappa);
    g.fallback = fallback;
    g.fallback_user = fallback_user;
    return 0;
}

// an octa_bif_fn: native where possible, the registered numpy callback for the rest
extern "C" void octa_bif_native(int n_req, const octa_bif_request *reqs, double *out6, void *) {
    std::vector<int> todo;
    for (int i = 0; i < n_req; i++) {
        bool ok = g.dgemm && eval_one(reqs[i], out6 + 6 * (size_t)i);
        if (!ok) todo.push_back(i);
    }
    g.n_native += n_req - (long)todo.size();
    g.n_fallback += (long)todo.size();
    if (!todo.empty() && g.fallback) {
        std::vector<octa_bif_request> sub(todo.size());
        std::vector<double> res(todo.size() * 6);
        for (size_t k = 0; k < todo.size(); k++) sub[k] = reqs[todo[k]];
        g.fallback((int)todo.size(), sub.data(), res.data(), g.fallback_user);
        for (size_t k = 0; k < todo.size(); k++) memcpy(out6 + 6 * (size_t)todo[k], &res[6 * k], 6 * sizeof(double));
    }
}

extern "C" int octa_bif_native_counts(int64_t *h_out2) {
    if (!h_out2) return -2;
    h_out2[0] = g.n_native;
    h_out2[1] = g.n_fallback;
    return 0;
}
