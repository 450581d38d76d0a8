// Register-resident 3x3 float linear algebra for the per-keypoint solves of the hot path:
//   * inverse(-H)            — Sift::_eliminateEdgeResponses, /root/reference/sift.cpp:306
//   * linearSolve(inv, D)    — sift.cpp:311
//   * linearSolve(a, b)      — alg::vertexParabola, /root/reference/algorithms.cpp:175
// The reference gets these from Vigra 1.11's Householder-QR (linear_solve.hxx:
// qrTransformToTriangularImpl, qrHouseholderStepImpl, linearSolveQRReplace, inverse).  Results
// must agree bit for bit with that arithmetic, so every step below keeps Vigra's operation order:
// float sequential sums from index 0, sqrtf norms, one rounding per operator (compile with
// -ffp-contract=off), column pivoting with "first strict maximum wins", the rank test
// minSV > m*maxSV*FLT_EPSILON evaluated in float and compared as double, and the minimum-norm path
// for rank-deficient systems.  Matrices are a[row][col]; all sizes are compile-time so everything
// lives in VGPRs (no runtime-indexed arrays, no scratch).
#pragma once
#include <float.h>

#include "fdlibm_atan2f.h"  // SIFT_HD

namespace sift_hip {

SIFT_HD float sqrt_rn(float x) { return __builtin_sqrtf(x); }

// One Householder step on column I of r (M x N).  Reflects the later columns of r and all NR
// columns of rhs; stores the Householder vector in hh[I..M-1][I] when STORE.
template <int I, int M, int N, int NR, bool STORE>
SIFT_HD bool householder_step(float (&r)[M][N], float (&rhs)[M][NR > 0 ? NR : 1], float (&hh)[M][N]) {
    constexpr int L = M - I;
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < L; ++t) s += r[I + t][I] * r[I + t][I];
    const float nrm = sqrt_rn(s);
    const float v0 = r[I][I];
    const float vnorm = (v0 > 0.0f) ? -nrm : nrm;
    const float f = sqrt_rn(vnorm * (vnorm - v0));
    float u[L];
    const bool nontrivial = !(f == 0.0f);
    if (nontrivial) {
        u[0] = (v0 - vnorm) / f;
#pragma unroll
        for (int t = 1; t < L; ++t) u[t] = r[I + t][I] / f;
    } else {
#pragma unroll
        for (int t = 0; t < L; ++t) u[t] = 0.0f;
    }
    r[I][I] = vnorm;
#pragma unroll
    for (int t = 1; t < L; ++t) r[I + t][I] = 0.0f;
    if (STORE) {
#pragma unroll
        for (int t = 0; t < L; ++t) hh[I + t][I] = u[t];
    }
    if (nontrivial) {
#pragma unroll
        for (int k = I + 1; k < N; ++k) {
            float d = 0.0f;
#pragma unroll
            for (int t = 0; t < L; ++t) d += r[I + t][k] * u[t];
#pragma unroll
            for (int t = 0; t < L; ++t) {
                const float prod = d * u[t];
                r[I + t][k] -= prod;
            }
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            float d = 0.0f;
#pragma unroll
            for (int t = 0; t < L; ++t) d += rhs[I + t][k] * u[t];
#pragma unroll
            for (int t = 0; t < L; ++t) {
                const float prod = d * u[t];
                rhs[I + t][k] -= prod;
            }
        }
    }
    return r[I][I] != 0.0f;
}

// argMax of csn[K..N-1]: strict >, start -FLT_MAX, first maximum wins; offset from K, -1 if none.
template <int K, int N>
SIFT_HD int argmax_from(const float (&csn)[N]) {
    float vopt = -FLT_MAX;
    int best = -1;
#pragma unroll
    for (int l = K; l < N; ++l)
        if (vopt < csn[l]) {
            vopt = csn[l];
            best = l - K;
        }
    return best;
}

template <int K, int M, int N>
SIFT_HD void swap_columns(float (&r)[M][N], float (&csn)[N], int (&perm)[N], int other) {
#pragma unroll
    for (int j = K + 1; j < N; ++j)
        if (j == other) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const float t = r[i][K];
                r[i][K] = r[i][j];
                r[i][j] = t;
            }
            const float c = csn[K];
            csn[K] = csn[j];
            csn[j] = c;
            const int p = perm[K];
            perm[K] = perm[j];
            perm[j] = p;
        }
}

template <int K, int M, int N, int NR, bool PIVOT, bool STORE>
struct QrLoop {
    SIFT_HD static void run(float (&r)[M][N], float (&rhs)[M][NR > 0 ? NR : 1], float (&hh)[M][N],
                            float (&csn)[N], int (&perm)[N], bool& pivoting, float& maxSV,
                            float& minSV, double& tol, int& rank) {
        if constexpr (K < (M < N ? M : N)) {
            if (PIVOT && pivoting) {
#pragma unroll
                for (int l = K; l < N; ++l) csn[l] -= r[K][l] * r[K][l];  // row K, as Vigra 1.11 does
                const int a = argmax_from<K, N>(csn);
                if (a > 0) swap_columns<K, M, N>(r, csn, perm, K + a);
            }
            householder_step<K, M, N, NR, STORE>(r, rhs, hh);
            const float nv = __builtin_fabsf(r[K][K]);
            maxSV = (nv < maxSV) ? maxSV : nv;   // std::max(nv, maxSV)
            minSV = (minSV < nv) ? minSV : nv;   // std::min(nv, minSV)
            tol = (double)((float)M * maxSV * FLT_EPSILON);
            if ((double)minSV > tol)
                ++rank;
            else
                pivoting = false;
            QrLoop<K + 1, M, N, NR, PIVOT, STORE>::run(r, rhs, hh, csn, perm, pivoting, maxSV, minSV,
                                                       tol, rank);
        }
    }
};

// detail::qrTransformToTriangularImpl for n < 4 (simple singular-value approximation), epsilon 0.
template <int M, int N, int NR, bool PIVOT, bool STORE>
SIFT_HD int qr_triangular(float (&r)[M][N], float (&rhs)[M][NR > 0 ? NR : 1], float (&hh)[M][N],
                          int (&perm)[N]) {
    float csn[N];
#pragma unroll
    for (int k = 0; k < N; ++k) csn[k] = 0.0f;
    bool pivoting = PIVOT;
    if (PIVOT) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < M; ++i) s += r[i][k] * r[i][k];
            csn[k] = s;
        }
        const int a = argmax_from<0, N>(csn);
        if (a > 0) swap_columns<0, M, N>(r, csn, perm, a);
    }
    householder_step<0, M, N, NR, STORE>(r, rhs, hh);
    int rank = 1;
    float maxSV = __builtin_fabsf(r[0][0]), minSV = maxSV;
    double tol = (double)((float)M * maxSV * FLT_EPSILON);
    if ((double)minSV <= tol) {
        rank = 0;
        pivoting = false;
    }
    QrLoop<1, M, N, NR, PIVOT, STORE>::run(r, rhs, hh, csn, perm, pivoting, maxSV, minSV, tol, rank);
    return rank;
}

// linalg::inverse for 3x3: res = R^-1 Q^T; false if rank < 3.
SIFT_HD bool inverse3(const float (&a)[3][3], float (&res)[3][3]) {
    float r[3][3], b[3][3], hh[3][3];
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            r[i][j] = a[i][j];
            b[i][j] = (i == j) ? 1.0f : 0.0f;
            hh[i][j] = 0.0f;
        }
    const int rank = qr_triangular<3, 3, 3, false, false>(r, b, hh, perm);
    if (rank != 3) return false;
    // linearSolveUpperTriangular(r, transpose(q) = b, res)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            float sum = b[i][k];
#pragma unroll
            for (int j = i + 1; j < 3; ++j) sum -= r[i][j] * res[j][k];
            res[i][k] = sum / r[i][i];
        }
    }
    return true;
}

// Minimum-norm tail of linearSolveQRReplace for rank RK in {1, 2}: QR of transpose(A[0:RK, 0:3])
// WITHOUT pivoting (the reference passes an empty rhs => empty permutation), forward
// substitution, then the stored Householder reflections applied to the solution.
template <int RK>
SIFT_HD void min_norm_tail(float (&A)[3][3], const float (&b)[3][1], float (&xp)[3]) {
    float rt[3][RK], hh[3][RK], none[3][1];
    int perm[RK];
#pragma unroll
    for (int j = 0; j < RK; ++j) perm[j] = j;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        none[i][0] = 0.0f;
#pragma unroll
        for (int j = 0; j < RK; ++j) {
            rt[i][j] = A[j][i];
            hh[i][j] = 0.0f;
        }
    }
    qr_triangular<3, RK, 0, false, true>(rt, none, hh, perm);
    // A[0:RK, 0:RK] is now lower triangular: A[j][i] = rt[i][j]
    bool ok = true;
#pragma unroll
    for (int i = 0; i < RK; ++i) {
        if (ok) {
            const float lii = rt[i][i];
            if (lii == 0.0f) {
                ok = false;  // linearSolveLowerTriangular returns early, rest stays 0
            } else {
                float sum = b[i][0];
#pragma unroll
                for (int j = 0; j < i; ++j) sum -= rt[j][i] * xp[j];
                xp[i] = sum / lii;
            }
        }
    }
    // applyHouseholderColumnReflections(hh (3 x RK), xp)
#pragma unroll
    for (int k = RK - 1; k >= 0; --k) {
        float d = 0.0f;
#pragma unroll
        for (int t = k; t < 3; ++t) d += xp[t] * hh[t][k];
#pragma unroll
        for (int t = k; t < 3; ++t) {
            const float prod = d * hh[t][k];
            xp[t] -= prod;
        }
    }
}

// linalg::linearSolve(A, b, res, "QR") for 3x3 / 3x1.  Returns rank == 3.  With DEFICIENT the
// rank-deficient minimum-norm solution is produced too (vertexParabola needs it); without, res
// is left unspecified when the function returns false (the edge filter discards it).
template <bool DEFICIENT>
SIFT_HD bool solve3(const float (&Ain)[3][3], const float (&bin)[3], float (&res)[3]) {
    float A[3][3], b[3][1], hh[3][3];
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        b[i][0] = bin[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[i][j] = Ain[i][j];
            hh[i][j] = 0.0f;
        }
    }
    const int rank = qr_triangular<3, 3, 1, true, false>(A, b, hh, perm);
    float xp[3] = {0.0f, 0.0f, 0.0f};
    if (rank == 3) {
        bool ok = true;
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            if (ok) {
                if (A[i][i] == 0.0f) {
                    ok = false;
                } else {
                    float sum = b[i][0];
#pragma unroll
                    for (int j = i + 1; j < 3; ++j) sum -= A[i][j] * xp[j];
                    xp[i] = sum / A[i][i];
                }
            }
        }
    } else if (DEFICIENT) {
        if (rank == 2) min_norm_tail<2>(A, b, xp);
        else if (rank == 1) min_norm_tail<1>(A, b, xp);
    }
    // inverseRowPermutation: res[perm[k]] = xp[k]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (perm[k] == j) res[j] = xp[k];
    }
    return rank == 3;
}

// alg::vertexParabola (algorithms.cpp:153-178)
SIFT_HD float vertex_parabola(unsigned short lnx, float lny, unsigned short px, float py,
                              unsigned short rnx, float rny) {
    float a[3][3], b[3], res[3] = {0.0f, 0.0f, 0.0f};
    a[0][0] = (float)((double)lnx * (double)lnx);
    a[1][0] = (float)((double)px * (double)px);
    a[2][0] = (float)((double)rnx * (double)rnx);
    a[0][1] = (float)lnx;
    a[1][1] = (float)px;
    a[2][1] = (float)rnx;
    a[0][2] = 0.0f;
    a[1][2] = 0.0f;
    a[2][2] = 0.0f;
    b[0] = lny;
    b[1] = py;
    b[2] = rny;
    solve3<true>(a, b, res);
    return -res[1] / (2.0f * res[0]);
}

}  // namespace sift_hip
