// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into, imported by or called from the
// product path (sift_amd/). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use anything under oracle/.
//
// PARITY PINNED AGAINST THE REFERENCE'S OWN BINARY.  The reference (snowiow/SIFT) has no tests or golden vectors and
// cannot be rebuilt here (needs Vigra 1.11, OpenCV, Boost; none present, no network), and its prebuilt executable
// /root/reference/bin/arch_x64/sift cannot be started (the same libraries are DT_NEEDED).  But Sift::calculate and
// the sift::alg functions INSIDE that executable, with every Vigra template they use compiled in, only need libc /
// libm / libstdc++: oracle/refexec maps the executable into its own process and calls them.  On every input of
// tests/golden/make_ref_pins.py (12 calculate() cases up to 1000x760, three of them ending in Vigra's exception;
// blur at 8 sigmas on 3 shapes, reduce / increaseToNextLevel, dog, vertexParabola) this oracle returns what the
// reference binary returns, bit for bit: point records, orientations, descriptors, every Gaussian level, exception
// text (tests/test_ref_pins.py against tests/golden/refpin.npz; the live comparison runs where the reference is
// mounted).  Not covered by the pin: the `u16_t size` truncation (App. B-7 needs > 65535 survivors, i.e. ~8 Mpx,
// hours in the reference's O(K x N) descriptor stage) and octaves > 4 at 4K.
//
// The restatement was written from the published algorithms of the Vigra 1.11 routines the reference calls,
// cross-checked against a static disassembly (objdump) of that binary.  Addresses "bin@0x..." below refer to it.
//
// Restated routines (Vigra 1.11, soname libvigraimpex.so.11; call sites in the reference:
// algorithms.cpp:13-19,33,46,175 and sift.cpp:306,311):
//   Kernel1D<float>::initGaussian           bin@0x41a900
//   separableConvolveX/Y -> convolveLine -> internalConvolveLineReflect   bin@0x4150d0,0x414710,0x41ba90
//   resizeImageNoInterpolation              bin@0x419ac0
//   linalg::inverse / qrDecomposition       bin@0x428b90 / 0x428900
//   linalg::linearSolve / linearSolveQRReplace   bin@0x41fe30 / 0x41f1c0
//   detail::qrTransformToTriangularImpl     bin@0x41e040
//   detail::qrHouseholderStepImpl           bin@0x41c9c0   (householderVector inlined)
//   detail::qrTransformToLowerTriangular    bin@0x41ec40
//   detail::applyHouseholderColumnReflections   bin@0x41c6f0
//   linearSolveUpper/LowerTriangular        bin@0x41a1b0 / 0x419fa0
//   dot, MultiArrayView::norm               bin@0x41c1d0 / 0x4174c0
//
// All float arithmetic is written one IEEE operation per C++ operator and must be compiled with
// -ffp-contract=off (the reference was built -O3 without -march => SSE2 scalar, no FMA).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace oracle {

// vigra::PreconditionViolation analogue (a std::exception in Vigra too). what() carries Vigra's
// message text ("Precondition violation!\n<message>").
struct PreconditionViolation : std::runtime_error {
    explicit PreconditionViolation(const std::string& msg)
        : std::runtime_error("Precondition violation!\n" + msg) {}
};

// ---------------------------------------------------------------------------------------------
// Image: contiguous float image, x fastest — layout of vigra::MultiArray<2,float> (img(x,y)).
// ---------------------------------------------------------------------------------------------
struct Img {
    long w = 0, h = 0;
    std::vector<float> d;
    Img() = default;
    Img(long w_, long h_) : w(w_), h(h_), d((size_t)w_ * (size_t)h_, 0.0f) {}
    float& operator()(long x, long y) { return d[(size_t)x + (size_t)y * (size_t)w]; }
    const float& operator()(long x, long y) const { return d[(size_t)x + (size_t)y * (size_t)w]; }
};

// ---------------------------------------------------------------------------------------------
// Kernel1D<float>::initGaussian(double std_dev) == initGaussian(std_dev, norm = 1.0f)
// (bin@0x41a900).  Gaussian<float> functor: sigma2 = float(-0.5/sigma/sigma) (double divisions),
// norm = float(1/sqrt(2 pi)/sigma), g(x) = norm * expf(x*x*sigma2); then normalize(1.0f) with a
// float running sum.
// ---------------------------------------------------------------------------------------------
struct Kernel1D {
    std::vector<float> k;  // taps for x = -radius .. +radius
    int radius = 0;
};

inline Kernel1D initGaussian(double std_dev) {
    if (!(std_dev >= 0.0))
        throw PreconditionViolation("Kernel1D::initGaussian(): Standard deviation must be >= 0.");
    Kernel1D ker;
    if (std_dev > 0.0) {
        const float sigma = (float)std_dev;
        const float sigma2 = (float)(-0.5 / sigma / sigma);
        const float norm = (float)(1.0 / std::sqrt(2.0 * M_PI) / sigma);
        int radius = (int)(3.0 * std_dev + 0.5);
        if (radius == 0) radius = 1;
        ker.radius = radius;
        ker.k.reserve((size_t)radius * 2 + 1);
        for (float x = -(float)radius; x <= (float)radius; ++x) {
            const float x2 = x * x;
            const float g = norm * std::exp(x2 * sigma2);  // float overload == expf
            ker.k.push_back(g);
        }
    } else {
        ker.k.push_back(1.0f);
        ker.radius = 0;
    }
    float sum = 0.0f;  // NumericTraits<float>::RealPromote == float
    for (float v : ker.k) sum += v;
    if (sum == 0.0f)
        throw PreconditionViolation(
            "Kernel1D<ARITHTYPE>::normalize(): Cannot normalize a kernel with sum = 0");
    sum = 1.0f / sum;
    for (float& v : ker.k) v = v * sum;
    return ker;
}

// ---------------------------------------------------------------------------------------------
// internalConvolveLineReflect (bin@0x41ba90) on a strided line of length w.  For every output x
// the source positions p = x-r .. x+r are visited in ASCENDING order; the kernel pointer starts
// at kernel[+r] and decrements.  Out-of-range p reflect without repeating the edge pixel:
// p < 0 -> -p ; p >= w -> 2(w-1)-p.  One mulss + one addss per term.
// ---------------------------------------------------------------------------------------------
inline void convolveLineReflect(const float* src, long sstride, long w, float* dst, long dstride,
                                const Kernel1D& ker) {
    const int r = ker.radius;
    const float* kc = ker.k.data() + r;  // kc[i], i in [-r, r]
    for (long x = 0; x < w; ++x) {
        float sum = 0.0f;
        long ik = r;  // kernel index, decrements as the source position ascends
        for (long p = x - r; p <= x + r; ++p, --ik) {
            long q = p;
            if (q < 0) q = -q;
            else if (q >= w) q = 2 * (w - 1) - q;
            sum += kc[ik] * src[q * sstride];
        }
        dst[x * dstride] = sum;
    }
}

// alg::convolveWithGauss (algorithms.cpp:10-22): separableConvolveX into tmp, then
// separableConvolveY into result.  Preconditions (bin@0x415353, 0x415525).
inline Img convolveWithGauss(const Img& img, float sigma) {
    const Kernel1D ker = initGaussian((double)sigma);
    Img tmp(img.w, img.h), res(img.w, img.h);
    if (!(img.w >= ker.radius + 1))
        throw PreconditionViolation("separableConvolveX(): kernel longer than line\n");
    for (long y = 0; y < img.h; ++y)
        convolveLineReflect(img.d.data() + (size_t)y * img.w, 1, img.w,
                            tmp.d.data() + (size_t)y * img.w, 1, ker);
    if (!(img.h >= ker.radius + 1))
        throw PreconditionViolation("separableConvolveY(): kernel longer than line\n");
    for (long x = 0; x < img.w; ++x)
        convolveLineReflect(tmp.d.data() + x, img.w, img.h, res.d.data() + x, img.w, ker);
    return res;
}

// resizeLineNoInterpolation index map (bin@0x419d98-0x419dc3): accumulated double.
inline std::vector<int> resizeIndexMap(long wold, long wnew) {
    std::vector<int> idx((size_t)wnew);
    if (wnew == 1) {
        idx[0] = 0;
        return idx;
    }
    const double dx = (double)(wold - 1) / (double)(wnew - 1);
    double x = 0.5;
    for (long i = 0; i < wnew; ++i, x += dx) idx[(size_t)i] = (int)x;
    return idx;
}

// resizeImageNoInterpolation (bin@0x419ac0): columns resampled in y first, then rows in x.
inline Img resizeImageNoInterpolation(const Img& src, long wnew, long hnew) {
    if (!(src.w > 1 && src.h > 1))
        throw PreconditionViolation("resizeImageNoInterpolation(): Source image too small.\n");
    if (!(wnew > 1 && hnew > 1))
        throw PreconditionViolation("resizeImageNoInterpolation(): Destination image too small.\n");
    const std::vector<int> iy = resizeIndexMap(src.h, hnew);
    const std::vector<int> ix = resizeIndexMap(src.w, wnew);
    Img tmp(src.w, hnew);
    for (long x = 0; x < src.w; ++x)
        for (long y = 0; y < hnew; ++y) tmp(x, y) = src(x, iy[(size_t)y]);
    Img out(wnew, hnew);
    for (long y = 0; y < hnew; ++y)
        for (long x = 0; x < wnew; ++x) out(x, y) = tmp(ix[(size_t)x], y);
    return out;
}

// ---------------------------------------------------------------------------------------------
// Small dense linear algebra (vigra::linalg).  vigra::Matrix(row, col) is stored with the ROW
// index fastest (MultiArray<2,T> first index).  MV is a strided view like MultiArrayView.
// ---------------------------------------------------------------------------------------------
struct MV {
    float* p = nullptr;
    long m = 0, n = 0;    // rows, columns
    long rs = 0, cs = 0;  // row stride, column stride
    float& operator()(long i, long j) const { return p[i * rs + j * cs]; }
    MV sub(long i0, long j0, long i1, long j1) const {
        return MV{p + i0 * rs + j0 * cs, i1 - i0, j1 - j0, rs, cs};
    }
    MV T() const { return MV{p, n, m, cs, rs}; }
};

struct Mat {
    long m, n;
    std::vector<float> d;
    Mat(long m_, long n_) : m(m_), n(n_), d((size_t)(m_ * n_), 0.0f) {}
    explicit Mat(const MV& v) : m(v.m), n(v.n), d((size_t)(v.m * v.n)) {
        for (long j = 0; j < n; ++j)
            for (long i = 0; i < m; ++i) d[(size_t)(i + j * m)] = v(i, j);
    }
    MV v() { return MV{d.data(), m, n, 1, m}; }
    float& operator()(long i, long j) { return d[(size_t)(i + j * m)]; }
};

// dot of two column vectors (bin@0x41c1d0): float, sequential from index 0.
inline float dotCol(const MV& x, const MV& y) {
    float ret = 0.0f;
    for (long i = 0; i < y.m; ++i) ret += x(i, 0) * y(i, 0);
    return ret;
}
// dot(row 1xn, column nx1)
inline float dotRowCol(const MV& x, const MV& y) {
    float ret = 0.0f;
    for (long i = 0; i < y.m; ++i) ret += x(0, i) * y(i, 0);
    return ret;
}
// squaredNorm / norm of a column vector (bin@0x4174c0): float sequential, sqrtf.
inline float sqNormCol(const MV& v) {
    float s = 0.0f;
    for (long i = 0; i < v.m; ++i) s += v(i, 0) * v(i, 0);
    return s;
}
inline float normCol(const MV& v) { return std::sqrt(sqNormCol(v)); }

// detail::qrHouseholderStepImpl with detail::householderVector inlined (bin@0x41c9c0).
inline bool qrHouseholderStep(long i, MV r, MV rhs, MV hh) {
    const long m = r.m, n = r.n, rhsCount = rhs.n;
    MV v = r.sub(i, i, m, i + 1);
    std::vector<float> u((size_t)(m - i), 0.0f);
    float vnorm = (v(0, 0) > 0.0f) ? -normCol(v) : normCol(v);
    const float f = std::sqrt(vnorm * (vnorm - v(0, 0)));
    bool nontrivial;
    if (f == 0.0f) {
        nontrivial = false;  // u stays 0
    } else {
        u[0] = (v(0, 0) - vnorm) / f;
        for (long k = 1; k < m - i; ++k) u[(size_t)k] = v(k, 0) / f;
        nontrivial = true;
    }
    r(i, i) = vnorm;
    for (long k = i + 1; k < m; ++k) r(k, i) = 0.0f;
    if (hh.n == n)
        for (long k = 0; k < m - i; ++k) hh(i + k, i) = u[(size_t)k];
    if (nontrivial) {
        MV uv{u.data(), m - i, 1, 1, m - i};
        for (long k = i + 1; k < n; ++k) {
            MV c = r.sub(i, k, m, k + 1);
            const float d = dotCol(c, uv);
            for (long t = 0; t < m - i; ++t) {
                const float prod = d * u[(size_t)t];  // temporary (dot * u) formed first
                c(t, 0) -= prod;
            }
        }
        for (long k = 0; k < rhsCount; ++k) {
            MV c = rhs.sub(i, k, m, k + 1);
            const float d = dotCol(c, uv);
            for (long t = 0; t < m - i; ++t) {
                const float prod = d * u[(size_t)t];
                c(t, 0) -= prod;
            }
        }
    }
    return r(i, i) != 0.0f;
}

// argMax over a float range (strict >, start value -FLT_MAX, first maximum wins); -1 if none.
inline int argMaxRange(const std::vector<float>& a, long from, long to) {
    float vopt = -FLT_MAX;
    int best = -1;
    for (long k = from; k < to; ++k)
        if (vopt < a[(size_t)k]) {
            vopt = a[(size_t)k];
            best = (int)(k - from);
        }
    return best;
}

// detail::qrTransformToTriangularImpl (bin@0x41e040).  permutation.size()>0 <=> column pivoting.
// NOTE (verified at bin@0x41e58e-0x41e5d1): the pivot-norm downdate subtracts r(k,l)^2 — row k,
// the loop index — exactly as Vigra 1.11's source does.
inline unsigned qrTransformToTriangularImpl(MV r, MV rhs, MV hh, std::vector<long>& permutation,
                                            double epsilon) {
    const long m = r.m, n = r.n;
    const long maxRank = m < n ? m : n;
    if (!(m >= n))
        throw PreconditionViolation(
            "qrTransformToTriangularImpl(): Coefficient matrix with at least as many rows as "
            "columns required.");
    bool pivoting = permutation.size() > 0;
    if (pivoting && n != (long)permutation.size())
        throw PreconditionViolation(
            "qrTransformToTriangularImpl(): Permutation array size mismatch.");
    if (n == 0) return 0;

    std::vector<float> csn;
    if (pivoting) {
        csn.resize((size_t)n);
        for (long k = 0; k < n; ++k) csn[(size_t)k] = sqNormCol(r.sub(0, k, m, k + 1));
        const int pivot = argMaxRange(csn, 0, n);
        if (pivot > 0) {  // pivot == -1 (all NaN) is UB in Vigra; treated as "no swap" here
            for (long i = 0; i < m; ++i) std::swap(r(i, 0), r(i, pivot));
            std::swap(csn[0], csn[(size_t)pivot]);
            std::swap(permutation[0], permutation[(size_t)pivot]);
        }
    }
    qrHouseholderStep(0, r, rhs, hh);

    long rank = 1;
    float maxSV = std::fabs(r(0, 0)), minSV = maxSV;
    double tolerance = (epsilon == 0.0) ? (double)((float)m * maxSV * FLT_EPSILON) : epsilon;
    // n < 4 => simple singular value approximation (the only case the reference reaches: n <= 3)
    if (n >= 4)
        throw std::logic_error("oracle: incremental SV approximation (n >= 4) not restated");
    if ((double)minSV <= tolerance) {
        rank = 0;
        pivoting = false;
    }
    for (long k = 1; k < maxRank; ++k) {
        if (pivoting) {
            for (long l = k; l < n; ++l) csn[(size_t)l] -= r(k, l) * r(k, l);
            const int a = argMaxRange(csn, k, n);
            const long pivot = k + a;
            if (a > 0) {
                for (long i = 0; i < m; ++i) std::swap(r(i, k), r(i, pivot));
                std::swap(csn[(size_t)k], csn[(size_t)pivot]);
                std::swap(permutation[(size_t)k], permutation[(size_t)pivot]);
            }
        }
        qrHouseholderStep(k, r, rhs, hh);
        const float nv = std::fabs(r(k, k));
        maxSV = (nv < maxSV) ? maxSV : nv;  // std::max(nv, maxSV) — maxss (bin@0x41e7eb)
        minSV = (minSV < nv) ? minSV : nv;  // std::min(nv, minSV) — minss (bin@0x41e7fb)
        if (epsilon == 0.0) tolerance = (double)((float)m * maxSV * FLT_EPSILON);
        if ((double)minSV > tolerance)
            ++rank;
        else
            pivoting = false;
    }
    return (unsigned)rank;
}

// linearSolveUpperTriangular (bin@0x41a1b0)
inline bool linearSolveUpperTriangular(const MV& r, const MV& b, MV x) {
    const long m = r.m, rhsCount = b.n;
    for (long k = 0; k < rhsCount; ++k)
        for (long i = m - 1; i >= 0; --i) {
            if (r(i, i) == 0.0f) return false;
            float sum = b(i, k);
            for (long j = i + 1; j < m; ++j) sum -= r(i, j) * x(j, k);
            x(i, k) = sum / r(i, i);
        }
    return true;
}
// linearSolveLowerTriangular (bin@0x419fa0)
inline bool linearSolveLowerTriangular(const MV& l, const MV& b, MV x) {
    const long m = l.n, n = b.n;
    for (long k = 0; k < n; ++k)
        for (long i = 0; i < m; ++i) {
            if (l(i, i) == 0.0f) return false;
            float sum = b(i, k);
            for (long j = 0; j < i; ++j) sum -= l(i, j) * x(j, k);
            x(i, k) = sum / l(i, i);
        }
    return true;
}

// linalg::inverse, square case (bin@0x428b90): q = I, r = v, Householder QR without pivoting
// applied to r and to transpose(q); false if rank < n; res = R^-1 * Q^T by back-substitution.
inline bool inverse(const MV& v, MV res) {
    const long n = v.n, m = v.m;
    if (m != n) throw std::logic_error("oracle: inverse() restated for square matrices only");
    Mat r(v), q(n, n);
    for (long i = 0; i < n; ++i) q(i, i) = 1.0f;
    MV tq = q.v().T();
    std::vector<long> noPivoting;
    MV noHouseholder{};
    const unsigned rank = qrTransformToTriangularImpl(r.v(), tq, noHouseholder, noPivoting, 0.0);
    if ((long)rank != n) return false;
    linearSolveUpperTriangular(r.v(), q.v().T(), res);
    return true;
}

// detail::applyHouseholderColumnReflections (bin@0x41c6f0)
inline void applyHouseholderColumnReflections(const MV& H, MV res) {
    const long m = H.m, n = H.n, rhsCount = res.n;
    for (long k = n - 1; k >= 0; --k) {
        MV u = H.sub(k, k, m, k + 1);
        for (long l = 0; l < rhsCount; ++l) {
            MV c = res.sub(k, l, m, l + 1);
            const float d = dotCol(c, u);
            for (long t = 0; t < m - k; ++t) {
                const float prod = d * u(t, 0);
                c(t, 0) -= prod;
            }
        }
    }
}

// linalg::linearSolve(A, b, res, "QR") (bin@0x41fe30 -> linearSolveQRReplace bin@0x41f1c0) for
// m >= n.  Returns rank == n.  Rank-deficient systems take the minimum-norm path; its
// qrTransformToLowerTriangular call passes an EMPTY rhs (bin@0x41f872-0x41f8b4), hence an empty
// permutation: the lower-triangular transform runs WITHOUT pivoting.
inline bool linearSolve(const MV& Ain, const MV& bin, MV res) {
    const long n = Ain.n, m = Ain.m;
    if (!(n <= m)) throw std::logic_error("oracle: linearSolve restated for m >= n only");
    Mat A(Ain), b(bin);
    const long rhsCount = res.n;
    std::vector<long> permutation((size_t)n);
    for (long k = 0; k < n; ++k) permutation[(size_t)k] = k;
    MV noHouseholder{};
    const long rank =
        (long)qrTransformToTriangularImpl(A.v(), b.v(), noHouseholder, permutation, 0.0);
    Mat permutedSolution(n, rhsCount);
    if (rank < n) {
        Mat householderMatrix(n, rank);
        if (rank > 0) {
            MV Asub = A.v().sub(0, 0, rank, n);
            // qrTransformToLowerTriangular(Asub, <empty rhs>, transpose(householderMatrix)):
            // QR of transpose(Asub) (n x rank), Householder vectors stored, no pivoting.
            std::vector<long> noPivoting;
            MV noRhs{};
            qrTransformToTriangularImpl(Asub.T(), noRhs, householderMatrix.v(), noPivoting, 0.0);
            linearSolveLowerTriangular(A.v().sub(0, 0, rank, rank), b.v().sub(0, 0, rank, rhsCount),
                                       permutedSolution.v().sub(0, 0, rank, rhsCount));
            applyHouseholderColumnReflections(householderMatrix.v(), permutedSolution.v());
        }
    } else {
        linearSolveUpperTriangular(A.v().sub(0, 0, rank, rank), b.v().sub(0, 0, rank, rhsCount),
                                   permutedSolution.v());
    }
    // detail::inverseRowPermutation
    for (long k = 0; k < n; ++k)
        for (long l = 0; l < rhsCount; ++l)
            res(permutation[(size_t)k], l) = permutedSolution(k, l);
    return rank == n;
}

}  // namespace oracle
