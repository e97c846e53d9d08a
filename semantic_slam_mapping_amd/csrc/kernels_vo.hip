// kernels_vo.hip -- stereo visual odometry on the quad matches: VisualOdometryStereo::estimateMotion
// (/root/reference/src/vo_stereo.cpp:47-152: 3-point RANSAC hypotheses by Gauss-Newton, consensus by reprojection
// error, refinement on the consensus set).  SURVEY.md s.8(f) rank 3; the contracts (sin/cos, LU, summation order of
// the refinement, the reference's rdrx11 term) are those of oracle/vo.c and are restated here line by line, in f64
// with -ffp-contract=off, so the result is bit-identical to the oracle.
//   vo_ransac_kernel : one wave per hypothesis.  Lane 0 runs the <= 22 Gauss-Newton steps on its 3 matches (12 rows,
//                      summed in row order like the reference); then the whole wave votes: every lane tests matches
//                      lane, lane+64, ... and the ballots give the consensus size.
//   vo_refine_kernel : one wave.  Picks the first hypothesis with the largest consensus, rebuilds its inlier list in
//                      index order (ballot prefix), then Gauss-Newton on all inliers: lane l accumulates the 42 normal-
//                      equation sums of inliers l, l+64, ... and a butterfly over lane distance 1..32 adds the lanes up
//                      (every lane ends with the same sums and solves the 6x6 system redundantly -- no broadcast).
// Tiny f64 work, latency-bound by construction; it exists so that the stereo path ends in a pose on the device.
#include "ssm_internal.h"
#include <cfloat>

struct VoRot {
    double r00, r01, r02, r10, r11, r12, r20, r21, r22;
    double rdrx10, rdrx11, rdrx12, rdrx20, rdrx21, rdrx22;
    double rdry00, rdry01, rdry02, rdry10, rdry11, rdry12, rdry20, rdry21, rdry22;
    double rdrz00, rdrz01, rdrz10, rdrz11, rdrz20, rdrz21;
    double tx, ty, tz;
};
// sin/cos contract of oracle/vo.c: Cody-Waite reduction by pi/2 in three pieces + the fdlibm kernel polynomials
__device__ __forceinline__ void vo_sincos64(double x, double* s, double* c)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double p1 = 1.57079632673412561417e+00, p2 = 6.07710050650619224932e-11, p3 = 2.02226624879595063154e-21;
    const double fn = rint(x * invpio2);
    double r = x - fn * p1;
    r = r - fn * p2;
    r = r - fn * p3;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sr = r + (r * z) * (S1 + z * ps);
    const double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    const double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    const long long k = (long long)fn;
    switch ((int)(k & 3)) {
        case 0: *s = sr;  *c = cr;  break;
        case 1: *s = cr;  *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}
__device__ __forceinline__ void vo_rot_make(const double tr[6], VoRot& R)
{
    double sx, cx, sy, cy, sz, cz;
    vo_sincos64(tr[0], &sx, &cx); vo_sincos64(tr[1], &sy, &cy); vo_sincos64(tr[2], &sz, &cz);
    R.tx = tr[3]; R.ty = tr[4]; R.tz = tr[5];
    R.r00 = +cy*cz;          R.r01 = -cy*sz;          R.r02 = +sy;
    R.r10 = +sx*sy*cz+cx*sz; R.r11 = -sx*sy*sz+cx*cz; R.r12 = -sx*cy;
    R.r20 = -cx*sy*cz+sx*sz; R.r21 = +cx*sy*sz+sx*cz; R.r22 = +cx*cy;
    R.rdrx10 = +cx*sy*cz-sx*sz; R.rdrx11 = -cx*sy*sz-sx*sz; R.rdrx12 = -cx*cy;        // rdrx11 as written at vo_stereo.cpp:289
    R.rdrx20 = +sx*sy*cz+cx*sz; R.rdrx21 = -sx*sy*sz+cx*cz; R.rdrx22 = -sx*cy;
    R.rdry00 = -sy*cz;          R.rdry01 = +sy*sz;          R.rdry02 = +cy;
    R.rdry10 = +sx*cy*cz;       R.rdry11 = -sx*cy*sz;       R.rdry12 = +sx*sy;
    R.rdry20 = -cx*cy*cz;       R.rdry21 = +cx*cy*sz;       R.rdry22 = -cx*sy;
    R.rdrz00 = -cy*sz;          R.rdrz01 = -cy*cz;
    R.rdrz10 = -sx*sy*sz+cx*cz; R.rdrz11 = -sx*sy*cz-cx*sz;
    R.rdrz20 = +cx*sy*sz+sx*cz; R.rdrz21 = +cx*sy*cz-sx*sz;
}
// one match: 4 predictions, optionally the 4 weighted residuals and the 4 x 6 Jacobian
template <bool WITH_J>
__device__ __forceinline__ void vo_point(const ssm_pmatch& m, const ssm_vo_params& P, const VoRot& R, double* J, double pred[4], double* res)
{
    const double dd = fmax((double)(m.u1p - m.u2p), 1.0);
    const double X1p = ((double)m.u1p - P.cu) * P.base / dd, Y1p = ((double)m.v1p - P.cv) * P.base / dd, Z1p = P.f * P.base / dd;
    const double X1c = R.r00*X1p + R.r01*Y1p + R.r02*Z1p + R.tx;
    const double Y1c = R.r10*X1p + R.r11*Y1p + R.r12*Z1p + R.ty;
    const double Z1c = R.r20*X1p + R.r21*Y1p + R.r22*Z1p + R.tz;
    const double obs[4] = { (double)m.u1c, (double)m.v1c, (double)m.u2c, (double)m.v2c };
    double weight = 1.0;
    if (P.reweighting) weight = 1.0 / (fabs(obs[0] - P.cu) / fabs(P.cu) + 0.05);
    const double X2c = X1c - P.base;
    if (WITH_J) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double X1cd, Y1cd, Z1cd;
            switch (j) {
                case 0: X1cd = 0; Y1cd = R.rdrx10*X1p + R.rdrx11*Y1p + R.rdrx12*Z1p; Z1cd = R.rdrx20*X1p + R.rdrx21*Y1p + R.rdrx22*Z1p; break;
                case 1: X1cd = R.rdry00*X1p + R.rdry01*Y1p + R.rdry02*Z1p; Y1cd = R.rdry10*X1p + R.rdry11*Y1p + R.rdry12*Z1p;
                        Z1cd = R.rdry20*X1p + R.rdry21*Y1p + R.rdry22*Z1p; break;
                case 2: X1cd = R.rdrz00*X1p + R.rdrz01*Y1p; Y1cd = R.rdrz10*X1p + R.rdrz11*Y1p; Z1cd = R.rdrz20*X1p + R.rdrz21*Y1p; break;
                case 3: X1cd = 1; Y1cd = 0; Z1cd = 0; break;
                case 4: X1cd = 0; Y1cd = 1; Z1cd = 0; break;
                default: X1cd = 0; Y1cd = 0; Z1cd = 1; break;
            }
            J[0 * 6 + j] = weight * P.f * (X1cd*Z1c - X1c*Z1cd) / (Z1c*Z1c);
            J[1 * 6 + j] = weight * P.f * (Y1cd*Z1c - Y1c*Z1cd) / (Z1c*Z1c);
            J[2 * 6 + j] = weight * P.f * (X1cd*Z1c - X2c*Z1cd) / (Z1c*Z1c);
            J[3 * 6 + j] = weight * P.f * (Y1cd*Z1c - Y1c*Z1cd) / (Z1c*Z1c);
        }
    }
    pred[0] = P.f * X1c / Z1c + P.cu; pred[1] = P.f * Y1c / Z1c + P.cv;
    pred[2] = P.f * X2c / Z1c + P.cu; pred[3] = P.f * Y1c / Z1c + P.cv;
    if (WITH_J) {
#pragma unroll
        for (int k = 0; k < 4; k++) res[k] = weight * (obs[k] - pred[k]);
    }
}
__device__ __forceinline__ bool vo_is_inlier(const ssm_pmatch& m, const ssm_vo_params& P, const VoRot& R)
{
    double pred[4];
    vo_point<false>(m, P, R, nullptr, pred, nullptr);
    const double d0 = (double)m.u1c - pred[0], d1 = (double)m.v1c - pred[1], d2 = (double)m.u2c - pred[2], d3 = (double)m.v2c - pred[3];
    return d0*d0 + d1*d1 + d2*d2 + d3*d3 < P.inlier_threshold * P.inlier_threshold;
}
// cv::solve(A, b, x, DECOMP_LU) as restated in oracle/vo.c (OpenCV 2.4 LU); b <- x; false = singular
__device__ __forceinline__ bool vo_solve6(double* A, double* b)
{
    const double eps = DBL_EPSILON * 100;
    for (int i = 0; i < 6; i++) {
        int k = i;
        for (int j = i + 1; j < 6; j++) if (fabs(A[j * 6 + i]) > fabs(A[k * 6 + i])) k = j;
        if (fabs(A[k * 6 + i]) < eps) return false;
        if (k != i) {
            for (int j = i; j < 6; j++) { const double t = A[i * 6 + j]; A[i * 6 + j] = A[k * 6 + j]; A[k * 6 + j] = t; }
            const double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        const double d = -1 / A[i * 6 + i];
        for (int j = i + 1; j < 6; j++) {
            const double alpha = A[j * 6 + i] * d;
            for (int kk = i + 1; kk < 6; kk++) A[j * 6 + kk] += alpha * A[i * 6 + kk];
            b[j] += alpha * b[i];
        }
        A[i * 6 + i] = -d;
    }
    for (int i = 5; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < 6; k++) s -= A[i * 6 + k] * b[k];
        b[i] = s * A[i * 6 + i];
    }
    return true;
}
enum { VO_UPDATED = 0, VO_FAILED = 1, VO_CONVERGED = 2 };
// adds the normal-equation terms of one match to acc (A row-major 36, then B 6), rows in order
__device__ __forceinline__ void vo_accumulate(const ssm_pmatch& m, const ssm_vo_params& P, const VoRot& R, double* acc)
{
    double J[24], pred[4], res[4];
    vo_point<true>(m, P, R, J, pred, res);
    for (int r = 0; r < 4; r++)
        for (int mm = 0; mm < 6; mm++) {
            for (int nn = 0; nn < 6; nn++) acc[mm * 6 + nn] += J[r * 6 + mm] * J[r * 6 + nn];
            acc[36 + mm] += J[r * 6 + mm] * res[r];
        }
}
__device__ __forceinline__ int vo_finish_step(double* acc, double tr[6], double eps)
{
    double A[36], b[6];
    for (int k = 0; k < 36; k++) A[k] = acc[k];
    for (int k = 0; k < 6; k++) b[k] = acc[36 + k];
    if (!vo_solve6(A, b)) return VO_FAILED;
    bool converged = true;
    for (int k = 0; k < 6; k++) { tr[k] += 1.0 * b[k]; if (fabs(b[k]) > eps) converged = false; }
    return converged ? VO_CONVERGED : VO_UPDATED;
}

__global__ void __launch_bounds__(64)
vo_ransac_kernel(const ssm_pmatch* __restrict__ m, int n, ssm_vo_params P, const int32_t* __restrict__ samples,
                 double* __restrict__ tr_all, int32_t* __restrict__ count)
{
    __shared__ double s_tr[6];
    __shared__ int s_result;
    const int k = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) {
        double tr[6] = {0, 0, 0, 0, 0, 0};
        int result = VO_UPDATED, iter = 0;
        while (result == VO_UPDATED) {
            VoRot R; vo_rot_make(tr, R);
            double acc[42];
            for (int q = 0; q < 42; q++) acc[q] = 0.0;
            for (int q = 0; q < 3; q++) vo_accumulate(m[samples[3 * k + q]], P, R, acc);
            result = vo_finish_step(acc, tr, 1e-6);
            if (iter++ > 20 || result == VO_CONVERGED) break;
        }
        for (int q = 0; q < 6; q++) { s_tr[q] = tr[q]; tr_all[6 * k + q] = tr[q]; }
        s_result = result;
    }
    __syncthreads();
    if (s_result == VO_FAILED) { if (lane == 0) count[k] = -1; return; }
    double tr[6];
    for (int q = 0; q < 6; q++) tr[q] = s_tr[q];
    VoRot R; vo_rot_make(tr, R);
    int c = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n && vo_is_inlier(m[i], P, R);
        c += __popcll(__ballot(in));
    }
    if (lane == 0) count[k] = c;
}
__global__ void __launch_bounds__(64)
vo_refine_kernel(const ssm_pmatch* __restrict__ m, int n, ssm_vo_params P, const double* __restrict__ tr_all, const int32_t* __restrict__ count, int iters,
                 double* __restrict__ tr_out, int32_t* __restrict__ inliers, int32_t* __restrict__ result /* [0] = n_inliers, [1] = success */)
{
    const int lane = threadIdx.x;
    // the first hypothesis with the largest consensus (the reference replaces only on a strictly larger set)
    int best = 0, bk = -1;
    for (int k = 0; k < iters; k++) { const int c = count[k]; if (c > best) { best = c; bk = k; } }
    double tr[6] = {0, 0, 0, 0, 0, 0};
    if (bk >= 0) for (int q = 0; q < 6; q++) tr[q] = tr_all[6 * bk + q];
    int na = 0;
    if (bk >= 0) {
        VoRot R; vo_rot_make(tr, R);
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool in = i < n && vo_is_inlier(m[i], P, R);
            const unsigned long long bal = __ballot(in);
            if (in) inliers[na + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            na += __popcll(bal);
        }
    }
    __syncthreads();                                         // the list is read back below by other lanes
    int success = 1;
    if (na >= 6) {
        int res = VO_UPDATED, iter = 0;
        while (res == VO_UPDATED) {
            VoRot R; vo_rot_make(tr, R);
            double acc[42];
            for (int q = 0; q < 42; q++) acc[q] = 0.0;
            for (int q = lane; q < na; q += 64) vo_accumulate(m[inliers[q]], P, R, acc);
            for (int s = 1; s < 64; s <<= 1)
                for (int q = 0; q < 42; q++) acc[q] = acc[q] + __shfl_xor(acc[q], s, 64);
            res = vo_finish_step(acc, tr, 1e-8);             // identical in every lane
            if (iter++ > 100 || res == VO_CONVERGED) break;
        }
        if (res != VO_CONVERGED) success = 0;
    } else success = 0;
    if (lane == 0) {
        for (int q = 0; q < 6; q++) tr_out[q] = tr[q];
        result[0] = na; result[1] = success;
    }
}
hipError_t k_vo_estimate(const ssm_pmatch* m, int n, const ssm_vo_params& P, const int32_t* samples, int iters,
                         double* tr_all, int32_t* count, double* tr_out, int32_t* inliers, int32_t* result, hipStream_t s)
{
    vo_ransac_kernel<<<iters, 64, 0, s>>>(m, n, P, samples, tr_all, count);
    vo_refine_kernel<<<1, 64, 0, s>>>(m, n, P, tr_all, count, iters, tr_out, inliers, result);
    return hipGetLastError();
}
