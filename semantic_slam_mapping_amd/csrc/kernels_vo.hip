// kernels_vo.hip -- stereo visual odometry on the quad matches: VisualOdometryStereo::estimateMotion
// (/root/reference/src/vo_stereo.cpp:47-152: 3-point RANSAC hypotheses by Gauss-Newton, consensus by reprojection
// error, refinement on the consensus set).  SURVEY.md s.8(f) rank 3; the contracts (sin/cos, LU, summation order of
// the refinement, the reference's rdrx11 term) are those of oracle/vo.c and are restated here line by line, in f64
// with -ffp-contract=off, so the result is bit-identical to the oracle.
//   vo_hyp_kernel    : one THREAD per hypothesis: the <= 22 Gauss-Newton steps on its 3 matches (12 rows, summed in row
//                      order like the reference)
//   vo_vote_kernel   : one wave per hypothesis: every lane tests matches lane, lane+64, ... and the ballots give the
//                      consensus size.
//   vo_refine_kernel : one wave.  Picks the first hypothesis with the largest consensus, rebuilds its inlier list in
//                      index order (ballot prefix), then Gauss-Newton on all inliers: lane l accumulates the 42 normal-
//                      equation sums of inliers l, l+64, ... and a butterfly over lane distance 1..32 adds the lanes up
//                      (every lane ends with the same sums and solves the 6x6 system redundantly -- no broadcast).
// Tiny f64 work, latency-bound by construction; it exists so that the stereo path ends in a pose on the device.
#include "ssm_internal.h"
#include <cfloat>

// Motion model of the stereo VO (the parametrisation of /root/reference/src/vo_stereo.cpp:280-345, own formulation): pose = Euler X-Y-Z rotation
// (alpha, beta, gamma) + translation.  rot[r][c] is the rotation, drot[p][r][c] its partial derivative by angle p.  The reference leaves some derivative
// terms out of its sums (d/d alpha does not touch row 0; d/d gamma has no column-2 term): VO_DTERMS[p][r] says how many leading terms of row r enter,
// so that the sums have the reference's operands in the reference's order (bit-exactness with oracle/vo.c rests on the order, not on the notation).
struct VoPose { double rot[3][3], t[3], drot[3][3][3]; };
__device__ __constant__ const int VO_DTERMS[3][3] = {{0, 3, 3}, {3, 3, 3}, {2, 2, 2}};
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
__device__ __forceinline__ void vo_pose_make(const double tr[6], VoPose& M)
{
    double sa, ca, sb, cb, sg, cg;                          // sines / cosines of alpha (about x), beta (about y), gamma (about z)
    vo_sincos64(tr[0], &sa, &ca); vo_sincos64(tr[1], &sb, &cb); vo_sincos64(tr[2], &sg, &cg);
    for (int r = 0; r < 3; r++) M.t[r] = tr[3 + r];
    const double rot[3][3] = {{cb*cg,                -cb*sg,                sb},
                              {sa*sb*cg + ca*sg,     -sa*sb*sg + ca*cg,     -sa*cb},
                              {-ca*sb*cg + sa*sg,    ca*sb*sg + sa*cg,      ca*cb}};
    // d/d alpha: row 0 does not depend on alpha.  Entry [1][1] is written -ca*sb*sg - sa*sg as the reference has it (vo_stereo.cpp:289; the
    // analytic derivative, and libviso2, end in sa*cg): bug-compatible by contract (oracle/vo.c header)
    const double da[3][3] = {{0.0,                   0.0,                   0.0},
                             {ca*sb*cg - sa*sg,      -ca*sb*sg - sa*sg,     -ca*cb},
                             {sa*sb*cg + ca*sg,      -sa*sb*sg + ca*cg,     -sa*cb}};
    const double db[3][3] = {{-sb*cg,                sb*sg,                 cb},
                             {sa*cb*cg,              -sa*cb*sg,             sa*sb},
                             {-ca*cb*cg,             ca*cb*sg,              -ca*sb}};
    const double dg[3][3] = {{-cb*sg,                -cb*cg,                0.0},
                             {-sa*sb*sg + ca*cg,     -sa*sb*cg - ca*sg,     0.0},
                             {ca*sb*sg + sa*cg,      ca*sb*cg - sa*sg,      0.0}};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { M.rot[r][c] = rot[r][c]; M.drot[0][r][c] = da[r][c]; M.drot[1][r][c] = db[r][c]; M.drot[2][r][c] = dg[r][c]; }
}
// one match: 4 predictions, optionally the 4 weighted residuals and the 4 x 6 Jacobian
template <bool WITH_J>
__device__ __forceinline__ void vo_point(const ssm_pmatch& m, const ssm_vo_params& P, const VoPose& M, double* J, double pred[4], double* res)
{
    // the match's 3-D point in the previous left camera, from its disparity (at least 1 px)
    const double disp = fmax((double)(m.u1p - m.u2p), 1.0);
    const double prev[3] = {((double)m.u1p - P.cu) * P.base / disp, ((double)m.v1p - P.cv) * P.base / disp, P.f * P.base / disp};
    double cur[3];                                           // the same point in the current left camera
#pragma unroll
    for (int r = 0; r < 3; r++) cur[r] = M.rot[r][0] * prev[0] + M.rot[r][1] * prev[1] + M.rot[r][2] * prev[2] + M.t[r];
    const double obs[4] = { (double)m.u1c, (double)m.v1c, (double)m.u2c, (double)m.v2c };
    double weight = 1.0;
    if (P.reweighting) weight = 1.0 / (fabs(obs[0] - P.cu) / fabs(P.cu) + 0.05);
    const double xr = cur[0] - P.base;                       // x in the current right camera
    if (WITH_J) {
        const double wf = weight * P.f, zz = cur[2] * cur[2];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double d[3];                                     // d cur / d parameter j
            if (j < 3) {
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const int nt = VO_DTERMS[j][r];
                    double v = 0.0;
                    if (nt >= 2) v = M.drot[j][r][0] * prev[0] + M.drot[j][r][1] * prev[1];
                    if (nt == 3) v = v + M.drot[j][r][2] * prev[2];
                    d[r] = v;
                }
            } else { d[0] = j == 3 ? 1.0 : 0.0; d[1] = j == 4 ? 1.0 : 0.0; d[2] = j == 5 ? 1.0 : 0.0; }
            // quotient rule on u = f x / z (+ cu), v = f y / z (+ cv); the two v rows are the same expression
            J[0 * 6 + j] = wf * (d[0] * cur[2] - cur[0] * d[2]) / zz;
            J[1 * 6 + j] = wf * (d[1] * cur[2] - cur[1] * d[2]) / zz;
            J[2 * 6 + j] = wf * (d[0] * cur[2] - xr * d[2]) / zz;
            J[3 * 6 + j] = J[1 * 6 + j];
        }
    }
    pred[0] = P.f * cur[0] / cur[2] + P.cu; pred[1] = P.f * cur[1] / cur[2] + P.cv;
    pred[2] = P.f * xr / cur[2] + P.cu; pred[3] = pred[1];
    if (WITH_J) {
#pragma unroll
        for (int k = 0; k < 4; k++) res[k] = weight * (obs[k] - pred[k]);
    }
}
__device__ __forceinline__ bool vo_is_inlier(const ssm_pmatch& m, const ssm_vo_params& P, const VoPose& R)
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
__device__ __forceinline__ void vo_accumulate(const ssm_pmatch& m, const ssm_vo_params& P, const VoPose& R, double* acc)
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

// Batched form: blockIdx.y = frame f of a sub-batch; its matches are m_all + f * stride, their number n_all[f] (n_all == nullptr: n_fixed, the
// per-call entry point).  The three sample indices of hypothesis k come either from an explicit table (samples, ssm_vo_estimate) or from the RAW
// rand() stream the way VisualOdometry::getRandomSample (src/vo.cpp:74-93) uses it: three draws r0, r1, r2 -> r0 % N of {0..N-1}, then r1 % (N-1)
// and r2 % (N-2) of what is left (erase() keeps the pool sorted, so "what is left" is the identity with the taken values skipped).  A frame's
// draws start at rand_off[f] of the stream: frames with fewer than 6 matches draw nothing (vo_stereo.cpp:61-63), which vo_offsets_kernel accounts for.
struct VoBatch {
    const ssm_pmatch* m_all; int stride; const int32_t* n_all; int n_fixed;
    const int32_t* samples; const uint32_t* rand_stream; const int32_t* rand_off; int iters;
};
__global__ void vo_offsets_kernel(const int32_t* __restrict__ n_all, int nb, int draws_per_frame, int32_t* __restrict__ consumed /* running draw count */, int32_t* __restrict__ rand_off)
{
    int c = *consumed;
    for (int f = 0; f < nb; f++) { rand_off[f] = c; if (n_all[f] >= 6) c += draws_per_frame; }
    *consumed = c;
}
// Round 4: the hypotheses of a frame are solved by one THREAD each (vo_hyp_kernel: the 3-point Gauss-Newton is a serial f64 chain; with one wave per hypothesis
// 63 lanes waited for lane 0, 12 k instructions per wave for one lane's work), then voted on by one WAVE each (vo_vote_kernel) -- the same arithmetic in the same
// order, 0.59 -> 0.1x ms per 64 frame pairs x 200 hypotheses.
__global__ void __launch_bounds__(64)
vo_hyp_kernel(VoBatch B, ssm_vo_params P, double* __restrict__ tr_all, int32_t* __restrict__ count)
{
    const int k = blockIdx.x * 64 + threadIdx.x, f = blockIdx.y;
    const ssm_pmatch* m = B.m_all + (size_t)f * B.stride;
    const int n = B.n_all ? B.n_all[f] : B.n_fixed;
    if (n < 6 || k >= B.iters) return;                       // no estimate for this frame (vo_refine_kernel writes the empty result)
    tr_all += (size_t)f * B.iters * 6; count += (size_t)f * B.iters;
    int smp[3];
    if (B.samples) { smp[0] = B.samples[3 * k]; smp[1] = B.samples[3 * k + 1]; smp[2] = B.samples[3 * k + 2]; }
    else {
        const uint32_t* r = B.rand_stream + B.rand_off[f] + 3 * k;
        const int v0 = (int)(r[0] % (uint32_t)n);
        int v1 = (int)(r[1] % (uint32_t)(n - 1)); v1 += v1 >= v0;
        const int lo = min(v0, v1), hi = max(v0, v1);
        int v2 = (int)(r[2] % (uint32_t)(n - 2)); v2 += v2 >= lo; v2 += v2 >= hi;
        smp[0] = v0; smp[1] = v1; smp[2] = v2;
    }
    const ssm_pmatch m0 = m[smp[0]], m1 = m[smp[1]], m2 = m[smp[2]];
    double tr[6] = {0, 0, 0, 0, 0, 0};
    int result = VO_UPDATED, iter = 0;
    while (result == VO_UPDATED) {
        VoPose R; vo_pose_make(tr, R);
        double acc[42];
        for (int q = 0; q < 42; q++) acc[q] = 0.0;
        vo_accumulate(m0, P, R, acc); vo_accumulate(m1, P, R, acc); vo_accumulate(m2, P, R, acc);
        result = vo_finish_step(acc, tr, 1e-6);
        if (iter++ > 20 || result == VO_CONVERGED) break;
    }
    for (int q = 0; q < 6; q++) tr_all[6 * k + q] = tr[q];
    count[k] = result == VO_FAILED ? -1 : 0;                 // (the vote replaces the 0)
}
__global__ void __launch_bounds__(256)
vo_vote_kernel(VoBatch B, ssm_vo_params P, const double* __restrict__ tr_all, int32_t* __restrict__ count)
{
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, f = blockIdx.y;
    const ssm_pmatch* m = B.m_all + (size_t)f * B.stride;
    const int n = B.n_all ? B.n_all[f] : B.n_fixed;
    if (n < 6 || k >= B.iters) return;
    tr_all += (size_t)f * B.iters * 6; count += (size_t)f * B.iters;
    if (count[k] < 0) return;                                // the Gauss-Newton failed: no vote (wave-uniform)
    double tr[6];
    for (int q = 0; q < 6; q++) tr[q] = tr_all[6 * k + q];
    VoPose R; vo_pose_make(tr, R);
    int c = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n && vo_is_inlier(m[i], P, R);
        c += __popcll(__ballot(in));
    }
    if (lane == 0) count[k] = c;
}
// blockIdx.x = frame
__global__ void __launch_bounds__(64)
vo_refine_kernel(VoBatch B, ssm_vo_params P, const double* __restrict__ tr_all, const int32_t* __restrict__ count,
                 double* __restrict__ tr_out, int32_t* __restrict__ inliers, int32_t* __restrict__ result /* per frame: [0] = n_inliers, [1] = success */)
{
    const int lane = threadIdx.x, f = blockIdx.x, iters = B.iters;
    const ssm_pmatch* m = B.m_all + (size_t)f * B.stride;
    const int n = B.n_all ? B.n_all[f] : B.n_fixed;
    tr_all += (size_t)f * iters * 6; count += (size_t)f * iters; tr_out += (size_t)f * 6; inliers += (size_t)f * B.stride; result += (size_t)f * 2;
    if (n < 6) {                                             // estimateMotion returns an empty vector (vo_stereo.cpp:61-63); also the frames without a previous frame (n = -1)
        if (lane == 0) { for (int q = 0; q < 6; q++) tr_out[q] = 0.0; result[0] = 0; result[1] = 0; }
        return;
    }
    // the first hypothesis with the largest consensus (the reference replaces only on a strictly larger set)
    int best = 0, bk = -1;
    for (int k = 0; k < iters; k++) { const int c = count[k]; if (c > best) { best = c; bk = k; } }
    double tr[6] = {0, 0, 0, 0, 0, 0};
    if (bk >= 0) for (int q = 0; q < 6; q++) tr[q] = tr_all[6 * bk + q];
    int na = 0;
    if (bk >= 0) {
        VoPose R; vo_pose_make(tr, R);
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
            VoPose R; vo_pose_make(tr, R);
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
// one call (ssm_vo_estimate): explicit samples, n on the host
hipError_t k_vo_estimate(const ssm_pmatch* m, int n, const ssm_vo_params& P, const int32_t* samples, int iters,
                         double* tr_all, int32_t* count, double* tr_out, int32_t* inliers, int32_t* result, hipStream_t s)
{
    VoBatch B; B.m_all = m; B.stride = n; B.n_all = nullptr; B.n_fixed = n; B.samples = samples; B.rand_stream = nullptr; B.rand_off = nullptr; B.iters = iters;
    if (iters > 0) { vo_hyp_kernel<<<dim3((iters + 63) / 64, 1), 64, 0, s>>>(B, P, tr_all, count); vo_vote_kernel<<<dim3((iters + 3) / 4, 1), 256, 0, s>>>(B, P, tr_all, count); }
    vo_refine_kernel<<<1, 64, 0, s>>>(B, P, tr_all, count, tr_out, inliers, result);
    return hipGetLastError();
}
// nb frames of the batched stereo path: matches m_all[f][stride] with n_all[f] entries (device), samples drawn from rand_stream as the host class
// would (consumed: running number of draws taken from the stream, updated); tr_all: nb*iters*6 doubles, count: nb*iters ints, rand_off: nb ints;
// outputs tr_out[f][6], inliers[f][stride], result[f][2]
hipError_t k_vo_estimate_batch(const ssm_pmatch* m_all, int stride, const int32_t* n_all, int nb, const ssm_vo_params& P, const uint32_t* rand_stream, int iters,
                               int32_t* consumed, int32_t* rand_off, double* tr_all, int32_t* count, double* tr_out, int32_t* inliers, int32_t* result, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    VoBatch B; B.m_all = m_all; B.stride = stride; B.n_all = n_all; B.n_fixed = 0; B.samples = nullptr; B.rand_stream = rand_stream; B.rand_off = rand_off; B.iters = iters;
    vo_offsets_kernel<<<1, 1, 0, s>>>(n_all, nb, 3 * iters, consumed, rand_off);
    if (iters > 0) { vo_hyp_kernel<<<dim3((iters + 63) / 64, nb), 64, 0, s>>>(B, P, tr_all, count); vo_vote_kernel<<<dim3((iters + 3) / 4, nb), 256, 0, s>>>(B, P, tr_all, count); }
    vo_refine_kernel<<<nb, 64, 0, s>>>(B, P, tr_all, count, tr_out, inliers, result);
    return hipGetLastError();
}
