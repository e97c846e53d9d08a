// ssm/pnp_core.h -- the arithmetic of rgbd_tutor::PnPSolver::solvePnP (reference src/pnp.cpp:5-118) as plain functions on plain structs, shared by
//   * the host class rgbd_tutor::PnPSolver (include/ssm/pnp.h: the per-frame Tracker),
//   * the bulk tracker inside libssm_hip.so (ssm_tracker_run: host path and the one-block device chain, csrc/kernels_pnp.hip),
// so that a pose solved per frame on the host and the same pose solved in bulk are the same bits.  g2o (un-vendored, absent) is restated as described in
// oracle/pnp.c; what this header adds is the NUMERIC CONTRACT that makes the solve identical on the CPU and on the GPU (both built -ffp-contract=off):
//   * sums over the edges (chi2; the 21 lower-triangle entries of H and the 6 of b) are LANE sums: edge i of the edge list belongs to lane i mod 1024,
//     a lane adds its edges in list order (an edge that is not active adds nothing), the 64 lanes of a group are added as a neighbour-first binary tree
//     (lane l + lane l^1, then ^2, ^4 ... ^32: the xor butterfly of a wavefront), the 16 group sums are added in group order.  On the device a lane is a
//     thread of the 1024-thread block; the host walks the same tree (lane_sum below).  g2o adds the edges one after the other: a rounding-level difference;
//   * sin / cos are ONE polynomial routine (Cody-Waite reduction by pi/2 + the fdlibm kernels: the stereo VO's contract, oracle/vo.c), within 1.1e-16 of libm;
//   * Levenberg's (2 gain - 1)^3 is t * t * t.
// Everything else is +, -, *, /, sqrt in IEEE double in the written order.  pnp.cpp's inlier bookkeeping is kept as written.
#pragma once
#include <math.h>
#include <float.h>
#include <stdint.h>
#ifndef SSM_HD
#  if defined(__HIPCC__)
#    define SSM_HD __host__ __device__ inline
#  else
#    define SSM_HD inline
#  endif
#endif
namespace ssm_pnp {
enum { LANES = 1024, GROUP = 64, NGROUP = LANES / GROUP, NACC = 27 };      // NACC: H lower triangle (21, row-major: 00 10 11 20 21 22 ...) + b (6)
struct Pose { double R[9], t[3]; };                                        // x_cam = R X + t, R row-major
struct Camera { double fx, fy, cx, cy; };
struct Edge { int32_t id, level, robust, pad; double X[3], u, v, e0, e1; };   // level 0 = in the optimisation; (e0, e1) = error at the last evaluation

SSM_HD void sincos64(double x, double& s, double& c)
{
    const double fn = rint(x * 6.36619772367581382433e-01);
    double r = x - fn * 1.57079632673412561417e+00; r = r - fn * 6.07710050650619224932e-11; r = r - fn * 2.02226624879595063154e-21;
    const double z = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double sr = r + (r * z) * (-1.66666666666666324348e-01 + z * ps);
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    switch ((int)((long long)fn & 3)) { case 0: s = sr; c = cr; break; case 1: s = cr; c = -sr; break; case 2: s = -sr; c = -cr; break; default: s = -cr; c = sr; }
}
SSM_HD void mat3_mul(const double* A, const double* B, double* C)
{
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
// T <- exp(d) T, d = (omega, upsilon): g2o::SE3Quat::exp (Rodrigues; below 1e-5 rad R = I + W + W W)
SSM_HD void pose_oplus(Pose& P, const double* d)
{
    const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double W[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
    double W2[9], dR[9], V[9]; mat3_mul(W, W, W2);
    if (th < 0.00001) { for (int k = 0; k < 9; k++) { dR[k] = (k % 4 == 0 ? 1.0 : 0.0) + W[k] + W2[k]; V[k] = dR[k]; } }
    else {
        double sn, cs; sincos64(th, sn, cs);
        const double a = sn / th, b = (1 - cs) / (th * th), c = (th - sn) / (th * th * th);
        for (int k = 0; k < 9; k++) { dR[k] = (k % 4 == 0 ? 1.0 : 0.0) + a * W[k] + b * W2[k]; V[k] = (k % 4 == 0 ? 1.0 : 0.0) + b * W[k] + c * W2[k]; }
    }
    double nR[9], nt[3]; mat3_mul(dR, P.R, nR);
    for (int r = 0; r < 3; r++) { const double vt = V[3 * r] * d[3] + V[3 * r + 1] * d[4] + V[3 * r + 2] * d[5]; nt[r] = dR[3 * r] * P.t[0] + dR[3 * r + 1] * P.t[1] + dR[3 * r + 2] * P.t[2] + vt; }
    for (int k = 0; k < 9; k++) P.R[k] = nR[k];
    for (int k = 0; k < 3; k++) P.t[k] = nt[k];
}
SSM_HD void edge_map(const Edge& e, const Pose& P, double* p) { for (int r = 0; r < 3; r++) p[r] = P.R[3 * r] * e.X[0] + P.R[3 * r + 1] * e.X[1] + P.R[3 * r + 2] * e.X[2] + P.t[r]; }
SSM_HD void edge_error(Edge& e, const Pose& P, const Camera& k)
{
    double p[3]; edge_map(e, P, p);
    e.e0 = e.u - (p[0] / p[2] * k.fx + k.cx); e.e1 = e.v - (p[1] / p[2] * k.fy + k.cy);
}
SSM_HD double edge_chi2(const Edge& e) { return e.e0 * e.e0 + e.e1 * e.e1; }
SSM_HD void huber(double e2, double delta, double& rho0, double& rho1)
{
    const double d2 = delta * delta;
    if (e2 <= d2) { rho0 = e2; rho1 = 1.0; } else { const double s = sqrt(e2); rho0 = 2 * s * delta - d2; rho1 = delta / s; }
}
// the edge's term of the (robustified) chi2 at P; leaves the error in the edge
SSM_HD double edge_rho(Edge& e, const Pose& P, const Camera& k, double delta)
{
    edge_error(e, P, k);
    const double e2 = edge_chi2(e);
    if (!e.robust) return e2;
    double r0, r1; huber(e2, delta, r0, r1); return r0;
}
// acc[0..20] += the edge's J^T w J (lower triangle), acc[21..26] += its J^T (-w e); the edge's error is current
SSM_HD void edge_accumulate(const Edge& e, const Pose& P, const Camera& k, double delta, double* acc)
{
    double p[3]; edge_map(e, P, p);
    const double x = p[0], y = p[1], iz = 1.0 / p[2], iz2 = iz * iz;
    const double J[2][6] = {{x * y * iz2 * k.fx, -(1 + (x * x * iz2)) * k.fx, y * iz * k.fx, -iz * k.fx, 0, x * iz2 * k.fx},
                            {(1 + y * y * iz2) * k.fy, -x * y * iz2 * k.fy, -x * iz * k.fy, 0, -iz * k.fy, y * iz2 * k.fy}};
    double w = 1.0;
    if (e.robust) { double r0; huber(edge_chi2(e), delta, r0, w); }
    const double er[2] = {e.e0, e.e1};
    // J[0][4] and J[1][3] are the literal zero: their products are +-0 and adding +-0 to a sum that started at +0 never changes it, so those 14 of the 54
    // accumulations are skipped (the compiler may not: x * 0 is not 0 for every x); everything else in the order written
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 2; r++) {
        const double wr = -er[r] * w;
        const int z = r == 0 ? 4 : 3;
        int q = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int a = 0; a < 6; a++) {
            if (a == z) { q += a + 1; continue; }
            acc[21 + a] += J[r][a] * wr;
            const double jw = J[r][a] * w;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int c = 0; c <= a; c++) { if (c != z) acc[q] += jw * J[r][c]; q++; }
        }
    }
}
// (H + lambda I) x = b with H given by its lower triangle, un-pivoted L D L^T (g2o: Eigen LDLT); false on a non-positive pivot
SSM_HD bool solve_ldlt(const double* Hl, double lambda, const double* b, double* x)
{
    double A[36], L[36], D[6], y[6];
    { int q = 0; for (int a = 0; a < 6; a++) for (int c = 0; c <= a; c++) { A[6 * a + c] = Hl[q]; A[6 * c + a] = Hl[q]; q++; } }
    for (int i = 0; i < 36; i++) L[i] = 0;
    for (int i = 0; i < 6; i++) A[7 * i] += lambda;
    for (int j = 0; j < 6; j++) {
        double d = A[6 * j + j]; for (int k = 0; k < j; k++) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        if (!(d > 0)) return false;
        D[j] = d; L[6 * j + j] = 1.0;
        for (int i = j + 1; i < 6; i++) { double s = A[6 * i + j]; for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k] * D[k]; L[6 * i + j] = s / d; }
    }
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[6 * i + k] * y[k]; y[i] = s; }
    for (int i = 0; i < 6; i++) y[i] /= D[i];
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k]; x[i] = s; }
    return true;
}
// the Levenberg bookkeeping of one trial (g2o OptimizationAlgorithmLevenberg::solve); returns true when the step is accepted
struct LmState { double lambda, nu; };
SSM_HD bool lm_update(LmState& s, double chi, double chi_new, bool solved, const double* x, const double* b, double& gain)
{
    if (!solved) chi_new = DBL_MAX;
    gain = chi - chi_new;
    double scale = 0; for (int j = 0; j < 6; j++) scale += x[j] * (s.lambda * x[j] + b[j]);
    scale += 1e-3; gain /= scale;
    if (gain > 0 && isfinite(chi_new)) {
        const double t = 2 * gain - 1;
        double alpha = 1. - t * t * t; alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
        s.lambda *= alpha > 1. / 3. ? alpha : 1. / 3.; s.nu = 2;
        return true;
    }
    s.lambda *= s.nu; s.nu *= 2;
    return false;
}
// rigid transforms as 4 x 4 column-major matrices, the operations of Eigen::Isometry3d as include/ssm/compat.h writes them
SSM_HD void iso_mul(const double* A, const double* B, double* C)
{
    double r[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[k * 4 + i] * B[j * 4 + k]; r[j * 4 + i] = s; }
    for (int k = 0; k < 16; k++) C[k] = r[k];
}
SSM_HD void iso_inverse(const double* M, double* out)
{
    double r[16];
    for (int k = 0; k < 16; k++) r[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[j * 4 + i] = M[i * 4 + j];
    for (int i = 0; i < 3; i++) r[12 + i] = -(r[0 * 4 + i] * M[12] + r[1 * 4 + i] * M[13] + r[2 * 4 + i] * M[14]);
    for (int k = 0; k < 16; k++) out[k] = r[k];
}
// (x, y, z, 1) through M: the first three components, in the operation order of Isometry3d * Vector4d
SSM_HD void iso_apply(const double* M, double x, double y, double z, double* out)
{
    const double p[4] = {x, y, z, 1.0};
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 4; k++) s += M[k * 4 + i] * p[k]; out[i] = s; }
}
SSM_HD void pose_from_iso(const double* T, Pose& P) { for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) P.R[3 * r + c] = T[c * 4 + r]; P.t[r] = T[12 + r]; } }
SSM_HD void pose_to_iso(const Pose& P, double* T)
{
    for (int k = 0; k < 16; k++) T[k] = (k == 15) ? 1.0 : 0.0;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[c * 4 + r] = P.R[3 * r + c]; T[12 + r] = P.t[r]; }
}

// ---- host side: the lane tree walked by one thread, and the whole solve
// sum of `nval` quantities over the edges: term(i, out) writes edge i's contribution (or returns false: no contribution)
template <int NVAL, class Term> inline void lane_sum(int ne, Term term, double* result)
{
    static thread_local double lane[LANES * NVAL];
    const int used = ne < LANES ? ne : LANES;
    for (int k = 0; k < used * NVAL; k++) lane[k] = 0.0;
    for (int i = 0; i < ne; i++) term(i, lane + (size_t)(i % LANES) * NVAL);
    double grp[NGROUP * NVAL];
    const int ngrp_used = (used + GROUP - 1) / GROUP;
    for (int g = 0; g < NGROUP; g++) {
        double* base = lane + (size_t)g * GROUP * NVAL;
        if (g >= ngrp_used) { for (int v = 0; v < NVAL; v++) grp[g * NVAL + v] = 0.0; continue; }
        const int live = used - g * GROUP < GROUP ? used - g * GROUP : GROUP;
        for (int k = live * NVAL; k < GROUP * NVAL; k++) base[k] = 0.0;         // lanes of the group without an edge hold 0
        for (int s = 1; s < GROUP; s <<= 1)
            for (int l = 0; l < GROUP; l += 2 * s)
                for (int v = 0; v < NVAL; v++) base[l * NVAL + v] = base[l * NVAL + v] + base[(l + s) * NVAL + v];
        for (int v = 0; v < NVAL; v++) grp[g * NVAL + v] = base[v];
    }
    for (int v = 0; v < NVAL; v++) { double s = grp[v]; for (int g = 1; g < NGROUP; g++) s = s + grp[g * NVAL + v]; result[v] = s; }
}
inline double active_chi2(Edge* E, int ne, const Pose& P, const Camera& k, double delta)
{
    double chi;
    lane_sum<1>(ne, [&](int i, double* acc) { if (E[i].level == 0) acc[0] += edge_rho(E[i], P, k, delta); }, &chi);
    return chi;
}
inline void build_system(const Edge* E, int ne, const Pose& P, const Camera& k, double delta, double* Hl, double* b)
{
    double acc[NACC];
    lane_sum<NACC>(ne, [&](int i, double* a) { if (E[i].level == 0) edge_accumulate(E[i], P, k, delta, a); }, acc);
    for (int q = 0; q < 21; q++) Hl[q] = acc[q];
    for (int q = 0; q < 6; q++) b[q] = acc[21 + q];
}
// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg on the level-0 edges
inline void lm_optimize(Edge* E, int ne, Pose& P, const Camera& k, double delta, int iterations)
{
    bool any = false; for (int i = 0; i < ne; i++) any = any || E[i].level == 0;
    if (!any) return;                                                       // initializeOptimization finds nothing to optimise
    LmState st; st.lambda = 0; st.nu = 2;
    for (int it = 0; it < iterations; it++) {
        double chi = active_chi2(E, ne, P, k, delta), Hl[21], b[6];
        build_system(E, ne, P, k, delta, Hl, b);
        if (it == 0) { double mx = 0; for (int j = 0; j < 6; j++) { const double dg = fabs(Hl[j * (j + 1) / 2 + j]); if (dg > mx) mx = dg; } st.lambda = 1e-5 * mx; st.nu = 2; }
        double gain = 0; int trials = 0;
        do {
            const Pose saved = P;
            double x[6] = {0, 0, 0, 0, 0, 0};
            const bool ok = solve_ldlt(Hl, st.lambda, b, x);
            pose_oplus(P, x);
            const double chi_new = active_chi2(E, ne, P, k, delta);
            if (lm_update(st, chi, chi_new, ok, x, b, gain)) chi = chi_new;
            else { P = saved; if (!isfinite(st.lambda)) break; }
            trials++;
        } while (gain < 0 && trials < 10);
        if (trials == 10 || gain == 0) break;                               // Terminate
    }
    active_chi2(E, ne, P, k, delta);                                        // the active edges carry the error at the final estimate
}
// img: n x (u, v); obj: n x (X, Y, Z), a (0, 0, 0) row = no depth; T: column-major 4 x 4, initial value in / estimate out; inl: n flags out;
// edges: scratch for n entries.  Returns the number of set flags; *success = the reference's return value (pnp.cpp:115: the vector's LENGTH)
inline int solve(const float* img, const float* obj, int n, const Camera& cam, int min_inliers, double* T, unsigned char* inl, Edge* E, int* success)
{
    const double delta = (double)(float)sqrt(5.991);
    int ne = 0, good = 0;
    for (int i = 0; i < n; i++) {
        inl[i] = 1;
        if (obj[3 * i] == 0.f && obj[3 * i + 1] == 0.f && obj[3 * i + 2] == 0.f) { inl[i] = 0; continue; }
        good++;
        Edge& e = E[ne++];
        e.id = i; e.level = 0; e.robust = 1; e.pad = 0; e.X[0] = obj[3 * i]; e.X[1] = obj[3 * i + 1]; e.X[2] = obj[3 * i + 2]; e.u = img[2 * i]; e.v = img[2 * i + 1]; e.e0 = e.e1 = 0;
    }
    Pose init, P; pose_from_iso(T, init); P = init;
    for (int it = 0; it < 4; it++) {
        P = init;                                                           // pnp.cpp:66: every round starts from the caller's transform
        lm_optimize(E, ne, P, cam, delta, 10);
        for (int i = 0; i < ne; i++) {
            Edge& e = E[i];
            if (inl[e.id]) edge_error(e, P, cam);
            if (edge_chi2(e) > 5.991) { inl[e.id] = 0; e.level = 1; good--; }
            else { inl[i] = 1; e.level = 0; }                               // [i], not [e.id]: as written at pnp.cpp:87
            if (it == 2) e.robust = 0;
        }
        if (good < 5) break;
    }
    pose_to_iso(P, T);
    int m = 0; for (int i = 0; i < n; i++) m += inl[i] != 0;
    if (success) *success = n > min_inliers;
    return m;
}
}  // namespace ssm_pnp
