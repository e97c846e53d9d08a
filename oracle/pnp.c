/* pnp.c -- CPU ORACLE (test infrastructure, see ssm_oracle.h) for the PnP consumer of the RGB-D match table:
 * rgbd_tutor::PnPSolver::solvePnP (/root/reference/src/pnp.cpp:5-118).  SURVEY.md s.8(f) rank 1.
 *
 * PARITY UNPINNED.  The in-tree part (pnp.cpp) is restated statement for statement, INCLUDING the inlier bookkeeping of SURVEY.md Appendix A
 * quirk 14 (bug-compatible, by decision):
 *   - every round restarts the vertex from the caller's initial transform (pnp.cpp:66);
 *   - an edge's error is recomputed only while `inliers[e->id()]` is true, chi2() of an edge that is already an outlier is therefore the STALE value
 *     that made it one, so it is counted out again (`good--`) in every later round (pnp.cpp:73-83);
 *   - an edge that passes marks `inliers[i]` with i = its POSITION in the edge list, not its id (pnp.cpp:87): behind a correspondence without depth the
 *     mark lands on the wrong entry (it can revive an entry that has no edge at all);
 *   - the robust kernels are dropped in the third round (pnp.cpp:92-93); the loop stops when good < 5 (pnp.cpp:97-98);
 *   - success is `inliers.size() > min_inliers`, the LENGTH of the flag vector (pnp.cpp:115).
 * The optimiser is g2o (un-vendored third-party dependency, no version pinned anywhere in the reference tree; the raw-pointer BlockSolver_6_3 /
 * OptimizationAlgorithmLevenberg constructors at pnp.cpp:9-11 are the pre-2017 API).  Its published algorithm is restated:
 *   - vertex: SE3 pose, update T <- exp(d) T with d = (omega, upsilon) (g2o::SE3Quat::exp: Rodrigues, V upsilon; below 1e-5 rad R = I + W + W W);
 *     the pose is kept as a rotation MATRIX here (g2o keeps a quaternion and re-normalises it: a rounding-level difference);
 *   - edge: EdgeSE3ProjectXYZOnlyPose, error = measurement - (fx x/z + cx, fy y/z + cy), its analytic 2 x 6 Jacobian, information = identity;
 *   - RobustKernelHuber with delta = (double)(float)sqrt(5.991) (pnp.cpp:28 stores it in a float): rho = (e2, 1) inside, (2 sqrt(e2) delta - delta^2,
 *     delta / sqrt(e2)) outside; the Hessian uses rho' * information only (g2o leaves the second-order term commented out);
 *   - OptimizationAlgorithmLevenberg::solve: lambda0 = 1e-5 * max diagonal of H at the first iteration of every optimize() call; each iteration
 *     builds (H, b) once, then tries (H + lambda I) x = b up to 10 times: gain = (chi - chi_new) / (sum x_j (lambda x_j + b_j) + 1e-3); gain > 0 accepts
 *     and scales lambda by clamp(1 - (2 gain - 1)^3, 1/3, 2/3); otherwise the step is undone, lambda *= nu, nu *= 2; optimize(10) stops early when an
 *     iteration used up its 10 trials or ended with gain == 0;
 *   - LinearSolverDense: Eigen's LDLT; restated as an un-pivoted L D L^T (fails on a non-positive pivot).
 * NUMERIC CONTRACT shared with the product (include/ssm/pnp_core.h: host class, bulk tracker, device chain), chosen so that a CPU and a 1024-thread
 * GPU block produce the same bits (g2o itself adds the edges one after the other and calls libm: rounding-level differences):
 *   - every sum over the edges (chi2, H, b) is a LANE sum: edge i of the edge list goes to lane i mod 1024, a lane adds its edges in list order, the 64
 *     lanes of a group are added as a neighbour-first binary tree, the 16 groups in group order (lane_reduce below);
 *   - sin / cos = sso_sincos64 (the stereo VO's polynomial routine, vo.c); (2 gain - 1)^3 = t * t * t;
 *   - only the lower triangle of H is formed (the L D L^T solve reads nothing else).
 * The host class include/ssm/pnp.h implements the same algorithm; tests/test_pnp.py compares the two; tests/golden/pyref.py is a second restatement.
 */
#include "ssm_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double R[9], t[3]; } pose_t;          /* row-major rotation, x_cam = R X + t */
typedef struct { int id, level, robust; double X[3], u, v, err[2]; } edge_t;

static void pose_from_colmajor(const double T[16], pose_t* P)
{
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) P->R[3 * r + c] = T[4 * c + r]; P->t[r] = T[12 + r]; }
}
static void pose_to_colmajor(const pose_t* P, double T[16])
{
    memset(T, 0, 16 * sizeof(double)); T[15] = 1.0;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[4 * c + r] = P->R[3 * r + c]; T[12 + r] = P->t[r]; }
}
static void mat3_mul(const double* A, const double* B, double* C)
{
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
/* T <- exp(d) * T, d = (omega, upsilon) */
static void pose_oplus(pose_t* P, const double d[6])
{
    const double w0 = d[0], w1 = d[1], w2 = d[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double W[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double W2[9]; mat3_mul(W, W, W2);
    double dR[9], V[9];
    if (theta < 0.00001) {
        for (int k = 0; k < 9; k++) dR[k] = (k % 4 == 0 ? 1.0 : 0.0) + W[k] + W2[k];
        memcpy(V, dR, sizeof(V));
    } else {
        double sn, cs; sso_sincos64(theta, &sn, &cs);
        const double a = sn / theta, b = (1 - cs) / (theta * theta), c = (theta - sn) / (theta * theta * theta);
        for (int k = 0; k < 9; k++) { dR[k] = (k % 4 == 0 ? 1.0 : 0.0) + a * W[k] + b * W2[k]; V[k] = (k % 4 == 0 ? 1.0 : 0.0) + b * W[k] + c * W2[k]; }
    }
    double nR[9], nt[3];
    mat3_mul(dR, P->R, nR);
    for (int r = 0; r < 3; r++) {
        const double vt = V[3 * r] * d[3] + V[3 * r + 1] * d[4] + V[3 * r + 2] * d[5];
        nt[r] = dR[3 * r] * P->t[0] + dR[3 * r + 1] * P->t[1] + dR[3 * r + 2] * P->t[2] + vt;
    }
    memcpy(P->R, nR, sizeof(nR)); memcpy(P->t, nt, sizeof(nt));
}
static void edge_map(const edge_t* e, const pose_t* P, double p[3])
{
    for (int r = 0; r < 3; r++) p[r] = P->R[3 * r] * e->X[0] + P->R[3 * r + 1] * e->X[1] + P->R[3 * r + 2] * e->X[2] + P->t[r];
}
static void edge_error(edge_t* e, const pose_t* P, const sso_camera* k)
{
    double p[3]; edge_map(e, P, p);
    e->err[0] = e->u - (p[0] / p[2] * k->fx + k->cx);
    e->err[1] = e->v - (p[1] / p[2] * k->fy + k->cy);
}
static double edge_chi2(const edge_t* e) { return e->err[0] * e->err[0] + e->err[1] * e->err[1]; }
static void huber(double e2, double delta, double rho[2])
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.0; }
    else { const double s = sqrt(e2); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; }
}
/* the lane sum of the contract: v[lane][nval] -> out[nval]; lanes >= the number of edges hold 0 */
#define PNP_LANES 1024
#define PNP_GROUP 64
static void lane_reduce(double* v, int nval, double* out)
{
    double grp[(PNP_LANES / PNP_GROUP) * 32];
    for (int g = 0; g < PNP_LANES / PNP_GROUP; g++) {
        double* base = v + (size_t)g * PNP_GROUP * nval;
        for (int s = 1; s < PNP_GROUP; s <<= 1)
            for (int l = 0; l < PNP_GROUP; l += 2 * s)
                for (int q = 0; q < nval; q++) base[l * nval + q] = base[l * nval + q] + base[(l + s) * nval + q];
        for (int q = 0; q < nval; q++) grp[g * nval + q] = base[q];
    }
    for (int q = 0; q < nval; q++) { double s = grp[q]; for (int g = 1; g < PNP_LANES / PNP_GROUP; g++) s = s + grp[g * nval + q]; out[q] = s; }
}
static double active_chi2(edge_t* E, int ne, const pose_t* P, const sso_camera* k, double delta)
{
    static _Thread_local double lane[PNP_LANES];
    memset(lane, 0, sizeof(lane));
    for (int i = 0; i < ne; i++) {
        if (E[i].level != 0) continue;
        edge_error(&E[i], P, k);
        const double e2 = edge_chi2(&E[i]);
        if (E[i].robust) { double rho[2]; huber(e2, delta, rho); lane[i % PNP_LANES] += rho[0]; } else lane[i % PNP_LANES] += e2;
    }
    double chi; lane_reduce(lane, 1, &chi);
    return chi;
}
/* H (6 x 6 row-major, lower triangle filled) and b from the active edges at P (their err[] is current) */
static void build_system(const edge_t* E, int ne, const pose_t* P, const sso_camera* k, double delta, double H[36], double b[6])
{
    static _Thread_local double lane[PNP_LANES * 27];
    memset(lane, 0, sizeof(lane));
    for (int i = 0; i < ne; i++) {
        const edge_t* e = &E[i];
        if (e->level != 0) continue;
        double* acc = lane + (size_t)(i % PNP_LANES) * 27;
        double p[3]; edge_map(e, P, p);
        const double x = p[0], y = p[1], iz = 1.0 / p[2], iz2 = iz * iz;
        const double J[2][6] = {
            { x * y * iz2 * k->fx, -(1 + (x * x * iz2)) * k->fx, y * iz * k->fx, -iz * k->fx, 0, x * iz2 * k->fx },
            { (1 + y * y * iz2) * k->fy, -x * y * iz2 * k->fy, -x * iz * k->fy, 0, -iz * k->fy, y * iz2 * k->fy } };
        double w = 1.0;
        if (e->robust) { double rho[2]; huber(edge_chi2(e), delta, rho); w = rho[1]; }
        for (int r = 0; r < 2; r++) {
            const double wr = -e->err[r] * w;                       /* omega_r = -information * error, robustified */
            int q = 0;
            for (int a = 0; a < 6; a++) { acc[21 + a] += J[r][a] * wr; for (int c = 0; c <= a; c++) acc[q++] += J[r][a] * w * J[r][c]; }
        }
    }
    double tot[27]; lane_reduce(lane, 27, tot);
    memset(H, 0, 36 * sizeof(double));
    { int q = 0; for (int a = 0; a < 6; a++) for (int c = 0; c <= a; c++) H[6 * a + c] = tot[q++]; }
    for (int a = 0; a < 6; a++) b[a] = tot[21 + a];
}
static int solve_ldlt(const double Hin[36], double lambda, const double b[6], double x[6])
{
    double L[36], D[6], A[36];
    memcpy(A, Hin, sizeof(A));
    for (int i = 0; i < 6; i++) A[7 * i] += lambda;
    memset(L, 0, sizeof(L));
    for (int j = 0; j < 6; j++) {
        double d = A[6 * j + j];
        for (int k = 0; k < j; k++) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        if (!(d > 0)) return 0;
        D[j] = d; L[6 * j + j] = 1.0;
        for (int i = j + 1; i < 6; i++) {
            double s = A[6 * i + j];
            for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[6 * i + k] * y[k]; y[i] = s; }
    for (int i = 0; i < 6; i++) y[i] /= D[i];
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k]; x[i] = s; }
    return 1;
}
/* SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg on the level-0 edges */
static void lm_optimize(edge_t* E, int ne, pose_t* P, const sso_camera* k, double delta, int iterations)
{
    int nactive = 0;
    for (int i = 0; i < ne; i++) nactive += E[i].level == 0;
    if (nactive == 0) return;                                       /* initializeOptimization finds nothing to optimise */
    double lambda = 0, nu = 2;
    for (int it = 0; it < iterations; it++) {
        double chi = active_chi2(E, ne, P, k, delta), chi_new = chi;
        double H[36], b[6];
        build_system(E, ne, P, k, delta, H, b);
        if (it == 0) { double mx = 0; for (int j = 0; j < 6; j++) if (fabs(H[7 * j]) > mx) mx = fabs(H[7 * j]); lambda = 1e-5 * mx; nu = 2; }
        double gain = 0; int trials = 0;
        do {
            const pose_t saved = *P;
            double x[6] = {0, 0, 0, 0, 0, 0};
            const int ok = solve_ldlt(H, lambda, b, x);
            pose_oplus(P, x);
            chi_new = active_chi2(E, ne, P, k, delta);
            if (!ok) chi_new = DBL_MAX;
            gain = chi - chi_new;
            double scale = 0;
            for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            gain /= scale;
            if (gain > 0 && isfinite(chi_new)) {
                const double t3 = 2 * gain - 1;
                double alpha = 1. - t3 * t3 * t3;
                alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                const double f = alpha > 1. / 3. ? alpha : 1. / 3.;
                lambda *= f; nu = 2; chi = chi_new;
            } else {
                lambda *= nu; nu *= 2; *P = saved;
                if (!isfinite(lambda)) break;
            }
            trials++;
        } while (gain < 0 && trials < 10);
        if (trials == 10 || gain == 0) break;                       /* Terminate */
    }
    active_chi2(E, ne, P, k, delta);                                /* the active edges carry the error at the final estimate */
}

/* img: n x (u, v); obj: n x (X, Y, Z) (a (0,0,0) row = no depth); T: column-major 4 x 4, initial value in / estimate out.
 * inliers_out (n ints) / *n_inliers: the indices whose flag is set at the end.  Returns the reference's success value. */
int sso_pnp_solve(const float* img, const float* obj, int n, const sso_camera* cam, int min_inliers, double T[16], int* inliers_out, int* n_inliers)
{
    const double delta = (double)(float)sqrt(5.991);
    edge_t* E = (edge_t*)calloc((size_t)(n > 0 ? n : 1), sizeof(edge_t));
    unsigned char* inl = (unsigned char*)malloc((size_t)(n > 0 ? n : 1));
    memset(inl, 1, (size_t)(n > 0 ? n : 1));
    int ne = 0, good = 0;
    for (int i = 0; i < n; i++) {
        if (obj[3 * i] == 0.f && obj[3 * i + 1] == 0.f && obj[3 * i + 2] == 0.f) { inl[i] = 0; continue; }
        good++;
        edge_t* e = &E[ne++];
        e->id = i; e->level = 0; e->robust = 1; e->u = img[2 * i]; e->v = img[2 * i + 1];
        e->X[0] = obj[3 * i]; e->X[1] = obj[3 * i + 1]; e->X[2] = obj[3 * i + 2];
    }
    pose_t init, P; pose_from_colmajor(T, &init); P = init;
    for (int it = 0; it < 4; it++) {
        P = init;
        lm_optimize(E, ne, &P, cam, delta, 10);
        for (int i = 0; i < ne; i++) {
            edge_t* e = &E[i];
            if (inl[e->id]) edge_error(e, &P, cam);
            if (edge_chi2(e) > 5.991) { inl[e->id] = 0; e->level = 1; good--; }
            else { inl[i] = 1; e->level = 0; }                      /* position i, not e->id: the reference's index mix-up */
            if (it == 2) e->robust = 0;
        }
        if (good < 5) break;
    }
    int m = 0;
    for (int i = 0; i < n; i++) if (inl[i]) inliers_out[m++] = i;
    *n_inliers = m;
    pose_to_colmajor(&P, T);
    free(E); free(inl);
    return n > min_inliers;
}
