// pnp_chain.h -- interface between the tracker's host orchestration (ssm_track.hip) and the device pose chain (kernels_pnp.hip).  Not installed.
#pragma once
#include "ssm_internal.h"
#include "../../include/ssm/pnp_core.h"
#define SSM_TRACK_MAXREF 64
// the Tracker's state while the chain runs on the device (device memory; the host uploads it before a run and reads it back after)
struct PnpState {
    double speed[16], last_pose[16];                 // column-major 4 x 4
    double ref_pose[SSM_TRACK_MAXREF][16];           // refFrames deque, oldest first
    int32_t ref_idx[SSM_TRACK_MAXREF];               // their frame indices relative to the current ssm_seq_process call (negative: frames of the previous call)
    int32_t nref, cnt_lost, stopped_at, pad;
    long long work[4];                               // out: fused passes, chi2 passes, active edges evaluated by the fused / by the chi2 passes of this launch
#ifdef SSM_PNP_PROF
    long long prof[32];                              // shader clocks per section (thread 0), ablation builds only
#endif
};
struct PnpChainArgs {
    const ssm_keypoint* kps; const float* pos3d; const ssm_dmatch* matches; const int32_t* nmatch;     // the call's outputs (device)
    const float* hist_pos3d;                         // R x cap x 3: positions of the deque members that precede the call, row idx + R
    int cap, R, f_begin, f_end, max_lost;
    ssm_pnp::Camera cam;
    PnpState* state; double* pose_out; ssm_track_info* info_out;
    float *img, *obj; uint8_t *inl, *dec;                                                              // scratch for R * cap correspondences
    struct LEdge* ledges; double2* err;              // the edge list when it does not fit in LDS (k_pnp_edge_bytes() each), and every edge's error
    int edges_in_lds;                                // set by k_pnp_chain
    // the cluster form: `blocks` (1, 2, 4 or 8) blocks run the chain together; state, img / obj / inl / dec / ledges / err hold `blocks` slices (slice b is block b's
    // private copy; the host fills every state slice and reads slice 0); xchg: k_pnp_xchg_bytes() of device memory, xfail = the word behind its ring
    int blocks; unsigned long long* xchg; unsigned* xfail;
};
size_t k_pnp_xchg_bytes(void);
hipError_t k_pnp_chain(const PnpChainArgs& a, hipStream_t s);
// one solvePnP on the device (ssm_pnp_solve): n correspondences, T (16 doubles, device) in / out
struct PnpSolveArgs {
    const float *img, *obj; int n; ssm_pnp::Camera cam; double* T; uint8_t *inl, *dec; struct LEdge* ledges; double2* err; int32_t* n_inliers;
    int edges_in_lds;                                // set by k_pnp_solve
    // the cluster form (round 6): `blocks` (1 or 8) blocks solve the list together like a chain's cluster does -- inl / dec / ledges / err hold `blocks` slices of
    // `slice` entries (block b's private copy; slice 0 is the result), xchg / xfail as in PnpChainArgs; T and n_inliers are written by block 0
    int blocks; size_t slice; unsigned long long* xchg; unsigned* xfail;
    unsigned seq_base;                               // first pass number of this launch: the ring is NOT zeroed per launch -- pass tags keep rising from launch to launch (the caller zeroes the ring when the numbers wrap), and xfail is a word the caller uploads as 0
};
hipError_t k_pnp_solve(const PnpSolveArgs& a, hipStream_t s);
size_t k_pnp_edge_bytes(void);
