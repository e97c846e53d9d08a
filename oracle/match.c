/* oracle/match.c -- CPU oracle (TEST INFRASTRUCTURE, see ssm_oracle.h) for K6:
 * OrbFeature::match, /root/reference/src/orb.cpp:16-29:
 *   matcher->knnMatch(desp1, desp2, matches_knn, 2);                       (orb.cpp:21)
 *   if (knn[i][0].distance < knn_match_ratio * knn[i][1].distance) keep    (orb.cpp:25)
 * with matcher = cv::DescriptorMatcher::create("BruteForce-Hamming") (include/orb.h:27) and descriptors stacked in
 * feature order (include/rgbdframe.h:78-86).  cv::BFMatcher (OpenCV 2.4) contract restated: exact Hamming distance
 * to every train row; K smallest kept by insertion with strict '<', so equal distances resolve to the LOWER
 * trainIdx; results ascending by distance; DMatch.distance is the integer distance stored as float; imgIdx 0.
 */
#include "ssm_oracle.h"
#include <stddef.h>
#include <string.h>

static inline int hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32); memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

int sso_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist)
{
    if (nt < 2) return -1;       /* the reference dereferences [1] unguarded (orb.cpp:25) */
    for (int i = 0; i < nq; i++) {
        int d0 = 1 << 30, d1 = 1 << 30, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            int d = hamming256(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
            else if (d < d1) { d1 = d; i1 = j; }
        }
        idx[2*i] = i0; idx[2*i+1] = i1; dist[2*i] = d0; dist[2*i+1] = d1;
    }
    return 0;
}

int sso_match(const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, sso_dmatch* out)
{
    if (nt < 2) return -1;
    int n = 0;
    for (int i = 0; i < nq; i++) {
        int32_t idx[2], dist[2];
        sso_hamming_knn2(q + (size_t)i * 32, 1, t, nt, idx, dist);
        float f0 = (float)dist[0], f1 = (float)dist[1];
        if ((double)f0 < ratio * (double)f1) {          /* float < double*float  ->  double compare (orb.cpp:25, orb.h:62) */
            out[n].queryIdx = i; out[n].trainIdx = idx[0]; out[n].imgIdx = 0; out[n].distance = f0; n++;
        }
    }
    return n;
}
