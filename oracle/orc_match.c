/* CPU ORACLE (test infrastructure) -- descriptor matching.
 * Restates corbslam_client/src/ORBmatcher.cc and Frame::ComputeStereoMatches (Frame.cc:470-644).
 * See orc.h for scope.  Compile with -ffp-contract=off. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define TH_HIGH 100          /* ORBmatcher.cc:37 */
#define TH_LOW 50            /* ORBmatcher.cc:38 */
#define HISTO_LENGTH 30      /* ORBmatcher.cc:39 */

/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:1792-1808): 8 x (u32 xor, SWAR popcount) */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb; memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* pyramid accessors implemented in orc_orb.c */
extern int orc_orb_level_dims(const OrcExtractor* ex, int level, int* w, int* h);
extern const uint8_t* orc_orb_level_data(const OrcExtractor* ex, int level);

typedef struct { int dist, idx; } DistIdx;
static int cmp_distidx(const void* a, const void* b)
{
    const DistIdx* x = (const DistIdx*)a, * y = (const DistIdx*)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* Frame::ComputeStereoMatches (Frame.cc:470-644) */
int orc_stereo_match(const OrcExtractor* left, const OrcExtractor* right,
                     const OrcKeyPoint* kl, const uint8_t* dl, int N,
                     const OrcKeyPoint* kr, const uint8_t* dr, int Nr,
                     const OrcStereoParams* p, float* u_right, float* depth)
{
    for (int i = 0; i < N; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    int w0, nRows; orc_orb_level_dims(left, 0, &w0, &nRows);
    /* row table (:481-497) */
    int* rcount = (int*)calloc(nRows + 1, sizeof(int));
    for (int pass = 0; pass < 1; pass++) {
        for (int iR = 0; iR < Nr; iR++) {
            const float kpY = kr[iR].y;
            const float r = 2.0f * p->scale[kr[iR].octave];
            const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
            for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) rcount[yi]++;   /* reference indexes unchecked */
        }
    }
    int* roff = (int*)malloc(sizeof(int) * (nRows + 1));
    roff[0] = 0; for (int y = 0; y < nRows; y++) roff[y + 1] = roff[y] + rcount[y];
    int* rows = (int*)malloc(sizeof(int) * (roff[nRows] > 0 ? roff[nRows] : 1));
    memset(rcount, 0, sizeof(int) * (nRows + 1));
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * p->scale[kr[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) rows[roff[yi] + rcount[yi]++] = iR;
    }
    const float minZ = p->mb, minD = 0;
    const float maxD = p->bf / minZ;
    DistIdx* vDistIdx = (DistIdx*)malloc(sizeof(DistIdx) * (N > 0 ? N : 1)); int nDist = 0;
    for (int iL = 0; iL < N; iL++) {
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        int row = (int)vL;
        if (row < 0 || row >= nRows) continue;
        const int* cand = rows + roff[row]; int ncand = roff[row + 1] - roff[row];
        if (ncand == 0) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH; int bestIdxR = 0;
        const uint8_t* dL = dl + (size_t)iL * 32;
        for (int iC = 0; iC < ncand; iC++) {
            const int iR = cand[iC];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(dL, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {                                          /* :556 */
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = p->inv_scale[levelL];
            const float scaleduL = roundf(kl[iL].x * scaleFactor);
            const float scaledvL = roundf(kl[iL].y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5;
            int lw, lh; orc_orb_level_dims(left, levelL, &lw, &lh);
            const uint8_t* imL = orc_orb_level_data(left, levelL);
            int rw, rh; orc_orb_level_dims(right, levelL, &rw, &rh);
            const uint8_t* imR = orc_orb_level_data(right, levelL);
            int y0 = (int)(scaledvL - w), x0 = (int)(scaleduL - w);
            int bestDistS = INT_MAX; int bestincR = 0;
            const int L = 5;
            float vDists[11];
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= rw) continue;
            /* defined guard: the reference would index outside the image here (cv::Mat::colRange throws) */
            if ((int)(scaleduR0 - L - w) < 0 || x0 < 0 || y0 < 0 || y0 + 10 >= lh || x0 + 10 >= lw) continue;
            float IL[11][11];
            { float c = (float)imL[(size_t)(y0 + w) * lw + x0 + w];
              for (int yy = 0; yy < 11; yy++) for (int xx = 0; xx < 11; xx++) IL[yy][xx] = (float)imL[(size_t)(y0 + yy) * lw + x0 + xx] - c; }
            for (int incR = -L; incR <= +L; incR++) {
                int xr0 = (int)(scaleduR0 + incR - w);
                float c = (float)imR[(size_t)(y0 + w) * rw + xr0 + w];
                double s = 0;                                   /* cv::norm(NORM_L1) on CV_32F accumulates in double */
                for (int yy = 0; yy < 11; yy++) for (int xx = 0; xx < 11; xx++) {
                    float ir = (float)imR[(size_t)(y0 + yy) * rw + xr0 + xx] - c;
                    s += fabs((double)(IL[yy][xx] - ir));
                }
                float dist = (float)s;
                if (dist < (float)bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = p->scale[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)(uL - 0.01); }
                depth[iL] = p->bf / disparity;
                u_right[iL] = bestuR;
                vDistIdx[nDist].dist = bestDistS; vDistIdx[nDist].idx = iL; nDist++;
            }
        }
    }
    int nvalid = nDist;
    if (nDist > 0) {                                             /* reference reads vDistIdx[0] of an empty vector here */
        qsort(vDistIdx, nDist, sizeof(DistIdx), cmp_distidx);
        const float median = (float)vDistIdx[nDist / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nDist - 1; i >= 0; i--) {
            if ((float)vDistIdx[i].dist < thDist) break;
            u_right[vDistIdx[i].idx] = -1; depth[vDistIdx[i].idx] = -1; nvalid--;
        }
    }
    free(rcount); free(roff); free(rows); free(vDistIdx);
    return nvalid;
}

/* ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1746-1787) on bin counts */
static void three_maxima(const int* hist, int L, int* ind1, int* ind2, int* ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* SearchByBoW (ORBmatcher.cc:162-291 / 294-423 variant 0 ; 657-790 variant 1) */
int orc_search_by_bow(int variant,
                      const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const OrcFeatVec* fv1,
                      const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const OrcFeatVec* fv2,
                      float nnratio, int check_ori, int32_t* match_out)
{
    /* variant 0: match_out has n2 entries (index of KF(1) feature per Frame(2) feature)
       variant 1: match_out has n1 entries (idx2 per idx1) */
    int nout = variant == 0 ? n2 : n1;
    for (int i = 0; i < nout; i++) match_out[i] = -1;
    uint8_t* matched2 = (uint8_t*)calloc(n2 > 0 ? n2 : 1, 1);
    int* hist_bin = (int*)malloc(sizeof(int) * (nout > 0 ? nout : 1));   /* bin per output slot, -1 none */
    for (int i = 0; i < nout; i++) hist_bin[i] = -1;
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    int nmatches = 0;
    int a = 0, b = 0;
    while (a < fv1->n_nodes && b < fv2->n_nodes) {
        if (fv1->node_id[a] == fv2->node_id[b]) {
            for (int i1 = fv1->offset[a]; i1 < fv1->offset[a + 1]; i1++) {
                const int idx1 = (int)fv1->idx[i1];
                if (!valid1[idx1]) continue;                       /* !pMP || pMP->isBad() */
                const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = fv2->offset[b]; i2 < fv2->offset[b + 1]; i2++) {
                    const int idx2 = (int)fv2->idx[i2];
                    if (variant == 0) { if (match_out[idx2] >= 0) continue; }
                    else { if (matched2[idx2] || !valid2[idx2]) continue; }
                    const int dist = orc_descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                int pass = variant == 0 ? (bestDist1 <= TH_LOW) : (bestDist1 < TH_LOW);
                if (pass && (float)bestDist1 < nnratio * (float)bestDist2) {
                    int slot;
                    if (variant == 0) { match_out[bestIdx2] = idx1; slot = bestIdx2; }
                    else { match_out[idx1] = bestIdx2; matched2[bestIdx2] = 1; slot = idx1; }
                    if (check_ori) { int bin = rot_bin(angle1[idx1], angle2[bestIdx2]); hist_bin[slot] = bin; hist[bin]++; }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (fv1->node_id[a] < fv2->node_id[b]) {
            while (a < fv1->n_nodes && fv1->node_id[a] < fv2->node_id[b]) a++;     /* lower_bound */
        } else {
            while (b < fv2->n_nodes && fv2->node_id[b] < fv1->node_id[a]) b++;
        }
    }
    if (check_ori) {
        int i1, i2, i3; three_maxima(hist, HISTO_LENGTH, &i1, &i2, &i3);
        for (int s = 0; s < nout; s++) {
            int bin = hist_bin[s];
            if (bin < 0 || bin == i1 || bin == i2 || bin == i3) continue;
            match_out[s] = -1; nmatches--;
        }
    }
    free(matched2); free(hist_bin);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine (ORBmatcher.cc:142-159) */
static int check_epipolar(const OrcKeyPoint* kp1, const OrcKeyPoint* kp2, const float* F12, const float* sigma2_2)
{
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2_2[kp2->octave];
}

/* ORBmatcher::SearchForTriangulation (ORBmatcher.cc:792-958) */
int orc_search_for_triangulation(
        const uint8_t* desc1, const OrcKeyPoint* kp1, const float* uright1, const uint8_t* has_mp1, int n1, const OrcFeatVec* fv1,
        const uint8_t* desc2, const OrcKeyPoint* kp2, const float* uright2, const uint8_t* has_mp2, int n2, const OrcFeatVec* fv2,
        const OrcTriParams* p, int only_stereo, int check_ori, int32_t* pairs_out)
{
    int* m12 = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    int* bins = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++) { m12[i] = -1; bins[i] = -1; }
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    int nmatches = 0;
    (void)n2;
    int a = 0, b = 0;
    while (a < fv1->n_nodes && b < fv2->n_nodes) {
        if (fv1->node_id[a] == fv2->node_id[b]) {
            for (int i1 = fv1->offset[a]; i1 < fv1->offset[a + 1]; i1++) {
                const int idx1 = (int)fv1->idx[i1];
                if (has_mp1[idx1]) continue;
                const int bStereo1 = uright1[idx1] >= 0;
                if (only_stereo && !bStereo1) continue;
                const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = fv2->offset[b]; i2 < fv2->offset[b + 1]; i2++) {
                    const int idx2 = (int)fv2->idx[i2];
                    if (has_mp2[idx2]) continue;                    /* vbMatched2 is never set (:812, :860) */
                    const int bStereo2 = uright2[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = orc_descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    if (!bStereo1 && !bStereo2) {
                        const float distex = p->ex - kp2[idx2].x, distey = p->ey - kp2[idx2].y;
                        if (distex * distex + distey * distey < 100 * p->scale2[kp2[idx2].octave]) continue;
                    }
                    if (check_epipolar(&kp1[idx1], &kp2[idx2], p->F12, p->sigma2_2)) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    m12[idx1] = bestIdx2; nmatches++;
                    if (check_ori) { int bin = rot_bin(kp1[idx1].angle, kp2[bestIdx2].angle); bins[idx1] = bin; hist[bin]++; }
                }
            }
            a++; b++;
        } else if (fv1->node_id[a] < fv2->node_id[b]) {
            while (a < fv1->n_nodes && fv1->node_id[a] < fv2->node_id[b]) a++;
        } else {
            while (b < fv2->n_nodes && fv2->node_id[b] < fv1->node_id[a]) b++;
        }
    }
    if (check_ori) {
        int i1, i2, i3; three_maxima(hist, HISTO_LENGTH, &i1, &i2, &i3);
        for (int s = 0; s < n1; s++) {
            int bin = bins[s];
            if (bin < 0 || bin == i1 || bin == i2 || bin == i3) continue;
            m12[s] = -1; nmatches--;
        }
    }
    int k = 0;
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) { pairs_out[2 * k] = i; pairs_out[2 * k + 1] = m12[i]; k++; }
    free(m12); free(bins);
    return nmatches;
}
