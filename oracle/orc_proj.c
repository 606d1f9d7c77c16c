/* CPU ORACLE (test infrastructure) -- projection-guided matchers of the Tracking thread.
 * Restates ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)   (corbslam_client/src/ORBmatcher.cc:45-131),
 *          ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)        (ORBmatcher.cc:1470-1614),
 *          Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea            (corbslam_client/src/Frame.cc:230-245, 331-395),
 *          SearchByProjection(Frame&, KeyFrame*, ...) (:1616-1744), Fuse x2 (:960-1241), SearchBySim3 (:1244-1468),
 *          KeyFrame::GetFeaturesInArea / IsInImage (KeyFrame.cc:700-739), MapPoint::PredictScale (MapPoint.cc:484-514).
 * Float arithmetic as written in the reference; the 3x3 * 3x1 + 3x1 products are cv::gemm on CV_32F (double accumulation,
 * one rounding to float).  See orc.h for scope.  Compile with -ffp-contract=off. */
#include "orc.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GRID_COLS 64        /* FRAME_GRID_COLS (Frame.h:39) */
#define GRID_ROWS 48        /* FRAME_GRID_ROWS (Frame.h:38) */
#define TH_HIGH 100
#define HISTO_LENGTH 30

typedef struct { int* off; int* idx; float winv, hinv; } Grid;

static void grid_build(const OrcFrameView* F, Grid* g)
{
    g->winv = (float)GRID_COLS / (F->max_x - F->min_x);        /* Frame.cc:101-102 */
    g->hinv = (float)GRID_ROWS / (F->max_y - F->min_y);
    g->off = (int*)calloc(GRID_COLS * GRID_ROWS + 1, sizeof(int));
    g->idx = (int*)malloc(sizeof(int) * (F->n > 0 ? F->n : 1));
    int* cell = (int*)malloc(sizeof(int) * (F->n > 0 ? F->n : 1));
    for (int i = 0; i < F->n; i++) {                            /* PosInGrid (Frame.cc:386-395) */
        int px = (int)roundf((F->keys_un[i].x - F->min_x) * g->winv);
        int py = (int)roundf((F->keys_un[i].y - F->min_y) * g->hinv);
        cell[i] = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
        if (cell[i] >= 0) g->off[cell[i] + 1]++;
    }
    for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) g->off[c + 1] += g->off[c];
    int* cur = (int*)malloc(sizeof(int) * GRID_COLS * GRID_ROWS);
    memcpy(cur, g->off, sizeof(int) * GRID_COLS * GRID_ROWS);
    for (int i = 0; i < F->n; i++) if (cell[i] >= 0) g->idx[cur[cell[i]]++] = i;   /* ascending feature index inside a cell */
    free(cur); free(cell);
}
static void grid_free(Grid* g) { free(g->off); free(g->idx); }

/* Frame::GetFeaturesInArea (Frame.cc:331-384): appends feature indices in the reference's visiting order */
static int features_in_area(const OrcFrameView* F, const Grid* g, float x, float y, float r, int minLevel, int maxLevel, int* out)
{
    int n = 0;
    int nMinCellX = (int)floorf((x - F->min_x - r) * g->winv); if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - F->min_x + r) * g->winv); if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - F->min_y - r) * g->hinv); if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - F->min_y + r) * g->hinv); if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (int j = g->off[ix * GRID_ROWS + iy]; j < g->off[ix * GRID_ROWS + iy + 1]; j++) {
                const int f = g->idx[j];
                const OrcKeyPoint* kp = &F->keys_un[f];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x, disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) out[n++] = f;
            }
    return n;
}

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

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)  (ORBmatcher.cc:45-131) */
int orc_search_by_projection_map(const OrcFrameView* F, const OrcTrackedPoint* mp, const uint8_t* mp_desc, int n_mp,
                                 float th, float nnratio, int32_t* match)
{
    Grid g; grid_build(F, &g);
    uint8_t* claimed = (uint8_t*)malloc(F->n > 0 ? F->n : 1);
    memcpy(claimed, F->claimed, F->n);
    for (int i = 0; i < F->n; i++) match[i] = -1;
    int* cand = (int*)malloc(sizeof(int) * (F->n > 0 ? F->n : 1));
    int nmatches = 0;
    const int bFactor = th != 1.0;
    for (int iMP = 0; iMP < n_mp; iMP++) {
        const OrcTrackedPoint* p = &mp[iMP];
        if (!p->valid) continue;                                  /* !mbTrackInView || isBad() */
        const int lvl = p->level;
        float r = ((double)p->view_cos > 0.998) ? 2.5f : 4.0f;    /* RadiusByViewingCos (:133-139): float compared with the double 0.998 */
        if (bFactor) r *= th;
        const float win = r * F->scale[lvl];
        const int nc = features_in_area(F, &g, p->proj_x, p->proj_y, win, lvl - 1, lvl, cand);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (claimed[idx]) continue;                           /* tMP && tMP->Observations()>0 */
            if (F->u_right[idx] > 0) {
                const float er = fabsf(p->proj_xr - F->u_right[idx]);
                if (er > win) continue;
            }
            const int dist = orc_descriptor_distance(mp_desc + (size_t)iMP * 32, F->desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F->keys_un[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = F->keys_un[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            match[bestIdx] = iMP;
            claimed[bestIdx] = p->claims;                         /* the assigned MapPoint's Observations()>0 */
            nmatches++;
        }
    }
    free(cand); free(claimed); grid_free(&g);
    return nmatches;
}

/* D = (float)(A(3x3 float) * x(3 float) + c(3 float)) with double accumulation: cv::gemm(A, x, 1, c, 1) on CV_32F */
static void gemm3(const float* A, int lda, const float* x, const float* c, float* d)
{
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)A[i * lda + k] * (double)x[k];
        d[i] = (float)(s + (c ? (double)c[i] : 0.0));
    }
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) (:1470-1614) */
int orc_search_by_projection_frame(const OrcFrameView* C, const float* Tcw, const float* Tlw, float fx, float fy, float cx, float cy,
                                   float bf, float mb, const OrcLastPoint* last, const uint8_t* last_desc, int n_last,
                                   float th, int bMono, int check_ori, int32_t* match)
{
    Grid g; grid_build(C, &g);
    uint8_t* claimed = (uint8_t*)malloc(C->n > 0 ? C->n : 1);
    memcpy(claimed, C->claimed, C->n);
    for (int i = 0; i < C->n; i++) match[i] = -1;
    int* cand = (int*)malloc(sizeof(int) * (C->n > 0 ? C->n : 1));
    int* ev_feat = (int*)malloc(sizeof(int) * (n_last > 0 ? n_last : 1)); int* ev_bin = (int*)malloc(sizeof(int) * (n_last > 0 ? n_last : 1)); int nev = 0;
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    int nmatches = 0;
    /* twc = -Rcw^T * tcw ; tlc = Rlw*twc + tlw   (:1480-1488) */
    float Rt[9], twc[3], ntcw[3] = { Tcw[3], Tcw[7], Tcw[11] }, tlc[3], tlw[3] = { Tlw[3], Tlw[7], Tlw[11] };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[i * 3 + j] = -Tcw[j * 4 + i];     /* -Rcw.t() (exact negation) */
    gemm3(Rt, 3, ntcw, NULL, twc);
    gemm3(Tlw, 4, twc, tlw, tlc);
    const int bForward = tlc[2] > mb && !bMono;
    const int bBackward = -tlc[2] > mb && !bMono;
    const float factor = 1.0f / HISTO_LENGTH;
    const float tcw[3] = { Tcw[3], Tcw[7], Tcw[11] };
    for (int i = 0; i < n_last; i++) {
        const OrcLastPoint* p = &last[i];
        if (!p->valid) continue;                                  /* pMP && !mvbOutlier[i] */
        float x3Dc[3];
        gemm3(Tcw, 4, p->world, tcw, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (u < C->min_x || u > C->max_x) continue;
        if (v < C->min_y || v > C->max_y) continue;
        const int nLastOctave = p->octave;
        const float radius = th * C->scale[nLastOctave];
        int nc;
        if (bForward) nc = features_in_area(C, &g, u, v, radius, nLastOctave, -1, cand);
        else if (bBackward) nc = features_in_area(C, &g, u, v, radius, 0, nLastOctave, cand);
        else nc = features_in_area(C, &g, u, v, radius, nLastOctave - 1, nLastOctave + 1, cand);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            if (claimed[i2]) continue;
            if (C->u_right[i2] > 0) {
                const float ur = u - bf * invzc;
                const float er = fabsf(ur - C->u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(last_desc + (size_t)i * 32, C->desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            match[bestIdx2] = i;
            claimed[bestIdx2] = p->claims;
            nmatches++;
            if (check_ori) {
                float rot = p->angle - C->keys_un[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                ev_feat[nev] = bestIdx2; ev_bin[nev] = bin; nev++; hist[bin]++;
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3; three_maxima(hist, HISTO_LENGTH, &i1, &i2, &i3);
        for (int e = 0; e < nev; e++)
            if (ev_bin[e] != i1 && ev_bin[e] != i2 && ev_bin[e] != i3) { match[ev_feat[e]] = -1; nmatches--; }
    }
    free(cand); free(claimed); free(ev_feat); free(ev_bin); grid_free(&g);
    return nmatches;
}

/* =====================================================================================================================
 * Matchers that project MapPoints into a KeyFrame (or, for relocalisation, into the current Frame).
 * Readings of the OpenCV expressions (2.4.8, CV_32F): A*x + b = cv::gemm, double accumulation, one rounding;
 * cv::norm = sqrt of a double sum of (double)v*v; Mat::dot = double sum of (double)a*b; s*M / M/s = per-element float
 * multiply by (float)s resp. (float)(1/s) (convertTo with a scale).
 * DEFINED: MapPoint::PredictScale calls libm log on a float (platform dependent): here (float)log((double)ratio). */
#define TH_LOW 50
static OrcFrameView as_frame(const OrcKeyFrameView* K, const uint8_t* claimed)
{
    OrcFrameView F; F.keys_un = K->keys_un; F.u_right = K->u_right; F.desc = K->desc; F.n = K->n; F.claimed = claimed;
    F.min_x = K->min_x; F.min_y = K->min_y; F.max_x = K->max_x; F.max_y = K->max_y; F.scale = K->scale; F.nlevels = K->nlevels;
    return F;
}
static float norm3(const float* v) { return (float)sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }
static int predict_scale(float max_distance, float dist, float log_scale_factor, int nlevels)
{
    const float ratio = max_distance / dist;
    const float lg = (float)log((double)ratio);
    int n = (int)ceilf(lg / log_scale_factor);
    if (n < 0) n = 0; else if (n >= nlevels) n = nlevels - 1;
    return n;
}
/* Ow = -Rcw^T * tcw (exact negation of the transposed rotation, then gemm) */
static void camera_centre(const float* Tcw, float* Ow)
{
    float Rt[9], t[3] = { Tcw[3], Tcw[7], Tcw[11] };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[i * 3 + j] = -Tcw[j * 4 + i];
    gemm3(Rt, 3, t, NULL, Ow);
}

int orc_search_by_projection_reloc(const OrcKeyFrameView* C, const uint8_t* claimed_in, const float* Tcw, const OrcMapPointView* pts,
                                   const uint8_t* desc, int n, float th, int orb_dist, int check_ori, int32_t* match)
{
    OrcFrameView F = as_frame(C, claimed_in);
    Grid g; grid_build(&F, &g);
    uint8_t* claimed = (uint8_t*)malloc(C->n > 0 ? C->n : 1);
    memcpy(claimed, claimed_in, C->n);
    for (int i = 0; i < C->n; i++) match[i] = -1;
    int* cand = (int*)malloc(sizeof(int) * (C->n > 0 ? C->n : 1));
    int* ev_feat = (int*)malloc(sizeof(int) * (n > 0 ? n : 1)); int* ev_bin = (int*)malloc(sizeof(int) * (n > 0 ? n : 1)); int nev = 0;
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    int nmatches = 0;
    float Ow[3]; camera_centre(Tcw, Ow);
    const float tcw[3] = { Tcw[3], Tcw[7], Tcw[11] };
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < n; i++) {
        const OrcMapPointView* p = &pts[i];
        if (!p->valid) continue;                                   /* pMP && !isBad() && !sAlreadyFound.count(pMP) */
        float x3Dc[3];
        gemm3(Tcw, 4, p->world, tcw, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        const float u = C->fx * xc * invzc + C->cx;
        const float v = C->fy * yc * invzc + C->cy;
        if (u < C->min_x || u > C->max_x) continue;
        if (v < C->min_y || v > C->max_y) continue;
        const float PO[3] = { p->world[0] - Ow[0], p->world[1] - Ow[1], p->world[2] - Ow[2] };
        const float dist3D = norm3(PO);
        const float maxDistance = 1.2f * p->max_distance, minDistance = 0.8f * p->min_distance;
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int lvl = predict_scale(p->max_distance, dist3D, C->log_scale_factor, C->nlevels);
        const float radius = th * C->scale[lvl];
        const int nc = features_in_area(&F, &g, u, v, radius, lvl - 1, lvl + 1, cand);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            if (claimed[i2]) continue;                             /* CurrentFrame.mvpMapPoints[i2].getMapPoint() */
            const int dist = orc_descriptor_distance(desc + (size_t)i * 32, C->desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {
            match[bestIdx2] = i; claimed[bestIdx2] = 1; nmatches++;
            if (check_ori) {
                float rot = p->angle - C->keys_un[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                ev_feat[nev] = bestIdx2; ev_bin[nev] = bin; nev++; hist[bin]++;
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3; three_maxima(hist, HISTO_LENGTH, &i1, &i2, &i3);
        for (int e = 0; e < nev; e++)
            if (ev_bin[e] != i1 && ev_bin[e] != i2 && ev_bin[e] != i3) { match[ev_feat[e]] = -1; nmatches--; }
    }
    free(cand); free(claimed); free(ev_feat); free(ev_bin); grid_free(&g);
    return nmatches;
}

/* int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize) (ORBmatcher.cc:540-655),
 * the monocular initialiser's matcher (Tracking::MonocularInitialization, C/src/Tracking.cc:606).  Level-0 features of F1 only (:558-560); the window is centred on
 * vbPrevMatched[i1] with GetFeaturesInArea(x, y, windowSize, 0, 0) (:562); a candidate whose current match is at least as close is skipped (:583); best and second-best
 * distance over the remaining candidates, TH_LOW and the ratio test `bestDist < (float)bestDist2 * mfNNratio` (:597-599; bestDist2 stays INT_MAX with one candidate);
 * a feature of F2 that is taken again loses its earlier partner (:601-605); every commit enters the rotation histogram -- a partner lost later stays in it (:610-619) --
 * and the bins outside the three maxima lose their matches (:625-648); vbPrevMatched of the matched features moves to the F2 keypoint (:651-653).
 * prev_matched: n1 x 2 floats, in / out.  match12[i1] = feature of F2 or -1; returns nmatches. */
int orc_search_for_initialization(const OrcFrameView* F1, const OrcFrameView* F2, float* prev_matched, int window_size, float nnratio, int check_ori, int32_t* match12)
{
    Grid g; grid_build(F2, &g);
    const int n1 = F1->n, n2 = F2->n;
    int* cand = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int* matched_dist = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int* match21 = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int* ev_i1 = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1)); int* ev_bin = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1)); int nev = 0;
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    for (int i = 0; i < n2; i++) { matched_dist[i] = INT_MAX; match21[i] = -1; }
    for (int i = 0; i < n1; i++) match12[i] = -1;
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        if (F1->keys_un[i1].octave > 0) continue;
        const int nc = features_in_area(F2, &g, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size, 0, 0, cand);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            const int dist = orc_descriptor_distance(F1->desc + (size_t)i1 * 32, F2->desc + (size_t)i2 * 32);
            if (matched_dist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {
            if (match21[bestIdx2] >= 0) { match12[match21[bestIdx2]] = -1; nmatches--; }
            match12[i1] = bestIdx2; match21[bestIdx2] = i1; matched_dist[bestIdx2] = bestDist; nmatches++;
            if (check_ori) {
                float rot = F1->keys_un[i1].angle - F2->keys_un[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                ev_i1[nev] = i1; ev_bin[nev] = bin; nev++; hist[bin]++;
            }
        }
    }
    if (check_ori) {
        int a, b, c; three_maxima(hist, HISTO_LENGTH, &a, &b, &c);
        for (int e = 0; e < nev; e++)
            if (ev_bin[e] != a && ev_bin[e] != b && ev_bin[e] != c && match12[ev_i1[e]] >= 0) { match12[ev_i1[e]] = -1; nmatches--; }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (match12[i1] >= 0) { prev_matched[2 * i1] = F2->keys_un[match12[i1]].x; prev_matched[2 * i1 + 1] = F2->keys_un[match12[i1]].y; }
    free(cand); free(matched_dist); free(match21); free(ev_i1); free(ev_bin); grid_free(&g);
    return nmatches;
}

/* best keyframe feature for one projected point: KeyFrame::GetFeaturesInArea(u, v, radius) + the per-candidate tests */
static int best_in_keyframe(const OrcKeyFrameView* K, const OrcFrameView* F, const Grid* g, int* cand, const uint8_t* d, float u, float v, float ur,
                            float radius, int lvl, int chi2_check, int* best_dist)
{
    const int nc = features_in_area(F, g, u, v, radius, -1, -1, cand);
    int bestDist = 256, bestIdx = -1;                              /* INT_MAX in two of the routines: same result, distances are <= 256 */
    for (int c = 0; c < nc; c++) {
        const int idx = cand[c];
        const OrcKeyPoint* kp = &K->keys_un[idx];
        const int kpLevel = kp->octave;
        if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
        if (chi2_check) {
            const float ex = u - kp->x, ey = v - kp->y;
            if (K->u_right[idx] >= 0) {
                const float er = ur - K->u_right[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if ((double)(e2 * K->inv_level_sigma2[kpLevel]) > 7.8) continue;
            } else {
                const float e2 = ex * ex + ey * ey;
                if ((double)(e2 * K->inv_level_sigma2[kpLevel]) > 5.99) continue;
            }
        }
        const int dist = orc_descriptor_distance(d, K->desc + (size_t)idx * 32);
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    *best_dist = bestDist;
    return bestIdx;
}

int orc_fuse(const OrcKeyFrameView* K, const float* T, const float* Ow_in, int sim3, const OrcMapPointView* pts, const uint8_t* desc, int n,
             float th, int32_t* best_idx, int32_t* best_dist)
{
    OrcFrameView F = as_frame(K, NULL);
    Grid g; grid_build(&F, &g);
    int* cand = (int*)malloc(sizeof(int) * (K->n > 0 ? K->n : 1));
    float M[16], Ow[3];
    if (sim3) {                                                    /* decompose Scw (:1124-1128) */
        const double dd = (double)T[0] * T[0] + (double)T[1] * T[1] + (double)T[2] * T[2];      /* sRcw.row(0).dot(sRcw.row(0)) */
        const float scw = (float)sqrt(dd);
        const float inv = (float)(1.0 / (double)scw);              /* M / s = M * (1/s): the scale is a double, applied as a float */
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[i * 4 + j] = T[i * 4 + j] * inv; M[i * 4 + 3] = T[i * 4 + 3] * inv; }
        M[12] = M[13] = M[14] = 0; M[15] = 1;
        camera_centre(M, Ow);
    } else { memcpy(M, T, sizeof(M)); Ow[0] = Ow_in[0]; Ow[1] = Ow_in[1]; Ow[2] = Ow_in[2]; }
    const float tcw[3] = { M[3], M[7], M[11] };
    int nFused = 0;
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        const OrcMapPointView* p = &pts[i];
        if (!p->valid) continue;
        float p3Dc[3];
        gemm3(M, 4, p->world, tcw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float invz = sim3 ? (float)(1.0 / p3Dc[2]) : 1 / p3Dc[2];     /* `1.0/z` (:1156) vs `1/z` (:996) */
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = K->fx * x + K->cx, v = K->fy * y + K->cy;
        if (!(u >= K->min_x && u < K->max_x && v >= K->min_y && v < K->max_y)) continue;          /* IsInImage */
        const float ur = u - K->bf * invz;
        const float PO[3] = { p->world[0] - Ow[0], p->world[1] - Ow[1], p->world[2] - Ow[2] };
        const float dist3D = norm3(PO);
        const float maxDistance = 1.2f * p->max_distance, minDistance = 0.8f * p->min_distance;
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const double dot = (double)PO[0] * p->normal[0] + (double)PO[1] * p->normal[1] + (double)PO[2] * p->normal[2];
        if (dot < 0.5 * dist3D) continue;                          /* viewing angle < 60 deg */
        const int lvl = predict_scale(p->max_distance, dist3D, K->log_scale_factor, K->nlevels);
        const float radius = th * K->scale[lvl];
        int bd;
        const int bi = best_in_keyframe(K, &F, &g, cand, desc + (size_t)i * 32, u, v, ur, radius, lvl, !sim3, &bd);
        best_dist[i] = bd;
        if (bi >= 0 && bd <= TH_LOW) { best_idx[i] = bi; nFused++; }
    }
    free(cand); grid_free(&g);
    return nFused;
}

/* SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th) (ORBmatcher.cc:425-538), the
 * projection matcher of the loop / map-fusion event (callers: LoopClosing.cc:377; S/src/GlobalOptimize.cpp:199).  Scw is decomposed as in Fuse(KeyFrame*, Scw, ...)
 * (:434-438 = :1124-1128); the gates are Fuse's (depth, IsInImage, distance invariance, viewing angle) with `1/z` as a FLOAT division (:467), the candidates are
 * GetFeaturesInArea(u, v, radius) filtered by octave in [level-1, level] inside the loop (:507-510), no chi2 test.  The loop is sequential and order dependent:
 * a feature that holds a point -- on entry (vpMatched[idx] != NULL, `claimed_in`) or since an earlier point of THIS call took it (:530) -- is skipped, the set of
 * already found points is fixed on entry (:441-442; the adapter folds it, with isBad(), into `valid`).  match[idx] = index of the point written into
 * vpMatched[idx] by this call or -1; returns nmatches. */
int orc_search_by_projection_scw(const OrcKeyFrameView* K, const uint8_t* claimed_in, const float* Scw, const OrcMapPointView* pts, const uint8_t* desc, int n,
                                 float th, int32_t* match)
{
    OrcFrameView F = as_frame(K, NULL);
    Grid g; grid_build(&F, &g);
    int* cand = (int*)malloc(sizeof(int) * (K->n > 0 ? K->n : 1));
    uint8_t* claimed = (uint8_t*)malloc(K->n > 0 ? K->n : 1);
    if (K->n > 0) memcpy(claimed, claimed_in, K->n);
    for (int i = 0; i < K->n; i++) match[i] = -1;
    float M[16], Ow[3];
    const double dd = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];      /* sRcw.row(0).dot(sRcw.row(0)) (:435) */
    const float scw = (float)sqrt(dd);
    const float inv = (float)(1.0 / (double)scw);                  /* M / s = M * (1/s): the scale is a double, applied as a float (:436-437) */
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[i * 4 + j] = Scw[i * 4 + j] * inv; M[i * 4 + 3] = Scw[i * 4 + 3] * inv; }
    M[12] = M[13] = M[14] = 0; M[15] = 1;
    camera_centre(M, Ow);                                          /* :438 */
    const float tcw[3] = { M[3], M[7], M[11] };
    int nmatches = 0;
    for (int i = 0; i < n; i++) {
        const OrcMapPointView* p = &pts[i];
        if (!p->valid) continue;                                   /* pMP->isBad() || spAlreadyFound.count(pMP) (:452) */
        float p3Dc[3];
        gemm3(M, 4, p->world, tcw, p3Dc);                          /* :459 */
        if (p3Dc[2] < 0.0f) continue;                              /* :462 */
        const float invz = 1 / p3Dc[2];                            /* :466, a float division */
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = K->fx * x + K->cx, v = K->fy * y + K->cy;
        if (!(u >= K->min_x && u < K->max_x && v >= K->min_y && v < K->max_y)) continue;          /* IsInImage (:474) */
        const float PO[3] = { p->world[0] - Ow[0], p->world[1] - Ow[1], p->world[2] - Ow[2] };
        const float dist3D = norm3(PO);
        const float maxDistance = 1.2f * p->max_distance, minDistance = 0.8f * p->min_distance;
        if (dist3D < minDistance || dist3D > maxDistance) continue;                                 /* :483 */
        const double dot = (double)PO[0] * p->normal[0] + (double)PO[1] * p->normal[1] + (double)PO[2] * p->normal[2];
        if (dot < 0.5 * dist3D) continue;                          /* :489 */
        const int lvl = predict_scale(p->max_distance, dist3D, K->log_scale_factor, K->nlevels);
        const float radius = th * K->scale[lvl];                   /* :495 */
        const int nc = features_in_area(&F, &g, u, v, radius, -1, -1, cand);
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (claimed[idx]) continue;                            /* vpMatched[idx] (:510) */
            const int kpLevel = K->keys_un[idx].octave;
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;      /* :515 */
            const int dist = orc_descriptor_distance(desc + (size_t)i * 32, K->desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { match[bestIdx] = i; claimed[bestIdx] = 1; nmatches++; }           /* :528-532 */
    }
    free(cand); free(claimed); grid_free(&g);
    return nmatches;
}

/* one direction of SearchBySim3: points of keyframe A (camera A from world, then A -> B by the similarity) into keyframe B */
static void sim3_direction(const OrcKeyFrameView* B, const float* TAw, const float* sR, const float* t, const OrcMapPointView* pts,
                           const uint8_t* desc, int n, float th, float fx, float fy, float cx, float cy, int32_t* out)
{
    OrcFrameView F = as_frame(B, NULL);
    Grid g; grid_build(&F, &g);
    int* cand = (int*)malloc(sizeof(int) * (B->n > 0 ? B->n : 1));
    const float tAw[3] = { TAw[3], TAw[7], TAw[11] };
    for (int i = 0; i < n; i++) {
        out[i] = -1;
        const OrcMapPointView* p = &pts[i];
        if (!p->valid) continue;                                   /* pMP && !vbAlreadyMatched && !isBad() */
        float pa[3], pb[3];
        gemm3(TAw, 4, p->world, tAw, pa);
        gemm3(sR, 3, pa, t, pb);
        if ((double)pb[2] < 0.0) continue;
        const float invz = (float)(1.0 / pb[2]);
        const float x = pb[0] * invz, y = pb[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= B->min_x && u < B->max_x && v >= B->min_y && v < B->max_y)) continue;
        const float dist3D = norm3(pb);
        const float maxDistance = 1.2f * p->max_distance, minDistance = 0.8f * p->min_distance;
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int lvl = predict_scale(p->max_distance, dist3D, B->log_scale_factor, B->nlevels);
        const float radius = th * B->scale[lvl];
        int bd;
        const int bi = best_in_keyframe(B, &F, &g, cand, desc + (size_t)i * 32, u, v, 0.f, radius, lvl, 0, &bd);
        if (bi >= 0 && bd <= TH_HIGH) out[i] = bi;
    }
    free(cand); grid_free(&g);
}

int orc_search_by_sim3(const OrcKeyFrameView* K1, const OrcKeyFrameView* K2, const float* T1w, const float* T2w,
                       const OrcMapPointView* pts1, const uint8_t* desc1, const OrcMapPointView* pts2, const uint8_t* desc2,
                       float s12, const float* R12, const float* t12, float th, int32_t* match12)
{
    /* sR12 = s12*R12 ; sR21 = (1.0/s12)*R12.t() ; t21 = -sR21*t12   (:1262-1264) */
    float sR12[9], sR21[9], t21[3], nsR21[9];
    const float is = (float)(1.0 / (double)s12);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { sR12[i * 3 + j] = R12[i * 3 + j] * s12; sR21[i * 3 + j] = R12[j * 3 + i] * is; }
    for (int i = 0; i < 9; i++) nsR21[i] = -sR21[i];
    gemm3(nsR21, 3, t12, NULL, t21);
    const int N1 = K1->n, N2 = K2->n;
    int32_t* m1 = (int32_t*)malloc(sizeof(int32_t) * (N1 > 0 ? N1 : 1));
    int32_t* m2 = (int32_t*)malloc(sizeof(int32_t) * (N2 > 0 ? N2 : 1));
    /* the intrinsics of BOTH directions are pKF1's (:1247-1250) */
    sim3_direction(K2, T1w, sR21, t21, pts1, desc1, N1, th, K1->fx, K1->fy, K1->cx, K1->cy, m1);
    sim3_direction(K1, T2w, sR12, t12, pts2, desc2, N2, th, K1->fx, K1->fy, K1->cx, K1->cy, m2);
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    free(m1); free(m2);
    return nFound;
}
