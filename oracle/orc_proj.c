/* CPU ORACLE (test infrastructure) -- projection-guided matchers of the Tracking thread.
 * Restates ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)   (corbslam_client/src/ORBmatcher.cc:45-131),
 *          ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)        (ORBmatcher.cc:1470-1614),
 *          Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea            (corbslam_client/src/Frame.cc:230-245, 331-395).
 * Float arithmetic as written in the reference; the 3x3 * 3x1 + 3x1 products are cv::gemm on CV_32F (double accumulation,
 * one rounding to float).  See orc.h for scope.  Compile with -ffp-contract=off. */
#include "orc.h"
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
