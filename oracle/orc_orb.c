/* CPU ORACLE (test infrastructure) -- ORB extraction.
 * Restates corbslam_client/src/ORBextractor.cc and the OpenCV 2.4.8 scalar routines it calls.
 * See orc.h for the scope / "parity unpinned" statement.  Compile with -ffp-contract=off. */
#include "orc.h"
#include "brief_pattern.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define PATCH_SIZE 31
#define HALF_PATCH_SIZE 15
#define EDGE_THRESHOLD 19
#define MAX_LEVELS 16

static inline int cv_round(double v) { return (int)lrint(v); }             /* cvRound: round-half-even */
static inline int cv_floor(double v) { int i = cv_round(v); float d = (float)(v - i); return i - (d < 0); }
static inline int cv_ceil(double v)  { int i = cv_round(v); float d = (float)(i - v); return i + (d < 0); }

struct OrcExtractor {
    int nfeatures, nlevels, ini_th, min_th;
    float scale_factor;
    float scale[MAX_LEVELS], inv_scale[MAX_LEVELS], sigma2[MAX_LEVELS], inv_sigma2[MAX_LEVELS];
    int quota[MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    /* state of the last extraction */
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    uint8_t* pyr[MAX_LEVELS];
    uint8_t* blur[MAX_LEVELS];
    OrcKeyPoint* cand[MAX_LEVELS]; int ncand[MAX_LEVELS];
    int nkp[MAX_LEVELS];
};

/* ------------------------------------------------------------------------------------------ */
/* ORBextractor::ORBextractor  (ORBextractor.cc:410-470)                                      */
OrcExtractor* orc_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    if (nlevels < 1 || nlevels > MAX_LEVELS) return NULL;
    OrcExtractor* ex = (OrcExtractor*)calloc(1, sizeof(*ex));
    ex->nfeatures = nfeatures; ex->scale_factor = scale_factor; ex->nlevels = nlevels;
    ex->ini_th = ini_th; ex->min_th = min_th;
    ex->scale[0] = 1.0f; ex->sigma2[0] = 1.0f;                                   /* :417-423 */
    for (int i = 1; i < nlevels; i++) {
        ex->scale[i] = ex->scale[i - 1] * scale_factor;
        ex->sigma2[i] = ex->scale[i] * ex->scale[i];
    }
    for (int i = 0; i < nlevels; i++) {                                          /* :427-431 */
        ex->inv_scale[i] = 1.0f / ex->scale[i];
        ex->inv_sigma2[i] = 1.0f / ex->sigma2[i];
    }
    float factor = 1.0f / scale_factor;                                          /* :436-446 */
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        ex->quota[level] = cv_round(nDesired);
        sum += ex->quota[level];
        nDesired *= factor;
    }
    ex->quota[nlevels - 1] = (nfeatures - sum) > 0 ? (nfeatures - sum) : 0;
    /* umax (:454-469) */
    int v, v0;
    int vmax = cv_floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) ex->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (ex->umax[v0] == ex->umax[v0 + 1]) ++v0;
        ex->umax[v] = v0;
        ++v0;
    }
    return ex;
}

static void free_state(OrcExtractor* ex)
{
    for (int l = 0; l < MAX_LEVELS; l++) {
        free(ex->pyr[l]); ex->pyr[l] = NULL;
        free(ex->blur[l]); ex->blur[l] = NULL;
        free(ex->cand[l]); ex->cand[l] = NULL;
        ex->ncand[l] = 0; ex->nkp[l] = 0; ex->lw[l] = ex->lh[l] = 0;
    }
}

void orc_orb_destroy(OrcExtractor* ex) { if (ex) { free_state(ex); free(ex); } }

void orc_orb_tables(const OrcExtractor* ex, float* scale, float* inv_scale, float* sigma2,
                    float* inv_sigma2, int* quota, int* umax16)
{
    for (int i = 0; i < ex->nlevels; i++) {
        if (scale) scale[i] = ex->scale[i];
        if (inv_scale) inv_scale[i] = ex->inv_scale[i];
        if (sigma2) sigma2[i] = ex->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = ex->inv_sigma2[i];
        if (quota) quota[i] = ex->quota[i];
    }
    if (umax16) for (int i = 0; i <= HALF_PATCH_SIZE; i++) umax16[i] = ex->umax[i];
}

int orc_orb_level_dims(const OrcExtractor* ex, int level, int* w, int* h)
{ if (level < 0 || level >= ex->nlevels) return -1; *w = ex->lw[level]; *h = ex->lh[level]; return 0; }
const uint8_t* orc_orb_level_data(const OrcExtractor* ex, int level) { return ex->pyr[level]; }
const uint8_t* orc_orb_blur_data(const OrcExtractor* ex, int level) { return ex->blur[level]; }
int orc_orb_level_count(const OrcExtractor* ex, int level) { return ex->nkp[level]; }
int orc_orb_level_candidates(const OrcExtractor* ex, int level, OrcKeyPoint* out, int cap)
{
    int n = ex->ncand[level];
    if (out) memcpy(out, ex->cand[level], sizeof(OrcKeyPoint) * (size_t)(n < cap ? n : cap));
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::resize(8UC1, INTER_LINEAR) -- OpenCV 2.4.8 imgproc/imgwarp.cpp resizeGeneric_<HResizeLinear,
 * VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>>, scalar path.  Call site :1120.   */
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { if (dx < xmax) xmax = dx; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx;
        ialpha[dx * 2]     = sat_short(cv_round((1.f - fx) * 2048));
        ialpha[dx * 2 + 1] = sat_short(cv_round(fx * 2048));
    }
    int* row0 = (int*)malloc(sizeof(int) * dw);
    int* row1 = (int*)malloc(sizeof(int) * dw);
    int cached0 = -1, cached1 = -1;   /* source rows held in row0/row1 (pure cache, no effect on values) */
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        short b0 = sat_short(cv_round((1.f - fy) * 2048));
        short b1 = sat_short(cv_round(fy * 2048));
        int sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);          /* clip(sy, 0, sh) */
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        for (int k = 0; k < 2; k++) {
            int syk = k ? sy1 : sy0; int* D = k ? row1 : row0; int* cached = k ? &cached1 : &cached0;
            if (*cached == syk) continue;
            const uint8_t* S = src + (size_t)syk * sstride;
            int dx = 0;
            for (; dx < xmax; dx++) { int sx = xofs[dx]; D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1]; }
            for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
            *cached = syk;
        }
        uint8_t* d = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            d[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(row0); free(row1);
}

/* ------------------------------------------------------------------------------------------ */
/* cv::GaussianBlur(Size(7,7), 2, 2, BORDER_REFLECT_101) on 8U -- OpenCV 2.4.8 scalar path:
 * getGaussianKernel(7,2,CV_32F) -> createSeparableLinearFilter with bits=8 fixed-point kernels,
 * SymmColumnFilter<FixedPtCastEx<int,uchar>>: (sum + 2^15) >> 16.  Call site :1086.            */
static inline int reflect101(int p, int len) { if (p < 0) p = -p; if (p >= len) p = 2 * len - 2 - p; return p; }

static void gauss7_kernel_q8(int k[7])
{
    /* getGaussianKernel: float taps exp(-x^2/(2 s^2)) normalised in double, stored float */
    float cf[7]; double sum = 0;
    double scale2X = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) { double x = i - 3.0; double t = exp(scale2X * x * x); cf[i] = (float)t; sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); k[i] = cv_round((double)cf[i] * 256.0); }
}

void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    int K[7]; gauss7_kernel_q8(K);           /* == {18,34,49,55,49,34,18} (checked in tests) */
    int* tmp = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = 0; k < 7; k++) acc += K[k] * s[reflect101(x + k - 3, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = 0; k < 7; k++) acc += K[k] * tmp[(size_t)reflect101(y + k - 3, h) * w + x];
            int v = (acc + (1 << 15)) >> 16;
            d[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* cv::FAST(img, kps, threshold, nonmaxSuppression) -- OpenCV 2.4.8 features2d/fast.cpp FAST_t<16>
 * + fast_score.cpp cornerScore<16>.  Call sites :809, :814.                                    */
static const int fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) a = a < d[k + j] ? a : d[k + j];
        int t = a < d[k] ? a : d[k];       if (t > a0) a0 = t;
        t = a < d[k + 9] ? a : d[k + 9];   if (t > a0) a0 = t;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; j++) b = b > d[k + j] ? b : d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) b = b > d[k + j] ? b : d[k + j];
        int t = b > d[k] ? b : d[k];       if (t < b0) b0 = t;
        t = b > d[k + 9] ? b : d[k + 9];   if (t < b0) b0 = t;
    }
    return -b0 - 1;
}

int orc_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, int nms,
                 OrcKeyPoint* out, int cap)
{
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = fast_dx[k] + fast_dy[k] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    if (w < 7 || h < 7) return 0;
    /* full-size score plane (zero = "not a corner"), then NMS; equivalent to the 3-row ring buffer */
    uint8_t* score = (uint8_t*)calloc((size_t)w * h, 1);
    uint8_t* iscorner = (uint8_t*)calloc((size_t)w * h, 1);
    for (int i = 3; i < h - 3; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        for (int j = 3; j < w - 3; j++, ptr++) {
            int v = ptr[0];
            int found = 0;
            /* darker run: x < v - threshold ; brighter run: x > v + threshold ; 9 contiguous of 16 */
            int vt = v - threshold, count = 0;
            for (int k = 0; k < N; k++) { int x = ptr[pixel[k]]; if (x < vt) { if (++count > K) { found = 1; break; } } else count = 0; }
            if (!found) {
                vt = v + threshold; count = 0;
                for (int k = 0; k < N; k++) { int x = ptr[pixel[k]]; if (x > vt) { if (++count > K) { found = 1; break; } } else count = 0; }
            }
            if (found) {
                iscorner[(size_t)i * w + j] = 1;
                if (nms) score[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
            }
        }
    }
    int n = 0;
    for (int i = 3; i < h - 3; i++) {
        for (int j = 3; j < w - 3; j++) {
            if (!iscorner[(size_t)i * w + j]) continue;
            int s = score[(size_t)i * w + j];
            int keep = 1;
            if (nms) {
                const uint8_t* p = score + (size_t)(i - 1) * w, * c = score + (size_t)i * w, * q = score + (size_t)(i + 1) * w;
                keep = s > p[j - 1] && s > p[j] && s > p[j + 1] && s > c[j - 1] && s > c[j + 1] &&
                       s > q[j - 1] && s > q[j] && s > q[j + 1];
            }
            if (keep) {
                if (n < cap) {
                    OrcKeyPoint kp = {(float)j, (float)i, 7.f, -1.f, (float)s, 0, -1};
                    out[n] = kp;
                }
                n++;
            }
        }
    }
    free(score); free(iscorner);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::fastAtan2 -- OpenCV 2.4.8 core/mathfuncs.cpp (polynomial version).  Call site :103.      */
float orc_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* IC_Angle (:77-104).  (cx,cy) = cvRound(pt) */
static float ic_angle(const uint8_t* img, int stride, int cx, int cy, const int* umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)cy * stride + cx;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

float orc_ic_angle(const uint8_t* img, int stride, int cx, int cy)
{
    static const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    return ic_angle(img, stride, cx, cy, umax);
}

/* ------------------------------------------------------------------------------------------ */
/* sin/cos for descriptor steering.  The reference calls libm cos/sin on a float (:113), whose
 * last-ulp behaviour depends on the platform libm.  The oracle DEFINES them as: double-precision
 * Cody-Waite reduction by pi/2 + Taylor polynomials evaluated with explicit fma, rounded to float.
 * The HIP product implements the same operation sequence, so results are bit-identical.         */
void orc_sincosf(float xf, float* s, float* c)
{
    static const double TWO_OVER_PI = 0.63661977236758134308;
    static const double PIO2_HI = 1.57079632679489655800e+00;   /* 0x3FF921FB54442D18 */
    static const double PIO2_LO = 6.12323399573676603587e-17;   /* pi/2 - PIO2_HI    */
    double x = (double)xf;
    double q = rint(x * TWO_OVER_PI);
    double r = fma(-q, PIO2_HI, x);
    r = fma(-q, PIO2_LO, r);
    double r2 = r * r;
    /* sin(r) = r + r^3 * P(r2),  cos(r) = 1 + r2 * Q(r2) ; Taylor to r^17 / r^16 */
    double ps = 2.81145725434552076320e-15;                     /*  1/17! */
    ps = fma(ps, r2, -7.64716373181981647590e-13);              /* -1/15! */
    ps = fma(ps, r2, 1.60590438368216145994e-10);               /*  1/13! */
    ps = fma(ps, r2, -2.50521083854417187751e-08);              /* -1/11! */
    ps = fma(ps, r2, 2.75573192239858906526e-06);               /*  1/9!  */
    ps = fma(ps, r2, -1.98412698412698412698e-04);              /* -1/7!  */
    ps = fma(ps, r2, 8.33333333333333333333e-03);               /*  1/5!  */
    ps = fma(ps, r2, -1.66666666666666666667e-01);              /* -1/3!  */
    double sr = fma(r * r2, ps, r);
    double pc = 4.77947733238738529744e-14;                     /*  1/16! */
    pc = fma(pc, r2, -1.14707455977297247139e-11);              /* -1/14! */
    pc = fma(pc, r2, 2.08767569878680989792e-09);               /*  1/12! */
    pc = fma(pc, r2, -2.75573192239858906526e-07);              /* -1/10! */
    pc = fma(pc, r2, 2.48015873015873015873e-05);               /*  1/8!  */
    pc = fma(pc, r2, -1.38888888888888888889e-03);              /* -1/6!  */
    pc = fma(pc, r2, 4.16666666666666666667e-02);               /*  1/4!  */
    pc = fma(pc, r2, -0.5);
    double cr = fma(r2, pc, 1.0);
    long n = (long)q;
    double sv, cv;
    switch (n & 3) {
        case 0: sv = sr;  cv = cr;  break;
        case 1: sv = cr;  cv = -sr; break;
        case 2: sv = -sr; cv = -cr; break;
        default: sv = -cr; cv = sr; break;
    }
    *s = (float)sv; *c = (float)cv;
}

/* computeOrbDescriptor (:108-147).  (cx,cy) = cvRound(pt) on the blurred level. */
static void orb_descriptor(float angle_deg, const uint8_t* img, int stride, int cx, int cy, uint8_t* desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a, b;
    orc_sincosf(angle, &b, &a);
    const uint8_t* center = img + (size_t)cy * stride + cx;
    const signed char* pat = orc_brief_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int bit = 0; bit < 8; bit++) {
            float x0 = (float)pat[bit * 4 + 0], y0 = (float)pat[bit * 4 + 1];
            float x1 = (float)pat[bit * 4 + 2], y1 = (float)pat[bit * 4 + 3];
            int t0 = center[cv_round(x0 * b + y0 * a) * stride + cv_round(x0 * a - y0 * b)];
            int t1 = center[cv_round(x1 * b + y1 * a) * stride + cv_round(x1 * a - y1 * b)];
            val |= (t0 < t1) << bit;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* ExtractorNode::DivideNode (:481-537) + ORBextractor::DistributeOctTree (:539-763)           */
typedef struct {
    int x0, x1, y0, y1;      /* UL.x, UR.x, UL.y, BL.y */
    int* keys; int nkeys;
    int no_more;
    int prev, next;          /* std::list links */
    int seq;                 /* creation order: DEFINED tie-break replacing the heap-pointer order */
    int alive;
} QNode;

typedef struct { QNode* n; int count, cap; int head, tail, size; } QList;

static int qnew(QList* L)
{
    if (L->count == L->cap) { L->cap = L->cap ? L->cap * 2 : 256; L->n = (QNode*)realloc(L->n, sizeof(QNode) * L->cap); }
    QNode* q = &L->n[L->count]; memset(q, 0, sizeof(*q));
    q->prev = q->next = -1; q->seq = L->count; q->alive = 0;
    return L->count++;
}
static void qpush_front(QList* L, int id)
{ QNode* q = &L->n[id]; q->prev = -1; q->next = L->head; if (L->head >= 0) L->n[L->head].prev = id; L->head = id; if (L->tail < 0) L->tail = id; L->size++; q->alive = 1; }
static void qpush_back(QList* L, int id)
{ QNode* q = &L->n[id]; q->next = -1; q->prev = L->tail; if (L->tail >= 0) L->n[L->tail].next = id; L->tail = id; if (L->head < 0) L->head = id; L->size++; q->alive = 1; }
static int qerase(QList* L, int id)     /* returns next */
{
    QNode* q = &L->n[id]; int nx = q->next;
    if (q->prev >= 0) L->n[q->prev].next = q->next; else L->head = q->next;
    if (q->next >= 0) L->n[q->next].prev = q->prev; else L->tail = q->prev;
    L->size--; q->alive = 0; free(q->keys); q->keys = NULL;
    return nx;
}

/* divides node `id` into 4 freshly created nodes (not yet in the list); returns their ids */
static void qdivide(QList* L, int id, const OrcKeyPoint* kp, int ch[4])
{
    for (int c = 0; c < 4; c++) ch[c] = qnew(L);
    QNode* p = &L->n[id];
    const int halfX = (int)ceilf((float)(p->x1 - p->x0) / 2);
    const int halfY = (int)ceilf((float)(p->y1 - p->y0) / 2);
    QNode* n1 = &L->n[ch[0]], * n2 = &L->n[ch[1]], * n3 = &L->n[ch[2]], * n4 = &L->n[ch[3]];
    n1->x0 = p->x0; n1->x1 = p->x0 + halfX; n1->y0 = p->y0; n1->y1 = p->y0 + halfY;
    n2->x0 = p->x0 + halfX; n2->x1 = p->x1; n2->y0 = p->y0; n2->y1 = p->y0 + halfY;
    n3->x0 = p->x0; n3->x1 = p->x0 + halfX; n3->y0 = p->y0 + halfY; n3->y1 = p->y1;
    n4->x0 = p->x0 + halfX; n4->x1 = p->x1; n4->y0 = p->y0 + halfY; n4->y1 = p->y1;
    for (int c = 0; c < 4; c++) { L->n[ch[c]].keys = (int*)malloc(sizeof(int) * (p->nkeys > 0 ? p->nkeys : 1)); L->n[ch[c]].nkeys = 0; }
    for (int i = 0; i < p->nkeys; i++) {
        const OrcKeyPoint* k = &kp[p->keys[i]];
        QNode* t;
        if (k->x < (float)n1->x1) t = (k->y < (float)n1->y1) ? n1 : n3;
        else t = (k->y < (float)n1->y1) ? n2 : n4;
        t->keys[t->nkeys++] = p->keys[i];
    }
    for (int c = 0; c < 4; c++) if (L->n[ch[c]].nkeys == 1) L->n[ch[c]].no_more = 1;
}

typedef struct { int size, seq, id; } SizeNode;
static int cmp_sizenode(const void* a, const void* b)
{
    const SizeNode* x = (const SizeNode*)a, * y = (const SizeNode*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

int orc_distribute_octree(const OrcKeyPoint* in, int n_in, int minX, int maxX, int minY, int maxY,
                          int N, OrcKeyPoint* out, int cap)
{
    QList L; memset(&L, 0, sizeof(L)); L.head = L.tail = -1;
    int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;            /* reference divides by zero here; defined as 1 */
    const float hX = (float)(maxX - minX) / nIni;
    int* ini = (int*)malloc(sizeof(int) * nIni);
    for (int i = 0; i < nIni; i++) {
        int id = qnew(&L); QNode* q = &L.n[id];
        q->x0 = (int)(hX * (float)i); q->x1 = (int)(hX * (float)(i + 1));
        q->y0 = 0; q->y1 = maxY - minY;
        q->keys = (int*)malloc(sizeof(int) * (n_in > 0 ? n_in : 1)); q->nkeys = 0;
        qpush_back(&L, id); ini[i] = id;
    }
    for (int i = 0; i < n_in; i++) {
        int b = (int)(in[i].x / hX);
        if (b >= nIni) b = nIni - 1;
        QNode* q = &L.n[ini[b]]; q->keys[q->nkeys++] = i;
    }
    free(ini);
    for (int lit = L.head; lit >= 0;) {
        QNode* q = &L.n[lit];
        if (q->nkeys == 1) { q->no_more = 1; lit = q->next; }
        else if (q->nkeys == 0) lit = qerase(&L, lit);
        else lit = q->next;
    }
    int finish = 0;
    SizeNode* vs = NULL; int nvs = 0, capvs = 0;
#define VS_PUSH(sz, nid) do { if (nvs == capvs) { capvs = capvs ? capvs * 2 : 256; vs = (SizeNode*)realloc(vs, sizeof(SizeNode) * capvs); } \
        vs[nvs].size = (sz); vs[nvs].seq = L.n[nid].seq; vs[nvs].id = (nid); nvs++; } while (0)
    while (!finish) {
        int prevSize = L.size;
        int nToExpand = 0;
        nvs = 0;
        for (int lit = L.head; lit >= 0;) {
            if (L.n[lit].no_more) { lit = L.n[lit].next; continue; }
            int ch[4]; qdivide(&L, lit, in, ch);
            for (int c = 0; c < 4; c++) {
                if (L.n[ch[c]].nkeys > 0) {
                    qpush_front(&L, ch[c]);
                    if (L.n[ch[c]].nkeys > 1) { nToExpand++; VS_PUSH(L.n[ch[c]].nkeys, ch[c]); }
                } else { free(L.n[ch[c]].keys); L.n[ch[c]].keys = NULL; }
            }
            lit = qerase(&L, lit);
        }
        if (L.size >= N || L.size == prevSize) finish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                int nprev = nvs;
                SizeNode* prev = (SizeNode*)malloc(sizeof(SizeNode) * (nprev > 0 ? nprev : 1));
                memcpy(prev, vs, sizeof(SizeNode) * nprev);
                nvs = 0;
                qsort(prev, nprev, sizeof(SizeNode), cmp_sizenode);
                for (int j = nprev - 1; j >= 0; j--) {
                    int ch[4]; qdivide(&L, prev[j].id, in, ch);
                    for (int c = 0; c < 4; c++) {
                        if (L.n[ch[c]].nkeys > 0) {
                            qpush_front(&L, ch[c]);
                            if (L.n[ch[c]].nkeys > 1) VS_PUSH(L.n[ch[c]].nkeys, ch[c]);
                        } else { free(L.n[ch[c]].keys); L.n[ch[c]].keys = NULL; }
                    }
                    qerase(&L, prev[j].id);
                    if (L.size >= N) break;
                }
                free(prev);
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
#undef VS_PUSH
    int n = 0;
    for (int lit = L.head; lit >= 0; lit = L.n[lit].next) {
        QNode* q = &L.n[lit];
        int best = q->keys[0]; float maxr = in[best].response;
        for (int k = 1; k < q->nkeys; k++) if (in[q->keys[k]].response > maxr) { best = q->keys[k]; maxr = in[best].response; }
        if (n < cap) out[n] = in[best];
        n++;
    }
    for (int i = 0; i < L.count; i++) free(L.n[i].keys);
    free(L.n); free(vs);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* ORBextractor::ComputePyramid (:1107-1132), ComputeKeyPointsOctTree (:765-853), operator() (:1043-1105) */
int orc_orb_extract(OrcExtractor* ex, const uint8_t* img, int w, int h, int stride,
                    OrcKeyPoint* kps, uint8_t* desc, int cap)
{
    free_state(ex);
    if (!img || w <= 0 || h <= 0) return 0;                 /* _image.empty() -> return (:1046) */
    /* pyramid: level sizes from the ORIGINAL size; resize chained from level-1 (:1111-1120).
     * The 19-px reflected halo (copyMakeBorder) is never read by the live path and is not built. */
    for (int l = 0; l < ex->nlevels; l++) {
        float sc = ex->inv_scale[l];
        ex->lw[l] = cv_round((float)w * sc);
        ex->lh[l] = cv_round((float)h * sc);
        ex->pyr[l] = (uint8_t*)malloc((size_t)ex->lw[l] * ex->lh[l]);
        if (l == 0) for (int y = 0; y < h; y++) memcpy(ex->pyr[0] + (size_t)y * w, img + (size_t)y * stride, w);
        else orc_resize_linear_u8(ex->pyr[l - 1], ex->lw[l - 1], ex->lh[l - 1], ex->lw[l - 1], ex->pyr[l], ex->lw[l], ex->lh[l], ex->lw[l]);
    }
    int total = 0;
    OrcKeyPoint** lev = (OrcKeyPoint**)calloc(ex->nlevels, sizeof(OrcKeyPoint*));
    const float W = 30;
    for (int level = 0; level < ex->nlevels; ++level) {
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = ex->lw[level] - EDGE_THRESHOLD + 3;
        const int maxBorderY = ex->lh[level] - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        int ccap = 1024, nc = 0;
        OrcKeyPoint* cand = (OrcKeyPoint*)malloc(sizeof(OrcKeyPoint) * ccap);
        if (nCols > 0 && nRows > 0) {
            const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
            OrcKeyPoint* cell = (OrcKeyPoint*)malloc(sizeof(OrcKeyPoint) * (size_t)(wCell + 6) * (hCell + 6));
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    const uint8_t* sub = ex->pyr[level] + (size_t)(int)iniY * ex->lw[level] + (int)iniX;
                    int cw = (int)maxX - (int)iniX, chh = (int)maxY - (int)iniY;
                    int ccellcap = (wCell + 6) * (hCell + 6);
                    int n = orc_fast9_16(sub, cw, chh, ex->lw[level], ex->ini_th, 1, cell, ccellcap);
                    if (n == 0) n = orc_fast9_16(sub, cw, chh, ex->lw[level], ex->min_th, 1, cell, ccellcap);
                    for (int k = 0; k < n; k++) {
                        cell[k].x += j * wCell; cell[k].y += i * hCell;
                        if (nc == ccap) { ccap *= 2; cand = (OrcKeyPoint*)realloc(cand, sizeof(OrcKeyPoint) * ccap); }
                        cand[nc++] = cell[k];
                    }
                }
            }
            free(cell);
        }
        ex->cand[level] = cand; ex->ncand[level] = nc;
        int ocap = ex->quota[level] + 8 + nc;   /* >= any possible node count */
        lev[level] = (OrcKeyPoint*)malloc(sizeof(OrcKeyPoint) * (ocap > 0 ? ocap : 1));
        int nk = nc > 0 ? orc_distribute_octree(cand, nc, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                                ex->quota[level], lev[level], ocap) : 0;
        const int scaledPatchSize = (int)(PATCH_SIZE * ex->scale[level]);
        for (int i = 0; i < nk; i++) {
            lev[level][i].x += minBorderX; lev[level][i].y += minBorderY;
            lev[level][i].octave = level; lev[level][i].size = (float)scaledPatchSize;
        }
        ex->nkp[level] = nk; total += nk;
    }
    for (int level = 0; level < ex->nlevels; ++level)               /* computeOrientation (:851-852) */
        for (int i = 0; i < ex->nkp[level]; i++)
            lev[level][i].angle = ic_angle(ex->pyr[level], ex->lw[level], cv_round(lev[level][i].x),
                                           cv_round(lev[level][i].y), ex->umax);
    int ret = total;
    if (total > cap) ret = -1;
    else {
        int offset = 0;
        for (int level = 0; level < ex->nlevels; ++level) {
            int nk = ex->nkp[level];
            if (nk == 0) continue;
            ex->blur[level] = (uint8_t*)malloc((size_t)ex->lw[level] * ex->lh[level]);
            orc_gaussian_blur7_u8(ex->pyr[level], ex->lw[level], ex->lh[level], ex->lw[level], ex->blur[level], ex->lw[level]);
            for (int i = 0; i < nk; i++)
                orb_descriptor(lev[level][i].angle, ex->blur[level], ex->lw[level], cv_round(lev[level][i].x),
                               cv_round(lev[level][i].y), desc + (size_t)(offset + i) * 32);
            if (level != 0) {
                float scale = ex->scale[level];
                for (int i = 0; i < nk; i++) { lev[level][i].x *= scale; lev[level][i].y *= scale; }
            }
            memcpy(kps + offset, lev[level], sizeof(OrcKeyPoint) * nk);
            offset += nk;
        }
    }
    for (int level = 0; level < ex->nlevels; ++level) free(lev[level]);
    free(lev);
    return ret;
}
