/*
 * slx_oracle.c -- CPU restatement of the DynaFrame static depth path.
 *
 * TEST INFRASTRUCTURE ONLY (see slx_oracle.h): the checker for the HIP path
 * and the timed CPU baseline of bench.py.  Never linked into the product.
 * PARITY STATUS: "parity unpinned" (no reference tests / golden vectors exist;
 * the reference is unbuildable here without OpenCV 2.4.9).
 *
 * R/ = /root/reference/DynaFrame/DynaFrame/.  Plain C, flat row-major arrays,
 * one pass per reference stage so that intermediate planes exist exactly where
 * the reference materialises them (that is what makes it a fair CPU "port").
 */
#include "slx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ a2 --- */
/* OpenCV 2.4.9 modules/core/src/mathfuncs.cpp, cv::fastAtan2 (cvFastArctan is
 * a thin C wrapper around it).  Called at R/CDecodePhase.cpp:67.  The four
 * coefficients are float products of a float literal and (float)(180/CV_PI),
 * evaluated in float (x64: FLT_EVAL_METHOD == 0). */
#define SLXO_PI 3.1415926535897932384626433832795
static const float k_scale = (float)(180.0 / SLXO_PI);

float slxo_fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * k_scale;
    const float p3 = -0.3258083974640975f * k_scale;
    const float p5 = 0.1555786518463281f * k_scale;
    const float p7 = -0.04432655554792128f * k_scale;
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
    if (x < 0)
        a = 180.f - a;
    if (y < 0)
        a = 360.f - a;
    return a;
}

/* R/CDecodePhase.cpp:67-75, the tail shared by a1 and x1: degrees -> pix. */
static inline double phase_tail(float sinValue, float cosValue, int period)
{
    float x = slxo_fast_atan2_deg(sinValue, cosValue);      /* :67 */
    float pix = (x) / (360) * (double)(period);             /* :69 float div, double mul, ->float */
    pix += 0.5;                                             /* :70 double add, ->float */
    if (pix > period) {                                     /* :71 int -> float compare */
        pix -= period;                                      /* :73 */
    }
    return (double)pix;                                     /* :75 */
}

/* ------------------------------------------------------------------ a1 --- */
void slxo_wrapped_phase_4step(const uint8_t *const img[4], size_t stride,
                              int width, int height, int period, double *pix)
{
    for (int i = 0; i < height; i++) {
        for (int j = 0; j < width; j++) {
            float g0 = img[0][(size_t)i * stride + j];       /* :59 */
            float g1 = img[1][(size_t)i * stride + j];       /* :60 */
            float g2 = img[2][(size_t)i * stride + j];       /* :61 */
            float g3 = img[3][(size_t)i * stride + j];       /* :62 */
            float sinValue = (g0 - g2) / 2;                  /* :64 */
            float cosValue = (g1 - g3) / 2;                  /* :65 */
            pix[(size_t)i * width + j] = phase_tail(sinValue, cosValue, period);
        }
    }
}

/* ------------------------------------------------------------------ x1 --- */
/* BUILD-DEFINED.  y = (2/N) sum_k I_k cos(2 pi k/N), x = (2/N) sum_k I_k sin(2 pi k/N),
 * float weights, k ascending, no contraction; for the reference's pattern model
 * g_k = (sin(phi + 2 pi k/N) + 1)*127 this is (127 sin phi, 127 cos phi) as in a1. */
void slxo_nstep_weights(int n_steps, float *wy, float *wx, float *scale)
{
    for (int k = 0; k < n_steps; k++) {
        double a = 2.0 * SLXO_PI * (double)k / (double)n_steps;
        double c = cos(a), s = sin(a);
        if (fabs(c) < 1e-9) c = 0.0;
        if (fabs(s) < 1e-9) s = 0.0;
        wy[k] = (float)c;
        wx[k] = (float)s;
    }
    *scale = 2.0f / (float)n_steps;
}

static void wrapped_phase_generic(const uint8_t *const *img, int n_steps, size_t stride,
                                  int width, int height, int period, double *pix)
{
    float wy[SLXO_MAX_STEPS], wx[SLXO_MAX_STEPS], scale;
    slxo_nstep_weights(n_steps, wy, wx, &scale);
    for (int i = 0; i < height; i++) {
        for (int j = 0; j < width; j++) {
            float sy = 0.0f, sx = 0.0f;
            for (int k = 0; k < n_steps; k++) {
                float g = img[k][(size_t)i * stride + j];
                sy = sy + g * wy[k];
                sx = sx + g * wx[k];
            }
            float sinValue = sy * scale;
            float cosValue = sx * scale;
            pix[(size_t)i * width + j] = phase_tail(sinValue, cosValue, period);
        }
    }
}

void slxo_wrapped_phase_nstep(const uint8_t *const *img, int n_steps, size_t stride,
                              int width, int height, int period, double *pix)
{
    if (n_steps == 4) {
        const uint8_t *four[4] = { img[0], img[1], img[2], img[3] };
        slxo_wrapped_phase_4step(four, stride, width, height, period, pix);
    } else {
        wrapped_phase_generic(img, n_steps, stride, width, height, period, pix);
    }
}

/* test hook: the generic x1 path at any N (used to show N == 4 reduces to a1) */
void slxo_wrapped_phase_generic(const uint8_t *const *img, int n_steps, size_t stride,
                                int width, int height, int period, double *pix)
{
    wrapped_phase_generic(img, n_steps, stride, width, height, period, pix);
}

/* ---------------------------------------------------------------- a3/a4 --- */
int slxo_gray_lut_from_rows(const int *rows_bin_gray, int n_rows, int16_t *lut)
{
    for (int i = 0; i < n_rows; i++) {                       /* R/CDecodeGray.cpp:120-125 */
        int binCode = rows_bin_gray[2 * i], grayCode = rows_bin_gray[2 * i + 1];
        if (grayCode < 0 || grayCode >= n_rows)
            return -1;
        lut[grayCode] = (int16_t)binCode;
    }
    return 0;
}

void slxo_gray_threshold(const uint8_t *pattern, const uint8_t *inverse, size_t stride,
                         int width, int height, uint8_t *bin)
{
    for (int i = 0; i < height; i++) {
        for (int j = 0; j < width; j++) {
            int a = pattern[(size_t)i * stride + j], b = inverse[(size_t)i * stride + j];
            uint8_t d = (uint8_t)(a > b ? a - b : 0);        /* :159 cv saturating u8 subtract */
            bin[(size_t)i * width + j] = d > 0 ? 0xFF : 0;   /* :167-171 */
        }
    }
}

void slxo_gray_count(const uint8_t *const *bin_planes, int bits, const int16_t *lut,
                     int stripe, int width, int height, double *gray)
{
    double pixPeriod = stripe;                               /* :181-185 (int division done by caller) */
    for (int i = 0; i < height; i++) {
        for (int j = 0; j < width; j++) {
            unsigned grayCode = 0;                           /* :192 */
            for (int b = 0; b < bits; b++) {
                if (bin_planes[b][(size_t)i * width + j] == 255)   /* :195 */
                    grayCode += 1u << b;                     /* :197 pair 0 = LSB */
            }
            gray[(size_t)i * width + j] = (double)lut[grayCode] * pixPeriod;  /* :200 */
        }
    }
}

/* ------------------------------------------------------------------ a5 --- */
void slxo_gray_phase_merge(const double *gray, const double *phase, int stripe, int period,
                           int width, int height, double *U)
{
    int vGrayPeriod = stripe, v_pixPeriod = period;          /* :550, :562-563 */
    for (int h = 0; h < height; h++) {
        for (int w = 0; w < width; w++) {
            double grayVal = gray[(size_t)h * width + w];
            double phaseVal = phase[(size_t)h * width + w];
            double ph = phaseVal;
            if ((int)(grayVal / vGrayPeriod) % 2 == 0) {     /* :570 */
                if (phaseVal > (double)v_pixPeriod * 0.75)   /* :572 */
                    ph = phaseVal - v_pixPeriod;
            } else {
                if (phaseVal < (double)v_pixPeriod * 0.25)   /* :579 */
                    ph = phaseVal + v_pixPeriod;
                ph = ph - 0.5 * v_pixPeriod;                 /* :583 */
            }
            U[(size_t)h * width + w] = grayVal + ph;         /* :587 */
        }
    }
}

/* ------------------------------------------------------------------ a6 --- */
void slxo_projection_matrix(const double pro[9], const double rot[9], const double trans[3],
                            double P[12])
{
    double RT[12];                                           /* :140-144, [R T] side by side */
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            RT[r * 4 + c] = rot[r * 3 + c];
        RT[r * 4 + 3] = trans[r];
    }
    for (int r = 0; r < 3; r++) {                            /* :145, k ascending */
        for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++)
                s = s + pro[r * 3 + k] * RT[k * 4 + c];
            P[r * 4 + c] = s;
        }
    }
}

void slxo_calib_tables(const slxo_config *cfg, double *cA, double *cB, double *cC, double *cD)
{
    double P[12];
    slxo_projection_matrix(cfg->pro, cfg->rot, cfg->trans, P);
    const double fu = cfg->cam[0], fv = cfg->cam[4], cx = cfg->cam[2], cy = cfg->cam[5];
    *cA = fu * fv * P[3];                                    /* :151 */
    *cB = fu * fv * P[11];                                   /* :152 */
    if (!cC || !cD)
        return;
    const int W = cfg->width, H = cfg->height;
    for (int uu = 0; uu < W; uu++) {                         /* :155 u outer */
        for (int vv = 0; vv < H; vv++) {
            int u = uu + cfg->col_offset, v = vv + cfg->row_offset;
            cC[(size_t)vv * W + uu] = (u - cx) * fv * P[0] + (v - cy) * fu * P[1] + fu * fv * P[2];   /* :159-161 */
            cD[(size_t)vv * W + uu] = (u - cx) * fv * P[8] + (v - cy) * fu * P[9] + fu * fv * P[10];  /* :162-164 */
        }
    }
}

/* ------------------------------------------------------------------ a7 --- */
static inline void tri_pixel(const slxo_config *cfg, size_t idx, const double *U, const uint8_t *mask,
                             double cA, double cB, const double *cC, const double *cD, double *z)
{
    double zz = 0;
    double Uv = U[idx];
    if (Uv == 0 || (mask && !mask[idx])) {                   /* :678 (z left as-is there; defined 0 here) */
        z[idx] = 0;
        return;
    }
    zz = -(cA - cB * Uv) / (cC[idx] - cD[idx] * Uv);         /* :686-687 */
    if ((zz < cfg->fov_min) || (zz > cfg->fov_max))          /* :701 */
        zz = 0;
    z[idx] = zz;                                             /* :706 */
}

void slxo_triangulate(const slxo_config *cfg, const double *U, const uint8_t *mask,
                      double cA, double cB, const double *cC, const double *cD,
                      double *z, double *x, double *y)
{
    const int W = cfg->width, H = cfg->height;
    if (cfg->faithful_order) {
        for (int u = 0; u < W; u++)                          /* :672 */
            for (int v = 0; v < H; v++)
                tri_pixel(cfg, (size_t)v * W + u, U, mask, cA, cB, cC, cD, z);
    } else {
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++)
                tri_pixel(cfg, (size_t)v * W + u, U, mask, cA, cB, cC, cD, z);
    }
    if (!x && !y)
        return;
    const double fu = cfg->cam[0], fv = cfg->cam[4], cx = cfg->cam[2], cy = cfg->cam[5];
    for (int uu = 0; uu < W; uu++) {                         /* :756 */
        for (int vv = 0; vv < H; vv++) {
            size_t idx = (size_t)vv * W + uu;
            double zz = z[idx];
            double uc = (uu + cfg->col_offset) - cx;         /* :762 */
            double vc = (vv + cfg->row_offset) - cy;         /* :763 */
            if (x) x[idx] = zz * uc / fu;                    /* :766 */
            if (y) y[idx] = zz * vc / fv;                    /* :767 */
        }
    }
}

/* -------------------------------------------------------------- Result --- */
size_t slxo_point_cloud(const slxo_config *cfg, const double *z, double *xyz)
{
    const int W = cfg->width, H = cfg->height;
    const double fu = cfg->cam[0], fv = cfg->cam[4], cx = cfg->cam[2], cy = cfg->cam[5];
    size_t n = 0;
    for (int u = 0; u < W; u++) {                            /* :335 */
        for (int v = 0; v < H; v++) {
            double valZ = z[(size_t)v * W + u];              /* :340 */
            if ((valZ < cfg->fov_min) || (valZ > cfg->fov_max))   /* :341 */
                continue;
            double uc = (u + cfg->col_offset) - cx, vc = (v + cfg->row_offset) - cy;   /* :762-763 */
            xyz[3 * n + 0] = valZ * uc / fu;                 /* :766, written at :347 */
            xyz[3 * n + 1] = valZ * vc / fv;                 /* :767 */
            xyz[3 * n + 2] = valZ;
            n++;
        }
    }
    return n;
}

/* ------------------------------------------------------- dynamic frames --- */
void slxo_strip_regression(const uint8_t *cam, size_t stride, int W, int H, int win, float *stripW, float *stripB)
{
    const int hw = win / 2;
    memset(stripW, 0, sizeof(float) * (size_t)W * H);                                /* :827-828 */
    memset(stripB, 0, sizeof(float) * (size_t)W * H);
    /* An image without an interior (no column or no row a whole window fits around): the reference's loops over w are empty when
     * W <= 2 hw, and when only H <= 2 hw its first loop reads rows past the image (:810, undefined behaviour there).  The strips
     * stay 0 here, which is what the product defines for that case (csrc/slx_track.hip: slx_launch_strip_regression_only). */
    if (win < 1 || H <= 2 * hw || W <= 2 * hw) return;
    float *valSum = (float *)calloc((size_t)W * H, sizeof(float));                 /* :799-801 setTo(0) */
    for (int w = hw; w < W - hw; w++) {                                              /* :802 */
        float sum = 0;
        for (int hc = 0; hc < win; hc++)
            sum += (float)cam[(size_t)hc * stride + w];                              /* :810 */
        valSum[(size_t)hw * W + w] = sum;                                            /* :812 */
    }
    for (int h = hw + 1; h < H - hw; h++)                                            /* :815 */
        for (int w = hw; w < W - hw; w++)
            valSum[(size_t)h * W + w] = valSum[(size_t)(h - 1) * W + w]
                - (float)cam[(size_t)(h - hw - 1) * stride + w]
                + (float)cam[(size_t)(h + hw) * stride + w];                         /* :820-822 */
    for (int h = hw; h < H - hw; h++) {
        for (int w = hw; w < W - hw; w++) {
            float max = valSum[(size_t)h * W + w], maxIdx = 0;                       /* :834-837 */
            float min = max, minIdx = 0;
            for (int i = -hw; i < hw; i++) {                                         /* :838 */
                float value = valSum[(size_t)h * W + w + i];
                if (value > max) { max = value; maxIdx = (float)i; }
                if (value < min) { min = value; minIdx = (float)i; }
            }
            stripB[(size_t)h * W + w] = minIdx;                                      /* :888 */
            stripW[(size_t)h * W + w] = maxIdx;                                      /* :889 */
        }
    }
    free(valSum);
}

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

void slxo_delta_p(const float *W0, const float *B0, const float *W1, const float *B1, int W, int H, float *deltaP)
{
    const size_t n = (size_t)W * H;
    float *tmp = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) {
        float f0W = W0[i], f0B = B0[i], f1W = W1[i], f1B = B1[i];                    /* :602-605 */
        float fBbias = fabsf(f0B - f1B), fWbias = fabsf(f0W - f1W);                  /* :607-608 */
        tmp[i] = (fBbias < fWbias) ? f0B - f1B : f0W - f1W;                          /* :610-617 */
    }
    const double scale = 1. / 9;                                                     /* cv::blur(.., Size(3,3)), :650 */
    for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++) {
            double s = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++)
                    s += (double)tmp[(size_t)reflect101(h + dy, H) * W + reflect101(w + dx, W)];
            deltaP[(size_t)h * W + w] = (float)(s * scale);
        }
    free(tmp);
}

void slxo_track_update(const double *Uprev, const float *deltaP, size_t n, double *U)
{
    for (size_t i = 0; i < n; i++) U[i] = Uprev[i] + deltaP[i];                      /* :656-658 */
}

/* ------------------------------------------------------------------ x2 --- */
/* BUILD-DEFINED (SURVEY.md section 8 a-ext x2): U_1 = pix_1;
 * k_f = (int)floor((U_{f-1} - pix_f)/T_f + 0.5); U_f = pix_f + k_f*T_f (double). */
void slxo_unwrap_multifreq(const double *pix, int n_freq, const int *period,
                           int width, int height, double *U, int32_t *k)
{
    size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; i++) {
        double Uf = pix[i];
        for (int f = 1; f < n_freq; f++) {
            double p = pix[(size_t)f * n + i];
            int kf = (int)floor((Uf - p) / period[f] + 0.5);
            Uf = p + (double)(kf * period[f]);
            if (k)
                k[(size_t)(f - 1) * n + i] = kf;
        }
        U[i] = Uf;
    }
}

/* ------------------------------------------------------------------ x3 --- */
/* BUILD-DEFINED: valid0 = |U - (gray + S/2)| <= S, then a 3-tap horizontal AND
 * (neighbours outside the tile do not veto). */
void slxo_gray_mask(const double *U, const double *gray, int stripe,
                    int width, int height, uint8_t *mask)
{
    uint8_t *v0 = (uint8_t *)malloc((size_t)width);
    for (int h = 0; h < height; h++) {
        const double *Ur = U + (size_t)h * width, *gr = gray + (size_t)h * width;
        for (int w = 0; w < width; w++)
            v0[w] = fabs(Ur[w] - (gr[w] + stripe * 0.5)) <= (double)stripe;
        for (int w = 0; w < width; w++) {
            int ok = v0[w];
            if (w > 0) ok = ok && v0[w - 1];
            if (w + 1 < width) ok = ok && v0[w + 1];
            mask[(size_t)h * width + w] = (uint8_t)ok;
        }
    }
    free(v0);
}

/* ------------------------------------------------------------- pipeline --- */
static int check_config(const slxo_config *c)
{
    if (c->width <= 0 || c->height <= 0) return -1;
    if (c->mode < SLXO_MODE_PHASE_ONLY || c->mode > SLXO_MODE_MULTIFREQ_GRAYMASK) return -2;
    int need_phase = c->mode != SLXO_MODE_GRAY_ONLY;
    int need_gray = c->mode == SLXO_MODE_GRAY_ONLY || c->mode == SLXO_MODE_GRAY_PHASE ||
                    c->mode == SLXO_MODE_MULTIFREQ_GRAYMASK;
    if (need_phase) {
        if (c->n_steps < 3 || c->n_steps > SLXO_MAX_STEPS) return -3;   /* R/CDecodePhase.cpp:122 rejects <=0; 3 is the algebraic minimum */
        if (c->n_freq < 1 || c->n_freq > SLXO_MAX_FREQ) return -4;
        for (int f = 0; f < c->n_freq; f++)
            if (c->period[f] <= 0) return -5;
        if ((c->mode == SLXO_MODE_PHASE_ONLY || c->mode == SLXO_MODE_GRAY_PHASE) && c->n_freq != 1) return -4;
    }
    if (need_gray) {
        if (c->gray_bits <= 0 || c->gray_bits > SLXO_MAX_GRAY_BITS) return -6;  /* R/CDecodeGray.cpp:39 */
        if (c->gray_stripe <= 0 || !c->gray_lut) return -7;
    }
    return 0;
}

int slxo_pipeline(const slxo_config *cfg, const uint8_t *const *phase_planes,
                  const uint8_t *const *gray_planes, size_t stride, slxo_outputs *out)
{
    int rc = check_config(cfg);
    if (rc) return rc;
    const int W = cfg->width, H = cfg->height, F = cfg->n_freq, N = cfg->n_steps, G = cfg->gray_bits;
    const size_t n = (size_t)W * H;
    const int mode = cfg->mode;
    const int has_gray = mode == SLXO_MODE_GRAY_ONLY || mode == SLXO_MODE_GRAY_PHASE ||
                         mode == SLXO_MODE_MULTIFREQ_GRAYMASK;
    const int has_phase = mode != SLXO_MODE_GRAY_ONLY;
    const int has_depth = mode >= SLXO_MODE_GRAY_PHASE;

    double *gray = NULL, *pix = NULL, *U = NULL, *cC = NULL, *cD = NULL, *z = NULL;
    uint8_t *mask = NULL;

    if (has_gray) {                                          /* CDecodeGray::Decode, R/CDecodeGray.cpp:108 */
        uint8_t **bin = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)G);
        for (int b = 0; b < G; b++) {
            bin[b] = (uint8_t *)malloc(n);
            slxo_gray_threshold(gray_planes[2 * b], gray_planes[2 * b + 1], stride, W, H, bin[b]);
        }
        gray = out->gray ? out->gray : (double *)malloc(n * sizeof(double));
        slxo_gray_count((const uint8_t *const *)bin, G, cfg->gray_lut, cfg->gray_stripe, W, H, gray);
        for (int b = 0; b < G; b++) free(bin[b]);
        free(bin);
    }
    if (has_phase) {                                         /* CDecodePhase::Decode, one per frequency */
        pix = out->pix ? out->pix : (double *)malloc(n * sizeof(double) * (size_t)F);
        for (int f = 0; f < F; f++)
            slxo_wrapped_phase_nstep(phase_planes + (size_t)f * N, N, stride, W, H, cfg->period[f],
                                     pix + (size_t)f * n);
    }
    if (has_depth) {
        U = out->U ? out->U : (double *)malloc(n * sizeof(double));
        if (mode == SLXO_MODE_GRAY_PHASE) {
            slxo_gray_phase_merge(gray, pix, cfg->gray_stripe, cfg->period[0], W, H, U);
        } else {
            slxo_unwrap_multifreq(pix, F, cfg->period, W, H, U, out->k);
        }
        if (mode == SLXO_MODE_MULTIFREQ_GRAYMASK) {
            mask = out->mask ? out->mask : (uint8_t *)malloc(n);
            slxo_gray_mask(U, gray, cfg->gray_stripe, W, H, mask);
        } else if (out->mask) {
            memset(out->mask, 1, n);
        }
        double cA, cB;
        cC = (double *)malloc(n * sizeof(double));
        cD = (double *)malloc(n * sizeof(double));
        slxo_calib_tables(cfg, &cA, &cB, cC, cD);
        z = out->z ? out->z : (double *)malloc(n * sizeof(double));
        slxo_triangulate(cfg, U, mask, cA, cB, cC, cD, z, out->x, out->y);
    }

    if (gray && gray != out->gray) free(gray);
    if (pix && pix != out->pix) free(pix);
    if (U && U != out->U) free(U);
    if (mask && mask != out->mask) free(mask);
    if (z && z != out->z) free(z);
    free(cC);
    free(cD);
    return 0;
}

int slxo_pipeline_mt(const slxo_config *cfg, const uint8_t *const *phase_planes,
                     const uint8_t *const *gray_planes, size_t stride, slxo_outputs *out,
                     int threads)
{
    int rc = check_config(cfg);
    if (rc) return rc;
    if (threads < 1) threads = 1;
    if (threads > cfg->height) threads = cfg->height;
    const int W = cfg->width, H = cfg->height, F = cfg->n_freq;
    const int n_phase = cfg->mode == SLXO_MODE_GRAY_ONLY ? 0 : F * cfg->n_steps;
    const int n_gray = (cfg->mode == SLXO_MODE_GRAY_ONLY || cfg->mode == SLXO_MODE_GRAY_PHASE ||
                        cfg->mode == SLXO_MODE_MULTIFREQ_GRAYMASK) ? 2 * cfg->gray_bits : 0;
    const size_t n = (size_t)W * H;
    int status = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static, 1)
#endif
    for (int t = 0; t < threads; t++) {
        int r0 = (int)((long long)H * t / threads), r1 = (int)((long long)H * (t + 1) / threads);
        if (r1 <= r0) continue;
        slxo_config sub = *cfg;
        sub.height = r1 - r0;
        sub.row_offset = cfg->row_offset + r0;
        const uint8_t *pp[SLXO_MAX_FREQ * SLXO_MAX_STEPS];
        const uint8_t *gp[2 * SLXO_MAX_GRAY_BITS];
        for (int i = 0; i < n_phase; i++) pp[i] = phase_planes[i] + (size_t)r0 * stride;
        for (int i = 0; i < n_gray; i++) gp[i] = gray_planes[i] + (size_t)r0 * stride;
        size_t off = (size_t)r0 * W, sub_n = (size_t)sub.height * W;
        /* per-thread scratch for multi-plane outputs, copied back plane by plane */
        slxo_outputs so;
        memset(&so, 0, sizeof so);
        so.z = out->z ? out->z + off : NULL;
        so.x = out->x ? out->x + off : NULL;
        so.y = out->y ? out->y + off : NULL;
        so.U = out->U ? out->U + off : NULL;
        so.gray = out->gray ? out->gray + off : NULL;
        so.mask = out->mask ? out->mask + off : NULL;
        so.pix = out->pix ? (double *)malloc(sub_n * sizeof(double) * (size_t)F) : NULL;
        so.k = (out->k && F > 1) ? (int32_t *)malloc(sub_n * sizeof(int32_t) * (size_t)(F - 1)) : NULL;
        int r = slxo_pipeline(&sub, pp, gp, stride, &so);
        if (r) status = r;
        if (so.pix) {
            for (int f = 0; f < F; f++)
                memcpy(out->pix + (size_t)f * n + off, so.pix + (size_t)f * sub_n, sub_n * sizeof(double));
            free(so.pix);
        }
        if (so.k) {
            for (int f = 0; f + 1 < F; f++)
                memcpy(out->k + (size_t)f * n + off, so.k + (size_t)f * sub_n, sub_n * sizeof(int32_t));
            free(so.k);
        }
    }
    return status;
}
