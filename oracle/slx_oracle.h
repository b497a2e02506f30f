/*
 * slx_oracle.h -- CPU restatement of the DynaFrame static depth path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference (R/ =
 * /root/reference/DynaFrame/DynaFrame/) ships no tests, no golden vectors and
 * cannot be built here (it needs OpenCV 2.4.9, which is absent; writing a
 * stand-in header is not allowed).  The only reference-held data are
 * R/Patterns/vGrayCode.txt and R/Result.yml; the oracle is checked against
 * both (tests/test_oracle_golden.py).  cvFastArctan lives in OpenCV 2.4.9
 * (opencv_core249, pinned by R/opencv_x64.props:8) and is restated here from
 * its published algorithm (modules/core/src/mathfuncs.cpp, cv::fastAtan2).
 *
 * Every function cites the reference file:line whose arithmetic it follows,
 * cast by cast.  Build with -ffp-contract=off (see Makefile): the reference
 * was built with MSVC /fp:precise on x64 = SSE2, no contraction.
 */
#ifndef SLX_ORACLE_H
#define SLX_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLXO_MAX_FREQ 4
#define SLXO_MAX_STEPS 16
#define SLXO_MAX_GRAY_BITS 16

enum {
    SLXO_MODE_PHASE_ONLY = 0,      /* a1 / x1                        */
    SLXO_MODE_GRAY_ONLY = 1,       /* a3 + a4                        */
    SLXO_MODE_GRAY_PHASE = 2,      /* x4: a1+a3+a4+a5+a7 (reference) */
    SLXO_MODE_MULTIFREQ = 3,       /* x2 (+x1) + a7                  */
    SLXO_MODE_MULTIFREQ_GRAYMASK = 4 /* x2 + a3/a4 + x3 + a7         */
};

typedef struct {
    int width, height;         /* camera tile: columns, rows               */
    int row_offset;            /* v of the tile's first row in the full frame */
    int col_offset;            /* u of the tile's first column             */
    int mode;
    int n_freq;                /* F */
    int n_steps;               /* N */
    int period[SLXO_MAX_FREQ]; /* T_f, projector px, coarse -> fine        */
    int gray_bits;             /* G */
    int gray_stripe;           /* S = projector_width / 2^G (int division) */
    const int16_t *gray_lut;   /* lut[gray] = bin, 2^G entries             */
    double fov_min, fov_max;
    double cam[9], pro[9], rot[9], trans[3]; /* CamMat, ProMat, R, T       */
    int faithful_order;        /* 1: walk u-outer/v-inner like the reference */
} slxo_config;

typedef struct {
    double *z, *x, *y, *U;     /* H*W each, may be NULL                    */
    double *pix;               /* F planes of H*W, may be NULL             */
    double *gray;              /* H*W, may be NULL                         */
    int32_t *k;                /* (F-1) planes of H*W, may be NULL         */
    uint8_t *mask;             /* H*W, 1 = valid, may be NULL              */
} slxo_outputs;

/* a2: cv::fastAtan2 / cvFastArctan of OpenCV 2.4.9, degrees in [0,360). */
float slxo_fast_atan2_deg(float y, float x);

/* a1: CDecodePhase::CountResult, R/CDecodePhase.cpp:48-80. */
void slxo_wrapped_phase_4step(const uint8_t *const img[4], size_t stride,
                              int width, int height, int period, double *pix);

/* x1: N-step generalisation; N == 4 takes the a1 path verbatim. */
void slxo_wrapped_phase_nstep(const uint8_t *const *img, int n_steps, size_t stride,
                              int width, int height, int period, double *pix);
void slxo_nstep_weights(int n_steps, float *wy, float *wx, float *scale);

/* R/CDecodeGray.cpp:120-125: fill lut[gray] = bin from "bin gray" rows. */
int slxo_gray_lut_from_rows(const int *rows_bin_gray, int n_rows, int16_t *lut);

/* a3: CDecodeGray::Grey2Bin, R/CDecodeGray.cpp:150-176 -> bin planes 0/0xFF. */
void slxo_gray_threshold(const uint8_t *pattern, const uint8_t *inverse, size_t stride,
                         int width, int height, uint8_t *bin);

/* a4: CDecodeGray::CountResult, R/CDecodeGray.cpp:179-204. */
void slxo_gray_count(const uint8_t *const *bin_planes, int bits, const int16_t *lut,
                     int stripe, int width, int height, double *gray);

/* a5: merge loop of CCalculation::FillFirstProjectorU, R/CCalculation.cpp:561-589. */
void slxo_gray_phase_merge(const double *gray, const double *phase, int stripe, int period,
                           int width, int height, double *U);

/* a6: calibration part of CCalculation::Init, R/CCalculation.cpp:134-166. */
void slxo_projection_matrix(const double pro[9], const double rot[9], const double trans[3],
                            double P[12]);
void slxo_calib_tables(const slxo_config *cfg, double *cA, double *cB, double *cC, double *cD);

/* a7: CCalculation::FillCoordinate, R/CCalculation.cpp:666-785. */
void slxo_triangulate(const slxo_config *cfg, const double *U, const uint8_t *mask,
                      double cA, double cB, const double *cC, const double *cD,
                      double *z, double *x, double *y);

/* CCalculation::Result, R/CCalculation.cpp:323-357, without the file: packed x y z of the depths inside the FOV,
 * u outer / v inner.  xyz: room for 3*width*height doubles.  Returns the number of points. */
size_t slxo_point_cloud(const slxo_config *cfg, const double *z, double *xyz);

/* Dynamic frames (CCalculation::CalculateOther, R/CCalculation.cpp:208-320).
 * StripRegression, R/CCalculation.cpp:789-892: 21-row sliding column sums, then per pixel the offset (in [-win/2, win/2))
 * of the largest (stripW) and smallest (stripB) sum among the horizontal neighbours; 0 outside the interior. */
void slxo_strip_regression(const uint8_t *cam, size_t stride, int width, int height, int win, float *stripW, float *stripB);
/* FillOtherDeltaProU, R/CCalculation.cpp:595-663: deltaP selection, cv::blur 3x3 (OpenCV 2.4.9 boxFilter, restated:
 * normalised, BORDER_REFLECT_101, double column sums scaled by 1./9 -- unpinned), U = Uprev + deltaP. */
void slxo_delta_p(const float *W0, const float *B0, const float *W1, const float *B1, int width, int height, float *deltaP);
void slxo_track_update(const double *Uprev, const float *deltaP, size_t n, double *U);

/* x2: hierarchical temporal unwrap (BUILD-DEFINED, SURVEY.md section 8 a-ext). */
void slxo_unwrap_multifreq(const double *pix, int n_freq, const int *period,
                           int width, int height, double *U, int32_t *k);

/* x3: Gray-code validity mask (BUILD-DEFINED). */
void slxo_gray_mask(const double *U, const double *gray, int stripe,
                    int width, int height, uint8_t *mask);

/* Whole path, one frame-set.  planes: phase planes first (f*N + k), then the
 * 2G Gray planes (2b = pattern, 2b+1 = inverse), each `stride` bytes per row.
 * Returns 0, or a negative value on a bad configuration. */
int slxo_pipeline(const slxo_config *cfg, const uint8_t *const *phase_planes,
                  const uint8_t *const *gray_planes, size_t stride, slxo_outputs *out);

/* Same, rows split over `threads` OpenMP threads (row-major walk). */
int slxo_pipeline_mt(const slxo_config *cfg, const uint8_t *const *phase_planes,
                     const uint8_t *const *gray_planes, size_t stride, slxo_outputs *out,
                     int threads);

#ifdef __cplusplus
}
#endif
#endif
