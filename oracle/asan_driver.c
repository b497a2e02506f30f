/* asan_driver.c -- TEST INFRASTRUCTURE (like everything under oracle/): runs the CPU restatement on odd tile shapes with
 * unstructured bytes under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`).
 * Not a parity check: it proves the oracle itself reads and writes inside its buffers and performs no undefined
 * operation (shifts, signed overflow, misaligned access) on any input, so that a difference between the HIP path and
 * the oracle is never the oracle's memory error.  Exit code 0 = no report. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "slx_oracle.h"

static uint64_t s_state = 0x5EED;
static uint32_t rnd(void)
{
    s_state += 0x9E3779B97F4A7C15ull;                 /* SplitMix64 */
    uint64_t z = s_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}

static int run(int mode, int F, int N, int G, int w, int h, int threads, int faithful)
{
    slxo_config c;
    memset(&c, 0, sizeof c);
    c.width = w; c.height = h; c.mode = mode; c.n_freq = F; c.n_steps = N; c.gray_bits = G; c.gray_stripe = G ? (1280 >> G) + 1 : 0;
    c.row_offset = (int)(rnd() % 50); c.faithful_order = faithful;
    for (int f = 0; f < F; f++) c.period[f] = 1 + (int)(rnd() % 2000);
    int16_t *lut = NULL;
    if (G) {
        lut = (int16_t *)malloc(sizeof(int16_t) << G);
        for (int i = 0; i < (1 << G); i++) lut[i] = (int16_t)(rnd() % (1u << G));
        c.gray_lut = lut;
    }
    c.fov_min = 100; c.fov_max = 1000;
    const double cam[9] = {1200, 0, w / 2.0, 0, 1210, h / 2.0, 0, 0, 1}, pro[9] = {2000, 0, 600, 0, 2010, 660, 0, 0, 1};
    const double rot[9] = {.99, -.01, .13, .02, .99, -.1, -.12, .1, .98}, tr[3] = {-31, -9, 39};
    memcpy(c.cam, cam, sizeof cam); memcpy(c.pro, pro, sizeof pro); memcpy(c.rot, rot, sizeof rot); memcpy(c.trans, tr, sizeof tr);
    const int np = mode == SLXO_MODE_GRAY_ONLY ? 0 : F * N, ng = (mode == SLXO_MODE_GRAY_ONLY || mode == SLXO_MODE_GRAY_PHASE || mode == SLXO_MODE_MULTIFREQ_GRAYMASK) ? 2 * G : 0;
    const size_t stride = (size_t)w + rnd() % 5, plane = stride * (size_t)h, hw = (size_t)w * (size_t)h;
    /* every plane in its own exact-size allocation: one byte past any of them is a report */
    const uint8_t **pp = (const uint8_t **)calloc((size_t)(np + ng) + 1, sizeof *pp);
    for (int i = 0; i < np + ng; i++) {
        uint8_t *p = (uint8_t *)malloc(plane - (stride - (size_t)w));      /* the last row has no padding */
        for (size_t j = 0; j < plane - (stride - (size_t)w); j++) p[j] = (uint8_t)rnd();
        pp[i] = p;
    }
    slxo_outputs o;
    memset(&o, 0, sizeof o);
    if (mode >= SLXO_MODE_GRAY_PHASE) { o.z = malloc(hw * 8); o.x = malloc(hw * 8); o.y = malloc(hw * 8); o.U = malloc(hw * 8); o.mask = malloc(hw); }
    if (np) o.pix = malloc(hw * 8 * (size_t)F);
    if (ng) o.gray = malloc(hw * 8);
    if (mode >= SLXO_MODE_MULTIFREQ && F > 1) o.k = malloc(hw * 4 * (size_t)(F - 1));
    const int rc = threads > 1 ? slxo_pipeline_mt(&c, pp, pp + np, stride, &o, threads) : slxo_pipeline(&c, pp, pp + np, stride, &o);
    if (rc == 0 && o.z) {
        double *xyz = malloc(hw * 24 + 8);
        const size_t n = slxo_point_cloud(&c, o.z, xyz);
        if (n > hw) { fprintf(stderr, "point cloud larger than the image\n"); return 1; }
        free(xyz);
    }
    for (int i = 0; i < np + ng; i++) free((void *)pp[i]);
    free(pp); free(lut); free(o.z); free(o.x); free(o.y); free(o.U); free(o.mask); free(o.pix); free(o.gray); free(o.k);
    return rc < 0 ? 2 : 0;
}

int main(void)
{
    const int shapes[][2] = {{1, 1}, {2, 3}, {17, 5}, {64, 48}, {130, 33}, {251, 9}, {5, 130}};
    int bad = 0;
    for (size_t s = 0; s < sizeof shapes / sizeof shapes[0]; s++) {
        const int w = shapes[s][0], h = shapes[s][1];
        for (int t = 1; t <= 3; t += 2) {
            bad |= run(SLXO_MODE_PHASE_ONLY, 1, 4, 0, w, h, t, 0);
            bad |= run(SLXO_MODE_PHASE_ONLY, 1, 3 + (int)(rnd() % 14), 0, w, h, t, 0);
            bad |= run(SLXO_MODE_GRAY_ONLY, 0, 4, 1 + (int)(rnd() % 16), w, h, t, 0);
            bad |= run(SLXO_MODE_GRAY_PHASE, 1, 4, 6, w, h, t, 1);
            bad |= run(SLXO_MODE_MULTIFREQ, 3, 4, 0, w, h, t, (int)(rnd() & 1));
            bad |= run(SLXO_MODE_MULTIFREQ, 4, 8, 0, w, h, t, 0);
            bad |= run(SLXO_MODE_MULTIFREQ_GRAYMASK, 3, 4, 6, w, h, t, 0);
            bad |= run(SLXO_MODE_MULTIFREQ_GRAYMASK, 2, 5, 10, w, h, t, 1);
        }
    }
    /* the tracker and its blur on a few shapes */
    /* ... including images the window does not fit in (fewer rows or columns than the window: strips of zeros, no access
     * outside the image -- tools/fuzz_track.py found the reference's out-of-image reads restated here) and other windows */
    for (int k = 0; k < 40; k++) {
        const int w = k < 6 ? 23 + (int)(rnd() % 80) : 1 + (int)(rnd() % 70), h = k < 6 ? 22 + (int)(rnd() % 60) : 1 + (int)(rnd() % 50);
        const int win = k < 6 ? 21 : 1 + 2 * (int)(rnd() % 16);
        uint8_t *cam = malloc((size_t)w * h);
        for (int i = 0; i < w * h; i++) cam[i] = (uint8_t)rnd();
        float *W0 = calloc((size_t)w * h, 4), *B0 = calloc((size_t)w * h, 4), *W1 = calloc((size_t)w * h, 4), *B1 = calloc((size_t)w * h, 4), *dP = calloc((size_t)w * h, 4);
        slxo_strip_regression(cam, (size_t)w, w, h, win, W0, B0);
        for (int i = 0; i < w * h; i++) cam[i] = (uint8_t)rnd();
        slxo_strip_regression(cam, (size_t)w, w, h, win, W1, B1);
        slxo_delta_p(W0, B0, W1, B1, w, h, dP);
        free(cam); free(W0); free(B0); free(W1); free(B1); free(dP);
    }
    if (bad) { fprintf(stderr, "asan_driver: a pipeline call failed (%d)\n", bad); return 1; }
    printf("oracle asan_driver ok\n");
    return 0;
}
