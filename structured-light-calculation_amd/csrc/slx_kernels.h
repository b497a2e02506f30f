// slx_kernels.h -- kernel parameter block shared by the HIP kernels and the C-ABI host code.
#ifndef SLX_KERNELS_H
#define SLX_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#include "slx.h"

#define SLX_MAX_PHASE_PLANES (SLX_MAX_FREQ * SLX_MAX_STEPS)
#define SLX_MAX_GRAY_PLANES (2 * SLX_MAX_GRAY_BITS)

// Pixels one lane owns per step: one dword of every 8-bit input plane.
#define SLX_QUAD 4
#define SLX_MAX_TIERS 4

struct SlxKParams {
    // inputs: plane p of frame-set s starts at plane[p] + s * set_stride
    const uint8_t *phase[SLX_MAX_PHASE_PLANES];
    const uint8_t *gray[SLX_MAX_GRAY_PLANES];
    size_t phase_set_stride, gray_set_stride;   // bytes
    size_t row_stride;                          // bytes between image rows of an input plane
    // outputs, dense [set][plane][H][W]; null = not produced
    double *z, *x, *y, *U, *pix, *gray_out;
    int32_t *k;
    uint8_t *mask;
    size_t out_set_stride;                      // pixels (= H*W)
    const int16_t *lut;                         // lut[gray] = bin
    int width, height, row_offset;
    unsigned quads_per_row;                     // ceil(W / 4)
    unsigned n_quads;                           // quads_per_row * H
    int aligned;                                // dword loads / 16-byte stores are legal
    int n_freq, n_steps;
    int period[SLX_MAX_FREQ];
    int gray_bits, gray_stripe;
    float wy[SLX_MAX_STEPS], wx[SLX_MAX_STEPS], wscale;   // x1 weights
    double fov_min, fov_max;
    // calibration scalars of R/CCalculation.cpp:151-164: cC = ((u-cx)*fv)*P00 + ((v-cy)*fu)*P01 + K1
    double cx, cy, fu, fv, P00, P01, K1, P20, P21, K2, cA, cB;
    // ---- fast path (slx_strip_kernel) ----
    // The planes of a group are equally spaced (slx_strip_eligible): plane k of the phase group starts phase_first + k *
    // phase_step bytes after plane_base, Gray plane k (when the Gray planes ride the ring) k * gray_step bytes after gray[0] -- two running scalar offsets in the kernel
    // instead of one register per plane.
    const uint8_t *plane_base;                  // lowest plane address: the buffer descriptor's base
    unsigned phase_first, phase_step;
    unsigned gray_step;                         // only when the Gray planes ride the ring
    int dma_imm;                                // planes are >= 256 bytes apart: the DMAs of a chunk share one M0 and step by the immediate offset
    double inv_period[SLX_MAX_FREQ];            // RN(1/T_f)
    double half_biased[SLX_MAX_FREQ];           // 0.5 + 2^-30/T_f
    int std_gray;                               // lut is the reflected Gray code: bin = prefix-xor(gray)
    unsigned interleave;                        // rows laid end to end so that their quads fill whole waves: 64 / gcd(quads_per_row, 64)
    unsigned chunks_per_group;                  // interleave * quads_per_row / 64 (64-quad chunks of a row group)
    // Work items come in up to SLX_MAX_TIERS tiers, dispatched one after the other: tier t covers the rows from tier_row0[t]
    // up to the next tier's first row of EVERY frame-set, in items of tier_rows[t] rows per lane.  Long items first (they
    // amortise the item start-up), ever shorter ones towards the end of the launch, where the chip drains for about one
    // item's lifetime.  A tier starts on a workgroup boundary: workgroups [tier_first_wg[t], tier_first_wg[t] + tier_wgs[t]).
    unsigned n_tiers;
    unsigned tier_rows[SLX_MAX_TIERS];          // rows one lane walks per work item
    unsigned tier_row0[SLX_MAX_TIERS];          // first image row of the tier
    unsigned tier_items_per_set[SLX_MAX_TIERS]; // row groups of the tier * chunks_per_group
    unsigned tier_items[SLX_MAX_TIERS];         // tier_items_per_set * frame-sets of the launch
    unsigned tier_first_wg[SLX_MAX_TIERS], tier_wgs[SLX_MAX_TIERS];
    int fast_arith;                             // generic kernel: use the bit-identical cheap unwrap / in-range division (host-checked)
    int plain_order;                            // Gray-mask strip kernel: items in plain order instead of XCD-grouped (slx_set_tuning, A/B only)
    // ---- stream kernel (slx_stream_kernel): resident waves take short items from per-column queues
    unsigned *sq_counters;                      // device: one word per queue, 32 words (128 B) apart; null = no stream kernel for this context
    unsigned sq_queues, sq_m;                   // queues = chunks_per_group * sq_m; queue q serves chunk column q % chunks_per_group, row groups = q / chunks_per_group (mod sq_m)
    unsigned sq_groups_per_set, sq_groups_total;// row groups (sq_rows * interleave rows each) per frame-set, and in the launch
    unsigned sq_rows;                           // rows per item, >= 2
    unsigned sq_magic;                          // floor(2^32 / sq_groups_per_set) + 1: G / groups_per_set = mulhi(G, magic) for every G of the launch
    unsigned n_cus;                             // compute units of the context's device (0: 256, an unpartitioned MI355X); sizes "one round of items"
    unsigned long long *stamps;                 // diagnostics: 4 words per work item (s_memtime / s_memrealtime at start, end) or null
    unsigned long long stamp_items;             // items the stamp buffer has room for
};

// Kernel variants (slx_set_variant): 0 = automatic (strip kernel when eligible, else the generic kernel with
// the cheap exact arithmetic when its preconditions hold), 1 = generic kernel with the reference's literal
// arithmetic, 2 = strip kernel only, 3 = generic kernel with the cheap exact arithmetic.
#define SLX_VARIANT_AUTO 0
#define SLX_VARIANT_GENERIC 1
#define SLX_VARIANT_STRIP 2
#define SLX_VARIANT_GENERIC_FAST 3

// True when unwrap_stage<true> and tri_depth<true> are bit-identical to the literal arithmetic for these
// parameters: every period <= 2^14, calibration magnitudes < 2^90.
bool slx_fast_arith_ok(const SlxKParams &kp);

// Point cloud compaction (R/CCalculation.cpp:323-357 order: column outer, row inner) over 64 x 64 tiles.
// counts: device array of slx_cloud_entries(width, height) unsigned; tiles: slx_cloud_tiles(width, height) unsigned;
// xyz: device, 3 doubles per point, or NULL to learn the number of points only.
int slx_cloud_entries(int width, int height);
int slx_cloud_tiles(int width, int height);
int slx_launch_cloud_count(const SlxKParams &kp, const double *z, unsigned *counts, unsigned *tiles, void *stream);
int slx_launch_cloud_write(const SlxKParams &kp, const double *z, const unsigned *counts, const unsigned *tiles, double *xyz, unsigned *total_dev,
                           unsigned *total_host, void *stream);   // total_host: pinned host word or NULL

// The same in ONE launch that reads the depth once (slx_cloud.hip): a workgroup owns 16 columns x rows_per_part rows (part p of column
// group g), counts while the rows come in, learns its offsets by a decoupled look-back over epoch-tagged words, writes its runs.
// words: device, slx_cloud_fused_words(groups, parts) x 8 bytes, zeroed once (and whenever epoch restarts at 0): the ticket counters
// (one per 16 words), then one total and 16 column counts per part.  epoch: launches since the words were zeroed.
#define SLX_CLOUD_THREADS 512        /* threads of a workgroup: 8 waves, a wave per column in the write phase */
#define SLX_CLOUD_COUNTERS 64        /* ticket counters (classes of workgroup indices), 128 bytes apart */
#define SLX_CLOUD_MAX_ROWS 448       /* rows of a part at most (the kernel keeps x, y of its columns in registers: 2 columns x 7 chunks of 64 rows per wave) */
struct SlxCloudFused {
    const double *z;
    double *xyz;                               // null: only the number of points
    unsigned long long *words;
    unsigned *total_dev, *total_host;          // the number of points (device word, pinned host word or null)
    int W, H, groups, parts, rows_per_part;
    unsigned epoch;
    int row_offset;
    double fov_min, fov_max, cx, cy, fu, fv;
    unsigned spin_limit;                       // rounds of polls a look-back wait may last before the workgroup gives up (slx_cloud.hip)
    unsigned *gave_up_host;                    // pinned host word: receives epoch + 1 when a workgroup gave up (the cloud of that launch is void)
    unsigned long long *stamps;                // diagnostics (slx_debug_stamps): 4 words per workgroup, or null
    unsigned long long stamp_items;
};
#define SLX_CLOUD_SPIN_LIMIT 16384u            /* ~15 ms of polling: three orders of magnitude above a wait when all is well */
// Plan of the fused cloud for a W x H map on a device of n_cus compute units (0: 256): false when the shape or the device is
// outside what the kernel's look-back may assume (more parts per column group than workgroups the device keeps resident, more than
// 16 parts) -- the two-launch path serves those.  Host arithmetic only (slx_plan.cpp).
bool slx_cloud_fused_plan(int W, int H, unsigned n_cus, int *groups, int *parts, int *rows_per_part);
size_t slx_cloud_fused_lds_bytes(int rows_per_part);          // dynamic LDS of a workgroup
size_t slx_cloud_fused_words(int groups, int parts);
int slx_launch_cloud_fused(const SlxCloudFused &q, void *stream);

// The point-cloud text of CCalculation::Result formatted on the device (slx_text.hip): n_points packed (x, y, z) triples -> "x y z\n" lines,
// every number as `ostream << double` prints it.  sums: slx_text_workgroups(n_points) words (device); flag: receives `tag` when a number is
// outside the device formatter's range (pinned host word: the text is then void); text: device, 4-byte aligned, room for
// n_points * SLX_TEXT_LINE_MAX (_MSVC) bytes; total_dev / total_host: the length of the text (device word; pinned host word or null).
#define SLX_TEXT_POINTS_PER_WG 1024
#define SLX_TEXT_LINE_MAX 39         /* three numbers of at most 12 characters ("-1.23457e-05"), two blanks, the newline */
#define SLX_TEXT_LINE_MAX_MSVC 43    /* SLX_TEXT_MSVC2013: "-1.23457e-005" is 13 characters, the line ends CR LF */
inline unsigned long long slx_text_workgroups(unsigned long long n_points) { return (n_points + SLX_TEXT_POINTS_PER_WG - 1ull) / SLX_TEXT_POINTS_PER_WG; }
#define SLX_TEXT_MAX_PIECES 16       /* pieces the text leaves in (slx_get_point_cloud_text: piece k crosses PCIe while piece k + 1 is formatted) */
#define SLX_TEXT_BASE_RUN 1024u      /* workgroups per run of `bases` (clouds of more than SLX_TEXT_BASES_FROM workgroups: the emit kernel's prefix stays linear) */
#define SLX_TEXT_BASES_FROM 4096u    /* 4 Mi points */
// bases: device, slx_text_workgroups(n) / SLX_TEXT_BASE_RUN + 1 words, or null for clouds below SLX_TEXT_BASES_FROM workgroups
int slx_launch_text_lengths(const double *xyz, const unsigned *n_dev, unsigned long long max_points, unsigned *sums, unsigned *flag, unsigned tag, unsigned pieces,
                            unsigned long long *offsets_host, unsigned long long *total_dev, int msvc, unsigned long long *bases, void *stream);
int slx_launch_text_piece(const double *xyz, unsigned long long n_points, const unsigned *sums, unsigned char *text, unsigned long long *total_dev, unsigned wg_base,
                          unsigned n_wgs, int msvc, const unsigned long long *bases, void *stream);
int slx_launch_text(const double *xyz, unsigned long long n_points, unsigned *sums, unsigned *flag, unsigned tag, unsigned char *text,
                    unsigned long long *total_dev, unsigned long long *total_host, int msvc, unsigned long long *bases, void *stream);   // msvc: enum slx_text_dialect

// Dynamic-frame tracker (slx_track.hip).  Device pointers; 0 or a hipError_t.
// prevW / prevB / raw non-null: also raw = the deltaP selection between the previous frame's strips and the new ones
// (fused into the strip kernel for the 21-pixel window, a second launch otherwise).
int slx_launch_strip_regression(const uint8_t *cam, size_t stride, int W, int H, int win, float *stripW, float *stripB, void *stream,
                                const float *prevW = nullptr, const float *prevB = nullptr, float *raw = nullptr);
int slx_launch_delta_p(const float *W0, const float *B0, const float *W1, const float *B1, size_t n, float *raw, void *stream);
int slx_launch_track_update(const SlxKParams &kp, const float *raw, float *deltaP, double *U, double *z, double *x, double *y, double *deltaZ,
                            void *stream);
// One dynamic frame in one launch (window 21, images larger than the window): strips, deltaP selection, blur, U, depth, deltaZ.
bool slx_track_fusable(int W, int H, int win);
int slx_launch_track_fused(const SlxKParams &kp, const uint8_t *cam, size_t stride, float *stripW, float *stripB, const float *prevW, const float *prevB,
                           float *deltaP, double *U, double *z, double *x, double *y, double *deltaZ, void *stream);

// Root-side row scatter of the staged depth-map gather (slx_gather.hip): segment k moves n_runs tiles of `run` doubles from
// stage + src + t * src_stride to full + dst + t * dst_stride.  Up to SLX_SCATTER_MAX_SEGS segments (peers) per launch.
#define SLX_SCATTER_MAX_SEGS 16
struct SlxScatterSeg { unsigned long long src, dst, run, n_runs, src_stride, dst_stride; };
struct SlxScatterSegs { int n; SlxScatterSeg seg[SLX_SCATTER_MAX_SEGS]; };
int slx_launch_row_scatter(const SlxScatterSegs &segs, const double *stage, double *full, void *stream);

// True when the strip kernel can run this configuration / these operands.
bool slx_strip_eligible(const SlxKParams &kp, int mode, bool aux);

// Launch-geometry overrides of the strip kernel (slx_set_tuning).  0 = the automatic choice.  They change how the
// work is cut into items, never a result; the process environment is not consulted anywhere.
struct SlxTuning {
    int strip_rows;      // rows per work item, 1..32
    int tail_pct;        // share of every frame-set's rows that goes into the shorter tiers, 1..99 (-1: none, one tier)
    int tail_rows;       // rows per item of the second tier (the following tiers quarter it again)
    int tiers;           // number of tiers, 1..SLX_MAX_TIERS
    int gray_plain;      // 1: Gray planes by ordinary loads instead of the DMA ring
    int strip_waves;     // waves per workgroup, 1..4
    int lds_pad_kib;     // extra LDS per workgroup (lowers the occupancy), 0..128
    int plain_order;     // 1: Gray-mask items in plain order instead of XCD-grouped
    int weave;           // rows woven into one row group, rounded down to a multiple of the smallest legal count, 1..64
    int stream;          // stream kernel: 0 automatic, 1 never, 2 whenever it can run
    int stream_rows;     // its rows per item, 2..16
    int cloud_passes;    // point cloud: 0 automatic (one launch where its plan allows), 1 the fused launch or an error, 2 the two-launch path
    int text_pieces;     // slx_get_point_cloud_text: pieces the text is formatted and copied in, 2..16 (1: no pipeline: cloud, text, copy one after the other); 0 = 2
    int cloud_spin;      // fused point cloud: rounds of polls a look-back wait may last, + 1 (1: none -- every workgroup that needs a word gives up); 0 = SLX_CLOUD_SPIN_LIMIT
};

// Waves per SIMD the VGPR count of a strip-kernel instantiation allows (host-side table, checked against the compiled kernels
// by tests/test_kernel_resources.py).
extern "C" unsigned slx_strip_waves_per_simd(int mode, int n_freq, int gray_ring_bits, int n_steps, int aux);

// Rows per work item the strip kernel's launcher picks (host-side model, exported for the CPU tests).
extern "C" unsigned slx_strip_rows_model(unsigned height, unsigned interleave, unsigned chunks_per_group, unsigned n_sets, unsigned slots_per_cu, unsigned preferred,
                                         unsigned n_cus);

// A planned decode launch (slx_plan.cpp, host arithmetic only): the parameter block with the work-item geometry filled in,
// which kernel family, the grid.  Returns 0, or non-zero when no plan exists (variant 2 on ineligible operands, ...).
struct SlxLaunchPlan {
    SlxKParams kp;
    int mode, aux;
    int strip;               // 1: slx_strip_kernel, 0: slx_fused_kernel
    int stream;              // 1: slx_stream_kernel, 2: slx_gstream_kernel (the reference's own mode) -- kp.sq_* filled in
    int gray_ring_bits;      // strip kernel: 6 when the Gray planes ride the DMA ring, else 0
    unsigned grid_x, grid_y, block;
    size_t lds_bytes;
};
int slx_plan_launch(const SlxKParams &kp, int mode, bool aux, int n_sets, int variant, const SlxTuning *tune, SlxLaunchPlan *plan);

// Launches the fused kernel for `n_sets` frame-sets on `stream` (hipStream_t).
// Returns 0, or a hipError_t value.  `variant` selects a kernel variant, `tune` (may be null) the item geometry.
// Host-side record of a context's queue counters (slx_stream_kernel): the geometry they were last zeroed for and the launches since.
#define SLX_STREAM_MAX_QUEUES 256
struct SlxStreamState {
    unsigned *counters = nullptr;               // device, SLX_STREAM_MAX_QUEUES * 32 words
    unsigned long long key = 0;                 // 0: the counters must be zeroed before the next launch (never used yet, or a launch failed); the kernels leave them at zero
    // what the last launch was (slx_last_kernel): 0 none, 1 slx_fused_kernel, 2 slx_strip_kernel, 3 slx_stream_kernel, 4 slx_decoder_strip_kernel,
    // 5 slx_gstream_kernel
    int last_kind = 0, last_rows = 0, last_weave = 0;
    // ... and which instantiation: the template arguments as rocprofv3 prints them (mode, frequencies, Gray bits on the DMA ring, steps, optional planes)
    int last_mode = 0, last_freq = 0, last_gray_ring_bits = 0, last_steps = 0, last_aux = 0;
};
int slx_launch_fused(const SlxKParams &kp, int mode, bool aux, int n_sets, int variant, void *stream, const SlxTuning *tune = nullptr,
                     SlxStreamState *stream_state = nullptr);

// For the other translation units of the library (slx_comm.cpp): the context's device and its own stream.
extern "C" int slx_internal_device(const slx_ctx *ctx);
extern "C" void *slx_internal_stream(const slx_ctx *ctx);
extern "C" void slx_internal_tile(const slx_ctx *ctx, int *width, int *height);

// Number of distinct variants slx_launch_fused understands.
int slx_num_variants(void);

#endif
