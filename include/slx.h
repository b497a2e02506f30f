/*
 * slx.h -- C ABI of the MI355X-native DynaFrame static depth path.
 *
 * Frame-in / depth-out surface of elevenface/Structured-Light-Calculation
 * (DynaFrame), re-presented as plain C so that any host loop (C, C++, ctypes,
 * cgo, JNI ...) can bind it.  The reference has no FFI of its own; each entry
 * point below names the reference call it replaces.  R/ =
 * DynaFrame/DynaFrame/ of the reference repository.
 *
 *   reference call (file:line)                         -> C ABI
 *   -------------------------------------------------------------------------
 *   CCalculation::Init            R/CCalculation.cpp:77   -> slx_create
 *   CDecodeGray::SetNumDigit      R/CDecodeGray.cpp:36    -> slx_config.gray_bits
 *   CDecodeGray::SetMatFileName   R/CDecodeGray.cpp:56    -> slx_config.gray_lut
 *   CDecodePhase::SetNumMat       R/CDecodePhase.cpp:119  -> slx_config.n_steps/period
 *   CSensor::LoadDatas(group)     R/CSensorV.cpp:60       -> `group` of slx_set_frame
 *   CDecode{Gray,Phase}::SetMat   R/CDecodeGray.cpp:24,
 *                                 R/CDecodePhase.cpp:107  -> slx_set_frame
 *   CDecode{Gray,Phase}::Decode + merge + FillCoordinate
 *        R/CDecodeGray.cpp:108, R/CDecodePhase.cpp:83,
 *        R/CCalculation.cpp:525-592, :666-785            -> slx_decode / slx_decode_batch
 *   CDecode*::GetResult, m_x/y/zMat, m_ProjectorU
 *        R/CDecodePhase.cpp:99, R/CCalculation.h:29-38    -> slx_get_output / slx_get_depth
 *   m_xMat / m_yMat / m_ProjectorU of a whole batch
 *        R/CCalculation.cpp:756-771, :589                 -> slx_decode_batch_ex (slx_batch_out)
 *   (no counterpart: the reference is one process)        -> slx_comm_*, slx_gather_depth, slx_decode_gather:
 *                                                           one rank per GPU, the final depth-map gather over RCCL
 *   cv::FileStorage (calibration) R/CCalculation.cpp:124  -> slx_read_calibration_yaml
 *   cv::imread (CSensor)          R/CSensorV.cpp:111      -> slx_read_bmp_gray / slx_read_pgm_gray (+ slx::CSensor, csrc/sensor.hpp)
 *   CCalculation::Result (file)   R/CCalculation.cpp:323  -> slx_get_point_cloud_text (formatted on the device), or
 *                                                           slx_get_point_cloud + slx_write_point_cloud_text (on the host)
 *   CSensor::GetCamPicture loop   R/CSensorV.cpp:171      -> slx_pipe_* (pinned host slots, copy/decode overlap)
 *   CCalculation::Result          R/CCalculation.cpp:323  -> slx_get_point_cloud / slx_point_cloud_of_depth (+ slx::CCalculation::Result text writer)
 *   CCalculation::CalculateOther  R/CCalculation.cpp:208  -> slx_track_begin / slx_track_next (+ slx::CCalculation::CalculateOther)
 *   the Mat GetCamPicture returns R/CSensorV.cpp:171-179  -> slx_track_image_buffer (the deep copy lands in the pinned slot directly)
 *   ~CCalculation / ReleaseSpace  R/CCalculation.cpp:30   -> slx_destroy
 *   ErrorHandling(msg)            R/GlobalFunction.cpp:3  -> int status + slx_last_error
 *                                                           (never prints, never blocks)
 *
 * Threading: a context is thread-compatible (one decode in flight per context,
 * like the reference's decoder objects); different contexts are independent.
 * The library needs an AMD GPU (gfx950); there is no CPU fallback: without a
 * device slx_create fails with SLX_ERR_NO_DEVICE.
 */
#ifndef SLX_H
#define SLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLX_VERSION_MAJOR 0
#define SLX_VERSION_MINOR 1

#define SLX_MAX_FREQ 4        /* frequencies of the temporal unwrap            */
#define SLX_MAX_STEPS 16      /* phase-shift steps per frequency               */
#define SLX_MAX_GRAY_BITS 16  /* R/CDecodeGray.cpp:39 accepts 1..16            */

typedef struct slx_ctx slx_ctx;

enum slx_status {
    SLX_OK = 0,
    SLX_ERR_INVALID_ARG = -1,     /* bad configuration / argument (reference: `return false`) */
    SLX_ERR_NOT_CONFIGURED = -2,  /* frame index outside what the mode uses (reference: SetMat before SetNum*) */
    SLX_ERR_MISSING_FRAME = -3,   /* decode before every input plane was set */
    SLX_ERR_NO_DEVICE = -4,       /* no usable HIP device: there is no CPU fallback */
    SLX_ERR_HIP = -5,             /* a HIP runtime call failed, see slx_last_error */
    SLX_ERR_OUT_OF_MEMORY = -6,
    SLX_ERR_NOT_DECODED = -7,     /* output requested before a decode */
    SLX_ERR_UNAVAILABLE = -8      /* output not produced by this mode / not enabled in aux_outputs */
};

/* What one decode computes.  Stage names follow SURVEY.md section 8(a). */
enum slx_mode {
    SLX_MODE_PHASE_ONLY = 0,          /* CDecodePhase alone: pix (f64)                      */
    SLX_MODE_GRAY_ONLY = 1,           /* CDecodeGray alone: gray (f64)                      */
    SLX_MODE_GRAY_PHASE = 2,          /* the reference's CalculateFirst(): Gray + 1 freq    */
    SLX_MODE_MULTIFREQ = 3,           /* F-frequency N-step temporal unwrap -> depth        */
    SLX_MODE_MULTIFREQ_GRAYMASK = 4   /* same + Gray-code validity mask                     */
};

enum slx_mem_kind { SLX_MEM_HOST = 0, SLX_MEM_DEVICE = 1 };

/* Image groups, numbered like CSensor::LoadDatas(groupNum), R/CSensorV.cpp:57-59. */
enum slx_group { SLX_GROUP_GRAY = 0, SLX_GROUP_PHASE = 1 };

enum slx_output {
    SLX_OUT_Z = 0,     /* f64 [H][W]   depth, 0 where invalid (m_zMat)              */
    SLX_OUT_X = 1,     /* f64 [H][W]   (m_xMat)                                     */
    SLX_OUT_Y = 2,     /* f64 [H][W]   (m_yMat)                                     */
    SLX_OUT_U = 3,     /* f64 [H][W]   projector column (m_ProjectorU)              */
    SLX_OUT_PIX = 4,   /* f64 [F][H][W] wrapped phase in projector px per frequency */
    SLX_OUT_GRAY = 5,  /* f64 [H][W]   Gray stripe left edge                        */
    SLX_OUT_K = 6,     /* i32 [F-1][H][W] fringe orders of the temporal unwrap      */
    SLX_OUT_MASK = 7,  /* u8  [H][W]   1 = valid                                    */
    /* dynamic frames (slx_track_*), available after slx_track_begin: */
    SLX_OUT_DELTAZ = 8,  /* f64 [H][W] z of this frame - z of the previous one (m_deltaZ)   */
    SLX_OUT_DELTAP = 9,  /* f32 [H][W] projector-column increment after the 3x3 blur (m_deltaP) */
    SLX_OUT_STRIPW = 10, /* f32 [H][W] offset of the brightest column sum (m_stripW)        */
    SLX_OUT_STRIPB = 11, /* f32 [H][W] offset of the darkest column sum (m_stripB)          */
    SLX_OUT_COUNT = 12
};

typedef struct slx_config {
    int width, height;             /* camera tile, pixels                                       */
    int row_offset;                /* image row v of the tile's first row (row-tile sharding)   */
    int mode;                      /* enum slx_mode                                             */
    int n_freq;                    /* F: 1 for PHASE_ONLY / GRAY_PHASE, 1..4 for MULTIFREQ*     */
    int n_steps;                   /* N: 3..16; the reference's value is 4                      */
    int period[SLX_MAX_FREQ];      /* T_f in projector px, coarse -> fine (m_pixPeroid)         */
    int gray_bits;                 /* G (GRAY_V_NUMDIGIT)                                       */
    int gray_stripe;               /* S = projector width / 2^G, integer division               */
    const int16_t *gray_lut;       /* lut[gray] = bin, 2^G entries; copied by slx_create        */
    double fov_min, fov_max;       /* FOV_MIN_DISTANCE / FOV_MAX_DISTANCE                       */
    double cam[9], pro[9], rot[9], trans[3];  /* CamMat, ProMat, R, T of the calibration file   */
    int device;                    /* HIP device ordinal, or -1 for the current device          */
    unsigned aux_outputs;          /* bit (1u << slx_output) for every extra output to produce  */
} slx_config;

/* Validates the configuration (no device needed): SLX_OK or SLX_ERR_INVALID_ARG,
 * with a message in msg (may be NULL). */
int slx_validate_config(const slx_config *cfg, char *msg, size_t msg_bytes);

int slx_create(const slx_config *cfg, slx_ctx **out);
void slx_destroy(slx_ctx *ctx);

/* Message of the last failure on this context (ctx == NULL: of the last failed slx_create
 * on this thread).  Never NULL. */
const char *slx_last_error(const slx_ctx *ctx);

/* Replaces the Gray table (n = 2^gray_bits entries, lut[gray] = bin).  The reference re-reads
 * its code file on every Decode (R/CDecodeGray.cpp:113-125); this is the hook for that. */
int slx_set_gray_lut(slx_ctx *ctx, const int16_t *lut, size_t n);

/* Hands over input plane `idx` of `group` (GRAY: 2b = pattern, 2b+1 = inverse of bit b,
 * bit 0 = LSB; PHASE: f*N + k).  SLX_MEM_HOST: the bytes are copied to the device before
 * the call returns to the caller's buffer being reusable (deep copy, like pic.copyTo).
 * SLX_MEM_DEVICE: the pointer is borrowed until the next decode has completed.  Frames may come with any row strides;
 * when the planes of a decode do not all share one (the kernels take one), the odd ones are copied to the context's
 * staging pitch on the device at decode time, as SetMat copies every image -- a common stride avoids that copy. */
int slx_set_frame(slx_ctx *ctx, int group, int idx, const uint8_t *data, size_t stride_bytes,
                  int mem_kind);

/* One frame-set: every stage of the mode, fused, on `stream` (a hipStream_t, or NULL for the
 * context's own stream, which is non-blocking: it does not order itself against work the caller has
 * queued on other streams, the legacy default stream included).  Asynchronous; outputs are read with
 * slx_get_output.
 * Ordering between launches of ONE context: every launch (slx_decode, slx_decode_batch*, the tracker, the point cloud) is
 * ordered on the device behind the context's previous launch, whichever streams the two ran on -- also two batch decodes
 * on two caller streams whose buffers have nothing in common.  A context is one in-order queue of work; a host that wants
 * two decodes to overlap uses two contexts (they share nothing).  The library keeps no handle of a caller's stream beyond
 * the call: the stream may be destroyed as soon as the call has returned.
 * Capture into a hipGraph (this call and slx_decode_batch / slx_decode_batch_ex, on a CALLER's stream between hipStreamBeginCapture and
 * hipStreamEndCapture): the launch becomes a kernel node and the context records nothing for it -- the graph orders it.  Call
 * slx_synchronize before the capture begins (a graph cannot depend on work queued outside it: SLX_ERR_INVALID_ARG otherwise), keep
 * slx_enable_timing off, and keep the frames and output buffers of the captured calls alive and in place for every replay.  A captured
 * batch takes the same kernel a plain launch would (the stream kernels leave their queue counters at zero, so a replay starts like
 * any launch) once the context has decoded a batch outside a capture (its counters are allocated then).  The getters
 * below (slx_get_output, slx_get_depth, slx_get_point_cloud*) know nothing of a graph's replays: after a captured decode the CALLER
 * synchronizes the replay stream before calling them -- they read the context's output planes as the last completed replay left them.
 * Worked example and timings: INTEGRATION.md, "A fixed sequence of decodes as a hipGraph". */
int slx_decode(slx_ctx *ctx, void *stream);

/* n_sets frame-sets resident in device memory, one launch.  Plane p of set s starts at
 * base + s*set_stride + p*height*row_stride.  z_out: device, f64 [n_sets][height][width].
 * gray_base may be NULL when the mode has no Gray planes (and phase_base for GRAY_ONLY). */
int slx_decode_batch(slx_ctx *ctx, int n_sets,
                     const uint8_t *phase_base, size_t phase_set_stride,
                     const uint8_t *gray_base, size_t gray_set_stride,
                     size_t row_stride, double *z_out, void *stream);

/* Output planes of slx_decode_batch_ex.  Every plane is device memory.  Plane q of frame-set s of an output with P planes
 * per set (P = 1; F-1 for k) starts `(s*P + q) * plane_stride` elements after the pointer, and the tile's row 0 at that
 * address: plane_stride = 0 means dense (height*width).  A larger plane_stride lets a row tile be decoded straight into its
 * rows of a full-height map (pointer = map + row0*width, plane_stride = full_height*width): the layout the gather below
 * delivers, with no copy in between.  z is required for the depth modes (for PHASE_ONLY / GRAY_ONLY it receives the mode's
 * primary output); the others are optional (NULL = not produced) and are what the reference computes beside z for every
 * frame: x, y (R/CCalculation.cpp:756-771), the projector column U (m_ProjectorU), the fringe orders k of the temporal
 * unwrap and the validity mask of the BUILD-DEFINED modes. */
typedef struct slx_batch_out {
    double *z;
    double *x, *y, *U;
    int32_t *k;
    uint8_t *mask;
    size_t plane_stride;
} slx_batch_out;
int slx_decode_batch_ex(slx_ctx *ctx, int n_sets,
                        const uint8_t *phase_base, size_t phase_set_stride,
                        const uint8_t *gray_base, size_t gray_set_stride,
                        size_t row_stride, const slx_batch_out *out, void *stream);

int slx_synchronize(slx_ctx *ctx);

/* The context's own stream (a hipStream_t), what slx_decode(ctx, NULL) and the tracker run on: for a host that wants to
 * record its own events on it or make other streams wait for it.  Launches on this stream need no completion event of the
 * library's own (a launch on a caller's stream records one: about 2 us between dependent launches). */
int slx_get_stream(slx_ctx *ctx, void **stream);

/* Copies an output of the last slx_decode (waits for it).  dst_bytes must be at least the
 * size listed at enum slx_output. */
int slx_get_output(slx_ctx *ctx, int which, void *dst, size_t dst_bytes, int mem_kind);
int slx_get_depth(slx_ctx *ctx, double *z, int mem_kind);

/* Point cloud of the last slx_decode, the data CCalculation::Result writes (R/CCalculation.cpp:323-357): packed
 * (x, y, z) f64 triples of every pixel whose depth lies in [fov_min, fov_max], in the reference's order (column u outer,
 * row v inner), x = z*(u-cx)/fu, y = z*(v-cy)/fv (R/CCalculation.cpp:756-771).  Compacted on the device.
 * xyz: room for `capacity_points` triples (host or device per mem_kind); *n_points receives the number of valid points
 * (also when it exceeds the capacity, in which case SLX_ERR_INVALID_ARG is returned and nothing is copied). */
int slx_get_point_cloud(slx_ctx *ctx, double *xyz, size_t capacity_points, size_t *n_points, int mem_kind);
/* The same cloud in pinned host memory that the CONTEXT owns: no buffer to size, no call to learn the count first.  *xyz stays
 * valid until the next point-cloud call on this context (or its destruction); the caller only reads it.  What
 * slx::CCalculation::Result formats its text file from. */
int slx_get_point_cloud_view(slx_ctx *ctx, const double **xyz, size_t *n_points);
/* The same cloud as the TEXT CCalculation::Result writes (R/CCalculation.cpp:323-357): "x y z" and a line end per point, every number as
 * `ostream << double` prints it (%g, 6 significant digits) -- formatted ON THE DEVICE (formatting is the cost of that function: 8-10 ms
 * per 1.3 M points on 16 host threads), byte for byte what slx_write_point_cloud_text(_ex) writes in the same dialect; only the text
 * crosses PCIe.  *text: pinned host memory the context owns, *n_bytes long (not NUL-terminated), valid until the next point-cloud call on
 * this context; write it to the file as it is (a BINARY-mode stream: the line ends are already in it).
 * SLX_ERR_UNAVAILABLE when a coordinate lies outside the device formatter's range (0 < |v| < 1e-5, |v| >= 1e15, NaN,
 * infinity): take slx_get_point_cloud_view + slx_write_point_cloud_text_ex for that frame (slx::CCalculation::Result does).
 *
 * WHICH bytes "ostream << double" means depends on the C++ runtime the reference is built with, and this library offers both:
 *   SLX_TEXT_LIBSTDCXX (default)  what that loop writes when compiled on Linux (libstdc++ over glibc's printf): two exponent digits
 *                                 ("5e-05"), '\n'.
 *   SLX_TEXT_MSVC2013             what the reference AS BUILT writes -- DynaFrame.vcxproj targets the MSVC 2013 runtime and opens the file
 *                                 in text mode: at least three exponent digits ("5e-005") and CR LF line ends.  Numbers in exponent
 *                                 notation do occur in a cloud (x of the columns next to cx).  The non-finite spellings of that runtime
 *                                 ("1.#INF", "-1.#IND", "1.#QNAN") come from the host formatter only.
 * Neither dialect is pinned by an output file of the reference (it ships none): both restate the documented behaviour of the runtimes. */
enum slx_text_dialect { SLX_TEXT_LIBSTDCXX = 0, SLX_TEXT_MSVC2013 = 1 };
/* The dialect of slx_get_point_cloud_text / slx_format_points_text on this context (default SLX_TEXT_LIBSTDCXX). */
int slx_set_text_dialect(slx_ctx *ctx, int dialect);
int slx_get_point_cloud_text(slx_ctx *ctx, const char **text, size_t *n_bytes, size_t *n_points);
/* The text of any n_points packed (x, y, z) triples in DEVICE memory (a cloud slx_point_cloud_of_depth left there), same contract. */
int slx_format_points_text(slx_ctx *ctx, const double *xyz_dev, size_t n_points, const char **text, size_t *n_bytes);
/* The same for any depth map of the context's geometry in device memory (height x width f64, contiguous): one plane of
 * slx_decode_batch's output, so that a batch host loop gets CCalculation::Result's data per frame-set without another
 * decode.  Ordered after the context's last launch (whatever stream it ran on); `depth` is borrowed until the call returns. */
int slx_point_cloud_of_depth(slx_ctx *ctx, const double *depth, double *xyz, size_t capacity_points, size_t *n_points, int mem_kind);

/* Dynamic frames, CCalculation::CalculateOther (R/CCalculation.cpp:208-320).  The context must be a depth mode created
 * with SLX_OUT_U in aux_outputs and hold a decoded frame 0.
 * slx_track_begin = StripRegression(0) (R/CCalculation.cpp:203): column-sum extrema of the first dynamic camera image.
 * slx_track_next  = StripRegression(fN) + FillOtherDeltaProU(fN) + FillCoordinate(fN) for the next camera image: updates
 * SLX_OUT_U / Z (/ X / Y when enabled) in place and produces SLX_OUT_DELTAZ, DELTAP, STRIPW, STRIPB.  `window` is
 * RECO_WINDOW_SIZE (21 in the reference; odd, 3..201).  Images: 8-bit, height x width, host (copied) or device (borrowed
 * until the call's work has completed; slx_synchronize). */
int slx_track_begin(slx_ctx *ctx, const uint8_t *image, size_t stride_bytes, int mem_kind, int window);
int slx_track_next(slx_ctx *ctx, const uint8_t *image, size_t stride_bytes, int mem_kind);
/* The pinned buffer (height x width bytes, *stride_bytes = width) the NEXT slx_track_begin / slx_track_next with a host
 * image stages through, for a host that can let its camera SDK or image reader (CSensor::GetCamPicture's deep copy, R/CSensorV.cpp:171-179) write
 * there: passing exactly this pointer and stride back as the SLX_MEM_HOST image skips the library's own copy into it.
 * The call returns once the transfer that last used the buffer (two frames ago) has left it; the pointer is valid for one
 * slx_track_* call and belongs to the context. */
int slx_track_image_buffer(slx_ctx *ctx, uint8_t **buffer, size_t *stride_bytes);

/* k dynamic frames per transfer.  The reference holds all of a run's dynaCam images in memory before it walks them
 * (CSensor::LoadDatas, R/CSensorV.cpp:60-133; the loop R/CCalculation.cpp:222-317), and the strips of frame fN depend on image fN
 * alone -- so k images can ride ONE host-to-device transfer instead of k (each single-image transfer pays its own hand-off to the
 * kernel that waits for it).  Images: n_frames x height x width bytes, rows stride_bytes apart, images image_stride_bytes apart.
 *  slx_track_next_batch   = n_frames x slx_track_next: one transfer (host images), then one launch per frame back to back on
 *                           the context's stream.  Afterwards the context's outputs are those of the LAST frame, exactly as after
 *                           n_frames calls of slx_track_next; deltaz_all (n_frames x height x width f64 in device or host memory,
 *                           deltaz_mem_kind; or NULL) receives every frame's deltaZ plane -- asynchronously on the context's
 *                           stream for a device buffer, complete on return for a host one.
 *  slx_track_stage_frames = only the transfer: the images land in a device slab of the context (*device_images, width bytes per row,
 *                           height x width per image), for a loop that needs every frame's outputs (CCalculation::CalculateOther
 *                           writes a point cloud per frame): slx_track_next(ctx, *device_images + f * height * width, width,
 *                           SLX_MEM_DEVICE) per frame.  The slab stays valid until the second staging call after this one -- or
 *                           until a staging / buffer call asks for more frames than any call before it (the slabs then grow).
 *  slx_track_frames_buffer= the pinned slab the NEXT staging call (or host-fed batch) of up to n_frames images copies from, for a
 *                           producer that can write there directly (passing it back skips the library's own copy). */
#define SLX_TRACK_MAX_BATCH 256
int slx_track_next_batch(slx_ctx *ctx, const uint8_t *images, size_t stride_bytes, size_t image_stride_bytes, int n_frames, int mem_kind,
                         double *deltaz_all, int deltaz_mem_kind);
int slx_track_stage_frames(slx_ctx *ctx, const uint8_t *images, size_t stride_bytes, size_t image_stride_bytes, int n_frames,
                           const uint8_t **device_images);
int slx_track_frames_buffer(slx_ctx *ctx, int n_frames, uint8_t **buffer, size_t *stride_bytes, size_t *image_stride_bytes);

/* Device pointer of an output buffer owned by the context (valid until slx_destroy). */
int slx_output_device_ptr(slx_ctx *ctx, int which, void **ptr);

/* Derived calibration constants, for inspection: P (3x4, row-major), cA, cB
 * (R/CCalculation.cpp:145,151,152). */
int slx_get_calibration(const slx_ctx *ctx, double P[12], double *cA, double *cB);

/* Launch duration of the most recent decode in milliseconds, measured with HIP events recorded
 * on the stream the kernel ran on (waits for it).  Off by default: slx_enable_timing(ctx, 1). */
int slx_enable_timing(slx_ctx *ctx, int on);
int slx_last_decode_ms(slx_ctx *ctx, float *ms);

/* Diagnostics: when set (device buffer of >= 32768 u64 words, or NULL to stop), the first n_words / 4 work items
 * (waves) of the strip kernel -- and the first n_words / 4 workgroups of the tracker's and the fused point cloud's launches -- record
 * s_memtime / s_memrealtime (100 MHz) at entry and exit into words [4*item .. 4*item+3]: first start to last end is the launch as the
 * shader sees it, without the dispatch and completion handling a profiler's kernel interval includes (tools/short_kernels.py). */
int slx_debug_stamps(slx_ctx *ctx, unsigned long long *device_words, size_t n_words);

/* Selects the kernel variant (0 = default); tuning / A-B benchmarking only. */
int slx_set_variant(slx_ctx *ctx, int variant);

/* Launch-geometry overrides of the fast kernel, for tuning / A-B benchmarking and for tests that force the rarely taken
 * item layouts on small tiles.  value 0 restores the automatic choice.  They change how the work is cut into items, never a
 * result.  This entry is the ONLY way to set them: the library does not read the process environment. */
enum slx_tuning_key {
    SLX_TUNE_STRIP_ROWS = 0,   /* rows per work item, 1..32                                             */
    SLX_TUNE_TAIL_PCT = 1,     /* percent of every frame-set's rows cut into shorter items, 1..99; -1 none */
    SLX_TUNE_TAIL_ROWS = 2,    /* rows per item of the second tier, 1..32 (later tiers quarter it)       */
    SLX_TUNE_GRAY_PLAIN = 3,   /* 1: Gray planes by ordinary loads instead of the LDS-DMA ring          */
    SLX_TUNE_STRIP_WAVES = 4,  /* waves per workgroup, 1..4                                             */
    SLX_TUNE_LDS_PAD_KIB = 5,  /* extra LDS per workgroup in KiB (lowers the occupancy), 1..128         */
    SLX_TUNE_PLAIN_ORDER = 6,  /* 1: Gray-mask work items in plain order instead of XCD-grouped         */
    SLX_TUNE_TIERS = 7,        /* tiers of ever shorter work items towards the end of a launch, 1..4    */
    SLX_TUNE_WEAVE = 8,        /* rows woven into one row group (a lane's rows are that far apart), 1..64 */
    SLX_TUNE_STREAM = 9,       /* stream kernels (resident waves, short items from queues): 0 automatic, 1 never, 2 whenever possible */
    SLX_TUNE_STREAM_ROWS = 10, /* their rows per work item, 2..16 (1..16 for the reference's own mode)   */
    SLX_TUNE_CLOUD_PASSES = 11,/* point cloud: 0 automatic, 1 the single fused launch (SLX_ERR_UNAVAILABLE where its plan refuses), 2 the count + write launches */
    SLX_TUNE_CLOUD_SPIN = 12,  /* fused point cloud: rounds of polls a look-back wait may last before the workgroup gives up and the frame is
                                  repeated on the count + write launches, + 1 (1 = no poll at all: tests force the fallback with it)     */
    SLX_TUNE_TEXT_PIECES = 13, /* slx_get_point_cloud_text: pieces the text is formatted and copied in (piece k crosses PCIe while k + 1 is
                                  formatted), 2..16; 1 = cloud, text and copy one after the other                                     */
    SLX_TUNE_COUNT = 14
};
int slx_set_tuning(slx_ctx *ctx, int key, int value);
/* Which kernel the context's last decode launch was -- the instantiation, spelled as rocprofv3's kernel trace prints it -- and how
 * its work was cut: "slx_stream_kernel<3>: resident waves, 2-row items from queues", "slx_gstream_kernel: resident waves, 1-row items from queues"
 * (the reference's own mode), "slx_strip_kernel<3, 3, 0, 4, false>: 16-row
 * items, 8 rows per row group" (<mode, frequencies, Gray bits on the DMA ring, steps, optional planes>), "slx_decoder_strip_kernel<0>:
 * 3-row items, 2 rows per row group", "slx_fused_kernel<3, 3, true, true>" (<mode, frequencies, 4 steps, optional planes>).  For bench
 * lines, profiles, and tests that must know a launch did not silently take another kernel. */
int slx_last_kernel(slx_ctx *ctx, char *buf, size_t buf_bytes);

/* ---- frame ingest pipeline: the live loop around the path -------------------------------------
 * Role of CSensor::GetCamPicture -> CDecode*::SetMat -> Decode in a capture loop (R/CSensorV.cpp:171-179,
 * R/CCalculation.cpp:171-205), for frame-sets that arrive in HOST memory: `slots` pinned host buffers, each holding
 * `sets_per_slot` frame-sets, move through copy-in -> decode -> copy-out on three HIP streams, so the PCIe transfers of
 * one slot overlap the decode of another.  Layout of a slot's input: [set][plane][height][pitch] bytes, planes in the
 * order phase f*N+k, then Gray 2b (pattern), 2b+1 (inverse); pitch = width rounded up to 4.  The result is the mode's
 * primary output (depth; pix for PHASE_ONLY; gray for GRAY_ONLY), [set][height][width] doubles.
 * Slot life cycle: acquire (host fills the pinned input) -> submit -> collect (oldest submitted first; the host reads
 * the pinned result) -> the slot is free again at its next acquire.  acquire fails with SLX_ERR_NOT_CONFIGURED when every
 * slot is submitted or collected-but-not-yet-reused in an order that leaves none free (collect first).
 * One pipe per context; the context's own frames / outputs (slx_set_frame, slx_decode) are not touched.  A pipe is used
 * only while its context lives; it may be DESTROYED before or after it. */
typedef struct slx_pipe slx_pipe;
typedef struct {
    int slots;            /* >= 2 */
    int sets_per_slot;    /* >= 1 frame-sets decoded by one launch */
    int host_result;      /* 1: copy the result back into pinned host memory; 0: leave it on the device */
} slx_pipe_config;
int slx_pipe_create(slx_ctx *ctx, const slx_pipe_config *cfg, slx_pipe **out);
void slx_pipe_destroy(slx_pipe *pipe);
/* Geometry of a slot: planes per frame-set, bytes between rows, planes and frame-sets of the input. */
int slx_pipe_layout(const slx_pipe *pipe, int *n_planes, size_t *pitch, size_t *plane_bytes, size_t *set_bytes);
/* Pinned input buffer of the next free slot. */
int slx_pipe_acquire(slx_pipe *pipe, uint8_t **host_in);
/* Hands the acquired slot to the GPU; returns at once.  n_sets <= sets_per_slot frame-sets are valid in it. */
int slx_pipe_submit(slx_pipe *pipe, int n_sets);
/* Waits for the oldest submitted slot.  host_result (pinned, NULL when host_result == 0), device_result and n_sets may be
 * NULL when not wanted.  The pointers stay valid until that slot is acquired again. */
int slx_pipe_collect(slx_pipe *pipe, const double **host_result, const double **device_result, int *n_sets);
const char *slx_pipe_last_error(const slx_pipe *pipe);

/* ---- multi-GPU: one process per GPU, the final depth-map gather over RCCL / xGMI -------------------------------------
 * The decode is pixel-independent, so ranks exchange nothing while computing; the only
 * collective is the gather of the finished depth maps.  Rank r holds the rows [row0, row0+rows) of the frame-sets
 * [set0, set0+n_sets) of a batch whose full result is f64 [total_sets][height][width]: whole frame-sets (rows == height)
 * or a row tile of every frame-set (the split BASELINE.json's north_star words).  The gather reassembles [set][H][W]
 * on `root` (or on every rank, root = -1): grouped ncclSend / ncclRecv, one message per (peer, frame-set) landing at
 * its row offset of that set, so the root's 7 xGMI links carry traffic at once and no staging copy or transpose follows.
 * The reference has no counterpart (single process); this is the drop-in for a host loop that spreads a batch over a node. */
typedef struct slx_comm slx_comm;
typedef struct slx_shard { int set0, n_sets, row0, rows; } slx_shard;
#define SLX_COMM_ID_BYTES 128
/* Rank 0 makes an id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by any means (a file, MPI, a socket ...). */
int slx_comm_unique_id(void *id, size_t id_bytes);
/* Collective over all `world` ranks: ncclCommInitRank on the context's device.  The comm owns a gather stream; it serves every
 * context of that device (a unique id makes ONE communicator: do not reuse it for a second slx_comm_create). */
int slx_comm_create(slx_ctx *ctx, const void *id, size_t id_bytes, int world, int rank, slx_comm **out);
/* Wraps a communicator the host already has (an ncclComm_t whose device is the context's); it is not destroyed with the wrapper. */
int slx_comm_adopt(slx_ctx *ctx, void *nccl_comm, slx_comm **out);
void slx_comm_destroy(slx_comm *comm);
int slx_comm_info(const slx_comm *comm, int *world, int *rank);
const char *slx_comm_last_error(const slx_comm *comm);   /* comm == NULL: of the last failed create on this thread */
/* Waits for everything queued on the comm's gather stream. */
int slx_comm_synchronize(slx_comm *comm);
/* One gather.  shards[world]: what every rank holds (the same table on every rank).  local: this rank's shard, plane
 * (s - set0) at local + (s - set0)*local_plane_stride doubles (0 = dense rows*width); it may point into `full`
 * (local == full + (set0*height + row0)*width, local_plane_stride == height*width: decoded in place, nothing is copied).
 * full: f64 [total_sets][height][width] on the destination ranks, ignored elsewhere.  Queued on `stream` (NULL: the comm's
 * gather stream); asynchronous. */
int slx_gather_depth(slx_comm *comm, const slx_shard *shards, int height, int width,
                     const double *local, size_t local_plane_stride, double *full, int root, void *stream);
/* The messages one group of the gather consists of for `rank`, without posting them (no GPU needed): the frame-sets
 * [first, first+count) counted within every shard, receives first, then sends, in posting order.  offset is in doubles into
 * `full` for a receive and into this rank's `local` for a send.  *n_out receives the number of messages (also when it exceeds
 * `capacity`).  For inspection and for checking the schedule of a world of N ranks on a machine without N GPUs. */
typedef struct slx_msg { int peer, send; unsigned long long offset, count; } slx_msg;   /* send: 0 receive into `full`, 1 send from `local`, 2 receive into the staging slot */
int slx_gather_plan(const slx_shard *shards, int world, int rank, int height, int width, int first, int count,
                    size_t local_plane_stride, int root, slx_msg *out, int capacity, int *n_out);
/* Two shapes of the same gather (the same bytes arrive in the same places):
 *   SLX_GATHER_IN_PLACE  one message per (peer, frame-set) of a row split, landing at the tile's rows of that set: no staging, no
 *                        second pass over the root's HBM; 7 x 256 = 1 792 messages of 2.3 MB per step for configuration 4 on 8 ranks.
 *   SLX_GATHER_STAGED    (gathers to ONE root; with root = -1 the in-place shape is used) one contiguous message per (peer, chunk)
 *                        into a staging slot of the root -- 224 messages of 18.4 MB for the same step in chunks of 8 -- and a
 *                        row-scatter kernel that moves the tiles to their rows while the next chunk's messages arrive in the
 *                        other slot.  Sending ranks must hold a dense tile stack (slx_decode_gather's scratch is).
 * Whole-frame shards are one message per peer in either shape.  The shape is a property of the communicator; every rank must
 * set the same one.  Default: SLX_GATHER_IN_PLACE. */
enum slx_gather_shape { SLX_GATHER_IN_PLACE = 0, SLX_GATHER_STAGED = 1 };
int slx_comm_set_gather_shape(slx_comm *comm, int shape);
/* The plan of a group for either shape.  Receives with send == 2 land `offset` doubles into the staging slot (of
 * *staging_doubles doubles); scatter_out lists how the slot then goes to the full array: n_runs runs of `run` doubles from
 * slot + src + t * src_stride to full + dst + t * dst_stride.  Counts are returned also when they exceed the capacities. */
typedef struct slx_scatter { unsigned long long src, dst, run, n_runs, src_stride, dst_stride; } slx_scatter;
int slx_gather_plan_ex(const slx_shard *shards, int world, int rank, int height, int width, int first, int count,
                       size_t local_plane_stride, int root, int shape, slx_msg *out, int capacity, int *n_out,
                       slx_scatter *scatter_out, int scatter_capacity, int *n_scatter_out, unsigned long long *staging_doubles);
/* Runs a scatter list of slx_gather_plan_ex on the context's device -- the root-side kernel of the staged shape by itself, for a
 * host that moves the messages with its own transport (MPI, its own RCCL calls) and for tests on a box with one GPU: staging and
 * full are device memory; asynchronous on `stream` (NULL: the context's stream). */
int slx_scatter_rows(slx_ctx *ctx, const slx_scatter *scatter, int n, const double *staging, double *full, void *stream);
/* Decode + gather of this rank's shard, pipelined: the shard's frame-sets are decoded `chunk_sets` at a time on `stream`
 * (NULL: the context's) and every finished chunk is gathered on the comm's stream while the next one decodes.  Inputs as
 * slx_decode_batch (this rank's shards[rank].n_sets frame-sets, tile height shards[rank].rows = the context's height).
 * Destination ranks decode in place into `full`; the others into `scratch` (f64 [n_sets][rows][width], may be NULL on
 * destination ranks).  Every rank must pass the same shards / chunk_sets / root.  Asynchronous: slx_comm_synchronize.
 * An error return on ANY rank (a failed decode of a chunk, a failed RCCL call) leaves the other ranks inside RCCL, waiting for messages
 * that rank will never post -- as with every collective.  The communicator is then unusable: destroy it on every rank (slx_comm_destroy,
 * or ncclCommAbort on an adopted one) and create a new one before the next gather; do not retry on the old communicator. */
int slx_decode_gather(slx_comm *comm, slx_ctx *ctx, const slx_shard *shards, int full_height, int chunk_sets,
                      const uint8_t *phase_base, size_t phase_set_stride,
                      const uint8_t *gray_base, size_t gray_set_stride, size_t row_stride,
                      double *scratch, double *full, int root, void *stream);

/* ---- file formats of a DynaFrame data directory (host only, no GPU needed) ----
 * 8-bit grey pixels of an uncompressed BMP (8-bit paletted or 24/32-bit colour, converted like
 * imread(..., CV_LOAD_IMAGE_GRAYSCALE), R/CSensorV.cpp:111-114), top-down, dense.  pixels == NULL: only the size.
 * SLX_ERR_UNAVAILABLE: missing or unsupported file. */
int slx_read_bmp_gray(const char *path, uint8_t *pixels, size_t capacity, int *rows, int *cols);
/* Same contract for a binary PGM (P5, maxval <= 255), the other 8-bit format cv::imread takes. */
int slx_read_pgm_gray(const char *path, uint8_t *pixels, size_t capacity, int *rows, int *cols);
/* The point-cloud text file of CCalculation::Result (R/CCalculation.cpp:323-357): "x y z\n" per point, every number as
 * `ostream << double` prints it (%g, 6 significant digits) -- the same bytes as that loop compiled with libstdc++ (SLX_TEXT_LIBSTDCXX;
 * the MSVC build's bytes: slx_write_point_cloud_text_ex), formatted by several threads and
 * written without its flush per line (2.27 M points: 0.06 s instead of 3.6 s).  xyz: host memory, 3 doubles per point, the
 * layout slx_get_point_cloud fills.  SLX_ERR_UNAVAILABLE: the file cannot be written. */
int slx_write_point_cloud_text(const char *path, const double *xyz, size_t n_points);
/* The same file in either dialect (enum slx_text_dialect above; SLX_ERR_INVALID_ARG for another value). */
int slx_write_point_cloud_text_ex(const char *path, const double *xyz, size_t n_points, int dialect);
/* CamMat, ProMat, R, T of the cv::FileStorage YAML Init reads (R/CCalculation.cpp:124-132; format of R/Result.yml). */
int slx_read_calibration_yaml(const char *path, double cam[9], double pro[9], double rot[9], double trans[3]);

/* The compiled-in configuration of the reference (R/StaticParameters.cpp) as the C++ mirror classes default to it, in this order:
 * PROJECTOR_RESLINE, PROJECTOR_RESROW, CAMERA_RESLINE, CAMERA_RESROW, GRAY_V_NUMDIGIT, PHASE_NUMDIGIT, FOV_MIN_DISTANCE,
 * FOV_MAX_DISTANCE, RECO_WINDOW_SIZE, DYNAFRAME_MAXNUM.  *n receives the count (10); SLX_ERR_INVALID_ARG when capacity is smaller. */
int slx_reference_defaults(int *values, int capacity, int *n);

int slx_version(void);

#ifdef __cplusplus
}
#endif
#endif
