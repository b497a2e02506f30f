// slx_api.cpp -- C-ABI host side of the MI355X-native DynaFrame static depth path.
//
// Owns what the reference's decoder objects own (R/CDecodePhase.cpp:19-45,
// R/CDecodeGray.cpp:65-105: the input-plane arrays; R/CCalculation.cpp:102-121:
// the result planes) as device buffers, does the one-off calibration algebra of
// CCalculation::Init (R/CCalculation.cpp:134-152) on the host, and launches the
// fused HIP kernel.  No OpenCV, no torch, no CPU fallback.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "slx.h"
#include "slx_kernels.h"

namespace {

thread_local std::string g_create_error;

struct Plane {
    const uint8_t *dev = nullptr;   // what the kernel reads
    uint8_t *owned = nullptr;       // staging area for host frames (deep copy): this plane's part of its group's slab
    size_t stride = 0;
    bool set = false;
};

}  // namespace

constexpr size_t kTextInfoWords = 2 + SLX_TEXT_MAX_PIECES + 1;

struct slx_ctx {
    slx_config cfg;
    SlxKParams kp;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // Completion of the most recent work that wrote the context's outputs or read its staged inputs, on whatever stream it
    // ran (the context's own or a caller's): what a later host copy / staging overwrite / launch on another stream waits for.
    // Work on the context's own stream needs no event: the stream itself can be waited for, and an event is recorded on it
    // only when another stream has to be ordered behind it (a per-launch record would cost a packet between dependent launches).
    // No handle of a caller's stream is ever kept: the caller may destroy the stream right after the call (and a new stream
    // may get the same handle value), so nothing here records on, waits on or compares against a remembered caller stream.
    hipEvent_t ev_done = nullptr;
    bool ev_pending = false;                   // the most recent work ran on a caller's stream and ev_done marks its end
    bool own_pending = false;                  // the most recent work ran on the context's own stream
    std::vector<int16_t> lut;
    int16_t *d_lut = nullptr;
    std::vector<Plane> phase, gray;
    // host frames are staged in ONE allocation per group, plane k at k * staging_pitch * height: equally spaced planes are
    // what the strip kernel's running plane offsets address
    uint8_t *phase_slab = nullptr, *gray_slab = nullptr;
    unsigned *d_cloud_counts = nullptr, *d_cloud_tiles = nullptr;   // slx_cloud_entries() / slx_cloud_tiles() + 1, point-cloud compaction
    double *d_cloud = nullptr;
    double *h_cloud = nullptr;                                        // pinned, one triple per pixel: slx_get_point_cloud_view
    unsigned *h_cloud_total = nullptr;                                // pinned: [0] the write kernel stores the point count here, [1] the fused kernel's "gave up" tag
    unsigned cloud_fallbacks = 0;                                     // frames repeated on the two-launch path because the fused launch gave up (diagnostics)
    unsigned long long *d_cloud_words = nullptr;                      // fused cloud: ticket counter + epoch-tagged counts (slx_cloud.hip)
    unsigned cloud_epoch = 0;                                         // launches since the words were zeroed
    // the cloud's text formatted on the device (slx_text.hip): device text + workgroup lengths, the text in pinned memory, [length, flag]
    unsigned char *d_text = nullptr;
    size_t d_text_capacity = 0;
    unsigned *d_text_sums = nullptr;                                  // slx_text_workgroups() words + the 8-byte length behind them
    size_t d_text_sums_capacity = 0;
    char *h_text = nullptr;
    size_t h_text_capacity = 0;
    unsigned long long *h_text_info = nullptr;                        // pinned: [0] the length of the text, [1] the range flag (a tag), [2 ..] the pieces' offsets
    unsigned text_tag = 0;
    int text_dialect = SLX_TEXT_LIBSTDCXX;                            // slx_set_text_dialect
    hipStream_t text_stream = nullptr;                                // the text's pieces cross PCIe on it while the next piece is formatted
    hipEvent_t ev_text[SLX_TEXT_MAX_PIECES] = {};
    size_t cloud_capacity = 0;
    // dynamic-frame tracker: previous frame's strips, unblurred deltaP, staged camera image
    float *d_stripW_prev = nullptr, *d_stripB_prev = nullptr, *d_deltaP_raw = nullptr;
    // camera images of host-fed dynamic frames: two pinned host buffers and two device buffers, used alternately, so that
    // frame n+1's copy-in (copy stream) overlaps frame n's kernels (the context's stream)
    uint8_t *h_track_img[2] = {nullptr, nullptr}, *d_track_img[2] = {nullptr, nullptr};
    hipEvent_t ev_track_copied[2] = {nullptr, nullptr}, ev_track_used[2] = {nullptr, nullptr};
    bool track_slot_used[2] = {false, false}, track_slot_waited[2] = {false, false};
    hipStream_t copy_stream = nullptr;
    // k dynamic frames per transfer (slx_track_stage_frames): two pinned + two device slabs of track_slab_frames images each
    uint8_t *h_track_slab[2] = {nullptr, nullptr}, *d_track_slab[2] = {nullptr, nullptr};
    hipEvent_t ev_slab_copied[2] = {nullptr, nullptr}, ev_slab_used[2] = {nullptr, nullptr};
    bool slab_used[2] = {false, false};
    int track_slab_frames = 0;
    unsigned track_slab = 0;
    unsigned track_slot = 0;
    int track_window = 0;
    void *out[SLX_OUT_COUNT] = {};
    size_t out_bytes[SLX_OUT_COUNT] = {};
    size_t staging_pitch = 0;
    double P[12] = {}, cA = 0, cB = 0;
    bool aux = false;
    bool decoded = false;
    int variant = 0;
    SlxTuning tune{};
    SlxStreamState stream_state;               // queue counters of the stream kernel (device words), allocated on first use
    std::string err;
};

namespace {

int fail(slx_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

int hip_fail(slx_ctx *ctx, hipError_t e, const char *what)
{
    if (e == hipErrorOutOfMemory) return fail(ctx, SLX_ERR_OUT_OF_MEMORY, "%s: %s", what, hipGetErrorString(e));
    return fail(ctx, SLX_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define SLX_HIP(ctx, call)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

bool mode_has_gray(int m) { return m == SLX_MODE_GRAY_ONLY || m == SLX_MODE_GRAY_PHASE || m == SLX_MODE_MULTIFREQ_GRAYMASK; }
bool mode_has_phase(int m) { return m != SLX_MODE_GRAY_ONLY; }
bool mode_has_depth(int m) { return m >= SLX_MODE_GRAY_PHASE; }

int validate(const slx_config *c, std::string &msg)
{
    char b[256];
    auto bad = [&](const char *fmt, auto... a) { snprintf(b, sizeof b, fmt, a...); msg = b; return (int)SLX_ERR_INVALID_ARG; };
    if (!c) return bad("config is NULL");
    if (c->width <= 0 || c->height <= 0) return bad("width/height must be positive (got %dx%d)", c->width, c->height);
    if ((long long)c->width * c->height > (1ll << 31)) return bad("tile too large");
    if (c->mode < SLX_MODE_PHASE_ONLY || c->mode > SLX_MODE_MULTIFREQ_GRAYMASK) return bad("unknown mode %d", c->mode);
    if (mode_has_phase(c->mode)) {
        // R/CDecodePhase.cpp:122 rejects numMat <= 0; three steps is the algebraic minimum
        if (c->n_steps < 3 || c->n_steps > SLX_MAX_STEPS) return bad("n_steps must be in [3,%d] (got %d)", SLX_MAX_STEPS, c->n_steps);
        if (c->n_freq < 1 || c->n_freq > SLX_MAX_FREQ) return bad("n_freq must be in [1,%d] (got %d)", SLX_MAX_FREQ, c->n_freq);
        if ((c->mode == SLX_MODE_PHASE_ONLY || c->mode == SLX_MODE_GRAY_PHASE) && c->n_freq != 1)
            return bad("mode %d decodes exactly one frequency (got n_freq=%d)", c->mode, c->n_freq);
        for (int f = 0; f < c->n_freq; f++)
            if (c->period[f] <= 0 || c->period[f] >= (1 << 24)) return bad("period[%d] must be in [1,2^24) (got %d)", f, c->period[f]);
    }
    if (mode_has_gray(c->mode)) {
        // R/CDecodeGray.cpp:39
        if (c->gray_bits <= 0 || c->gray_bits > SLX_MAX_GRAY_BITS) return bad("gray_bits must be in [1,%d] (got %d)", SLX_MAX_GRAY_BITS, c->gray_bits);
        if (c->gray_stripe <= 0) return bad("gray_stripe must be positive (got %d)", c->gray_stripe);
        if (!c->gray_lut) return bad("gray_lut is NULL (the reference fails when the code file is missing, R/CDecodeGray.cpp:115)");
    }
    if (mode_has_depth(c->mode)) {
        if (!(c->fov_min <= c->fov_max)) return bad("fov_min must not exceed fov_max");
        if (c->cam[0] == 0.0 || c->cam[4] == 0.0) return bad("camera focal lengths must be non-zero");
    }
    const unsigned allowed = [&] {
        unsigned a = 0;
        if (mode_has_depth(c->mode)) a |= 1u << SLX_OUT_Z | 1u << SLX_OUT_X | 1u << SLX_OUT_Y | 1u << SLX_OUT_U | 1u << SLX_OUT_MASK;
        if (mode_has_phase(c->mode)) a |= 1u << SLX_OUT_PIX;
        if (mode_has_gray(c->mode)) a |= 1u << SLX_OUT_GRAY;
        if (mode_has_depth(c->mode) && c->mode != SLX_MODE_GRAY_PHASE && c->n_freq > 1) a |= 1u << SLX_OUT_K;
        return a;
    }();
    if (c->aux_outputs & ~allowed) return bad("aux_outputs 0x%x names outputs this mode does not produce (allowed 0x%x)", c->aux_outputs, allowed);
    return SLX_OK;
}

size_t out_elem_bytes(int which)
{
    if (which == SLX_OUT_K || which == SLX_OUT_DELTAP || which == SLX_OUT_STRIPW || which == SLX_OUT_STRIPB) return 4;
    return which == SLX_OUT_MASK ? 1 : 8;
}

size_t out_planes(const slx_config &c, int which)
{
    if (which == SLX_OUT_PIX) return (size_t)c.n_freq;
    if (which == SLX_OUT_K) return (size_t)(c.n_freq - 1);
    return 1;
}

// x1 weights; must stay the same formula as the documented spec (DESIGN.md "x1").
void nstep_weights(int n, float *wy, float *wx, float *scale)
{
    const double pi = 3.1415926535897932384626433832795;
    for (int k = 0; k < n; k++) {
        const double a = 2.0 * pi * (double)k / (double)n;
        double c = std::cos(a), s = std::sin(a);
        if (std::fabs(c) < 1e-9) c = 0.0;
        if (std::fabs(s) < 1e-9) s = 0.0;
        wy[k] = (float)c;
        wx[k] = (float)s;
    }
    *scale = 2.0f / (float)n;
}

// R/CCalculation.cpp:134-152: P = ProMat * [R T] (k ascending), cA, cB and the scalars the
// kernel rebuilds cC/cD from.  Plain double arithmetic, no contraction (-ffp-contract=off).
void calibrate(slx_ctx *ctx)
{
    const slx_config &c = ctx->cfg;
    double RT[12];
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) RT[r * 4 + k] = c.rot[r * 3 + k];
        RT[r * 4 + 3] = c.trans[r];
    }
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 4; col++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++) s = s + c.pro[r * 3 + k] * RT[k * 4 + col];
            ctx->P[r * 4 + col] = s;
        }
    const double fu = c.cam[0], fv = c.cam[4];
    ctx->cA = fu * fv * ctx->P[3];
    ctx->cB = fu * fv * ctx->P[11];
    SlxKParams &kp = ctx->kp;
    kp.cx = c.cam[2];
    kp.cy = c.cam[5];
    kp.fu = fu;
    kp.fv = fv;
    kp.P00 = ctx->P[0];
    kp.P01 = ctx->P[1];
    kp.K1 = fu * fv * ctx->P[2];
    kp.P20 = ctx->P[8];
    kp.P21 = ctx->P[9];
    kp.K2 = fu * fv * ctx->P[10];
    kp.cA = ctx->cA;
    kp.cB = ctx->cB;
}

// Host-side check that every operand the kernel will touch matches the grid it is launched on.
int check_launch_shapes(slx_ctx *ctx, const SlxKParams &kp, int n_phase, int n_gray, int n_sets)
{
    const slx_config &c = ctx->cfg;
    if (n_sets <= 0 || n_sets > 65535) return fail(ctx, SLX_ERR_INVALID_ARG, "n_sets must be in [1,65535] (got %d)", n_sets);
    if (kp.row_stride < (size_t)c.width) return fail(ctx, SLX_ERR_INVALID_ARG, "row stride %zu is smaller than the width %d", kp.row_stride, c.width);
    for (int i = 0; i < n_phase; i++)
        if (!kp.phase[i]) return fail(ctx, SLX_ERR_MISSING_FRAME, "phase plane %d was never set", i);
    for (int i = 0; i < n_gray; i++)
        if (!kp.gray[i]) return fail(ctx, SLX_ERR_MISSING_FRAME, "gray plane %d was never set", i);
    if (kp.quads_per_row != (unsigned)((c.width + SLX_QUAD - 1) / SLX_QUAD) || kp.n_quads != kp.quads_per_row * (unsigned)c.height)
        return fail(ctx, SLX_ERR_INVALID_ARG, "internal: quad geometry does not match the tile");
    return SLX_OK;
}

// lut[g] == inverse of g = b ^ (b >> 1) for every entry (the table of R/Patterns/vGrayCode.txt)
bool is_reflected_gray(const std::vector<int16_t> &lut)
{
    for (size_t b = 0; b < lut.size(); b++)
        if (lut[b ^ (b >> 1)] != (int16_t)b) return false;
    return lut.size() <= 32768;
}

bool ptr_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// The three uses of ev_done (see slx_ctx).
int mark_done(slx_ctx *ctx, hipStream_t s)
{
    if (s == ctx->stream) {
        ctx->own_pending = true;               // (ordered behind any earlier caller-stream work by order_after_done)
        ctx->ev_pending = false;
        return SLX_OK;
    }
    SLX_HIP(ctx, hipEventRecord(ctx->ev_done, s));
    ctx->ev_pending = true;
    ctx->own_pending = false;
    return SLX_OK;
}
int wait_done_host(slx_ctx *ctx)
{
    if (ctx->own_pending) SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    else if (ctx->ev_pending) SLX_HIP(ctx, hipEventSynchronize(ctx->ev_done));
    ctx->own_pending = ctx->ev_pending = false;                      // waited for: nothing later has to be ordered behind it
    return SLX_OK;
}
// A caller's stream that is being captured into a hipGraph: the launch becomes a node of the graph, ordered by the graph; the context
// neither waits for events recorded outside the capture (it could not) nor records one of its own for it.
bool capturing(slx_ctx *ctx, hipStream_t s)
{
    if (s == ctx->stream) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}
int order_after_done(slx_ctx *ctx, hipStream_t s)
{
    if (ctx->own_pending) {
        if (s == ctx->stream) return SLX_OK;   // same stream: already ordered
        SLX_HIP(ctx, hipEventRecord(ctx->ev_done, ctx->stream));
        SLX_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_done, 0));
    } else if (ctx->ev_pending) {
        // always through the event, also when s looks like the stream it was recorded on: the runtime knows whether the
        // event's queue is this stream's (and then enqueues nothing); a remembered handle value would not -- the caller may
        // have destroyed that stream and been handed the same value for a new one while the old work is still in flight
        SLX_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_done, 0));
    }
    return SLX_OK;
}

}  // namespace

extern "C" {

int slx_version(void) { return SLX_VERSION_MAJOR * 100 + SLX_VERSION_MINOR; }

int slx_validate_config(const slx_config *cfg, char *msg, size_t msg_bytes)
{
    std::string m;
    int rc = validate(cfg, m);
    if (msg && msg_bytes) {
        snprintf(msg, msg_bytes, "%s", m.c_str());
    }
    return rc;
}

const char *slx_last_error(const slx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void slx_destroy(slx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // first every wait -- work on a caller's stream (through its event: the stream itself may be gone by now), the context's
    // own streams -- and only then the frees
    if (ctx->ev_pending) (void)hipEventSynchronize(ctx->ev_done);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->text_stream) (void)hipStreamSynchronize(ctx->text_stream);
    if (ctx->phase_slab) (void)hipFree(ctx->phase_slab);
    if (ctx->gray_slab) (void)hipFree(ctx->gray_slab);
    for (void *o : ctx->out)
        if (o) (void)hipFree(o);
    if (ctx->d_lut) (void)hipFree(ctx->d_lut);
    if (ctx->stream_state.counters) (void)hipFree(ctx->stream_state.counters);
    for (void *q : {(void *)ctx->d_cloud_counts, (void *)ctx->d_cloud_tiles, (void *)ctx->d_cloud, (void *)ctx->d_cloud_words, (void *)ctx->d_stripW_prev,
                    (void *)ctx->d_stripB_prev, (void *)ctx->d_deltaP_raw, (void *)ctx->d_track_img[0], (void *)ctx->d_track_img[1]})
        if (q) (void)hipFree(q);
    for (uint8_t *h : ctx->h_track_img)
        if (h) (void)hipHostFree(h);
    for (int i = 0; i < 2; i++) {
        if (ctx->h_track_slab[i]) (void)hipHostFree(ctx->h_track_slab[i]);
        if (ctx->d_track_slab[i]) (void)hipFree(ctx->d_track_slab[i]);
        if (ctx->ev_slab_copied[i]) (void)hipEventDestroy(ctx->ev_slab_copied[i]);
        if (ctx->ev_slab_used[i]) (void)hipEventDestroy(ctx->ev_slab_used[i]);
    }
    if (ctx->h_cloud_total) (void)hipHostFree(ctx->h_cloud_total);
    if (ctx->h_cloud) (void)hipHostFree(ctx->h_cloud);
    if (ctx->d_text) (void)hipFree(ctx->d_text);
    if (ctx->d_text_sums) (void)hipFree(ctx->d_text_sums);
    if (ctx->h_text) (void)hipHostFree(ctx->h_text);
    if (ctx->h_text_info) (void)hipHostFree(ctx->h_text_info);
    for (hipEvent_t e : {ctx->ev0, ctx->ev1, ctx->ev_done, ctx->ev_track_copied[0], ctx->ev_track_copied[1], ctx->ev_track_used[0], ctx->ev_track_used[1]})
        if (e) (void)hipEventDestroy(e);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->text_stream) (void)hipStreamDestroy(ctx->text_stream);
    for (hipEvent_t ev : ctx->ev_text)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int slx_create(const slx_config *cfg, slx_ctx **out)
{
    if (!out) return fail(nullptr, SLX_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    std::string msg;
    int rc = validate(cfg, msg);
    if (rc != SLX_OK) return fail(nullptr, rc, "%s", msg.c_str());

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, SLX_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    int dev = cfg->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= n_dev) return fail(nullptr, SLX_ERR_INVALID_ARG, "device %d does not exist (%d devices)", dev, n_dev);

    slx_ctx *ctx = new slx_ctx;
    ctx->cfg = *cfg;
    ctx->device = dev;
    auto bail = [&](int code) {
        g_create_error = ctx->err;
        slx_destroy(ctx);
        return code;
    };
#define SLX_TRY(call)                                                  \
    do {                                                               \
        hipError_t e2_ = (call);                                       \
        if (e2_ != hipSuccess) return bail(hip_fail(ctx, e2_, #call)); \
    } while (0)

    SLX_TRY(hipSetDevice(dev));
    SLX_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    SLX_TRY(hipEventCreate(&ctx->ev0));
    SLX_TRY(hipEventCreate(&ctx->ev1));
    SLX_TRY(hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming));

    const slx_config &c = ctx->cfg;
    const int n_phase = mode_has_phase(c.mode) ? c.n_freq * c.n_steps : 0;
    const int n_gray = mode_has_gray(c.mode) ? 2 * c.gray_bits : 0;
    ctx->phase.resize((size_t)n_phase);
    ctx->gray.resize((size_t)n_gray);
    ctx->staging_pitch = ((size_t)c.width + 3) & ~(size_t)3;

    SlxKParams &kp = ctx->kp;
    std::memset(&kp, 0, sizeof kp);
    kp.width = c.width;
    kp.height = c.height;
    kp.row_offset = c.row_offset;
    kp.quads_per_row = (unsigned)((c.width + SLX_QUAD - 1) / SLX_QUAD);
    kp.n_quads = kp.quads_per_row * (unsigned)c.height;
    kp.n_freq = mode_has_phase(c.mode) ? c.n_freq : 0;
    kp.n_steps = mode_has_phase(c.mode) ? c.n_steps : 4;
    for (int f = 0; f < SLX_MAX_FREQ; f++) kp.period[f] = f < c.n_freq ? c.period[f] : 1;
    kp.gray_bits = mode_has_gray(c.mode) ? c.gray_bits : 0;
    kp.gray_stripe = c.gray_stripe;
    kp.fov_min = c.fov_min;
    kp.fov_max = c.fov_max;
    kp.out_set_stride = (size_t)c.width * (size_t)c.height;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) kp.n_cus = (unsigned)cus;
    }
    if (mode_has_phase(c.mode)) nstep_weights(c.n_steps, kp.wy, kp.wx, &kp.wscale);
    if (mode_has_depth(c.mode)) calibrate(ctx);
    for (int f = 0; f < SLX_MAX_FREQ; f++) {
        // constants of the fast temporal unwrap (slx_kernels.hip: unwrap_stage<true>)
        kp.inv_period[f] = 1.0 / (double)kp.period[f];
        kp.half_biased[f] = 0.5 + 0x1p-30 / (double)kp.period[f];
    }

    if (mode_has_gray(c.mode)) {
        const size_t n = (size_t)1 << c.gray_bits;
        ctx->lut.assign(c.gray_lut, c.gray_lut + n);
        ctx->cfg.gray_lut = ctx->lut.data();
        SLX_TRY(hipMalloc((void **)&ctx->d_lut, n * sizeof(int16_t)));
        SLX_TRY(hipMemcpy(ctx->d_lut, ctx->lut.data(), n * sizeof(int16_t), hipMemcpyHostToDevice));
        kp.lut = ctx->d_lut;
        kp.std_gray = is_reflected_gray(ctx->lut) ? 1 : 0;
    }

    // result planes (the reference allocates them in Init, R/CCalculation.cpp:110-121)
    unsigned want = c.aux_outputs;
    if (mode_has_depth(c.mode)) want |= 1u << SLX_OUT_Z;
    if (c.mode == SLX_MODE_PHASE_ONLY) want |= 1u << SLX_OUT_PIX;
    if (c.mode == SLX_MODE_GRAY_ONLY) want |= 1u << SLX_OUT_GRAY;
    const size_t hw = (size_t)c.width * (size_t)c.height;
    for (int w = 0; w < SLX_OUT_COUNT; w++) {
        if (!(want & (1u << w))) continue;
        const size_t bytes = hw * out_planes(c, w) * out_elem_bytes(w);
        if (bytes == 0) continue;
        SLX_TRY(hipMalloc(&ctx->out[w], bytes));
        SLX_TRY(hipMemset(ctx->out[w], 0, bytes));
        ctx->out_bytes[w] = bytes;
    }
    ctx->aux = mode_has_depth(c.mode) && (c.aux_outputs & ~(1u << SLX_OUT_Z)) != 0;
    kp.z = (double *)ctx->out[SLX_OUT_Z];
    kp.x = (double *)ctx->out[SLX_OUT_X];
    kp.y = (double *)ctx->out[SLX_OUT_Y];
    kp.U = (double *)ctx->out[SLX_OUT_U];
    kp.pix = (double *)ctx->out[SLX_OUT_PIX];
    kp.gray_out = (double *)ctx->out[SLX_OUT_GRAY];
    kp.k = (int32_t *)ctx->out[SLX_OUT_K];
    kp.mask = (uint8_t *)ctx->out[SLX_OUT_MASK];
#undef SLX_TRY
    *out = ctx;
    return SLX_OK;
}

int slx_set_frame(slx_ctx *ctx, int group, int idx, const uint8_t *data, size_t stride_bytes, int mem_kind)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (group != SLX_GROUP_GRAY && group != SLX_GROUP_PHASE) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown group %d", group);
    std::vector<Plane> &v = group == SLX_GROUP_GRAY ? ctx->gray : ctx->phase;
    if (v.empty())   // reference: "grePicture Space is not allocated", R/CDecodePhase.cpp:109-113
        return fail(ctx, SLX_ERR_NOT_CONFIGURED, "this mode has no %s planes", group == SLX_GROUP_GRAY ? "gray" : "phase");
    if (idx < 0 || (size_t)idx >= v.size()) return fail(ctx, SLX_ERR_NOT_CONFIGURED, "plane index %d outside [0,%zu)", idx, v.size());
    if (!data) return fail(ctx, SLX_ERR_INVALID_ARG, "data is NULL");
    if (stride_bytes < (size_t)ctx->cfg.width) return fail(ctx, SLX_ERR_INVALID_ARG, "stride %zu is smaller than the width %d", stride_bytes, ctx->cfg.width);
    if (mem_kind != SLX_MEM_HOST && mem_kind != SLX_MEM_DEVICE) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
    Plane &p = v[(size_t)idx];
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (mem_kind == SLX_MEM_DEVICE) {
        p.dev = data;
        p.stride = stride_bytes;
    } else {
        if (!p.owned) {
            uint8_t *&slab = group == SLX_GROUP_GRAY ? ctx->gray_slab : ctx->phase_slab;
            const size_t plane_bytes = ctx->staging_pitch * (size_t)ctx->cfg.height;
            if (!slab) SLX_HIP(ctx, hipMalloc((void **)&slab, plane_bytes * v.size()));
            p.owned = slab + (size_t)idx * plane_bytes;
        }
        // the previous decode (asynchronous, on the context's or a caller's stream) may still be reading the staging buffer
        if (int rc = wait_done_host(ctx)) return rc;
        // deep copy, complete before return (pic.copyTo, R/CDecodePhase.cpp:114)
        SLX_HIP(ctx, hipMemcpy2D(p.owned, ctx->staging_pitch, data, stride_bytes, (size_t)ctx->cfg.width,
                                 (size_t)ctx->cfg.height, hipMemcpyHostToDevice));
        p.dev = p.owned;
        p.stride = ctx->staging_pitch;
    }
    p.set = true;
    return SLX_OK;
}

int slx_set_gray_lut(slx_ctx *ctx, const int16_t *lut, size_t n)
{
    if (!ctx || !lut) return SLX_ERR_INVALID_ARG;
    if (!ctx->d_lut) return fail(ctx, SLX_ERR_NOT_CONFIGURED, "this mode has no Gray table");
    if (n != ctx->lut.size()) return fail(ctx, SLX_ERR_INVALID_ARG, "table has %zu entries, expected %zu", n, ctx->lut.size());
    ctx->lut.assign(lut, lut + n);
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = wait_done_host(ctx)) return rc;          // the last decode may still be reading the table, on any stream
    SLX_HIP(ctx, hipMemcpy(ctx->d_lut, ctx->lut.data(), n * sizeof(int16_t), hipMemcpyHostToDevice));
    ctx->kp.std_gray = is_reflected_gray(ctx->lut) ? 1 : 0;
    return SLX_OK;
}

static int launch(slx_ctx *ctx, SlxKParams &kp, int n_sets, bool aux, void *stream)
{
    const slx_config &c = ctx->cfg;
    const int n_phase = (int)ctx->phase.size(), n_gray = (int)ctx->gray.size();
    int rc = check_launch_shapes(ctx, kp, n_phase, n_gray, n_sets);
    if (rc != SLX_OK) return rc;
    // dword input loads and 16-byte output stores need aligned bases, strides and width
    bool al = (c.width % SLX_QUAD) == 0 && (kp.row_stride % 4) == 0 && (kp.phase_set_stride % 4) == 0 && (kp.gray_set_stride % 4) == 0;
    for (int i = 0; i < n_phase; i++) al = al && ptr_aligned(kp.phase[i], 4);
    for (int i = 0; i < n_gray; i++) al = al && ptr_aligned(kp.gray[i], 4);
    // stores are 16 bytes wide for the f64 and i32 planes and 4 bytes for the mask: the planes' bases, and the start of every
    // plane of every frame-set ((set * planes + plane) * out_set_stride elements in), must be aligned accordingly
    auto plane_ok = [&](const void *o, size_t elem, size_t planes, size_t align) {
        if (!o) return true;
        if (!ptr_aligned(o, align)) return false;
        return (n_sets == 1 && planes <= 1) || (kp.out_set_stride * elem) % align == 0;
    };
    al = al && plane_ok(kp.z, 8, 1, 16) && plane_ok(kp.x, 8, 1, 16) && plane_ok(kp.y, 8, 1, 16) && plane_ok(kp.U, 8, 1, 16) &&
         plane_ok(kp.gray_out, 8, 1, 16) && plane_ok(kp.pix, 8, (size_t)kp.n_freq, 16) &&
         plane_ok(kp.k, 4, (size_t)(kp.n_freq > 1 ? kp.n_freq - 1 : 1), 16) && plane_ok(kp.mask, 1, 1, 4);
    kp.aligned = al ? 1 : 0;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->variant == SLX_VARIANT_STRIP && !slx_strip_eligible(kp, c.mode, aux))
        return fail(ctx, SLX_ERR_UNAVAILABLE, "variant %d (strip kernel) cannot run this configuration or these operands", ctx->variant);
    if (ctx->variant == SLX_VARIANT_GENERIC_FAST && !(mode_has_depth(c.mode) && slx_fast_arith_ok(kp)))
        return fail(ctx, SLX_ERR_UNAVAILABLE, "variant %d (cheap exact arithmetic) needs a depth mode, periods <= 2^14 and moderate calibration magnitudes", ctx->variant);
    // ordered after the last work on the context's outputs when that ran on another stream (tracker / cloud kernels on the
    // context's stream, an earlier decode on a caller's stream)
    const bool captured = capturing(ctx, s);
    if (captured) {
        // everything the context itself has in flight must be done before the capture began (the graph cannot depend on it)
        if (ctx->own_pending || ctx->ev_pending)
            return fail(ctx, SLX_ERR_INVALID_ARG, "capture into a hipGraph: call slx_synchronize before the capture begins (work of this context is still in flight)");
        if (ctx->timed) return fail(ctx, SLX_ERR_INVALID_ARG, "capture into a hipGraph: switch slx_enable_timing off");
    } else if (int rc2 = order_after_done(ctx, s)) {
        return rc2;
    }
    if (ctx->timed) SLX_HIP(ctx, hipEventRecord(ctx->ev0, s));
    if (!captured && !ctx->stream_state.counters && n_sets > 1 && (c.mode == SLX_MODE_MULTIFREQ || c.mode == SLX_MODE_GRAY_PHASE)) {
        // queue counters of the stream kernels, zeroed here once: the kernels leave them at zero (the wave that draws a queue's last ticket
        // of a launch resets it), so a launch -- also one captured into a hipGraph and replayed -- always starts from zero.  (A capture
        // that meets a context without counters -- no batch decoded outside a capture yet -- keeps the strip kernel: no allocation inside a capture.)
        SLX_HIP(ctx, hipMalloc((void **)&ctx->stream_state.counters, (size_t)SLX_STREAM_MAX_QUEUES * 32u * sizeof(unsigned)));
        SLX_HIP(ctx, hipMemset(ctx->stream_state.counters, 0, (size_t)SLX_STREAM_MAX_QUEUES * 32u * sizeof(unsigned)));
        ctx->stream_state.key = 1;
    }
    SlxTuning tune = ctx->tune;
    if (captured && ctx->stream_state.key == 0) tune.stream = 1;      // counters that a failed launch left dirty are not zeroed inside a capture
    int e = slx_launch_fused(kp, c.mode, aux, n_sets, ctx->variant, s, &tune, &ctx->stream_state);
    if (e != 0) ctx->stream_state.key = 0;                   // whatever a failed launch left in the counters is not trusted: zeroed before the next one
    if (e != 0) return hip_fail(ctx, (hipError_t)e, "kernel launch");
    if (ctx->timed) SLX_HIP(ctx, hipEventRecord(ctx->ev1, s));
    return captured ? SLX_OK : mark_done(ctx, s);
}

int slx_decode(slx_ctx *ctx, void *stream)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    SlxKParams kp = ctx->kp;
    size_t stride = 0;
    bool mixed = false;
    for (auto *v : {&ctx->phase, &ctx->gray}) {
        for (size_t i = 0; i < v->size(); i++) {
            const Plane &p = (*v)[i];
            if (!p.set) return fail(ctx, SLX_ERR_MISSING_FRAME, "%s plane %zu was never set", v == &ctx->phase ? "phase" : "gray", i);
            if (stride == 0) stride = p.stride;
            mixed = mixed || p.stride != stride;
            (v == &ctx->phase ? kp.phase : kp.gray)[i] = p.dev;
        }
    }
    if (mixed) {
        // The kernels take ONE row stride for all planes of a decode.  Frames of different pitches (device images borrowed from
        // several buffers, or mixed with host frames, which sit in the staging slab at its pitch) are brought to the staging pitch
        // the way SetMat brings every image into the decoder (pic.copyTo, R/CDecodePhase.cpp:114): a device-to-device copy into
        // the plane's own staging slot, on the decode's stream, ahead of the launch.  The borrowed pointer stays what it was.
        SLX_HIP(ctx, hipSetDevice(ctx->device));
        hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
        if (int rc0 = order_after_done(ctx, s)) return rc0;            // the last decode may still read the staging slots
        stride = ctx->staging_pitch;
        for (auto *v : {&ctx->phase, &ctx->gray}) {
            for (size_t i = 0; i < v->size(); i++) {
                Plane &p = (*v)[i];
                if (p.stride == stride) continue;
                if (!p.owned) {
                    uint8_t *&slab = v == &ctx->gray ? ctx->gray_slab : ctx->phase_slab;
                    const size_t plane_bytes = ctx->staging_pitch * (size_t)ctx->cfg.height;
                    if (!slab) SLX_HIP(ctx, hipMalloc((void **)&slab, plane_bytes * v->size()));
                    p.owned = slab + i * plane_bytes;
                }
                SLX_HIP(ctx, hipMemcpy2DAsync(p.owned, stride, p.dev, p.stride, (size_t)ctx->cfg.width, (size_t)ctx->cfg.height, hipMemcpyDeviceToDevice, s));
                (v == &ctx->phase ? kp.phase : kp.gray)[i] = p.owned;
            }
        }
    }
    kp.row_stride = stride;
    kp.phase_set_stride = kp.gray_set_stride = 0;
    int rc = launch(ctx, kp, 1, ctx->aux, stream);
    if (rc == SLX_OK) ctx->decoded = true;
    return rc;
}

int slx_decode_batch_ex(slx_ctx *ctx, int n_sets, const uint8_t *phase_base, size_t phase_set_stride, const uint8_t *gray_base,
                        size_t gray_set_stride, size_t row_stride, const slx_batch_out *out, void *stream)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    const slx_config &c = ctx->cfg;
    const size_t n_phase = ctx->phase.size(), n_gray = ctx->gray.size();
    if (!out) return fail(ctx, SLX_ERR_INVALID_ARG, "out is NULL");
    if (n_phase && !phase_base) return fail(ctx, SLX_ERR_MISSING_FRAME, "phase_base is NULL");
    if (n_gray && !gray_base) return fail(ctx, SLX_ERR_MISSING_FRAME, "gray_base is NULL");
    if (row_stride < (size_t)c.width) return fail(ctx, SLX_ERR_INVALID_ARG, "row stride %zu is smaller than the width %d", row_stride, c.width);
    const size_t plane_bytes = row_stride * (size_t)c.height;
    if (n_sets > 1 && n_phase && phase_set_stride < n_phase * plane_bytes) return fail(ctx, SLX_ERR_INVALID_ARG, "phase_set_stride %zu overlaps frame-sets", phase_set_stride);
    if (n_sets > 1 && n_gray && gray_set_stride < n_gray * plane_bytes) return fail(ctx, SLX_ERR_INVALID_ARG, "gray_set_stride %zu overlaps frame-sets", gray_set_stride);
    const size_t hw = (size_t)c.width * (size_t)c.height;
    if (out->plane_stride != 0 && out->plane_stride < hw) return fail(ctx, SLX_ERR_INVALID_ARG, "plane_stride %zu is smaller than the tile (%zu pixels)", out->plane_stride, hw);
    SlxKParams kp = ctx->kp;
    for (size_t i = 0; i < n_phase; i++) kp.phase[i] = phase_base + i * plane_bytes;
    for (size_t i = 0; i < n_gray; i++) kp.gray[i] = gray_base + i * plane_bytes;
    kp.phase_set_stride = n_phase ? phase_set_stride : 0;
    kp.gray_set_stride = n_gray ? gray_set_stride : 0;
    kp.row_stride = row_stride;
    kp.out_set_stride = out->plane_stride ? out->plane_stride : hw;
    double *primary = out->z;
    if (!primary) return fail(ctx, SLX_ERR_INVALID_ARG, "z_out is NULL");
    kp.x = kp.y = kp.U = nullptr;
    kp.k = nullptr;
    kp.mask = nullptr;
    kp.z = kp.pix = kp.gray_out = nullptr;
    bool aux = false;
    if (c.mode == SLX_MODE_PHASE_ONLY) kp.pix = primary;
    else if (c.mode == SLX_MODE_GRAY_ONLY) kp.gray_out = primary;
    else {
        kp.z = primary;
        kp.x = out->x;
        kp.y = out->y;
        kp.U = out->U;
        kp.mask = out->mask;
        if (out->k) {
            if (!(c.mode != SLX_MODE_GRAY_PHASE && c.n_freq > 1)) return fail(ctx, SLX_ERR_UNAVAILABLE, "this mode has no fringe orders (k)");
            kp.k = out->k;
        }
        aux = out->x || out->y || out->U || out->mask || out->k;
    }
    if (!mode_has_depth(c.mode) && (out->x || out->y || out->U || out->mask || out->k))
        return fail(ctx, SLX_ERR_UNAVAILABLE, "mode %d produces no depth-side outputs", c.mode);
    return launch(ctx, kp, n_sets, aux, stream);
}

int slx_decode_batch(slx_ctx *ctx, int n_sets, const uint8_t *phase_base, size_t phase_set_stride,
                     const uint8_t *gray_base, size_t gray_set_stride, size_t row_stride, double *z_out, void *stream)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (!z_out) return fail(ctx, SLX_ERR_INVALID_ARG, "z_out is NULL");
    slx_batch_out out{};
    out.z = z_out;
    return slx_decode_batch_ex(ctx, n_sets, phase_base, phase_set_stride, gray_base, gray_set_stride, row_stride, &out, stream);
}

int slx_synchronize(slx_ctx *ctx)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = wait_done_host(ctx)) return rc;          // the last launch, whichever stream it ran on
    SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLX_OK;
}

int slx_get_stream(slx_ctx *ctx, void **stream)
{
    if (!ctx || !stream) return SLX_ERR_INVALID_ARG;
    *stream = (void *)ctx->stream;
    return SLX_OK;
}

int slx_output_device_ptr(slx_ctx *ctx, int which, void **ptr)
{
    if (!ctx || !ptr) return SLX_ERR_INVALID_ARG;
    if (which < 0 || which >= SLX_OUT_COUNT) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown output %d", which);
    if (!ctx->out[which]) return fail(ctx, SLX_ERR_UNAVAILABLE, "output %d is not produced by this context (mode %d, aux_outputs 0x%x)", which, ctx->cfg.mode, ctx->cfg.aux_outputs);
    *ptr = ctx->out[which];
    return SLX_OK;
}

int slx_get_output(slx_ctx *ctx, int which, void *dst, size_t dst_bytes, int mem_kind)
{
    if (!ctx || !dst) return SLX_ERR_INVALID_ARG;
    if (which < 0 || which >= SLX_OUT_COUNT) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown output %d", which);
    if (!ctx->out[which]) return fail(ctx, SLX_ERR_UNAVAILABLE, "output %d is not produced by this context (mode %d, aux_outputs 0x%x)", which, ctx->cfg.mode, ctx->cfg.aux_outputs);
    if (!ctx->decoded) return fail(ctx, SLX_ERR_NOT_DECODED, "no decode has run yet");
    if (dst_bytes < ctx->out_bytes[which]) return fail(ctx, SLX_ERR_INVALID_ARG, "destination holds %zu bytes, output needs %zu", dst_bytes, ctx->out_bytes[which]);
    if (mem_kind != SLX_MEM_HOST && mem_kind != SLX_MEM_DEVICE) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = wait_done_host(ctx)) return rc;          // the producer of the outputs, on whichever stream it ran; nothing else is stalled
    SLX_HIP(ctx, hipMemcpy(dst, ctx->out[which], ctx->out_bytes[which],
                           mem_kind == SLX_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    return SLX_OK;
}

int slx_get_depth(slx_ctx *ctx, double *z, int mem_kind)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    return slx_get_output(ctx, SLX_OUT_Z, z, ctx->out_bytes[SLX_OUT_Z], mem_kind);
}

int slx_get_point_cloud(slx_ctx *ctx, double *xyz, size_t capacity_points, size_t *n_points, int mem_kind)
{
    if (!ctx || !n_points) return SLX_ERR_INVALID_ARG;
    if (!mode_has_depth(ctx->cfg.mode)) return fail(ctx, SLX_ERR_UNAVAILABLE, "mode %d produces no depth", ctx->cfg.mode);
    if (!ctx->decoded) return fail(ctx, SLX_ERR_NOT_DECODED, "no decode has run yet");
    return slx_point_cloud_of_depth(ctx, (const double *)ctx->out[SLX_OUT_Z], xyz, capacity_points, n_points, mem_kind);
}

// The fused point-cloud launch (slx_cloud.hip), queued on the context's stream; nothing is waited for.  fq: the plan (groups, parts,
// rows_per_part filled in).  *tag receives what the launch's "gave up" flag (h_cloud_total[1]) would carry.
static int launch_cloud_fused_async(slx_ctx *ctx, SlxCloudFused fq, const double *z, double *target, unsigned *total_dev, unsigned *tag)
{
    const slx_config &c = ctx->cfg;
    const size_t n_words = slx_cloud_fused_words(fq.groups, fq.parts);
    if (!ctx->d_cloud_words || ctx->cloud_epoch >= (1u << 30)) {
        // first use, or before the epoch tags could repeat: the ticket counter and every tagged word start from zero
        if (!ctx->d_cloud_words) SLX_HIP(ctx, hipMalloc((void **)&ctx->d_cloud_words, n_words * sizeof(unsigned long long)));
        SLX_HIP(ctx, hipMemsetAsync(ctx->d_cloud_words, 0, n_words * sizeof(unsigned long long), ctx->stream));
        ctx->cloud_epoch = 0;
    }
    fq.z = z;
    fq.xyz = target;
    fq.words = ctx->d_cloud_words;
    fq.total_dev = total_dev;
    fq.total_host = ctx->h_cloud_total;
    fq.W = c.width;
    fq.H = c.height;
    fq.epoch = ctx->cloud_epoch++;
    *tag = fq.epoch + 1u;
    fq.spin_limit = ctx->tune.cloud_spin > 0 ? (unsigned)ctx->tune.cloud_spin - 1u : SLX_CLOUD_SPIN_LIMIT;
    fq.gave_up_host = ctx->h_cloud_total + 1;
    fq.stamps = ctx->kp.stamps;
    fq.stamp_items = ctx->kp.stamp_items;
    ctx->h_cloud_total[1] = 0u;                                     // (no cloud launch of this context is in flight here: every cloud call ends with its launches drained)
    fq.row_offset = ctx->kp.row_offset;
    fq.fov_min = ctx->kp.fov_min;
    fq.fov_max = ctx->kp.fov_max;
    fq.cx = ctx->kp.cx;
    fq.cy = ctx->kp.cy;
    fq.fu = ctx->kp.fu;
    fq.fv = ctx->kp.fv;
    const int e2 = slx_launch_cloud_fused(fq, ctx->stream);
    if (e2 != 0) {
        ctx->cloud_epoch = 1u << 30;                                // whatever a failed launch left in the words is not trusted: zero them next time
        return hip_fail(ctx, (hipError_t)e2, "point-cloud launch");
    }
    return SLX_OK;
}

// Device / pinned buffers of the text of up to max_points points in the context's dialect.
static int ensure_text_buffers(slx_ctx *ctx, size_t max_points)
{
    const size_t line_max = ctx->text_dialect == SLX_TEXT_MSVC2013 ? SLX_TEXT_LINE_MAX_MSVC : SLX_TEXT_LINE_MAX;
    const size_t need = max_points * line_max + 16, wgs = (size_t)slx_text_workgroups(max_points);
    if (ctx->d_text_capacity < need) {
        SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_text) (void)hipFree(ctx->d_text);
        ctx->d_text = nullptr;
        ctx->d_text_capacity = 0;
        SLX_HIP(ctx, hipMalloc((void **)&ctx->d_text, need));
        ctx->d_text_capacity = need;
    }
    if (ctx->d_text_sums_capacity < wgs) {
        SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_text_sums) (void)hipFree(ctx->d_text_sums);
        ctx->d_text_sums = nullptr;
        ctx->d_text_sums_capacity = 0;
        // + the length of the text (8-byte aligned) + the run bases of a very large cloud (slx_text_bases_kernel)
        SLX_HIP(ctx, hipMalloc((void **)&ctx->d_text_sums, (wgs + 4) * sizeof(unsigned) + (wgs / SLX_TEXT_BASE_RUN + 2) * sizeof(unsigned long long)));
        ctx->d_text_sums_capacity = wgs;
    }
    if (!ctx->h_text_info) {
        SLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_text_info, kTextInfoWords * sizeof(unsigned long long), hipHostMallocDefault));
        for (size_t k = 0; k < kTextInfoWords; k++) ctx->h_text_info[k] = 0;
    }
    return SLX_OK;
}

// the 8-byte words behind the workgroup lengths: [0] the length of the text, [1 ..] the run bases (null below SLX_TEXT_BASES_FROM workgroups)
static unsigned long long *text_total_word(slx_ctx *ctx) { return (unsigned long long *)(ctx->d_text_sums + ((ctx->d_text_sums_capacity + 1) & ~(size_t)1)); }
static unsigned long long *text_bases(slx_ctx *ctx, size_t max_points)
{
    return slx_text_workgroups(max_points) > SLX_TEXT_BASES_FROM ? text_total_word(ctx) + 1 : nullptr;
}

static int ensure_host_text(slx_ctx *ctx, size_t total)
{
    if (ctx->h_text_capacity >= total) return SLX_OK;
    if (ctx->h_text) (void)hipHostFree(ctx->h_text);
    ctx->h_text = nullptr;
    ctx->h_text_capacity = 0;
    const size_t cap = total + total / 8 + 4096;                      // successive frames differ by a few per cent
    SLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_text, cap, hipHostMallocDefault));
    ctx->h_text_capacity = cap;
    return SLX_OK;
}

int slx_point_cloud_of_depth(slx_ctx *ctx, const double *depth, double *xyz, size_t capacity_points, size_t *n_points, int mem_kind)
{
    if (!ctx || !n_points) return SLX_ERR_INVALID_ARG;
    const slx_config &c = ctx->cfg;
    if (!mode_has_depth(c.mode)) return fail(ctx, SLX_ERR_UNAVAILABLE, "mode %d produces no depth", c.mode);
    if (!depth) return fail(ctx, SLX_ERR_INVALID_ARG, "depth is NULL");
    if ((uintptr_t)depth % sizeof(double)) return fail(ctx, SLX_ERR_INVALID_ARG, "depth is not aligned to 8 bytes");
    if (mem_kind != SLX_MEM_HOST && mem_kind != SLX_MEM_DEVICE) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = order_after_done(ctx, ctx->stream)) return rc;   // the decode may have run on a caller stream: device-side wait only
    const int entries = slx_cloud_entries(c.width, c.height), n_tiles = slx_cloud_tiles(c.width, c.height);
    if (!ctx->d_cloud_tiles) SLX_HIP(ctx, hipMalloc((void **)&ctx->d_cloud_tiles, ((size_t)n_tiles + 1) * sizeof(unsigned)));   // + the total
    if (!ctx->h_cloud_total) {                                      // pinned: [0] the point count, [1] the fused kernel's "gave up" tag
        SLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_cloud_total, 2 * sizeof(unsigned), hipHostMallocDefault));
        ctx->h_cloud_total[0] = ctx->h_cloud_total[1] = 0u;
    }
    unsigned *total_dev = ctx->d_cloud_tiles + n_tiles;
    const double *z = depth;
    // One launch that reads the depth once (slx_cloud.hip) wherever its plan allows; else count + write (the depth read twice)
    SlxCloudFused fq{};
    const bool can_fuse = slx_cloud_fused_plan(c.width, c.height, ctx->kp.n_cus, &fq.groups, &fq.parts, &fq.rows_per_part);
    if (ctx->tune.cloud_passes == 1 && !can_fuse) return fail(ctx, SLX_ERR_UNAVAILABLE, "the fused point-cloud launch has no plan for a %d x %d map on this device", c.width, c.height);
    const bool fused = can_fuse && ctx->tune.cloud_passes != 2;
    unsigned fused_tag = 0;                                         // epoch + 1 of the last fused launch: what its "gave up" flag would carry
    auto launch_cloud = [&](double *target, bool two_launches) -> int {
        if (two_launches) {
            if (!ctx->d_cloud_counts) SLX_HIP(ctx, hipMalloc((void **)&ctx->d_cloud_counts, (size_t)entries * sizeof(unsigned)));
            int e2 = slx_launch_cloud_count(ctx->kp, z, ctx->d_cloud_counts, ctx->d_cloud_tiles, ctx->stream);
            if (e2 != 0) return hip_fail(ctx, (hipError_t)e2, "point-cloud count");
            e2 = slx_launch_cloud_write(ctx->kp, z, ctx->d_cloud_counts, ctx->d_cloud_tiles, target, total_dev, ctx->h_cloud_total, ctx->stream);   // target NULL: the total only
            if (e2 != 0) return hip_fail(ctx, (hipError_t)e2, "point-cloud write");
            return SLX_OK;
        }
        return launch_cloud_fused_async(ctx, fq, z, target, total_dev, &fused_tag);
    };
    // The points can be written at once when the target cannot overflow (a device buffer for every pixel, or the
    // context's own staging buffer, which is sized for every pixel): one pass over the GPU, one wait.
    const size_t all = (size_t)c.width * c.height;
    double *dst = nullptr;
    if (mem_kind == SLX_MEM_DEVICE && xyz && capacity_points >= all) {
        dst = xyz;
    } else if (mem_kind == SLX_MEM_HOST && xyz) {
        if (ctx->cloud_capacity < all) {
            if (ctx->d_cloud) (void)hipFree(ctx->d_cloud);
            ctx->d_cloud = nullptr;
            ctx->cloud_capacity = 0;
            SLX_HIP(ctx, hipMalloc((void **)&ctx->d_cloud, all * 3 * sizeof(double)));
            ctx->cloud_capacity = all;
        }
        dst = ctx->d_cloud;
    }
    // One fused launch, or -- where its plan refuses, on request, or when a workgroup of the fused launch gave up waiting for its
    // look-back (slx_cloud.hip: a bounded spin, so that a dispatcher that does not behave as the kernel assumes costs a repeat of the
    // frame instead of a hung GPU) -- the count + write launches.  Returns with the stream drained and the total in pinned memory.
    auto run_cloud = [&](double *target) -> int {
        if (int rc = launch_cloud(target, !fused)) return rc;
        SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (fused && *(volatile unsigned *)(ctx->h_cloud_total + 1) == fused_tag) {
            ctx->cloud_epoch = 1u << 30;                            // the tagged words of the abandoned launch are not trusted: zeroed before the next fused launch
            ctx->cloud_fallbacks++;
            if (int rc = launch_cloud(target, true)) return rc;
            SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->err = "point cloud: a workgroup of the fused launch gave up waiting for its look-back; the frame was repeated on the count + write launches (result complete)";
        }
        return SLX_OK;
    };
    if (int rc = run_cloud(dst)) return rc;                         // dst NULL: the total only
    const unsigned total = *(volatile unsigned *)ctx->h_cloud_total;   // stored by the write kernel, visible once the stream has drained
    *n_points = total;
    if (total == 0) return SLX_OK;
    if (!xyz || capacity_points < total) return fail(ctx, SLX_ERR_INVALID_ARG, "the cloud has %u points, the buffer holds %zu", total, capacity_points);
    if (!dst) {                                                     // a device buffer smaller than the frame, now known to be large enough
        return run_cloud(xyz);
    } else if (mem_kind == SLX_MEM_HOST) {
        SLX_HIP(ctx, hipMemcpyAsync(xyz, dst, (size_t)total * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    } else {
        return SLX_OK;                                              // already written and waited for
    }
    SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLX_OK;
}

// Text of n packed (x, y, z) triples in device memory, formatted there (slx_text.hip), in the context's pinned buffer.
int slx_format_points_text(slx_ctx *ctx, const double *xyz_dev, size_t n_points, const char **text, size_t *n_bytes)
{
    if (!ctx || !text || !n_bytes) return SLX_ERR_INVALID_ARG;
    *text = "";
    *n_bytes = 0;
    if (n_points == 0) return SLX_OK;
    if (!xyz_dev) return fail(ctx, SLX_ERR_INVALID_ARG, "xyz is NULL");
    if ((uintptr_t)xyz_dev % sizeof(double)) return fail(ctx, SLX_ERR_INVALID_ARG, "xyz is not aligned to 8 bytes");
    const size_t line_max = ctx->text_dialect == SLX_TEXT_MSVC2013 ? SLX_TEXT_LINE_MAX_MSVC : SLX_TEXT_LINE_MAX;
    if (n_points >= (1ull << 40) / line_max) return fail(ctx, SLX_ERR_INVALID_ARG, "%zu points: too many for one text", n_points);
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t need = n_points * line_max + 16;
    if (int rc = ensure_text_buffers(ctx, n_points)) return rc;
    if (int rc = order_after_done(ctx, ctx->stream)) return rc;       // the points may come from a launch on a caller's stream
    unsigned long long *total_dev = text_total_word(ctx);
    if (++ctx->text_tag == 0) ctx->text_tag = 1;                       // (the flag word starts as 0 and keeps the last raised tag)
    // (The emit kernel storing straight into pinned host memory instead -- no device text, no copy, one wait fewer -- was measured: 1.38 ms
    // against 1.18 ms per 56 MB text; the kernel's stores cross PCIe slower than the copy engine.)
    const int e = slx_launch_text(xyz_dev, n_points, ctx->d_text_sums, (unsigned *)&ctx->h_text_info[1], ctx->text_tag, ctx->d_text, total_dev,
                                  &ctx->h_text_info[0], ctx->text_dialect, text_bases(ctx, n_points), ctx->stream);
    if (e != 0) return hip_fail(ctx, (hipError_t)e, "point-cloud text launch");
    SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*(volatile unsigned *)&ctx->h_text_info[1] == ctx->text_tag)
        return fail(ctx, SLX_ERR_UNAVAILABLE, "a coordinate outside the device formatter's range (|v| < 1e-5, |v| >= 1e15, NaN or infinity): format this cloud on the host");
    const size_t total = (size_t)*(volatile unsigned long long *)&ctx->h_text_info[0];
    if (total > need) return fail(ctx, SLX_ERR_HIP, "the device reports %zu bytes of text for %zu points", total, n_points);
    if (int rc = ensure_host_text(ctx, total)) return rc;
    SLX_HIP(ctx, hipMemcpyAsync(ctx->h_text, ctx->d_text, total, hipMemcpyDeviceToHost, ctx->stream));
    SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *text = ctx->h_text;
    *n_bytes = total;
    return SLX_OK;
}

int slx_set_text_dialect(slx_ctx *ctx, int dialect)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (dialect != SLX_TEXT_LIBSTDCXX && dialect != SLX_TEXT_MSVC2013) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown text dialect %d", dialect);
    ctx->text_dialect = dialect;
    return SLX_OK;
}

int slx_get_point_cloud_text(slx_ctx *ctx, const char **text, size_t *n_bytes, size_t *n_points)
{
    if (!ctx || !text || !n_bytes) return SLX_ERR_INVALID_ARG;
    *text = "";
    *n_bytes = 0;
    if (n_points) *n_points = 0;
    if (!mode_has_depth(ctx->cfg.mode)) return fail(ctx, SLX_ERR_UNAVAILABLE, "mode %d produces no depth", ctx->cfg.mode);
    if (!ctx->decoded) return fail(ctx, SLX_ERR_NOT_DECODED, "no decode has run yet");
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t all = (size_t)ctx->cfg.width * (size_t)ctx->cfg.height;
    if (ctx->cloud_capacity < all) {                                  // the context's device buffer for a cloud: one triple per pixel
        SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_cloud) (void)hipFree(ctx->d_cloud);
        ctx->d_cloud = nullptr;
        ctx->cloud_capacity = 0;
        SLX_HIP(ctx, hipMalloc((void **)&ctx->d_cloud, all * 3 * sizeof(double)));
        ctx->cloud_capacity = all;
    }
    // The pipeline (where the fused cloud launch has a plan): cloud -> text lengths -> piece offsets are queued back to back -- the
    // length kernel takes the number of points from the device word the cloud kernel leaves -- and the host waits ONCE for all three;
    // then the characters are formatted piece by piece and every piece crosses PCIe (its own stream) while the next one is formatted.
    // Before: cloud | wait | lengths, characters | wait | copy | wait, i.e. the kernels (0.15 ms) in front of the copy (1.05 ms) instead of beside it.
    const slx_config &c = ctx->cfg;
    SlxCloudFused fq{};
    const unsigned pieces = ctx->tune.text_pieces > 0 ? (unsigned)ctx->tune.text_pieces : 2u;   // measured: 1 piece 1.174 ms, 2 1.167, 4 1.168, 8 1.211, 16 1.297 (every copy costs ~20 us to set up)
    if (pieces >= 2 && all >= 65536 && slx_cloud_fused_plan(c.width, c.height, ctx->kp.n_cus, &fq.groups, &fq.parts, &fq.rows_per_part) && ctx->tune.cloud_passes != 2) {
        const int msvc = ctx->text_dialect == SLX_TEXT_MSVC2013 ? 1 : 0;
        if (int rc = ensure_text_buffers(ctx, all)) return rc;
        const int n_tiles = slx_cloud_tiles(c.width, c.height);
        if (!ctx->d_cloud_tiles) SLX_HIP(ctx, hipMalloc((void **)&ctx->d_cloud_tiles, ((size_t)n_tiles + 1) * sizeof(unsigned)));
        if (!ctx->h_cloud_total) {
            SLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_cloud_total, 2 * sizeof(unsigned), hipHostMallocDefault));
            ctx->h_cloud_total[0] = ctx->h_cloud_total[1] = 0u;
        }
        if (!ctx->text_stream) SLX_HIP(ctx, hipStreamCreateWithFlags(&ctx->text_stream, hipStreamNonBlocking));
        for (unsigned k = 0; k < pieces; k++)
            if (!ctx->ev_text[k]) SLX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_text[k], hipEventDisableTiming));
        if (int rc = order_after_done(ctx, ctx->stream)) return rc;
        unsigned *n_dev = ctx->d_cloud_tiles + n_tiles;
        unsigned long long *total_dev = text_total_word(ctx);
        unsigned long long *bases = text_bases(ctx, all);
        unsigned fused_tag = 0;
        if (int rc = launch_cloud_fused_async(ctx, fq, (const double *)ctx->out[SLX_OUT_Z], ctx->d_cloud, n_dev, &fused_tag)) return rc;
        if (++ctx->text_tag == 0) ctx->text_tag = 1;
        int e = slx_launch_text_lengths(ctx->d_cloud, n_dev, all, ctx->d_text_sums, (unsigned *)&ctx->h_text_info[1], ctx->text_tag, pieces, &ctx->h_text_info[2], total_dev,
                                        msvc, bases, ctx->stream);
        if (e != 0) return hip_fail(ctx, (hipError_t)e, "point-cloud text launch");
        SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));                // the one wait in front of the copy
        if (*(volatile unsigned *)(ctx->h_cloud_total + 1) != fused_tag) {
            const size_t n = *(volatile unsigned *)ctx->h_cloud_total;
            if (n_points) *n_points = n;
            if (n == 0) return SLX_OK;
            if (*(volatile unsigned *)&ctx->h_text_info[1] == ctx->text_tag)
                return fail(ctx, SLX_ERR_UNAVAILABLE, "a coordinate outside the device formatter's range (|v| < 1e-5, |v| >= 1e15, NaN or infinity): format this cloud on the host");
            const unsigned long long *off = ctx->h_text_info + 2;
            const size_t total = (size_t)off[pieces];
            const size_t line_max = msvc ? SLX_TEXT_LINE_MAX_MSVC : SLX_TEXT_LINE_MAX;
            if (total > n * line_max || total < n * 6) return fail(ctx, SLX_ERR_HIP, "the device reports %zu bytes of text for %zu points", total, n);
            if (int rc = ensure_host_text(ctx, total)) return rc;
            const unsigned wgs = (unsigned)slx_text_workgroups(n), per = (wgs + pieces - 1u) / pieces;   // the cut slx_text_bounds_kernel made
            for (unsigned k = 0; k < pieces; k++) {
                const unsigned a0 = std::min(k * per, wgs), b0 = std::min(a0 + per, wgs);
                if (b0 == a0) continue;
                e = slx_launch_text_piece(ctx->d_cloud, n, ctx->d_text_sums, ctx->d_text, total_dev, a0, b0 - a0, msvc, bases, ctx->stream);
                if (e != 0) return hip_fail(ctx, (hipError_t)e, "point-cloud text launch");
                SLX_HIP(ctx, hipEventRecord(ctx->ev_text[k], ctx->stream));
                SLX_HIP(ctx, hipStreamWaitEvent(ctx->text_stream, ctx->ev_text[k], 0));
                // a piece's first and last dword are shared with its neighbours (the kernels write those bytes one by one): the copy takes
                // whole bytes [off[k], off[k + 1]) -- exactly what this piece's workgroups wrote
                if (off[k + 1] > off[k])
                    SLX_HIP(ctx, hipMemcpyAsync(ctx->h_text + off[k], ctx->d_text + off[k], (size_t)(off[k + 1] - off[k]), hipMemcpyDeviceToHost, ctx->text_stream));
            }
            SLX_HIP(ctx, hipStreamSynchronize(ctx->text_stream));       // ... and the one behind it (the pieces' kernels ended before their copies)
            *text = ctx->h_text;
            *n_bytes = total;
            return SLX_OK;
        }
        // a workgroup of the fused cloud launch gave up its look-back (slx_cloud.hip): this frame the slow way, on the count + write launches
        ctx->cloud_epoch = 1u << 30;
        ctx->cloud_fallbacks++;
    }
    size_t n = 0;
    int rc = slx_point_cloud_of_depth(ctx, (const double *)ctx->out[SLX_OUT_Z], ctx->d_cloud, all, &n, SLX_MEM_DEVICE);
    if (rc != SLX_OK) return rc;
    if (n_points) *n_points = n;
    return slx_format_points_text(ctx, ctx->d_cloud, n, text, n_bytes);
}

int slx_get_point_cloud_view(slx_ctx *ctx, const double **xyz, size_t *n_points)
{
    if (!ctx || !xyz || !n_points) return SLX_ERR_INVALID_ARG;
    *xyz = nullptr;
    *n_points = 0;
    if (!mode_has_depth(ctx->cfg.mode)) return fail(ctx, SLX_ERR_UNAVAILABLE, "mode %d produces no depth", ctx->cfg.mode);
    if (!ctx->decoded) return fail(ctx, SLX_ERR_NOT_DECODED, "no decode has run yet");
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t all = (size_t)ctx->cfg.width * (size_t)ctx->cfg.height;
    if (!ctx->h_cloud) SLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_cloud, all * 3 * sizeof(double), hipHostMallocDefault));
    // one pass over the device (count, write into the context's device buffer, the total in a pinned word), one copy of exactly
    // the points into pinned memory: no second pass to learn the size first, no pageable bounce, nothing to zero
    size_t n = 0;
    const int rc = slx_point_cloud_of_depth(ctx, (const double *)ctx->out[SLX_OUT_Z], ctx->h_cloud, all, &n, SLX_MEM_HOST);
    if (rc != SLX_OK) return rc;
    *xyz = ctx->h_cloud;
    *n_points = n;
    return SLX_OK;
}

// Camera image of a dynamic frame on the device: borrowed, or staged.  A host image is copied into one of two pinned
// buffers (the caller's buffer is free again on return, like CSensor::GetCamPicture's deep copy, R/CSensorV.cpp:171-179)
// and goes to the device on the copy stream; the context's stream waits for that copy on the device, not the host, so the
// call returns while the previous frame's kernels are still running and frame n+1's copy-in overlaps them.
// A transfer of a few tens of microseconds: poll before falling back to hipEventSynchronize, whose wake-up after a real wait
// costs more than the transfer itself (measured: 190 us per frame with it, against the 42 us copy).
static hipError_t wait_event_spinning(hipEvent_t ev)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) return hipEventSynchronize(ev);
    }
}

// Makes pinned slot i of the tracker feed writable by the host: allocated, and its transfer of two frames ago finished.
static int track_slot_ready(slx_ctx *ctx, unsigned i)
{
    const slx_config &c = ctx->cfg;
    const size_t bytes = (size_t)c.width * c.height;
    if (!ctx->copy_stream) SLX_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->h_track_img[i]) {
        SLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_track_img[i], bytes, hipHostMallocDefault));
        SLX_HIP(ctx, hipMalloc((void **)&ctx->d_track_img[i], bytes));
        SLX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_track_copied[i], hipEventDisableTiming));
        SLX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_track_used[i], hipEventDisableTiming));
    }
    if (ctx->track_slot_used[i] && !ctx->track_slot_waited[i]) {
        // slot i was last used two frames ago: its kernels must have read the device copy (device-side wait on the copy
        // stream), and its transfer must have left the pinned buffer before the host overwrites it
        SLX_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_track_used[i], 0));
        SLX_HIP(ctx, wait_event_spinning(ctx->ev_track_copied[i]));
        ctx->track_slot_waited[i] = true;
    }
    return SLX_OK;
}

static int track_image(slx_ctx *ctx, const uint8_t *image, size_t stride, int mem_kind, const uint8_t **dev, size_t *dev_stride, int *slot_out)
{
    const slx_config &c = ctx->cfg;
    *slot_out = -1;
    if (!image) return fail(ctx, SLX_ERR_INVALID_ARG, "image is NULL");
    if (stride < (size_t)c.width) return fail(ctx, SLX_ERR_INVALID_ARG, "stride %zu is smaller than the width %d", stride, c.width);
    if (mem_kind == SLX_MEM_DEVICE) {
        *dev = image;
        *dev_stride = stride;
        return SLX_OK;
    }
    if (mem_kind != SLX_MEM_HOST) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
    const unsigned i = ctx->track_slot & 1u;
    if (int rc = track_slot_ready(ctx, i)) return rc;
    ctx->track_slot++;
    const size_t bytes = (size_t)c.width * c.height;
    // an image the caller wrote straight into the slot slx_track_image_buffer handed out is already where the transfer reads it
    if (!(image == ctx->h_track_img[i] && stride == (size_t)c.width))
        for (int r = 0; r < c.height; r++) std::memcpy(ctx->h_track_img[i] + (size_t)r * c.width, image + (size_t)r * stride, (size_t)c.width);
    SLX_HIP(ctx, hipMemcpyAsync(ctx->d_track_img[i], ctx->h_track_img[i], bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    SLX_HIP(ctx, hipEventRecord(ctx->ev_track_copied[i], ctx->copy_stream));
    SLX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_track_copied[i], 0));
    ctx->track_slot_used[i] = true;
    ctx->track_slot_waited[i] = false;
    *dev = ctx->d_track_img[i];
    *dev_stride = (size_t)c.width;
    *slot_out = (int)i;
    return SLX_OK;
}

int slx_track_image_buffer(slx_ctx *ctx, uint8_t **buffer, size_t *stride_bytes)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (!buffer) return fail(ctx, SLX_ERR_INVALID_ARG, "buffer is NULL");
    if (!mode_has_depth(ctx->cfg.mode) || !ctx->out[SLX_OUT_U])
        return fail(ctx, SLX_ERR_UNAVAILABLE, "dynamic frames need a depth mode created with SLX_OUT_U in aux_outputs");
    const unsigned i = ctx->track_slot & 1u;
    if (int rc = track_slot_ready(ctx, i)) return rc;
    *buffer = ctx->h_track_img[i];
    if (stride_bytes) *stride_bytes = (size_t)ctx->cfg.width;
    return SLX_OK;
}

int slx_track_begin(slx_ctx *ctx, const uint8_t *image, size_t stride_bytes, int mem_kind, int window)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    const slx_config &c = ctx->cfg;
    if (!mode_has_depth(c.mode) || !ctx->out[SLX_OUT_U])
        return fail(ctx, SLX_ERR_UNAVAILABLE, "dynamic frames need a depth mode created with SLX_OUT_U in aux_outputs");
    if (!ctx->decoded) return fail(ctx, SLX_ERR_NOT_DECODED, "decode frame 0 first");
    if (window < 3 || window > 201 || (window & 1) == 0) return fail(ctx, SLX_ERR_INVALID_ARG, "window must be odd and in [3,201] (got %d)", window);
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc0 = order_after_done(ctx, ctx->stream)) return rc0;   // frame 0 may have been decoded on a caller stream: device-side wait
    const size_t hw = (size_t)c.width * c.height;
    for (int w : {SLX_OUT_DELTAZ, SLX_OUT_DELTAP, SLX_OUT_STRIPW, SLX_OUT_STRIPB}) {
        if (ctx->out[w]) continue;
        const size_t bytes = hw * out_elem_bytes(w);
        SLX_HIP(ctx, hipMalloc(&ctx->out[w], bytes));
        ctx->out_bytes[w] = bytes;
    }
    for (float **q : {&ctx->d_stripW_prev, &ctx->d_stripB_prev, &ctx->d_deltaP_raw})
        if (!*q) SLX_HIP(ctx, hipMalloc((void **)q, hw * sizeof(float)));
    for (int w : {SLX_OUT_DELTAZ, SLX_OUT_DELTAP}) SLX_HIP(ctx, hipMemsetAsync(ctx->out[w], 0, ctx->out_bytes[w], ctx->stream));
    const uint8_t *img;
    size_t istride;
    int slot;
    int rc = track_image(ctx, image, stride_bytes, mem_kind, &img, &istride, &slot);
    if (rc != SLX_OK) return rc;
    ctx->track_window = window;
    int e = slx_launch_strip_regression(img, istride, c.width, c.height, window, (float *)ctx->out[SLX_OUT_STRIPW], (float *)ctx->out[SLX_OUT_STRIPB], ctx->stream);
    if (e != 0) return hip_fail(ctx, (hipError_t)e, "strip regression");
    if (slot >= 0) SLX_HIP(ctx, hipEventRecord(ctx->ev_track_used[slot], ctx->stream));
    return mark_done(ctx, ctx->stream);
}

// One dynamic frame from a camera image in device memory: StripRegression(fN) + FillOtherDeltaProU(fN) + FillCoordinate(fN)
// (R/CCalculation.cpp:789-892, 595-663, 666-785) on the context's stream.  deltaz: where this frame's deltaZ plane goes.
static int track_step(slx_ctx *ctx, const uint8_t *img, size_t istride, double *deltaz)
{
    const slx_config &c = ctx->cfg;
    // the strips of the previous frame move aside; the new ones start from 0 (R/CCalculation.cpp:827-828)
    float *curW = (float *)ctx->out[SLX_OUT_STRIPW], *curB = (float *)ctx->out[SLX_OUT_STRIPB];
    std::swap(curW, ctx->d_stripW_prev);
    std::swap(curB, ctx->d_stripB_prev);
    ctx->out[SLX_OUT_STRIPW] = curW;
    ctx->out[SLX_OUT_STRIPB] = curB;
    // (the kernels write every pixel of both planes, zeros outside the interior)
    int e;
    const bool divisors_in_range = std::fabs(ctx->kp.fu) > 0x1p-90 && std::fabs(ctx->kp.fv) > 0x1p-90 && slx_fast_arith_ok(ctx->kp);
    if (divisors_in_range && slx_track_fusable(c.width, c.height, ctx->track_window)) {
        e = slx_launch_track_fused(ctx->kp, img, istride, curW, curB, ctx->d_stripW_prev, ctx->d_stripB_prev, (float *)ctx->out[SLX_OUT_DELTAP],
                                   (double *)ctx->out[SLX_OUT_U], (double *)ctx->out[SLX_OUT_Z], (double *)ctx->out[SLX_OUT_X],
                                   (double *)ctx->out[SLX_OUT_Y], deltaz, ctx->stream);
    } else {
        e = slx_launch_strip_regression(img, istride, c.width, c.height, ctx->track_window, curW, curB, ctx->stream, ctx->d_stripW_prev,
                                        ctx->d_stripB_prev, ctx->d_deltaP_raw);
        if (e == 0)
            e = slx_launch_track_update(ctx->kp, ctx->d_deltaP_raw, (float *)ctx->out[SLX_OUT_DELTAP], (double *)ctx->out[SLX_OUT_U],
                                        (double *)ctx->out[SLX_OUT_Z], (double *)ctx->out[SLX_OUT_X], (double *)ctx->out[SLX_OUT_Y],
                                        deltaz, ctx->stream);
    }
    if (e != 0) return hip_fail(ctx, (hipError_t)e, "dynamic-frame kernels");
    return SLX_OK;
}

// A kernel reading `img` has just been queued on the context's stream: when the image lies in one of the staging slabs, that is
// the work the slab's next transfer has to wait for.
static int note_slab_read(slx_ctx *ctx, const uint8_t *img)
{
    const size_t cap = (size_t)ctx->track_slab_frames * (size_t)ctx->cfg.width * (size_t)ctx->cfg.height;
    for (int k = 0; k < 2; k++)
        if (ctx->d_track_slab[k] && img >= ctx->d_track_slab[k] && img < ctx->d_track_slab[k] + cap)
            SLX_HIP(ctx, hipEventRecord(ctx->ev_slab_used[k], ctx->stream));
    return SLX_OK;
}

int slx_track_next(slx_ctx *ctx, const uint8_t *image, size_t stride_bytes, int mem_kind)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (ctx->track_window == 0) return fail(ctx, SLX_ERR_NOT_CONFIGURED, "slx_track_begin has not been called");
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc0 = order_after_done(ctx, ctx->stream)) return rc0;
    const uint8_t *img;
    size_t istride;
    int slot;
    int rc = track_image(ctx, image, stride_bytes, mem_kind, &img, &istride, &slot);
    if (rc != SLX_OK) return rc;
    rc = track_step(ctx, img, istride, (double *)ctx->out[SLX_OUT_DELTAZ]);
    if (rc != SLX_OK) return rc;
    if (slot >= 0) SLX_HIP(ctx, hipEventRecord(ctx->ev_track_used[slot], ctx->stream));
    if (int rs = note_slab_read(ctx, img)) return rs;
    return mark_done(ctx, ctx->stream);
}

// The slab pair of slx_track_stage_frames: allocated for at least n frames, slab i's last transfer and the kernels that read it done.
static int track_slab_ready(slx_ctx *ctx, unsigned i, int n_frames)
{
    const size_t bytes = (size_t)ctx->cfg.width * ctx->cfg.height;
    if (!ctx->copy_stream) SLX_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (n_frames > ctx->track_slab_frames) {
        // grow both slabs: nothing may still be using the old ones
        SLX_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        if (int rc = wait_done_host(ctx)) return rc;
        SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < 2; k++) {
            if (ctx->h_track_slab[k]) (void)hipHostFree(ctx->h_track_slab[k]);
            if (ctx->d_track_slab[k]) (void)hipFree(ctx->d_track_slab[k]);
            ctx->h_track_slab[k] = ctx->d_track_slab[k] = nullptr;
            ctx->slab_used[k] = false;
        }
        ctx->track_slab_frames = 0;
        for (int k = 0; k < 2; k++) {
            SLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_track_slab[k], bytes * (size_t)n_frames, hipHostMallocDefault));
            SLX_HIP(ctx, hipMalloc((void **)&ctx->d_track_slab[k], bytes * (size_t)n_frames));
            if (!ctx->ev_slab_copied[k]) SLX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_slab_copied[k], hipEventDisableTiming));
            if (!ctx->ev_slab_used[k]) SLX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_slab_used[k], hipEventDisableTiming));
        }
        ctx->track_slab_frames = n_frames;
    }
    if (ctx->slab_used[i]) {
        // slab i was staged two calls ago.  The kernels that read its device copy (slx_track_next_batch, or slx_track_next on an
        // image inside the slab: both record ev_slab_used after their launches) must be done before the next transfer overwrites
        // it -- a device-side wait on the copy stream, which leaves the OTHER slab's kernels free to run beside this transfer --
        // and the slab's own transfer must have left the pinned half before the host writes there
        SLX_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_slab_used[i], 0));
        SLX_HIP(ctx, wait_event_spinning(ctx->ev_slab_copied[i]));
    }
    return SLX_OK;
}

int slx_track_frames_buffer(slx_ctx *ctx, int n_frames, uint8_t **buffer, size_t *stride_bytes, size_t *image_stride_bytes)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (!buffer) return fail(ctx, SLX_ERR_INVALID_ARG, "buffer is NULL");
    if (n_frames < 1 || n_frames > SLX_TRACK_MAX_BATCH) return fail(ctx, SLX_ERR_INVALID_ARG, "n_frames must be in [1,%d] (got %d)", SLX_TRACK_MAX_BATCH, n_frames);
    if (!mode_has_depth(ctx->cfg.mode) || !ctx->out[SLX_OUT_U])
        return fail(ctx, SLX_ERR_UNAVAILABLE, "dynamic frames need a depth mode created with SLX_OUT_U in aux_outputs");
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned i = ctx->track_slab & 1u;
    if (int rc = track_slab_ready(ctx, i, n_frames)) return rc;
    *buffer = ctx->h_track_slab[i];
    if (stride_bytes) *stride_bytes = (size_t)ctx->cfg.width;
    if (image_stride_bytes) *image_stride_bytes = (size_t)ctx->cfg.width * ctx->cfg.height;
    return SLX_OK;
}

int slx_track_stage_frames(slx_ctx *ctx, const uint8_t *images, size_t stride_bytes, size_t image_stride_bytes, int n_frames,
                           const uint8_t **device_images)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    const slx_config &c = ctx->cfg;
    if (!images || !device_images) return fail(ctx, SLX_ERR_INVALID_ARG, "images / device_images is NULL");
    if (n_frames < 1 || n_frames > SLX_TRACK_MAX_BATCH) return fail(ctx, SLX_ERR_INVALID_ARG, "n_frames must be in [1,%d] (got %d)", SLX_TRACK_MAX_BATCH, n_frames);
    if (stride_bytes < (size_t)c.width) return fail(ctx, SLX_ERR_INVALID_ARG, "stride %zu is smaller than the width %d", stride_bytes, c.width);
    if (n_frames > 1 && image_stride_bytes < stride_bytes * (size_t)c.height) return fail(ctx, SLX_ERR_INVALID_ARG, "image stride %zu overlaps images", image_stride_bytes);
    if (!mode_has_depth(c.mode) || !ctx->out[SLX_OUT_U])
        return fail(ctx, SLX_ERR_UNAVAILABLE, "dynamic frames need a depth mode created with SLX_OUT_U in aux_outputs");
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned i = ctx->track_slab & 1u;
    if (n_frames > ctx->track_slab_frames) {
        // the slabs are about to be replaced by larger ones: a pointer into a slab slx_track_frames_buffer handed out for fewer
        // frames would dangle the moment they are freed (and the images behind it with them)
        const size_t cap = (size_t)c.width * c.height * (size_t)ctx->track_slab_frames;
        for (int k = 0; k < 2; k++)
            if (ctx->h_track_slab[k] && images >= ctx->h_track_slab[k] && images < ctx->h_track_slab[k] + cap)
                return fail(ctx, SLX_ERR_INVALID_ARG, "images points into the pinned slab slx_track_frames_buffer handed out for %d frames; ask it for %d first",
                            ctx->track_slab_frames, n_frames);
    }
    if (int rc = track_slab_ready(ctx, i, n_frames)) return rc;
    ctx->track_slab++;
    const size_t bytes = (size_t)c.width * c.height;
    // images the caller wrote straight into the slab slx_track_frames_buffer handed out are already where the transfer reads them
    if (!(images == ctx->h_track_slab[i] && stride_bytes == (size_t)c.width && (n_frames == 1 || image_stride_bytes == bytes)))
        for (int f = 0; f < n_frames; f++)
            for (int r = 0; r < c.height; r++)
                std::memcpy(ctx->h_track_slab[i] + (size_t)f * bytes + (size_t)r * c.width, images + (size_t)f * image_stride_bytes + (size_t)r * stride_bytes,
                            (size_t)c.width);
    // ONE transfer for all of them (a 2.3 MB image per transfer pays ~20 us of hand-off between its end and the kernel that waits
    // for it, every frame; k images pay it once)
    SLX_HIP(ctx, hipMemcpyAsync(ctx->d_track_slab[i], ctx->h_track_slab[i], bytes * (size_t)n_frames, hipMemcpyHostToDevice, ctx->copy_stream));
    SLX_HIP(ctx, hipEventRecord(ctx->ev_slab_copied[i], ctx->copy_stream));
    SLX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_slab_copied[i], 0));
    // until a kernel reads the slab, "used" is the point where the stream learnt of the transfer
    SLX_HIP(ctx, hipEventRecord(ctx->ev_slab_used[i], ctx->stream));
    ctx->slab_used[i] = true;
    *device_images = ctx->d_track_slab[i];
    return SLX_OK;
}

int slx_track_next_batch(slx_ctx *ctx, const uint8_t *images, size_t stride_bytes, size_t image_stride_bytes, int n_frames, int mem_kind,
                         double *deltaz_all, int deltaz_mem_kind)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    const slx_config &c = ctx->cfg;
    if (ctx->track_window == 0) return fail(ctx, SLX_ERR_NOT_CONFIGURED, "slx_track_begin has not been called");
    if (!images) return fail(ctx, SLX_ERR_INVALID_ARG, "images is NULL");
    if (n_frames < 1 || n_frames > SLX_TRACK_MAX_BATCH) return fail(ctx, SLX_ERR_INVALID_ARG, "n_frames must be in [1,%d] (got %d)", SLX_TRACK_MAX_BATCH, n_frames);
    if (mem_kind != SLX_MEM_HOST && mem_kind != SLX_MEM_DEVICE) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
    if (stride_bytes < (size_t)c.width) return fail(ctx, SLX_ERR_INVALID_ARG, "stride %zu is smaller than the width %d", stride_bytes, c.width);
    if (n_frames > 1 && image_stride_bytes < stride_bytes * (size_t)c.height) return fail(ctx, SLX_ERR_INVALID_ARG, "image stride %zu overlaps images", image_stride_bytes);
    if (deltaz_all && deltaz_mem_kind != SLX_MEM_HOST && deltaz_mem_kind != SLX_MEM_DEVICE) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown deltaz_mem_kind %d", deltaz_mem_kind);
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc0 = order_after_done(ctx, ctx->stream)) return rc0;
    const size_t hw = (size_t)c.width * c.height;
    // every frame's deltaZ for a host caller: collected in a device buffer of the call, copied out at its end (synchronous, like slx_get_output)
    double *host_dz = nullptr, *dz_dev = deltaz_all;
    if (deltaz_all && deltaz_mem_kind == SLX_MEM_HOST) {
        host_dz = deltaz_all;
        SLX_HIP(ctx, hipMalloc((void **)&dz_dev, (size_t)n_frames * hw * sizeof(double)));
    }
    struct Scratch {                                   // freed on every way out
        double *p;
        ~Scratch() { if (p) (void)hipFree(p); }
    } scratch{host_dz ? dz_dev : nullptr};
    const uint8_t *dev = images;
    size_t dstride = stride_bytes, dimage = image_stride_bytes;
    if (mem_kind == SLX_MEM_HOST) {
        int rc = slx_track_stage_frames(ctx, images, stride_bytes, image_stride_bytes, n_frames, &dev);
        if (rc != SLX_OK) return rc;
        dstride = (size_t)c.width;
        dimage = (size_t)c.width * c.height;
    }
    for (int f = 0; f < n_frames; f++) {
        // every frame's deltaZ goes straight into its plane of the collection; the last frame's also stays in the context (SLX_OUT_DELTAZ)
        double *dz = (dz_dev && f + 1 < n_frames) ? dz_dev + (size_t)f * hw : (double *)ctx->out[SLX_OUT_DELTAZ];
        int rc = track_step(ctx, dev + (size_t)f * dimage, dstride, dz);
        if (rc != SLX_OK) return rc;
    }
    if (int rs = note_slab_read(ctx, dev)) return rs;
    if (dz_dev)
        SLX_HIP(ctx, hipMemcpyAsync(dz_dev + (size_t)(n_frames - 1) * hw, ctx->out[SLX_OUT_DELTAZ], hw * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    int rc = mark_done(ctx, ctx->stream);
    if (rc == SLX_OK && host_dz) {
        SLX_HIP(ctx, hipMemcpyAsync(host_dz, dz_dev, (size_t)n_frames * hw * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return rc;
}

// ---- frame ingest pipeline ---------------------------------------------------------
// Three streams, one event per stage and slot:
//   copy-in  (h2d stream):    hipMemcpyAsync pinned -> device input               -> ev_in
//   decode   (compute stream): waits ev_in, slx_decode_batch on the slot's buffers  -> ev_dec
//   copy-out (d2h stream):    waits ev_dec, hipMemcpyAsync device result -> pinned  -> ev_out
// A slot's buffers are touched again only after slx_pipe_collect has waited for its last event, so consecutive slots
// overlap freely: while slot k decodes, slot k+1 copies in and slot k-1 copies out (PCIe is full duplex).
struct slx_pipe {
    enum State { FREE, ACQUIRED, SUBMITTED, COLLECTED };
    struct Slot {
        uint8_t *h_in = nullptr, *d_in = nullptr;
        double *h_out = nullptr, *d_out = nullptr;
        hipEvent_t ev_in = nullptr, ev_dec = nullptr, ev_out = nullptr;
        State state = FREE;
        int n_sets = 0;
        unsigned long long ticket = 0;                             // submit order
    };
    slx_ctx *ctx = nullptr;
    int device = 0;                                                // the context's device, kept here: slx_pipe_destroy may run after slx_destroy
    slx_pipe_config cfg{};
    std::vector<Slot> slot;
    hipStream_t s_in = nullptr, s_dec = nullptr, s_out = nullptr;
    int n_phase = 0, n_gray = 0;
    size_t pitch = 0, plane_bytes = 0, set_bytes = 0, out_set_bytes = 0;
    int acquired = -1;
    unsigned long long next_ticket = 1;
    std::string err;
};

namespace {

int pipe_fail(slx_pipe *p, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (p) p->err = buf;
    else g_create_error = buf;
    return code;
}

#define SLX_PIPE_HIP(p, call)                                                                                   \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            return pipe_fail(p, e_ == hipErrorOutOfMemory ? SLX_ERR_OUT_OF_MEMORY : SLX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

void slx_pipe_destroy(slx_pipe *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);                                 // not through p->ctx: the context may be gone already
    for (hipStream_t s : {p->s_in, p->s_dec, p->s_out})
        if (s) (void)hipStreamSynchronize(s);
    for (auto &sl : p->slot) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        for (hipEvent_t e : {sl.ev_in, sl.ev_dec, sl.ev_out})
            if (e) (void)hipEventDestroy(e);
    }
    for (hipStream_t s : {p->s_in, p->s_dec, p->s_out})
        if (s) (void)hipStreamDestroy(s);
    delete p;
}

int slx_pipe_create(slx_ctx *ctx, const slx_pipe_config *cfg, slx_pipe **out)
{
    if (!ctx || !cfg || !out) return SLX_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->slots < 2 || cfg->slots > 64) return fail(ctx, SLX_ERR_INVALID_ARG, "a pipe needs 2..64 slots (got %d)", cfg->slots);
    if (cfg->sets_per_slot < 1 || cfg->sets_per_slot > 4096) return fail(ctx, SLX_ERR_INVALID_ARG, "sets_per_slot must be in [1,4096] (got %d)", cfg->sets_per_slot);
    slx_pipe *p = new slx_pipe;
    p->ctx = ctx;
    p->device = ctx->device;
    p->cfg = *cfg;
    p->n_phase = (int)ctx->phase.size();
    p->n_gray = (int)ctx->gray.size();
    p->pitch = ctx->staging_pitch;
    p->plane_bytes = p->pitch * (size_t)ctx->cfg.height;
    p->set_bytes = p->plane_bytes * (size_t)(p->n_phase + p->n_gray);
    p->out_set_bytes = (size_t)ctx->cfg.width * (size_t)ctx->cfg.height * sizeof(double);
    p->slot.resize((size_t)cfg->slots);
    auto bail = [&](int rc) {
        ctx->err = p->err;
        slx_pipe_destroy(p);
        return rc;
    };
#define SLX_PIPE_TRY(call)                                                                           \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            pipe_fail(p, 0, "%s: %s", #call, hipGetErrorString(e_));                                 \
            return bail(e_ == hipErrorOutOfMemory ? SLX_ERR_OUT_OF_MEMORY : SLX_ERR_HIP);            \
        }                                                                                            \
    } while (0)
    SLX_PIPE_TRY(hipSetDevice(ctx->device));
    SLX_PIPE_TRY(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
    SLX_PIPE_TRY(hipStreamCreateWithFlags(&p->s_dec, hipStreamNonBlocking));
    SLX_PIPE_TRY(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    const size_t in_bytes = p->set_bytes * (size_t)cfg->sets_per_slot, out_bytes = p->out_set_bytes * (size_t)cfg->sets_per_slot;
    for (auto &sl : p->slot) {
        SLX_PIPE_TRY(hipHostMalloc((void **)&sl.h_in, in_bytes, hipHostMallocDefault));
        SLX_PIPE_TRY(hipMalloc((void **)&sl.d_in, in_bytes));
        SLX_PIPE_TRY(hipMalloc((void **)&sl.d_out, out_bytes));
        if (cfg->host_result) SLX_PIPE_TRY(hipHostMalloc((void **)&sl.h_out, out_bytes, hipHostMallocDefault));
        SLX_PIPE_TRY(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
        SLX_PIPE_TRY(hipEventCreateWithFlags(&sl.ev_dec, hipEventDisableTiming));
        SLX_PIPE_TRY(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
    }
#undef SLX_PIPE_TRY
    *out = p;
    return SLX_OK;
}

const char *slx_pipe_last_error(const slx_pipe *p) { return p ? p->err.c_str() : g_create_error.c_str(); }

int slx_pipe_layout(const slx_pipe *p, int *n_planes, size_t *pitch, size_t *plane_bytes, size_t *set_bytes)
{
    if (!p) return SLX_ERR_INVALID_ARG;
    if (n_planes) *n_planes = p->n_phase + p->n_gray;
    if (pitch) *pitch = p->pitch;
    if (plane_bytes) *plane_bytes = p->plane_bytes;
    if (set_bytes) *set_bytes = p->set_bytes;
    return SLX_OK;
}

int slx_pipe_acquire(slx_pipe *p, uint8_t **host_in)
{
    if (!p || !host_in) return SLX_ERR_INVALID_ARG;
    if (p->acquired >= 0) return pipe_fail(p, SLX_ERR_INVALID_ARG, "slot %d is already acquired: submit it first", p->acquired);
    // a free slot, else the collected slot whose result is the oldest (its reader has had it the longest)
    int pick = -1;
    for (size_t i = 0; i < p->slot.size(); i++)
        if (p->slot[i].state == slx_pipe::FREE) { pick = (int)i; break; }
    if (pick < 0) {
        for (size_t i = 0; i < p->slot.size(); i++)
            if (p->slot[i].state == slx_pipe::COLLECTED && (pick < 0 || p->slot[i].ticket < p->slot[(size_t)pick].ticket)) pick = (int)i;
    }
    if (pick < 0) return pipe_fail(p, SLX_ERR_NOT_CONFIGURED, "all %zu slots are in flight: collect one first", p->slot.size());
    p->slot[(size_t)pick].state = slx_pipe::ACQUIRED;
    p->acquired = pick;
    *host_in = p->slot[(size_t)pick].h_in;
    return SLX_OK;
}

int slx_pipe_submit(slx_pipe *p, int n_sets)
{
    if (!p) return SLX_ERR_INVALID_ARG;
    if (p->acquired < 0) return pipe_fail(p, SLX_ERR_INVALID_ARG, "no slot is acquired");
    if (n_sets < 1 || n_sets > p->cfg.sets_per_slot) return pipe_fail(p, SLX_ERR_INVALID_ARG, "n_sets must be in [1,%d] (got %d)", p->cfg.sets_per_slot, n_sets);
    slx_pipe::Slot &sl = p->slot[(size_t)p->acquired];
    slx_ctx *ctx = p->ctx;
    SLX_PIPE_HIP(p, hipSetDevice(ctx->device));
    SLX_PIPE_HIP(p, hipMemcpyAsync(sl.d_in, sl.h_in, p->set_bytes * (size_t)n_sets, hipMemcpyHostToDevice, p->s_in));
    SLX_PIPE_HIP(p, hipEventRecord(sl.ev_in, p->s_in));
    SLX_PIPE_HIP(p, hipStreamWaitEvent(p->s_dec, sl.ev_in, 0));
    const uint8_t *phase_base = p->n_phase ? sl.d_in : nullptr;
    const uint8_t *gray_base = p->n_gray ? sl.d_in + (size_t)p->n_phase * p->plane_bytes : nullptr;
    const int rc = slx_decode_batch(ctx, n_sets, phase_base, p->set_bytes, gray_base, p->set_bytes, p->pitch, sl.d_out, p->s_dec);
    if (rc != SLX_OK) return pipe_fail(p, rc, "decode: %s", ctx->err.c_str());
    SLX_PIPE_HIP(p, hipEventRecord(sl.ev_dec, p->s_dec));
    if (sl.h_out) {
        SLX_PIPE_HIP(p, hipStreamWaitEvent(p->s_out, sl.ev_dec, 0));
        SLX_PIPE_HIP(p, hipMemcpyAsync(sl.h_out, sl.d_out, p->out_set_bytes * (size_t)n_sets, hipMemcpyDeviceToHost, p->s_out));
        SLX_PIPE_HIP(p, hipEventRecord(sl.ev_out, p->s_out));
    }
    sl.n_sets = n_sets;
    sl.ticket = p->next_ticket++;
    sl.state = slx_pipe::SUBMITTED;
    p->acquired = -1;
    return SLX_OK;
}

int slx_pipe_collect(slx_pipe *p, const double **host_result, const double **device_result, int *n_sets)
{
    if (!p) return SLX_ERR_INVALID_ARG;
    int pick = -1;
    for (size_t i = 0; i < p->slot.size(); i++)
        if (p->slot[i].state == slx_pipe::SUBMITTED && (pick < 0 || p->slot[i].ticket < p->slot[(size_t)pick].ticket)) pick = (int)i;
    if (pick < 0) return pipe_fail(p, SLX_ERR_NOT_DECODED, "nothing has been submitted");
    slx_pipe::Slot &sl = p->slot[(size_t)pick];
    SLX_PIPE_HIP(p, hipSetDevice(p->ctx->device));
    SLX_PIPE_HIP(p, hipEventSynchronize(sl.h_out ? sl.ev_out : sl.ev_dec));
    sl.state = slx_pipe::COLLECTED;
    if (host_result) *host_result = sl.h_out;
    if (device_result) *device_result = sl.d_out;
    if (n_sets) *n_sets = sl.n_sets;
    return SLX_OK;
}

int slx_get_calibration(const slx_ctx *ctx, double P[12], double *cA, double *cB)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (!mode_has_depth(ctx->cfg.mode)) return SLX_ERR_UNAVAILABLE;
    if (P) std::memcpy(P, ctx->P, sizeof ctx->P);
    if (cA) *cA = ctx->cA;
    if (cB) *cB = ctx->cB;
    return SLX_OK;
}

int slx_enable_timing(slx_ctx *ctx, int on)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    ctx->timed = on != 0;
    return SLX_OK;
}

int slx_last_decode_ms(slx_ctx *ctx, float *ms)
{
    if (!ctx || !ms) return SLX_ERR_INVALID_ARG;
    if (!ctx->timed) return fail(ctx, SLX_ERR_UNAVAILABLE, "timing is off (slx_enable_timing)");
    SLX_HIP(ctx, hipSetDevice(ctx->device));
    SLX_HIP(ctx, hipEventSynchronize(ctx->ev1));
    SLX_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return SLX_OK;
}

int slx_debug_stamps(slx_ctx *ctx, unsigned long long *device_words, size_t n_words)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (device_words && n_words < 4 * 8192) return fail(ctx, SLX_ERR_INVALID_ARG, "stamp buffer needs at least 32768 words");
    ctx->kp.stamps = device_words;
    ctx->kp.stamp_items = device_words ? n_words / 4 : 0;
    return SLX_OK;
}

int slx_last_kernel(slx_ctx *ctx, char *buf, size_t buf_bytes)
{
    if (!ctx || !buf || buf_bytes == 0) return SLX_ERR_INVALID_ARG;
    const SlxStreamState &st = ctx->stream_state;
    static const char *const names[] = {"none", "slx_fused_kernel", "slx_strip_kernel", "slx_stream_kernel", "slx_decoder_strip_kernel"};
    // the instantiation, written the way rocprofv3's kernel trace demangles it
    char inst[64];
    const char *aux = st.last_aux ? "true" : "false";
    switch (st.last_kind) {
    case 1: snprintf(inst, sizeof inst, "%s<%d, %d, %s, %s>", names[1], st.last_mode, st.last_freq, (st.last_steps == 4 || st.last_mode == SLX_MODE_GRAY_ONLY) ? "true" : "false", st.last_mode >= SLX_MODE_GRAY_PHASE ? aux : "false"); break;
    case 2: snprintf(inst, sizeof inst, "%s<%d, %d, %d, %d, %s>", names[2], st.last_mode, st.last_freq, st.last_gray_ring_bits, st.last_steps, aux); break;
    case 3: snprintf(inst, sizeof inst, "%s<%d>", names[3], st.last_freq); break;
    case 4: snprintf(inst, sizeof inst, "%s<%d>", names[4], st.last_mode); break;
    case 5: snprintf(inst, sizeof inst, "slx_gstream_kernel"); break;
    default: snprintf(inst, sizeof inst, "%s", names[0]); break;
    }
    if (st.last_kind == 3 || st.last_kind == 5) snprintf(buf, buf_bytes, "%s: resident waves, %d-row items from queues", inst, st.last_rows);
    else if (st.last_kind == 2 || st.last_kind == 4) snprintf(buf, buf_bytes, "%s: %d-row items, %d rows per row group", inst, st.last_rows, st.last_weave);
    else snprintf(buf, buf_bytes, "%s", inst);
    return SLX_OK;
}

int slx_set_variant(slx_ctx *ctx, int variant)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    if (variant < 0 || variant >= slx_num_variants()) return fail(ctx, SLX_ERR_INVALID_ARG, "variant %d outside [0,%d)", variant, slx_num_variants());
    ctx->variant = variant;
    return SLX_OK;
}

int slx_internal_device(const slx_ctx *ctx) { return ctx->device; }
void *slx_internal_stream(const slx_ctx *ctx) { return (void *)ctx->stream; }
void slx_internal_tile(const slx_ctx *ctx, int *width, int *height)
{
    *width = ctx->cfg.width;
    *height = ctx->cfg.height;
}

int slx_set_tuning(slx_ctx *ctx, int key, int value)
{
    if (!ctx) return SLX_ERR_INVALID_ARG;
    struct Range { int *field; int lo, hi; };
    SlxTuning &t = ctx->tune;
    const Range r[SLX_TUNE_COUNT] = {{&t.strip_rows, 0, 32}, {&t.tail_pct, -1, 99}, {&t.tail_rows, 0, 32}, {&t.gray_plain, 0, 1},
                                     {&t.strip_waves, 0, 4}, {&t.lds_pad_kib, 0, 128}, {&t.plain_order, 0, 1}, {&t.tiers, 0, SLX_MAX_TIERS},
                                     {&t.weave, 0, 64}, {&t.stream, 0, 2}, {&t.stream_rows, 0, 16}, {&t.cloud_passes, 0, 2}, {&t.cloud_spin, 0, 1 << 30}, {&t.text_pieces, 0, SLX_TEXT_MAX_PIECES}};
    if (key < 0 || key >= SLX_TUNE_COUNT) return fail(ctx, SLX_ERR_INVALID_ARG, "unknown tuning key %d", key);
    if (value < r[key].lo || value > r[key].hi) return fail(ctx, SLX_ERR_INVALID_ARG, "tuning key %d takes values in [%d,%d] (got %d)", key, r[key].lo, r[key].hi, value);
    *r[key].field = value;
    return SLX_OK;
}

}  // extern "C"
