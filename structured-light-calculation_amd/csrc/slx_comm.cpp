// slx_comm.cpp -- the one collective of the path: the gather of finished depth maps over RCCL / xGMI.
//
// The reference is a single process (no distributed backend); BASELINE.json's north_star spreads a
// batch over the 8 GPUs of a node by row tile and gathers the depth maps at the end.  Every decode stage is
// pixel-independent, so this file is the only place where ranks talk to each other.
//
// xGMI on MI355X is point to point (7 links per GPU): a gather to one root is bound by the root's 7 ingest links, and
// a ring would be bound by ONE link.  So the gather is a single group of ncclSend / ncclRecv -- every peer sends straight
// to the root at once, one message per (peer, frame-set) that lands at the tile's row offset inside that set of the
// full [set][H][W] array.  No staging buffer, no transpose, no second pass over HBM.
//
// That is the IN-PLACE shape (SLX_GATHER_IN_PLACE, the default).  A row split makes it many messages: 8 ranks x 256 frame-sets =
// 1 792 messages of 2.3 MB per step at the root.  The STAGED shape (SLX_GATHER_STAGED, gathers to one root) trades them for one
// contiguous message per (peer, chunk) -- 224 of 18.4 MB for the same step in chunks of 8 -- into a staging slot of the root, and a
// row-scatter kernel (slx_gather.hip) that moves the tiles to their rows while the next chunk's messages arrive in the other
// slot: one more pass over the root's HBM (read + write of what arrived) against 8 x fewer, 8 x longer messages.  Both shapes come
// out of the same planner (plan_range) and deliver the same bytes; which one wins is a property of the fabric and of RCCL's
// per-message cost, so bench.py times both.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "slx.h"
#include "slx_kernels.h"

namespace {
thread_local std::string g_comm_create_error;
}

struct slx_comm {
    slx_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool owned = false;
    int world = 0, rank = 0, device = 0;
    hipStream_t stream = nullptr;            // the gather stream
    std::vector<hipEvent_t> ev_chunk;        // decode of chunk i finished (recorded on the decode stream)
    hipEvent_t ev_gathered = nullptr;        // everything queued on the gather stream so far has finished
    bool gathered_pending = false;
    // the staged shape: two staging slots on the receiving rank, filled and scattered alternately
    int shape = SLX_GATHER_IN_PLACE;
    hipStream_t scatter_stream = nullptr;
    double *stage[2] = {nullptr, nullptr};
    size_t stage_doubles = 0;                // capacity of each slot
    hipEvent_t ev_recv[2] = {nullptr, nullptr};        // the group that filled slot i has completed (gather stream)
    hipEvent_t ev_scattered[2] = {nullptr, nullptr};   // slot i has been scattered into the full array (scatter stream)
    bool scattered_pending[2] = {false, false};
    unsigned stage_turn = 0;
    std::string err;
};

namespace {

int cfail(slx_comm *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_comm_create_error = buf;
    return code;
}

#define SLXC_HIP(c, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return cfail(c, SLX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));          \
    } while (0)
#define SLXC_NCCL(c, call)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) return cfail(c, SLX_ERR_HIP, "%s: %s", #call, ncclGetErrorString(r_));        \
    } while (0)

// The shard table must describe a partition every rank can act on without talking: the same frame width everywhere,
// rows inside the frame, no rank beyond the communicator.
int check_shards(slx_comm *c, const slx_shard *shards, int height, int width, int root)
{
    if (!shards) return cfail(c, SLX_ERR_INVALID_ARG, "shards is NULL");
    if (height <= 0 || width <= 0) return cfail(c, SLX_ERR_INVALID_ARG, "height/width must be positive");
    if (root < -1 || root >= c->world) return cfail(c, SLX_ERR_INVALID_ARG, "root %d outside [-1,%d)", root, c->world);
    for (int r = 0; r < c->world; r++) {
        const slx_shard &s = shards[r];
        if (s.n_sets < 0 || s.set0 < 0 || s.rows < 0 || s.row0 < 0 || s.row0 + s.rows > height)
            return cfail(c, SLX_ERR_INVALID_ARG, "shard %d (sets %d+%d, rows %d+%d) does not fit a %d-row frame", r, s.set0, s.n_sets, s.row0, s.rows, height);
    }
    return SLX_OK;
}

// The messages of one group, as data: what gather_range below posts, in the order it posts them (receives first, then sends).
// Kept apart from the posting so that the schedule can be checked without a GPU: tests/test_gather_plan.py plays every rank's
// plan against the others' for worlds of 2..8 -- every send must meet a receive of the same length, in the same order per
// pair of ranks, and the replayed copies must reassemble [set][H][W].
// shape SLX_GATHER_STAGED (a gather to ONE root; with root < 0 every rank receives in place and the shape is the in-place one):
// a row-tile peer's frame-sets of the range travel as one message into the root's staging slot (slx_msg.send == 2, offset in
// doubles into the slot), and `scat` lists how the slot is then scattered into the full array; the sender's tile stack must be
// dense.  Whole-frame shards are one message per peer in either shape and land in place.
int plan_range(const slx_shard *shards, int world, int me, int height, int width, int first, int count, size_t local_plane_stride, int root, int shape,
               std::vector<slx_msg> &out, std::vector<slx_scatter> &scat, unsigned long long &staging_doubles, std::string &why)
{
    const bool staged = shape == SLX_GATHER_STAGED && root >= 0;
    staging_doubles = 0;
    const bool i_receive = root < 0 || root == me;
    const size_t W = (size_t)width, H = (size_t)height;
    auto clip = [&](const slx_shard &sh, int &lo, int &n) {          // sets of the shard that fall into the range
        lo = std::min(first, sh.n_sets);
        n = std::min(first + count, sh.n_sets) - lo;
    };
    const slx_shard &mine = shards[me];
    const size_t lstride = local_plane_stride ? local_plane_stride : (size_t)mine.rows * W;
    int my_lo, my_n;
    clip(mine, my_lo, my_n);
    // whole-frame shards travel as ONE message per peer (the receiver posts one receive for the run of sets), so they must be dense
    if ((size_t)mine.rows == H && lstride != H * W) {
        why = "a whole-frame shard must be dense";
        return SLX_ERR_INVALID_ARG;
    }
    if (i_receive) {
        for (int p = 0; p < world; p++) {
            if (p == me) continue;
            const slx_shard &sh = shards[p];
            int lo, n;
            clip(sh, lo, n);
            if (n <= 0 || sh.rows == 0) continue;
            if ((size_t)sh.rows == H) {                              // whole frames: the peer's sets are one contiguous run
                out.push_back({p, 0, (unsigned long long)((size_t)(sh.set0 + lo) * H * W), (unsigned long long)((size_t)n * H * W)});
            } else if (staged) {                                     // the peer's n tiles as one message into the staging slot
                const unsigned long long run = (unsigned long long)sh.rows * W;
                out.push_back({p, 2, staging_doubles, (unsigned long long)n * run});
                scat.push_back({staging_doubles, (unsigned long long)(((size_t)(sh.set0 + lo) * H + (size_t)sh.row0) * W), run, (unsigned long long)n, run,
                                (unsigned long long)(H * W)});
                staging_doubles += (unsigned long long)n * run;
            } else {
                for (int k = 0; k < n; k++)
                    out.push_back({p, 0, (unsigned long long)(((size_t)(sh.set0 + lo + k) * H + (size_t)sh.row0) * W), (unsigned long long)((size_t)sh.rows * W)});
            }
        }
    }
    if (my_n > 0 && mine.rows > 0) {
        for (int d = 0; d < world; d++) {
            if (d == me || !(root < 0 || root == d)) continue;
            if ((size_t)mine.rows == H) {
                out.push_back({d, 1, (unsigned long long)((size_t)my_lo * lstride), (unsigned long long)((size_t)my_n * H * W)});
            } else if (staged) {
                if (lstride != (size_t)mine.rows * W) {
                    why = "the staged gather sends a chunk's tiles as one message: the sending rank's tile stack must be dense";
                    return SLX_ERR_INVALID_ARG;
                }
                out.push_back({d, 1, (unsigned long long)((size_t)my_lo * lstride), (unsigned long long)((size_t)my_n * (size_t)mine.rows * W)});
            } else {
                for (int k = 0; k < my_n; k++)
                    out.push_back({d, 1, (unsigned long long)((size_t)(my_lo + k) * lstride), (unsigned long long)((size_t)mine.rows * W)});
            }
        }
    }
    return SLX_OK;
}

// One group of sends / receives for the frame-sets [first, first + count) of every rank's shard, counted within the shard
// (chunk c of a pipelined gather = sets [c * chunk, (c+1) * chunk) of each shard).  `local` addresses this rank's shard as
// documented at slx_gather_depth.
int gather_range(slx_comm *c, const slx_shard *shards, int height, int width, int first, int count, const double *local,
                 size_t local_plane_stride, double *full, int root, hipStream_t s)
{
    const int me = c->rank;
    const bool i_receive = root < 0 || root == me;
    const size_t W = (size_t)width, H = (size_t)height;
    const slx_shard &mine = shards[me];
    const size_t lstride = local_plane_stride ? local_plane_stride : (size_t)mine.rows * W;
    const int my_lo = std::min(first, mine.n_sets), my_n = std::min(first + count, mine.n_sets) - my_lo;
    const bool in_place = i_receive && local == full + ((size_t)mine.set0 * H + (size_t)mine.row0) * W && lstride == H * W;
    std::vector<slx_msg> plan;
    std::vector<slx_scatter> scat;
    unsigned long long staging = 0;
    std::string why;
    int rc = plan_range(shards, c->world, me, height, width, first, count, local_plane_stride, root, c->shape, plan, scat, staging, why);
    if (rc != SLX_OK) return cfail(c, rc, "%s", why.c_str());

    // the staging slot of this group (staged shape, receiving rank): large enough, and scattered empty since its last use
    double *slot_base = nullptr;
    unsigned slot = 0;
    if (!scat.empty()) {
        if (!c->scatter_stream) {
            SLXC_HIP(c, hipStreamCreateWithFlags(&c->scatter_stream, hipStreamNonBlocking));
            for (int k = 0; k < 2; k++) {
                SLXC_HIP(c, hipEventCreateWithFlags(&c->ev_recv[k], hipEventDisableTiming));
                SLXC_HIP(c, hipEventCreateWithFlags(&c->ev_scattered[k], hipEventDisableTiming));
            }
        }
        if ((size_t)staging > c->stage_doubles) {
            // grow both slots: nothing may still be arriving in or leaving the old ones
            SLXC_HIP(c, hipStreamSynchronize(s));
            SLXC_HIP(c, hipStreamSynchronize(c->stream));
            SLXC_HIP(c, hipStreamSynchronize(c->scatter_stream));
            for (int k = 0; k < 2; k++) {
                if (c->stage[k]) (void)hipFree(c->stage[k]);
                c->stage[k] = nullptr;
                c->scattered_pending[k] = false;
            }
            c->stage_doubles = 0;
            for (int k = 0; k < 2; k++) SLXC_HIP(c, hipMalloc((void **)&c->stage[k], (size_t)staging * sizeof(double)));
            c->stage_doubles = (size_t)staging;
        }
        slot = c->stage_turn++ & 1u;
        slot_base = c->stage[slot];
        if (c->scattered_pending[slot]) SLXC_HIP(c, hipStreamWaitEvent(s, c->ev_scattered[slot], 0));
    }

    SLXC_NCCL(c, ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    for (const slx_msg &m : plan) {
        if (r != ncclSuccess) break;
        if (m.send == 1) r = ncclSend(local + m.offset, (size_t)m.count, ncclDouble, m.peer, c->comm, s);
        else if (m.send == 2) r = ncclRecv(slot_base + m.offset, (size_t)m.count, ncclDouble, m.peer, c->comm, s);
        else r = ncclRecv(full + m.offset, (size_t)m.count, ncclDouble, m.peer, c->comm, s);
    }
    const ncclResult_t rg = ncclGroupEnd();
    if (r != ncclSuccess) return cfail(c, SLX_ERR_HIP, "ncclSend/ncclRecv: %s", ncclGetErrorString(r));
    if (rg != ncclSuccess) return cfail(c, SLX_ERR_HIP, "ncclGroupEnd: %s", ncclGetErrorString(rg));
    if (!scat.empty()) {
        // the tiles go to their rows on the scatter stream, behind this group and beside the next one (which fills the other slot)
        SLXC_HIP(c, hipEventRecord(c->ev_recv[slot], s));
        SLXC_HIP(c, hipStreamWaitEvent(c->scatter_stream, c->ev_recv[slot], 0));
        for (size_t k0 = 0; k0 < scat.size(); k0 += SLX_SCATTER_MAX_SEGS) {
            SlxScatterSegs segs;
            segs.n = (int)std::min<size_t>(SLX_SCATTER_MAX_SEGS, scat.size() - k0);
            for (int k = 0; k < segs.n; k++) {
                const slx_scatter &q = scat[k0 + (size_t)k];
                segs.seg[k] = {q.src, q.dst, q.run, q.n_runs, q.src_stride, q.dst_stride};
            }
            const int e = slx_launch_row_scatter(segs, slot_base, full, c->scatter_stream);
            if (e != 0) return cfail(c, SLX_ERR_HIP, "row scatter launch: %s", hipGetErrorString((hipError_t)e));
        }
        SLXC_HIP(c, hipEventRecord(c->ev_scattered[slot], c->scatter_stream));
        c->scattered_pending[slot] = true;
    }
    // this rank's own tile, when it was not decoded in place
    if (i_receive && full && local && !in_place && my_n > 0 && mine.rows > 0) {
        SLXC_HIP(c, hipMemcpy2DAsync(full + ((size_t)(mine.set0 + my_lo) * H + (size_t)mine.row0) * W, H * W * sizeof(double),
                                     local + (size_t)my_lo * lstride, lstride * sizeof(double), (size_t)mine.rows * W * sizeof(double),
                                     (size_t)my_n, hipMemcpyDeviceToDevice, s));
    }
    return SLX_OK;
}

// Everything the scatter stream still owes the full array is ordered before whatever is queued on `s` next.
int join_scatters(slx_comm *c, hipStream_t s)
{
    for (int k = 0; k < 2; k++)
        if (c->scattered_pending[k]) SLXC_HIP(c, hipStreamWaitEvent(s, c->ev_scattered[k], 0));
    return SLX_OK;
}

int finish_create(slx_comm *c, slx_comm **out)
{
    SLXC_HIP(c, hipSetDevice(c->device));
    SLXC_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    SLXC_HIP(c, hipEventCreateWithFlags(&c->ev_gathered, hipEventDisableTiming));
    *out = c;
    return SLX_OK;
}

}  // namespace

extern "C" {

int slx_comm_unique_id(void *id, size_t id_bytes)
{
    if (!id || id_bytes < SLX_COMM_ID_BYTES) return cfail(nullptr, SLX_ERR_INVALID_ARG, "id buffer must hold %d bytes", SLX_COMM_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) == SLX_COMM_ID_BYTES, "SLX_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId u;
    SLXC_NCCL(nullptr, ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return SLX_OK;
}

void slx_comm_destroy(slx_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->scatter_stream) (void)hipStreamSynchronize(c->scatter_stream);
    if (c->owned && c->comm) (void)ncclCommDestroy(c->comm);
    for (int k = 0; k < 2; k++) {
        if (c->stage[k]) (void)hipFree(c->stage[k]);
        if (c->ev_recv[k]) (void)hipEventDestroy(c->ev_recv[k]);
        if (c->ev_scattered[k]) (void)hipEventDestroy(c->ev_scattered[k]);
    }
    if (c->scatter_stream) (void)hipStreamDestroy(c->scatter_stream);
    for (hipEvent_t e : c->ev_chunk)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_gathered) (void)hipEventDestroy(c->ev_gathered);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int slx_comm_create(slx_ctx *ctx, const void *id, size_t id_bytes, int world, int rank, slx_comm **out)
{
    if (!out) return cfail(nullptr, SLX_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!ctx || !id || id_bytes < SLX_COMM_ID_BYTES) return cfail(nullptr, SLX_ERR_INVALID_ARG, "ctx / id missing, or id shorter than %d bytes", SLX_COMM_ID_BYTES);
    if (world < 1 || rank < 0 || rank >= world) return cfail(nullptr, SLX_ERR_INVALID_ARG, "bad rank %d of %d", rank, world);
    slx_comm *c = new slx_comm;
    c->ctx = ctx;
    c->world = world;
    c->rank = rank;
    c->device = slx_internal_device(ctx);
    c->owned = true;
    auto bail = [&](int rc) {
        g_comm_create_error = c->err;
        slx_comm_destroy(c);
        return rc;
    };
    if (hipSetDevice(c->device) != hipSuccess) return bail(cfail(c, SLX_ERR_HIP, "hipSetDevice(%d) failed", c->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        return bail(cfail(c, SLX_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, c->device, ncclGetErrorString(r)));
    }
    // world and rank as the communicator RCCL built reports them -- what slx_comm_info returns and the bench prints as
    // "rccl_world_size" -- not the numbers the caller passed in; a communicator that disagrees with them is refused
    int got_world = -1, got_rank = -1;
    if (ncclCommCount(c->comm, &got_world) != ncclSuccess || ncclCommUserRank(c->comm, &got_rank) != ncclSuccess)
        return bail(cfail(c, SLX_ERR_HIP, "ncclCommCount / ncclCommUserRank failed on the new communicator"));
    if (got_world != world || got_rank != rank)
        return bail(cfail(c, SLX_ERR_HIP, "RCCL built a communicator of %d ranks (this one %d), asked for %d (%d)", got_world, got_rank, world, rank));
    c->world = got_world;
    c->rank = got_rank;
    int rc = finish_create(c, out);
    return rc == SLX_OK ? rc : bail(rc);
}

int slx_comm_adopt(slx_ctx *ctx, void *nccl_comm, slx_comm **out)
{
    if (!out) return cfail(nullptr, SLX_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!ctx || !nccl_comm) return cfail(nullptr, SLX_ERR_INVALID_ARG, "ctx / communicator is NULL");
    slx_comm *c = new slx_comm;
    c->ctx = ctx;
    c->comm = (ncclComm_t)nccl_comm;
    c->owned = false;
    c->device = slx_internal_device(ctx);
    auto bail = [&](int rc) {
        g_comm_create_error = c->err;
        slx_comm_destroy(c);
        return rc;
    };
    int dev = -1;
    if (ncclCommCount(c->comm, &c->world) != ncclSuccess || ncclCommUserRank(c->comm, &c->rank) != ncclSuccess || ncclCommCuDevice(c->comm, &dev) != ncclSuccess)
        return bail(cfail(c, SLX_ERR_INVALID_ARG, "not a usable ncclComm_t"));
    if (dev != c->device) return bail(cfail(c, SLX_ERR_INVALID_ARG, "the communicator lives on device %d, the context on device %d", dev, c->device));
    int rc = finish_create(c, out);
    return rc == SLX_OK ? rc : bail(rc);
}

int slx_gather_plan_ex(const slx_shard *shards, int world, int rank, int height, int width, int first, int count, size_t local_plane_stride,
                       int root, int shape, slx_msg *out, int capacity, int *n_out, slx_scatter *scatter_out, int scatter_capacity, int *n_scatter_out,
                       unsigned long long *staging_doubles)
{
    if (!shards || !n_out || world < 1 || rank < 0 || rank >= world || height <= 0 || width <= 0 || root < -1 || root >= world) return SLX_ERR_INVALID_ARG;
    if (shape != SLX_GATHER_IN_PLACE && shape != SLX_GATHER_STAGED) return SLX_ERR_INVALID_ARG;
    std::vector<slx_msg> plan;
    std::vector<slx_scatter> scat;
    unsigned long long staging = 0;
    std::string why;
    const int rc = plan_range(shards, world, rank, height, width, first, count, local_plane_stride, root, shape, plan, scat, staging, why);
    if (rc != SLX_OK) return rc;
    *n_out = (int)plan.size();
    if (out)
        for (int i = 0; i < (int)plan.size() && i < capacity; i++) out[i] = plan[(size_t)i];
    if (n_scatter_out) *n_scatter_out = (int)scat.size();
    if (scatter_out)
        for (int i = 0; i < (int)scat.size() && i < scatter_capacity; i++) scatter_out[i] = scat[(size_t)i];
    if (staging_doubles) *staging_doubles = staging;
    return SLX_OK;
}

int slx_gather_plan(const slx_shard *shards, int world, int rank, int height, int width, int first, int count, size_t local_plane_stride,
                    int root, slx_msg *out, int capacity, int *n_out)
{
    return slx_gather_plan_ex(shards, world, rank, height, width, first, count, local_plane_stride, root, SLX_GATHER_IN_PLACE, out, capacity, n_out, nullptr, 0,
                              nullptr, nullptr);
}

int slx_scatter_rows(slx_ctx *ctx, const slx_scatter *scatter, int n, const double *staging, double *full, void *stream)
{
    if (!ctx || n < 0 || (n > 0 && (!scatter || !staging || !full))) return SLX_ERR_INVALID_ARG;
    if (hipSetDevice(slx_internal_device(ctx)) != hipSuccess) return SLX_ERR_HIP;
    hipStream_t s = stream ? (hipStream_t)stream : (hipStream_t)slx_internal_stream(ctx);
    for (int k0 = 0; k0 < n; k0 += SLX_SCATTER_MAX_SEGS) {
        SlxScatterSegs segs;
        segs.n = std::min(SLX_SCATTER_MAX_SEGS, n - k0);
        for (int k = 0; k < segs.n; k++) {
            const slx_scatter &q = scatter[k0 + k];
            if (q.n_runs > 65535ull) return SLX_ERR_INVALID_ARG;
            segs.seg[k] = {q.src, q.dst, q.run, q.n_runs, q.src_stride, q.dst_stride};
        }
        if (slx_launch_row_scatter(segs, staging, full, s) != 0) return SLX_ERR_HIP;
    }
    return SLX_OK;
}

int slx_comm_set_gather_shape(slx_comm *c, int shape)
{
    if (!c) return SLX_ERR_INVALID_ARG;
    if (shape != SLX_GATHER_IN_PLACE && shape != SLX_GATHER_STAGED) return cfail(c, SLX_ERR_INVALID_ARG, "unknown gather shape %d", shape);
    c->shape = shape;
    return SLX_OK;
}

// Asks the communicator itself (ncclCommCount / ncclCommUserRank), every time: the numbers are RCCL's, not remembered arguments.
int slx_comm_info(const slx_comm *c, int *world, int *rank)
{
    if (!c || !c->comm) return SLX_ERR_INVALID_ARG;
    int w = 0, r = 0;
    if (ncclCommCount(c->comm, &w) != ncclSuccess || ncclCommUserRank(c->comm, &r) != ncclSuccess) return SLX_ERR_HIP;
    if (world) *world = w;
    if (rank) *rank = r;
    return SLX_OK;
}

const char *slx_comm_last_error(const slx_comm *c) { return c ? c->err.c_str() : g_comm_create_error.c_str(); }

int slx_comm_synchronize(slx_comm *c)
{
    if (!c) return SLX_ERR_INVALID_ARG;
    SLXC_HIP(c, hipSetDevice(c->device));
    SLXC_HIP(c, hipStreamSynchronize(c->stream));
    return SLX_OK;
}

int slx_gather_depth(slx_comm *c, const slx_shard *shards, int height, int width, const double *local, size_t local_plane_stride,
                     double *full, int root, void *stream)
{
    if (!c) return SLX_ERR_INVALID_ARG;
    int rc = check_shards(c, shards, height, width, root);
    if (rc != SLX_OK) return rc;
    const bool i_receive = root < 0 || root == c->rank;
    if (i_receive && !full) return cfail(c, SLX_ERR_INVALID_ARG, "full is NULL on a destination rank");
    if (!local && shards[c->rank].n_sets > 0 && shards[c->rank].rows > 0) return cfail(c, SLX_ERR_INVALID_ARG, "local is NULL but this rank holds a shard");
    SLXC_HIP(c, hipSetDevice(c->device));
    int most = 0;
    for (int r = 0; r < c->world; r++) most = std::max(most, shards[r].n_sets);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    rc = gather_range(c, shards, height, width, 0, most, local, local_plane_stride, full, root, s);
    if (rc != SLX_OK) return rc;
    if ((rc = join_scatters(c, s)) != SLX_OK) return rc;             // staged shape: the tiles are in place before anything queued behind this call
    if (s == c->stream) {
        SLXC_HIP(c, hipEventRecord(c->ev_gathered, c->stream));
        c->gathered_pending = true;
    }
    return SLX_OK;
}

int slx_decode_gather(slx_comm *c, slx_ctx *ctx, const slx_shard *shards, int full_height, int chunk_sets, const uint8_t *phase_base,
                      size_t phase_set_stride, const uint8_t *gray_base, size_t gray_set_stride, size_t row_stride, double *scratch,
                      double *full, int root, void *stream)
{
    if (!c || !ctx) return SLX_ERR_INVALID_ARG;
    if (slx_internal_device(ctx) != c->device) return cfail(c, SLX_ERR_INVALID_ARG, "the communicator lives on device %d, the context on device %d", c->device, slx_internal_device(ctx));
    if (chunk_sets < 1) return cfail(c, SLX_ERR_INVALID_ARG, "chunk_sets must be positive (got %d)", chunk_sets);
    int width = 0, tile_rows = 0;
    slx_internal_tile(ctx, &width, &tile_rows);
    int rc = check_shards(c, shards, full_height, width, root);
    if (rc != SLX_OK) return rc;
    const slx_shard &mine = shards[c->rank];
    if (mine.rows != tile_rows) return cfail(c, SLX_ERR_INVALID_ARG, "this rank's shard has %d rows, its context decodes %d", mine.rows, tile_rows);
    const bool i_receive = root < 0 || root == c->rank;
    if (i_receive && !full) return cfail(c, SLX_ERR_INVALID_ARG, "full is NULL on a destination rank");
    if (!i_receive && !scratch && mine.n_sets > 0) return cfail(c, SLX_ERR_INVALID_ARG, "scratch is NULL on a rank that does not receive");
    SLXC_HIP(c, hipSetDevice(c->device));
    hipStream_t ds = stream ? (hipStream_t)stream : (hipStream_t)slx_internal_stream(ctx);

    const size_t W = (size_t)width, H = (size_t)full_height;
    // where this rank decodes: in place on a destination rank, else the scratch tile stack
    double *local = i_receive ? full + ((size_t)mine.set0 * H + (size_t)mine.row0) * W : scratch;
    const size_t lstride = i_receive ? H * W : (size_t)mine.rows * W;

    int most = 0;
    for (int r = 0; r < c->world; r++) most = std::max(most, shards[r].n_sets);
    const int n_chunks = (most + chunk_sets - 1) / chunk_sets;
    while ((int)c->ev_chunk.size() < n_chunks) {
        hipEvent_t e;
        SLXC_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_chunk.push_back(e);
    }
    // the previous gather may still be sending from / receiving into the buffers this decode is about to overwrite
    if (c->gathered_pending) SLXC_HIP(c, hipStreamWaitEvent(ds, c->ev_gathered, 0));
    for (int k = 0; k < n_chunks; k++) {
        const int first = k * chunk_sets;
        const int n = std::min(first + chunk_sets, mine.n_sets) - std::min(first, mine.n_sets);
        if (n > 0) {
            slx_batch_out out{};
            out.z = local + (size_t)first * lstride;
            out.plane_stride = lstride;
            rc = slx_decode_batch_ex(ctx, n, phase_base ? phase_base + (size_t)first * phase_set_stride : nullptr, phase_set_stride,
                                     gray_base ? gray_base + (size_t)first * gray_set_stride : nullptr, gray_set_stride, row_stride, &out, ds);
            if (rc != SLX_OK) return cfail(c, rc, "decode of chunk %d: %s", k, slx_last_error(ctx));
        }
        SLXC_HIP(c, hipEventRecord(c->ev_chunk[(size_t)k], ds));
        SLXC_HIP(c, hipStreamWaitEvent(c->stream, c->ev_chunk[(size_t)k], 0));
        // chunk k travels on the gather stream while chunk k+1 decodes on ds
        rc = gather_range(c, shards, full_height, width, first, chunk_sets, local, lstride, full, root, c->stream);
        if (rc != SLX_OK) return rc;
    }
    if ((rc = join_scatters(c, c->stream)) != SLX_OK) return rc;     // staged shape: the last chunks' tiles are in place before "gathered"
    SLXC_HIP(c, hipEventRecord(c->ev_gathered, c->stream));
    c->gathered_pending = true;
    return SLX_OK;
}

}  // extern "C"
