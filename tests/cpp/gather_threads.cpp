// gather_threads.cpp -- TEST INFRASTRUCTURE: every rank of the multi-GPU host loop as a THREAD of one process on one GPU, with an
// in-process loopback standing in for the wire.
//
// The 1-GPU boxes cannot host an RCCL run with more than one rank (ncclCommInitRank refuses two ranks on one device), so the
// library's gather code (csrc/slx_comm.cpp: gather_range, the staging slots of the staged shape, the scatter stream, the events
// that order slot reuse, the chunk pipeline of slx_decode_gather) never runs with a peer there.  This program runs exactly that
// code -- libslx.so as built, through the C ABI -- and replaces only the transport: the dozen nccl* entry points the library
// calls are defined HERE (an executable's definitions come before a shared library's in symbol resolution, so libslx's calls land
// on them; librccl stays loaded and unused).  Semantics kept from NCCL: operations are queued between ncclGroupStart / End and
// take effect at the outermost End; a send meets the receive of the same ordered pair of ranks in posting order; the counts must
// agree; data is read / written in stream order (an event on the poster's stream marks "ready", the copy runs on a stream of its
// own behind both sides' events, both sides' streams wait for the copy).  Not kept: anything about time.  ncclGroupEnd blocks the
// host until this rank's operations have met their peers -- every rank posts its groups in the same order, as NCCL requires.
//
//   gather_threads <world> <rows|framesets> <in_place|staged> <W> <H> <sets> <chunk> <in.bin> <out.bin>
// in.bin: [sets][12][H][W] u8; out.bin: rank 0's gathered [sets][H][W] f64 (after it has been checked against a second gather
// from separate local buffers through slx_gather_depth, chunk by chunk).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "slx.h"

// ------------------------------------------------------------------------------------------------ the loopback "RCCL"
namespace {

struct Posted {                        // one side of a message, waiting for the other
    void *buf;
    size_t bytes;
    hipEvent_t ready;                  // recorded on the poster's stream: its buffer may be read / written from here on
    hipEvent_t *done_out;              // where the poster finds the copy's completion event
    bool *matched;
};

struct World {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int joined = 0;
    hipStream_t copy_stream = nullptr;
    std::map<std::pair<int, int>, std::deque<Posted>> sends, recvs;    // (src, dst) -> FIFO
    std::string error;
};

std::mutex g_worlds_m;
std::map<std::string, World *> g_worlds;
std::atomic<int> g_next_id{1};
std::atomic<bool> g_abort{false};        // a rank has failed: nobody waits for it any longer

struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<std::pair<ncclComm_t, Op>> t_ops;

}  // namespace

struct ncclComm {
    World *w;
    int rank;
};

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "loopback-%d", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    World *w;
    {
        std::lock_guard<std::mutex> g(g_worlds_m);
        World *&slot = g_worlds[std::string(id.internal, sizeof id.internal)];
        if (!slot) {
            slot = new World;
            slot->world = nranks;
            if (hipStreamCreateWithFlags(&slot->copy_stream, hipStreamNonBlocking) != hipSuccess) return ncclUnhandledCudaError;
        }
        w = slot;
    }
    if (w->world != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lk(w->m);
    w->joined++;
    w->cv.notify_all();
    w->cv.wait(lk, [&] { return w->joined >= w->world || g_abort.load(); });   // communicator creation is collective
    if (g_abort.load()) return ncclInternalError;
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { *count = comm->w->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) { *rank = comm->rank; return ncclSuccess; }
ncclResult_t ncclCommCuDevice(const ncclComm_t, int *device) { return hipGetDevice(device) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "loopback transport error"; }
ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }

static size_t type_bytes(ncclDataType_t t) { return t == ncclDouble ? 8 : t == ncclFloat ? 4 : 1; }

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (t_depth == 0) return ncclInvalidUsage;                      // the library always groups
    t_ops.push_back({comm, Op{true, const_cast<void *>(buf), count * type_bytes(type), peer, stream}});
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (t_depth == 0) return ncclInvalidUsage;
    t_ops.push_back({comm, Op{false, buf, count * type_bytes(type), peer, stream}});
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (--t_depth > 0) return ncclSuccess;
    std::vector<std::pair<ncclComm_t, Op>> ops;
    ops.swap(t_ops);
    const size_t n = ops.size();
    std::vector<hipEvent_t> done(n, nullptr);
    std::vector<char> matched(n, 0);
    ncclResult_t rc = ncclSuccess;
    for (size_t i = 0; i < n; i++) {
        World *w = ops[i].first->w;
        const Op &op = ops[i].second;
        const int me = ops[i].first->rank;
        if (op.peer < 0 || op.peer >= w->world || op.peer == me) return ncclInvalidArgument;
        hipEvent_t ready;
        if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(ready, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        const std::pair<int, int> key = op.send ? std::make_pair(me, op.peer) : std::make_pair(op.peer, me);
        std::lock_guard<std::mutex> g(w->m);
        auto &mine = op.send ? w->sends[key] : w->recvs[key];
        auto &theirs = op.send ? w->recvs[key] : w->sends[key];
        mine.push_back(Posted{op.buf, op.bytes, ready, &done[i], reinterpret_cast<bool *>(&matched[i])});
        // messages of a pair meet in posting order: match the heads while both queues have one
        auto &S = w->sends[key];
        auto &R = w->recvs[key];
        while (!S.empty() && !R.empty()) {
            Posted s = S.front(), r = R.front();
            S.pop_front();
            R.pop_front();
            hipEvent_t fin = nullptr;
            if (s.bytes != r.bytes) {
                w->error = "a send of " + std::to_string(s.bytes) + " bytes met a receive of " + std::to_string(r.bytes);
            } else if (hipStreamWaitEvent(w->copy_stream, s.ready, 0) != hipSuccess || hipStreamWaitEvent(w->copy_stream, r.ready, 0) != hipSuccess ||
                       hipMemcpyAsync(r.buf, s.buf, s.bytes, hipMemcpyDeviceToDevice, w->copy_stream) != hipSuccess ||
                       hipEventCreateWithFlags(&fin, hipEventDisableTiming) != hipSuccess || hipEventRecord(fin, w->copy_stream) != hipSuccess) {
                w->error = "the loopback copy failed";
            }
            *s.done_out = fin;
            *r.done_out = fin;                                      // (one event for both sides; never destroyed: a test process)
            *s.matched = true;
            *r.matched = true;
        }
        (void)theirs;
        w->cv.notify_all();
    }
    // this rank's operations take effect in stream order behind the copies: wait (on the host) until every one has met its peer,
    // then make its stream wait for the copy
    for (size_t i = 0; i < n; i++) {
        World *w = ops[i].first->w;
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return matched[i] != 0 || !w->error.empty() || g_abort.load(); });
        if (g_abort.load() && !matched[i] && w->error.empty()) w->error = "a peer rank failed";
        if (!w->error.empty()) { std::fprintf(stderr, "loopback: %s\n", w->error.c_str()); rc = ncclInternalError; continue; }
        lk.unlock();
        if (done[i] && hipStreamWaitEvent(ops[i].second.stream, done[i], 0) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ a rank
#define RK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); return 1; } } while (0)
#define RK_SLX(x, who) do { int rc_ = (x); if (rc_ != SLX_OK) { std::fprintf(stderr, "rank %d: %s: %d %s\n", rank, #x, rc_, who); return 1; } } while (0)

struct Job {
    int world, W, H, sets, chunk, shape;
    std::string split;
    const std::vector<uint8_t> *all;
    char id[SLX_COMM_ID_BYTES];
    std::vector<double> out;                                        // rank 0's gathered maps
};

static int run_rank(int rank, Job &job)
{
    const int world = job.world, W = job.W, H = job.H, sets = job.sets;
    RK_HIP(hipSetDevice(0));
    std::vector<slx_shard> shards((size_t)world);
    for (int r = 0; r < world; r++) {
        auto cut = [&](int n, int &lo, int &cnt) { const int b = n / world, rem = n % world; lo = r * b + (r < rem ? r : rem); cnt = b + (r < rem ? 1 : 0); };
        if (job.split == "rows") { shards[r].set0 = 0; shards[r].n_sets = sets; cut(H, shards[r].row0, shards[r].rows); }
        else { cut(sets, shards[r].set0, shards[r].n_sets); shards[r].row0 = 0; shards[r].rows = H; }
    }
    const slx_shard mine = shards[(size_t)rank];
    slx_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.width = W; cfg.height = mine.rows > 0 ? mine.rows : 1; cfg.row_offset = mine.row0;
    cfg.mode = SLX_MODE_MULTIFREQ; cfg.n_freq = 3; cfg.n_steps = 4;
    cfg.period[0] = 1920; cfg.period[1] = 240; cfg.period[2] = 30;
    cfg.fov_min = -1e300; cfg.fov_max = 1e300; cfg.device = 0;
    const double cam[9] = {3600, 0, (W - 1) / 2.0, 0, 3600, (H - 1) / 2.0, 0, 0, 1}, pro[9] = {3000, 0, 900, 0, 3000, 600, 0, 0, 1};
    const double rot[9] = {0.99, -0.01, 0.13, 0.02, 0.99, -0.1, -0.13, 0.1, 0.98}, trans[3] = {-31.7, -9.3, 39.4};
    std::memcpy(cfg.cam, cam, sizeof cam); std::memcpy(cfg.pro, pro, sizeof pro); std::memcpy(cfg.rot, rot, sizeof rot); std::memcpy(cfg.trans, trans, sizeof trans);
    slx_ctx *ctx = nullptr;
    RK_SLX(slx_create(&cfg, &ctx), slx_last_error(nullptr));
    slx_comm *comm = nullptr;
    RK_SLX(slx_comm_create(ctx, job.id, sizeof job.id, world, rank, &comm), slx_comm_last_error(nullptr));
    RK_SLX(slx_comm_set_gather_shape(comm, job.shape), slx_comm_last_error(comm));

    const size_t full_plane = (size_t)W * H, tile_plane = (size_t)W * mine.rows;
    std::vector<uint8_t> tile((size_t)mine.n_sets * 12 * tile_plane);
    for (int s = 0; s < mine.n_sets; s++)
        for (int p = 0; p < 12; p++)
            std::memcpy(tile.data() + ((size_t)s * 12 + p) * tile_plane, job.all->data() + ((size_t)(mine.set0 + s) * 12 + p) * full_plane + (size_t)mine.row0 * W, tile_plane);
    uint8_t *d_in = nullptr;
    double *d_full = nullptr, *d_scratch = nullptr, *d_local = nullptr;
    RK_HIP(hipMalloc((void **)&d_in, tile.size() ? tile.size() : 1));
    RK_HIP(hipMemcpy(d_in, tile.data(), tile.size(), hipMemcpyHostToDevice));
    const size_t full_elems = (size_t)sets * full_plane, local_elems = (size_t)mine.n_sets * tile_plane;
    if (rank == 0) RK_HIP(hipMalloc((void **)&d_full, full_elems * sizeof(double)));
    else RK_HIP(hipMalloc((void **)&d_scratch, (local_elems ? local_elems : 1) * sizeof(double)));
    if (rank == 0) RK_HIP(hipMemset(d_full, 0xff, full_elems * sizeof(double)));
    RK_HIP(hipDeviceSynchronize());

    // decode + gather, pipelined in chunks; twice (the second call must wait for the first one's gather before it overwrites, and the
    // staging slots are reused)
    for (int rep = 0; rep < 2; rep++) {
        if (mine.rows > 0)
            RK_SLX(slx_decode_gather(comm, ctx, shards.data(), H, job.chunk, d_in, 12 * tile_plane, nullptr, 0, (size_t)W, d_scratch, d_full, 0, nullptr),
                   slx_comm_last_error(comm));
        else
            RK_SLX(slx_gather_depth(comm, shards.data(), H, W, nullptr, 0, d_full, 0, nullptr), slx_comm_last_error(comm));   // an empty tile: nothing to decode
    }
    RK_SLX(slx_comm_synchronize(comm), slx_comm_last_error(comm));
    std::vector<double> a(rank == 0 ? full_elems : 0), b(a.size());
    if (rank == 0) RK_HIP(hipMemcpy(a.data(), d_full, a.size() * sizeof(double), hipMemcpyDeviceToHost));

    // the same from separate local buffers through ONE plain gather per rank: must deliver the same array
    RK_HIP(hipMalloc((void **)&d_local, (local_elems ? local_elems : 1) * sizeof(double)));
    if (mine.rows > 0 && mine.n_sets > 0) {
        RK_SLX(slx_decode_batch(ctx, mine.n_sets, d_in, 12 * tile_plane, nullptr, 0, (size_t)W, d_local, nullptr), slx_last_error(ctx));
        RK_SLX(slx_synchronize(ctx), slx_last_error(ctx));
    }
    if (rank == 0) RK_HIP(hipMemset(d_full, 0xff, full_elems * sizeof(double)));
    RK_HIP(hipDeviceSynchronize());
    RK_SLX(slx_gather_depth(comm, shards.data(), H, W, d_local, 0, d_full, 0, nullptr), slx_comm_last_error(comm));
    RK_SLX(slx_comm_synchronize(comm), slx_comm_last_error(comm));
    if (rank == 0) {
        RK_HIP(hipMemcpy(b.data(), d_full, b.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) != 0) { std::fprintf(stderr, "the pipelined and the plain gather differ\n"); return 1; }
        job.out.swap(a);
    }
    slx_comm_destroy(comm);
    slx_destroy(ctx);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 10) { std::fprintf(stderr, "usage: world rows|framesets in_place|staged W H sets chunk in out\n"); return 2; }
    Job job;
    job.world = std::atoi(argv[1]);
    job.split = argv[2];
    job.shape = std::string(argv[3]) == "staged" ? SLX_GATHER_STAGED : SLX_GATHER_IN_PLACE;
    job.W = std::atoi(argv[4]); job.H = std::atoi(argv[5]); job.sets = std::atoi(argv[6]); job.chunk = std::atoi(argv[7]);
    if (job.world < 1 || job.world > 16) return 2;
    std::vector<uint8_t> all((size_t)job.sets * 12 * job.W * job.H);
    FILE *f = std::fopen(argv[8], "rb");
    if (!f || std::fread(all.data(), 1, all.size(), f) != all.size()) { std::fprintf(stderr, "cannot read %s\n", argv[8]); return 1; }
    std::fclose(f);
    job.all = &all;
    if (slx_comm_unique_id(job.id, sizeof job.id) != SLX_OK) return 1;
    std::vector<int> rc((size_t)job.world, 0);
    std::vector<std::thread> threads;
    for (int r = 0; r < job.world; r++)
        threads.emplace_back([&, r] {
            rc[(size_t)r] = run_rank(r, job);
            if (rc[(size_t)r] != 0) {                                // wake whoever waits for this rank
                g_abort.store(true);
                std::lock_guard<std::mutex> g(g_worlds_m);
                for (auto &kv : g_worlds) { std::lock_guard<std::mutex> g2(kv.second->m); kv.second->cv.notify_all(); }
            }
        });
    for (auto &t : threads) t.join();
    for (int r = 0; r < job.world; r++)
        if (rc[(size_t)r] != 0) { std::fprintf(stderr, "rank %d failed\n", r); return 1; }
    FILE *o = std::fopen(argv[9], "wb");
    if (!o || std::fwrite(job.out.data(), sizeof(double), job.out.size(), o) != job.out.size()) return 1;
    std::fclose(o);
    std::printf("%d ranks as threads ok (%s, %s)\n", job.world, job.split.c_str(), argv[3]);
    return 0;
}
