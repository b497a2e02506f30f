// A C++ host loop that spreads a batch over the ranks of a node and gathers the depth maps over RCCL, through the C ABI only
// (include/slx.h: slx_comm_*, slx_decode_gather, slx_gather_depth).  One process = one rank = one GPU; with one GPU on the
// box this runs as a world of one, which still exercises communicator creation from a unique id, the chunked decode + gather
// pipeline (in-place path) and the copy path of slx_gather_depth.  N > 1: start N copies with the same id file,
//   gather_host_loop <rank> <world> <id file> <rows|framesets> <W> <H> <sets> <in.bin> <out.bin>
// rank 0 writes the id file (ncclGetUniqueId) and, at the end, the gathered [sets][H][W] f64 maps.
//   in.bin: [sets][12][H][W] u8 -- every rank reads the whole batch and decodes its shard (row tile or frame-sets).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "slx.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_SLX(x, who) do { int rc_ = (x); if (rc_ != SLX_OK) { std::fprintf(stderr, "%s: %d %s\n", #x, rc_, who); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 10) { std::fprintf(stderr, "usage: rank world idfile rows|framesets W H sets in out\n"); return 2; }
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const std::string idfile = argv[3], split = argv[4];
    const int W = std::atoi(argv[5]), H = std::atoi(argv[6]), sets = std::atoi(argv[7]);
    int n_dev = 0;
    CHECK_HIP(hipGetDeviceCount(&n_dev));
    CHECK_HIP(hipSetDevice(rank % n_dev));

    // the shard table: the same on every rank
    std::vector<slx_shard> shards((size_t)world);
    for (int r = 0; r < world; r++) {
        auto cut = [&](int n, int &lo, int &cnt) { const int b = n / world, rem = n % world; lo = r * b + (r < rem ? r : rem); cnt = b + (r < rem ? 1 : 0); };
        if (split == "rows") { shards[r].set0 = 0; shards[r].n_sets = sets; cut(H, shards[r].row0, shards[r].rows); }
        else { cut(sets, shards[r].set0, shards[r].n_sets); shards[r].row0 = 0; shards[r].rows = H; }
    }
    const slx_shard mine = shards[(size_t)rank];

    slx_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.width = W; cfg.height = mine.rows; cfg.row_offset = mine.row0;      // a row tile keeps (v - cy) of the full frame
    cfg.mode = SLX_MODE_MULTIFREQ; cfg.n_freq = 3; cfg.n_steps = 4;
    cfg.period[0] = 1920; cfg.period[1] = 240; cfg.period[2] = 30;
    cfg.fov_min = -1e300; cfg.fov_max = 1e300; cfg.device = -1;
    const double cam[9] = {3600, 0, (W - 1) / 2.0, 0, 3600, (H - 1) / 2.0, 0, 0, 1}, pro[9] = {3000, 0, 900, 0, 3000, 600, 0, 0, 1};
    const double rot[9] = {0.99, -0.01, 0.13, 0.02, 0.99, -0.1, -0.13, 0.1, 0.98}, trans[3] = {-31.7, -9.3, 39.4};
    std::memcpy(cfg.cam, cam, sizeof cam); std::memcpy(cfg.pro, pro, sizeof pro); std::memcpy(cfg.rot, rot, sizeof rot); std::memcpy(cfg.trans, trans, sizeof trans);
    slx_ctx *ctx = nullptr;
    CHECK_SLX(slx_create(&cfg, &ctx), slx_last_error(nullptr));

    // rendezvous: 128 bytes through a file (any channel will do)
    char id[SLX_COMM_ID_BYTES];
    if (rank == 0) {
        CHECK_SLX(slx_comm_unique_id(id, sizeof id), slx_comm_last_error(nullptr));
        FILE *f = std::fopen((idfile + ".tmp").c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
        std::fclose(f);
        std::rename((idfile + ".tmp").c_str(), idfile.c_str());
    } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(idfile.c_str(), "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!f || std::fread(id, 1, sizeof id, f) != sizeof id) return 1;
        std::fclose(f);
    }
    slx_comm *comm = nullptr;
    CHECK_SLX(slx_comm_create(ctx, id, sizeof id, world, rank, &comm), slx_comm_last_error(nullptr));
    int w2 = 0, r2 = -1;
    slx_comm_info(comm, &w2, &r2);
    if (w2 != world || r2 != rank) return 1;

    // this rank's input: its row tile / its frame-sets of the batch, [n][12][rows][W]
    const size_t full_plane = (size_t)W * H, tile_plane = (size_t)W * mine.rows;
    std::vector<uint8_t> all((size_t)sets * 12 * full_plane), tile((size_t)mine.n_sets * 12 * tile_plane);
    FILE *f = std::fopen(argv[8], "rb");
    if (!f || std::fread(all.data(), 1, all.size(), f) != all.size()) return 1;
    std::fclose(f);
    for (int s = 0; s < mine.n_sets; s++)
        for (int p = 0; p < 12; p++)
            std::memcpy(tile.data() + ((size_t)s * 12 + p) * tile_plane, all.data() + ((size_t)(mine.set0 + s) * 12 + p) * full_plane + (size_t)mine.row0 * W, tile_plane);
    uint8_t *d_in = nullptr;
    double *d_full = nullptr, *d_scratch = nullptr, *d_local = nullptr;
    CHECK_HIP(hipMalloc((void **)&d_in, tile.size() ? tile.size() : 1));
    CHECK_HIP(hipMemcpy(d_in, tile.data(), tile.size(), hipMemcpyHostToDevice));
    const size_t full_elems = (size_t)sets * full_plane, local_elems = (size_t)mine.n_sets * tile_plane;
    if (rank == 0) CHECK_HIP(hipMalloc((void **)&d_full, full_elems * sizeof(double)));
    else CHECK_HIP(hipMalloc((void **)&d_scratch, (local_elems ? local_elems : 1) * sizeof(double)));
    if (rank == 0) CHECK_HIP(hipMemset(d_full, 0xff, full_elems * sizeof(double)));
    // the fill runs on the NULL stream, asynchronously; the decode and the gather run on non-blocking streams that do not order
    // themselves behind it: wait, or the poison could land on top of the result
    CHECK_HIP(hipDeviceSynchronize());

    // decode + gather, pipelined in chunks of 2 frame-sets (the last chunk may be ragged)
    CHECK_SLX(slx_decode_gather(comm, ctx, shards.data(), H, 2, d_in, 12 * tile_plane, nullptr, 0, (size_t)W, d_scratch, d_full, 0, nullptr),
              slx_comm_last_error(comm));
    CHECK_SLX(slx_comm_synchronize(comm), slx_comm_last_error(comm));
    std::vector<double> a(rank == 0 ? full_elems : 0), b(a.size());
    if (rank == 0) CHECK_HIP(hipMemcpy(a.data(), d_full, a.size() * sizeof(double), hipMemcpyDeviceToHost));

    // the same through a separate local buffer and one plain gather: must deliver the same array
    CHECK_HIP(hipMalloc((void **)&d_local, (local_elems ? local_elems : 1) * sizeof(double)));
    CHECK_SLX(slx_decode_batch(ctx, mine.n_sets, d_in, 12 * tile_plane, nullptr, 0, (size_t)W, d_local, nullptr), slx_last_error(ctx));
    CHECK_SLX(slx_synchronize(ctx), slx_last_error(ctx));
    if (rank == 0) CHECK_HIP(hipMemset(d_full, 0xff, full_elems * sizeof(double)));
    CHECK_HIP(hipDeviceSynchronize());                             // as above: the fill must have landed before the gather writes
    CHECK_SLX(slx_gather_depth(comm, shards.data(), H, W, d_local, 0, d_full, 0, nullptr), slx_comm_last_error(comm));
    CHECK_SLX(slx_comm_synchronize(comm), slx_comm_last_error(comm));
    if (rank == 0) {
        CHECK_HIP(hipMemcpy(b.data(), d_full, b.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) != 0) { std::fprintf(stderr, "the two gathers differ\n"); return 1; }
        FILE *o = std::fopen(argv[9], "wb");
        if (!o || std::fwrite(a.data(), sizeof(double), a.size(), o) != a.size()) return 1;
        std::fclose(o);
    }
    slx_comm_destroy(comm);
    slx_destroy(ctx);
    std::printf("rank %d of %d ok (%s)\n", rank, world, split.c_str());
    return 0;
}
