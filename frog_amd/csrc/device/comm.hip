// comm.hip -- libfrog_comm.so (include/frog_comm.h): the collectives of a one-process, one-thread-per-GPU
// registration over RCCL, and a loopback stand-in with the same interface for contexts that share a device.
//
// Only the public ABI of libfrog_hip.so is used (frog_comm_buffer for the device pointers, frog_get_stream for the
// stream the operation is enqueued on), so this is also the template for a host that drives the library with its own
// communicator.  RCCL usage is the single-process / multi-thread form: ncclCommInitAll once, then every rank's thread
// calls the same collective on its own communicator and stream; ragged all-gathers are n grouped broadcasts.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/frog_comm.h"

namespace frog {
void set_last_error(const std::string &s);      // libfrog_hip.so
}

namespace {

struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, waiting = 0;
    uint64_t generation = 0;
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = generation;
        if (++waiting == n) { waiting = 0; generation++; cv.notify_all(); return; }
        cv.wait(lk, [&] { return generation != g; });
    }
};

struct Shared {
    int n = 0;
    bool rccl = false;
    bool one_rank_per_process = false;                  // frog_comm_create_rank: no other rank's thread to meet at a barrier
    Barrier barrier;
    std::vector<size_t> row_begin, row_end;             // xyz2 rows of every rank
    // loopback staging
    std::vector<std::vector<unsigned char>> stage;      // one buffer per rank
    std::vector<float> xyz2_all;
    double box[64][6];
};

int comm_fail(int code, const std::string &msg)
{
    frog::set_last_error(msg);
    return code;
}

#define COMM_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return comm_fail(FROG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define COMM_NCCL(expr)                                                                                  \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return comm_fail(FROG_E_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

} // namespace

struct frog_comm {
    std::shared_ptr<Shared> sh;
    int rank = 0;
    int device = 0;
    ncclComm_t nccl = nullptr;
    frog_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    double *d_box = nullptr;        // [6] max xyz, -min xyz
    double *h_box = nullptr;        // pinned
};

extern "C" {

int frog_comm_create_rccl(int n, const int *devices, frog_comm **out)
{
    if (n < 1 || n > 64 || !devices || !out) return comm_fail(FROG_E_INVALID, "bad communicator arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return comm_fail(FROG_E_NODEVICE, "no HIP device");
    for (int r = 0; r < n; r++) {
        if (devices[r] < 0 || devices[r] >= ndev) return comm_fail(FROG_E_INVALID, "device index out of range");
        for (int q = 0; q < r; q++)
            if (devices[q] == devices[r]) return comm_fail(FROG_E_INVALID, "RCCL needs one distinct device per rank (use the loopback communicator to share a device)");
    }
    std::vector<ncclComm_t> comms(n);
    COMM_NCCL(ncclCommInitAll(comms.data(), n, devices));
    auto sh = std::make_shared<Shared>();
    sh->n = n; sh->rccl = true; sh->barrier.n = n;
    sh->row_begin.assign(n, 0); sh->row_end.assign(n, 0);
    for (int r = 0; r < n; r++) {
        frog_comm *c = new frog_comm;
        c->sh = sh; c->rank = r; c->device = devices[r]; c->nccl = comms[r];
        out[r] = c;
    }
    return FROG_OK;
}

int frog_comm_create_loopback(int n, frog_comm **out)
{
    if (n < 1 || n > 64 || !out) return comm_fail(FROG_E_INVALID, "bad communicator arguments");
    auto sh = std::make_shared<Shared>();
    sh->n = n; sh->rccl = false; sh->barrier.n = n;
    sh->row_begin.assign(n, 0); sh->row_end.assign(n, 0);
    sh->stage.resize(n);
    for (int r = 0; r < n; r++) {
        frog_comm *c = new frog_comm;
        c->sh = sh; c->rank = r;
        out[r] = c;
    }
    return FROG_OK;
}

int frog_comm_unique_id(unsigned char id_out[128])
{
    if (!id_out) return comm_fail(FROG_E_INVALID, "null id");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    COMM_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return FROG_OK;
}

int frog_comm_create_rank(int n, int rank, const unsigned char id_in[128], int device, frog_comm **out)
{
    if (n < 1 || n > 4096 || rank < 0 || rank >= n || !id_in || !out) return comm_fail(FROG_E_INVALID, "bad communicator arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return comm_fail(FROG_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return comm_fail(FROG_E_INVALID, "device index out of range");
    COMM_HIP(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof id);
    ncclComm_t nc = nullptr;
    COMM_NCCL(ncclCommInitRank(&nc, n, id, rank));
    auto sh = std::make_shared<Shared>();
    sh->n = n; sh->rccl = true; sh->one_rank_per_process = true; sh->barrier.n = 1;
    sh->row_begin.assign(n, 0); sh->row_end.assign(n, 0);
    frog_comm *c = new frog_comm;
    c->sh = sh; c->rank = rank; c->device = device; c->nccl = nc;
    *out = c;
    return FROG_OK;
}

int frog_comm_set_rows(frog_comm *c, const uint64_t *row_begin)
{
    if (!c || !row_begin) return comm_fail(FROG_E_INVALID, "null argument");
    for (int r = 0; r < c->sh->n; r++) {
        if (row_begin[r + 1] < row_begin[r]) return comm_fail(FROG_E_INVALID, "rows must be ascending");
        c->sh->row_begin[r] = (size_t)row_begin[r]; c->sh->row_end[r] = (size_t)row_begin[r + 1];
    }
    return FROG_OK;
}

void frog_comm_destroy_all(int n, frog_comm **comms)
{
    if (!comms) return;
    for (int r = 0; r < n; r++) {
        frog_comm *c = comms[r];
        if (!c) continue;
        if (c->d_box || c->h_box || c->nccl) (void)hipSetDevice(c->device);
        if (c->d_box) (void)hipFree(c->d_box);
        if (c->h_box) (void)hipHostFree(c->h_box);
        if (c->nccl) (void)ncclCommDestroy(c->nccl);
        delete c;
        comms[r] = nullptr;
    }
}

int frog_comm_bind(frog_comm *c, frog_ctx *ctx, const uint32_t *image_begin)
{
    if (!c || !ctx || !image_begin) return comm_fail(FROG_E_INVALID, "null argument");
    c->ctx = ctx;
    void *s = nullptr;
    int rc = frog_get_stream(ctx, &s, &c->device);
    if (rc) return rc;
    c->stream = (hipStream_t)s;
    COMM_HIP(hipSetDevice(c->device));
    size_t rb = 0, re = 0;
    rc = frog_comm_buffer(ctx, FROG_BUF_XYZ2, nullptr, nullptr, &rb, &re);
    if (rc) return rc;
    c->sh->row_begin[c->rank] = rb; c->sh->row_end[c->rank] = re;
    if (!c->d_box) {
        COMM_HIP(hipMalloc((void **)&c->d_box, 6 * sizeof(double)));
        COMM_HIP(hipMemset(c->d_box, 0, 6 * sizeof(double)));       // frog_comm_barrier all-reduces its first word
    }
    if (!c->h_box) COMM_HIP(hipHostMalloc((void **)&c->h_box, 6 * sizeof(double)));
    if (c->sh->one_rank_per_process) return FROG_OK;    // the other ranks' rows come through frog_comm_set_rows
    c->sh->barrier.wait();          // every rank's rows are known to all
    for (int r = 0; r + 1 < c->sh->n; r++)
        if (c->sh->row_end[r] != c->sh->row_begin[r + 1]) return comm_fail(FROG_E_INVALID, "shards must be contiguous and in rank order");
    (void)image_begin;
    return FROG_OK;
}

int frog_comm_barrier(frog_comm *c)
{
    if (!c) return comm_fail(FROG_E_INVALID, "null communicator");
    if (c->sh->one_rank_per_process) {
        // across processes: a one-element all-reduce, awaited
        if (!c->d_box || !c->ctx) return comm_fail(FROG_E_STATE, "communicator not bound");      // (the stream may be the null stream)
        COMM_HIP(hipSetDevice(c->device));
        COMM_NCCL(ncclAllReduce(c->d_box, c->d_box, 1, ncclDouble, ncclMax, c->nccl, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        return FROG_OK;
    }
    c->sh->barrier.wait();
    return FROG_OK;
}

int frog_comm_all_gather_xyz2(frog_comm *c)
{
    if (!c || !c->ctx) return comm_fail(FROG_E_INVALID, "communicator not bound");
    if (c->sh->n == 1) return FROG_OK;
    COMM_HIP(hipSetDevice(c->device));
    void *p = nullptr;
    size_t bytes = 0;
    int rc = frog_comm_buffer(c->ctx, FROG_BUF_XYZ2, &p, &bytes, nullptr, nullptr);
    if (rc) return rc;
    float *base = (float *)p;
    Shared &sh = *c->sh;
    if (sh.rccl) {
        // ragged shards: rank q broadcasts its rows, in place (send = receive = the rows' place in the replica);
        // the n broadcasts are one grouped operation
        COMM_NCCL(ncclGroupStart());
        for (int q = 0; q < sh.n; q++) {
            const size_t cnt = (sh.row_end[q] - sh.row_begin[q]) * 3;
            if (!cnt) continue;
            float *rows = base + sh.row_begin[q] * 3;
            COMM_NCCL(ncclBroadcast(rows, rows, cnt, ncclFloat, q, c->nccl, c->stream));
        }
        COMM_NCCL(ncclGroupEnd());
        return FROG_OK;
    }
    // loopback: own rows -> host, barrier, the other ranks' rows <- host
    const size_t P = bytes / (3 * sizeof(float));
    COMM_HIP(hipStreamSynchronize(c->stream));
    if (c->rank == 0 && sh.xyz2_all.size() != 3 * P) sh.xyz2_all.assign(3 * P, 0.f);
    sh.barrier.wait();
    const size_t b = sh.row_begin[c->rank], e = sh.row_end[c->rank];
    if (e > b) COMM_HIP(hipMemcpy(sh.xyz2_all.data() + 3 * b, base + 3 * b, (e - b) * 3 * sizeof(float), hipMemcpyDeviceToHost));
    sh.barrier.wait();
    for (int q = 0; q < sh.n; q++) {
        if (q == c->rank || sh.row_end[q] == sh.row_begin[q]) continue;
        COMM_HIP(hipMemcpy(base + 3 * sh.row_begin[q], sh.xyz2_all.data() + 3 * sh.row_begin[q],
                           (sh.row_end[q] - sh.row_begin[q]) * 3 * sizeof(float), hipMemcpyHostToDevice));
    }
    sh.barrier.wait();
    return FROG_OK;
}

int frog_comm_all_reduce(frog_comm *c, int which)
{
    if (!c || !c->ctx) return comm_fail(FROG_E_INVALID, "communicator not bound");
    if (c->sh->n == 1) return FROG_OK;
    if (which != FROG_BUF_EM && which != FROG_BUF_ENERGY && which != FROG_BUF_GRIDSUM) return comm_fail(FROG_E_INVALID, "buffer is not reducible");
    COMM_HIP(hipSetDevice(c->device));
    void *p = nullptr;
    size_t bytes = 0;
    int rc = frog_comm_buffer(c->ctx, which, &p, &bytes, nullptr, nullptr);
    if (rc) return rc;
    const bool f32 = which == FROG_BUF_EM;
    const size_t count = bytes / (f32 ? sizeof(float) : sizeof(double));
    Shared &sh = *c->sh;
    if (sh.rccl) {
        COMM_NCCL(ncclAllReduce(p, p, count, f32 ? ncclFloat : ncclDouble, ncclSum, c->nccl, c->stream));
        return FROG_OK;
    }
    COMM_HIP(hipStreamSynchronize(c->stream));
    sh.stage[c->rank].resize(bytes);
    COMM_HIP(hipMemcpy(sh.stage[c->rank].data(), p, bytes, hipMemcpyDeviceToHost));
    sh.barrier.wait();
    std::vector<unsigned char> out(bytes);
    if (f32) {                      // rank order, the same on every rank
        float *o = (float *)out.data();
        for (size_t k = 0; k < count; k++) { float s = 0; for (int q = 0; q < sh.n; q++) s += ((const float *)sh.stage[q].data())[k]; o[k] = s; }
    } else {
        double *o = (double *)out.data();
        for (size_t k = 0; k < count; k++) { double s = 0; for (int q = 0; q < sh.n; q++) s += ((const double *)sh.stage[q].data())[k]; o[k] = s; }
    }
    COMM_HIP(hipMemcpy(p, out.data(), bytes, hipMemcpyHostToDevice));
    sh.barrier.wait();              // nobody refills its stage before everybody has read it
    return FROG_OK;
}

int frog_comm_all_reduce_bounds(frog_comm *c, double mins[3], double maxs[3])
{
    if (!c || !c->ctx || !mins || !maxs) return comm_fail(FROG_E_INVALID, "communicator not bound");
    if (c->sh->n == 1) return FROG_OK;
    Shared &sh = *c->sh;
    if (sh.rccl) {
        COMM_HIP(hipSetDevice(c->device));
        for (int k = 0; k < 3; k++) { c->h_box[k] = maxs[k]; c->h_box[3 + k] = -mins[k]; }
        COMM_HIP(hipMemcpyAsync(c->d_box, c->h_box, 6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        COMM_NCCL(ncclAllReduce(c->d_box, c->d_box, 6, ncclDouble, ncclMax, c->nccl, c->stream));
        COMM_HIP(hipMemcpyAsync(c->h_box, c->d_box, 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        for (int k = 0; k < 3; k++) { maxs[k] = c->h_box[k]; mins[k] = -c->h_box[3 + k]; }
        return FROG_OK;
    }
    for (int k = 0; k < 3; k++) { sh.box[c->rank][k] = mins[k]; sh.box[c->rank][3 + k] = maxs[k]; }
    sh.barrier.wait();
    for (int q = 0; q < sh.n; q++)
        for (int k = 0; k < 3; k++) {
            if (sh.box[q][k] < mins[k]) mins[k] = sh.box[q][k];
            if (sh.box[q][3 + k] > maxs[k]) maxs[k] = sh.box[q][3 + k];
        }
    sh.barrier.wait();
    return FROG_OK;
}

} // extern "C"
