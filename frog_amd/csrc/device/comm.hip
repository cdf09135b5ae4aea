// comm.hip -- libfrog_comm.so (include/frog_comm.h): the collectives of a one-process, one-thread-per-GPU
// registration over RCCL, and a loopback stand-in with the same interface for contexts that share a device.
//
// Only the public ABI of libfrog_hip.so is used (frog_comm_buffer for the device pointers, frog_get_stream for the
// stream the operation is enqueued on), so this is also the template for a host that drives the library with its own
// communicator.  RCCL usage is the single-process / multi-thread form: ncclCommInitAll once, then every rank's thread
// calls the same collective on its own communicator and stream; ragged all-gathers are one equal-size all-gather of padded slots.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
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
    std::vector<unsigned char> slab_host;               // loopback: the slab's host image
    double box[64][6];
    // host-staged collectives ACROSS PROCESSES (frog_comm_create_shm): a control block and a data area in POSIX shared memory
    struct ShmCtl {
        std::atomic<uint32_t> arrived, generation;
        double box[64][6];
    };
    bool shm = false;
    std::string shm_name;
    ShmCtl *ctl = nullptr;
    unsigned char *area = nullptr;      // current data area (n slots of area_slot bytes), re-created larger on demand
    size_t area_bytes = 0;
    bool area_pinned = false;           // hipHostRegister'ed in this process
    unsigned area_seq = 0;
    double barrier_timeout_s = 120.0;
};

// FROG_COMM_EXERCISE_SINGLE_RANK=1 (tests): a communicator of ONE rank over RCCL issues its collectives for real instead of
// returning at once -- what a one-GPU box can execute of the calls a multi-GPU run makes (ncclAllGather in place on the
// slab, ncclAllReduce on the library's buffers), on the context's stream, in the hosts' own order
bool skip_single_rank(const Shared &sh)
{
    static const bool exercise = getenv("FROG_COMM_EXERCISE_SINGLE_RANK") != nullptr;
    return sh.n == 1 && !(exercise && sh.rccl);
}

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
    // the coordinate gather's slab (include/frog_hip.h: world slots of FROG_SLAB_SLOT_BYTES(slot_rows) bytes), on this rank's device
    unsigned char *slab = nullptr;
    uint64_t slot_rows = 0;
    std::vector<uint64_t> rows;     // row_begin of every rank + the end: what frog_comm_unpack_slab_step wants
    // device time of the collectives (frog_comm_timing): every TIMING_SAMPLE-th call of a kind is bracketed by an event pair
    bool timing = false;
    struct Timed { hipEvent_t a, b; int kind; };
    std::vector<Timed> timed;
    uint64_t calls[4] = {}, sampled[4] = {};
    double ms[4] = {};
};

namespace {
constexpr int TIMING_SAMPLE = 8;
// kinds: 0 all_gather_xyz2, 1 all_reduce EM, 2 all_reduce ENERGY, 3 all_reduce GRIDSUM
struct TimeSpan {
    frog_comm *c; hipEvent_t a = nullptr, b = nullptr; int kind;
    TimeSpan(frog_comm *comm, int k) : c(comm), kind(k)
    {
        if (!c->timing || !c->stream) { c = nullptr; return; }
        if (c->calls[k]++ % TIMING_SAMPLE) { c = nullptr; return; }
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { c = nullptr; return; }
        (void)hipEventRecord(a, c->stream);
    }
    ~TimeSpan() { if (c) { (void)hipEventRecord(b, c->stream); c->timed.push_back({ a, b, kind }); } }
};

// barrier between the ranks' PROCESSES: sense reversal on two counters in shared memory; bounded (a rank that died must not
// leave the others spinning for ever)
int shm_barrier(Shared &sh)
{
    Shared::ShmCtl *ctl = sh.ctl;
    const uint32_t g = ctl->generation.load(std::memory_order_acquire);
    if (ctl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)sh.n) {
        ctl->arrived.store(0, std::memory_order_relaxed);
        ctl->generation.fetch_add(1, std::memory_order_release);
        return FROG_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (ctl->generation.load(std::memory_order_acquire) == g) {
        if ((++spins & 63u) == 0u) sched_yield();
        if ((spins & 0xFFFFu) == 0u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > sh.barrier_timeout_s)
            return comm_fail(FROG_E_STATE, "shared-memory barrier timed out: a rank is missing");
    }
    return FROG_OK;
}

void *shm_map(const std::string &name, size_t bytes, bool create, std::string &err)
{
    int fd = shm_open(name.c_str(), create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) { err = "shm_open " + name + ": " + strerror(errno); return nullptr; }
    if (create && ftruncate(fd, (off_t)bytes) != 0) { err = "ftruncate " + name + ": " + strerror(errno); close(fd); return nullptr; }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { err = "mmap " + name + ": " + strerror(errno); return nullptr; }
    return p;
}

// a data area of at least `bytes`, the same on every rank (all ranks make the same calls in the same order): rank 0 creates
// the larger segment, the others attach to it
int shm_area(Shared &sh, int rank, size_t bytes)
{
    if (sh.area_bytes >= bytes) return FROG_OK;
    int rc = shm_barrier(sh);           // nobody still reads the old area
    if (rc) return rc;
    const size_t want = std::max(bytes, (size_t)1 << 20) * 2;
    const std::string old_name = sh.shm_name + "_d" + std::to_string(sh.area_seq);
    if (sh.area) {
        if (sh.area_pinned) (void)hipHostUnregister(sh.area);
        munmap(sh.area, sh.area_bytes); sh.area = nullptr; sh.area_pinned = false;
        if (rank == 0) shm_unlink(old_name.c_str());
    }
    sh.area_seq++;
    const std::string name = sh.shm_name + "_d" + std::to_string(sh.area_seq);
    std::string err;
    if (rank == 0) {
        sh.area = (unsigned char *)shm_map(name, want, true, err);
        if (!sh.area) return comm_fail(FROG_E_HIP, err);
    }
    rc = shm_barrier(sh);
    if (rc) return rc;
    if (rank != 0) {
        sh.area = (unsigned char *)shm_map(name, want, false, err);
        if (!sh.area) return comm_fail(FROG_E_HIP, err);
    }
    sh.area_bytes = want;
    // page-locked for this process's device copies (a pageable staging area goes through the runtime's own bounce buffers:
    // the 24 MB coordinate gather of cfg 3 took 2 ms per iteration that way); best effort
    sh.area_pinned = hipHostRegister(sh.area, want, hipHostRegisterPortable) == hipSuccess;
    if (!sh.area_pinned) (void)hipGetLastError();
    return shm_barrier(sh);
}
} // namespace

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

int frog_comm_create_shm(int n, int rank, const char *name, int device, frog_comm **out)
{
    if (n < 1 || n > 64 || rank < 0 || rank >= n || !name || !name[0] || !out) return comm_fail(FROG_E_INVALID, "bad communicator arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return comm_fail(FROG_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return comm_fail(FROG_E_INVALID, "device index out of range");
    auto sh = std::make_shared<Shared>();
    sh->n = n; sh->rccl = false; sh->shm = true; sh->one_rank_per_process = true; sh->barrier.n = 1;
    sh->row_begin.assign(n, 0); sh->row_end.assign(n, 0);
    sh->shm_name = std::string(name[0] == '/' ? "" : "/") + name;
    std::string err;
    // rank 0 creates the control block (zero-filled by ftruncate); the others wait for it to appear
    const std::string ctl_name = sh->shm_name + "_ctl";
    if (rank == 0) {
        shm_unlink(ctl_name.c_str());
        sh->ctl = (Shared::ShmCtl *)shm_map(ctl_name, sizeof(Shared::ShmCtl), true, err);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (!sh->ctl) {
            struct stat st;
            int fd = shm_open(ctl_name.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                const bool ready = fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Shared::ShmCtl);
                close(fd);
                if (ready) { sh->ctl = (Shared::ShmCtl *)shm_map(ctl_name, sizeof(Shared::ShmCtl), false, err); break; }
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > sh->barrier_timeout_s) { err = "rank 0 never created " + ctl_name; break; }
            usleep(2000);
        }
    }
    if (!sh->ctl) return comm_fail(FROG_E_HIP, err);
    frog_comm *c = new frog_comm;
    c->sh = sh; c->rank = rank; c->device = device;
    *out = c;
    const int rc = shm_barrier(*sh);            // everybody is attached
    if (rc) { delete c; *out = nullptr; return rc; }
    if (rank == 0) shm_unlink(ctl_name.c_str());        // the mappings keep it alive; nothing is left behind in /dev/shm
    return FROG_OK;
}

int frog_comm_timing(frog_comm *c, int on)
{
    if (!c) return comm_fail(FROG_E_INVALID, "null communicator");
    c->timing = on != 0;
    return FROG_OK;
}

int frog_comm_timing_read(frog_comm *c, double ms_out[4], uint64_t calls_out[4], uint64_t sampled_out[4])
{
    if (!c) return comm_fail(FROG_E_INVALID, "null communicator");
    if (c->stream || c->ctx) { COMM_HIP(hipSetDevice(c->device)); COMM_HIP(hipStreamSynchronize(c->stream)); }
    for (auto &t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { c->ms[t.kind] += (double)ms; c->sampled[t.kind]++; }
        (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b);
    }
    c->timed.clear();
    for (int k = 0; k < 4; k++) {
        if (ms_out) ms_out[k] = c->ms[k];
        if (calls_out) calls_out[k] = c->calls[k];
        if (sampled_out) sampled_out[k] = c->sampled[k];
    }
    return FROG_OK;
}

int frog_comm_set_rows(frog_comm *c, const uint64_t *row_begin)
{
    if (!c || !row_begin) return comm_fail(FROG_E_INVALID, "null argument");
    for (int r = 0; r < c->sh->n; r++) {
        if (row_begin[r + 1] < row_begin[r]) return comm_fail(FROG_E_INVALID, "rows must be ascending");
        c->sh->row_begin[r] = (size_t)row_begin[r]; c->sh->row_end[r] = (size_t)row_begin[r + 1];
    }
    c->slot_rows = 0;               // the slab follows the rows: re-made on demand
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
        if (c->slab) { (void)hipSetDevice(c->device); (void)hipFree(c->slab); }
        if (c->h_box) (void)hipHostFree(c->h_box);
        if (c->nccl) (void)ncclCommDestroy(c->nccl);
        for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
        if (c->sh && c->sh->shm) {
            Shared &sh = *c->sh;
            if (sh.area) {
                if (sh.area_pinned) (void)hipHostUnregister(sh.area);
                munmap(sh.area, sh.area_bytes);
                if (c->rank == 0) shm_unlink((sh.shm_name + "_d" + std::to_string(sh.area_seq)).c_str());
                sh.area = nullptr;
            }
            if (sh.ctl) { munmap(sh.ctl, sizeof(Shared::ShmCtl)); sh.ctl = nullptr; }
        }
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
    if (c->sh->shm) {
        if (c->ctx) { COMM_HIP(hipSetDevice(c->device)); COMM_HIP(hipStreamSynchronize(c->stream)); }
        return shm_barrier(*c->sh);
    }
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

static int ensure_slab(frog_comm *c);

int frog_comm_all_gather_xyz2(frog_comm *c)
{
    if (!c || !c->ctx) return comm_fail(FROG_E_INVALID, "communicator not bound");
    if (skip_single_rank(*c->sh)) return FROG_OK;
    COMM_HIP(hipSetDevice(c->device));
    void *p = nullptr;
    size_t bytes = 0;
    int rc = frog_comm_buffer(c->ctx, FROG_BUF_XYZ2, &p, &bytes, nullptr, nullptr);
    if (rc) return rc;
    float *base = (float *)p;
    Shared &sh = *c->sh;
    TimeSpan span(c, 0);
    if (sh.shm) {
        // across processes, host-staged: own rows -> the shared table, barrier, the other ranks' rows <- the shared table
        rc = shm_area(sh, c->rank, bytes);
        if (rc) return rc;
        const size_t b = sh.row_begin[c->rank], e = sh.row_end[c->rank];
        if (e > b) COMM_HIP(hipMemcpyAsync(sh.area + 3 * b * sizeof(float), base + 3 * b, (e - b) * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        rc = shm_barrier(sh);
        if (rc) return rc;
        // the rows before and after the own ones are contiguous in the table: two copies
        if (b > 0) COMM_HIP(hipMemcpyAsync(base, sh.area, 3 * b * sizeof(float), hipMemcpyHostToDevice, c->stream));
        const size_t P = bytes / (3 * sizeof(float));
        if (e < P) COMM_HIP(hipMemcpyAsync(base + 3 * e, sh.area + 3 * e * sizeof(float), 3 * (P - e) * sizeof(float), hipMemcpyHostToDevice, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        return shm_barrier(sh);         // nobody overwrites its rows before everybody has read them
    }
    if (sh.rccl) {
        // ragged shards, ONE collective: the own rows into the rank's slot of the slab, an equal-size all-gather of the slots,
        // every slot's rows into the replica in one launch (frog_comm_unpack_slab_step; the own rows come back unchanged).
        // Until round 5: n grouped in-place broadcasts -- n collective launches per iteration.
        rc = ensure_slab(c);
        if (rc) return rc;
        const size_t slot = FROG_SLAB_SLOT_BYTES(c->slot_rows);
        const size_t b = sh.row_begin[c->rank], e = sh.row_end[c->rank];
        if (e > b) COMM_HIP(hipMemcpyAsync(c->slab + slot * (size_t)c->rank, base + 3 * b, (e - b) * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        COMM_NCCL(ncclAllGather(c->slab + slot * (size_t)c->rank, c->slab, slot, ncclChar, c->nccl, c->stream));
        return frog_comm_unpack_slab_step(c->ctx, c->slab, c->slot_rows, (uint32_t)sh.n, c->rows.data(), (uint32_t)c->rank, 0u);
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

// the slab of the padded gather: slot_rows = the longest shard, every rank's rows known (frog_comm_bind / frog_comm_set_rows)
static int ensure_slab(frog_comm *c)
{
    Shared &sh = *c->sh;
    uint64_t longest = 0;
    for (int r = 0; r < sh.n; r++) longest = std::max<uint64_t>(longest, sh.row_end[r] - sh.row_begin[r]);
    if (c->slab && c->slot_rows == longest && (int)c->rows.size() == sh.n + 1) return FROG_OK;
    COMM_HIP(hipSetDevice(c->device));
    if (c->slab) { COMM_HIP(hipStreamSynchronize(c->stream)); COMM_HIP(hipFree(c->slab)); c->slab = nullptr; }
    const size_t bytes = FROG_SLAB_SLOT_BYTES(longest) * (size_t)sh.n;
    COMM_HIP(hipMalloc((void **)&c->slab, bytes));
    COMM_HIP(hipMemsetAsync(c->slab, 0, bytes, c->stream));
    c->slot_rows = longest;
    c->rows.resize(sh.n + 1);
    for (int r = 0; r < sh.n; r++) c->rows[r] = sh.row_begin[r];
    c->rows[sh.n] = sh.row_end[sh.n - 1];
    return FROG_OK;
}

int frog_comm_slab(frog_comm *c, void **slab, uint64_t *slot_rows)
{
    if (!c || !c->ctx) return comm_fail(FROG_E_INVALID, "communicator not bound");
    const int rc = ensure_slab(c);
    if (rc) return rc;
    if (slab) *slab = c->slab;
    if (slot_rows) *slot_rows = c->slot_rows;
    return FROG_OK;
}

int frog_comm_all_gather_slab(frog_comm *c)
{
    if (!c || !c->ctx) return comm_fail(FROG_E_INVALID, "communicator not bound");
    int rc = ensure_slab(c);
    if (rc) return rc;
    Shared &sh = *c->sh;
    if (skip_single_rank(sh)) return FROG_OK;
    COMM_HIP(hipSetDevice(c->device));
    const size_t slot = FROG_SLAB_SLOT_BYTES(c->slot_rows), all = slot * (size_t)sh.n;
    unsigned char *mine = c->slab + slot * (size_t)c->rank;
    TimeSpan span(c, 0);
    if (sh.rccl) {
        // ONE equal-size all-gather, in place (send = the rank's slot inside the receive buffer): ragged shards cost padding,
        // not a broadcast per rank -- each collective launch costs tens of microseconds over xGMI, a rank's whole share of an
        // iteration at eight GPUs
        COMM_NCCL(ncclAllGather(mine, c->slab, slot, ncclChar, c->nccl, c->stream));
        return FROG_OK;
    }
    if (sh.shm) {
        rc = shm_area(sh, c->rank, all);
        if (rc) return rc;
        COMM_HIP(hipMemcpyAsync(sh.area + slot * (size_t)c->rank, mine, slot, hipMemcpyDeviceToHost, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        rc = shm_barrier(sh);
        if (rc) return rc;
        // the slots before and after the own one: two copies
        if (c->rank > 0) COMM_HIP(hipMemcpyAsync(c->slab, sh.area, slot * (size_t)c->rank, hipMemcpyHostToDevice, c->stream));
        if (c->rank + 1 < sh.n)
            COMM_HIP(hipMemcpyAsync(mine + slot, sh.area + slot * (size_t)(c->rank + 1), slot * (size_t)(sh.n - 1 - c->rank), hipMemcpyHostToDevice, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        return shm_barrier(sh);         // nobody refills its slot before everybody has read it
    }
    // loopback (threads of one process): the same through a host image of the slab
    COMM_HIP(hipStreamSynchronize(c->stream));
    if (c->rank == 0 && sh.slab_host.size() != all) sh.slab_host.assign(all, 0);
    sh.barrier.wait();
    COMM_HIP(hipMemcpy(sh.slab_host.data() + slot * (size_t)c->rank, mine, slot, hipMemcpyDeviceToHost));
    sh.barrier.wait();
    if (c->rank > 0) COMM_HIP(hipMemcpy(c->slab, sh.slab_host.data(), slot * (size_t)c->rank, hipMemcpyHostToDevice));
    if (c->rank + 1 < sh.n)
        COMM_HIP(hipMemcpy(mine + slot, sh.slab_host.data() + slot * (size_t)(c->rank + 1), slot * (size_t)(sh.n - 1 - c->rank), hipMemcpyHostToDevice));
    sh.barrier.wait();
    return FROG_OK;
}

int frog_comm_gather_points(frog_comm *c, int apply, int after_step, uint32_t sum_mask)
{
    if (!c || !c->ctx) return comm_fail(FROG_E_INVALID, "communicator not bound");
    int rc = ensure_slab(c);
    if (rc) return rc;
    rc = frog_transform_points_slab(c->ctx, apply, after_step, c->slab, c->slot_rows, (uint32_t)c->rank);
    if (rc) return rc;
    rc = frog_comm_all_gather_slab(c);
    if (rc) return rc;
    return frog_comm_unpack_slab_step(c->ctx, c->slab, c->slot_rows, (uint32_t)c->sh->n, c->rows.data(), (uint32_t)c->rank, sum_mask);
}

int frog_comm_all_reduce(frog_comm *c, int which)
{
    if (!c || !c->ctx) return comm_fail(FROG_E_INVALID, "communicator not bound");
    if (skip_single_rank(*c->sh)) return FROG_OK;
    if (which != FROG_BUF_EM && which != FROG_BUF_ENERGY && which != FROG_BUF_GRIDSUM) return comm_fail(FROG_E_INVALID, "buffer is not reducible");
    COMM_HIP(hipSetDevice(c->device));
    void *p = nullptr;
    size_t bytes = 0;
    int rc = frog_comm_buffer(c->ctx, which, &p, &bytes, nullptr, nullptr);
    if (rc) return rc;
    const bool f32 = which == FROG_BUF_EM;
    const size_t count = bytes / (f32 ? sizeof(float) : sizeof(double));
    Shared &sh = *c->sh;
    TimeSpan span(c, which == FROG_BUF_EM ? 1 : which == FROG_BUF_ENERGY ? 2 : 3);
    if (sh.shm) {
        // across processes, host-staged: every rank's buffer into its slot, barrier, every rank adds the slots in rank order
        rc = shm_area(sh, c->rank, bytes * (size_t)sh.n);
        if (rc) return rc;
        COMM_HIP(hipMemcpyAsync(sh.area + bytes * (size_t)c->rank, p, bytes, hipMemcpyDeviceToHost, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        rc = shm_barrier(sh);
        if (rc) return rc;
        std::vector<unsigned char> out(bytes);
        if (f32) {
            float *o = (float *)out.data();
            for (size_t k = 0; k < count; k++) { float s = 0; for (int q = 0; q < sh.n; q++) s += ((const float *)(sh.area + bytes * (size_t)q))[k]; o[k] = s; }
        } else {
            double *o = (double *)out.data();
            for (size_t k = 0; k < count; k++) { double s = 0; for (int q = 0; q < sh.n; q++) s += ((const double *)(sh.area + bytes * (size_t)q))[k]; o[k] = s; }
        }
        COMM_HIP(hipMemcpyAsync(p, out.data(), bytes, hipMemcpyHostToDevice, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        return shm_barrier(sh);         // nobody refills its slot before everybody has read it
    }
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
    if (skip_single_rank(*c->sh)) return FROG_OK;
    Shared &sh = *c->sh;
    if (sh.shm) {
        for (int k = 0; k < 3; k++) { sh.ctl->box[c->rank][k] = mins[k]; sh.ctl->box[c->rank][3 + k] = maxs[k]; }
        int rc = shm_barrier(sh);
        if (rc) return rc;
        for (int q = 0; q < sh.n; q++)
            for (int k = 0; k < 3; k++) {
                if (sh.ctl->box[q][k] < mins[k]) mins[k] = sh.ctl->box[q][k];
                if (sh.ctl->box[q][3 + k] > maxs[k]) maxs[k] = sh.ctl->box[q][3 + k];
            }
        return shm_barrier(sh);
    }
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
