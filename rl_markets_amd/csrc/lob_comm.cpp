// Multi-GPU weight exchange (include/lob_comm.h): RCCL communicator per process and the
// all-reduce of the engine's delta-theta buffer over xGMI.  Host code only; the collective runs on
// the engine's own HIP stream, between the engine's delta kernels.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lob_comm.h"
#include "lob_internal.h"

static_assert(sizeof(ncclUniqueId) == LOB_COMM_ID_BYTES, "rendezvous token size");

// the phases of one exchange, stamped on the engine's stream (read back by lob_comm_exchange_stats)
struct ExchangeStamp { hipEvent_t ev[4]; int64_t bytes; int sparse; };

struct lob_comm {
    std::vector<ExchangeStamp> stamps;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;  // for the small host-side reductions
    double* scratch = nullptr;     // 64 doubles in HBM
    int mode = -1;                 // the exchange all ranks agreed on at their first lob_theta_allreduce: 1 sparse, 0 dense (-1: not yet)
};

#define NCCLCHK(expr)                                                                         \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess) {                                                              \
            lob_set_error(std::string(#expr) + ": " + ncclGetErrorString(_r));                \
            return LOB_EHIP;                                                                  \
        }                                                                                     \
    } while (0)
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            lob_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return LOB_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

extern "C" {

int lob_comm_get_id(uint8_t id[LOB_COMM_ID_BYTES]) {
    if (!id) return LOB_EINVAL;
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return LOB_OK;
}

int lob_comm_create(const uint8_t id[LOB_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, lob_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { lob_set_error("lob_comm_create: bad argument"); return LOB_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        lob_set_error("lob_comm_create: no such HIP device (one process per GPU; RCCL refuses two ranks on one device)");
        return LOB_ENODEV;
    }
    HIPCHK(hipSetDevice(device));
    lob_comm* c = new lob_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        lob_set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
        delete c;
        return LOB_EHIP;
    }
    if (hipStreamCreate(&c->stream) != hipSuccess || hipMalloc((void**)&c->scratch, 64 * sizeof(double)) != hipSuccess) {
        lob_set_error("lob_comm_create: stream / scratch allocation failed");
        lob_comm_destroy(c);
        return LOB_EHIP;
    }
    *out = c;
    return LOB_OK;
}

int lob_comm_create_file(const char* path, int32_t rank, int32_t world, int32_t device, int32_t timeout_s, lob_comm** out) {
    if (!path || !out || world < 1 || rank < 0 || rank >= world) { lob_set_error("lob_comm_create_file: bad argument"); return LOB_EINVAL; }
    uint8_t id[LOB_COMM_ID_BYTES];
    if (rank == 0) {
        int rc = lob_comm_get_id(id);
        if (rc) return rc;
        // the token is written to a fresh file of our own (O_EXCL | O_NOFOLLOW: never through a link somebody left at a
        // predictable name in a shared directory) and published by rename; callers put `path` in a private directory
        // (launch.py, lob_run: mkdtemp) or at least give it a per-run nonce
        const std::string tmp = std::string(path) + ".tmp";
        unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd < 0 || write(fd, id, sizeof id) != (ssize_t)sizeof id) {
            if (fd >= 0) { close(fd); unlink(tmp.c_str()); }
            lob_set_error(std::string("lob_comm_create_file: cannot write ") + tmp);
            return LOB_EINVAL;
        }
        close(fd);
        if (rename(tmp.c_str(), path) != 0) { lob_set_error(std::string("lob_comm_create_file: cannot publish ") + path); return LOB_EINVAL; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (true) {
            struct stat st;
            if (stat(path, &st) == 0 && st.st_size == (off_t)sizeof id) {
                const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
                const bool ok = fd >= 0 && read(fd, id, sizeof id) == (ssize_t)sizeof id;
                if (fd >= 0) close(fd);
                if (ok) break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > (double)timeout_s) {
                lob_set_error(std::string("lob_comm_create_file: rank 0 never published ") + path);
                return LOB_ESTATE;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    int rc = lob_comm_create(id, rank, world, device, out);
    if (rc == LOB_OK && rank == 0) unlink(path);  // ncclCommInitRank returned on rank 0: every rank has read the token
    return rc;
}

// Bind the calling thread to the cores next to GPU `device` (its PCIe function's local_cpulist in sysfs): launches, the
// exchange's one synchronisation and the event reads then never cross a socket.  Best effort: LOB_OK with *n_cpus = 0 when the
// topology cannot be read.
int lob_comm_pin_host_thread(int32_t device, int32_t* n_cpus) {
    if (n_cpus) *n_cpus = 0;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) return LOB_OK;
    for (char* p = bdf; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return LOB_OK;
    char line[1024] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return LOB_OK;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    for (char* tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k == 1) b = a;
        if (k < 1) continue;
        for (int i = a; i <= b && i < CPU_SETSIZE; i++) { CPU_SET(i, &set); n++; }
    }
    if (n > 0 && sched_setaffinity(0, sizeof set, &set) == 0 && n_cpus) *n_cpus = n;
    return LOB_OK;
}

void lob_comm_destroy(lob_comm* c) {
    if (!c) return;
    hipSetDevice(c->device);
    for (auto& y : c->stamps) for (auto& v : y.ev) hipEventDestroy(v);
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if (c->scratch) hipFree(c->scratch);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
}

int32_t lob_comm_rank(const lob_comm* c) { return c ? c->rank : -1; }
int32_t lob_comm_world(const lob_comm* c) { return c ? c->world : 0; }

int lob_comm_allreduce_f64(lob_comm* c, double* dev_buf, int64_t count, void* hip_stream) {
    if (!c || !dev_buf || count < 0) { lob_set_error("lob_comm_allreduce_f64: bad argument"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    NCCLCHK(ncclAllReduce(dev_buf, dev_buf, (size_t)count, ncclDouble, ncclSum, c->comm, (hipStream_t)hip_stream));
    return LOB_OK;
}

int lob_comm_reduce_host_f64(lob_comm* c, double* vals, int32_t n, int32_t op) {
    if (!c || !vals || n < 1 || n > 64 || (op != LOB_COMM_SUM && op != LOB_COMM_MAX)) { lob_set_error("lob_comm_reduce_host_f64: bad argument"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->scratch, vals, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(ncclAllReduce(c->scratch, c->scratch, (size_t)n, ncclDouble, op == LOB_COMM_SUM ? ncclSum : ncclMax, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(vals, c->scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LOB_OK;
}

int lob_comm_barrier(lob_comm* c) {
    double z = 0.0;
    return lob_comm_reduce_host_f64(c, &z, 1, LOB_COMM_SUM);
}

int lob_comm_allgather_u32(lob_comm* c, const uint32_t* dev_send, uint32_t* dev_recv, int64_t count, void* hip_stream) {
    if (!c || !dev_send || !dev_recv || count < 0) { lob_set_error("lob_comm_allgather_u32: bad argument"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    NCCLCHK(ncclAllGather(dev_send, dev_recv, (size_t)count, ncclUint32, c->comm, (hipStream_t)hip_stream));
    return LOB_OK;
}

static hipEvent_t new_event() { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableSystemFence); return e; }

static int theta_allreduce_impl(lob_engine* e, lob_comm* c, ExchangeStamp& x) {
    hipStream_t st = (hipStream_t)lob_stream(e);
    int rc;
    if (c->mode < 0) {
        // Which exchange: decided ONCE, by all ranks together -- sparse only if every rank can and wants to (a rank whose engine
        // has no exact map, or whose environment says LOB_DENSE_EXCHANGE=1, would otherwise enter a different collective than
        // its peers and all of them would wait for ever).  One tiny host-side all-reduce, at the first exchange.
        // The same reduction carries every rank's allocation result: the sparse exchange's buffers (the gathered maps, the union,
        // the block offsets) are made HERE, before any collective of the data path, and a rank that could not makes every rank
        // return the error together -- from then on nothing is allocated between two collectives.
        const char* df = getenv("LOB_DENSE_EXCHANGE");
        double dense_wanted = ((df && df[0] == '1') || !lob_delta_sparse_supported(e)) ? 1.0 : 0.0;
        int rc_alloc = LOB_OK;
        if (dense_wanted == 0.0) {
            uint32_t *own = nullptr, *gather = nullptr;
            int64_t words = 0;
            rc_alloc = lob_delta_sparse_maps(e, c->world, &own, &gather, &words);
        }
        double agreed = rc_alloc != LOB_OK ? 2.0 : dense_wanted;
        rc = lob_comm_reduce_host_f64(c, &agreed, 1, LOB_COMM_MAX);
        if (rc) return rc;
        if (agreed > 1.0) {
            if (rc_alloc == LOB_OK) lob_set_error("lob_theta_allreduce: another rank could not allocate the exchange's buffers");
            return rc_alloc != LOB_OK ? rc_alloc : LOB_ENOMEM;
        }
        c->mode = agreed > 0.0 ? 0 : 1;
    }
    hipEventRecord(x.ev[0], st);
    if (c->mode == 1) {
        // maps all-gathered -> union -> packed deltas all-reduced -> scattered back (include/lob_engine.h)
        uint32_t *own = nullptr, *gather = nullptr;
        int64_t words = 0;
        // (no allocation from here on -- see above --: what can still fail between the two collectives is the HIP / RCCL runtime
        // itself, and that ends the communicator for every rank: ncclCommAbort territory, not a recoverable local error)
        rc = lob_delta_sparse_maps(e, c->world, &own, &gather, &words);
        if (rc) return rc;
        rc = lob_comm_allgather_u32(c, own, gather, words, st);
        if (rc) return rc;
        double* buf = nullptr;
        int64_t count = 0;
        rc = lob_delta_sparse_pack(e, c->world, &buf, &count);   // (the union is the same on every rank: so is `count`)
        if (rc) return rc;
        hipEventRecord(x.ev[1], st);
        if (count > 0) rc = lob_comm_allreduce_f64(c, buf, count, st);
        if (rc) return rc;
        hipEventRecord(x.ev[2], st);
        rc = lob_delta_sparse_apply(e);
        x.bytes = count * 8 + words * 4 * c->world;
        x.sparse = 1;
    } else {
        double* delta = nullptr;
        int64_t count = 0;
        rc = lob_delta_begin_async(e, &delta, &count);  // delta = theta - theta_sync, on the engine stream
        if (rc) return rc;
        hipEventRecord(x.ev[1], st);
        rc = lob_comm_allreduce_f64(c, delta, count, st);
        if (rc) return rc;
        hipEventRecord(x.ev[2], st);
        rc = lob_delta_apply(e);                          // theta = theta_sync + sum(delta); theta_sync = theta
        x.bytes = count * 8;
        x.sparse = 0;
    }
    hipEventRecord(x.ev[3], st);
    return rc;
}

int lob_theta_allreduce(lob_engine* e, lob_comm* c) {
    if (!e || !c) { lob_set_error("lob_theta_allreduce: bad argument"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    ExchangeStamp x;
    for (auto& v : x.ev) v = new_event();
    const int rc = theta_allreduce_impl(e, c, x);
    if (rc != LOB_OK || c->stamps.size() >= 4096) {  // a failed exchange, or nobody reads the statistics: no stamps kept
        for (auto& v : x.ev) hipEventDestroy(v);
        if (rc == LOB_OK && c->stamps.size() >= 4096) {
            for (auto& y : c->stamps) for (auto& v : y.ev) hipEventDestroy(v);
            c->stamps.clear();
        }
        return rc;
    }
    c->stamps.push_back(x);
    return rc;
}

int lob_comm_exchange_stats(lob_comm* c, double out[7]) {
    if (!c || !out) return LOB_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    for (int i = 0; i < 7; i++) out[i] = 0.0;
    for (auto& x : c->stamps) {
        hipEventSynchronize(x.ev[3]);
        float a = 0, b = 0, d = 0;
        hipEventElapsedTime(&a, x.ev[0], x.ev[1]);
        hipEventElapsedTime(&b, x.ev[1], x.ev[2]);
        hipEventElapsedTime(&d, x.ev[2], x.ev[3]);
        out[0] += 1.0; out[1] += a; out[2] += b; out[3] += d; out[4] += (double)x.bytes; out[5] += x.sparse;
        for (auto& v : x.ev) hipEventDestroy(v);
    }
    c->stamps.clear();
    int n = 0;
    if (ncclCommCount(c->comm, &n) == ncclSuccess) out[6] = n;  // ranks as RCCL sees them
    return LOB_OK;
}

}  // extern "C"
