// fake_rccl.cpp -- TEST DOUBLE, not a product path.  The nccl* entry points lsnShard* binds (exchange.hip: struct Rccl),
// implemented for ranks that are processes on ONE host sharing ONE GPU: every all-gather is staged through a POSIX
// shared-memory segment (device -> shm slot of the rank, barrier, every slot -> device).  RCCL refuses two ranks on one
// device, so this is the only way the world > 1 code of lsnShardStep (rank offsets, the grouped collectives, the chunked
// second-stream pipeline) can run on a one-GPU box.  Selected with $LSN_RCCL_LIBRARY=<path of this .so>; the collectives
// block the calling thread (a legal, if slow, implementation of the stream semantics: the work is complete when the call
// returns, so everything later on any stream sees it).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr size_t kSlot = 8u << 20;   // bytes one rank contributes per round
constexpr int kMaxRanks = 8;

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
};

struct Op {
    const void *send;
    void *recv;
    size_t bytes;
    hipStream_t stream;
};

}  // namespace

struct ncclComm {
    int rank = 0, world = 1;
    char name[64] = {0};
    Header *hdr = nullptr;
    unsigned char *data = nullptr;
    size_t map_bytes = 0;
};

namespace {

thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm *, Op>> g_queue;

// false: a peer did not arrive within two minutes (it died): the caller reports an error instead of spinning for ever
bool barrier(ncclComm *c)
{
    const int gen = c->hdr->generation.load(std::memory_order_acquire);
    if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
        c->hdr->arrived.store(0, std::memory_order_relaxed);
        c->hdr->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    const time_t t0 = time(nullptr);
    for (unsigned int spins = 0; c->hdr->generation.load(std::memory_order_acquire) == gen; spins++) {
        sched_yield();
        if ((spins & 0xFFFF) == 0xFFFF && time(nullptr) - t0 > 120) return false;
    }
    return true;
}

ncclResult_t run(ncclComm *c, const Op &op)
{
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;   // everything the send buffer depends on has run
    for (size_t off = 0; off < op.bytes || off == 0; off += kSlot) {
        const size_t n = op.bytes - off < kSlot ? op.bytes - off : kSlot;
        if (n && hipMemcpy(c->data + (size_t)c->rank * kSlot, (const char *)op.send + off, n, hipMemcpyDeviceToHost) != hipSuccess)
            return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        for (int q = 0; q < c->world && n; q++)
            if (hipMemcpy((char *)op.recv + (size_t)q * op.bytes + off, c->data + (size_t)q * kSlot, n, hipMemcpyHostToDevice) != hipSuccess)
                return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        if (op.bytes == 0) break;
    }
    return ncclSuccess;
}

size_t size_of(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

}  // namespace

extern "C" ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    unsigned int r = 0;
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { (void)!fread(&r, sizeof(r), 1, f); fclose(f); }
    snprintf(id->internal, sizeof(id->internal), "/lsn_fake_rccl_%d_%08x", (int)getpid(), r);
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (getenv("FAKE_RCCL_HANG_INIT"))   // a rendezvous that never completes (what a first run on an unknown node may look like): the caller's guard is under test
        for (;;) sleep(3600);
    ncclComm *c = new ncclComm();
    c->rank = rank;
    c->world = nranks;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->map_bytes = 4096 + (size_t)nranks * kSlot;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);    // a fresh segment reads as zeros: the header needs no initialiser
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->hdr = static_cast<Header *>(p);
    c->data = static_cast<unsigned char *>(p) + 4096;
    c->hdr->attached.fetch_add(1);
    if (!barrier(c)) {                                            // like the real call: returns once every rank has arrived
        ncclCommDestroy(c);
        return ncclSystemError;
    }
    *comm = c;
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    if (c->hdr->attached.fetch_sub(1) == 1) shm_unlink(c->name);  // the last rank out removes the segment
    munmap(c->hdr, c->map_bytes);
    delete c;
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommCount(const ncclComm_t c, int *count)
{
    if (!c || !count) return ncclInvalidArgument;
    *count = c->hdr->attached.load();     // the ranks attached to the segment right now, not the number this rank was told
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommUserRank(const ncclComm_t c, int *rank)
{
    if (!c || !rank) return ncclInvalidArgument;
    *rank = c->rank;
    return ncclSuccess;
}

extern "C" ncclResult_t ncclGroupStart()
{
    g_depth++;
    return ncclSuccess;
}

extern "C" ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (auto &e : g_queue)
        if (rc == ncclSuccess) rc = run(e.first, e.second);
    g_queue.clear();
    return rc;
}

extern "C" ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream)
{
    const size_t sz = size_of(type);
    if (!c || !sz || (count && (!send || !recv))) return ncclInvalidArgument;
    const Op op{send, recv, count * sz, stream};
    if (g_depth > 0) {
        g_queue.push_back({c, op});
        return ncclSuccess;
    }
    return run(c, op);
}

extern "C" const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
    case ncclSystemError: return "fake rccl: shared-memory segment unavailable, or a peer rank did not arrive";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
    }
}
