// fake_rccl.cc -- TEST DOUBLE for librccl.so.1 (tests/test_gpu_rccl_path.py).  NOT a product component and not RCCL.
//
// RCCL refuses two ranks on one device, so on a one-GPU box libcgx's CGX_COMM_RCCL transport can only ever run with one rank:
// the sequence prefold kernel -> ncclAllGather (in place, equal segments) -> K3, the scalar all-gather of the verification
// phase and the wire-up through ncclGetUniqueId / ncclCommInitRank had never met a second rank.  This library implements the
// entry points libcgx binds (csrc/cgx_rccl.cpp) with the semantics the NCCL API documents, for ranks that are separate OS
// processes sharing ONE GPU: a POSIX shared-memory segment named after the unique id holds a barrier and every rank's IPC
// handle of a staging buffer; ncclAllGather = copy the contribution into the own staging buffer on the
// caller's stream and wait for it, barrier, copy every peer's staging buffer into place on that stream and wait, barrier.  Synchronous where RCCL is stream-ordered -- a
// caller cannot tell the difference -- and every wait is bounded.  It says nothing about RCCL's performance or its xGMI path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

namespace {

constexpr int kMaxRanks = 16;
constexpr size_t kStage = 8u << 20;   // bytes of staging per rank

struct Shared {
    std::atomic<int> arrived;       // barrier: arrivals of the current generation
    std::atomic<int> generation;
    std::atomic<int> ready[kMaxRanks];
    hipIpcMemHandle_t handle[kMaxRanks];
};

}  // namespace

struct ncclComm {
    int nranks = 0, rank = 0;
    std::string name;
    Shared *sh = nullptr;
    unsigned char *stage[kMaxRanks] = {nullptr};
    bool failed = false;
};

namespace {

bool barrier(ncclComm *c)
{
    Shared *s = c->sh;
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == c->nranks) {
        s->arrived.store(0);
        s->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return false;   // a peer died
        std::this_thread::yield();
    }
    return true;
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) { if (version) *version = 22707; return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl: a peer did not arrive or a HIP call failed"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm *c = new ncclComm;
    c->nranks = nranks;
    c->rank = rank;
    c->name = std::string(id.internal, strnlen(id.internal, sizeof id.internal));
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return ncclSystemError;   // a fresh segment is zero-filled
    c->sh = static_cast<Shared *>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    close(fd);
    if (c->sh == MAP_FAILED) return ncclSystemError;
    if (hipMalloc(reinterpret_cast<void **>(&c->stage[rank]), kStage) != hipSuccess) return ncclUnhandledCudaError;
    if (hipIpcGetMemHandle(&c->sh->handle[rank], c->stage[rank]) != hipSuccess) return ncclUnhandledCudaError;
    c->sh->ready[rank].store(1);
    if (!barrier(c)) return ncclSystemError;
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        if (hipIpcOpenMemHandle(reinterpret_cast<void **>(&c->stage[q]), c->sh->handle[q], hipIpcMemLazyEnablePeerAccess) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    if (!barrier(c)) return ncclSystemError;
    fprintf(stderr, "fake_rccl: rank %d of %d wired (test double, tests/fake_rccl)\n", rank, nranks);
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    (void)barrier(c);                        // nobody unmaps a buffer a peer may still be reading
    for (int q = 0; q < c->nranks; ++q)
        if (q != c->rank && c->stage[q]) (void)hipIpcCloseMemHandle(c->stage[q]);
    (void)barrier(c);
    (void)hipFree(c->stage[c->rank]);
    if (c->rank == 0) shm_unlink(c->name.c_str());
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count) { *count = c->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *rank) { *rank = c->rank; return ncclSuccess; }
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

// Every rank's `sendcount` elements, rank order, into recvbuff on every rank; in place when sendbuff == recvbuff + rank * bytes
// (the NCCL contract libcgx's gather_segments relies on).
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t type, ncclComm_t c, hipStream_t stream)
{
    const size_t bytes = sendcount * type_size(type);
    if (!c || c->failed || bytes > kStage) return ncclInvalidArgument;
    // every copy is enqueued on the CALLER's stream (behind the producers of sendbuff, in front of the consumers of recvbuff:
    // the stream order RCCL gives) and the host waits for it before it tells the peers
    if (hipMemcpyAsync(c->stage[c->rank], sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) { c->failed = true; return ncclSystemError; }
    unsigned char *dst = static_cast<unsigned char *>(recvbuff);
    for (int q = 0; q < c->nranks; ++q) {
        if (q == c->rank && static_cast<const unsigned char *>(sendbuff) == dst + (size_t)q * bytes) continue;   // in place
        if (hipMemcpyAsync(dst + (size_t)q * bytes, c->stage[q], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) { c->failed = true; return ncclSystemError; }                      // staging buffers free again
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t type, int root, ncclComm_t c, hipStream_t stream)
{
    const size_t bytes = count * type_size(type);
    if (!c || c->failed || bytes > kStage) return ncclInvalidArgument;
    if (c->rank == root && hipMemcpyAsync(c->stage[root], sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if ((c->rank != root || sendbuff != recvbuff) && hipMemcpyAsync(recvbuff, c->stage[root], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

}  // extern "C"
