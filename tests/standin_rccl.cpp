// TEST INFRASTRUCTURE -- not part of the product.  A stand-in for librccl.so.1 that lets N ranks of a job share ONE GPU, so that the
// product's own multi-rank code -- rd_rccl_init, the receiver's half of rd_rccl_bcast_model, rd_rccl_allreduce_max, rd_rccl_barrier,
// rd_rccl_comm_count, the launcher, the transport agreement -- runs with N > 1 on a one-GPU box.  Real RCCL refuses two ranks on one
// device ("Duplicate GPU detected"); on the 8-GPU node the real library is used and this file plays no part.
//
// It implements the seven entry points the product resolves with dlsym (csrc/api.hip rccl_load) with the signatures of
// /opt/rocm/include/rccl/rccl.h, over a POSIX shared-memory segment named after the unique id: a collective synchronises the caller's
// stream, stages the data through the segment in chunks with hipMemcpy, and meets the other ranks at a counting barrier.  Semantics
// (element counts, data types, root, in-place buffers, max-reduction of doubles) are RCCL's; performance is irrelevant.
//
// Built by tests/test_gpu_standin_rccl.py into a private directory as librccl.so.1; the ranks find it through LD_LIBRARY_PATH.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace {

constexpr size_t kChunk = 32u << 20;       // staging area per segment
constexpr int kMaxRanks = 16;

struct Shared {
    std::atomic<int> arrived;              // counting barrier
    std::atomic<int> generation;
    std::atomic<int> joined;
    double slots[kMaxRanks][64];           // all-reduce operands (the product reduces a handful of doubles)
    alignas(64) unsigned char data[kChunk];
};

struct Comm {
    Shared* sh = nullptr;
    int rank = 0, nranks = 1;
    char name[80] = {0};
};

bool barrier(Comm* c, double timeout_s = 120.0)
{
    Shared* s = c->sh;
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == c->nranks) {
        s->arrived.store(0);
        s->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    }
    return true;
}

size_t type_size(ncclDataType_t t)
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

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/standin_rccl_%d_%ld", (int)getpid(), (long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) {
        delete c;
        return ncclSystemError;
    }
    if (ftruncate(fd, sizeof(Shared)) != 0) {      // (a new segment is zero-filled: the atomics start at 0 on every rank's view)
        close(fd);
        delete c;
        return ncclSystemError;
    }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->sh = (Shared*)p;
    c->sh->joined.fetch_add(1);
    if (!barrier(c)) {                               // like the real call: returns when every rank has joined
        munmap(p, sizeof(Shared));
        delete c;
        return ncclSystemError;
    }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    if (c->sh) {
        if (c->sh->joined.fetch_sub(1) == 1) shm_unlink(c->name);   // the last one out removes the name
        munmap(c->sh, sizeof(Shared));
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((const Comm*)comm)->nranks;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = (Comm*)comm;
    const size_t es = type_size(datatype);
    if (!c || !es || root < 0 || root >= c->nranks || (count && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const size_t bytes = count * es;
    for (size_t off = 0; off < bytes || off == 0; off += kChunk) {
        const size_t n = bytes - off < kChunk ? bytes - off : kChunk;
        if (c->rank == root && n && hipMemcpy(c->sh->data, (const char*)sendbuff + off, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        if (n && (c->rank != root || recvbuff != sendbuff) &&
            hipMemcpy((char*)recvbuff + off, c->sh->data, n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        if (bytes == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = (Comm*)comm;
    if (!c || datatype != ncclFloat64 || op != ncclMax || count < 1 || count > 64 || !sendbuff || !recvbuff) return ncclInvalidArgument;   // what the product uses
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->sh->slots[c->rank], sendbuff, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    double out[64];
    for (size_t i = 0; i < count; i++) {
        double m = c->sh->slots[0][i];
        for (int r = 1; r < c->nranks; r++) m = c->sh->slots[r][i] > m ? c->sh->slots[r][i] : m;
        out[i] = m;
    }
    if (!barrier(c)) return ncclSystemError;           // everyone has read the slots before anyone's next collective overwrites them
    if (hipMemcpy(recvbuff, out, count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclInvalidArgument: return "invalid argument (stand-in rccl)";
        case ncclSystemError: return "system error / a rank did not arrive (stand-in rccl)";
        case ncclUnhandledCudaError: return "HIP error (stand-in rccl)";
        default: return "error (stand-in rccl)";
    }
}

}  // extern "C"
