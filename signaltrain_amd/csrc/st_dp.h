// st_dp.h -- data-parallel exchange step inside the library: RCCL over xGMI, driven from C (no Python between buckets).
//
// The reference's only multi-GPU mechanism is a disabled nn.DataParallel stub (train.py:259-263); SURVEY.md 8(b)/(e)
// specify the replacement: one process per GPU, the flat fp32 gradient sum-all-reduced bucket by bucket as the backward
// makes the buckets final, 1/world scaling, L1 clip AFTER the reduction (same norm on every rank, no second collective),
// replicated Adam.  The communicator, its side stream and the ordering events live here; the caller supplies the 128-byte
// RCCL unique id (created on rank 0 by st_dp_unique_id and handed to the other ranks over whatever bootstrap channel the host
// has -- torch.distributed's store, MPI, a file) and owns every buffer.
//
// RCCL is bound at run time (dlopen): the library has no link-time dependency on librccl, and a process that already
// carries one (PyTorch-ROCm does) shares that copy instead of loading a second RCCL.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>
#include "st_common.h"

struct st_dp {
    void* lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*GetVersion)(int*);     // optional (evidence only: bench.py prints which RCCL the ranks met on)
    ncclComm_t comm;
    int rank, world;
    hipStream_t cs;               // communicator stream: collectives run here, beside the compute stream
    hipEvent_t ready[4], done;    // compute -> comm ("bucket final") and comm -> compute ("all reduced")
    hipEvent_t wgfree;            // comm -> compute: the synthesis slab sum (run on the communicator stream) has read the weight-gradient slabs
    int n_issued;
};

namespace stdp {

static void* open_rccl()
{
    // ST_RCCL_LIB: bind THIS library instead (same six entry points).  tests/fake_rccl.cpp uses it to run the world > 1 C path with two
    // processes on one GPU; a site could point it at a differently built RCCL.
    const char* ov = getenv("ST_RCCL_LIB");
    if (ov && *ov) return dlopen(ov, RTLD_NOW | RTLD_LOCAL);
    // already in the process (PyTorch-ROCm links it)?  else the system copy
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (h) return h; }
    for (const char* n : names) { void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) return h; }
    return nullptr;
}

static int bind(st_dp* p)
{
    p->lib = open_rccl();
    if (!p->lib) return st_fail(ST_ERR_UNSUPPORTED, "st_dp: librccl.so not found (%s)", dlerror());
#define ST_SYM(field_, name_) do { *(void**)(&p->field_) = dlsym(p->lib, name_); \
        if (!p->field_) return st_fail(ST_ERR_UNSUPPORTED, "st_dp: symbol %s missing in librccl", name_); } while (0)
    ST_SYM(GetUniqueId, "ncclGetUniqueId"); ST_SYM(CommInitRank, "ncclCommInitRank"); ST_SYM(CommDestroy, "ncclCommDestroy");
    ST_SYM(AllReduce, "ncclAllReduce"); ST_SYM(Broadcast, "ncclBroadcast"); ST_SYM(GetErrorString, "ncclGetErrorString");
#undef ST_SYM
    *(void**)(&p->GetVersion) = dlsym(p->lib, "ncclGetVersion");
    return ST_OK;
}

#define ST_NCCL(p_, call_, what_) do { const ncclResult_t r_ = (call_); \
        if (r_ != ncclSuccess) return st_fail(ST_ERR_LAUNCH, "st_dp %s: %s", what_, (p_)->GetErrorString(r_)); } while (0)
#define ST_HIP(call_, what_) do { const hipError_t e_ = (call_); \
        if (e_ != hipSuccess) return st_fail(ST_ERR_LAUNCH, "st_dp %s: %s", what_, hipGetErrorString(e_)); } while (0)

}  // namespace stdp
