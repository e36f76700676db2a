// fake_rccl.cpp -- TEST DOUBLE for librccl (tests only; never shipped, never linked into the product library).
//
// libsignaltrain_hip.so binds RCCL at run time (st_dp.h: dlopen + dlsym of six symbols).  With ST_RCCL_LIB pointing at this library the
// data-parallel C path -- st_dp_init / st_dp_broadcast / st_dp_train_step with its three exchanges issued from the middle of the
// backward -- runs with world > 1 on ONE GPU: every rank is a process with its own HIP context on the same device, and the
// "fabric" is a POSIX shared-memory segment.  The collectives are stream-ordered exactly like RCCL's:
//     all-reduce  =  copy my buffer into my slot of the segment (D2H, on the caller's stream)
//                 -> host function on the stream: arrive + wait for every rank ("all slots written")
//                 -> copy the peers' slots into a device scratch (H2D), sum slot 0 + slot 1 + ... in RANK ORDER (bit-identical on all ranks)
//                 -> host function: arrive + wait ("all slots read": the segment may be overwritten)
// Deliberately SLOW (PCIe both ways) and optionally skewed (ST_FAKE_RCCL_DELAY_RANK / ST_FAKE_RCCL_DELAY_US: that rank sleeps before
// it arrives): a consumer that does not wait for the collective, or a collective that does not wait for its producer, computes with
// stale data and the world-2 == single-process comparison of tests/test_dp_world2.py fails.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclFloat32 = 7, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;

#define FAKE_MAX_RANKS 8
#define FAKE_SLOT_BYTES ((size_t)64 << 20)          // per-rank slot: 64 MiB (the largest exchange of the tests is 8.4 MB)

struct Control {
    std::atomic<uint64_t> arrived[FAKE_MAX_RANKS];  // monotone per-rank phase counters
    std::atomic<uint32_t> attached;
};
struct fakeComm {
    int rank, world;
    char name[64];
    Control* ctl; char* slots; size_t map_bytes;
    float* scratch;                                 // device: one slot
    uint64_t phase;                                 // phases issued so far (host side, in issue order)
    long delay_us;
};
typedef fakeComm* ncclComm_t;

static void sleep_us(long us) { struct timespec ts = {us / 1000000, (us % 1000000) * 1000}; nanosleep(&ts, nullptr); }

struct Arrive { fakeComm* c; uint64_t phase; bool delay; };
static void arrive_and_wait(void* p)
{
    Arrive* a = (Arrive*)p;
    fakeComm* c = a->c;
    if (a->delay && c->delay_us > 0) sleep_us(c->delay_us);
    c->ctl->arrived[c->rank].store(a->phase, std::memory_order_release);
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < c->world; ++r)
        while (c->ctl->arrived[r].load(std::memory_order_acquire) < a->phase) {
            sleep_us(50);
            struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) > 120.0) { fprintf(stderr, "fake_rccl: rank %d timed out waiting for rank %d (phase %llu)\n", c->rank, r, (unsigned long long)a->phase); abort(); }
        }
    delete a;
}
static hipError_t barrier_on_stream(fakeComm* c, hipStream_t s, bool delay)
{
    Arrive* a = new Arrive{c, ++c->phase, delay};
    return hipLaunchHostFunc(s, arrive_and_wait, a);
}

__global__ void fake_sum_kernel(const float* __restrict__ mine, const float* __restrict__ others, float* __restrict__ out, size_t n, int rank, int world)
{
    // slot r of `others` holds rank r's data (the own slot is not used: `mine` is the same data); summed in rank order
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < world; ++r) s += (r == rank) ? mine[i] : others[(size_t)r * n + i];
        out[i] = s;
    }
}

// bfloat16 payloads (the packed last exchange of the 16-bit configurations): summed the way a RING reduction sums them -- every hop adds two bfloat16
// values and forwards the bfloat16-ROUNDED partial sum (RCCL's reduce-scatter moves the payload type), so the result carries world - 1 roundings, not one
// (a single final rounding is strictly more accurate than the real collective and would not bound its error).  Rank order.
__device__ __forceinline__ unsigned short fake_bf16_rne(float s)
{
    unsigned u = __float_as_uint(s);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__global__ void fake_sum_bf16_kernel(const unsigned short* __restrict__ mine, const unsigned short* __restrict__ others, unsigned short* __restrict__ out, size_t n, int rank, int world)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned short acc = (0 == rank) ? mine[i] : others[i];
        for (int r = 1; r < world; ++r) {
            const unsigned short v = (r == rank) ? mine[i] : others[(size_t)r * n + i];
            acc = fake_bf16_rne(__uint_as_float((unsigned)acc << 16) + __uint_as_float((unsigned)v << 16));
        }
        out[i] = acc;
    }
}

ncclResult_t ncclGetVersion(int* v) { if (!v) return ncclInvalidArgument; *v = -1; return ncclSuccess; }      // -1: "this is the test double"

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/st_fake_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
    if (world < 1 || world > FAKE_MAX_RANKS || rank < 0 || rank >= world) return ncclInvalidArgument;
    fakeComm* c = new fakeComm; memset(c, 0, sizeof(*c));
    c->rank = rank; c->world = world; strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->map_bytes = 4096 + (size_t)world * FAKE_SLOT_BYTES;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { perror("fake_rccl shm_open"); return ncclSystemError; }
    } else {
        for (int tries = 0; tries < 6000 && fd < 0; ++tries) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) sleep_us(10000); }
        struct stat st;
        for (int tries = 0; tries < 6000; ++tries) { if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break; sleep_us(10000); }
        if (fd < 0) { perror("fake_rccl shm_open (peer)"); return ncclSystemError; }
    }
    void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { perror("fake_rccl mmap"); return ncclSystemError; }
    c->ctl = (Control*)m; c->slots = (char*)m + 4096;
    if (hipHostRegister(m, c->map_bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); }   // pageable copies still work (synchronously)
    if (hipMalloc(&c->scratch, (size_t)world * FAKE_SLOT_BYTES) != hipSuccess) return ncclUnhandledCudaError;
    const char* dr = getenv("ST_FAKE_RCCL_DELAY_RANK"); const char* du = getenv("ST_FAKE_RCCL_DELAY_US");
    if (dr && du && atoi(dr) == rank) c->delay_us = atol(du);
    c->ctl->attached.fetch_add(1);
    for (int tries = 0; tries < 12000 && c->ctl->attached.load() < (uint32_t)world; ++tries) sleep_us(10000);      // init is a collective
    if (c->ctl->attached.load() < (uint32_t)world) { fprintf(stderr, "fake_rccl: rank %d: peers never attached\n", rank); return ncclSystemError; }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    (void)hipHostUnregister(c->ctl);
    if (c->scratch) (void)hipFree(c->scratch);
    munmap(c->ctl, c->map_bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    const size_t esz = dt == ncclBfloat16 ? 2 : 4;
    if ((dt != ncclFloat32 && dt != ncclBfloat16) || op != ncclSum || count * esz > FAKE_SLOT_BYTES) return ncclInvalidArgument;
    const size_t bytes = count * esz;
    if (hipMemcpyAsync(c->slots + (size_t)c->rank * FAKE_SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    if (barrier_on_stream(c, s, true) != hipSuccess) return ncclUnhandledCudaError;                 // all slots written
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && hipMemcpyAsync((char*)c->scratch + (size_t)r * bytes, c->slots + (size_t)r * FAKE_SLOT_BYTES, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    if (dt == ncclBfloat16) hipLaunchKernelGGL(fake_sum_bf16_kernel, dim3(256), dim3(256), 0, s, (const unsigned short*)send, (const unsigned short*)c->scratch, (unsigned short*)recv, count, c->rank, c->world);
    else
    hipLaunchKernelGGL(fake_sum_kernel, dim3(256), dim3(256), 0, s, (const float*)send, (const float*)c->scratch, (float*)recv, count, c->rank, c->world);
    if (barrier_on_stream(c, s, false) != hipSuccess) return ncclUnhandledCudaError;                // all slots read
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t s)
{
    if (dt != ncclFloat32 || count * 4 > FAKE_SLOT_BYTES || root < 0 || root >= c->world) return ncclInvalidArgument;
    const size_t bytes = count * 4;
    if (c->rank == root && hipMemcpyAsync(c->slots + (size_t)root * FAKE_SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    if (barrier_on_stream(c, s, true) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank != root && hipMemcpyAsync(recv, c->slots + (size_t)root * FAKE_SLOT_BYTES, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    if (barrier_on_stream(c, s, false) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // extern "C"
