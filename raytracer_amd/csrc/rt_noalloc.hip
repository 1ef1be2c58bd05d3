// rt_noalloc.hip - device memory that exists without any allocation call, for the one entry point
// of librt_amd.so that callers who cannot allocate may use (code under HIP graph capture, tight
// serving loops, harnesses that forbid hipMalloc): rt_frame_stats_noalloc (rt_host.cpp).
//
// The counting build of the megakernel normally adds every wave's work counters to one DevCounters
// record the library allocated at rt_create. Here every wave stores its partial counters into its
// own slot of a file-scope device array instead, and rt_sum_partials adds the slots up into the
// caller's output. The array is part of the code object: every GPU the library runs on holds its own
// copy (RT_NOALLOC_SCRATCH_BYTES, 4 MiB) from the moment the module is loaded until the process ends.
//
// On the host, the array's name is only a handle for hipGetSymbolAddress; it is never passed to a
// kernel or a copy. rt_noalloc_setup() asks, once per device and with that device current, for the
// address of that device's copy and keeps it; everything else uses the kept pointer.
//
// Every call on a GPU shares that GPU's array: two calls that overlap on one GPU (two contexts, two
// streams) race. Calls on one stream are ordered. A call overwrites the slots it reads before it
// reads them (every launched wave stores its slot, rt_sum_partials reads exactly that many), so
// nothing has to be cleared between calls.
#include <hip/hip_runtime.h>

#include <mutex>

#include "rt_device.h"

#define RT_NOALLOC_MAX_WAVES 131072   /* 8x8-pixel waves of one dispatch: 8 388 608 pixels, e.g. 3840x2160 */
#define RT_NOALLOC_SCRATCH_BYTES (RT_NOALLOC_MAX_WAVES * sizeof(DevCounters))

static __device__ __align__(256) unsigned long long scratch[RT_NOALLOC_MAX_WAVES * (sizeof(DevCounters) / sizeof(unsigned long long))];

namespace {

const int kMaxDevices = 64;
void* g_scratchPtr[kMaxDevices];   // per device: address of that device's copy of `scratch`; null until set up
std::mutex g_setupLock;

// out = sum of n per-wave counter records. One block of 256 threads.
__global__ __launch_bounds__(256) void rt_sum_partials(const DevCounters* __restrict__ partials, int n, DevCounters* __restrict__ out)
{
    __shared__ unsigned long long part[4][256];
    unsigned long long r = 0, nd = 0, t = 0, s = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const DevCounters c = partials[i];
        r += c.rays; nd += c.nodeTests; t += c.triTests; s += c.samples;
    }
    part[0][threadIdx.x] = r; part[1][threadIdx.x] = nd; part[2][threadIdx.x] = t; part[3][threadIdx.x] = s;
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {
        if ((int)threadIdx.x < half)
            for (int k = 0; k < 4; ++k) part[k][threadIdx.x] += part[k][threadIdx.x + half];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out->rays = part[0][0]; out->nodeTests = part[1][0]; out->triTests = part[2][0]; out->samples = part[3][0];
    }
}

} // namespace

// Called by rt_create with `device` current. Idempotent per device.
extern "C" hipError_t rt_noalloc_setup(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_setupLock);
    if (g_scratchPtr[device]) return hipSuccess;
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratchPtr[device] = p;
    return hipSuccess;
}

// The kept device address (null: rt_noalloc_setup has not succeeded on that device) and the
// number of per-wave slots behind it.
extern "C" void* rt_noalloc_scratch(int device)
{
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> g(g_setupLock);
    return g_scratchPtr[device];
}

extern "C" long long rt_noalloc_max_waves(void) { return RT_NOALLOC_MAX_WAVES; }

// partials: the pointer rt_noalloc_scratch returned; out: device-accessible DevCounters.
extern "C" hipError_t rt_noalloc_sum(const void* partials, int n, void* out, hipStream_t stream)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    if (!partials || !out || n < 0 || n > RT_NOALLOC_MAX_WAVES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rt_sum_partials, dim3(1), dim3(256), 0, stream, static_cast<const DevCounters*>(partials), n,
                       static_cast<DevCounters*>(out));
    return hipGetLastError();
}
