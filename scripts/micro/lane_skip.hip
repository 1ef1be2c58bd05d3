// Micro-benchmark: does a wave64 VALU instruction cost less when whole 16-lane groups are inactive?
// Every wave runs a long dependent chain of v_fma_f32 under an exec mask given by `pattern`:
//   0: all 64 lanes   1: lanes 0-15   2: lanes 0,16,32,48   3: lanes 0-3   4: lanes 0-31   5: one lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) void chain(float* out, int iters, unsigned long long mask)
{
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-3f, b = 0.999f, c = 1e-3f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            a = a * b + c; a = a * b + c; a = a * b + c; a = a * b + c;
            a = a * b + c; a = a * b + c; a = a * b + c; a = a * b + c;
        }
    }
    out[blockIdx.x * 64 + lane] = a;
}

// the same number of v_fma_f32, as two independent chains per lane
__global__ __launch_bounds__(64) void chain2(float* out, int iters)
{
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-3f, d = 2.0f + lane * 1e-3f, b = 0.999f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {
        a = a * b + c; d = d * b + c; a = a * b + c; d = d * b + c;
        a = a * b + c; d = d * b + c; a = a * b + c; d = d * b + c;
    }
    out[blockIdx.x * 64 + lane] = a + d;
}

int main()
{
    const int blocks = 256 * 16, iters = 20000;
    float* out;
    if (hipMalloc(&out, blocks * 64 * sizeof(float)) != hipSuccess) return 1;
    const unsigned long long masks[6] = { ~0ull, 0xffffull, 0x0001000100010001ull, 0xfull, 0xffffffffull, 1ull };
    const char* names[6] = { "all 64 lanes", "lanes 0-15", "lanes 0,16,32,48", "lanes 0-3", "lanes 0-31", "one lane" };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int p = 0; p < 6; ++p) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, out, iters, masks[p]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) std::printf("%-18s %8.3f ms\n", names[p], ms);
        }
    for (int waves = 1; waves <= 8; waves *= 2) {   // waves per SIMD: 4 * waves blocks per CU
        const int nb = 256 * 4 * waves;
        for (int k = 0; k < 2; ++k) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL(chain, dim3(nb), dim3(64), 0, 0, out, iters, ~0ull);
            else hipLaunchKernelGGL(chain2, dim3(nb), dim3(64), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::printf("%d wave(s) per SIMD, %s: %8.3f ms -> %.2f cycles per wave instruction per SIMD at 2.4 GHz\n", waves,
                        k ? "two chains" : "one chain ", ms, ms * 1e-3 * 2.4e9 / ((double)waves * iters * 8));
        }
    }
    hipFree(out);
    return 0;
}
