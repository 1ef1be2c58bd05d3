// Micro-benchmark: what does a divergent 16-byte vector load cost, and does the cost depend on how many lanes are active?
// Every active lane chases pointers through a 512 KB table of float4 records (each record holds the index of the next
// one): a chain of dependent global_load_dwordx4, addresses scattered over the table, which is L2-resident and mostly
// L1-resident - the access pattern of a BVH walk. 16 waves per CU, as in the renderer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void chase(const float4* __restrict__ table, float* out, int iters, unsigned long long mask, int loadsPerStep)
{
    const int lane = threadIdx.x;
    int idx = (blockIdx.x * 64 + lane) * 97 & 32767;
    float acc = 0.0f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            float4 r = table[idx];
            if (loadsPerStep > 1) { float4 r2 = table[idx ^ 1]; float4 r3 = table[idx ^ 2]; float4 r4 = table[idx ^ 3]; acc += r2.x + r3.x + r4.x; }
            acc += r.x;
            idx = __float_as_int(r.w);
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main()
{
    const int N = 32768;   // 512 KB
    std::vector<float4> h(N);
    unsigned s = 12345;
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u;
        const int next = (int)((s >> 8) % N);
        h[i] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
        reinterpret_cast<int*>(&h[i])[3] = next;
    }
    float4* table; float* out;
    const int blocks = 256 * 16, iters = 2000;
    hipMalloc(&table, N * sizeof(float4)); hipMalloc(&out, blocks * 64 * sizeof(float));
    hipMemcpy(table, h.data(), N * sizeof(float4), hipMemcpyHostToDevice);
    const unsigned long long masks[5] = { ~0ull, 0xffffull, 0x0001000100010001ull, 0xfull, 1ull };
    const char* names[5] = { "all 64 lanes", "lanes 0-15", "lanes 0,16,32,48", "lanes 0-3", "one lane" };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lps = 1; lps <= 4; lps += 3)
        for (int rep = 0; rep < 2; ++rep)
            for (int p = 0; p < 5; ++p) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, table, out, iters, masks[p], lps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                // 16 waves per CU share one texture addresser: cycles of that unit per load instruction if it were the limit
                if (rep) std::printf("%d load(s) per step, %-18s %8.3f ms  -> %.1f cycles per load instruction per CU at 2.4 GHz\n", lps, names[p], ms,
                                     ms * 1e-3 * 2.4e9 / (16.0 * iters * lps));
            }
    return 0;
}
