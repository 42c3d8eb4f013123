// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access pattern of the traversal kernels:
// every lane gathers 48-byte items (three global_load_dwordx4, 16-byte aligned) scattered over a
// table far larger than L2 and the Infinity Cache.  The program knows exactly which bytes it asks
// for, so the counter of a run under `rocprofv3 --pmc FETCH_SIZE` can be put beside
//   - the algorithmic bytes                     (items x 48),
//   - the 64-byte sectors the items overlap     (x 64),
//   - the 128-byte lines the items overlap      (x 128),
// and beside a plain coalesced stream over the same table (16 bytes per lane), for which the counter
// is known to report half the bytes.
//
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   ./fetch_calib            prints the known byte counts of both kernels (one dispatch each)
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib        (and a second run with the TCC request counters)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr uint32_t kItems = 1u << 25;        // 48-byte items: 1.61 GB
constexpr uint32_t kUnitsPerItem = 3;        // 16-byte units
constexpr int kBlocks = 2048, kThreads = 256, kPerThread = 64;

__host__ __device__ inline uint32_t pcg(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
__host__ __device__ inline uint32_t item_of(uint32_t thread, uint32_t k) { return pcg(thread * 64u + k) & (kItems - 1u); }

__global__ __launch_bounds__(256) void gather_kernel(const float4 *table, float *out)
{
    const uint32_t tid = blockIdx.x * kThreads + threadIdx.x;
    float acc = 0.f;
    for (int k = 0; k < kPerThread; k++) {
        const float4 *p = table + (size_t)item_of(tid, (uint32_t)k) * kUnitsPerItem;
        float4 a, b, c;
        asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                     "global_load_dwordx4 %1, %3, off offset:16\n\t"
                     "global_load_dwordx4 %2, %3, off offset:32\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c)
                     : "v"(p)
                     : "memory");
        acc += (a.x + b.y) + c.z;
    }
    out[tid] = acc;
}

__global__ __launch_bounds__(256) void stream_kernel(const float4 *table, size_t units, float *out)
{
    const size_t tid = (size_t)blockIdx.x * kThreads + threadIdx.x, stride = (size_t)gridDim.x * kThreads;
    float acc = 0.f;
    for (size_t i = tid; i < units; i += stride) {
        const float4 a = table[i];
        acc += a.x;
    }
    out[tid] = acc;
}

int main()
{
    const size_t units = (size_t)kItems * kUnitsPerItem, bytes = units * 16;
    float4 *table;
    float *out;
    CHECK(hipMalloc(&table, bytes + 256));
    CHECK(hipMalloc(&out, (size_t)kBlocks * kThreads * sizeof(float)));
    CHECK(hipMemset(table, 0, bytes + 256));
    CHECK(hipDeviceSynchronize());

    // what the gather asks for
    const uint64_t base = (uint64_t)(uintptr_t)table;
    uint64_t items = 0, sectors64 = 0, lines128 = 0;
    for (uint32_t t = 0; t < (uint32_t)(kBlocks * kThreads); t++)
        for (uint32_t k = 0; k < (uint32_t)kPerThread; k++) {
            const uint64_t a = base + (uint64_t)item_of(t, k) * 48u, b = a + 47u;
            items++;
            sectors64 += (b >> 6) - (a >> 6) + 1;
            lines128 += (b >> 7) - (a >> 7) + 1;
        }
    hipLaunchKernelGGL(gather_kernel, dim3(kBlocks), dim3(kThreads), 0, 0, table, out);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(stream_kernel, dim3(kBlocks), dim3(kThreads), 0, 0, table, units, out);
    CHECK(hipDeviceSynchronize());
    printf("table_bytes %zu\n", bytes);
    printf("gather_items %llu\n", (unsigned long long)items);
    printf("gather_algorithmic_bytes %llu\n", (unsigned long long)(items * 48u));
    printf("gather_sector64_bytes %llu\n", (unsigned long long)(sectors64 * 64u));
    printf("gather_line128_bytes %llu\n", (unsigned long long)(lines128 * 128u));
    printf("stream_bytes %zu\n", bytes);
    CHECK(hipFree(table));
    CHECK(hipFree(out));
    return 0;
}
