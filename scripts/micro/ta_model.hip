// Microbenchmark of the vector-memory front end on gfx950: how many clocks does a wave pay to bring one
// 64-byte item per lane into registers, as a function of how the 16-byte pieces are spread over lanes?
//   A  every lane loads the four 16-byte pieces of its own item (4 x global_load_dwordx4, scattered lanes)
//   B  quad-cooperative: in load i the four lanes of a quad load the four pieces of the item of quad lane i
//      (each quad touches one contiguous 64-byte segment per load); data ends up transposed
//   C  as B but straight into LDS (global_load_lds_dwordx4), then each lane reads its item back with
//      4 x ds_read_b128
//   D  as A with only `active` of 64 lanes enabled
// Items are chosen pseudo-randomly in a buffer of `bytes` bytes (64-byte aligned).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ inline float sum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ inline uint32_t rng(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int I> __device__ inline uint32_t quad_bcast(uint32_t v)
{
    return __builtin_amdgcn_mov_dpp(v, I * 0x55, 0xf, 0xf, true); // quad_perm [I,I,I,I]
}
__device__ inline uint32_t quad_lane(uint32_t v, int i)
{
    return i == 0 ? quad_bcast<0>(v) : i == 1 ? quad_bcast<1>(v) : i == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v);
}

template <int VARIANT>
__global__ __launch_bounds__(256) void bench(const float4 *buf, uint32_t items, int iters, int active, float *out)
{
    __shared__ float4 stage[4][4][64]; // [wave][load][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    const bool on = VARIANT != 3 || ((lane * 37) & 63) < active;
    for (int it = 0; it < iters; it++) {
        const uint32_t item = rng(s) % items;     // this lane's 64-byte item
        const float4 *p = buf + (size_t)item * 4;
        if (VARIANT == 0 || VARIANT == 3) {
            if (on) {
                float4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += (sum4(a) + sum4(b)) + (sum4(c) + sum4(d));
            }
        } else if (VARIANT == 1) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t other = quad_lane(item, i);
                v[i] = buf[(size_t)other * 4 + (lane & 3)];
            }
            acc += (sum4(v[0]) + sum4(v[1])) + (sum4(v[2]) + sum4(v[3]));
        } else if (VARIANT == 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t other = quad_lane(item, i);
                __builtin_amdgcn_global_load_lds((glb_void *)(buf + (size_t)other * 4 + (lane & 3)),
                                                 (lds_void *)&stage[wave][i][0], 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00 | 0xc000); // vmcnt(0)
            const float4 *mine = &stage[wave][lane & 3][(lane >> 2) * 4];
            float4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
            acc += (sum4(a) + sum4(b)) + (sum4(c) + sum4(d));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int iters = 2000;
    const int blocks = 256 * 5;
    float *out;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    const size_t sizes[] = {16u << 10, 1u << 20, 64u << 20};
    for (size_t bytes : sizes) {
        float4 *buf;
        CHECK(hipMalloc(&buf, bytes + 64));
        CHECK(hipMemset(buf, 0, bytes + 64));
        const uint32_t items = (uint32_t)(bytes / 64);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        auto run = [&](int variant, int active, const char *name) {
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                switch (variant) {
                case 0: bench<0><<<blocks, 256>>>(buf, items, iters, active, out); break;
                case 1: bench<1><<<blocks, 256>>>(buf, items, iters, active, out); break;
                case 2: bench<2><<<blocks, 256>>>(buf, items, iters, active, out); break;
                default: bench<3><<<blocks, 256>>>(buf, items, iters, active, out); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
            }
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            // waves per CU = 20; clocks per wave-iteration per CU at 2.4 GHz
            const double clk = ms * 1e-3 * 2.4e9 / ((double)iters * 20);
            printf("%8zu KiB  %-28s %8.3f ms  %7.1f CU-clocks per wave-iteration\n", bytes >> 10, name, ms, clk);
        };
        run(0, 64, "A own item, 4 loads");
        run(1, 64, "B quad-cooperative");
        run(2, 64, "C quad-cooperative via LDS");
        run(3, 32, "D own item, 32 lanes");
        run(3, 16, "D own item, 16 lanes");
        CHECK(hipFree(buf));
    }
    return 0;
}
