// Ceiling probe for the search kernel's memory pattern: every wavefront gathers random 512-byte rows
// (16 lanes x 16 B per row, 4 rows per load instruction, `inflight` rows issued before the first is
// consumed) with no search logic around the loads.  Occupancy is set the way the search kernel's is: by
// the dynamic LDS size per single-wave workgroup.  Prints GB/s per (queries-per-CU, rows in flight).
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench tools/gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int GROUPS>   // GROUPS load instructions (4 rows each) in flight
__global__ __launch_bounds__(64) void gather_kernel(const float4* __restrict__ rows, unsigned n_rows, unsigned row_f4,
                                                    int iters, float* __restrict__ out) {
    extern __shared__ float lds[];
    const unsigned lane = threadIdx.x, h = lane >> 4, c = lane & 15;
    float acc = 0.f;
    unsigned seed = blockIdx.x * 2654435761U + 12345U;
    for (int it = 0; it < iters; ++it) {
        float4 v[GROUPS][2];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            unsigned r = mix(seed + (unsigned)(it * GROUPS * 4 + g * 4) + h) % n_rows;
            const float4* p = rows + (size_t)r * row_f4;
            v[g][0] = p[c];
            v[g][1] = p[c + 16];
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
            acc += v[g][0].x + v[g][0].w + v[g][1].y + v[g][1].z;
    }
    if (acc == 12345.678f) lds[lane] = acc;     // never true; keeps the LDS allocation and the sums alive
    if (lane == 0) out[blockIdx.x] = acc;
}

// Second pattern: two lanes per row, 8 bytes per lane per instruction; instruction k reads bytes [16k, 16k+16)
// of each of 32 rows, so a 128-byte line is touched by 8 consecutive instructions and the L1 has to
// merge them.  32 instructions (one 512-byte row per lane pair) are in flight together.
__global__ __launch_bounds__(64) void gather_b64_kernel(const float2* __restrict__ rows, unsigned n_rows, unsigned row_f2,
                                                        int iters, float* __restrict__ out) {
    extern __shared__ float lds[];
    const unsigned lane = threadIdx.x, r = lane >> 1, p = lane & 1;
    float acc = 0.f;
    unsigned seed = blockIdx.x * 2654435761U + 12345U;
    for (int it = 0; it < iters; ++it) {
        const unsigned row = mix(seed + (unsigned)(it * 32) + r) % n_rows;
        const float2* q = rows + (size_t)row * row_f2 + p;
        float2 v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = q[2 * k];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(v[k].x, v[k].x, acc) + v[k].y;
    }
    if (acc == 12345.678f) lds[lane] = acc;
    if (lane == 0) out[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const unsigned n_rows = argc > 1 ? atoi(argv[1]) : 1000000, row_f4 = 32;   // 512-byte rows
    const int n_blocks = 10000, iters_total_rows = 1184;                         // ~ one SIFT-1M query's row count
    float4* rows; float* out;
    CHECK(hipMalloc((void**)&rows, (size_t)n_rows * row_f4 * 16));
    CHECK(hipMemset(rows, 0, (size_t)n_rows * row_f4 * 16));
    CHECK(hipMalloc((void**)&out, n_blocks * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int per_cu[] = {4, 8, 10, 13, 16, 20, 26, 32};
    printf("rows=%u x 512 B, %d single-wave workgroups x %d rows each\n", n_rows, n_blocks, iters_total_rows);
    for (int groups : {4, 8, 16}) {
        for (int q : per_cu) {
            unsigned lds_bytes = (160 * 1024 / q) & ~255u;
            if (lds_bytes > 64 * 1024) lds_bytes = 64 * 1024;
            int iters = iters_total_rows / (groups * 4);
            auto launch = [&]() {
                if (groups == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(n_blocks), dim3(64), lds_bytes, 0, rows, n_rows, row_f4, iters, out);
                else if (groups == 8) hipLaunchKernelGGL(gather_kernel<8>, dim3(n_blocks), dim3(64), lds_bytes, 0, rows, n_rows, row_f4, iters, out);
                else hipLaunchKernelGGL(gather_kernel<16>, dim3(n_blocks), dim3(64), lds_bytes, 0, rows, n_rows, row_f4, iters, out);
            };
            if (groups == 4) CHECK(hipFuncSetAttribute((const void*)gather_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            if (groups == 8) CHECK(hipFuncSetAttribute((const void*)gather_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            if (groups == 16) CHECK(hipFuncSetAttribute((const void*)gather_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) launch();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            double bytes = (double)n_blocks * iters * groups * 4 * 512;
            printf("rows_in_flight=%2d queries_per_cu<=%2d lds=%6u B  %.3f ms  %.0f GB/s\n", groups * 4, q, lds_bytes, ms, bytes / ms / 1e6);
            fflush(stdout);
        }
    }
    CHECK(hipFuncSetAttribute((const void*)gather_b64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int q : per_cu) {
        unsigned lds_bytes = (160 * 1024 / q) & ~255u;
        if (lds_bytes > 64 * 1024) lds_bytes = 64 * 1024;
        const int iters = iters_total_rows / 32;
        auto launch = [&]() { hipLaunchKernelGGL(gather_b64_kernel, dim3(n_blocks), dim3(64), lds_bytes, 0, (const float2*)rows, n_rows, row_f4 * 2, iters, out); };
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        double bytes = (double)n_blocks * iters * 32 * 512;
        printf("pair-per-row b64, 32 rows in flight, queries_per_cu<=%2d lds=%6u B  %.3f ms  %.0f GB/s\n", q, lds_bytes, ms, bytes / ms / 1e6);
        fflush(stdout);
    }
    return 0;
}
