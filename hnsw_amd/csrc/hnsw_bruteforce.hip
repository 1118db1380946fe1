// Exact k-nearest-neighbour scan (squared L2) on the GPU: the ground truth recall is
// measured against, and the only matrix-core user of this library.
//
//   d2(q, x) = |q|^2 + |x|^2 - 2 q.x
//
// The q.x part is a GEMM computed with the f32-input MFMA (v_mfma_f32_32x32x2_f32: f32 in,
// f32 accumulate, bit-for-bit a k-ordered fmaf chain), so for the integer-valued sets every
// distance is exact.  Base rows are processed in chunks: a tile kernel writes the chunk's
// distance matrix, a selection kernel (one wavefront per query) folds it into the running
// top-k.  The reference has no counterpart: it reads ground truth from the dataset's
// "neighbors" array (src/main.cpp:190-195, src/hnsw.cpp:89-91).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>
#include <string>

#include "../../include/hnsw_amd.h"

namespace hnsw_amd {

#define BF_CHECK(expr)                                                                               \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            throw std::runtime_error(std::string("HIP: ") + #expr + ": " + hipGetErrorString(e_));   \
    } while (0)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128;        // queries per block tile
constexpr int kBN = 128;        // base rows per block tile
constexpr int kBK = 32;         // k per staging step
constexpr int kLd = kBM + 4;    // LDS row length (floats): tiles are stored k-major, [k][row]

__global__ void row_norms_kernel(const float* x, uint32_t n, uint32_t d, float* out)
{
    const uint32_t row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    float s = 0.0f;
    for (uint32_t j = lane; j < d; j += 64) { const float v = x[(size_t)row * d + j]; s = __builtin_fmaf(v, v, s); }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s;
}

// D[q][j] = qn[q] + xn[x0 + j] - 2 * dot(Q[q], X[x0 + j]) for q < nq, j < cw.
// 256 threads = 4 waves in 2 x 2; each wave owns a 64 x 64 part = 2 x 2 MFMA tiles of 32 x 32.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void dist_tile_kernel(const float* __restrict__ Q, const float* __restrict__ X,
                                                        const float* __restrict__ qn, const float* __restrict__ xn,
                                                        uint32_t nq, uint32_t n, uint32_t d, uint32_t x0, uint32_t cw,
                                                        float* __restrict__ D)
{
    __shared__ float As[kBK * kLd];
    __shared__ float Bs[kBK * kLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const uint32_t q0 = blockIdx.y * kBM, j0 = blockIdx.x * kBN;       // tile origin (query, column of the chunk)

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    // staging role: row r of the tile, 16 consecutive k starting at 16*kh
    const int sr = tid & 127, kh = tid >> 7;
    const uint32_t qrow = q0 + sr, xcol = j0 + sr;
    const bool qok = qrow < nq, xok = xcol < cw && x0 + xcol < n;
    const float* qp = Q + (size_t)(qok ? qrow : 0) * d;
    const float* xp = X + (size_t)(xok ? x0 + xcol : 0) * d;
    // compute role
    const int fi = lane & 31, fk = lane >> 5;

    // The 16 + 16 values of the next k step are fetched into registers while the matrix cores work on
    // the current one (16-byte loads when the rows allow it).
    float ra[16], rb[16];
    const bool vec = (d & 3u) == 0;
    auto fetch = [&](uint32_t k0) {
        const uint32_t kb = k0 + 16 * kh;
        if (vec) {
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
                if (kb + j < d) {                       // d % 4 == 0: a group of 4 is inside the row or outside it
                    if (qok) va = *reinterpret_cast<const float4*>(qp + kb + j);
                    if (xok) vb = *reinterpret_cast<const float4*>(xp + kb + j);
                }
                ra[j] = va.x; ra[j + 1] = va.y; ra[j + 2] = va.z; ra[j + 3] = va.w;
                rb[j] = vb.x; rb[j + 1] = vb.y; rb[j + 2] = vb.z; rb[j + 3] = vb.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t k = kb + j;
                ra[j] = (qok && k < d) ? qp[k] : 0.0f;
                rb[j] = (xok && k < d) ? xp[k] : 0.0f;
            }
        }
    };
    fetch(0);
    for (uint32_t k0 = 0; k0 < d; k0 += kBK) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            As[(16 * kh + j) * kLd + sr] = ra[j];
            Bs[(16 * kh + j) * kLd + sr] = rb[j];
        }
        __syncthreads();
        if (k0 + kBK < d) fetch(k0 + kBK);
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                a[t] = As[(kk + fk) * kLd + wr * 64 + t * 32 + fi];
                b[t] = Bs[(kk + fk) * kLd + wc * 64 + t * 32 + fi];
            }
#pragma unroll
            for (int ta = 0; ta < 2; ta++)
#pragma unroll
                for (int tb = 0; tb < 2; tb++) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
    }
    // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int ta = 0; ta < 2; ta++)
#pragma unroll
        for (int tb = 0; tb < 2; tb++) {
            const uint32_t col = j0 + wc * 64 + tb * 32 + (lane & 31);
            if (col >= cw || x0 + col >= n) continue;
            const float xv = xn[x0 + col];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t row = q0 + wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < nq) D[(size_t)row * cw + col] = (qn[row] + xv) - 2.0f * acc[ta][tb][r];
            }
        }
}

// One wavefront per query: fold the chunk's distances into the query's running top-k
// (best_d/best_i: k entries ascending by (distance, id), +inf / 0xFFFFFFFF when empty).
constexpr int kSelBuf = 64;
__global__ __launch_bounds__(64) void select_topk_kernel(const float* __restrict__ D, uint32_t cw, uint32_t x0, uint32_t id_offset,
                                                         uint32_t n_valid, uint32_t k, float* best_d, uint32_t* best_i)
{
    extern __shared__ __align__(8) uint2 pool[];     // k + kSelBuf entries: (distance bits, id)
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    for (uint32_t i = lane; i < k; i += 64) pool[i] = make_uint2(__float_as_uint(best_d[(size_t)q * k + i]), best_i[(size_t)q * k + i]);
    __syncthreads();
    uint32_t cnt = 0;                                 // buffered candidates beyond the k kept ones
    uint2 worst = pool[k - 1];
    auto better = [](uint2 a, uint2 b) {              // a strictly before b in (distance, id) order; distances may be negative zero-ish
        const float fa = __uint_as_float(a.x), fb = __uint_as_float(b.x);
        return fa < fb || (fa == fb && a.y < b.y);
    };
    auto fold = [&]() {
        // rank sort of the k + cnt pooled entries, keep the first k
        const uint32_t total = k + cnt;
        uint2 mine[3]; uint32_t rk[3]; int nm = 0;
        for (uint32_t i = lane; i < total; i += 64) {
            const uint2 me = pool[i];
            uint32_t r = 0;
            for (uint32_t j = 0; j < total; j++) { const uint2 o = pool[j]; r += (better(o, me) || (o.x == me.x && o.y == me.y && j < i)) ? 1u : 0u; }
            mine[nm] = me; rk[nm] = r; nm++;
        }
        __syncthreads();
        for (int t = 0; t < nm; t++) if (rk[t] < k) pool[rk[t]] = mine[t];
        __syncthreads();
        cnt = 0;
        worst = pool[k - 1];
    };
    for (uint32_t c0 = 0; c0 < n_valid; c0 += 64) {
        const uint32_t j = c0 + lane;
        uint2 e = make_uint2(0x7F800000u, 0xFFFFFFFFu);
        if (j < n_valid) e = make_uint2(__float_as_uint(D[(size_t)q * cw + j]), id_offset + x0 + j);
        const bool cand = j < n_valid && better(e, worst);
        const unsigned long long m = __ballot(cand);
        const uint32_t add = (uint32_t)__popcll(m);
        if (add == 0) continue;
        if (cnt + add > (uint32_t)kSelBuf) fold();
        // `worst` may have tightened: re-test
        const bool cand2 = cand && better(e, worst);
        const unsigned long long m2 = __ballot(cand2);
        if (cand2) pool[k + cnt + (uint32_t)__popcll(m2 & ((1ull << lane) - 1ull))] = e;
        cnt += (uint32_t)__popcll(m2);
        __syncthreads();
    }
    if (cnt) fold();
    for (uint32_t i = lane; i < k; i += 64) { best_d[(size_t)q * k + i] = __uint_as_float(pool[i].x); best_i[(size_t)q * k + i] = pool[i].y; }
}

__global__ void fill_topk_kernel(float* d, uint32_t* i, size_t count)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) { d[t] = __builtin_inff(); i[t] = 0xFFFFFFFFu; }
}

}  // namespace

// Returns the milliseconds spent in the MFMA tile kernel (device time), for the utilisation figure.
float bruteforce_knn(const float* d_base, uint32_t n, const float* d_queries, uint32_t nq, uint32_t d, uint32_t k,
                     uint32_t id_offset, uint32_t* d_out_ids, float* d_out_dist, void* stream_v)
{
    if (!d_base || !d_queries || !d_out_ids || !d_out_dist) throw std::invalid_argument("bruteforce_knn: NULL argument");
    if (n == 0 || nq == 0 || d == 0 || k == 0 || k > 128) throw std::invalid_argument("bruteforce_knn: need n, nq, d >= 1 and 1 <= k <= 128");
    hipStream_t stream = (hipStream_t)stream_v;
    // chunk width: keep the distance matrix of one chunk around 1 GiB
    uint32_t cw = (uint32_t)std::min<uint64_t>(n, std::max<uint64_t>(kBN, ((1ull << 28) / nq) / kBN * kBN));
    float *qn = nullptr, *xn = nullptr, *D = nullptr;
    BF_CHECK(hipMalloc((void**)&qn, (size_t)nq * 4));
    BF_CHECK(hipMalloc((void**)&xn, (size_t)n * 4));
    BF_CHECK(hipMalloc((void**)&D, (size_t)nq * cw * 4));
    hipEvent_t e0, e1;
    BF_CHECK(hipEventCreate(&e0)); BF_CHECK(hipEventCreate(&e1));
    float mfma_ms = 0.0f;
    try {
        hipLaunchKernelGGL(row_norms_kernel, dim3((nq + 3) / 4), dim3(256), 0, stream, d_queries, nq, d, qn);
        hipLaunchKernelGGL(row_norms_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, d_base, n, d, xn);
        const size_t cnt = (size_t)nq * k;
        hipLaunchKernelGGL(fill_topk_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, d_out_dist, d_out_ids, cnt);
        for (uint32_t x0 = 0; x0 < n; x0 += cw) {
            const uint32_t valid = std::min(cw, n - x0);
            BF_CHECK(hipEventRecord(e0, stream));
            hipLaunchKernelGGL(dist_tile_kernel, dim3((valid + kBN - 1) / kBN, (nq + kBM - 1) / kBM), dim3(256), 0, stream,
                               d_queries, d_base, qn, xn, nq, n, d, x0, cw, D);
            BF_CHECK(hipEventRecord(e1, stream));
            hipLaunchKernelGGL(select_topk_kernel, dim3(nq), dim3(64), (size_t)(k + kSelBuf) * sizeof(uint2), stream, D, cw, x0,
                               id_offset, valid, k, d_out_dist, d_out_ids);
            BF_CHECK(hipGetLastError());
            BF_CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            BF_CHECK(hipEventElapsedTime(&ms, e0, e1));
            mfma_ms += ms;
        }
        BF_CHECK(hipStreamSynchronize(stream));
    } catch (...) {
        (void)hipFree(qn); (void)hipFree(xn); (void)hipFree(D); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        throw;
    }
    (void)hipFree(qn); (void)hipFree(xn); (void)hipFree(D); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return mfma_ms;
}

}  // namespace hnsw_amd
