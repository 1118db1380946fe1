// Exact k-nearest-neighbour scan (squared L2) on the GPU: the ground truth recall is
// measured against, and the only matrix-core user of this library.
//
//   d2(q, x) = |q|^2 + |x|^2 - 2 q.x
//
// The q.x part is a GEMM computed with the f32-input MFMA (v_mfma_f32_32x32x2_f32: f32 in,
// f32 accumulate, bit-for-bit a k-ordered fmaf chain), so for the integer-valued sets every
// distance is exact.  Base rows are processed in chunks.  The first chunk (a few thousand rows)
// goes the plain way: a tile kernel writes its distance matrix, a selection kernel (one
// wavefront per query) folds it into the running top-k.  From then on every query has a
// k-th-best distance, and the tile kernel itself drops everything beyond it: of a 128 x 128 tile
// only the handful of (distance, id) pairs that can still enter a top-k list leave the chip,
// appended to per-query candidate lists that the selection kernel folds in.  Chunks grow
// fourfold, so a chunk offers a query about 4 k survivors whatever the size of the base, and the
// n x nq distance matrix is never written.  (A query whose list overflows — rows arriving in
// adverse order — is redone for that chunk the plain way.)  The kept lists carry a margin of
// 8 entries beyond k; at the end their distances are recomputed directly as sum (q - x)^2 in
// double precision and the lists sorted by that, so that for real-valued data the rounding of
// |q|^2 + |x|^2 - 2 q.x cannot reorder near ties.
// The reference has no counterpart: it reads ground truth from the dataset's "neighbors" array
// (src/main.cpp:190-195, src/hnsw.cpp:89-91).
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

// What the tile kernel does with a tile's distances when it filters them itself (EMIT).
struct EmitArgs {
    const float* best_d;   // nq x kp running lists: a query's threshold is its last entry
    uint32_t kp;
    uint2* cand;           // nq x cap (distance bits, id)
    uint32_t* cnt;         // nq: entries appended (may exceed cap: the list then counts as overflowed)
    uint32_t cap;
    uint32_t id_offset;
};

// dist(q, j) = qn[q] + xn[x0 + j] - 2 * dot(Q[q], X[x0 + j]) for q < nq, j < cw.
// EMIT = false: written to D[q][j].  EMIT = true: appended to query q's candidate list when it does not exceed the
// query's threshold.
// 256 threads = 4 waves in 2 x 2; each wave owns a 64 x 64 part = 2 x 2 MFMA tiles of 32 x 32.
template <bool EMIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void dist_tile_kernel(const float* __restrict__ Q, const float* __restrict__ X,
                                                        const float* __restrict__ qn, const float* __restrict__ xn,
                                                        uint32_t nq, uint32_t n, uint32_t d, uint32_t x0, uint32_t cw,
                                                        float* __restrict__ D, EmitArgs em)
{
    __shared__ __align__(16) float As[kBK * kLd];
    __shared__ __align__(16) float Bs[kBK * kLd];
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
            // always load (rows outside the tile read row 0, groups past the end of a row its last group) and zero
            // afterwards: selects instead of a branch around every load
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const bool inside = kb + j < d;                 // d % 4 == 0: a group of 4 is inside the row or outside it
                const uint32_t kc = inside ? kb + j : d - 4u;
                float4 va = *reinterpret_cast<const float4*>(qp + kc);
                float4 vb = *reinterpret_cast<const float4*>(xp + kc);
                if (!(inside && qok)) va = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!(inside && xok)) vb = make_float4(0.f, 0.f, 0.f, 0.f);
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
    // the tile's query norms and (EMIT) thresholds, staged once in the LDS the operands no longer need
    float* qn_s = As;
    float* thr_s = As + kBM;
    if (tid < kBM) {
        const uint32_t row = q0 + tid;
        qn_s[tid] = row < nq ? qn[row] : 0.0f;
        if (EMIT) thr_s[tid] = row < nq ? em.best_d[(size_t)row * em.kp + em.kp - 1] : -__builtin_inff();
    }
    __syncthreads();
    // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): a lane's 16
    // registers are 4 groups of 4 consecutive rows, so the rows' norms and thresholds are 4 + 4 16-byte LDS reads
#pragma unroll
    for (int ta = 0; ta < 2; ta++) {
        float4 qv[4], tv[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int base = wr * 64 + ta * 32 + 8 * g + 4 * (lane >> 5);
            qv[g] = *reinterpret_cast<const float4*>(qn_s + base);
            if (EMIT) tv[g] = *reinterpret_cast<const float4*>(thr_s + base);
        }
        auto pick = [](const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); };
#pragma unroll
        for (int tb = 0; tb < 2; tb++) {
            const uint32_t col = j0 + wc * 64 + tb * 32 + (lane & 31);
            const bool col_ok = col < cw && x0 + col < n;
            const float xv = col_ok ? xn[x0 + col] : 0.0f;
            float dist[16];
            uint32_t pass = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                dist[r] = (pick(qv[r >> 2], r & 3) + xv) - 2.0f * acc[ta][tb][r];
                if (EMIT) pass |= (dist[r] <= pick(tv[r >> 2], r & 3) ? 1u : 0u) << r;      // (rows past nq have threshold -inf)
            }
            if constexpr (EMIT) {
                if (!col_ok) pass = 0;
                if (__ballot(pass != 0u) == 0ull) continue;          // nothing of this 32 x 32 part survives: the common case late in the scan
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if ((pass >> r) & 1u) {
                        const uint32_t row = q0 + wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const uint32_t at = atomicAdd(&em.cnt[row], 1u);
                        if (at < em.cap) em.cand[(size_t)row * em.cap + at] = make_uint2(__float_as_uint(dist[r]), em.id_offset + x0 + col);
                    }
            } else {
                if (!col_ok) continue;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t row = q0 + wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < nq) D[(size_t)row * cw + col] = dist[r];
                }
            }
        }
    }
}

// PAIRS = false: the candidates are row q of the chunk's distance matrix D (n_valid columns, ids id_offset + x0 + j);
//   with `only_flagged`, queries whose flag is clear are skipped (they took the other path for this chunk).
// PAIRS = true: the candidates are query q's list cand[q][0 .. cnt[q]); a list that overflowed (cnt[q] > cap) is not
//   folded at all and the query's flag is set instead.
constexpr int kSelBuf = 64;
template <bool PAIRS>
__global__ __launch_bounds__(64) void select_topk_kernel(const float* __restrict__ D, uint32_t cw, uint32_t x0, uint32_t id_offset,
                                                         uint32_t n_valid, uint32_t k, float* best_d, uint32_t* best_i,
                                                         const uint2* __restrict__ cand, const uint32_t* __restrict__ cnt_in, uint32_t cap,
                                                         uint32_t* flags, uint32_t* any_flag, int only_flagged)
{
    extern __shared__ __align__(8) uint2 pool[];     // k + kSelBuf entries: (distance bits, id)
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    if constexpr (PAIRS) {
        n_valid = cnt_in[q];
        if (n_valid > cap) {
            if (lane == 0) { flags[q] = 1u; atomicOr(any_flag, 1u); }
            return;
        }
        if (lane == 0) flags[q] = 0u;
        if (n_valid == 0) return;
    } else {
        if (only_flagged && flags[q] == 0u) return;
    }
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
        uint2 mine[4]; uint32_t rk[4]; int nm = 0;      // k + kSelBuf <= 136 + 64 entries over 64 lanes
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
        if (j < n_valid) {
            if constexpr (PAIRS) e = cand[(size_t)q * cap + j];
            else e = make_uint2(__float_as_uint(D[(size_t)q * cw + j]), id_offset + x0 + j);
        }
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

// One wavefront per query: the kp kept entries get their distance recomputed as sum (q - x)^2 in double precision,
// are sorted by (that distance, id), and the first k are written out.
__global__ __launch_bounds__(64) void rescore_kernel(const float* __restrict__ Q, const float* __restrict__ X, uint32_t d, uint32_t id_offset,
                                                     const uint32_t* __restrict__ best_i, uint32_t kp, uint32_t k,
                                                     float* __restrict__ out_d, uint32_t* __restrict__ out_i)
{
    extern __shared__ __align__(8) uint2 pool[];     // kp entries
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const float* qp = Q + (size_t)q * d;
    for (uint32_t j = 0; j < kp; j++) {
        const uint32_t id = best_i[(size_t)q * kp + j];
        if (id == 0xFFFFFFFFu) { if (lane == 0) pool[j] = make_uint2(0x7F800000u, id); continue; }
        const float* xp = X + (size_t)(id - id_offset) * d;
        double s = 0.0;
        for (uint32_t t = lane; t < d; t += 64) { const double df = (double)qp[t] - (double)xp[t]; s += df * df; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) pool[j] = make_uint2(__float_as_uint((float)s), id);
    }
    __syncthreads();
    for (uint32_t i = lane; i < kp; i += 64) {
        const uint2 me = pool[i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < kp; j++) {
            const uint2 o = pool[j];
            r += (o.x < me.x || (o.x == me.x && (o.y < me.y || (o.y == me.y && j < i)))) ? 1u : 0u;     // (distances are non-negative: bits order)
        }
        if (r < k) { out_d[(size_t)q * k + r] = __uint_as_float(me.x); out_i[(size_t)q * k + r] = me.y; }
    }
}

}  // namespace

// Returns the milliseconds spent in the MFMA tile kernels (device time), for the utilisation figure.
float bruteforce_knn(const float* d_base, uint32_t n, const float* d_queries, uint32_t nq, uint32_t d, uint32_t k,
                     uint32_t id_offset, uint32_t* d_out_ids, float* d_out_dist, void* stream_v)
{
    if (!d_base || !d_queries || !d_out_ids || !d_out_dist) throw std::invalid_argument("bruteforce_knn: NULL argument");
    if (n == 0 || nq == 0 || d == 0 || k == 0 || k > 128) throw std::invalid_argument("bruteforce_knn: need n, nq, d >= 1 and 1 <= k <= 128");
    hipStream_t stream = (hipStream_t)stream_v;
    const uint32_t kp = k + 8;                        // kept entries: a margin beyond k for the final re-scoring
    constexpr uint32_t kCap = 1024;                   // candidate list per query and chunk
    // plain path: distance matrix of at most about 1 GiB at a time
    const uint32_t cw = (uint32_t)std::min<uint64_t>(n, std::max<uint64_t>(kBN, ((1ull << 28) / nq) / kBN * kBN));
    const uint32_t first = std::min<uint32_t>(n, std::max<uint32_t>(8192u, 8u * kp));
    float *qn = nullptr, *xn = nullptr, *D = nullptr, *best_d = nullptr;
    uint32_t *best_i = nullptr, *cnt = nullptr, *flags = nullptr, *h_any = nullptr;
    uint2* cand = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float mfma_ms = 0.0f;
    auto release = [&]() {
        for (void* p : {(void*)qn, (void*)xn, (void*)D, (void*)best_d, (void*)best_i, (void*)cnt, (void*)flags, (void*)cand}) if (p) (void)hipFree(p);
        if (h_any) (void)hipHostFree(h_any);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    try {
        BF_CHECK(hipMalloc((void**)&qn, (size_t)nq * 4));
        BF_CHECK(hipMalloc((void**)&xn, (size_t)n * 4));
        BF_CHECK(hipMalloc((void**)&D, (size_t)nq * std::min(cw, std::max(first, cw)) * 4));
        BF_CHECK(hipMalloc((void**)&best_d, (size_t)nq * kp * 4));
        BF_CHECK(hipMalloc((void**)&best_i, (size_t)nq * kp * 4));
        BF_CHECK(hipMalloc((void**)&flags, ((size_t)nq + 1) * 4));
        BF_CHECK(hipHostMalloc((void**)&h_any, 64, hipHostMallocDefault));
        BF_CHECK(hipEventCreate(&e0)); BF_CHECK(hipEventCreate(&e1));
        uint32_t* any_flag = flags + nq;
        hipLaunchKernelGGL(row_norms_kernel, dim3((nq + 3) / 4), dim3(256), 0, stream, d_queries, nq, d, qn);
        hipLaunchKernelGGL(row_norms_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, d_base, n, d, xn);
        const size_t total = (size_t)nq * kp;
        hipLaunchKernelGGL(fill_topk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, best_d, best_i, total);
        const size_t pool_bytes = (size_t)(kp + kSelBuf) * sizeof(uint2);
        auto timed = [&](auto&& launch) {
            BF_CHECK(hipEventRecord(e0, stream));
            launch();
            BF_CHECK(hipEventRecord(e1, stream));
            BF_CHECK(hipGetLastError());
            BF_CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            BF_CHECK(hipEventElapsedTime(&ms, e0, e1));
            mfma_ms += ms;
        };
        // rows [lo, hi) the plain way (for every query, or only for those whose flag is set)
        auto plain = [&](uint32_t lo, uint32_t hi, int only_flagged) {
            for (uint32_t x0 = lo; x0 < hi; x0 += cw) {
                const uint32_t valid = std::min(cw, hi - x0);
                timed([&] {
                    hipLaunchKernelGGL(dist_tile_kernel<false>, dim3((valid + kBN - 1) / kBN, (nq + kBM - 1) / kBM), dim3(256), 0, stream,
                                       d_queries, d_base, qn, xn, nq, n, d, x0, cw, D, EmitArgs{});
                });
                hipLaunchKernelGGL(select_topk_kernel<false>, dim3(nq), dim3(64), pool_bytes, stream, D, cw, x0, id_offset, valid, kp, best_d,
                                   best_i, (const uint2*)nullptr, (const uint32_t*)nullptr, 0u, flags, any_flag, only_flagged);
            }
        };
        plain(0, first, 0);
        uint32_t seen = first;
        if (seen < n) {
            BF_CHECK(hipMalloc((void**)&cand, (size_t)nq * kCap * sizeof(uint2)));
            BF_CHECK(hipMalloc((void**)&cnt, (size_t)nq * 4));
        }
        while (seen < n) {
            // (a chunk four times what has been seen offers every query about 4 kp survivors)
            const uint32_t width = (uint32_t)std::min<uint64_t>(n - seen, 4ull * seen);
            BF_CHECK(hipMemsetAsync(cnt, 0, (size_t)nq * 4, stream));
            BF_CHECK(hipMemsetAsync(any_flag, 0, 4, stream));
            const EmitArgs em{best_d, kp, cand, cnt, kCap, id_offset};
            timed([&] {
                hipLaunchKernelGGL(dist_tile_kernel<true>, dim3((width + kBN - 1) / kBN, (nq + kBM - 1) / kBM), dim3(256), 0, stream,
                                   d_queries, d_base, qn, xn, nq, n, d, seen, width, (float*)nullptr, em);
            });
            hipLaunchKernelGGL(select_topk_kernel<true>, dim3(nq), dim3(64), pool_bytes, stream, (const float*)nullptr, 0u, 0u, id_offset, 0u, kp,
                               best_d, best_i, (const uint2*)cand, (const uint32_t*)cnt, kCap, flags, any_flag, 0);
            BF_CHECK(hipMemcpyAsync(h_any, any_flag, 4, hipMemcpyDeviceToHost, stream));
            BF_CHECK(hipStreamSynchronize(stream));
            if (h_any[0]) plain(seen, seen + width, 1);       // some lists overflowed: those queries see the chunk the plain way
            seen += width;
        }
        hipLaunchKernelGGL(rescore_kernel, dim3(nq), dim3(64), (size_t)kp * sizeof(uint2), stream, d_queries, d_base, d, id_offset,
                           (const uint32_t*)best_i, kp, k, d_out_dist, d_out_ids);
        BF_CHECK(hipGetLastError());
        BF_CHECK(hipStreamSynchronize(stream));
    } catch (...) {
        release();
        throw;
    }
    release();
    return mfma_ms;
}

}  // namespace hnsw_amd
