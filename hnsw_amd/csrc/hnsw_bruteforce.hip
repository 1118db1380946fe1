// Exact k-nearest-neighbour scan (squared L2) on the GPU: the ground truth recall is
// measured against, and the only matrix-core user of this library.
//
//   d2(q, x) = |q|^2 + |x|^2 - 2 q.x
//
// The q.x part is a GEMM computed with the f32-input MFMA (v_mfma_f32_32x32x2_f32: f32 in,
// f32 accumulate, bit-for-bit an fmaf chain over k in a fixed order), so for the integer-valued sets every
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
#include <cstdlib>
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
constexpr int kLd = kBK + 4;    // LDS row length (floats): tiles are stored as they lie in memory, [row][k]; 36 makes the 16-byte reads
                                // of a lane group (16 rows, 4 banks each) cover the 64 banks exactly once

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

__device__ __forceinline__ float el(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

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
// VEC: rows are a multiple of 4 floats long (16-byte loads).
// Grid: x = tile of 128 queries, y = worker; a worker takes the column tiles y, y + gridDim.y, ... of the chunk one after
// the other (the blocks that share a column tile sit next to each other in launch order, so a tile of base rows is
// fetched from memory about once and then served by the L2s), and asks for its next tile's first operands before it
// turns to the distances of the current one.
// TAIL: the rows end inside a k step (d % 32 != 0): the groups of 8 k past the end of a row are skipped.
// FOLD: d % 32 == 4 (d = 100): instead of a step of its own for four dims, the last full step carries them along — they
//       sit in the four floats of padding every LDS row has, and cost two matrix instructions per tile pair instead of
//       a staging step with its two barriers.
template <bool EMIT, bool VEC, bool TAIL, bool FOLD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void dist_tile_kernel(const float* __restrict__ Q, const float* __restrict__ X,
                                                        const float* __restrict__ qn, const float* __restrict__ xn,
                                                        uint32_t nq, uint32_t n, uint32_t d, uint32_t x0, uint32_t cw, uint32_t ntx,
                                                        float* __restrict__ D, EmitArgs em)
{
    __shared__ __align__(16) float As[kBM * kLd];
    __shared__ __align__(16) float Bs[kBN * kLd];
    __shared__ __align__(16) float qn_s[kBM];
    __shared__ __align__(16) float thr_s[kBM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const uint32_t q0 = blockIdx.x * kBM;                  // first query of the tile
    uint32_t jt = blockIdx.y;                              // column tile in hand
    if (jt >= ntx) return;

    // the tile's query norms and (EMIT) thresholds (the first barrier of the k loop publishes them)
    if (tid < kBM) {
        const uint32_t row = q0 + tid;
        qn_s[tid] = row < nq ? qn[row] : 0.0f;
        if (EMIT) thr_s[tid] = row < nq ? em.best_d[(size_t)row * em.kp + em.kp - 1] : -__builtin_inff();
    }

    // Staging role: the group of 4 consecutive k numbered sc, of rows sr0, sr0 + 32, sr0 + 64, sr0 + 96 of both tiles:
    // eight lanes fetch the 128 contiguous bytes a row contributes to a k step, and store them to LDS as they are.
    // Rows outside the tile read the tile's first row instead: what is computed from them is never looked at (the
    // epilogue skips columns and rows outside), and only k beyond the end of a row has to contribute zero — to one operand.
    const int sc = tid & 7, sr0 = tid >> 3;
    const uint32_t dm = FOLD ? d - 4u : d;                 // the dims the k steps cover
    const float* qt = Q + (size_t)q0 * d;
    const float* xt = nullptr;
    uint32_t xj0 = 0;                         // first column of the tile the staging role points at
    auto aim = [&](uint32_t tile) {           // point the staging role at a column tile
        xj0 = tile * kBN;
        xt = X + ((size_t)x0 + xj0) * d;
    };
    // (row offsets are recomputed at every fetch: eight registers less to carry through the matrix instructions)
    auto qoff = [&](int i) { const uint32_t r = (uint32_t)(sr0 + 32 * i); return (q0 + r < nq ? r : 0u) * d; };
    auto xoff = [&](int i) { const uint32_t r = (uint32_t)(sr0 + 32 * i); return ((xj0 + r < cw && x0 + xj0 + r < n) ? r : 0u) * d; };
    // compute role
    const int fi = lane & 31, fk = lane >> 5;

    // The 4 + 4 groups of the next k step are fetched into registers while the matrix cores work on the current one
    // (16-byte loads when the rows allow it).
    // Nothing touches the fetched registers before the store to LDS one step later (the zero for k beyond the end of a row
    // is selected there): a use any earlier would put the wait for the loads in front of the matrix instructions.
    float ra[16], rb[16];
    float4 rx = make_float4(0.f, 0.f, 0.f, 0.f);      // FOLD: the last four dims of row tid of the A tile (tid < 128) or row tid - 128 of the B tile
    bool inside = true;                 // of the values in flight: is their k inside the row
    auto fetch = [&](uint32_t k0) {
        if constexpr (FOLD) {
            if (k0 + (uint32_t)kBK >= dm) {             // the last step: its four extra dims
                const uint32_t r = (uint32_t)tid & 127u;
                if (tid < 128) rx = *reinterpret_cast<const float4*>(qt + (q0 + r < nq ? r : 0u) * d + dm);
                else rx = *reinterpret_cast<const float4*>(xt + ((xj0 + r < cw && x0 + xj0 + r < n) ? r : 0u) * d + dm);
            }
        }
        const uint32_t kb = k0 + 4u * (uint32_t)sc;
        if constexpr (VEC) {
            // always load (groups past the end of a row read its last group)
            inside = kb < d;                                    // d % 4 == 0: a group of 4 is inside the row or outside it
            const uint32_t kc = inside ? kb : d - 4u;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 va = *reinterpret_cast<const float4*>(qt + qoff(i) + kc);
                const float4 vb = *reinterpret_cast<const float4*>(xt + xoff(i) + kc);
                ra[4 * i] = va.x; ra[4 * i + 1] = va.y; ra[4 * i + 2] = va.z; ra[4 * i + 3] = va.w;
                rb[4 * i] = vb.x; rb[4 * i + 1] = vb.y; rb[4 * i + 2] = vb.z; rb[4 * i + 3] = vb.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float* qr = qt + qoff(i);
                const float* xr = xt + xoff(i);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    ra[4 * i + e] = kb + e < d ? qr[kb + e] : 0.0f;
                    rb[4 * i + e] = kb + e < d ? xr[kb + e] : 0.0f;
                }
            }
        }
    };
    aim(jt);
    fetch(0);
    for (; jt < ntx; jt += gridDim.y) {
        const uint32_t j0 = jt * kBN;
        // the norms of this lane's two columns, asked for now and used after the k loop
        float xv[2];
        bool col_ok[2];
#pragma unroll
        for (int tb = 0; tb < 2; tb++) {
            const uint32_t col = j0 + wc * 64 + tb * 32 + (lane & 31);
            col_ok[tb] = col < cw && x0 + col < n;
            xv[tb] = xn[col_ok[tb] ? x0 + col : x0 + j0];
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

        for (uint32_t k0 = 0; k0 < dm; k0 += kBK) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                *reinterpret_cast<float4*>(&As[(sr0 + 32 * i) * kLd + 4 * sc]) =
                    make_float4(inside ? ra[4 * i] : 0.f, inside ? ra[4 * i + 1] : 0.f, inside ? ra[4 * i + 2] : 0.f, inside ? ra[4 * i + 3] : 0.f);
                *reinterpret_cast<float4*>(&Bs[(sr0 + 32 * i) * kLd + 4 * sc]) = make_float4(rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]);
            }
            const bool last_step = FOLD && k0 + (uint32_t)kBK >= dm;
            if (last_step) *reinterpret_cast<float4*>(&(tid < 128 ? As : Bs)[(tid & 127) * kLd + kBK]) = rx;
            __syncthreads();
            if (k0 + kBK < dm) fetch(k0 + kBK);
            else if (jt + gridDim.y < ntx) { aim(jt + gridDim.y); fetch(0); }       // the next tile's first step
            // One 16-byte LDS read hands a lane the operands of four matrix instructions: the lane of row fi and half fk takes
            // k = 8 j + 4 fk + e for e = 0..3, so instruction (j, e) multiplies k = 8 j + e (lower half of the wave) and
            // k = 8 j + 4 + e (upper half) — every k of the step exactly once, in an order that is the same for every output.
            // (the reads of group j + 1 are issued before the matrix instructions of group j)
            float4 a[2], b[2];
            auto read_group = [&](int j, float4 (&av)[2], float4 (&bv)[2]) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    av[t] = *reinterpret_cast<const float4*>(&As[(wr * 64 + t * 32 + fi) * kLd + 8 * j + 4 * fk]);
                    bv[t] = *reinterpret_cast<const float4*>(&Bs[(wc * 64 + t * 32 + fi) * kLd + 8 * j + 4 * fk]);
                }
            };
            // (a row that ends inside this step: the groups of 8 k past its end hold nothing but zeros and are skipped,
            // so d = 100 costs 13 groups, not 16)
            const int nj = (!TAIL || d - k0 >= (uint32_t)kBK) ? kBK / 8 : (int)((d - k0 + 7u) >> 3);
            read_group(0, a, b);
#pragma unroll
            for (int j = 0; j < kBK / 8; j++) {
                if (TAIL && j >= nj) break;
                float4 an[2], bn[2];
                if (TAIL ? j + 1 < nj : j + 1 < kBK / 8) read_group(j + 1, an, bn);
                __builtin_amdgcn_sched_barrier(0);          // (the scheduler would otherwise sink the reads to where their registers free up)
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int ta = 0; ta < 2; ta++)
#pragma unroll
                        for (int tb = 0; tb < 2; tb++)
                            acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(el(a[ta], e), el(b[tb], e), acc[ta][tb], 0, 0, 0);
                if (TAIL ? j + 1 < nj : j + 1 < kBK / 8) {
#pragma unroll
                    for (int t = 0; t < 2; t++) { a[t] = an[t]; b[t] = bn[t]; }
                }
            }
            if (last_step) {
                // k = 32 .. 35 of the step: the lane of half fk takes k = 32 + 2 fk + e for e = 0, 1, so instruction e multiplies
                // k = 32 + e (lower half of the wave) and k = 34 + e (upper half)
                float2 a2[2], b2[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    a2[t] = *reinterpret_cast<const float2*>(&As[(wr * 64 + t * 32 + fi) * kLd + kBK + 2 * fk]);
                    b2[t] = *reinterpret_cast<const float2*>(&Bs[(wc * 64 + t * 32 + fi) * kLd + kBK + 2 * fk]);
                }
#pragma unroll
                for (int ta = 0; ta < 2; ta++)
#pragma unroll
                    for (int tb = 0; tb < 2; tb++) {
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[ta].x, b2[tb].x, acc[ta][tb], 0, 0, 0);
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[ta].y, b2[tb].y, acc[ta][tb], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
        // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): a lane's 16
        // registers are 4 groups of 4 consecutive rows, so a group's norms and thresholds are one 16-byte LDS read each
        // (read where they are used: the operands of the next tile are in flight and registers are short)
        uint32_t row_base = q0 + (uint32_t)(wr * 64 + 4 * (lane >> 5));
        asm volatile("" : "+v"(row_base));        // (keeps the 64 row addresses of the rare emission from being carried through the k loop)
#pragma unroll
        for (int ta = 0; ta < 2; ta++) {
#pragma unroll
            for (int tb = 0; tb < 2; tb++) {
                const uint32_t col = j0 + wc * 64 + tb * 32 + (lane & 31);
                float dist[16];
                uint32_t pass = 0;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int base = wr * 64 + ta * 32 + 8 * g + 4 * (lane >> 5);
                    const float4 qv = *reinterpret_cast<const float4*>(qn_s + base);
                    float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (EMIT) tv = *reinterpret_cast<const float4*>(thr_s + base);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * g + e;
                        dist[r] = (el(qv, e) + xv[tb]) - 2.0f * acc[ta][tb][r];
                        if (EMIT) pass |= (dist[r] <= el(tv, e) ? 1u : 0u) << r;      // (rows past nq have threshold -inf)
                    }
                }
                if constexpr (EMIT) {
                    if (!col_ok[tb]) pass = 0;
                    if (__ballot(pass != 0u) == 0ull) continue;          // nothing of this 32 x 32 part survives: the common case late in the scan
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        if ((pass >> r) & 1u) {
                            const uint32_t row = row_base + (uint32_t)(ta * 32 + (r & 3) + 8 * (r >> 2));
                            const uint32_t at = atomicAdd(&em.cnt[row], 1u);
                            if (at < em.cap) em.cand[(size_t)row * em.cap + at] = make_uint2(__float_as_uint(dist[r]), em.id_offset + x0 + col);
                        }
                } else {
                    if (!col_ok[tb]) continue;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const uint32_t row = row_base + (uint32_t)(ta * 32 + (r & 3) + 8 * (r >> 2));
                        if (row < nq) D[(size_t)row * cw + col] = dist[r];
                    }
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
    const bool vec = (d & 3u) == 0 && ((uintptr_t)d_base & 15u) == 0 && ((uintptr_t)d_queries & 15u) == 0;
    // four dims past the last full k step ride along with it (HNSW_AMD_BF_NO_FOLD: measure without)
    const bool fold = vec && d >= 36u && (d % (uint32_t)kBK) == 4u && getenv("HNSW_AMD_BF_NO_FOLD") == nullptr;
    const bool tail = !fold && (d % (uint32_t)kBK) != 0;                  // rows end inside a k step
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
        // column-tile workers of a launch: up to 8 tiles each, as long as that leaves the chip several rounds of blocks
        const uint32_t nty = (nq + kBM - 1) / kBM;
        auto workers_for = [&](uint32_t ntx) {
            uint32_t per = 8;
            while (per > 1 && (uint64_t)((ntx + per - 1) / per) * nty < 4096u) per >>= 1;
            if (const char* e = getenv("HNSW_AMD_BF_TILES_PER_WORKER")) per = (uint32_t)std::max(1, atoi(e));     // tests: the multi-tile path on small inputs
            return (ntx + per - 1) / per;
        };
        // rows [lo, hi) the plain way (for every query, or only for those whose flag is set)
        auto plain = [&](uint32_t lo, uint32_t hi, int only_flagged) {
            for (uint32_t x0 = lo; x0 < hi; x0 += cw) {
                const uint32_t valid = std::min(cw, hi - x0);
                timed([&] {
                    const uint32_t ntx = (valid + kBN - 1) / kBN;
                    hipLaunchKernelGGL((fold ? dist_tile_kernel<false, true, false, true> : tail ? (vec ? dist_tile_kernel<false, true, true> : dist_tile_kernel<false, false, true>) : (vec ? dist_tile_kernel<false, true, false> : dist_tile_kernel<false, false, false>)), dim3((nq + kBM - 1) / kBM, workers_for(ntx)), dim3(256), 0, stream,
                                       d_queries, d_base, qn, xn, nq, n, d, x0, cw, ntx, D, EmitArgs{});
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
                const uint32_t ntx = (width + kBN - 1) / kBN;
                hipLaunchKernelGGL((fold ? dist_tile_kernel<true, true, false, true> : tail ? (vec ? dist_tile_kernel<true, true, true> : dist_tile_kernel<true, false, true>) : (vec ? dist_tile_kernel<true, true, false> : dist_tile_kernel<true, false, false>)), dim3((nq + kBM - 1) / kBM, workers_for(ntx)), dim3(256), 0, stream,
                                   d_queries, d_base, qn, xn, nq, n, d, seen, width, ntx, (float*)nullptr, em);
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
