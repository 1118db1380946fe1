// HIP kernels of the HNSW query path for gfx950 (MI355X), and their launch code.
//
// One 64-lane wavefront (= one workgroup) runs one query from entry point to sorted
// result, entirely on chip except for the adjacency rows and base vectors it gathers
// from HBM:
//   - query vector, candidate heap C, result heap W, exact visited hash set: LDS;
//   - a neighbour list is one coalesced read (lane j reads neighbour j); the unvisited
//     neighbours are compacted with a ballot, and their base-vector rows are gathered
//     16 lanes per row (256 contiguous bytes per row per instruction, 4 rows per
//     instruction), all rows of an expansion in flight together;
//   - distances reproduce the reference's float arithmetic bit for bit: the 4 lane sums of
//     its SSE loop are 4 sequential fused-multiply-add chains per row, so the differences
//     are transposed through a small LDS tile and 64 lanes each run one (row, lane-sum)
//     chain (16 rows x 4 sums at a time);
//   - heap updates follow libstdc++'s push_heap/pop_heap step by step (one lane), because
//     with the reference's int32 candidate keys ties are the rule and their pop order
//     decides which nodes are expanded.
// Follows reference src/hnsw.cpp:373-449 (aproximateKnn), :676-717 (searchLayerOne),
// :725-787 (searchLayer), src/hnsw.h:334-400 (distance), :55-125 (visited set).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "hnsw_device.h"

namespace hnsw_amd {

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            throw std::runtime_error(std::string("HIP: ") + #expr + ": " + hipGetErrorString(e_));   \
    } while (0)

namespace {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;

__device__ __forceinline__ int32_t trunc_key(float f)
{
    // float -> int32 as x86 cvttss2si does it for the reference (src/hnsw.cpp:732,767)
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return (int32_t)0x80000000;
    return (int32_t)f;
}

// The workgroup is one wavefront, and a wavefront's LDS instructions execute in program
// order, so data one lane wrote is visible to any lane's later read without a barrier or a
// wait: only the compiler has to be kept from reordering the accesses.
#define LDS_SYNC() asm volatile("" ::: "memory")

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rflf(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// ---- W: max-heap on the float distance; entries (dist bits, id).  libstdc++ __push_heap /
// __adjust_heap with comp(a,b) = a.dist < b.dist (reference src/node.h:48-52). ----
__device__ __forceinline__ void w_push(uint2* a, uint32_t n)
{
    uint2 v = a[n - 1];
    float vd = __uint_as_float(v.x);
    uint32_t hole = n - 1;
    while (hole > 0) {
        uint32_t parent = (hole - 1) >> 1;
        uint2 p = a[parent];
        if (!(__uint_as_float(p.x) < vd)) break;
        a[hole] = p; hole = parent;
    }
    a[hole] = v;
}

__device__ __forceinline__ void w_pop(uint2* a, uint32_t n)
{
    if (n <= 1) return;
    uint32_t len = n - 1;
    uint2 v = a[len];
    float vd = __uint_as_float(v.x);
    a[len] = a[0];
    uint32_t hole = 0, child = 0;
    while (child < (len - 1) >> 1) {
        child = 2 * (child + 1);
        uint2 r = a[child], l = a[child - 1];
        if (__uint_as_float(r.x) < __uint_as_float(l.x)) { child--; r = l; }
        a[hole] = r; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) >> 1) {
        child = 2 * (child + 1);
        a[hole] = a[child - 1]; hole = child - 1;
    }
    while (hole > 0) {
        uint32_t parent = (hole - 1) >> 1;
        uint2 p = a[parent];
        if (!(__uint_as_float(p.x) < vd)) break;
        a[hole] = p; hole = parent;
    }
    a[hole] = v;
}

// ---- C: min-heap on an int32 key; entries (key, id); comp(a,b) = a.key > b.key
// (reference src/hnsw.h:210-215). ----
__device__ __forceinline__ void c_push(uint2* a, uint32_t n)
{
    uint2 v = a[n - 1];
    int32_t vk = (int32_t)v.x;
    uint32_t hole = n - 1;
    while (hole > 0) {
        uint32_t parent = (hole - 1) >> 1;
        uint2 p = a[parent];
        if (!((int32_t)p.x > vk)) break;
        a[hole] = p; hole = parent;
    }
    a[hole] = v;
}

__device__ __forceinline__ void c_pop(uint2* a, uint32_t n)
{
    if (n <= 1) return;
    uint32_t len = n - 1;
    uint2 v = a[len];
    int32_t vk = (int32_t)v.x;
    a[len] = a[0];
    uint32_t hole = 0, child = 0;
    while (child < (len - 1) >> 1) {
        child = 2 * (child + 1);
        uint2 r = a[child], l = a[child - 1];
        if ((int32_t)r.x > (int32_t)l.x) { child--; r = l; }
        a[hole] = r; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) >> 1) {
        child = 2 * (child + 1);
        a[hole] = a[child - 1]; hole = child - 1;
    }
    while (hole > 0) {
        uint32_t parent = (hole - 1) >> 1;
        uint2 p = a[parent];
        if (!((int32_t)p.x > vk)) break;
        a[hole] = p; hole = parent;
    }
    a[hole] = v;
}

// ---------------------------------------------------------------------------------------
// The same two libstdc++ routines executed by the whole wavefront in a constant number of
// LDS round trips instead of one per tree level.  They leave the array exactly as
// std::push_heap / std::pop_heap do:
//  - push: the new element climbs the fixed chain of ancestors of index n-1 while
//    comp(ancestor, value); lane k owns ancestor k, one ballot gives the climb length.
//  - pop: __adjust_heap first walks the hole from the root to a leaf choosing, at every
//    node, a child by comparing the two SIBLINGS only (never the moving value), so every
//    node's choice can be evaluated at once (one ballot per 64 internal nodes) and the
//    walk becomes bit tests; then the saved last element climbs back up that same path
//    (again one ballot).  Lane k owns step k of the path.
// Both are called with wave-uniform arguments by all 64 lanes.
// ---------------------------------------------------------------------------------------
struct WOrder {   // comp(a,b) = a.dist < b.dist
    static __device__ __forceinline__ bool before(uint2 a, uint2 b) { return __uint_as_float(a.x) < __uint_as_float(b.x); }
};
struct COrder {   // comp(a,b) = a.key > b.key
    static __device__ __forceinline__ bool before(uint2 a, uint2 b) { return (int32_t)a.x > (int32_t)b.x; }
};

// Maximum of an unsigned value over the 64 lanes with DPP moves (no LDS round trips):
// within quads, half rows and rows, then across the four rows; the result is read from lane 63.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
    uint32_t y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false); x = x > y ? x : y;    // quad_perm [1,0,3,2]
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false); x = x > y ? x : y;    // quad_perm [2,3,0,1]
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, false); x = x > y ? x : y;   // row_half_mirror
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, false); x = x > y ? x : y;   // row_mirror
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x142, 0xA, 0xF, false); x = x > y ? x : y;   // row_bcast15 -> rows 1,3
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x143, 0xC, 0xF, false); x = x > y ? x : y;   // row_bcast31 -> rows 2,3
    return __builtin_amdgcn_readlane(x, 63);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
    uint32_t y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x142, 0xA, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x143, 0xC, 0xF, false); x = x < y ? x : y;
    return __builtin_amdgcn_readlane(x, 63);
}

constexpr uint32_t kWaveHeapMax = 513;   // larger heaps use the one-lane routines

template <class Ord>
__device__ __forceinline__ void heap_push_wave(uint2* a, uint32_t n /* size including the new element */, uint2 v, int lane)
{
    const uint32_t depth = 31u - (uint32_t)__clz((int)n);        // levels above index n-1
    uint2 x = make_uint2(0, 0);
    bool c = false;
    if ((uint32_t)lane < depth) { x = a[(n >> (lane + 1)) - 1]; c = Ord::before(x, v); }
    const unsigned long long m = __ballot(c);
    const uint32_t t = (uint32_t)__builtin_ctzll(~m);            // ancestors that move down
    if ((uint32_t)lane < t) a[(n >> lane) - 1] = x;
    if ((uint32_t)lane == t) a[(n >> t) - 1] = v;
}

// Returns the new root (valid when n > 1).
template <class Ord>
__device__ __forceinline__ uint2 heap_pop_wave(uint2* a, uint32_t n /* size before the pop */, int lane)
{
    const uint32_t len = n - 1;
    const uint2 v = a[len];
    const uint32_t nint = (len - 1) >> 1;      // nodes with both children inside [0, len)
    // lane i (+64, +128, +192): the child node i hands the hole to
    uint32_t nx0 = 0, nx1 = 0, nx2 = 0, nx3 = 0;
    {
        uint32_t i = lane;
        if (i < nint) nx0 = 2 * i + 2 - (Ord::before(a[2 * i + 2], a[2 * i + 1]) ? 1u : 0u);
        if (nint > 64)  { i = lane + 64;  if (i < nint) nx1 = 2 * i + 2 - (Ord::before(a[2 * i + 2], a[2 * i + 1]) ? 1u : 0u); }
        if (nint > 128) { i = lane + 128; if (i < nint) nx2 = 2 * i + 2 - (Ord::before(a[2 * i + 2], a[2 * i + 1]) ? 1u : 0u); }
        if (nint > 192) { i = lane + 192; if (i < nint) nx3 = 2 * i + 2 - (Ord::before(a[2 * i + 2], a[2 * i + 1]) ? 1u : 0u); }
    }
    // walk the hole down; lane k records step k of the path (from -> to)
    uint32_t cur = 0, d = 0, my_from = 0, my_to = 0;
    while (cur < nint) {
        const uint32_t w = cur >> 6, l = cur & 63;
        uint32_t nxt;
        if (w == 0) nxt = __builtin_amdgcn_readlane(nx0, l);
        else if (w == 1) nxt = __builtin_amdgcn_readlane(nx1, l);
        else if (w == 2) nxt = __builtin_amdgcn_readlane(nx2, l);
        else nxt = __builtin_amdgcn_readlane(nx3, l);
        if ((uint32_t)lane == d) { my_from = nxt; my_to = cur; }
        cur = nxt; d++;
    }
    if ((len & 1u) == 0 && cur == (len - 2) >> 1) {      // a last node with a left child only
        const uint32_t nxt = 2 * cur + 1;
        if ((uint32_t)lane == d) { my_from = nxt; my_to = cur; }
        cur = nxt; d++;
    }
    uint2 x = make_uint2(0, 0);
    bool c = false;
    if ((uint32_t)lane < d) { x = a[my_from]; c = Ord::before(x, v); }
    const unsigned long long cm = __ballot(c);
    uint32_t s = 0;                                       // path elements (from the leaf up) that stay put
    if (d > 0) { s = (uint32_t)__builtin_clzll(~(cm << (64 - d))); if (s > d) s = d; }
    const uint32_t moved = d - s;
    if ((uint32_t)lane < moved) a[my_to] = x;
    if (s == 0) { if (lane == 0) a[cur] = v; }
    else if ((uint32_t)lane == moved) a[my_to] = v;
    // new root: the first path element if it moved up, else the re-inserted value
    uint2 root;
    root.x = moved > 0 ? rfl(x.x) : v.x;
    root.y = moved > 0 ? rfl(x.y) : v.y;
    return root;
}

// Exact visited set: open addressing, linear probing, one LDS compare-and-swap per probe.
// Returns true when `id` was not in the set (and now is).
__device__ __forceinline__ bool visited_insert(uint32_t* hash, uint32_t mask, uint32_t shift, uint32_t id)
{
    uint32_t slot = (id * 0x9E3779B1u) >> shift;
    for (;;) {
        uint32_t old = atomicCAS(&hash[slot], kEmpty, id);
        if (old == kEmpty) return true;
        if (old == id) return false;
        slot = (slot + 1) & mask;
    }
}

// Diagnostic build only (-DHNSW_DIAG): the counters struct is overwritten with cycle sums
// per phase (s_memtime), never in the shipped library.
#ifdef HNSW_DIAG
#define DIAG_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define DIAG_ADD(acc, a, b) acc += (uint32_t)((b) - (a))
#else
#define DIAG_STAMP(var)
#define DIAG_ADD(acc, a, b)
#endif

// How the 64 lanes share the rows of one batch of neighbours.
//  SCALAR=false (the reference's SSE loop, src/hnsw.h:334-383): the 4 lane sums of a row are 4
//    independent sequential fused-multiply-add chains over the row's 16-byte chunks in index
//    order.  Two lanes per row: lane p of the pair takes the 8 bytes of every chunk that feed
//    sums 2p and 2p+1, so each lane runs two chains and nothing is exchanged until the final
//    ((s0+s1)+s2)+s3.  32 rows per batch.
//  SCALAR=true (the -DNO_MANUAL_VECTORIZATION build, src/hnsw.h:387-400): one running sum per
//    row, so one lane per row reads whole 16-byte chunks.  64 rows per batch.
// One load instruction therefore reads 16 bytes of each row of the batch; the 8 instructions
// that touch one 128-byte line are adjacent in the instruction stream and merge in the
// vector L1, and all loads of two groups of chunks are in flight before the first is used.
template <bool SCALAR>
struct RowMap {
    static constexpr int kLanesPerRow = SCALAR ? 1 : 2;
    static constexpr int kRowsPerBatch = 64 / kLanesPerRow;
    static constexpr int kGroupChunks = SCALAR ? 8 : 16;      // 16-byte chunks per group (32 registers per lane)
    static constexpr int kGroupFloats = kGroupChunks * 4;     // dims per group
};

struct DistCtx {
    const float* qs;        // query in LDS, zero padded
    uint32_t nfull;         // groups of chunks processed without per-chunk tests
    uint32_t ngroups;       // nfull (+1 when tail_chunks > 0)
    uint32_t tail_chunks;   // chunks of the last, partial group
    uint32_t d4;            // scalar mode: dims whose square is rounded before the add
#ifdef HNSW_DIAG
    mutable uint32_t cy_load = 0, cy_chain = 0;
#endif
};

template <bool SCALAR>
__device__ __forceinline__ DistCtx make_dist_ctx(const float* qs, uint32_t d)
{
    DistCtx c;
    c.qs = qs;
    c.d4 = d & ~3u;
    if (SCALAR) {
        // every dim takes part; only the groups entirely below d4 are free of the fma tail
        const uint32_t chunks = (d + 3u) >> 2;
        c.nfull = c.d4 / RowMap<true>::kGroupFloats;
        c.tail_chunks = chunks - c.nfull * RowMap<true>::kGroupChunks;
    } else {
        const uint32_t d_eff = d & ~15u;          // the SSE loop drops the last d mod 16 dims
        c.nfull = d_eff / RowMap<false>::kGroupFloats;
        c.tail_chunks = (d_eff % RowMap<false>::kGroupFloats) >> 2;
    }
    c.ngroups = c.nfull + (c.tail_chunks ? 1u : 0u);
    return c;
}

// SSE order: this lane's 8 bytes of every chunk.
template <bool FULL>
__device__ __forceinline__ void issue_group(const float* rp, uint32_t nch, float2 (&v)[16])
{
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (FULL || (uint32_t)k < nch) v[k] = *reinterpret_cast<const float2*>(rp + 4 * k);
}

template <bool FULL>
__device__ __forceinline__ void consume_group(const float* qp, uint32_t nch, const float2 (&v)[16], float& a0, float& a1)
{
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (FULL || (uint32_t)k < nch) {
            const float2 q = *reinterpret_cast<const float2*>(qp + 4 * k);
            const float t0 = v[k].x - q.x, t1 = v[k].y - q.y;
            a0 = __builtin_fmaf(t0, t0, a0);
            a1 = __builtin_fmaf(t1, t1, a1);
        }
}

// Scalar order: whole chunks, one running sum; squares are rounded before the add for the
// first 4*(d/4) dims and fused for the last d%4 (what g++ 11 emits for that loop).
template <bool FULL>
__device__ __forceinline__ void issue_group(const float* rp, uint32_t nch, float4 (&v)[8])
{
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (FULL || (uint32_t)k < nch) v[k] = *reinterpret_cast<const float4*>(rp + 4 * k);
}

template <bool FULL>
__device__ __forceinline__ void consume_group(const float* qp, uint32_t nch, uint32_t e0, uint32_t d4, const float4 (&v)[8], float& a)
{
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (FULL || (uint32_t)k < nch) {
            const float4 q = *reinterpret_cast<const float4*>(qp + 4 * k);
            const float t0 = v[k].x - q.x, t1 = v[k].y - q.y, t2 = v[k].z - q.z, t3 = v[k].w - q.w;
            if (FULL || e0 + 4 * k < d4) {
                const float s0 = t0 * t0, s1 = t1 * t1, s2 = t2 * t2, s3 = t3 * t3;
                a = a + s0; a = a + s1; a = a + s2; a = a + s3;
            } else {
                a = __builtin_fmaf(t0, t0, a); a = __builtin_fmaf(t1, t1, a);
                a = __builtin_fmaf(t2, t2, a); a = __builtin_fmaf(t3, t3, a);
            }
        }
}

// Distance from the query to the row `row` points at, for the lanes with `active` set; the
// result is valid in the first lane of every row's lane group.  Called by all 64 lanes.
template <bool SCALAR>
__device__ __forceinline__ float row_distance(const DistCtx& c, const float* row, bool active, int lane)
{
    typedef RowMap<SCALAR> RM;
    float a0 = 0.0f, a1 = 0.0f;
    DIAG_STAMP(d0);
    if (active) {
        const uint32_t ng = c.ngroups, nf = c.nfull, tc = c.tail_chunks;
        if (SCALAR) {
            float4 A[8], B[8];
            const float* qs = c.qs;
            if (0 < nf) issue_group<true>(row, 0, A); else issue_group<false>(row, tc, A);
            if (ng > 1) { if (1 < nf) issue_group<true>(row + RM::kGroupFloats, 0, B); else issue_group<false>(row + RM::kGroupFloats, tc, B); }
            for (uint32_t g = 0; g < ng; g += 2) {
                if (g < nf) consume_group<true>(qs + g * RM::kGroupFloats, 0, 0, 0, A, a0);
                else consume_group<false>(qs + g * RM::kGroupFloats, tc, g * RM::kGroupFloats, c.d4, A, a0);
                if (g + 2 < ng) {
                    if (g + 2 < nf) issue_group<true>(row + (g + 2) * RM::kGroupFloats, 0, A);
                    else issue_group<false>(row + (g + 2) * RM::kGroupFloats, tc, A);
                }
                if (g + 1 < ng) {
                    if (g + 1 < nf) consume_group<true>(qs + (g + 1) * RM::kGroupFloats, 0, 0, 0, B, a0);
                    else consume_group<false>(qs + (g + 1) * RM::kGroupFloats, tc, (g + 1) * RM::kGroupFloats, c.d4, B, a0);
                    if (g + 3 < ng) {
                        if (g + 3 < nf) issue_group<true>(row + (g + 3) * RM::kGroupFloats, 0, B);
                        else issue_group<false>(row + (g + 3) * RM::kGroupFloats, tc, B);
                    }
                }
            }
        } else {
            float2 A[16], B[16];
            const int p = lane & 1;
            const float* rp = row + 2 * p;
            const float* qs = c.qs + 2 * p;
            if (0 < nf) issue_group<true>(rp, 0, A); else issue_group<false>(rp, tc, A);
            if (ng > 1) { if (1 < nf) issue_group<true>(rp + RM::kGroupFloats, 0, B); else issue_group<false>(rp + RM::kGroupFloats, tc, B); }
            for (uint32_t g = 0; g < ng; g += 2) {
                if (g < nf) consume_group<true>(qs + g * RM::kGroupFloats, 0, A, a0, a1);
                else consume_group<false>(qs + g * RM::kGroupFloats, tc, A, a0, a1);
                if (g + 2 < ng) {
                    if (g + 2 < nf) issue_group<true>(rp + (g + 2) * RM::kGroupFloats, 0, A);
                    else issue_group<false>(rp + (g + 2) * RM::kGroupFloats, tc, A);
                }
                if (g + 1 < ng) {
                    if (g + 1 < nf) consume_group<true>(qs + (g + 1) * RM::kGroupFloats, 0, B, a0, a1);
                    else consume_group<false>(qs + (g + 1) * RM::kGroupFloats, tc, B, a0, a1);
                    if (g + 3 < ng) {
                        if (g + 3 < nf) issue_group<true>(rp + (g + 3) * RM::kGroupFloats, 0, B);
                        else issue_group<false>(rp + (g + 3) * RM::kGroupFloats, tc, B);
                    }
                }
            }
        }
    }
    DIAG_STAMP(d1);
    DIAG_ADD(c.cy_chain, d0, d1);
    if (SCALAR) return a0;
    // sums 0,1 are in the even lane of the pair, sums 2,3 in the odd one: ((s0 + s1) + s2) + s3
    const float s2 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(a0), 0xF5, 0xF, 0xF, false));   // quad_perm [1,1,3,3]
    const float s3 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(a1), 0xF5, 0xF, 0xF, false));
    return ((a0 + a1) + s2) + s3;
}

// WR = 0: the result list W is the libstdc++ max-heap in LDS, maintained exactly as the
//         reference does (so even the choice among several equally distant worst entries
//         is the reference's).
// WR > 0: W is an unordered set, WR entries per lane in registers (ef <= 64*WR).  Only the
//         largest distance f and the size of W steer the search, and both evolve
//         identically; the one observable difference is which of several entries with
//         exactly the worst distance is dropped.
template <bool SCALAR, int WR>
__global__ __launch_bounds__(64) void hnsw_search_kernel(SearchArgs a)
{
    typedef RowMap<SCALAR> RM;
    extern __shared__ __align__(16) uint32_t smem[];
    const int lane = threadIdx.x;
    const uint32_t r = (uint32_t)lane / RM::kLanesPerRow;              // row of the batch this lane works on
    const bool owner = (lane % RM::kLanesPerRow) == 0;                // the lane that holds the row's id and distance
    uint32_t qi = blockIdx.x;
    if (a.qmap) qi = a.qmap[qi];

    // LDS: query | W (exact-heap mode only) | C | visited.  The register mode needs W in LDS only for the
    // final sort, when C is dead, so it borrows C's space.
    float* qs = reinterpret_cast<float*>(smem);
    uint2* W = reinterpret_cast<uint2*>(smem + a.qpad);
    uint2* C = W + (WR == 0 ? a.cap_w : 0u);
    uint32_t* hash = reinterpret_cast<uint32_t*>(C + a.cap_c);
    const uint32_t H = 1u << a.hash_bits, hmask = H - 1, hshift = 32 - a.hash_bits;
    const uint32_t visit_limit = H - (H >> 3);       // keep the table at most 7/8 full
    const float* const vectors = a.g.vectors;
    const uint32_t row_stride = a.g.row_stride;
    const DistCtx dc = make_dist_ctx<SCALAR>(qs, a.g.d);
    const bool float_keys = (a.flags & HNSW_AMD_KEYS_FLOAT) != 0;
    const uint32_t ef = a.ef;

    // query into LDS (zero padded), visited set empty
    {
        const float* q = a.queries + (size_t)qi * a.g.d;
        for (uint32_t i = lane; i < a.qpad; i += 64) qs[i] = i < a.g.d ? q[i] : 0.0f;
        for (uint32_t i = lane; i < H; i += 64) hash[i] = kEmpty;
    }
    LDS_SYNC();

    hnsw_amd_query_stats st = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t fail = 0;
#ifdef HNSW_DIAG
    uint32_t cy_pop = 0, cy_adj = 0, cy_dist = 0, cy_adm = 0, cy_up = 0, cy_fin = 0;
#endif
    DIAG_STAMP(t_begin);

    // ---- entry point of the top layer (src/hnsw.cpp:385-398) ----
    const int L = (int)a.g.nlayers - 1;
    uint32_t cur_slot = a.g.top_ep_slot, cur_id = a.g.top_ep_order;
    float cur_dist;
    if (lane == 0) visited_insert(hash, hmask, hshift, cur_id);
    {
        const float dj = row_distance<SCALAR>(dc, vectors + (size_t)cur_id * row_stride, r == 0, lane);
        cur_dist = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dj), 0));
    }
    st.n_dist = 1;
    uint32_t nvis = 1;
    LDS_SYNC();

    // ---- greedy descent through layers L..1; visited shared by all of them
    //      (src/hnsw.cpp:403-412, 676-717) ----
    for (int l = L; l > 0 && !fail; l--) {
        const uint32_t base = a.g.up_off[l];
        bool change = true;
        while (change) {
            change = false;
            if (nvis + 64 > visit_limit) { fail = 1; break; }
            st.n_sweeps++;
            // The sweep covers the whole list of the node current at its start; the running best moves
            // only on a strict improvement, i.e. to the first minimum of the list.
            const size_t rowi = (size_t)(base + cur_slot) * a.g.degu;
            for (uint32_t b0 = 0; b0 < a.g.degu; b0 += RM::kRowsPerBatch) {
                uint32_t my_order = kEmpty, my_slot = kEmpty;
                if (b0 + r < a.g.degu) { my_order = a.g.up_adj_order[rowi + b0 + r]; my_slot = a.g.up_adj_slot[rowi + b0 + r]; }
                const bool live = my_order != kEmpty;
                bool fresh = live && owner && visited_insert(hash, hmask, hshift, my_order);
                if (RM::kLanesPerRow == 2) fresh = __builtin_amdgcn_update_dpp(0, (int)fresh, 0xA0, 0xF, 0xF, false) != 0;   // quad_perm [0,0,2,2]
                const uint32_t m = (uint32_t)__popcll(__ballot(fresh && owner));
                st.sum_degu += (uint32_t)__popcll(__ballot(live && owner));
                st.n_dist += m; nvis += m;
                LDS_SYNC();
                if (m == 0) continue;
                const float dj = row_distance<SCALAR>(dc, vectors + (size_t)my_order * row_stride, fresh, lane);
                const bool mine = fresh && owner && dj == dj;
                const uint32_t key = mine ? __float_as_uint(dj) : 0x7F800000u;     // non-negative floats order as their bit patterns
                const uint32_t kmin = wave_min_u32(key);
                const float bd = __uint_as_float(kmin);
                if (bd < cur_dist) {
                    const unsigned long long who = __ballot(mine && key == kmin);
                    const int j = __builtin_ctzll(who);
                    change = true;
                    cur_dist = bd;
                    cur_id = __builtin_amdgcn_readlane(my_order, j);
                    cur_slot = __builtin_amdgcn_readlane(my_slot, j);
                }
            }
        }
        if (!fail) cur_slot = a.g.up_down[base + cur_slot];      // changeLayer, src/hnsw.h:575-587
    }
    st.n_visited_up = nvis;
    DIAG_STAMP(t_up_end);
    DIAG_ADD(cy_up, t_begin, t_up_end);

    // ---- layer 0 beam search (src/hnsw.cpp:413-415, 725-787) ----
    uint32_t nW = 0, nC = 0;
    float f = 0.0f;
    uint32_t wd[WR > 0 ? WR : 1], wi[WR > 0 ? WR : 1];     // register form of W: distance bits, id
    if (!fail) {
        LDS_SYNC();
        for (uint32_t i = lane; i < H; i += 64) hash[i] = kEmpty;
        LDS_SYNC();
        if (lane == 0) {
            if (WR == 0) W[0] = make_uint2(__float_as_uint(cur_dist), cur_id);
            C[0] = make_uint2(float_keys ? __float_as_uint(cur_dist) : (uint32_t)trunc_key(cur_dist), cur_id);
            visited_insert(hash, hmask, hshift, cur_id);
        }
#pragma unroll
        for (int w = 0; w < (WR > 0 ? WR : 1); w++) { wd[w] = 0; wi[w] = 0; }
        if (WR > 0 && lane == 0) { wd[0] = __float_as_uint(cur_dist); wi[0] = cur_id; }
        nW = 1; nC = 1; nvis = 1; f = cur_dist; st.max_c = 1;
        LDS_SYNC();
        while (nC > 0 && !fail) {
            DIAG_STAMP(t0);
            uint32_t ckey, cid;
            uint32_t my = kEmpty;
            {
                const uint2 c = C[0];
                ckey = rfl(c.x); cid = rfl(c.y);
                // the candidate's neighbour list is requested now and arrives while the heap is repaired
                if (r < a.g.deg0) my = a.g.adj0[(size_t)cid * a.g.deg0 + r];
                if (nC > 1) {
                    if (nC <= kWaveHeapMax) heap_pop_wave<COrder>(C, nC, lane);
                    else if (lane == 0) c_pop(C, nC);
                }
                nC--;
                LDS_SYNC();
            }
            DIAG_STAMP(t1);
            DIAG_ADD(cy_pop, t0, t1);
            if (float_keys) { if (__uint_as_float(ckey) > f) break; }
            else { if ((float)(int32_t)ckey > f) break; }
            st.n_expand++;
            for (uint32_t b0 = 0;;) {
                DIAG_STAMP(t1b);
                if (nvis + 64 > visit_limit) { fail = 1; break; }
                const bool live = my != kEmpty;
                const uint32_t nlive = (uint32_t)__popcll(__ballot(live && owner));
                bool fresh = live && owner && visited_insert(hash, hmask, hshift, my);
                if (RM::kLanesPerRow == 2) fresh = __builtin_amdgcn_update_dpp(0, (int)fresh, 0xA0, 0xF, 0xF, false) != 0;   // quad_perm [0,0,2,2]
                const uint32_t m = (uint32_t)__popcll(__ballot(fresh && owner));
                st.sum_deg0 += nlive; st.n_dist += m; nvis += m;
                LDS_SYNC();
                DIAG_STAMP(t2);
                DIAG_ADD(cy_adj, t1b, t2);
                if (m != 0) {
                    const float dj = row_distance<SCALAR>(dc, vectors + (size_t)my * row_stride, fresh, lane);
                    DIAG_STAMP(t3);
                    DIAG_ADD(cy_dist, t2, t3);
                    // Admission in list order (src/hnsw.cpp:765-777).  A neighbour that fails
                    // `dist < f` against the f at the start of the expansion can only be admitted
                    // while W is not full; f never grows once W is full, so skipping the others is exact.
                    unsigned long long todo = __ballot(fresh && owner && (dj < f || nW < ef));
                    if (nC + (uint32_t)__popcll(todo) > a.cap_c) { fail = 2; break; }
                    while (todo) {
                        const int j = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        const float dist = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dj), j));
                        if (dist < f || nW < ef) {
                            const uint32_t id = __builtin_amdgcn_readlane(my, j);
                            const uint2 ce = make_uint2(float_keys ? __float_as_uint(dist) : (uint32_t)trunc_key(dist), id);
                            const uint2 we = make_uint2(__float_as_uint(dist), id);
                            nC++;
                            if (WR > 0) {
                                if (nC <= kWaveHeapMax) heap_push_wave<COrder>(C, nC, ce, lane);
                                else { if (lane == 0) { C[nC - 1] = ce; c_push(C, nC); } }
                                if (nW < ef) {
                                    // W not full: append; f stays the largest distance in W
#pragma unroll
                                    for (int w = 0; w < WR; w++)
                                        if ((nW >> 6) == (uint32_t)w && (nW & 63u) == (uint32_t)lane) { wd[w] = we.x; wi[w] = we.y; }
                                    nW++;
                                    f = dist > f ? dist : f;
                                } else {
                                    // W full and dist < f: the new entry replaces one entry at distance f
                                    const uint32_t fb = __float_as_uint(f);
                                    bool done = false;
                                    uint32_t mx = 0;
#pragma unroll
                                    for (int w = 0; w < WR; w++) {
                                        const unsigned long long hit = done ? 0ull : __ballot(wd[w] == fb);
                                        if (hit) { if (lane == __builtin_ctzll(hit)) { wd[w] = we.x; wi[w] = we.y; } done = true; }
                                        mx = wd[w] > mx ? wd[w] : mx;
                                    }
                                    f = __uint_as_float(wave_max_u32(mx));   // non-negative floats order as their bit patterns
                                }
                                LDS_SYNC();
                            } else {
                                nW++;
                                if (nC <= kWaveHeapMax && nW <= kWaveHeapMax) {
                                    heap_push_wave<COrder>(C, nC, ce, lane);
                                    heap_push_wave<WOrder>(W, nW, we, lane);
                                    LDS_SYNC();
                                    if (nW > ef) {
                                        const uint2 root = heap_pop_wave<WOrder>(W, nW, lane);
                                        nW--;
                                        f = __uint_as_float(root.x);
                                        LDS_SYNC();
                                    } else {
                                        f = __uint_as_float(W[0].x);
                                    }
                                } else {
                                    if (lane == 0) {
                                        C[nC - 1] = ce; c_push(C, nC);
                                        W[nW - 1] = we; w_push(W, nW);
                                        if (nW > ef) w_pop(W, nW);
                                    }
                                    if (nW > ef) nW--;
                                    LDS_SYNC();
                                    f = __uint_as_float(W[0].x);
                                }
                            }
                            if (nC > st.max_c) st.max_c = nC;
                        }
                    }
                    nW = rfl(nW); nC = rfl(nC); f = rflf(f); st.max_c = rfl(st.max_c);
                    LDS_SYNC();
                    DIAG_STAMP(t4);
                    DIAG_ADD(cy_adm, t3, t4);
                }
                // the live entries of a list are a prefix: a batch with a gap was the last one
                b0 += RM::kRowsPerBatch;
                if (b0 >= a.g.deg0 || nlive < (uint32_t)RM::kRowsPerBatch) break;
                my = b0 + r < a.g.deg0 ? a.g.adj0[(size_t)cid * a.g.deg0 + b0 + r] : kEmpty;
            }
        }
    }
    st.n_visited0 = nvis;
    DIAG_STAMP(t_l0_end);

    if (fail) {
        if (lane == 0) {
            uint32_t at = atomicAdd(a.fail_count, 1u);
            a.fail_list[at] = qi;
            if (a.out_count) a.out_count[qi] = kEmpty;     // marks the row as not produced by this launch
        }
        return;
    }

    // ---- sort W ascending by (distance, id) (src/hnsw.cpp:447) and write the first out_k ----
    LDS_SYNC();
    if (WR > 0) {
#pragma unroll
        for (int w = 0; w < WR; w++)
            if ((uint32_t)(w * 64 + lane) < nW) W[w * 64 + lane] = make_uint2(wd[w], wi[w]);
        LDS_SYNC();
    }
    for (uint32_t i = lane; i < nW; i += 64) {
        const uint2 me = W[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nW; j++) {
            const uint2 o = W[j];
            rank += (o.x < me.x || (o.x == me.x && o.y < me.y)) ? 1u : 0u;   // non-negative floats order as their bits
        }
        if (rank < a.out_k) {
            a.out_ids[(size_t)qi * a.out_k + rank] = me.y;
            a.out_dist[(size_t)qi * a.out_k + rank] = __uint_as_float(me.x);
        }
    }
    for (uint32_t i = nW + lane; i < a.out_k; i += 64) {
        a.out_ids[(size_t)qi * a.out_k + i] = kEmpty;
        a.out_dist[(size_t)qi * a.out_k + i] = __builtin_inff();
    }
#ifdef HNSW_DIAG
    {
        DIAG_STAMP(t_end);
        DIAG_ADD(cy_fin, t_l0_end, t_end);
        st.max_c = st.n_expand;
        st.n_dist = dc.cy_chain; st.n_expand = cy_adj; st.n_sweeps = cy_dist; st.sum_deg0 = cy_adm;
        st.sum_degu = cy_pop; st.n_visited0 = cy_fin; st.n_visited_up = (uint32_t)(t_end - t_begin);
        (void)cy_up;
    }
#endif
    if (lane == 0) {
        if (a.out_count) a.out_count[qi] = nW;
        if (a.stats) a.stats[qi] = st;
    }
}

}  // namespace

size_t search_lds_bytes(const SearchArgs& a, bool w_in_lds)
{
    size_t dwords = (size_t)a.qpad + (w_in_lds ? 2 * (size_t)a.cap_w : 0) + 2 * (size_t)a.cap_c + ((size_t)1 << a.hash_bits);
    return dwords * 4;
}

// ---------------------------------------------------------------------------------------
DeviceIndex::~DeviceIndex() { release(); }

void DeviceIndex::release()
{
    if (device >= 0) (void)hipSetDevice(device);
    for (auto& b : bufs_) { if (b) (void)hipFree(b); b = nullptr; }
    if (d_fail) (void)hipFree(d_fail);
    if (h_fail) (void)hipHostFree(h_fail);
    d_fail = nullptr; h_fail = nullptr; fail_cap = 0;
    version = ~0ull;
}

template <class T>
static T* to_device(const std::vector<T>& v, void*& keep)
{
    T* p = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), bytes));
    keep = p;
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

void DeviceIndex::upload(const HostIndex& h, int dev)
{
    release();
    HIP_CHECK(hipSetDevice(dev));
    device = dev;
    FlatGraph fg;
    h.export_flat(fg);
    if (fg.nlayers == 0 || fg.n == 0) throw std::invalid_argument("search: the index is empty");
    if (fg.deg0 > 64 || fg.degu > 64) throw std::invalid_argument("search: neighbour lists longer than 64 are not supported");
    g = DeviceGraph{};
    g.n = fg.n; g.d = fg.d; g.nlayers = fg.nlayers; g.deg0 = fg.deg0; g.degu = fg.degu;
    g.row_stride = (fg.d + 31u) & ~31u;
    g.top_ep_slot = fg.ep_slot[fg.nlayers - 1]; g.top_ep_order = fg.ep_order[fg.nlayers - 1];
    // base vectors: row-major, every row padded with zeros to whole 128-byte lines
    {
        float* dv = nullptr;
        const size_t bytes = (size_t)g.n * g.row_stride * sizeof(float);
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dv), bytes));
        bufs_[0] = dv;
        if (g.row_stride == g.d) {
            HIP_CHECK(hipMemcpy(dv, h.vector_data_.data(), bytes, hipMemcpyHostToDevice));
        } else {
            HIP_CHECK(hipMemset(dv, 0, bytes));
            HIP_CHECK(hipMemcpy2D(dv, (size_t)g.row_stride * 4, h.vector_data_.data(), (size_t)g.d * 4, (size_t)g.d * 4, g.n,
                                  hipMemcpyHostToDevice));
        }
        g.vectors = dv;
    }
    g.adj0 = to_device(fg.adj0, bufs_[1]);
    g.up_off = to_device(fg.up_off, bufs_[2]);
    g.up_down = to_device(fg.up_down, bufs_[3]);
    g.up_adj_slot = to_device(fg.up_adj_slot, bufs_[4]);
    g.up_adj_order = to_device(fg.up_adj_order, bufs_[5]);
    HIP_CHECK(hipDeviceSynchronize());
    version = h.version_;
    hint_tier = 0; hint_ef = 0; hint_flags = 0;
}

static ScratchTier tier_for(int ef, int tier)
{
    // Sized from measured search statistics (visited ~ 16-25 x ef on graphs of 10^4..10^6
    // nodes, candidate heap <= ~6 x ef on integer data); a query that outgrows a tier is
    // re-run on the next one.
    uint32_t want_visit = (uint32_t)ef * 28u + 256u;
    uint32_t bits = 10;
    while ((1u << bits) - (1u << (bits - 3)) < want_visit + 64) bits++;
    uint32_t cap_c = ((uint32_t)ef * 4u + 63u) & ~63u;
    bits += (uint32_t)tier;
    cap_c <<= tier;
    return ScratchTier{bits, cap_c};
}

uint32_t DeviceIndex::search(const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t flags, uint32_t* d_ids,
                             float* d_dist, uint32_t* d_count, hnsw_amd_query_stats* d_stats, void* stream_v, bool sync_and_retry)
{
    if (device < 0) throw std::logic_error("search: index not on a device");
    if (nq == 0) return 0;
    if (ef < 1 || out_k < 1 || out_k > (uint32_t)ef) throw std::invalid_argument("search: need 1 <= out_k <= ef");
    HIP_CHECK(hipSetDevice(device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (fail_cap < nq + 1) {
        if (d_fail) HIP_CHECK(hipFree(d_fail));
        d_fail = nullptr;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_fail), ((size_t)nq + 1) * 4 * 2));
        fail_cap = nq + 1;
        if (!h_fail) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_fail), 64, hipHostMallocDefault));
    }
    uint32_t* d_fail_alt = d_fail + fail_cap;   // second list: a retry launch reads one and appends to the other

    SearchArgs a{};
    a.g = g; a.queries = d_queries; a.qmap = nullptr; a.nq_launch = nq;
    a.ef = (uint32_t)ef; a.out_k = out_k; a.flags = flags;
    a.cap_w = ((uint32_t)ef + 2u) & ~1u;
    a.qpad = (g.d + 63u) & ~63u;
    a.out_ids = d_ids; a.out_dist = d_dist; a.out_count = d_count; a.stats = d_stats;
    const bool scalar = (flags & HNSW_AMD_DIST_SCALAR) != 0;

    int max_lds = 0;
    HIP_CHECK(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device));

    uint32_t retried = 0;
    uint32_t* cur = d_fail;
    uint32_t* other = d_fail_alt;
    // Start where most queries of the previous batch at a comparable ef ended up: on data where
    // the search visits far more nodes than the tier-0 estimate (unit-norm vectors with the
    // reference's int32 keys), running every query twice would double the work.
    const int first_tier = (hint_tier > 0 && 4 * ef >= 3 * hint_ef && flags == hint_flags) ? hint_tier : 0;
    int crowded_tier = first_tier;
    for (int tier = first_tier;; tier++) {
        ScratchTier t = tier_for(ef, tier);
        a.hash_bits = t.hash_bits; a.cap_c = t.cap_c;
        const bool std_heap = (flags & HNSW_AMD_W_STD_HEAP) != 0 || ef > 256;
        if (!std_heap && a.cap_w > a.cap_c) throw std::logic_error("search: W does not fit the space it borrows for the final sort");
        size_t lds = search_lds_bytes(a, std_heap);
        if (lds > (size_t)max_lds) {
            throw std::length_error("search: " + std::to_string(a.nq_launch) +
                                    " queries outgrew the largest on-chip scratch configuration (ef too large for LDS?)");
        }
        a.fail_count = cur; a.fail_list = cur + 1;
        HIP_CHECK(hipMemsetAsync(cur, 0, 4, stream));
        void (*kern)(SearchArgs);
        if (scalar) kern = std_heap ? hnsw_search_kernel<true, 0> : (ef <= 64 ? hnsw_search_kernel<true, 1> : (ef <= 128 ? hnsw_search_kernel<true, 2> : hnsw_search_kernel<true, 4>));
        else kern = std_heap ? hnsw_search_kernel<false, 0> : (ef <= 64 ? hnsw_search_kernel<false, 1> : (ef <= 128 ? hnsw_search_kernel<false, 2> : hnsw_search_kernel<false, 4>));
        if (lds > 32768)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (const char* pad = getenv("HNSW_AMD_LDS_PAD")) lds += (size_t)atoi(pad);      // occupancy experiments
        if (getenv("HNSW_AMD_VERBOSE")) {
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 64, lds);
            fprintf(stderr, "[hnsw_amd] search launch: %u queries, ef=%d, tier %d (hash 2^%u, C %u), LDS %zu B, resident queries per CU %d\n",
                    a.nq_launch, ef, tier, a.hash_bits, a.cap_c, lds, nb);
        }
        hipLaunchKernelGGL(kern, dim3(a.nq_launch), dim3(64), lds, stream, a);
        HIP_CHECK(hipGetLastError());
        if (!sync_and_retry) return 0;
        HIP_CHECK(hipMemcpyAsync(h_fail, cur, 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        uint32_t nfail = h_fail[0];
        if (8 * (size_t)nfail >= nq) crowded_tier = tier + 1;      // an eighth of the batch or more needs the next tier
        if (nfail == 0) break;
        retried += nfail;
        a.qmap = cur + 1; a.nq_launch = nfail;
        std::swap(cur, other);
    }
    if (crowded_tier != first_tier || first_tier == 0) { hint_tier = crowded_tier; hint_ef = ef; hint_flags = flags; }
    return retried;
}

}  // namespace hnsw_amd
