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
constexpr int kTileStride = 68;            // dwords per row of the transposition tile (64 + 4: conflict-free chain reads)
constexpr int kTileDwords = 16 * kTileStride;
constexpr int kSeg = 64;                   // floats per segment (16 lanes x float4)

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

struct WaveCtx {
    float* tile;          // kTileDwords
    float* qs;            // qpad floats, zero padded
    uint32_t* nb_id;      // 64
    float* nb_dist;       // 64
    uint32_t* nb_slot;    // 64
    const float* vectors;
    uint32_t row_stride;
    uint32_t table_bytes; // size of the vector table if it is below 4 GiB (32-bit offsets), else 0
    uint32_t d_eff;       // dims the distance uses
    uint32_t nseg;        // ceil(d_eff / 64)
    uint32_t d4;          // scalar mode: dims whose square is rounded before the add
    int lane;
#ifdef HNSW_DIAG
    mutable uint32_t cy_load = 0, cy_chain = 0, cy_fin = 0;
#endif
};

struct RowRegs { float4 r[4][2]; };    // one lane's share of 16 rows x 2 segments (128 dims)

// Issue the loads of rows nb_id[b0 .. b0+16) for segments s0, s0+1: lane (h, c) takes the
// 16-byte chunk c of row 4g+h for g = 0..3, so one instruction reads 256 contiguous bytes of
// each of 4 rows.
__device__ __forceinline__ void load_rows(const WaveCtx& w, uint32_t m, uint32_t b0, uint32_t s0, RowRegs& R)
{
    const int h = w.lane >> 4, c = w.lane & 15;
    if (w.table_bytes != 0) {
        // The whole table is addressable with a 32-bit byte offset: buffer loads.  A lane with
        // nothing to fetch (row index past m, chunk past the used dims) gets an offset beyond
        // the table; the hardware range check returns zeros for it without touching memory,
        // so there is no branching around the loads.
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.vectors), 0, (int)w.table_bytes, 0x00020000);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t r = b0 + 4 * g + h;
            const uint32_t id = w.nb_id[r & 63u];
            const uint32_t row_off = id * (w.row_stride * 4u) + (uint32_t)c * 16u;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const uint32_t e0 = (s0 + s) * kSeg + c * 4;
                const uint32_t off = (r < m && e0 < w.d_eff) ? row_off + (s0 + s) * (kSeg * 4u) : 0xFFFFFFFFu;
                const u4v v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
                R.r[g][s] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t r = b0 + 4 * g + h;
        const bool ok = r < m;
        const uint32_t id = ok ? w.nb_id[r] : 0u;
        const float* rp = w.vectors + (size_t)id * w.row_stride + c * 4;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const uint32_t e0 = (s0 + s) * kSeg + c * 4;
            R.r[g][s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && e0 < w.d_eff) R.r[g][s] = *reinterpret_cast<const float4*>(rp + (s0 + s) * kSeg);
        }
    }
}

// Differences to the query -> transposition tile -> the per-(row, lane-sum) chains, for the
// segments s0, s0+1 held in R.  Returns the running sum of this lane's chain.
// SCALAR=false: the reference's SSE loop order (4 lane sums, fma, left-to-right final add).
// SCALAR=true : the reference's scalar loop as its -DNO_MANUAL_VECTORIZATION build runs it:
//               one running sum; squares rounded before the add for the first 4*(d/4)
//               dims, fused multiply-add for the last d%4.
template <bool SCALAR>
__device__ __forceinline__ float chain_rows(const WaveCtx& w, uint32_t s0, const RowRegs& R, float acc)
{
    const int h = w.lane >> 4, c = w.lane & 15;     // loader role
    const int cr = w.lane >> 2, cj = w.lane & 3;    // chain role: row cr of the batch, lane-sum cj
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (s0 + s < w.nseg) {
            const uint32_t seg = s0 + s;
            const uint32_t e0 = seg * kSeg + c * 4;
            const float4 qv = *reinterpret_cast<const float4*>(w.qs + e0);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 v = R.r[g][s];
                float4 t;
                t.x = v.x - qv.x; t.y = v.y - qv.y; t.z = v.z - qv.z; t.w = v.w - qv.w;
                if (SCALAR && e0 < w.d4) { t.x = t.x * t.x; t.y = t.y * t.y; t.z = t.z * t.z; t.w = t.w * t.w; }
                *reinterpret_cast<float4*>(w.tile + (4 * g + h) * kTileStride + c * 4) = t;
            }
            LDS_SYNC();
            const uint32_t left = w.d_eff - seg * kSeg;          // dims of this segment
            if (!SCALAR) {
                const uint32_t steps = left >= kSeg ? 16u : (left >> 2);
                const float* tp = w.tile + cr * kTileStride + cj;
                if (steps == 16) {
#pragma unroll
                    for (int t = 0; t < 16; t++) { float x = tp[4 * t]; acc = __builtin_fmaf(x, x, acc); }
                } else {
                    for (uint32_t t = 0; t < steps; t++) { float x = tp[4 * t]; acc = __builtin_fmaf(x, x, acc); }
                }
            } else if (cj == 0) {
                const uint32_t steps = left >= kSeg ? (uint32_t)kSeg : left;
                const float* tp = w.tile + cr * kTileStride;
                for (uint32_t t = 0; t < steps; t++) {
                    float x = tp[t];
                    if (seg * kSeg + t < w.d4) acc = acc + x; else acc = __builtin_fmaf(x, x, acc);
                }
            }
            LDS_SYNC();
        }
    }
    return acc;
}

template <bool SCALAR>
__device__ __forceinline__ void finish_rows(const WaveCtx& w, uint32_t m, uint32_t b0, float acc)
{
    const int cr = w.lane >> 2, cj = w.lane & 3;
    float res = acc;
    if (!SCALAR) {
        // lane sums 0..3 of a row sit in 4 adjacent lanes: ((s0 + s1) + s2) + s3
        // (quad broadcasts by DPP: no LDS round trip)
        const int ai = (int)__float_as_uint(acc);
        const float a0 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0x00, 0xF, 0xF, false));
        const float a1 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0x55, 0xF, 0xF, false));
        const float a2 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0xAA, 0xF, 0xF, false));
        const float a3 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0xFF, 0xF, 0xF, false));
        res = ((a0 + a1) + a2) + a3;
    }
    if (cj == 0 && b0 + cr < m) w.nb_dist[b0 + cr] = res;
}

// Distances from the query to the rows nb_id[0..m) -> nb_dist[0..m).  m <= 64, wave-uniform.
// Rows of up to 128 used dims: the loads of 32 rows (a whole layer-0 list at M=16) are in
// flight together before the first is consumed.  Wider rows go 16 rows x 128 dims at a time.
template <bool SCALAR>
__device__ __forceinline__ void compute_distances(const WaveCtx& w, uint32_t m)
{
    if (w.nseg <= 2) {
        for (uint32_t b0 = 0; b0 < m; b0 += 32) {
            RowRegs A, B;
            DIAG_STAMP(d0);
            load_rows(w, m, b0, 0, A);
            const bool two = b0 + 16 < m;
            if (two) load_rows(w, m, b0 + 16, 0, B);
#ifdef HNSW_DIAG
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            DIAG_STAMP(d1);
            DIAG_ADD(w.cy_load, d0, d1);
            float accA = chain_rows<SCALAR>(w, 0, A, 0.0f);
            float accB = 0.0f;
            if (two) accB = chain_rows<SCALAR>(w, 0, B, 0.0f);
            DIAG_STAMP(d2);
            DIAG_ADD(w.cy_chain, d1, d2);
            finish_rows<SCALAR>(w, m, b0, accA);
            if (two) finish_rows<SCALAR>(w, m, b0 + 16, accB);
            DIAG_STAMP(d3);
            DIAG_ADD(w.cy_fin, d2, d3);
        }
    } else {
        for (uint32_t b0 = 0; b0 < m; b0 += 16) {
            float acc = 0.0f;
            for (uint32_t s0 = 0; s0 < w.nseg; s0 += 2) {
                RowRegs A;
                load_rows(w, m, b0, s0, A);
                acc = chain_rows<SCALAR>(w, s0, A, acc);
            }
            finish_rows<SCALAR>(w, m, b0, acc);
        }
    }
    LDS_SYNC();
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
    extern __shared__ __align__(16) uint32_t smem[];
    const int lane = threadIdx.x;
    uint32_t qi = blockIdx.x;
    if (a.qmap) qi = a.qmap[qi];

    WaveCtx w;
    w.tile = reinterpret_cast<float*>(smem);
    w.qs = w.tile + kTileDwords;
    w.nb_id = smem + kTileDwords + a.qpad;
    w.nb_dist = reinterpret_cast<float*>(w.nb_id + 64);
    w.nb_slot = w.nb_id + 128;
    // W in LDS: the exact-heap mode keeps it there throughout; the register mode only needs it
    // for the final sort, after the last use of the tile, so it shares the tile's space
    uint2* W = WR == 0 ? reinterpret_cast<uint2*>(w.nb_id + 192) : reinterpret_cast<uint2*>(smem);
    uint2* C = reinterpret_cast<uint2*>(w.nb_id + 192) + (WR == 0 ? a.cap_w : 0u);
    uint32_t* hash = reinterpret_cast<uint32_t*>(C + a.cap_c);
    const uint32_t H = 1u << a.hash_bits, hmask = H - 1, hshift = 32 - a.hash_bits;
    const uint32_t visit_limit = H - (H >> 3);       // keep the table at most 7/8 full
    w.vectors = a.g.vectors; w.row_stride = a.g.row_stride; w.lane = lane;
    {
        const unsigned long long tb = (unsigned long long)a.g.n * a.g.row_stride * 4ull;
        w.table_bytes = tb < 0xFFFFFFF0ull ? (uint32_t)tb : 0u;
    }
    w.d_eff = SCALAR ? a.g.d : (a.g.d & ~15u);
    w.nseg = (w.d_eff + kSeg - 1) / kSeg;
    w.d4 = a.g.d & ~3u;
    const bool float_keys = (a.flags & HNSW_AMD_KEYS_FLOAT) != 0;
    const uint32_t ef = a.ef;

    // query into LDS (zero padded), visited set empty
    {
        const float* q = a.queries + (size_t)qi * a.g.d;
        for (uint32_t i = lane; i < a.qpad; i += 64) w.qs[i] = i < a.g.d ? q[i] : 0.0f;
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
    if (lane == 0) { w.nb_id[0] = cur_id; visited_insert(hash, hmask, hshift, cur_id); }
    LDS_SYNC();
    compute_distances<SCALAR>(w, 1);
    cur_dist = w.nb_dist[0];
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
            const size_t rowi = (size_t)(base + cur_slot) * a.g.degu;
            uint32_t my_order = kEmpty, my_slot = kEmpty;
            if ((uint32_t)lane < a.g.degu) { my_order = a.g.up_adj_order[rowi + lane]; my_slot = a.g.up_adj_slot[rowi + lane]; }
            const bool live = my_order != kEmpty;
            const bool fresh = live && visited_insert(hash, hmask, hshift, my_order);
            const unsigned long long fm = __ballot(fresh);
            const uint32_t m = (uint32_t)__popcll(fm);
            const uint32_t rank = (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
            if (fresh) { w.nb_id[rank] = my_order; w.nb_slot[rank] = my_slot; }
            st.n_sweeps++; st.sum_degu += (uint32_t)__popcll(__ballot(live));
            st.n_dist += m; nvis += m;
            LDS_SYNC();
            if (m == 0) continue;
            compute_distances<SCALAR>(w, m);
            // first minimum among the new distances, taken only if it improves (strict <)
            float bd = (uint32_t)lane < m ? w.nb_dist[lane] : __builtin_inff();
            if (!(bd == bd)) bd = __builtin_inff();
            uint32_t bj = lane;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                float od = __shfl_xor(bd, off);
                uint32_t oj = __shfl_xor(bj, off);
                if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
            }
            bd = rflf(bd); bj = rfl(bj);
            if (bd < cur_dist) {
                change = true;
                cur_dist = bd; cur_id = w.nb_id[bj]; cur_slot = w.nb_slot[bj];
            }
            LDS_SYNC();
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
        for (int r = 0; r < (WR > 0 ? WR : 1); r++) { wd[r] = 0; wi[r] = 0; }
        if (WR > 0 && lane == 0) { wd[0] = __float_as_uint(cur_dist); wi[0] = cur_id; }
        nW = 1; nC = 1; nvis = 1; f = cur_dist; st.max_c = 1;
        LDS_SYNC();
        while (nC > 0) {
            DIAG_STAMP(t0);
            uint32_t ckey, cid;
            uint32_t my = kEmpty;
            {
                const uint2 c = C[0];
                ckey = rfl(c.x); cid = rfl(c.y);
                // the candidate's neighbour list is requested now and arrives while the heap is repaired
                if ((uint32_t)lane < a.g.deg0) my = a.g.adj0[(size_t)cid * a.g.deg0 + lane];
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
            if (nvis + 64 > visit_limit) { fail = 1; break; }
            const bool live = my != kEmpty;
            const bool fresh = live && visited_insert(hash, hmask, hshift, my);
            const unsigned long long fm = __ballot(fresh);
            const uint32_t m = (uint32_t)__popcll(fm);
            const uint32_t rank = (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
            if (fresh) w.nb_id[rank] = my;
            st.n_expand++; st.sum_deg0 += (uint32_t)__popcll(__ballot(live));
            st.n_dist += m; nvis += m;
            LDS_SYNC();
            DIAG_STAMP(t2);
            DIAG_ADD(cy_adj, t1, t2);
            if (m == 0) continue;
            compute_distances<SCALAR>(w, m);
            DIAG_STAMP(t3);
            DIAG_ADD(cy_dist, t2, t3);
            // Admission in list order (src/hnsw.cpp:765-777).  A neighbour that fails
            // `dist < f` against the f at the start of the expansion can only be admitted
            // while W is not full; f never grows once W is full, so skipping the others is exact.
            const float dj = (uint32_t)lane < m ? w.nb_dist[lane] : 0.0f;
            const uint32_t idj = (uint32_t)lane < m ? w.nb_id[lane] : 0u;
            unsigned long long todo = __ballot((uint32_t)lane < m && (dj < f || nW < ef));
            if (nC + (uint32_t)__popcll(todo) > a.cap_c) { fail = 2; break; }
            while (todo) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1;
                const float dist = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dj), j));
                if (dist < f || nW < ef) {
                    const uint32_t id = __builtin_amdgcn_readlane(idj, j);
                    const uint2 ce = make_uint2(float_keys ? __float_as_uint(dist) : (uint32_t)trunc_key(dist), id);
                    const uint2 we = make_uint2(__float_as_uint(dist), id);
                    nC++;
                    if (WR > 0) {
                        if (nC <= kWaveHeapMax) heap_push_wave<COrder>(C, nC, ce, lane);
                        else { if (lane == 0) { C[nC - 1] = ce; c_push(C, nC); } }
                        if (nW < ef) {
                            // W not full: append; f stays the largest distance in W
#pragma unroll
                            for (int r = 0; r < WR; r++)
                                if ((nW >> 6) == (uint32_t)r && (nW & 63u) == (uint32_t)lane) { wd[r] = we.x; wi[r] = we.y; }
                            nW++;
                            f = dist > f ? dist : f;
                        } else {
                            // W full and dist < f: the new entry replaces one entry at distance f
                            const uint32_t fb = __float_as_uint(f);
                            bool done = false;
                            uint32_t mx = 0;
#pragma unroll
                            for (int r = 0; r < WR; r++) {
                                const unsigned long long hit = done ? 0ull : __ballot(wd[r] == fb);
                                if (hit) { if (lane == __builtin_ctzll(hit)) { wd[r] = we.x; wi[r] = we.y; } done = true; }
                                mx = wd[r] > mx ? wd[r] : mx;
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
        for (int r = 0; r < WR; r++)
            if ((uint32_t)(r * 64 + lane) < nW) W[r * 64 + lane] = make_uint2(wd[r], wi[r]);
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
        st.n_dist = cy_pop; st.n_expand = cy_adj; st.n_sweeps = cy_dist; st.sum_deg0 = cy_adm;
        st.sum_degu = cy_up; st.n_visited0 = cy_fin; st.n_visited_up = (uint32_t)(t_end - t_begin);
        // finer split of the distance phase, in place of two of the coarse sums
        st.n_dist = w.cy_load; st.sum_degu = w.cy_chain; st.n_visited0 = w.cy_fin;
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
    size_t dwords = (size_t)kTileDwords + a.qpad + 192 + (w_in_lds ? 2 * (size_t)a.cap_w : 0) + 2 * (size_t)a.cap_c +
                    ((size_t)1 << a.hash_bits);
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
        if (!std_heap && 2 * (size_t)a.cap_w > (size_t)kTileDwords) throw std::logic_error("search: W does not fit the tile");
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
