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

struct WaveCtx {
    float* tile;          // kTileDwords
    float* qs;            // qpad floats, zero padded
    uint32_t* nb_id;      // 64
    float* nb_dist;       // 64
    uint32_t* nb_slot;    // 64
    const float* vectors;
    uint32_t row_stride;
    uint32_t d_eff;       // dims the distance uses
    uint32_t nseg;        // ceil(d_eff / 64)
    uint32_t d4;          // scalar mode: dims whose square is rounded before the add
    int lane;
};

// Distances from the query to the rows nb_id[0..m) -> nb_dist[0..m).  m <= 64, wave-uniform.
// SCALAR=false: the reference's SSE loop order (4 lane sums, fma, left-to-right final add).
// SCALAR=true : the reference's scalar loop as its -DNO_MANUAL_VECTORIZATION build runs it:
//               one running sum; squares rounded before the add for the first 4*(d/4)
//               dims, fused multiply-add for the last d%4.
template <bool SCALAR>
__device__ __forceinline__ void compute_distances(const WaveCtx& w, uint32_t m)
{
    const int lane = w.lane;
    const int h = lane >> 4, c = lane & 15;     // loader role: row h of a 4-row group, 16-byte chunk c of a segment
    const int cr = lane >> 2, cj = lane & 3;    // chain role: row cr of the batch, lane-sum cj
    for (uint32_t b0 = 0; b0 < m; b0 += 16) {
        const float* rp[4];
        bool ok[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            uint32_t r = b0 + 4 * g + h;
            ok[g] = r < m;
            uint32_t id = ok[g] ? w.nb_id[r] : 0u;
            rp[g] = w.vectors + (size_t)id * w.row_stride + c * 4;
        }
        float acc = 0.0f;
        for (uint32_t s0 = 0; s0 < w.nseg; s0 += 2) {
            float4 reg[4][2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const uint32_t e0 = (s0 + s) * kSeg + c * 4;
                const bool in = e0 < w.d_eff;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    reg[g][s] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok[g] && in) reg[g][s] = *reinterpret_cast<const float4*>(rp[g] + (s0 + s) * kSeg);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (s0 + s < w.nseg) {
                    const uint32_t seg = s0 + s;
                    const uint32_t e0 = seg * kSeg + c * 4;
                    const float4 qv = *reinterpret_cast<const float4*>(w.qs + e0);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        float4 v = reg[g][s];
                        float4 t;
                        t.x = v.x - qv.x; t.y = v.y - qv.y; t.z = v.z - qv.z; t.w = v.w - qv.w;
                        if (SCALAR && e0 < w.d4) { t.x = t.x * t.x; t.y = t.y * t.y; t.z = t.z * t.z; t.w = t.w * t.w; }
                        *reinterpret_cast<float4*>(w.tile + (4 * g + h) * kTileStride + c * 4) = t;
                    }
                    __syncthreads();
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
                    __syncthreads();
                }
            }
        }
        float res;
        if (!SCALAR) {
            const int q0 = lane & ~3;
            float a0 = __shfl(acc, q0), a1 = __shfl(acc, q0 + 1), a2 = __shfl(acc, q0 + 2), a3 = __shfl(acc, q0 + 3);
            res = ((a0 + a1) + a2) + a3;
        } else {
            res = acc;
        }
        if (cj == 0 && b0 + cr < m) w.nb_dist[b0 + cr] = res;
    }
    __syncthreads();
}

template <bool SCALAR>
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
    uint2* W = reinterpret_cast<uint2*>(w.nb_id + 192);
    uint2* C = W + a.cap_w;
    uint32_t* hash = reinterpret_cast<uint32_t*>(C + a.cap_c);
    const uint32_t H = 1u << a.hash_bits, hmask = H - 1, hshift = 32 - a.hash_bits;
    const uint32_t visit_limit = H - (H >> 3);       // keep the table at most 7/8 full
    w.vectors = a.g.vectors; w.row_stride = a.g.row_stride; w.lane = lane;
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
    __syncthreads();

    hnsw_amd_query_stats st = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t fail = 0;

    // ---- entry point of the top layer (src/hnsw.cpp:385-398) ----
    const int L = (int)a.g.nlayers - 1;
    uint32_t cur_slot = a.g.top_ep_slot, cur_id = a.g.top_ep_order;
    float cur_dist;
    if (lane == 0) { w.nb_id[0] = cur_id; visited_insert(hash, hmask, hshift, cur_id); }
    __syncthreads();
    compute_distances<SCALAR>(w, 1);
    cur_dist = w.nb_dist[0];
    st.n_dist = 1;
    uint32_t nvis = 1;
    __syncthreads();

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
            __syncthreads();
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
            __syncthreads();
        }
        if (!fail) cur_slot = a.g.up_down[base + cur_slot];      // changeLayer, src/hnsw.h:575-587
    }
    st.n_visited_up = nvis;

    // ---- layer 0 beam search (src/hnsw.cpp:413-415, 725-787) ----
    uint32_t nW = 0, nC = 0;
    float f = 0.0f;
    if (!fail) {
        __syncthreads();
        for (uint32_t i = lane; i < H; i += 64) hash[i] = kEmpty;
        __syncthreads();
        if (lane == 0) {
            W[0] = make_uint2(__float_as_uint(cur_dist), cur_id);
            C[0] = make_uint2(float_keys ? __float_as_uint(cur_dist) : (uint32_t)trunc_key(cur_dist), cur_id);
            visited_insert(hash, hmask, hshift, cur_id);
        }
        nW = 1; nC = 1; nvis = 1; f = cur_dist; st.max_c = 1;
        __syncthreads();
        while (nC > 0) {
            uint32_t ckey = 0, cid = 0;
            if (lane == 0) { uint2 c = C[0]; ckey = c.x; cid = c.y; c_pop(C, nC); }
            nC--;
            ckey = rfl(ckey); cid = rfl(cid);
            if (float_keys) { if (__uint_as_float(ckey) > f) break; }
            else { if ((float)(int32_t)ckey > f) break; }
            if (nvis + 64 > visit_limit) { fail = 1; break; }
            uint32_t my = kEmpty;
            if ((uint32_t)lane < a.g.deg0) my = a.g.adj0[(size_t)cid * a.g.deg0 + lane];
            const bool live = my != kEmpty;
            const bool fresh = live && visited_insert(hash, hmask, hshift, my);
            const unsigned long long fm = __ballot(fresh);
            const uint32_t m = (uint32_t)__popcll(fm);
            const uint32_t rank = (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
            if (fresh) w.nb_id[rank] = my;
            st.n_expand++; st.sum_deg0 += (uint32_t)__popcll(__ballot(live));
            st.n_dist += m; nvis += m;
            __syncthreads();
            if (m == 0) continue;
            compute_distances<SCALAR>(w, m);
            // Admission in list order (src/hnsw.cpp:765-777).  A neighbour that fails
            // `dist < f` against the f at the start of the expansion can only be admitted
            // while W is not full; f never grows once W is full, so skipping the others is exact.
            const float dj = (uint32_t)lane < m ? w.nb_dist[lane] : 0.0f;
            unsigned long long todo = __ballot((uint32_t)lane < m && (dj < f || nW < ef));
            if (nC + (uint32_t)__popcll(todo) > a.cap_c) { fail = 2; break; }
            if (lane == 0) {
                while (todo) {
                    const int j = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const float dist = w.nb_dist[j];
                    if (dist < f || nW < ef) {
                        const uint32_t id = w.nb_id[j];
                        C[nC] = make_uint2(float_keys ? __float_as_uint(dist) : (uint32_t)trunc_key(dist), id);
                        nC++;
                        c_push(C, nC);
                        if (nC > st.max_c) st.max_c = nC;
                        W[nW] = make_uint2(__float_as_uint(dist), id);
                        nW++;
                        w_push(W, nW);
                        if (nW > ef) { w_pop(W, nW); nW--; }
                        f = __uint_as_float(W[0].x);
                    }
                }
            }
            nW = rfl(nW); nC = rfl(nC); f = rflf(f); st.max_c = rfl(st.max_c);
            __syncthreads();
        }
    }
    st.n_visited0 = nvis;

    if (fail) {
        if (lane == 0) {
            uint32_t at = atomicAdd(a.fail_count, 1u);
            a.fail_list[at] = qi;
            if (a.out_count) a.out_count[qi] = kEmpty;     // marks the row as not produced by this launch
        }
        return;
    }

    // ---- sort W ascending by (distance, id) (src/hnsw.cpp:447) and write the first out_k ----
    __syncthreads();
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
    if (lane == 0) {
        if (a.out_count) a.out_count[qi] = nW;
        if (a.stats) a.stats[qi] = st;
    }
}

}  // namespace

size_t search_lds_bytes(const SearchArgs& a)
{
    size_t dwords = (size_t)kTileDwords + a.qpad + 192 + 2 * (size_t)a.cap_w + 2 * (size_t)a.cap_c + ((size_t)1 << a.hash_bits);
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
}

static ScratchTier tier_for(int ef, int tier)
{
    // Sized from measured search statistics (visited ~ 16-25 x ef on graphs of 10^4..10^6
    // nodes, candidate heap <= ~6 x ef on integer data); a query that outgrows a tier is
    // re-run on the next one.
    uint32_t want_visit = (uint32_t)ef * 28u + 256u;
    uint32_t bits = 10;
    while ((1u << bits) - (1u << (bits - 3)) < want_visit + 64) bits++;
    uint32_t cap_c = ((uint32_t)ef * 6u + 127u) & ~63u;
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
    for (int tier = 0;; tier++) {
        ScratchTier t = tier_for(ef, tier);
        a.hash_bits = t.hash_bits; a.cap_c = t.cap_c;
        size_t lds = search_lds_bytes(a);
        if (lds > (size_t)max_lds) {
            throw std::length_error("search: " + std::to_string(a.nq_launch) +
                                    " queries outgrew the largest on-chip scratch configuration (ef too large for LDS?)");
        }
        a.fail_count = cur; a.fail_list = cur + 1;
        HIP_CHECK(hipMemsetAsync(cur, 0, 4, stream));
        auto kern = scalar ? hnsw_search_kernel<true> : hnsw_search_kernel<false>;
        if (lds > 32768)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(a.nq_launch), dim3(64), lds, stream, a);
        HIP_CHECK(hipGetLastError());
        if (!sync_and_retry) return 0;
        HIP_CHECK(hipMemcpyAsync(h_fail, cur, 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        uint32_t nfail = h_fail[0];
        if (nfail == 0) break;
        retried += nfail;
        a.qmap = cur + 1; a.nq_launch = nfail;
        std::swap(cur, other);
    }
    return retried;
}

}  // namespace hnsw_amd
