// Layer 0 of a GPU-assisted build kept, selected and linked on the device.
//
// The reference's insert (src/hnsw.cpp:163-334) does three things for a new vector in layer 0: searchLayer with
// ef = efConstruction (:246-262), selectNeighbors among the result (:263, 1078-1126), and linking — the new node's own
// list, then the new node appended to every selected neighbour's list, with selectNeighbors run again over a list that
// is full (:266-312).  Here a sub-batch of new vectors goes through the three steps as three device passes over the
// graph as it stands:
//   1. the batched search kernel (hnsw_kernels.hip) with ef = efConstruction, results left in device memory;
//   2. select_new_kernel: one wavefront per new vector runs the selection rule over its candidate list, writes the
//      vector's own list and emits one (neighbour, new vector, distance) triple per selected neighbour;
//   3. the triples are sorted by neighbour (radix sort), and backlink_kernel gives every neighbour that received
//      links to one wavefront, which appends them in the order of the new vectors' ids and re-selects whenever the
//      list is full — no two wavefronts ever write the same list, so there are no locks and the result does not
//      depend on timing.
// The vectors of one sub-batch do not see each other (as in the host-linked assisted build), upper layers stay on the
// host.  Distances between two stored rows are plain fused-multiply-add sums over the dims the build's distance uses;
// they are not bit-equal to the host's (another summation order), so a comparison that is a near-tie on the host may
// fall the other way here: a graph of the same rule, not the same graph.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "hnsw_device.h"

namespace hnsw_amd {

namespace {

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); \
    } while (0)

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kMaxList = 32;        // longest layer-0 list the device linking handles (one lane per entry, plus the newcomer)

struct Layer0 {
    uint32_t n, d, d_lim, row_stride, deg0;   // d_lim: the dims that take part in a distance (d, or 16 * floor(d / 16))
    const float* vectors;                     // stored rows (hnsw_kernels.hip: stored_pos)
    uint32_t* adj0;                           // n x deg0
    float* dist0;                             // n x deg0: distance stored on the edge (feeds re-selection)
};

// Squared L2 between the row staged in LDS and the stored row `b`, computed by the four lanes of a quad (lane s takes
// every fourth 16-byte piece) and summed over the quad; every lane of the wavefront must call it.  Both rows are in
// the stored order, which permutes the dims inside a 128-byte line only: a sum over all of them needs no un-permuting,
// and the padding of both rows is zero.  Only a build whose distance drops the last d mod 16 dims (src/hnsw.h:340-342)
// has to know which stored positions those are: float 8 j + k of a line is dim 4 k + j.
__device__ __forceinline__ float quad_row_distance(const float* a_lds, const float* __restrict__ b, const Layer0& g, int s)
{
    float acc = 0.0f;
    const uint32_t pieces = g.row_stride >> 2;
    const bool masked = g.d_lim < g.d;
    for (uint32_t m = (uint32_t)s; m < pieces; m += 4u) {
        const float4 x = *reinterpret_cast<const float4*>(b + 4u * m);
        const float4 y = *reinterpret_cast<const float4*>(a_lds + 4u * m);
        float dx[4] = {x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w};
        if (masked) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t p = 4u * m + (uint32_t)e;
                const uint32_t dim = (p & ~31u) | ((p & 7u) << 2) | ((p >> 3) & 3u);
                if (dim >= g.d_lim) dx[e] = 0.0f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) acc = __builtin_fmaf(dx[e], dx[e], acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    return acc;
}

// selectNeighbors (src/hnsw.cpp:1078-1126) by one wavefront.  cand_id / cand_dist (LDS): nc candidates ascending by
// distance to the vector the list belongs to.  Fewer candidates than M: all of them.  Else the two nearest, then
// every further candidate, in order, that is not closer to an already selected one than to the owner, until M are
// selected.  A candidate's row is staged in LDS and the selected rows are compared with it 16 at a time, a quad each.
// Result in sel_id / sel_dist (LDS, room for M); returns how many.  Called by all 64 lanes of a one-wavefront workgroup.
__device__ uint32_t select_neighbors_wave(const Layer0& g, const uint32_t* cand_id, const float* cand_dist, uint32_t nc, uint32_t M,
                                          uint32_t* sel_id, float* sel_dist, float* rowbuf, int lane)
{
    if (nc < M) {
        for (uint32_t i = (uint32_t)lane; i < nc; i += 64u) { sel_id[i] = cand_id[i]; sel_dist[i] = cand_dist[i]; }
        __syncthreads();
        return nc;
    }
    if (lane < 2 && (uint32_t)lane < nc) { sel_id[lane] = cand_id[lane]; sel_dist[lane] = cand_dist[lane]; }
    uint32_t ns = nc < 2u ? nc : 2u;
    __syncthreads();
    const int quad = lane >> 2, s = lane & 3;
    for (uint32_t i = 2; i < nc && ns < M; i++) {
        const float* crow = g.vectors + (size_t)cand_id[i] * g.row_stride;
        for (uint32_t p = 4u * (uint32_t)lane; p < g.row_stride; p += 256u)
            *reinterpret_cast<float4*>(rowbuf + p) = *reinterpret_cast<const float4*>(crow + p);
        __syncthreads();
        const float own = cand_dist[i];
        bool drop = false;
        for (uint32_t r0 = 0; r0 < ns && !drop; r0 += 16u) {
            const uint32_t idx = r0 + (uint32_t)quad;
            const bool valid = idx < ns;
            const float* brow = g.vectors + (size_t)sel_id[valid ? idx : 0u] * g.row_stride;
            const float dd = quad_row_distance(rowbuf, brow, g, s);
            drop = __ballot(valid && dd < own) != 0ull;
        }
        __syncthreads();                        // (the staged row is overwritten by the next candidate)
        if (!drop) {
            if (lane == 0) { sel_id[ns] = cand_id[i]; sel_dist[ns] = own; }
            ns++;
            __syncthreads();
        }
    }
    return ns;
}

// Step 2: one wavefront per new vector.  Candidates = the search's result list of that vector (ascending), without
// the vector itself.  Writes the vector's own list (src/hnsw.cpp:266-276 for the new node) and M triples, unused ones
// with an all-ones key (they sort to the end).
__global__ __launch_bounds__(64) void select_new_kernel(Layer0 g, uint32_t first_id, uint32_t ef, uint32_t M,
                                                        const uint32_t* __restrict__ cand_ids, const float* __restrict__ cand_dist,
                                                        const uint32_t* __restrict__ cand_cnt,
                                                        unsigned long long* __restrict__ keys, float* __restrict__ vals)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float* rowbuf = reinterpret_cast<float*>(lds_raw);
    uint32_t* cid = reinterpret_cast<uint32_t*>(rowbuf + g.row_stride);
    float* cdist = reinterpret_cast<float*>(cid + ef);
    uint32_t* sel_id = reinterpret_cast<uint32_t*>(cdist + ef);
    float* sel_dist = reinterpret_cast<float*>(sel_id + kMaxList);
    const int lane = (int)threadIdx.x;
    const uint32_t j = blockIdx.x, self = first_id + j;
    const uint32_t raw = cand_cnt[j] < ef ? cand_cnt[j] : ef;
    uint32_t nc = 0;
    for (uint32_t b = 0; b < raw; b += 64u) {
        const uint32_t i = b + (uint32_t)lane;
        const uint32_t id = i < raw ? cand_ids[(size_t)j * ef + i] : kNone;
        const bool keep = id != kNone && id != self && id < g.n;
        const unsigned long long mask = __ballot(keep);
        const uint32_t at = nc + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (keep) { cid[at] = id; cdist[at] = cand_dist[(size_t)j * ef + i]; }
        nc += (uint32_t)__popcll(mask);
    }
    __syncthreads();
    const uint32_t ns = select_neighbors_wave(g, cid, cdist, nc, M, sel_id, sel_dist, rowbuf, lane);
    if ((uint32_t)lane < g.deg0) {
        const bool live = (uint32_t)lane < ns;
        g.adj0[(size_t)self * g.deg0 + lane] = live ? sel_id[lane] : kNone;
        g.dist0[(size_t)self * g.deg0 + lane] = live ? sel_dist[lane] : 0.0f;
    }
    if ((uint32_t)lane < M) {
        const bool live = (uint32_t)lane < ns;
        keys[(size_t)j * M + lane] = live ? (((unsigned long long)sel_id[lane] << 32) | self) : ~0ull;
        vals[(size_t)j * M + lane] = live ? sel_dist[lane] : 0.0f;
    }
}

// Step 3: the triples sorted by (neighbour, new vector).  The wavefront of a triple that starts a neighbour's run
// handles the whole run (the others leave at once): it holds the neighbour's list one entry per lane, appends the
// newcomers in turn and, when the list is full, runs the selection over the list plus the newcomer (src/hnsw.cpp:277-302:
// the list sorted by distance, then selectNeighbors with M = the list's capacity).
__global__ __launch_bounds__(64) void backlink_kernel(Layer0 g, const unsigned long long* __restrict__ keys, const float* __restrict__ vals,
                                                      uint32_t npairs)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float* rowbuf = reinterpret_cast<float*>(lds_raw);
    uint32_t* lid = reinterpret_cast<uint32_t*>(rowbuf + g.row_stride);     // the list (+ the newcomer)
    float* ldist = reinterpret_cast<float*>(lid + kMaxList + 1);
    uint32_t* cid = reinterpret_cast<uint32_t*>(ldist + kMaxList + 1);      // ... sorted
    float* cdist = reinterpret_cast<float*>(cid + kMaxList + 1);
    uint32_t* sel_id = reinterpret_cast<uint32_t*>(cdist + kMaxList + 1);
    float* sel_dist = reinterpret_cast<float*>(sel_id + kMaxList);
    const int lane = (int)threadIdx.x;
    const uint32_t p = blockIdx.x;
    const unsigned long long key0 = keys[p];
    const uint32_t nb = (uint32_t)(key0 >> 32);
    if (nb == kNone || nb >= g.n) return;
    if (p > 0 && (uint32_t)(keys[p - 1] >> 32) == nb) return;
    const uint32_t cap = g.deg0;
    uint32_t my_id = kNone; float my_dist = 0.0f;
    if ((uint32_t)lane < cap) { my_id = g.adj0[(size_t)nb * cap + lane]; my_dist = g.dist0[(size_t)nb * cap + lane]; }
    uint32_t cnt = (uint32_t)__popcll(__ballot(my_id != kNone));       // (the live entries are a prefix)
    if ((uint32_t)lane < cap) { lid[lane] = my_id; ldist[lane] = my_dist; }
    __syncthreads();
    for (uint32_t e = p; e < npairs; e++) {
        const unsigned long long key = keys[e];
        if ((uint32_t)(key >> 32) != nb) break;
        const uint32_t q = (uint32_t)key;
        const float dq = vals[e];
        if (cnt < cap) {
            if (lane == 0) { lid[cnt] = q; ldist[cnt] = dq; }
            cnt++;
            __syncthreads();
            continue;
        }
        // full: the cap + 1 entries in ascending order of (distance, id), then the selection rule
        if (lane == 0) { lid[cap] = q; ldist[cap] = dq; }
        __syncthreads();
        const uint32_t total = cap + 1u;
        if ((uint32_t)lane < total) {
            const uint32_t id = lid[lane]; const float dd = ldist[lane];
            uint32_t rank = 0;
            for (uint32_t o = 0; o < total; o++) {
                const float od = ldist[o]; const uint32_t oi = lid[o];
                rank += (od < dd || (od == dd && oi < id)) ? 1u : 0u;
            }
            cid[rank] = id; cdist[rank] = dd;
        }
        __syncthreads();
        cnt = select_neighbors_wave(g, cid, cdist, total, cap, sel_id, sel_dist, rowbuf, lane);
        if ((uint32_t)lane < cap) {
            lid[lane] = (uint32_t)lane < cnt ? sel_id[lane] : kNone;
            ldist[lane] = (uint32_t)lane < cnt ? sel_dist[lane] : 0.0f;
        }
        __syncthreads();
    }
    if ((uint32_t)lane < cap) {
        const bool live = (uint32_t)lane < cnt;
        g.adj0[(size_t)nb * cap + lane] = live ? lid[lane] : kNone;
        g.dist0[(size_t)nb * cap + lane] = live ? ldist[lane] : 0.0f;
    }
}

// Layer 1's lists, which the index keeps by slot, written by node_order: row order1[s] of adj1 = list of slot s.
__global__ void layer1_by_order_kernel(const uint32_t* __restrict__ order1, const uint32_t* __restrict__ up_adj_order, uint32_t n1, uint32_t degu,
                                       uint32_t n, uint32_t* __restrict__ adj1)
{
    const size_t total = (size_t)n1 * degu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t s = (uint32_t)(i / degu), j = (uint32_t)(i % degu);
        const uint32_t o = order1[s];
        if (o < n) adj1[(size_t)o * degu + j] = up_adj_order[i];
    }
}

template <class T>
void grow(T*& p, size_t& have, size_t want)
{
    if (have >= want) return;
    if (p) HIP_CHECK(hipFree(p));
    p = nullptr; have = 0;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T)));
    have = want;
}

}  // namespace

bool DeviceBuild::supports(const HostIndex& h)
{
    return h.Mmax0_ >= 2 && h.Mmax0_ <= kMaxList && h.M_ >= 2 && h.M_ <= h.Mmax0_ && h.efConstruction_ >= h.M_ && h.efConstruction_ <= 4096;
}

void DeviceBuild::release()
{
    for (void* p : {(void*)dist0, (void*)d_q, (void*)d_cids, (void*)d_cdist, (void*)d_ccnt, (void*)d_keys, (void*)d_keys2, (void*)d_vals,
                    (void*)d_vals2, d_tmp, (void*)adj1, (void*)d_order1, (void*)v_up_off, (void*)v_up_down})
        if (p) (void)hipFree(p);
    *this = DeviceBuild{};
}

// The edge distances of layer 0 as the host has them (the lists themselves are the index's own device copy, which
// the caller has just brought up to date).
void DeviceBuild::from_host(DeviceIndex& dev, const HostIndex& h)
{
    if (dev.device < 0 || !dev.g.adj0) throw std::logic_error("device build: index not on a device");
    HIP_CHECK(hipSetDevice(dev.device));
    const size_t words = (size_t)dev.g.n * dev.g.deg0;
    const std::vector<float>& hd = h.layers_[0].nb_dist;
    if (hd.size() < words) throw std::logic_error("device build: layer 0 is shorter than the index");
    grow(dist0, dist0_words, words);
    HIP_CHECK(hipMemcpy(dist0, hd.data(), words * sizeof(float), hipMemcpyHostToDevice));
}

// Layer 1 as a graph of its own: the search kernel walks a "layer 0" whose lists are indexed by node_order, so layer
// 1's lists are re-written that way (adj1), and the layers above it become the view's upper layers — the same arrays,
// entered one layer later, except that the lowest of them must lead down to node_orders instead of layer-1 slots.
// `upper`: the flat form of the upper layers that dev.refresh_graph has just put on the device.
void DeviceBuild::upper_view(DeviceIndex& dev, const FlatGraph& upper)
{
    view_ok = false;
    if (upper.nlayers < 2) return;                       // no layer 1 yet
    HIP_CHECK(hipSetDevice(dev.device));
    const uint32_t n = dev.g.n, degu = dev.g.degu, nl = upper.nlayers;
    const uint32_t n1 = upper.up_off[2] - upper.up_off[1];
    if (n1 == 0) return;
    const bool fresh = adj1_words < (size_t)n * degu;
    grow(adj1, adj1_words, (size_t)n * degu);
    if (fresh) HIP_CHECK(hipMemset(adj1, 0xFF, (size_t)n * degu * 4));
    grow(d_order1, order1_words, (size_t)n1 + (n1 >> 3) + 1024);
    HIP_CHECK(hipMemcpy(d_order1, upper.up_order.data(), (size_t)n1 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layer1_by_order_kernel, dim3(1024), dim3(256), 0, nullptr, d_order1, dev.g.up_adj_order, n1, degu, n, adj1);
    HIP_CHECK(hipGetLastError());
    // the view's upper layers: real layers 2 .. nl-1 as layers 1 .. nl-2
    const uint32_t off2 = upper.up_off[2], U2 = upper.up_off[nl] - off2;
    std::vector<uint32_t> voff(nl, 0u), vdown(std::max<uint32_t>(U2, 1u), 0u);
    for (uint32_t l = 1; l + 1 <= nl - 1; l++) voff[l + 1] = upper.up_off[l + 2] - off2;      // (voff[0] = voff[1] = 0)
    for (uint32_t i = 0; i < U2; i++) {
        const bool lowest = nl > 2 && off2 + i < upper.up_off[3];
        vdown[i] = lowest ? upper.up_order[off2 + i] : upper.up_down[off2 + i];
    }
    grow(v_up_off, v_up_off_words, (size_t)nl + 8);
    grow(v_up_down, v_up_down_words, (size_t)vdown.size() + (vdown.size() >> 3) + 1024);
    HIP_CHECK(hipMemcpy(v_up_off, voff.data(), voff.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(v_up_down, vdown.data(), vdown.size() * 4, hipMemcpyHostToDevice));
    view = dev.g;
    view.nlayers = nl - 1;
    view.deg0 = degu;
    view.adj0 = adj1;
    view.up_off = v_up_off;
    view.up_down = v_up_down;
    view.up_adj_slot = dev.g.up_adj_slot + (size_t)off2 * degu;
    view.up_adj_order = dev.g.up_adj_order + (size_t)off2 * degu;
    view.top_ep_slot = nl > 2 ? upper.ep_slot[nl - 1] : upper.ep_order[1];
    view.top_ep_order = nl > 2 ? upper.ep_order[nl - 1] : upper.ep_order[1];
    HIP_CHECK(hipDeviceSynchronize());
    view_ok = true;
}

// (Runs while host threads change the index's upper layers: reads vector_data_ and the build parameters, nothing else of `h`.)
void DeviceBuild::step(DeviceIndex& dev, const HostIndex& h, uint32_t first_id, uint32_t cnt, uint32_t flags,
                       const uint32_t* upper_ids, uint32_t n_upper, uint32_t* cand_ids, float* cand_dist, uint32_t* cand_cnt)
{
    if (cnt == 0) return;
    if (!dist0 || dist0_words < (size_t)dev.g.n * dev.g.deg0) throw std::logic_error("device build: step before from_host");
    if ((uint64_t)first_id + cnt > dev.g.n) throw std::logic_error("device build: vectors beyond the device's copy of the index");
    HIP_CHECK(hipSetDevice(dev.device));
    const uint32_t ef = (uint32_t)h.efConstruction_, M = (uint32_t)h.M_, d = dev.g.d;
    if (!supports(h) || dev.g.deg0 > (uint32_t)kMaxList || M > dev.g.deg0)
        throw std::logic_error("device build: list lengths the kernels are not built for");
    grow(d_q, q_words, (size_t)cnt * d);
    grow(d_cids, cand_words, (size_t)cnt * ef);
    grow(d_cdist, cdist_words, (size_t)cnt * ef);
    grow(d_ccnt, ccnt_words, (size_t)cnt);
    const size_t npairs = (size_t)cnt * M;
    grow(d_keys, keys_words, npairs); grow(d_keys2, keys2_words, npairs);
    grow(d_vals, vals_words, npairs); grow(d_vals2, vals2_words, npairs);
    HIP_CHECK(hipMemcpy(d_q, h.row(first_id), (size_t)cnt * d * sizeof(float), hipMemcpyHostToDevice));
    dev.search(d_q, cnt, (int)ef, ef, flags, d_cids, d_cdist, d_ccnt, nullptr, nullptr);

    Layer0 g;
    g.n = dev.g.n; g.d = d; g.row_stride = dev.g.row_stride; g.deg0 = dev.g.deg0;
    g.d_lim = (flags & HNSW_AMD_DIST_SCALAR) ? d : (d & ~15u);
    g.vectors = dev.g.vectors; g.adj0 = const_cast<uint32_t*>(dev.g.adj0); g.dist0 = dist0;
    const size_t lds_a = ((size_t)g.row_stride + 2 * (size_t)ef + 2 * kMaxList) * 4;
    const size_t lds_b = ((size_t)g.row_stride + 4 * (kMaxList + 1) + 2 * kMaxList) * 4;
    if (lds_a > 65536 || lds_b > 65536) throw std::length_error("device build: a row and its candidate list do not fit the LDS");
    hipLaunchKernelGGL(select_new_kernel, dim3(cnt), dim3(64), lds_a, nullptr, g, first_id, ef, M, d_cids, d_cdist, d_ccnt,
                       reinterpret_cast<unsigned long long*>(d_keys), d_vals);
    HIP_CHECK(hipGetLastError());
    size_t need = 0;
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, reinterpret_cast<unsigned long long*>(d_keys),
                                                 reinterpret_cast<unsigned long long*>(d_keys2), d_vals, d_vals2, (int)npairs, 0, 64, nullptr));
    if (need > tmp_bytes) {
        if (d_tmp) HIP_CHECK(hipFree(d_tmp));
        d_tmp = nullptr; tmp_bytes = 0;
        HIP_CHECK(hipMalloc(&d_tmp, need));
        tmp_bytes = need;
    }
    size_t have = tmp_bytes;
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(d_tmp, have, reinterpret_cast<unsigned long long*>(d_keys),
                                                 reinterpret_cast<unsigned long long*>(d_keys2), d_vals, d_vals2, (int)npairs, 0, 64, nullptr));
    hipLaunchKernelGGL(backlink_kernel, dim3((uint32_t)npairs), dim3(64), lds_b, nullptr, g,
                       reinterpret_cast<const unsigned long long*>(d_keys2), d_vals2, (uint32_t)npairs);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());

    // the construction searches of the vectors that also live in layer 1, on the view of that layer; the host selects
    // and links from the lists (the buffers of the layer-0 candidates are free again: n_upper <= cnt)
    if (n_upper == 0) return;
    for (uint32_t j = 0; j < n_upper; j++) cand_cnt[j] = 0;
    if (!view_ok || n_upper > cnt) return;
    std::vector<float> rows((size_t)n_upper * d);
    for (uint32_t j = 0; j < n_upper; j++) std::memcpy(&rows[(size_t)j * d], h.row(upper_ids[j]), (size_t)d * sizeof(float));
    HIP_CHECK(hipMemcpy(d_q, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
    dev.search_on(view, d_q, n_upper, (int)ef, ef, flags, d_cids, d_cdist, d_ccnt, nullptr);
    HIP_CHECK(hipMemcpy(cand_ids, d_cids, (size_t)n_upper * ef * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(cand_dist, d_cdist, (size_t)n_upper * ef * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(cand_cnt, d_ccnt, (size_t)n_upper * 4, hipMemcpyDeviceToHost));
}

// Layer 0 back into the host's arrays (lists, edge distances, counts): what every later host-side step — a sequential
// insert, the flat export, the graph file — reads.
void DeviceBuild::to_host(const DeviceIndex& dev, HostIndex& h)
{
    if (!dist0) throw std::logic_error("device build: to_host before from_host");
    HIP_CHECK(hipSetDevice(dev.device));
    LayerStore& L = h.layers_[0];
    const uint32_t n = dev.g.n, deg = dev.g.deg0;
    const size_t words = (size_t)n * deg;
    if (L.nb_order.size() < words || L.nb_slot.size() < words || L.nb_dist.size() < words || L.cnt.size() < n || L.deg != deg)
        throw std::logic_error("device build: the host's layer 0 does not match the device's");
    HIP_CHECK(hipMemcpy(L.nb_order.data(), dev.g.adj0, words * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(L.nb_dist.data(), dist0, words * 4, hipMemcpyDeviceToHost));
    std::memcpy(L.nb_slot.data(), L.nb_order.data(), words * 4);           // (layer 0: slot = node_order, same padding)
    for (uint32_t i = 0; i < n; i++) {
        uint32_t c = 0;
        while (c < deg && L.nb_order[(size_t)i * deg + c] != kPad) c++;
        L.cnt[i] = c;
        if (i < L.sorted.size()) L.sorted[i] = 0;
    }
}

}  // namespace hnsw_amd
