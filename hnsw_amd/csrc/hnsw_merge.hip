// On-device k-way merge of per-shard top-k lists (the combine step of the sharded
// multi-GPU layout: every GPU searches its own sub-graph, the lists are all-gathered
// over RCCL, and this kernel turns S lists per query into one).
// The reference is a single in-memory index and has no counterpart of this step.
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>

#include "../../include/hnsw_amd.h"

namespace hnsw_amd {

namespace {

constexpr int kMaxShards = 64;
struct MergeArgs {
    const float* dist;       // S x nq x k
    const uint32_t* ids;     // S x nq x k, local to the shard; 0xFFFFFFFF = padding
    uint32_t offset[kMaxShards];   // first global id of each shard
    uint32_t S, nq, k, out_k;
    size_t shard_stride;     // elements between consecutive shards' blocks (>= nq*k)
    uint32_t* out_ids;       // nq x out_k
    float* out_dist;
};

// One wavefront per query.  All S*k candidates go to LDS as (distance bits, global id);
// each lane ranks its candidates against all of them (ties broken by id, so ranks are
// unique) and writes those that land in the first out_k places.
__global__ __launch_bounds__(64) void merge_topk_kernel(MergeArgs a)
{
    extern __shared__ __align__(8) uint2 cand[];
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint32_t total = a.S * a.k;
    for (uint32_t i = lane; i < total; i += 64) {
        const uint32_t s = i / a.k, j = i - s * a.k;
        const size_t at = (size_t)s * a.shard_stride + (size_t)q * a.k + j;
        const uint32_t id = a.ids[at];
        const float d = a.dist[at];
        // padding sorts last; distances are non-negative, so their bit patterns order like the floats
        cand[i] = id == 0xFFFFFFFFu ? make_uint2(0x7F800000u, 0xFFFFFFFFu) : make_uint2(__float_as_uint(d), id + a.offset[s]);
    }
    __syncthreads();
    for (uint32_t i = lane; i < total; i += 64) {
        const uint2 me = cand[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < total; j++) {
            const uint2 o = cand[j];
            rank += (o.x < me.x || (o.x == me.x && (o.y < me.y || (o.y == me.y && j < i)))) ? 1u : 0u;
        }
        if (rank < a.out_k) {
            a.out_ids[(size_t)q * a.out_k + rank] = me.y;
            a.out_dist[(size_t)q * a.out_k + rank] = me.y == 0xFFFFFFFFu ? __builtin_inff() : __uint_as_float(me.x);
        }
    }
    for (uint32_t i = total + lane; i < a.out_k; i += 64) {
        a.out_ids[(size_t)q * a.out_k + i] = 0xFFFFFFFFu;
        a.out_dist[(size_t)q * a.out_k + i] = __builtin_inff();
    }
}

}  // namespace

void merge_topk(const float* d_dist, const uint32_t* d_ids, const uint32_t* shard_offset, uint32_t S, uint32_t nq, uint32_t k,
                uint32_t out_k, size_t shard_stride, uint32_t* d_out_ids, float* d_out_dist, void* stream)
{
    if (shard_stride == 0) shard_stride = (size_t)nq * k;
    if (shard_stride < (size_t)nq * k) throw std::invalid_argument("merge_topk: shard_stride smaller than one block");
    if (S < 1 || S > (uint32_t)kMaxShards) throw std::invalid_argument("merge_topk: 1..64 shards");
    if (k < 1 || out_k < 1 || (size_t)S * k > 8192) throw std::invalid_argument("merge_topk: S*k must be 1..8192");
    if (nq == 0) return;
    MergeArgs a{};
    a.dist = d_dist; a.ids = d_ids; a.S = S; a.nq = nq; a.k = k; a.out_k = out_k; a.out_ids = d_out_ids; a.out_dist = d_out_dist; a.shard_stride = shard_stride;
    for (uint32_t s = 0; s < S; s++) a.offset[s] = shard_offset[s];
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(64), (size_t)S * k * sizeof(uint2), (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP: merge_topk launch: ") + hipGetErrorString(e));
}

}  // namespace hnsw_amd
