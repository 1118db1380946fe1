// HIP kernels of the HNSW query path for gfx950 (MI355X), and their launch code.
//
// One 64-lane wavefront (= one workgroup) runs one query from entry point to sorted
// result, entirely on chip except for the adjacency rows and base vectors it gathers
// from HBM:
//   - query vector, candidate heap C and the exact visited hash set: LDS; neighbour ids,
//     distances and the result list W: registers;
//   - a neighbour list is one coalesced read; rows already visited are masked out of the row
//     loads (no compaction), and all row loads of an expansion are in flight together;
//   - distances reproduce the reference's float arithmetic bit for bit: the 4 lane sums of
//     its SSE loop are 4 sequential fused-multiply-add chains per row, run by the 4 lanes of
//     a quad, each reading its own 32 bytes of every 128-byte line (the lines are stored
//     regrouped by lane sum, see stored_pos);
//   - heap updates follow libstdc++'s push_heap/pop_heap exactly, executed by the whole
//     wavefront in a constant number of LDS round trips, because with the reference's
//     int32 candidate keys ties are the rule and their pop order decides which nodes are
//     expanded.
// Follows reference src/hnsw.cpp:373-449 (aproximateKnn), :676-717 (searchLayerOne),
// :725-787 (searchLayer), src/hnsw.h:334-400 (distance), :55-125 (visited set).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <vector>

#include "hnsw_device.h"

// Scratch of the search entry point that allocates nothing (hnsw_amd_search_batch_device_noalloc): the two
// overflow lists of the retry protocol (a few header words, then query indices; twice), which the other entry
// points keep in a buffer they allocate.  One copy per GPU, held until the process ends (512 KB).  In host
// code this name is only a handle for hipGetSymbolAddress: the address a kernel can use is looked up once
// per device (noalloc_scratch_for) and passed as an argument like any other buffer.
constexpr uint32_t kNoAllocMaxQueries = 65535;
constexpr uint32_t kFailHeader = 8;      // words before the list: [0] count, [1] largest visited set, [2] most room C was asked for in the
                                         // launch, [3] spill tables handed out, [4] sum of visited counts / 16 and [5] number of a sample (one in 32) of the queries that finished,
                                         // [6] queries handed back because the integer arithmetic of the byte rows does not apply to them, [7] always zero
static __device__ __align__(256) uint32_t scratch[2 * (kNoAllocMaxQueries + kFailHeader)];

namespace hnsw_amd {

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            throw std::runtime_error(std::string("HIP: ") + #expr + ": " + hipGetErrorString(e_));   \
    } while (0)

namespace {

// Device address of `scratch` on each GPU (every GPU holds its own copy at its own address); filled when an
// index is first uploaded to that GPU, with that GPU current.
constexpr int kMaxDevices = 64;
uint32_t* g_noalloc_scratch[kMaxDevices] = {};
bool g_noalloc_headers_zero[kMaxDevices] = {};      // both list headers in the array are known to be zero (every launch leaves them so)
std::mutex g_noalloc_mutex;

uint32_t* noalloc_scratch_for(int dev, bool lookup)
{
    if (dev < 0 || dev >= kMaxDevices) throw std::invalid_argument("search: device ordinal beyond the supported 64");
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_scratch[dev] && lookup) {
        uint32_t* ptr = nullptr;
        HIP_CHECK(hipSetDevice(dev));
        HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
        g_noalloc_scratch[dev] = ptr;
    }
    return g_noalloc_scratch[dev];
}

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

// (Loads of lists and rows are left as generic loads on purpose.  Giving them the device address space lets the
// compiler wait for exactly the loads it needs, but also lets it move them past the LDS work they were placed in
// front of, and the overlap written into the code — a neighbour list requested before the heap is repaired, rows
// requested before the query is read — is lost: measured 9 % slower at ef=20 and 40 % on an earlier form of the scalar-order path.)
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rflf(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// ---------------------------------------------------------------------------------------
// The heaps: W is a max-heap on the float distance (entries (dist bits, id), comp(a,b) = a.dist < b.dist,
// reference src/node.h:48-52), C a min-heap on an int32 key (entries (key, id), comp(a,b) = a.key > b.key,
// src/hnsw.h:210-215).  libstdc++'s __push_heap / __adjust_heap are executed by the whole wavefront in a
// constant number of LDS round trips instead of one per tree level (below the eighth level of a very
// large heap: one per level).  They leave the array exactly as std::push_heap / std::pop_heap do:
//  - push: the new element climbs the fixed chain of ancestors of index n-1 while
//    comp(ancestor, value); lane k owns ancestor k, one ballot gives the climb length.
//  - pop: __adjust_heap first walks the hole from the root to a leaf choosing, at every
//    node, a child by comparing the two SIBLINGS only (never the moving value), so every
//    node's choice can be evaluated at once (lane i of a register holds the child node i hands
//    the hole to) and the walk becomes one lane read per level; then the saved last element
//    climbs back up that same path (one ballot).  Lane k owns step k of the path.
// Both are called with wave-uniform arguments by all 64 lanes.
// ---------------------------------------------------------------------------------------
struct WOrder {   // comp(a,b) = a.dist < b.dist
    static __device__ __forceinline__ bool before(uint2 a, uint2 b) { return __uint_as_float(a.x) < __uint_as_float(b.x); }
};
struct COrder {   // comp(a,b) = a.key > b.key
    static __device__ __forceinline__ bool before(uint2 a, uint2 b) { return (int32_t)a.x > (int32_t)b.x; }
};
// Candidate-heap entries of four bytes: key in the top byte, node_order in the other three (an index of fewer than 2^24
// rows whose keys provably stay below 256; the launcher decides, see DeviceIndex::launch_tier).  Same order, same heap
// steps; half the LDS.
struct COrderPacked {
    static __device__ __forceinline__ bool before(uint32_t a, uint32_t b) { return (a >> 24) > (b >> 24); }
};
__device__ __forceinline__ uint2 rfl_entry(uint2 e) { return make_uint2(rfl(e.x), rfl(e.y)); }
__device__ __forceinline__ uint32_t rfl_entry(uint32_t e) { return rfl(e); }

// Minimum of an unsigned value over the 64 lanes with DPP moves (no LDS round trips):
// within quads, half rows and rows, then across the four rows; the result is read from lane 63.
// (Lanes a step does not write take all ones, the identity of the minimum: each step is one v_min_u32 with a DPP operand.)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
    uint32_t y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0xB1, 0xF, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x4E, 0xF, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x141, 0xF, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x140, 0xF, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x142, 0xA, 0xF, false); x = x < y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x143, 0xC, 0xF, false); x = x < y ? x : y;
    return __builtin_amdgcn_readlane(x, 63);
}


// push in two halves, so that independent work can be placed between the LDS read and its use
// (E: the entry type, uint2 or the packed uint32_t)
template <class Ord, class E>
__device__ __forceinline__ E heap_push_read(const E* a, uint32_t n /* size including the new element */, int lane)
{
    // every lane reads (the lanes past the chain of ancestors read some valid entry and are masked out of the
    // ballot): no divergent region around the load
    const uint32_t anc = n >> ((lane + 1) & 31);
    return a[(anc > 0u ? anc : 1u) - 1u];
}

template <class Ord, class E>
__device__ __forceinline__ void heap_push_finish(E* a, uint32_t n, E v, E x /* from heap_push_read */, int lane)
{
    const uint32_t depth = 31u - (uint32_t)__clz((int)n);        // levels above index n-1
    const unsigned long long m = __ballot(Ord::before(x, v)) & ((1ull << depth) - 1ull);
    const uint32_t t = (uint32_t)__builtin_ctzll(~m);            // ancestors that move down
    // lanes below t hand their ancestor down one level, lane t drops the new element into the gap
    if ((uint32_t)lane <= t) a[(n >> lane) - 1] = (uint32_t)lane < t ? x : v;
}

template <class Ord, class E>
__device__ __forceinline__ void heap_push_wave(E* a, uint32_t n /* size including the new element */, E v, int lane)
{
    heap_push_finish<Ord, E>(a, n, v, heap_push_read<Ord, E>(a, n, lane), lane);
}

// Returns the new root (valid when n > 1).
template <class Ord, class E>
__device__ __forceinline__ E heap_pop_wave(E* a, uint32_t n /* size before the pop */, int lane)
{
    const uint32_t len = n - 1;
    const E v = a[len];
    const uint32_t nint = (len - 1) >> 1;      // nodes with both children inside [0, len)
    const uint32_t nint1 = nint < 256u ? nint : 256u;      // nodes 0..255, the first eight levels, are decided at once
    // Every node of the first eight levels decides at once which child the hole moves to (node 64b + lane in register
    // b): next = 2i + 2 - comp(right, left).  The reads of the first two registers (heaps of up to 258 entries) are
    // issued together without a test, those of the other two under one.  (Nodes past nint1 read node 0's children;
    // the walk never looks at their entries.)
    uint32_t ch0, ch1, ch2 = 0, ch3 = 0;
    {
        auto children = [&](uint32_t base, E& r, E& l) {
            const uint32_t i = base + (uint32_t)lane, ic = i < nint1 ? i : 0u;
            r = a[2 * ic + 2]; l = a[2 * ic + 1];
        };
        auto next_of = [&](uint32_t base, const E& r, const E& l) -> uint32_t {
            return 2u * (base + (uint32_t)lane) + 2u - (Ord::before(r, l) ? 1u : 0u);
        };
        E r0, l0, r1, l1;
        children(0, r0, l0);
        children(64, r1, l1);
        ch0 = next_of(0, r0, l0);
        ch1 = next_of(64, r1, l1);
        if (nint1 > 128) {
            E r2, l2, r3, l3;
            children(128, r2, l2);
            children(192, r3, l3);
            ch2 = next_of(128, r2, l2);
            ch3 = next_of(192, r3, l3);
        }
    }
    // Walk the hole down: one lane read per level (the index of the node the hole is at selects the lane).
    uint32_t idx = 0;                                         // 0-based node the hole is at
    {
        const uint32_t lim0 = nint1 < 64u ? nint1 : 64u;     // nodes 0..63 are in ch0
        while (idx < lim0) idx = __builtin_amdgcn_readlane(ch0, idx);
        while (idx < nint1) {                                // (at most three steps: nodes 64..255 are levels 6 and 7)
            const uint32_t blk = idx >> 6;
            uint32_t t = ch1;                                // the register is picked with two selects, not with branches
            t = blk >= 2u ? ch2 : t;
            t = blk >= 3u ? ch3 : t;
            idx = __builtin_amdgcn_readlane(t, idx & 63u);
        }
        // a heap of more than 513 entries: the levels below the eighth, one LDS round trip each
        while (idx < nint) {
            const E r = a[2 * idx + 2], l = a[2 * idx + 1];
            const bool left = Ord::before(rfl_entry(r), rfl_entry(l));
            idx = 2 * idx + 2 - (left ? 1u : 0u);
        }
    }
    uint32_t p = idx + 1;                                     // 1-based: a leading 1 followed by one bit per step
    uint32_t d = 31u - (uint32_t)__clz((int)p);               // steps taken
    if ((len & 1u) == 0 && p - 1 == (len - 2) >> 1) { p = 2 * p; d++; }     // a last node with a left child only
    // lane k owns step k of the path: the element at my_from moves up to my_to
    const uint32_t sh = d - (uint32_t)lane;                                  // meaningful for lane < d
    const uint32_t my_to = (p >> (sh & 31u)) - 1;
    const uint32_t from1 = p >> ((sh - 1) & 31u);          // 1-based; lanes past the path get some index <= p
    const E x = a[(from1 > 0u ? from1 : 1u) - 1u];         // read by every lane, the ballot is masked instead
    const unsigned long long cm = __ballot(Ord::before(x, v)) & ((1ull << d) - 1ull);
    uint32_t s = 0;                                       // path elements (from the leaf up) that stay put
    if (d > 0) { s = (uint32_t)__builtin_clzll(~(cm << (64 - d))); if (s > d) s = d; }
    const uint32_t moved = d - s;
    // the first `moved` path elements climb one step; the saved last element lands on step `moved`
    // (for moved == d that is the end of the path)
    if ((uint32_t)lane <= moved) a[my_to] = (uint32_t)lane < moved ? x : v;
    // new root: the first path element if it moved up, else the re-inserted value
    return moved > 0 ? rfl_entry(x) : v;
}

// Checking build only (-DHNSW_CHECKS, `make -C hnsw_amd/csrc checks`): invariants of the lock-step visited sets are
// re-read from LDS and a violation is counted in a device word that the launcher turns into an error.  The shipped
// library contains none of this.
#ifdef HNSW_CHECKS
__device__ uint32_t g_check_failures = 0;
#define HNSW_CHECK_THAT(cond) do { if (!(cond)) atomicAdd(&g_check_failures, 1u); } while (0)
#else
#define HNSW_CHECK_THAT(cond) do { } while (0)
#endif

// Exact visited set: open addressing, linear probing, one LDS compare-and-swap per probe.
// Returns true when `id` was not in the set (and now is).
// The table has `size` slots, any number (no power of two needed): the start slot is the
// multiplicative hash scaled to the table, and the probe wraps by comparison.
__device__ __forceinline__ bool visited_insert32(uint32_t* hash, uint32_t size, uint32_t id)
{
    uint32_t slot = __umulhi(id * 0x9E3779B1u, size);
    uint32_t old;
    do {                                    // one exit: the lanes' loop masks stay simple
        old = atomicCAS(&hash[slot], kEmpty, id);
        slot = slot + 1 == size ? 0u : slot + 1;
    } while (old != kEmpty && old != id);
    return old == kEmpty;
}

// Half-size form of the same exact set for large ef, where LDS decides how many queries a CU holds:
// 2^rb regions selected by the low rb bits of the id, 16-bit slots holding the remaining bits
// (the host checks that they fit below 0xFFFF = empty), linear probing inside the region.
// No atomic is needed: the workgroup is one wavefront whose lanes run the loop in lockstep, so all
// reads of an iteration see the table before any of that iteration's writes.  A lane that finds its
// slot empty writes its tag (a 2-byte LDS store) and looks again in the next iteration: if several
// lanes wrote different tags to one slot, one of them survives, its lane sees its own tag and is
// done, the others see a foreign tag and probe on.  (Ids within one neighbour list are distinct, so
// a tag read back as one's own after writing was put there by this lane.)  A region can fill up
// while the table as a whole has room: that is reported through `overflow` (the query is then
// re-run with larger tables, like any other scratch overflow).
__device__ __forceinline__ bool visited_insert16(uint32_t* hash, uint32_t rb, uint32_t per_region, uint32_t id, bool& overflow)
{
    uint16_t* h16 = reinterpret_cast<uint16_t*>(hash);
    const uint32_t tag = id >> rb;
    // the region's slots are probed two at a time (one 32-bit word; regions hold an even number of slots)
    const uint32_t pairs = per_region >> 1;
    const uint32_t first = (id & ((1u << rb) - 1u)) * pairs;      // in words
    const uint32_t last = first + pairs;
    uint32_t word = first + __umulhi(tag * 0x9E3779B1u, pairs);
    uint32_t probes = 0, wrote_at = 0xFFFFFFFFu;      // the slot this lane last wrote its tag to
    bool hit, fresh = false;
    do {
        const uint32_t w = __hip_atomic_load(&hash[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
        const bool hit_lo = lo == tag, hit_hi = hi == tag;
        hit = hit_lo || hit_hi;
        fresh = hit && wrote_at == 2u * word + (hit_lo ? 0u : 1u);
        const bool empty_lo = lo == 0xFFFFu, empty = empty_lo || hi == 0xFFFFu;
        const uint32_t target = 2u * word + (empty_lo ? 0u : 1u);
        if (!hit && empty) __hip_atomic_store(&h16[target], (uint16_t)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        wrote_at = (!hit && empty) ? target : wrote_at;
        const bool advance = !hit && !empty;
        const uint32_t next = word + 1 == last ? first : word + 1;
        probes += advance ? 1u : 0u;
        word = advance ? next : word;
    } while (!hit && probes < pairs);
    overflow = overflow || !hit;
    // a tag counted as this lane's own insertion sits where the lane wrote it
    HNSW_CHECK_THAT(!fresh || (wrote_at != 0xFFFFFFFFu && h16[wrote_at] == (uint16_t)tag));
    HNSW_CHECK_THAT(tag < 0xFFFFu);
    return fresh;
}

// Bucket form of the 16-bit set, for tables that must run nearly full (an index of many millions of rows needs
// so many regions for its ids to fit 16-bit tags that a region's share of the visited ids is small, and sizing
// every region for its worst case would give the saving away): 2^rb regions selected by the low rb bits of the
// id, each `8 * nchunk` 16-bit slots of which the last holds the region's fill count minus one (so that the
// cleared value 0xFFFF means empty) and the others are filled front to back.  A lookup reads the whole region
// (nchunk 16-byte LDS reads) and tests all tags at once with word arithmetic, so it costs the same however
// full the region is.  An id whose region is full goes to a small 32-bit table of the other kind (`ovf`); with
// regions sized a little above their expected share only a few per cent of the ids end up there.
// `regions` is any number (region = id mod regions, tag = id div regions < 0xFFFF; `magic` =
// floor(2^32 / regions) gives the quotient to within one).
// Once the overflow table has reached its limit the query is given a table in device memory (`spill`, same
// 32-bit form): from then on the overflow table is only read, and what it does not hold is looked up in and
// added to the spill table.  Only the few queries that visit far more nodes than most get there, late in their
// search, so the tables in LDS can be sized for the bulk of the queries instead of the worst one.
// Insertion follows the lock-step argument of visited_insert16: the lanes that did not find their tag write it to
// the slot the count points at, write the count, and read both back; of several lanes that wrote to one
// slot one finds its own tag there and is done, the others find a foreign tag and the advanced count and try the
// next slot.  (A tag read back as one's own was put there by this lane: the ids of one neighbour list are
// distinct, and slots at or past the count never hold a tag.)  `used_ovf` reports that the id was looked up in
// (and, when fresh, added to) the overflow table.
__device__ __forceinline__ bool visited_insert_bucket(uint32_t* hash, uint32_t regions, uint32_t magic, uint32_t nchunk, uint32_t* ovf,
                                                      uint32_t ovf_size, uint32_t* spill, uint32_t spill_slots, uint32_t id, bool& used_ovf)
{
    uint32_t tag = __umulhi(id, magic);
    uint32_t region = id - tag * regions;
    if (region >= regions) { tag++; region -= regions; }
    const uint32_t S = nchunk * 8u;
    uint16_t* h16 = reinterpret_cast<uint16_t*>(hash) + region * S;
    const uint4* reg = reinterpret_cast<const uint4*>(hash) + region * nchunk;
    const uint32_t tt = tag * 0x00010001u;
    // a half-word of x = w ^ tt is zero where the slot holds the tag; (x - 0x00010001) & ~x & 0x80008000 is
    // non-zero exactly when some half-word of x is zero
    // All four chunks a region can have are read at once, whatever the region's size (the reads past a region's end
    // land in the next region or in the overflow table behind the last one, and are ignored): one LDS round trip
    // instead of one per chunk, and no branch on the chunk count.
    uint32_t z = 0, last = 0;
    LDS_SYNC();
    const uint4 vc[4] = {reg[0], reg[1], reg[2], reg[3]};
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        const uint4 v = vc[c];
        const bool used = c < 2u || c < nchunk;                 // (a region has at least two chunks)
        const bool is_last = c >= 1u && c + 1u == nchunk;
        uint32_t x, zc = 0;
        x = v.x ^ tt; zc |= (x - 0x00010001u) & ~x;
        x = v.y ^ tt; zc |= (x - 0x00010001u) & ~x;
        x = v.z ^ tt; zc |= (x - 0x00010001u) & ~x;
        x = (v.w ^ tt) | (is_last ? 0x80000000u : 0u);          // the upper half of the last word is the count, not a tag
        zc |= (x - 0x00010001u) & ~x;
        z |= used ? zc : 0u;
        last = is_last ? v.w : last;
    }
    const bool hit = (z & 0x80008000u) != 0u;
    uint32_t count = ((last >> 16) + 1u) & 0xFFFFu;
    bool fresh = false, full = false, done = hit;
    while (!done) {                         // (one exit) the lanes still inserting run in lock-step
        const bool can = count < S - 1u;
        if (can) { h16[count] = (uint16_t)tag; h16[S - 1u] = (uint16_t)count; }
        LDS_SYNC();
        const uint32_t back = h16[count], hdr = h16[S - 1u];
        LDS_SYNC();
        fresh = can && back == tag;
        full = !can;
        done = fresh || full;
        count = (hdr + 1u) & 0xFFFFu;
    }
    used_ovf = full;
    HNSW_CHECK_THAT(tag < 0xFFFFu && region < regions);
    // the region's count never passes its usable slots, and every slot below it holds a tag
    HNSW_CHECK_THAT((((uint32_t)h16[S - 1u] + 1u) & 0xFFFFu) <= S - 1u);
    HNSW_CHECK_THAT(!(fresh || hit) || [&] { bool there = false; for (uint32_t i = 0; i + 1u < S; i++) there = there || h16[i] == (uint16_t)tag; return there; }());
    if (!full) return fresh;
    if (!spill) return visited_insert32(ovf, ovf_size, id);
    // the overflow table is closed: look only
    {
        uint32_t slot = __umulhi(id * 0x9E3779B1u, ovf_size), old;
        do {
            old = ovf[slot];
            slot = slot + 1 == ovf_size ? 0u : slot + 1;
        } while (old != kEmpty && old != id);
        if (old == id) return false;
    }
    return visited_insert32(spill, spill_slots, id);
}

// The visited set of one query, in one of its three forms (chosen by the host per launch).
struct VisitedSet {
    uint32_t* hash;       // primary table
    uint32_t size;        // its slots (32-bit form: words; 16-bit forms: half-words)
    uint32_t rb;          // 0xFFFFFFFF: 32-bit slots, otherwise region bits of the 16-bit probing form
    uint32_t nchunk;      // bucket form: 16-byte chunks per region; 0: a probing form
    uint32_t regions, magic;   // bucket form: number of regions, floor(2^32 / regions)
    uint32_t* ovf;        // bucket form: overflow table (32-bit slots)
    uint32_t ovf_size;
    uint32_t spill_slots; // bucket form: slots of a spill table in device memory
};

// `spill`: the query's spill table once it has one (bucket form only), else nullptr.  The bucket form has its own
// instantiations of the search kernel (BUCKET), so the probing forms do not carry its code and bookkeeping.
template <bool BUCKET>
__device__ __forceinline__ bool visited_insert(const VisitedSet& v, uint32_t* spill, uint32_t id, bool& overflow, bool& used_ovf)
{
    if constexpr (BUCKET) {
        return visited_insert_bucket(v.hash, v.regions, v.magic, v.nchunk, v.ovf, v.ovf_size, spill, v.spill_slots, id, used_ovf);
    } else {
        if (v.rb == 0xFFFFFFFFu) return visited_insert32(v.hash, v.size, id);
        return visited_insert16(v.hash, v.rb, v.size >> v.rb, id, overflow);
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

// Stored form of a base vector (and of the query copy in LDS): every 128-byte line holds 32
// consecutive dims, regrouped by the lane sum they feed in the reference's SSE loop
// (src/hnsw.h:349-380): dim 32*l + 4*k + j (k = 0..7, j = 0..3) sits at float 32*l + 8*j + k.
// The 8 elements that sum j takes from a line are then 32 contiguous bytes.
__host__ __device__ __forceinline__ uint32_t stored_pos(uint32_t dim)
{
    return (dim & ~31u) | ((dim & 3u) << 3) | ((dim >> 2) & 7u);
}

// Byte form of a base vector whose values are all integers in 0..255 (then float(byte) is the stored
// float, exactly): one byte per dim, rows padded to 128 bytes, every 128-dim block regrouped by
// lane sum: dim 128*b + 4*k + j (k = 0..31) sits at byte 128*b + 32*j + k, so chain j reads 32
// contiguous bytes per block.  The query copy in LDS uses the same order (as floats).
__host__ __device__ __forceinline__ uint32_t stored_pos_u8(uint32_t dim)
{
    return (dim & ~127u) | ((dim & 3u) << 5) | ((dim >> 2) & 31u);
}

// How the 64 lanes share the rows of a neighbour list.
//  SCALAR=false (the SSE loop): the 4 lane sums of a row are 4 independent sequential fused-multiply-add
//    chains in index order.  Four lanes per row, lane j runs chain j: it reads its 32 bytes of
//    every line (2 x 16 B) and nothing is exchanged until the final ((s0+s1)+s2)+s3.  A load
//    instruction covers 16 rows x 64 bytes, so a line is touched by 2 instructions.  A list is
//    handled 32 rows at a time as two sub-batches of 16 whose loads are in flight together.
//  SCALAR=true (the -DNO_MANUAL_VECTORIZATION build, src/hnsw.h:387-400): one running sum per row over all dims
//    in index order.  The same four lanes read the same bytes (coalesced 64-byte pieces, every line touched by two
//    instructions instead of eight); dim 4k + j of a line is lane j's k-th element, so the running sum visits the
//    lanes of the quad in turn: it travels as a rotating DPP operand (consume_item_scalar), one add per dim per wave
//    instruction for 16 rows at once.
template <bool SCALAR>
struct RowMap {
    static constexpr int kLanesPerRow = 4;
    static constexpr int kSub = 2;                                // sub-batches in flight
    static constexpr int kRowsPerSub = 64 / kLanesPerRow;
    static constexpr int kRowsPerBatch = kSub * kRowsPerSub;
    static constexpr int kItemFloats = 128;                       // dims one "item" (8 x 16-byte loads per lane) covers
};

struct DistCtx {
    const float* qs;        // query in LDS, stored form, zero padded
    uint32_t nfull;         // items processed without per-load tests
    uint32_t nitems;        // nfull (+1 when tail_loads > 0)
    uint32_t tail_loads;    // 16-byte loads per lane of the last, partial item
    uint32_t d4;            // scalar mode: dims whose square is rounded before the add
    uint32_t kf;            // scalar mode: element index (dim / 4) of the fused group of the last d % 4 dims, or all ones
};

template <bool SCALAR>
__device__ __forceinline__ DistCtx make_dist_ctx(const float* qs, uint32_t d)
{
    DistCtx c;
    c.qs = qs;
    c.d4 = d & ~3u;
    c.kf = (d & 3u) ? (c.d4 >> 2) : 0xFFFFFFFFu;
    if (SCALAR) {
        // every dim takes part; the last item is loaded in whole 16-dim pieces (rows and query are zero padded, and
        // adding the square of 0 changes nothing); only items entirely below d4 are free of the fused tail
        c.nfull = c.d4 >> 7;
        c.tail_loads = (d - 128u * c.nfull + 15u) >> 4;
    } else {
        const uint32_t d_eff = d & ~15u;          // the SSE loop drops the last d mod 16 dims
        c.nfull = d_eff >> 7;
        c.tail_loads = (d_eff & 127u) >> 4;       // one load per lane = 4 elements of its chain = 16 dims of the row
    }
    c.nitems = c.nfull + (c.tail_loads ? 1u : 0u);
    return c;
}

// The lane's 2 x 16 bytes of each of 4 lines.
// NLD: the loads a full item consists of (8, or fewer in a kernel built for one row shape)
template <bool SCALAR, bool FULL, int NLD = 8>
__device__ __forceinline__ void issue_item(const float* p, uint32_t nld, float4 (&v)[8])
{
#pragma unroll
    for (int t = 0; t < NLD; t++)
        if (FULL || (uint32_t)t < nld) v[t] = *reinterpret_cast<const float4*>(p + 32 * (t >> 1) + 4 * (t & 1));
}

// A kernel built for one row shape keeps the lane's pieces of the query in registers for the whole search (all six
// of a 96-dim row; the first two of a 128-dim row, whose kernels have less room), so that a distance batch does not
// wait for them to come out of LDS piece by piece.
// (Byte rows of 128 dims: all eight pieces, in the byte rows' own order of the query copy; those kernels have the room.)
#ifndef HNSW_QK8
#define HNSW_QK8 2       // pieces of a 128-dim float query a lane keeps in registers (0, 2 and 4 measure within 1 %; 8 spills in the bucket-form kernels)
#endif
// RK: 0 = float rows, 1 = byte rows read through the float arithmetic, 2 = byte rows through integer dot products (the
// lane's 32 bytes of each 128-dim block of the query as packed bytes, plus the query's squared norm).
template <int NL, int RK = 0>
struct QRegs {
    // (the scalar-order kernels, NL = 7, read the query from LDS)
    static constexpr int K = RK == 2 ? 0 : (RK != 0 ? (NL == 8 ? 8 : 0) : (NL == 6 ? 6 : (NL == 8 ? HNSW_QK8 : 0)));
    float4 q[K ? K : 1];
    static constexpr int KB = RK == 2 && NL == 8 ? 2 : 0;
    uint4 qb[KB ? KB : 1];
    uint32_t qq;
};
// Kernels whose QRegs hold every piece of the query (float rows of 96 used dims, SSE order, one wavefront per query): the
// query's copy in LDS is dead after the registers are loaded, and the tables take its place.
__host__ __device__ constexpr bool query_overlaid(int NL, int ROWS, bool SCALAR, int NBUF) { return NL == 6 && ROWS == 0 && !SCALAR && NBUF == 2; }
// which of the three a kernel's ROWS parameter means (ROWS: 0 float rows, 1 exact byte rows, 2 quantised byte rows,
// 3 exact byte rows through integer dot products)
__host__ __device__ constexpr int row_kind(int ROWS) { return ROWS == 0 ? 0 : (ROWS >= 3 ? 2 : 1); }
__host__ __device__ constexpr bool rows_quantised(int ROWS) { return ROWS == 2 || ROWS == 4; }
template <int NL, int RK>
__device__ __forceinline__ void qregs_load(QRegs<NL, RK>& qr, const float* qs, int lane)
{
#pragma unroll
    for (int t = 0; t < QRegs<NL, RK>::K; t++)
        qr.q[t] = *reinterpret_cast<const float4*>(RK != 0 ? qs + 32 * (lane & 3) + 4 * t : qs + 8 * (lane & 3) + 32 * (t >> 1) + 4 * (t & 1));
#pragma unroll
    for (int t = 0; t < QRegs<NL, RK>::KB; t++)
        qr.qb[t] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(qs) + 32 * (lane & 3) + 16 * t);
}

// SSE order, lane j of the row's quad: p / qp point at this lane's 8 floats of the item's first line.
// IP: the chain accumulates the products q_i * x_i instead of the squared differences (inner-product metric).
// QK: the first QK pieces of the query come from registers (qk), the others from LDS.
template <bool FULL, bool IP, int NLD = 8, int QK = 0>
__device__ __forceinline__ void consume_item_sse(const float* qp, uint32_t nld, const float4 (&v)[8], float& a, const float4 (&qk)[QK ? QK : 1])
{
#pragma unroll
    for (int t = 0; t < NLD; t++)
        if (FULL || (uint32_t)t < nld) {
            const float4 q = t < QK ? qk[t] : *reinterpret_cast<const float4*>(qp + 32 * (t >> 1) + 4 * (t & 1));
            if constexpr (IP) {
                a = __builtin_fmaf(v[t].x, q.x, a); a = __builtin_fmaf(v[t].y, q.y, a);
                a = __builtin_fmaf(v[t].z, q.z, a); a = __builtin_fmaf(v[t].w, q.w, a);
                continue;
            }
            // (plain subtracts and fused multiply-adds: the packed two-element f32 instructions cost several times the
            // issue time of the two plain ones they replace on this chip; the file is built with -fno-slp-vectorize)
            const float dx = v[t].x - q.x, dy = v[t].y - q.y, dz = v[t].z - q.z, dw = v[t].w - q.w;
            a = __builtin_fmaf(dx, dx, a); a = __builtin_fmaf(dy, dy, a);
            a = __builtin_fmaf(dz, dz, a); a = __builtin_fmaf(dw, dw, a);
        }
}

// Byte rows, SSE order: an item is one 128-dim block, 2 x 16 bytes per lane; a dword holds 4
// consecutive elements of the lane's chain (16 dims of the row).
__device__ __forceinline__ void issue_item_u8(const uint8_t* p, uint4 (&v)[2])
{
    v[0] = *reinterpret_cast<const uint4*>(p);
    v[1] = *reinterpret_cast<const uint4*>(p + 16);
}

__device__ __forceinline__ uint32_t u4_get(const uint4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

template <bool FULL, int QK = 0>
__device__ __forceinline__ void consume_item_u8(const float* qp, uint32_t nwords, const uint4 (&v)[2], float& a, const float4 (&qk)[QK ? QK : 1])
{
#pragma unroll
    for (int w = 0; w < 8; w++)
        if (FULL || (uint32_t)w < nwords) {
            const float4 q = w < QK ? qk[w] : *reinterpret_cast<const float4*>(qp + 4 * w);
            const uint32_t word = u4_get(v[w >> 2], w & 3);
            const float dx = (float)(word & 0xFFu) - q.x, dy = (float)((word >> 8) & 0xFFu) - q.y;
            const float dz = (float)((word >> 16) & 0xFFu) - q.z, dw = (float)(word >> 24) - q.w;
            a = __builtin_fmaf(dx, dx, a); a = __builtin_fmaf(dy, dy, a);
            a = __builtin_fmaf(dz, dz, a); a = __builtin_fmaf(dw, dw, a);
        }
}

__device__ __forceinline__ float f4_get(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// Scalar order over the four lanes of a row: the running sum of the reference's plain loop visits the dims in index
// order, and dim 4k + j of a line is the k-th element of lane j.  `x` is the rotating sum: every wave instruction
// `x = x(from the lane before, in the quad) + mine` advances the chain by one dim in the lane whose turn it is (the
// other three lanes compute values that are overwritten before anybody reads them), so after the four
// instructions of element k the sum over dims .. 4k+3 sits in lane 3, where lane 0 picks it up for element k+1.
// Squares are rounded before the add for dims below d4 and fused for the last d % 4 (what g++ 11 emits for that loop):
// `kf` is the element index (dim / 4, counted over the whole row) of that fused group, or all ones when d % 4 == 0;
// its unused lanes hold padding zeros, which leave the sum as it is either way.
__device__ __forceinline__ float quad_from_previous_lane(float x)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x93, 0xF, 0xF, false));   // quad_perm [3,0,1,2]
}

// FULL: all NLD loads of the item, no fused group in it.  Otherwise `nld` loads; FUSED: the item holds the fused group.
template <bool FULL, int NLD = 8, bool FUSED = false>
__device__ __forceinline__ void consume_item_scalar(const float* qp, uint32_t k0, uint32_t kf, uint32_t nld, const float4 (&v)[8], float& x)
{
#pragma unroll
    for (int t = 0; t < NLD; t++)
        if (FULL || (uint32_t)t < nld) {
            const float4 q = *reinterpret_cast<const float4*>(qp + 32 * (t >> 1) + 4 * (t & 1));
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float tt = f4_get(v[t], c) - f4_get(q, c);
                if (!FUSED || k0 + 4 * t + c != kf) {
                    const float sq = tt * tt;
#pragma unroll
                    for (int j = 0; j < 4; j++) x = quad_from_previous_lane(x) + sq;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) x = __builtin_fmaf(tt, tt, quad_from_previous_lane(x));
                }
            }
        }
}

// NL: 0, or the row shape the kernel is built for: the number of 16-byte loads per lane that make up a row of at most
// 128 dims (8: 128 dims, 6: 96 dims, 7: 100 dims in scalar order), or 100 F + T for a row of F whole 128-dim items and
// a last one of T loads (704: 960 dims).
template <bool SCALAR, int NBUF, bool IP = false, int SUB = -1, int NL = 0>
__device__ __forceinline__ void batch_distance(const DistCtx& c, const float* const (&row)[RowMap<SCALAR>::kSub],
                                               const bool (&act)[RowMap<SCALAR>::kSub], float (&out)[RowMap<SCALAR>::kSub], int lane,
                                               const QRegs<NL>& qr = QRegs<NL>())
{
    typedef RowMap<SCALAR> RM;
    // (a kernel built for one row shape knows the items at compile time — NL < 100: one item of NL loads; NL = 100 F + T: F
    // whole items and a last one of T loads — so the loops below unroll into straight-line code without a test per load)
    static_assert(NL == 0 || NL >= 100 || (NBUF == 2 && SUB < 0), "single-item row-shape kernels run one wavefront per query");
    constexpr int kNF = NL >= 100 ? NL / 100 : 0;          // whole items before the last one
    constexpr int kTL = NL >= 100 ? NL % 100 : NL;         // loads of the last (or only) item
    constexpr int NLD = kTL ? kTL : 8;
    constexpr int kUnroll = NL >= 100 ? 32 : 1;            // (a multi-item shape: the item loop unrolls completely)
    const uint32_t ni = NL ? (uint32_t)(kNF + (kTL ? 1 : 0)) : c.nitems, nf = NL ? (uint32_t)kNF : c.nfull, tl = NL ? 0u : c.tail_loads;
    const float* qs = c.qs + 8 * (lane & 3);
    const uint32_t kf = c.kf;
    float acc[RM::kSub];
#pragma unroll
    for (int s = 0; s < RM::kSub; s++) acc[s] = 0.0f;
    float4 buf[NBUF][8];
    // item i of the sequence: the sub-batches take turns, item i / kSub of sub-batch i % kSub
    const uint32_t total = (SUB < 0 ? RM::kSub : 1) * ni;
    auto issue = [&](float4 (&v)[8], uint32_t i) {
        const uint32_t s = SUB < 0 ? (i & 1u) : (uint32_t)SUB, g = SUB >= 0 ? i : (i >> 1);
        const float* p = (s ? row[RM::kSub - 1] : row[0]) + g * RM::kItemFloats;
        if (s ? act[RM::kSub - 1] : act[0]) { if (g < nf) issue_item<SCALAR, true>(p, 0, v); else if (NL) issue_item<SCALAR, true, NLD>(p, 0, v); else issue_item<SCALAR, false>(p, tl, v); }
    };
    auto consume = [&](const float4 (&v)[8], uint32_t i) {
        const uint32_t s = SUB < 0 ? (i & 1u) : (uint32_t)SUB, g = SUB >= 0 ? i : (i >> 1);
        const float* qp = qs + g * RM::kItemFloats;
        float a = s ? acc[RM::kSub - 1] : acc[0];
        if (s ? act[RM::kSub - 1] : act[0]) {
            if (SCALAR) {
                if (g < nf) consume_item_scalar<true>(qp, 0, 0, 0, v, a);
                else if (NL) consume_item_scalar<true, NLD>(qp, 0, 0, 0, v, a);
                else if (kf - 32u * g < 32u) consume_item_scalar<false, 8, true>(qp, 32u * g, kf, tl, v, a);      // the item with the fused group
                else consume_item_scalar<false>(qp, 32u * g, kf, tl, v, a);
            }
            else {
                if (g < nf) consume_item_sse<true, IP>(qp, 0, v, a, QRegs<0>().q);
                else if (NL) consume_item_sse<true, IP, NLD, QRegs<NL>::K>(qp, 0, v, a, qr.q);
                else consume_item_sse<false, IP>(qp, tl, v, a, QRegs<0>().q);
            }
        }
        if (s) acc[RM::kSub - 1] = a; else acc[0] = a;
    };
    // (A pipeline in which every step issues the same number of loads, so that the wait before a piece is used
    // need not cover the loads issued after it, was measured on 960-dim rows: slower, by the loads it adds to keep
    // the count fixed.)
#pragma unroll
    for (int b = 0; b < NBUF; b++)
        if ((uint32_t)b < total) issue(buf[b], b);
#pragma unroll kUnroll
    for (uint32_t i = 0; i < total; i += NBUF) {
#pragma unroll
        for (int b = 0; b < NBUF; b++)
            if (i + b < total) {
                consume(buf[b], i + b);
                if (i + b + NBUF < total) issue(buf[b], i + b + NBUF);
            }
    }
#pragma unroll
    for (int s = 0; s < RM::kSub; s++) {
        if (SCALAR) {
            // the running sum ended its last round in lane 3 of the quad
            out[s] = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(acc[s]), 0xFF, 0xF, 0xF, false));
            continue;
        }
        // sums 0..3 of a row sit in the 4 lanes of its quad: ((s0 + s1) + s2) + s3
        const int ai = (int)__float_as_uint(acc[s]);
        const float a0 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0x00, 0xF, 0xF, false));
        const float a1 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0x55, 0xF, 0xF, false));
        const float a2 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0xAA, 0xF, 0xF, false));
        const float a3 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0xFF, 0xF, 0xF, false));
        const float sum = ((a0 + a1) + a2) + a3;
        // inner-product metric: 1 - q.x, held at 0 from below (the heaps and W order distances by their bit patterns);
        // on unit vectors this is half the squared L2 distance
        out[s] = IP ? __builtin_fmaxf(1.0f - sum, 0.0f) : sum;
    }
}

// Scalar order on TWO lanes per row.  The running sum of the reference's plain loop (src/hnsw.h:387-400) is one sequential
// chain of d additions per row, so a wave instruction can advance at most as many rows as it has lanes working on
// different rows: with four lanes per row (the form above) a list of 32 rows takes two passes in each of which three
// lanes of four compute values nobody reads.  Here the two rows of a quad are served at once: lanes j and j + 2 of quad r
// hold row 2r + (j & 1) (lane j & 1 is the lane that holds the row's id in merged form), the lower lane ("A") the
// elements that feed lane sums 0 and 1 of the SSE layout — the first 64 bytes of every stored 128-byte line — the upper
// ("B") those of sums 2 and 3.  Dims 4k, 4k+1 of a line are then A's k-th elements, dims 4k+2, 4k+3 B's, and the running
// sum alternates: two additions in A, over to B (a DPP operand: quad_perm [2,3,0,1]), two additions in B, back to A.  One
// instruction stream serves both: x = partner(x) + p0; x = x + p1, twice per k — the lane whose turn it is not computes
// values that its next "partner(x)" overwrites.  Per 4 dims of 32 rows: 2 sub + 2 mul + 4 add = 8 instructions, against
// 12 for the four-lane form (two passes of 6); a line is touched by four load instructions of 32 bytes per row.
// `rowp`: this lane's row + 16 floats for a B lane; `act`: the row takes part.  The result is returned in the A lane.
// NL = 7: rows of 97..112 dims without a fused tail (d = 100): three whole lines and the first four k-steps of the fourth
// (what lies past d there is zero padding in row and query: adding the square of 0 leaves the sum as it is).
__device__ __forceinline__ float quad_partner(float x)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
}

template <int KSTEPS, bool FUSED>
__device__ __forceinline__ void consume_line_scalar2(const float* qp /* the lane's 16 floats of the query's line */, uint32_t k0 /* element index of the line's first k-step */,
                                                     uint32_t kf, const float4 (&v)[4], float& x)
{
    // v[0], v[1]: the lane's first sum's elements k = 0..7; v[2], v[3]: its second sum's
#pragma unroll
    for (int k = 0; k < KSTEPS; k++) {
        const float4 qa = *reinterpret_cast<const float4*>(qp + 4 * (k >> 2)), qb = *reinterpret_cast<const float4*>(qp + 8 + 4 * (k >> 2));
        const float t0 = f4_get(v[k >> 2], k & 3) - f4_get(qa, k & 3);
        const float t1 = f4_get(v[2 + (k >> 2)], k & 3) - f4_get(qb, k & 3);
        if (!FUSED || k0 + (uint32_t)k != kf) {
            const float p0 = t0 * t0, p1 = t1 * t1;
            x = quad_partner(x) + p0; x = x + p1;
            x = quad_partner(x) + p0; x = x + p1;
        } else {
            x = __builtin_fmaf(t0, t0, quad_partner(x)); x = __builtin_fmaf(t1, t1, x);
            x = __builtin_fmaf(t0, t0, quad_partner(x)); x = __builtin_fmaf(t1, t1, x);
        }
    }
}

__device__ __forceinline__ void issue_line_scalar2(const float* p, float4 (&v)[4])
{
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = *reinterpret_cast<const float4*>(p + 4 * t);
}

template <int NL>
__device__ __forceinline__ float batch_distance_scalar2(const DistCtx& c, const float* rowp, bool act, int lane)
{
    static_assert(NL == 0 || NL == 7, "scalar order: the kernel for any row length, or the one for 97..112 dims");
    const float* qp = c.qs + 16 * ((lane >> 1) & 1);
    float x = 0.0f;
    if constexpr (NL == 7) {
        float4 b0[4], b1[4], b2[4], b3[4];
        if (act) {
            issue_line_scalar2(rowp, b0); issue_line_scalar2(rowp + 32, b1); issue_line_scalar2(rowp + 64, b2);
            b3[0] = *reinterpret_cast<const float4*>(rowp + 96); b3[2] = *reinterpret_cast<const float4*>(rowp + 96 + 8);
            b3[1] = b3[0]; b3[3] = b3[2];
        }
        if (act) {
            consume_line_scalar2<8, false>(qp, 0, 0, b0, x);
            consume_line_scalar2<8, false>(qp + 32, 0, 0, b1, x);
            consume_line_scalar2<8, false>(qp + 64, 0, 0, b2, x);
            consume_line_scalar2<4, false>(qp + 96, 0, 0, b3, x);
        }
    } else {
        // any row length: lines of 32 dims, four in flight; the k-step that holds the last d % 4 dims is fused
        // (c.kf, counted in k-steps over the whole row), and nothing past the last line that holds a dim is read
        const uint32_t nlines = (c.d4 + (c.kf != 0xFFFFFFFFu ? 4u : 0u) + 31u) >> 5;
        float4 buf[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++)
            if ((uint32_t)b < nlines && act) issue_line_scalar2(rowp + 32 * b, buf[b]);
        for (uint32_t l = 0; l < nlines; l += 4) {
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (l + b < nlines) {
                    if (act) {
                        if (c.kf - 8u * (l + b) < 8u) consume_line_scalar2<8, true>(qp + 32 * (l + b), 8u * (l + b), c.kf, buf[b], x);
                        else consume_line_scalar2<8, false>(qp + 32 * (l + b), 0, 0, buf[b], x);
                        if (l + b + 4 < nlines) issue_line_scalar2(rowp + 32 * (l + b + 4), buf[b]);
                    }
                }
        }
    }
    // After a whole k-step the sum sits in the B lane: one more exchange hands it to the A lane, which is the lane that
    // holds the row in merged form and the only one whose distance is read (the B lane receives a value nobody uses).
    // Unconditional on purpose: a cross-lane read under a lane-dependent condition is executed with the source lanes
    // switched off and returns nothing.
    return quad_partner(x);
}

// The same for byte rows (SSE order only): row[s] points at this lane's 32 bytes of the row's first block.
// NL = 8: a kernel built for rows of 128 dims (one block, known at compile time: the loop below unrolls into straight-line code).
template <int NL = 0>
__device__ __forceinline__ void batch_distance_u8(const DistCtx& c, const uint8_t* const (&row)[2], const bool (&act)[2],
                                                  float (&out)[2], int lane, const QRegs<NL, 1>& qr = QRegs<NL, 1>())
{
    static_assert(NL == 0 || NL == 8, "byte rows: the kernel for any row length, or the one for 128 dims");
    const uint32_t ni = NL ? 1u : c.nitems, nf = NL ? 1u : c.nfull, tl = NL ? 0u : c.tail_loads;
    const float* qs = c.qs + 32 * (lane & 3);
    float acc[2] = {0.0f, 0.0f};
    uint4 A[2], B[2];
    const uint32_t total = 2 * ni;
    auto issue = [&](uint4 (&v)[2], uint32_t i) {
        const uint32_t s = i >= ni ? 1u : 0u, g = i - s * ni;
        if (s ? act[1] : act[0]) issue_item_u8((s ? row[1] : row[0]) + g * 128u, v);
    };
    auto consume = [&](const uint4 (&v)[2], uint32_t i) {
        const uint32_t s = i >= ni ? 1u : 0u, g = i - s * ni;
        float a = s ? acc[1] : acc[0];
        if (s ? act[1] : act[0]) { if (NL || g < nf) consume_item_u8<true, QRegs<NL, 1>::K>(qs + g * 128u, 0, v, a, qr.q); else consume_item_u8<false>(qs + g * 128u, tl, v, a, QRegs<0, 1>().q); }
        if (s) acc[1] = a; else acc[0] = a;
    };
    issue(A, 0);
    issue(B, 1);
    for (uint32_t i = 0; i < total; i += 2) {
        consume(A, i);
        if (i + 2 < total) issue(A, i + 2);
        consume(B, i + 1);
        if (i + 3 < total) issue(B, i + 3);
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int ai = (int)__float_as_uint(acc[s]);
        const float a0 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0x00, 0xF, 0xF, false));
        const float a1 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0x55, 0xF, 0xF, false));
        const float a2 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0xAA, 0xF, 0xF, false));
        const float a3 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, ai, 0xFF, 0xF, 0xF, false));
        out[s] = ((a0 + a1) + a2) + a3;
    }
}

// Byte rows through integer dot products.  When every stored value and every element of the query is an integer in 0..255
// and the largest possible distance, 255^2 per used dim, stays below 2^24, every partial sum of the reference's four float
// chains (src/hnsw.h:349-380) is an integer that a float holds exactly: no step of the chain rounds, so the result is the
// exact integer sum of (q_i - x_i)^2 whatever the order of the additions, and
//     sum (q - x)^2 = sum q^2 - 2 sum q x + sum x^2
// computed in integers and converted once has the same bits.  v_dot4_u32_u8 takes four products per instruction: a dword
// of the row costs two instructions (q.x and x.x) instead of four conversions, four subtractions and four multiply-adds.
// row[s] points at this lane's 32 bytes of the row's first 128-dim block; the lane's partial sums meet in the quad.
// `qbytes`: the query as packed bytes in LDS, in the byte rows' order.
template <int NL = 0>
__device__ __forceinline__ void batch_distance_dot4(const DistCtx& c, const uint8_t* const (&row)[2], const bool (&act)[2],
                                                    float (&out)[2], int lane, const QRegs<NL, 2>& qr)
{
    static_assert(NL == 0 || NL == 8, "byte rows: the kernel for any row length, or the one for 128 dims");
    const uint32_t ni = NL ? 1u : c.nitems, nf = NL ? 1u : c.nfull, tl = NL ? 0u : c.tail_loads;
    const uint8_t* qbytes = reinterpret_cast<const uint8_t*>(c.qs) + 32 * (lane & 3);
    int32_t acc[2] = {0, 0};
    uint4 A[2], B[2];
    const uint32_t total = 2 * ni;
    auto issue = [&](uint4 (&v)[2], uint32_t i) {
        const uint32_t s = i >= ni ? 1u : 0u, g = i - s * ni;
        if (s ? act[1] : act[0]) issue_item_u8((s ? row[1] : row[0]) + g * 128u, v);
    };
    auto consume = [&](const uint4 (&v)[2], uint32_t i) {
        const uint32_t s = i >= ni ? 1u : 0u, g = i - s * ni;
        if (s ? act[1] : act[0]) {
            uint32_t qx = 0, xx = 0;
            uint4 q[2];
            if constexpr (NL != 0) { q[0] = qr.qb[0]; q[1] = qr.qb[1]; }
            else { q[0] = *reinterpret_cast<const uint4*>(qbytes + g * 128u); q[1] = *reinterpret_cast<const uint4*>(qbytes + g * 128u + 16); }
            const uint32_t nw = (NL || g < nf) ? 8u : tl;
#pragma unroll
            for (int w = 0; w < 8; w++)
                if (NL || (uint32_t)w < nw) {
                    const uint32_t x = u4_get(v[w >> 2], w & 3);
                    qx = __builtin_amdgcn_udot4(x, u4_get(q[w >> 2], w & 3), qx, false);
                    xx = __builtin_amdgcn_udot4(x, x, xx, false);
                }
            const int32_t t = (int32_t)xx - 2 * (int32_t)qx;
            if (s) acc[1] += t; else acc[0] += t;
        }
    };
    issue(A, 0);
    issue(B, 1);
    for (uint32_t i = 0; i < total; i += 2) {
        consume(A, i);
        if (i + 2 < total) issue(A, i + 2);
        consume(B, i + 1);
        if (i + 3 < total) issue(B, i + 3);
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
        int t = acc[s];
        t += __builtin_amdgcn_update_dpp(0, t, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
        t += __builtin_amdgcn_update_dpp(0, t, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
        out[s] = (float)(t + (int32_t)qr.qq);
    }
}

// Distances to the rows `id[s]` (sub-batch s; act[s]: the lane's row takes part), from the float or the byte table.
template <bool SCALAR, int ROWS, int NBUF = 2, bool IP = false, int SUB = -1, int NL = 0>
__device__ __forceinline__ void rows_distance(const DeviceGraph& g, const DistCtx& c, const uint32_t (&id)[RowMap<SCALAR>::kSub],
                                              const bool (&act)[RowMap<SCALAR>::kSub], float (&out)[RowMap<SCALAR>::kSub], int lane,
                                              const QRegs<NL, row_kind(ROWS)>& qr = QRegs<NL, row_kind(ROWS)>())
{
    typedef RowMap<SCALAR> RM;
    if constexpr (ROWS != 0) {
        static_assert(!SCALAR, "byte rows are read in SSE order only");
        const uint8_t* const table = rows_quantised(ROWS) ? g.vectors_q8 : g.vectors_u8;
        const uint8_t* row[2];
#pragma unroll
        for (int s = 0; s < 2; s++) row[s] = table + (size_t)id[s] * g.row_stride_u8 + 32u * (lane & 3);
        if constexpr (ROWS >= 3) batch_distance_dot4<NL>(c, row, act, out, lane, qr);
        else batch_distance_u8<NL>(c, row, act, out, lane, qr);
    } else {
        if constexpr (SCALAR) {
            // two lanes per row: lanes j and j + 2 of a quad serve the row that lane j & 1 holds in merged form
            // (one wavefront per query: the two-wavefront form, SUB >= 0, reads float rows in SSE order only and never gets here)
            // (plain values first: a lane-dependent choice between two array elements would be made in scratch memory)
            const bool odd = (lane & 1) != 0;
            const uint32_t id_even = id[0], id_odd = id[RM::kSub - 1];
            const bool act_even = act[0], act_odd = act[RM::kSub - 1];
            const uint32_t my_id = odd ? id_odd : id_even;
            const bool my_act = odd ? act_odd : act_even;
            const float* rowp = g.vectors + (size_t)my_id * g.row_stride + 16u * ((lane >> 1) & 1);
            const float r = batch_distance_scalar2<NL>(c, rowp, my_act, lane);
#pragma unroll
            for (int s = 0; s < RM::kSub; s++) out[s] = r;
        } else {
            const float* row[RM::kSub];
#pragma unroll
            for (int s = 0; s < RM::kSub; s++) row[s] = g.vectors + (size_t)id[s] * g.row_stride + 8u * (lane & 3);
            batch_distance<SCALAR, NBUF, IP, SUB, NL>(c, row, act, out, lane, qr);
        }
    }
}

// A lane's view of one batch of a neighbour list.  In SSE mode lane (r, j) of quad r serves row 2r
// (sub-batch 0) and row 2r + 1 (sub-batch 1); the id, freshness and distance of sub-batch s are
// kept "merged" in lane j == s of the quad, so one register and one ballot describe both, and
// lane order is list order.
template <bool SCALAR>
struct BatchLanes {
    typedef RowMap<SCALAR> RM;
    // row of the list this lane holds in merged form, or >= deg when none
    static __device__ __forceinline__ uint32_t merged_row(int lane)
    {
        const uint32_t j = lane & 3, r = (uint32_t)lane >> 2;
        return j < 2 ? 2u * r + j : 0xFFFFu;
    }
    // value of sub-batch s for every lane of the quad, from the merged register
    static __device__ __forceinline__ uint32_t spread(uint32_t merged, int s)
    {
        if (s == 0) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)merged, 0x00, 0xF, 0xF, false);   // quad_perm [0,0,0,0]
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)merged, 0x55, 0xF, 0xF, false);               // quad_perm [1,1,1,1]
    }
};

// The result list W in registers: entry e lives in register e / 64, lane e % 64, and the nW valid
// entries are kept ascending by distance.  Insert = count the entries not larger than the new one
// (one ballot per register), move the rest up by one (a wave-wide DPP shift, lane 63 carried into
// the next register) and drop the new entry into the gap; with W full the largest entry falls
// off the end.  Non-negative floats order as their bit patterns.
template <int WR>
__device__ __forceinline__ void w_sorted_insert(uint32_t (&wd)[WR], uint32_t (&wi)[WR], uint32_t nW, uint32_t db, uint32_t id, int lane)
{
    uint32_t pos = 0;
#pragma unroll
    for (int w = 0; w < WR; w++) {
        const uint32_t left = nW > 64u * w ? nW - 64u * w : 0u;
        const unsigned long long valid = left >= 64u ? ~0ull : (1ull << left) - 1ull;
        pos += (uint32_t)__popcll(__ballot(wd[w] <= db) & valid);
    }
#pragma unroll
    for (int w = WR - 1; w >= 0; w--) {
        if (WR > 1 && 64u * (uint32_t)(w + 1) <= pos) break;     // this register and the ones below hold entries before the gap only
        uint32_t sd = (uint32_t)__builtin_amdgcn_update_dpp((int)wd[w], (int)wd[w], 0x138, 0xF, 0xF, false);   // wave_shr:1
        uint32_t si = (uint32_t)__builtin_amdgcn_update_dpp((int)wi[w], (int)wi[w], 0x138, 0xF, 0xF, false);
        if (w > 0) {
            const uint32_t cd = __builtin_amdgcn_readlane(wd[w - 1], 63), ci = __builtin_amdgcn_readlane(wi[w - 1], 63);
            if (lane == 0) { sd = cd; si = ci; }
        }
        const uint32_t e = 64u * w + (uint32_t)lane;
        if (e > pos) { wd[w] = sd; wi[w] = si; }
        if (e == pos) { wd[w] = db; wi[w] = id; }
    }
}

template <int WR>
__device__ __forceinline__ uint32_t w_entry_dist(const uint32_t (&wd)[WR], uint32_t e)
{
    uint32_t v = 0;
#pragma unroll
    for (int w = 0; w < WR; w++)
        if ((e >> 6) == (uint32_t)w) v = __builtin_amdgcn_readlane(wd[w], e & 63u);
    return v;
}

// Maximum of an unsigned value over the 64 lanes (same DPP steps as wave_min_u32).  The lanes a step does not write
// take 0, the identity of the maximum, so that every step can be one v_max_u32 with a DPP operand.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
    uint32_t y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false); x = x > y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false); x = x > y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false); x = x > y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false); x = x > y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); x = x > y ? x : y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); x = x > y ? x : y;
    return __builtin_amdgcn_readlane(x, 63);
}

// W over more than one register per lane (ef > 64) is kept as an UNORDERED set: keeping several registers sorted
// costs a shift through all of them per admitted neighbour, while the search itself only ever needs the largest
// distance f and the size.  A full W admits by overwriting an entry that holds the largest distance (its bits
// are f's) and taking the maximum again; the order is made once, when the results are written.  Slots past the
// current size hold distance bits 0, which never win a maximum.
template <int WR>
__device__ __forceinline__ void w_set_append(uint32_t (&wd)[WR], uint32_t (&wi)[WR], uint32_t slot, uint32_t db, uint32_t id, int lane)
{
#pragma unroll
    for (int w = 0; w < WR; w++)
        if (64u * w + (uint32_t)lane == slot) { wd[w] = db; wi[w] = id; }
}

// Replaces one entry whose distance bits are `fb` (the current maximum of a full W) and returns the new maximum.
// Straight-line: one comparison per register, the first register that holds `fb` and the first such lane in it are
// picked with scalar arithmetic, and every lane selects between its old entry and the new one.  (`fb` is never 0
// here - nothing is admitted below a maximum of 0 - so the unused slots past the end of W, which hold 0, never match.)
template <int WR>
__device__ __forceinline__ uint32_t w_set_replace_max(uint32_t (&wd)[WR], uint32_t (&wi)[WR], uint32_t fb, uint32_t db, uint32_t id, int lane)
{
    bool found = false;                                       // wave-uniform
#pragma unroll
    for (int w = 0; w < WR; w++) {
        const unsigned long long b = __ballot(wd[w] == fb);
        const bool here = !found && b != 0ull;
        const bool mine = here && lane == __builtin_ctzll(b | (1ull << 63));
        wd[w] = mine ? db : wd[w];
        wi[w] = mine ? id : wi[w];
        found = found || b != 0ull;
    }
    uint32_t m = wd[0];
#pragma unroll
    for (int w = 1; w < WR; w++) m = m > wd[w] ? m : wd[w];
    return wave_max_u32(m);
}

// WR = 0: the result list W is the libstdc++ max-heap in LDS, maintained exactly as the
//         reference does (so even the choice among several equally distant worst entries
//         is the reference's).
// WR > 0: W is a sorted list, WR entries per lane in registers (ef <= 64*WR).  Only the
//         largest distance f and the size of W steer the search, and both evolve
//         identically; the one observable difference is which of several entries with
//         exactly the worst distance is dropped.
// ROWS = 0: rows are read from the float table.
// ROWS = 1: from the exact byte table (DeviceGraph::vectors_u8): same arithmetic, same results.
// ROWS = 4: the quantised byte table through integer dot products (the query rounded to the table's grid), results re-scored like ROWS = 2.
// ROWS = 2: from the quantised byte table (vectors_q8) with the query mapped to the same grid, so the
//           search runs on approximate distances; its ef results are then re-scored with the exact
//           float distance and sorted by it.
// BUCKET: the visited set is in its bucket form (16-bit slots in small regions, overflow table, spill tables).
// NBUF: row pieces (8 x 16-byte loads per lane each) in flight during a distance batch.  2 is all a 128-dim row
//       has to offer; the deep form (6) is for wide rows in a batch too small to fill the chip, where a CU holds
//       only a few queries, registers are plentiful and nothing but a longer queue of loads hides the memory latency.
//       The deep form also runs TWO wavefronts per query (workgroups of 128): a row is one serial chain of loads and
//       fused multiply-adds per lane, a distance batch gives every lane two rows, and the second wavefront takes the
//       second of them (GIST-shaped 1M x 960, 1000 queries: 0.79 -> 1.06 M queries/s).
// IP: inner-product metric (distance 1 - q.x) instead of squared L2; float rows, SSE order only.
// (Except for the deep-prefetch form the kernels are held to 128 registers and pinned at four wavefronts per SIMD = 16
// queries per CU.)
// NL: 0 = rows of any length; 8 / 6 = a kernel for rows of exactly 128 / 96 used dims (float rows, SSE order; 8 also for byte
//     rows), 7 = for the scalar order on rows of 97..112 dims without a fused tail (d = 100), 704 = the deep form for rows of
//     960 used dims (7 whole items + 4 loads); their distance batches are straight-line code.
// WPS: wavefronts per SIMD the kernel is pinned at (one-wavefront workgroups: four times as many queries per CU).  4 for
//      all but the kernels that read byte rows through integer dot products, which need half the registers of the others
//      and also exist pinned at 6 for ef <= 64, where a query's LDS can be small enough for 24 queries per CU.  (A kernel
//      is pinned because one that may run more wavefronts per SIMD than LDS lets the CU hold fills the SIMDs unevenly.)
// PACKC: candidate-heap entries of four bytes (COrderPacked); float rows, int32 keys only.
template <bool SCALAR, int WR, int ROWS, bool BUCKET, int NBUF, bool IP, int NL, int WPS = 4, bool PACKC = false>
__global__ __launch_bounds__(NBUF == 2 ? 64 : 128) __attribute__((amdgpu_waves_per_eu(NBUF == 2 ? WPS : 1, NBUF == 2 ? WPS : 2))) void hnsw_search_kernel(SearchArgs a)
{
    typedef RowMap<SCALAR> RM;
    typedef BatchLanes<SCALAR> BL;
    typedef typename std::conditional<PACKC, uint32_t, uint2>::type CE;          // an entry of the candidate heap
    typedef typename std::conditional<PACKC, COrderPacked, COrder>::type COrd;
    static_assert(!PACKC || (ROWS == 0 && !IP && NBUF == 2 && WR != 0), "packed candidate entries: float rows, squared L2, W in registers");
    auto c_entry = [](uint32_t key, uint32_t id) -> CE { if constexpr (PACKC) return (key << 24) | id; else return make_uint2(key, id); };
    extern __shared__ __align__(16) uint32_t smem[];
    // The deep form (wide rows, few queries per CU) runs two wavefronts per query: the first walks the graph and takes the
    // rows of sub-batch 0 of every distance batch, the second does nothing but the rows of sub-batch 1 — a row is a serial
    // chain of loads and fused multiply-adds per lane, so halving the rows per lane halves that chain.  They talk through
    // a mailbox at the end of the query's LDS: [0,64) the batch's ids in merged form, [64,128) its freshness flags,
    // [128,192) the second wavefront's distances, [192] the command (1 = a batch, 0 = the query is over).
    constexpr bool kHelper = NBUF != 2;
    static_assert(!kHelper || (!SCALAR && ROWS == 0), "the two-wavefront form reads float rows in SSE order");
    const int lane = threadIdx.x & 63;
    uint32_t* const mail = smem + a.mail_off;
    if (kHelper && threadIdx.x >= 64) {
        for (;;) {
            __syncthreads();                                 // a request (or the end) is in the mailbox
            if (mail[192] == 0u) break;
            const uint32_t my = mail[lane];
            const uint32_t fresh = mail[64 + lane];
            uint32_t rid[RM::kSub]; bool act[RM::kSub]; float dj[RM::kSub];
#pragma unroll
            for (int s = 0; s < RM::kSub; s++) { rid[s] = BL::spread(my, s); act[s] = BL::spread(fresh, s) != 0; }
            const DistCtx dc = make_dist_ctx<SCALAR>(reinterpret_cast<float*>(smem), a.g.d);
            rows_distance<SCALAR, ROWS, NBUF, IP, 1, (kHelper ? NL : 0)>(a.g, dc, rid, act, dj, lane);
            if ((lane & 3) == 1) mail[128 + lane] = __float_as_uint(dj[RM::kSub - 1]);
            __syncthreads();                                 // the distances are in the mailbox
        }
        return;
    }
    // (first wavefront, two-wavefront form) hand sub-batch 1 over, compute sub-batch 0, merge
    auto batch_with_helper = [&](uint32_t my, bool fresh, const DistCtx& dc) -> float {
        mail[lane] = my; mail[64 + lane] = fresh ? 1u : 0u;
        if (lane == 0) mail[192] = 1u;
        __syncthreads();
        uint32_t rid[RM::kSub]; bool act[RM::kSub]; float dj[RM::kSub];
#pragma unroll
        for (int s = 0; s < RM::kSub; s++) { rid[s] = BL::spread(my, s); act[s] = BL::spread((uint32_t)fresh, s) != 0; }
        rows_distance<SCALAR, ROWS, NBUF, IP, 0, (kHelper ? NL : 0)>(a.g, dc, rid, act, dj, lane);
        __syncthreads();
        return (lane & 3) == 1 ? __uint_as_float(mail[128 + lane]) : dj[0];
    };
    auto release_helper = [&]() {
        if (kHelper) { if (lane == 0) mail[192] = 0u; __syncthreads(); }
    };
    const uint32_t mrow = BL::merged_row(lane);
    uint32_t qi = blockIdx.x;
    if (a.qmap) qi = a.qmap[qi];

    // LDS: query | W (exact-heap mode only) | C | visited.  The register mode needs W in LDS only for the
    // final sort, when C is dead, so it borrows C's space.
    // A kernel that keeps the whole query in registers (query_overlaid) lets W / C / visited start where the query's
    // copy was: that copy is dead once the registers are loaded.
    constexpr bool kOverlayQ = query_overlaid(NL, ROWS, SCALAR, NBUF);
    float* qs = reinterpret_cast<float*>(smem);
    uint2* W = reinterpret_cast<uint2*>(smem + (kOverlayQ ? 0u : a.qpad));
    CE* C = reinterpret_cast<CE*>(W + (WR == 0 ? a.cap_w : 0u));
    uint32_t* hash = reinterpret_cast<uint32_t*>(C + a.cap_c);
    const uint32_t H = a.hash_size, vis_rb = a.vis_rb;
    const uint32_t primary_words = vis_rb == 0xFFFFFFFFu ? H : H >> 1;
    const uint32_t hash_words = primary_words + (BUCKET ? a.ovf_size : 0u);
    const VisitedSet vs = {hash, H, vis_rb, a.vis_chunks, a.vis_regions, a.vis_magic, hash + primary_words, a.ovf_size, a.spill_slots};
    uint32_t* spill = nullptr;           // this query's spill table, once it needed one (layer 0 only)
    uint32_t n_spill = 0;
    const uint32_t spill_limit = a.spill_slots - (a.spill_slots >> 3);
    // the probing forms are kept at most 7/8 full; the bucket form fills its regions completely and keeps its
    // overflow table at most 7/8 full instead
    const uint32_t visit_limit = BUCKET ? 0xFFFFFFFFu : H - (H >> 3);
    const uint32_t ovf_limit = BUCKET ? a.ovf_size - (a.ovf_size >> 3) : 0xFFFFFFFFu;
    uint32_t n_ovf = 0;
    bool vis_ovf = false, used_ovf = false;
    const DistCtx dc = make_dist_ctx<SCALAR>(qs, a.g.d);
    const bool float_keys = (a.flags & HNSW_AMD_KEYS_FLOAT) != 0;
    const uint32_t ef = a.ef;

    // query into LDS in stored form (zero padded), visited set empty
    QRegs<NL, row_kind(ROWS)> qr;
    if constexpr (ROWS >= 3) {
        // as packed bytes (a.qpad counts dwords of four); a query with an element that is not an integer in 0..255 cannot
        // take the integer arithmetic: it is handed back like a query that outgrew its scratch (the re-run uses floats)
        const float* q = a.queries + (size_t)qi * a.g.d;
        const uint32_t d_used = a.g.d & ~15u;
        uint8_t* qb = reinterpret_cast<uint8_t*>(qs);
        bool inexact = false;
        uint32_t sq = 0;
        for (uint32_t i = lane; i < 4u * a.qpad; i += 64) {
            float v = i < d_used ? q[i] : 0.0f;
            if (ROWS == 4 && i < d_used) v = __builtin_rintf((v - a.g.q8_lo) * a.g.q8_inv_step);      // the query on the table's grid, rounded
            if (ROWS == 4) v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
            const uint32_t b = (uint32_t)(v >= 0.0f && v <= 255.0f ? (int)v : 0);
            inexact = inexact || (ROWS == 3 && (float)b != v);
            sq += b * b;
            qb[stored_pos_u8(i)] = (uint8_t)b;
        }
        for (uint32_t i = lane; i < hash_words; i += 64) hash[i] = kEmpty;
        if (__ballot(inexact)) {
            if (lane == 0) {
                const uint32_t at = atomicAdd(a.fail_count, 1u);
                a.fail_list[at] = qi;
                atomicAdd(&a.fail_count[6], 1u);
                if (a.out_count) a.out_count[qi] = kEmpty;
            }
            return;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += (uint32_t)__shfl_xor((int)sq, off);
        qr.qq = rfl(sq);
    } else {
        const float* q = a.queries + (size_t)qi * a.g.d;
        float qn = 0.0f;
        for (uint32_t i = lane; i < a.qpad; i += 64) {
            float v = i < a.g.d ? q[i] : 0.0f;
            if (PACKC) qn = __builtin_fmaf(v, v, qn);
            if (ROWS == 2 && i < a.g.d) v = (v - a.g.q8_lo) * a.g.q8_inv_step;
            qs[ROWS != 0 ? stored_pos_u8(i) : stored_pos(i)] = v;
        }
        if constexpr (PACKC) {
            // Keys must stay below 256: no row is farther from the query than (|q| + the longest row)^2.  A query for
            // which that bound does not hold is handed back (the re-run uses eight-byte entries).
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off);
            const float reach = __builtin_sqrtf(qn) * 1.001f + a.g.max_row_norm;
            if (!(reach * reach < 250.0f)) {
                if (lane == 0) {
                    const uint32_t at = atomicAdd(a.fail_count, 1u);
                    a.fail_list[at] = qi;
                    atomicAdd(&a.fail_count[6], 1u);
                    if (a.out_count) a.out_count[qi] = kEmpty;
                }
                return;
            }
        }
        if (!kOverlayQ) for (uint32_t i = lane; i < hash_words; i += 64) hash[i] = kEmpty;
    }
    LDS_SYNC();
    qregs_load(qr, qs, lane);
    if constexpr (kOverlayQ) {
        // (the registers hold the query; its place in LDS now belongs to the tables)
        LDS_SYNC();
        for (uint32_t i = lane; i < hash_words; i += 64) hash[i] = kEmpty;
        LDS_SYNC();
    }

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
    if (lane == 0) visited_insert<BUCKET>(vs, nullptr, cur_id, vis_ovf, used_ovf);
    if (BUCKET) n_ovf += __ballot(lane == 0 && used_ovf) ? 1u : 0u;
    {
        uint32_t rid[RM::kSub]; bool act[RM::kSub]; float dj[RM::kSub];
#pragma unroll
        for (int s = 0; s < RM::kSub; s++) { rid[s] = cur_id; act[s] = s == 0 && lane < RM::kLanesPerRow; }
        rows_distance<SCALAR, ROWS, NBUF, IP, -1, NL>(a.g, dc, rid, act, dj, lane, qr);
        cur_dist = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dj[0]), 0));
    }
    st.n_dist = 1;
    uint32_t nvis = 1;
    LDS_SYNC();

    // ---- greedy descent through layers L..1; visited shared by all of them
    //      (src/hnsw.cpp:403-412, 676-717) ----
    // Lists of at most 16 rows (M = 16: every upper-layer list) go one row per quad into sub-batch 0 alone: one chain of
    // loads and multiply-adds per lane instead of two half-empty ones.  Lane order is still list order.  (Not in the
    // two-wavefront form, whose second wavefront lives on sub-batch 1.)
    const bool narrow_up = !kHelper && a.g.degu <= (uint32_t)RM::kRowsPerSub;
    const uint32_t mrow_up = narrow_up ? ((lane & 3) == 0 ? (uint32_t)lane >> 2 : 0xFFFFu) : mrow;
    const uint32_t step_up = narrow_up ? (uint32_t)RM::kRowsPerSub : (uint32_t)RM::kRowsPerBatch;
    for (int l = L; l > 0 && !fail; l--) {
        const uint32_t base = a.g.up_off[l];
        bool change = true;
        while (change && !fail) {
            change = false;
            if (nvis + 64 > visit_limit || (BUCKET && n_ovf + 64 > ovf_limit)) { fail = 1; break; }
            st.n_sweeps++;
            // The sweep covers the whole list of the node current at its start; the running best moves
            // only on a strict improvement, i.e. to the first minimum of the list.
            const size_t rowi = (size_t)(base + cur_slot) * a.g.degu;
            for (uint32_t b0 = 0; b0 < a.g.degu; b0 += step_up) {
                uint32_t my_order = kEmpty, my_slot = kEmpty;           // merged form
                if (b0 + mrow_up < a.g.degu) { my_order = a.g.up_adj_order[rowi + b0 + mrow_up]; my_slot = a.g.up_adj_slot[rowi + b0 + mrow_up]; }
                const bool live = my_order != kEmpty;
                used_ovf = false;
                const bool fresh = live && visited_insert<BUCKET>(vs, nullptr, my_order, vis_ovf, used_ovf);
                if (!BUCKET && __ballot(vis_ovf)) { fail = 1; break; }
                if (BUCKET) n_ovf += (uint32_t)__popcll(__ballot(fresh && used_ovf));
                const uint32_t m = (uint32_t)__popcll(__ballot(fresh));
                st.sum_degu += (uint32_t)__popcll(__ballot(live));
                st.n_dist += m; nvis += m;
                LDS_SYNC();
                if (m == 0) continue;
                float dm;                                                                      // merged form
                if constexpr (kHelper) {
                    dm = batch_with_helper(my_order, fresh, dc);
                } else {
                    uint32_t rid[RM::kSub]; bool act[RM::kSub]; float dj[RM::kSub];
#pragma unroll
                    for (int s = 0; s < RM::kSub; s++) { rid[s] = BL::spread(my_order, s); act[s] = BL::spread((uint32_t)fresh, s) != 0; }
                    rows_distance<SCALAR, ROWS, NBUF, IP, -1, NL>(a.g, dc, rid, act, dj, lane, qr);
                    dm = (lane & 3) == 1 ? dj[RM::kSub - 1] : dj[0];
                }
                const bool mine = fresh && dm == dm;
                const uint32_t key = mine ? __float_as_uint(dm) : 0x7F800000u;     // non-negative floats order as their bit patterns
                const uint32_t kmin = wave_min_u32(key);
                const float bd = __uint_as_float(kmin);
                if (bd < cur_dist) {
                    const int j = __builtin_ctzll(__ballot(mine && key == kmin));
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
    uint32_t nW = 0, nC = 0, c_need = 1;
    float f = 0.0f;
    uint32_t wd[WR > 0 ? WR : 1], wi[WR > 0 ? WR : 1];     // register form of W: distance bits, id
    if (!fail) {
        LDS_SYNC();
        for (uint32_t i = lane; i < hash_words; i += 64) hash[i] = kEmpty;
        LDS_SYNC();
        if (lane == 0) {
            if (WR == 0) W[0] = make_uint2(__float_as_uint(cur_dist), cur_id);
            C[0] = c_entry(float_keys ? __float_as_uint(cur_dist) : (uint32_t)trunc_key(cur_dist), cur_id);
            visited_insert<BUCKET>(vs, nullptr, cur_id, vis_ovf, used_ovf);
        }
        if (BUCKET) n_ovf = __ballot(lane == 0 && used_ovf) ? 1u : 0u;
#pragma unroll
        for (int w = 0; w < (WR > 0 ? WR : 1); w++) { wd[w] = 0; wi[w] = 0; }
        if (WR > 0 && lane == 0) { wd[0] = __float_as_uint(cur_dist); wi[0] = cur_id; }
        nW = 1; nC = 1; nvis = 1; f = cur_dist; st.max_c = 1;
        LDS_SYNC();
        while (nC > 0 && !fail) {
            DIAG_STAMP(t0);
            uint32_t ckey, cid;
            uint32_t my = kEmpty;                                   // merged form
            {
                const CE c = C[0];
                if constexpr (PACKC) { const uint32_t cw = rfl(c); ckey = cw >> 24; cid = cw & 0xFFFFFFu; }
                else { ckey = rfl(c.x); cid = rfl(c.y); }
                // the candidate's neighbour list is requested now and arrives while the heap is repaired
                if (mrow < a.g.deg0) my = a.g.adj0[(size_t)cid * a.g.deg0 + mrow];
                if (nC > 1) {
                    heap_pop_wave<COrd, CE>(C, nC, lane);
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
                if (BUCKET && n_ovf + 64 > ovf_limit) {
                    // the overflow table is at its limit: continue in a spill table in device memory, taken from the
                    // launch's pool (the queries that get here are few; one that finds the pool empty is re-run)
                    if (!spill) {
                        uint32_t t = 0;
                        if (lane == 0) t = atomicAdd(&a.fail_count[3], 1u);
                        t = rfl(t);
                        if (t >= a.spill_tables) { fail = 1; break; }
                        spill = a.spill_pool + (size_t)t * a.spill_slots;
                        for (uint32_t i = lane; i < a.spill_slots; i += 64) spill[i] = kEmpty;
                        __threadfence();
                    } else if (n_spill + 64 > spill_limit) { fail = 1; break; }
                }
                const bool live = my != kEmpty;
                const uint32_t nlive = (uint32_t)__popcll(__ballot(live));
                used_ovf = false;
                const bool fresh = live && visited_insert<BUCKET>(vs, spill, my, vis_ovf, used_ovf);
                if (!BUCKET && __ballot(vis_ovf)) { fail = 1; break; }
                if (BUCKET) {
                    const uint32_t added = (uint32_t)__popcll(__ballot(fresh && used_ovf));
                    if (spill) n_spill += added; else n_ovf += added;
                }
                const uint32_t m = (uint32_t)__popcll(__ballot(fresh));
                st.sum_deg0 += nlive; st.n_dist += m; nvis += m;
                LDS_SYNC();
                DIAG_STAMP(t2);
                DIAG_ADD(cy_adj, t1b, t2);
                if (m != 0) {
                    // At most 16 fresh rows (the rule at large ef): they are moved, in list order, one per quad into
                    // sub-batch 0 (lane 4r takes the r-th fresh row through one cross-lane push), so that the batch is one
                    // chain of loads and multiply-adds per lane instead of two sparsely filled ones.  What follows takes
                    // the batch in either form: lane order is list order in both.
                    uint32_t my_b = my;
                    bool fresh_b = fresh;
                    if (!kHelper && WR != 1 && m <= (uint32_t)RM::kRowsPerSub) {      // (ef <= 64: most batches hold more, measured no gain)
                        const unsigned long long fm = __ballot(fresh);
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                        const uint32_t to = fresh ? 4u * rank : 63u;          // (lane 63 holds no row in this form)
                        my_b = (uint32_t)__builtin_amdgcn_ds_permute((int)(to << 2), (int)my);
                        fresh_b = (lane & 3) == 0 && ((uint32_t)lane >> 2) < m;
                    }
                    float dm;                                                                  // merged form
                    if constexpr (kHelper) {
                        dm = batch_with_helper(my, fresh, dc);
                    } else {
                        uint32_t rid[RM::kSub]; bool act[RM::kSub]; float dj[RM::kSub];
#pragma unroll
                        for (int s = 0; s < RM::kSub; s++) { rid[s] = BL::spread(my_b, s); act[s] = BL::spread((uint32_t)fresh_b, s) != 0; }
                        rows_distance<SCALAR, ROWS, NBUF, IP, -1, NL>(a.g, dc, rid, act, dj, lane, qr);
                        dm = (lane & 3) == 1 ? dj[RM::kSub - 1] : dj[0];
                    }
                    DIAG_STAMP(t3);
                    DIAG_ADD(cy_dist, t2, t3);
                    // Admission in list order (src/hnsw.cpp:765-777).  A neighbour that fails
                    // `dist < f` against the f at the start of the expansion can only be admitted
                    // while W is not full; f never grows once W is full, so skipping the others is exact.
                    unsigned long long todo = __ballot(fresh_b && (dm < f || nW < ef));
                    const uint32_t c_room = nC + (uint32_t)__popcll(todo);      // what the heap must have room for
                    if (c_room > a.cap_c) { fail = 2; break; }
                    c_need = c_room > c_need ? c_room : c_need;
                    const uint32_t keym = float_keys ? __float_as_uint(dm) : (uint32_t)trunc_key(dm);
                    if (WR > 0 && nW == ef) {
                        // W is full (the usual state): its size and the place of its largest entry are fixed
                        while (todo) {
                            const int j = __builtin_ctzll(todo);
                            todo &= todo - 1;
                            const uint32_t db = __builtin_amdgcn_readlane(__float_as_uint(dm), j);
                            if (__uint_as_float(db) < f) {
                                const uint32_t id = __builtin_amdgcn_readlane(my_b, j);
                                const CE ce = c_entry(__builtin_amdgcn_readlane(keym, j), id);
                                nC++;
                                const CE anc = heap_push_read<COrd, CE>(C, nC, lane);      // in flight during the insert into W
                                if constexpr (WR >= 2) {
                                    f = __uint_as_float(w_set_replace_max<(WR > 0 ? WR : 1)>(wd, wi, __float_as_uint(f), db, id, lane));
                                } else {
                                    w_sorted_insert<(WR > 0 ? WR : 1)>(wd, wi, ef, db, id, lane);
                                    f = __uint_as_float(w_entry_dist<(WR > 0 ? WR : 1)>(wd, ef - 1));
                                }
                                heap_push_finish<COrd, CE>(C, nC, ce, anc, lane);
                                LDS_SYNC();
                            }
                        }
                        if (nC > st.max_c) st.max_c = nC;     // C only grows inside the loop
                    } else
                    while (todo) {
                        const int j = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        const float dist = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dm), j));
                        if (dist < f || nW < ef) {
                            const uint32_t id = __builtin_amdgcn_readlane(my_b, j);
                            const CE ce = c_entry(__builtin_amdgcn_readlane(keym, j), id);
                            const uint2 we = make_uint2(__float_as_uint(dist), id);
                            nC++;
                            if (WR > 0) {
                                heap_push_wave<COrd, CE>(C, nC, ce, lane);
                                if constexpr (WR >= 2) {
                                    if (nW < ef) {
                                        w_set_append<(WR > 0 ? WR : 1)>(wd, wi, nW, we.x, we.y, lane);
                                        nW++;
                                        f = dist > f ? dist : f;          // (the largest distance in W, as W_.front() of the reference's heap)
                                    } else {
                                        f = __uint_as_float(w_set_replace_max<(WR > 0 ? WR : 1)>(wd, wi, __float_as_uint(f), we.x, we.y, lane));
                                    }
                                } else {
                                    w_sorted_insert<(WR > 0 ? WR : 1)>(wd, wi, nW, we.x, we.y, lane);
                                    if (nW < ef) nW++;
                                    f = __uint_as_float(w_entry_dist<(WR > 0 ? WR : 1)>(wd, nW - 1));
                                }
                                LDS_SYNC();
                            } else {
                                nW++;
                                heap_push_wave<COrd, CE>(C, nC, ce, lane);
                                heap_push_wave<WOrder, uint2>(W, nW, we, lane);
                                LDS_SYNC();
                                if (nW > ef) {
                                    const uint2 root = heap_pop_wave<WOrder, uint2>(W, nW, lane);
                                    nW--;
                                    f = __uint_as_float(root.x);
                                    LDS_SYNC();
                                } else {
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
                my = b0 + mrow < a.g.deg0 ? a.g.adj0[(size_t)cid * a.g.deg0 + b0 + mrow] : kEmpty;
            }
        }
    }
    st.n_visited0 = nvis;
    DIAG_STAMP(t_l0_end);

    release_helper();                    // (two-wavefront form) no distance batch follows
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
    // Only entries that can rank below out_k take part in the sort: those not farther than the out_k-th smallest
    // distance `dk`.  (The quantised-row mode re-scores everything first, so it sorts everything.)
    uint32_t nsort = nW;
    if constexpr (WR >= 2 && !rows_quantised(ROWS)) {
        // W is an unordered set in registers: find dk by taking minima (counting ties) until out_k entries are
        // covered, then write the entries up to dk to LDS back to back
        uint32_t dk = 0xFFFFFFFFu;
        if (a.out_k < nW) {
            uint32_t covered = 0, t = 0;
            bool first = true;
            while (covered < a.out_k) {
                uint32_t m = 0xFFFFFFFFu;
#pragma unroll
                for (int w = 0; w < WR; w++) {
                    const bool in = (uint32_t)(w * 64 + lane) < nW && (first || wd[w] > t);
                    m = in && wd[w] < m ? wd[w] : m;
                }
                t = wave_min_u32(m);
                first = false;
#pragma unroll
                for (int w = 0; w < WR; w++)
                    covered += (uint32_t)__popcll(__ballot((uint32_t)(w * 64 + lane) < nW && wd[w] == t));
            }
            dk = t;
        }
        uint32_t base = 0;
#pragma unroll
        for (int w = 0; w < WR; w++) {
            const bool sel = (uint32_t)(w * 64 + lane) < nW && wd[w] <= dk;
            const unsigned long long b = __ballot(sel);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (sel) W[base + before] = make_uint2(wd[w], wi[w]);
            base += (uint32_t)__popcll(b);
        }
        nsort = base;
        LDS_SYNC();
    } else if (WR > 0) {
#pragma unroll
        for (int w = 0; w < WR; w++)
            if ((uint32_t)(w * 64 + lane) < nW) W[w * 64 + lane] = make_uint2(wd[w], wi[w]);
        LDS_SYNC();
    }
    if constexpr (rows_quantised(ROWS)) {
        // the search ran on the quantised table: replace every result's distance by the exact one
        // (float rows, the reference's SSE order) before sorting
        const float* q = a.queries + (size_t)qi * a.g.d;
        const uint32_t qpad_f = (a.g.d + 63u) & ~63u;
        for (uint32_t i = lane; i < qpad_f; i += 64) qs[stored_pos(i)] = i < a.g.d ? q[i] : 0.0f;
        LDS_SYNC();
        const DistCtx dx = make_dist_ctx<false>(qs, a.g.d);
        for (uint32_t b0 = 0; b0 < nW; b0 += 32) {
            const uint32_t e = b0 + mrow;                                 // merged form: lane 4r+s <-> entry b0 + 2r + s
            const uint32_t my = e < nW ? W[e].y : kEmpty;
            uint32_t rid[2]; bool act[2]; float dj[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) { rid[s2] = BL::spread(my, s2); act[s2] = rid[s2] != kEmpty; }
            rows_distance<false, 0>(a.g, dx, rid, act, dj, lane);
            const float dm = (lane & 3) == 1 ? dj[1] : dj[0];
            if (my != kEmpty) W[e].x = __float_as_uint(dm);
            st.n_dist += (uint32_t)__popcll(__ballot(my != kEmpty));
        }
        LDS_SYNC();
    }
    // The one-register form wrote W in ascending order of distance, so the entries up to dk are the first ones.
    if (WR == 1 && !rows_quantised(ROWS) && a.out_k < nW) {
        const uint32_t dk = w_entry_dist<(WR > 0 ? WR : 1)>(wd, a.out_k - 1);
        nsort = 0;
#pragma unroll
        for (int w = 0; w < (WR > 0 ? WR : 1); w++) {
            const uint32_t left = nW > 64u * w ? nW - 64u * w : 0u;
            const unsigned long long valid = left >= 64u ? ~0ull : (1ull << left) - 1ull;
            nsort += (uint32_t)__popcll(__ballot(wd[w] <= dk) & valid);
        }
    }
    for (uint32_t i = lane; i < nsort; i += 64) {
        const uint2 me = W[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nsort; j++) {
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
        st.n_dist = cy_up; st.n_expand = cy_adj; st.n_sweeps = cy_dist; st.sum_deg0 = cy_adm;
        st.sum_degu = cy_pop; st.n_visited0 = cy_fin; st.n_visited_up = (uint32_t)(t_end - t_begin);
    }
#endif
    if (lane == 0) {
        if (a.out_count) a.out_count[qi] = nW;
        if (a.stats) a.stats[qi] = st;
#ifndef HNSW_DIAG
        // What this query needed of its scratch, for the sizing of later batches.  Only a query that goes beyond what
        // the launcher already knows (a.known_*) touches device memory for it — in the steady state none does — and even
        // then the word is read first: most queries stay below the running maximum.
        const uint32_t vmax = st.n_visited0 > st.n_visited_up ? st.n_visited0 : st.n_visited_up;
        if (vmax > a.known_visit && vmax > __hip_atomic_load(&a.fail_count[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&a.fail_count[1], vmax);
        if (c_need > a.known_c && c_need > __hip_atomic_load(&a.fail_count[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&a.fail_count[2], c_need);
        // (the mean from one query in 32: atomics to one address are served one after the other, and ten thousand
        // of them take longer than a whole batch)
        if ((qi & 31u) == 0u) { atomicAdd(&a.fail_count[4], vmax >> 4); atomicAdd(&a.fail_count[5], 1u); }
#endif
    }
}

// Diagnostic: what the memory system delivers to the search kernel's row reads when nothing else is in the
// way.  Every wavefront computes distances to `iters` batches of 32 uniformly random rows of the index's own
// vector table with the search kernel's own loads and arithmetic (rows_distance: four lanes per row, 32 rows
// in flight), but with no heaps, no visited set, no neighbour lists and no dependence of one batch on the
// previous one.  Residency is set the way the search kernel's is, by the dynamic LDS size.
template <int NL, bool SCALAR = false>
__global__ __launch_bounds__(64) void gather_ceiling_kernel(DeviceGraph g, uint32_t iters, float* __restrict__ out)
{
    extern __shared__ __align__(16) uint32_t smem[];
    float* qs = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x;
    const uint32_t qpad = (g.d + 63u) & ~63u;
    for (uint32_t i = lane; i < qpad; i += 64) qs[i] = 0.0f;
    LDS_SYNC();
    const DistCtx dc = make_dist_ctx<SCALAR>(qs, g.d);
    QRegs<NL> qr;
    qregs_load(qr, qs, lane);
    auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
    const uint32_t seed = blockIdx.x * 2654435761u + 12345u, r = (uint32_t)lane >> 2;
    const bool act[2] = {true, true};
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t rid[2]; float dj[2];
        rid[0] = __umulhi(mix(seed + it * 32u + 2u * r), g.n);
        rid[1] = __umulhi(mix(seed + it * 32u + 2u * r + 1u), g.n);
        rows_distance<SCALAR, 0, 2, false, -1, NL>(g, dc, rid, act, dj, lane, qr);
        acc += dj[0] + dj[1];
    }
    if (lane == 0) out[blockIdx.x] = acc;
}

// The end of a launch: its report (the header of its overflow list) goes to the host, and the header is cleared for the
// next launch that uses the list.  `host_words` is pinned host memory the host is watching: the first seven words are
// written, then — after a system-wide fence — the eighth, whose arrival tells the host that the others are there.
// One small dispatch behind the search kernel on the same stream; it replaces a clear in front of the launch and a
// copy engine's transfer behind it.
__global__ __launch_bounds__(64) void report_kernel(uint32_t* __restrict__ header, uint32_t* __restrict__ host_words)
{
    const uint32_t i = threadIdx.x;
    uint32_t v = 0;
    if (i < kFailHeader) v = header[i];
    if (i + 1 < kFailHeader) __hip_atomic_store(&host_words[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (i < kFailHeader) header[i] = 0u;
    if (i + 1 == kFailHeader) __hip_atomic_store(&host_words[i], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Rows as the caller has them (n x d, dense) -> stored form (n x row_stride, regrouped lines, zero padded).
__global__ void store_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, uint32_t rows, uint32_t d, uint32_t row_stride)
{
    const size_t total = (size_t)rows * row_stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / row_stride), dim = (uint32_t)(i % row_stride);
        dst[(size_t)row * row_stride + stored_pos(dim)] = dim < d ? src[(size_t)row * d + dim] : 0.0f;
    }
}

// Largest Euclidean length of a row (as the bits of a non-negative float, which order like the float): what the
// launcher needs to know to promise that candidate keys stay small.
__global__ void row_norm_max_kernel(const float* __restrict__ src, uint32_t rows, uint32_t d, uint32_t* __restrict__ out_bits)
{
    float m = 0.0f;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        float sq = 0.0f;
        for (uint32_t i = 0; i < d; i++) { const float v = src[r * d + i]; sq = __builtin_fmaf(v, v, sq); }
        const float len = sq == sq ? __builtin_sqrtf(sq) : __builtin_inff();
        m = len > m ? len : m;
    }
    atomicMax(out_bits, __float_as_uint(m * 1.001f));
}

// The byte table, and whether it is exact: *bad is raised if any value is not an integer in 0..255.
__global__ void store_rows_u8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, uint32_t rows, uint32_t d, uint32_t row_stride_u8,
                                     uint32_t* __restrict__ bad)
{
    const size_t total = (size_t)rows * row_stride_u8;
    bool any_bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / row_stride_u8), dim = (uint32_t)(i % row_stride_u8);
        uint8_t b = 0;
        if (dim < d) {
            const float v = src[(size_t)row * d + dim];
            b = (uint8_t)(v >= 0.0f && v <= 255.0f ? (int)v : 0);
            if (!((float)b == v)) any_bad = true;
        }
        dst[(size_t)row * row_stride_u8 + stored_pos_u8(dim)] = b;
    }
    if (any_bad) atomicOr(bad, 1u);
}

// The quantised byte table: byte = clamp(round((v - lo) * inv_step), 0, 255).
__global__ void store_rows_q8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, uint32_t rows, uint32_t d, uint32_t row_stride_u8,
                                     float lo, float inv_step)
{
    const size_t total = (size_t)rows * row_stride_u8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / row_stride_u8), dim = (uint32_t)(i % row_stride_u8);
        uint8_t b = 0;
        if (dim < d) {
            const float t = rintf((src[(size_t)row * d + dim] - lo) * inv_step);
            b = (uint8_t)(t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t));
        }
        dst[(size_t)row * row_stride_u8 + stored_pos_u8(dim)] = b;
    }
}

}  // namespace

// Rows of exactly 128 or 96 used dims (the SSE loop uses the first 16 * floor(d / 16)) have kernels of their own whose
// distance batches are straight-line code; 0 = the kernel for any row length.
static int row_shape_loads(uint32_t d, bool scalar = false)
{
    if (scalar) return (d > 96u && d <= 112u && (d & 3u) == 0u) ? 7 : 0;      // scalar order: all dims, 7 loads (d = 100), no fused tail
    const uint32_t d_eff = d & ~15u;
    return d_eff == 128u ? 8 : (d_eff == 96u ? 6 : 0);
}

size_t search_lds_bytes(const SearchArgs& a, bool w_in_lds)
{
    size_t dwords = (a.q_overlay ? 0 : (size_t)a.qpad) + (w_in_lds ? 2 * (size_t)a.cap_w : 0) + (size_t)a.c_words * a.cap_c +
                    (a.vis_rb == 0xFFFFFFFFu ? (size_t)a.hash_size : (size_t)a.hash_size / 2) + (size_t)a.ovf_size;
    return std::max<size_t>(dwords, a.qpad) * 4;         // (the query is written there first, whatever takes its place later)
}

// ---------------------------------------------------------------------------------------
DeviceIndex::~DeviceIndex()
{
    // (a destructor cannot refuse: calls still in flight are abandoned)
    for (SearchCall& c : calls_) c.busy = false;
    noalloc_call_.busy = false;
    release();
}

void DeviceIndex::release()
{
    if (calls_in_flight()) throw std::logic_error("the index was changed (or destroyed) while a search call on it was in flight: finish the call first");
    if (device >= 0) (void)hipSetDevice(device);
    for (auto& b : bufs_) { if (b) (void)hipFree(b); b = nullptr; }
    auto drop = [](SearchCall& c) {
        if (c.d_lists) (void)hipFree(c.d_lists);
        if (c.d_spill) (void)hipFree(c.d_spill);
        if (c.h_report) (void)hipHostFree(c.h_report);
        c = SearchCall{};
    };
    for (SearchCall& c : calls_) drop(c);
    noalloc_call_.d_spill = nullptr;          // (borrowed, never owned)
    drop(noalloc_call_);
    if (device >= 0 && device < kMaxDevices) g_noalloc_headers_zero[device] = false;      // (be sure rather than remember)
    up_lists_cap_ = 0;
    version = ~0ull; rows_epoch = 0;
}

bool DeviceIndex::calls_in_flight() const
{
    for (const SearchCall& c : calls_) if (c.busy) return true;
    return noalloc_call_.busy;
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
    // base vectors: row-major, every row padded with zeros to whole 128-byte lines, the dims of a
    // line regrouped by lane sum (stored_pos); converted on the device, a slab of rows at a time
    {
        float* dv = nullptr;
        const size_t bytes = (size_t)g.n * g.row_stride * sizeof(float);
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dv), bytes));
        bufs_[0] = dv;
        const uint32_t slab = (uint32_t)std::max<size_t>(1, std::min<size_t>(g.n, ((size_t)64 << 20) / g.d));
        // second copy, one byte per dim; kept only if it is exact (every value an integer in 0..255)
        g.row_stride_u8 = (g.d + 127u) & ~127u;
        uint8_t* du = nullptr;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&du), (size_t)g.n * g.row_stride_u8));
        bufs_[6] = du;
        uint32_t* d_bad = nullptr;          // [0] some value is not an integer in 0..255, [1] bits of the largest row length
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_bad), 8));
        bufs_[7] = d_bad;
        HIP_CHECK(hipMemset(d_bad, 0, 8));
        float* tmp = nullptr;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)slab * g.d * sizeof(float)));
        for (uint32_t r0 = 0; r0 < g.n; r0 += slab) {
            const uint32_t rows = std::min(slab, g.n - r0);
            hipError_t e = hipMemcpy(tmp, h.vector_data_.data() + (size_t)r0 * g.d, (size_t)rows * g.d * sizeof(float), hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(store_rows_kernel, dim3(4096), dim3(256), 0, 0, tmp, dv + (size_t)r0 * g.row_stride, rows, g.d, g.row_stride);
                hipLaunchKernelGGL(store_rows_u8_kernel, dim3(4096), dim3(256), 0, 0, tmp, du + (size_t)r0 * g.row_stride_u8, rows, g.d,
                                   g.row_stride_u8, d_bad);
                hipLaunchKernelGGL(row_norm_max_kernel, dim3(1024), dim3(256), 0, 0, tmp, rows, g.d, d_bad + 1);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) { (void)hipFree(tmp); HIP_CHECK(e); }
        }
        HIP_CHECK(hipFree(tmp));
        g.vectors = dv;
        uint32_t bad = 1, both[2] = {1, 0x7F800000u};
        HIP_CHECK(hipMemcpy(both, d_bad, 8, hipMemcpyDeviceToHost));
        bad = both[0];
        std::memcpy(&g.max_row_norm, &both[1], 4);
        HIP_CHECK(hipFree(d_bad)); bufs_[7] = nullptr;
        if (bad) { HIP_CHECK(hipFree(du)); bufs_[6] = nullptr; du = nullptr; }
        g.vectors_u8 = du;
    }
    g.adj0 = to_device(fg.adj0, bufs_[1]);
    g.up_off = to_device(fg.up_off, bufs_[2]);
    g.up_down = to_device(fg.up_down, bufs_[3]);
    g.up_adj_slot = to_device(fg.up_adj_slot, bufs_[4]);
    g.up_adj_order = to_device(fg.up_adj_order, bufs_[5]);
    up_lists_cap_ = fg.up_off[fg.nlayers];
    HIP_CHECK(hipDeviceSynchronize());
    // set-up for the search calls: the pinned word the overflow count is read into, and this GPU's address of the
    // static scratch array (so that a search call itself never has to allocate or look anything up)
    if (!noalloc_call_.h_report) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&noalloc_call_.h_report), 64, hipHostMallocDefault));
    (void)noalloc_scratch_for(dev, true);
    tune.read();
    dot4_off = false; pack_off = false;
    version = h.version_;
    rows_epoch = h.rows_epoch_;
    learn_flags = 0; learn_ef = 0; learn_visit = 0; learn_c = 0; learn_mean = 0; learn_notes.clear();
    shape_full = BucketShape{}; shape_bulk = BucketShape{};
}

// The graph of an index whose vectors are on the device already (same rows, same list lengths): the neighbour lists
// and the upper layers are copied again, nothing else.  The host keeps layer 0's lists in the device's own layout
// (node_order per slot, 0xFFFFFFFF padding), so that part is one copy.
void DeviceIndex::refresh_graph(const HostIndex& h, bool with_layer0, FlatGraph* exported)
{
    if (device < 0 || !g.vectors) throw std::logic_error("refresh_graph: index not on a device");
    if (calls_in_flight()) throw std::logic_error("the index was changed while a search call on it was in flight: finish the call first");
    HIP_CHECK(hipSetDevice(device));
    // (the upper layers' lists are read from the host's per-layer arrays directly, which have the device's layout:
    // [slot][degu], 0xFFFFFFFF behind the live entries)
    FlatGraph fg;
    h.export_flat(fg, false, false);
    if (fg.n != g.n || fg.d != g.d || fg.deg0 != g.deg0 || fg.degu != g.degu || fg.nlayers == 0)
        throw std::logic_error("refresh_graph: the index changed shape");
    if (with_layer0) {
        const std::vector<uint32_t>& lists0 = h.layers_[0].nb_order;
        if (lists0.size() < (size_t)g.n * g.deg0) throw std::logic_error("refresh_graph: layer 0 is shorter than the index");
        HIP_CHECK(hipMemcpy(const_cast<uint32_t*>(g.adj0), lists0.data(), (size_t)g.n * g.deg0 * 4, hipMemcpyHostToDevice));
    }
    for (int b = 2; b <= 3; b++) { if (bufs_[b]) HIP_CHECK(hipFree(bufs_[b])); bufs_[b] = nullptr; }
    g.nlayers = fg.nlayers;
    g.top_ep_slot = fg.ep_slot[fg.nlayers - 1]; g.top_ep_order = fg.ep_order[fg.nlayers - 1];
    g.up_off = to_device(fg.up_off, bufs_[2]);
    g.up_down = to_device(fg.up_down, bufs_[3]);
    // the two list tables keep their buffers from refresh to refresh (grown with headroom when the layers outgrow them)
    const size_t U = fg.up_off[fg.nlayers];
    if (U > up_lists_cap_ || !bufs_[4] || !bufs_[5]) {
        for (int b = 4; b <= 5; b++) { if (bufs_[b]) HIP_CHECK(hipFree(bufs_[b])); bufs_[b] = nullptr; }
        up_lists_cap_ = 0;
        const size_t cap = U + U / 8 + 1024;
        HIP_CHECK(hipMalloc(&bufs_[4], cap * g.degu * 4));
        HIP_CHECK(hipMalloc(&bufs_[5], cap * g.degu * 4));
        up_lists_cap_ = cap;
    }
    uint32_t* d_slot = static_cast<uint32_t*>(bufs_[4]);
    uint32_t* d_order = static_cast<uint32_t*>(bufs_[5]);
    for (uint32_t l = 1; l < fg.nlayers; l++) {
        const LayerStore& L = h.layers_[l];
        const size_t words = (size_t)L.count * g.degu;
        if (L.deg != g.degu || L.nb_slot.size() < words || L.nb_order.size() < words) throw std::logic_error("refresh_graph: an upper layer's lists do not have the index's shape");
        if (words == 0) continue;
        HIP_CHECK(hipMemcpy(d_slot + (size_t)fg.up_off[l] * g.degu, L.nb_slot.data(), words * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_order + (size_t)fg.up_off[l] * g.degu, L.nb_order.data(), words * 4, hipMemcpyHostToDevice));
    }
    g.up_adj_slot = d_slot;
    g.up_adj_order = d_order;
    HIP_CHECK(hipDeviceSynchronize());
    if (with_layer0) version = h.version_;
    if (exported) *exported = std::move(fg);
}

uint32_t DeviceIndex::search_on(const DeviceGraph& view, const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t flags,
                                uint32_t* d_ids, float* d_dist, uint32_t* d_count, void* stream)
{
    if (calls_in_flight()) throw std::logic_error("search_on: a search call on the index is in flight");
    const DeviceGraph own = g;
    g = view;
    uint32_t r = 0;
    try {
        r = search(d_queries, nq, ef, out_k, flags | kFlagView, d_ids, d_dist, d_count, nullptr, stream);
    } catch (...) {
        g = own;
        throw;
    }
    g = own;
    return r;
}

void DeviceIndex::ensure_quantized(const HostIndex& h)
{
    if (device < 0) throw std::logic_error("ensure_quantized: index not on a device");
    if (g.vectors_q8) return;
    HIP_CHECK(hipSetDevice(device));
    // One grid for the whole table: [lo, hi] = the 0.1 % and 99.9 % quantiles of a sample of the stored
    // values (outliers are clamped), 255 steps in between.
    const size_t total = (size_t)g.n * g.d;
    const size_t want = std::min<size_t>(total, (size_t)1 << 20);
    std::vector<float> sample(want);
    const size_t stride = total / want;
    for (size_t i = 0; i < want; i++) sample[i] = h.vector_data_[i * stride];
    const size_t klo = want / 1000, khi = want - 1 - want / 1000;
    std::nth_element(sample.begin(), sample.begin() + klo, sample.end());
    const float lo = sample[klo];
    std::nth_element(sample.begin(), sample.begin() + khi, sample.end());
    float hi = sample[khi];
    if (!(hi > lo)) hi = lo + 1.0f;
    const float step = (hi - lo) / 255.0f;
    uint8_t* dq = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dq), (size_t)g.n * g.row_stride_u8));
    bufs_[7] = dq;
    const uint32_t slab = (uint32_t)std::max<size_t>(1, std::min<size_t>(g.n, ((size_t)64 << 20) / g.d));
    float* tmp = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)slab * g.d * sizeof(float)));
    for (uint32_t r0 = 0; r0 < g.n; r0 += slab) {
        const uint32_t rows = std::min(slab, g.n - r0);
        hipError_t e = hipMemcpy(tmp, h.vector_data_.data() + (size_t)r0 * g.d, (size_t)rows * g.d * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(store_rows_q8_kernel, dim3(4096), dim3(256), 0, 0, tmp, dq + (size_t)r0 * g.row_stride_u8, rows, g.d,
                               g.row_stride_u8, lo, 1.0f / step);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipFree(tmp); HIP_CHECK(e); }
    }
    HIP_CHECK(hipFree(tmp));
    g.q8_lo = lo; g.q8_inv_step = 1.0f / step;
    g.vectors_q8 = dq;
}

float DeviceIndex::gather_ceiling(uint32_t n_waves, uint32_t rows_per_wave, uint32_t lds_bytes, void* stream_v, uint64_t& bytes, bool scalar)
{
    if (device < 0) throw std::logic_error("gather_ceiling: index not on a device");
    if (n_waves == 0 || rows_per_wave < 32) throw std::invalid_argument("gather_ceiling: need at least one wavefront and 32 rows each");
    HIP_CHECK(hipSetDevice(device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const uint32_t iters = rows_per_wave / 32u;
    const size_t min_lds = (size_t)((g.d + 63u) & ~63u) * 4;
    size_t lds = std::max<size_t>(lds_bytes, min_lds);
    if (lds > 65536) lds = 65536;
    float* d_out = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n_waves * 4));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.0f;
    try {
        // (the same row-shape kernels the search uses)
        typedef void (*CeilingKernel)(DeviceGraph, uint32_t, float*);
        const int nl = row_shape_loads(g.d, scalar);
        const CeilingKernel ck = scalar ? (nl == 7 ? gather_ceiling_kernel<7, true> : gather_ceiling_kernel<0, true>)
                                        : (nl == 8 ? gather_ceiling_kernel<8> : (nl == 6 ? gather_ceiling_kernel<6> : gather_ceiling_kernel<0>));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ck), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(ck, dim3(n_waves), dim3(64), lds, stream, g, iters, d_out);       // warm-up
        HIP_CHECK(hipEventRecord(e0, stream));
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(ck, dim3(n_waves), dim3(64), lds, stream, g, iters, d_out);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(e1, stream));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    } catch (...) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(d_out);
        throw;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIP_CHECK(hipFree(d_out));
    bytes = (uint64_t)n_waves * iters * 32u * (scalar ? g.d : (g.d & ~15u)) * 4u;      // the bytes the search's counters would charge for these rows
    return ms / 3.0f;
}

void DeviceIndex::last_launch(uint32_t& lds_bytes, uint32_t& resident_per_cu, uint32_t& visited_form) const
{
    lds_bytes = last_lds; visited_form = (uint32_t)last_form; resident_per_cu = 0;
    if (!last_kernel) return;
    HIP_CHECK(hipSetDevice(device));
    int nb = 0;
    if (last_lds <= 65536) HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, last_kernel, (int)last_block, last_lds));
    else nb = (int)(163840 / last_lds);
    if (last_block == 64 && nb > 4 * last_wps) nb = 4 * last_wps;        // (the kernel is pinned at last_wps wavefronts per SIMD)
    resident_per_cu = (uint32_t)nb;
}

typedef void (*SearchKernel)(SearchArgs);

template <bool SCALAR, int ROWS, bool BUCKET, int NBUF, bool IP = false, int NL = 0>
static SearchKernel pick_wr(int wr, int wps = 4, bool packc = false)
{
    if constexpr (ROWS >= 3) { if (wr == 1 && wps == 6) return hnsw_search_kernel<SCALAR, 1, ROWS, BUCKET, NBUF, IP, NL, 6>; }
    if constexpr (ROWS == 0 && !IP && NBUF == 2) {
        if (packc && wr == 1) return hnsw_search_kernel<SCALAR, 1, 0, BUCKET, 2, false, NL, 4, true>;
        if (packc && wr == 2) return hnsw_search_kernel<SCALAR, 2, 0, BUCKET, 2, false, NL, 4, true>;
    }
    switch (wr) {
        case 0: return hnsw_search_kernel<SCALAR, 0, ROWS, BUCKET, NBUF, IP, NL>;
        case 1: return hnsw_search_kernel<SCALAR, 1, ROWS, BUCKET, NBUF, IP, NL>;
        case 2: return hnsw_search_kernel<SCALAR, 2, ROWS, BUCKET, NBUF, IP, NL>;
        default: return hnsw_search_kernel<SCALAR, 4, ROWS, BUCKET, NBUF, IP, NL>;
    }
}

template <bool BUCKET>
static SearchKernel pick_rows(bool scalar, int wr, int rows, bool ip, int nl, int wps, bool packc)
{
    if (ip && nl == 8) return pick_wr<false, 0, BUCKET, 2, true, 8>(wr);
    if (ip && nl == 6) return pick_wr<false, 0, BUCKET, 2, true, 6>(wr);
    if (ip) return pick_wr<false, 0, BUCKET, 2, true>(wr);
    if (scalar && nl == 7) return pick_wr<true, 0, BUCKET, 2, false, 7>(wr, 4, packc);
    if (scalar) return pick_wr<true, 0, BUCKET, 2>(wr, 4, packc);
    if (rows == 3 && nl == 8) return pick_wr<false, 3, BUCKET, 2, false, 8>(wr, wps);
    if (rows == 3) return pick_wr<false, 3, BUCKET, 2>(wr, wps);
    if (rows == 4 && nl == 8) return pick_wr<false, 4, BUCKET, 2, false, 8>(wr, wps);
    if (rows == 4) return pick_wr<false, 4, BUCKET, 2>(wr, wps);
    if (rows == 1 && nl == 8) return pick_wr<false, 1, BUCKET, 2, false, 8>(wr);
    if (rows == 2 && nl == 8) return pick_wr<false, 2, BUCKET, 2, false, 8>(wr);
    if (rows == 1) return pick_wr<false, 1, BUCKET, 2>(wr);
    if (rows == 2) return pick_wr<false, 2, BUCKET, 2>(wr);
    if (nl == 8) return pick_wr<false, 0, BUCKET, 2, false, 8>(wr, wps, packc);
    if (nl == 6) return pick_wr<false, 0, BUCKET, 2, false, 6>(wr, 4, packc);
    return pick_wr<false, 0, BUCKET, 2>(wr, 4, packc);
}

// deep: the float-row, probing-form kernel with six row pieces in flight and two wavefronts per query (see hnsw_search_kernel)
static SearchKernel pick_kernel(bool scalar, int wr, int rows, bool bucket, bool ip, uint32_t d, bool generic /* the kernel for any row length */, bool deep = false, int wps = 4, bool packc = false)
{
    if (deep && !scalar && rows == 0 && !bucket && !ip) {
        // (960 used dims = seven whole 128-dim items and one of four loads: the shape of the configuration this form exists for)
        if (!generic && (d & ~15u) == 960u) return pick_wr<false, 0, false, 6, false, 704>(wr);
        return pick_wr<false, 0, false, 6>(wr);
    }
    const int nl = generic ? 0 : row_shape_loads(d, scalar);
    return bucket ? pick_rows<true>(scalar, wr, rows, ip, nl, wps, packc) : pick_rows<false>(scalar, wr, rows, ip, nl, wps, packc);
}

constexpr size_t kLdsPerCu = 163840;     // gfx950: 160 KB

// First-batch estimates of what a search at `ef` needs, before anything was observed on this index.  On a
// 10^6-node graph of degree 32 the largest visited count of 10^4 queries is close to 18.7 x ef + 470 for
// ef = 50..300 and the largest candidate heap close to 2.2 x ef + 120; small ef stays below 28 x ef + 192
// visited nodes.  Every launch reports the largest visited count and heap size its queries reached, and later
// batches are sized from those (DeviceIndex::learn_*), so these figures only decide how the first batch starts.
static uint32_t visit_estimate(int ef) { const uint32_t e = (uint32_t)ef; return std::min(e * 28u + 192u, (e * 39u) / 2u + 560u); }
// (the overflow test adds a whole neighbour list to the heap's size; 4 x ef in steps of 64 is the smaller bound for small ef)
static uint32_t heap_estimate(int ef) { const uint32_t e = (uint32_t)ef; return std::min((e * 4u + 63u) & ~63u, (e * 12u) / 5u + 150u); }

// 32-bit probing table for `want_visit` ids: at most 7/8 full, with room for one more neighbour list of 64
static uint32_t hash32_slots(uint32_t want_visit) { return (((want_visit + 64u) * 8u + 6u) / 7u + 63u) & ~63u; }

// The half-size (16-bit slot) probing form of the visited table for `want_visit` ids below n, if the ids fit:
// region bits and total slots (regions x an even number of slots each).  Regions are sized for their
// expected share plus four standard deviations; a region that overflows anyway re-runs the query.
static bool visited16_shape(uint32_t n, uint32_t want_visit, uint32_t rb_min, uint32_t& rb, uint32_t& slots)
{
    rb = rb_min;
    while (rb < 12 && ((n - 1u) >> rb) >= 0xFFFFu) rb++;
    if (((n - 1u) >> rb) >= 0xFFFFu) return false;
    const float share = (float)(want_visit + 64u) / (float)(1u << rb);
    uint32_t per = (uint32_t)((share + 4.0f * sqrtf(share)) * 8.0f / 7.0f) + 2u;
    per = (per + 1u) & ~1u;
    if (per < 96u) return false;             // too small for the law of large numbers to help
    slots = per << rb;
    return true;
}

// The bucket form for `want_visit` ids below n: `regions` regions of 8 * nchunk 16-bit slots plus an overflow table
// of ovf_slots 32-bit slots (one slot of every region holds its fill count), the combination with the fewest bytes.
// A region receives a Poisson share of the visited ids (the ids of the nodes a search meets are as good as random);
// what does not fit goes to the overflow table, which is sized for the expected total plus four standard deviations
// plus one batch.
static bool bucket_shape(uint32_t n, uint32_t want_visit, uint32_t rb_min, uint32_t& regions, uint32_t& nchunk, uint32_t& ovf_slots)
{
    const uint32_t r_min = std::max<uint32_t>({(n - 1u) / 0xFFFFu + 1u, 1u << std::min(rb_min, 20u), 4u});   // tags stay below 0xFFFF
    const uint32_t r_max = std::max<uint32_t>(r_min, (want_visit + 64u) / 4u);
    size_t best = ~(size_t)0;
    for (uint32_t r = r_min; r <= r_max; r += std::max(1u, r / 32u)) {
        const double R = (double)r, lambda = (double)(want_visit + 64u) / R;
        for (uint32_t c = 2; c <= 4; c++) {
            const uint32_t S = 8 * c;
            if (lambda > 1.2 * S || lambda < 0.4 * S) continue;
            // first and second moment of (X - U)+ for X ~ Poisson(lambda), U = S - 1 usable slots
            const uint32_t U = S - 1u;
            double pk = exp(-lambda), e1 = 0.0, e2 = 0.0;
            for (uint32_t k = 1; k <= U + 96; k++) {
                pk *= lambda / (double)k;
                if (k > U) { const double x = (double)(k - U); e1 += x * pk; e2 += x * x * pk; }
            }
            const double total = R * e1, sd = sqrt(std::max(0.0, R * (e2 - e1 * e1)));
            const uint32_t need = (uint32_t)(total + 4.0 * sd) + 64u;
            const uint32_t ovf = std::max(128u, hash32_slots(need));
            const size_t bytes = (size_t)r * S * 2 + (size_t)ovf * 4;
            if (bytes < best) { best = bytes; regions = r; nchunk = c; ovf_slots = ovf; }
        }
    }
    return best != ~(size_t)0;
}

// Spill tables of the bucket form (device memory): a pool of 64 MB per search call in flight
constexpr uint32_t kSpillTables = 1024, kSpillSlots = 16384;

// The switches a developer can set in the environment (diagnostics, tests, tuning; the C ABI has no knob for them).
// Read when an index is put on a device and again by hnsw_amd_reload_env, never inside a search call.
void Tunables::read()
{
    auto on = [](const char* name) { return getenv(name) != nullptr; };
    auto num = [](const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; };
    generic_rows = on("HNSW_AMD_GENERIC_ROWS");      // the kernel for any row length instead of the one built for the row shape
    no_deep_rows = on("HNSW_AMD_NO_DEEP_ROWS");      // never the two-wavefront form
    visited32 = on("HNSW_AMD_VISITED32"); visited16 = on("HNSW_AMD_VISITED16"); visitedb = on("HNSW_AMD_VISITEDB");   // force one form of the visited set
    no_spill = on("HNSW_AMD_NO_SPILL");              // bucket form without spill tables
    verbose = on("HNSW_AMD_VERBOSE");                // launch configuration and kernel times on stderr
    no_pack_c = on("HNSW_AMD_NO_PACK_C");            // candidate heap entries of 8 bytes even where 4 would do
    no_dot4 = on("HNSW_AMD_NO_DOT4");
    const char* bf = getenv("HNSW_AMD_BULK_FRAC");
    bulk_frac = bf ? (float)atof(bf) : 0.30f;        // (0.45 and 0.30 measure alike on the SIFT-shaped set; 0.30 lets the 12.5 M-row DEEP-shaped shard hold more queries per CU)
    want_visit = num("HNSW_AMD_WANT_VISIT", -1);     // a fixed size of the visited set
    want_bulk = num("HNSW_AMD_WANT_BULK", -1);       // tests: spill whatever it gains
    spill_tables = num("HNSW_AMD_SPILL_TABLES", -1); // tests: a pool that runs dry
    rb_min = num("HNSW_AMD_VIS_RB_MIN", 0);          // tests: the region count of a larger index
    lds_budget = num("HNSW_AMD_LDS_BUDGET", -1);     // tests: bare estimates
    lds_pad = num("HNSW_AMD_LDS_PAD", 0);            // occupancy experiments
}

// The largest dynamic LDS size a kernel was enabled for is a property of the process (per device and kernel), not of an
// index: the table is shared by every index, and the size is only ever raised.
void DeviceIndex::set_max_dynamic_lds(const void* kern, size_t lds)
{
    if (lds <= 32768) return;
    struct Enabled { int device; const void* kernel; size_t bytes; };
    static std::vector<Enabled> table;
    static std::mutex table_mutex;
    std::lock_guard<std::mutex> lock(table_mutex);
    for (Enabled& e : table) if (e.device == device && e.kernel == kern) {
        if (e.bytes >= lds) return;
        HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        e.bytes = lds;
        return;
    }
    HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    table.push_back({device, kern, lds});
}

// ---- one search call = begin (size the scratch from what is known, launch, queue the report) + finish (wait for the
//      report, re-run what outgrew its scratch on larger tables, remember what the batch needed) ----

SearchCall* DeviceIndex::begin(const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t flags, uint32_t* d_ids,
                               float* d_dist, uint32_t* d_count, hnsw_amd_query_stats* d_stats, void* stream_v,
                               bool no_alloc, const CopyBack* copy_back)
{
    if (device < 0) throw std::logic_error("search: index not on a device");
    if (no_alloc && nq > kNoAllocMaxQueries)
        throw std::invalid_argument("search (no-allocation entry point): at most " + std::to_string(kNoAllocMaxQueries) +
                                    " queries per call, got " + std::to_string(nq));
    if (nq != 0) {
        if (ef < 1 || out_k < 1 || out_k > (uint32_t)ef) throw std::invalid_argument("search: need 1 <= out_k <= ef");
        if (ef > (1 << 20)) throw std::length_error("search: ef beyond anything the on-chip scratch can hold");   // (and the size estimates stay in range)
    }
    SearchCall* cp = nullptr;
    if (no_alloc) {
        cp = &noalloc_call_;
        if (cp->busy) throw std::logic_error("search (no-allocation entry point): the previous call has not finished");
    } else {
        for (SearchCall& c : calls_) if (!c.busy) { cp = &c; break; }
        if (!cp) throw std::logic_error("search: " + std::to_string(kMaxCallsInFlight) + " calls are in flight on this index already; finish one first");
    }
    SearchCall& c = *cp;
    c.nq = nq; c.ef = ef; c.flags = flags; c.no_alloc = no_alloc; c.retried = 0; c.tier = 0; c.launched = false;
    c.seen_visit = 0; c.seen_c = 0; c.seen_sum16 = 0; c.seen_done = 0;
    c.has_copy_back = copy_back != nullptr;
    if (copy_back) c.copy_back = *copy_back;
    c.stream = stream_v;
    if (nq == 0) { c.busy = true; return cp; }
    HIP_CHECK(hipSetDevice(device));
    // the two overflow lists (kFailHeader words of header, then query indices): a retry launch reads one and
    // appends to the other
    if (no_alloc) {
        // this GPU's copy of the static scratch array, through the address looked up at upload (never the array's name)
        uint32_t* lists = noalloc_scratch_for(device, false);
        if (!lists || !c.h_report) throw std::logic_error("search (no-allocation entry point): the index was not set up on this device");
        c.cur = lists;
        c.other = lists + (kNoAllocMaxQueries + kFailHeader);
        // (spill tables only if an allocating call left a pool behind that nobody is using now)
        c.d_spill = calls_[0].busy ? nullptr : calls_[0].d_spill;
    } else {
        if (c.list_cap < nq + kFailHeader) {
            if (c.d_lists) HIP_CHECK(hipFree(c.d_lists));
            c.d_lists = nullptr; c.list_cap = 0;
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c.d_lists), ((size_t)nq + kFailHeader) * 4 * 2));
            c.list_cap = nq + kFailHeader;
            c.headers_zero = false;
        }
        if (!c.h_report) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&c.h_report), 64, hipHostMallocDefault));
        c.cur = c.d_lists;
        c.other = c.d_lists + c.list_cap;
        c.headers_zero_flag = &c.headers_zero;
    }
    if (no_alloc) c.headers_zero_flag = &g_noalloc_headers_zero[device];
    if (!c.h_report_dev) HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&c.h_report_dev), c.h_report, 0));

    SearchArgs& a = c.a;
    a = SearchArgs{};
    a.g = g; a.queries = d_queries; a.qmap = nullptr; a.nq_launch = nq;
    a.ef = (uint32_t)ef; a.out_k = out_k; a.flags = flags;
    a.cap_w = ((uint32_t)ef + 2u) & ~1u;
    const bool q8 = (flags & HNSW_AMD_ROWS_Q8) != 0;
    const bool u8 = (flags & HNSW_AMD_ROWS_U8) != 0;
    if ((u8 || q8) && (flags & HNSW_AMD_DIST_SCALAR)) throw std::invalid_argument("search: byte rows go with HNSW_AMD_DIST_SSE only");
    if (u8 && q8) throw std::invalid_argument("search: HNSW_AMD_ROWS_U8 and HNSW_AMD_ROWS_Q8 exclude each other");
    if (u8 && !g.vectors_u8)
        throw std::invalid_argument("search: HNSW_AMD_ROWS_U8 needs base vectors whose values are all integers in 0..255");
    if (q8 && !g.vectors_q8) throw std::logic_error("search: the quantised table was not built");
    const bool ip = (flags & HNSW_AMD_METRIC_IP) != 0;
    if (ip && (u8 || q8 || (flags & HNSW_AMD_DIST_SCALAR)))
        throw std::invalid_argument("search: HNSW_AMD_METRIC_IP goes with float rows and HNSW_AMD_DIST_SSE only");
    a.out_ids = d_ids; a.out_dist = d_dist; a.out_count = d_count; a.stats = d_stats;

    if (dev_max_lds == 0) {
        HIP_CHECK(hipDeviceGetAttribute(&dev_max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
        HIP_CHECK(hipDeviceGetAttribute(&dev_num_cus, hipDeviceAttributeMultiprocessorCount, device));
        if (dev_num_cus < 1) dev_num_cus = 1;
    }

    // ---- what the queries of this batch are expected to need ----
    // Sized from what earlier batches on this index were seen to need (the kernel reports the largest visited
    // count and candidate heap of every launch): the same ef and flags with a twentieth of headroom, another ef by
    // the ratio of observation to estimate at the ef it was made at (data on which searches visit several
    // times the estimate, such as unit-norm vectors under the reference's int32 keys, do so at every ef).
    if (flags != learn_flags || (learn_ef != 0 && learn_ef != ef)) {
        // another kind of search than the last one: put what was learnt aside and take up what an earlier batch of this
        // kind left (a caller that alternates between a few ef values keeps all of them sized)
        if (learn_ef != 0) {
            LearnNote note = {learn_flags, learn_ef, learn_visit, learn_c, learn_mean, learn_age};
            bool put = false;
            for (LearnNote& o : learn_notes) if (o.flags == note.flags && o.ef == note.ef) { o = note; put = true; }
            if (!put) {
                if (learn_notes.size() < 16) learn_notes.push_back(note);
                else learn_notes[learn_notes_next++ % 16u] = note;
            }
        }
        bool found = false;
        for (const LearnNote& o : learn_notes)
            if (o.flags == flags && o.ef == ef) {
                learn_flags = o.flags; learn_ef = o.ef; learn_visit = o.visit; learn_c = o.c; learn_mean = o.mean; learn_age = o.age;
                found = true;
            }
        // (nothing on this kind yet: with the same flags the last kind's figures, scaled, are the first guess)
        if (!found && flags != learn_flags) { learn_flags = flags; learn_ef = 0; learn_visit = 0; learn_c = 0; learn_mean = 0; }
    }
    c.want_visit0 = visit_estimate(ef); c.want_c0 = heap_estimate(ef);
    a.known_visit = 0; a.known_c = 0;
    // what the bulk of the queries needs (bucket form with spill tables: the few beyond it continue in device memory)
    c.want_bulk0 = 0;
    if (learn_ef == ef) {
        if (learn_mean > 0 && learn_mean < learn_visit) c.want_bulk0 = learn_mean + (uint32_t)(tune.bulk_frac * (float)(learn_visit - learn_mean));
        c.want_visit0 = learn_visit + learn_visit / 20u + 16u;
        // (queries report only what goes beyond what is known; one batch in 64 reports in full, so that the known
        // values can also come down)
        if ((++learn_age & 63u) != 0u) { a.known_visit = learn_visit; a.known_c = learn_c; }
        c.want_c0 = learn_c + learn_c / 32u + 4u;          // (learn_c is what the overflow test asked for, not the heap's size)
    } else if (learn_ef != 0) {
        const float rv = std::max(1.0f, 1.2f * (float)learn_visit / (float)visit_estimate(learn_ef));
        const float rc = std::max(1.0f, 1.2f * (float)learn_c / (float)heap_estimate(learn_ef));
        c.want_visit0 = (uint32_t)((float)c.want_visit0 * std::min(rv, 16.0f));
        c.want_c0 = (uint32_t)((float)c.want_c0 * std::min(rc, 16.0f));
    }
    if (tune.want_visit >= 0) c.want_visit0 = (uint32_t)tune.want_visit;
    if (tune.want_bulk >= 0) c.want_bulk0 = (uint32_t)tune.want_bulk;
    c.busy = true;
    try {
        launch_tier(c);
    } catch (...) {
        c.busy = false;
        throw;
    }
    return cp;
}

// One launch of the call: tier 0 over all its queries, tier t > 0 over the queries that outgrew tier t - 1's tables
// (listed in c.cur's twin), on tables 2^t times the size.  Queues the launch's report behind it.
void DeviceIndex::launch_tier(SearchCall& c)
{
    SearchArgs& a = c.a;
    const int tier = c.tier, ef = c.ef;
    const uint32_t flags = c.flags;
    const bool no_alloc = c.no_alloc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(c.stream);
    const bool q8 = (flags & HNSW_AMD_ROWS_Q8) != 0, u8 = (flags & HNSW_AMD_ROWS_U8) != 0;
    const bool ip = (flags & HNSW_AMD_METRIC_IP) != 0, scalar = (flags & HNSW_AMD_DIST_SCALAR) != 0;
    const int max_lds = dev_max_lds, num_cus = dev_num_cus;
    const uint32_t rb_min = (uint32_t)tune.rb_min;
    const uint32_t spill_tables = tune.spill_tables >= 0 ? std::min<uint32_t>(kSpillTables, (uint32_t)tune.spill_tables) : kSpillTables;
    const bool force_bulk = tune.want_bulk >= 0;

    // Exact byte rows through integer dot products where that is provably the same arithmetic: at most 255^2 per used
    // dim must stay below 2^24 (d <= 256 + 15).  The first launch only: a query that is not integer-valued is handed back
    // and re-run with the float arithmetic; an index whose queries mostly are not stops trying.
    const bool dot4 = u8 && tier == 0 && !tune.no_dot4 && !dot4_off && (uint64_t)(g.d & ~15u) * 65025ull < (1ull << 24);
    c.dot4 = dot4;
    // ... and the quantised rows likewise (an approximation either way: the query is rounded to the table's grid, the
    // kernel re-scores its results with the float rows, for which the query's LDS region must hold the float query)
    const bool dot4q = q8 && tier == 0 && !tune.no_dot4 && (uint64_t)(g.d & ~15u) * 65025ull < (1ull << 24);
    a.qpad = dot4 ? ((g.d + 127u) & ~127u) / 4u : (dot4q ? std::max(((g.d + 127u) & ~127u) / 4u, (g.d + 63u) & ~63u) : ((u8 || q8) ? ((g.d + 127u) & ~127u) : ((g.d + 63u) & ~63u)));
    const uint32_t want_visit = c.want_visit0 << tier;
    const bool std_heap = (flags & HNSW_AMD_W_STD_HEAP) != 0 || ef > 256;
    const int wr = std_heap ? 0 : (ef <= 64 ? 1 : (ef <= 128 ? 2 : 4));
    const int rows_kind = dot4 ? 3 : (dot4q ? 4 : (u8 ? 1 : (q8 ? 2 : 0)));
    const bool wide_small = g.d > 256 && (uint64_t)a.nq_launch <= 4ull * (uint64_t)num_cus && !tune.no_deep_rows;   // (the two-wavefront form)
    // Candidate-heap entries of four bytes (key in the top byte, node_order below) where that is provably enough: fewer
    // than 2^24 rows, int32 keys, and no row so long that a query of ordinary length could be 250 away from it (the
    // kernel checks its own query against the longest row and hands it back otherwise).  First launch only.
    const bool packc = rows_kind == 0 && !ip && !(flags & HNSW_AMD_KEYS_FLOAT) && (wr == 1 || wr == 2) && !wide_small && tier == 0 &&
                       g.n < (1u << 24) && g.max_row_norm <= 6.0f && !pack_off && !tune.no_pack_c;
    c.packc = packc;
    const int nl_shape = tune.generic_rows ? 0 : row_shape_loads(g.d, scalar);
    a.c_words = packc ? 1u : 2u;
    a.q_overlay = query_overlaid(nl_shape, rows_kind, scalar, (wide_small && !scalar && !ip) ? 6 : 2) ? 1u : 0u;
    constexpr size_t kMailBytes = 1024;
    SearchKernel kern_p = nullptr, kern_b = nullptr;
    bool deep_rows = false, deep_kernel = false;
    size_t lds = 0;
    const SearchArgs a_in = a;
    // Wavefronts per SIMD of the kernel: 6 (24 queries per CU) where such a kernel exists (integer dot products, ef <= 64)
    // and the scratch of a query then fits a 24th of the LDS, else 4 (16 per CU).
    int wps = (rows_kind >= 3 && wr == 1 && (uint64_t)a.nq_launch >= 16ull * (uint64_t)num_cus && tune.lds_budget < 0) ? 6 : 4;
    for (;; wps = 4) {
        a = a_in;
        a.spill_pool = nullptr; a.spill_tables = 0; a.spill_slots = 0; a.vis_regions = 0; a.vis_magic = 0;
        a.hash_size = hash32_slots(want_visit); a.vis_rb = 0xFFFFFFFFu; a.vis_chunks = 0; a.ovf_size = 0;
        a.cap_c = std::max(((c.want_c0 << tier) + 31u) & ~31u, ((uint32_t)ef + 33u) & ~31u);      // W is sorted in C's space at the end
        if (packc) a.cap_c = std::max(a.cap_c, 2u * a.cap_w);                                     // (W's entries are eight bytes)
        if (!std_heap && a.cap_w > a.cap_c) throw std::logic_error("search: W does not fit the space it borrows for the final sort");
        lds = search_lds_bytes(a, std_heap);
        // wide rows in a batch that leaves most of the chip idle: more loads in flight per query
        // (its 240 registers and two wavefronts per query leave a CU four queries: batches of up to four per CU)
        deep_rows = wide_small;
        kern_p = pick_kernel(scalar, wr, rows_kind, false, ip, g.d, tune.generic_rows, deep_rows, wps, packc);     // probing forms
        kern_b = pick_kernel(scalar, wr, rows_kind, true, ip, g.d, tune.generic_rows, false, wps, packc);      // bucket form
        // LDS a query may use without lowering residency: the float-row kernels need 110-118 VGPRs, i.e. 4 waves per
        // SIMD = 16 one-wave workgroups per CU (the byte-row kernels could run 24, measured no faster than 16), and a
        // batch of fewer than 16 queries per CU needs even fewer slots (bounded, so that clearing the tables stays cheap).
        const uint32_t res_cap = 4u * (uint32_t)wps;
        const uint32_t per_cu = std::min<uint32_t>(res_cap, (a.nq_launch + (uint32_t)num_cus - 1u) / (uint32_t)num_cus);
        size_t budget = std::min<size_t>(kLdsPerCu / per_cu, 40960);
        if (tune.lds_budget >= 0) budget = std::min<size_t>(budget, (size_t)tune.lds_budget);
        // (the deep form runs two wavefronts per query, which talk through a mailbox behind the query's tables)
        deep_kernel = deep_rows && !scalar && !u8 && !q8 && !ip;
        if (deep_kernel && budget > 4 * kMailBytes) budget -= kMailBytes;
        {
            // 16-bit visited slots where they let a CU hold more queries: the probing form (large regions) or the
            // bucket form (small regions read whole, nearly full, plus an overflow table), whichever needs less LDS.
            const bool force16 = tune.visited16, forceb = tune.visitedb;
            if (!tune.visited32 && (force16 || forceb || lds > budget)) {
                auto occupancy = [&](size_t bytes, bool bucket = false) {
                    if (bytes > 65536) return (int)(kLdsPerCu / bytes);
                    // (asked of the runtime once per kernel and size: the call costs microseconds, a batch a few hundred)
                    const void* fn = reinterpret_cast<const void*>(bucket ? kern_b : kern_p);
                    for (const OccupancyNote& o : occupancy_notes) if (o.kernel == fn && o.bytes == bytes) return o.blocks;
                    int occ = 0;
                    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64, bytes));
                    if (occupancy_notes.size() >= 64) occupancy_notes.clear();
                    occupancy_notes.push_back({fn, bytes, occ});
                    return occ;
                };
                auto bucket_args = [&](uint32_t want, BucketShape& sh, SearchArgs& out) -> size_t {
                    if (sh.n != g.n || sh.want != want || sh.rb_min != rb_min) {
                        sh.ok = bucket_shape(g.n, want, rb_min, sh.regions, sh.chunks, sh.ovf);
                        sh.n = g.n; sh.want = want; sh.rb_min = rb_min;
                    }
                    if (!sh.ok) return ~(size_t)0;
                    out = a;
                    out.hash_size = 8u * sh.chunks * sh.regions; out.vis_rb = 0; out.vis_chunks = sh.chunks; out.ovf_size = sh.ovf;
                    out.vis_regions = sh.regions; out.vis_magic = (uint32_t)(0x100000000ull / sh.regions);
                    return search_lds_bytes(out, std_heap);
                };
                SearchArgs b16 = a, bb = a;
                size_t lds16 = ~(size_t)0, ldsb = ~(size_t)0;
                uint32_t rb = 0, slots = 0;
                if (!forceb && visited16_shape(g.n, want_visit, rb_min, rb, slots)) {
                    b16.hash_size = slots; b16.vis_rb = rb;
                    lds16 = search_lds_bytes(b16, std_heap);
                }
                if (!force16) {
                    ldsb = bucket_args(want_visit, shape_full, bb);
                    // Tables for the bulk of the queries and spill tables for the rest, where that lets a CU hold more
                    // queries (first tier only; HNSW_AMD_NO_SPILL: never)
                    const bool spill_ok = tier == 0 && c.want_bulk0 > 0 && !tune.no_spill && (c.d_spill != nullptr || !no_alloc);
                    if (spill_ok && ldsb != ~(size_t)0 && ldsb <= lds16) {
                        SearchArgs bs = a;
                        const size_t ldss = bucket_args(c.want_bulk0, shape_bulk, bs);
                        if (ldss != ~(size_t)0 && (force_bulk || occupancy(ldss, true) > occupancy(ldsb, true))) {
                            if (!c.d_spill) HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c.d_spill), (size_t)kSpillTables * kSpillSlots * 4));
                            bs.spill_pool = c.d_spill; bs.spill_tables = spill_tables; bs.spill_slots = kSpillSlots;
                            bb = bs; ldsb = ldss;
                        }
                    }
                }
                const bool use_b = ldsb < lds16;
                const size_t ldsn = use_b ? ldsb : lds16;
                if (ldsn != ~(size_t)0) {
                    const int occ32 = occupancy(lds), occn = occupancy(ldsn, use_b);
                    // (beyond 16 queries per CU the cheaper 32-bit probe wins: measured on the byte-row kernel, 16 -> 20)
                    if (force16 || forceb || (occn > occ32 && occ32 < (int)res_cap)) { a = use_b ? bb : b16; lds = ldsn; }
                }
            }
        }
        // What the estimate leaves of the budget goes to the tables (a quarter to the heap): a lower load makes the
        // visited probes shorter, and the occasional far-reaching query is spared its re-run.
        if (a.vis_rb == 0xFFFFFFFFu && lds < budget) {
            const uint32_t slack = (uint32_t)(budget - lds);
            const uint32_t more_c = (slack / 4u / 8u) & ~31u;
            a.cap_c += more_c;
            a.hash_size += ((slack - more_c * 8u) / 4u) & ~63u;
            lds = search_lds_bytes(a, std_heap);
        }
        if (lds > (size_t)max_lds) {
            throw std::length_error("search: " + std::to_string(a.nq_launch) +
                                    " queries outgrew the largest on-chip scratch configuration (ef too large for LDS?)");
        }
        if (wps == 4 || lds <= kLdsPerCu / (4u * (size_t)wps)) break;
    }
    const SearchKernel kern = a.vis_chunks ? kern_b : kern_p;
    const bool two_waves = deep_kernel && !a.vis_chunks;
    const unsigned block_threads = two_waves ? 128u : 64u;
    a.mail_off = 0;
    if (two_waves) {
        a.mail_off = (uint32_t)(lds / 4);
        lds += kMailBytes;
        if (lds > (size_t)max_lds) throw std::length_error("search: the query's scratch and the two-wavefront mailbox do not fit the LDS");
    }
    a.fail_count = c.cur; a.fail_list = c.cur + kFailHeader;
    // The headers of both lists are zero between launches: each launch's report kernel clears the one it used.  Only a
    // list that is new, or whose last launch did not get as far as its report, is cleared here.
    if (!*c.headers_zero_flag) {
        HIP_CHECK(hipMemsetAsync(c.cur, 0, 4 * kFailHeader, stream));
        HIP_CHECK(hipMemsetAsync(c.other, 0, 4 * kFailHeader, stream));
    }
    *c.headers_zero_flag = false;
    // The report's words are preset to all ones on the host; the device's copy holds zero in the last word (cleared
    // with the rest of the header, never written by the kernel) and counts that cannot be all ones in the others.
    for (uint32_t w = 0; w < kFailHeader; w++) c.h_report[w] = 0xFFFFFFFFu;
    set_max_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    lds += (size_t)tune.lds_pad;
    if (tune.verbose) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), (int)block_threads, lds);
        char form[96];
        if (a.vis_rb == 0xFFFFFFFFu) snprintf(form, sizeof form, "%u slots of 32 bits", a.hash_size);
        else if (a.vis_chunks) snprintf(form, sizeof form, "%u regions x %u slots of 16 bits + %u overflow%s", a.vis_regions, 8u * a.vis_chunks, a.ovf_size,
                                        a.spill_tables ? " + spill tables" : "");
        else snprintf(form, sizeof form, "%u slots of 16 bits in %u regions", a.hash_size, 1u << a.vis_rb);
        fprintf(stderr, "[hnsw_amd] search launch: %u queries, ef=%d, tier %d (visited: %s for %u ids, C %u x %u B%s), LDS %zu B, resident queries per CU %d%s%s\n",
                a.nq_launch, ef, tier, form, want_visit, a.cap_c, 4u * a.c_words, a.q_overlay ? ", query's copy overlaid" : "", lds, nb,
                two_waves ? ", two wavefronts per query" : "", dot4 ? ", integer dot products" : "");
    }
    c.timed = tune.verbose && !no_alloc;
    if (c.timed) {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        HIP_CHECK(hipEventCreate(&ev0)); HIP_CHECK(hipEventCreate(&ev1));
        c.ev0 = ev0; c.ev1 = ev1;
        HIP_CHECK(hipEventRecord(ev0, stream));
    }
    hipLaunchKernelGGL(kern, dim3(a.nq_launch), dim3(block_threads), lds, stream, a);
    HIP_CHECK(hipGetLastError());
    if (tier == 0) { last_kernel = reinterpret_cast<const void*>(kern); last_lds = (uint32_t)lds; last_block = block_threads; last_wps = wps; last_form = a.vis_rb == 0xFFFFFFFFu ? 0 : (a.vis_chunks ? 2 : 1); }
    if (c.timed) HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(c.ev1), stream));
    if (c.has_copy_back) HIP_CHECK(hipMemcpyAsync(c.copy_back.dst, c.copy_back.src, c.copy_back.bytes, hipMemcpyDeviceToHost, stream));
    hipLaunchKernelGGL(report_kernel, dim3(1), dim3(64), 0, stream, c.cur, c.h_report_dev);
    HIP_CHECK(hipGetLastError());
    c.launched = true;
}

// The launch's report is written into a pinned host buffer by a one-wavefront kernel queued behind the search kernel
// (report_kernel), and the host waits by watching that buffer: no stream synchronisation (the caller's stream is often the
// legacy default stream, whose synchronisation also covers whatever other threads of the process keep adding to it or to
// streams it synchronises with), no event, no copy engine, nothing to wake up — measured 20 us per call less than an event
// wait.  The wait spins for the length of an ordinary batch, then yields the core between looks, then sleeps between them;
// the stream is queried now and then so that an error ends the wait.  Every word read afterwards is itself waited for (all
// were preset to a value the device never sends), so nothing depends on the order in which the words land.
void DeviceIndex::wait_report(SearchCall& c)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(c.stream);
    volatile uint32_t* report = c.h_report;
    auto arrived = [&]() {
        if (report[kFailHeader - 1] != 0u) return false;
        for (uint32_t w = 0; w + 1 < kFailHeader; w++) if (report[w] == 0xFFFFFFFFu) return false;
        return true;
    };
    for (uint32_t spins = 0; !arrived(); spins++) {
        if (spins < (1u << 16)) __builtin_ia32_pause();                 // (a few milliseconds)
        else if (spins < (1u << 16) + 4096u) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 0xFFFu) == 0xFFFu || spins > (1u << 16) + 4096u) {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) {
                // everything queued has run: the report is in memory (or never will be)
                if (!arrived()) HIP_CHECK(hipStreamSynchronize(stream));
                if (!arrived()) throw std::runtime_error("HIP: the launch's report did not arrive");
                break;
            }
            if (q != hipErrorNotReady) HIP_CHECK(q);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *c.headers_zero_flag = true;       // (the report kernel cleared the header it sent)
}

uint32_t DeviceIndex::finish(SearchCall* cp)
{
    if (!cp || !cp->busy) throw std::logic_error("search: finish without a call in flight");
    SearchCall& c = *cp;
    struct Release {                       // whatever happens, the call's slot is free afterwards
        SearchCall& c;
        ~Release() {
            if (c.ev0) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(c.ev0));
            if (c.ev1) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(c.ev1));
            c.ev0 = c.ev1 = nullptr;
            c.busy = false;
        }
    } release{c};
    if (c.nq == 0) return 0;
    HIP_CHECK(hipSetDevice(device));
    SearchArgs& a = c.a;
    for (;;) {
        wait_report(c);
        if (c.timed) {
            float ms = 0.0f;
            HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(c.ev1)));
            HIP_CHECK(hipEventElapsedTime(&ms, reinterpret_cast<hipEvent_t>(c.ev0), reinterpret_cast<hipEvent_t>(c.ev1)));
            fprintf(stderr, "[hnsw_amd] search kernel: %.4f ms for %u queries\n", ms, a.nq_launch);
            (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(c.ev0)); (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(c.ev1));
            c.ev0 = c.ev1 = nullptr;
        }
#ifdef HNSW_CHECKS
        {
            uint32_t bad = 0;
            HIP_CHECK(hipMemcpyFromSymbol(&bad, HIP_SYMBOL(g_check_failures), 4));
            if (bad) throw std::logic_error("search (checking build): " + std::to_string(bad) + " invariant violations in the visited set");
        }
#endif
        const uint32_t* r = c.h_report;
        const uint32_t nfail = r[0];
        c.seen_visit = std::max(c.seen_visit, r[1]);
        c.seen_c = std::max(c.seen_c, r[2]);
        c.seen_sum16 += r[4]; c.seen_done += r[5];
        if (tune.verbose)
            fprintf(stderr, "[hnsw_amd]   %u of %u queries outgrew their scratch, %u used a spill table; largest visited set %u, room asked of C %u\n",
                    nfail, a.nq_launch, std::min(r[3], a.spill_tables), r[1], r[2]);
        if (c.dot4 && c.tier == 0 && r[6] > c.nq / 2u) dot4_off = true;
        if (c.packc && c.tier == 0 && r[6] > c.nq / 2u) pack_off = true;
        if (nfail == 0) break;
        c.retried += nfail;
        a.qmap = c.cur + kFailHeader; a.nq_launch = nfail;
        std::swap(c.cur, c.other);
        c.tier++;
        launch_tier(c);
    }
    // remember what this batch needed (the maxima are those of the queries that finished, i.e. of all of them)
    const int ef = c.ef;
    const uint32_t nq = c.nq;
    const uint32_t seen_mean = c.seen_done ? (uint32_t)(c.seen_sum16 * 16u / c.seen_done) : 0u;
    if (c.flags != learn_flags) {
        // (another kind of search was begun since this one: what it learnt goes to the notes)
        bool put = false;
        for (LearnNote& o : learn_notes) if (o.flags == c.flags && o.ef == ef) {
            o.visit = std::max(o.visit, c.seen_visit); o.c = std::max(o.c, c.seen_c); put = true;
        }
        if (!put && c.seen_visit) {
            const LearnNote note = {c.flags, ef, c.seen_visit, c.seen_c, nq >= 64 ? seen_mean : 0u, 1u};
            if (learn_notes.size() < 16) learn_notes.push_back(note);
            else learn_notes[learn_notes_next++ % 16u] = note;
        }
    } else if (c.seen_visit == 0 && learn_ef != ef) {
        // (a diagnostic build of the kernel reports nothing)
    } else if (learn_ef == ef) {
        // Grows at once, shrinks slowly (one far-reaching query should not cost residency for ever).  A launch that
        // was told the known maxima reports only what went beyond them.
        if (a.known_visit) {
            learn_visit = std::max(learn_visit, c.seen_visit); learn_c = std::max(learn_c, c.seen_c);
        } else if (c.seen_visit) {
            learn_visit = std::max(c.seen_visit, (uint32_t)(((uint64_t)learn_visit * 15u + c.seen_visit) / 16u));
            learn_c = std::max(c.seen_c, (uint32_t)(((uint64_t)learn_c * 15u + c.seen_c) / 16u));
        }
        if (nq >= 64) learn_mean = (uint32_t)(((uint64_t)learn_mean * 3u + seen_mean) / 4u);
    } else if (nq >= 64 || learn_ef == 0) {
        learn_ef = ef; learn_visit = c.seen_visit; learn_c = c.seen_c; learn_mean = nq >= 64 ? seen_mean : 0u;
    }
    return c.retried;
}

uint32_t DeviceIndex::search(const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t flags, uint32_t* d_ids,
                             float* d_dist, uint32_t* d_count, hnsw_amd_query_stats* d_stats, void* stream_v,
                             bool no_alloc, const CopyBack* copy_back)
{
    return finish(begin(d_queries, nq, ef, out_k, flags, d_ids, d_dist, d_count, d_stats, stream_v, no_alloc, copy_back));
}

}  // namespace hnsw_amd
