// Host-side HNSW index: construction, serialisation, flat export.  See hnsw_host.h.
#include "hnsw_host.h"

#include <immintrin.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <fstream>
#include <mutex>
#include <shared_mutex>
#include <thread>

namespace hnsw_amd {

// ---------------------------------------------------------------------------------------
// Level generator.  The reference draws a node's level from glibc rand() with the default
// seed, one call per insert (src/hnsw.cpp:165).  This is glibc's additive-feedback
// generator (random_r TYPE_3, r[i] = r[i-31] + r[i-3], output >> 1) restated so the
// sequence does not depend on, or disturb, the process-wide rand() state.
// ---------------------------------------------------------------------------------------
class GlibcRand {
    uint32_t r_[31];
    int pos_ = 0;   // index of r[i-31] for the next output

public:
    explicit GlibcRand(uint32_t seed = 1) {
        std::vector<uint32_t> t(344);
        t[0] = seed;
        for (int i = 1; i < 31; i++) {
            int64_t w = (16807LL * (int32_t)t[i - 1]) % 2147483647LL;
            if (w < 0) w += 2147483647LL;
            t[i] = (uint32_t)w;
        }
        for (int i = 31; i < 34; i++) t[i] = t[i - 31];
        for (int i = 34; i < 344; i++) t[i] = t[i - 31] + t[i - 3];
        for (int i = 0; i < 31; i++) r_[i] = t[344 - 31 + i];
        pos_ = 0;
    }
    int next() {
        // r_[pos_] holds r[i-31]; r[i-3] sits 28 places further on in the ring
        uint32_t v = r_[pos_] + r_[(pos_ + 28) % 31];
        r_[pos_] = v;
        pos_ = (pos_ + 1) % 31;
        return (int)(v >> 1);
    }
};

// ---------------------------------------------------------------------------------------
// Squared L2 in the exact operation order of the reference's shipped build
// (src/hnsw.h:334-383 compiled with FMA contraction): 4 lane sums, element i in lane i%4,
// fused multiply-add, dimensions beyond 16*floor(d/16) ignored, lanes added left to right.
// ---------------------------------------------------------------------------------------
float distance_sse_order(const float* a, const float* b, uint32_t d)
{
    const uint32_t d16 = d & ~15u;
    __m128 sum = _mm_setzero_ps();
    for (uint32_t i = 0; i < d16; i += 4) {
        __m128 diff = _mm_sub_ps(_mm_loadu_ps(a + i), _mm_loadu_ps(b + i));
        sum = _mm_fmadd_ps(diff, diff, sum);
    }
    alignas(16) float t[4];
    _mm_store_ps(t, sum);
    return ((t[0] + t[1]) + t[2]) + t[3];
}

// The plain loop of src/hnsw.h:387-400 as g++ 11 compiles it for the reference's scalar build: all d dims, one running
// sum in index order; the loop is unrolled by four with each square rounded before it is added, and only the remainder
// loop (the last d % 4 dims) is contracted into fused multiply-adds.  (This file is built with -ffp-contract=off: the
// fused steps are the ones spelled out.)
float distance_scalar_order(const float* a, const float* b, uint32_t d)
{
    float r = 0.0f;
    const uint32_t d4 = d & ~3u;
    uint32_t i = 0;
    for (; i < d4; i++) { const float t = a[i] - b[i]; const float p = t * t; r = r + p; }
    for (; i < d; i++) { const float t = a[i] - b[i]; r = std::fmaf(t, t, r); }
    return r;
}

float HostIndex::dist(const float* a, const float* b) const
{
    return dist_scalar_ ? distance_scalar_order(a, b, vector_size_) : distance_sse_order(a, b, vector_size_);
}

// Four rows against one query at once: same arithmetic per row, the four dependent
// FMA chains interleaved so the core's FMA latency is hidden.
static inline void distance_sse_order_x4(const float* q, const float* r0, const float* r1, const float* r2,
                                         const float* r3, uint32_t d, float out[4])
{
    const uint32_t d16 = d & ~15u;
    __m128 s0 = _mm_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    for (uint32_t i = 0; i < d16; i += 4) {
        __m128 qv = _mm_loadu_ps(q + i);
        __m128 d0 = _mm_sub_ps(qv, _mm_loadu_ps(r0 + i));
        __m128 d1 = _mm_sub_ps(qv, _mm_loadu_ps(r1 + i));
        __m128 d2 = _mm_sub_ps(qv, _mm_loadu_ps(r2 + i));
        __m128 d3 = _mm_sub_ps(qv, _mm_loadu_ps(r3 + i));
        s0 = _mm_fmadd_ps(d0, d0, s0); s1 = _mm_fmadd_ps(d1, d1, s1);
        s2 = _mm_fmadd_ps(d2, d2, s2); s3 = _mm_fmadd_ps(d3, d3, s3);
    }
    alignas(16) float t[4];
    _mm_store_ps(t, s0); out[0] = ((t[0] + t[1]) + t[2]) + t[3];
    _mm_store_ps(t, s1); out[1] = ((t[0] + t[1]) + t[2]) + t[3];
    _mm_store_ps(t, s2); out[2] = ((t[0] + t[1]) + t[2]) + t[3];
    _mm_store_ps(t, s3); out[3] = ((t[0] + t[1]) + t[2]) + t[3];
}

// float -> int32 of the candidate-heap key (src/hnsw.cpp:732,767: a float passed to an
// int32_t tuple member), with the x86 conversion's out-of-range result.
static inline int32_t trunc_key(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}

struct Cand { int32_t key; uint32_t slot; };
// std::*_heap orderings of the reference: W_ max-heap on distance (src/node.h:48-52),
// C min-heap on the int key (src/hnsw.h:210-215).
struct WLess { bool operator()(const Neighbor& a, const Neighbor& b) const { return a.distance < b.distance; } };
struct CGreater { bool operator()(const Cand& a, const Cand& b) const { return a.key > b.key; } };

// ---------------------------------------------------------------------------------------
void LayerStore::reserve_nodes(size_t n)
{
    order.reserve(n); down.reserve(n); cnt.reserve(n); sorted.reserve(n);
    nb_slot.reserve(n * deg); nb_order.reserve(n * deg); nb_dist.reserve(n * deg);
}

uint32_t LayerStore::add_node(uint32_t node_order, uint32_t down_slot)
{
    uint32_t s = count++;
    order.push_back(node_order); down.push_back(down_slot); cnt.push_back(0); sorted.push_back(0);
    nb_slot.resize((size_t)count * deg, kPad); nb_order.resize((size_t)count * deg, kPad);
    nb_dist.resize((size_t)count * deg, 0.0f);
    return s;
}

// per-thread working state of an insert
struct HostIndex::Scratch {
    std::vector<uint32_t> stamp; uint32_t epoch = 0;
    std::vector<Neighbor> W, R, Wov, Rov;
    std::vector<Cand> C;
    std::vector<uint32_t> ids, slots; std::vector<float> dist;
    void reset_visited(size_t n) {
        if (stamp.size() < n) { stamp.assign(n, 0); epoch = 0; }
        if (++epoch == 0) { std::fill(stamp.begin(), stamp.end(), 0); epoch = 1; }
    }
    bool test_and_set(uint32_t id) { if (stamp[id] == epoch) return true; stamp[id] = epoch; return false; }
};

struct SpinGuard {
    std::atomic_flag* f;
    explicit SpinGuard(std::atomic_flag* fl) : f(fl) { if (f) while (f->test_and_set(std::memory_order_acquire)) { _mm_pause(); } }
    ~SpinGuard() { if (f) f->clear(std::memory_order_release); }
};

HostIndex::HostIndex(int M, int Mmax, int efConstruction)
    : M_(M), Mmax_(Mmax), Mmax0_(Mmax * 2), efConstruction_(efConstruction)
{
    ml_ = (float)(1 / std::log(0.8 * M));   // src/hnsw.h:279: double expression stored to float
    rng_ = std::make_unique<GlibcRand>(1);
    scratch_ = std::make_unique<Scratch>();
}

HostIndex::~HostIndex() = default;

static std::atomic<uint64_t> g_rows_epoch{0};

void HostIndex::init(uint32_t vector_size, uint32_t max_node_count)
{
    if (vector_size == 0) throw std::invalid_argument("init: vector_size must be positive");
    vector_size_ = vector_size;
    max_node_count_ = (int)max_node_count;
    vector_data_.assign((size_t)max_node_count * vector_size, 0.0f);
    layers_.clear();
    actual_node_count_ = -1;
    rng_ = std::make_unique<GlibcRand>(1);
    rows_epoch_ = ++g_rows_epoch;
    build_failed_ = false;
    version_++;
}

int HostIndex::draw_level()
{
    // src/hnsw.cpp:165, float arithmetic as written there (std::log on a float is logf)
    float x = ((float)(rng_->next() % 10000 + 1)) / 10000;
    return (int32_t)(-std::log(x) * ml_);
}

void HostIndex::change_layer(int layer, std::vector<Neighbor>& W) const
{
    // src/hnsw.h:575-587: every entry moves to its twin in the layer below
    const LayerStore& L = layers_[layer];
    for (auto& w : W) w.slot = L.down[w.slot];
}

// src/hnsw.cpp:676-717
template <bool MT>
void HostIndex::search_layer_one(int layer, const float* q, Scratch& s)
{
    LayerStore& L = layers_[layer];
    Neighbor cur = s.W.front();
    s.W.pop_back();
    uint32_t loc_slot[64], loc_order[64];
    bool change = true;
    while (change) {
        change = false;
        // the sweep walks the list of the node that was current when the sweep began
        const uint32_t sweep = cur.slot;
        uint32_t cnt;
        const uint32_t *ps, *po;
        if (MT) {
            SpinGuard g(&locks_[layer][sweep]);
            cnt = L.cnt[sweep];
            std::memcpy(loc_slot, &L.nb_slot[(size_t)sweep * L.deg], cnt * 4);
            std::memcpy(loc_order, &L.nb_order[(size_t)sweep * L.deg], cnt * 4);
            ps = loc_slot; po = loc_order;
        } else {
            cnt = L.cnt[sweep];
            ps = &L.nb_slot[(size_t)sweep * L.deg]; po = &L.nb_order[(size_t)sweep * L.deg];
        }
        for (uint32_t j = 0; j < cnt; j++) {
            uint32_t id = po[j];
            if (s.test_and_set(id)) continue;
            float dist = this->dist(row(id), q);
            if (dist < cur.distance) { change = true; cur.slot = ps[j]; cur.distance = dist; cur.node_order = id; }
        }
    }
    s.W.clear();
    s.W.push_back(cur);
}

// src/hnsw.cpp:725-787.  Distances of a candidate's unvisited neighbours are computed as a
// group before the admission loop; the loop itself keeps the reference's order and tests.
template <bool MT>
void HostIndex::search_layer(int layer, const float* q, int ef, Scratch& s)
{
    LayerStore& L = layers_[layer];
    auto& W = s.W; auto& C = s.C;
    C.clear();
    // candidate keys: the reference's int32 truncation, or (option) the float distance itself, stored as its bit
    // pattern: non-negative floats order like their bit patterns read as int32
    const bool fk = build_float_keys_;
    auto key_of = [fk](float dist) -> int32_t { int32_t b; std::memcpy(&b, &dist, 4); return fk ? b : trunc_key(dist); };
    auto key_as_float = [fk](int32_t key) -> float { float v; std::memcpy(&v, &key, 4); return fk ? v : (float)key; };
    for (auto& n : W) { C.push_back(Cand{key_of(n.distance), n.slot}); s.test_and_set(n.node_order); }
    std::make_heap(W.begin(), W.end(), WLess());
    std::make_heap(C.begin(), C.end(), CGreater());
    float f = W.front().distance;
    const uint32_t d = vector_size_;
    s.ids.resize(L.deg); s.slots.resize(L.deg); s.dist.resize(L.deg + 4);
    while (!C.empty()) {
        Cand c = C.front();
        std::pop_heap(C.begin(), C.end(), CGreater());
        C.pop_back();
        if (key_as_float(c.key) > f) break;
        uint32_t m = 0;
        {
            SpinGuard g(MT ? &locks_[layer][c.slot] : nullptr);
            const uint32_t cnt = L.cnt[c.slot];
            const uint32_t* ps = &L.nb_slot[(size_t)c.slot * L.deg];
            const uint32_t* po = &L.nb_order[(size_t)c.slot * L.deg];
            for (uint32_t j = 0; j < cnt; j++) {
                uint32_t id = po[j];
                if (s.test_and_set(id)) continue;
                s.ids[m] = id; s.slots[m] = ps[j]; m++;
                _mm_prefetch((const char*)row(id), _MM_HINT_T0);
            }
        }
        uint32_t j = 0;
        if (!dist_scalar_)
            for (; j + 4 <= m; j += 4)
                distance_sse_order_x4(q, row(s.ids[j]), row(s.ids[j + 1]), row(s.ids[j + 2]), row(s.ids[j + 3]), d, &s.dist[j]);
        for (; j < m; j++) s.dist[j] = this->dist(q, row(s.ids[j]));
        for (j = 0; j < m; j++) {
            const float dist = s.dist[j];
            if (dist < f || W.size() < (size_t)ef) {
                C.push_back(Cand{key_of(dist), s.slots[j]});
                std::push_heap(C.begin(), C.end(), CGreater());
                W.push_back(Neighbor{dist, s.ids[j], s.slots[j]});
                std::push_heap(W.begin(), W.end(), WLess());
                if (W.size() > (size_t)ef) { std::pop_heap(W.begin(), W.end(), WLess()); W.pop_back(); }
                f = W.front().distance;
            }
        }
    }
}

// src/hnsw.cpp:1078-1126 ("Algorithm 4" variant)
void HostIndex::select_neighbors(std::vector<Neighbor>& W, std::vector<Neighbor>& R, int M, int /*layer*/) const
{
    if ((int)W.size() < M) { R = W; return; }
    R.clear();
    std::sort(W.begin(), W.end(), WLess());
    R.push_back(W[0]);
    R.push_back(W[1]);
    int sel = 2;
    for (size_t i = 2; i < W.size() && sel < M; i++) {
        bool keep = true;
        for (size_t j = 0; j < R.size(); j++) {
            float dist = this->dist(row(W[i].node_order), row(R[j].node_order));
            if (dist < W[i].distance) { keep = false; break; }
        }
        if (keep) { R.push_back(W[i]); sel++; }
    }
}

// src/hnsw.cpp:266-312: connect the new node with every selected neighbour, re-selecting a
// neighbour's list when it is full.
template <bool MT>
void HostIndex::link(int layer, uint32_t new_slot, uint32_t id, const std::vector<Neighbor>& R, Scratch& s)
{
    LayerStore& L = layers_[layer];
    const uint32_t cap = layer == 0 ? (uint32_t)Mmax0_ : (uint32_t)Mmax_;
    // The new node's own list first (the reference appends to it inside the loop below, with the
    // same outcome).  In a concurrent build another insert can already have reached this slot through
    // the node's link in the layer above and appended itself here: those entries are kept, behind the
    // selected ones, unless the selection holds them as well.
    {
        SpinGuard g(MT ? &locks_[layer][new_slot] : nullptr);
        if (R.size() > L.deg) throw std::logic_error("link: selected more neighbours than a list holds");
        const size_t base = (size_t)new_slot * L.deg;
        std::vector<Neighbor>& early = s.Rov;
        early.clear();
        if (MT)
            for (uint32_t j = 0; j < L.cnt[new_slot]; j++) early.push_back(Neighbor{L.nb_dist[base + j], L.nb_order[base + j], L.nb_slot[base + j]});
        uint32_t k = 0;
        for (size_t i = 0; i < R.size(); i++, k++) {
            L.nb_slot[base + k] = R[i].slot; L.nb_order[base + k] = R[i].node_order; L.nb_dist[base + k] = R[i].distance;
        }
        for (const Neighbor& x : early) {
            bool dup = false;
            for (size_t i = 0; i < R.size() && !dup; i++) dup = R[i].node_order == x.node_order;
            if (dup || k >= cap) continue;
            L.nb_slot[base + k] = x.slot; L.nb_order[base + k] = x.node_order; L.nb_dist[base + k] = x.distance; k++;
        }
        for (uint32_t j = k; j < L.deg; j++) { L.nb_slot[base + j] = kPad; L.nb_order[base + j] = kPad; L.nb_dist[base + j] = 0.0f; }
        L.cnt[new_slot] = k;
    }
    for (size_t i = 0; i < R.size(); i++) {
        const Neighbor& e = R[i];
        SpinGuard g(MT ? &locks_[layer][e.slot] : nullptr);
        const size_t base = (size_t)e.slot * L.deg;
        uint32_t ecnt = L.cnt[e.slot];
        if (MT) {
            // that neighbour may have linked to this node on its own already (see above)
            bool have = false;
            for (uint32_t j = 0; j < ecnt && !have; j++) have = L.nb_order[base + j] == id;
            if (have) continue;
        }
        if (ecnt == cap) {
            s.Wov.resize(ecnt);
            for (uint32_t j = 0; j < ecnt; j++) s.Wov[j] = Neighbor{L.nb_dist[base + j], L.nb_order[base + j], L.nb_slot[base + j]};
            if (!L.sorted[e.slot]) {
                // the reference sorts the stored list itself once (src/hnsw.cpp:278-282)
                std::sort(s.Wov.begin(), s.Wov.end(), WLess());
                L.sorted[e.slot] = 1;
            }
            s.Wov.push_back(Neighbor{e.distance, id, new_slot});
            select_neighbors(s.Wov, s.Rov, (int)cap, layer);
            uint32_t k = 0;
            for (auto& r : s.Rov) { L.nb_slot[base + k] = r.slot; L.nb_order[base + k] = r.node_order; L.nb_dist[base + k] = r.distance; k++; }
            for (uint32_t j = k; j < L.deg; j++) { L.nb_slot[base + j] = kPad; L.nb_order[base + j] = kPad; L.nb_dist[base + j] = 0.0f; }
            L.cnt[e.slot] = k;
        } else {
            L.nb_slot[base + ecnt] = new_slot; L.nb_order[base + ecnt] = id; L.nb_dist[base + ecnt] = e.distance;
            L.cnt[e.slot] = ecnt + 1;
        }
    }
}

// src/hnsw.cpp:163-334 for the vector with id `id`, level `level`.  new_slots (concurrent
// path only): the slot reserved for this vector in each layer 0..min(level, top).
template <bool MT>
void HostIndex::insert_one(uint32_t id, const float* q, int level, const uint32_t* new_slots, Scratch& s,
                           const uint32_t* cand_ids, const float* cand_dist, uint32_t cand_cnt, bool upper_layers_only, const uint32_t* slot1_of)
{
    // the layer whose candidate list comes from outside: 0, or 1 when layer 0 is not this function's business at all
    const int cand_layer = upper_layers_only ? 1 : 0;
    const int L = (int)layers_.size() - 1;
    s.reset_visited((size_t)max_node_count_);
    s.W.clear();
    if (L >= 0) {
        const LayerStore& top = layers_[L];
        float dist = this->dist(row(top.ep_order), q);
        s.test_and_set(top.ep_order);
        s.W.push_back(Neighbor{dist, top.ep_order, top.ep_slot});
    }
    for (int i = L; i > level; i--) {
        search_layer_one<MT>(i, q, s);
        change_layer(i, s.W);
    }
    uint32_t prev_slot = kPad; int prev_layer = -1;
    uint32_t top_new_slot = kPad;     // the reference's down_node
    // (upper_layers_only: layer 0 is searched and linked elsewhere)
    for (int i = std::min(L, level); i >= (upper_layers_only ? 1 : 0); i--) {
        LayerStore& Ls = layers_[i];
        uint32_t ns;
        if (MT) ns = new_slots[i];
        else ns = Ls.add_node(id, kPad);
        if (level > L && i == L) top_new_slot = ns;
        s.reset_visited((size_t)max_node_count_);
        // (concurrent build: another insert may have reached this vector through its node in the layer above and linked to
        // it already, so the search could arrive at the vector itself — it is not its own neighbour)
        if (MT) s.test_and_set(id);
        if (i == cand_layer && cand_cnt > 0 && (i == 0 || slot1_of)) {
            // this layer's candidate list comes from outside (a search from the top of the graph as it stood; the reference
            // seeds this search with the whole result of the layer above instead, src/hnsw.cpp:246-262)
            s.W.clear();
            for (uint32_t c = 0; c < cand_cnt; c++)
                if (cand_ids[c] != id) {
                    const uint32_t sl = i == 0 ? cand_ids[c] : slot1_of[cand_ids[c]];                 // (layer 0: slot = node_order)
                    if (sl != kPad) s.W.push_back(Neighbor{cand_dist[c], cand_ids[c], sl});
                }
        } else {
            search_layer<MT>(i, q, efConstruction_, s);
        }
        select_neighbors(s.W, s.R, M_, i);
        link<MT>(i, ns, id, s.R, s);
        if (!MT && prev_layer >= 0) layers_[prev_layer].down[prev_slot] = ns;      // (the concurrent path set it in add_node, before any other thread could see the node)
        prev_slot = ns; prev_layer = i;
        if (i > 0) change_layer(i, s.W);
    }
    if (level > L) {
        // new top layers, each holding only this vector, which stays their entry point
        // for good (src/hnsw.cpp:324-333)
        uint32_t below = top_new_slot;
        for (int i = L + 1; i <= level; i++) {
            layers_.emplace_back();
            LayerStore& Ls = layers_.back();
            Ls.deg = i == 0 ? (uint32_t)Mmax0_ : (uint32_t)Mmax_;
            uint32_t ns = Ls.add_node(id, below);
            Ls.ep_slot = ns; Ls.ep_order = id;
            below = ns;
        }
    }
}

void HostIndex::insert(const float* q)
{
    if (max_node_count_ <= 0) throw std::invalid_argument("insert: init() has not been called");
    if (build_failed_) throw std::logic_error("insert: an earlier concurrent build failed half way; init() the index again");
    // the reference draws the level before it checks capacity (src/hnsw.cpp:165,175-179)
    int level = draw_level();
    if (actual_node_count_ + 1 >= max_node_count_) throw std::runtime_error("HNSW: exceeded maximal number of nodes\n");
    uint32_t id = (uint32_t)++actual_node_count_;
    std::memcpy(&vector_data_[(size_t)id * vector_size_], q, sizeof(float) * vector_size_);
    insert_one<false>(id, q, level, nullptr, *scratch_);
    version_++;
}

void HostIndex::ensure_locks()
{
    locks_.resize(layers_.size());
    lock_cap_.resize(layers_.size(), 0);
    for (size_t l = 0; l < layers_.size(); l++) {
        size_t need = layers_[l].order.capacity();
        if (lock_cap_[l] < need) {
            locks_[l].reset(new std::atomic_flag[need]);
            for (size_t i = 0; i < need; i++) locks_[l][i].clear();
            lock_cap_[l] = need;
        }
    }
}

void HostIndex::insert_batch(const float* rows, uint32_t n, int threads, CandidateSearchFn candidates, void* ctx, const Layer0Linker* layer0)
{
    if (max_node_count_ <= 0) throw std::invalid_argument("insert_batch: init() has not been called");
    if (build_failed_) throw std::logic_error("insert_batch: an earlier concurrent build failed half way; init() the index again");
    if ((int64_t)actual_node_count_ + 1 + n > (int64_t)max_node_count_)
        throw std::runtime_error("HNSW: exceeded maximal number of nodes\n");
    if (threads <= 1 || n < 2048) {
        for (uint32_t i = 0; i < n; i++) insert(rows + (size_t)i * vector_size_);
        return;
    }
    // Concurrent construction.  Levels are still drawn in id order from the same
    // generator, so layer membership and slots equal the sequential build's; what
    // differs is which already-inserted nodes a search can see.
    const uint32_t first = (uint32_t)(actual_node_count_ + 1);
    std::vector<int> level(n);
    for (uint32_t i = 0; i < n; i++) level[i] = draw_level();
    std::memcpy(&vector_data_[(size_t)first * vector_size_], rows, sizeof(float) * (size_t)n * vector_size_);

    // a sequential prefix builds the upper structure of an index that is still small
    uint32_t warm = first >= 1024u ? 0u : std::min<uint32_t>(n, 1024u - first);
    for (uint32_t i = 0; i < warm; i++) {
        uint32_t id = (uint32_t)++actual_node_count_;
        insert_one<false>(id, row(id), level[i], nullptr, *scratch_);
    }
    // Reserve every remaining slot.  A vector that would open a new top layer is
    // inserted alone (exclusive lock) so the layer list only grows while no search runs.
    int top = (int)layers_.size() - 1;
    std::vector<std::vector<uint32_t>> slots(n);      // [i][layer] for layers 0..min(level, top at that time)
    std::vector<uint8_t> raises(n, 0);
    {
        std::vector<uint32_t> next(layers_.size());
        for (size_t l = 0; l < layers_.size(); l++) next[l] = layers_[l].count;
        int cur_top = top;
        for (uint32_t i = warm; i < n; i++) {
            int lv = level[i];
            if (lv > cur_top) { raises[i] = 1; }
            int upto = std::min(lv, cur_top);
            slots[i].resize(upto + 1);
            for (int l = 0; l <= upto; l++) slots[i][l] = next[l]++;
            if (lv > cur_top) { for (int l = cur_top + 1; l <= lv; l++) next.push_back(1); cur_top = lv; }
        }
        for (size_t l = 0; l < layers_.size(); l++) {
            LayerStore& Ls = layers_[l];
            size_t total = next[l];
            Ls.reserve_nodes(total);
        }
    }
    // Process in segments separated by layer-raising vectors.  If anything throws from here on (the candidate
    // searches can: device memory, a HIP error) the segment's nodes are in the layers but not all linked: the index
    // is marked as failed, and the version moves so that no device copy passes for current.
    uint32_t seg_begin = warm;
    std::vector<std::unique_ptr<Scratch>> scratch_pool;      // one per worker thread, for the whole call
    std::vector<uint32_t> slot1_of;                   // (device-linked build) node_order -> slot in layer 1
    std::vector<uint32_t> upper_rank(layer0 ? n : 0u, 0u);   // ... rank of a vector among those of its sub-batch that live in layer 1
    try {
    while (seg_begin < n) {
        uint32_t seg_end = seg_begin;
        while (seg_end < n && !raises[seg_end]) seg_end++;
        if (seg_end > seg_begin) {
            // add the segment's nodes to every layer they live in, in id order
            for (uint32_t i = seg_begin; i < seg_end; i++) {
                uint32_t id = first + i;
                for (size_t l = 0; l < slots[i].size(); l++) {
                    uint32_t sl = layers_[l].add_node(id, l == 0 ? kPad : slots[i][l - 1]);
                    if (sl != slots[i][l]) throw std::logic_error("insert_batch: slot reservation out of step");
                }
            }
            ensure_locks();
            actual_node_count_ = (int)(first + seg_end - 1);
            if (layer0 && layers_.size() > 1) {
                // layer-1 slot of every vector that has one (candidate lists name vectors, lists hold slots)
                const LayerStore& L1 = layers_[1];
                slot1_of.assign((size_t)max_node_count_, kPad);
                for (uint32_t sl = 0; sl < L1.count; sl++) slot1_of[L1.order[sl]] = sl;
            }
            // vectors [lo, hi) of the batch, by a pool of threads; `cand` (optional): for vector lo + j the result list
            // of a layer-0 search with ef = efConstruction_ run elsewhere, cand_cnt[j] entries at cand_ids/cand_dist + j * ef
            auto run_range = [&](uint32_t lo, uint32_t hi, const uint32_t* cand_ids, const float* cand_dist, const uint32_t* cand_cnt,
                                 bool upper_only = false) {
                std::atomic<uint32_t> cursor{lo};
                std::vector<std::thread> pool;
                std::exception_ptr err; std::mutex err_mu;
                int nt = std::min<int>(threads, 256);
                // (a thread's working state outlives the sub-batch: its visited stamps are as long as the index)
                while ((int)scratch_pool.size() < nt) scratch_pool.emplace_back(new Scratch());
                for (int t = 0; t < nt; t++) {
                    pool.emplace_back([&, t]() {
                        Scratch& s = *scratch_pool[t];
                        try {
                            for (;;) {
                                uint32_t i = cursor.fetch_add(1);
                                if (i >= hi) break;
                                uint32_t id = first + i;
                                const uint32_t j = i - lo;
                                const size_t at = (size_t)j * (size_t)efConstruction_;
                                if (upper_only) {
                                    // layer 0 is linked elsewhere: the layers above for the one vector in sixteen that has any, with
                                    // layer 1's candidate list from there as well (cand_cnt is indexed by the vector's rank among them)
                                    if (slots[i].size() > 1) {
                                        const uint32_t u = upper_rank[i];
                                        const size_t ua = (size_t)u * (size_t)efConstruction_;
                                        if (cand_ids && cand_cnt[u] > 0)
                                            insert_one<true>(id, row(id), level[i], slots[i].data(), s, cand_ids + ua, cand_dist + ua, cand_cnt[u], true, slot1_of.data());
                                        else
                                            insert_one<true>(id, row(id), level[i], slots[i].data(), s, nullptr, nullptr, 0, true);
                                    }
                                } else if (cand_ids && slots[i].size() == 1 && cand_cnt[j] > 0) {
                                    // a vector that lives in layer 0 only: the reference's insert would run exactly that
                                    // search (src/hnsw.cpp:241-262); select and link as usual
                                    s.W.clear();
                                    for (uint32_t c = 0; c < cand_cnt[j]; c++)
                                        if (cand_ids[at + c] != id)
                                            s.W.push_back(Neighbor{cand_dist[at + c], cand_ids[at + c], cand_ids[at + c]});   // (layer 0: slot = node_order)
                                    select_neighbors(s.W, s.R, M_, 0);
                                    link<true>(0, slots[i][0], id, s.R, s);
                                } else if (cand_ids) {
                                    // upper layers on the host (they are small), layer 0 from the same candidate list
                                    insert_one<true>(id, row(id), level[i], slots[i].data(), s, cand_ids + at, cand_dist + at, cand_cnt[j]);
                                } else {
                                    insert_one<true>(id, row(id), level[i], slots[i].data(), s);
                                }
                            }
                        } catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
                    });
                }
                for (auto& th : pool) th.join();
                if (err) std::rethrow_exception(err);
            };
            // (a sixteenth of what is linked, 16384 at most — beyond a million linked vectors a sixty-fourth, 65536 at most, so
            // that the per-sub-batch hand-overs stay a small part of a large build; while the graph is small a quarter of it, up
            // to 1024.  With layer 0 on the device a sub-batch costs little besides its kernels, and the vectors of a sub-batch
            // do not see each other: a sixty-fourth throughout — SIFT-shaped 1M, recall@10 at ef = 50: 0.9547 / 0.9554 / 0.9559 /
            // 0.9562 with a 16th / 32nd / 64th / 128th, 0.9563 from the host-only build; 1.3 / 1.3 / 1.6 / 1.9 s.)
            const uint32_t sb_div = layer0 ? 64u : 16u;
            auto sub_batch = [&](uint32_t pos) {
                const uint32_t linked = first + pos;
                const uint32_t cap = std::min<uint32_t>(65536u, std::max<uint32_t>(16384u, linked / 64u));
                return std::min<uint32_t>(cap, std::max<uint32_t>(std::min<uint32_t>(1024u, std::max<uint32_t>(64u, linked / 4u)), linked / sb_div));
            };
            if (layer0) {
                // Layer 0 on the device from here to the end of the segment.  A sub-batch is searched, selected and linked
                // there on the graph as it stands (its vectors do not see each other); only after that are its vectors linked
                // into the layers above on the host — while the device works on the next sub-batch, which descends through
                // the upper layers as they were handed over before that linking began.  So no search ever descends into a
                // vector whose layer-0 list does not exist yet.
                layer0->from_host(layer0->ctx);
                uint32_t prev_pos = 0, prev_end = 0;                      // the sub-batch whose upper layers are still to be linked
                // candidate lists of the layer-1 searches: of the sub-batch in the device's hands, and of the one before it
                std::vector<uint32_t> uids, cids[2], ccnt[2]; std::vector<float> cdist[2];
                int cur = 0;
                for (uint32_t pos = seg_begin; pos < seg_end;) {
                    const uint32_t end = std::min<uint32_t>(seg_end, pos + sub_batch(pos));
                    const auto t0 = std::chrono::steady_clock::now();
                    uids.clear();
                    for (uint32_t i = pos; i < end; i++)
                        if (slots[i].size() > 1) { upper_rank[i] = (uint32_t)uids.size(); uids.push_back(first + i); }
                    cids[cur].resize(uids.size() * (size_t)efConstruction_); cdist[cur].resize(uids.size() * (size_t)efConstruction_);
                    ccnt[cur].assign(uids.size(), 0u);
                    layer0->upper_layers(layer0->ctx);
                    std::exception_ptr dev_err;
                    double dev_s = 0.0;
                    std::thread dev_thread([&]() {
                        try {
                            layer0->step(layer0->ctx, first + pos, end - pos, uids.data(), (uint32_t)uids.size(), cids[cur].data(), cdist[cur].data(),
                                         ccnt[cur].data());
                        } catch (...) { dev_err = std::current_exception(); }
                        dev_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    });
                    const auto t1 = std::chrono::steady_clock::now();
                    std::exception_ptr host_err;
                    try {
                        if (prev_end > prev_pos) run_range(prev_pos, prev_end, cids[cur ^ 1].data(), cdist[cur ^ 1].data(), ccnt[cur ^ 1].data(), true);
                    } catch (...) { host_err = std::current_exception(); }
                    const auto t2 = std::chrono::steady_clock::now();
                    dev_thread.join();
                    if (dev_err) std::rethrow_exception(dev_err);
                    if (host_err) std::rethrow_exception(host_err);
                    assisted_search_s += dev_s;
                    assisted_link_s += std::chrono::duration<double>(t2 - t1).count();
                    assisted_batches++;
                    prev_pos = pos; prev_end = end;
                    cur ^= 1;
                    pos = end;
                }
                if (prev_end > prev_pos) {
                    const auto t1 = std::chrono::steady_clock::now();
                    run_range(prev_pos, prev_end, cids[cur ^ 1].data(), cdist[cur ^ 1].data(), ccnt[cur ^ 1].data(), true);
                    assisted_link_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
                }
                layer0->to_host(layer0->ctx);
            } else if (!candidates) {
                run_range(seg_begin, seg_end, nullptr, nullptr, nullptr);
            } else {
                // Candidate lists from outside (the GPU): the graph is frozen while a sub-batch is searched, so the vectors
                // of a sub-batch do not see each other; sub-batches stay small next to what is linked already.
                std::vector<uint32_t> cids, ccnt; std::vector<float> cdist;
                for (uint32_t pos = seg_begin; pos < seg_end;) {
                    const uint32_t end = std::min<uint32_t>(seg_end, pos + sub_batch(pos));
                    const uint32_t cnt = end - pos;
                    cids.resize((size_t)cnt * efConstruction_); cdist.resize((size_t)cnt * efConstruction_); ccnt.assign(cnt, 0);
                    const auto t0 = std::chrono::steady_clock::now();
                    candidates(ctx, row(first + pos), cnt, efConstruction_, cids.data(), cdist.data(), ccnt.data());
                    const auto t1 = std::chrono::steady_clock::now();
                    run_range(pos, end, cids.data(), cdist.data(), ccnt.data());
                    const auto t2 = std::chrono::steady_clock::now();
                    assisted_search_s += std::chrono::duration<double>(t1 - t0).count();
                    assisted_link_s += std::chrono::duration<double>(t2 - t1).count();
                    assisted_batches++;
                    pos = end;
                }
            }
        }
        if (seg_end < n) {
            // the layer-raising vector, alone
            uint32_t i = seg_end;
            uint32_t id = first + i;
            actual_node_count_ = (int)id;
            insert_one<false>(id, row(id), level[i], nullptr, *scratch_);
            seg_end++;
        }
        seg_begin = seg_end;
    }
    } catch (...) {
        build_failed_ = true;
        version_++;
        throw;
    }
    actual_node_count_ = (int)(first + n - 1);
    version_++;
    if ((candidates || layer0) && getenv("HNSW_AMD_VERBOSE"))
        fprintf(stderr, "[hnsw_amd] assisted build: %u sub-batches so far, %.2f s in %s, %.2f s selecting and linking on the host%s\n",
                assisted_batches, assisted_search_s, layer0 ? "the device steps (search, selection, linking of layer 0)" : "candidate searches (graph refresh included)",
                assisted_link_s, layer0 ? " (upper layers, at the same time as the device steps)" : "");
}

// ---------------------------------------------------------------------------------------
// Flat export / import
// ---------------------------------------------------------------------------------------
void HostIndex::export_flat(FlatGraph& g, bool with_layer0, bool with_upper_lists) const
{
    const uint32_t nl = (uint32_t)layers_.size();
    g = FlatGraph();
    g.n = size(); g.d = vector_size_; g.nlayers = nl; g.deg0 = (uint32_t)Mmax0_; g.degu = (uint32_t)Mmax_;
    if (nl == 0) { g.up_off = {0}; return; }
    const LayerStore& L0 = layers_[0];
    if (L0.count != g.n) throw std::logic_error("export_flat: layer 0 does not hold every vector");
    if (with_layer0) {
        g.adj0.assign(L0.nb_order.begin(), L0.nb_order.begin() + (size_t)g.n * g.deg0);
        g.cnt0.assign(L0.cnt.begin(), L0.cnt.begin() + g.n);
    }
    g.up_off.assign(nl + 1, 0);
    for (uint32_t l = 1; l < nl; l++) g.up_off[l + 1] = g.up_off[l] + layers_[l].count;
    const uint32_t U = g.up_off[nl];
    g.up_order.resize(U); g.up_down.resize(U);
    if (with_upper_lists) {
        g.up_cnt.resize(U);
        g.up_adj_slot.assign((size_t)U * g.degu, kPad); g.up_adj_order.assign((size_t)U * g.degu, kPad);
    }
    g.ep_slot.resize(nl); g.ep_order.resize(nl);
    for (uint32_t l = 0; l < nl; l++) { g.ep_slot[l] = layers_[l].ep_slot; g.ep_order[l] = layers_[l].ep_order; }
    for (uint32_t l = 1; l < nl; l++) {
        const LayerStore& L = layers_[l];
        const uint32_t off = g.up_off[l];
        for (uint32_t s = 0; s < L.count; s++) {
            g.up_order[off + s] = L.order[s]; g.up_down[off + s] = L.down[s];
            if (!with_upper_lists) continue;
            g.up_cnt[off + s] = L.cnt[s];
            for (uint32_t j = 0; j < L.cnt[s]; j++) {
                g.up_adj_slot[(size_t)(off + s) * g.degu + j] = L.nb_slot[(size_t)s * L.deg + j];
                g.up_adj_order[(size_t)(off + s) * g.degu + j] = L.nb_order[(size_t)s * L.deg + j];
            }
        }
    }
}

void HostIndex::import_flat(const FlatGraph& g, const float* vectors)
{
    if (g.d == 0 || g.nlayers == 0 || g.n == 0) throw std::invalid_argument("import_flat: empty graph");
    if (g.deg0 == 0 || g.degu == 0) throw std::invalid_argument("import_flat: zero degree");
    init(g.d, g.n);
    Mmax0_ = (int)g.deg0; Mmax_ = (int)g.degu;
    std::memcpy(vector_data_.data(), vectors, sizeof(float) * (size_t)g.n * g.d);
    actual_node_count_ = (int)g.n - 1;
    layers_.resize(g.nlayers);
    for (uint32_t l = 0; l < g.nlayers; l++) {
        LayerStore& L = layers_[l];
        L.deg = l == 0 ? g.deg0 : g.degu;
        const uint32_t cnt = l == 0 ? g.n : g.up_off[l + 1] - g.up_off[l];
        const uint32_t below = l == 0 ? 0 : (l == 1 ? g.n : g.up_off[l] - g.up_off[l - 1]);
        if (cnt == 0) throw std::invalid_argument("import_flat: empty layer");
        L.reserve_nodes(cnt);
        for (uint32_t s = 0; s < cnt; s++) {
            const uint32_t u = l == 0 ? s : g.up_off[l] + s;
            const uint32_t order = l == 0 ? s : g.up_order[u];
            const uint32_t down = l == 0 ? kPad : g.up_down[u];
            const uint32_t c = l == 0 ? g.cnt0[s] : g.up_cnt[u];
            if (order >= g.n || c > L.deg || (l > 0 && down >= below)) throw std::invalid_argument("import_flat: index out of range");
            L.add_node(order, down);
            L.cnt[s] = c;
            for (uint32_t j = 0; j < c; j++) {
                uint32_t no = l == 0 ? g.adj0[(size_t)s * g.deg0 + j] : g.up_adj_order[(size_t)u * g.degu + j];
                uint32_t ns = l == 0 ? no : g.up_adj_slot[(size_t)u * g.degu + j];
                if (no >= g.n || ns >= cnt) throw std::invalid_argument("import_flat: neighbour out of range");
                for (uint32_t jj = 0; jj < j; jj++)
                    if (L.nb_order[(size_t)s * L.deg + jj] == no) throw std::invalid_argument("import_flat: duplicate neighbour in a list");
                L.nb_slot[(size_t)s * L.deg + j] = ns; L.nb_order[(size_t)s * L.deg + j] = no;
            }
        }
        L.ep_slot = g.ep_slot[l]; L.ep_order = g.ep_order[l];
        if (L.ep_slot >= cnt || L.ep_order >= g.n || L.order[L.ep_slot] != L.ep_order)
            throw std::invalid_argument("import_flat: bad entry point");
    }
    // a neighbour's slot must stand for the neighbour's node_order
    for (uint32_t l = 1; l < g.nlayers; l++) {
        const LayerStore& L = layers_[l];
        for (uint32_t s = 0; s < L.count; s++)
            for (uint32_t j = 0; j < L.cnt[s]; j++)
                if (L.order[L.nb_slot[(size_t)s * L.deg + j]] != L.nb_order[(size_t)s * L.deg + j])
                    throw std::invalid_argument("import_flat: neighbour slot and node_order disagree");
        const LayerStore& B = layers_[l - 1];
        for (uint32_t s = 0; s < L.count; s++)
            if (B.order[L.down[s]] != L.order[s]) throw std::invalid_argument("import_flat: down link points at another vector");
    }
    version_++;
}

// ---------------------------------------------------------------------------------------
// saveGraph / loadGraph in the reference's file format (src/hnsw.cpp:1189-1232, 1254-1302):
//   int max_node_count; uint32 vector_size; int actual_node_count; float[max*d];
//   size_t nlayers; per layer: size_t n_nodes, per node size_t degree;
//   per layer: uint32 ep_node_order, int ep_global_index, per node: int lower_global_index,
//   size_t degree, degree x (int neighbour_global_index, uint32 neighbour_node_order).
// "global index" = position of a node in the concatenation of all layers' node lists.
// ---------------------------------------------------------------------------------------
void HostIndex::save_graph(const char* filename) const
{
    std::ofstream f(filename, std::ios::binary);
    if (!f.is_open()) throw std::ios_base::failure(std::string("Save file ") + filename + " was not successfully opened");
    auto wr = [&](const void* p, size_t n) { f.write(reinterpret_cast<const char*>(p), (std::streamsize)n); };
    int32_t maxn = max_node_count_, act = actual_node_count_;
    uint32_t d = vector_size_;
    wr(&maxn, 4); wr(&d, 4); wr(&act, 4);
    wr(vector_data_.data(), sizeof(float) * (size_t)maxn * d);
    uint64_t nl = layers_.size();
    wr(&nl, 8);
    std::vector<uint64_t> base(layers_.size() + 1, 0);
    for (size_t l = 0; l < layers_.size(); l++) {
        const LayerStore& L = layers_[l];
        uint64_t c = L.count;
        wr(&c, 8);
        for (uint32_t s = 0; s < L.count; s++) { uint64_t deg = L.cnt[s]; wr(&deg, 8); }
        base[l + 1] = base[l] + L.count;
    }
    for (size_t l = 0; l < layers_.size(); l++) {
        const LayerStore& L = layers_[l];
        wr(&L.ep_order, 4);
        int32_t ep = (int32_t)(base[l] + L.ep_slot);
        wr(&ep, 4);
        for (uint32_t s = 0; s < L.count; s++) {
            // layer-0 nodes have no lower twin; the reference writes 0 for them
            int32_t low = l == 0 ? 0 : (int32_t)(base[l - 1] + L.down[s]);
            wr(&low, 4);
            uint64_t deg = L.cnt[s];
            wr(&deg, 8);
            for (uint32_t j = 0; j < L.cnt[s]; j++) {
                int32_t gi = (int32_t)(base[l] + L.nb_slot[(size_t)s * L.deg + j]);
                wr(&gi, 4); wr(&L.nb_order[(size_t)s * L.deg + j], 4);
            }
        }
    }
    if (!f) throw std::ios_base::failure(std::string("write to ") + filename + " failed");
}

// Everything a search trusts about a graph that came from a file: the same checks import_flat applies to
// flat arrays.  (The GPU kernels index the vector table and the adjacency arrays with these numbers.)
void HostIndex::validate_loaded(const char* filename) const
{
    auto bad = [&](const char* what) { return std::domain_error(std::string(filename) + ": " + what); };
    const uint32_t n = layers_[0].count;
    if (n == 0 || n > (uint32_t)max_node_count_) throw bad("layer 0 is empty or larger than the vector table");
    for (size_t l = 0; l < layers_.size(); l++) {
        const LayerStore& L = layers_[l];
        if (L.deg > 64) throw bad("neighbour lists longer than 64 are not supported");
        if (l > 0 && L.count > layers_[l - 1].count) throw bad("a layer larger than the one below it");
        for (uint32_t s = 0; s < L.count; s++) {
            if (L.order[s] >= n) throw bad("node outside layer 0");
            const uint32_t* no = &L.nb_order[(size_t)s * L.deg];
            const uint32_t* ns = &L.nb_slot[(size_t)s * L.deg];
            for (uint32_t j = 0; j < L.cnt[s]; j++) {
                if (no[j] >= n || ns[j] >= L.count) throw bad("neighbour outside the filled part of the index");
                if (L.order[ns[j]] != no[j]) throw bad("a neighbour's position and its node_order disagree");
                for (uint32_t jj = 0; jj < j; jj++)
                    if (no[jj] == no[j]) throw bad("duplicate neighbour in a list");
            }
        }
        if (L.ep_slot >= L.count || L.order[L.ep_slot] != L.ep_order) throw bad("entry point and its node_order disagree");
    }
}

// The file is read into a separate object and adopted only when it parsed and validated: an error leaves this
// index as it was.
void HostIndex::load_graph(const char* filename)
{
    HostIndex tmp(M_, Mmax_ , efConstruction_);
    tmp.Mmax0_ = Mmax0_;
    tmp.parse_graph_file(filename);
    tmp.validate_loaded(filename);
    Mmax_ = tmp.Mmax_; Mmax0_ = tmp.Mmax0_;
    vector_size_ = tmp.vector_size_; max_node_count_ = tmp.max_node_count_; actual_node_count_ = tmp.actual_node_count_;
    layers_ = std::move(tmp.layers_);
    vector_data_ = std::move(tmp.vector_data_);
    rows_epoch_ = tmp.rows_epoch_;      // (unique in the process: tmp's init drew it)
    build_failed_ = false;
    rng_ = std::make_unique<GlibcRand>(1);
    locks_.clear(); lock_cap_.clear();
    version_++;
}

void HostIndex::parse_graph_file(const char* filename)
{
    std::ifstream f(filename, std::ios::binary);
    if (!f.is_open()) throw std::ios_base::failure(std::string("Save file ") + filename + " was not successfully opened");
    auto rd = [&](void* p, size_t n) {
        f.read(reinterpret_cast<char*>(p), (std::streamsize)n);
        if ((size_t)f.gcount() != n) throw std::ios_base::failure(std::string(filename) + ": truncated graph file");
    };
    int32_t maxn, act; uint32_t d;
    rd(&maxn, 4); rd(&d, 4); rd(&act, 4);
    if (maxn <= 0 || d == 0 || d > (1u << 20)) throw std::ios_base::failure(std::string(filename) + ": implausible header");
    init(d, (uint32_t)maxn);
    rd(vector_data_.data(), sizeof(float) * (size_t)maxn * d);
    uint64_t nl;
    rd(&nl, 8);
    if (nl == 0 || nl > 64) throw std::ios_base::failure(std::string(filename) + ": implausible layer count");
    layers_.resize(nl);
    std::vector<uint64_t> base(nl + 1, 0);
    for (size_t l = 0; l < nl; l++) {
        LayerStore& L = layers_[l];
        L.deg = l == 0 ? (uint32_t)Mmax0_ : (uint32_t)Mmax_;
        uint64_t c;
        rd(&c, 8);
        if (c == 0 || c > (uint64_t)maxn) throw std::ios_base::failure(std::string(filename) + ": implausible layer size");
        std::vector<uint64_t> degs(c);
        rd(degs.data(), 8 * c);
        for (uint64_t s = 0; s < c; s++) if (degs[s] > L.deg) {
            // a file written with a larger Mmax than this object was constructed with (64 is what the searches,
            // host and device, are built for)
            if (degs[s] > 64) throw std::domain_error(std::string(filename) + ": neighbour lists longer than 64 are not supported");
            L.deg = (uint32_t)degs[s];
        }
        L.reserve_nodes(c);
        for (uint64_t s = 0; s < c; s++) L.add_node(kPad, kPad);
        base[l + 1] = base[l] + c;
    }
    if (layers_[0].deg > (uint32_t)Mmax0_) Mmax0_ = (int)layers_[0].deg;
    for (size_t l = 1; l < nl; l++) if (layers_[l].deg > (uint32_t)Mmax_) Mmax_ = (int)layers_[l].deg;
    for (size_t l = 1; l < nl; l++) if (layers_[l].deg != (uint32_t)Mmax_) {
        // widen to the common upper-layer degree (lists are still empty here)
        LayerStore& L = layers_[l];
        L.deg = (uint32_t)Mmax_;
        L.nb_slot.assign((size_t)L.count * L.deg, kPad); L.nb_order.assign((size_t)L.count * L.deg, kPad);
        L.nb_dist.assign((size_t)L.count * L.deg, 0.0f);
    }
    for (size_t l = 0; l < nl; l++) {
        LayerStore& L = layers_[l];
        int32_t ep;
        rd(&L.ep_order, 4); rd(&ep, 4);
        if (ep < (int64_t)base[l] || ep >= (int64_t)base[l + 1]) throw std::ios_base::failure(std::string(filename) + ": entry point outside its layer");
        L.ep_slot = (uint32_t)(ep - base[l]);
        for (uint32_t s = 0; s < L.count; s++) {
            int32_t low; uint64_t deg;
            rd(&low, 4); rd(&deg, 8);
            if (deg > L.deg) throw std::ios_base::failure(std::string(filename) + ": degree larger than declared");
            if (l > 0) {
                if (low < (int64_t)base[l - 1] || low >= (int64_t)base[l]) throw std::ios_base::failure(std::string(filename) + ": lower link outside the layer below");
                L.down[s] = (uint32_t)(low - base[l - 1]);
            }
            L.cnt[s] = (uint32_t)deg;
            for (uint32_t j = 0; j < deg; j++) {
                int32_t gi; uint32_t no;
                rd(&gi, 4); rd(&no, 4);
                if (gi < (int64_t)base[l] || gi >= (int64_t)base[l + 1] || no >= (uint32_t)maxn)
                    throw std::ios_base::failure(std::string(filename) + ": neighbour outside its layer");
                L.nb_slot[(size_t)s * L.deg + j] = (uint32_t)(gi - base[l]);
                L.nb_order[(size_t)s * L.deg + j] = no;
                L.nb_dist[(size_t)s * L.deg + j] = 0.0f;   // not stored in the file (src/hnsw.cpp:1299)
            }
        }
    }
    // A node's own node_order is not in the file: layer 0 slot s is vector s; upper nodes
    // inherit it through their lower links.
    for (uint32_t s = 0; s < layers_[0].count; s++) layers_[0].order[s] = s;
    for (size_t l = 1; l < nl; l++)
        for (uint32_t s = 0; s < layers_[l].count; s++) layers_[l].order[s] = layers_[l - 1].order[layers_[l].down[s]];
    // the stored counter also counts queries run before the save (src/hnsw.cpp:380); the
    // number of vectors is the size of layer 0
    actual_node_count_ = (int)layers_[0].count - 1;
    (void)act;
    version_++;
}

}  // namespace hnsw_amd
