// libpsn host side: the C ABI declared in include/psn.h.  Owns device memory, validates
// what crosses the boundary, records the sweep sequence into a hipGraph and replays it
// until the device-resident control block reports completion.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <atomic>
#include <cmath>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/psn.h"
#include "psn_device.h"
#include <chrono>
#include <cstdio>
#include "psn_kernels.hip"  // one translation unit: kernels are launched from this file

using namespace psn;

struct psn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    psn_config cfg{};
    std::string err;
    // sizes
    int n = 0;
    long long m2 = 0;
    long long high_items = 0;
    bool uploaded = false, solved = false;
    DevGraph g{};
    // owned device buffers
    std::vector<void *> bufs;
    i64 *d_supply = nullptr;
    int *d_cost = nullptr, *d_ocap = nullptr, *d_head = nullptr, *d_rev = nullptr;  // non-const aliases of g.cost / g.ocap / ...
    i64 *d_scalar = nullptr;                   // [0] total cost, [1] max violation, [2..] k_cost_stats output
    unsigned long long *d_tally_out = nullptr;
    int tally_slots = 0;
    // class x machine admission mask (label selectors), applied by the cost kernel when set
    unsigned *d_allow = nullptr;
    int allow_words = 0, allow_classes = 0;  // shape the buffer was allocated for
    bool allow_on = false;
    // placements scratch
    int *d_place = nullptr, *d_slot_machine = nullptr;
    long long place_count = 0, slot_stride = 0, slot_classes = 0;
    // graph replay
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    int rounds = 8;
    int launches_per_replay = 0;
    Ctrl *h_ctrl = nullptr;  // pinned, two slots: replays are kept two deep in the stream and read back alternately
    hipEvent_t ev_rep[2] = {nullptr, nullptr};
    // per-kernel-class timing (PSN_CFG_KTIME, eager launches only): event pairs per launch
    std::vector<hipEvent_t> kt_ev;
    std::vector<int> kt_cls;
    size_t kt_used = 0;
    double kt_ms[PSN_KT_CLASSES] = {0};
    uint64_t kt_calls[PSN_KT_CLASSES] = {0};
    // host copy of the row offsets of the uploaded network: layouts handed to the cost model and
    // to placement extraction are checked against it before any kernel indexes with them
    std::vector<unsigned> h_row;
    // grow-only device staging for the per-round update calls (kept across uploads), and a
    // pinned word block for small read-backs
    char *d_stage = nullptr;
    size_t stage_bytes = 0;
    unsigned *h_flag = nullptr;  // pinned, kFlagWords words
    // identity for caches that outlive a call: never reused, bumped on every upload
    unsigned long long serial = 0, upload_gen = 0;
    // vertex cover behind the cost scale: flags per vertex, size found, size the current scale allows for
    unsigned char *d_in_cover = nullptr;
    unsigned char *d_cls = nullptr;    // vertex class bytes (the device copy the kernels read; k_lean_flag sets a bit in them)
    unsigned *d_lean_stop = nullptr;
    int *d_hub_of = nullptr;           // hub slots from the host, pull marks from k_pull_mark
    int2 *d_pull_list = nullptr;
    unsigned *d_pull_cnt = nullptr;
    long long cover_size = 0, cover_cap = 0;
    // ... and the part of the cover that is not in the independent subset (see k_cover_best), with its head-room
    unsigned long long *d_cover_best = nullptr;
    unsigned char *d_not_indep = nullptr;
    long long noni_size = 0, noni_cap = 0;
    struct GroupCache *gcache = nullptr;  // lockstep group led by this context, kept between batch calls
    // the slot-block layout the cost kernel last verified on the device (k_check_slot_layout re-reads
    // the pairing of every slot arc: 150 MB at 10 000 machines): valid for this upload until arcs are patched
    unsigned long long layout_ok_gen = ~0ull;
    int layout_ok_E = 0, layout_ok_N = 0, layout_ok_K = 0;
    std::vector<int64_t> layout_ok_base;
};

namespace {

int fail(psn_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
    if (c) {
        c->err = what ? what : "";
        if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
    }
    return code;
}
#define HIPCHK(c, call)                                                   \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return fail((c), PSN_EDEVICE, #call, e_);   \
    } while (0)

// Every device or pinned allocation the library makes goes through counted_malloc /
// counted_host_malloc, so that a caller (or a test) can see that a warm scheduling round
// allocates nothing: psn_debug_alloc_count().
std::atomic<unsigned long long> g_alloc_calls{0};
std::atomic<unsigned long long> g_ctx_serial{0};
hipError_t counted_malloc(void **p, size_t bytes) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    return hipMalloc(p, bytes);
}
hipError_t counted_host_malloc(void **p, size_t bytes) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}
constexpr int kFlagWords = 16;

// ---- scratch of the entry point that allocates nothing (psn_get_placements_device_noalloc)
// One array per GPU the library runs on, held until the process ends.  Host code never uses the
// array's name as an address: the per-device address is looked up once per device when the first
// context on that device is created, and that pointer is what kernels and copies are given.
constexpr size_t kNoAllocInts = (size_t)1 << 20;  // classes x tasks the entry point can serve: 4 MiB per GPU
constexpr int kMaxDevices = 64;
static __device__ __align__(256) int scratch[kNoAllocInts];
int *g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;
// Device `device` must be current.
hipError_t resolve_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return hipSuccess;
    int *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

template <typename T>
int dalloc(psn_ctx *c, T **p, size_t count) {
    void *q = nullptr;
    hipError_t e = counted_malloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return fail(c, PSN_ENOMEM, "hipMalloc", e);
    c->bufs.push_back(q);
    *p = static_cast<T *>(q);
    return PSN_OK;
}
// Device staging of at least `bytes` for one update call: grows by half when it has to, else reused.
int stage(psn_ctx *c, size_t bytes, char **out) {
    if (bytes > c->stage_bytes) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(c, PSN_EDEVICE, "hipStreamSynchronize", e);
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr;
        c->stage_bytes = 0;
        const size_t want = std::max<size_t>((size_t)1 << 20, bytes + bytes / 2);
        void *q = nullptr;
        e = counted_malloc(&q, want);
        if (e != hipSuccess) return fail(c, PSN_ENOMEM, "hipMalloc (staging)", e);
        c->d_stage = static_cast<char *>(q);
        c->stage_bytes = want;
    }
    *out = c->d_stage;
    return PSN_OK;
}
void drop_group_cache(psn_ctx *c);
void free_graph_buffers(psn_ctx *c) {
    drop_group_cache(c);
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    for (void *p : c->bufs) (void)hipFree(p);
    c->bufs.clear();
    c->h_row.clear();
    ++c->upload_gen;
    c->uploaded = c->solved = false;
    c->d_place = c->d_slot_machine = nullptr;
    c->d_in_cover = nullptr; c->d_cls = nullptr; c->d_lean_stop = nullptr; c->d_hub_of = nullptr; c->d_pull_list = nullptr; c->d_pull_cnt = nullptr; c->cover_size = c->cover_cap = 0; c->d_cover_best = nullptr; c->d_not_indep = nullptr; c->noni_size = c->noni_cap = 0;
    c->d_allow = nullptr; c->allow_words = c->allow_classes = 0; c->allow_on = false;
    c->place_count = c->slot_stride = c->slot_classes = 0;
}

// The fixed launch sequence of one replay; `count_only` just counts the launches.
int enqueue_replay(psn_ctx *c, hipStream_t s) {
    const DevGraph &g = c->g;
    int launches = 0;
    const int gv = std::max(1, std::min(kGrid, (g.n + kBlock * vpt_of(g.n) - 1) / (kBlock * vpt_of(g.n))));  // vertex sweeps
    const int gv1 = std::max(1, std::min(kGrid, (g.n + kBlock - 1) / kBlock));
    // arc sweeps: enough workgroups for the network, no more (small networks then leave room
    // for the kernels of other contexts running beside them)
    int gs = (int)std::max<long long>(32, std::min<long long>(kGrid, (c->m2 + kBlock * 16 - 1) / (kBlock * 16)));
    if (const char *ev = std::getenv("PSN_GRID")) gs = std::max(1, std::min(kGrid, atoi(ev)));
    const bool timed = (c->cfg.flags & PSN_CFG_KTIME) && !c->gexec;
    auto mark = [&](int cls, bool begin) {
        if (!timed) return;
        if (c->kt_used == c->kt_ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            c->kt_ev.push_back(e);
            c->kt_cls.push_back(0);
        }
        c->kt_cls[c->kt_used] = begin ? cls : -1;
        (void)hipEventRecord(c->kt_ev[c->kt_used++], s);
    };
#define L(cls, kern, grid, ...)                                           \
    do {                                                                  \
        mark(cls, true);                                                  \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, s, __VA_ARGS__); \
        mark(cls, false);                                                 \
        ++launches;                                                       \
    } while (0)
    hipLaunchKernelGGL(k_begin, dim3(1), dim3(1), 0, s, g, c->rounds); ++launches;
    L(PSN_KT_OTHER, k_flip, gv1, g, 0);
    L(PSN_KT_SATURATE, k_saturate, gs, g);
    L(PSN_KT_OTHER, k_gu_init, gv, g);
    for (int k = 0; k < kGuIters; ++k) L(PSN_KT_GU_EXPAND, k_gu_expand, gs, g, k);
    L(PSN_KT_OTHER, k_gu_apply, gv1, g, kGuIters);
    L(PSN_KT_RELABEL, k_relabel_compact, gv, g, 0, 0);
    for (int k = 0; k < c->rounds; ++k) {
        L(PSN_KT_PUSH, k_push, gs, g, k);
        L(PSN_KT_RELABEL, k_relabel_compact, gv, g, k, 1);
    }
#undef L
    c->launches_per_replay = launches;
    return PSN_OK;
}

int build_graph(psn_ctx *c) {
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    if (c->cfg.flags & PSN_CFG_EAGER) return PSN_OK;
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    enqueue_replay(c, c->stream);
    const hipError_t e_launch = hipGetLastError();
    const hipError_t e_end = hipStreamEndCapture(c->stream, &c->graph);  // always, so that the stream leaves capture mode
    HIPCHK(c, e_launch);
    HIPCHK(c, e_end);
    HIPCHK(c, hipGraphInstantiate(&c->gexec, c->graph, nullptr, nullptr, 0));
    return PSN_OK;
}

// Upload checks on the device: the host has verified the row offsets, everything read through an
// index taken from the input is range-checked before it is used.
__global__ void k_fill_tail(const unsigned *row, int n, unsigned m2, int *tail) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < m2; a += gridDim.x * kBlock) {
        int lo = 0, hi = n;  // row[lo] <= a < row[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row[mid] <= a) lo = mid; else hi = mid; }
        tail[a] = lo;
    }
}
__device__ __forceinline__ bool arc_is_sound(int n, unsigned m2, const int *head, const int *cap, const int *cost, const int *rev,
                                              const int *tail, unsigned a) {
    const int r = rev[a];
    if (head[a] < 0 || head[a] >= n) return false;
    if (r < 0 || (unsigned)r >= m2 || rev[r] != (int)a) return false;
    if ((unsigned)r == a && (head[a] != tail[a] || cap[a] != 0 || cost[a] != 0)) return false;
    if (head[r] != tail[a] || cost[r] != -cost[a] || cap[a] < 0) return false;
    return (long long)cap[a] + cap[r] <= INT32_MAX;
}
// out[0]: lowest index of an unsound arc (starts at ~0), out[1]: largest |cost|
__global__ void k_check_arcs(int n, unsigned m2, const int *head, const int *cap, const int *cost, const int *rev, const int *tail,
                             unsigned long long *out) {
    unsigned long long bad = ~0ull, mx = 0;
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < m2; a += gridDim.x * kBlock) {
        if (!arc_is_sound(n, m2, head, cap, cost, rev, tail, a) && a < bad) bad = a;
        const long long k = cost[a];
        const unsigned long long ak = (unsigned long long)(k < 0 ? -k : k);
        mx = ak > mx ? ak : mx;
    }
    for (int d = 32; d; d >>= 1) {
        const unsigned long long b2 = __shfl_xor(bad, d), m2x = __shfl_xor(mx, d);
        bad = b2 < bad ? b2 : bad; mx = m2x > mx ? m2x : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad != ~0ull) __hip_atomic_fetch_min(&out[0], bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mx) __hip_atomic_fetch_max(&out[1], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Vertex cover for the cost scale.  A flow that is 1-optimal for costs multiplied by S is optimal
// as soon as S exceeds the length of every simple residual cycle: such a cycle of (unscaled) cost
// <= -1 would have scaled cost <= -S, while its reduced costs sum to >= -length.  A simple cycle
// passes every vertex once and each of its arcs has an endpoint in any vertex cover, so its length
// is at most twice the size of the cover.  The cover taken here is the complement of the
// independent set { v : every neighbour has a longer row than v (ties: a larger id) }; capacities
// play no part, so the bound holds whatever the cost model or an update does to them.  In a
// scheduling network the tasks and the machines stay out of the cover: S drops from n + 1 to
// about 2 * (classes + jobs), which is one refine phase less.
__global__ void k_cover_mark(const unsigned *row, int n, unsigned m2, const int *head, const int *tail, unsigned char *in_cover) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < m2; a += gridDim.x * kBlock) {
        const int t = tail[a], h = head[a];
        if (h == t || (unsigned)h >= (unsigned)n) continue;  // placeholder self-loop / rejected elsewhere
        const unsigned dt = row[t + 1] - row[t], dh = row[h + 1] - row[h];
        if (!(dh > dt || (dh == dt && h > t))) in_cover[t] = 1;  // a neighbour that is not "larger": t cannot stay out
    }
}
// A sharper bound from the same cover C.  Between two cover vertices that follow each other on a
// simple cycle lies at most one other vertex (the vertices outside C are pairwise non-adjacent), so
// a cycle with k cover vertices has at most 2k arcs.  Let I be a subset of C no two members of which
// are adjacent or share a neighbour outside C: two members of I cannot follow each other on a
// cycle, so at most half of a cycle's cover vertices are in I, k <= 2 |C \ I|, and no simple cycle
// has more than 4 |C \ I| arcs (two arcs between one pair of vertices are a cycle of 2).  I is taken
// as the cover vertices v such that, with key(u) = (row length, id): every neighbour in C has a larger
// key than v, and every neighbour w outside C has v as the smallest-keyed of w's cover neighbours
// (so w borders no second member of I).  In a scheduling network I holds the unscheduled
// aggregators (each task has one, the sink and the classes have larger rows), C \ I the classes
// and the sink: 4 * 65 arcs instead of 2 * 2 065 at 10 000 machines - four bits of cost scale,
// one refine phase.
__device__ __forceinline__ unsigned long long cover_key(const unsigned *row, int v) {
    return ((unsigned long long)(row[v + 1] - row[v]) << 32) | (unsigned)v;
}
__global__ void k_cover_best(const unsigned *row, int n, unsigned m2, const int *head, const int *tail, const unsigned char *in_cover,
                             unsigned long long *best) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < m2; a += gridDim.x * kBlock) {
        const int t = tail[a], h = head[a];
        if (h == t || (unsigned)h >= (unsigned)n) continue;
        if (!in_cover[t] && in_cover[h]) {
            const unsigned long long k = cover_key(row, h);
            if (k < best[t]) __hip_atomic_fetch_min(&best[t], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // best only falls: a stale read just costs an atomic
        }
    }
}
__global__ void k_cover_indep(const unsigned *row, int n, unsigned m2, const int *head, const int *tail, const unsigned char *in_cover,
                              const unsigned long long *best, unsigned char *not_indep) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < m2; a += gridDim.x * kBlock) {
        const int t = tail[a], h = head[a];
        if (h == t || (unsigned)h >= (unsigned)n || !in_cover[t]) continue;
        const unsigned long long kt = cover_key(row, t);
        const bool ok = in_cover[h] ? kt < cover_key(row, h) : best[h] == kt;
        if (!ok) not_indep[t] = 1;
    }
}
__global__ void k_cover_count_noni(const unsigned char *in_cover, const unsigned char *not_indep, int n, unsigned *out) {
    unsigned c = 0;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) c += in_cover[v] && not_indep[v];
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0 && c) __hip_atomic_fetch_add(out, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Rows made of aligned groups of 8 parallel arcs (the 8 slot arcs between a task class and a
// machine, seen from either end) let the expansion sweep fetch one head per lane instead of eight
// and take a two-candidate path.  lean_stop[v] = first arc index past the leading such groups of
// v's row; a vertex that is one wave item gets the kClsLean bit when every step of its row
// qualifies (the 8-arc parts of its steps all lie inside that prefix).
__global__ void k_lean_init(const unsigned *row, int n, unsigned *lean_stop) {
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) lean_stop[v] = row[v] + ((row[v + 1] - row[v]) & ~7u);
}
__global__ void k_lean_scan(const unsigned *row, unsigned m2, const int *head, const int *tail, unsigned *lean_stop) {
    // a wave's 64 consecutive arcs mostly belong to one vertex: its lanes agree on the first group that
    // is not uniform and one of them reports it (thousands of same-address atomics per large row otherwise)
    const unsigned m2_round = (m2 + 63u) & ~63u;
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < m2_round; a += gridDim.x * kBlock) {
        const bool in = a < m2;
        const int v = tail[in ? a : m2 - 1];
        unsigned bad = 0xFFFFFFFFu;
        if (in && !((a - row[v]) & 7u) && a + 8 <= row[v + 1] && lean_stop[v] > a) {  // first arc of a complete group inside the prefix so far
            bool same = true;
            for (int j = 1; j < 8; ++j) same &= head[a + j] == head[a];
            if (!same) bad = a;
        }
        const int v0 = __builtin_amdgcn_readfirstlane(v);
        if (__all(v == v0)) {
            unsigned m = bad;
            for (int d = 32; d; d >>= 1) { const unsigned y = __shfl_xor(m, d); m = y < m ? y : m; }
            if ((threadIdx.x & 63) == 0 && m != 0xFFFFFFFFu) atomicMin(&lean_stop[v0], m);
        } else if (bad != 0xFFFFFFFFu) {
            atomicMin(&lean_stop[v], bad);
        }
    }
}
__global__ void k_lean_flag(const unsigned *row, int n, const unsigned *lean_stop, unsigned char *cls) {
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) {
        const unsigned char c = cls[v] & kClsMask;
        const unsigned deg = row[v + 1] - row[v];
        bool lean = false;
        if (c == kClsMed && deg > 0) {
            const unsigned last_step = (deg - 1) / kStep * kStep;  // offset of the row's last step
            lean = lean_stop[v] - row[v] >= min(deg, last_step + 512u);
        }
        cls[v] = c | (lean ? kClsLean : 0);
    }
}
// Pull vertices (DevGraph::pull_list): degree <= kLowMax and every neighbour of a larger degree.
__global__ void k_pull_mark(const unsigned *row, int n, const int *head, const unsigned char *cls, int *hub_of, int2 *pull_list,
                            unsigned *pull_cnt) {
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) {
        if ((cls[v] & kClsMask) != kClsLow) continue;
        const unsigned r0 = row[v], r1 = row[v + 1];
        bool ok = r1 > r0;
        for (unsigned a = r0; a < r1; ++a) {
            const int h = head[a];
            if ((unsigned)h >= (unsigned)n || (cls[h] & kClsMask) == kClsLow) ok = false;
        }
        hub_of[v] = ok ? kHubPull : -1;
        if (ok) pull_list[atomicAdd(pull_cnt, 1u)] = make_int2(v | (int)((r1 - r0) << 28), (int)r0);
    }
}
__global__ void k_cover_count(const unsigned char *in_cover, int n, unsigned *out) {
    unsigned c = 0;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) c += in_cover[v];
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0 && c) __hip_atomic_fetch_add(out, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// After a patch: do the tables that follow from the heads (vertex cover, its independent subset, the
// prefixes of parallel-arc groups, the pull vertices) still stand, so that they need not be made again
// from all arcs?  Arcs that went away only loosen what the tables claim; an arc (t, h) that is there
// now must have an endpoint in the cover, must not join two members of the independent subset I nor
// give a vertex outside the cover a second neighbour in I (h's row is looked through when it is short;
// a long one counts as broken), must lie behind its row's prefix of parallel groups, and must not
// give a pull vertex a low-degree neighbour.  broken[0] counts the arcs that fail.
__device__ __forceinline__ bool independent_subset_stands(const unsigned *row, int n, const int *head, const unsigned char *in_cover,
                                                          const unsigned char *not_indep, int t, int h) {
    if (!(in_cover[t] && !not_indep[t])) return true;  // t is not in I: nothing to ask from its side
    if (in_cover[h]) return not_indep[h] != 0;
    const unsigned r0 = row[h], r1 = row[h + 1];
    if (r1 - r0 > 16u) return false;
    for (unsigned b = r0; b < r1; ++b) {
        const int x = head[b];
        if (x != t && x != h && (unsigned)x < (unsigned)n && in_cover[x] && !not_indep[x]) return false;
    }
    return true;
}
__global__ void k_patch_keeps_tables(const unsigned *row, int n, long long count, const i64 *idx, const int *head, const int *tail,
                                     const unsigned char *in_cover, const unsigned char *not_indep, const unsigned *lean_stop,
                                     const int *hub_of, const unsigned char *cls, unsigned *broken) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const unsigned a = (unsigned)idx[i];
        const int t = tail[a], h = head[a];
        if (h == t || (unsigned)h >= (unsigned)n) continue;  // placeholder self-loop (checked elsewhere)
        bool ok = in_cover[t] || in_cover[h];
        ok = ok && independent_subset_stands(row, n, head, in_cover, not_indep, t, h) && independent_subset_stands(row, n, head, in_cover, not_indep, h, t);
        if (a < lean_stop[t]) ok = false;
        if (hub_of[t] == kHubPull && (cls[h] & kClsMask) == kClsLow) ok = false;
        if (hub_of[h] == kHubPull && (cls[t] & kClsMask) == kClsLow) ok = false;
        if (!ok) atomicAdd(broken, 1u);
    }
}

// Cost statistics of the arcs that can carry flow: out[0] = max |cost|, out[1] = number with a
// negative cost, out[2 + b] = number with 2^(b-1) < cost <= 2^b (b = 0 holds costs 0 and 1).
__device__ __forceinline__ void cost_stats_body(const DevGraph &g, i64 *out) {
    __shared__ unsigned s_hist[33];
    if (threadIdx.x < 33) s_hist[threadIdx.x] = 0;
    __syncthreads();
    i64 mx = 0;
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        const bool fwd = g.ocap[a] > 0;
        if (fwd || g.ocap[g.rev[a]] > 0) { i64 c = g.cost[a]; c = c < 0 ? -c : c; mx = c > mx ? c : mx; }
        if (fwd) {
            const int c = g.cost[a];
            const int b = c < 0 ? 32 : (c <= 1 ? 0 : 32 - __clz(c - 1));
            atomicAdd(&s_hist[b], 1u);
        }
    }
    for (int d = 32; d; d >>= 1) { i64 y = __shfl_xor(mx, d); mx = y > mx ? y : mx; }
    if ((threadIdx.x & 63) == 0 && mx > 0) __hip_atomic_fetch_max(out, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x < 33 && s_hist[threadIdx.x]) {
        i64 *dst = threadIdx.x == 32 ? out + 1 : out + 2 + threadIdx.x;
        __hip_atomic_fetch_add(dst, (i64)s_hist[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Largest amount by which a residual arc's reduced cost is below zero (warm start).
__device__ __forceinline__ void max_violation_body(const DevGraph &g, i64 *out) {
    i64 mx = 0;
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        if (g.cap[a] > 0) {
            i64 rc = (i64)g.cost[a] * g.scale + g.price[g.tail[a]] - g.price[g.head[a]];
            if (-rc > mx) mx = -rc;
        }
    }
    for (int d = 32; d; d >>= 1) { i64 y = __shfl_xor(mx, d); mx = y > mx ? y : mx; }
    if ((threadIdx.x & 63) == 0 && mx > 0) __hip_atomic_fetch_max(out, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of the supplies of the resident network (psn_update_supplies may have unbalanced them) and
// the number of vertices that hold supply.
__device__ __forceinline__ void sum_supply_body(const i64 *supply, int n, i64 *out, i64 *n_sources, i64 *units) {
    i64 sm = 0, un = 0;
    unsigned src = 0;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) {
        sm += supply[v];
        src += supply[v] > 0;
        un += supply[v] > 0 ? supply[v] : 0;
    }
    sm = (i64)wave_sum_u64((unsigned long long)sm);
    un = (i64)wave_sum_u64((unsigned long long)un);
    src = wave_sum_u32(src);
    if ((threadIdx.x & 63) == 0 && sm) __hip_atomic_fetch_add(out, sm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((threadIdx.x & 63) == 0 && src) {
        __hip_atomic_fetch_add(n_sources, (i64)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(units, un, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr int kScalarSlots = 48;
constexpr int kSupplySumSlot = 38;  // d_scalar: [0] total cost, [1] max violation, [2..35] cost statistics, [36] vertices with supply, [38] supply sum
constexpr int kSourcesSlot = 36;
constexpr int kSupplyUnitsSlot = 40;  // sum of the positive supplies
__global__ void k_cost_stats(DevGraph g, i64 *out) { cost_stats_body(g, out); }
__global__ void k_max_violation(DevGraph g, i64 *out) { max_violation_body(g, out); }
__global__ void k_sum_supply(const i64 *supply, int n, i64 *out, i64 *n_sources, i64 *units) { sum_supply_body(supply, n, out, n_sources, units); }
// for the members of a lockstep group (blockIdx.y picks the member): the statistics the first
// epsilon comes from - of the costs for a cold member, of the kept prices' violations for a warm one
__global__ void k_start_stats_group(const DevGraph *__restrict__ gs, const GroupAux *__restrict__ aux) {
    const GroupAux &x = aux[blockIdx.y];
    if (x.warm) max_violation_body(gs[blockIdx.y], x.scalar + 1);
    else cost_stats_body(gs[blockIdx.y], x.scalar + 2);
}
__global__ void k_sum_supply_group(const DevGraph *__restrict__ gs, const GroupAux *__restrict__ aux) {
    const GroupAux &x = aux[blockIdx.y];
    sum_supply_body(x.supply, gs[blockIdx.y].n, x.scalar + kSupplySumSlot, x.scalar + kSourcesSlot, x.scalar + kSupplyUnitsSlot);
}

// Marks and counts the vertex cover of the resident network on the context's stream; the count is
// on its way to h_flag[2] when this returns (d_scalar slot kCoverSlot is the device counter).
constexpr int kCoverSlot = 37;
int launch_cover(psn_ctx *c) {
    const DevGraph &g = c->g;
    unsigned *d_cnt = reinterpret_cast<unsigned *>(c->d_scalar + kCoverSlot);
    HIPCHK(c, hipMemsetAsync(c->d_in_cover, 0, (size_t)g.n, c->stream));
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, sizeof(unsigned), c->stream));
    if (g.m2) hipLaunchKernelGGL(k_cover_mark, dim3(kGrid), dim3(kBlock), 0, c->stream, g.row, g.n, g.m2, g.head, g.tail, c->d_in_cover);
    hipLaunchKernelGGL(k_cover_count, dim3(std::max(1, std::min(256, (g.n + kBlock - 1) / kBlock))), dim3(kBlock), 0, c->stream, c->d_in_cover, g.n, d_cnt);
    HIPCHK(c, hipMemcpyAsync(c->h_flag + 2, d_cnt, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    // the cover vertices outside the independent subset (count on its way to h_flag[3])
    unsigned *d_cnt2 = d_cnt + 1;
    HIPCHK(c, hipMemsetAsync(c->d_cover_best, 0xFF, sizeof(unsigned long long) * (size_t)g.n, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_not_indep, 0, (size_t)g.n, c->stream));
    HIPCHK(c, hipMemsetAsync(d_cnt2, 0, sizeof(unsigned), c->stream));
    if (g.m2) {
        hipLaunchKernelGGL(k_cover_best, dim3(kGrid), dim3(kBlock), 0, c->stream, g.row, g.n, g.m2, g.head, g.tail, c->d_in_cover, c->d_cover_best);
        hipLaunchKernelGGL(k_cover_indep, dim3(kGrid), dim3(kBlock), 0, c->stream, g.row, g.n, g.m2, g.head, g.tail, c->d_in_cover, c->d_cover_best, c->d_not_indep);
    }
    hipLaunchKernelGGL(k_cover_count_noni, dim3(std::max(1, std::min(256, (g.n + kBlock - 1) / kBlock))), dim3(kBlock), 0, c->stream, c->d_in_cover, c->d_not_indep, g.n, d_cnt2);
    HIPCHK(c, hipMemcpyAsync(c->h_flag + 3, d_cnt2, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    // the other table that follows from the heads alone: where rows stop being groups of parallel arcs
    const int gvn = std::max(1, std::min(kGrid, (g.n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_lean_init, dim3(gvn), dim3(kBlock), 0, c->stream, g.row, g.n, c->d_lean_stop);
    if (g.m2) hipLaunchKernelGGL(k_lean_scan, dim3(kGrid), dim3(kBlock), 0, c->stream, g.row, g.m2, g.head, g.tail, c->d_lean_stop);
    hipLaunchKernelGGL(k_lean_flag, dim3(gvn), dim3(kBlock), 0, c->stream, g.row, g.n, c->d_lean_stop, c->d_cls);
    HIPCHK(c, hipMemsetAsync(c->d_pull_cnt, 0, sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_pull_mark, dim3(gvn), dim3(kBlock), 0, c->stream, g.row, g.n, g.head, c->d_cls, c->d_hub_of, c->d_pull_list, c->d_pull_cnt);
    return PSN_OK;
}
// The cost scale that goes with a cover of `cover` vertices, with room for the cover to grow a
// little (tasks and jobs that arrive later) before the scale has to change.
void scale_for_cover(psn_ctx *c, long long cover, long long noni) {
    c->cover_size = cover;
    c->cover_cap = cover + std::max<long long>(64, cover / 8);
    c->noni_size = noni;
    c->noni_cap = noni + std::max<long long>(16, noni / 8);
    i64 len = std::min<i64>((i64)c->g.n, 2 * (i64)c->cover_cap);  // no simple cycle is longer
    if (!std::getenv("PSN_SCALE_COVER")) len = std::min<i64>(len, 4 * (i64)c->noni_cap);  // (development: the plain cover bound)
    len = std::max<i64>(len, 2);
    if (std::getenv("PSN_SCALE_N")) len = c->g.n;                 // diagnostics: the textbook scale n + 1
    c->g.scale = len + 1;
}

int ceil_log2(i64 x) {
    int s = 0;
    while (((i64)1 << s) < x && s < 62) ++s;
    return s;
}

}  // namespace

extern "C" {

int psn_version(void) { return 310; }
const char *psn_kernel_list(void) {
    return "k_begin,k_flip,k_saturate,k_gu_init,k_gu_expand,k_gu_apply,k_relabel_compact,k_push";
}

const char *psn_strerror(int code) {
    switch (code) {
        case PSN_OK: return "ok";
        case PSN_EINVAL: return "invalid argument or malformed graph";
        case PSN_ENOMEM: return "out of memory";
        case PSN_EDEVICE: return "HIP runtime error";
        case PSN_EINFEASIBLE: return "supplies cannot be routed (infeasible)";
        case PSN_EOVERFLOW: return "scaled costs do not fit 64-bit arithmetic";
        case PSN_ESTATE: return "call out of order";
        case PSN_ELIMIT: return "iteration guard tripped";
        default: return "unknown error";
    }
}

const char *psn_last_error(const psn_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int psn_create(const psn_config *cfg, psn_ctx **out) {
    if (!out) return PSN_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PSN_EDEVICE;  // no GPU: fail loudly
    psn_ctx *c = new (std::nothrow) psn_ctx();
    if (!c) return PSN_ENOMEM;
    if (cfg) c->cfg = *cfg;
    if (c->cfg.alpha_log2 <= 0 || c->cfg.alpha_log2 > 8) c->cfg.alpha_log2 = kDefaultAlphaLog2;
    c->rounds = c->cfg.rounds_per_batch > 0 ? c->cfg.rounds_per_batch : kDefaultRounds;
    if (const char *ev = std::getenv("PSN_ROUNDS")) c->rounds = std::max(1, std::min(64, atoi(ev)));  // tuning aid
    if (c->rounds & 1) ++c->rounds;  // list parity must be the same at every replay start
    if (c->rounds > 62) c->rounds = 62;
    c->device = c->cfg.device;
    if (c->device < 0 || c->device >= ndev) { delete c; return PSN_EINVAL; }
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = counted_host_malloc((void **)&c->h_ctrl, 2 * sizeof(Ctrl));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_rep[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_rep[1], hipEventDisableTiming);
    if (e == hipSuccess) e = counted_host_malloc((void **)&c->h_flag, kFlagWords * sizeof(unsigned));
    if (e == hipSuccess) e = resolve_scratch(c->device);
    if (e != hipSuccess) { psn_destroy(c); return PSN_EDEVICE; }
    c->serial = g_ctx_serial.fetch_add(1) + 1;
    *out = c;
    return PSN_OK;
}

void psn_destroy(psn_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : c->kt_ev) (void)hipEventDestroy(e);
    free_graph_buffers(c);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->h_ctrl) (void)hipHostFree(c->h_ctrl);
    for (int i = 0; i < 2; ++i) if (c->ev_rep[i]) (void)hipEventDestroy(c->ev_rep[i]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Everything about a network that follows from its row offsets: vertex classes, hubs and their
// segments, every device buffer, and the small per-vertex tables uploaded.  The arc arrays
// (head, cost, rev, ocap, cap, supply) are left for the caller to fill on the stream.
static int setup_rows(psn_ctx *c, int32_t n, int64_t m2, const int64_t *row_off) {
    if (n <= 0 || n >= kMaxVertices || m2 < 0 || m2 >= (int64_t)0x7FFFFFF0 || !row_off) return fail(c, PSN_EINVAL, "null or out-of-range argument");
    if (row_off[0] != 0 || row_off[n] != m2) return fail(c, PSN_EINVAL, "row_off[0] != 0 or row_off[n] != m2");
    for (int32_t v = 0; v < n; ++v)
        if (row_off[v + 1] < row_off[v]) return fail(c, PSN_EINVAL, "row_off not monotone");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_graph_buffers(c);
    c->n = n;
    c->m2 = m2;
    // ---- classify vertices by degree
    std::vector<unsigned> &row = c->h_row;
    row.assign((size_t)n + 1, 0u);
    std::vector<unsigned char> cls((size_t)n);
    long long high_items = 0;
    for (int32_t v = 0; v <= n; ++v) row[(size_t)v] = (unsigned)row_off[v];
    for (int32_t v = 0; v < n; ++v) {
        const unsigned deg = row[(size_t)v + 1] - row[(size_t)v];
        cls[(size_t)v] = deg <= kLowMax ? kClsLow : (deg <= kMedMax ? kClsMed : kClsHigh);
        if (deg > kMedMax) high_items += (deg + kWSeg - 1) / kWSeg;
    }
    c->high_items = high_items;
    // hubs: the kMaxHubs highest-degree vertices among those split into several items
    std::vector<int> hub_of((size_t)n, -1), hub_vertex((size_t)kMaxHubs, 0);
    int nhubs = 0;
    {
        std::vector<std::pair<unsigned, int>> big;
        for (int32_t v = 0; v < n; ++v)
            if (cls[(size_t)v] == kClsHigh) big.emplace_back(row[(size_t)v + 1] - row[(size_t)v], v);
        std::sort(big.begin(), big.end(), [](const std::pair<unsigned, int> &x, const std::pair<unsigned, int> &y) {
            return x.first != y.first ? x.first > y.first : x.second < y.second;
        });
        for (size_t i = 0; i < big.size() && nhubs < kMaxHubs; ++i) {
            hub_of[(size_t)big[i].second] = nhubs;
            hub_vertex[(size_t)nhubs++] = big[i].second;
        }
    }
    std::vector<unsigned> hub_row0((size_t)kMaxHubs, 0u);
    std::vector<int> hub_seg0((size_t)kMaxHubs, 0);
    int n_hub_segs = 0;
    for (int i = 0; i < nhubs; ++i) {
        const int hv = hub_vertex[(size_t)i];
        hub_row0[(size_t)i] = row[(size_t)hv];
        hub_seg0[(size_t)i] = n_hub_segs;
        n_hub_segs += (int)((row[(size_t)hv + 1] - row[(size_t)hv] + kWSeg - 1) / kWSeg);
    }
    if (n_hub_segs >= (1 << 23) - 1) return fail(c, PSN_EINVAL, "too many hub segments");  // an item's flags hold 1 + the segment index in 23 bits
    std::vector<int> seg1_of((size_t)n, 0);
    for (int i = 0; i < nhubs; ++i) seg1_of[(size_t)hub_vertex[(size_t)i]] = hub_seg0[(size_t)i] + 1;
    // ---- allocate
    DevGraph &g = c->g;
    g = DevGraph{};
    g.n = n;
    g.m2 = (unsigned)m2;
    g.scale = (i64)n + 1;
    unsigned *d_row, *d_hub_row0; int *d_head, *d_rev, *d_tail, *d_hub_of, *d_hub_vertex, *d_hub_seg0, *d_seg1_of; unsigned char *d_cls;
    int rc;
#define A(p, cnt) if ((rc = dalloc(c, &(p), (size_t)(cnt))) != PSN_OK) { free_graph_buffers(c); return rc; }
    A(d_row, n + 1) A(d_head, m2 + kArcPad) A(c->d_cost, m2 + kArcPad) A(d_rev, m2 + kArcPad) A(d_tail, m2) A(c->d_ocap, m2 + kArcPad) A(d_cls, n) A(d_hub_of, n) A(d_hub_vertex, kMaxHubs)
    A(d_hub_row0, kMaxHubs) A(d_hub_seg0, kMaxHubs) A(d_seg1_of, n) A(g.seg_scan, n_hub_segs) A(g.seg_add[0], n_hub_segs) A(g.seg_add[1], n_hub_segs)
    A(g.hub_purse, kMaxHubs * kPurseStride) A(g.hub_acc_add, kHubShards * kMaxHubs) A(g.hub_acc_ext, kHubShards * kMaxHubs)
    A(g.cap, m2 + kArcPad) A(g.rcap, m2 + kArcPad) A(g.price, n) A(g.excess, n) A(g.cand, n) A(g.dist, n)
    A(c->d_supply, n) A(c->d_scalar, kScalarSlots) A(c->d_in_cover, n) A(c->d_cover_best, n) A(c->d_not_indep, n) A(c->d_lean_stop, n) A(c->d_pull_list, n) A(c->d_pull_cnt, 1)
    for (int i = 0; i < 2; ++i) { A(g.list_low[i], n) A(g.list_seg[i], (size_t)n + high_items) }
    for (int i = 0; i < 3; ++i) { A(g.fr_low[i], n) A(g.fr_seg[i], (size_t)n + high_items) }
    c->tally_slots = kSweepWaves;
    A(g.dmax_part, (size_t)kSweepWaves)
    A(g.tally, (size_t)c->tally_slots * kTallyN) A(c->d_tally_out, kTallyN) A(g.ctrl, 1)
    i64 *d_spare;
    A(d_spare, 8)
    if (std::getenv("PSN_TRACE")) { A(g.dbg, kDbgCap) A(g.dbg_cnt, 1) }
#undef A
    c->d_head = d_head; c->d_rev = d_rev;
    // purses and sharded accumulators for the hubs pay from about two million residual arcs (measured: 8 % less
    // solve time at 10.7 million, 5 % more at 1.1 million)
    g.hub_shards = m2 >= (int64_t)kHubShardsMinArcs ? kHubShards : 0;
    if (const char *ev = std::getenv("PSN_HUB_SHARDS")) g.hub_shards = atoi(ev) ? kHubShards : 0;  // development
    g.row = d_row; g.head = d_head; g.cost = c->d_cost; g.rev = d_rev; g.tail = d_tail; g.ocap = c->d_ocap; g.cls = d_cls; c->d_cls = d_cls; g.lean_stop = c->d_lean_stop;
    g.hub_of = d_hub_of; c->d_hub_of = d_hub_of; g.pull_list = c->d_pull_list; g.pull_cnt = c->d_pull_cnt; g.hub_vertex = d_hub_vertex; g.nhubs = nhubs;
    g.hub_row0 = d_hub_row0; g.hub_seg0 = d_hub_seg0; g.seg1_of = d_seg1_of; g.n_hub_segs = n_hub_segs;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(d_row, row.data(), sizeof(unsigned) * ((size_t)n + 1), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_cls, cls.data(), (size_t)n, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_hub_of, hub_of.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_hub_vertex, hub_vertex.data(), sizeof(int) * (size_t)kMaxHubs, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_hub_row0, hub_row0.data(), sizeof(unsigned) * (size_t)kMaxHubs, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_hub_seg0, hub_seg0.data(), sizeof(int) * (size_t)kMaxHubs, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_seg1_of, seg1_of.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(g.price, 0, sizeof(i64) * (size_t)n, s));
    HIPCHK(c, hipMemsetAsync(d_spare, 0, sizeof(i64) * 8, s));
    g.spare = d_spare;
    if (m2 > 0) hipLaunchKernelGGL(k_fill_tail, dim3(kGrid), dim3(kBlock), 0, s, d_row, (int)n, (unsigned)m2, d_tail);
    HIPCHK(c, hipStreamSynchronize(s));  // the staging vectors above die at return
    return PSN_OK;
}

int psn_upload_csr(psn_ctx *c, int32_t n, int64_t m2, const int64_t *row_off, const int32_t *head, const int32_t *cap,
                   const int32_t *cost, const int32_t *rev, const int64_t *supply) {
    if (!c) return PSN_EINVAL;
    const bool dbg_up = std::getenv("PSN_DEBUG_UPLOAD") != nullptr;
    auto now_up = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tu0 = now_up();
    if (n <= 0 || !row_off || !supply || (m2 > 0 && (!head || !cap || !cost || !rev)))
        return fail(c, PSN_EINVAL, "null or out-of-range argument");
    // ---- validate: offsets and supplies here, the arcs on the device once they are there (the solver
    // kernels only ever see a network that passed; a rejected upload leaves the context empty)
    int64_t ssum = 0;
    for (int32_t v = 0; v < n; ++v) ssum += supply[v];
    if (ssum != 0) return fail(c, PSN_EINFEASIBLE, "supplies do not sum to zero");
    int rc = setup_rows(c, n, m2, row_off);
    if (rc != PSN_OK) return rc;
    const double tu3 = now_up();
    DevGraph &g = c->g;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->d_head, head, sizeof(int) * (size_t)m2, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_cost, cost, sizeof(int) * (size_t)m2, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_rev, rev, sizeof(int) * (size_t)m2, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_ocap, cap, sizeof(int) * (size_t)m2, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(g.cap, cap, sizeof(int) * (size_t)m2, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_supply, supply, sizeof(i64) * (size_t)n, hipMemcpyHostToDevice, s));
    // arc checks (the far end of every pair is a random read: 46 ms for 10 M arcs on the host, well under 1 ms here)
    unsigned long long *d_chk = reinterpret_cast<unsigned long long *>(c->d_scalar), h_chk[2] = {~0ull, 0};
    HIPCHK(c, hipMemcpyAsync(d_chk, h_chk, sizeof(h_chk), hipMemcpyHostToDevice, s));
    if (m2 > 0)
        hipLaunchKernelGGL(k_check_arcs, dim3(kGrid), dim3(kBlock), 0, s, (int)n, (unsigned)m2, c->d_head, g.cap, c->d_cost, c->d_rev, g.tail, d_chk);
    HIPCHK(c, hipMemcpyAsync(h_chk, d_chk, sizeof(h_chk), hipMemcpyDeviceToHost, s));
    if ((rc = launch_cover(c)) != PSN_OK) { free_graph_buffers(c); return rc; }
    HIPCHK(c, hipStreamSynchronize(s));
    if (h_chk[0] != ~0ull) {  // name what is wrong with the first unsound arc
        free_graph_buffers(c);
        const int64_t a = (int64_t)h_chk[0];
        const int32_t r = rev[a];
        const int32_t ta = (int32_t)(std::upper_bound(row_off, row_off + n + 1, a) - row_off) - 1;
        if (head[a] < 0 || head[a] >= n) return fail(c, PSN_EINVAL, "head out of range");
        if (r < 0 || r >= m2 || rev[r] != a) return fail(c, PSN_EINVAL, "rev is not an involution");
        if (r == a && (head[a] != ta || cap[a] != 0 || cost[a] != 0))
            return fail(c, PSN_EINVAL, "an arc may pair with itself only as an unused placeholder (self-loop, cap 0, cost 0)");
        if (head[r] != ta) return fail(c, PSN_EINVAL, "rev does not point at the reverse arc");
        if (cost[r] != -cost[a]) return fail(c, PSN_EINVAL, "cost[rev[a]] != -cost[a]");
        if (cap[a] < 0) return fail(c, PSN_EINVAL, "negative capacity");
        return fail(c, PSN_EOVERFLOW, "capacity pair overflows int32");
    }
    {   // price magnitudes stay below ~ 8 * n * max|cost| * (n+1)
        const long double bound = 8.0L * (long double)n * (long double)(h_chk[1] + 1) * (long double)(n + 1);
        if (bound > 4.0e18L) { free_graph_buffers(c); return fail(c, PSN_EOVERFLOW, "n^2 * max|cost| too large for int64 prices"); }
    }
    scale_for_cover(c, (long long)c->h_flag[2], (long long)c->h_flag[3]);
    const double tu4 = now_up();
    if ((rc = build_graph(c)) != PSN_OK) { free_graph_buffers(c); return rc; }
    if (dbg_up) fprintf(stderr, "psn_upload_csr: checks + rows %.1f copies + arc checks %.1f graph capture %.1f ms\n", tu3 - tu0, tu4 - tu3, now_up() - tu4);
    c->uploaded = true;
    return PSN_OK;
}

// ---- the scheduling network built on the device
// The residual CSR of a scheduling network is a function of its layout and of a few words per task.
// The host works out the row offsets and each task's place in its class row and job row; the arcs,
// five arrays of them, are written where they are used instead of being assembled on the host and
// copied: 300 MB less over PCIe at 10 000 machines.  Rows may have head-room (reverse places that no
// task uses yet, task slots that are free): those arcs are placeholders, self-loops paired with
// themselves.
__global__ void k_build_placeholders(DevGraph g, int *mhead, int *mcost, int *mrev, int *mocap) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        mhead[a] = g.tail[a]; mrev[a] = (int)a; mcost[a] = 0; mocap[a] = 0; g.cap[a] = 0;
    }
}
__global__ void k_build_task_arcs(DevGraph g, long long count, int v_ec, int v_un, int v_t, int unsched_cost, const int *task_slot,
                                  const int *task_class, const int *task_job, const int *pos_class, const int *pos_job, int *mhead, int *mcost,
                                  int *mrev, int *mocap, i64 *supply) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const int e = task_class[i], j = task_job[i], vt = v_t + task_slot[i];
        const unsigned ta = g.row[vt], re = g.row[v_ec + e] + (unsigned)pos_class[i], rj = g.row[v_un + j] + (unsigned)pos_job[i];
        mhead[ta] = v_ec + e; mocap[ta] = 1; g.cap[ta] = 1; mcost[ta] = 0; mrev[ta] = (int)re;
        mhead[re] = vt; mocap[re] = 0; g.cap[re] = 0; mcost[re] = 0; mrev[re] = (int)ta;
        mhead[ta + 1] = v_un + j; mocap[ta + 1] = 1; g.cap[ta + 1] = 1; mcost[ta + 1] = unsched_cost; mrev[ta + 1] = (int)rj;
        mhead[rj] = vt; mocap[rj] = 0; g.cap[rj] = 0; mcost[rj] = -unsched_cost; mrev[rj] = (int)(ta + 1);
        supply[vt] = 1;
    }
}
__global__ void k_build_slot_arcs(DevGraph g, int N, int E, int K, int v_ec, const i64 *ec_arc_base, int *mhead, int *mcost, int *mrev, int *mocap) {
    const long long total = (long long)E * N * K;
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < 2 * total; i += (long long)gridDim.x * kBlock) {
        const bool fwd = i < total;
        const long long x = fwd ? i : i - total;
        // forward threads run machine-major, slot-minor inside a class; reverse threads class-major, slot-minor inside a machine
        int e, m, k;
        if (fwd) { e = (int)(x / ((long long)N * K)); const long long r = x - (long long)e * N * K; m = (int)(r / K); k = (int)(r % K); }
        else { m = (int)(x / ((long long)E * K)); const long long r = x - (long long)m * E * K; e = (int)(r / K); k = (int)(r % K); }
        const unsigned a = (unsigned)(ec_arc_base[e] + (i64)m * K + k), r = g.row[1 + m] + (unsigned)e * (unsigned)K + (unsigned)k;
        if (fwd) { mhead[a] = 1 + m; mocap[a] = 0; g.cap[a] = 0; mcost[a] = 0; mrev[a] = (int)r; }
        else { mhead[r] = v_ec + e; mocap[r] = 0; g.cap[r] = 0; mcost[r] = 0; mrev[r] = (int)a; }
    }
}
__global__ void k_build_sink_arcs(DevGraph g, int N, int E, int J, int K, int v_un, const int *room_job, const int *machine_sink_cap,
                                  const int *job_sink_cap, int *mhead, int *mcost, int *mrev, int *mocap) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N + J; i += gridDim.x * kBlock) {
        const bool machine = i < N;
        const int v = machine ? 1 + i : v_un + (i - N);
        const unsigned a = machine ? g.row[v] + (unsigned)E * (unsigned)K : g.row[v] + (unsigned)room_job[i - N];
        const unsigned r = g.row[0] + (unsigned)i;
        const int cp = machine ? machine_sink_cap[i] : job_sink_cap[i - N];
        mhead[a] = 0; mocap[a] = cp; g.cap[a] = cp; mcost[a] = 0; mrev[a] = (int)r;
        mhead[r] = v; mocap[r] = 0; g.cap[r] = 0; mcost[r] = 0; mrev[r] = (int)a;
    }
}

int psn_build_network_rooms_device(psn_ctx *c, const psn_layout *lay, int32_t unsched_cost, const int32_t *room_class,
                                   const int32_t *room_job, const int32_t *machine_sink_cap, const int32_t *job_sink_cap, int64_t n_attached,
                                   const int32_t *task_slot, const int32_t *task_class, const int32_t *task_job, const int32_t *place_class,
                                   const int32_t *place_job, int64_t *ec_arc_base_out) {
    if (!c || !lay || !room_class || !room_job || !machine_sink_cap || !job_sink_cap || unsched_cost < 0 || n_attached < 0 ||
        (n_attached > 0 && (!task_slot || !task_class || !task_job || !place_class || !place_job)))
        return fail(c, PSN_EINVAL, "null or out-of-range argument");
    const int64_t N = lay->n_machines, E = lay->n_classes, J = lay->n_jobs, P = lay->n_tasks, K = lay->k_slots;
    if (N <= 0 || E <= 0 || J <= 0 || P <= 0 || K <= 0 || n_attached > P) return fail(c, PSN_EINVAL, "bad layout");
    const int64_t nn = 1 + N + E + J + P;
    if (nn > INT32_MAX) return fail(c, PSN_EOVERFLOW, "network too large");
    {   // price magnitudes stay below ~ 8 * n * max|cost| * (n+1); slot costs never exceed 2000 + rounding
        const long double bound = 8.0L * (long double)nn * (long double)(std::max<int64_t>(unsched_cost, 4096) + 1) * (long double)(nn + 1);
        if (bound > 4.0e18L) return fail(c, PSN_EOVERFLOW, "n^2 * max|cost| too large for int64 prices");
    }
    const int32_t n = (int32_t)nn;
    const int64_t v_ec = 1 + N, v_un = 1 + N + E, v_t = 1 + N + E + J;
    std::vector<int64_t> row_off((size_t)n + 1);
    row_off[0] = 0;
    row_off[1] = N + J;
    for (int64_t mm = 0; mm < N; ++mm) {
        if (machine_sink_cap[mm] < 0) return fail(c, PSN_EINVAL, "negative capacity");
        row_off[(size_t)(1 + mm + 1)] = row_off[(size_t)(1 + mm)] + E * K + 1;
    }
    for (int64_t e = 0; e < E; ++e) {
        if (room_class[e] < 0) return fail(c, PSN_EINVAL, "negative room");
        row_off[(size_t)(v_ec + e + 1)] = row_off[(size_t)(v_ec + e)] + room_class[e] + N * K;
    }
    for (int64_t j = 0; j < J; ++j) {
        if (room_job[j] < 0 || job_sink_cap[j] < 0) return fail(c, PSN_EINVAL, "negative room or capacity");
        row_off[(size_t)(v_un + j + 1)] = row_off[(size_t)(v_un + j)] + room_job[j] + 1;
    }
    for (int64_t t = 0; t < P; ++t) row_off[(size_t)(v_t + t + 1)] = row_off[(size_t)(v_t + t)] + 2;
    const int64_t m2 = row_off[(size_t)n];
    if (m2 >= (int64_t)0x7FFFFFF0) return fail(c, PSN_EOVERFLOW, "network too large");
    // every task takes a slot and a place of its own; places lie inside the rooms
    {
        std::vector<char> slot_taken((size_t)P, 0);
        std::vector<std::vector<char>> seen_e((size_t)E), seen_j((size_t)J);
        for (int64_t i = 0; i < n_attached; ++i) {
            const int32_t sl = task_slot[i], e = task_class[i], j = task_job[i];
            if (sl < 0 || sl >= P || e < 0 || e >= E || j < 0 || j >= J || place_class[i] < 0 || place_class[i] >= room_class[e] ||
                place_job[i] < 0 || place_job[i] >= room_job[j] || slot_taken[(size_t)sl])
                return fail(c, PSN_EINVAL, "task slot, class, job or place out of range or taken twice");
            slot_taken[(size_t)sl] = 1;
            auto &se = seen_e[(size_t)e]; if (se.empty()) se.assign((size_t)room_class[e], 0);
            auto &sj = seen_j[(size_t)j]; if (sj.empty()) sj.assign((size_t)room_job[j], 0);
            if (se[(size_t)place_class[i]] || sj[(size_t)place_job[i]]) return fail(c, PSN_EINVAL, "a place in a class or job row is taken twice");
            se[(size_t)place_class[i]] = 1; sj[(size_t)place_job[i]] = 1;
        }
    }
    std::vector<int64_t> base((size_t)E);
    for (int64_t e = 0; e < E; ++e) base[(size_t)e] = row_off[(size_t)(v_ec + e)] + room_class[e];
    int rc = setup_rows(c, n, m2, row_off.data());
    if (rc != PSN_OK) return rc;
    // staging: class arc bases; job rooms, sink capacities; the five task tables
    const size_t A = (size_t)n_attached;
    const size_t stage_bytes = sizeof(i64) * (size_t)E + sizeof(int) * (5 * A + 2 * (size_t)J + (size_t)N) + 16;
    char *d;
    if ((rc = stage(c, stage_bytes, &d)) != PSN_OK) { free_graph_buffers(c); return rc; }
    i64 *d_base = reinterpret_cast<i64 *>(d);
    int *d_roomj = reinterpret_cast<int *>(d_base + E), *d_msink = d_roomj + J, *d_jsink = d_msink + N;
    int *d_slot = d_jsink + J, *d_class = d_slot + A, *d_job = d_class + A, *d_pe = d_job + A, *d_pj = d_pe + A;
    hipStream_t s = c->stream;
    DevGraph &g = c->g;
    hipError_t e = hipMemcpyAsync(d_base, base.data(), sizeof(i64) * (size_t)E, hipMemcpyHostToDevice, s);
    auto up = [&](int *dst, const int32_t *src, size_t cnt) { if (e == hipSuccess && cnt) e = hipMemcpyAsync(dst, src, sizeof(int) * cnt, hipMemcpyHostToDevice, s); };
    up(d_roomj, room_job, (size_t)J); up(d_msink, machine_sink_cap, (size_t)N); up(d_jsink, job_sink_cap, (size_t)J);
    up(d_slot, task_slot, A); up(d_class, task_class, A); up(d_job, task_job, A); up(d_pe, place_class, A); up(d_pj, place_job, A);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_supply, 0, sizeof(i64) * (size_t)n, s);
    const i64 sink = -(i64)n_attached;
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_supply, &sink, sizeof(i64), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_build_placeholders, dim3(kGrid), dim3(kBlock), 0, s, g, c->d_head, c->d_cost, c->d_rev, c->d_ocap);
        if (A) hipLaunchKernelGGL(k_build_task_arcs, dim3((unsigned)std::min<int64_t>(kGrid, ((int64_t)A + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, g, (long long)A,
                                  (int)v_ec, (int)v_un, (int)v_t, (int)unsched_cost, d_slot, d_class, d_job, d_pe, d_pj, c->d_head, c->d_cost, c->d_rev,
                                  c->d_ocap, c->d_supply);
        hipLaunchKernelGGL(k_build_slot_arcs, dim3(4096), dim3(kBlock), 0, s, g, (int)N, (int)E, (int)K, (int)v_ec, d_base, c->d_head, c->d_cost,
                           c->d_rev, c->d_ocap);
        hipLaunchKernelGGL(k_build_sink_arcs, dim3((unsigned)std::min<int64_t>(kGrid, (N + J + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, g, (int)N, (int)E,
                           (int)J, (int)K, (int)v_un, d_roomj, d_msink, d_jsink, c->d_head, c->d_cost, c->d_rev, c->d_ocap);
        e = hipGetLastError();
    }
    if (e == hipSuccess && launch_cover(c) != PSN_OK) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // the host staging vectors die at return
    if (e != hipSuccess) { free_graph_buffers(c); return fail(c, PSN_EDEVICE, "psn_build_network_rooms_device", e); }
    scale_for_cover(c, (long long)c->h_flag[2], (long long)c->h_flag[3]);
    if ((rc = build_graph(c)) != PSN_OK) { free_graph_buffers(c); return rc; }
    if (ec_arc_base_out) std::copy(base.begin(), base.end(), ec_arc_base_out);
    c->uploaded = true;
    return PSN_OK;
}

int psn_build_network_device(psn_ctx *c, const psn_layout *lay, int32_t slots, int32_t unsched_cost, const int32_t *task_class,
                             const int32_t *task_job, int64_t *ec_arc_base_out) {
    if (!c || !lay || !task_class || !task_job || slots < 0 || unsched_cost < 0) return fail(c, PSN_EINVAL, "null or out-of-range argument");
    const int64_t N = lay->n_machines, E = lay->n_classes, J = lay->n_jobs, P = lay->n_tasks, K = lay->k_slots;
    if (N <= 0 || E <= 0 || J <= 0 || P <= 0 || K <= 0) return fail(c, PSN_EINVAL, "bad layout");
    // the exact layout: rooms as large as the membership, tasks in arrival order
    std::vector<int32_t> cnt_e((size_t)E, 0), cnt_j((size_t)J, 0), slot((size_t)P), pe((size_t)P), pj((size_t)P), msink((size_t)N, slots);
    for (int64_t t = 0; t < P; ++t) {
        if (task_class[t] < 0 || task_class[t] >= E || task_job[t] < 0 || task_job[t] >= J) return fail(c, PSN_EINVAL, "task class or job out of range");
        slot[(size_t)t] = (int32_t)t;
        pe[(size_t)t] = cnt_e[(size_t)task_class[t]]++;
        pj[(size_t)t] = cnt_j[(size_t)task_job[t]]++;
    }
    return psn_build_network_rooms_device(c, lay, unsched_cost, cnt_e.data(), cnt_j.data(), msink.data(), cnt_j.data(), P, slot.data(), task_class,
                                          task_job, pe.data(), pj.data(), ec_arc_base_out);
}

static int eval_costs_impl(psn_ctx *c, const psn_layout *lay, const int64_t *cap_cpu, const int64_t *cap_mem, const int64_t *used_cpu,
                           const int64_t *used_mem, const int64_t *req_cpu, const int64_t *req_mem, const int64_t *ec_arc_base,
                           int keep_flow) {
    if (!c || !lay || !cap_cpu || !cap_mem || !used_cpu || !used_mem || !req_cpu || !req_mem || !ec_arc_base)
        return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    const int N = lay->n_machines, E = lay->n_classes, K = lay->k_slots;
    if (N <= 0 || E <= 0 || K <= 0) return fail(c, PSN_EINVAL, "bad layout");
    for (int e = 0; e < E; ++e) {
        if (req_cpu[e] <= 0 || req_mem[e] <= 0) return fail(c, PSN_EINVAL, "request must be positive");
        if (ec_arc_base[e] < 0 || ec_arc_base[e] + (int64_t)N * K > c->m2) return fail(c, PSN_EINVAL, "ec_arc_base out of range");
    }
    for (int m = 0; m < N; ++m) {
        if (cap_cpu[m] <= 0 || cap_mem[m] <= 0 || used_cpu[m] < 0 || used_mem[m] < 0)
            return fail(c, PSN_EINVAL, "machine capacity must be positive, usage non-negative");
        // largest intermediate: 1000 * (used + K * req) must stay in int64 and the cost in int32
        if (cap_mem[m] > (int64_t)4e15 || used_mem[m] > (int64_t)4e15 || cap_cpu[m] > (int64_t)4e15 || used_cpu[m] > (int64_t)4e15)
            return fail(c, PSN_EOVERFLOW, "machine quantity too large");
    }
    if ((long long)1 + N + E > c->n) return fail(c, PSN_EINVAL, "layout does not match the uploaded network");
    // rows of the uploaded network must hold what the layout says: E * K reverse slot arcs at the start of
    // every machine row, N * K slot arcs from ec_arc_base[e] on inside the row of class e
    for (int m = 0; m < N; ++m)
        if ((long long)c->h_row[(size_t)m + 2] - c->h_row[(size_t)m + 1] < (long long)E * K)
            return fail(c, PSN_EINVAL, "a machine row is shorter than classes x slots: layout does not match the uploaded network");
    for (int e = 0; e < E; ++e) {
        const long long lo = c->h_row[(size_t)1 + N + e], hi = c->h_row[(size_t)2 + N + e];
        if (ec_arc_base[e] < lo || ec_arc_base[e] + (int64_t)N * K > hi)
            return fail(c, PSN_EINVAL, "ec_arc_base does not lie inside the row of its class");
    }
    // an admission mask belongs to one (classes, machines) shape; pricing another shape under it would
    // silently lift the selectors
    if (c->allow_on && (c->allow_classes != E || c->allow_words != (N + 31) / 32))
        return fail(c, PSN_EINVAL, "admission mask was set for another layout");
    HIPCHK(c, hipSetDevice(c->device));
    // staging: 4 machine arrays, 2 shape arrays, bases, layout-check counter
    const size_t cnt = 4 * (size_t)N + 3 * (size_t)E + 1;
    char *dst;
    int rcs = stage(c, cnt * sizeof(i64), &dst);
    if (rcs != PSN_OK) return rcs;
    i64 *d = reinterpret_cast<i64 *>(dst);
    std::vector<i64> h(cnt);
    std::copy(cap_cpu, cap_cpu + N, h.begin());
    std::copy(cap_mem, cap_mem + N, h.begin() + N);
    std::copy(used_cpu, used_cpu + N, h.begin() + 2 * (size_t)N);
    std::copy(used_mem, used_mem + N, h.begin() + 3 * (size_t)N);
    std::copy(req_cpu, req_cpu + E, h.begin() + 4 * (size_t)N);
    std::copy(req_mem, req_mem + E, h.begin() + 4 * (size_t)N + E);
    std::copy(ec_arc_base, ec_arc_base + E, h.begin() + 4 * (size_t)N + 2 * (size_t)E);
    h[cnt - 1] = 0;
    unsigned *d_bad = reinterpret_cast<unsigned *>(d + cnt - 1);
    hipError_t e = hipMemcpyAsync(d, h.data(), cnt * sizeof(i64), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const long long total = 2LL * E * N;
        const int grid = (int)std::min<long long>((total + kBlock - 1) / kBlock, 8192);
        const i64 *d_base = d + 4 * (size_t)N + 2 * (size_t)E;
        // the layout pass writes nothing but its counter; the cost kernel returns at once when it is nonzero.
        // A layout that passed once stands until the network is uploaded again or arcs are rewired.
        const bool verified = c->layout_ok_gen == c->upload_gen && c->layout_ok_E == E && c->layout_ok_N == N && c->layout_ok_K == K &&
                              c->layout_ok_base.size() == (size_t)E && std::equal(ec_arc_base, ec_arc_base + E, c->layout_ok_base.begin());
        if (!verified)
            hipLaunchKernelGGL(k_check_slot_layout, dim3(std::max(1, grid / 2)), dim3(kBlock), 0, c->stream, c->g, E, N, K, d_base, d_bad);
        hipLaunchKernelGGL(k_eval_costs, dim3(grid), dim3(kBlock), 0, c->stream, c->g, E, N, K, d, d + N, d + 2 * (size_t)N,
                           d + 3 * (size_t)N, d + 4 * (size_t)N, d + 4 * (size_t)N + E, d_base, c->d_cost,
                           c->d_ocap, keep_flow, c->allow_on ? c->d_allow : nullptr, c->allow_words, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_flag, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // also: the host staging vector dies at return
    if (e != hipSuccess) return fail(c, PSN_EDEVICE, "psn_eval_costs", e);
    if (c->h_flag[0])  // nothing was written: the resident network and its solved state stand as they were
        return fail(c, PSN_EINVAL, "slot arcs and their reverse arcs do not pair up class-major, slot-minor at the start of the machine rows");
    c->layout_ok_gen = c->upload_gen; c->layout_ok_E = E; c->layout_ok_N = N; c->layout_ok_K = K;
    c->layout_ok_base.assign(ec_arc_base, ec_arc_base + E);
    if (!keep_flow) c->solved = false;  // the flow on re-priced arcs was reset
    return PSN_OK;
}

int psn_eval_costs(psn_ctx *c, const psn_layout *lay, const int64_t *cap_cpu, const int64_t *cap_mem, const int64_t *used_cpu,
                   const int64_t *used_mem, const int64_t *req_cpu, const int64_t *req_mem, const int64_t *ec_arc_base) {
    return eval_costs_impl(c, lay, cap_cpu, cap_mem, used_cpu, used_mem, req_cpu, req_mem, ec_arc_base, 0);
}

int psn_reprice_costs(psn_ctx *c, const psn_layout *lay, const int64_t *cap_cpu, const int64_t *cap_mem, const int64_t *used_cpu,
                      const int64_t *used_mem, const int64_t *req_cpu, const int64_t *req_mem, const int64_t *ec_arc_base) {
    return eval_costs_impl(c, lay, cap_cpu, cap_mem, used_cpu, used_mem, req_cpu, req_mem, ec_arc_base, 1);
}

int psn_set_admission_mask(psn_ctx *c, const psn_layout *lay, const uint32_t *mask) {
    if (!c || !lay) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    const int N = lay->n_machines, E = lay->n_classes;
    if (N <= 0 || E <= 0) return fail(c, PSN_EINVAL, "bad layout");
    if (!mask) { c->allow_on = false; return PSN_OK; }  // every class may use every machine again
    if ((long long)1 + N + E > c->n) return fail(c, PSN_EINVAL, "layout does not match the uploaded network");
    const int words = (N + 31) / 32;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_allow || c->allow_words != words || c->allow_classes != E) {
        int rc = dalloc(c, &c->d_allow, (size_t)E * words);  // freed with the network's other buffers
        if (rc != PSN_OK) return rc;
        c->allow_words = words;
        c->allow_classes = E;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_allow, mask, (size_t)E * words * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->allow_on = true;
    return PSN_OK;
}

int psn_patch_arcs(psn_ctx *c, int64_t count, const int64_t *arc_index, const int32_t *head, const int32_t *cap, const int32_t *cost,
                   const int32_t *rev) {
    if (!c || count < 0 || (count > 0 && (!arc_index || !head || !cap || !cost || !rev))) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    if (count == 0) return PSN_OK;
    // 8 n (|cost| + 1) (n + 1) must stay below 4e18: the largest |cost| that does, worked out once
    const long double cost_room = 4.0e18L / (8.0L * (long double)c->n * (long double)(c->n + 1)) - 1.0L;
    const int64_t max_abs_cost = cost_room >= 2147483648.0L ? (int64_t)2147483648LL : (cost_room < 0 ? -1 : (int64_t)cost_room);
    for (int64_t i = 0; i < count; ++i) {
        if (arc_index[i] < 0 || arc_index[i] >= c->m2 || rev[i] < 0 || rev[i] >= c->m2 || head[i] < 0 || head[i] >= c->n || cap[i] < 0)
            return fail(c, PSN_EINVAL, "patched arc out of range");
        const int64_t ac = cost[i] < 0 ? -(int64_t)cost[i] : cost[i];
        if (ac > max_abs_cost) return fail(c, PSN_EOVERFLOW, "n^2 * |cost| too large for int64 prices");
    }
    const bool dbg_t = std::getenv("PSN_DEBUG_SVC") != nullptr;
    auto now_p = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = now_p();
    HIPCHK(c, hipSetDevice(c->device));
    // one staging block: indices (8 B) then four int arrays
    const size_t bytes = (size_t)count * (sizeof(i64) + 4 * sizeof(int)) + 2 * sizeof(unsigned);
    char *d;
    { const int rcs = stage(c, bytes, &d); if (rcs != PSN_OK) return rcs; }
    i64 *d_idx = (i64 *)d;
    int *d_h = (int *)(d + (size_t)count * sizeof(i64)), *d_c = d_h + count, *d_k = d_c + count, *d_r = d_k + count;
    unsigned *d_bad = (unsigned *)(d_r + count);  // [0] arcs that do not pair up, [1] arcs that break the tables made from the heads
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(d_idx, arc_index, (size_t)count * sizeof(i64), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_h, head, (size_t)count * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, cap, (size_t)count * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_k, cost, (size_t)count * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_r, rev, (size_t)count * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 2 * sizeof(unsigned), s);
    unsigned bad = 0;
    if (e == hipSuccess) {
        const int grid = (int)std::min<long long>((count + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(k_patch_arcs, dim3(grid), dim3(kBlock), 0, s, c->g, (long long)count, d_idx, d_h, d_c, d_k, d_r, c->d_head,
                           c->d_cost, c->d_rev, c->d_ocap);
        hipLaunchKernelGGL(k_check_patched, dim3(grid), dim3(kBlock), 0, s, c->g, (long long)count, d_idx, d_bad);
        hipLaunchKernelGGL(k_patch_keeps_tables, dim3(grid), dim3(kBlock), 0, s, c->g.row, c->g.n, (long long)count, d_idx, c->g.head, c->g.tail,
                           c->d_in_cover, c->d_not_indep, c->d_lean_stop, c->d_hub_of, c->d_cls, d_bad + 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_flag, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_flag + 4, d_bad + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s);
    c->layout_ok_gen = ~0ull;  // pairings change: the slot-block layout is verified again by the next pricing
    // The tables made from the heads (vertex cover and what goes with it: four sweeps over all arcs, 0.7 ms
    // at 13 M) are made again only if a patched arc breaks one of them (PSN_PATCH_TABLES=1: always).
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    bool tables_again = false;
    if (e == hipSuccess && c->h_flag[0] == 0 && (c->h_flag[4] != 0 || std::getenv("PSN_PATCH_TABLES"))) {
        tables_again = true;
        const int rcl = launch_cover(c);  // (the error text is launch_cover's)
        if (rcl != PSN_OK) return rcl;
        e = hipStreamSynchronize(s);
    }
    if (dbg_t) fprintf(stderr, "    patch_arcs: %.3f ms, tables made again: %d\n", now_p() - tp0, (int)tables_again);
    if (e != hipSuccess) return fail(c, PSN_EDEVICE, "psn_patch_arcs", e);
    bad = c->h_flag[0];
    if (bad) {
        c->uploaded = c->solved = false;  // the device network is inconsistent: upload again
        return fail(c, PSN_EINVAL, "patched arcs do not pair up; the network must be uploaded again");
    }
    if (tables_again) {
        c->cover_size = (long long)c->h_flag[2];
        c->noni_size = (long long)c->h_flag[3];
    }
    if (c->cover_size > c->cover_cap || c->noni_size > c->noni_cap) {
        // longer residual cycles are possible now: a larger cost scale, under which the kept prices mean
        // nothing, so the next solve starts cold; the replay graph carries the scale and is captured anew
        scale_for_cover(c, c->cover_size, c->noni_size);
        c->solved = false;
        ++c->upload_gen;
        drop_group_cache(c);
        const int rcg = build_graph(c);
        if (rcg != PSN_OK) return rcg;
    }
    return PSN_OK;
}

int psn_settle_sources(psn_ctx *c, int64_t count, const int32_t *vertex, int64_t *skipped) {
    if (!c || count < 0 || (count > 0 && !vertex)) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    if (skipped) *skipped = 0;
    if (count == 0) return PSN_OK;
    for (int64_t i = 0; i < count; ++i)
        if (vertex[i] < 0 || vertex[i] >= c->n) return fail(c, PSN_EINVAL, "vertex out of range");
    HIPCHK(c, hipSetDevice(c->device));
    char *d;
    { const int rcs = stage(c, (size_t)count * sizeof(int) + sizeof(unsigned), &d); if (rcs != PSN_OK) return rcs; }
    int *d_v = (int *)d;
    unsigned *d_skip = (unsigned *)(d_v + count);
    unsigned skip = 0;
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(d_v, vertex, (size_t)count * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_skip, 0, sizeof(unsigned), s);
    if (e == hipSuccess) {
        const int grid = (int)std::min<long long>((count + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(k_settle_sources, dim3(grid), dim3(kBlock), 0, s, c->g, (long long)count, d_v, d_skip);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_flag, d_skip, sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(c, PSN_EDEVICE, "psn_settle_sources", e);
    skip = c->h_flag[0];
    if (skipped) *skipped = (int64_t)skip;
    return PSN_OK;
}

int psn_update_caps(psn_ctx *c, int64_t count, const int64_t *arc_index, const int32_t *new_cap) {
    if (!c || count < 0 || (count > 0 && (!arc_index || !new_cap))) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    if (count == 0) return PSN_OK;
    for (int64_t i = 0; i < count; ++i)
        if (arc_index[i] < 0 || arc_index[i] >= c->m2 || new_cap[i] < 0) return fail(c, PSN_EINVAL, "arc index or capacity out of range");
    HIPCHK(c, hipSetDevice(c->device));
    char *d;
    { const int rcs = stage(c, (size_t)count * (sizeof(i64) + sizeof(int)), &d); if (rcs != PSN_OK) return rcs; }
    i64 *d_idx = (i64 *)d;
    int *d_cap = (int *)(d_idx + count);
    hipError_t e = hipMemcpyAsync(d_idx, arc_index, (size_t)count * sizeof(i64), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cap, new_cap, (size_t)count * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const int grid = (int)std::min<long long>((count + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(k_scatter_caps, dim3(grid), dim3(kBlock), 0, c->stream, c->g, (long long)count, d_idx, d_cap, c->d_ocap);
        hipLaunchKernelGGL(k_clamp_flow, dim3(kGrid), dim3(kBlock), 0, c->stream, c->g);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, PSN_EDEVICE, "psn_update_caps", e);
    return PSN_OK;
}

int psn_update_costs(psn_ctx *c, int64_t count, const int64_t *arc_index, const int32_t *new_cost) {
    if (!c || count < 0 || (count > 0 && (!arc_index || !new_cost))) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    if (count == 0) return PSN_OK;
    for (int64_t i = 0; i < count; ++i) {
        if (arc_index[i] < 0 || arc_index[i] >= c->m2 || new_cost[i] == INT32_MIN) return fail(c, PSN_EINVAL, "arc index or cost out of range");
        const int64_t ac = new_cost[i] < 0 ? -(int64_t)new_cost[i] : new_cost[i];
        const long double bound = 8.0L * (long double)c->n * (long double)(ac + 1) * (long double)(c->n + 1);
        if (bound > 4.0e18L) return fail(c, PSN_EOVERFLOW, "n^2 * |cost| too large for int64 prices");
    }
    HIPCHK(c, hipSetDevice(c->device));
    char *d;
    { const int rcs = stage(c, (size_t)count * (sizeof(i64) + sizeof(int)), &d); if (rcs != PSN_OK) return rcs; }
    i64 *d_idx = (i64 *)d;
    int *d_cost = (int *)(d_idx + count);
    hipError_t e = hipMemcpyAsync(d_idx, arc_index, (size_t)count * sizeof(i64), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cost, new_cost, (size_t)count * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const int grid = (int)std::min<long long>((count + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(k_scatter_costs, dim3(grid), dim3(kBlock), 0, c->stream, c->g, (long long)count, d_idx, d_cost, c->d_cost);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, PSN_EDEVICE, "psn_update_costs", e);
    return PSN_OK;
}

int psn_update_supplies(psn_ctx *c, int64_t count, const int32_t *vertex, const int64_t *new_supply) {
    if (!c || count < 0 || (count > 0 && (!vertex || !new_supply))) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    if (count == 0) return PSN_OK;
    for (int64_t i = 0; i < count; ++i)
        if (vertex[i] < 0 || vertex[i] >= c->n) return fail(c, PSN_EINVAL, "vertex out of range");
    HIPCHK(c, hipSetDevice(c->device));
    char *d;
    { const int rcs = stage(c, (size_t)count * (sizeof(i64) + sizeof(int)), &d); if (rcs != PSN_OK) return rcs; }
    i64 *d_s = (i64 *)d;
    int *d_v = (int *)(d_s + count);
    hipError_t e = hipMemcpyAsync(d_v, vertex, (size_t)count * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_s, new_supply, (size_t)count * sizeof(i64), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const int grid = (int)std::min<long long>((count + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(k_scatter_supply, dim3(grid), dim3(kBlock), 0, c->stream, (long long)count, d_v, d_s, c->d_supply);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, PSN_EDEVICE, "psn_update_supplies", e);
    return PSN_OK;
}

// ---- a solve in pieces, shared by the single-network call and the lockstep group
// Launches that bring the working state to the start of a solve and ask for the statistics the
// first epsilon is derived from (h_scalar: kScalarSlots words, read after the stream has drained).
static int solve_prepare(psn_ctx *c, hipStream_t s, bool warm, i64 *h_scalar) {
    const DevGraph &g = c->g;
    hipLaunchKernelGGL(k_reset, dim3(kGrid), dim3(kBlock), 0, s, g, c->d_supply, warm ? 1 : 0);
    hipLaunchKernelGGL(k_mirror, dim3(kGrid), dim3(kBlock), 0, s, g, c->d_supply, warm ? 1 : 0);
    if (warm) hipLaunchKernelGGL(k_excess_from_flow, dim3(kGrid), dim3(kBlock), 0, s, g);
    HIPCHK(c, hipMemsetAsync(c->d_scalar, 0, kScalarSlots * sizeof(i64), s));
    HIPCHK(c, hipMemsetAsync(g.tally, 0, (size_t)c->tally_slots * kTallyN * sizeof(unsigned long long), s));
    if (g.dbg_cnt) HIPCHK(c, hipMemsetAsync(g.dbg_cnt, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_cost_stats, dim3(kGrid), dim3(kBlock), 0, s, g, c->d_scalar + 2);  // (a warm solve compares its first epsilon with a cold one's)
    if (warm) hipLaunchKernelGGL(k_max_violation, dim3(kGrid), dim3(kBlock), 0, s, g, c->d_scalar + 1);
    hipLaunchKernelGGL(k_sum_supply, dim3(std::max(1, std::min(64, (g.n + kBlock - 1) / kBlock))), dim3(kBlock), 0, s, c->d_supply, g.n, c->d_scalar + kSupplySumSlot, c->d_scalar + kSourcesSlot, c->d_scalar + kSupplyUnitsSlot);
    HIPCHK(c, hipMemcpyAsync(h_scalar, c->d_scalar, kScalarSlots * sizeof(i64), hipMemcpyDeviceToHost, s));
    return PSN_OK;
}
// log2 of the epsilon the first phase of a cold solve runs at.
// Zero flow with zero prices is 0-optimal when no usable arc has a negative cost, so the
// first phase may start anywhere; starting at the 90th-percentile cost bucket instead of
// the largest cost skips the phases in which only a few expensive arcs (the unscheduled
// penalty) are distinguishable.  Any starting epsilon is correct: every phase first
// saturates the arcs that violate its epsilon.
static int cold_first_shift(const psn_ctx *c, const i64 *h_scalar, int alpha_log2) {
    i64 c0 = std::max<i64>(1, h_scalar[2]);
    if (h_scalar[3] == 0 && !(c->cfg.flags & PSN_CFG_EPS_FROM_MAX)) {
        i64 total = 0, run = 0;
        for (int b = 0; b < 32; ++b) total += h_scalar[4 + b];
        for (int b = 0; b < 32; ++b) {
            run += h_scalar[4 + b];
            if (run * 10 >= total * 9) { c0 = std::min<i64>(c0, (i64)1 << b); break; }
        }
    }
    return std::max(0, ceil_log2(c0 * c->g.scale) - alpha_log2);
}
// ... and of a warm one: the kept pseudoflow is v-optimal for v = largest violation; resume one step below it.
static int warm_first_shift(const i64 *h_scalar, int alpha_log2) {
    return h_scalar[1] > 1 ? std::max(0, ceil_log2(h_scalar[1]) - alpha_log2) : 0;
}
static int warm_alpha_log2(const psn_ctx *c) {
    if (const char *ev = std::getenv("PSN_WARM_ALPHA_LOG2")) return std::max(1, std::min(12, atoi(ev)));
    return c->cfg.alpha_log2;
}
// The control block a solve starts from (into the context's pinned copy) and its upload.
static Ctrl solve_init(psn_ctx *c, bool warm, const i64 *h_scalar) {
    const DevGraph &g = c->g;
    Ctrl init{};
    init.alpha_log2 = c->cfg.alpha_log2;
    {
        // Phases other than the last stop when little excess is left on its way (see k_begin).  The
        // last phase has to route what they leave, one bounce after the other; how much it takes
        // without costing more than the tails saved grows a little faster than the square root of
        // the supply (measured with phases that step by 8: about 20 units of 1 000, 60-95 of 10 000,
        // 350-450 of 100 000).
        const i64 supply_units = h_scalar[kSupplyUnitsSlot];
        double tf = kTailFactor, tp = kTailPower;
        if (const char *ev = std::getenv("PSN_TAIL_FACTOR")) tf = atof(ev);  // development: the law's two constants
        if (const char *ev = std::getenv("PSN_TAIL_POWER")) tp = atof(ev);
        i64 u = (i64)(tf * std::pow((double)std::max<i64>(0, supply_units), tp));
        if (const char *ev = std::getenv("PSN_TAIL_DIV")) u = supply_units / std::max(1, atoi(ev));  // tests: far more
        if (const char *ev = std::getenv("PSN_TAIL_UNITS")) u = std::max(0, atoi(ev));
        init.tail_units = (int)std::min<i64>(65534, u);
    }
    int shift0;
    if (!warm) {
        shift0 = cold_first_shift(c, h_scalar, init.alpha_log2);
    } else {
        init.alpha_log2 = warm_alpha_log2(c);
        shift0 = warm_first_shift(h_scalar, init.alpha_log2);
    }
    init.eps_shift = shift0;
    init.eps = (i64)1 << shift0;
    init.need_sat = 1;
    init.first_units = ~0u;
    // replays on the reversed network (see Orient in the kernels and k_begin): from an epsilon of 64 cost units down
    init.mirror_max_shift = ceil_log2(g.scale) + 6;
    init.mirror_ratio = 3;  // in quarters: as soon as the deficits sit at no more than 4/3 as many vertices as this side has work items
    if (const char *ev = std::getenv("PSN_MIRROR_DELTA")) init.mirror_max_shift = ceil_log2(g.scale) + atoi(ev);  // development: relative to one cost unit
    if (const char *ev = std::getenv("PSN_MIRROR_SHIFT")) init.mirror_max_shift = atoi(ev);  // development: -1 never, 62 in every phase
    if (const char *ev = std::getenv("PSN_MIRROR_RATIO")) init.mirror_ratio = std::max(0, atoi(ev));
    // supplies that do not sum to zero cannot be routed: the solve ends at its first control step
    if (h_scalar[kSupplySumSlot] != 0) init.error = PSN_EINFEASIBLE_DEV;
    return init;
}
static int solve_start(psn_ctx *c, hipStream_t s, bool warm, const i64 *h_scalar) {
    *c->h_ctrl = solve_init(c, warm, h_scalar);
    HIPCHK(c, hipMemcpyAsync(c->g.ctrl, c->h_ctrl, sizeof(Ctrl), hipMemcpyHostToDevice, s));
    return PSN_OK;
}
// Objective and work counters of the finished solve (h_cost: 1 word, h_tally: kTallyN words, and
// the context's pinned control block; all valid after the stream has drained).
static int solve_finish(psn_ctx *c, hipStream_t s, i64 *h_cost, unsigned long long *h_tally) {
    const DevGraph &g = c->g;
    hipLaunchKernelGGL(k_flip, dim3(std::max(1, std::min(kGrid, (g.n + kBlock - 1) / kBlock))), dim3(kBlock), 0, s, g, 1);  // prices and excesses the way the network is stored
    hipLaunchKernelGGL(k_total_cost, dim3(kGrid), dim3(kBlock), 0, s, g, c->d_scalar);
    hipLaunchKernelGGL(k_tally_sum, dim3(1), dim3(64 * kTallyN), 0, s, g, c->d_tally_out, c->tally_slots);
    HIPCHK(c, hipMemcpyAsync(h_cost, c->d_scalar, sizeof(i64), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(h_tally, c->d_tally_out, kTallyN * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_ctrl, g.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, s));
    return PSN_OK;
}
static void solve_report(psn_ctx *c, psn_result *out, int status, i64 cost, const unsigned long long *h_tally, long long replays,
                         int launches_per_replay, float ms, bool warm, int first_shift) {
    out->warm_started = warm ? 1 : 0;
    out->total_cost = cost;
    out->status = status;
    out->refines = (int32_t)c->h_ctrl->tally_refines;
    out->arc_scans = h_tally[kTallyScans];
    out->pushes = h_tally[kTallyPushes];
    out->relabels = h_tally[kTallyRelabels];
    out->vertex_visits = h_tally[kTallyVisits];
    out->gu_arc_scans = h_tally[kTallyGuScans];
    out->gu_relaxations = h_tally[kTallyGuRelax];
    out->sat_arc_scans = h_tally[kTallySat];
    out->rounds = h_tally[kTallyRounds];
    out->gu_iterations = h_tally[kTallyGuIters];
    out->global_updates = c->h_ctrl->tally_gus;
    out->launches = (uint64_t)replays * (uint64_t)launches_per_replay;
    out->solve_ms = ms;
    out->gu_arc_skips = h_tally[kTallyGuSkips];
    out->arc_skips = h_tally[kTallySkips];
    out->mirrored_replays = (int32_t)c->h_ctrl->tally_mirror;
    out->replays = (uint64_t)replays;
    out->first_units = c->h_ctrl->first_units == ~0u ? 0 : c->h_ctrl->first_units;
    out->last_phase_replays = (int32_t)c->h_ctrl->last_phase_replays;
    out->first_eps_log2 = first_shift;
    c->solved = status == PSN_OK;
}

int psn_solve(psn_ctx *c, uint32_t flags, psn_result *out) {
    if (!c || !out) return fail(c, PSN_EINVAL, "null argument");
    std::memset(out, 0, sizeof(*out));
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    bool warm = (flags & PSN_WARM_START) && c->solved;
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < PSN_KT_CLASSES; ++i) { c->kt_ms[i] = 0; c->kt_calls[i] = 0; }
    hipStream_t s = c->stream;
    const DevGraph &g = c->g;
    HIPCHK(c, hipEventRecord(c->ev0, s));
    i64 h_scalar[kScalarSlots] = {};
    int rc = solve_prepare(c, s, warm, h_scalar);
    if (rc != PSN_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(s));
    bool declined = false;
    if (warm) {
        // A kept state that is violated by nearly as much as zero flow with zero prices is no head start: the
        // scaling runs the same phases from it and, measured on re-priced 10 000-machine clusters, more
        // replays (37-44 against 33).  Within kWarmDeclineSteps powers of two of a cold solve's first epsilon
        // the solve starts cold instead; the result says so.
        int steps = kWarmDeclineSteps;
        if (const char *ev = std::getenv("PSN_WARM_DECLINE")) steps = atoi(ev);  // development: -1 never declines
        if (steps >= 0 && warm_first_shift(h_scalar, warm_alpha_log2(c)) + steps >= cold_first_shift(c, h_scalar, c->cfg.alpha_log2)) {
            warm = false;
            declined = true;
            if ((rc = solve_prepare(c, s, false, h_scalar)) != PSN_OK) return rc;
            HIPCHK(c, hipStreamSynchronize(s));
        }
    }
    if ((rc = solve_start(c, s, warm, h_scalar)) != PSN_OK) return rc;
    const int first_shift = c->h_ctrl->eps_shift;
    // ---- replay until done
    long long max_replays = 200000;
    if (const char *ev = std::getenv("PSN_MAX_REPLAYS")) max_replays = std::max(1, atoi(ev));  // development: stop a solve half way (PSN_ELIMIT)
    long long replays = 0;
    int status = PSN_OK;
    const bool dbg_phases = std::getenv("PSN_DEBUG_SOLVE") != nullptr;
    const auto t_host0 = std::chrono::steady_clock::now();
    std::string phase_log;
    int last_shift = -1; long long phase_replays = 0;
    // Graph replays are kept two deep in the stream: while the host looks at the control block replay i
    // left behind, replay i + 1 is already running, so the device never waits for the host between
    // replays.  A replay issued after the solve is over finds every stage switched off.
    const int depth = c->gexec ? 2 : 1;
    long long issued = 0, checked = 0;
    auto issue = [&]() -> int {
        const int slot = (int)(issued & 1);
        if (c->gexec) {
            HIPCHK(c, hipGraphLaunch(c->gexec, s));
        } else {
            enqueue_replay(c, s);
            HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, hipMemcpyAsync(&c->h_ctrl[slot], g.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->ev_rep[slot], s));
        ++issued;
        return PSN_OK;
    };
    for (int i = 0; i < depth; ++i) if ((rc = issue()) != PSN_OK) return rc;
    for (;;) {
        const Ctrl &h = c->h_ctrl[checked & 1];
        HIPCHK(c, hipEventSynchronize(c->ev_rep[checked & 1]));
        ++checked;
        if (depth == 1) {
            for (size_t i = 0; i + 1 < c->kt_used; i += 2) {  // event pairs of this replay
                float ems = 0.f;
                const int cls = c->kt_cls[i];
                if (cls >= 0 && hipEventElapsedTime(&ems, c->kt_ev[i], c->kt_ev[i + 1]) == hipSuccess) {
                    c->kt_ms[cls] += ems;
                    c->kt_calls[cls] += 1;
                }
            }
            c->kt_used = 0;
        }
        if (dbg_phases) {
            {
                const unsigned *lc = h.list_cnt[c->rounds & 1];
                if (h.rounds_on && std::getenv("PSN_DEBUG_PHASES")) {
                    const double at = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
                    fprintf(stderr, "     shift %d%s: items %u units %u deficit vertices %u  at %.2f ms (host clock)\n", h.eps_shift, h.mirror ? " mirrored" : "", lc[kSlotLow] + lc[kSlotSeg], lc[kSlotUnits], lc[kSlotDeficits], at);
                }
            }
            if (h.eps_shift != last_shift && last_shift >= 0) { phase_log += " " + std::to_string(last_shift) + ":" + std::to_string(phase_replays + 1); phase_replays = 0; }
            else ++phase_replays;
            last_shift = h.eps_shift;
        }
        if (h.error) { status = h.error == PSN_EINFEASIBLE_DEV ? PSN_EINFEASIBLE : PSN_EDEVICE; break; }
        if (h.done) break;
        {   // the rounds of this replay emptied the work list at epsilon 1: solved (what the next k_begin would find)
            const unsigned *lc = h.list_cnt[c->rounds & 1];
            if (h.rounds_on && lc[kSlotLow] + lc[kSlotSeg] == 0 && h.eps_shift == 0) break;
        }
        if (checked >= max_replays) { status = PSN_ELIMIT; break; }
        if ((rc = issue()) != PSN_OK) return rc;
    }
    replays = issued;
    unsigned long long h_tally[kTallyN];
    i64 h_cost = 0;
    if ((rc = solve_finish(c, s, &h_cost, h_tally)) != PSN_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, s));
    HIPCHK(c, hipStreamSynchronize(s));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    solve_report(c, out, status, h_cost, h_tally, replays, c->launches_per_replay, ms, warm, first_shift);
    out->warm_declined = declined ? 1 : 0;
    if (std::getenv("PSN_DEBUG_SOLVE"))
        fprintf(stderr, "psn_solve: %s first shift %d (max violation %lld) refines %d replays %lld rounds %llu gus %llu gu_iters %llu pushes %llu relabels %llu %.2f ms\n",
                warm ? "warm" : "cold", ceil_log2(std::max<i64>(1, warm ? h_scalar[1] : h_scalar[2] * g.scale)) - c->cfg.alpha_log2, (long long)h_scalar[1], out->refines, replays,
                (unsigned long long)out->rounds, (unsigned long long)out->global_updates, (unsigned long long)out->gu_iterations, (unsigned long long)out->pushes,
                (unsigned long long)out->relabels, ms);
    if (dbg_phases) fprintf(stderr, "   replays per phase (shift:replays)%s %d:%lld\n", phase_log.c_str(), last_shift, phase_replays);
    if (status != PSN_OK) return fail(c, status, h_scalar[kSupplySumSlot] != 0 ? "supplies do not sum to zero" : "solve did not converge");
    return PSN_OK;
}

// ---- lockstep group: several resident networks advance through one replay sequence
// One launch per stage serves every network of the group (blockIdx.y picks it), so a group of
// G small networks costs the launches of one.  Members must live on one device and share the
// solver configuration; everything runs on the first member's stream.
namespace {
constexpr int kMaxGroup = 256;

int group_enqueue_replay(const std::vector<psn_ctx *> &m, const DevGraph *d_graphs, int *d_status, hipStream_t s, int *launches_out) {
    const unsigned G = (unsigned)m.size();
    int gv = 1, gv1 = 1, gs = 32, launches = 0;
    for (psn_ctx *c : m) {
        gv = std::max(gv, std::min(kGrid, (c->g.n + kBlock * 2 - 1) / (kBlock * 2)));
        gv1 = std::max(gv1, std::min(kGrid, (c->g.n + kBlock - 1) / kBlock));
        gs = std::max<int>(gs, (int)std::min<long long>(kGrid, (c->m2 + kBlock * 16 - 1) / (kBlock * 16)));
    }
    // Arc sweeps of a group: the networks bring the parallelism, so each gets only its share of a
    // launch that fills the chip a few times over (a workgroup that finds nothing to do still
    // costs its start-up, and a big group would start tens of thousands of them).
    int target = 2048;
    if (const char *ev = std::getenv("PSN_GROUP_BLOCKS")) target = std::max(256, atoi(ev));
    gs = std::max(16, std::min(gs, (target + (int)G - 1) / (int)G));
    const int rounds = m[0]->rounds;
#define LG(kern, grid, ...) do { hipLaunchKernelGGL(kern, dim3(grid, G), dim3(kBlock), 0, s, d_graphs, ##__VA_ARGS__); ++launches; } while (0)
    hipLaunchKernelGGL(k_begin_group, dim3(1, G), dim3(1), 0, s, d_graphs, rounds); ++launches;
    LG(k_flip_group, gv1, 0);
    LG(k_saturate_group, gs);
    LG(k_gu_init_group, gv);
    for (int k = 0; k < kGuIters; ++k) LG(k_gu_expand_group, gs, k);
    LG(k_gu_apply_group, gv1, kGuIters);
    LG(k_relabel_compact_group, gv, 0, 0);
    for (int k = 0; k < rounds; ++k) {
        LG(k_push_group, gs, k);
        LG(k_relabel_compact_group, gv, k, 1);
    }
    hipLaunchKernelGGL(k_group_status, dim3(1), dim3(kMaxGroup), 0, s, d_graphs, (int)G, rounds, d_status); ++launches;
#undef LG
    *launches_out = launches;
    return PSN_OK;
}

}  // namespace

// What a lockstep group needs between replays, kept by the group's first member from one batch
// call to the next: the table of member networks in device memory, the status words and the
// instantiated replay graph.  Valid while the same contexts, in the same order, hold the same
// uploads; anything else builds it anew.
struct GroupCache {
    std::vector<std::pair<unsigned long long, unsigned long long>> key;  // (serial, upload generation) per member
    DevGraph *d_graphs = nullptr;
    int *d_status = nullptr, *h_status = nullptr;
    // set-up and read-out of all members in one launch / one copy each: per-member records, and
    // blocks of the members' scalar words, work counters and control blocks (device + pinned)
    GroupAux *d_aux = nullptr, *h_aux = nullptr;
    i64 *d_scalars = nullptr, *h_scalars = nullptr;
    unsigned long long *d_tally = nullptr, *h_tally = nullptr;
    Ctrl *d_ctrls = nullptr, *h_ctrls = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    int launches = 0;
};

namespace {
void drop_group_cache(psn_ctx *c) {
    GroupCache *gc = c->gcache;
    if (!gc) return;
    if (gc->gexec) (void)hipGraphExecDestroy(gc->gexec);
    if (gc->graph) (void)hipGraphDestroy(gc->graph);
    if (gc->d_graphs) (void)hipFree(gc->d_graphs);
    if (gc->d_status) (void)hipFree(gc->d_status);
    if (gc->h_status) (void)hipHostFree(gc->h_status);
    if (gc->d_aux) (void)hipFree(gc->d_aux);
    if (gc->d_scalars) (void)hipFree(gc->d_scalars);
    if (gc->d_tally) (void)hipFree(gc->d_tally);
    if (gc->d_ctrls) (void)hipFree(gc->d_ctrls);
    if (gc->h_aux) (void)hipHostFree(gc->h_aux);
    if (gc->h_scalars) (void)hipHostFree(gc->h_scalars);
    if (gc->h_tally) (void)hipHostFree(gc->h_tally);
    if (gc->h_ctrls) (void)hipHostFree(gc->h_ctrls);
    delete gc;
    c->gcache = nullptr;
}

// Solves the members together; out[i] belongs to m[i].  Returns the first member's error, if any.
int solve_group(const std::vector<psn_ctx *> &m, uint32_t flags, psn_result *const *out) {
    psn_ctx *lead = m[0];
    const size_t G = m.size();
    hipStream_t s = lead->stream;
    HIPCHK(lead, hipSetDevice(lead->device));
    std::vector<i64> h_scalar(G * kScalarSlots, 0), h_cost(G, 0);
    std::vector<unsigned long long> h_tally(G * kTallyN, 0);
    std::vector<char> warm(G, 0);
    std::vector<int> status(G, PSN_OK);
    int rc = PSN_OK;
    long long replays = 0;
    float ms = 0.f;
    // On a failed call: the stream is drained first (copies into the cache's pinned blocks may still be in
    // flight), then the cache goes, then the error is reported against the group's first member.
#define GCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipStreamSynchronize(s); drop_group_cache(lead); return fail(lead, PSN_EDEVICE, #call, e_); } } while (0)
    const bool dbg_t = std::getenv("PSN_DEBUG_GROUP") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_a = now(), t_b = 0, t_c = 0, t_d = 0, t_e = 0;
    std::vector<std::pair<unsigned long long, unsigned long long>> key(G);
    for (size_t i = 0; i < G; ++i) key[i] = std::make_pair(m[i]->serial, m[i]->upload_gen);
    for (size_t i = 1; i < G; ++i) GCHK(hipStreamSynchronize(m[i]->stream));  // members' own earlier work (pricing) is done
    bool fresh = false;
    if (!lead->gcache || lead->gcache->key != key) {
        drop_group_cache(lead);
        GroupCache *gc = new (std::nothrow) GroupCache();
        if (!gc) return fail(lead, PSN_ENOMEM, "group cache");
        lead->gcache = gc;
        gc->key = key;
        GCHK(counted_malloc((void **)&gc->d_graphs, G * sizeof(DevGraph)));
        GCHK(counted_malloc((void **)&gc->d_status, 2 * G * sizeof(int)));
        GCHK(counted_host_malloc((void **)&gc->h_status, 2 * 2 * G * sizeof(int)));  // two read-back slots
        GCHK(counted_malloc((void **)&gc->d_aux, G * sizeof(GroupAux)));
        GCHK(counted_host_malloc((void **)&gc->h_aux, G * sizeof(GroupAux)));
        GCHK(counted_malloc((void **)&gc->d_scalars, G * kScalarSlots * sizeof(i64)));
        GCHK(counted_host_malloc((void **)&gc->h_scalars, G * kScalarSlots * sizeof(i64)));
        GCHK(counted_malloc((void **)&gc->d_tally, G * kTallyN * sizeof(unsigned long long)));
        GCHK(counted_host_malloc((void **)&gc->h_tally, G * kTallyN * sizeof(unsigned long long)));
        GCHK(counted_malloc((void **)&gc->d_ctrls, G * sizeof(Ctrl)));
        GCHK(counted_host_malloc((void **)&gc->h_ctrls, G * sizeof(Ctrl)));
        std::vector<DevGraph> h_graphs(G);
        for (size_t i = 0; i < G; ++i) h_graphs[i] = m[i]->g;
        GCHK(hipMemcpyAsync(gc->d_graphs, h_graphs.data(), G * sizeof(DevGraph), hipMemcpyHostToDevice, s));
        GCHK(hipStreamSynchronize(s));  // the host table dies with this block
        fresh = true;
    }
    GroupCache *gc = lead->gcache;
    for (size_t i = 0; i < G; ++i) {
        std::memset(out[i], 0, sizeof(psn_result));
        warm[i] = (flags & PSN_WARM_START) && m[i]->solved;
    }
    GCHK(hipEventRecord(lead->ev0, s));
    // set-up of all members: one launch per step for the whole group, one copy of the statistics back
    bool any_warm = false;
    long long max_m2 = 1;
    for (size_t i = 0; i < G; ++i) {
        gc->h_aux[i] = GroupAux{m[i]->d_supply, gc->d_scalars + i * kScalarSlots, gc->d_tally + i * kTallyN, gc->d_ctrls + i, warm[i] ? 1 : 0};
        any_warm = any_warm || warm[i];
        max_m2 = std::max<long long>(max_m2, m[i]->m2);
    }
    GCHK(hipMemcpyAsync(gc->d_aux, gc->h_aux, G * sizeof(GroupAux), hipMemcpyHostToDevice, s));
    GCHK(hipMemsetAsync(gc->d_scalars, 0, G * kScalarSlots * sizeof(i64), s));
    const dim3 ga((unsigned)std::max<long long>(8, std::min<long long>(256, (max_m2 + kBlock * 16 - 1) / (kBlock * 16))), (unsigned)G);
    hipLaunchKernelGGL(k_reset_group, ga, dim3(kBlock), 0, s, gc->d_graphs, gc->d_aux);
    hipLaunchKernelGGL(k_mirror_group, ga, dim3(kBlock), 0, s, gc->d_graphs, gc->d_aux);
    if (any_warm) hipLaunchKernelGGL(k_excess_from_flow_group, ga, dim3(kBlock), 0, s, gc->d_graphs, gc->d_aux);
    hipLaunchKernelGGL(k_start_stats_group, ga, dim3(kBlock), 0, s, gc->d_graphs, gc->d_aux);
    hipLaunchKernelGGL(k_sum_supply_group, dim3(16, (unsigned)G), dim3(kBlock), 0, s, gc->d_graphs, gc->d_aux);
    GCHK(hipMemcpyAsync(gc->h_scalars, gc->d_scalars, G * kScalarSlots * sizeof(i64), hipMemcpyDeviceToHost, s));
    for (size_t i = 0; i < G; ++i)
        if (m[i]->g.dbg_cnt) GCHK(hipMemsetAsync(m[i]->g.dbg_cnt, 0, sizeof(unsigned), s));
    GCHK(hipStreamSynchronize(s));
    std::memcpy(h_scalar.data(), gc->h_scalars, G * kScalarSlots * sizeof(i64));
    std::vector<int> first_shifts(G, 0);
    for (size_t i = 0; i < G; ++i) { gc->h_ctrls[i] = solve_init(m[i], warm[i], &h_scalar[i * kScalarSlots]); first_shifts[i] = gc->h_ctrls[i].eps_shift; }
    GCHK(hipMemcpyAsync(gc->d_ctrls, gc->h_ctrls, G * sizeof(Ctrl), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_ctrl_scatter_group, dim3((unsigned)((G + 63) / 64)), dim3(64), 0, s, gc->d_graphs, gc->d_ctrls, (int)G);
    t_b = now();
    if (fresh) {
        GCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rce = group_enqueue_replay(m, gc->d_graphs, gc->d_status, s, &gc->launches);
        const hipError_t e_launch = hipGetLastError();  // a launch the capture refused
        const hipError_t e_end = hipStreamEndCapture(s, &gc->graph);  // always: the stream must leave capture mode before anything else runs on it
        if (rce != PSN_OK || e_launch != hipSuccess) GCHK(e_launch != hipSuccess ? e_launch : hipErrorUnknown);
        GCHK(e_end);
        GCHK(hipGraphInstantiate(&gc->gexec, gc->graph, nullptr, nullptr, 0));
    }
    t_c = now();
    // replays are kept two deep in the stream (see psn_solve); status words are read back alternately
    int *h_status = gc->h_status;
    const long long max_replays = 200000;
    long long issued = 0, checked = 0;
    auto issue = [&]() -> hipError_t {
        const int slot = (int)(issued & 1);
        hipError_t e = hipGraphLaunch(gc->gexec, s);
        if (e == hipSuccess) e = hipMemcpyAsync(gc->h_status + (size_t)slot * 2 * G, gc->d_status, 2 * G * sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(lead->ev_rep[slot], s);
        ++issued;
        return e;
    };
    GCHK(issue());
    GCHK(issue());
    for (;;) {
        h_status = gc->h_status + (size_t)(checked & 1) * 2 * G;
        GCHK(hipEventSynchronize(lead->ev_rep[checked & 1]));
        ++checked;
        bool all = true;
        for (size_t i = 0; i < G; ++i) all = all && (h_status[2 * i] || h_status[2 * i + 1]);
        if (all) break;
        if (checked >= max_replays) {
            for (size_t i = 0; i < G; ++i) if (!h_status[2 * i] && !h_status[2 * i + 1]) status[i] = PSN_ELIMIT;
            break;
        }
        GCHK(issue());
    }
    replays = issued;
    t_d = now();
    for (size_t i = 0; i < G; ++i)
        if (h_status[2 * i + 1]) status[i] = h_status[2 * i + 1] == PSN_EINFEASIBLE_DEV ? PSN_EINFEASIBLE : PSN_EDEVICE;
    // read-out of all members: objective (scalar word 0 of each), work counters, control blocks
    GCHK(hipMemsetAsync(gc->d_scalars, 0, G * kScalarSlots * sizeof(i64), s));
    hipLaunchKernelGGL(k_flip_group, ga, dim3(kBlock), 0, s, gc->d_graphs, 1);
    hipLaunchKernelGGL(k_total_cost_group, ga, dim3(kBlock), 0, s, gc->d_graphs, gc->d_aux);
    hipLaunchKernelGGL(k_tally_sum_group, dim3(1, (unsigned)G), dim3(64 * kTallyN), 0, s, gc->d_graphs, gc->d_aux, kSweepWaves);
    GCHK(hipMemcpyAsync(gc->h_scalars, gc->d_scalars, G * kScalarSlots * sizeof(i64), hipMemcpyDeviceToHost, s));
    GCHK(hipMemcpyAsync(gc->h_tally, gc->d_tally, G * kTallyN * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    GCHK(hipMemcpyAsync(gc->h_ctrls, gc->d_ctrls, G * sizeof(Ctrl), hipMemcpyDeviceToHost, s));
    GCHK(hipEventRecord(lead->ev1, s));
    GCHK(hipStreamSynchronize(s));
    for (size_t i = 0; i < G; ++i) {
        h_cost[i] = gc->h_scalars[i * kScalarSlots];
        std::memcpy(&h_tally[i * kTallyN], gc->h_tally + i * kTallyN, kTallyN * sizeof(unsigned long long));
        *m[i]->h_ctrl = gc->h_ctrls[i];
    }
    GCHK(hipEventElapsedTime(&ms, lead->ev0, lead->ev1));
#undef GCHK
    t_e = now();
    if (dbg_t) fprintf(stderr, "group of %zu (%s): prepare %.2f capture+instantiate %.2f replays(%lld) %.2f finish %.2f ms\n", G, fresh ? "new" : "cached", t_b - t_a, t_c - t_b, replays, t_d - t_c, t_e - t_d);
    rc = PSN_OK;
    for (size_t i = 0; i < G; ++i) {
        solve_report(m[i], out[i], status[i], h_cost[i], &h_tally[i * kTallyN], replays, gc->launches, ms, warm[i] != 0, first_shifts[i]);
        if (status[i] != PSN_OK) {
            (void)fail(m[i], status[i], h_scalar[i * kScalarSlots + kSupplySumSlot] != 0 ? "supplies do not sum to zero" : "solve did not converge");
            if (rc == PSN_OK) rc = status[i];
        }
    }
    return rc;
}
}  // namespace

int psn_solve_batch(psn_ctx *const *ctxs, int32_t count, uint32_t flags, int32_t workers, psn_result *out) {
    if (count < 0 || (count > 0 && (!ctxs || !out))) return PSN_EINVAL;
    for (int32_t i = 0; i < count; ++i) if (!ctxs[i]) return PSN_EINVAL;
    for (int32_t i = 0; i < count; ++i)
        for (int32_t j = i + 1; j < count; ++j)
            if (ctxs[i] == ctxs[j]) return PSN_EINVAL;  // a context runs one solve at a time
    for (int32_t i = 0; i < count; ++i) if (!ctxs[i]->uploaded) return fail(ctxs[i], PSN_ESTATE, "no graph uploaded");
    // one host thread drives one group at a time; with contexts on several devices every device
    // should have a thread of its own so that the devices work side by side
    const bool workers_default = workers <= 0;
    int32_t n_devices = 1;
    {
        std::vector<int> devs;
        for (int32_t i = 0; i < count; ++i) if (std::find(devs.begin(), devs.end(), ctxs[i]->device) == devs.end()) devs.push_back(ctxs[i]->device);
        n_devices = std::max<int32_t>(1, (int32_t)devs.size());
        if (workers <= 0) workers = std::max<int32_t>(4, 2 * n_devices);
        workers = std::max<int32_t>(workers, n_devices);
    }
    // Contexts that can share a replay sequence (same device, same solver configuration, graph
    // replay on) are solved in lockstep groups; the groups are spread over the worker threads.
    int group_cap = 128;
    if (const char *ev = std::getenv("PSN_BATCH_GROUP")) group_cap = std::max(1, std::min(kMaxGroup, atoi(ev)));
    std::vector<std::vector<int32_t>> groups;
    size_t n_lockstep = 0;
    {
        std::vector<char> taken((size_t)count, 0);
        // even group sizes: ceil(eligible / ceil(eligible / cap)) members each
        for (int32_t i = 0; i < count; ++i) {
            if (taken[(size_t)i]) continue;
            std::vector<int32_t> same{i};
            taken[(size_t)i] = 1;
            const psn_ctx *a = ctxs[i];
            const bool groupable = !(a->cfg.flags & (PSN_CFG_EAGER | PSN_CFG_KTIME)) && group_cap > 1;
            for (int32_t j = i + 1; groupable && j < count; ++j) {
                const psn_ctx *b = ctxs[j];
                if (!taken[(size_t)j] && b->device == a->device && b->cfg.flags == a->cfg.flags && b->cfg.alpha_log2 == a->cfg.alpha_log2 &&
                    b->rounds == a->rounds) { same.push_back(j); taken[(size_t)j] = 1; }
            }
            // A lockstep group pays for itself from about four members on: its launches carry every
            // member's idle stages, and two or three networks finish sooner one after the other.
            const size_t min_group = (size_t)std::min(4, group_cap);
            // Two groups side by side on a device fill the gaps in each other's replays (measured: 7-13 %
            // less time than one group of all, from 8 networks to 128); a third or fourth costs more than
            // it fills, so the rest of a larger batch waits its turn (see the worker count below).
            size_t parts = (same.size() + (size_t)group_cap - 1) / (size_t)group_cap;
            if (parts == 1 && same.size() >= 2 * min_group && !std::getenv("PSN_BATCH_GROUP")) parts = 2;
            const size_t per = (same.size() + parts - 1) / parts;
            for (size_t at = 0; at < same.size(); at += per) {
                const size_t end = std::min(same.size(), at + per);
                if (end - at >= min_group) groups.emplace_back(same.begin() + (long)at, same.begin() + (long)end);
                else for (size_t x = at; x < end; ++x) groups.emplace_back(1, same[x]);
            }
        }
        // groups of different devices take turns in the work queue, so that the first `workers` groups
        // picked up already cover every device
        std::stable_sort(groups.begin(), groups.end(), [](const std::vector<int32_t> &a, const std::vector<int32_t> &b) { return a.size() > b.size(); });
        std::vector<std::vector<int32_t>> by_turn;
        std::vector<char> used(groups.size(), 0);
        for (size_t left = groups.size(); left;) {
            std::vector<int> seen;
            for (size_t gi = 0; gi < groups.size(); ++gi) {
                if (used[gi]) continue;
                const int dev = ctxs[groups[gi][0]]->device;
                if (std::find(seen.begin(), seen.end(), dev) != seen.end()) continue;
                seen.push_back(dev); used[gi] = 1; --left;
                by_turn.push_back(std::move(groups[gi]));
            }
        }
        groups.swap(by_turn);
    }
    workers = std::min<int32_t>(workers, std::max<int32_t>((int32_t)groups.size(), 1));
    for (const auto &gr : groups) n_lockstep += gr.size() > 1;
    if (workers_default && n_lockstep == groups.size()) workers = std::min<int32_t>(workers, 2 * n_devices);  // two groups at a time per device
    std::atomic<size_t> next{0};
    std::atomic<int> first_error{PSN_OK};
    auto work = [&]() {
        for (;;) {
            const size_t gi = next.fetch_add(1);
            if (gi >= groups.size()) return;
            const std::vector<int32_t> &ids = groups[gi];
            int rc;
            if (ids.size() == 1) {
                rc = psn_solve(ctxs[ids[0]], flags, &out[ids[0]]);
            } else {
                std::vector<psn_ctx *> m;
                std::vector<psn_result *> o;
                for (int32_t id : ids) { m.push_back(ctxs[id]); o.push_back(&out[id]); }
                rc = solve_group(m, flags, o.data());
            }
            if (rc != PSN_OK) { int expected = PSN_OK; first_error.compare_exchange_strong(expected, rc); }
        }
    };
    std::vector<std::thread> pool;
    for (int32_t w = 1; w < workers; ++w) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return first_error.load();
}

int psn_get_flow(psn_ctx *c, int32_t *flow) {
    if (!c || !flow) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    int *d;
    HIPCHK(c, hipMalloc((void **)&d, std::max<size_t>(1, (size_t)c->m2) * sizeof(int)));
    hipLaunchKernelGGL(k_flow, dim3(kGrid), dim3(kBlock), 0, c->stream, c->g, d);
    hipError_t e = hipMemcpyAsync(flow, d, (size_t)c->m2 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, PSN_EDEVICE, "psn_get_flow", e);
    return PSN_OK;
}

int psn_get_residual(psn_ctx *c, int32_t *cap) {
    if (!c || !cap) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(cap, c->g.cap, (size_t)c->m2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSN_OK;
}

int psn_get_prices(psn_ctx *c, int64_t *price) {
    if (!c || !price) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(price, c->g.price, (size_t)c->n * sizeof(i64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSN_OK;
}

int psn_price_scale(psn_ctx *c, int64_t *scale) {
    if (!c || !scale) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    *scale = c->g.scale;
    return PSN_OK;
}

int psn_get_structure(psn_ctx *c, int32_t *head, int32_t *rev) {
    if (!c || (!head && !rev)) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (head) HIPCHK(c, hipMemcpyAsync(head, c->d_head, (size_t)c->m2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (rev) HIPCHK(c, hipMemcpyAsync(rev, c->d_rev, (size_t)c->m2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSN_OK;
}

int psn_get_rows(psn_ctx *c, int64_t *row_off, int64_t *supply) {
    if (!c || (!row_off && !supply)) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    if (row_off) for (size_t v = 0; v < c->h_row.size(); ++v) row_off[v] = c->h_row[v];
    if (supply) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipMemcpyAsync(supply, c->d_supply, (size_t)c->n * sizeof(i64), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return PSN_OK;
}

int psn_get_costs(psn_ctx *c, int32_t *cost, int32_t *cap_orig) {
    if (!c || (!cost && !cap_orig)) return fail(c, PSN_EINVAL, "null argument");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (cost) HIPCHK(c, hipMemcpyAsync(cost, c->d_cost, (size_t)c->m2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (cap_orig) HIPCHK(c, hipMemcpyAsync(cap_orig, c->d_ocap, (size_t)c->m2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PSN_OK;
}

constexpr int kPlaceBadSlot = 39;  // d_scalar word that counts arcs k_placements found out of place
// Host-side check of a layout against the uploaded network, before any kernel indexes with it.
static int check_place_layout(psn_ctx *c, const psn_layout *lay) {
    if (!lay) return fail(c, PSN_EINVAL, "null layout");
    if (!c->uploaded) return fail(c, PSN_ESTATE, "no graph uploaded");
    const int N = lay->n_machines, E = lay->n_classes, J = lay->n_jobs, P = lay->n_tasks, K = lay->k_slots;
    if (N <= 0 || E <= 0 || J < 0 || P <= 0 || K <= 0 || (long long)1 + N + E + J + P != c->n)
        return fail(c, PSN_EINVAL, "layout does not match the uploaded network");
    // every class row must end with its N * K slot arcs (the kernel splits the row there)
    for (int e = 0; e < E; ++e)
        if ((long long)c->h_row[(size_t)2 + N + e] - c->h_row[(size_t)1 + N + e] < (long long)N * K)
            return fail(c, PSN_EINVAL, "a class row is shorter than machines x slots: layout does not match the uploaded network");
    return PSN_OK;
}
// K5 on the context's stream: placements into dst (device, int32[P]); slot_machine (device,
// int32[E * stride]) holds each class's units of flow in arc order while the kernel runs.  The
// count of arcs found out of place is on its way to h_flag[1] when this returns.
static int launch_placements(psn_ctx *c, const psn_layout *lay, int *dst, int *slot_machine, long long stride) {
    const int N = lay->n_machines, E = lay->n_classes, J = lay->n_jobs, P = lay->n_tasks, K = lay->k_slots;
    unsigned *d_bad = reinterpret_cast<unsigned *>(c->d_scalar + kPlaceBadSlot);
    HIPCHK(c, hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_fill_i32, dim3(std::max(1, std::min(256, (P + kBlock - 1) / kBlock))), dim3(kBlock), 0, c->stream, dst, -1, (long long)P);
    hipLaunchKernelGGL(k_placements, dim3(E), dim3(kPlaceBlock), 0, c->stream, c->g, N, E, K, 1 + N + E + J, P, slot_machine, (int)stride, dst, d_bad);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_flag + 1, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    return PSN_OK;
}
static int finish_placements(psn_ctx *c) {  // after the stream has drained
    if (c->h_flag[1]) return fail(c, PSN_EINVAL, "slot arcs or task arcs are not where the layout puts them");
    return PSN_OK;
}
static int run_placements(psn_ctx *c, const psn_layout *lay) {
    int rc = check_place_layout(c, lay);
    if (rc != PSN_OK) return rc;
    const int E = lay->n_classes, P = lay->n_tasks;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->place_count != P || c->slot_stride != P || c->slot_classes != E) {
        if ((rc = dalloc(c, &c->d_place, (size_t)P)) != PSN_OK) return rc;
        if ((rc = dalloc(c, &c->d_slot_machine, (size_t)E * (size_t)P)) != PSN_OK) return rc;
        c->place_count = P; c->slot_stride = P; c->slot_classes = E;
    }
    return launch_placements(c, lay, c->d_place, c->d_slot_machine, P);
}

int psn_get_placements(psn_ctx *c, const psn_layout *lay, int32_t *task_to_machine) {
    if (!c || !task_to_machine) return fail(c, PSN_EINVAL, "null argument");
    int rc = run_placements(c, lay);
    if (rc != PSN_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(task_to_machine, c->d_place, (size_t)lay->n_tasks * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return finish_placements(c);
}

// Development aid (PSN_TRACE=1 in the environment at upload time): one record per sweep
// launch that had work: kind (1 push, 2 global-update expand), k, lane-list size, item-list size.
int psn_debug_trace(psn_ctx *c, int32_t *out4, int64_t max_records, int64_t *count) {
    if (!c || !out4 || !count) return fail(c, PSN_EINVAL, "null argument");
    *count = 0;
    if (!c->uploaded || !c->g.dbg) return PSN_OK;
    unsigned n = 0;
    HIPCHK(c, hipMemcpy(&n, c->g.dbg_cnt, sizeof(unsigned), hipMemcpyDeviceToHost));
    n = std::min<unsigned>(n, kDbgCap);
    n = (unsigned)std::min<int64_t>(n, max_records);
    HIPCHK(c, hipMemcpy(out4, c->g.dbg, (size_t)n * sizeof(int4), hipMemcpyDeviceToHost));
    *count = n;
    return PSN_OK;
}


int psn_kernel_times(psn_ctx *c, double *ms, uint64_t *calls) {
    if (!c || !ms || !calls) return fail(c, PSN_EINVAL, "null argument");
    for (int i = 0; i < PSN_KT_CLASSES; ++i) { ms[i] = c->kt_ms[i]; calls[i] = c->kt_calls[i]; }
    return PSN_OK;
}

int psn_get_placements_device(psn_ctx *c, const psn_layout *lay, void *dst_device) {
    if (!c || !dst_device) return fail(c, PSN_EINVAL, "null argument");
    int rc = run_placements(c, lay);
    if (rc != PSN_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(dst_device, c->d_place, (size_t)lay->n_tasks * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return finish_placements(c);
}

int psn_gather_placements(psn_ctx *const *ctxs, int32_t count, const psn_layout *layouts, int64_t stride, int32_t dst_device,
                          void *dst) {
    if (count < 0 || stride < 0 || (count > 0 && (!ctxs || !layouts || !dst))) return PSN_EINVAL;
    for (int32_t i = 0; i < count; ++i) {
        if (!ctxs[i]) return PSN_EINVAL;
        if (layouts[i].n_tasks > stride) return fail(ctxs[i], PSN_EINVAL, "stride is smaller than the number of tasks");
    }
    int ndev = 0;
    if (dst_device >= 0 && (hipGetDeviceCount(&ndev) != hipSuccess || dst_device >= ndev)) return PSN_EINVAL;
    // extraction kernels of all contexts first (each on its own stream and device), then the copies
    for (int32_t i = 0; i < count; ++i) {
        const int rc = run_placements(ctxs[i], &layouts[i]);
        if (rc != PSN_OK) return rc;
    }
    int first = PSN_OK;
    for (int32_t i = 0; i < count; ++i) {
        psn_ctx *c = ctxs[i];
        const size_t bytes = (size_t)layouts[i].n_tasks * sizeof(int);
        int32_t *row = static_cast<int32_t *>(dst) + (size_t)i * (size_t)stride;
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) {
            if (dst_device < 0) e = hipMemcpyAsync(row, c->d_place, bytes, hipMemcpyDeviceToHost, c->stream);
            else if (dst_device == c->device) e = hipMemcpyAsync(row, c->d_place, bytes, hipMemcpyDeviceToDevice, c->stream);
            else e = hipMemcpyPeerAsync(row, dst_device, c->d_place, c->device, bytes, c->stream);
        }
        if (e != hipSuccess && first == PSN_OK) first = fail(c, PSN_EDEVICE, "psn_gather_placements: copy", e);
    }
    for (int32_t i = 0; i < count; ++i) {
        psn_ctx *c = ctxs[i];
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess && first == PSN_OK) first = fail(c, PSN_EDEVICE, "psn_gather_placements: synchronize", e);
        if (e == hipSuccess && first == PSN_OK) first = finish_placements(c);
    }
    return first;
}

uint64_t psn_debug_alloc_count(void) { return g_alloc_calls.load(); }

int psn_get_placements_device_noalloc(psn_ctx *c, const psn_layout *lay, void *dst_device) {
    if (!c || !dst_device) return fail(c, PSN_EINVAL, "null argument");
    int rc = check_place_layout(c, lay);
    if (rc != PSN_OK) return rc;
    if ((size_t)lay->n_classes * (size_t)lay->n_tasks > kNoAllocInts)
        return fail(c, PSN_EINVAL, "classes x tasks exceeds the scratch of the no-allocation entry point (1048576 entries)");
    HIPCHK(c, hipSetDevice(c->device));
    int *ptr = c->device < kMaxDevices ? g_scratch_ptr[c->device] : nullptr;
    if (!ptr) return fail(c, PSN_ESTATE, "scratch address of this device was not resolved");
    // the unit lists of the classes live in the per-device scratch while the kernel runs; nothing in it
    // needs to start at zero (every entry that is read has been written by the same launch)
    if ((rc = launch_placements(c, lay, static_cast<int *>(dst_device), ptr, lay->n_tasks)) != PSN_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return finish_placements(c);
}

int psn_placements_device_ptr(psn_ctx *c, void **dptr, int64_t *count) {
    if (!c || !dptr || !count) return fail(c, PSN_EINVAL, "null argument");
    if (!c->d_place) return fail(c, PSN_ESTATE, "no placements computed yet");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *dptr = c->d_place;
    *count = c->place_count;
    return PSN_OK;
}

#ifdef PSN_UBENCH
// Development builds only (-DPSN_UBENCH): the frontier expansions of one global update, seeded at
// `sink`, on the state a finished solve left, `reps` times over - one launch per expansion, to be
// timed expansion by expansion from a kernel trace (tools/ubench/gu_bench.py).  The context's
// solution is destroyed.
namespace psn {
__global__ void k_ub_seed(DevGraph g, int sink, int eps_shift) {
    Ctrl *ct = g.ctrl;
    ct->done = ct->error = 0;
    ct->gu_on = ct->gu_init_on = ct->gu_active = 1;
    ct->gu_stamp_base = 0;
    ct->eps_shift = eps_shift;
    ct->eps = (i64)1 << eps_shift;
    ct->epoch++;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) ct->fr_cnt[i][j] = 0;
    ct->gu_phase[0] = ct->gu_phase[1] = 0;
    g.excess[sink] = -1;
}
}  // namespace psn
int psn_debug_gu_bench(psn_ctx *c, int reps, int eps_shift, int sink, int n_expansions) {
    if (!c || !c->uploaded) return PSN_ESTATE;
    HIPCHK(c, hipSetDevice(c->device));
    const DevGraph &g = c->g;
    const int gv = std::max(1, std::min(kGrid, (g.n + kBlock * vpt_of(g.n) - 1) / (kBlock * vpt_of(g.n))));
    const int gs = (int)std::max<long long>(32, std::min<long long>(kGrid, (c->m2 + kBlock * 16 - 1) / (kBlock * 16)));
    hipStream_t s = c->stream;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(psn::k_ub_seed, dim3(1), dim3(1), 0, s, g, sink, eps_shift);
        hipLaunchKernelGGL(k_gu_init, dim3(gv), dim3(kBlock), 0, s, g);
        for (int k = 0; k < n_expansions; ++k) hipLaunchKernelGGL(k_gu_expand, dim3(gs), dim3(kBlock), 0, s, g, k);
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return PSN_OK;
}
#endif

}  // extern "C"
