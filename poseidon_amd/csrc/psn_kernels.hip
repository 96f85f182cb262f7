// libpsn device code: cost-scaling push-relabel min-cost flow for gfx950 (MI355X).
//
// Everything here is wave-centric (64 lanes): a work item is a vertex of degree <= 4
// (one lane) or a segment of up to 1024 arcs of a larger vertex (one wave, 8 arcs per
// lane per step).  The sweep kernels need no workgroup barrier.
//
// The algorithm is bulk-synchronous so that the result does not depend on timing:
//   push phase     prices frozen; every active vertex pushes on admissible arcs
//                  (reduced cost < 0).  A residual capacity is written only by the
//                  wave that owns the arc's tail (an arc and its reverse cannot both be
//                  admissible under one price vector), excess moves by 64-bit atomics.
//   relabel phase  residual graph frozen; an active vertex whose scan found no
//                  admissible residual arc gets price = max(price[v] - cost) - eps.
// Epsilon is a power of two; costs are multiplied by S = 2 |C| + 1 for a vertex cover C of the
// network (a simple residual cycle has at most 2 |C| arcs), so a flow that is 1-optimal is optimal.
// All loop control (epsilon schedule, when to saturate, when to run a global price
// update, termination) lives in a device-resident Ctrl block that tiny single-thread
// kernels advance, so one hipGraph can be replayed until Ctrl.done is set.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "psn_device.h"

namespace psn {

static constexpr i64 kInf = INT64_MAX;
static constexpr i64 kNegInf = INT64_MIN;
// Global-update distances are stored packed: (distance << 16) | stamp of the sweep that last
// lowered it, so that one atomic min both lowers the distance and tells whether the vertex
// was already queued during this sweep.  Arc lengths are cut at kLenMax: shortest distances
// under shorter lengths still satisfy d(u) <= d(v) + length(u,v) for the true lengths, which
// is all the price update needs; with at most 2^28 vertices distances stay below 2^46.
static constexpr i64 kDistInf = (i64)1 << 46;
#ifndef PSN_LENMAX_LOG2
#define PSN_LENMAX_LOG2 18
#endif
static constexpr i64 kLenMax = (i64)1 << PSN_LENMAX_LOG2;
__device__ __forceinline__ i64 dist_pack(i64 d, unsigned stamp16) { return (d << 16) | (i64)stamp16; }
__device__ __forceinline__ i64 dist_of(i64 packed) { return packed >> 16; }

typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));

// Predicated loads without control flow.  `need ? *p : dflt` written as a branch around the load
// lets the compiler sink whatever consumes the value into the branch, and the wait for the load
// with it: every such load then becomes a memory round trip of its own (nine in a row for the
// per-arc lookups of a lane).  Selecting the address instead keeps the loads unconditional: a
// lane that does not need the value reads a spare word (one shared cache line per wave) and the
// result is selected after all loads of the step have been issued.
template <typename T>
__device__ __forceinline__ T ld_sel(bool need, const T *p, const void *spare) {
    return *(need ? p : static_cast<const T *>(spare));
}

// ------------------------------------------------------------------ wave primitives
// Control words a kernel reads at its start are asked for together and waited for once: read one
// after the other behind the branches that look at them, each costs a round trip to memory of its
// own (the launch before wrote them), which is most of the duration of a launch with little work.
#define CTRL_TOGETHER4(a, b, c, d) asm volatile("" ::"s"(a), "s"(b), "s"(c), "s"(d))
#define CTRL_TOGETHER5(a, b, c, d, e) asm volatile("" ::"s"(a), "s"(b), "s"(c), "s"(d), "s"(e))
#define CTRL_TOGETHER8(a, b, c, d, e, f, g_, h) asm volatile("" ::"s"(a), "s"(b), "s"(c), "s"(d), "s"(e), "s"(f), "s"(g_), "s"(h))
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ i64 wave_incl_scan(i64 x) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        i64 y = __shfl_up(x, d);
        if (l >= d) x += y;
    }
    return x;
}
__device__ __forceinline__ i64 wave_max(i64 x) {
#pragma unroll
    for (int d = 32; d; d >>= 1) { i64 y = __shfl_xor(x, d); x = y > x ? y : x; }
    return x;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long x) {
#pragma unroll
    for (int d = 32; d; d >>= 1) x += __shfl_xor(x, d);
    return x;
}
__device__ __forceinline__ i64 amax64(i64 *p, i64 v) { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ i64 amin64(i64 *p, i64 v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ i64 aadd64(i64 *p, i64 v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned aadd32(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned axchg32(unsigned *p, unsigned v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned mix32(unsigned a, unsigned b) {
    unsigned long long z = ((unsigned long long)a << 32 | b) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    return (unsigned)z;
}

// Per-wave work counters: 32-bit while a sweep runs (one wave scans far fewer than 2^32 arcs in
// one launch), added to the wave's 64-bit slot in memory when it ends (no atomics: wave w of block
// b owns slot (b, w) in every launch, and launches on one stream are ordered).  A wave that
// counted nothing skips the fold.
struct Tally { unsigned c[kTallyN]; __device__ Tally() { for (int i = 0; i < kTallyN; ++i) c[i] = 0; } };
__device__ __forceinline__ unsigned wave_sum_u32(unsigned x) {
#pragma unroll
    for (int d = 32; d; d >>= 1) x += __shfl_xor(x, d);
    return x;
}
template <int... Idx>
__device__ __forceinline__ void tally_flush(const DevGraph &g, Tally &t) {
    unsigned any = 0;
    ((any |= t.c[Idx]), ...);
    if (!__ballot(any != 0)) return;
    ((t.c[Idx] = wave_sum_u32(t.c[Idx])), ...);
    if (lane_id() == 0) {
        unsigned long long *slot = g.tally + ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * kTallyN;
        ((slot[Idx] += t.c[Idx]), ...);
    }
}

// ------------------------------------------------------------------------ work items
// Lane item  (degree <= kLowMax): int2 { v | degree << 28, first arc }.
// Wave item  (everything else):   int4 { v, first arc, end arc, flags } covering at most
// kWSeg arcs; flags bit 0 = the item is the whole vertex ("solo"), bit 1 (solo items of the
// global-update frontier) = the row is made of aligned groups of 8 parallel arcs.  Items carry their arc
// range so that a sweep does not pay a dependent row-offset load before it can stream.
__device__ __forceinline__ int2 low_item(int v, unsigned r0, unsigned r1) { return make_int2(v | (int)((r1 - r0) << 28), (int)r0); }
__device__ __forceinline__ int4 solo_item(int v, unsigned r0, unsigned r1, bool lean = false) { return make_int4(v, (int)r0, (int)r1, lean ? 3 : 1); }
// A vertex larger than one item: its segments, starting at a per-round rotated one so that
// vertices with common neighbours do not all begin at the same arcs.
// Bits 1..23 of the flags = 1 + index of the segment's bound slots when the vertex is a hub (seg1 = 1 +
// index of its first segment), else 0; bits 24..30 = the hub's slot (work lists of the push rounds only).
constexpr int kItemHubShift = 24, kItemSegMask = (1 << 23) - 1;
static_assert(kMaxHubs <= 128, "the hub slot takes 7 bits of an item's flags");
__device__ __forceinline__ void append_split(unsigned *cnt, int4 *segl, int v, unsigned r0, unsigned r1, unsigned rot, int seg1 = 0) {
    const unsigned nseg = (r1 - r0 + kWSeg - 1) / kWSeg;
    const unsigned base = aadd32(&cnt[kSlotSeg], nseg);
    unsigned sg = rot % nseg;
    for (unsigned s = 0; s < nseg; ++s) {
        const unsigned a = r0 + sg * kWSeg;
        segl[base + s] = make_int4(v, (int)a, (int)min(r1, a + kWSeg), seg1 ? (seg1 + (int)sg) << 1 : 0);
        if (++sg == nseg) sg = 0;
    }
}

__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned x) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned y = __shfl_up(x, d); if (l >= d) x += y; }
    return x;
}

// ------------------------------------------------------- active-list compaction (K4)
// Vertex sweep helper: each thread brings VPT vertices (take[i], class, arc range); lane
// and wave items are numbered by a popcount-style prefix inside the wave, waves are
// chained through LDS, and the workgroup takes its slots with one atomic per list.
// Split vertices (hubs) are staged in LDS first: one thread writing the ~70 segments of a
// task class kept its whole workgroup alive three times longer than the rest of the sweep,
// so the workgroup takes the slots for all staged vertices with one atomic and writes the
// segments with all its threads.
constexpr int kSplitCap = 128;  // split vertices staged per call; any beyond are written by their own thread
struct AppendLds {
    unsigned sums[2][kBlock / 64 + 1];
    int v[kSplitCap];
    unsigned r0[kSplitCap], r1[kSplitCap], rot[kSplitCap], pref[kSplitCap + 1];
    unsigned n, nsegs;
};
// Items written here carry no bound slots: the relabel sweep handles the hubs, whose segments
// do, outside this helper (one workgroup per hub).
template <int VPT>
__device__ __forceinline__ void block_append(unsigned *cnt, int2 *lowl, int4 *segl, const bool (&take)[VPT], const int (&v)[VPT],
                                             const int (&cls)[VPT], const unsigned (&r0)[VPT], const unsigned (&r1)[VPT],
                                             unsigned rot, AppendLds &lds, unsigned extra_segs = 0, unsigned *extra_base = nullptr) {
    // extra_segs (workgroup-uniform): further wave-item slots the caller fills itself, taken with
    // the same atomic; *extra_base receives the first of them.
    unsigned nl = 0, ns = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) { nl += take[i] && (cls[i] & kClsMask) == kClsLow; ns += take[i] && (cls[i] & kClsMask) == kClsMed; }
    const unsigned il = wave_incl_scan_u32(nl), is = wave_incl_scan_u32(ns);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 63) { lds.sums[0][w] = il; lds.sums[1][w] = is; }
    if (threadIdx.x == 0) { lds.n = 0; lds.nsegs = 0; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPT; ++i) {  // stage the split vertices
        if (take[i] && (cls[i] & kClsMask) == kClsHigh) {
            const unsigned at = __hip_atomic_fetch_add(&lds.n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned vrot = rot ^ (unsigned)v[i] * 0x9E3779B9u;
            if (at < (unsigned)kSplitCap) {
                lds.v[at] = v[i]; lds.r0[at] = r0[i]; lds.r1[at] = r1[i]; lds.rot[at] = vrot;
                __hip_atomic_fetch_add(&lds.nsegs, (r1[i] - r0[i] + kWSeg - 1) / kWSeg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                append_split(cnt, segl, v[i], r0[i], r1[i], vrot);
            }
        }
    }
    __syncthreads();
    // One atomic per list takes the slots of the whole workgroup: lane items; wave items of the
    // unsplit vertices followed by the segments of the staged ones.
    const unsigned nsplit = min(lds.n, (unsigned)kSplitCap), nsplit_segs = lds.nsegs;  // workgroup-uniform
    unsigned my_base = 0, my_tot = 0;
    if (threadIdx.x < 2) {
        for (int i = 0; i < kBlock / 64; ++i) my_tot += lds.sums[threadIdx.x][i];
        const unsigned want = my_tot + (threadIdx.x == 1 ? nsplit_segs + extra_segs : 0u);
        if (want) my_base = aadd32(&cnt[threadIdx.x == 0 ? kSlotLow : kSlotSeg], want);
    }
    if (nsplit && threadIdx.x < 64) {  // meanwhile: exclusive prefix of the staged vertices' segment counts
        unsigned run = 0;
        for (unsigned b = 0; b < nsplit; b += 64) {
            const unsigned i = b + threadIdx.x;
            const unsigned c = i < nsplit ? (lds.r1[i] - lds.r0[i] + kWSeg - 1) / kWSeg : 0u;
            const unsigned incl = wave_incl_scan_u32(c);
            if (i < nsplit) lds.pref[i] = run + incl - c;
            run += __shfl(incl, 63);
        }
        if (threadIdx.x == 0) lds.pref[nsplit] = run;
    }
    if (threadIdx.x < 2) lds.sums[threadIdx.x][kBlock / 64] = my_base;
    __syncthreads();
    unsigned pl = lds.sums[0][kBlock / 64] + il - nl, ps = lds.sums[1][kBlock / 64] + is - ns;
    unsigned med_total = 0;
    for (int i = 0; i < kBlock / 64; ++i) med_total += lds.sums[1][i];
    for (int i = 0; i < w; ++i) { pl += lds.sums[0][i]; ps += lds.sums[1][i]; }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        if (!take[i]) continue;
        if ((cls[i] & kClsMask) == kClsLow) lowl[pl++] = low_item(v[i], r0[i], r1[i]);
        else if ((cls[i] & kClsMask) == kClsMed) segl[ps++] = solo_item(v[i], r0[i], r1[i], (cls[i] & kClsLean) != 0);
    }
    if (extra_base) *extra_base = lds.sums[1][kBlock / 64] + med_total + nsplit_segs;
    if (nsplit) {
        // Wave w writes the staged vertices w, w + 4, ...: lane l first fetches the record of the
        // l-th of them, the loop then hands one record at a time to the whole wave (lane reads, no
        // LDS round trip per vertex) and the lanes write its segments side by side.
        const unsigned base = lds.sums[1][kBlock / 64] + med_total;
        const int l = lane_id();
        const unsigned mine = (unsigned)w + (kBlock / 64) * (unsigned)l;
        const bool have = mine < nsplit;
        const int m_v = have ? lds.v[mine] : 0;
        const int m_r0 = have ? (int)lds.r0[mine] : 0, m_r1 = have ? (int)lds.r1[mine] : 0;
        const int m_rot = have ? (int)lds.rot[mine] : 0, m_pref = have ? (int)lds.pref[mine] : 0;
        const unsigned mine_cnt = nsplit > (unsigned)w ? (nsplit - (unsigned)w + kBlock / 64 - 1) / (kBlock / 64) : 0u;
        for (unsigned j = 0; j < mine_cnt; ++j) {
            const int hv = __builtin_amdgcn_readlane(m_v, j);
            const unsigned hr0 = (unsigned)__builtin_amdgcn_readlane(m_r0, j), hr1 = (unsigned)__builtin_amdgcn_readlane(m_r1, j);
            const unsigned hrot = (unsigned)__builtin_amdgcn_readlane(m_rot, j), hpref = (unsigned)__builtin_amdgcn_readlane(m_pref, j);
            const unsigned nseg = (hr1 - hr0 + kWSeg - 1) / kWSeg, first = hrot % nseg;
            for (unsigned i = (unsigned)l; i < nseg; i += 64) {
                unsigned sg = first + i;  // same rotated order as append_split
                if (sg >= nseg) sg -= nseg;
                const unsigned a = hr0 + sg * kWSeg;
                segl[base + hpref + i] = make_int4(hv, (int)a, (int)min(hr1, a + kWSeg), 0);
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ hub accumulators
// A few vertices (the sink, the task classes) receive most pushes and relaxations; their
// same-address atomics would serialise at the memory side.  Each workgroup combines what
// it sends to a hub in LDS and issues one global atomic per hub at the end of the kernel.
struct HubAcc { i64 *add; i64 *ext; i64 *out; };  // LDS, kMaxHubs entries each: sum received, and max (push) or min (distances); push rounds: sum drawn from the hub's purse
__device__ __forceinline__ void hub_init(const HubAcc &hb, i64 ext_init) {
    for (int i = threadIdx.x; i < kMaxHubs; i += kBlock) { hb.add[i] = 0; hb.ext[i] = ext_init; if (hb.out) hb.out[i] = 0; }
    __syncthreads();
}
__device__ __forceinline__ void lds_add(i64 *p, i64 v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_max(i64 *p, i64 v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_min(i64 *p, i64 v) { __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void excess_add(const DevGraph &g, const HubAcc &hb, int v, i64 d) {
    const int h = g.hub_of[v];
    if (h >= 0) lds_add(&hb.add[h], d); else aadd64(&g.excess[v], d);
}
// The vertex of hub slot threadIdx.x (kMaxHubs <= kBlock), fetched when a sweep starts so that
// the flush at its end does not wait for a lookup.
static_assert(kMaxHubs <= kBlock, "one thread per hub slot");
__device__ __forceinline__ int hub_vertex_of_thread(const DevGraph &g) {
    return (int)threadIdx.x < g.nhubs ? g.hub_vertex[threadIdx.x] : -1;
}
// end of a push/saturate kernel: one atomic per hub touched by this workgroup - on the hub's own words, or
// (PUSH_ROUND on a network with hub_shards: a relabel sweep always follows and folds them) on this
// workgroup's copy of the accumulators
template <bool PUSH_ROUND>
__device__ __forceinline__ void hub_flush_push(const DevGraph &g, const HubAcc &hb, int my_hub_vertex) {
    __syncthreads();
    if (my_hub_vertex >= 0) {
        const int i = threadIdx.x;
        const bool sharded = PUSH_ROUND && g.hub_shards != 0;
        const int at = sharded ? (int)(blockIdx.x % (unsigned)kHubShards) * kMaxHubs + i : 0;
        const i64 net = hb.add[i] - (PUSH_ROUND ? hb.out[i] : 0);
        if (net) aadd64(sharded ? &g.hub_acc_add[at] : &g.excess[my_hub_vertex], net);
        if (hb.ext[i] > kNegInf) amax64(sharded ? &g.hub_acc_ext[at] : &g.cand[my_hub_vertex], hb.ext[i]);
    }
}

// One step of a wave item covers up to kStep = 576 arcs starting at h: lane p owns the 8
// consecutive arcs h + 8p .. h + 8p + 7 (two 16-byte loads per array) and arc h + 512 + p.
// The ninth arc lets a 513-arc row (64 classes x 8 slots + the sink arc of a machine) go
// through in one step.  Slot j of a lane: arc index arc9(a0, at, j).
__device__ __forceinline__ unsigned arc9(unsigned a0, unsigned at, int j) { return j < 8 ? a0 + (unsigned)j : at; }
__device__ __forceinline__ void load9(const int *p, unsigned a0, unsigned lim8, unsigned at, unsigned limt, int (&x)[9],
                                      bool need = true, unsigned safe = 0) {
    // No control flow around the loads (see ld_sel): a lane outside the range, or one that does not
    // need the array, reads at `safe` (an index inside the arrays, the same for the whole wave).
    // A lane whose 8 arcs straddle the end of the range reads up to 7 entries past it: the arc
    // arrays are allocated with kArcPad spare entries.
    const bool in8 = need && a0 < lim8, intl = need && at < limt;
    const int *q = p + (in8 ? a0 : safe);
    const int4u lo = *reinterpret_cast<const int4u *>(q);
    const int4u hi = *reinterpret_cast<const int4u *>(q + 4);
    const int tl = p[intl ? at : safe];
    const unsigned left = in8 ? lim8 - a0 : 0u;
    x[0] = left > 0 ? lo.x : 0; x[1] = left > 1 ? lo.y : 0; x[2] = left > 2 ? lo.z : 0; x[3] = left > 3 ? lo.w : 0;
    x[4] = left > 4 ? hi.x : 0; x[5] = left > 5 ? hi.y : 0; x[6] = left > 6 ? hi.z : 0; x[7] = left > 7 ? hi.w : 0;
    x[8] = intl ? tl : 0;
}
__device__ __forceinline__ unsigned count9(unsigned a0, unsigned lim8, unsigned at, unsigned limt) {
    return (a0 < lim8 ? min(8u, lim8 - a0) : 0u) + (at < limt ? 1u : 0u);
}

// Per-vertex values for the 9 arcs of a lane.  Parallel arcs (same head) sit next to each
// other in a row, so a lane fetches once per run of equal heads among its 8 consecutive
// arcs and copies the value along; the tail arc (slot 8) is fetched on its own.
template <typename T, typename F>
__device__ __forceinline__ void gather9(const int (&hd)[9], const bool (&need)[9], T (&out)[9], T dflt, F load) {
    bool have[9], reuse[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        reuse[j] = j > 0 && j < 8 && have[j > 0 ? j - 1 : 0] && hd[j] == hd[j > 0 ? j - 1 : 0];
        have[j] = reuse[j] || need[j];
    }
    T raw[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) raw[j] = load(need[j] && !reuse[j], hd[j]);  // all loads in flight together, none behind a branch
#pragma unroll
    for (int j = 0; j < 9; ++j) out[j] = (need[j] && !reuse[j]) ? raw[j] : dflt;
#pragma unroll
    for (int j = 1; j < 8; ++j) if (reuse[j]) out[j] = out[j - 1];
}

// A push over (u,v) makes the reverse arc r = (v,u) residual; if v is a hub, the bound of the
// segment of v's row that holds r must cover r's key.
__device__ __forceinline__ void seg_note(const DevGraph &g, int hub, int r, i64 key, int par) {
    const unsigned s = (unsigned)g.hub_seg0[hub] + ((unsigned)r - g.hub_row0[hub]) / kWSeg;
    amax64(&g.seg_add[par][s], key);
}


// ------------------------------------------------------------------- orientation
// The sweeps can work on the network as it is or on its mirror image: every arc reversed, supplies
// and prices negated.  In the residual CSR the mirror needs no second copy: the reverse of arc a
// sits at rev[a] in the row of a's head, so row u of the mirror is row u of the original with
// cap and rcap (= cap[rev]) exchanged and the costs negated; price[] and excess[] hold the
// negated values while the mirror is worked on (k_flip negates them in place), and negating the
// cost scale negates every cost * scale product.  Pushing excess in the mirror is pulling flow
// towards the deficits of the original: a vertex that holds a large deficit (a task class whose
// arcs a saturation filled) spreads it over its neighbours in a round or two, where the excesses
// that look for it from the other side get through one unit per global update.
struct Orient { int *cap, *rcap; i64 S; };
__device__ __forceinline__ Orient orient_of(const DevGraph &g, int mirror) {
    Orient o;
    o.cap = mirror ? g.rcap : g.cap;
    o.rcap = mirror ? g.cap : g.rcap;
    o.S = mirror ? -g.scale : g.scale;
    return o;
}

// --------------------------------------------------------------------------- pushes
// Low-degree vertex: one lane, all loads of its (at most kLowMax) arcs issued together.
__device__ __forceinline__ void push_low(const DevGraph &g, const Orient &o, const HubAcc &hb, int2 it, i64 S, int par, Tally &t) {
    const int u = it.x & 0x0FFFFFFF;
    const unsigned deg = (unsigned)it.x >> 28, r0 = (unsigned)it.y;
    const i64 pu = g.price[u];
    const i64 e0 = g.excess[u];
    int cp[kLowMax], hd[kLowMax], cs[kLowMax];
#pragma unroll
    for (unsigned j = 0; j < kLowMax; ++j) {
        const bool ok = j < deg;
        cp[j] = ok ? o.cap[r0 + j] : 0; hd[j] = ok ? g.head[r0 + j] : 0; cs[j] = ok ? g.cost[r0 + j] : 0;
    }
    i64 key[kLowMax];
#pragma unroll
    for (unsigned j = 0; j < kLowMax; ++j) key[j] = cp[j] > 0 ? g.price[hd[j]] - (i64)cs[j] * S : kNegInf;
    t.c[kTallyVisits]++;
    t.c[kTallyScans] += deg;
    i64 e = e0, best = kNegInf;
#pragma unroll
    for (unsigned j = 0; j < kLowMax; ++j) {
        if (key[j] > pu && e > 0) {
            const i64 d = e < cp[j] ? e : cp[j];
            const unsigned a = r0 + j;
            const int r = g.rev[a];
            const int na = cp[j] - (int)d, nr = o.rcap[a] + (int)d;
            o.cap[a] = na; o.cap[r] = nr; o.rcap[a] = nr; o.rcap[r] = na;
            const int h = g.hub_of[hd[j]];
            const i64 kr = pu + (i64)cs[j] * S;
            if (h >= 0) { lds_add(&hb.add[h], d); lds_max(&hb.ext[h], kr); seg_note(g, h, r, kr, par); }
            else { aadd64(&g.excess[hd[j]], d); amax64(&g.cand[hd[j]], kr); }
            e -= d; cp[j] = na;
            t.c[kTallyPushes]++;
        }
        if (cp[j] > 0 && key[j] > best) best = key[j];
    }
    if (best > kNegInf) amax64(&g.cand[u], best);
    if (e != e0) aadd64(&g.excess[u], e - e0);
}

// One step of a solo item whose lanes found at most 64 residual arcs between them (a machine's
// row: 512 reverse arcs, of which the few that carry flow are residual, and the arc to the sink).
// The residual arcs are handed out one per lane through the wave's LDS record, so the step costs
// one arc's worth of instructions per lane instead of nine; the prefix order of the grants is
// the order of the records, which is the (rotated) lane order of the dense step.
struct StepOut { i64 here; bool stop; };
__device__ __forceinline__ StepOut push_step_sparse(const DevGraph &g, const Orient &o, const HubAcc &hb, int2 *slive, const int (&cp)[9], unsigned lm,
                                                    unsigned off, unsigned total_live, unsigned a0, unsigned at, unsigned safe, i64 pu,
                                                    i64 S, i64 &e, int par, Tally &t) {
    const int lane = lane_id();
#pragma unroll
    for (int j = 0; j < 9; ++j)
        if ((lm >> j) & 1u) { slive[off] = make_int2((int)arc9(a0, at, j), cp[j]); ++off; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool has = (unsigned)lane < total_live;
    const int2 rec = slive[has ? lane : 0];
    const unsigned a = has ? (unsigned)rec.x : safe;
    const int c = has ? rec.y : 0;
    const int hd = g.head[a], cs = g.cost[a], rv = g.rev[a], cr = o.rcap[a];  // one round trip for all four
    const i64 ph = ld_sel(has, &g.price[hd], g.spare);
    const int hub = ld_sel(has, &g.hub_of[hd], g.spare);
    const i64 key = ph - (i64)cs * S;
    const bool adm = has && key > pu;
    const i64 want = adm ? (i64)c : 0;
    int d = 0;
    bool stop = false;
    if (__ballot(want > 0)) {
        const i64 incl = wave_incl_scan(want);
        const i64 total = __shfl(incl, 63);
        const i64 granted = total < e ? total : e;
        e -= granted;
        const i64 mine = granted - (incl - want);  // what this lane may push
        if (adm && mine > 0) {
            d = (int)(mine < (i64)c ? mine : (i64)c);
            const int na = c - d, nr = cr + d;
            o.cap[a] = na; o.cap[rv] = nr; o.rcap[a] = nr; o.rcap[rv] = na;
            const i64 kr = pu + (i64)cs * S;  // key of the new residual arc (v,u)
            if (hub >= 0) { lds_add(&hb.add[hub], (i64)d); lds_max(&hb.ext[hub], kr); seg_note(g, hub, rv, kr, par); }
            else { aadd64(&g.excess[hd], (i64)d); amax64(&g.cand[hd], kr); }
            t.c[kTallyPushes]++;
        }
        stop = granted < total || e == 0;  // out of excess
    }
    // the lane's share of the relabel candidate, as the dense step works it out from its nine arcs
    StepOut so;
    so.here = !has ? kNegInf : (adm ? (c - d > 0 ? key : kNegInf) : key);
    so.stop = stop;
    __builtin_amdgcn_wave_barrier();  // the record is read before the next step or item overwrites it
    return so;
}

// One wave item.  A solo vertex keeps its excess in a register; a vertex split over
// several items has its waves claim what they want to push from excess[u] and hand back
// what was not granted.  `rot` permutes which lane owns which arcs, so that the prefix
// order (= who gets the excess first) changes from round to round.
template <bool SOLO>
__device__ __forceinline__ void push_item_t(const DevGraph &g, const Orient &o, const HubAcc &hb, int2 *slive, int4 it, i64 S, unsigned rot, int par, Tally &t) {
    const int lane = lane_id();
    const int u = it.x;
    const unsigned r0 = (unsigned)it.y, r1 = (unsigned)it.z;
    constexpr bool solo = SOLO;
    const int seg1 = SOLO ? 0 : (it.w >> 1) & kItemSegMask;  // 1 + index of this hub segment's bound slots, 0: none (a solo item has none)
    const unsigned plane = (unsigned)(lane + rot) & 63u;
    const i64 pu = g.price[u];
    // a hub's segment draws on the hub's purse and reports through the workgroup's accumulators (see DevGraph::hub_purse)
    const int hu = seg1 && g.hub_shards ? it.w >> kItemHubShift : -1;  // the hub's slot rides in the item: no lookup before the purse can be asked
    i64 *const purse = hu >= 0 ? &g.hub_purse[hu * kPurseStride] : &g.excess[u];
    int cp[9];
    if constexpr (SOLO) {
        // The only item of its vertex: listed because it holds excess, and nobody else takes excess
        // away, so nothing has to be known before the capacities of the first step are asked for.
        // They travel with the vertex words; heads and costs follow for the lanes that found a
        // residual arc (the rows of machines are mostly reverse arcs without flow).
        load9(o.cap, r0 + plane * 8, min(r1, r0 + 512), r0 + 512 + plane, min(r1, r0 + kStep), cp);
    }
    if (seg1) {
        // no residual arc of the segment has a key above the bound: with bound <= price[u] none is
        // admissible, and the bound stands in for the segment's relabel candidate
        i64 b = g.seg_scan[seg1 - 1];
        const i64 b0 = g.seg_add[0][seg1 - 1], b1 = g.seg_add[1][seg1 - 1];
        b = b > b0 ? b : b0;
        b = b > b1 ? b : b1;
        if (b <= pu) {
            if (lane == 0 && b > kNegInf) { if (hu >= 0) lds_max(&hb.ext[hu], b); else amax64(&g.cand[u], b); }
            return;
        }
    }
    i64 e0 = __hip_atomic_load(purse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    e0 = __shfl(e0, 0);
    if (lane == 0) t.c[kTallyVisits]++;
    if (!SOLO && e0 <= 0) {  // drained by the vertex's other items: this one stays unscanned
        if (lane == 0) { if (hu >= 0) lds_max(&hb.ext[hu], kInf); else amax64(&g.cand[u], kInf); }
        return;
    }
    i64 e = e0, best = kNegInf, drawn = 0;
    bool incomplete = false;
    for (unsigned h = r0; h < r1; h += kStep) {
        const unsigned a0 = h + plane * 8, lim8 = min(r1, h + 512), at = h + 512 + plane, limt = min(r1, h + kStep);
        int hd[9], cs[9];
        if (!SOLO || h != r0) load9(o.cap, a0, lim8, at, limt, cp);
        t.c[kTallyScans] += count9(a0, lim8, at, limt);
        bool lane_live = true;
        if constexpr (SOLO) {
            unsigned lm = 0;
#pragma unroll
            for (int j = 0; j < 9; ++j) lm |= (cp[j] > 0 ? 1u : 0u) << j;
            lane_live = lm != 0;
            const unsigned nl = (unsigned)__popc(lm), incl_l = wave_incl_scan_u32(nl), total_live = (unsigned)__shfl((int)incl_l, 63);
            if (total_live <= 64u) {  // wave-uniform
                t.c[kTallySkips] += count9(a0, lim8, at, limt) - nl;
                const StepOut so = push_step_sparse(g, o, hb, slive, cp, lm, incl_l - nl, total_live, a0, at, h, pu, S, e, par, t);
                best = so.here > best ? so.here : best;
                if (so.stop) {
                    if (h + kStep < r1) incomplete = true;
                    break;
                }
                continue;
            }
        }
        load9(g.head, a0, lim8, at, limt, hd, lane_live, h);
        load9(g.cost, a0, lim8, at, limt, cs, lane_live, h);
        // Keys price[head] - cost * S are not kept per arc: what later steps need is which arcs are
        // admissible (a bit mask), the largest key of all live arcs and the largest key among the
        // live arcs that are not admissible.  If an admissible arc is still residual after the
        // pushes, the vertex needs no relabel and any value above its price serves as candidate
        // (the largest key of all does, and it also bounds the segment from above); otherwise the
        // candidate is exactly the largest non-admissible key.
        i64 ph[9], want = 0, kmax_all = kNegInf, kmax_rest = kNegInf;
        unsigned adm = 0;
        bool live[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) live[j] = cp[j] > 0;
        gather9(hd, live, ph, (i64)0, [&](bool c, int v) { return ld_sel(c, &g.price[v], g.spare); });
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (live[j]) {
                const i64 key = ph[j] - (i64)cs[j] * S;
                kmax_all = key > kmax_all ? key : kmax_all;
                if (key > pu) { adm |= 1u << j; want += cp[j]; }
                else kmax_rest = key > kmax_rest ? key : kmax_rest;
            }
        }
        bool stop = false;
        if (__ballot(want > 0)) {
            const i64 incl = wave_incl_scan(want);
            const i64 total = __shfl(incl, 63);
            i64 granted;
            if (solo) {
                granted = total < e ? total : e;
                e -= granted;
            } else {
                granted = 0;
                if (lane == 0) {
                    const i64 old = aadd64(purse, -total);
                    granted = old <= 0 ? 0 : (old < total ? old : total);
                    if (granted < total) aadd64(purse, total - granted);
                }
                granted = __shfl(granted, 0);
                drawn += granted;
            }
            i64 mine = granted - (incl - want);  // what this lane may push
            if (mine > 0) {
                // three batches of three slots (twelve registers less than two of five and four): amounts
                // first, then every load of the batch before any dependent store
#pragma unroll
                for (int b0 = 0; b0 < 9; b0 += 3) {
                    constexpr int W = 3;
                    int d[W], rv[W], cr[W], hub[W];
#pragma unroll
                    for (int x = 0; x < W; ++x) {
                        const int j = b0 + x;
                        d[x] = 0;
                        if (j < 9 && ((adm >> j) & 1u) && mine > 0) { d[x] = (int)(mine < cp[j < 9 ? j : 0] ? mine : cp[j < 9 ? j : 0]); mine -= d[x]; }
                    }
#pragma unroll
                    for (int x = 0; x < W; ++x) {  // rcap mirrors cap[rev]: no dependent load for the reverse capacity
                        const int j = b0 + x < 9 ? b0 + x : 8;
                        const bool on = d[x] > 0;
                        const unsigned ax = on ? arc9(a0, at, j) : h;  // no control flow around the loads (see ld_sel)
                        const int rvx = g.rev[ax], crx = o.rcap[ax], hbx = ld_sel(on, &g.hub_of[hd[j]], g.spare);
                        rv[x] = on ? rvx : 0; hub[x] = on ? hbx : -1; cr[x] = on ? crx : 0;
                    }
#pragma unroll
                    for (int x = 0; x < W; ++x) {
                        const int j = b0 + x < 9 ? b0 + x : 8;
                        if (d[x] > 0) {
                            const unsigned a = arc9(a0, at, j);
                            const int na = cp[j] - d[x], nr = cr[x] + d[x];
                            o.cap[a] = na; o.cap[rv[x]] = nr; o.rcap[a] = nr; o.rcap[rv[x]] = na;
                            const i64 kr = pu + (i64)cs[j] * S;  // key of the new residual arc (v,u)
                            if (hub[x] >= 0) { lds_add(&hb.add[hub[x]], (i64)d[x]); lds_max(&hb.ext[hub[x]], kr); seg_note(g, hub[x], rv[x], kr, par); }
                            else { aadd64(&g.excess[hd[j]], (i64)d[x]); amax64(&g.cand[hd[j]], kr); }
                            cp[j] = na;
                            t.c[kTallyPushes]++;
                        }
                    }
                }
            }
            stop = granted < total || (solo && e == 0);  // out of excess
        }
        bool adm_left = false;
#pragma unroll
        for (int j = 0; j < 9; ++j) adm_left |= ((adm >> j) & 1u) && cp[j] > 0;
        const i64 here = adm_left ? kmax_all : kmax_rest;
        best = here > best ? here : best;
        if (stop) {
            if (h + kStep < r1) incomplete = true;
            break;
        }
    }
    best = wave_max(best);
    if (lane == 0) {
        if (seg1 && !incomplete) {
            // every arc of the segment was looked at: the bound is exact again; keys noted during
            // earlier rounds of the other parity are covered by it (this round's notes are not)
            g.seg_scan[seg1 - 1] = best;
            g.seg_add[par ^ 1][seg1 - 1] = kNegInf;
        }
        if (incomplete) best = kInf;  // arcs not looked at: no relabel this round
        if (best > kNegInf) { if (hu >= 0) lds_max(&hb.ext[hu], best); else amax64(&g.cand[u], best); }
        if (solo && e != e0) aadd64(&g.excess[u], e - e0);
        if (hu >= 0 && drawn) lds_add(&hb.out[hu], drawn);  // what left the purse leaves the hub's excess
    }
}

__device__ __forceinline__ void push_item(const DevGraph &g, const Orient &o, const HubAcc &hb, int2 *slive, int4 it, i64 S, unsigned rot, int par, Tally &t) {
    if (it.w & 1) push_item_t<true>(g, o, hb, slive, it, S, rot, par, t);
    else push_item_t<false>(g, o, hb, slive, it, S, rot, par, t);
}

__device__ __forceinline__ void push_body(const DevGraph &g, int k) {
    Ctrl *ct = g.ctrl;
    const unsigned *cnt = ct->list_cnt[k & 1];
    const int on = ct->rounds_on;
    const unsigned nlow = cnt[kSlotLow], nseg = cnt[kSlotSeg], epoch = ct->epoch;
    const int mirror = ct->mirror;
    CTRL_TOGETHER5(on, nlow, nseg, epoch, mirror);
    if (!on) return;
    const Orient o = orient_of(g, mirror);
    if (blockIdx.x == 0 && threadIdx.x < kCntSlots) ct->list_cnt[(k + 1) & 1][cnt_slot((int)threadIdx.x)] = 0;
    if (k == 0 && blockIdx.x == 0 && threadIdx.x == 0 && ct->first_units == ~0u) ct->first_units = cnt[kSlotUnits];
    if (nlow + nseg == 0) return;
    if (g.dbg && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned i = aadd32(g.dbg_cnt, 1u);
        if (i < kDbgCap) g.dbg[i] = make_int4(1, k, (int)nlow, (int)nseg);
    }
    __shared__ i64 s_add[kMaxHubs], s_max[kMaxHubs], s_out[kMaxHubs];
    __shared__ int2 s_live[kBlock / 64][64];  // per wave: the residual arcs (index, capacity) of a sparse step
    const HubAcc hb{s_add, s_max, s_out};
    hub_init(hb, kNegInf);
    const i64 S = o.S;
    const unsigned salt = epoch * 64u + (unsigned)k;
    const int2 *lowl = g.list_low[k & 1];
    const int4 *segl = g.list_seg[k & 1];
    const unsigned wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * kBlock) >> 6;
    Tally t;
    const int my_hub_vertex = hub_vertex_of_thread(g);
    int4 it_next = wave < nseg ? segl[wave] : make_int4(0, 0, 0, 0);
    for (unsigned i = wave; i < nseg; i += nwaves) {
        const int4 it = it_next;
        if (i + nwaves < nseg) it_next = segl[i + nwaves];  // on its way while this item is worked on
        push_item(g, o, hb, s_live[threadIdx.x >> 6], it, S, mix32((unsigned)it.x, salt) >> 16, k & 1, t);
    }
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < nlow; i += gridDim.x * kBlock) push_low(g, o, hb, lowl[i], S, k & 1, t);
    hub_flush_push<true>(g, hb, my_hub_vertex);
    if (blockIdx.x == 0 && threadIdx.x == 0) t.c[kTallyRounds]++;
    tally_flush<kTallyScans, kTallyPushes, kTallyVisits, kTallyRounds, kTallySkips>(g, t);
}

// Relabel (optional) + rebuild of the active lists for the next round: one sweep over
// the vertices, kVpt per thread.  cand[v] is +inf for vertices outside the list, so a
// vertex that only became active during the push phase is never relabelled on partial
// information.
// Hubs are left out of the vertex sweep: workgroup h looks after hub h with all its threads.
// The task classes sit next to each other, so the vertex sweep would hand all of them to one
// workgroup, and one compute unit writes the ~4500 items of their segments several times
// slower than the rest of the launch takes altogether.
struct HubTurn { int v; bool act; unsigned r0, r1, nseg, first; int seg1; i64 e, c, c_stored, p; };  // c: the candidate with the round's keys folded in; c_stored: what cand[] holds
// The hub of slot `slot`: its words, with what the push round left in the accumulator copies folded in
// (excess[] is brought up to date, the copies are emptied; the caller resets cand[]).
struct HubShards { i64 add[kHubShards], ext[kHubShards]; };
// (their addresses follow from the slot alone, so a sweep asks for them before it knows anything else)
__device__ __forceinline__ void hub_shards_load(const DevGraph &g, int slot, bool have, HubShards &sh) {
#pragma unroll
    for (int k = 0; k < kHubShards; ++k) {
        sh.add[k] = have ? g.hub_acc_add[k * kMaxHubs + slot] : 0;
        sh.ext[k] = have ? g.hub_acc_ext[k * kMaxHubs + slot] : kNegInf;
    }
}
__device__ __forceinline__ HubTurn hub_fetch(const DevGraph &g, int hv, int slot, const HubShards &sh) {
    HubTurn h;
    h.v = hv; h.e = g.excess[hv]; h.c = h.c_stored = g.cand[hv]; h.p = g.price[hv];
    h.r0 = g.row[hv]; h.r1 = g.row[hv + 1]; h.seg1 = g.seg1_of[hv];
    const i64 (&add)[kHubShards] = sh.add, (&ext)[kHubShards] = sh.ext;
    i64 sum = 0;
#pragma unroll
    for (int k = 0; k < kHubShards; ++k) {
        sum += add[k];
        h.c = ext[k] > h.c ? ext[k] : h.c;
        if (add[k]) g.hub_acc_add[k * kMaxHubs + slot] = 0;
        if (ext[k] > kNegInf) g.hub_acc_ext[k * kMaxHubs + slot] = kNegInf;
    }
    if (sum) { h.e += sum; g.excess[hv] = h.e; }
    return h;
}
__device__ __forceinline__ unsigned excess_units(i64 e) { return e > 0 ? (unsigned)(e < 65535 ? e : 65535) : 0u; }
// The thread that looks after a hub relabels it and resets its candidate.
__device__ __forceinline__ void hub_decide(const DevGraph &g, HubTurn &h, bool do_relabel, i64 eps, unsigned rot, Tally &t, unsigned &units, int slot) {
    h.act = h.e > 0;
    h.nseg = (h.r1 - h.r0 + kWSeg - 1) / kWSeg;
    h.first = (rot ^ (unsigned)h.v * 0x9E3779B9u) % h.nseg;
    if (h.act) units += excess_units(h.e);
    if (h.act && do_relabel && h.c != kInf && h.c != kNegInf && h.c <= h.p) { g.price[h.v] = h.c - eps; t.c[kTallyRelabels]++; }
    const i64 nc = h.act ? kNegInf : kInf;
    if (h.c_stored != nc) g.cand[h.v] = nc;
    if (g.hub_shards) g.hub_purse[slot * kPurseStride] = h.act ? h.e : 0;  // what the hub's segments may push in the coming round
}
// The hubs of a workgroup (hub slots blockIdx.x, blockIdx.x + gridDim.x, ...: one thread each) leave
// their records in LDS; their segments take consecutive slots of the wave-item list.
struct HubRec { int v, seg1, slot; unsigned r0, r1, first, nseg, off; };  // nseg = 0: not active
struct HubLds { HubRec rec[kMaxHubs]; unsigned total, base; };
__device__ __forceinline__ void append_hubs(const HubLds &hl, unsigned count, unsigned slot_base, int4 *segl) {
    const unsigned w = threadIdx.x >> 6, l = (unsigned)lane_id();
    for (unsigned j = w; j < count; j += kBlock / 64) {  // a wave per hub, its lanes side by side
        const HubRec h = hl.rec[j];
        for (unsigned i = l; i < h.nseg; i += 64) {
            unsigned sg = h.first + i;  // rotated order, as append_split
            if (sg >= h.nseg) sg -= h.nseg;
            const unsigned a = h.r0 + sg * kWSeg;
            segl[slot_base + h.off + i] = make_int4(h.v, (int)a, (int)min(h.r1, a + kWSeg), ((h.seg1 + (int)sg) << 1) | (h.slot << kItemHubShift));
        }
    }
}

template <int kVpt>
__device__ __forceinline__ void relabel_compact_body_t(const DevGraph &g, int k, int do_relabel) {
    Ctrl *ct = g.ctrl;
    // The first batch of vertex words is asked for before the control words are known: both
    // arrive in one round trip, and a launch that turns out to be switched off just drops them.
    const int span = kBlock * kVpt;
    int v[kVpt], cls[kVpt], seg1[kVpt];
    unsigned r0[kVpt], r1[kVpt];
    i64 e[kVpt], c[kVpt], p[kVpt];
    auto fetch = [&](int base) {
#pragma unroll
        for (int i = 0; i < kVpt; ++i) {  // everything in one round trip: the sweep is latency-bound
            v[i] = base + i * kBlock + (int)threadIdx.x;
            const bool in = v[i] < g.n;
            e[i] = in ? g.excess[v[i]] : 0;
            c[i] = in ? g.cand[v[i]] : kInf;
            cls[i] = in ? (int)(g.cls[v[i]] & kClsMask) : 0;  // no lean bit in the push lists: their flags hold bound slots
            p[i] = in ? g.price[v[i]] : 0;
            r0[i] = in ? g.row[v[i]] : 0;
            r1[i] = in ? g.row[v[i] + 1] : 0;
            seg1[i] = in ? g.seg1_of[v[i]] : 1;
        }
    };
    fetch((int)blockIdx.x * span);
    // the hubs of this workgroup, one per thread: their vertex is asked for with the control words as well
    const int my_hs = (int)blockIdx.x + (int)threadIdx.x * (int)gridDim.x;
    const bool have_hub = threadIdx.x < (unsigned)kMaxHubs && my_hs < g.nhubs;
    const int hub_v = have_hub ? g.hub_vertex[my_hs] : -1;
    HubShards hub_sh;
    hub_shards_load(g, my_hs, have_hub && g.hub_shards != 0, hub_sh);
    const unsigned *cnt = ct->list_cnt[k & 1];
    const int on = ct->rounds_on;
    const unsigned nlow = cnt[kSlotLow], nseg = cnt[kSlotSeg], epoch = ct->epoch;
    const i64 eps = ct->eps;
    CTRL_TOGETHER5(on, nlow, nseg, epoch, eps);
    if (!on) return;
    const int dst = do_relabel ? (k + 1) & 1 : 0;
    if (do_relabel && nlow + nseg == 0) return;  // nothing was active
    const unsigned rot = mix32(epoch, (unsigned)k);
    Tally t;
    unsigned units_acc = 0, ndef_acc = 0;  // this thread's share of the two totals, added up once per workgroup at the end
    __shared__ AppendLds s_app;
    __shared__ HubLds s_hub;
    __shared__ unsigned s_tot[2][kBlock / 64];
    const unsigned n_my_hubs = (int)blockIdx.x < g.nhubs ? min((unsigned)kMaxHubs, ((unsigned)g.nhubs - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0u;
    if (n_my_hubs) {  // workgroup-uniform
        HubTurn hub{};
        if (have_hub) hub = hub_fetch(g, hub_v, my_hs, hub_sh);
        if (threadIdx.x == 0) s_hub.total = 0;
        __syncthreads();
        if (have_hub) {
            hub_decide(g, hub, do_relabel, eps, rot, t, units_acc, my_hs);
            HubRec r;
            r.v = hub.v; r.seg1 = hub.seg1; r.slot = my_hs; r.r0 = hub.r0; r.r1 = hub.r1; r.first = hub.first;
            r.nseg = hub.act ? hub.nseg : 0u;
            r.off = r.nseg ? __hip_atomic_fetch_add(&s_hub.total, r.nseg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
            s_hub.rec[threadIdx.x] = r;
        }
        __syncthreads();
    }
    const unsigned hub_total = n_my_hubs ? s_hub.total : 0u;  // their slots ride on the sweep's own slot atomic
    bool hubs_pending = n_my_hubs != 0;
    for (int base = blockIdx.x * span; base < g.n; base += gridDim.x * span) {
        if (base != (int)blockIdx.x * span) fetch(base);
        bool act[kVpt];
        unsigned hub_base = 0;
#pragma unroll
        for (int i = 0; i < kVpt; ++i) {
            act[i] = e[i] > 0 && seg1[i] == 0;  // not a hub (and inside the vertex range)
            if (!(act[i] && do_relabel && c[i] != kInf && c[i] != kNegInf)) p[i] = kNegInf;
        }
#pragma unroll
        for (int i = 0; i < kVpt; ++i) {
            if (p[i] != kNegInf && c[i] <= p[i]) { g.price[v[i]] = c[i] - eps; t.c[kTallyRelabels]++; }
            const i64 nc = act[i] ? kNegInf : kInf;
            if (seg1[i] == 0 && c[i] != nc) g.cand[v[i]] = nc;
        }
#pragma unroll
        for (int i = 0; i < kVpt; ++i) { units_acc += act[i] ? excess_units(e[i]) : 0u; ndef_acc += e[i] < 0; }  // e is 0 outside the vertex range
        block_append<kVpt>(ct->list_cnt[dst], g.list_low[dst], g.list_seg[dst], act, v, cls, r0, r1, rot, s_app, hubs_pending ? hub_total : 0u, &hub_base);
        if (hubs_pending) { if (hub_total) append_hubs(s_hub, n_my_hubs, hub_base, g.list_seg[dst]); hubs_pending = false; }
    }
    if (hubs_pending && hub_total) {  // this workgroup had no vertices to sweep: the hubs' slots are taken on their own
        if (threadIdx.x == 0) s_hub.base = aadd32(&ct->list_cnt[dst][kSlotSeg], hub_total);
        __syncthreads();
        append_hubs(s_hub, n_my_hubs, s_hub.base, g.list_seg[dst]);
    }
    {
        // one atomic per workgroup and total (a wave each would put thousands of them on one cache line)
        units_acc = wave_sum_u32(units_acc);
        ndef_acc = wave_sum_u32(ndef_acc);
        if (lane_id() == 0) { s_tot[0][threadIdx.x >> 6] = units_acc; s_tot[1][threadIdx.x >> 6] = ndef_acc; }
        __syncthreads();
        if (threadIdx.x < 2) {
            unsigned x = 0;
            for (int i = 0; i < kBlock / 64; ++i) x += s_tot[threadIdx.x][i];
            if (x) __hip_atomic_fetch_add(&ct->list_cnt[dst][threadIdx.x == 0 ? kSlotUnits : kSlotDeficits], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    tally_flush<kTallyRelabels>(g, t);
}

// A small network solved on its own takes one vertex per thread (twice the workgroups, half the work behind each
// wait: 1-2 % of a solve); large ones, and the members of a lockstep group (whose launches already hold a hundred
// networks' workgroups: 5 % the other way), take two.
__device__ __forceinline__ void relabel_compact_body(const DevGraph &g, int k, int do_relabel) {
    if (g.n <= kVptSmallN) relabel_compact_body_t<1>(g, k, do_relabel);
    else relabel_compact_body_t<2>(g, k, do_relabel);
}

// ------------------------------------------------------------- per-phase saturation
// refine(eps) starts from an eps-optimal pseudoflow: every residual arc whose reduced
// cost is below -eps is saturated.  Four consecutive arcs per thread; only the thread of
// arc a writes cap[a] and cap[rev[a]] (the reverse has reduced cost > eps and is left alone).
__device__ __forceinline__ void saturate_body(const DevGraph &g) {
    Ctrl *ct = g.ctrl;
    if (!ct->sat_on) return;
    // capacities change wholesale below: the hub segment bounds start over
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n_hub_segs; i += gridDim.x * kBlock) {
        g.seg_scan[i] = kInf; g.seg_add[0][i] = kNegInf; g.seg_add[1][i] = kNegInf;
    }
    const Orient o = orient_of(g, ct->mirror);
    const i64 eps = ct->eps, S = o.S;
    Tally t;
    __shared__ i64 s_add[kMaxHubs], s_max[kMaxHubs];
    const HubAcc hb{s_add, s_max, nullptr};
    hub_init(hb, kNegInf);
    const unsigned nquad = (g.m2 + 3) / 4;
    for (unsigned q = blockIdx.x * kBlock + threadIdx.x; q < nquad; q += gridDim.x * kBlock) {
        const unsigned a0 = q * 4;
        int cp[4], tl[4], hd[4], cs[4];
        if (a0 + 3 < g.m2) {
            const int4u x = *reinterpret_cast<const int4u *>(o.cap + a0);
            cp[0] = x.x; cp[1] = x.y; cp[2] = x.z; cp[3] = x.w;
            t.c[kTallySat] += 4;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { cp[j] = a0 + j < g.m2 ? o.cap[a0 + j] : 0; if (a0 + j < g.m2) t.c[kTallySat]++; }
        }
        if ((cp[0] | cp[1] | cp[2] | cp[3]) == 0) continue;  // capacities are never negative
        if (a0 + 3 < g.m2) {
            const int4u y = *reinterpret_cast<const int4u *>(g.tail + a0);
            const int4u z = *reinterpret_cast<const int4u *>(g.head + a0);
            const int4u w = *reinterpret_cast<const int4u *>(g.cost + a0);
            tl[0] = y.x; tl[1] = y.y; tl[2] = y.z; tl[3] = y.w;
            hd[0] = z.x; hd[1] = z.y; hd[2] = z.z; hd[3] = z.w;
            cs[0] = w.x; cs[1] = w.y; cs[2] = w.z; cs[3] = w.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = a0 + j < g.m2;
                tl[j] = ok ? g.tail[a0 + j] : 0; hd[j] = ok ? g.head[a0 + j] : 0; cs[j] = ok ? g.cost[a0 + j] : 0;
            }
        }
        i64 rc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rc[j] = cp[j] > 0 ? (i64)cs[j] * S + g.price[tl[j]] - g.price[hd[j]] : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (cp[j] > 0 && rc[j] < -eps) {
                const unsigned a = a0 + j;
                const int r = g.rev[a];
                const int nr = o.cap[r] + cp[j];
                o.cap[a] = 0; o.cap[r] = nr; o.rcap[a] = nr; o.rcap[r] = 0;
                excess_add(g, hb, tl[j], -(i64)cp[j]);
                excess_add(g, hb, hd[j], (i64)cp[j]);
            }
        }
    }
    hub_flush_push<false>(g, hb, hub_vertex_of_thread(g));
    tally_flush<kTallySat>(g, t);
}


// ------------------------------------------------------- global price update (K3)
// Distances (in units of eps) to the nearest deficit over reversed residual arcs with
// length floor(rc/eps)+1 >= 0, by frontier relaxation to a fixed point; then
// price[v] -= eps*dist[v].  Newly improved vertices are staged in a per-wave LDS queue
// and flushed to the global frontier in coalesced bursts with one atomic per burst.
struct WaveQ {
    int2 *low;            // LDS staging, kQCap lane items
    int4 *seg;            // LDS staging, kQCap wave items
    unsigned nlow, nseg;  // wave-uniform fill counts
};
template <typename T>
__device__ __forceinline__ void waveq_flush_one(T *q, unsigned &n, unsigned *cnt, T *dst) {
    if (n == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    unsigned base = 0;
    if (lane_id() == 0) base = aadd32(cnt, n);
    base = __shfl(base, 0);
    for (unsigned i = lane_id(); i < n; i += 64) dst[base + i] = q[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    n = 0;
}

// Distance pre-checks use plain loads.  The vector L1 is not coherent with the atomics that
// lower distances during a sweep, but it starts every kernel empty and distances only go down
// within one update, so a cached value can only be too large: the check then lets a
// candidate through to the atomic min, which decides.  (Agent-scope atomic loads would be
// fresher, but the compiler waits for each of them separately: nine serial round trips per step.)
__device__ __forceinline__ i64 ld_dist(const DevGraph &g, int u) { return dist_of(g.dist[u]); }
__device__ __forceinline__ i64 arc_len(i64 rc, int sh) {
    i64 l = (rc >> sh) + 1;
    return l < 0 ? 0 : (l > kLenMax ? kLenMax : l);
}

// A value the optimiser cannot see through.  `x = dflt; if (c) x = atomic(...)` with a constant
// default lets it fold whatever consumes x into the branch, and the wait for the atomic's result
// with it: NJ atomics then take NJ round trips one after the other.
__device__ __forceinline__ i64 opaque(i64 x) { asm("" : "+v"(x)); return x; }

// Commit up to NJ tentative distances per lane (nd[j] == kDistInf: none) and queue the
// vertices this wave is the first to improve in this iteration.  The atomic mins of all NJ
// candidates and the class / row lookups their vertices will need if they have to be queued are
// issued together, before any result is used: a lane pays one round trip per step for the
// commit, not one per arc and not a second one for the lookups.
// Hubs (hub[j] >= 0) are only lowered in LDS here; the workgroup commits them at the end.
template <int NJ>
__device__ __forceinline__ void gu_commit_batch(const DevGraph &g, WaveQ &q, unsigned *dcnt, int2 *dlow, int4 *dseg, i64 *hub_min,
                                                const int (&hd)[NJ], const int (&hub)[NJ], const i64 (&nd)[NJ], unsigned stamp,
                                                unsigned rot, i64 &wmax, Tally &t) {
    bool anyc = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) anyc |= nd[j] < kDistInf;
    if (!__ballot(anyc)) return;
    const i64 none = opaque(kNegInf);
    i64 old[NJ];
    bool far[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const bool c = nd[j] < kDistInf && hub[j] != kHubPull;  // pull vertices fetch their distance themselves
        far[j] = c && hub[j] < 0;
        old[j] = none;
        if (c && hub[j] >= 0) lds_min(&hub_min[hub[j]], nd[j]);
        if (far[j]) old[j] = amin64(&g.dist[hd[j]], dist_pack(nd[j], stamp));
    }
    int clsb[NJ];
    unsigned r0[NJ], r1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {  // in flight together with the atomics
        clsb[j] = (int)ld_sel(far[j], &g.cls[hd[j]], g.spare);
        r0[j] = ld_sel(far[j], &g.row[hd[j]], g.spare);
        r1[j] = ld_sel(far[j], &g.row[hd[j] + 1], g.spare);
    }
    int cls[NJ];
    unsigned nl = 0, ns = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        // improved: the stored distance was larger; first: nobody lowered it during this sweep
        // yet, so this lane queues the vertex
        const bool improved = far[j] && nd[j] < dist_of(old[j]);
        const bool first = improved && (unsigned)(old[j] & 0xFFFF) != stamp;
        if (improved) {
            t.c[kTallyGuRelax]++;
            if (nd[j] > wmax) wmax = nd[j];
        }
        cls[j] = first ? (clsb[j] & kClsMask) : -1;
        nl += cls[j] == kClsLow;
        ns += cls[j] == kClsMed;
    }
    if (__ballot(nl != 0)) {
        const unsigned incl = wave_incl_scan_u32(nl);
        const unsigned tot = __shfl(incl, 63);
        if (q.nlow + tot > kQCap) waveq_flush_one(q.low, q.nlow, &dcnt[kSlotLow], dlow);
        if (tot > kQCap) {  // more than the staging queue holds: straight to memory
            unsigned base = 0;
            if (lane_id() == 0) base = aadd32(&dcnt[kSlotLow], tot);
            unsigned pos = __shfl(base, 0) + incl - nl;
#pragma unroll
            for (int j = 0; j < NJ; ++j) if (cls[j] == kClsLow) dlow[pos++] = low_item(hd[j], r0[j], r1[j]);
        } else {
            unsigned pos = q.nlow + incl - nl;
#pragma unroll
            for (int j = 0; j < NJ; ++j) if (cls[j] == kClsLow) q.low[pos++] = low_item(hd[j], r0[j], r1[j]);
            q.nlow += tot;
        }
    }
    if (__ballot(ns != 0)) {
        const unsigned incl = wave_incl_scan_u32(ns);
        const unsigned tot = __shfl(incl, 63);
        if (q.nseg + tot > kQCap) waveq_flush_one(q.seg, q.nseg, &dcnt[kSlotSeg], dseg);
        if (tot > kQCap) {
            unsigned base = 0;
            if (lane_id() == 0) base = aadd32(&dcnt[kSlotSeg], tot);
            unsigned pos = __shfl(base, 0) + incl - ns;
#pragma unroll
            for (int j = 0; j < NJ; ++j) if (cls[j] == kClsMed) dseg[pos++] = solo_item(hd[j], r0[j], r1[j], (clsb[j] & kClsLean) != 0);
        } else {
            unsigned pos = q.nseg + incl - ns;
#pragma unroll
            for (int j = 0; j < NJ; ++j) if (cls[j] == kClsMed) q.seg[pos++] = solo_item(hd[j], r0[j], r1[j], (clsb[j] & kClsLean) != 0);
            q.nseg += tot;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) if (cls[j] == kClsHigh) append_split(dcnt, dseg, hd[j], r0[j], r1[j], rot ^ (unsigned)hd[j]);
}

__device__ __forceinline__ void gu_expand_body(const DevGraph &g, int k) {
    Ctrl *ct = g.ctrl;
    const int src = k % 3, dst = (k + 1) % 3, clr = (k + 2) % 3;
    const unsigned *cnt = ct->fr_cnt[src];
    const int on = ct->gu_on;
    const unsigned nlow = cnt[kSlotLow], nseg = cnt[kSlotSeg];
    const int phase = ct->gu_phase[k & 1];
    const unsigned npull = *g.pull_cnt;
    const int sh = ct->eps_shift;
    const unsigned stamp_base = ct->gu_stamp_base, epoch = ct->epoch;
    const int mirror = ct->mirror;
    CTRL_TOGETHER8(on, nlow, nseg, phase, npull, sh, stamp_base, epoch);
    CTRL_TOGETHER4(mirror, on, nlow, nseg);
    if (!on) return;
    const Orient o = orient_of(g, mirror);
    if (blockIdx.x == 0 && threadIdx.x < 2) ct->fr_cnt[clr][threadIdx.x] = 0;
    // What this launch does follows from words it does not write: the frontier it reads and the
    // phase the launch before it left.  A dry frontier is followed by one sweep over the pull
    // vertices; if that queues nothing either, the update has converged.
    const bool dry = nlow + nseg == 0;
    const bool do_pull = phase == 0 && dry && npull != 0;
    const int phase_out = phase == 2 ? 2 : (!dry ? 0 : (do_pull ? 1 : 2));
    if (blockIdx.x == 0 && threadIdx.x == 0) ct->gu_phase[(k + 1) & 1] = phase_out;
    if (phase == 2 || (dry && !do_pull)) return;
    if (g.dbg && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned i = aadd32(g.dbg_cnt, 1u);
        if (i < kDbgCap) g.dbg[i] = do_pull ? make_int4(3, k, (int)npull, 0) : make_int4(2, k, (int)nlow, (int)nseg);
    }
    __shared__ int2 lds_low[(kBlock / 64) * kQCap];
    __shared__ int4 lds_seg[(kBlock / 64) * kQCap];
    __shared__ i64 s_hmin[kMaxHubs];
    for (int i = threadIdx.x; i < kMaxHubs; i += kBlock) s_hmin[i] = kDistInf;
    __syncthreads();
    WaveQ q;
    q.low = lds_low + (threadIdx.x >> 6) * kQCap;
    q.seg = lds_seg + (threadIdx.x >> 6) * kQCap;
    q.nlow = q.nseg = 0;
    unsigned *dcnt = ct->fr_cnt[dst];
    int2 *dlow = g.fr_low[dst];
    int4 *dseg = g.fr_seg[dst];
    const i64 S = o.S;
    // under a negative scale (mirror) the cheapest, not the dearest, of a group of parallel arcs has the shortest length
    const int sgn = S < 0 ? -1 : 1;
    const i64 Sabs = S < 0 ? -S : S;
    const unsigned stamp = (stamp_base + (unsigned)k) % 65535u + 1u;  // 16 bits, never 0
    const unsigned rot = mix32(epoch, stamp);
    const int lane = lane_id();
    const unsigned wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * kBlock) >> 6;
    Tally t;
    const int my_hub_vertex = hub_vertex_of_thread(g);
    i64 wmax = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) t.c[kTallyGuIters]++;
    if (do_pull) {
        const int2 *pl = g.pull_list;
        const unsigned np_round = (npull + 63) & ~63u;
        for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < np_round; i += gridDim.x * kBlock) {
            const bool have = i < npull;
            const int2 it = pl[have ? i : 0];
            const int v = it.x & 0x0FFFFFFF;
            const unsigned deg = have ? (unsigned)it.x >> 28 : 0u, r0 = (unsigned)it.y;
            const i64 dcur = dist_of(g.dist[v]), pv = g.price[v];
            int rc4[kLowMax], cp4[kLowMax], hd[kLowMax], cs[kLowMax], hub[kLowMax];
            i64 pu[kLowMax], du[kLowMax], nd[kLowMax];
#pragma unroll
            for (unsigned j = 0; j < kLowMax; ++j) {  // entries past the row's end are read (the arrays have spare entries) and dropped
                const bool ok = j < deg;
                const int rcj = o.rcap[r0 + j], cpj = o.cap[r0 + j], hdj = g.head[r0 + j], csj = g.cost[r0 + j];
                rc4[j] = ok ? rcj : 0; cp4[j] = ok ? cpj : 0; hd[j] = ok ? hdj : 0; cs[j] = ok ? csj : 0;
            }
            t.c[kTallyGuScans] += deg;
#pragma unroll
            for (unsigned j = 0; j < kLowMax; ++j) {
                const bool ok = rc4[j] > 0 || cp4[j] > 0;
                pu[j] = ld_sel(ok, &g.price[hd[j]], g.spare); du[j] = dist_of(ld_sel(ok, &g.dist[hd[j]], g.spare));
                const int hb = ld_sel(rc4[j] > 0, &g.hub_of[hd[j]], g.spare);
                hub[j] = rc4[j] > 0 ? hb : -1;
                if (!ok) du[j] = kDistInf;
            }
            // the vertex's own distance, over its residual arcs
            i64 best = dcur;
#pragma unroll
            for (unsigned j = 0; j < kLowMax; ++j) {
                if (cp4[j] > 0 && du[j] < kDistInf) {
                    const i64 l = arc_len((i64)cs[j] * S + pv - pu[j], sh);
                    if (du[j] + l < best) best = du[j] + l;
                }
            }
            const bool changed = best < dcur;
            if (changed) {
                g.dist[v] = dist_pack(best, stamp);  // nobody else writes a pull vertex's distance
                t.c[kTallyGuRelax]++;
                if (best > wmax) wmax = best;
            }
            // and from there its neighbours', as an expansion of the vertex would
#pragma unroll
            for (unsigned j = 0; j < kLowMax; ++j) {
                nd[j] = kDistInf;
                if (changed && rc4[j] > 0) {
                    const i64 l = arc_len(-(i64)cs[j] * S + pu[j] - pv, sh);
                    if (best + l < du[j]) nd[j] = best + l;
                }
            }
            gu_commit_batch<(int)kLowMax>(g, q, dcnt, dlow, dseg, s_hmin, hd, hub, nd, stamp, rot, wmax, t);
        }
    }
    const int4 *segl = g.fr_seg[src];
    int4 it_next = wave < nseg ? segl[wave] : make_int4(0, 0, 0, 0);
    for (unsigned i = wave; i < nseg; i += nwaves) {
        // the item is the same for the whole wave: keep it in scalar registers, and fetch the next
        // one now so that its round trip is over when this item is
        const int v = __builtin_amdgcn_readfirstlane(it_next.x);
        const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane(it_next.y), r1 = (unsigned)__builtin_amdgcn_readfirstlane(it_next.z);
        const int flags = __builtin_amdgcn_readfirstlane(it_next.w);
        if (i + nwaves < nseg) it_next = segl[i + nwaves];
        const bool solo = flags & 1;
        // first arc index past the part of v's row that is made of aligned groups of 8 parallel arcs
        // (solo items carry the answer for their whole row in the flags)
        const unsigned lean_stop = solo ? ((flags & 2) ? r1 : r0) : g.lean_stop[v];
        const i64 dv = dist_of(g.dist[v]), pv = g.price[v];
        if (r1 - r0 <= 64u) {
            // Short rows (a job's aggregator: some fifty tasks) go one arc per lane: eight arcs per
            // lane would leave seven lanes doing all the atomics and lookups of the row one after
            // the other, and such a row then costs more than a full 576-arc step.
            const unsigned a = r0 + (unsigned)lane;
            const bool in = a < r1;
            const unsigned ac = in ? a : r0;
            const int rc_raw = o.rcap[ac], hd_raw = g.head[ac], cs_raw = g.cost[ac];
            const int rc = in ? rc_raw : 0;
            int hd1[1] = {in ? hd_raw : 0}, hub1[1] = {-1};
            const int cs1 = in ? cs_raw : 0;
            t.c[kTallyGuScans] += in;
            const bool on = rc > 0;
            const i64 p1 = ld_sel(on, &g.price[hd1[0]], g.spare), d1 = dist_of(ld_sel(on, &g.dist[hd1[0]], g.spare));
            const int hb1 = ld_sel(on, &g.hub_of[hd1[0]], g.spare);
            if (on) hub1[0] = hb1;
            i64 nd1[1] = {kDistInf};
            if (on) { const i64 l = arc_len(-(i64)cs1 * S + p1 - pv, sh); if (dv + l < d1) nd1[0] = dv + l; }
            gu_commit_batch<1>(g, q, dcnt, dlow, dseg, s_hmin, hd1, hub1, nd1, stamp, rot, wmax, t);
            continue;
        }
        int rc_next[9];
        load9(o.rcap, r0 + lane * 8, min(r1, r0 + 512), r0 + 512 + lane, min(r1, r0 + kStep), rc_next, true, r0);
        for (unsigned h = r0; h < r1; h += kStep) {
            const unsigned a0 = h + lane * 8, lim8 = min(r1, h + 512), at = h + 512 + lane, limt = min(r1, h + kStep);
            int rc9[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) rc9[j] = rc_next[j];
            if (h + kStep < r1) {  // the capacities of the item's next step travel while this one is worked on
                const unsigned hn = h + kStep;
                load9(o.rcap, hn + lane * 8, min(r1, hn + 512), hn + 512 + lane, min(r1, hn + kStep), rc_next, true, hn);
            }
            const unsigned mine = count9(a0, lim8, at, limt);
            t.c[kTallyGuScans] += mine;
            if (lim8 <= lean_stop) {
                // Aligned groups of 8 parallel arcs (the slot arcs between a task class and a machine,
                // seen from either side): a lane's 8 consecutive arcs share their head, and since the
                // arc length falls as the cost grows only the dearest arc with residual capacity
                // matters.  One head and the costs are fetched, by the lanes that hold such an arc
                // (whole-vertex items fetch them at once, without waiting for the capacities); the
                // tail arc of the lane is a candidate of its own.
                int cs[9];
                int hd2[2], hub2[2];
                const bool in8 = a0 < lim8, intl = at < limt;
                bool need8 = in8, needt = intl;
                if (!solo) {
                    bool any8 = false;
#pragma unroll
                    for (int j = 0; j < 8; ++j) any8 |= rc9[j] > 0;
                    need8 = any8;
                    needt = rc9[8] > 0;
                    if (!need8 && !needt) t.c[kTallyGuSkips] += mine;  // only the 4-byte capacities were read
                    if (!__ballot(need8 || needt)) continue;
                }
                {
                    const int *qc = g.cost + (need8 ? a0 : h);
                    const int4u lo = *reinterpret_cast<const int4u *>(qc), hi = *reinterpret_cast<const int4u *>(qc + 4);
                    cs[0] = lo.x; cs[1] = lo.y; cs[2] = lo.z; cs[3] = lo.w; cs[4] = hi.x; cs[5] = hi.y; cs[6] = hi.z; cs[7] = hi.w;
                    cs[8] = g.cost[needt ? at : h];
                    hd2[0] = g.head[need8 ? a0 : h];
                    hd2[1] = g.head[needt ? at : h];
                }
                int c0 = INT32_MIN;  // largest sgn * cost among the group's residual arcs
#pragma unroll
                for (int j = 0; j < 8; ++j) if (rc9[j] > 0 && sgn * cs[j] > c0) c0 = sgn * cs[j];  // rc9 is 0 past the end of the range
                const bool on0 = need8 && c0 != INT32_MIN, on1 = needt && rc9[8] > 0;
                i64 nd2[2] = {kDistInf, kDistInf};
                const i64 p0 = ld_sel(on0, &g.price[hd2[0]], g.spare), p1 = ld_sel(on1, &g.price[hd2[1]], g.spare);
                const i64 d0 = dist_of(ld_sel(on0, &g.dist[hd2[0]], g.spare)), d1 = dist_of(ld_sel(on1, &g.dist[hd2[1]], g.spare));
                const int hb0 = ld_sel(on0, &g.hub_of[hd2[0]], g.spare), hb1 = ld_sel(on1, &g.hub_of[hd2[1]], g.spare);
                hub2[0] = on0 ? hb0 : -1;
                hub2[1] = on1 ? hb1 : -1;
                if (on0) { const i64 l = arc_len(-(i64)c0 * Sabs + p0 - pv, sh); if (dv + l < d0) nd2[0] = dv + l; }
                if (on1) { const i64 l = arc_len(-(i64)cs[8] * S + p1 - pv, sh); if (dv + l < d1) nd2[1] = dv + l; }
                gu_commit_batch<2>(g, q, dcnt, dlow, dseg, s_hmin, hd2, hub2, nd2, stamp, rot, wmax, t);
                continue;
            }
            // Any other row: the lane's nine arcs in three batches of three, each batch with its
            // heads, costs and the labels of its heads in flight together; a batch in which no lane
            // of the wave holds an arc with residual capacity costs nothing.
            bool any9 = false;
#pragma unroll
            for (int j = 0; j < 9; ++j) any9 |= rc9[j] > 0;
            if (!any9) t.c[kTallyGuSkips] += mine;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                bool live[3];
                unsigned arc[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) { live[j] = rc9[3 * b + j] > 0; arc[j] = arc9(a0, at, 3 * b + j); }
                if (!__ballot(live[0] || live[1] || live[2])) continue;
                int hd3[3], cs3[3], hub3[3];
                i64 nd3[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) { hd3[j] = g.head[live[j] ? arc[j] : h]; cs3[j] = g.cost[live[j] ? arc[j] : h]; }
                i64 pu[3], du[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    pu[j] = ld_sel(live[j], &g.price[hd3[j]], g.spare);
                    du[j] = dist_of(ld_sel(live[j], &g.dist[hd3[j]], g.spare));
                    hub3[j] = ld_sel(live[j], &g.hub_of[hd3[j]], g.spare);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    nd3[j] = kDistInf;
                    if (!live[j]) hub3[j] = -1;
                    if (live[j]) { const i64 l = arc_len(-(i64)cs3[j] * S + pu[j] - pv, sh); if (dv + l < du[j]) nd3[j] = dv + l; }
                }
                gu_commit_batch<3>(g, q, dcnt, dlow, dseg, s_hmin, hd3, hub3, nd3, stamp, rot, wmax, t);
            }
        }
    }
    const int2 *lowl = g.fr_low[src];
    const unsigned nlow_round = (nlow + 63) & ~63u;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < nlow_round; i += gridDim.x * kBlock) {
        const bool have = i < nlow;
        const int2 it = have ? lowl[i] : make_int2(0, 0);
        const int v = it.x & 0x0FFFFFFF;
        const unsigned deg = have ? (unsigned)it.x >> 28 : 0u, r0 = (unsigned)it.y;
        const i64 dv = dist_of(g.dist[v]), pv = g.price[v];  // v = 0 for the padding lanes
        int rc4[kLowMax], hd[kLowMax], cs[kLowMax], hub[kLowMax];
        i64 pu[kLowMax], du[kLowMax], nd[kLowMax];
#pragma unroll
        for (unsigned j = 0; j < kLowMax; ++j) {  // entries past the row's end are read (the arrays have spare entries) and dropped
            const bool ok = j < deg;
            const int rcj = o.rcap[r0 + j], hdj = g.head[r0 + j], csj = g.cost[r0 + j];
            rc4[j] = ok ? rcj : 0; hd[j] = ok ? hdj : 0; cs[j] = ok ? csj : 0;
        }
        t.c[kTallyGuScans] += deg;
#pragma unroll
        for (unsigned j = 0; j < kLowMax; ++j) {
            const bool ok = rc4[j] > 0;
            pu[j] = ld_sel(ok, &g.price[hd[j]], g.spare); du[j] = dist_of(ld_sel(ok, &g.dist[hd[j]], g.spare));
            const int hb = ld_sel(ok, &g.hub_of[hd[j]], g.spare);
            hub[j] = ok ? hb : -1;
        }
#pragma unroll
        for (unsigned j = 0; j < kLowMax; ++j) {
            nd[j] = kDistInf;
            if (rc4[j] > 0) {
                const i64 l = arc_len(-(i64)cs[j] * S + pu[j] - pv, sh);
                if (dv + l < du[j]) nd[j] = dv + l;
            }
        }
        gu_commit_batch<(int)kLowMax>(g, q, dcnt, dlow, dseg, s_hmin, hd, hub, nd, stamp, rot, wmax, t);
    }
    __syncthreads();  // hub minima of the whole workgroup are in LDS
    // The end of the sweep is a handful of independent read-modify-writes (hub distance, the two
    // queue counters, the largest-distance slot, the work counters).  Their first halves are all
    // issued before any result is used, so the wave waits one round trip for them, not five.
    const bool hub_cand = my_hub_vertex >= 0 && s_hmin[threadIdx.x] < kDistInf;
    const i64 hub_nd = hub_cand ? s_hmin[threadIdx.x] : kDistInf;
    // one check per hub and workgroup, against a fresh value (most workgroups arrive after
    // the first few have lowered the hub's distance and can skip the same-address atomic)
    const i64 hub_seen = hub_cand ? __hip_atomic_load(&g.dist[my_hub_vertex], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the staged queue entries are in LDS
    unsigned base_low = 0, base_seg = 0;
    if (lane == 0 && q.nlow) base_low = aadd32(&dcnt[kSlotLow], q.nlow);
    if (lane == 0 && q.nseg) base_seg = aadd32(&dcnt[kSlotSeg], q.nseg);
    i64 *dmax_slot = g.dmax_part + wave;
    unsigned long long *tslot = g.tally + (size_t)wave * kTallyN;  // this wave's work counters
    i64 dmax_seen = 0;
    unsigned long long t_scans = 0, t_relax = 0, t_iters = 0, t_skips = 0;
    if (lane == 0) { dmax_seen = *dmax_slot; t_scans = tslot[kTallyGuScans]; t_relax = tslot[kTallyGuRelax]; t_iters = tslot[kTallyGuIters]; t_skips = tslot[kTallyGuSkips]; }
    // second halves
    i64 hub_gain = 0;
    if (hub_cand && hub_nd < dist_of(hub_seen)) {
        const int u = my_hub_vertex;
        const i64 old = amin64(&g.dist[u], dist_pack(hub_nd, stamp));
        if (hub_nd < dist_of(old)) {
            t.c[kTallyGuRelax]++;
            hub_gain = hub_nd;
            if ((unsigned)(old & 0xFFFF) != stamp)  // first to lower it in this sweep; hubs are always split vertices
                append_split(dcnt, dseg, u, g.row[u], g.row[u + 1], rot ^ (unsigned)u);  // once per hub and sweep: not worth two registers held through the sweep
        }
    }
    base_low = __shfl(base_low, 0);
    base_seg = __shfl(base_seg, 0);
    for (unsigned i = lane; i < q.nlow; i += 64) dlow[base_low + i] = q.low[i];
    for (unsigned i = lane; i < q.nseg; i += 64) dseg[base_seg + i] = q.seg[i];
    // Largest distance handed out: kept per wave slot (no shared counter: thousands of waves
    // raising one word serialise at the memory side); k_gu_apply folds the slots.
    wmax = wave_max(wmax > hub_gain ? wmax : hub_gain);
    const unsigned n_scans = wave_sum_u32(t.c[kTallyGuScans]), n_relax = wave_sum_u32(t.c[kTallyGuRelax]), n_iters = wave_sum_u32(t.c[kTallyGuIters]);
    const unsigned n_skips = wave_sum_u32(t.c[kTallyGuSkips]);
    if (lane == 0) {
        if (wmax > dmax_seen) *dmax_slot = wmax;
        if (n_skips) tslot[kTallyGuSkips] = t_skips + n_skips;
        if (n_scans) tslot[kTallyGuScans] = t_scans + n_scans;
        if (n_relax) tslot[kTallyGuRelax] = t_relax + n_relax;
        if (n_iters) tslot[kTallyGuIters] = t_iters + n_iters;
    }
}

// dist = 0 at deficits (they seed frontier 0), "infinite" elsewhere.
template <int kVpt>
__device__ __forceinline__ void gu_init_body_t(const DevGraph &g) {
    Ctrl *ct = g.ctrl;
    if (!ct->gu_init_on) return;
    __shared__ AppendLds s_app;
    const unsigned rot = mix32(ct->epoch, 0x5EEDu);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < kSweepWaves; i += gridDim.x * kBlock) g.dmax_part[i] = 0;
    const int span = kBlock * kVpt;
    for (int base = blockIdx.x * span; base < g.n; base += gridDim.x * span) {
        bool def[kVpt];
        int v[kVpt], cls[kVpt];
        unsigned r0[kVpt], r1[kVpt];
#pragma unroll
        for (int i = 0; i < kVpt; ++i) {
            v[i] = base + i * kBlock + (int)threadIdx.x;
            const bool in = v[i] < g.n;
            def[i] = in && g.excess[v[i]] < 0;
            cls[i] = in ? (int)g.cls[v[i]] : 0;
            if (in) g.dist[v[i]] = def[i] ? 0 : dist_pack(kDistInf, 0);
        }
#pragma unroll
        for (int i = 0; i < kVpt; ++i) { r0[i] = def[i] ? g.row[v[i]] : 0; r1[i] = def[i] ? g.row[v[i] + 1] : 0; }
        block_append<kVpt>(ct->fr_cnt[0], g.fr_low[0], g.fr_seg[0], def, v, cls, r0, r1, rot, s_app);
    }
}

__device__ __forceinline__ void gu_init_body(const DevGraph &g) {
    if (g.n <= kVptSmallN) gu_init_body_t<1>(g);
    else gu_init_body_t<2>(g);
}

// price[v] -= eps * dist[v]; vertices that reach no deficit move by one more than the
// largest distance handed out (any value >= the largest final distance keeps every
// residual arc eps-optimal).
__device__ __forceinline__ void gu_apply_body(const DevGraph &g, int n_iters) {
    Ctrl *ct = g.ctrl;
    // Control step between the expansions and the rounds: every workgroup works the decision out
    // from words this kernel does not write; the first thread of the launch records it.
    if (ct->done || ct->error) return;
    bool apply = false, rounds = true, give_up = false;
    if (ct->gu_on) {
        const unsigned *c = ct->fr_cnt[n_iters % 3];
        const int phase = ct->gu_phase[n_iters & 1];
        const bool dry = c[kSlotLow] + c[kSlotSeg] == 0;
        // converged: seen by an expansion already, or the last one was a pull sweep that queued
        // nothing (or found the frontier dry with no pull vertices to ask)
        if (phase == 2 || (dry && (phase == 1 || *g.pull_cnt == 0))) apply = true;
        // the 16-bit sweep stamps are about to repeat: give this update up (leaving the prices
        // alone is always valid) and go on with push/relabel rounds
        else if (ct->gu_stamp_base + 2u * kGuIters > 60000u) give_up = true;
        else rounds = false;  // the frontier is not empty yet: the update goes on in the next replay
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (apply || give_up) { ct->gu_active = 0; ct->need_gu = 0; }
        if (rounds) ct->rounds_on = 1;
        for (int j = 0; j < kCntSlots; ++j) ct->list_cnt[0][cnt_slot(j)] = 0;
    }
    if (!apply) return;
    const i64 eps = ct->eps;
    __shared__ i64 s_dmax[kBlock / 64];
    i64 dm = 0;
    for (int i = threadIdx.x; i < kSweepWaves; i += kBlock) { const i64 x = g.dmax_part[i]; dm = x > dm ? x : dm; }
    dm = wave_max(dm);
    if (lane_id() == 0) s_dmax[threadIdx.x >> 6] = dm;
    __syncthreads();
    dm = 0;
    for (int i = 0; i < kBlock / 64; ++i) dm = s_dmax[i] > dm ? s_dmax[i] : dm;
    const i64 dfar = dm + 1;
    bool bad = false;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < g.n; v += gridDim.x * kBlock) {
        const i64 d = dist_of(g.dist[v]);
        if (d >= kDistInf && g.excess[v] > 0) bad = true;  // supply that reaches no deficit
        const i64 dd = d < kDistInf ? d : dfar;
        if (dd) g.price[v] -= eps * dd;
    }
    if (bad) ct->error = PSN_EINFEASIBLE_DEV;
}

// ------------------------------------------------------------- control state machine
// The rounds of a replay end with the work list the last relabel sweep built: empty means the
// phase is over (the solve too, at epsilon 1), else a global update comes first next time.
// Evaluated at the start of the next replay; the host (or k_group_status) reads the same words
// to see that the solve is over, so no replay is spent on finding out.
__device__ __forceinline__ bool rounds_closed_solve(const Ctrl *ct, int n_rounds) {
    if (ct->done || ct->error || !ct->rounds_on) return false;
    const unsigned *c = ct->list_cnt[n_rounds & 1];
    return c[kSlotLow] + c[kSlotSeg] == 0 && ct->eps_shift == 0;
}
__device__ __forceinline__ void begin_body(const DevGraph &g, int n_rounds) {
    Ctrl *ct = g.ctrl;
    // what the last relabel sweep counted sits on three cache lines: asked for together, ahead of the branches that use it
    const unsigned *const c = ct->list_cnt[n_rounds & 1];
    const unsigned c_low = c[kSlotLow], c_seg = c[kSlotSeg], c_units = c[kSlotUnits], c_ndef = c[kSlotDeficits];
    CTRL_TOGETHER4(c_low, c_seg, c_units, c_ndef);
    if (!ct->done && !ct->error && ct->rounds_on) {  // close the rounds of the previous replay
        const unsigned left = c_low + c_seg;
        // A phase other than the last one need not route its last units of excess: they join what
        // the next phase's saturation sets free (every phase starts by restoring its own
        // epsilon-optimality, whatever pseudoflow it is given).  The long tail of a phase - a
        // handful of units moving one arc per round - is thereby run once, at the end of the last
        // phase, instead of once per phase.  What is left is measured in units of excess at the
        // active vertices, not in work items: a few large vertices that still hold thousands of
        // units (an over-subscribed cluster) are not a tail.
        if (left == 0 || (ct->eps_shift > 0 && c_units <= (unsigned)ct->tail_units)) {
            if (ct->eps_shift == 0) {
                ct->done = 1;
            } else {
                // next phase: epsilon / alpha; a last step that would be less than half a phase is
                // folded into this one (a phase costs about the same whatever its step)
                int nxt = ct->eps_shift > ct->alpha_log2 ? ct->eps_shift - ct->alpha_log2 : 0;
                if (2 * nxt <= ct->alpha_log2) nxt = 0;
                ct->eps_shift = nxt;
                ct->eps = (i64)1 << ct->eps_shift;
                ct->need_sat = 1;
            }
        } else {
            ct->need_gu = 1;
        }
    }
    ct->epoch++;
    ct->sat_on = ct->gu_on = ct->gu_init_on = ct->rounds_on = ct->flip_on = 0;
    if (ct->done || ct->error) return;
    // Which way round this replay works.  A phase starts the way the network is stored; a replay
    // that opens a new global update inside a phase, once epsilon is within mirror_max_shift,
    // changes sides when the other side's open units sit at few vertices compared with the work
    // this side has queued (both counted by the last relabel sweep: work items of this side - a
    // split vertex counts once per segment -, vertices with a deficit).  Pushing from the side
    // where the units are concentrated fans them out; from the other side many units look for
    // the same few vertices and get through one per update.  Measured over three sizes of
    // scheduling networks and several seeds each: 2 deficit vertices per work item, from 64
    // cost units of epsilon down, is the best of the settings tried (about a sixth less time).
    {
        int want = ct->mirror;
        if (ct->need_sat) {
            want = 0;
        } else if (ct->need_gu && !ct->gu_active && ct->eps_shift <= ct->mirror_max_shift && ct->mirror_ratio > 0) {
            const unsigned long long items = (unsigned long long)c_low + c_seg, ndef = c_ndef;
            if (ndef > 0 && ndef * (unsigned long long)ct->mirror_ratio <= items * 4ull) want = !ct->mirror;
        }
        if (want != ct->mirror) { ct->mirror = want; ct->flip_on = 1; }
    }
    if (ct->need_sat) { ct->sat_on = 1; ct->need_sat = 0; ct->need_gu = 1; ct->tally_refines++; }
    if (ct->need_gu) {
        ct->gu_on = 1;
        if (!ct->gu_active) {
            ct->gu_active = 1;
            ct->gu_init_on = 1;
            ct->tally_gus++;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) ct->fr_cnt[i][j] = 0;
            ct->gu_phase[0] = ct->gu_phase[1] = 0;
            ct->gu_stamp_base = 0;
        } else {
            ct->gu_stamp_base += kGuIters;  // sweep stamps stay unique within one update
        }
    }
    if (ct->mirror) ct->tally_mirror++;
    if (ct->eps_shift == 0) ct->last_phase_replays++;
}

// Negates prices and excesses in place when the orientation changes (or, with `unflip`, when a
// finished solve was left mirrored), and forgets the hub segment bounds: they bound keys of the
// other orientation.
__device__ __forceinline__ void flip_body(const DevGraph &g, int unflip) {
    const Ctrl *ct = g.ctrl;
    if (unflip ? !ct->mirror : !ct->flip_on) return;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < g.n; v += gridDim.x * kBlock) {
        g.price[v] = -g.price[v];
        g.excess[v] = -g.excess[v];
    }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n_hub_segs; i += gridDim.x * kBlock) {
        g.seg_scan[i] = kInf; g.seg_add[0][i] = kNegInf; g.seg_add[1][i] = kNegInf;
    }
}

// ------------------------------------------------------------ setup / readout kernels
__device__ __forceinline__ void reset_body(const DevGraph &g, const i64 *supply, int keep_flow, int zero_tally) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        if (!keep_flow) g.cap[a] = g.ocap[a];
    }
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < g.n; v += gridDim.x * kBlock) {
        if (!keep_flow) { g.price[v] = 0; g.excess[v] = supply[v]; }
        g.cand[v] = kInf;
    }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n_hub_segs; i += gridDim.x * kBlock) {
        g.seg_scan[i] = kInf; g.seg_add[0][i] = kNegInf; g.seg_add[1][i] = kNegInf;
    }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < kHubShards * kMaxHubs; i += gridDim.x * kBlock) { g.hub_acc_add[i] = 0; g.hub_acc_ext[i] = kNegInf; }
    if (zero_tally)
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < kSweepWaves * kTallyN; i += gridDim.x * kBlock) g.tally[i] = 0;
}
// rcap mirrors cap[rev]; excess from scratch for a kept flow: supply - outflow + inflow,
// with outflow(a) = ocap[a] - cap[a] counted on arcs where it is positive.
__device__ __forceinline__ void mirror_body(const DevGraph &g, const i64 *supply, int recompute_excess) {
    if (recompute_excess)
        for (int v = blockIdx.x * kBlock + threadIdx.x; v < g.n; v += gridDim.x * kBlock) g.excess[v] = supply[v];
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) g.rcap[a] = g.cap[g.rev[a]];
}
__device__ __forceinline__ void excess_from_flow_body(const DevGraph &g) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        const i64 f = (i64)g.ocap[a] - g.cap[a];
        if (f > 0) { aadd64(&g.excess[g.tail[a]], -f); aadd64(&g.excess[g.head[a]], f); }
    }
}
__device__ __forceinline__ void total_cost_body(const DevGraph &g, i64 *out) {
    i64 s = 0;
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        const i64 f = (i64)g.ocap[a] - g.cap[a];
        if (f > 0) s += f * g.cost[a];
    }
    s = (i64)wave_sum_u64((unsigned long long)s);
    if (lane_id() == 0 && s) aadd64(out, s);
}
__global__ __launch_bounds__(kBlock) void k_reset(DevGraph g, const i64 *supply, int keep_flow) { reset_body(g, supply, keep_flow, 0); }
__global__ __launch_bounds__(kBlock) void k_mirror(DevGraph g, const i64 *supply, int recompute_excess) { mirror_body(g, supply, recompute_excess); }
__global__ __launch_bounds__(kBlock) void k_excess_from_flow(DevGraph g) { excess_from_flow_body(g); }
__global__ __launch_bounds__(kBlock) void k_total_cost(DevGraph g, i64 *out) { total_cost_body(g, out); }
// The same for the members of a lockstep group, one launch for all of them (blockIdx.y picks the
// member; what differs per member comes from its GroupAux record).
__global__ __launch_bounds__(kBlock) void k_reset_group(const DevGraph *__restrict__ gs, const GroupAux *__restrict__ aux) {
    reset_body(gs[blockIdx.y], aux[blockIdx.y].supply, aux[blockIdx.y].warm, 1);
}
__global__ __launch_bounds__(kBlock) void k_mirror_group(const DevGraph *__restrict__ gs, const GroupAux *__restrict__ aux) {
    mirror_body(gs[blockIdx.y], aux[blockIdx.y].supply, aux[blockIdx.y].warm);
}
__global__ __launch_bounds__(kBlock) void k_excess_from_flow_group(const DevGraph *__restrict__ gs, const GroupAux *__restrict__ aux) {
    if (aux[blockIdx.y].warm) excess_from_flow_body(gs[blockIdx.y]);
}
__global__ __launch_bounds__(kBlock) void k_total_cost_group(const DevGraph *__restrict__ gs, const GroupAux *__restrict__ aux) {
    total_cost_body(gs[blockIdx.y], aux[blockIdx.y].scalar);
}
__global__ __launch_bounds__(kBlock) void k_flow(DevGraph g, int *flow) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) flow[a] = g.ocap[a] - g.cap[a];
}
__device__ __forceinline__ void tally_sum_body(const DevGraph &g, unsigned long long *out, int nslots) {
    // one wave per counter
    const int w = threadIdx.x >> 6;
    if (w >= kTallyN) return;
    unsigned long long s = 0;
    for (int i = lane_id(); i < nslots; i += 64) s += g.tally[(size_t)i * kTallyN + w];
    s = wave_sum_u64(s);
    if (lane_id() == 0) out[w] = s;
}
__global__ void k_tally_sum(DevGraph g, unsigned long long *out, int nslots) { tally_sum_body(g, out, nslots); }
// work counters and control block of every member of a group, into the group's contiguous read-back blocks
__global__ void k_tally_sum_group(const DevGraph *__restrict__ gs, const GroupAux *__restrict__ aux, int nslots) {
    const DevGraph &g = gs[blockIdx.y];
    if (threadIdx.x == 0) *aux[blockIdx.y].ctrl_out = *g.ctrl;
    tally_sum_body(g, aux[blockIdx.y].tally_out, nslots);
}
// the control blocks the members of a group start from, out of the group's upload block
__global__ void k_ctrl_scatter_group(const DevGraph *__restrict__ gs, const Ctrl *__restrict__ init, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) *gs[i].ctrl = init[i];
}

// ------------------------------------------------------------- arc-cost model (K1)
// Slot arcs (task class e -> machine m, slot k), int64 arithmetic throughout:
//   fit  = min((cap_cpu-used_cpu)/req_cpu, (cap_mem-used_mem)/req_mem)
//   cap  = k < fit;  cost = 1000*(used_cpu+(k+1)*req_cpu)/cap_cpu + 1000*(used_mem+(k+1)*req_mem)/cap_mem
// The numerators grow linearly with k, so a thread divides twice per dimension (first
// numerator, step) and carries quotient and remainder along its K slots.  Two kinds of
// threads, both writing K consecutive entries next to their neighbours' (coalesced): one per
// (e, m) for the forward arcs in the class rows, one per (m, e) for the paired reverse arcs,
// which sit class-major, slot-minor at the start of the machine rows.  A kept flow is read
// from the thread's own entries (forward: original minus residual capacity; reverse: its
// residual capacity) and cut to the new capacity.
struct SlotCosts {
    i64 q, r, dq, dr, c;  // cost term of the current slot = q; next: q += dq + carry
    __device__ SlotCosts(i64 used, i64 req, i64 cap) : c(cap) {
        const i64 n0 = 1000 * (used + req), st = 1000 * req;
        q = n0 / cap; r = n0 - q * cap;
        dq = st / cap; dr = st - dq * cap;
    }
    __device__ void next() {
        q += dq; r += dr;
        if (r >= c) { r -= c; ++q; }
    }
};
// Layout check of the slot block, run as a pass of its own before the cost kernel writes
// anything: the K reverse arcs of every (machine, class) pair must sit class-major, slot-minor
// at the start of the machine's row and pair up with the K forward arcs at ec_arc_base[e] + m * K.
// bad[0] counts the pairs that do not; the cost kernel returns without a write when it is nonzero.
__global__ __launch_bounds__(kBlock) void k_check_slot_layout(DevGraph g, int E, int N, int K, const i64 *ec_arc_base, unsigned *bad) {
    const long long groups = (long long)E * N;
    unsigned mine = 0;
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < groups; i += (long long)gridDim.x * kBlock) {
        const int e = (int)(i % E), m = (int)(i / E);
        const unsigned a0 = (unsigned)(ec_arc_base[e] + (i64)m * K);
        const unsigned r0 = g.row[1 + m] + (unsigned)e * (unsigned)K;
        bool ok = r0 + (unsigned)K <= g.row[2 + m] && a0 + (unsigned)K <= g.m2;
        for (int k = 0; ok && k < K; ++k) ok = (unsigned)g.rev[r0 + k] == a0 + (unsigned)k && (unsigned)g.rev[a0 + k] == r0 + (unsigned)k;
        mine += !ok;
    }
    if (__ballot(mine != 0)) {
        mine = wave_sum_u32(mine);
        if (lane_id() == 0) aadd32(bad, mine);
    }
}
__global__ __launch_bounds__(kBlock) void k_eval_costs(DevGraph g, int E, int N, int K, const i64 *cap_cpu, const i64 *cap_mem,
                                                       const i64 *used_cpu, const i64 *used_mem, const i64 *req_cpu,
                                                       const i64 *req_mem, const i64 *ec_arc_base, int *mcost, int *mocap,
                                                       int keep_flow, const unsigned *allow, int allow_words, const unsigned *bad_layout) {
    const long long groups = (long long)E * N;
    if (*bad_layout) return;  // the layout pass found pairs that do not line up: nothing is written
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < 2 * groups; i += (long long)gridDim.x * kBlock) {
        const bool fwd = i < groups;
        const long long gi = fwd ? i : i - groups;
        // forward threads run machine-fastest inside a class, reverse threads class-fastest inside a machine
        const int e = fwd ? (int)(gi / N) : (int)(gi % E), m = fwd ? (int)(gi % N) : (int)(gi / E);
        const i64 cc = cap_cpu[m], cm = cap_mem[m], uc = used_cpu[m], um = used_mem[m];
        const i64 rc = req_cpu[e], rm = req_mem[e];
        const i64 fc = (cc - uc) / rc, fm = (cm - um) / rm;
        i64 fit = fc < fm ? fc : fm;
        if (allow && !((allow[(size_t)e * allow_words + (m >> 5)] >> (m & 31)) & 1u)) fit = 0;  // selector mismatch: no capacity
        const unsigned a0 = (unsigned)(ec_arc_base[e] + (i64)m * K);
        const unsigned r0 = g.row[1 + m] + (unsigned)e * (unsigned)K;
        SlotCosts sc(uc, rc, cc), sm(um, rm, cm);
        const unsigned b0 = fwd ? a0 : r0;  // first of this thread's K entries
        if (K == 8) {  // the usual shape: eight slots, written as two 16-byte stores per array
            int c[8], cp[8], f[8], oc[8], old[8];
            if (keep_flow) {
                const int4u x0 = *reinterpret_cast<const int4u *>(g.cap + b0), x1 = *reinterpret_cast<const int4u *>(g.cap + b0 + 4);
                old[0] = x0.x; old[1] = x0.y; old[2] = x0.z; old[3] = x0.w; old[4] = x1.x; old[5] = x1.y; old[6] = x1.z; old[7] = x1.w;
                if (fwd) {
                    const int4u y0 = *reinterpret_cast<const int4u *>(mocap + b0), y1 = *reinterpret_cast<const int4u *>(mocap + b0 + 4);
                    oc[0] = y0.x; oc[1] = y0.y; oc[2] = y0.z; oc[3] = y0.w; oc[4] = y1.x; oc[5] = y1.y; oc[6] = y1.z; oc[7] = y1.w;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c[k] = (int)(sc.q + sm.q);
                cp[k] = k < fit ? 1 : 0;
                f[k] = 0;  // flow kept on the arc, cut to the new capacity
                if (keep_flow) { f[k] = fwd ? oc[k] - old[k] : old[k]; f[k] = f[k] < 0 ? 0 : (f[k] > cp[k] ? cp[k] : f[k]); }
                sc.next(); sm.next();
            }
            int4u v0, v1;
#define PUT8(dst, expr)                                                                          \
            { int k = 0; v0.x = (expr); k = 1; v0.y = (expr); k = 2; v0.z = (expr); k = 3; v0.w = (expr);   \
              k = 4; v1.x = (expr); k = 5; v1.y = (expr); k = 6; v1.z = (expr); k = 7; v1.w = (expr);       \
              *reinterpret_cast<int4u *>((dst) + b0) = v0; *reinterpret_cast<int4u *>((dst) + b0 + 4) = v1; }
            if (fwd) {
                PUT8(mcost, c[k]) PUT8(mocap, cp[k]) PUT8(g.cap, cp[k] - f[k]) PUT8(g.rcap, f[k])
            } else {
                PUT8(mcost, -c[k]) PUT8(mocap, 0 * k) PUT8(g.cap, f[k]) PUT8(g.rcap, cp[k] - f[k])
            }
#undef PUT8
            continue;
        }
        for (int k = 0; k < K; ++k) {
            const int c = (int)(sc.q + sm.q);
            const int cp = k < fit ? 1 : 0;
            if (fwd) {
                const unsigned a = a0 + k;
                int f = 0;
                if (keep_flow) { f = mocap[a] - g.cap[a]; f = f < 0 ? 0 : (f > cp ? cp : f); }
                mcost[a] = c; mocap[a] = cp; g.cap[a] = cp - f; g.rcap[a] = f;
            } else {
                const unsigned r = r0 + k;
                int f = 0;
                if (keep_flow) { f = g.cap[r]; f = f < 0 ? 0 : (f > cp ? cp : f); }
                mcost[r] = -c; mocap[r] = 0; g.cap[r] = f; g.rcap[r] = cp - f;
            }
            sc.next(); sm.next();
        }
    }
}

// ----------------------------------------------------- placement extraction (K5)
// Tasks of one class are interchangeable, so the class's routed tasks (in arc order) are
// matched to its units of outgoing flow (in arc order).  One workgroup per class:
// pass 1 numbers the units of flow on slot arcs, pass 2 numbers the routed tasks.
constexpr int kPlaceBlock = 1024;  // threads per class in k_placements
__device__ __forceinline__ unsigned block_excl_scan(unsigned x, unsigned *lds, unsigned &total) {
    const int lane = lane_id(), w = threadIdx.x >> 6;
    unsigned incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    unsigned off = 0, tot = 0;
    for (int i = 0; i < kPlaceBlock / 64; ++i) { if (i < w) off += lds[i]; tot += lds[i]; }
    __syncthreads();
    total = tot;
    return off + incl - x;
}
// bad[0] counts arcs whose head is not where the layout says it should be (a slot arc that does
// not end at a machine, a reverse arc that does not end at a task); nothing is written for them.
__global__ __launch_bounds__(kPlaceBlock) void k_placements(DevGraph g, int N, int E, int K, int v_task0, int n_tasks, int *slot_machine,
                                                            int slot_stride, int *place, unsigned *bad) {
    __shared__ unsigned lds[kPlaceBlock / 64];
    const int e = blockIdx.x;
    if (e >= E) return;
    const int u = 1 + N + e;
    const unsigned r0 = g.row[u], r1 = g.row[u + 1];
    const unsigned nslot = (unsigned)N * K;
    const unsigned t1 = r1 - nslot;  // [r0,t1): reverse arcs to tasks, [t1,r1): slot arcs
    int *sm = slot_machine + (size_t)e * slot_stride;
    unsigned run = 0;
    for (unsigned b = t1; b < r1; b += kPlaceBlock * 4) {  // four consecutive slot arcs per thread
        const unsigned a0 = b + threadIdx.x * 4;
        int fl[4] = {0, 0, 0, 0};
        if (a0 + 3 < r1) {
            const int4u oc = *reinterpret_cast<const int4u *>(g.ocap + a0), cp = *reinterpret_cast<const int4u *>(g.cap + a0);
            fl[0] = oc.x - cp.x; fl[1] = oc.y - cp.y; fl[2] = oc.z - cp.z; fl[3] = oc.w - cp.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (a0 + j < r1) fl[j] = g.ocap[a0 + j] - g.cap[a0 + j];
        }
        unsigned f = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { fl[j] = fl[j] > 0 ? fl[j] : 0; f += (unsigned)fl[j]; }
        unsigned tot;
        unsigned pos = run + block_excl_scan(f, lds, tot);
        if (f) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (fl[j]) {
                    const int mch = g.head[a0 + j] - 1;
                    if ((unsigned)mch >= (unsigned)N) { aadd32(bad, 1u); pos += (unsigned)fl[j]; continue; }
                    for (int x = 0; x < fl[j]; ++x, ++pos) if (pos < (unsigned)slot_stride) sm[pos] = mch;
                }
        }
        run += tot;
    }
    __syncthreads();
    const unsigned units = run;
    run = 0;
    for (unsigned b = r0; b < t1; b += kPlaceBlock) {
        const unsigned a = b + threadIdx.x;
        unsigned f = 0;
        if (a < t1) f = g.cap[a] > 0 ? 1u : 0u;  // residual on the reverse arc == task routed here
        unsigned tot;
        unsigned pos = run + block_excl_scan(f, lds, tot);
        if (f) {
            const int tk = g.head[a] - v_task0;
            if ((unsigned)tk >= (unsigned)n_tasks) aadd32(bad, 1u);
            else place[tk] = pos < units && pos < (unsigned)slot_stride ? sm[pos] : -1;
        }
        run += tot;
    }
}
__global__ __launch_bounds__(kBlock) void k_fill_i32(int *p, int v, long long n) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) p[i] = v;
}
__global__ __launch_bounds__(kBlock) void k_scatter_caps(DevGraph g, long long count, const i64 *idx, const int *ncap, int *mocap) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const unsigned a = (unsigned)idx[i];
        const int flow = mocap[a] - g.cap[a];
        mocap[a] = ncap[i];
        g.cap[a] = ncap[i] - flow;  // may go negative: repaired by k_clamp_flow
    }
}
// New cost of an arc (and its negation on the paired arc); flow and prices stay, a warm start
// repairs whatever optimality conditions the change broke.
__global__ __launch_bounds__(kBlock) void k_scatter_costs(DevGraph g, long long count, const i64 *idx, const int *ncost, int *mcost) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const unsigned a = (unsigned)idx[i];
        mcost[a] = ncost[i];
        mcost[g.rev[a]] = -ncost[i];
    }
}
// After capacities shrank under a kept flow: cancel the flow that no longer fits.
__global__ __launch_bounds__(kBlock) void k_clamp_flow(DevGraph g) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        const int c = g.cap[a];
        if (c < 0) { g.cap[a] = 0; g.cap[g.rev[a]] += c; }
    }
}
// Overwrite whole residual arcs (head, capacity, cost, pairing); any flow on them is dropped.
__global__ __launch_bounds__(kBlock) void k_patch_arcs(DevGraph g, long long count, const i64 *idx, const int *nhead, const int *ncap,
                                                       const int *ncost, const int *nrev, int *mhead, int *mcost, int *mrev, int *mocap) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const unsigned a = (unsigned)idx[i];
        mhead[a] = nhead[i]; mcost[a] = ncost[i]; mrev[a] = nrev[i]; mocap[a] = ncap[i]; g.cap[a] = ncap[i];
    }
}
// Patched arcs must still pair up: rev is an involution onto the reverse arc with the
// negated cost; an arc may pair with itself only as an unused placeholder (self-loop, no
// capacity, no cost).  bad[0] counts violations.
__global__ __launch_bounds__(kBlock) void k_check_patched(DevGraph g, long long count, const i64 *idx, unsigned *bad) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const unsigned a = (unsigned)idx[i];
        const int r = g.rev[a];
        bool ok = r >= 0 && (unsigned)r < g.m2 && g.head[a] >= 0 && g.head[a] < g.n;
        if (ok) {
            if ((unsigned)r == a) ok = g.head[a] == g.tail[a] && g.ocap[a] == 0 && g.cost[a] == 0;
            else ok = (unsigned)g.rev[r] == a && g.head[r] == g.tail[a] && g.head[a] == g.tail[r] && g.cost[r] == -g.cost[a] &&
                      g.ocap[a] >= 0 && g.ocap[r] >= 0;
        }
        if (!ok) aadd32(bad, 1u);
    }
}
// A vertex that no residual arc enters (a source that has not sent anything yet) can take any
// price without breaking optimality conditions of other arcs; give it the lowest price that
// makes all of its own residual arcs non-negative, so a warm start sees no violation there.
__global__ __launch_bounds__(kBlock) void k_settle_sources(DevGraph g, long long count, const int *vtx, unsigned *skipped) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const int v = vtx[i];
        bool entered = false;
        i64 best = kNegInf;
        for (unsigned a = g.row[v]; a < g.row[v + 1]; ++a) {
            if (g.cap[g.rev[a]] > 0 && (unsigned)g.rev[a] != a) entered = true;
            if (g.cap[a] > 0) { const i64 key = g.price[g.head[a]] - (i64)g.cost[a] * g.scale; best = key > best ? key : best; }
        }
        if (entered) aadd32(skipped, 1u);
        else if (best != kNegInf) g.price[v] = best;
    }
}
__global__ __launch_bounds__(kBlock) void k_scatter_supply(long long count, const int *vtx, const i64 *val, i64 *supply) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) supply[vtx[i]] = val[i];
}


// ------------------------------------------------------------------ kernel entry points
// Every stage of the replay comes in two forms: one network handed over by value, and a group
// of networks in lockstep (blockIdx.y picks the network from a table in device memory; the
// networks share nothing, each follows its own control block, and one that has finished or
// is waiting for another stage returns at once).  The group form divides the number of
// launches per network by the group size, which is what bounds a batch of small networks.
__global__ __launch_bounds__(kBlock) void k_push(DevGraph g, int k) { push_body(g, k); }
__global__ __launch_bounds__(kBlock) void k_push_group(const DevGraph *__restrict__ gs, int k) { push_body(gs[blockIdx.y], k); }
__global__ __launch_bounds__(kBlock) void k_relabel_compact(DevGraph g, int k, int do_relabel) { relabel_compact_body(g, k, do_relabel); }
__global__ __launch_bounds__(kBlock) void k_relabel_compact_group(const DevGraph *__restrict__ gs, int k, int do_relabel) { relabel_compact_body_t<2>(gs[blockIdx.y], k, do_relabel); }
__global__ __launch_bounds__(kBlock) void k_saturate(DevGraph g) { saturate_body(g); }
__global__ __launch_bounds__(kBlock) void k_saturate_group(const DevGraph *__restrict__ gs) { saturate_body(gs[blockIdx.y]); }
__global__ __launch_bounds__(kBlock, 4) void k_gu_expand(DevGraph g, int k) { gu_expand_body(g, k); }
__global__ __launch_bounds__(kBlock, 4) void k_gu_expand_group(const DevGraph *__restrict__ gs, int k) { gu_expand_body(gs[blockIdx.y], k); }
__global__ __launch_bounds__(kBlock) void k_gu_init(DevGraph g) { gu_init_body(g); }
__global__ __launch_bounds__(kBlock) void k_gu_init_group(const DevGraph *__restrict__ gs) { gu_init_body_t<2>(gs[blockIdx.y]); }
__global__ __launch_bounds__(kBlock) void k_gu_apply(DevGraph g, int n_iters) { gu_apply_body(g, n_iters); }
__global__ __launch_bounds__(kBlock) void k_gu_apply_group(const DevGraph *__restrict__ gs, int n_iters) { gu_apply_body(gs[blockIdx.y], n_iters); }
__global__ void k_begin(DevGraph g, int n_rounds) { begin_body(g, n_rounds); }
__global__ __launch_bounds__(kBlock) void k_flip(DevGraph g, int unflip) { flip_body(g, unflip); }
__global__ __launch_bounds__(kBlock) void k_flip_group(const DevGraph *__restrict__ gs, int unflip) { flip_body(gs[blockIdx.y], unflip); }
__global__ void k_begin_group(const DevGraph *__restrict__ gs, int n_rounds) { begin_body(gs[blockIdx.y], n_rounds); }
// done / error of every network of a group, side by side for one read-back per replay
__global__ void k_group_status(const DevGraph *gs, int count, int n_rounds, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { const Ctrl *ct = gs[i].ctrl; status[2 * i] = ct->done || rounds_closed_solve(ct, n_rounds); status[2 * i + 1] = ct->error; }
}

}  // namespace psn
