// libpsn device code: cost-scaling push-relabel min-cost flow for gfx950 (MI355X).
//
// Everything here is wave-centric (64 lanes): a work item is a vertex (one lane for
// degree <= 4, one wave up to degree 1024) or a 512-arc segment of a high-degree
// vertex handled by one wave.  No kernel needs a workgroup barrier.
//
// The algorithm is bulk-synchronous so that the result does not depend on timing:
//   push phase     prices frozen; every active vertex pushes on admissible arcs
//                  (reduced cost < 0).  A residual capacity is written only by the
//                  wave that owns the arc's tail (an arc and its reverse cannot both be
//                  admissible under one price vector), excess moves by 64-bit atomics.
//   relabel phase  residual graph frozen; an active vertex whose scan found no
//                  admissible residual arc gets price = max(price[v] - cost) - eps.
// Epsilon is a power of two; costs are multiplied by S = n + 1, so eps = 1 is optimal.
// All loop control (epsilon schedule, when to saturate, when to run a global price
// update, termination) lives in a device-resident Ctrl block that tiny single-thread
// kernels advance, so one hipGraph can be replayed until Ctrl.done is set.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "psn_device.h"

namespace psn {

static constexpr i64 kInf = INT64_MAX;
static constexpr i64 kNegInf = INT64_MIN;
static constexpr i64 kDistInf = INT64_MAX / 4;

typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));

// ------------------------------------------------------------------ wave primitives
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

// Per-thread work counters; folded into a per-block slot (no atomics: block b owns
// slot b in every launch, and launches on one stream are ordered).
struct Tally { unsigned long long c[kTallyN]; __device__ Tally() { for (int i = 0; i < kTallyN; ++i) c[i] = 0; } };
__device__ __forceinline__ void tally_flush(const DevGraph &g, Tally &t) {
    bool any = false;
#pragma unroll
    for (int i = 0; i < kTallyN; ++i) { t.c[i] = wave_sum_u64(t.c[i]); any |= t.c[i] != 0; }
    if (lane_id() == 0 && any) {
        unsigned long long *slot = g.tally + ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * kTallyN;
#pragma unroll
        for (int i = 0; i < kTallyN; ++i) if (t.c[i]) slot[i] += t.c[i];
    }
}

// ------------------------------------------------------- active-list compaction (K4)
// Appends the vertices flagged by `take` to the class lists: one ballot per class,
// one atomic per wave per non-empty class, slots by popcount prefix.
__device__ __forceinline__ void wave_append(const DevGraph &g, unsigned *cnt, int *lowl, int *medl, int2 *highl,
                                            bool take, int v, int cls) {
    const unsigned long long lt = lanemask_lt();
    unsigned long long m = __ballot(take && cls == kClsLow);
    if (m) {
        unsigned base = 0;
        if (lane_id() == __ffsll((long long)m) - 1) base = aadd32(&cnt[kClsLow], (unsigned)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (take && cls == kClsLow) lowl[base + __popcll(m & lt)] = v;
    }
    m = __ballot(take && cls == kClsMed);
    if (m) {
        unsigned base = 0;
        if (lane_id() == __ffsll((long long)m) - 1) base = aadd32(&cnt[kClsMed], (unsigned)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (take && cls == kClsMed) medl[base + __popcll(m & lt)] = v;
    }
    if (take && cls == kClsHigh) {  // rare: a handful of vertices own most arcs
        unsigned deg = g.row[v + 1] - g.row[v];
        unsigned nseg = (deg + kWSeg - 1) / kWSeg;
        unsigned base = aadd32(&cnt[kClsHigh], nseg);
        for (unsigned s = 0; s < nseg; ++s) highl[base + s] = make_int2(v, (int)s);
    }
}

// --------------------------------------------------------------------------- pushes
// One unit of push bookkeeping: the caller owns arc a (tail u); r = rev[a] is only
// written here because (v,u) cannot be admissible while (u,v) is.
__device__ __forceinline__ void do_push(const DevGraph &g, unsigned a, int v, int cap_before, i64 d, i64 pu, i64 cS) {
    const int r = g.rev[a];
    const int na = cap_before - (int)d;
    const int nr = g.cap[r] + (int)d;
    g.cap[a] = na;
    g.cap[r] = nr;
    g.rcap[a] = nr;
    g.rcap[r] = na;
    aadd64(&g.excess[v], d);
    amax64(&g.cand[v], pu + cS);  // key of the new residual arc (v,u): price[u] - cost(v,u)
}

// Low-degree vertex: one lane walks its arcs.
__device__ __forceinline__ void push_low(const DevGraph &g, int u, i64 S, Tally &t) {
    const unsigned r0 = g.row[u], r1 = g.row[u + 1];
    const i64 pu = g.price[u];
    const i64 e0 = g.excess[u];
    i64 e = e0, best = kNegInf;
    t.c[kTallyVisits]++;
    for (unsigned a = r0; a < r1; ++a) {
        int cp = g.cap[a];
        t.c[kTallyScans]++;
        if (cp <= 0) continue;
        const int v = g.head[a];
        const i64 cS = (i64)g.cost[a] * S;
        const i64 key = g.price[v] - cS;
        if (key > pu && e > 0) {
            i64 d = e < cp ? e : cp;
            do_push(g, a, v, cp, d, pu, cS);
            e -= d; cp -= (int)d;
            t.c[kTallyPushes]++;
        }
        if (cp > 0 && key > best) best = key;
    }
    if (best > kNegInf) amax64(&g.cand[u], best);
    if (e != e0) aadd64(&g.excess[u], e - e0);
}

// Medium-degree vertex: one wave, 64 arcs per step, chunk order rotated per round.
__device__ __forceinline__ void push_med(const DevGraph &g, int u, i64 S, unsigned salt, Tally &t) {
    const int lane = lane_id();
    const unsigned r0 = g.row[u], r1 = g.row[u + 1];
    const unsigned nch = (r1 - r0 + 63) >> 6;
    const i64 pu = g.price[u];
    const i64 e0 = g.excess[u];
    i64 e = e0, best = kNegInf;
    bool incomplete = false;
    if (lane == 0) t.c[kTallyVisits]++;
    unsigned c = nch > 1 ? mix32((unsigned)u, salt) % nch : 0;
    for (unsigned ci = 0; ci < nch; ++ci) {
        const unsigned a = r0 + (c << 6) + lane;
        if (++c == nch) c = 0;
        const bool valid = a < r1;
        int cp = valid ? g.cap[a] : 0;
        if (valid) t.c[kTallyScans]++;
        int v = 0; i64 cS = 0, key = kNegInf;
        if (cp > 0) { v = g.head[a]; cS = (i64)g.cost[a] * S; key = g.price[v] - cS; }
        const bool adm = cp > 0 && key > pu;
        if (__ballot(adm)) {
            const i64 want = adm ? cp : 0;
            const i64 incl = wave_incl_scan(want);
            const i64 total = __shfl(incl, 63);
            i64 give = e - (incl - want);
            give = give < 0 ? 0 : (give > want ? want : give);
            if (give > 0) { do_push(g, a, v, cp, give, pu, cS); cp -= (int)give; t.c[kTallyPushes]++; }
            e -= total < e ? total : e;
        }
        if (cp > 0 && key > best) best = key;
        if (e == 0 && ci + 1 < nch) { incomplete = true; break; }
    }
    best = wave_max(best);
    if (lane == 0) {
        if (incomplete) best = kInf;  // arcs not looked at: no relabel this round
        if (best > kNegInf) amax64(&g.cand[u], best);
        if (e != e0) aadd64(&g.excess[u], e - e0);
    }
}

// High-degree vertex, one 512-arc segment per wave.  Several waves work on one vertex,
// so each claims what it wants to push from excess[u] and hands back what it cannot use.
__device__ __forceinline__ void push_high(const DevGraph &g, int u, unsigned seg, i64 S, Tally &t) {
    const int lane = lane_id();
    const unsigned r0 = g.row[u] + seg * kWSeg;
    const unsigned r1 = min(g.row[u + 1], r0 + kWSeg);
    const i64 pu = g.price[u];
    if (lane == 0) t.c[kTallyVisits]++;
    i64 avail = __hip_atomic_load(&g.excess[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (avail <= 0) {  // drained by the other segments: this one stays unscanned
        if (lane == 0) amax64(&g.cand[u], kInf);
        return;
    }
    i64 best = kNegInf;
    for (unsigned h = 0; h < kWSeg; h += 256) {
        const unsigned a0 = r0 + h + lane * 4;
        int cp[4], hd[4], cs[4];
        if (a0 + 3 < r1) {
            int4u x = *reinterpret_cast<const int4u *>(g.cap + a0);
            int4u y = *reinterpret_cast<const int4u *>(g.head + a0);
            int4u z = *reinterpret_cast<const int4u *>(g.cost + a0);
            cp[0] = x.x; cp[1] = x.y; cp[2] = x.z; cp[3] = x.w;
            hd[0] = y.x; hd[1] = y.y; hd[2] = y.z; hd[3] = y.w;
            cs[0] = z.x; cs[1] = z.y; cs[2] = z.z; cs[3] = z.w;
            t.c[kTallyScans] += 4;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = a0 + j < r1;
                cp[j] = ok ? g.cap[a0 + j] : 0;
                hd[j] = ok ? g.head[a0 + j] : 0;
                cs[j] = ok ? g.cost[a0 + j] : 0;
                if (ok) t.c[kTallyScans]++;
            }
        }
        i64 key[4], want = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            key[j] = cp[j] > 0 ? g.price[hd[j]] - (i64)cs[j] * S : kNegInf;
            if (cp[j] > 0 && key[j] > pu) want += cp[j];
        }
        if (__ballot(want > 0)) {
            const i64 incl = wave_incl_scan(want);
            const i64 total = __shfl(incl, 63);
            i64 granted = 0;
            if (lane == 0) {
                const i64 old = aadd64(&g.excess[u], -total);
                granted = old <= 0 ? 0 : (old < total ? old : total);
                if (granted < total) aadd64(&g.excess[u], total - granted);
            }
            granted = __shfl(granted, 0);
            i64 mine = granted - (incl - want);  // what this lane may push
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (cp[j] > 0 && key[j] > pu && mine > 0) {
                    const i64 d = mine < cp[j] ? mine : cp[j];
                    do_push(g, a0 + j, hd[j], cp[j], d, pu, (i64)cs[j] * S);
                    cp[j] -= (int)d; mine -= d;
                    t.c[kTallyPushes]++;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (cp[j] > 0 && key[j] > best) best = key[j];
        if (r0 + h + 256 >= r1) break;
    }
    best = wave_max(best);
    if (lane == 0 && best > kNegInf) amax64(&g.cand[u], best);
}

__global__ __launch_bounds__(kBlock) void k_push(DevGraph g, int k) {
    Ctrl *ct = g.ctrl;
    if (!ct->rounds_on) return;
    const unsigned *cnt = ct->list_cnt[k & 1];
    const unsigned nlow = cnt[kClsLow], nmed = cnt[kClsMed], nhigh = cnt[kClsHigh];
    if (blockIdx.x == 0 && threadIdx.x < 4) ct->list_cnt[(k + 1) & 1][threadIdx.x] = 0;
    if (nlow + nmed + nhigh == 0) return;
    const i64 S = g.scale;
    const unsigned salt = ct->epoch * 64u + (unsigned)k;
    const int *lowl = g.list_low[k & 1];
    const int *medl = g.list_med[k & 1];
    const int2 *highl = g.list_high[k & 1];
    const unsigned wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * kBlock) >> 6;
    Tally t;
    // segments of high-degree vertices first (largest items), start rotated per round so
    // that vertices sharing neighbours do not all begin at the same ones
    for (unsigned i = wave; i < nhigh; i += nwaves) {
        const int2 it = highl[i];
        const unsigned deg = g.row[it.x + 1] - g.row[it.x];
        const unsigned nseg = (deg + kWSeg - 1) / kWSeg;
        const unsigned seg = ((unsigned)it.y + mix32((unsigned)it.x, salt)) % nseg;
        push_high(g, it.x, seg, S, t);
    }
    for (unsigned i = wave; i < nmed; i += nwaves) push_med(g, medl[i], S, salt, t);
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < nlow; i += gridDim.x * kBlock) push_low(g, lowl[i], S, t);
    if (blockIdx.x == 0 && threadIdx.x == 0) t.c[kTallyRounds]++;
    tally_flush(g, t);
}

// Relabel (optional) + rebuild of the active lists for the next round: one sweep over
// the vertices.  cand[v] is +inf for vertices outside the list, so a vertex that only
// became active during the push phase is never relabelled on partial information.
__global__ __launch_bounds__(kBlock) void k_relabel_compact(DevGraph g, int k, int do_relabel) {
    Ctrl *ct = g.ctrl;
    if (!ct->rounds_on) return;
    const int dst = do_relabel ? (k + 1) & 1 : 0;
    if (do_relabel) {
        const unsigned *cnt = ct->list_cnt[k & 1];
        if (cnt[kClsLow] + cnt[kClsMed] + cnt[kClsHigh] == 0) return;  // nothing was active
    }
    const i64 eps = ct->eps;
    Tally t;
    const int nround = (g.n + 63) & ~63;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < nround; v += gridDim.x * kBlock) {
        bool act = false;
        int cls = 0;
        if (v < g.n) {
            const i64 e = g.excess[v];
            act = e > 0;
            const i64 c = g.cand[v];
            if (act && do_relabel && c != kInf && c != kNegInf) {
                const i64 p = g.price[v];
                if (c <= p) { g.price[v] = c - eps; t.c[kTallyRelabels]++; }
            }
            const i64 nc = act ? kNegInf : kInf;
            if (c != nc) g.cand[v] = nc;
            cls = g.cls[v];
        }
        wave_append(g, ct->list_cnt[dst], g.list_low[dst], g.list_med[dst], g.list_high[dst], act, v, cls);
    }
    tally_flush(g, t);
}

// ------------------------------------------------------------- per-phase saturation
// refine(eps) starts from an eps-optimal pseudoflow: every residual arc whose reduced
// cost is below -eps is saturated.  One thread per arc; only the thread of arc a writes
// cap[a] and cap[rev[a]] (the reverse has reduced cost > eps and is left alone).
__global__ __launch_bounds__(kBlock) void k_saturate(DevGraph g) {
    Ctrl *ct = g.ctrl;
    if (!ct->sat_on) return;
    const i64 eps = ct->eps, S = g.scale;
    Tally t;
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        const int cp = g.cap[a];
        t.c[kTallySat]++;
        if (cp <= 0) continue;
        const int u = g.tail[a], v = g.head[a];
        const i64 rc = (i64)g.cost[a] * S + g.price[u] - g.price[v];
        if (rc < -eps) {
            const int r = g.rev[a];
            const int nr = g.cap[r] + cp;
            g.cap[a] = 0; g.cap[r] = nr; g.rcap[a] = nr; g.rcap[r] = 0;
            aadd64(&g.excess[u], -(i64)cp);
            aadd64(&g.excess[v], (i64)cp);
        }
    }
    tally_flush(g, t);
}

// ------------------------------------------------------- global price update (K3)
// Distances (in units of eps) to the nearest deficit over reversed residual arcs with
// length floor(rc/eps)+1 >= 0, by frontier relaxation to a fixed point; then
// price[v] -= eps*dist[v].  Newly improved vertices are staged in a per-wave LDS queue
// and flushed to the global frontier in coalesced bursts with one atomic per burst.
struct WaveQ {
    int *low, *med;       // LDS staging, kQCap entries each
    unsigned nlow, nmed;  // wave-uniform fill counts
};
__device__ __forceinline__ void waveq_flush_one(int *q, unsigned &n, unsigned *cnt, int *dst) {
    if (n == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    unsigned base = 0;
    if (lane_id() == 0) base = aadd32(cnt, n);
    base = __shfl(base, 0);
    for (unsigned i = lane_id(); i < n; i += 64) dst[base + i] = q[i];
    n = 0;
}
__device__ __forceinline__ void waveq_push(const DevGraph &g, WaveQ &q, unsigned *cnt, int *lowl, int *medl, int2 *highl,
                                           bool take, int v) {
    int cls = take ? g.cls[v] : 0;
    const unsigned long long lt = lanemask_lt();
    unsigned long long m = __ballot(take && cls == kClsLow);
    if (m) {
        if (take && cls == kClsLow) q.low[q.nlow + __popcll(m & lt)] = v;
        q.nlow += (unsigned)__popcll(m);
        if (q.nlow > kQCap - 64) waveq_flush_one(q.low, q.nlow, &cnt[kClsLow], lowl);
    }
    m = __ballot(take && cls == kClsMed);
    if (m) {
        if (take && cls == kClsMed) q.med[q.nmed + __popcll(m & lt)] = v;
        q.nmed += (unsigned)__popcll(m);
        if (q.nmed > kQCap - 64) waveq_flush_one(q.med, q.nmed, &cnt[kClsMed], medl);
    }
    if (take && cls == kClsHigh) {
        unsigned deg = g.row[v + 1] - g.row[v];
        unsigned nseg = (deg + kWSeg - 1) / kWSeg;
        unsigned base = aadd32(&cnt[kClsHigh], nseg);
        for (unsigned s = 0; s < nseg; ++s) highl[base + s] = make_int2(v, (int)s);
    }
}

// Relax the tail of reversed arc a = (v,u): residual arc (u,v) has capacity rcap[a].
__device__ __forceinline__ bool gu_relax(const DevGraph &g, unsigned a, i64 dv, i64 pv, i64 S, int sh, unsigned stamp,
                                         int &u_out, Tally &t) {
    const int u = g.head[a];
    const i64 rc = -(i64)g.cost[a] * S + g.price[u] - pv;
    i64 l = (rc >> sh) + 1;
    if (l < 0) l = 0;
    const i64 nd = dv + l;
    if (nd >= g.dist[u]) return false;
    const i64 old = amin64(&g.dist[u], nd);
    if (nd >= old) return false;
    t.c[kTallyGuRelax]++;
    u_out = u;
    return axchg32(&g.fstamp[u], stamp) != stamp;  // first improver in this iteration queues u
}

__global__ __launch_bounds__(kBlock) void k_gu_expand(DevGraph g, int k) {
    Ctrl *ct = g.ctrl;
    if (!ct->gu_on) return;
    const int src = k % 3, dst = (k + 1) % 3, clr = (k + 2) % 3;
    const unsigned *cnt = ct->fr_cnt[src];
    const unsigned nlow = cnt[kClsLow], nmed = cnt[kClsMed], nhigh = cnt[kClsHigh];
    if (blockIdx.x == 0 && threadIdx.x < 4) ct->fr_cnt[clr][threadIdx.x] = 0;
    if (nlow + nmed + nhigh == 0) return;
    __shared__ int lds_q[(kBlock / 64) * 2 * kQCap];
    WaveQ q;
    q.low = lds_q + (threadIdx.x >> 6) * 2 * kQCap;
    q.med = q.low + kQCap;
    q.nlow = q.nmed = 0;
    unsigned *dcnt = ct->fr_cnt[dst];
    int *dlow = g.fr_low[dst], *dmed = g.fr_med[dst];
    int2 *dhigh = g.fr_high[dst];
    const i64 S = g.scale;
    const int sh = ct->eps_shift;
    const unsigned stamp = ct->gu_stamp_base + (unsigned)k + 1u;
    const int lane = lane_id();
    const unsigned wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * kBlock) >> 6;
    Tally t;
    if (blockIdx.x == 0 && threadIdx.x == 0) t.c[kTallyGuIters]++;
    const int2 *highl = g.fr_high[src];
    for (unsigned i = wave; i < nhigh; i += nwaves) {
        const int2 it = highl[i];
        const int v = it.x;
        const unsigned r0 = g.row[v] + (unsigned)it.y * kWSeg;
        const unsigned r1 = min(g.row[v + 1], r0 + kWSeg);
        const i64 dv = g.dist[v], pv = g.price[v];
        for (unsigned h = 0; h < kWSeg; h += 256) {
            const unsigned a0 = r0 + h + lane * 4;
            int rc4[4];
            if (a0 + 3 < r1) {
                int4u x = *reinterpret_cast<const int4u *>(g.rcap + a0);
                rc4[0] = x.x; rc4[1] = x.y; rc4[2] = x.z; rc4[3] = x.w;
                t.c[kTallyGuScans] += 4;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const bool ok = a0 + j < r1; rc4[j] = ok ? g.rcap[a0 + j] : 0; if (ok) t.c[kTallyGuScans]++; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int u = 0;
                bool enq = false;
                if (rc4[j] > 0) enq = gu_relax(g, a0 + j, dv, pv, S, sh, stamp, u, t);
                if (__ballot(enq)) waveq_push(g, q, dcnt, dlow, dmed, dhigh, enq, u);
            }
            if (r0 + h + 256 >= r1) break;
        }
    }
    const int *medl = g.fr_med[src];
    for (unsigned i = wave; i < nmed; i += nwaves) {
        const int v = medl[i];
        const unsigned r0 = g.row[v], r1 = g.row[v + 1];
        const i64 dv = g.dist[v], pv = g.price[v];
        for (unsigned a = r0 + lane; a - lane < r1; a += 64) {
            int u = 0;
            bool enq = false;
            if (a < r1) {
                t.c[kTallyGuScans]++;
                if (g.rcap[a] > 0) enq = gu_relax(g, a, dv, pv, S, sh, stamp, u, t);
            }
            if (__ballot(enq)) waveq_push(g, q, dcnt, dlow, dmed, dhigh, enq, u);
        }
    }
    const int *lowl = g.fr_low[src];
    const unsigned nlow_round = (nlow + 63) & ~63u;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < nlow_round; i += gridDim.x * kBlock) {
        const bool have = i < nlow;
        const int v = have ? lowl[i] : 0;
        const unsigned r0 = have ? g.row[v] : 0, r1 = have ? g.row[v + 1] : 0;
        const i64 dv = have ? g.dist[v] : 0, pv = have ? g.price[v] : 0;
        for (unsigned j = 0; j < kLowMax; ++j) {
            const unsigned a = r0 + j;
            int u = 0;
            bool enq = false;
            if (a < r1) {
                t.c[kTallyGuScans]++;
                if (g.rcap[a] > 0) enq = gu_relax(g, a, dv, pv, S, sh, stamp, u, t);
            }
            if (__ballot(enq)) waveq_push(g, q, dcnt, dlow, dmed, dhigh, enq, u);
        }
    }
    waveq_flush_one(q.low, q.nlow, &dcnt[kClsLow], dlow);
    waveq_flush_one(q.med, q.nmed, &dcnt[kClsMed], dmed);
    tally_flush(g, t);
}

// dist = 0 at deficits (they seed frontier 0), "infinite" elsewhere.
__global__ __launch_bounds__(kBlock) void k_gu_init(DevGraph g) {
    Ctrl *ct = g.ctrl;
    if (!ct->gu_init_on) return;
    const int nround = (g.n + 63) & ~63;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < nround; v += gridDim.x * kBlock) {
        bool def = false;
        int cls = 0;
        if (v < g.n) {
            def = g.excess[v] < 0;
            g.dist[v] = def ? 0 : kDistInf;
            cls = g.cls[v];
        }
        wave_append(g, ct->fr_cnt[0], g.fr_low[0], g.fr_med[0], g.fr_high[0], def, v, cls);
    }
}

__global__ __launch_bounds__(kBlock) void k_gu_dmax(DevGraph g) {
    Ctrl *ct = g.ctrl;
    if (!ct->gu_apply_on) return;
    i64 mx = 0;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < g.n; v += gridDim.x * kBlock) {
        const i64 d = g.dist[v];
        if (d < kDistInf && d > mx) mx = d;
    }
    mx = wave_max(mx);
    if (lane_id() == 0 && mx > 0) amax64(&ct->dmax, mx);
}

__global__ __launch_bounds__(kBlock) void k_gu_apply(DevGraph g) {
    Ctrl *ct = g.ctrl;
    if (!ct->gu_apply_on) return;
    const i64 eps = ct->eps;
    const i64 dfar = ct->dmax + 1;
    bool bad = false;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < g.n; v += gridDim.x * kBlock) {
        const i64 d = g.dist[v];
        if (d >= kDistInf && g.excess[v] > 0) bad = true;  // supply that reaches no deficit
        const i64 dd = d < kDistInf ? d : dfar;
        if (dd) g.price[v] -= eps * dd;
    }
    if (bad) ct->error = PSN_EINFEASIBLE_DEV;
}

// ------------------------------------------------------------- control state machine
__global__ void k_begin(DevGraph g) {
    Ctrl *ct = g.ctrl;
    ct->epoch++;
    ct->sat_on = ct->gu_on = ct->gu_init_on = ct->gu_apply_on = ct->rounds_on = 0;
    if (ct->done || ct->error) return;
    if (ct->need_sat) { ct->sat_on = 1; ct->need_sat = 0; ct->need_gu = 1; ct->tally_refines++; }
    if (ct->need_gu) {
        ct->gu_on = 1;
        if (!ct->gu_active) {
            ct->gu_active = 1;
            ct->gu_init_on = 1;
            ct->dmax = 0;
            ct->tally_gus++;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) ct->fr_cnt[i][j] = 0;
        }
        ct->gu_stamp_base += 64;  // fresh dedupe stamps for this replay's iterations
    }
}
__global__ void k_mid(DevGraph g, int n_iters) {
    Ctrl *ct = g.ctrl;
    if (ct->done || ct->error) return;
    if (ct->gu_on) {
        const unsigned *c = ct->fr_cnt[n_iters % 3];
        if (c[kClsLow] + c[kClsMed] + c[kClsHigh] == 0) {
            ct->gu_apply_on = 1; ct->gu_active = 0; ct->need_gu = 0; ct->rounds_on = 1;
        }
    } else {
        ct->rounds_on = 1;
    }
    for (int j = 0; j < 4; ++j) ct->list_cnt[0][j] = 0;
}
__global__ void k_end(DevGraph g, int n_rounds) {
    Ctrl *ct = g.ctrl;
    if (ct->done || ct->error || !ct->rounds_on) return;
    const unsigned *c = ct->list_cnt[n_rounds & 1];
    if (c[kClsLow] + c[kClsMed] + c[kClsHigh] == 0) {
        if (ct->eps_shift == 0) { ct->done = 1; return; }
        ct->eps_shift = ct->eps_shift > ct->alpha_log2 ? ct->eps_shift - ct->alpha_log2 : 0;
        ct->eps = (i64)1 << ct->eps_shift;
        ct->need_sat = 1;
    } else {
        ct->need_gu = 1;
    }
}

// ------------------------------------------------------------ setup / readout kernels
__global__ __launch_bounds__(kBlock) void k_reset(DevGraph g, const i64 *supply, int keep_flow) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        if (!keep_flow) g.cap[a] = g.ocap[a];
    }
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < g.n; v += gridDim.x * kBlock) {
        if (!keep_flow) { g.price[v] = 0; g.excess[v] = supply[v]; }
        g.cand[v] = kInf;
        g.fstamp[v] = 0;
    }
}
// rcap mirrors cap[rev]; excess from scratch for a kept flow: supply - outflow + inflow,
// with outflow(a) = ocap[a] - cap[a] counted on arcs where it is positive.
__global__ __launch_bounds__(kBlock) void k_mirror(DevGraph g, const i64 *supply, int recompute_excess) {
    if (recompute_excess)
        for (int v = blockIdx.x * kBlock + threadIdx.x; v < g.n; v += gridDim.x * kBlock) g.excess[v] = supply[v];
    (void)supply;
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) g.rcap[a] = g.cap[g.rev[a]];
}
__global__ __launch_bounds__(kBlock) void k_excess_from_flow(DevGraph g) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        const i64 f = (i64)g.ocap[a] - g.cap[a];
        if (f > 0) { aadd64(&g.excess[g.tail[a]], -f); aadd64(&g.excess[g.head[a]], f); }
    }
}
__global__ __launch_bounds__(kBlock) void k_total_cost(DevGraph g, i64 *out) {
    i64 s = 0;
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        const i64 f = (i64)g.ocap[a] - g.cap[a];
        if (f > 0) s += f * g.cost[a];
    }
    s = (i64)wave_sum_u64((unsigned long long)s);
    if (lane_id() == 0 && s) aadd64(out, s);
}
__global__ __launch_bounds__(kBlock) void k_flow(DevGraph g, int *flow) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) flow[a] = g.ocap[a] - g.cap[a];
}
__global__ void k_tally_sum(DevGraph g, unsigned long long *out, int nslots) {
    // one wave per counter
    const int w = threadIdx.x >> 6;
    if (w >= kTallyN) return;
    unsigned long long s = 0;
    for (int i = lane_id(); i < nslots; i += 64) s += g.tally[(size_t)i * kTallyN + w];
    s = wave_sum_u64(s);
    if (lane_id() == 0) out[w] = s;
}

// ------------------------------------------------------------- arc-cost model (K1)
// One thread per (task class e, machine m): K slot arcs, int64 arithmetic throughout.
//   fit  = min((cap_cpu-used_cpu)/req_cpu, (cap_mem-used_mem)/req_mem)
//   cap  = k < fit;  cost = 1000*(used_cpu+(k+1)*req_cpu)/cap_cpu + 1000*(used_mem+(k+1)*req_mem)/cap_mem
__global__ __launch_bounds__(kBlock) void k_eval_costs(DevGraph g, int E, int N, int K, const i64 *cap_cpu, const i64 *cap_mem,
                                                       const i64 *used_cpu, const i64 *used_mem, const i64 *req_cpu,
                                                       const i64 *req_mem, const i64 *ec_arc_base, int *mcost, int *mocap) {
    const long long total = (long long)E * N;
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int e = (int)(i / N), m = (int)(i % N);
        const i64 cc = cap_cpu[m], cm = cap_mem[m], uc = used_cpu[m], um = used_mem[m];
        const i64 rc = req_cpu[e], rm = req_mem[e];
        const i64 fc = (cc - uc) / rc, fm = (cm - um) / rm;
        const i64 fit = fc < fm ? fc : fm;
        const unsigned a0 = (unsigned)(ec_arc_base[e] + (i64)m * K);
        for (int k = 0; k < K; ++k) {
            const i64 c = (1000 * (uc + (k + 1) * rc)) / cc + (1000 * (um + (k + 1) * rm)) / cm;
            const int cp = k < fit ? 1 : 0;
            const unsigned a = a0 + k;
            const int r = g.rev[a];
            mcost[a] = (int)c; mcost[r] = -(int)c;
            mocap[a] = cp; mocap[r] = 0;
            g.cap[a] = cp; g.cap[r] = 0;
            g.rcap[a] = 0; g.rcap[r] = cp;
        }
    }
}

// ----------------------------------------------------- placement extraction (K5)
// Tasks of one class are interchangeable, so the class's routed tasks (in arc order) are
// matched to its units of outgoing flow (in arc order).  One workgroup per class:
// pass 1 numbers the units of flow on slot arcs, pass 2 numbers the routed tasks.
__device__ __forceinline__ unsigned block_excl_scan(unsigned x, unsigned *lds, unsigned &total) {
    const int lane = lane_id(), w = threadIdx.x >> 6;
    unsigned incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    unsigned off = 0, tot = 0;
    for (int i = 0; i < kBlock / 64; ++i) { if (i < w) off += lds[i]; tot += lds[i]; }
    __syncthreads();
    total = tot;
    return off + incl - x;
}
__global__ __launch_bounds__(kBlock) void k_placements(DevGraph g, int N, int E, int K, int v_task0, int *slot_machine,
                                                       int slot_stride, int *place) {
    __shared__ unsigned lds[kBlock / 64];
    const int e = blockIdx.x;
    if (e >= E) return;
    const int u = 1 + N + e;
    const unsigned r0 = g.row[u], r1 = g.row[u + 1];
    const unsigned nslot = (unsigned)N * K;
    const unsigned t1 = r1 - nslot;  // [r0,t1): reverse arcs to tasks, [t1,r1): slot arcs
    int *sm = slot_machine + (size_t)e * slot_stride;
    unsigned run = 0;
    for (unsigned b = t1; b < r1; b += kBlock) {
        const unsigned a = b + threadIdx.x;
        unsigned f = 0;
        if (a < r1) { int fl = g.ocap[a] - g.cap[a]; f = fl > 0 ? (unsigned)fl : 0; }
        unsigned tot;
        unsigned pos = run + block_excl_scan(f, lds, tot);
        for (unsigned j = 0; j < f; ++j) if (pos + j < (unsigned)slot_stride) sm[pos + j] = g.head[a] - 1;
        run += tot;
    }
    __syncthreads();
    const unsigned units = run;
    run = 0;
    for (unsigned b = r0; b < t1; b += kBlock) {
        const unsigned a = b + threadIdx.x;
        unsigned f = 0;
        if (a < t1) f = g.cap[a] > 0 ? 1u : 0u;  // residual on the reverse arc == task routed here
        unsigned tot;
        unsigned pos = run + block_excl_scan(f, lds, tot);
        if (f) place[g.head[a] - v_task0] = pos < units ? sm[pos] : -1;
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
// After capacities shrank under a kept flow: cancel the flow that no longer fits.
__global__ __launch_bounds__(kBlock) void k_clamp_flow(DevGraph g) {
    for (unsigned a = blockIdx.x * kBlock + threadIdx.x; a < g.m2; a += gridDim.x * kBlock) {
        const int c = g.cap[a];
        if (c < 0) { g.cap[a] = 0; g.cap[g.rev[a]] += c; }
    }
}
__global__ __launch_bounds__(kBlock) void k_scatter_supply(long long count, const int *vtx, const i64 *val, i64 *supply) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) supply[vtx[i]] = val[i];
}

}  // namespace psn
