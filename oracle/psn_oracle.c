/*
 * ORACLE - test infrastructure only.  Nothing under poseidon_amd/ links, loads
 * or calls this file; it exists so that tests, the smoke check and the
 * cpu_baseline leg of the benchmark have an independent CPU answer.
 *
 * Parity status: PARITY UNPINNED.  The reference repository contains no
 * min-cost-flow solver: Firmament is an absent git submodule
 * (/root/reference/.gitmodules:1-3) and its cs2 / Flowlessly solvers are
 * third-party code named only in /root/reference/LICENSE:18-26, with no pinned
 * version recoverable.  The only call sites are the Schedule RPC
 * (/root/reference/pkg/firmament/firmament_client.go:29-35) and its consumer
 * (/root/reference/cmd/poseidon/poseidon.go:34-68); every reference test mocks
 * the solver away (/root/reference/pkg/firmament/firmament_client_test.go:132-139).
 * What is restated here is therefore the published mathematics:
 *   - psn_oracle_ssp:  successive shortest paths / primal-dual with Dijkstra
 *     and node potentials (Edmonds-Karp 1972, Tomizawa 1971);
 *   - psn_oracle_cost_scaling: Goldberg-Tarjan successive approximation with
 *     push/relabel refine and global price updates (Goldberg, "An efficient
 *     implementation of a scaling minimum-cost flow algorithm", J. Algorithms
 *     22, 1997) - the class of algorithm cs2 implements; this is the timed
 *     single-thread CPU baseline;
 *   - psn_oracle_check: feasibility + optimality certificate that trusts no
 *     solver (conservation, capacity bounds, non-negative reduced cost on every
 *     residual arc for potentials it derives itself).
 * For integer data the optimal objective value is unique, so any exact solver
 * pins the number the GPU path must reproduce bit for bit.
 *
 * Graph encoding (same as include/psn.h): residual arcs grouped by tail,
 * row_off[n+1], head/cap/cost/rev[m2]; cost[rev[a]] == -cost[a]; supply[n].
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

typedef int64_t i64;
typedef int32_t i32;

#define ORC_OK 0
#define ORC_EINFEASIBLE -4
#define ORC_ENOMEM -2
#define ORC_EBAD -1

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------ hashing */
uint64_t psn_oracle_fnv1a64(const void *data, i64 nbytes, uint64_t h) {
    const unsigned char *p = (const unsigned char *)data;
    for (i64 i = 0; i < nbytes; ++i) { h ^= p[i]; h *= 0x100000001B3ULL; }
    return h;
}

/* --------------------------------------------------------------- cost model */
/* CPU/memory slot cost for (shape e, machine m, slot k); int64 arithmetic only. */
void psn_oracle_eval_costs(i32 E, i32 N, i32 K, const i64 *req_cpu, const i64 *req_mem,
                           const i64 *cap_cpu, const i64 *cap_mem, const i64 *used_cpu,
                           const i64 *used_mem, i32 *cost_out, i32 *cap_out) {
    for (i32 e = 0; e < E; ++e)
        for (i32 m = 0; m < N; ++m) {
            i64 fc = (cap_cpu[m] - used_cpu[m]) / req_cpu[e];
            i64 fm = (cap_mem[m] - used_mem[m]) / req_mem[e];
            i64 fit = fc < fm ? fc : fm;
            for (i32 k = 0; k < K; ++k) {
                i64 c = (1000 * (used_cpu[m] + (k + 1) * req_cpu[e])) / cap_cpu[m] +
                        (1000 * (used_mem[m] + (k + 1) * req_mem[e])) / cap_mem[m];
                size_t o = ((size_t)e * N + m) * K + k;
                cost_out[o] = (i32)c;
                cap_out[o] = k < fit ? 1 : 0;
            }
        }
}

/* --------------------------------------------------------------- binary heap */
typedef struct { i64 key; i32 v; } hitem;
typedef struct { hitem *a; i64 n, cap; } heap;
static int heap_push(heap *h, i64 key, i32 v) {
    if (h->n == h->cap) {
        i64 nc = h->cap ? h->cap * 2 : 1024;
        hitem *na = (hitem *)realloc(h->a, nc * sizeof(hitem));
        if (!na) return -1;
        h->a = na; h->cap = nc;
    }
    i64 i = h->n++;
    while (i > 0) {
        i64 p = (i - 1) / 2;
        if (h->a[p].key <= key) break;
        h->a[i] = h->a[p]; i = p;
    }
    h->a[i].key = key; h->a[i].v = v;
    return 0;
}
static hitem heap_pop(heap *h) {
    hitem top = h->a[0];
    hitem last = h->a[--h->n];
    i64 i = 0;
    for (;;) {
        i64 c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && h->a[c + 1].key < h->a[c].key) ++c;
        if (h->a[c].key >= last.key) break;
        h->a[i] = h->a[c]; i = c;
    }
    if (h->n) h->a[i] = last;
    return top;
}

/* ------------------------------------------------- successive shortest paths */
/* Primal-dual: one multi-source Dijkstra on reduced costs per phase, then DFS
 * augmentations over zero-reduced-cost residual arcs until none is left.
 * cap[] is updated in place to the final residual capacities.
 * pot_out (may be NULL) receives potentials with cost+pot[u]-pot[v] >= 0 on
 * every residual arc. */
typedef struct {
    i32 n; const i64 *row_off; const i32 *head; i32 *cap; const i32 *cost; const i32 *rev;
    i64 *pot; i64 *excess; i64 *cur; unsigned char *dead; i32 *stk_v; i64 *stk_a;
} ssp_ctx;

/* One augmentation from s over zero-reduced-cost residual arcs (iterative DFS with
 * per-phase current-arc pointers).  dead[v]: 1 = exhausted, 2 = on the DFS stack.
 * Returns the amount pushed (0 if s cannot reach a deficit). */
static i64 ssp_augment(ssp_ctx *c, i32 s) {
    i64 depth = 0;
    c->stk_v[0] = s;
    c->dead[s] = 2;
    for (;;) {
        i32 u = c->stk_v[depth];
        if (c->excess[u] < 0) { /* reached a deficit: push the bottleneck */
            i64 amt = c->excess[s] < -c->excess[u] ? c->excess[s] : -c->excess[u];
            for (i64 d = 0; d < depth; ++d)
                if (c->cap[c->stk_a[d]] < amt) amt = c->cap[c->stk_a[d]];
            for (i64 d = 0; d < depth; ++d) {
                i64 a = c->stk_a[d];
                c->cap[a] -= (i32)amt;
                c->cap[c->rev[a]] += (i32)amt;
            }
            c->excess[s] -= amt;
            c->excess[u] += amt;
            for (i64 d = 0; d <= depth; ++d) c->dead[c->stk_v[d]] = 0;
            return amt;
        }
        int advanced = 0;
        for (; c->cur[u] < c->row_off[u + 1]; ++c->cur[u]) {
            i64 a = c->cur[u];
            i32 v = c->head[a];
            if (c->cap[a] <= 0 || c->dead[v]) continue;
            if ((i64)c->cost[a] + c->pot[u] - c->pot[v] != 0) continue;
            c->stk_a[depth] = a;
            c->stk_v[++depth] = v;
            c->dead[v] = 2;
            advanced = 1;
            break;
        }
        if (advanced) continue;
        c->dead[u] = 1;
        if (depth == 0) return 0;
        --depth;
        ++c->cur[c->stk_v[depth]];
    }
}

int psn_oracle_ssp(i32 n, i64 m2, const i64 *row_off, const i32 *head, i32 *cap,
                   const i32 *cost, const i32 *rev, const i64 *supply, i64 *total_cost,
                   i64 *pot_out) {
    const i64 INF = INT64_MAX / 4;
    i32 *cap0 = (i32 *)malloc(sizeof(i32) * (size_t)(m2 ? m2 : 1));
    i64 *pot = (i64 *)calloc(n, sizeof(i64));
    i64 *dist = (i64 *)malloc(sizeof(i64) * n);
    i64 *excess = (i64 *)malloc(sizeof(i64) * n);
    i64 *cur = (i64 *)malloc(sizeof(i64) * n);
    unsigned char *dead = (unsigned char *)calloc(n, 1);
    unsigned char *done = (unsigned char *)malloc(n);
    i32 *stk_v = (i32 *)malloc(sizeof(i32) * ((size_t)n + 1));
    i64 *stk_a = (i64 *)malloc(sizeof(i64) * ((size_t)n + 1));
    heap h = {0, 0, 0};
    int rc = ORC_OK;
    if (!cap0 || !pot || !dist || !excess || !cur || !dead || !done || !stk_v || !stk_a) { rc = ORC_ENOMEM; goto out; }
    memcpy(cap0, cap, sizeof(i32) * (size_t)m2);
    memcpy(excess, supply, sizeof(i64) * n);
    /* negative costs: Bellman-Ford for initial potentials */
    {
        int neg = 0;
        for (i64 a = 0; a < m2; ++a) if (cap[a] > 0 && cost[a] < 0) { neg = 1; break; }
        for (int it = 0; neg && it <= n; ++it) {
            int ch = 0;
            for (i32 u = 0; u < n; ++u)
                for (i64 a = row_off[u]; a < row_off[u + 1]; ++a)
                    if (cap[a] > 0 && pot[u] + cost[a] < pot[head[a]]) { pot[head[a]] = pot[u] + cost[a]; ch = 1; }
            if (!ch) break;
            if (it == n) { rc = ORC_EBAD; goto out; } /* negative cycle */
        }
        /* pot now satisfies pot[v] <= pot[u] + cost  <=> cost + pot[u] - pot[v] >= 0 */
    }
    ssp_ctx c = {n, row_off, head, cap, cost, rev, pot, excess, cur, dead, stk_v, stk_a};
    for (;;) {
        i64 remaining = 0;
        for (i32 v = 0; v < n; ++v) if (excess[v] > 0) remaining += excess[v];
        if (remaining == 0) break;
        for (i32 v = 0; v < n; ++v) { dist[v] = INF; done[v] = 0; }
        h.n = 0;
        for (i32 v = 0; v < n; ++v)
            if (excess[v] > 0) { dist[v] = 0; if (heap_push(&h, 0, v)) { rc = ORC_ENOMEM; goto out; } }
        i64 dt = INF;
        while (h.n) {
            hitem it = heap_pop(&h);
            i32 u = it.v;
            if (done[u] || it.key != dist[u]) continue;
            done[u] = 1;
            if (excess[u] < 0) { dt = dist[u]; break; }
            for (i64 a = row_off[u]; a < row_off[u + 1]; ++a) {
                if (cap[a] <= 0) continue;
                i32 v = head[a];
                i64 nd = dist[u] + cost[a] + pot[u] - pot[v];
                if (nd < dist[v]) { dist[v] = nd; if (heap_push(&h, nd, v)) { rc = ORC_ENOMEM; goto out; } }
            }
        }
        if (dt == INF) { rc = ORC_EINFEASIBLE; goto out; }
        /* pot'[v] = pot[v] + min(dist[v], dt): keeps cost + pot[u] - pot[v] >= 0 on
         * residual arcs and makes a shortest excess->deficit path zero-reduced-cost */
        for (i32 v = 0; v < n; ++v) {
            i64 d = (done[v] && dist[v] < dt) ? dist[v] : dt;
            pot[v] += d;
        }
        memset(dead, 0, n);
        for (i32 v = 0; v < n; ++v) cur[v] = row_off[v];
        i64 pushed_phase = 0;
        for (i32 s = 0; s < n; ++s)
            while (excess[s] > 0 && dead[s] != 1) {
                i64 got = ssp_augment(&c, s);
                if (got <= 0) break;
                pushed_phase += got;
            }
        if (pushed_phase == 0) { rc = ORC_EINFEASIBLE; goto out; }
    }
    {
        i64 tc = 0;
        for (i64 a = 0; a < m2; ++a) {
            i64 f = (i64)cap0[a] - cap[a];
            if (f > 0) tc += f * cost[a];
        }
        *total_cost = tc;
    }
    if (pot_out) memcpy(pot_out, pot, sizeof(i64) * n);
out:
    free(cap0); free(pot); free(dist); free(excess); free(cur); free(dead); free(done); free(stk_v); free(stk_a); free(h.a);
    return rc;
}

/* ------------------------------------------------------------- cost scaling */
/* Sequential Goldberg-Tarjan cost scaling.  Costs are multiplied by (n+1) so that
 * a 1-optimal flow is optimal.  Reduced cost of residual arc a=(u,v):
 *   rc(a) = cost[a]*(n+1) + price[u] - price[v];   admissible iff rc(a) < 0.
 * refine(eps): saturate every residual arc with rc < 0, then discharge active
 * vertices FIFO with current-arc pointers; relabel sets
 *   price[u] = max over residual (u,v) of (price[v] - cost*(n+1)) - eps.
 * Global price update (set-relabel): bucketed Dijkstra from the deficit vertices
 * over reversed residual arcs with lengths floor(rc/eps)+1, then
 *   price[v] -= eps * dist[v].
 */
typedef struct {
    i64 pushes, relabels, arc_scans, global_updates, refines;
    double solve_ms;
} psn_oracle_stats;

typedef struct {
    i32 n; i64 m2; const i64 *row_off; const i32 *head; i32 *cap; const i32 *cost; const i32 *rev;
    i64 S; i64 *price; i64 *excess; i64 *cur;
    i32 *q; i64 qh, qt, qcap; unsigned char *inq;
    i32 *bnext, *bprev; i64 *dist; i32 *bucket; i64 nbuckets; unsigned char *settled;
    psn_oracle_stats st;
} cs_ctx;

static inline i64 cs_rc(const cs_ctx *c, i64 a, i32 u) {
    return (i64)c->cost[a] * c->S + c->price[u] - c->price[c->head[a]];
}
static inline void cs_enqueue(cs_ctx *c, i32 v) {
    if (c->inq[v]) return;
    c->inq[v] = 1;
    c->q[c->qt] = v;
    if (++c->qt == c->qcap) c->qt = 0;
}
static inline i64 floordiv(i64 a, i64 b) { /* b > 0 */
    i64 q = a / b;
    if ((a % b) != 0 && a < 0) --q;
    return q;
}

static void cs_bucket_insert(cs_ctx *c, i32 v, i64 b) {
    c->bprev[v] = -1;
    c->bnext[v] = c->bucket[b];
    if (c->bucket[b] >= 0) c->bprev[c->bucket[b]] = v;
    c->bucket[b] = v;
}
static void cs_bucket_remove(cs_ctx *c, i32 v, i64 b) {
    if (c->bprev[v] >= 0) c->bnext[c->bprev[v]] = c->bnext[v];
    else c->bucket[b] = c->bnext[v];
    if (c->bnext[v] >= 0) c->bprev[c->bnext[v]] = c->bprev[v];
}

static void cs_global_update(cs_ctx *c, i64 eps) {
    const i32 n = c->n;
    const i64 NB = c->nbuckets;
    const i64 DINF = INT64_MAX / 4;
    c->st.global_updates++;
    for (i64 b = 0; b < NB; ++b) c->bucket[b] = -1;
    i64 n_active = 0;
    for (i32 v = 0; v < n; ++v) {
        c->settled[v] = 0;
        if (c->excess[v] < 0) { c->dist[v] = 0; cs_bucket_insert(c, v, 0); }
        else { c->dist[v] = DINF; if (c->excess[v] > 0) ++n_active; }
    }
    i64 dlast = 0;
    for (i64 b = 0; b < NB; ++b) {
        while (c->bucket[b] >= 0) {
            i32 v = c->bucket[b];
            cs_bucket_remove(c, v, b);
            c->settled[v] = 1;
            dlast = b;
            if (c->excess[v] > 0) --n_active;
            for (i64 a = c->row_off[v]; a < c->row_off[v + 1]; ++a) {
                i32 u = c->head[a];
                i64 r = c->rev[a]; /* residual arc u -> v */
                c->st.arc_scans++;
                if (c->cap[r] <= 0 || c->settled[u]) continue;
                i64 rc = (i64)c->cost[r] * c->S + c->price[u] - c->price[v];
                i64 l = floordiv(rc, eps) + 1;
                if (l < 0) l = 0;
                i64 nd = b + l;
                if (nd >= c->dist[u] || nd >= NB) continue;
                if (c->dist[u] < NB) cs_bucket_remove(c, u, c->dist[u]);
                c->dist[u] = nd;
                cs_bucket_insert(c, u, nd);
            }
        }
        if (n_active == 0) break; /* bucket b is fully drained: every dist <= b is settled */
    }
    for (i32 v = 0; v < n; ++v) {
        c->price[v] -= eps * (c->settled[v] ? c->dist[v] : dlast + 1);
        c->cur[v] = c->row_off[v];
    }
}

static void cs_relabel(cs_ctx *c, i32 u, i64 eps) {
    i64 best = INT64_MIN;
    for (i64 a = c->row_off[u]; a < c->row_off[u + 1]; ++a) {
        c->st.arc_scans++;
        if (c->cap[a] <= 0) continue;
        i64 k = c->price[c->head[a]] - (i64)c->cost[a] * c->S;
        if (k > best) best = k;
    }
    if (best != INT64_MIN) c->price[u] = best - eps;
    else c->price[u] -= eps; /* isolated: cannot happen in a feasible instance */
    c->cur[u] = c->row_off[u];
    c->st.relabels++;
}

static int cs_refine(cs_ctx *c, i64 eps) {
    const i32 n = c->n;
    c->st.refines++;
    /* 1. saturate arcs with negative reduced cost -> 0-optimal pseudoflow */
    for (i32 u = 0; u < n; ++u)
        for (i64 a = c->row_off[u]; a < c->row_off[u + 1]; ++a) {
            if (c->cap[a] > 0 && cs_rc(c, a, u) < 0) {
                i64 d = c->cap[a];
                c->cap[a] = 0;
                c->cap[c->rev[a]] += (i32)d;
                c->excess[u] -= d;
                c->excess[c->head[a]] += d;
            }
        }
    c->qh = c->qt = 0;
    memset(c->inq, 0, n);
    for (i32 v = 0; v < n; ++v) {
        c->cur[v] = c->row_off[v];
        if (c->excess[v] > 0) cs_enqueue(c, v);
    }
    i64 relabels_since = 0;
    const i64 gu_period = (i64)n / 2 + 1;
    i64 guard = 0;
    cs_global_update(c, eps);
    while (c->qh != c->qt) {
        i32 u = c->q[c->qh];
        if (++c->qh == c->qcap) c->qh = 0;
        c->inq[u] = 0;
        while (c->excess[u] > 0) {
            i64 a = c->cur[u];
            if (a == c->row_off[u + 1]) {
                cs_relabel(c, u, eps);
                if (++relabels_since >= gu_period) {
                    relabels_since = 0;
                    cs_global_update(c, eps);
                }
                if (++guard > (i64)4e11) return ORC_EBAD;
                continue;
            }
            c->st.arc_scans++;
            if (c->cap[a] > 0 && cs_rc(c, a, u) < 0) {
                i32 v = c->head[a];
                i64 d = c->excess[u] < c->cap[a] ? c->excess[u] : c->cap[a];
                c->cap[a] -= (i32)d;
                c->cap[c->rev[a]] += (i32)d;
                c->excess[u] -= d;
                c->excess[v] += d;
                c->st.pushes++;
                if (c->excess[v] > 0 && c->excess[v] <= d) cs_enqueue(c, v);
                if (c->cap[a] == 0) c->cur[u] = a + 1;
            } else {
                c->cur[u] = a + 1;
            }
        }
    }
    return ORC_OK;
}

/* Price refinement (Goldberg 1997, section 5, in label-correcting form): before refine(eps)
 * moves any flow, look for prices under which the CURRENT flow is already eps-optimal, i.e.
 * d with d(u) <= d(v) + rc(u,v) + eps on every residual arc (u,v); then price += d does it and
 * the phase is skipped.  FIFO label correcting from d = 0; gives up (returns 0, prices
 * untouched) when the scan budget is used up, which is what a negative cycle looks like. */
static int cs_price_refine(cs_ctx *c, i64 eps) {
    const i32 n = c->n;
    for (i32 v = 0; v < n; ++v) if (c->excess[v] != 0) return 0; /* needs a flow, not a pseudoflow */
    i64 *d = c->dist;
    memset(c->inq, 0, n);
    c->qh = c->qt = 0;
    for (i32 v = 0; v < n; ++v) { d[v] = 0; cs_enqueue(c, v); }
    i64 budget = 3 * c->m2 + 16 * (i64)n;
    while (c->qh != c->qt) {
        i32 v = c->q[c->qh];
        if (++c->qh == c->qcap) c->qh = 0;
        c->inq[v] = 0;
        /* arcs (u,v) entering v: the reverse entries of v's row */
        for (i64 a = c->row_off[v]; a < c->row_off[v + 1]; ++a) {
            i64 r = c->rev[a];
            c->st.arc_scans++;
            if (c->cap[r] <= 0) continue;
            i32 u = c->head[a];
            i64 l = (i64)c->cost[r] * c->S + c->price[u] - c->price[v] + eps;
            if (d[v] + l < d[u]) { d[u] = d[v] + l; cs_enqueue(c, u); }
        }
        budget -= c->row_off[v + 1] - c->row_off[v];
        if (budget < 0) { memset(c->inq, 0, n); c->qh = c->qt = 0; return 0; }
    }
    for (i32 v = 0; v < n; ++v) c->price[v] += d[v];
    return 1;
}

#define ORC_CS_EPS_PERCENTILE 1 /* first epsilon from the 90th-percentile cost (valid when no usable arc has a negative cost) */
#define ORC_CS_PRICE_REFINE 2   /* try price refinement before every phase but the first */
#define ORC_CS_COVER_SCALE 4    /* multiply costs by (cycle-length bound + 1) instead of n + 1 */
i64 psn_oracle_cycle_bound(i32 n, const i64 *row_off, const i32 *head);

/* cap[] in: original residual capacities; out: final residual capacities.
 * price_out (may be NULL): scaled prices, 1-optimal w.r.t. cost*(n+1).
 * flags = 0 is the plain textbook variant the parity tests use as the oracle; the flags add the
 * heuristics a tuned sequential solver has, for the timed CPU baseline. */
int psn_oracle_cost_scaling_ex(i32 n, i64 m2, const i64 *row_off, const i32 *head, i32 *cap,
                               const i32 *cost, const i32 *rev, const i64 *supply, i32 alpha, i32 flags,
                               i64 *total_cost, i64 *price_out, psn_oracle_stats *stats);
int psn_oracle_cost_scaling(i32 n, i64 m2, const i64 *row_off, const i32 *head, i32 *cap,
                            const i32 *cost, const i32 *rev, const i64 *supply, i32 alpha,
                            i64 *total_cost, i64 *price_out, psn_oracle_stats *stats) {
    return psn_oracle_cost_scaling_ex(n, m2, row_off, head, cap, cost, rev, supply, alpha, 0, total_cost, price_out, stats);
}
int psn_oracle_cost_scaling_ex(i32 n, i64 m2, const i64 *row_off, const i32 *head, i32 *cap,
                               const i32 *cost, const i32 *rev, const i64 *supply, i32 alpha, i32 flags,
                               i64 *total_cost, i64 *price_out, psn_oracle_stats *stats) {
    cs_ctx c;
    memset(&c, 0, sizeof(c));
    int rc = ORC_OK;
    i32 *cap0 = (i32 *)malloc(sizeof(i32) * (size_t)(m2 ? m2 : 1));
    c.n = n; c.m2 = m2; c.row_off = row_off; c.head = head; c.cap = cap; c.cost = cost; c.rev = rev;
    c.S = (i64)n + 1;
    c.price = (i64 *)calloc(n, sizeof(i64));
    c.excess = (i64 *)malloc(sizeof(i64) * n);
    c.cur = (i64 *)malloc(sizeof(i64) * n);
    c.qcap = (i64)n + 1;
    c.q = (i32 *)malloc(sizeof(i32) * c.qcap);
    c.inq = (unsigned char *)calloc(n, 1);
    c.settled = (unsigned char *)calloc(n, 1);
    c.bnext = (i32 *)malloc(sizeof(i32) * n);
    c.bprev = (i32 *)malloc(sizeof(i32) * n);
    c.dist = (i64 *)malloc(sizeof(i64) * n);
    c.nbuckets = 4 * (i64)n + 64;
    c.bucket = (i32 *)malloc(sizeof(i32) * c.nbuckets);
    if (!cap0 || !c.price || !c.excess || !c.cur || !c.q || !c.inq || !c.settled || !c.bnext ||
        !c.bprev || !c.dist || !c.bucket) { rc = ORC_ENOMEM; goto out; }
    if (alpha < 2) alpha = 12;
    if (flags & ORC_CS_COVER_SCALE) c.S = psn_oracle_cycle_bound(n, row_off, head) + 1;
    memcpy(cap0, cap, sizeof(i32) * (size_t)m2);
    memcpy(c.excess, supply, sizeof(i64) * n);
    {
        i64 sum = 0;
        for (i32 v = 0; v < n; ++v) sum += supply[v];
        if (sum != 0) { rc = ORC_EINFEASIBLE; goto out; }
    }
    double t0 = now_ms();
    i64 eps = 0;
    for (i64 a = 0; a < m2; ++a) {
        i64 cc = cost[a] < 0 ? -(i64)cost[a] : cost[a];
        if (cap[a] > 0 && cc * c.S > eps) eps = cc * c.S;
    }
    if (eps < 1) eps = 1;
    if (flags & ORC_CS_EPS_PERCENTILE) {
        /* zero flow with zero prices is 0-optimal when no usable arc has a negative cost, so the
         * first phase may start anywhere; the 90th-percentile cost skips the phases in which only
         * the few most expensive arcs are distinguishable */
        i64 hist[33], total = 0, run = 0, neg = 0;
        memset(hist, 0, sizeof(hist));
        for (i64 a = 0; a < m2; ++a) {
            if (cap[a] <= 0) continue;
            if (cost[a] < 0) { ++neg; continue; }
            int b = 0;
            while (b < 32 && ((i64)1 << b) < cost[a]) ++b;
            hist[b]++; total++;
        }
        if (!neg)
            for (int b = 0; b < 32; ++b) {
                run += hist[b];
                if (run * 10 >= total * 9) { i64 e2 = ((i64)1 << b) * c.S; if (e2 < eps) eps = e2; break; }
            }
    }
    {
        int first = 1;
        do {
            eps = eps / alpha;
            if (eps < 1) eps = 1;
            if (!first && (flags & ORC_CS_PRICE_REFINE) && cs_price_refine(&c, eps)) { first = 0; continue; }
            first = 0;
            rc = cs_refine(&c, eps);
            if (rc) goto out;
        } while (eps > 1);
    }
    c.st.solve_ms = now_ms() - t0;
    {
        i64 tc = 0;
        for (i64 a = 0; a < m2; ++a) {
            i64 f = (i64)cap0[a] - cap[a];
            if (f > 0) tc += f * cost[a];
        }
        *total_cost = tc;
        for (i32 v = 0; v < n; ++v) if (c.excess[v] != 0) { rc = ORC_EINFEASIBLE; break; }
    }
    if (price_out) memcpy(price_out, c.price, sizeof(i64) * n);
    if (stats) *stats = c.st;
out:
    free(cap0); free(c.price); free(c.excess); free(c.cur); free(c.q); free(c.inq); free(c.settled);
    free(c.bnext); free(c.bprev); free(c.dist); free(c.bucket);
    return rc;
}

/* ------------------------------------------------------------- certificate */
/* Verifies a claimed solution without trusting any solver:
 *   1 conservation: supply[v] + inflow - outflow == 0 for every v
 *   2 bounds: 0 <= cap_final[a], and flow pairs are antisymmetric
 *             (cap_final[a] + cap_final[rev[a]] == cap_orig[a] + cap_orig[rev[a]])
 *   3 optimality: derives integer potentials by label-correcting passes on the
 *     residual graph (start: hint/(n+1) if a scaled price hint is given, else 0)
 *     and checks cost + pot[u] - pot[v] >= 0 on every residual arc; a negative
 *     residual cycle makes the passes fail to converge within n rounds.
 * Returns 0 and the objective in *total_cost; 1/2/3 = which check failed. */
int psn_oracle_check(i32 n, i64 m2, const i64 *row_off, const i32 *head, const i32 *cap_orig,
                     const i32 *cap_final, const i32 *cost, const i32 *rev, const i64 *supply,
                     const i64 *price_hint, i64 *total_cost, i64 *pot_out) {
    i64 *bal = (i64 *)malloc(sizeof(i64) * n);
    i64 *pot = (i64 *)malloc(sizeof(i64) * n);
    int rc = 0;
    if (!bal || !pot) { free(bal); free(pot); return ORC_ENOMEM; }
    memcpy(bal, supply, sizeof(i64) * n);
    i64 tc = 0;
    for (i32 u = 0; u < n && !rc; ++u)
        for (i64 a = row_off[u]; a < row_off[u + 1]; ++a) {
            i64 r = rev[a];
            if (cap_final[a] < 0) { rc = 2; break; }
            if ((i64)cap_final[a] + cap_final[r] != (i64)cap_orig[a] + cap_orig[r]) { rc = 2; break; }
            if (cost[r] != -cost[a] || rev[r] != a || head[r] != u) { rc = 2; break; }
            i64 f = (i64)cap_orig[a] - cap_final[a];
            if (f > 0) { bal[u] -= f; bal[head[a]] += f; tc += f * cost[a]; }
        }
    if (!rc) for (i32 v = 0; v < n; ++v) if (bal[v] != 0) { rc = 1; break; }
    if (!rc) {
        for (i32 v = 0; v < n; ++v)
            pot[v] = price_hint ? -floordiv(-price_hint[v], (i64)n + 1) : 0;
        /* label correcting: want cost + pot[u] - pot[v] >= 0, i.e. pot[v] <= pot[u] + cost */
        int it;
        for (it = 0; it <= n; ++it) {
            int ch = 0;
            for (i32 u = 0; u < n; ++u)
                for (i64 a = row_off[u]; a < row_off[u + 1]; ++a)
                    if (cap_final[a] > 0 && pot[u] + cost[a] < pot[head[a]]) { pot[head[a]] = pot[u] + cost[a]; ch = 1; }
            if (!ch) break;
        }
        if (it > n) rc = 3;
        for (i32 u = 0; u < n && !rc; ++u)
            for (i64 a = row_off[u]; a < row_off[u + 1]; ++a)
                if (cap_final[a] > 0 && (i64)cost[a] + pot[u] - pot[head[a]] < 0) { rc = 3; break; }
    }
    if (total_cost) *total_cost = tc;
    if (pot_out && !rc) memcpy(pot_out, pot, sizeof(i64) * n);
    free(bal); free(pot);
    return rc;
}

/* Upper bound on the length of a simple cycle of the network, from a vertex cover: every arc of
 * a cycle has an endpoint in the cover and a simple cycle passes a vertex once, so it has at most
 * 2 * |cover| arcs (and at most n).  The cover is the complement of the independent set
 * { v : every neighbour has a longer row than v, ties broken by id }; self-loops are ignored. */
i64 psn_oracle_cycle_bound(i32 n, const i64 *row_off, const i32 *head) {
    /* Sharper than 2 * |cover|: outside the cover no two vertices are adjacent, so between two cover
     * vertices that follow each other on a simple cycle lies at most one other vertex.  Take a subset
     * I of the cover no two members of which are adjacent or share a neighbour outside the cover: two
     * members of I never follow each other on a cycle, so at most half of the cover vertices of a
     * cycle are in I, and a cycle has at most 4 * |cover \ I| arcs.  I = cover vertices v such that
     * (row length, id) of v is below that of every neighbour in the cover, and v is the smallest by
     * that key among the cover neighbours of each of its neighbours outside the cover.  Two arcs
     * between the same pair of vertices form a cycle of length 2, hence the floor. */
    unsigned char *in_cover = (unsigned char *)calloc((size_t)n + 1, 1);
    i32 *best = (i32 *)malloc(sizeof(i32) * ((size_t)n + 1));
    if (!in_cover || !best) { free(in_cover); free(best); return (i64)n; }
    i64 cover = 0, noni = 0;
#define ROWLEN(x) (row_off[(x) + 1] - row_off[(x)])
#define KEY_LT(a, b) (ROWLEN(a) < ROWLEN(b) || (ROWLEN(a) == ROWLEN(b) && (a) < (b)))
    for (i32 v = 0; v < n; ++v) {
        int out = 1;
        for (i64 a = row_off[v]; a < row_off[v + 1] && out; ++a) {
            const i32 u = head[a];
            if (u == v) continue;
            if (!KEY_LT(v, u)) out = 0;
        }
        in_cover[v] = !out;
        cover += !out;
    }
    for (i32 w = 0; w < n; ++w) { /* smallest-keyed cover neighbour of every vertex outside the cover */
        best[w] = -1;
        if (in_cover[w]) continue;
        for (i64 a = row_off[w]; a < row_off[w + 1]; ++a) {
            const i32 u = head[a];
            if (u == w || !in_cover[u]) continue;
            if (best[w] < 0 || KEY_LT(u, best[w])) best[w] = u;
        }
    }
    for (i32 v = 0; v < n; ++v) {
        if (!in_cover[v]) continue;
        int indep = 1;
        for (i64 a = row_off[v]; a < row_off[v + 1] && indep; ++a) {
            const i32 x = head[a];
            if (x == v) continue;
            if (in_cover[x]) { if (!KEY_LT(v, x)) indep = 0; }
            else if (best[x] != v) indep = 0;
        }
        noni += !indep;
    }
#undef KEY_LT
#undef ROWLEN
    free(in_cover); free(best);
    i64 bound = 2 * cover < (i64)n ? 2 * cover : (i64)n;
    if (4 * noni < bound) bound = 4 * noni;
    return bound < 2 ? 2 : bound;
}

/* Scaled-price certificate for an arbitrary cost multiplier `scale`: every residual arc has
 * cost*scale + price[u] - price[v] >= -eps, and eps * (longest simple cycle) < scale, which for
 * integer costs implies optimality (a residual cycle of cost <= -1 would have scaled cost
 * <= -scale, but its reduced costs sum to >= -eps * length).  Returns the number of violating
 * arcs, -1 when eps is too large for the scale. */
i64 psn_oracle_check_eps_scaled(i32 n, i64 m2, const i64 *row_off, const i32 *head, const i32 *cap_final,
                                const i32 *cost, const i64 *price, i64 eps, i64 scale) {
    i64 bad = 0;
    (void)m2;
    if (eps * psn_oracle_cycle_bound(n, row_off, head) >= scale) return -1;
    for (i32 u = 0; u < n; ++u)
        for (i64 a = row_off[u]; a < row_off[u + 1]; ++a)
            if (cap_final[a] > 0 && (i64)cost[a] * scale + price[u] - price[head[a]] < -eps) ++bad;
    return bad;
}

/* Scaled-price certificate: every residual arc has cost*(n+1)+price[u]-price[v] >= -eps
 * with eps*n < n+1, which for integer costs implies optimality (Goldberg-Tarjan 1990,
 * Thm 3.3 restated for costs multiplied by n+1).  Returns the number of violating arcs. */
i64 psn_oracle_check_eps(i32 n, i64 m2, const i64 *row_off, const i32 *head, const i32 *cap_final,
                         const i32 *cost, const i64 *price, i64 eps) {
    i64 bad = 0;
    (void)m2;
    if (eps * (i64)n >= (i64)n + 1) return -1;
    for (i32 u = 0; u < n; ++u)
        for (i64 a = row_off[u]; a < row_off[u + 1]; ++a)
            if (cap_final[a] > 0 && (i64)cost[a] * ((i64)n + 1) + price[u] - price[head[a]] < -eps) ++bad;
    return bad;
}
