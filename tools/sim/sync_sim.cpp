// Design-study tool (not part of the product, not an oracle): a sequential model of the
// bulk-synchronous cost-scaling push-relabel that the HIP kernels implement, used to count
// rounds / arc scans for heuristic choices before spending GPU time.
//   g++ -O2 -o /tmp/sync_sim tools/sim/sync_sim.cpp && /tmp/sync_sim /tmp/g1.bin [alpha] [gu_period] [sat_mode]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <chrono>
typedef int64_t i64;
typedef int32_t i32;

struct Graph {
    i64 n, m2;
    std::vector<i64> row, supply;
    std::vector<i32> head, cap, cost, rev;
};
static bool load(const char *p, Graph &g) {
    FILE *f = fopen(p, "rb");
    if (!f) return false;
    i64 h[2];
    if (fread(h, 8, 2, f) != 2) return false;
    g.n = h[0]; g.m2 = h[1];
    g.row.resize(g.n + 1); g.head.resize(g.m2); g.cap.resize(g.m2); g.cost.resize(g.m2); g.rev.resize(g.m2); g.supply.resize(g.n);
    bool ok = fread(g.row.data(), 8, g.n + 1, f) == (size_t)g.n + 1 && fread(g.head.data(), 4, g.m2, f) == (size_t)g.m2 &&
              fread(g.cap.data(), 4, g.m2, f) == (size_t)g.m2 && fread(g.cost.data(), 4, g.m2, f) == (size_t)g.m2 &&
              fread(g.rev.data(), 4, g.m2, f) == (size_t)g.m2 && fread(g.supply.data(), 8, g.n, f) == (size_t)g.n;
    fclose(f);
    return ok;
}
static inline i64 floordiv(i64 a, i64 b) { i64 q = a / b; if ((a % b) != 0 && a < 0) --q; return q; }

struct Sim {
    Graph &g;
    i64 S;
    std::vector<i64> price, excess, delta, cand, dist;
    std::vector<i32> cap;
    int alpha = 8, gu_period = 8, sat_mode = 1;
    i64 eps0_cap = 0;  // if > 0: first epsilon is min(max scaled cost, eps0_cap * S)
    int small_active = 0, small_period = 0; int walk = 0; i64 maxwl = 0; int chase = 0; i64 hubdeg = 1152; int disch = 0; i64 tailcut = 0, tailact = 0; int deep = 0; i64 deep_mindeg = 0, deep_relabels = 0; i64 kernels = 0; std::vector<char> inX, hasYnbr; std::vector<i32> pend;  // if > 0: with fewer active vertices than small_active, update every small_period rounds
    // stats
    i64 rounds = 0, scans = 0, pushes = 0, relabels = 0, gu_iters = 0, gu_scans = 0, gus = 0, sat_scans = 0;
    Sim(Graph &g_) : g(g_), S(getenv("SCALE") ? atoll(getenv("SCALE")) : g_.n + 1), price(g_.n, 0), excess(g_.supply), delta(g_.n, 0), cand(g_.n), dist(g_.n), cap(g_.cap) {}
    inline i64 rc(i64 a, i32 u) const { return (i64)g.cost[a] * S + price[u] - price[g.head[a]]; }

    void saturate(i64 eps) {
        i64 thr = sat_mode ? -eps : 0;
        for (i32 u = 0; u < g.n; ++u)
            for (i64 a = g.row[u]; a < g.row[u + 1]; ++a) {
                ++sat_scans;
                if (cap[a] > 0 && rc(a, u) < thr) {
                    i64 d = cap[a];
                    cap[a] = 0; cap[g.rev[a]] += (i32)d;
                    excess[u] -= d; excess[g.head[a]] += d;
                }
            }
    }
    void global_update(i64 eps) {
        const i64 INF = INT64_MAX / 4;
        ++gus;
        std::vector<i32> fr, nx;
        std::vector<char> innx(g.n, 0);
        for (i32 v = 0; v < g.n; ++v) { if (excess[v] < 0) { dist[v] = 0; fr.push_back(v); } else dist[v] = INF; }
        while (!fr.empty()) {
            ++gu_iters;
            nx.clear();
            for (i32 v : fr) {
                i64 dv = dist[v];
                for (i64 a = g.row[v]; a < g.row[v + 1]; ++a) {
                    ++gu_scans;
                    i64 r = g.rev[a];
                    if (cap[r] <= 0) continue;
                    i32 u = g.head[a];
                    i64 rcv = (i64)g.cost[r] * S + price[u] - price[v];
                    i64 l = floordiv(rcv, eps) + 1;
                    if (l < 0) l = 0;
                    if (dv + l < dist[u]) { dist[u] = dv + l; if (!innx[u]) { innx[u] = 1; nx.push_back(u); } }
                }
            }
            for (i32 u : nx) innx[u] = 0;
            fr.swap(nx);
        }
        i64 dmax = 0;
        for (i32 v = 0; v < g.n; ++v) if (dist[v] < INF) dmax = std::max(dmax, dist[v]);
        for (i32 v = 0; v < g.n; ++v) price[v] -= eps * (dist[v] < INF ? dist[v] : dmax + 1);
    }
    // push-through: v was not active when the round began and has just received excess; the pusher
    // forwards what v now holds along v's admissible arcs (prices frozen), up to `depth` hops
    std::vector<char> chased, was_active;
    i64 chase_hops = 0;
    void chase_from(i32 v, int depth) {
        if (was_active[v] || chased[v]) return;
        if (g.row[v + 1] - g.row[v] > hubdeg) return;
        i64 e = excess[v] + delta[v];
        if (e <= 0) return;
        chased[v] = 1;
        ++chase_hops;
        for (i64 a = g.row[v]; a < g.row[v + 1] && e > 0; ++a) {
            ++scans;
            if (cap[a] <= 0) continue;
            i32 w = g.head[a];
            i64 key = price[w] - (i64)g.cost[a] * S;
            if (key > price[v]) {
                i64 d = std::min<i64>(e, cap[a]);
                cap[a] -= (i32)d; cap[g.rev[a]] += (i32)d;
                e -= d; delta[v] -= d; delta[w] += d; ++pushes;
                i64 kr = price[v] + (i64)g.cost[a] * S;
                if (kr > cand[w]) cand[w] = kr;
                if (depth > 1) chase_from(w, depth - 1);
            }
        }
    }
    void refine(i64 eps) {
        i64 r0 = rounds, s0 = scans, p0 = pushes, l0 = relabels, gi0 = gu_iters, gs0 = gu_scans;
        saturate(eps);
        global_update(eps);
        std::vector<i32> act;
        i64 local_round = 0, lastp = pushes, lastr = relabels;
        for (;;) {
            act.clear();
            for (i32 v = 0; v < g.n; ++v) if (excess[v] > 0) act.push_back(v);
            if (act.empty()) break;
            if (tailact > 0 && eps > 1 && (i64)act.size() <= tailact) break;
            if (tailcut > 0 && eps > 1) { i64 tot = 0; for (i32 v : act) tot += excess[v]; if (tot <= tailcut) { printf("    (phase left with %ld units of excess at %zu vertices)\n", (long)tot, act.size()); break; } }
            ++rounds; ++local_round;
            if (disch) {
                // two-colour rounds: side X = independent set (no arcs inside it), side Y = the rest.
                // Kernel X: every active X vertex discharges completely (push, relabel, push ... with its
                // neighbours' prices frozen; nobody pushes into an X vertex meanwhile).  Kernel Y: active Y
                // vertices that are not hubs and have no active... (here: all non-hub Y vertices) discharge
                // completely too if none of their neighbours is a Y vertex that can relabel; hubs push once and relabel at the end.
                std::vector<char> act0(g.n, 0);
                for (i32 v = 0; v < g.n; ++v) act0[v] = excess[v] > 0;
                for (int side = 0; side < 2; ++side) {
                    std::vector<i32> lst;
                    for (i32 v = 0; v < g.n; ++v) if (excess[v] > 0 && (int)inX[v] == (side == 0) && (!getenv("NOAPPEND") || act0[v])) lst.push_back(v);
                    if (lst.empty()) continue;
                    ++kernels;
                    std::vector<i64> arrive(g.n, 0);
                    for (i32 u : lst) {
                        i64 e = excess[u];
                        i64 deg = g.row[u + 1] - g.row[u];
                        bool full = deg <= hubdeg && (side == 0 || (!hasYnbr[u] && !getenv("YSINGLE")));
                        for (int it = 0; it < (full ? 100000 : 1) && e > 0; ++it) {
                            i64 best = INT64_MIN;
                            for (i64 a = g.row[u]; a < g.row[u + 1]; ++a) {
                                ++scans;
                                if (cap[a] <= 0) continue;
                                i32 v = g.head[a];
                                i64 key = price[v] - (i64)g.cost[a] * S;
                                if (key > price[u] && e > 0) {
                                    i64 d = std::min<i64>(e, cap[a]);
                                    cap[a] -= (i32)d; cap[g.rev[a]] += (i32)d;
                                    e -= d; arrive[v] += d; ++pushes;
                                }
                                if (cap[a] > 0 && key > best) best = key;
                            }
                            if (e > 0) {
                                if (best == INT64_MIN) break;
                                if (best <= price[u]) { if (full) { price[u] = best - eps; ++relabels; } else { pend.push_back(u); } }
                                else if (!full) break;
                            }
                        }
                        excess[u] = e;
                    }
                    for (i32 v = 0; v < g.n; ++v) excess[v] += arrive[v];
                    // hubs that ended without an admissible arc relabel at the kernel boundary (graph frozen)
                    for (i32 u : pend) {
                        if (excess[u] <= 0) continue;
                        i64 best = INT64_MIN;
                        for (i64 a = g.row[u]; a < g.row[u + 1]; ++a) { if (cap[a] <= 0) continue; i64 key = price[g.head[a]] - (i64)g.cost[a] * S; if (key > best) best = key; }
                        if (best != INT64_MIN && best <= price[u]) { price[u] = best - eps; ++relabels; }
                    }
                    pend.clear();
                }
            } else if (!walk) {
            chased.assign(g.n, 0); was_active.assign(g.n, 0);
            for (i32 u : act) was_active[u] = 1;
            // push phase: prices frozen, excess snapshot (arrivals go to delta)
            for (i32 u : act) {
                i64 e = excess[u], e0 = e;
                i64 best = INT64_MIN;
                i64 deg = g.row[u + 1] - g.row[u];
                i64 off = deg ? (i64)(((uint64_t)u * 0x9E3779B97F4A7C15ULL + (uint64_t)rounds * 0xD1B54A32D192ED03ULL) >> 33) % deg : 0;
                std::vector<i64> order;
                if (deep) {  // deepest arcs first
                    for (i64 a = g.row[u]; a < g.row[u + 1]; ++a) order.push_back(a);
                    std::sort(order.begin(), order.end(), [&](i64 x, i64 y) { return price[g.head[x]] - (i64)g.cost[x] * S > price[g.head[y]] - (i64)g.cost[y] * S; });
                }
                for (i64 k = 0; k < deg; ++k) {
                    i64 a = deep ? order[k] : g.row[u] + (off + k) % deg;
                    ++scans;
                    if (cap[a] <= 0) continue;
                    i32 v = g.head[a];
                    i64 key = price[v] - (i64)g.cost[a] * S;
                    if (key > price[u] && e > 0) { // admissible
                        i64 d = std::min<i64>(e, cap[a]);
                        cap[a] -= (i32)d; cap[g.rev[a]] += (i32)d;
                        e -= d; delta[v] += d; ++pushes;
                        i64 kr = price[u] + (i64)g.cost[a] * S;
                        if (kr > cand[v]) cand[v] = kr;
                        if (cap[a] > 0 && key > best) best = key;
                        if (chase > 0) chase_from(v, chase);
                    } else if (key > best) best = key;
                }
                delta[u] -= (e0 - e);
                if (best > cand[u]) cand[u] = best;
            }
            for (i32 u : act) { excess[u] += delta[u]; delta[u] = 0; }
            for (i32 v = 0; v < g.n; ++v) if (delta[v]) { excess[v] += delta[v]; delta[v] = 0; }
            for (i32 u : act) {
                if (excess[u] > 0 && cand[u] <= price[u] && cand[u] != INT64_MIN) {
                    i64 np = cand[u] - eps;
                    if (deep && (g.row[u + 1] - g.row[u] >= deep_mindeg)) {
                        // discharge-depth relabel: down to the level at which the residual capacity of
                        // the admissible arcs covers (a fraction of) the excess
                        std::vector<std::pair<i64, i64>> kc;
                        for (i64 a = g.row[u]; a < g.row[u + 1]; ++a) if (cap[a] > 0) kc.push_back({price[g.head[a]] - (i64)g.cost[a] * S, cap[a]});
                        std::sort(kc.begin(), kc.end(), [](auto &x, auto &y) { return x.first > y.first; });
                        i64 need = std::max<i64>(1, excess[u] * deep / 100), cum = 0, T = kc.empty() ? cand[u] : kc[0].first;
                        for (auto &x : kc) { T = x.first; cum += x.second; if (cum >= need) break; }
                        np = std::min(np, T - eps);
                        ++deep_relabels;
                    }
                    price[u] = np; ++relabels;
                }
            }
            } else {
            // walk phase: prices frozen; arrivals are usable at once, newly active vertices join this sweep's list
            std::vector<i32> wl(act);
            std::vector<char> inl(g.n, 0);
            for (i32 u : wl) inl[u] = 1;
            i64 hops = 0;
            for (size_t qi = 0; qi < wl.size(); ++qi) {
                i32 u = wl[qi];
                inl[u] = 0;
                i64 e = excess[u];
                if (e <= 0) continue;
                ++hops;
                i64 deg = g.row[u + 1] - g.row[u];
                for (i64 k = 0; k < deg && e > 0; ++k) {
                    i64 a = g.row[u] + k;
                    ++scans;
                    if (cap[a] <= 0) continue;
                    i32 v = g.head[a];
                    i64 key = price[v] - (i64)g.cost[a] * S;
                    if (key > price[u]) {
                        i64 d = std::min<i64>(e, cap[a]);
                        cap[a] -= (i32)d; cap[g.rev[a]] += (i32)d;
                        e -= d; ++pushes;
                        bool was = excess[v] > 0;
                        excess[v] += d;
                        if (!was && excess[v] > 0 && !inl[v]) { inl[v] = 1; wl.push_back(v); }
                    }
                }
                excess[u] = e;
            }
            maxwl = std::max<i64>(maxwl, (i64)wl.size());
            // relabel phase: every vertex still active has no admissible residual arc
            for (i32 u = 0; u < g.n; ++u) {
                if (excess[u] <= 0) continue;
                i64 best = INT64_MIN;
                for (i64 a = g.row[u]; a < g.row[u + 1]; ++a) {
                    ++scans;
                    if (cap[a] <= 0) continue;
                    i64 key = price[g.head[a]] - (i64)g.cost[a] * S;
                    if (key > best) best = key;
                }
                if (best != INT64_MIN && best <= price[u]) { price[u] = best - eps; ++relabels; }
            }
            }
            for (i32 v = 0; v < g.n; ++v) cand[v] = INT64_MIN;
            if (getenv("TRACE")) printf("    r%ld act %zu pushes %ld relabels %ld\n", (long)local_round, act.size(), (long)(pushes - lastp), (long)(relabels - lastr));
            lastp = pushes; lastr = relabels;
            {
                static i64 since = 0;
                ++since;
                const i64 per = (small_active > 0 && (i64)act.size() < small_active) ? small_period : gu_period;
                if (per > 0 && since >= per) { global_update(eps); since = 0; if (getenv("TRACE")) printf("    -- GU iters so far %ld\n", (long)gu_iters); }
            }
        }
        printf("  eps %-14ld rounds %-6ld scans %-12ld pushes %-9ld relabels %-9ld gu_iters %-5ld gu_scans %-11ld\n", (long)eps,
               (long)(rounds - r0), (long)(scans - s0), (long)(pushes - p0), (long)(relabels - l0), (long)(gu_iters - gi0), (long)(gu_scans - gs0));
    }
    i64 solve() {
        i64 eps = 1;
        for (i64 a = 0; a < g.m2; ++a) if (cap[a] > 0) eps = std::max<i64>(eps, std::abs((i64)g.cost[a]) * S);
        std::fill(cand.begin(), cand.end(), INT64_MIN);
        if (eps0_cap > 0) eps = std::min<i64>(eps, eps0_cap * S);
        if (const char *sc = getenv("EPS_SCHED")) {  // explicit shifts, e.g. "20,16,12,0"
            std::vector<int> sh; const char *q = sc;
            while (*q) { sh.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; }
            for (int x : sh) refine((i64)1 << x);
        } else
        do { eps = std::max<i64>(1, eps / alpha); refine(eps); } while (eps > 1);
        i64 tc = 0;
        for (i64 a = 0; a < g.m2; ++a) { i64 f = (i64)g.cap[a] - cap[a]; if (f > 0) tc += f * g.cost[a]; }
        return tc;
    }
};

int main(int argc, char **argv) {
    Graph g;
    if (argc < 2 || !load(argv[1], g)) { fprintf(stderr, "usage: sync_sim graph.bin [alpha] [gu_period] [sat_mode]\n"); return 1; }
    Sim s(g);
    if (argc > 2) s.alpha = atoi(argv[2]);
    if (argc > 3) s.gu_period = atoi(argv[3]);
    if (argc > 4) s.sat_mode = atoi(argv[4]);
    if (argc > 5) s.eps0_cap = atoll(argv[5]);
    if (argc > 7) { s.small_active = atoi(argv[6]); s.small_period = atoi(argv[7]); }
    if (getenv("WALK")) s.walk = 1;
    if (getenv("CHASE")) s.chase = atoi(getenv("CHASE"));
    if (getenv("DISCH")) {
        s.disch = 1;
        s.inX.assign(g.n, 1); s.hasYnbr.assign(g.n, 0);
        for (i32 v = 0; v < g.n; ++v) { i64 dv = g.row[v+1]-g.row[v]; for (i64 a = g.row[v]; a < g.row[v+1]; ++a) { i32 u = g.head[a]; if (u == v) continue; i64 du = g.row[u+1]-g.row[u]; if (!(du > dv || (du == dv && u > v))) s.inX[v] = 0; } }
        for (i32 v = 0; v < g.n; ++v) if (!s.inX[v]) for (i64 a = g.row[v]; a < g.row[v+1]; ++a) { i32 u = g.head[a]; if (u != v && !s.inX[u] && g.supply[u] >= 0) s.hasYnbr[v] = 1; }
    }
    if (getenv("TAILACT")) s.tailact = atoll(getenv("TAILACT"));
    if (getenv("TAILCUT")) s.tailcut = atoll(getenv("TAILCUT"));
    if (getenv("DEEP")) s.deep = atoi(getenv("DEEP"));
    if (getenv("DEEPDEG")) s.deep_mindeg = atoll(getenv("DEEPDEG"));
    if (getenv("HUBDEG")) s.hubdeg = atoll(getenv("HUBDEG"));
    auto t0 = std::chrono::steady_clock::now();
    i64 tc = s.solve();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (s.disch) printf("kernels %ld\n", (long)s.kernels);
    printf("cost %ld  rounds %ld scans %ld (%.1f x m2) pushes %ld relabels %ld | gus %ld gu_iters %ld gu_scans %ld (%.1f x m2) | sat_scans %ld | sim %.0f ms\n",
           (long)tc, (long)s.rounds, (long)s.scans, (double)s.scans / g.m2, (long)s.pushes, (long)s.relabels, (long)s.gus, (long)s.gu_iters,
           (long)s.gu_scans, (double)s.gu_scans / g.m2, (long)s.sat_scans, ms);
    return 0;
}
