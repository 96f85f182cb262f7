// DIMACS minimum-cost-flow text format, host side (include/psn_builder.h):
//   p min <vertices> <arcs> / n <vertex> <supply> / a <tail> <head> <low> <cap> <cost>
// with 1-based vertex ids, and the residual-CSR builder for plain arc lists that a reader
// needs.  This is the interchange format of the CPU solvers Firmament drives (named at
// /root/reference/LICENSE:18-26), so a network solved here can be handed to one of them
// and back.
#include <algorithm>
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/psn.h"
#include "../../include/psn_builder.h"

namespace {
// One line of at most 510 characters; a longer one is refused rather than split (its tail would be
// parsed as a record of its own).  1: a line was read, 0: end of file, -1: over-long line.
int read_line(FILE *f, char *line, size_t size) {
    if (!std::fgets(line, (int)size, f)) return 0;
    const size_t len = std::strlen(line);
    if (len + 1 == size && line[len - 1] != '\n') {
        const int c = std::fgetc(f);
        if (c != EOF) return -1;
    }
    return 1;
}
}  // namespace

extern "C" {

int psn_csr_from_arcs(int32_t n, int64_t m, const int32_t *tail, const int32_t *head, const int32_t *cap, const int32_t *cost,
                      int64_t *row_off, int32_t *r_head, int32_t *r_cap, int32_t *r_cost, int32_t *r_rev, int64_t *fwd_pos) {
    if (n <= 0 || m < 0 || 2 * m >= (int64_t)0x7FFFFFF0 || !row_off || (m > 0 && (!tail || !head || !cap || !cost || !r_head || !r_cap || !r_cost || !r_rev)))
        return PSN_EINVAL;
    std::fill(row_off, row_off + n + 1, (int64_t)0);
    for (int64_t i = 0; i < m; ++i) {
        if (tail[i] < 0 || tail[i] >= n || head[i] < 0 || head[i] >= n || cap[i] < 0) return PSN_EINVAL;
        ++row_off[tail[i] + 1];
        ++row_off[head[i] + 1];
    }
    for (int32_t v = 0; v < n; ++v) row_off[v + 1] += row_off[v];
    std::vector<int64_t> fill(row_off, row_off + n);
    for (int64_t i = 0; i < m; ++i) {  // row contents in arc-list order, forward entry and reverse entry alike
        const int64_t a = fill[tail[i]]++, r = fill[head[i]]++;
        r_head[a] = head[i]; r_cap[a] = cap[i]; r_cost[a] = cost[i]; r_rev[a] = (int32_t)r;
        r_head[r] = tail[i]; r_cap[r] = 0; r_cost[r] = -cost[i]; r_rev[r] = (int32_t)a;
        if (fwd_pos) fwd_pos[i] = a;
    }
    return PSN_OK;
}

int psn_dimacs_write(const char *path, int32_t n, int64_t m2, const int64_t *row_off, const int32_t *head, const int32_t *cap,
                     const int32_t *cost, const int64_t *supply, const char *comment) {
    if (!path || n <= 0 || m2 < 0 || !row_off || !supply || (m2 > 0 && (!head || !cap || !cost))) return PSN_EINVAL;
    FILE *f = std::fopen(path, "w");
    if (!f) return PSN_EINVAL;
    int64_t arcs = 0;
    for (int64_t a = 0; a < m2; ++a) arcs += cap[a] > 0;
    if (comment) std::fprintf(f, "c %s\n", comment);
    std::fprintf(f, "p min %d %" PRId64 "\n", n, arcs);
    for (int32_t v = 0; v < n; ++v)
        if (supply[v]) std::fprintf(f, "n %d %" PRId64 "\n", v + 1, supply[v]);
    for (int32_t v = 0; v < n; ++v)
        for (int64_t a = row_off[v]; a < row_off[v + 1]; ++a)
            if (cap[a] > 0) std::fprintf(f, "a %d %d 0 %d %d\n", v + 1, head[a] + 1, cap[a], cost[a]);
    const bool bad = std::ferror(f) != 0;
    return (std::fclose(f) != 0 || bad) ? PSN_EINVAL : PSN_OK;
}

int psn_dimacs_size(const char *path, int32_t *n, int64_t *m) {
    if (!path || !n || !m) return PSN_EINVAL;
    FILE *f = std::fopen(path, "r");
    if (!f) return PSN_EINVAL;
    char line[512];
    int rc = PSN_EINVAL;
    while (read_line(f, line, sizeof line) == 1) {
        if (line[0] != 'p') continue;
        long long nn = 0, mm = 0;
        char kind[16] = {0};
        if (std::sscanf(line, "p %15s %lld %lld", kind, &nn, &mm) == 3 && std::strcmp(kind, "min") == 0 && nn > 0 && nn <= INT32_MAX && mm >= 0) {
            *n = (int32_t)nn; *m = mm; rc = PSN_OK;
        }
        break;
    }
    std::fclose(f);
    return rc;
}

static int dimacs_read_impl(const char *path, int32_t n, int64_t m, int32_t *tail, int32_t *head, int32_t *cap, int32_t *cost,
                            int64_t *supply, int64_t *base_cost, int32_t *low) {
    if (!path || n <= 0 || m < 0 || !supply || (m > 0 && (!tail || !head || !cap || !cost))) return PSN_EINVAL;
    FILE *f = std::fopen(path, "r");
    if (!f) return PSN_EINVAL;
    std::fill(supply, supply + n, (int64_t)0);
    std::vector<int64_t> shift((size_t)n, 0);  // supply moved by lower bounds; added at the end, whatever order the lines come in
    int64_t base = 0, got = 0;
    char line[512];
    int rc = PSN_OK, lr = 0;
    while (rc == PSN_OK && (lr = read_line(f, line, sizeof line)) != 0) {
        if (lr < 0) { rc = PSN_EINVAL; break; }
        if (line[0] == 'n') {
            long long v = 0, s = 0;
            if (std::sscanf(line, "n %lld %lld", &v, &s) != 2 || v < 1 || v > n) rc = PSN_EINVAL;
            else supply[v - 1] = s;
        } else if (line[0] == 'a') {
            long long u = 0, v = 0, lo = 0, hi = 0, c = 0;
            if (std::sscanf(line, "a %lld %lld %lld %lld %lld", &u, &v, &lo, &hi, &c) != 5 || u < 1 || u > n || v < 1 || v > n ||
                lo < 0 || hi < lo || hi - lo > INT32_MAX || c < -(long long)INT32_MAX || c > INT32_MAX || got >= m) { rc = PSN_EINVAL; break; }
            // a lower bound is flow that must be there: route it up front
            shift[(size_t)(u - 1)] -= lo; shift[(size_t)(v - 1)] += lo; base += lo * c;
            tail[got] = (int32_t)(u - 1); head[got] = (int32_t)(v - 1); cap[got] = (int32_t)(hi - lo); cost[got] = (int32_t)c;
            if (low) low[got] = (int32_t)std::min<long long>(lo, INT32_MAX);
            ++got;
        }
    }
    std::fclose(f);
    if (rc == PSN_OK && got != m) rc = PSN_EINVAL;
    for (int32_t v = 0; v < n; ++v) supply[v] += shift[(size_t)v];
    if (base_cost) *base_cost = base;
    return rc;
}

int psn_dimacs_read(const char *path, int32_t n, int64_t m, int32_t *tail, int32_t *head, int32_t *cap, int32_t *cost,
                    int64_t *supply, int64_t *base_cost) {
    return dimacs_read_impl(path, n, m, tail, head, cap, cost, supply, base_cost, nullptr);
}
int psn_dimacs_read_lower(const char *path, int32_t n, int64_t m, int32_t *tail, int32_t *head, int32_t *cap, int32_t *cost,
                          int64_t *supply, int64_t *base_cost, int32_t *low) {
    if (m > 0 && !low) return PSN_EINVAL;
    return dimacs_read_impl(path, n, m, tail, head, cap, cost, supply, base_cost, low);
}

// ---- change records: a second file that edits a problem already read
//   x <tail> <head> <low> <cap> <cost> [<k>]   new capacity and cost of the k-th (default first) arc tail -> head
//                                              of the problem file, counting parallel arcs in file order; low must be 0
//   n <vertex> <supply>                         new supply of a vertex
// Arcs are neither added nor removed (capacity 0 takes an arc out): the resident network keeps its
// layout, which is what lets the device copy be edited in place and the solve warm-started.
namespace {
struct ChangeLine { char kind; long long a, b, lo, hi, c, k; };
int parse_changes(const char *path, std::vector<ChangeLine> &out) {
    FILE *f = std::fopen(path, "r");
    if (!f) return PSN_EINVAL;
    char line[512];
    int rc = PSN_OK, lr = 0;
    while (rc == PSN_OK && (lr = read_line(f, line, sizeof line)) != 0) {
        if (lr < 0) { rc = PSN_EINVAL; break; }
        ChangeLine ch{};
        if (line[0] == 'x') {
            ch.kind = 'x';
            const int got = std::sscanf(line, "x %lld %lld %lld %lld %lld %lld", &ch.a, &ch.b, &ch.lo, &ch.hi, &ch.c, &ch.k);
            if (got < 5 || ch.lo != 0 || ch.hi < 0 || ch.hi > INT32_MAX || ch.c < -(long long)INT32_MAX || ch.c > INT32_MAX || (got == 6 && ch.k < 0)) rc = PSN_EINVAL;
            if (got < 6) ch.k = 0;
            out.push_back(ch);
        } else if (line[0] == 'n') {
            ch.kind = 'n';
            if (std::sscanf(line, "n %lld %lld", &ch.a, &ch.b) != 2) rc = PSN_EINVAL;
            out.push_back(ch);
        } else if (line[0] == 'a' || line[0] == 'p' || line[0] == 'r' || line[0] == 'd') {
            rc = PSN_EINVAL;  // adding or removing arcs / vertices is not an in-place edit
        }
    }
    std::fclose(f);
    return rc;
}
}  // namespace

int psn_dimacs_changes_size(const char *path, int64_t *n_arc_changes, int64_t *n_supply_changes) {
    if (!path || !n_arc_changes || !n_supply_changes) return PSN_EINVAL;
    std::vector<ChangeLine> ch;
    const int rc = parse_changes(path, ch);
    if (rc != PSN_OK) return rc;
    *n_arc_changes = *n_supply_changes = 0;
    for (const ChangeLine &c : ch) ++*(c.kind == 'x' ? n_arc_changes : n_supply_changes);
    return PSN_OK;
}

static int read_changes_impl(const char *path, int32_t n, int64_t m, const int32_t *tail, const int32_t *head, const int32_t *low, int64_t *arc,
                             int32_t *new_cap, int32_t *new_cost, int32_t *vertex, int64_t *new_supply) {
    if (!path || n <= 0 || m < 0 || (m > 0 && (!tail || !head))) return PSN_EINVAL;
    std::vector<ChangeLine> ch;
    int rc = parse_changes(path, ch);
    if (rc != PSN_OK) return rc;
    // arcs of the problem by (tail, head), parallel arcs in file order
    std::vector<int64_t> order((size_t)m);
    for (int64_t i = 0; i < m; ++i) order[(size_t)i] = i;
    auto key = [&](int64_t i) { return ((uint64_t)(uint32_t)tail[i] << 32) | (uint32_t)head[i]; };
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return key(x) != key(y) ? key(x) < key(y) : x < y; });
    int64_t na = 0, ns = 0;
    for (const ChangeLine &c : ch) {
        if (c.kind == 'n') {
            if (c.a < 1 || c.a > n || !vertex || !new_supply) return PSN_EINVAL;
            vertex[ns] = (int32_t)(c.a - 1); new_supply[ns] = c.b; ++ns;
            continue;
        }
        if (c.a < 1 || c.a > n || c.b < 1 || c.b > n || !arc || !new_cap || !new_cost) return PSN_EINVAL;
        const uint64_t want = ((uint64_t)(uint32_t)(c.a - 1) << 32) | (uint32_t)(c.b - 1);
        auto it = std::lower_bound(order.begin(), order.end(), want, [&](int64_t i, uint64_t w) { return key(i) < w; });
        if (it == order.end() || key(*it) != want || (order.end() - it) <= c.k || key(*(it + c.k)) != want) return PSN_EINVAL;  // no such arc
        // An arc whose lower bound the reader folded away (capacity hi - lo, lo units moved into the
        // supplies and the base cost) cannot be edited by a record that only carries the new upper
        // bound and cost: the residual capacity and the base cost would both come out wrong.
        if (low && low[*(it + c.k)] != 0) return PSN_EINVAL;
        arc[na] = *(it + c.k); new_cap[na] = (int32_t)c.hi; new_cost[na] = (int32_t)c.c; ++na;
    }
    return PSN_OK;
}
int psn_dimacs_read_changes(const char *path, int32_t n, int64_t m, const int32_t *tail, const int32_t *head, int64_t *arc,
                            int32_t *new_cap, int32_t *new_cost, int32_t *vertex, int64_t *new_supply) {
    return read_changes_impl(path, n, m, tail, head, nullptr, arc, new_cap, new_cost, vertex, new_supply);
}
int psn_dimacs_read_changes_lower(const char *path, int32_t n, int64_t m, const int32_t *tail, const int32_t *head, const int32_t *low,
                                  int64_t *arc, int32_t *new_cap, int32_t *new_cost, int32_t *vertex, int64_t *new_supply) {
    return read_changes_impl(path, n, m, tail, head, low, arc, new_cap, new_cost, vertex, new_supply);
}

}  // extern "C"
