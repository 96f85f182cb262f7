// Flow-graph builder (host side above the solver ABI): turns the cluster state that the
// TaskSubmitted / NodeAdded stream accumulates
// (/root/reference/pkg/k8sclient/podwatcher.go:377-410 builds the task descriptors,
// /root/reference/pkg/k8sclient/nodewatcher.go:292-339 the machine + single PU topology)
// into the residual CSR that psn_upload_csr takes, and generates the synthetic cluster
// family psn-synth-v1 used by the benchmark.  Slot-arc costs are left at zero here; the
// device prices them (psn_eval_costs).
//
// Vertex ids: 0 sink | 1..N machines | then E task classes | then J unscheduled
// aggregators | then P tasks.  Row contents, in order:
//   sink      reverse of machine->sink (by machine), reverse of unsched->sink (by job)
//   machine   reverse of class->machine slot arcs (class-major, slot-minor), machine->sink
//   class e   reverse of task->class (tasks of the class, by task id), slot arcs (machine-major)
//   unsched j reverse of task->unsched (tasks of the job, by task id), unsched->sink
//   task      task->class, task->unsched
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/psn.h"
#include "../../include/psn_builder.h"

namespace {
const uint64_t kMagic = 0x506F736569646F6EULL;
const int64_t kCpuChoices[5] = {4000, 8000, 16000, 32000, 64000};
const int64_t kGibPerCore[3] = {2, 4, 8};
const int64_t kReqCpu[8] = {10, 50, 100, 250, 500, 1000, 2000, 4000};
const int64_t kReqMemMi[8] = {10, 64, 128, 256, 512, 1024, 2048, 4096};

struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
};
}  // namespace

extern "C" {

int psn_synth_cluster(int32_t config_id, uint64_t seed, const psn_layout *lay, int64_t *cap_cpu, int64_t *cap_mem,
                      int64_t *used_cpu, int64_t *used_mem, int64_t *req_cpu, int64_t *req_mem, int32_t *task_class,
                      int32_t *task_job) {
    if (!lay || !cap_cpu || !cap_mem || !used_cpu || !used_mem || !req_cpu || !req_mem || !task_class || !task_job) return PSN_EINVAL;
    const int N = lay->n_machines, P = lay->n_tasks, J = lay->n_jobs, E = lay->n_classes;
    if (N <= 0 || P <= 0 || J <= 0 || E != 64) return PSN_EINVAL;
    SplitMix64 r{kMagic ^ ((uint64_t)(uint32_t)config_id << 32) ^ seed};
    for (int m = 0; m < N; ++m) {
        cap_cpu[m] = kCpuChoices[r.next() % 5];
        cap_mem[m] = (cap_cpu[m] / 1000) * kGibPerCore[r.next() % 3] * 1048576;
        used_cpu[m] = cap_cpu[m] * (int64_t)(r.next() % 601) / 1000;
        used_mem[m] = cap_mem[m] * (int64_t)(r.next() % 601) / 1000;
    }
    for (int t = 0; t < P; ++t) {
        task_class[t] = (int32_t)(r.next() % (uint64_t)E);
        task_job[t] = (int32_t)(r.next() % (uint64_t)J);
    }
    for (int e = 0; e < E; ++e) {
        req_cpu[e] = kReqCpu[e / 8];
        req_mem[e] = kReqMemMi[e % 8] * 1024;
    }
    return PSN_OK;
}

int psn_network_size(const psn_layout *lay, int32_t *n, int64_t *m2) {
    if (!lay || !n || !m2) return PSN_EINVAL;
    const int64_t N = lay->n_machines, E = lay->n_classes, J = lay->n_jobs, P = lay->n_tasks, K = lay->k_slots;
    if (N <= 0 || E <= 0 || J <= 0 || P <= 0 || K <= 0) return PSN_EINVAL;
    const int64_t nn = 1 + N + E + J + P;
    const int64_t m = 2 * P + E * N * K + N + J;
    if (nn > INT32_MAX || 2 * m >= (int64_t)0x7FFFFFF0) return PSN_EOVERFLOW;
    *n = (int32_t)nn;
    *m2 = 2 * m;
    return PSN_OK;
}

int psn_build_network(const psn_layout *lay, int32_t slots, int32_t unsched_cost, const int32_t *task_class,
                      const int32_t *task_job, int64_t *row_off, int32_t *head, int32_t *cap, int32_t *cost, int32_t *rev,
                      int64_t *supply, int64_t *ec_arc_base) {
    int32_t n;
    int64_t m2;
    int rc = psn_network_size(lay, &n, &m2);
    if (rc != PSN_OK) return rc;
    if (!task_class || !task_job || !row_off || !head || !cap || !cost || !rev || !supply || !ec_arc_base || slots < 0 || unsched_cost < 0)
        return PSN_EINVAL;
    const int64_t N = lay->n_machines, E = lay->n_classes, J = lay->n_jobs, P = lay->n_tasks, K = lay->k_slots;
    std::vector<int64_t> cnt_e(E, 0), cnt_j(J, 0);
    for (int64_t t = 0; t < P; ++t) {
        if (task_class[t] < 0 || task_class[t] >= E || task_job[t] < 0 || task_job[t] >= J) return PSN_EINVAL;
        ++cnt_e[task_class[t]];
        ++cnt_j[task_job[t]];
    }
    const int64_t v_m = 1, v_ec = 1 + N, v_un = 1 + N + E, v_t = 1 + N + E + J;
    // row offsets
    row_off[0] = 0;
    row_off[1] = N + J;
    for (int64_t m = 0; m < N; ++m) row_off[v_m + m + 1] = row_off[v_m + m] + E * K + 1;
    for (int64_t e = 0; e < E; ++e) row_off[v_ec + e + 1] = row_off[v_ec + e] + cnt_e[e] + N * K;
    for (int64_t j = 0; j < J; ++j) row_off[v_un + j + 1] = row_off[v_un + j] + cnt_j[j] + 1;
    for (int64_t t = 0; t < P; ++t) row_off[v_t + t + 1] = row_off[v_t + t] + 2;
    if (row_off[n] != m2) return PSN_EINVAL;
    std::memset(supply, 0, sizeof(int64_t) * (size_t)n);
    supply[0] = -P;
    auto put = [&](int64_t a, int64_t r, int64_t to, int64_t from, int32_t cp, int32_t cs) {
        head[a] = (int32_t)to; cap[a] = cp; cost[a] = cs; rev[a] = (int32_t)r;
        head[r] = (int32_t)from; cap[r] = 0; cost[r] = -cs; rev[r] = (int32_t)a;
    };
    std::vector<int64_t> fill_e(E, 0), fill_j(J, 0);
    for (int64_t t = 0; t < P; ++t) {
        const int64_t e = task_class[t], j = task_job[t];
        const int64_t ta = row_off[v_t + t];
        put(ta, row_off[v_ec + e] + fill_e[e]++, v_ec + e, v_t + t, 1, 0);
        put(ta + 1, row_off[v_un + j] + fill_j[j]++, v_un + j, v_t + t, 1, unsched_cost);
        supply[v_t + t] = 1;
    }
    for (int64_t e = 0; e < E; ++e) {
        const int64_t base = row_off[v_ec + e] + cnt_e[e];
        ec_arc_base[e] = base;
        for (int64_t m = 0; m < N; ++m)
            for (int64_t k = 0; k < K; ++k)
                put(base + m * K + k, row_off[v_m + m] + e * K + k, v_m + m, v_ec + e, 0, 0);
    }
    for (int64_t m = 0; m < N; ++m) put(row_off[v_m + m] + E * K, row_off[0] + m, 0, v_m + m, slots, 0);
    for (int64_t j = 0; j < J; ++j) put(row_off[v_un + j] + cnt_j[j], row_off[0] + N + j, 0, v_un + j, (int32_t)cnt_j[j], 0);
    return PSN_OK;
}

uint64_t psn_fnv1a64(const void *data, int64_t nbytes, uint64_t h) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (int64_t i = 0; i < nbytes; ++i) { h ^= p[i]; h *= 0x100000001B3ULL; }
    return h;
}

}  // extern "C"
