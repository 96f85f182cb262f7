// Scheduler service core (include/psn_service.h): task / machine state machine with the
// reply codes of the Firmament scheduler RPCs, a resident flow network with head-room that
// is edited in place, and delta generation from consecutive placements.
//
// Network layout (vertex ids as psn_layout describes, sized by capacities, not by what is
// in use): 0 sink | machines | classes | jobs (unscheduled aggregators) | task slots.
//   sink row      reverse of machine->sink, reverse of job->sink
//   machine row   reverse of class->machine slot arcs (class-major, slot-minor), machine->sink
//   class row     `room` reverse places for member tasks, then slot arcs (machine-major)
//   job row       `room` reverse places for member tasks, then job->sink
//   task row      task->class, task->job
// A reverse place that no task uses, and both arcs of a free task slot, are placeholders:
// self-loops paired with themselves, capacity 0.  A task that arrives takes a free task
// slot and one free place in its class row and in its job row: four arcs change.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <thread>
#include <vector>

#include "../../include/psn.h"
#include "../../include/psn_builder.h"
#include "../../include/psn_service.h"

namespace {

typedef int64_t i64;
const i64 kDeadQuantity = (i64)1 << 40;  // capacity == usage: nothing fits on a machine that is not there

enum TaskState { kRunnable = 0, kCompleted = 1, kFailed = 2 };

struct Selector {  // /root/reference/pkg/firmament/label_selector.proto:24-35
    int type = 0;
    std::string key;
    std::vector<std::string> values;  // sorted
};
struct Machine {
    std::string uuid;
    std::map<std::string, std::string> labels;
    i64 cap_cpu = 0, cap_mem = 0, used_cpu = 0, used_mem = 0;
    i64 meas_cpu = -1, meas_mem = -1;          // measured usage of the whole machine (ResourceStats), -1: none reported
    i64 committed_cpu = 0, committed_mem = 0;  // requests of the tasks running here (keep-running mode)
    int running = 0;                           // how many they are: each holds one of the machine's slots
    bool live = false;
    uint32_t gen = 0;  // bumped whenever the index changes hands
};
struct Pool {  // reverse places of one class or job row
    i64 first = 0;       // arc index of place 0
    int room = 0;
    int count = 0;       // places in use
    std::vector<int> free_places;
};
struct Class {
    i64 cpu = 0, ram = 0;
    std::vector<Selector> selectors;  // all must hold
    Pool pool;
};
struct Job {
    std::string uuid;
    bool used = false;
    int known = 0;  // tasks of the job the service knows, runnable or not
    Pool pool;
};
struct Task {
    uint64_t uid = 0;
    int state = kRunnable;
    int cls = -1, job = -1;
    int slot = -1;               // task slot in the network, -1: not in the network
    i64 place_class = -1, place_job = -1;  // arc indices of its reverse places
    int machine = -1;            // placement after the last round
    uint32_t machine_gen = 0;
    bool pinned = false;         // running on `machine` and therefore outside the network (keep-running mode)
    i64 obs_cpu = -1, obs_mem = -1;  // observed usage (TaskStats), -1: none reported
    i64 cpu = 0, ram = 0;
};
struct Delta { uint64_t uid; int32_t type; int machine; };

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct psn_svc {
    psn_svc_config cfg{};
    psn_ctx *ctx = nullptr;
    std::string err;
    // entities
    std::vector<Machine> machines;
    std::unordered_map<std::string, int> machine_by_uuid;
    std::vector<int> free_machines;
    std::vector<Class> classes;
    std::map<std::pair<std::pair<i64, i64>, std::string>, int> class_by_shape;  // (cpu, ram, canonical selectors)
    std::vector<uint32_t> allow;  // admission mask, E rows of (N + 31) / 32 words
    bool any_selector = false, mask_dirty = false;
    std::vector<Job> jobs;
    std::unordered_map<std::string, int> job_by_uuid;
    std::vector<int> free_jobs;
    std::unordered_map<uint64_t, Task> tasks;
    std::vector<uint64_t> slot_uid;  // task slot -> uid (0: free)
    std::vector<char> slot_used;
    std::vector<Task *> slot_task;   // task slot -> its entry in `tasks` (map nodes do not move)
    std::vector<int> free_slots;
    i64 runnable = 0;
    // network mirror
    bool laid_out = false, need_layout = true, resident = false;
    int N = 0, E = 0, J = 0, P = 0, K = 8;
    psn_layout lay{};
    int32_t n = 0;
    i64 m2 = 0;
    int v_ec = 0, v_un = 0, v_t = 0;
    // The network is kept as what determines it (row offsets, rooms, which task sits where); the arc
    // arrays themselves exist on the device, and on the host only when somebody asks for them
    // (materialize).  Arcs rewired since the device copy was last in sync are kept by value.
    std::vector<i64> row_off, supply, ec_arc_base;
    std::vector<int32_t> room_e, room_j;
    struct ArcValue { int32_t head, cap, cost; i64 rev; };
    // changes since the device copy was last in sync
    std::unordered_map<i64, ArcValue> dirty_arcs;
    std::unordered_map<i64, int32_t> cap_updates;
    std::unordered_set<int32_t> dirty_supply;
    std::vector<int32_t> new_sources;
    bool reprice = false;
    bool settled = false;  // the last round ran to its end and nothing has been queued since
    bool orphans = false;  // a machine went away: tasks pinned to it have to come back into the network
    i64 attached = 0;      // tasks in the network
    i64 pinned_total = 0;  // tasks running (outside the network)
    // results
    std::vector<Delta> deltas;
    psn_svc_stats st{};
    // usage of each machine by things the service does not manage, as the cost model and the admission
    // check see it; follows the AddNodeStats figures, or the measured whole-machine usage less what the
    // service's own running tasks use there
    std::vector<i64> ext_cpu, ext_mem;
    bool any_measured = false;

    int fail(int code, const std::string &what) { err = what; return code; }

    // ---------------------------------------------------------------- mirror edits
    void set_arc(i64 a, int h, int cp, int cs, i64 r) {
        if (resident) dirty_arcs[a] = ArcValue{h, cp, cs, r};
    }
    void park(i64 a, int own_vertex) { set_arc(a, own_vertex, 0, 0, a); }
    void set_cap(i64 a, int cp) {
        if (!resident) return;
        auto it = dirty_arcs.find(a);
        if (it != dirty_arcs.end()) it->second.cap = cp; else cap_updates[a] = cp;
    }
    void set_supply(int v, i64 s) {
        supply[(size_t)v] = s;
        if (resident) dirty_supply.insert(v);
    }
    i64 take_place(Pool &p) {
        if (p.free_places.empty()) return -1;
        const int i = p.free_places.back();
        p.free_places.pop_back();
        ++p.count;
        return p.first + i;
    }
    void give_place(Pool &p, i64 arc) {
        p.free_places.push_back((int)(arc - p.first));
        --p.count;
    }
    i64 job_sink_arc(int j) const { return jobs[(size_t)j].pool.first + jobs[(size_t)j].pool.room; }
    i64 machine_sink_arc(int m) const { return row_off[(size_t)(1 + m)] + (i64)E * K; }

    // Put a runnable task into the laid-out network; false when some head-room is used up.
    bool attach(Task &t) {
        if (free_slots.empty()) return false;
        Class &c = classes[(size_t)t.cls];
        Job &jb = jobs[(size_t)t.job];
        if (c.pool.free_places.empty() || jb.pool.free_places.empty()) return false;
        t.slot = free_slots.back();
        free_slots.pop_back();
        slot_uid[(size_t)t.slot] = t.uid;
        slot_used[(size_t)t.slot] = 1;
        slot_task[(size_t)t.slot] = &t;
        t.place_class = take_place(c.pool);
        t.place_job = take_place(jb.pool);
        const int vt = v_t + t.slot;
        const i64 ta = row_off[(size_t)vt];
        set_arc(ta, v_ec + t.cls, 1, 0, t.place_class);
        set_arc(t.place_class, vt, 0, 0, ta);
        set_arc(ta + 1, v_un + t.job, 1, cfg.unsched_cost, t.place_job);
        set_arc(t.place_job, vt, 0, -cfg.unsched_cost, ta + 1);
        set_cap(job_sink_arc(t.job), jb.pool.count);
        set_supply(vt, 1);
        set_supply(0, supply[0] - 1);
        if (resident) new_sources.push_back(vt);
        ++attached;
        return true;
    }
    void detach(Task &t) {
        if (t.slot < 0) return;
        const int vt = v_t + t.slot;
        const i64 ta = row_off[(size_t)vt];
        park(ta, vt);
        park(ta + 1, vt);
        park(t.place_class, v_ec + t.cls);
        park(t.place_job, v_un + t.job);
        give_place(classes[(size_t)t.cls].pool, t.place_class);
        give_place(jobs[(size_t)t.job].pool, t.place_job);
        set_cap(job_sink_arc(t.job), jobs[(size_t)t.job].pool.count);
        set_supply(vt, 0);
        set_supply(0, supply[0] + 1);
        slot_uid[(size_t)t.slot] = 0;
        slot_used[(size_t)t.slot] = 0;
        slot_task[(size_t)t.slot] = nullptr;
        free_slots.push_back(t.slot);
        t.slot = -1;
        t.place_class = t.place_job = -1;
        --attached;
    }

    // ---------------------------------------------------------------- running tasks (keep-running mode)
    bool keep_running() const { return cfg.reschedule_running == 0; }
    bool on_its_machine(const Task &t) const {
        return t.machine >= 0 && (size_t)t.machine < machines.size() && machines[(size_t)t.machine].gen == t.machine_gen;
    }
    int sink_capacity(int m) const {
        if ((size_t)m >= machines.size() || !machines[(size_t)m].live) return 0;
        return std::max(0, cfg.slots_per_machine - machines[(size_t)m].running);
    }
    void account(int m, i64 dcpu, i64 dmem, int drunning) {
        Machine &mc = machines[(size_t)m];
        mc.committed_cpu += dcpu; mc.committed_mem += dmem; mc.running += drunning;
        if (laid_out && !need_layout && m < N) {
            if (drunning) set_cap(machine_sink_arc(m), sink_capacity(m));
            reprice = true;
        }
    }
    // The task starts running on machine m: it leaves the network and its request counts as usage there.
    void pin(Task &t, int m) {
        if (t.slot >= 0) {
            if (need_layout) {  // the network is about to be laid out again: no point in editing it
                slot_uid[(size_t)t.slot] = 0; slot_used[(size_t)t.slot] = 0; slot_task[(size_t)t.slot] = nullptr;
                t.slot = -1; t.place_class = t.place_job = -1;
                --attached;
            } else {
                detach(t);
            }
        }
        t.pinned = true;
        ++pinned_total;
        account(m, t.cpu, t.ram, 1);
    }
    // The task stops running (it left, or is about to be described anew).
    void unpin(Task &t) {
        if (!t.pinned) return;
        t.pinned = false;
        --pinned_total;
        if (on_its_machine(t)) account(t.machine, -t.cpu, -t.ram, -1);
    }

    // Tasks that ran on a machine that went away are pending again: back into the network.
    void readmit_orphans() {
        if (!orphans) return;
        for (auto &kv : tasks) {
            Task &t = kv.second;
            if (!t.pinned || on_its_machine(t)) continue;
            t.pinned = false;
            --pinned_total;
            t.machine = -1;
            if (laid_out && !need_layout && !attach(t)) need_layout = true;
        }
        orphans = false;
    }

    // ---------------------------------------------------------------- entity tables
    static std::string canonical(std::vector<Selector> &sels) {
        for (auto &x : sels) std::sort(x.values.begin(), x.values.end());
        std::sort(sels.begin(), sels.end(), [](const Selector &a, const Selector &b) {
            return std::tie(a.key, a.type, a.values) < std::tie(b.key, b.type, b.values);
        });
        std::string out;
        for (auto &x : sels) {
            out += (char)('0' + x.type); out += '\x1f'; out += x.key;
            for (auto &v : x.values) { out += '\x1e'; out += v; }
            out += '\x1d';
        }
        return out;
    }
    bool admits(const Class &c, const Machine &m) const {
        for (const Selector &x : c.selectors) {
            auto it = m.labels.find(x.key);
            const bool has = it != m.labels.end();
            const bool in = has && std::binary_search(x.values.begin(), x.values.end(), it->second);
            switch (x.type) {
                case 0: if (!in) return false; break;   // IN_SET
                case 1: if (in) return false; break;    // NOT_IN_SET
                case 2: if (!has) return false; break;  // EXISTS_KEY
                case 3: if (has) return false; break;   // NOT_EXISTS_KEY
                default: return false;
            }
        }
        return true;
    }
    int mask_words() const { return (N + 31) / 32; }
    void mask_set(int e, int m, bool on) {
        if (!laid_out || e >= E || m >= N) return;
        uint32_t &w = allow[(size_t)e * mask_words() + (m >> 5)];
        const uint32_t bit = 1u << (m & 31), old = w;
        w = on ? (w | bit) : (w & ~bit);
        if (w != old) { mask_dirty = true; reprice = true; }
    }
    void mask_machine(int m) {  // column of one machine
        const Machine &mc = machines[(size_t)m];
        for (size_t e = 0; e < classes.size(); ++e) mask_set((int)e, m, mc.live && admits(classes[e], mc));
    }
    void mask_class(int e) {  // row of one class
        for (size_t m = 0; m < machines.size(); ++m) mask_set(e, (int)m, machines[m].live && admits(classes[(size_t)e], machines[m]));
    }
    int class_of(i64 cpu, i64 ram, std::vector<Selector> sels = {}) {
        const std::string canon = canonical(sels);
        const auto key = std::make_pair(std::make_pair(cpu, ram), canon);
        auto it = class_by_shape.find(key);
        if (it != class_by_shape.end()) return it->second;
        const int e = (int)classes.size();
        Class c;
        c.cpu = cpu; c.ram = ram;
        c.selectors = std::move(sels);
        if (!c.selectors.empty()) any_selector = true;
        const bool fits = laid_out && e < E;
        if (fits) {  // an unused class row of the resident network
            c.pool = spare_class_pool(e);
            reprice = true;
        } else {
            need_layout = true;
        }
        classes.push_back(c);
        class_by_shape[key] = e;
        if (fits) mask_class(e);
        return e;
    }
    Pool spare_class_pool(int e) const {
        Pool p;
        p.first = row_off[(size_t)(v_ec + e)];
        p.room = (int)(row_off[(size_t)(v_ec + e + 1)] - p.first - (i64)N * K);
        for (int i = p.room - 1; i >= 0; --i) p.free_places.push_back(i);
        return p;
    }
    Pool spare_job_pool(int j) const {
        Pool p;
        p.first = row_off[(size_t)(v_un + j)];
        p.room = (int)(row_off[(size_t)(v_un + j + 1)] - p.first - 1);
        for (int i = p.room - 1; i >= 0; --i) p.free_places.push_back(i);
        return p;
    }
    int job_of(const std::string &uuid) {
        auto it = job_by_uuid.find(uuid);
        if (it != job_by_uuid.end()) return it->second;
        int j;
        if (!free_jobs.empty()) { j = free_jobs.back(); free_jobs.pop_back(); }
        else { j = (int)jobs.size(); jobs.emplace_back(); }
        Job &jb = jobs[(size_t)j];
        jb.uuid = uuid;
        jb.used = true;
        if (laid_out && j < J) jb.pool = spare_job_pool(j);
        else { jb.pool = Pool(); need_layout = true; }
        job_by_uuid[uuid] = j;
        return j;
    }
    void release_job_if_empty(int j) {
        Job &jb = jobs[(size_t)j];
        if (jb.known > 0) return;
        job_by_uuid.erase(jb.uuid);
        jb.uuid.clear();
        jb.used = false;
        free_jobs.push_back(j);
    }

    // ---------------------------------------------------------------- layout
    // Lay the whole network out again from the entity tables, with fresh head-room.
    int layout() {
        const double t0 = now_ms();
        K = cfg.k_slots;
        i64 live_tasks = 0;
        for (auto &kv : tasks) if (kv.second.state == kRunnable && !kv.second.pinned) ++live_tasks;
        auto grown = [](i64 have, i64 floor_) { return (int)std::max<i64>(floor_, have + have / 4 + 1); };
        N = grown((i64)machines.size(), std::max(cfg.max_machines, 1));
        E = std::max<int>(cfg.max_classes, (int)((classes.size() + 63) / 64 * 64));
        J = grown((i64)jobs.size(), std::max(cfg.max_jobs, 1));
        P = grown(live_tasks, std::max(cfg.max_tasks, 1));
        lay.n_machines = N; lay.n_classes = E; lay.n_jobs = J; lay.n_tasks = P; lay.k_slots = K;
        v_ec = 1 + N; v_un = 1 + N + E; v_t = 1 + N + E + J;
        const i64 nn = (i64)1 + N + E + J + P;
        if (nn > INT32_MAX) return fail(PSN_EOVERFLOW, "too many vertices");
        n = (int32_t)nn;
        // members per class / job among runnable tasks
        std::vector<i64> cnt_e((size_t)E, 0), cnt_j((size_t)J, 0);
        for (auto &kv : tasks)
            if (kv.second.state == kRunnable && !kv.second.pinned) { ++cnt_e[(size_t)kv.second.cls]; ++cnt_j[(size_t)kv.second.job]; }
        const i64 avg_job = jobs.empty() ? 16 : std::max<i64>(16, live_tasks / (i64)std::max<size_t>(1, job_by_uuid.size()));
        row_off.assign((size_t)n + 1, 0);
        row_off[1] = (i64)N + J;
        for (int m = 0; m < N; ++m) row_off[(size_t)(1 + m + 1)] = row_off[(size_t)(1 + m)] + (i64)E * K + 1;
        room_e.assign((size_t)E, 0); room_j.assign((size_t)J, 0);
        for (int e = 0; e < E; ++e) {
            room_e[(size_t)e] = (int)(cnt_e[(size_t)e] + std::max<i64>(64, cnt_e[(size_t)e] / 4));
            row_off[(size_t)(v_ec + e + 1)] = row_off[(size_t)(v_ec + e)] + room_e[(size_t)e] + (i64)N * K;
        }
        for (int j = 0; j < J; ++j) {
            room_j[(size_t)j] = (int)(cnt_j[(size_t)j] ? cnt_j[(size_t)j] + std::max<i64>(16, cnt_j[(size_t)j] / 4) : avg_job + avg_job / 4);
            row_off[(size_t)(v_un + j + 1)] = row_off[(size_t)(v_un + j)] + room_j[(size_t)j] + 1;
        }
        for (int t = 0; t < P; ++t) row_off[(size_t)(v_t + t + 1)] = row_off[(size_t)(v_t + t)] + 2;
        m2 = row_off[(size_t)n];
        if (m2 >= (i64)0x7FFFFFF0) return fail(PSN_EOVERFLOW, "too many arcs");
        supply.assign((size_t)n, 0);
        ec_arc_base.assign((size_t)E, 0);
        resident = false;  // the device copy is built from this state at the next round
        dirty_arcs.clear(); cap_updates.clear(); dirty_supply.clear(); new_sources.clear();
        laid_out = true;
        attached = 0;
        for (int e = 0; e < E; ++e) {
            ec_arc_base[(size_t)e] = row_off[(size_t)(v_ec + e)] + room_e[(size_t)e];
            if ((size_t)e < classes.size()) classes[(size_t)e].pool = spare_class_pool(e);
        }
        for (int j = 0; j < J; ++j)
            if ((size_t)j < jobs.size()) jobs[(size_t)j].pool = spare_job_pool(j);
        slot_uid.assign((size_t)P, 0);
        slot_used.assign((size_t)P, 0);
        slot_task.assign((size_t)P, nullptr);
        free_slots.clear();
        for (int t = P - 1; t >= 0; --t) free_slots.push_back(t);
        // runnable tasks in uid order, so that a layout is a function of the state alone
        std::vector<uint64_t> uids;
        uids.reserve(tasks.size());
        for (auto &kv : tasks) { kv.second.slot = -1; if (kv.second.state == kRunnable && !kv.second.pinned) uids.push_back(kv.first); }
        std::sort(uids.begin(), uids.end());
        for (uint64_t u : uids)
            if (!attach(tasks[u])) return fail(PSN_EINVAL, "internal: fresh layout without room");
        allow.assign((size_t)E * mask_words(), 0u);
        for (size_t e = 0; e < classes.size(); ++e)
            for (size_t m = 0; m < machines.size(); ++m)
                if (machines[m].live && admits(classes[e], machines[m])) allow[e * mask_words() + (m >> 5)] |= 1u << (m & 31);
        for (size_t e = classes.size(); e < (size_t)E; ++e)  // unused class rows admit everything (they carry no tasks)
            for (int w = 0; w < mask_words(); ++w) allow[e * mask_words() + w] = ~0u;
        mask_dirty = any_selector;
        need_layout = false;
        reprice = false;
        ++st.rebuilds;
        st.build_ms = now_ms() - t0;
        return PSN_OK;
    }

    // The arc arrays of the network as psn_upload_csr would take them (slot arcs unpriced), from the
    // state: placeholders everywhere, then slot arcs, sink arcs and the arcs of the attached tasks.
    void materialize(std::vector<int32_t> &head, std::vector<int32_t> &cap, std::vector<int32_t> &cost, std::vector<int32_t> &rev) const {
        head.assign((size_t)m2, 0); cap.assign((size_t)m2, 0); cost.assign((size_t)m2, 0); rev.assign((size_t)m2, 0);
        for (int v = 0; v < n; ++v)
            for (i64 a = row_off[(size_t)v]; a < row_off[(size_t)v + 1]; ++a) { head[(size_t)a] = v; rev[(size_t)a] = (int32_t)a; }
        auto pair_up = [&](i64 a, i64 r, int to, int from, int cp, int cs) {
            head[(size_t)a] = to; cap[(size_t)a] = cp; cost[(size_t)a] = cs; rev[(size_t)a] = (int32_t)r;
            head[(size_t)r] = from; cap[(size_t)r] = 0; cost[(size_t)r] = -cs; rev[(size_t)r] = (int32_t)a;
        };
        for (int e = 0; e < E; ++e)
            for (i64 m = 0; m < N; ++m)
                for (i64 k = 0; k < K; ++k)
                    pair_up(ec_arc_base[(size_t)e] + m * K + k, row_off[(size_t)(1 + m)] + (i64)e * K + k, (int)(1 + m), v_ec + e, 0, 0);
        for (int m = 0; m < N; ++m) pair_up(machine_sink_arc(m), row_off[0] + m, 0, 1 + m, sink_capacity(m), 0);
        for (int j = 0; j < J; ++j)
            pair_up(row_off[(size_t)(v_un + j)] + room_j[(size_t)j], row_off[0] + N + j, 0, v_un + j,
                    (size_t)j < jobs.size() && jobs[(size_t)j].used ? jobs[(size_t)j].pool.count : 0, 0);
        for (int t = 0; t < P; ++t) {
            if (!slot_used[(size_t)t]) continue;
            const Task &tk = *slot_task[(size_t)t];
            const int vt = v_t + t;
            const i64 ta = row_off[(size_t)vt];
            pair_up(ta, tk.place_class, v_ec + tk.cls, vt, 1, 0);
            pair_up(ta + 1, tk.place_job, v_un + tk.job, vt, 1, cfg.unsched_cost);
        }
    }
    // Build the device copy from the state (no arc array crosses PCIe).
    int build_on_device() {
        std::vector<int32_t> msink((size_t)N), jsink((size_t)J), slot, cls, job, pe, pj;
        for (int m = 0; m < N; ++m) msink[(size_t)m] = sink_capacity(m);
        for (int j = 0; j < J; ++j) jsink[(size_t)j] = (size_t)j < jobs.size() && jobs[(size_t)j].used ? jobs[(size_t)j].pool.count : 0;
        slot.reserve((size_t)attached); cls.reserve((size_t)attached); job.reserve((size_t)attached); pe.reserve((size_t)attached); pj.reserve((size_t)attached);
        for (int t = 0; t < P; ++t) {
            if (!slot_used[(size_t)t]) continue;
            const Task &tk = *slot_task[(size_t)t];
            slot.push_back(t); cls.push_back(tk.cls); job.push_back(tk.job);
            pe.push_back((int32_t)(tk.place_class - row_off[(size_t)(v_ec + tk.cls)]));
            pj.push_back((int32_t)(tk.place_job - row_off[(size_t)(v_un + tk.job)]));
        }
        std::vector<i64> base((size_t)E);
        const int rc = psn_build_network_rooms_device(ctx, &lay, cfg.unsched_cost, room_e.data(), room_j.data(), msink.data(), jsink.data(),
                                                      (i64)slot.size(), slot.data(), cls.data(), job.data(), pe.data(), pj.data(), base.data());
        if (rc == PSN_OK && base != ec_arc_base) return PSN_EINVAL;
        return rc;
    }

    // External usage per machine.  A measured figure covers everything on the machine, the tasks the
    // service runs there included; their share is taken out again: the observed usage of a task where
    // TaskStats reported one, its request otherwise (requests stay what is reserved for a running task).
    void refresh_external() {
        ext_cpu.assign(machines.size(), 0); ext_mem.assign(machines.size(), 0);
        std::vector<i64> own_cpu, own_mem;
        any_measured = false;  // a measurement may have been replaced by a plain usage figure since
        for (const Machine &mc : machines) any_measured = any_measured || mc.meas_cpu >= 0 || mc.meas_mem >= 0;
        if (any_measured) {
            own_cpu.assign(machines.size(), 0); own_mem.assign(machines.size(), 0);
            for (auto &kv : tasks) {
                const Task &t = kv.second;
                if (!t.pinned || !on_its_machine(t)) continue;
                own_cpu[(size_t)t.machine] += t.obs_cpu >= 0 ? t.obs_cpu : t.cpu;
                own_mem[(size_t)t.machine] += t.obs_mem >= 0 ? t.obs_mem : t.ram;
            }
        }
        for (size_t m = 0; m < machines.size(); ++m) {
            const Machine &mc = machines[m];
            ext_cpu[m] = mc.meas_cpu >= 0 ? std::max<i64>(0, mc.meas_cpu - own_cpu[m]) : mc.used_cpu;
            ext_mem[m] = mc.meas_mem >= 0 ? std::max<i64>(0, mc.meas_mem - own_mem[m]) : mc.used_mem;
        }
    }
    void cost_inputs(std::vector<i64> &cc, std::vector<i64> &cm, std::vector<i64> &uc, std::vector<i64> &um, std::vector<i64> &rc,
                     std::vector<i64> &rm) {
        refresh_external();
        cc.assign((size_t)N, kDeadQuantity); cm.assign((size_t)N, kDeadQuantity);
        uc.assign((size_t)N, kDeadQuantity); um.assign((size_t)N, kDeadQuantity);
        for (size_t m = 0; m < machines.size() && m < (size_t)N; ++m) {
            const Machine &mc = machines[m];
            if (!mc.live) continue;
            cc[m] = mc.cap_cpu; cm[m] = mc.cap_mem;
            // usage = what the statistics report for things the service does not manage + the requests
            // of the tasks it has running there
            uc[m] = std::min(ext_cpu[m] + mc.committed_cpu, mc.cap_cpu); um[m] = std::min(ext_mem[m] + mc.committed_mem, mc.cap_mem);
        }
        rc.assign((size_t)E, 1); rm.assign((size_t)E, 1);
        for (size_t e = 0; e < classes.size() && e < (size_t)E; ++e) { rc[e] = classes[e].cpu; rm[e] = classes[e].ram; }
    }
};

namespace {

int add_runnable(psn_svc *s, Task &t) {
    ++s->runnable;
    if (s->laid_out && !s->need_layout) {
        if (!s->attach(t)) s->need_layout = true;
    }
    return PSN_OK;
}
void drop_runnable(psn_svc *s, Task &t) {
    --s->runnable;
    if (t.pinned) s->unpin(t);
    else if (s->laid_out && t.slot >= 0) s->detach(t);
    t.slot = -1;
}
bool bad_uuid(const char *u) { return !u || !*u || std::strlen(u) > 36; }

}  // namespace

extern "C" {

int psn_svc_create(const psn_svc_config *cfg, psn_svc **out) {
    if (!out) return PSN_EINVAL;
    *out = nullptr;
    psn_svc *s = new (std::nothrow) psn_svc();
    if (!s) return PSN_ENOMEM;
    if (cfg) s->cfg = *cfg;
    if (s->cfg.max_classes <= 0) s->cfg.max_classes = PSN_NCLASSES;
    if (s->cfg.k_slots <= 0) s->cfg.k_slots = PSN_KSLOTS;
    if (s->cfg.slots_per_machine <= 0) s->cfg.slots_per_machine = PSN_SLOTS;
    if (s->cfg.unsched_cost <= 0) s->cfg.unsched_cost = PSN_UNSCHED_COST;
    if (s->cfg.device >= 0) {
        psn_config sc = s->cfg.solver;
        sc.device = s->cfg.device;
        const int rc = psn_create(&sc, &s->ctx);
        if (rc != PSN_OK) { delete s; return rc; }
    }
    *out = s;
    return PSN_OK;
}

void psn_svc_destroy(psn_svc *s) {
    if (!s) return;
    if (s->ctx) psn_destroy(s->ctx);
    delete s;
}

const char *psn_svc_last_error(const psn_svc *s) { return s ? s->err.c_str() : "null service"; }

static bool read_selectors(const psn_selector *sels, int32_t n, std::vector<Selector> &out) {
    if (n < 0 || (n > 0 && !sels)) return false;
    for (int32_t i = 0; i < n; ++i) {
        if (sels[i].type < 0 || sels[i].type > 3 || !sels[i].key || sels[i].n_values < 0 || (sels[i].n_values > 0 && !sels[i].values)) return false;
        Selector x;
        x.type = sels[i].type;
        x.key = sels[i].key;
        for (int32_t v = 0; v < sels[i].n_values; ++v) {
            if (!sels[i].values[v]) return false;
            x.values.emplace_back(sels[i].values[v]);
        }
        out.push_back(std::move(x));
    }
    return true;
}

int psn_svc_task_submitted(psn_svc *s, uint64_t uid, const char *job_uuid, int32_t state, int64_t cpu, int64_t ram) {
    return psn_svc_task_submitted_sel(s, uid, job_uuid, state, cpu, ram, nullptr, 0);
}

int psn_svc_task_submitted_sel(psn_svc *s, uint64_t uid, const char *job_uuid, int32_t state, int64_t cpu, int64_t ram,
                               const psn_selector *sels, int32_t n_sels) {
    if (!s || bad_uuid(job_uuid) || uid == 0 || cpu < 0 || ram < 0) return s ? s->fail(PSN_EINVAL, "bad task description") : PSN_EINVAL;
    std::vector<Selector> selectors;
    if (!read_selectors(sels, n_sels, selectors)) return s->fail(PSN_EINVAL, "bad label selector");
    if (s->tasks.count(uid)) return PSN_TASK_ALREADY_SUBMITTED;
    if (state != PSN_TASK_STATE_CREATED) return PSN_TASK_STATE_NOT_CREATED;
    Task t;
    t.uid = uid;
    t.cpu = std::max<i64>(cpu, 1);  // a zero request still takes a slot: price it as the smallest request
    t.ram = std::max<i64>(ram, 1);
    t.cls = s->class_of(t.cpu, t.ram, std::move(selectors));
    t.job = s->job_of(job_uuid);
    Task &ref = s->tasks[uid] = t;
    ++s->jobs[(size_t)t.job].known;
    add_runnable(s, ref);
    return PSN_TASK_SUBMITTED_OK;
}

int psn_svc_task_updated(psn_svc *s, uint64_t uid, const char *job_uuid, int64_t cpu, int64_t ram) {
    if (!s || bad_uuid(job_uuid) || cpu < 0 || ram < 0) return s ? s->fail(PSN_EINVAL, "bad task description") : PSN_EINVAL;
    auto it = s->tasks.find(uid);
    if (it == s->tasks.end()) return PSN_TASK_NOT_FOUND;
    Task &t = it->second;
    auto jt = s->job_by_uuid.find(job_uuid);
    if (jt == s->job_by_uuid.end() || jt->second != t.job) return PSN_TASK_JOB_NOT_FOUND;
    cpu = std::max<i64>(cpu, 1);
    ram = std::max<i64>(ram, 1);
    if (cpu == t.cpu && ram == t.ram) return PSN_TASK_UPDATED_OK;
    if (t.pinned) {  // running: only what it holds on its machine changes
        if (s->on_its_machine(t)) s->account(t.machine, cpu - t.cpu, ram - t.ram, 0);
        t.cpu = cpu; t.ram = ram;
        t.cls = s->class_of(cpu, ram, s->classes[(size_t)t.cls].selectors);
        return PSN_TASK_UPDATED_OK;
    }
    const bool in_network = t.state == kRunnable;
    if (in_network) drop_runnable(s, t);
    t.cpu = cpu; t.ram = ram;
    t.cls = s->class_of(cpu, ram, s->classes[(size_t)t.cls].selectors);  // the selectors stay
    if (in_network) add_runnable(s, t);
    return PSN_TASK_UPDATED_OK;
}

static int task_leaves(psn_svc *s, uint64_t uid, int new_state, int ok_code, bool forget) {
    if (!s) return PSN_EINVAL;
    auto it = s->tasks.find(uid);
    if (it == s->tasks.end()) return PSN_TASK_NOT_FOUND;
    Task &t = it->second;
    if (t.job < 0 || (size_t)t.job >= s->jobs.size() || !s->jobs[(size_t)t.job].used) return PSN_TASK_JOB_NOT_FOUND;
    if (t.state == kRunnable) drop_runnable(s, t);
    t.state = new_state;
    t.machine = -1;
    if (forget) {
        const int j = t.job;
        s->tasks.erase(it);
        --s->jobs[(size_t)j].known;
        s->release_job_if_empty(j);
    }
    return ok_code;
}
int psn_svc_task_completed(psn_svc *s, uint64_t uid) { return task_leaves(s, uid, kCompleted, PSN_TASK_COMPLETED_OK, false); }
int psn_svc_task_failed(psn_svc *s, uint64_t uid) { return task_leaves(s, uid, kFailed, PSN_TASK_FAILED_OK, false); }
int psn_svc_task_removed(psn_svc *s, uint64_t uid) { return task_leaves(s, uid, kFailed, PSN_TASK_REMOVED_OK, true); }
int psn_svc_add_task_stats(psn_svc *s, uint64_t uid) {
    if (!s) return PSN_EINVAL;
    return s->tasks.count(uid) ? PSN_TASK_UPDATED_OK : PSN_TASK_NOT_FOUND;
}

int psn_svc_node_added(psn_svc *s, const char *uuid, int64_t cpu, int64_t ram) {
    if (!s || bad_uuid(uuid) || cpu <= 0 || ram <= 0 || cpu > (i64)4e15 || ram > (i64)4e15)
        return s ? s->fail(PSN_EINVAL, "bad machine description") : PSN_EINVAL;
    if (s->machine_by_uuid.count(uuid)) return PSN_NODE_ALREADY_EXISTS;
    int m;
    if (!s->free_machines.empty()) { m = s->free_machines.back(); s->free_machines.pop_back(); }
    else { m = (int)s->machines.size(); s->machines.emplace_back(); }
    Machine &mc = s->machines[(size_t)m];
    mc.uuid = uuid; mc.cap_cpu = cpu; mc.cap_mem = ram; mc.used_cpu = mc.used_mem = 0; mc.live = true; ++mc.gen;
    mc.committed_cpu = mc.committed_mem = 0; mc.running = 0;
    mc.meas_cpu = mc.meas_mem = -1;
    s->machine_by_uuid[uuid] = m;
    mc.labels.clear();
    if (s->laid_out && m < s->N) { s->set_cap(s->machine_sink_arc(m), s->sink_capacity(m)); s->reprice = true; s->mask_machine(m); }
    else s->need_layout = true;
    return PSN_NODE_ADDED_OK;
}

int psn_svc_node_updated(psn_svc *s, const char *uuid, int64_t cpu, int64_t ram) {
    if (!s || bad_uuid(uuid) || cpu <= 0 || ram <= 0 || cpu > (i64)4e15 || ram > (i64)4e15)
        return s ? s->fail(PSN_EINVAL, "bad machine description") : PSN_EINVAL;
    auto it = s->machine_by_uuid.find(uuid);
    if (it == s->machine_by_uuid.end()) return PSN_NODE_NOT_FOUND;
    Machine &mc = s->machines[(size_t)it->second];
    if (mc.cap_cpu != cpu || mc.cap_mem != ram) { mc.cap_cpu = cpu; mc.cap_mem = ram; s->reprice = true; }
    return PSN_NODE_UPDATED_OK;
}

int psn_svc_add_node_stats(psn_svc *s, const char *uuid, int64_t used_cpu, int64_t used_ram) {
    if (!s || bad_uuid(uuid) || used_cpu < 0 || used_ram < 0) return s ? s->fail(PSN_EINVAL, "bad machine statistics") : PSN_EINVAL;
    auto it = s->machine_by_uuid.find(uuid);
    if (it == s->machine_by_uuid.end()) return PSN_NODE_NOT_FOUND;
    Machine &mc = s->machines[(size_t)it->second];
    if (mc.used_cpu != used_cpu || mc.used_mem != used_ram || mc.meas_cpu >= 0 || mc.meas_mem >= 0) {
        mc.used_cpu = used_cpu; mc.used_mem = used_ram; mc.meas_cpu = mc.meas_mem = -1; s->reprice = true;
    }
    return PSN_NODE_UPDATED_OK;
}

// A utilisation figure of ResourceStats as a quantity: the message describes it as a fraction of the
// capacity (/root/reference/pkg/firmament/resource_stats.proto:36-39,56-59), the reference's own
// vectors carry absolute quantities (/root/reference/pkg/stats/stats_test.go:100-127: utilisation
// 2000.0 of capacity 10000).  psn_svc_config::stats_utilization says which reading applies; the
// default (0) takes values up to 1 as fractions and larger ones as quantities, which misreads an
// absolute reading of at most one millicore or one Kb (a machine that is idle to the last unit is
// priced as full): a caller that knows its feed sets 1 (always a fraction) or 2 (always a quantity).
static i64 stats_quantity(double utilization, i64 capacity, int mode) {
    if (!(utilization > 0)) return 0;
    const bool fraction = mode == 1 || (mode != 2 && utilization <= 1.0);
    const double q = fraction ? utilization * (double)capacity : utilization;
    const double capped = std::min(q, (double)capacity);
    return (i64)(capped + 0.5);
}

int psn_svc_add_node_stats_raw(psn_svc *s, const char *uuid, int32_t n_cpus, const int64_t *cpu_capacity, const double *cpu_utilization,
                               int64_t mem_capacity, double mem_utilization) {
    if (!s || bad_uuid(uuid) || n_cpus < 0 || (n_cpus > 0 && (!cpu_capacity || !cpu_utilization)) || mem_capacity < 0)
        return s ? s->fail(PSN_EINVAL, "bad machine statistics") : PSN_EINVAL;
    auto it = s->machine_by_uuid.find(uuid);
    if (it == s->machine_by_uuid.end()) return PSN_NODE_NOT_FOUND;
    Machine &mc = s->machines[(size_t)it->second];
    i64 cpu = 0;
    for (int32_t i = 0; i < n_cpus; ++i) {  // one entry per CPU, or one for the whole node as the reference sends it (stats.go:60-66)
        if (cpu_capacity[i] < 0) return s->fail(PSN_EINVAL, "bad machine statistics");
        cpu += stats_quantity(cpu_utilization[i], cpu_capacity[i], s->cfg.stats_utilization);
    }
    const i64 mem = stats_quantity(mem_utilization, mem_capacity, s->cfg.stats_utilization);
    if (mc.meas_cpu != cpu || mc.meas_mem != mem) { mc.meas_cpu = cpu; mc.meas_mem = mem; s->any_measured = true; s->reprice = true; }
    return PSN_NODE_UPDATED_OK;
}

int psn_svc_add_task_stats_raw(psn_svc *s, uint64_t uid, int64_t cpu_usage, int64_t mem_usage) {
    if (!s || cpu_usage < 0 || mem_usage < 0) return s ? s->fail(PSN_EINVAL, "bad task statistics") : PSN_EINVAL;
    auto it = s->tasks.find(uid);
    if (it == s->tasks.end()) return PSN_TASK_NOT_FOUND;
    Task &t = it->second;
    if (t.obs_cpu != cpu_usage || t.obs_mem != mem_usage) {
        t.obs_cpu = cpu_usage; t.obs_mem = mem_usage;
        if (t.pinned && s->any_measured) s->reprice = true;  // its share of a measured machine changed
    }
    return PSN_TASK_UPDATED_OK;
}

static int node_leaves(psn_svc *s, const char *uuid, int ok_code) {
    if (!s || bad_uuid(uuid)) return s ? s->fail(PSN_EINVAL, "bad resource uid") : PSN_EINVAL;
    auto it = s->machine_by_uuid.find(uuid);
    if (it == s->machine_by_uuid.end()) return PSN_NODE_NOT_FOUND;
    const int m = it->second;
    Machine &mc = s->machines[(size_t)m];
    mc.live = false;
    mc.uuid.clear();
    ++mc.gen;  // placements on it no longer count as "previous machine"
    if (mc.running > 0) s->orphans = true;  // the tasks that ran here are pending again
    mc.committed_cpu = mc.committed_mem = 0; mc.running = 0;
    s->machine_by_uuid.erase(it);
    s->free_machines.push_back(m);
    mc.labels.clear();
    if (s->laid_out && m < s->N) { s->set_cap(s->machine_sink_arc(m), 0); s->reprice = true; s->mask_machine(m); }
    return ok_code;
}
int psn_svc_node_failed(psn_svc *s, const char *uuid) { return node_leaves(s, uuid, PSN_NODE_FAILED_OK); }
int psn_svc_node_removed(psn_svc *s, const char *uuid) { return node_leaves(s, uuid, PSN_NODE_REMOVED_OK); }

int psn_svc_node_labels(psn_svc *s, const char *uuid, const char *const *keys, const char *const *values, int32_t n_labels) {
    if (!s || bad_uuid(uuid) || n_labels < 0 || (n_labels > 0 && (!keys || !values))) return s ? s->fail(PSN_EINVAL, "bad labels") : PSN_EINVAL;
    auto it = s->machine_by_uuid.find(uuid);
    if (it == s->machine_by_uuid.end()) return PSN_NODE_NOT_FOUND;
    Machine &mc = s->machines[(size_t)it->second];
    std::map<std::string, std::string> labels;
    for (int32_t i = 0; i < n_labels; ++i) {
        if (!keys[i] || !values[i]) return s->fail(PSN_EINVAL, "bad labels");
        labels[keys[i]] = values[i];
    }
    if (labels != mc.labels) {
        mc.labels.swap(labels);
        if (s->laid_out && it->second < s->N) s->mask_machine(it->second);
    }
    return PSN_NODE_UPDATED_OK;
}

int psn_svc_export_mask(psn_svc *s, uint32_t *mask) {
    if (!s || !mask) return PSN_EINVAL;
    s->readmit_orphans();
    if (!s->laid_out || s->need_layout) { const int rc = s->layout(); if (rc != PSN_OK) return rc; }
    std::copy(s->allow.begin(), s->allow.end(), mask);
    return PSN_OK;
}

int psn_svc_check(const psn_svc *s) {
    if (!s) return PSN_SERVING_UNKNOWN;
    return s->ctx ? PSN_SERVING : PSN_NOT_SERVING;
}

int psn_svc_network_size(psn_svc *s, psn_layout *lay, int32_t *n, int64_t *m2) {
    if (!s) return PSN_EINVAL;
    s->readmit_orphans();
    if (!s->laid_out || s->need_layout) { const int rc = s->layout(); if (rc != PSN_OK) return rc; }
    if (lay) *lay = s->lay;
    if (n) *n = s->n;
    if (m2) *m2 = s->m2;
    return PSN_OK;
}

int psn_svc_export_network(psn_svc *s, int64_t *row_off, int32_t *head, int32_t *cap, int32_t *cost, int32_t *rev, int64_t *supply,
                           int64_t *ec_arc_base, int64_t *cap_cpu, int64_t *cap_mem, int64_t *used_cpu, int64_t *used_mem,
                           int64_t *req_cpu, int64_t *req_mem) {
    if (!s) return PSN_EINVAL;
    s->readmit_orphans();
    if (!s->laid_out || s->need_layout) { const int rc = s->layout(); if (rc != PSN_OK) return rc; }
    auto put64 = [](int64_t *dst, const std::vector<i64> &v) { if (dst) std::copy(v.begin(), v.end(), dst); };
    auto put32 = [](int32_t *dst, const std::vector<int32_t> &v) { if (dst) std::copy(v.begin(), v.end(), dst); };
    put64(row_off, s->row_off);
    if (head || cap || cost || rev) {
        std::vector<int32_t> h, c, k, r;
        s->materialize(h, c, k, r);
        put32(head, h); put32(cap, c); put32(cost, k); put32(rev, r);
    }
    put64(supply, s->supply); put64(ec_arc_base, s->ec_arc_base);
    std::vector<i64> cc, cm, uc, um, rc, rm;
    s->cost_inputs(cc, cm, uc, um, rc, rm);
    put64(cap_cpu, cc); put64(cap_mem, cm); put64(used_cpu, uc); put64(used_mem, um); put64(req_cpu, rc); put64(req_mem, rm);
    return PSN_OK;
}

int64_t psn_svc_schedule(psn_svc *s) {
    if (!s) return PSN_EINVAL;
    if (!s->ctx) return s->fail(PSN_ESTATE, "no device: the service was created with device = -1");
    const double t_round = now_ms();
    struct RoundClock { psn_svc *s; double t0; ~RoundClock() { s->st.total_ms = now_ms() - t0; } } round_clock{s, t_round};
    ++s->st.rounds;
    s->st.build_ms = s->st.upload_ms = s->st.update_ms = s->st.solve_ms = s->st.extract_ms = 0;
    s->st.patched_arcs = s->st.cap_updates = s->st.supply_updates = 0;
    s->deltas.clear();
    s->st.tasks_deferred = 0;
    int rc;
    s->readmit_orphans();
    // Keep-running mode with nothing pending: there is nothing to decide, whatever is queued (and a
    // layout that is due) waits for the round that has something to place.
    if (s->keep_running() && s->runnable == s->pinned_total && !s->cfg.emit_noop) {
        s->st.last_round_warm = 1;
        ++s->st.warm_rounds;
        s->st.tasks_placed = 0;
        return 0;
    }
    if (!s->laid_out || s->need_layout) { if ((rc = s->layout()) != PSN_OK) return rc; }
    // Nothing was submitted, removed, re-priced or re-labelled since a round that ran to its end:
    // the resident optimum still stands and there is nothing to report (Poseidon calls Schedule
    // on a timer, /root/reference/cmd/poseidon/poseidon.go:32-72; most rounds of a quiet cluster are such).
    if (s->settled && s->resident && !s->cfg.emit_noop && s->dirty_arcs.empty() && s->cap_updates.empty() && s->dirty_supply.empty() &&
        s->new_sources.empty() && !s->mask_dirty && !s->reprice) {
        s->st.last_round_warm = 1;
        ++s->st.warm_rounds;
        return 0;
    }
    s->settled = false;
    std::vector<i64> cc, cm, uc, um, rq, rm;
    psn_result res{};
    bool warm = false;
    if (!s->resident) {
        double t0 = now_ms();
        rc = s->build_on_device();
        if (rc != PSN_OK) return s->fail(rc, std::string("build on device: ") + psn_last_error(s->ctx));
        if (s->any_selector) {
            rc = psn_set_admission_mask(s->ctx, &s->lay, s->allow.data());
            if (rc != PSN_OK) return s->fail(rc, std::string("admission mask: ") + psn_last_error(s->ctx));
        }
        s->mask_dirty = false;
        s->cost_inputs(cc, cm, uc, um, rq, rm);
        rc = psn_eval_costs(s->ctx, &s->lay, cc.data(), cm.data(), uc.data(), um.data(), rq.data(), rm.data(), s->ec_arc_base.data());
        if (rc != PSN_OK) return s->fail(rc, std::string("eval_costs: ") + psn_last_error(s->ctx));
        s->st.upload_ms = now_ms() - t0;
        s->resident = true;
        s->reprice = false;
    } else {
        double t0 = now_ms();
        // PSN_DEBUG_SVC: the parts of an in-place update on stderr
        const bool dbg_svc = std::getenv("PSN_DEBUG_SVC") != nullptr;
        double t_lap = t0;
        auto lap = [&](const char *what) { if (dbg_svc) { const double t = now_ms(); fprintf(stderr, "  update: %-10s %.3f ms\n", what, t - t_lap); t_lap = t; } };
        if (!s->dirty_arcs.empty()) {
            // in the table's order: every entry patches words of its own, so the order is free (sorting 23 000
            // entries of a 100 000-pod round cost more than the patch kernel saves by it)
            const size_t nd = s->dirty_arcs.size();
            std::vector<i64> idx(nd);
            std::vector<int32_t> h(nd), c(nd), k(nd), r(nd);
            size_t i = 0;
            for (const auto &kv : s->dirty_arcs) {
                const psn_svc::ArcValue &av = kv.second;
                idx[i] = kv.first; h[i] = av.head; c[i] = av.cap; k[i] = av.cost; r[i] = (int32_t)av.rev;
                ++i;
            }
            lap("patch prep");
            rc = psn_patch_arcs(s->ctx, (i64)idx.size(), idx.data(), h.data(), c.data(), k.data(), r.data());
            if (rc != PSN_OK) { s->resident = false; return s->fail(rc, std::string("patch: ") + psn_last_error(s->ctx)); }
            s->st.patched_arcs = (i64)idx.size();
        }
        lap("patch");
        if (!s->cap_updates.empty()) {
            std::vector<i64> idx;
            std::vector<int32_t> cp;
            for (auto &kv : s->cap_updates)
                if (!s->dirty_arcs.count(kv.first)) { idx.push_back(kv.first); cp.push_back(kv.second); }
            rc = psn_update_caps(s->ctx, (i64)idx.size(), idx.data(), cp.data());
            if (rc != PSN_OK) { s->resident = false; return s->fail(rc, std::string("update_caps: ") + psn_last_error(s->ctx)); }
            s->st.cap_updates = (i64)idx.size();
        }
        if (!s->dirty_supply.empty()) {
            std::vector<int32_t> v(s->dirty_supply.begin(), s->dirty_supply.end());
            std::vector<i64> sv(v.size());
            for (size_t i = 0; i < v.size(); ++i) sv[i] = s->supply[(size_t)v[i]];
            rc = psn_update_supplies(s->ctx, (i64)v.size(), v.data(), sv.data());
            if (rc != PSN_OK) { s->resident = false; return s->fail(rc, std::string("update_supplies: ") + psn_last_error(s->ctx)); }
            s->st.supply_updates = (i64)v.size();
        }
        lap("caps+supply");
        if (s->mask_dirty && s->any_selector) {
            rc = psn_set_admission_mask(s->ctx, &s->lay, s->allow.data());
            if (rc != PSN_OK) { s->resident = false; return s->fail(rc, std::string("admission mask: ") + psn_last_error(s->ctx)); }
            s->reprice = true;
        }
        s->mask_dirty = false;
        if (s->reprice) {
            s->cost_inputs(cc, cm, uc, um, rq, rm);
            rc = psn_reprice_costs(s->ctx, &s->lay, cc.data(), cm.data(), uc.data(), um.data(), rq.data(), rm.data(),
                                   s->ec_arc_base.data());
            if (rc != PSN_OK) { s->resident = false; return s->fail(rc, std::string("reprice: ") + psn_last_error(s->ctx)); }
            s->reprice = false;
        }
        lap("reprice");
        if (!s->new_sources.empty()) {
            // a slot may have been taken and given back since: settle only vertices that are sources now
            std::vector<int32_t> v;
            for (int32_t x : s->new_sources) if (s->supply[(size_t)x] > 0) v.push_back(x);
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            int64_t skipped = 0;
            rc = psn_settle_sources(s->ctx, (i64)v.size(), v.data(), &skipped);
            if (rc != PSN_OK) { s->resident = false; return s->fail(rc, std::string("settle: ") + psn_last_error(s->ctx)); }
        }
        lap("settle");
        s->dirty_arcs.clear(); s->cap_updates.clear(); s->dirty_supply.clear(); s->new_sources.clear();
        s->st.update_ms = now_ms() - t0;
        warm = true;
    }
    const bool in_place = warm;  // the network stayed resident: tasks have machines from the round before
    {
        const double t0 = now_ms();
        rc = psn_solve(s->ctx, warm ? PSN_WARM_START : 0u, &res);
        s->st.solve_ms = now_ms() - t0;
        if (rc != PSN_OK) { s->resident = false; return s->fail(rc, std::string("solve: ") + psn_last_error(s->ctx)); }
    }
    // what the solver did, not what was asked for: a patch that outgrew the cost scale's head-room makes
    // the solve after it a cold one although the network stayed resident, and so does a kept state that is
    // no head start (solve_warm_declined)
    warm = res.warm_started != 0;
    s->st.last_round_warm = warm ? 1 : 0;
    if (warm) ++s->st.warm_rounds;
    s->st.total_cost = res.total_cost;
    s->st.solve_replays = (i64)res.replays; s->st.solve_last_phase_replays = res.last_phase_replays; s->st.solve_refines = res.refines;
    s->st.solve_rounds = (i64)res.rounds; s->st.solve_global_updates = (i64)res.global_updates; s->st.solve_pushes = (i64)res.pushes;
    s->st.solve_first_units = (i64)res.first_units; s->st.solve_first_eps_log2 = res.first_eps_log2; s->st.solve_device_ms = res.solve_ms;
    s->st.solve_warm_declined = res.warm_declined;
    // ---- placements -> deltas
    const double t0 = now_ms();
    const bool dbg_x = std::getenv("PSN_DEBUG_SVC") != nullptr;
    double tx = t0;
    auto lapx = [&](const char *what) { if (dbg_x) { const double t = now_ms(); fprintf(stderr, "  extract: %-10s %.3f ms\n", what, t - tx); tx = t; } };
    std::vector<int32_t> place((size_t)s->P, -1);
    rc = psn_get_placements(s->ctx, &s->lay, place.data());
    if (rc != PSN_OK) return s->fail(rc, std::string("placements: ") + psn_last_error(s->ctx));
    lapx("placements");
    // Tasks of a class are interchangeable: hand each class's routed units first to the tasks
    // that already sit on the unit's machine, so that an unchanged optimum yields no deltas.
    std::vector<std::vector<int>> members((size_t)s->E);
    for (int t = 0; t < s->P; ++t)
        if (s->slot_used[(size_t)t]) members[(size_t)s->slot_task[(size_t)t]->cls].push_back(t);
    lapx("members");
    i64 placed = 0;
    std::vector<int32_t> final_place((size_t)s->P, -1);
    // Classes are independent of each other here (a task belongs to one class and final_place is
    // written per task), so a large cluster's classes are shared out over a few host threads.
    std::atomic<int> next_class{0};
    std::atomic<bool> bad_machine{false};
    auto match_classes = [&]() {
        std::vector<int> units(s->machines.size(), 0);  // units of the class on each machine; cleared through `touched`
        std::vector<int> touched, rest;
        std::vector<char> done;
        for (int e = next_class.fetch_add(1); e < s->E; e = next_class.fetch_add(1)) {
            const std::vector<int> &mem = members[(size_t)e];
            i64 total = 0;
            touched.clear();
            for (int t : mem) {
                const int m = place[(size_t)t];
                if (m < 0) continue;
                if ((size_t)m >= units.size()) { bad_machine = true; return; }
                if (units[(size_t)m]++ == 0) touched.push_back(m);
                ++total;
            }
            if (!total) continue;
            done.assign(mem.size(), 0);
            for (size_t i = 0; i < mem.size(); ++i) {  // keep what can be kept
                const Task &tk = *s->slot_task[(size_t)mem[i]];
                if (tk.machine < 0 || (size_t)tk.machine >= s->machines.size() || s->machines[(size_t)tk.machine].gen != tk.machine_gen) continue;
                if (units[(size_t)tk.machine] == 0) continue;
                --units[(size_t)tk.machine]; --total;
                final_place[(size_t)mem[i]] = tk.machine;
                done[i] = 1;
            }
            // the rest of the units, machine by machine, to the tasks the solver routed, then to any other
            std::sort(touched.begin(), touched.end());
            rest.clear();
            for (int m : touched) { for (int c = 0; c < units[(size_t)m]; ++c) rest.push_back(m); units[(size_t)m] = 0; }
            size_t ri = 0;
            for (int pass = 0; pass < 2 && ri < rest.size(); ++pass)
                for (size_t i = 0; i < mem.size() && ri < rest.size(); ++i) {
                    if (done[i]) continue;
                    if (pass == 0 && place[(size_t)mem[i]] < 0) continue;
                    final_place[(size_t)mem[i]] = rest[ri++];
                    done[i] = 1;
                }
        }
    };
    {
        // (a first round has nothing to keep: its matching is too short to be worth the threads' start)
        const int helpers = in_place && s->P >= 20000 && s->E >= 8 ? 3 : 0;
        std::vector<std::thread> pool;
        for (int i = 0; i < helpers; ++i) pool.emplace_back(match_classes);
        match_classes();
        for (std::thread &th : pool) th.join();
        if (bad_machine) return s->fail(PSN_EDEVICE, "placement names a machine that does not exist");
    }
    // Keep-running mode: a placement is admitted only if the task's request still fits next to what
    // the machine already holds, tasks admitted earlier in this round included.  One class never
    // over-commits a machine (the slot arcs that do not fit have no capacity), two classes placed on
    // one machine in the same round can; the task that does not fit stays pending and is priced
    // against the new usage next round.
    lapx("matching");
    const bool keep = s->keep_running();
    std::vector<std::pair<Task *, int>> to_pin;
    std::vector<i64> free_cpu, free_mem;
    std::vector<int> free_slot;
    if (keep) {
        s->refresh_external();
        free_cpu.resize(s->machines.size()); free_mem.resize(s->machines.size()); free_slot.resize(s->machines.size());
        for (size_t m = 0; m < s->machines.size(); ++m) {
            const Machine &mc = s->machines[m];
            free_cpu[m] = mc.cap_cpu - std::min(s->ext_cpu[m] + mc.committed_cpu, mc.cap_cpu);
            free_mem[m] = mc.cap_mem - std::min(s->ext_mem[m] + mc.committed_mem, mc.cap_mem);
            free_slot[m] = s->sink_capacity((int)m);
        }
    }
    for (int t = 0; t < s->P; ++t) {
        if (!s->slot_used[(size_t)t]) continue;
        Task &tk = *s->slot_task[(size_t)t];
        const bool had = s->on_its_machine(tk);
        int now = final_place[(size_t)t];
        if (keep && now >= 0) {
            const size_t m = (size_t)now;
            if (tk.cpu > free_cpu[m] || tk.ram > free_mem[m] || free_slot[m] <= 0) { now = -1; ++s->st.tasks_deferred; }
            else { free_cpu[m] -= tk.cpu; free_mem[m] -= tk.ram; --free_slot[m]; to_pin.emplace_back(&tk, now); }
        }
        int type = PSN_DELTA_NOOP;
        if (!had && now >= 0) type = PSN_DELTA_PLACE;
        else if (had && now < 0) type = PSN_DELTA_PREEMPT;
        else if (had && now >= 0 && now != tk.machine) type = PSN_DELTA_MIGRATE;
        if (type != PSN_DELTA_NOOP || s->cfg.emit_noop) s->deltas.push_back(Delta{tk.uid, type, now >= 0 ? now : (had ? tk.machine : -1)});
        tk.machine = now;
        tk.machine_gen = now >= 0 ? s->machines[(size_t)now].gen : 0;
        if (now >= 0) ++placed;
    }
    if (keep && !to_pin.empty()) {
        // most of the network leaving at once (a first round): lay the rest out afresh instead of parking arc by arc
        if ((i64)to_pin.size() * 4 > s->attached) s->need_layout = true;
        for (auto &pr : to_pin) s->pin(*pr.first, pr.second);
    }
    lapx("deltas");
    s->st.extract_ms = now_ms() - t0;
    s->st.tasks_placed = placed;
    s->st.tasks_running = s->pinned_total;
    s->st.tasks_runnable = s->runnable;
    s->st.machines_live = (i64)s->machine_by_uuid.size();
    s->st.n = s->n;
    s->st.m2 = s->m2;
    s->settled = true;
    return (int64_t)s->deltas.size();
}

int psn_svc_get_deltas(psn_svc *s, int64_t first, int64_t count, uint64_t *task_uid, int32_t *type, char *resource_uuid) {
    if (!s || first < 0 || count < 0 || first + count > (int64_t)s->deltas.size()) return s ? s->fail(PSN_EINVAL, "delta range") : PSN_EINVAL;
    for (int64_t i = 0; i < count; ++i) {
        const Delta &d = s->deltas[(size_t)(first + i)];
        if (task_uid) task_uid[i] = d.uid;
        if (type) type[i] = d.type;
        if (resource_uuid) {
            char *dst = resource_uuid + i * 37;
            std::memset(dst, 0, 37);
            if (d.type != PSN_DELTA_PREEMPT && d.machine >= 0 && (size_t)d.machine < s->machines.size())
                std::strncpy(dst, s->machines[(size_t)d.machine].uuid.c_str(), 36);
        }
    }
    return PSN_OK;
}

int psn_svc_get_stats(const psn_svc *s, psn_svc_stats *out) {
    if (!s || !out) return PSN_EINVAL;
    *out = s->st;
    out->tasks_runnable = s->runnable;
    out->tasks_running = s->pinned_total;
    out->machines_live = (i64)s->machine_by_uuid.size();
    return PSN_OK;
}

int psn_svc_task_placement(psn_svc *s, uint64_t uid, char *resource_uuid) {
    if (!s || !resource_uuid) return PSN_EINVAL;
    std::memset(resource_uuid, 0, 37);
    auto it = s->tasks.find(uid);
    if (it == s->tasks.end()) return PSN_TASK_NOT_FOUND;
    const Task &t = it->second;
    if (t.machine >= 0 && (size_t)t.machine < s->machines.size() && s->machines[(size_t)t.machine].gen == t.machine_gen)
        std::strncpy(resource_uuid, s->machines[(size_t)t.machine].uuid.c_str(), 36);
    return PSN_OK;
}

psn_ctx *psn_svc_solver(psn_svc *s) { return s ? s->ctx : nullptr; }

}  // extern "C"
