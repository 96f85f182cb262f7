// Shared host/device declarations of libpsn: device-resident graph view, control
// block and kernel prototypes.  Included by the kernels and by the C-ABI host code.
#ifndef PSN_DEVICE_H_
#define PSN_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psn {

typedef long long i64;

constexpr int kBlock = 256;      // threads per workgroup (4 waves)
constexpr int kGrid = 1024;      // workgroups per sweep launch: 16 waves per CU on 256 CUs
constexpr int kSweepWaves = kGrid * 4;  // per-wave slots (work counters, largest distance): waves of the largest sweep grid
constexpr unsigned kLowMax = 4;  // degree <= kLowMax: one lane per vertex
constexpr unsigned kStep = 576;  // arcs per wave step: 8 consecutive per lane + one strided
constexpr unsigned kArcPad = 8;   // spare entries after the arc arrays: a lane may load its 8 arcs past the end of a range
constexpr unsigned kWSeg = 2 * kStep;  // arcs per wave work item; larger vertices are split into several items
constexpr unsigned kMedMax = kWSeg;  // degree <= kMedMax: the whole vertex is one wave item
constexpr unsigned kQCap = 256;  // entries per per-wave LDS frontier staging queue
constexpr int kVptSmallN = 32768; // vertex sweeps: one vertex per thread up to this many vertices, two beyond
__host__ __device__ constexpr int vpt_of(int n) { return n <= kVptSmallN ? 1 : 2; }
constexpr int kMaxVertices = 1 << 28;  // lane items pack the degree above bit 28
constexpr int kClsLow = 0, kClsMed = 1, kClsHigh = 2;  // vertex classes by degree
constexpr int kClsMask = 3;      // the class in a cls[] byte
constexpr int kClsLean = 4;      // cls[] bit: every step of the row is made of aligned groups of 8 parallel arcs
// Counters that many workgroups add to within one launch sit 128 bytes apart: atomics of the whole GPU
// on one cache line are served one after the other (about 12 ns each, measured by
// tools/ubench/atomic_contention.hip), whichever words of the line they name.
// The two list sizes stay side by side: every sweep reads both when it starts, and a line more to wait for
// costs each launch more than their appends cost each other.
constexpr int kCntLine = 32;     // unsigned words per cache line
constexpr int kSlotLow = 0, kSlotSeg = 1;                                // list counters: lane items, wave items
constexpr int kSlotUnits = 1 * kCntLine, kSlotDeficits = 2 * kCntLine;  // work lists only: units of excess at the listed vertices (a vertex counts for at most 2^16 - 1); vertices that hold a deficit (hubs included)
constexpr int kCntWords = 3 * kCntLine;  // words per counter block
__host__ __device__ constexpr int cnt_slot(int j) { return j == 0 ? kSlotLow : (j == 1 ? kSlotSeg : (j == 2 ? kSlotUnits : kSlotDeficits)); }
constexpr int kCntSlots = 4;
constexpr int kHubShards = 16;   // copies of the per-hub accumulators the push sweep flushes into (workgroup b: copy b mod kHubShards)
constexpr int kPurseStride = 16; // 64-bit words between the purses of two hubs (one cache line each)
constexpr long long kHubShardsMinArcs = 2 << 20;  // residual arcs from which a network's hubs get purses and sharded accumulators
constexpr int kHubPull = -2;     // hub_of[] value of a pull vertex (see DevGraph::pull_list); -1: an ordinary vertex
constexpr int kMaxHubs = 128;    // highest-degree vertices whose atomics are pre-combined in LDS per workgroup
constexpr double kTailFactor = 0.1, kTailPower = 0.74;  // Ctrl::tail_units = kTailFactor * (total supply)^kTailPower
constexpr int kDefaultRounds = 14;  // push/relabel rounds per replay, i.e. between two global updates
constexpr int kDefaultAlphaLog2 = 4;  // epsilon / 16 per phase
constexpr int kWarmDeclineSteps = 1;  // a warm start whose first epsilon is within this many powers of two of a cold solve's starts cold
constexpr int kGuIters = 12;     // frontier expansions per replay (multiple of 3)
constexpr int PSN_EINFEASIBLE_DEV = -4;

// tally slots (per wave, summed at the end of a solve)
enum { kTallyScans = 0, kTallyPushes, kTallyRelabels, kTallyVisits, kTallyGuScans, kTallyGuRelax,
       kTallySat, kTallyRounds, kTallyGuIters, kTallyGuSkips, kTallySkips, kTallyN };

// Device-resident loop control.  k_begin (one thread) and the first thread of k_gu_apply are the
// only writers of the flags; sweep kernels read them and return at once when their
// stage is off, which is what lets a fixed hipGraph be replayed until `done`.
struct Ctrl {
    i64 eps;
    int eps_shift, alpha_log2;
    int tail_units;  // a phase that is not the last one ends when the active vertices hold no more than this many units of excess
    int need_sat, need_gu, gu_active;
    int sat_on, gu_on, gu_init_on, rounds_on;
    int done, error;
    unsigned epoch, gu_stamp_base;
    alignas(128) unsigned list_cnt[2][kCntWords];  // per list parity, at kSlotLow / kSlotSeg / kSlotUnits / kSlotDeficits
    alignas(128) unsigned fr_cnt[3][kCntLine];     // per frontier buffer, at kSlotLow / kSlotSeg
    alignas(128) int gu_phase[2];  // global update, read by expansion k at [k & 1], written at [(k + 1) & 1]: 0 expanding, 1 the pull sweep just ran, 2 converged
    unsigned long long tally_refines, tally_gus;
    // Orientation (see Orient in the kernels): mirror = 1 while the sweeps work on the reversed
    // network; k_begin decides per replay, k_flip (flip_on) negates prices and excesses in place.
    int mirror, flip_on;
    int mirror_max_shift;   // replays may run mirrored in phases with eps_shift <= this (-1: never)
    int mirror_ratio;       // ... changing sides when (vertices holding a deficit) * mirror_ratio / 4 <= work items of the current side (0: never)
    unsigned long long tally_mirror;  // replays run mirrored
    unsigned first_units;             // units of excess on the work list when the solve's first round started (~0: not yet)
    unsigned last_phase_replays;      // replays begun at epsilon = 1
};

// Plain-pointer view of one flow network in HBM (struct of arrays).
//   static:  row[n+1] (u32 offsets), head/cost/rev/tail/ocap[m2], cls[n], hub_of[n]
//   dynamic: cap[m2] residual capacity, rcap[m2] = cap[rev[a]] (kept in step by every
//            push so that the global update streams 4 B per arc instead of gathering),
//            price/excess/cand/dist[n]
struct DevGraph {
    int n;
    unsigned m2;
    i64 scale;  // S = 2 |C| + 1, C a vertex cover (at most n + 1); negative while the reversed network is being worked on
    const unsigned *row;
    const int *head, *cost, *rev, *tail, *ocap;
    const unsigned char *cls;  // class | kClsLean
    const unsigned *lean_stop;  // per vertex: first arc index past the leading aligned groups of 8 parallel arcs of its row
    const int *hub_of;      // hub slot of a vertex; kHubPull for a pull vertex; -1 for the rest
    // Pull vertices: degree <= kLowMax and no neighbour of degree <= kLowMax (the tasks of a scheduling
    // network).  The frontier expansions of a global update do not relax them; when the frontier
    // runs dry one sweep over this static list works their distances out from their neighbours'
    // (no atomics: each is written by its own lane) and relaxes their neighbours in turn.
    const int2 *pull_list;  // lane items
    const unsigned *pull_cnt;
    const int *hub_vertex;  // vertex of each hub slot
    const int *seg1_of;     // per vertex: 1 + index of its first segment's bound slots when it is a hub, else 0
    int nhubs;
    // Per hub segment (wave item of a hub row): upper bounds on the largest key
    // price[head] - cost * S over the segment's residual arcs, so that the push sweep can skip
    // segments that cannot hold an admissible arc.  seg_scan: value at the segment's last
    // complete scan; seg_add[p]: keys of arcs that regained capacity during rounds of parity p.
    const unsigned *hub_row0;  // first arc of each hub's row
    const int *hub_seg0;       // index of each hub's first segment
    i64 *seg_scan, *seg_add[2];
    int n_hub_segs;
    // Hubs are neighbours in the vertex numbering (the task classes), so their excess[] and cand[] words
    // share cache lines, and a line serves the atomics of the whole GPU one at a time.  During a push
    // round a hub's words are therefore not touched: the waves that push from its segments draw on
    // hub_purse[slot * kPurseStride] (the excess the hub held when the round's lists were built), and
    // everything that is sent to the hub or taken from it is summed per workgroup and flushed to one of
    // kHubShards copies [copy * kMaxHubs + slot] of a sum and a largest-key accumulator.  The relabel
    // sweep that follows folds the copies into excess[] and cand[] and empties them.
    // hub_shards = 0 (networks whose sweeps are a few hundred workgroups: the lines are not contended, and the
    // fold would cost the relabel sweep more than it saves): none of this, the hubs' own words are used.
    i64 *hub_purse, *hub_acc_add, *hub_acc_ext;
    int hub_shards;  // kHubShards or 0
    i64 *dmax_part;  // largest distance handed out, one slot per wave of the largest sweep grid
    int *cap, *rcap;
    i64 *price, *excess, *cand, *dist;
    int2 *list_low[2];  // active vertices of degree <= kLowMax (one lane each)
    int4 *list_seg[2];  // active wave items (vertex, arc range, flags)
    int2 *fr_low[3];    // global-update frontier, same split, triple buffered
    int4 *fr_seg[3];
    unsigned long long *tally;
    Ctrl *ctrl;
    const void *spare;  // 64 bytes of zeros: where a lane that needs no value points its load
    int4 *dbg;          // optional launch trace (kind, k, lane-list size, item-list size); null when off
    unsigned *dbg_cnt;
};
constexpr unsigned kDbgCap = 1u << 16;

// Per member of a lockstep group: what the group's set-up and read-out launches need besides the
// member's DevGraph (the scalar / counter / control words live in blocks of the group, so that
// one copy serves all members).
struct GroupAux {
    const i64 *supply;
    i64 *scalar;                    // kScalarSlots words of this member
    unsigned long long *tally_out;  // kTallyN words
    Ctrl *ctrl_out;
    int warm;
};

__global__ void k_push(DevGraph g, int k);
__global__ void k_relabel_compact(DevGraph g, int k, int do_relabel);
__global__ void k_saturate(DevGraph g);
__global__ void k_gu_init(DevGraph g);
__global__ void k_gu_expand(DevGraph g, int k);
__global__ void k_gu_apply(DevGraph g, int n_iters);
__global__ void k_begin(DevGraph g, int n_rounds);
__global__ void k_flip(DevGraph g, int unflip);
__global__ void k_push_group(const DevGraph *gs, int k);
__global__ void k_relabel_compact_group(const DevGraph *gs, int k, int do_relabel);
__global__ void k_saturate_group(const DevGraph *gs);
__global__ void k_gu_init_group(const DevGraph *gs);
__global__ void k_gu_expand_group(const DevGraph *gs, int k);
__global__ void k_gu_apply_group(const DevGraph *gs, int n_iters);
__global__ void k_begin_group(const DevGraph *gs, int n_rounds);
__global__ void k_flip_group(const DevGraph *gs, int unflip);
__global__ void k_group_status(const DevGraph *gs, int count, int n_rounds, int *status);
__global__ void k_reset(DevGraph g, const i64 *supply, int keep_flow);
__global__ void k_mirror(DevGraph g, const i64 *supply, int recompute_excess);
__global__ void k_excess_from_flow(DevGraph g);
__global__ void k_total_cost(DevGraph g, i64 *out);
__global__ void k_flow(DevGraph g, int *flow);
__global__ void k_tally_sum(DevGraph g, unsigned long long *out, int nslots);
__global__ void k_reset_group(const DevGraph *gs, const GroupAux *aux);
__global__ void k_mirror_group(const DevGraph *gs, const GroupAux *aux);
__global__ void k_excess_from_flow_group(const DevGraph *gs, const GroupAux *aux);
__global__ void k_total_cost_group(const DevGraph *gs, const GroupAux *aux);
__global__ void k_tally_sum_group(const DevGraph *gs, const GroupAux *aux, int nslots);
__global__ void k_ctrl_scatter_group(const DevGraph *gs, const Ctrl *init, int count);
__global__ void k_eval_costs(DevGraph g, int E, int N, int K, const i64 *cap_cpu, const i64 *cap_mem, const i64 *used_cpu,
                             const i64 *used_mem, const i64 *req_cpu, const i64 *req_mem, const i64 *ec_arc_base, int *mcost,
                             int *mocap);
__global__ void k_placements(DevGraph g, int N, int E, int K, int v_task0, int n_tasks, int *slot_machine, int slot_stride, int *place,
                             unsigned *bad);
__global__ void k_fill_i32(int *p, int v, long long n);
__global__ void k_scatter_caps(DevGraph g, long long count, const i64 *idx, const int *ncap, int *mocap);
__global__ void k_clamp_flow(DevGraph g);
__global__ void k_scatter_supply(long long count, const int *vtx, const i64 *val, i64 *supply);

}  // namespace psn
#endif
