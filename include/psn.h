/*
 * libpsn - MI355X (gfx950) min-cost max-flow scheduling solver: C ABI.
 *
 * This is the drop-in boundary for the scheduling round that Poseidon triggers with
 * the Schedule RPC (/root/reference/pkg/firmament/firmament_scheduler.proto:17,
 * client wrapper /root/reference/pkg/firmament/firmament_client.go:29-35, caller
 * /root/reference/cmd/poseidon/poseidon.go:34).  In the reference everything behind
 * that RPC (flow-graph update, arc pricing, min-cost max-flow solve, flow -> deltas)
 * runs in an external Firmament process that is not part of the repository
 * (/root/reference/.gitmodules:1-3, /root/reference/deploy/firmament-deployment.yaml:29-30).
 * A Go server implementing FirmamentSchedulerServer
 * (/root/reference/pkg/firmament/firmament_scheduler.pb.go:600-614) binds these
 * entry points through cgo; the declarations use plain pointers and sizes only.
 *
 * Conventions
 *  - every function returns PSN_OK (0) or a negative PSN_E* code; nothing aborts or
 *    throws across the boundary; psn_strerror() gives the text, psn_last_error()
 *    the detail (HIP error string) of the last failure on a context;
 *  - the caller owns every host buffer; the library copies inputs before returning
 *    and never keeps a caller pointer (cgo rule); outputs are caller-allocated;
 *  - the library owns all device memory, inside psn_ctx;
 *  - a psn_ctx is bound to one GPU and is not re-entrant: one call at a time.
 */
#ifndef PSN_H_
#define PSN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSN_OK 0
#define PSN_EINVAL (-1)      /* bad argument / malformed graph */
#define PSN_ENOMEM (-2)      /* host or device allocation failed */
#define PSN_EDEVICE (-3)     /* HIP runtime error (see psn_last_error) */
#define PSN_EINFEASIBLE (-4) /* supplies cannot be routed */
#define PSN_EOVERFLOW (-5)   /* 8 n (n+1) max|cost| > 4e18: scaled prices might not fit int64 (the cost scale
                              * is 2|C|+1 for a vertex cover C, n+1 at most; the check assumes the worst) */
#define PSN_ESTATE (-6)      /* call out of order (e.g. solve before upload) */
#define PSN_ELIMIT (-7)      /* iteration guard tripped without convergence */

typedef struct psn_ctx psn_ctx;

typedef struct psn_config {
    int32_t device;          /* HIP device ordinal */
    int32_t alpha_log2;      /* epsilon divisor per refine, alpha = 2^alpha_log2; 0 -> default alpha_log2 = 4, i.e. alpha = 16 */
    int32_t rounds_per_batch;/* push/relabel rounds between global price updates (0 -> default) */
    int32_t flags;           /* PSN_CFG_* */
} psn_config;
#define PSN_CFG_EAGER 1      /* plain launches instead of hipGraph replay (debugging) */
#define PSN_CFG_EPS_FROM_MAX 4 /* first epsilon from the largest |cost| (textbook) instead of the 90th-percentile cost */
#define PSN_CFG_KTIME 2      /* with PSN_CFG_EAGER: HIP-event time every sweep launch (psn_kernel_times) */

/* kernel classes reported by psn_kernel_times */
#define PSN_KT_PUSH 0        /* push sweep over the active lists */
#define PSN_KT_GU_EXPAND 1   /* global price update: frontier expansion */
#define PSN_KT_RELABEL 2     /* relabel + active-list compaction */
#define PSN_KT_SATURATE 3    /* per-phase saturation sweep */
#define PSN_KT_OTHER 4
#define PSN_KT_CLASSES 5

/* Result of one solve.  Counters are what the kernels actually did; bytes follow
 * from them (20 B per arc scan, 36 B per push, 32 B per vertex visit, 8 B per
 * relabel). */
typedef struct psn_result {
    int64_t total_cost;      /* sum over arcs of flow * cost */
    int32_t status;          /* PSN_OK or PSN_E* */
    int32_t refines;         /* epsilon phases run */
    uint64_t arc_scans;      /* residual arcs examined by push/relabel sweeps */
    uint64_t pushes;
    uint64_t relabels;
    uint64_t vertex_visits;  /* active vertices picked up by sweeps */
    uint64_t gu_arc_scans;   /* residual arcs examined by global price updates */
    uint64_t gu_relaxations; /* successful distance improvements */
    uint64_t sat_arc_scans;  /* arcs examined by the per-phase saturation sweep */
    uint64_t rounds;         /* push/relabel rounds that had work */
    uint64_t gu_iterations;  /* frontier expansions that had work */
    uint64_t global_updates;
    uint64_t launches;       /* kernel launches issued (graph nodes included) */
    double solve_ms;         /* device time, first refine kernel -> cost readback */
    uint64_t gu_arc_skips;   /* of gu_arc_scans: arcs in 576-arc steps of split rows where no lane saw residual
                              * capacity, so only the 4-byte capacity word was read (not head, cost, distance) */
    uint64_t arc_skips;      /* of arc_scans: arcs without residual capacity in rows whose residual arcs a push sweep
                              * handed out one per lane; only their 4-byte capacity word was read */
    int32_t warm_started;    /* 1: the solve resumed from the kept flow and prices; 0: it started cold (no
                              * PSN_WARM_START, nothing solved before, or an update invalidated the kept state) */
    int32_t mirrored_replays; /* replays run on the reversed network (deficits pushed towards the excesses) */
    uint64_t replays;        /* passes over the fixed launch sequence (one global update + a batch of rounds each) */
    uint64_t first_units;    /* units of excess at the work-list vertices when the first round of the solve started,
                              * i.e. what the first phase's saturation (plus, warm, the changes since the last solve) set free */
    int32_t first_eps_log2;  /* log2 of the epsilon the first phase ran at (scaled costs) */
    int32_t last_phase_replays; /* replays spent at epsilon = 1 */
    int32_t warm_declined;   /* 1: PSN_WARM_START was asked for and the kept state was there, but it is violated by about as
                              * much as an empty one (its first epsilon within a power of two of a cold solve's), so the
                              * solve started cold (warm_started = 0) */
    int32_t reserved_;
} psn_result;

#define PSN_WARM_START 1u    /* psn_solve: keep prices and flow from the previous solve (psn_solve declines a kept state that is no
                              * head start and says so in psn_result::warm_declined; psn_solve_batch resumes whenever asked) */

/* Shape of a psn flow network (vertex id ranges), needed by cost evaluation and
 * placement extraction.  Vertex ids: 0 sink | 1..N machines | N+1..N+E task
 * classes | then J unscheduled aggregators | then P tasks. */
typedef struct psn_layout {
    int32_t n_machines, n_classes, n_jobs, n_tasks, k_slots;
} psn_layout;

int psn_version(void);
/* Names of the kernels one solve launches (comma separated, without namespace or arguments), so that a
 * profile taken from one build can be told from a profile of another: bench.py refuses to quote HBM
 * traffic from a committed counter pass whose kernel list differs. */
const char *psn_kernel_list(void);
const char *psn_strerror(int code);
const char *psn_last_error(const psn_ctx *ctx);

/* Lifetime.  Replaces connecting to the Firmament service
 * (/root/reference/pkg/firmament/firmament_client.go:211-221). */
int psn_create(const psn_config *cfg, psn_ctx **out);
void psn_destroy(psn_ctx *ctx);

/* Hand the flow network to the device as residual CSR: arcs grouped by tail,
 * row_off[n+1]; for each residual arc head, residual capacity, cost (reverse arc
 * carries -cost) and the index of its paired arc.  supply[n] sums to zero.
 * This is what the reference's TaskSubmitted / NodeAdded stream
 * (/root/reference/pkg/firmament/firmament_scheduler.proto:26,31) accumulates
 * into inside Firmament before a Schedule call. */
int psn_upload_csr(psn_ctx *ctx, int32_t n, int64_t m2, const int64_t *row_off,
                   const int32_t *head, const int32_t *cap, const int32_t *cost,
                   const int32_t *rev, const int64_t *supply);

/* The same hand-over for a scheduling network, built on the device: the residual CSR that
 * psn_build_network (include/psn_builder.h) would lay out for this layout and these tasks is
 * written straight into the context, arc for arc the same, from two words per task (class and
 * job index) instead of being assembled on the host and copied (300 MB at 10 000 machines).
 * Machine -> sink arcs get capacity `slots`, task -> unscheduled arcs cost `unsched_cost`; slot
 * arcs are left unpriced for psn_eval_costs, whose ec_arc_base argument is returned in
 * ec_arc_base_out (n_classes entries, may be NULL).  This is the flow-graph construction that
 * Firmament does from the TaskSubmitted / NodeAdded stream
 * (/root/reference/pkg/firmament/firmament_scheduler.proto:26,31) before a Schedule call. */
int psn_build_network_device(psn_ctx *ctx, const psn_layout *layout, int32_t slots, int32_t unsched_cost,
                             const int32_t *task_class, const int32_t *task_job, int64_t *ec_arc_base_out);

/* psn_build_network_device for a network laid out with head-room, the way a service keeps it
 * resident between rounds: class e's row has room_class[e] reverse places for member tasks before
 * its slot arcs, job j's row room_job[j] before its arc to the sink, and there are n_tasks task
 * slots; n_attached tasks are given by slot, class, job and the place each takes in its class row
 * and job row (all distinct).  Places and slots that nobody takes are placeholders (self-loops
 * paired with themselves, capacity 0) which psn_patch_arcs can hand to tasks that arrive later.
 * Machine m's arc to the sink gets capacity machine_sink_cap[m], job j's job_sink_cap[j]. */
int psn_build_network_rooms_device(psn_ctx *ctx, const psn_layout *layout, int32_t unsched_cost,
                                   const int32_t *room_class, const int32_t *room_job,
                                   const int32_t *machine_sink_cap, const int32_t *job_sink_cap,
                                   int64_t n_attached, const int32_t *task_slot, const int32_t *task_class,
                                   const int32_t *task_job, const int32_t *place_class, const int32_t *place_job,
                                   int64_t *ec_arc_base_out);

/* CPU/memory arc-cost model on the device: prices every (task class e, machine m,
 * slot k) arc of the uploaded network in place (cost and capacity of the forward
 * arc, -cost / 0 on its reverse).  ec_arc_base[e] is the index of class e's first
 * slot arc; arcs of one class are laid out machine-major, slot-minor, and their paired
 * reverse arcs open the row of their machine (vertex 1 + m) class-major, slot-minor, as
 * psn_build_network lays them out (anything else is PSN_EINVAL).  Stands in
 * for Firmament's cpu/mem cost model selected at
 * /root/reference/deploy/firmament-deployment.yaml:29 with the units of
 * /root/reference/pkg/k8sclient/podwatcher.go:382-386 (millicores, Kb). */
int psn_eval_costs(psn_ctx *ctx, const psn_layout *layout, const int64_t *cap_cpu,
                   const int64_t *cap_mem, const int64_t *used_cpu, const int64_t *used_mem,
                   const int64_t *req_cpu, const int64_t *req_mem, const int64_t *ec_arc_base);

/* Overwrite residual capacities / supplies of the uploaded network (same layout),
 * for incremental re-solves: entries are (arc index, new original capacity) and
 * (vertex, new supply).  With PSN_WARM_START the next solve repairs the kept flow. */
int psn_update_caps(psn_ctx *ctx, int64_t count, const int64_t *arc_index, const int32_t *new_cap);
int psn_update_supplies(psn_ctx *ctx, int64_t count, const int32_t *vertex, const int64_t *new_supply);
/* New costs of residual arcs (arc index, cost; the paired arc gets the negated cost).  Flow and
 * prices are kept: with PSN_WARM_START the next solve repairs the optimality conditions the
 * change broke.  With psn_update_caps this is what a DIMACS change record amounts to
 * (include/psn_builder.h). */
int psn_update_costs(psn_ctx *ctx, int64_t count, const int64_t *arc_index, const int32_t *new_cost);

/* Label selectors as an admission mask: bit (m & 31) of word mask[e * ((N+31)/32) + m/32]
 * says whether tasks of class e may run on machine m.  Once set, psn_eval_costs and
 * psn_reprice_costs give the slot arcs of a forbidden (class, machine) pair capacity 0;
 * NULL lifts the mask.  The mask belongs to the uploaded network and goes away with it.
 * The reference forwards a pod's nodeSelector as IN_SET selectors
 * (/root/reference/pkg/k8sclient/podwatcher.go:399,455-465,
 * /root/reference/pkg/firmament/label_selector.proto:24-35) and node labels
 * (/root/reference/pkg/k8sclient/nodewatcher.go:309-315); the caller folds the selector
 * set into the class key and evaluates the match per (class, machine). */
int psn_set_admission_mask(psn_ctx *ctx, const psn_layout *layout, const uint32_t *mask);

/* Same pricing as psn_eval_costs, for a network that keeps its state between rounds: the
 * flow on every slot arc is kept (cut down to the new capacity), prices and the solved
 * state stay, so the next psn_solve(PSN_WARM_START) repairs instead of starting over.
 * This is the path of a machine whose usage or capacity changed (NodeUpdated,
 * /root/reference/pkg/firmament/firmament_scheduler.proto:37; AddNodeStats, :41). */
int psn_reprice_costs(psn_ctx *ctx, const psn_layout *layout, const int64_t *cap_cpu,
                      const int64_t *cap_mem, const int64_t *used_cpu, const int64_t *used_mem,
                      const int64_t *req_cpu, const int64_t *req_mem, const int64_t *ec_arc_base);

/* Rewire residual arcs in place: entry i overwrites arc arc_index[i] with (head, original
 * capacity, cost, index of its paired arc); the tail of an arc is fixed by its row.  Flow
 * on a patched arc is dropped, so both arcs of a pair must be patched together; an unused
 * slot is parked as a placeholder paired with itself (head = its own tail, capacity 0,
 * cost 0).  The device checks that every patched arc pairs up again; if not, PSN_EINVAL is
 * returned and the network has to be uploaded again.  This is how a task that arrives,
 * leaves or changes class (TaskSubmitted / TaskRemoved,
 * /root/reference/pkg/firmament/firmament_scheduler.proto:24,26) changes the resident
 * network without a new upload. */
int psn_patch_arcs(psn_ctx *ctx, int64_t count, const int64_t *arc_index, const int32_t *head,
                   const int32_t *cap, const int32_t *cost, const int32_t *rev);

/* Give vertices that no residual arc enters (new tasks) the lowest price that leaves none
 * of their own arcs violated, so that a warm start does not restart from a large epsilon
 * because of them.  Vertices that are entered are left alone and counted in *skipped. */
int psn_settle_sources(psn_ctx *ctx, int64_t count, const int32_t *vertex, int64_t *skipped);

/* The scheduling solve itself: cost-scaling push-relabel on the device.  Replaces
 * the solver run inside Firmament's Schedule handler
 * (/root/reference/pkg/firmament/firmament_scheduler.proto:17). */
int psn_solve(psn_ctx *ctx, uint32_t flags, psn_result *out);

/* Solve `count` independent networks, one per context (each uploaded and priced
 * beforehand, all contexts distinct).  Contexts on one device with the same solver
 * configuration are solved in lockstep groups of up to 128 - two groups of half the contexts
 * each from eight contexts on, since two groups side by side fill each other's gaps
 * (environment PSN_BATCH_GROUP: 1..256 fixes the largest group instead; 1 = one at a time):
 * every stage of the solve is one launch for the whole group,
 * which is what bounds many small networks; the networks share nothing and each stops
 * when it is done.  Fewer than four contexts that could share a group are solved one after
 * the other instead (a group's launches carry every member's idle stages).  The contexts may
 * live on several GPUs of the node: grouping is per device, and the groups are spread over
 * `workers` host threads (0 -> two per device when every context is in a group, else 4 or
 * two per device when that is more; never fewer than one per device), each driving one
 * stream, so the devices work side by side.  A group's replay
 * graph is kept by its first context and reused while the same contexts, in the same order,
 * keep their uploads.  out[i] receives the result of ctxs[i] (solve_ms of a group member is the
 * group's time); the first error code is returned, the other solves still run.  This is
 * the batch of what-if scheduling rounds: each is the same Schedule solve
 * (/root/reference/pkg/firmament/firmament_scheduler.proto:17) on its own cluster state. */
int psn_solve_batch(psn_ctx *const *ctxs, int32_t count, uint32_t flags, int32_t workers,
                    psn_result *out);

/* Read back: flow on every residual arc (original minus residual capacity, >= 0 on
 * arcs carrying flow, <= 0 on their pairs), scaled prices (1-optimal for costs
 * multiplied by the factor psn_price_scale reports), and task -> machine index (-1 = unscheduled) for psn-layout
 * networks.  Placements feed the SchedulingDelta list the caller returns
 * (/root/reference/pkg/firmament/scheduling_delta.proto:25-41). */
int psn_get_flow(psn_ctx *ctx, int32_t *flow /* m2 */);
/* The factor S the solver multiplies costs by: the prices psn_get_prices returns are 1-optimal
 * for costs * S (cost[a] * S + price[tail] - price[head] >= -1 on every residual arc), and S
 * exceeds the length of every simple cycle of the network, which makes such a flow optimal.  S is
 * n + 1 at most; the library derives a smaller bound from a vertex cover of the network (a simple
 * cycle is at most twice as long as a vertex cover is large), which saves refine phases. */
int psn_price_scale(psn_ctx *ctx, int64_t *scale);
int psn_get_residual(psn_ctx *ctx, int32_t *cap /* m2 */);
int psn_get_prices(psn_ctx *ctx, int64_t *price /* n */);
int psn_get_costs(psn_ctx *ctx, int32_t *cost /* m2 */, int32_t *cap_orig /* m2 */);
/* The structure of the resident network as psn_upload_csr takes it: heads and paired-arc
 * indices (m2 entries each), row offsets (n + 1) and supplies (n); any pointer may be NULL.
 * With psn_get_costs this reads back everything psn_build_network_device wrote. */
int psn_get_structure(psn_ctx *ctx, int32_t *head /* m2 */, int32_t *rev /* m2 */);
int psn_get_rows(psn_ctx *ctx, int64_t *row_off /* n + 1 */, int64_t *supply /* n */);
int psn_get_placements(psn_ctx *ctx, const psn_layout *layout, int32_t *task_to_machine /* P */);

/* Same as psn_get_placements but into caller-owned device memory (int32[P]) on this
 * context's GPU, e.g. the send buffer of an all-gather; returns after the copy is done. */
int psn_get_placements_device(psn_ctx *ctx, const psn_layout *layout, void *dst_device);

/* psn_get_placements_device for callers that cannot allocate (tight serving loops, harnesses
 * that forbid hipMalloc): same result into the same kind of
 * caller-owned device buffer, but the per-class unit lists the extraction kernel needs live in a
 * fixed 4 MiB device array that every GPU the library runs on holds from the first psn_create
 * on it until the process ends, and the call makes no device or host allocation of any kind.
 * Limit: n_classes * n_tasks <= 1 048 576 (PSN_EINVAL beyond it, nothing launched).  All
 * contexts of one GPU share that array: calls that overlap on one GPU race, the caller
 * serialises them.  The call waits for its stream (it reads back the kernel's layout verdict), so it
 * cannot itself run under HIP stream capture.  This entry point has no counterpart in the reference. */
int psn_get_placements_device_noalloc(psn_ctx *ctx, const psn_layout *layout, void *dst_device);

/* Placements of `count` solved contexts into one buffer, row i (int32, `stride` entries apart,
 * layouts[i].n_tasks of them written) for ctxs[i].  The contexts may live on different GPUs of
 * the node: every context runs its extraction on its own device and stream, then the rows are
 * copied to host memory (dst_device < 0) or into memory of GPU dst_device (peer copies over
 * xGMI for contexts on other GPUs).  With psn_solve_batch this is the whole multi-GPU batch of
 * what-if rounds from one process: each row is what a Schedule reply would carry as deltas
 * (/root/reference/pkg/firmament/scheduling_delta.proto:25-41) for its cluster state. */
int psn_gather_placements(psn_ctx *const *ctxs, int32_t count, const psn_layout *layouts, int64_t stride,
                          int32_t dst_device, void *dst);

/* Number of device / pinned-host allocations the library has made in this process so far.  A
 * caller can bracket a warm scheduling round with it: the per-round calls (psn_patch_arcs,
 * psn_update_*, psn_reprice_costs, psn_settle_sources, psn_solve, psn_get_placements, and
 * psn_solve_batch over an unchanged set of contexts) reuse what earlier rounds allocated. */
uint64_t psn_debug_alloc_count(void);

/* Per-kernel-class device time (ms) and launch counts of the last psn_solve, measured
 * with HIP events on the solver's stream; all zero unless the context was created with
 * PSN_CFG_EAGER | PSN_CFG_KTIME.  Arrays of PSN_KT_CLASSES entries. */
int psn_kernel_times(psn_ctx *ctx, double *ms, uint64_t *calls);

/* Development aid: with PSN_TRACE=1 in the environment when a network is uploaded, one
 * record (kind: 1 push / 2 global-update expand, k, lane-list size, item-list size) per
 * sweep launch that had work during the last solve. */
int psn_debug_trace(psn_ctx *ctx, int32_t *out4, int64_t max_records, int64_t *count);

/* Device pointer to the last placements (int32[P]) for collectives that gather
 * placements across GPUs without a host round trip; valid until the next call. */
int psn_placements_device_ptr(psn_ctx *ctx, void **dptr, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* PSN_H_ */
