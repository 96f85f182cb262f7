/*
 * libpsn scheduler service core: the state machine behind the Firmament scheduler RPCs,
 * host-side C ABI above the solver (include/psn.h).
 *
 * One psn_svc is what a Go type implementing FirmamentSchedulerServer
 * (/root/reference/pkg/firmament/firmament_scheduler.pb.go:600-614) holds: every RPC of
 * the service (/root/reference/pkg/firmament/firmament_scheduler.proto:15-45) maps to one
 * call below, with the protobuf payload reduced to the fields the reference actually fills
 * in (/root/reference/pkg/k8sclient/podwatcher.go:377-410,
 * /root/reference/pkg/k8sclient/nodewatcher.go:292-339).  The calls return the reply enum
 * of the RPC (values of TaskReplyType / NodeReplyType,
 * /root/reference/pkg/firmament/firmament_scheduler.proto:110-129) or a negative PSN_E*
 * code; the unmodified client accepts exactly those replies
 * (/root/reference/pkg/firmament/firmament_client.go:43-179).
 *
 * The service keeps the flow network resident on the GPU between rounds.  Tasks, jobs,
 * classes and machines own fixed places in a network that is laid out with head-room;
 * an RPC edits the host mirror and queues the few arcs, capacities and supplies it
 * touched; Schedule ships those to the device and repairs the previous solution
 * (warm start).  Only when head-room runs out is the network laid out and uploaded again.
 *
 * Not thread-safe: the caller serialises calls (one mutex around the service, as
 * Firmament's own scheduling lock does).
 */
#ifndef PSN_SERVICE_H_
#define PSN_SERVICE_H_

#include <stdint.h>
#include "psn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* TaskReplyType, /root/reference/pkg/firmament/firmament_scheduler.proto:110-120 */
#define PSN_TASK_COMPLETED_OK 0
#define PSN_TASK_SUBMITTED_OK 1
#define PSN_TASK_REMOVED_OK 2
#define PSN_TASK_FAILED_OK 3
#define PSN_TASK_UPDATED_OK 4
#define PSN_TASK_NOT_FOUND 5
#define PSN_TASK_JOB_NOT_FOUND 6
#define PSN_TASK_ALREADY_SUBMITTED 7
#define PSN_TASK_STATE_NOT_CREATED 8
/* NodeReplyType, :122-129 */
#define PSN_NODE_ADDED_OK 0
#define PSN_NODE_FAILED_OK 1
#define PSN_NODE_REMOVED_OK 2
#define PSN_NODE_UPDATED_OK 3
#define PSN_NODE_NOT_FOUND 4
#define PSN_NODE_ALREADY_EXISTS 5
/* ServingStatus, :131-135 */
#define PSN_SERVING_UNKNOWN 0
#define PSN_SERVING 1
#define PSN_NOT_SERVING 2
/* SchedulingDelta.ChangeType, /root/reference/pkg/firmament/scheduling_delta.proto:26-35 */
#define PSN_DELTA_NOOP 0
#define PSN_DELTA_PLACE 1
#define PSN_DELTA_PREEMPT 2
#define PSN_DELTA_MIGRATE 3
/* TaskDescriptor.TaskState CREATED, /root/reference/pkg/firmament/task_desc.proto:32-33 */
#define PSN_TASK_STATE_CREATED 0

typedef struct psn_svc psn_svc;

typedef struct psn_svc_config {
    int32_t device;            /* HIP device ordinal; -1: no device (state machine and network
                                  export only, psn_svc_schedule fails with PSN_ESTATE) */
    int32_t max_machines;      /* initial capacities of the resident network; 0: sized from */
    int32_t max_tasks;         /*   what is known at the first Schedule plus 25 % */
    int32_t max_jobs;
    int32_t max_classes;       /* distinct (cpu, memory) request shapes; 0 -> 64 */
    int32_t k_slots;           /* parallel unit arcs per (class, machine); 0 -> 8 */
    int32_t slots_per_machine; /* machine -> sink capacity; 0 -> 12 */
    int32_t unsched_cost;      /* cost of leaving a task unscheduled; 0 -> 2 560 000 */
    int32_t emit_noop;         /* 1: report unchanged tasks as NOOP deltas too */
    int32_t reschedule_running;/* 0 (default): a task that has been placed keeps its machine until it
                                  completes, fails, is removed or the machine goes away; it leaves the
                                  network, its request counts as usage of the machine and it holds one of
                                  the machine's slots, so a round only ever answers PLACE (the unmodified
                                  client turns MIGRATE and PREEMPT into pod deletions,
                                  /root/reference/cmd/poseidon/poseidon.go:52-63).  1: every runnable task
                                  is re-optimised in every round and may be migrated or preempted (what-if
                                  and defragmentation runs; the incremental benchmark configuration). */
    int32_t stats_utilization; /* how psn_svc_add_node_stats_raw reads a utilisation figure: 1 always as a
                                  fraction of the capacity (/root/reference/pkg/firmament/resource_stats.proto:36-39,56-59),
                                  2 always as an absolute quantity (what /root/reference/pkg/stats/stats_test.go:100-127
                                  carries), 0 (default) by value: up to 1.0 a fraction, above it a quantity -
                                  ambiguous for absolute readings of at most one unit */
    psn_config solver;         /* passed to psn_create (device field overwritten) */
} psn_svc_config;

/* One label selector of a task (/root/reference/pkg/firmament/label_selector.proto:24-35;
 * type 0 IN_SET, 1 NOT_IN_SET, 2 EXISTS_KEY, 3 NOT_EXISTS_KEY).  The reference sends a pod's
 * nodeSelector as one IN_SET selector per key
 * (/root/reference/pkg/k8sclient/podwatcher.go:455-465). */
typedef struct psn_selector {
    int32_t type;
    const char *key;
    const char *const *values;
    int32_t n_values;
} psn_selector;

typedef struct psn_svc_stats {
    int64_t rounds, rebuilds, warm_rounds; /* Schedule calls, full uploads, warm-started solves */
    int64_t patched_arcs, cap_updates, supply_updates; /* records shipped by the last Schedule */
    int64_t tasks_runnable, tasks_placed, machines_live;
    int64_t n, m2;                          /* resident network size */
    double build_ms, upload_ms, update_ms, solve_ms, extract_ms; /* last Schedule, host clock */
    int64_t total_cost;
    int32_t last_round_warm;
    int64_t tasks_running;   /* tasks that hold a machine outside the network (reschedule_running = 0) */
    double total_ms;         /* last Schedule from entry to return, host clock */
    int64_t tasks_deferred;  /* last Schedule: placements not admitted because the request no longer fitted next
                                to what other classes were given on that machine in the same round */
    /* the solve of the last Schedule (psn_result fields of the same names) */
    int64_t solve_replays, solve_last_phase_replays, solve_refines, solve_rounds, solve_global_updates, solve_pushes,
        solve_first_units, solve_first_eps_log2;
    double solve_device_ms;
    int64_t solve_warm_declined;  /* 1: the round was applied in place and offered its kept state, the solver started cold (psn_result::warm_declined) */
} psn_svc_stats;

int psn_svc_create(const psn_svc_config *cfg, psn_svc **out);
void psn_svc_destroy(psn_svc *svc);
const char *psn_svc_last_error(const psn_svc *svc);

/* TaskSubmitted (/root/reference/pkg/firmament/firmament_scheduler.proto:26): task uid
 * (/root/reference/pkg/k8sclient/podwatcher.go:420-422), job uuid, TaskDescriptor.state,
 * cpu request in millicores and memory request in Kb (podwatcher.go:382-386). */
int psn_svc_task_submitted(psn_svc *svc, uint64_t task_uid, const char *job_uuid, int32_t state,
                           int64_t cpu_millicores, int64_t ram_kb);
/* TaskSubmitted with the task's label selectors: the task may only be placed on machines
 * whose labels satisfy all of them.  Tasks with the same request and the same selectors
 * share a class. */
int psn_svc_task_submitted_sel(psn_svc *svc, uint64_t task_uid, const char *job_uuid, int32_t state,
                               int64_t cpu_millicores, int64_t ram_kb, const psn_selector *selectors,
                               int32_t n_selectors);
/* TaskUpdated (:28): same payload; the task may change its request shape. */
int psn_svc_task_updated(psn_svc *svc, uint64_t task_uid, const char *job_uuid, int64_t cpu_millicores,
                         int64_t ram_kb);
/* TaskCompleted / TaskFailed / TaskRemoved (:20-24). */
int psn_svc_task_completed(psn_svc *svc, uint64_t task_uid);
int psn_svc_task_failed(psn_svc *svc, uint64_t task_uid);
int psn_svc_task_removed(psn_svc *svc, uint64_t task_uid);

/* NodeAdded / NodeUpdated (:31,37): resource uuid of the machine
 * (/root/reference/pkg/k8sclient/nodewatcher.go:296-306) and its capacity in millicores / Kb
 * (nodewatcher.go:300-303).  NodeFailed / NodeRemoved (:33-35). */
int psn_svc_node_added(psn_svc *svc, const char *resource_uuid, int64_t cpu_millicores, int64_t ram_kb);
int psn_svc_node_updated(psn_svc *svc, const char *resource_uuid, int64_t cpu_millicores, int64_t ram_kb);
int psn_svc_node_failed(psn_svc *svc, const char *resource_uuid);
int psn_svc_node_removed(psn_svc *svc, const char *resource_uuid);
/* Labels of a machine as NodeAdded / NodeUpdated carry them
 * (/root/reference/pkg/k8sclient/nodewatcher.go:309-315): replaces the machine's label set.
 * Reply as for NodeUpdated. */
int psn_svc_node_labels(psn_svc *svc, const char *resource_uuid, const char *const *keys,
                        const char *const *values, int32_t n_labels);
/* AddNodeStats (:42): resources in use on the machine by anything that is not one of the
 * tasks known to the service; feeds the cost model, where the requests of the tasks the service
 * has running on the machine are added to it.  Reply as for NodeUpdated. */
int psn_svc_add_node_stats(psn_svc *svc, const char *resource_uuid, int64_t used_cpu_millicores, int64_t used_ram_kb);
/* AddNodeStats with the fields of the message itself (ResourceStats,
 * /root/reference/pkg/firmament/resource_stats.proto:22-59, filled at
 * /root/reference/pkg/stats/stats.go:60-75): per-CPU (or whole-node, one entry) capacity in
 * millicores and utilisation, memory capacity in Kb and utilisation.  A utilisation up to 1 is
 * read as a fraction of the capacity (what the message says), a larger one as a quantity (what
 * the reference's vectors carry, /root/reference/pkg/stats/stats_test.go:100-127).  The figure
 * measures the whole machine, the tasks the service runs there included; their share (observed
 * usage where AddTaskStats reported it, the request otherwise) is taken out to get the usage by
 * everything else, to which the requests of the running tasks are then added as for
 * psn_svc_add_node_stats.  A later psn_svc_add_node_stats replaces the measurement. */
int psn_svc_add_node_stats_raw(psn_svc *svc, const char *resource_uuid, int32_t n_cpus, const int64_t *cpu_capacity,
                               const double *cpu_utilization, int64_t mem_capacity, double mem_utilization);
/* AddTaskStats (:40).  psn_svc_add_task_stats only acknowledges (the client ignores the reply,
 * /root/reference/pkg/firmament/firmament_client.go:183-188); the _raw form records the task's
 * observed cpu_usage (millicores) and mem_usage (Kb) (TaskStats,
 * /root/reference/pkg/firmament/task_stats.proto:22-50, filled at
 * /root/reference/pkg/stats/stats.go:33-57).  What a running task holds on its machine stays
 * its request (that is what is reserved for it); the observed usage tells how much of a measured
 * machine figure is the task's own. */
int psn_svc_add_task_stats(psn_svc *svc, uint64_t task_uid);
int psn_svc_add_task_stats_raw(psn_svc *svc, uint64_t task_uid, int64_t cpu_usage_millicores, int64_t mem_usage_kb);
/* Check (:44). */
int psn_svc_check(const psn_svc *svc);

/* Schedule (:17): one scheduling round.  Returns the number of deltas (>= 0) or a negative
 * PSN_E* code; the deltas stay readable until the next round.  A round with nothing queued
 * since the last completed one (no task, node, label or statistics RPC) returns 0 at once:
 * the resident optimum still stands (not so with emit_noop, which reports every task). */
int64_t psn_svc_schedule(psn_svc *svc);
/* Copy deltas [first, first+count): task uid, change type and the 36-character resource
 * uuid of the machine (NUL-terminated, 37 bytes per delta; empty for PREEMPT). */
int psn_svc_get_deltas(psn_svc *svc, int64_t first, int64_t count, uint64_t *task_uid, int32_t *type,
                       char *resource_uuid /* count * 37 */);
int psn_svc_get_stats(const psn_svc *svc, psn_svc_stats *out);

/* Inspection: the host mirror of the resident network as psn_upload_csr would take it
 * (slot arcs unpriced: cost 0, capacity 0) plus what psn_eval_costs needs.  Sizes first
 * (any pointer may be NULL), then the arrays. */
int psn_svc_network_size(psn_svc *svc, psn_layout *layout, int32_t *n, int64_t *m2);
int psn_svc_export_network(psn_svc *svc, int64_t *row_off, int32_t *head, int32_t *cap, int32_t *cost,
                           int32_t *rev, int64_t *supply, int64_t *ec_arc_base, int64_t *cap_cpu,
                           int64_t *cap_mem, int64_t *used_cpu, int64_t *used_mem, int64_t *req_cpu,
                           int64_t *req_mem);
/* Admission mask of the resident network as psn_set_admission_mask takes it:
 * n_classes rows of (n_machines + 31) / 32 words. */
int psn_svc_export_mask(psn_svc *svc, uint32_t *mask);
/* Current placement of a task: machine uuid (37 bytes) or empty when unscheduled/unknown. */
int psn_svc_task_placement(psn_svc *svc, uint64_t task_uid, char *resource_uuid /* 37 */);
/* The solver context of the service (NULL without a device), for read-backs in tests. */
psn_ctx *psn_svc_solver(psn_svc *svc);

#ifdef __cplusplus
}
#endif
#endif /* PSN_SERVICE_H_ */
