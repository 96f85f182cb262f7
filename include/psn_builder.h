/*
 * libpsn flow-graph builder: host-side C ABI above the solver (include/psn.h).
 *
 * In the new split the Go server keeps the protobuf surface
 * (/root/reference/pkg/firmament/firmament_scheduler.pb.go:600-614) and turns the
 * accumulated tasks (/root/reference/pkg/k8sclient/podwatcher.go:377-410) and machines
 * (/root/reference/pkg/k8sclient/nodewatcher.go:292-339, one PU per machine) into the
 * task -> class -> machine -> sink flow network.  These entry points are that builder,
 * written in C++ because no Go toolchain is available; the Go side would call them
 * through cgo exactly as it calls psn_upload_csr.
 */
#ifndef PSN_BUILDER_H_
#define PSN_BUILDER_H_

#include <stdint.h>
#include "psn.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PSN_UNSCHED_COST 2560000 /* cost of leaving a task unscheduled */
#define PSN_SLOTS 12             /* task slots per machine (machine -> sink capacity) */
#define PSN_KSLOTS 8             /* parallel unit arcs per (class, machine) pair */
#define PSN_NCLASSES 64          /* request shapes: 8 cpu x 8 memory sizes */

/* Deterministic synthetic cluster (family psn-synth-v1): splitmix64 seeded with
 * 0x506F736569646F6E ^ (config_id << 32) ^ seed; per machine four draws (cpu size,
 * GiB per core, used cpu permille, used mem permille), then per task two draws
 * (class, job); every draw is next() % range.  CPU in millicores and memory in Kb,
 * the units the reference sends (/root/reference/pkg/k8sclient/podwatcher.go:382-386).
 * All arrays are caller-allocated: machines N, classes E (= 64), tasks P. */
int psn_synth_cluster(int32_t config_id, uint64_t seed, const psn_layout *layout, int64_t *cap_cpu,
                      int64_t *cap_mem, int64_t *used_cpu, int64_t *used_mem, int64_t *req_cpu,
                      int64_t *req_mem, int32_t *task_class, int32_t *task_job);

/* Vertex and residual-arc counts of the network for a layout. */
int psn_network_size(const psn_layout *layout, int32_t *n, int64_t *m2);

/* Residual CSR of the network (arrays sized by psn_network_size).  Slot arcs
 * (class -> machine) are emitted with cost 0 and capacity 0; psn_eval_costs prices
 * them on the device.  ec_arc_base[E] receives the index of each class's first slot arc. */
int psn_build_network(const psn_layout *layout, int32_t slots, int32_t unsched_cost,
                      const int32_t *task_class, const int32_t *task_job, int64_t *row_off,
                      int32_t *head, int32_t *cap, int32_t *cost, int32_t *rev, int64_t *supply,
                      int64_t *ec_arc_base);

/* Residual CSR from a plain arc list (0-based ids): rows keep arc-list order, each arc
 * contributes its forward entry to the row of its tail and its reverse entry (capacity 0,
 * negated cost) to the row of its head.  Output arrays hold 2m entries, row_off n+1;
 * fwd_pos (optional, m entries) receives the index of each arc's forward entry. */
int psn_csr_from_arcs(int32_t n, int64_t m, const int32_t *tail, const int32_t *head, const int32_t *cap,
                      const int32_t *cost, int64_t *row_off, int32_t *r_head, int32_t *r_cap,
                      int32_t *r_cost, int32_t *r_rev, int64_t *fwd_pos);

/* DIMACS minimum-cost-flow text ("p min", "n", "a" lines, 1-based ids), the input format of
 * the CPU solvers Firmament runs (/root/reference/LICENSE:18-26).  psn_dimacs_write emits
 * one arc per residual arc that has capacity; psn_dimacs_size reads the problem line;
 * psn_dimacs_read fills an arc list of that size.  Lower bounds are folded away: the
 * bound is subtracted from the capacity, moved into the supplies, and its cost is returned
 * in *base_cost (to be added to the optimum of the reduced problem). */
int psn_dimacs_write(const char *path, int32_t n, int64_t m2, const int64_t *row_off, const int32_t *head,
                     const int32_t *cap, const int32_t *cost, const int64_t *supply, const char *comment);
int psn_dimacs_size(const char *path, int32_t *n, int64_t *m);
int psn_dimacs_read(const char *path, int32_t n, int64_t m, int32_t *tail, int32_t *head, int32_t *cap,
                    int32_t *cost, int64_t *supply, int64_t *base_cost);
/* The same, also returning the lower bound each arc had in the file (low[m], file order).  Lines
 * longer than 510 characters are refused (PSN_EINVAL) by every reader here. */
int psn_dimacs_read_lower(const char *path, int32_t n, int64_t m, int32_t *tail, int32_t *head, int32_t *cap,
                          int32_t *cost, int64_t *supply, int64_t *base_cost, int32_t *low);

/* Change records: a second file that edits a problem already read, so that the resident network
 * can be edited in place and the solve warm-started: a later scheduling round described as changes
 * to the previous one (for the same CPU solvers, /root/reference/LICENSE:18-26).  Lines:
 *   x <tail> <head> 0 <cap> <cost> [<k>]   new capacity and cost of the k-th (default first) arc
 *                                          tail -> head, counting parallel arcs in file order
 *   n <vertex> <supply>                    new supply
 * Arcs and vertices are neither added nor removed (capacity 0 takes an arc out); any other
 * record is PSN_EINVAL.  psn_dimacs_changes_size counts the records, psn_dimacs_read_changes
 * resolves them against the arc list psn_dimacs_read produced: arc[i] indexes that list.  Apply
 * with psn_update_caps / psn_update_costs (arc positions from psn_csr_from_arcs' fwd_pos) and
 * psn_update_supplies. */
int psn_dimacs_changes_size(const char *path, int64_t *n_arc_changes, int64_t *n_supply_changes);
int psn_dimacs_read_changes(const char *path, int32_t n, int64_t m, const int32_t *tail, const int32_t *head,
                            int64_t *arc, int32_t *new_cap, int32_t *new_cost, int32_t *vertex, int64_t *new_supply);
/* The same with the lower bounds psn_dimacs_read_lower returned: an `x` record that targets an arc
 * whose lower bound in the problem file was not 0 is PSN_EINVAL (the reader folded that bound
 * into the capacity, the supplies and the base cost; a record that only carries a new upper bound
 * and cost cannot be applied to it).  psn_dimacs_read_changes cannot tell and applies the record
 * as if the bound had been 0: use it only for problems without lower bounds. */
int psn_dimacs_read_changes_lower(const char *path, int32_t n, int64_t m, const int32_t *tail, const int32_t *head,
                                  const int32_t *low, int64_t *arc, int32_t *new_cap, int32_t *new_cost,
                                  int32_t *vertex, int64_t *new_supply);

/* FNV-1a-64 over a byte range, chained through h (start with 0xCBF29CE484222325). */
uint64_t psn_fnv1a64(const void *data, int64_t nbytes, uint64_t h);

#ifdef __cplusplus
}
#endif
#endif /* PSN_BUILDER_H_ */
