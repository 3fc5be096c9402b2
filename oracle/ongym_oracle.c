/*
 * ongym_oracle.c — CPU restatement of the reference's QRMSA per-request hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (libongym_hip.so and the Python package) never imports,
 * links or executes anything under oracle/.
 *
 * Parity status: PINNED.  The reference ships no golden vectors for this path, so the oracle is pinned
 * against outputs of the reference itself, generated in the build container by tests/golden/make_golden.py (the
 * reference compiled unmodified under /tmp) and committed as data under tests/golden/: GN known-answer tests,
 * candidate-scan cases, action codec cases, full trajectories (first fit, load balancing, highest SNR, with
 * measure_disruptions and with defragmentation), observation vectors + action masks, and the per-state decisions of
 * the remaining heuristics (tests/test_oracle_golden.py).
 *
 * It deliberately keeps the reference's data structures and operation order — int32 slot grid with 1 = free, per-link
 * running-service lists in insertion order, a binary heap of departures, fp64 GN model looped span by span — so that
 * floating-point results agree to the last bits.  Each function cites the reference lines it follows (paths relative
 * to the reference repository root, optical_networking_gym/...).
 *
 * Plain C (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ongym.h"
#include "../include/ongym_traffic.h"

#define ORC_MAX_HOPS 64

typedef struct {
    int32_t id;         /* Service.service_id = episode_services_processed at creation (envs/qrmsa.pyx:1092) */
    int32_t src, dst;
    float arrival_time, holding_time, bit_rate; /* C floats in the reference (envs/qrmsa.pyx:35-37, 1068-1075) */
    int32_t path_id, slot, nslots, mod;
    double center_frequency, bandwidth, launch_power;
    double osnr, ase, nli;
    int accepted;
    int disrupted;      /* member of disrupted_services_list (envs/qrmsa.pyx:949-952) */
} orc_service;

typedef struct {
    double key;    /* heap key; re-pushed entries carry the float32-rounded key (envs/qrmsa.pyx:1114-1121) */
    int32_t id;
    int32_t svc;   /* index into pool */
} orc_event;

typedef struct orc_env {
    ongym_config cfg; /* pointers re-targeted to owned copies */
    int32_t *pair_paths, *path_hops, *path_links, *link_nspans, *mod_se;
    double *link_span_km, *link_alpha, *link_nf, *mod_thr, *bit_rates, *bit_rate_cum, *node_cum;
    double launch_power, margin, load, mean_iat;
    /* dynamic state */
    int32_t *grid;          /* [E*S], 1 = free: topology.graph["available_slots"] (envs/qrmsa.pyx:306-309) */
    orc_service *pool;      /* running services live here */
    int32_t *pool_free, n_pool_free, pool_cap;
    int32_t *run;           /* [E][cap] per-link running_services, insertion order (envs/qrmsa.pyx:1304-1305) */
    int32_t *run_cnt;       /* [E] */
    int32_t n_running;      /* len(topology.graph["running_services"]) */
    int32_t *glist;         /* topology.graph["running_services"] itself, insertion order (envs/qrmsa.pyx:1306, 1350) */
    orc_event *heap; int32_t n_heap;
    orc_service cur;        /* current_service */
    int new_service;        /* _new_service */
    double current_time;
    int max_mod_idx;
    /* counters (envs/qrmsa.pyx:364-371, 408-410) */
    int64_t services_processed, services_accepted, ep_processed, ep_accepted;
    double bit_rate_requested, bit_rate_provisioned, ep_bit_rate_requested, ep_bit_rate_provisioned;
    int64_t bl_reject, ep_mod_hist[8];
    double ep_osnr_sum; int64_t ep_services_listed; /* for mean_gsnr: topology.graph["services"] */
    int64_t episodes_completed;
    int64_t disrupted_services, ep_disrupted_services;   /* envs/qrmsa.pyx:315-316, 468-469 */
    int64_t ep_defrag_cycles, ep_reallocations;          /* envs/qrmsa.pyx:412-413, 438-439, 1546, 1635 */
    int64_t step_defrag_cycles, step_reallocations;      /* ... as the last step's info dict saw them (:1008-1009) */
    ongym_stats last; /* snapshot at last terminal step */
    int64_t total_steps, total_accepted, total_gn, total_terms, total_paths, total_hops, total_active_sum;
    /* request source */
    int mode;               /* 0 none, 1 rng, 2 trace */
    uint64_t key, req_index;
    const ongym_request *trace; int64_t trace_n, trace_pos;
    int flags;
    void *scratch_intf; int32_t *scratch_avail;   /* preallocated work buffers */
} orc_env;

static void *dup_mem(const void *p, size_t n) { void *q = malloc(n ? n : 1); if (p && n) memcpy(q, p, n); return q; }

orc_env *orc_create(const ongym_config *c, int replica) {
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->cfg = *c;
    int N = c->n_nodes, E = c->n_links, P = c->n_paths, K = c->k_paths, H = c->max_hops, M = c->n_mods;
    e->pair_paths = dup_mem(c->pair_paths, sizeof(int32_t) * N * N * K);
    e->path_hops = dup_mem(c->path_hops, sizeof(int32_t) * P);
    e->path_links = dup_mem(c->path_links, sizeof(int32_t) * P * H);
    e->link_nspans = dup_mem(c->link_nspans, sizeof(int32_t) * E);
    e->link_span_km = dup_mem(c->link_span_km, sizeof(double) * E);
    e->link_alpha = dup_mem(c->link_alpha, sizeof(double) * E);
    e->link_nf = dup_mem(c->link_nf, sizeof(double) * E);
    e->mod_se = dup_mem(c->mod_se, sizeof(int32_t) * M);
    e->mod_thr = dup_mem(c->mod_min_osnr, sizeof(double) * M);
    e->bit_rates = dup_mem(c->bit_rates, sizeof(double) * c->n_bit_rates);
    e->bit_rate_cum = dup_mem(c->bit_rate_cum, sizeof(double) * c->n_bit_rates);
    e->node_cum = dup_mem(c->node_cum, sizeof(double) * N);
    e->launch_power = c->replica_launch_power_w ? c->replica_launch_power_w[replica] : c->launch_power_w;
    e->margin = c->replica_margin ? c->replica_margin[replica] : c->margin;
    e->load = c->replica_load ? c->replica_load[replica] : c->load;
    /* set_load, envs/qrmsa.pyx:1124-1132 */
    e->mean_iat = 1 / (e->load / c->mean_holding_time);
    e->grid = (int32_t *)malloc(sizeof(int32_t) * E * c->n_slots);
    e->pool_cap = c->capacity;
    e->pool = (orc_service *)calloc(e->pool_cap, sizeof(orc_service));
    e->pool_free = (int32_t *)malloc(sizeof(int32_t) * e->pool_cap);
    e->run = (int32_t *)malloc(sizeof(int32_t) * E * e->pool_cap);
    e->run_cnt = (int32_t *)calloc(E, sizeof(int32_t));
    e->glist = (int32_t *)malloc(sizeof(int32_t) * e->pool_cap);
    e->heap = (orc_event *)malloc(sizeof(orc_event) * e->pool_cap);
    e->max_mod_idx = M - 1;
    e->current_time = 0.0;
    e->scratch_intf = malloc(32 * ((size_t)e->pool_cap + 1) * (size_t)H);
    e->scratch_avail = (int32_t *)malloc(sizeof(int32_t) * c->n_slots);
    return e;
}

void orc_destroy(orc_env *e) {
    if (!e) return;
    free(e->pair_paths); free(e->path_hops); free(e->path_links); free(e->link_nspans); free(e->link_span_km);
    free(e->link_alpha); free(e->link_nf); free(e->mod_se); free(e->mod_thr); free(e->bit_rates);
    free(e->bit_rate_cum); free(e->node_cum); free(e->grid); free(e->pool); free(e->pool_free); free(e->run);
    free(e->run_cnt); free(e->glist); free(e->heap); free(e->scratch_intf); free(e->scratch_avail); free(e);
}

void orc_seed(orc_env *e, uint64_t seed, uint64_t replica) {
    e->mode = 1; e->key = ongym_stream_key(seed, replica); e->req_index = 0;
}
void orc_set_trace(orc_env *e, const ongym_request *reqs, int64_t n) {
    e->mode = 2; e->trace = reqs; e->trace_n = n; e->trace_pos = 0;
}

/* ---- departures heap: heapq of (release_time, service_id, service) (envs/qrmsa.pyx:1327-1330) ------------------ */
static int ev_less(const orc_event *a, const orc_event *b) {
    return a->key < b->key || (a->key == b->key && a->id < b->id);
}
static void heap_push(orc_env *e, orc_event ev) {
    int i = e->n_heap++;
    e->heap[i] = ev;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!ev_less(&e->heap[i], &e->heap[p])) break;
        orc_event t = e->heap[i]; e->heap[i] = e->heap[p]; e->heap[p] = t; i = p;
    }
}
static orc_event heap_pop(orc_env *e) {
    orc_event top = e->heap[0];
    e->heap[0] = e->heap[--e->n_heap];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < e->n_heap && ev_less(&e->heap[l], &e->heap[m])) m = l;
        if (r < e->n_heap && ev_less(&e->heap[r], &e->heap[m])) m = r;
        if (m == i) break;
        orc_event t = e->heap[i]; e->heap[i] = e->heap[m]; e->heap[m] = t; i = m;
    }
    return top;
}

/* ---- _release_path (envs/qrmsa.pyx:1332-1350): frees [slot, slot+n+1) (numpy clamps at S) on every link -------- */
static void release_path(orc_env *e, int32_t si) {
    orc_service *s = &e->pool[si];
    int S = e->cfg.n_slots, H = e->cfg.max_hops;
    int hops = e->path_hops[s->path_id];
    for (int h = 0; h < hops; h++) {
        int l = e->path_links[s->path_id * H + h];
        int end = s->slot + s->nslots + 1; if (end > S) end = S;
        for (int j = s->slot; j < end; j++) e->grid[l * S + j] = 1;
        /* list.remove(service): first match, order of the rest preserved */
        int32_t *lst = &e->run[(size_t)l * e->pool_cap]; int n = e->run_cnt[l], k = 0;
        while (k < n && lst[k] != si) k++;
        for (; k + 1 < n; k++) lst[k] = lst[k + 1];
        e->run_cnt[l] = n - 1;
    }
    {   /* topology.graph["running_services"].remove(service), :1350 */
        int n = e->n_running, k = 0;
        while (k < n && e->glist[k] != si) k++;
        for (; k + 1 < n; k++) e->glist[k] = e->glist[k + 1];
    }
    e->n_running--;
    e->pool_free[e->n_pool_free++] = si;
}

/* ---- get_number_slots (envs/qrmsa.pyx:1198-1205), bands unset: ceil(bit_rate / (SE * channel_width)) ----------- */
int orc_number_slots(const orc_env *e, float bit_rate, int mod) {
    /* with `bands` the reference divides by the C band's width in Hz (quirk Q9): cfg.nslots_channel_width */
    const double width = e->cfg.nslots_channel_width > 0 ? e->cfg.nslots_channel_width : e->cfg.channel_width;
    double required = (double)bit_rate / ((double)e->mod_se[mod] * width);
    return (int)ceil(required);
}

static void defragment(orc_env *e, int num_services);

/* ---- _next_service (envs/qrmsa.pyx:1067-1122) ---------------------------------------------------------------- */
static int next_service(orc_env *e) {
    if (e->new_service) return 0;                                  /* :1077-1078 */
    float at, ht, br; int src, dst;
    if (e->mode == 1) {
        ongym_traffic_params tp = { e->mean_iat, e->cfg.mean_holding_time, e->node_cum, e->cfg.n_nodes,
            e->cfg.bit_rate_mode, e->bit_rates, e->bit_rate_cum, e->cfg.n_bit_rates, e->cfg.bit_rate_lo,
            e->cfg.bit_rate_hi };
        ongym_drawn_request r = ongym_draw_request(e->key, e->req_index++, e->current_time, &tp);
        at = r.arrival_time; ht = r.holding_time; br = r.bit_rate; src = r.source; dst = r.destination;
    } else if (e->mode == 2) {
        if (e->trace_pos >= e->trace_n) { e->flags |= ONGYM_F_NO_REQUEST; return -1; }
        const ongym_request *q = &e->trace[e->trace_pos++];
        at = q->arrival_time; ht = q->holding_time; br = q->bit_rate; src = q->source; dst = q->destination;
    } else return -1;
    e->current_time = at;                                          /* :1079-1081: cdef float at -> double */
    memset(&e->cur, 0, sizeof(e->cur));
    e->cur.id = (int32_t)e->ep_processed;                          /* :1092 */
    e->cur.src = src; e->cur.dst = dst; e->cur.arrival_time = at; e->cur.holding_time = ht; e->cur.bit_rate = br;
    e->cur.path_id = -1; e->cur.slot = 0; e->cur.mod = -1;
    e->new_service = 1;
    e->services_processed += 1; e->ep_processed += 1;              /* :1104-1105 */
    e->bit_rate_requested += e->cur.bit_rate; e->ep_bit_rate_requested += e->cur.bit_rate; /* :1107-1108 */
    while (e->n_heap > 0) {                                        /* :1113-1122 */
        orc_event ev = heap_pop(e);
        float time = (float)ev.key;                                /* `cdef float time` */
        if (time <= e->current_time) {
            release_path(e, ev.svc);
            if (e->cfg.defragmentation &&                          /* :1117-1119 */
                (e->cfg.n_defrag_services == 0 || e->ep_processed % e->cfg.n_defrag_services == 0))
                defragment(e, e->cfg.n_defrag_services);
        } else {
            ev.key = time; heap_push(e, ev);
            break;
        }
    }
    return 0;
}

/* ---- reset (envs/qrmsa.pyx:427-504) --------------------------------------------------------------------------- */
int orc_reset(orc_env *e) {
    int E = e->cfg.n_links, S = e->cfg.n_slots;
    e->ep_bit_rate_requested = 0.0; e->ep_bit_rate_provisioned = 0.0;
    e->ep_processed = 0; e->ep_accepted = 0;
    e->n_heap = 0; e->bl_reject = 0; e->max_mod_idx = e->cfg.n_mods - 1;
    memset(e->ep_mod_hist, 0, sizeof(e->ep_mod_hist));
    e->bit_rate_requested = 0.0; e->bit_rate_provisioned = 0.0;   /* :466-467 */
    e->ep_osnr_sum = 0.0; e->ep_services_listed = 0;              /* topology.graph["services"] = [] */
    e->disrupted_services = 0; e->ep_disrupted_services = 0;      /* :432, 468-469 */
    e->ep_defrag_cycles = 0; e->ep_reallocations = 0;             /* :438-439 */
    e->n_running = 0; memset(e->run_cnt, 0, sizeof(int32_t) * E);
    e->n_pool_free = 0;
    for (int i = e->pool_cap - 1; i >= 0; i--) e->pool_free[e->n_pool_free++] = i;
    for (int i = 0; i < E * S; i++) e->grid[i] = 1;               /* :481-484 */
    e->new_service = 0;                                            /* :499-500 */
    return next_service(e);
}

/* reset(options={"only_episode_counters": True}) (envs/qrmsa.pyx:427-464): the first part of reset() only; note
 * `self._events = []` — the departure heap is dropped, so the services running now are never released. */
void orc_reset_counters(orc_env *e) {
    e->ep_bit_rate_requested = 0.0; e->ep_bit_rate_provisioned = 0.0;
    e->ep_processed = 0; e->ep_accepted = 0;
    e->ep_disrupted_services = 0;
    e->n_heap = 0; e->bl_reject = 0; e->max_mod_idx = e->cfg.n_mods - 1;
    e->ep_defrag_cycles = 0; e->ep_reallocations = 0;
    memset(e->ep_mod_hist, 0, sizeof(e->ep_mod_hist));
}

/* ---- get_available_slots (envs/qrmsa.pyx:1482-1512): product of the path's link rows --------------------------- */
void orc_available(const orc_env *e, int path_id, int32_t *out) {
    int S = e->cfg.n_slots, H = e->cfg.max_hops, hops = e->path_hops[path_id];
    const int32_t *r0 = &e->grid[e->path_links[path_id * H] * S];
    for (int j = 0; j < S; j++) out[j] = r0[j];
    for (int h = 1; h < hops; h++) {
        const int32_t *r = &e->grid[e->path_links[path_id * H + h] * S];
        for (int j = 0; j < S; j++) out[j] *= r[j];
    }
}

/* ---- rle (utils.pyx:44-58) + _get_candidates (envs/qrmsa.pyx:515-541) ------------------------------------------ */
/* Walks the runs of `row` (the numpy rle returns (positions, values, lengths)); a free run [start, start+len) yields
 * starts start..start+len-n when it touches total_slots, else start..start+len-(n+1).  Returns the count; writes up
 * to max_out starts (ascending). */
int orc_candidates(const int32_t *row, int total_slots, int n, int32_t *out, int max_out) {
    int cnt = 0, i = 0;
    while (i < total_slots) {
        int start = i, val = row[i];
        while (i < total_slots && row[i] == val) i++;
        int length = i - start;
        if (val == 1) {
            int need = (start + length == total_slots) ? n : n + 1;
            if (length >= need)
                for (int c = start; c <= start + length - need; c++) { if (cnt < max_out) out[cnt] = c; cnt++; }
        }
    }
    return cnt;
}

/* ---- is_path_free (envs/qrmsa.pyx:1248-1264) ------------------------------------------------------------------ */
int orc_is_path_free(const orc_env *e, int path_id, int slot, int n) {
    int S = e->cfg.n_slots, H = e->cfg.max_hops;
    int end = slot + n;
    if (end > S) return 0;
    if (end < S) end += 1;
    for (int h = 0; h < e->path_hops[path_id]; h++) {
        const int32_t *r = &e->grid[e->path_links[path_id * H + h] * S];
        for (int j = slot; j < end; j++) if (r[j] == 0) return 0;
    }
    return 1;
}

/* ---- calculate_osnr (core/osnr.pyx:21-142), literal: link loop, span loop, interferer loop, fp64 --------------- */
static const double PHI_MOD[6] = {1.0, 1.0, 2.0 / 3.0, 17.0 / 25.0, 69.0 / 100.0, 13.0 / 21.0}; /* core/osnr.pyx:38-41 */

typedef struct { double fc, bw; int se; int id; } orc_intf;

static void gn_core(const orc_env *e, int path_id, double fc, double bw, double P, int self_id,
                    const orc_intf *const *lists, const int *counts, double out[3], int64_t *terms) {
    const double beta_2 = -21.3e-27, gamma = 1.3e-3, h_plank = 6.626e-34, pi = M_PI;
    double acc_gsnr = 0.0, acc_ase = 0.0, acc_nli = 0.0;
    int H = e->cfg.max_hops, hops = e->path_hops[path_id];
    for (int h = 0; h < hops; h++) {
        int l = e->path_links[path_id * H + h];
        double alpha = e->link_alpha[l], L = e->link_span_km[l], nf = e->link_nf[l];
        for (int sp = 0; sp < e->link_nspans[l]; sp++) {
            double l_eff_a = 1.0 / (2.0 * alpha);
            double l_eff = (1.0 - exp(-2.0 * alpha * L * 1e3)) / (2.0 * alpha);
            double sum_phi = asinh(pi * pi * fabs(beta_2) * (bw * bw) / (4.0 * alpha));
            for (int k = 0; k < counts[h]; k++) {
                const orc_intf *r = &lists[h][k];
                if (r->id == self_id) continue;                    /* core/osnr.pyx:65 */
                double phi = (asinh(pi * pi * fabs(beta_2) * l_eff_a * r->bw * (r->fc - fc + (r->bw / 2.0)))
                              - asinh(pi * pi * fabs(beta_2) * l_eff_a * r->bw * (r->fc - fc - (r->bw / 2.0))))
                             - (PHI_MOD[r->se - 1] * (r->bw / fabs(r->fc - fc)) * (5.0 / 3.0) * (l_eff / (L * 1e3)));
                sum_phi += phi;
                if (terms && sp == 0) (*terms)++;   /* interferer x link terms (the span loop repeats them) */
            }
            double ratio = P / bw;
            double power_nli_span = (ratio * ratio * ratio) * (8.0 / (27.0 * pi * fabs(beta_2))) * (gamma * gamma)
                                    * l_eff * sum_phi * bw;
            double power_ase = bw * h_plank * fc * (exp(2.0 * alpha * L * 1e3) - 1.0) * nf;
            acc_gsnr += 1.0 / (P / (power_ase + power_nli_span));
            acc_ase += 1.0 / (P / power_ase);
            acc_nli += 1.0 / (P / power_nli_span);
        }
    }
    out[0] = 10.0 * log10(1.0 / acc_gsnr);
    out[1] = 10.0 * log10(1.0 / acc_ase);
    out[2] = 10.0 * log10(1.0 / acc_nli);
}

/* center frequency / bandwidth of an allocation (envs/qrmsa.pyx:901-906; heuristics/heuristics.py:947-952) */
static double center_freq(const orc_env *e, int slot, int n) {
    return e->cfg.frequency_start + (e->cfg.slot_bandwidth * slot) + (e->cfg.slot_bandwidth * (n / 2.0));
}

/* GN of a candidate against the CURRENT state; `count` = add to the evaluation / term statistics */
static void gn_state(orc_env *e, int path_id, int slot, int n, double out[3], int count) {
    int H = e->cfg.max_hops, hops = e->path_hops[path_id];
    const orc_intf *lists[ORC_MAX_HOPS]; int counts[ORC_MAX_HOPS];
    orc_intf *buf = (orc_intf *)e->scratch_intf;
    size_t off = 0;
    for (int h = 0; h < hops; h++) {
        int l = e->path_links[path_id * H + h];
        lists[h] = &buf[off]; counts[h] = e->run_cnt[l];
        for (int k = 0; k < e->run_cnt[l]; k++) {
            const orc_service *s = &e->pool[e->run[(size_t)l * e->pool_cap + k]];
            buf[off].fc = s->center_frequency; buf[off].bw = s->bandwidth; buf[off].se = e->mod_se[s->mod];
            buf[off].id = s->id; off++;
        }
    }
    if (count) e->total_gn++;
    /* the candidate is current_service: running services with ITS service_id are skipped (core/osnr.pyx:65, quirk Q12).
     * Ids are unique among the running services of an episode, so this only ever matters after
     * reset(options={"only_episode_counters": True}) restarted the ids under services that keep running. */
    gn_core(e, path_id, center_freq(e, slot, n), e->cfg.slot_bandwidth * n, e->launch_power, e->cur.id, lists, counts, out,
            count ? &e->total_terms : 0);
}
void orc_gn(orc_env *e, int path_id, int slot, int n, double out[3]) { gn_state(e, path_id, slot, n, out, 0); }

/* GN of a RUNNING service against the current state, itself excluded by service_id (core/osnr.pyx:65) */
static void gn_running(orc_env *e, int32_t si, double out[3]) {
    const orc_service *y = &e->pool[si];
    int H = e->cfg.max_hops, hops = e->path_hops[y->path_id];
    const orc_intf *lists[ORC_MAX_HOPS]; int counts[ORC_MAX_HOPS];
    orc_intf *buf = (orc_intf *)e->scratch_intf;
    size_t off = 0;
    for (int h = 0; h < hops; h++) {
        int l = e->path_links[y->path_id * H + h];
        lists[h] = &buf[off]; counts[h] = e->run_cnt[l];
        for (int k = 0; k < e->run_cnt[l]; k++) {
            const orc_service *s = &e->pool[e->run[(size_t)l * e->pool_cap + k]];
            buf[off].fc = s->center_frequency; buf[off].bw = s->bandwidth; buf[off].se = e->mod_se[s->mod];
            buf[off].id = s->id; off++;
        }
    }
    gn_core(e, y->path_id, y->center_frequency, y->bandwidth, y->launch_power, y->id, lists, counts, out, 0);
}

/* ---- defragment (envs/qrmsa.pyx:1545-1639) ------------------------------------------------------------------------
 * Every running service, in the order of topology.graph["running_services"], is offered the candidate starts of its own
 * path for its own slot count (its current slots still count as occupied); the lowest start below its present one whose
 * GSNR — evaluated with the service moved there, itself skipped by service_id — is not below minimum_osnr (NO margin)
 * wins: old [slot, slot+n+1) freed (clamped at S), new [start, start+n(+1 unless it ends at S)) taken, the service goes to
 * the END of its links' running lists, Service.OSNR/ASE/NLI are overwritten. */
static void defragment(orc_env *e, int num_services) {
    int S = e->cfg.n_slots, H = e->cfg.max_hops;
    e->ep_defrag_cycles += 1;
    if (num_services == 0) num_services = 1000000;
    int moved = 0, n_active = e->n_running;
    int32_t *active = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_active + 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1));
    memcpy(active, e->glist, sizeof(int32_t) * (size_t)n_active);      /* list(...) copy, :1557 */
    for (int a = 0; a < n_active; a++) {
        orc_service *s = &e->pool[active[a]];
        if (moved >= num_services) break;
        int p = s->path_id, n = s->nslots, old_slot = s->slot;
        double old_fc = s->center_frequency, old_bw = s->bandwidth;
        orc_available(e, p, e->scratch_avail);
        int nc = orc_candidates(e->scratch_avail, S, n, cand, S + 1);
        for (int q = 0; q < nc; q++) {
            int start = cand[q];
            if (start >= s->slot) continue;
            int end = start + n;
            if (end < S) end += 1; else if (end > S) continue;
            s->slot = start;
            s->center_frequency = center_freq(e, start, n);
            s->bandwidth = e->cfg.slot_bandwidth * n;
            s->launch_power = e->launch_power;
            double o[3];
            gn_running(e, active[a], o);
            if (o[0] < e->mod_thr[s->mod]) {
                s->slot = old_slot; s->center_frequency = old_fc; s->bandwidth = old_bw;
                continue;
            }
            for (int h = 0; h < e->path_hops[p]; h++) {
                int l = e->path_links[p * H + h];
                int oe = old_slot + n + 1; if (oe > S) oe = S;
                for (int j = old_slot; j < oe; j++) e->grid[l * S + j] = 1;
                int32_t *lst = &e->run[(size_t)l * e->pool_cap]; int cnt = e->run_cnt[l], k = 0;
                while (k < cnt && lst[k] != active[a]) k++;
                for (; k + 1 < cnt; k++) lst[k] = lst[k + 1];
                e->run_cnt[l] = cnt - 1;
            }
            for (int h = 0; h < e->path_hops[p]; h++) {
                int l = e->path_links[p * H + h];
                for (int j = start; j < end; j++) e->grid[l * S + j] = 0;
                e->run[(size_t)l * e->pool_cap + e->run_cnt[l]++] = active[a];
            }
            e->ep_osnr_sum += o[0] - s->osnr;                      /* the object in topology.graph["services"] is updated */
            s->osnr = o[0]; s->ase = o[1]; s->nli = o[2];
            moved += 1; e->ep_reallocations += 1;
            break;
        }
    }
    free(active); free(cand);
}

/* GN with explicit per-link interferer lists (for the captured known-answer tests):
 * intf = flat (slot, n, se) triples, link h owns counts[h] consecutive triples. */
void orc_gn_lists(orc_env *e, int path_id, int slot, int n, const int32_t *counts_in, const int16_t *intf,
                  double out[3]) {
    int hops = e->path_hops[path_id];
    const orc_intf *lists[ORC_MAX_HOPS]; int counts[ORC_MAX_HOPS];
    size_t total = 0;
    for (int h = 0; h < hops; h++) total += counts_in[h];
    orc_intf *buf = (orc_intf *)malloc(sizeof(orc_intf) * (total + 1));
    size_t off = 0;
    for (int h = 0; h < hops; h++) {
        lists[h] = &buf[off]; counts[h] = counts_in[h];
        for (int k = 0; k < counts_in[h]; k++, off++) {
            buf[off].fc = center_freq(e, intf[off * 3], intf[off * 3 + 1]);
            buf[off].bw = e->cfg.slot_bandwidth * intf[off * 3 + 1];
            buf[off].se = intf[off * 3 + 2]; buf[off].id = 1000000 + (int)off;
        }
    }
    gn_core(e, path_id, center_freq(e, slot, n), e->cfg.slot_bandwidth * n, e->launch_power, -1, lists, counts, out, 0);
    free(buf);
}

/* ---- action codec (heuristics/heuristics.py:36-54; envs/qrmsa.pyx:801-834) ------------------------------------- */
/* modulations_to_consider = min(kwarg, len(modulations)) (envs/qrmsa.pyx:313) */
static int mods_consider(const orc_env *e) {
    int mc = e->cfg.n_mods_consider;
    return (mc <= 0 || mc > e->cfg.n_mods) ? e->cfg.n_mods : mc;
}
int orc_encode_action(const orc_env *e, int path_index, int mod_index, int slot) {
    int rel = e->max_mod_idx - mod_index;
    return path_index * mods_consider(e) * e->cfg.n_slots + rel * e->cfg.n_slots + slot;
}
void orc_decode_action(const orc_env *e, int action, int out[3]) {
    int Mc = mods_consider(e), S = e->cfg.n_slots, K = e->cfg.k_paths;
    int slot = action % S; action /= S;
    int r = action % Mc; action /= Mc;
    int route = action % K;
    int mod = (e->max_mod_idx > 1) ? e->max_mod_idx - r : (Mc - 1) - r;  /* :821-825 */
    out[0] = route; out[1] = mod; out[2] = slot;
}
int orc_reject_action(const orc_env *e) { return e->cfg.k_paths * mods_consider(e) * e->cfg.n_slots; }
int orc_max_modulation_idx(const orc_env *e) { return e->max_mod_idx; }

/* ---- heuristic_shortest_available_path_first_fit_best_modulation (heuristics/heuristics.py:923-966) ------------ */
int orc_policy_first_fit(orc_env *e, int *blocked_resources, int *blocked_osnr) {
    int bres = 0, bosnr = 0;
    int S = e->cfg.n_slots, K = e->cfg.k_paths, N = e->cfg.n_nodes;
    int32_t *avail = e->scratch_avail;
    int action = orc_reject_action(e);
    for (int k = 0; k < K; k++) {
        int p = e->pair_paths[(e->cur.src * N + e->cur.dst) * K + k];
        if (p < 0) break;                                          /* fewer than k paths exist */
        e->total_paths++; e->total_hops += e->path_hops[p];
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m);
            if (req <= 0) continue;
            orc_available(e, p, avail);
            int32_t first;
            int cnt = orc_candidates(avail, S, req, &first, 1);
            if (cnt == 0) { bres = 1; continue; }
            double o[3];
            gn_state(e, p, first, req, o, 1);   /* statistics count the policy's evaluations only: the fused device
                                                   path does not repeat step()'s identical re-evaluation (:909) */
            double threshold = e->mod_thr[m] + e->margin;
            if (o[0] >= threshold) {
                action = orc_encode_action(e, k, m, first);
                *blocked_resources = 0; *blocked_osnr = 0;
                return action;
            }
            bosnr = 1;
            if (bres) bres = 0;
        }
    }
    *blocked_resources = bres; *blocked_osnr = bosnr;
    return action;
}

/* ---- load_balancing_best_modulation (heuristics/heuristics.py:547-627) ------------------------------------------ */
int orc_policy_load_balancing(orc_env *e, int *blocked_resources, int *blocked_osnr) {
    int any_res = 0, any_osnr = 0, have = 0;
    int S = e->cfg.n_slots, K = e->cfg.k_paths, N = e->cfg.n_nodes;
    int32_t *avail = e->scratch_avail;
    int solution = orc_reject_action(e);
    double lowest_load = INFINITY;
    for (int k = 0; k < K; k++) {
        int p = e->pair_paths[(e->cur.src * N + e->cur.dst) * K + k];
        if (p < 0) break;
        e->total_paths++; e->total_hops += e->path_hops[p];
        orc_available(e, p, avail);
        int busy = 0;
        for (int j = 0; j < S; j++) busy += (avail[j] == 0);
        double current_load = (double)busy / (double)e->path_hops[p];   /* np.sum(avail == 0) / len(path.links) */
        if (current_load >= lowest_load) continue;
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m);
            if (req <= 0) continue;
            int32_t first;
            int cnt = orc_candidates(avail, S, req, &first, 1);
            if (cnt == 0) { any_res = 1; continue; }
            double o[3];
            gn_state(e, p, first, req, o, 1);
            if (o[0] >= e->mod_thr[m] + e->margin) {
                lowest_load = current_load;
                solution = orc_encode_action(e, k, m, first);
                have = 1;
                break;
            }
            any_osnr = 1;
        }
    }
    if (have) { *blocked_resources = 0; *blocked_osnr = 0; return solution; }
    if (any_osnr) any_res = 0;
    *blocked_resources = any_res; *blocked_osnr = any_osnr;
    return solution;
}

/* ---- heuristic_highest_snr (heuristics/heuristics.py:272-328): every valid start of every (path, modulation) --------- */
int orc_policy_highest_snr(orc_env *e, int *blocked_resources, int *blocked_osnr) {
    int any_res = 0, any_osnr = 0, have = 0;
    int S = e->cfg.n_slots, K = e->cfg.k_paths, N = e->cfg.n_nodes;
    int32_t *avail = e->scratch_avail;
    int32_t *starts = (int32_t *)malloc(sizeof(int32_t) * (S + 1));
    int best_action = orc_reject_action(e);
    double best_osnr = -INFINITY;
    for (int k = 0; k < K; k++) {
        int p = e->pair_paths[(e->cur.src * N + e->cur.dst) * K + k];
        if (p < 0) break;
        e->total_paths++; e->total_hops += e->path_hops[p];
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m);
            if (req <= 0) continue;
            orc_available(e, p, avail);
            int cnt = orc_candidates(avail, S, req, starts, S + 1);
            if (cnt == 0) { any_res = 1; continue; }
            for (int i = 0; i < cnt; i++) {
                double o[3];
                gn_state(e, p, starts[i], req, o, 1);
                if (o[0] >= e->mod_thr[m] + e->margin) {
                    if (o[0] > best_osnr || (o[0] == best_osnr && !have)) {
                        best_osnr = o[0]; best_action = orc_encode_action(e, k, m, starts[i]); have = 1;
                    }
                } else any_osnr = 1;
            }
        }
    }
    free(starts);
    if (have) { *blocked_resources = 0; *blocked_osnr = 0; return best_action; }
    if (any_osnr) any_res = 0;
    *blocked_resources = any_res; *blocked_osnr = any_osnr;
    return best_action;
}

/* ---- the cheaper remaining policies of heuristics/heuristics.py ------------------------------------------------------ */
static int path_of(const orc_env *e, int k) {
    return e->pair_paths[(e->cur.src * e->cfg.n_nodes + e->cur.dst) * e->cfg.k_paths + k];
}
static int passes(orc_env *e, int p, int slot, int n, int m) {
    double o[3];
    gn_state(e, p, slot, n, o, 1);
    return o[0] >= e->mod_thr[m] + e->margin;
}
/* run-length view of a row (utils.rle, utils.pyx:44-58): starts/lengths of the FREE runs, left to right */
static int free_runs(const int32_t *row, int S, int32_t *start, int32_t *len) {
    int n = 0;
    for (int j = 0; j < S;) {
        if (row[j] == 0) { j++; continue; }
        int b = j;
        while (j < S && row[j] != 0) j++;
        start[n] = b; len[n] = j - b; n++;
    }
    return n;
}
static int largest_free_run(const int32_t *row, int S) {             /* _get_largest_contiguous_block, :751-763 */
    int best = 0, cur = 0;
    for (int j = 0; j < S; j++) { cur = row[j] ? cur + 1 : 0; if (cur > best) best = cur; }
    return best;
}

/* shortest_available_path_lowest_spectrum_best_modulation (:431-490): the first-fit walk, other flag rule */
static int orc_policy_lowest_spectrum(orc_env *e, int *bres_out, int *bosnr_out) {
    int bres = 0, bosnr = 0, S = e->cfg.n_slots;
    int32_t *avail = e->scratch_avail;
    for (int k = 0; k < e->cfg.k_paths; k++) {
        int p = path_of(e, k);
        if (p < 0) break;
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m);
            if (req <= 0) continue;
            orc_available(e, p, avail);
            int32_t first;
            if (orc_candidates(avail, S, req, &first, 1) == 0) { bres = 1; continue; }
            if (passes(e, p, first, req, m)) { *bres_out = 0; *bosnr_out = 0; return orc_encode_action(e, k, m, first); }
            bosnr = 1;
        }
    }
    if (bosnr) bres = 0;
    *bres_out = bres; *bosnr_out = bosnr;
    return orc_reject_action(e);
}

/* heuristic_load_balancing_first_fit (:202-269): routes by (occupied fraction, index), then first fit */
static int orc_policy_lb_first_fit(orc_env *e, int *bres_out, int *bosnr_out) {
    int S = e->cfg.n_slots, K = e->cfg.k_paths;
    int32_t *avail = e->scratch_avail;
    int order[64], busy[64], nk = 0;
    for (int k = 0; k < K && k < 64; k++) {
        int p = path_of(e, k);
        if (p < 0) break;
        orc_available(e, p, avail);
        busy[k] = 0;
        for (int j = 0; j < S; j++) busy[k] += (avail[j] == 0);
        order[nk++] = k;
    }
    for (int i = 1; i < nk; i++)                                   /* stable insertion sort: (load, index) ascending */
        for (int j = i; j > 0 && busy[order[j]] < busy[order[j - 1]]; j--) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    for (int i = 0; i < nk; i++) {
        int k = order[i], p = path_of(e, k);
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m);
            if (req <= 0) continue;
            orc_available(e, p, avail);
            int32_t first;
            if (orc_candidates(avail, S, req, &first, 1) == 0) continue;
            if (passes(e, p, first, req, m)) { *bres_out = 0; *bosnr_out = 0; return orc_encode_action(e, k, m, first); }
        }
    }
    *bres_out = 1; *bosnr_out = 0;
    return orc_reject_action(e);
}

/* best_modulation_load_balancing (:491-545): ALL modulations outer, routes inner; first free run of >= slots+1 */
static int orc_policy_best_mod_lb(orc_env *e, int *bres_out, int *bosnr_out) {
    int S = e->cfg.n_slots;
    int32_t *avail = e->scratch_avail;
    int32_t *rs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1)), *rl = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1));
    int action = orc_reject_action(e);
    *bres_out = 0; *bosnr_out = 0;
    for (int m = e->cfg.n_mods - 1; m >= 0 && action == orc_reject_action(e); m--) {
        int req = orc_number_slots(e, e->cur.bit_rate, m);
        for (int k = 0; k < e->cfg.k_paths; k++) {
            int p = path_of(e, k);
            if (p < 0) break;
            orc_available(e, p, avail);
            int nr = free_runs(avail, S, rs, rl), slot = -1;
            for (int i = 0; i < nr; i++) if (rl[i] >= req + 1) { slot = rs[i]; break; }
            if (slot < 0) continue;
            if (passes(e, p, slot, req, m)) { action = orc_encode_action(e, k, m, slot); break; }
        }
    }
    free(rs); free(rl);
    return action;
}

/* heuristic_mscl_simplified (:765-839) / heuristic_mscl_sequential_simplified (:841-921) */
static int orc_policy_mscl_simplified(orc_env *e, int sequential, int *bres_out, int *bosnr_out) {
    int bres = 0, bosnr = 0, S = e->cfg.n_slots;
    int32_t *avail = e->scratch_avail;
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
    int best_action = -1, best_score = -1;
    for (int k = 0; k < e->cfg.k_paths; k++) {
        int p = path_of(e, k);
        if (p < 0) break;
        if (sequential) { best_action = -1; best_score = -1; }
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m);
            if (req <= 0) continue;
            orc_available(e, p, avail);
            int32_t first;
            if (orc_candidates(avail, S, req, &first, 1) == 0) { bres = 1; continue; }
            if (passes(e, p, first, req, m)) {
                memcpy(tmp, avail, sizeof(int32_t) * (size_t)S);
                for (int j = first; j < first + req && j < S; j++) tmp[j] = 0;
                int score = largest_free_run(tmp, S);
                if (score > best_score) { best_score = score; best_action = orc_encode_action(e, k, m, first); }
            } else bosnr = 1;
        }
        if (sequential && best_action >= 0) break;
    }
    free(tmp);
    if (best_action >= 0) { *bres_out = 0; *bosnr_out = 0; return best_action; }
    *bres_out = bres; *bosnr_out = bosnr;
    return orc_reject_action(e);
}

/* heuristic_psr (:1019-1119), default coefficients: the route cost never excludes a route (best_cost stays infinite until
 * the search ends), so: first route, best modulation, lowest start - ALL starts of a modulation are tried - that passes */
static int orc_policy_psr(orc_env *e, int *bres_out, int *bosnr_out) {
    int S = e->cfg.n_slots;
    int32_t *avail = e->scratch_avail;
    int32_t *starts = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1));
    int action = orc_reject_action(e), found = 0;
    for (int k = 0; k < e->cfg.k_paths && !found; k++) {
        int p = path_of(e, k);
        if (p < 0) break;
        for (int m = e->max_mod_idx; m >= 0 && !found; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m);
            if (req <= 0) continue;
            orc_available(e, p, avail);
            int cnt = orc_candidates(avail, S, req, starts, S + 1);
            for (int i = 0; i < cnt; i++)
                if (passes(e, p, starts[i], req, m)) { action = orc_encode_action(e, k, m, starts[i]); found = 1; break; }
        }
    }
    free(starts);
    *bres_out = found ? 0 : 1; *bosnr_out = 0;
    return action;
}

/* heuristic_exact_fit (:1121-1227): first free run of exactly the needed length, else the smallest that is long enough */
static int orc_policy_exact_fit(orc_env *e, int *bres_out, int *bosnr_out) {
    int bres = 0, bosnr = 0, S = e->cfg.n_slots;
    int32_t *avail = e->scratch_avail;
    int32_t *rs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1)), *rl = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1));
    int action = -1;
    for (int k = 0; k < e->cfg.k_paths && action < 0; k++) {
        int p = path_of(e, k);
        if (p < 0) break;
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m);
            if (req <= 0) continue;
            orc_available(e, p, avail);
            int nr = free_runs(avail, S, rs, rl), slot = -1;
            if (nr == 0) { bres = 1; continue; }
            for (int i = 0; i < nr; i++) if (rl[i] == req) { slot = rs[i]; break; }
            if (slot < 0) {
                int best = 0x7fffffff;
                for (int i = 0; i < nr; i++) if (rl[i] >= req && rl[i] < best) { best = rl[i]; slot = rs[i]; }
                if (slot < 0) { bres = 1; continue; }
            }
            if (passes(e, p, slot, req, m)) { action = orc_encode_action(e, k, m, slot); break; }
            bosnr = 1;
        }
    }
    free(rs); free(rl);
    if (action >= 0) { *bres_out = 0; *bosnr_out = 0; return action; }
    if (bosnr) bres = 0;
    *bres_out = bres; *bosnr_out = bosnr;
    return orc_reject_action(e);
}

/* heuristic_lowest_fragmentation (:330-414), literally: every candidate start of every format of every route gets the score
 * 0.33*mean link entropy + 0.33*cuts + 0.34*rss of the route's link rows with the trial block PAINTED 1 (quirk: 1 is "free",
 * the block is free already, and utils.pyx:61-107 take the runs of value 0 for the "free blocks"); the request is sized
 * slots + 1 and the GN model evaluated at that width; strict `<` keeps the first of the lowest scores. */
static int orc_policy_lowest_fragmentation(orc_env *e, int *bres_out, int *bosnr_out) {
    int S = e->cfg.n_slots, H = e->cfg.max_hops, bres = 0, bosnr = 0;
    int32_t *avail = e->scratch_avail;
    int32_t *starts = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1));
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)H * (size_t)S);
    double best_score = INFINITY;
    int best_action = -1;
    for (int k = 0; k < e->cfg.k_paths; k++) {
        int p = path_of(e, k);
        if (p < 0) break;
        int hops = e->path_hops[p];
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int req = orc_number_slots(e, e->cur.bit_rate, m) + 1;                     /* :357 */
            if (req <= 0) continue;
            orc_available(e, p, avail);
            int cnt = orc_candidates(avail, S, req, starts, S + 1);
            if (cnt == 0) { bres = 1; continue; }
            for (int i = 0; i < cnt; i++) {
                int start = starts[i];
                for (int h = 0; h < hops; h++) {                                       /* :371-372 */
                    memcpy(tmp + (size_t)h * S, e->grid + (size_t)e->path_links[p * H + h] * S, sizeof(int32_t) * (size_t)S);
                    for (int j = start; j < start + req && j < S; j++) tmp[(size_t)h * S + j] = 1;
                }
                double ent_sum = 0.0, sum_sq = 0.0, sum_len = 0.0;
                long cuts = 0;
                for (int h = 0; h < hops; h++) {
                    const int32_t *row = tmp + (size_t)h * S;
                    double entropy = 0.0;                                              /* link_shannon_entropy_, utils.pyx:61-79 */
                    for (int j = 0; j < S;) {
                        if (row[j] != 0) { j++; continue; }
                        int b = j;
                        while (j < S && row[j] == 0) j++;
                        double pr = (double)(j - b) / (double)S;
                        volatile double term = pr * log(pr);                          /* two roundings, as in Python */
                        entropy += term;
                        cuts += 1;                                                     /* fragmentation_route_cuts, :82-90 */
                        sum_sq += (double)(j - b) * (double)(j - b);                   /* fragmentation_route_rss, :92-107 */
                        sum_len += (double)(j - b);
                    }
                    entropy = entropy != 0.0 ? -entropy : 0.0;
                    ent_sum = h == 0 ? entropy : ent_sum + entropy;                    /* sum(): 0 + e0 + e1 + ... */
                }
                double se = hops ? ent_sum / (double)hops : 0.0;
                double rss = sum_len == 0.0 ? 0.0 : sqrt(sum_sq) / sum_len;
                volatile double t1 = 0.33 * se, t2 = 0.33 * (double)cuts, t3 = 0.34 * rss;   /* no contraction */
                volatile double t12 = t1 + t2;
                double score = t12 + t3;
                if (!passes(e, p, start, req, m)) { bosnr = 1; continue; }
                if (score < best_score) { best_score = score; best_action = orc_encode_action(e, k, m, start); }
            }
        }
    }
    free(starts); free(tmp);
    if (best_action >= 0) { *bres_out = 0; *bosnr_out = 0; return best_action; }
    if (bosnr) bres = 0;
    *bres_out = bres; *bosnr_out = bosnr;
    return orc_reject_action(e);
}

/* heuristic_mscl (:647-749): among the candidates whose GSNR passes, the one that destroys the fewest placements
 * (_calculate_allocation_possibilities, :629-645: free runs of length >= w hold len - w + 1 placements) summed over the
 * configured bit rates at the candidate's format and over every route of the network - both directions of every pair,
 * as the reference's dict holds them - that shares a link with the candidate route, when [start, start+n) is blocked.
 * before - after = the free windows of width w that overlap the block = prefix-count difference (the reference recounts
 * both rows for every candidate; same integers). */
static int orc_policy_mscl(orc_env *e, int *bres_out, int *bosnr_out) {
    int S = e->cfg.n_slots, H = e->cfg.max_hops, N = e->cfg.n_nodes, K = e->cfg.k_paths, NP = e->cfg.n_paths;
    int bres = 0, bosnr = 0;
    int32_t *avail = e->scratch_avail;
    int32_t *starts = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1));
    int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)NP * (size_t)S);      /* get_available_slots of every route */
    int64_t *acc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(S + 1));
    for (int q = 0; q < NP; q++) orc_available(e, q, rows + (size_t)q * S);
    int64_t best_loss = INT64_MAX;
    int best_action = -1;
    for (int k = 0; k < K; k++) {
        int p = path_of(e, k);
        if (p < 0) break;
        for (int m = e->max_mod_idx; m >= 0; m--) {
            int n = orc_number_slots(e, e->cur.bit_rate, m);
            if (n <= 0) continue;
            orc_available(e, p, avail);
            int cnt = orc_candidates(avail, S, n, starts, S + 1);
            if (cnt == 0) { bres = 1; continue; }
            int64_t *loss = (int64_t *)calloc((size_t)cnt, sizeof(int64_t));
            int have_loss = 0;
            for (int i = 0; i < cnt; i++) {
                int start = starts[i];
                if (!passes(e, p, start, n, m)) { bosnr = 1; continue; }
                if (!have_loss) {
                    have_loss = 1;
                    for (int b = 0; b < e->cfg.n_bit_rates; b++) {
                        int w = orc_number_slots(e, (float)e->bit_rates[b], m);
                        if (w <= 0) continue;
                        memset(acc, 0, sizeof(int64_t) * (size_t)(S + 1));
                        for (int a = 0; a < N; a++) for (int d = 0; d < N; d++) for (int kk = 0; kk < K; kk++) {
                            int q = e->pair_paths[(a * N + d) * K + kk];
                            if (q < 0) continue;
                            int share = 0;
                            for (int h = 0; h < e->path_hops[p] && !share; h++)
                                for (int g = 0; g < e->path_hops[q] && !share; g++)
                                    share = e->path_links[p * H + h] == e->path_links[q * H + g];
                            if (!share) continue;
                            const int32_t *row = rows + (size_t)q * S;
                            int64_t c = 0;
                            int run = 0;                    /* window [t, t+w) free <=> the free run ending at t+w-1 is >= w long */
                            for (int j = 0; j < S; j++) {   /* acc[i] += number of free windows starting before i */
                                run = row[j] ? run + 1 : 0;
                                int t = j - w + 1;
                                if (t >= 0) { if (run >= w) c++; acc[t + 1] += c; }
                            }
                            for (int t = S - w + 2; t <= S; t++) if (t >= 1) acc[t] += c;
                        }
                        for (int j = 0; j < cnt; j++) {
                            int hi = starts[j] + n < S ? starts[j] + n : S, lo = starts[j] - w + 1 > 0 ? starts[j] - w + 1 : 0;
                            loss[j] += acc[hi] - acc[lo];
                        }
                    }
                }
                if (loss[i] < best_loss) { best_loss = loss[i]; best_action = orc_encode_action(e, k, m, start); }
            }
            free(loss);
        }
    }
    free(starts); free(rows); free(acc);
    if (best_action >= 0) { *bres_out = 0; *bosnr_out = 0; return best_action; }
    *bres_out = bres; *bosnr_out = bosnr;
    return orc_reject_action(e);
}

int orc_policy(orc_env *e, int policy, int *bres, int *bosnr) {
    if (policy == ONGYM_POLICY_LOWEST_FRAGMENTATION) return orc_policy_lowest_fragmentation(e, bres, bosnr);
    if (policy == ONGYM_POLICY_MSCL) return orc_policy_mscl(e, bres, bosnr);
    if (policy == ONGYM_POLICY_LOAD_BALANCING) return orc_policy_load_balancing(e, bres, bosnr);
    if (policy == ONGYM_POLICY_HIGHEST_SNR) return orc_policy_highest_snr(e, bres, bosnr);
    if (policy == ONGYM_POLICY_LOWEST_SPECTRUM) return orc_policy_lowest_spectrum(e, bres, bosnr);
    if (policy == ONGYM_POLICY_LB_FIRST_FIT) return orc_policy_lb_first_fit(e, bres, bosnr);
    if (policy == ONGYM_POLICY_BEST_MOD_LB) return orc_policy_best_mod_lb(e, bres, bosnr);
    if (policy == ONGYM_POLICY_MSCL_SIMPLIFIED) return orc_policy_mscl_simplified(e, 0, bres, bosnr);
    if (policy == ONGYM_POLICY_MSCL_SEQUENTIAL) return orc_policy_mscl_simplified(e, 1, bres, bosnr);
    if (policy == ONGYM_POLICY_PSR) return orc_policy_psr(e, bres, bosnr);
    if (policy == ONGYM_POLICY_EXACT_FIT) return orc_policy_exact_fit(e, bres, bosnr);
    return orc_policy_first_fit(e, bres, bosnr);
}

/* ---- reward (envs/qrmsa.pyx:1266-1285): only the not-accepted branch returns a value (quirk Q1) --------------- */
static double reward(const orc_env *e) {
    double failed_ratio = (double)(e->ep_processed - e->ep_accepted) / (double)e->ep_processed;
    if (!e->cur.accepted) return -3.0 * (1.0 + failed_ratio);
    return 0.0;
}

/* ---- _provision_path + _add_release (envs/qrmsa.pyx:1288-1330) -------------------------------------------------- */
static int provision(orc_env *e, int path_id, int slot, int n) {
    int S = e->cfg.n_slots, H = e->cfg.max_hops;
    if (e->n_pool_free == 0) { e->flags |= ONGYM_F_OVERFLOW; return -1; }
    int end = slot + n;
    if (end < S) end += 1;
    int32_t si = e->pool_free[--e->n_pool_free];
    e->cur.path_id = path_id; e->cur.slot = slot; e->cur.nslots = n;
    e->cur.center_frequency = center_freq(e, slot, n);
    e->cur.bandwidth = e->cfg.slot_bandwidth * n;
    e->pool[si] = e->cur;
    for (int h = 0; h < e->path_hops[path_id]; h++) {
        int l = e->path_links[path_id * H + h];
        for (int j = slot; j < end; j++) e->grid[l * S + j] = 0;
        e->run[(size_t)l * e->pool_cap + e->run_cnt[l]++] = si;
    }
    e->glist[e->n_running++] = si;
    e->services_accepted += 1; e->ep_accepted += 1;
    e->bit_rate_provisioned += e->cur.bit_rate;
    e->ep_bit_rate_provisioned = (double)(int64_t)(e->ep_bit_rate_provisioned + e->cur.bit_rate); /* :1319-1321 */
    orc_event ev; ev.key = (double)(e->cur.arrival_time + e->cur.holding_time); /* float + float, :1329 */
    ev.id = e->cur.id; ev.svc = si;
    heap_push(e, ev);
    return 0;
}

static void snapshot_terminal(orc_env *e) {
    ongym_stats *s = &e->last;
    s->last_episode_processed = e->ep_processed; s->last_episode_accepted = e->ep_accepted;
    s->last_rejected = e->bl_reject;
    s->last_service_blocking_rate = e->services_processed > 0
        ? (double)(e->services_processed - e->services_accepted) / (double)e->services_processed : 0.0;
    s->last_episode_service_blocking_rate = e->ep_processed > 0
        ? (double)(e->ep_processed - e->ep_accepted) / (double)e->ep_processed : 0.0;
    s->last_bit_rate_blocking_rate = e->bit_rate_requested > 0
        ? (e->bit_rate_requested - e->bit_rate_provisioned) / e->bit_rate_requested : 0.0;
    s->last_episode_bit_rate_blocking_rate = e->ep_bit_rate_requested > 0
        ? (e->ep_bit_rate_requested - e->ep_bit_rate_provisioned) / e->ep_bit_rate_requested : 0.0;
    memcpy(s->last_modulation_hist, e->ep_mod_hist, sizeof(e->ep_mod_hist));
    s->last_mean_gsnr = e->ep_services_listed ? e->ep_osnr_sum / (double)e->ep_services_listed : 0.0;
    s->last_episode_disrupted = e->ep_disrupted_services;
    s->last_episode_defrag_cycles = e->ep_defrag_cycles; s->last_episode_service_reallocations = e->ep_reallocations;
}

/* ---- step (envs/qrmsa.pyx:838-1065), gen_observation=False, measure_disruptions=False, no CSV ------------------ */
int orc_step(orc_env *e, int action, ongym_step_rec *out) {
    ongym_step_rec r; memset(&r, 0, sizeof(r));
    r.action = action; r.route = -1; r.modulation = -1; r.slot = -1;
    int reject = orc_reject_action(e);
    double osnr = 0.0;
    if (action == reject) {
        e->cur.accepted = 0; e->bl_reject += 1;                   /* :861-865 */
    } else {
        int d[3]; orc_decode_action(e, action, d);
        int route = d[0], m = d[1], slot = d[2];
        int N = e->cfg.n_nodes, K = e->cfg.k_paths;
        int p = e->pair_paths[(e->cur.src * N + e->cur.dst) * K + route];
        int n = orc_number_slots(e, e->cur.bit_rate, m);
        double osnr_req = e->mod_thr[m] + e->margin;
        if (p < 0 || !orc_is_path_free(e, p, slot, n)) {          /* :886-897 (quirk Q5) */
            e->cur.accepted = 0;
            r.reward = reward(e); r.retry = 1; r.flags |= ONGYM_F_BLOCKED_RESOURCES;
            r.active = e->n_running;
            if (out) *out = r;
            return 0;
        }
        double o[3];
        orc_gn(e, p, slot, n, o);                                  /* :909 */
        r.route = (int16_t)route; r.slot = (int16_t)slot;
        if (o[0] >= osnr_req) {                                    /* :911-924 */
            e->cur.accepted = 1; e->cur.osnr = o[0]; e->cur.ase = o[1]; e->cur.nli = o[2]; e->cur.mod = m;
            e->cur.launch_power = e->launch_power;
            e->ep_mod_hist[m] += 1;
            if (provision(e, p, slot, n) < 0) { e->cur.accepted = 0; r.flags |= ONGYM_F_OVERFLOW; }
            else { r.modulation = (int16_t)m; r.nslots = (int16_t)n; r.osnr = o[0]; r.ase = o[1]; r.nli = o[2];
                   osnr = o[0]; e->total_accepted++; }
        } else {                                                   /* :925-929: raises ValueError */
            r.flags |= ONGYM_F_QOT_ERROR; r.osnr = o[0];
            if (out) *out = r;
            return ONGYM_E_STATE;
        }
    }
    /* measure_disruptions (envs/qrmsa.pyx:937-952): services on the new service's links (itself included, it is already in
     * the running lists) that are not yet in the disrupted list are re-evaluated against minimum_osnr (no margin) */
    if (e->cfg.measure_disruptions && e->cur.accepted) {
        int H = e->cfg.max_hops, p = e->cur.path_id;
        int32_t *todo = (int32_t *)malloc(sizeof(int32_t) * (size_t)(e->n_running + 1));
        int nt = 0;
        for (int h = 0; h < e->path_hops[p]; h++) {
            int l = e->path_links[p * H + h];
            for (int k = 0; k < e->run_cnt[l]; k++) {
                int32_t si = e->run[(size_t)l * e->pool_cap + k];
                int seen = e->pool[si].disrupted;
                for (int q = 0; q < nt && !seen; q++) seen = todo[q] == si;
                if (!seen) todo[nt++] = si;
            }
        }
        for (int q = 0; q < nt; q++) {
            double o[3];
            gn_running(e, todo[q], o);
            if (o[0] < e->mod_thr[e->pool[todo[q]].mod]) {
                e->pool[todo[q]].disrupted = 1; e->disrupted_services++; e->ep_disrupted_services++;
            }
        }
        free(todo);
    }
    r.accepted = (uint8_t)e->cur.accepted;
    if (!e->cur.accepted) { r.route = (action == reject) ? -1 : r.route; }
    r.reward = (action != reject) ? reward(e) : -6.0;              /* :992-995 */
    e->ep_osnr_sum += e->cur.accepted ? osnr : 0.0;               /* Service.OSNR is reset to 0 when not accepted :963 */
    e->ep_services_listed += 1;                                    /* :1053 */
    e->new_service = 0;                                            /* :1052 */
    e->total_steps++;
    /* info rates are computed BEFORE _next_service (:996-1050) */
    e->step_defrag_cycles = e->ep_defrag_cycles; e->step_reallocations = e->ep_reallocations;
    int will_terminate = (e->ep_processed + 1 == e->cfg.episode_length);
    if (will_terminate) snapshot_terminal(e);
    next_service(e);                                               /* :1054 */
    r.terminated = (uint8_t)(e->ep_processed == e->cfg.episode_length); /* :1056 */
    if (r.terminated) {
        e->episodes_completed++;
        /* graph_load.py:181-185 reads Service.OSNR AFTER the loop, i.e. after this step's _next_service: the departures
         * it processed may have run defragment(), which rewrites the OSNR of the services it moves */
        e->last.last_mean_gsnr = e->ep_services_listed ? e->ep_osnr_sum / (double)e->ep_services_listed : 0.0;
    }
    r.active = e->n_running;
    e->total_active_sum += e->n_running;
    if (out) *out = r;
    return 0;
}

void orc_stats(const orc_env *e, ongym_stats *s) {
    *s = e->last;
    s->services_processed = e->services_processed; s->services_accepted = e->services_accepted;
    s->episode_services_processed = e->ep_processed; s->episode_services_accepted = e->ep_accepted;
    s->bit_rate_requested = e->bit_rate_requested; s->bit_rate_provisioned = e->bit_rate_provisioned;
    s->episode_bit_rate_requested = e->ep_bit_rate_requested;
    s->episode_bit_rate_provisioned = e->ep_bit_rate_provisioned;
    s->rejected = e->bl_reject;
    memcpy(s->episode_modulation_hist, e->ep_mod_hist, sizeof(e->ep_mod_hist));
    s->episode_osnr_sum = e->ep_osnr_sum; s->episodes_completed = e->episodes_completed;
    s->disrupted_services = e->disrupted_services; s->episode_disrupted_services = e->ep_disrupted_services;
    s->episode_defrag_cycles = e->ep_defrag_cycles; s->episode_service_reallocations = e->ep_reallocations;
    s->step_defrag_cycles = e->step_defrag_cycles; s->step_service_reallocations = e->step_reallocations;
    s->total_steps = e->total_steps; s->total_accepted = e->total_accepted; s->total_gn_evals = e->total_gn;
    s->total_interferer_terms = e->total_terms; s->total_paths_tried = e->total_paths;
    s->total_path_hops = e->total_hops; s->total_active_sum = e->total_active_sum;
    s->total_gn_shortcuts = 0; s->current_time = e->current_time; s->active = e->n_running;
    s->max_modulation_idx = e->max_mod_idx;
    s->flags = e->flags;
}

void orc_grid(const orc_env *e, int32_t *out) {
    memcpy(out, e->grid, sizeof(int32_t) * e->cfg.n_links * e->cfg.n_slots);
}
void orc_request(const orc_env *e, ongym_request *q) {
    q->arrival_time = e->cur.arrival_time; q->holding_time = e->cur.holding_time; q->bit_rate = e->cur.bit_rate;
    q->source = (int16_t)e->cur.src; q->destination = (int16_t)e->cur.dst;
}
/* running services in global insertion order is not kept by the reference beyond a list; export unordered set */
int orc_services(const orc_env *e, ongym_service *out) {
    int n = 0;
    for (int i = 0; i < e->n_heap; i++) {
        const orc_service *s = &e->pool[e->heap[i].svc];
        out[n].path_id = s->path_id; out[n].slot = (int16_t)s->slot; out[n].nslots = (int16_t)s->nslots;
        out[n].modulation = (int16_t)s->mod; out[n].reserved = 0; out[n].release_time = (float)e->heap[i].key;
        out[n].service_id = e->cfg.defragmentation ? s->id : -1; out[n].pad_ = 0;   /* like the device: kept only then */
        out[n].osnr = e->cfg.defragmentation ? s->osnr : 0.0; n++;
    }
    return n;
}


/* ---- observation() + action mask (envs/qrmsa.pyx:583-781) and calculate_osnr_observation (core/osnr.pyx:259-369) ----
 * gen_observation=True path. obs: float32[1 + 2 + k + k*M*12], mask: uint8[k*M*S + 1].
 * path_len_norm[p] = (path length - min link length) / (max link length - min link length)  (:692-705, quirk Q10)
 * max_bit_rate = max(bit_rates) (:679). */
static double osnr_observation(const orc_env *e, int path_id, double bw, double fc, double P, double gsnr_th) {
    const double beta_2 = -21.3e-27, gamma = 1.3e-3, h_plank = 6.626e-34, pi = M_PI;
    double acc_gsnr = 0.0;
    int H = e->cfg.max_hops, hops = e->path_hops[path_id];
    for (int h = 0; h < hops; h++) {
        int l = e->path_links[path_id * H + h];
        double alpha = e->link_alpha[l], L = e->link_span_km[l], nf = e->link_nf[l];
        for (int sp = 0; sp < e->link_nspans[l]; sp++) {
            double l_eff_a = 1.0 / (2.0 * alpha);
            double l_eff = (1.0 - exp(-2.0 * alpha * L * 1e3)) / (2.0 * alpha);
            double sum_phi = asinh(pi * pi * fabs(beta_2) * (bw * bw) / (4.0 * alpha));
            for (int k = 0; k < e->run_cnt[l]; k++) {
                const orc_service *r = &e->pool[e->run[(size_t)l * e->pool_cap + k]];
                double rb = r->bandwidth, rf = r->center_frequency;
                double phi = (asinh(pi * pi * fabs(beta_2) * l_eff_a * rb * (rf - fc + (rb / 2.0)))
                              - asinh(pi * pi * fabs(beta_2) * l_eff_a * rb * (rf - fc - (rb / 2.0))))
                             - (PHI_MOD[e->mod_se[r->mod] - 1] * (rb / fabs(rf - fc)) * (5.0 / 3.0) * (l_eff / (L * 1e3)));
                sum_phi += phi;
            }
            double ratio = P / bw;
            double power_nli_span = (ratio * ratio * ratio) * (8.0 / (27.0 * pi * fabs(beta_2))) * (gamma * gamma)
                                    * l_eff * sum_phi * bw;
            double power_ase = bw * h_plank * fc * (exp(2.0 * alpha * L * 1e3) - 1.0) * nf;
            acc_gsnr += 1.0 / (P / (power_ase + power_nli_span));
        }
    }
    double gsnr = 10.0 * log10(1.0 / acc_gsnr);
    return nearbyint(((gsnr - gsnr_th) / fabs(gsnr_th)) * 1e10) / 1e10;   /* np.round(x, 10), core/osnr.pyx:368 */
}

/* get_max_modulation_index (envs/qrmsa.pyx:543-581): path-major, best modulation first, candidates ascending; the first
 * candidate whose calculate_osnr reaches minimum_osnr + margin fixes max_modulation_idx = max(its modulation index,
 * modulations_to_consider - 1); none: modulations_to_consider - 1. */
void orc_get_max_modulation_index(orc_env *e) {
    int S = e->cfg.n_slots, K = e->cfg.k_paths, M = e->cfg.n_mods, N = e->cfg.n_nodes, Mc = mods_consider(e);
    int32_t *avail = e->scratch_avail;
    int32_t *starts = (int32_t *)malloc(sizeof(int32_t) * (S + 1));
    for (int k = 0; k < K; k++) {
        int p = e->pair_paths[(e->cur.src * N + e->cur.dst) * K + k];
        if (p < 0) break;
        orc_available(e, p, avail);
        for (int m = M - 1; m >= 0; m--) {
            int n = orc_number_slots(e, e->cur.bit_rate, m);
            int cnt = orc_candidates(avail, S, n, starts, S + 1);
            for (int i = 0; i < cnt; i++) {
                double o[3];
                gn_state(e, p, starts[i], n, o, 0);
                if (o[0] >= e->mod_thr[m] + e->margin) {
                    e->max_mod_idx = m > Mc - 1 ? m : Mc - 1;
                    free(starts);
                    return;
                }
            }
        }
    }
    e->max_mod_idx = Mc - 1;
    free(starts);
}

void orc_observe(orc_env *e, const double *path_len_norm, double max_bit_rate, float *obs, uint8_t *mask) {
    int S = e->cfg.n_slots, K = e->cfg.k_paths, Mall = e->cfg.n_mods, N = e->cfg.n_nodes;
    const int M = mods_consider(e);                                     /* num_mod_to_consider */
    orc_get_max_modulation_index(e);                                    /* :680 */
    /* the window of formats the observation describes (:712-717) */
    const int start_index = e->max_mod_idx <= 1 ? 0 : (e->max_mod_idx - (M - 1) > 0 ? e->max_mod_idx - (M - 1) : 0);
    (void)Mall;
    int32_t *avail = e->scratch_avail;
    int32_t *starts = (int32_t *)malloc(sizeof(int32_t) * (S + 1));
    double *vals = (double *)malloc(sizeof(double) * (S + 1));
    int o = 0;
    obs[o++] = (float)(e->cur.bit_rate / max_bit_rate);                 /* :688 */
    obs[o++] = (float)((double)e->cur.src / (N - 1));                    /* :682-686 */
    obs[o++] = (float)((double)e->cur.dst / (N - 1));
    for (int k = 0; k < K; k++) {
        int p = e->pair_paths[(e->cur.src * N + e->cur.dst) * K + k];
        obs[o++] = p >= 0 ? (float)path_len_norm[p] : 0.0f;
    }
    memset(mask, 0, (size_t)K * M * S + 1);
    double Pw = e->launch_power;
    for (int k = 0; k < K; k++) {
        int p = e->pair_paths[(e->cur.src * N + e->cur.dst) * K + k];
        for (int mi = 0; mi < M; mi++) {
            float *f = &obs[o]; o += 12;
            for (int j = 0; j < 12; j++) f[j] = -1.0f;
            if (p < 0) continue;
            int m = start_index + M - 1 - mi;                            /* mod_list = reversed(modulations[start:start+M]) :716-717 */
            int n = orc_number_slots(e, e->cur.bit_rate, m);
            orc_available(e, p, avail);
            int cnt = orc_candidates(avail, S, n, starts, S + 1);
            double mean_s = 0, std_s = 0, max_s = 0;
            if (cnt > 0) {
                for (int i = 0; i < cnt; i++) { mean_s += starts[i]; if (starts[i] > max_s) max_s = starts[i]; }
                mean_s /= cnt;
                for (int i = 0; i < cnt; i++) std_s += (starts[i] - mean_s) * (starts[i] - mean_s);
                std_s = sqrt(std_s / cnt);
            }
            double best = 0.0, om = 0.0, ov = 0.0;
            double bw = n * e->cfg.channel_width * 1e9;
            for (int i = 0; i < cnt; i++) {
                double fc = e->cfg.frequency_start + (e->cfg.channel_width * 1e9 * starts[i])
                            + (e->cfg.channel_width * 1e9 * (n / 2.0));
                vals[i] = osnr_observation(e, p, bw, fc, Pw, e->mod_thr[m]);
                if (vals[i] > best) best = vals[i];
                om += vals[i];
                if (vals[i] >= 0) mask[((size_t)k * M + mi) * S + starts[i]] = 1;   /* :743-763 */
            }
            if (cnt > 0) {
                om /= cnt;
                for (int i = 0; i < cnt; i++) ov += (vals[i] - om) * (vals[i] - om);
                ov /= cnt;
            }
            double tot = 0; for (int j = 0; j < S; j++) tot += avail[j];
            /* block sizes :635-650 */
            int nb = 0, cur_len = 0; double bsum = 0;
            int32_t *blk = starts;   /* reuse */
            for (int j = 0; j < S; j++) {
                if (avail[j] == 1) cur_len++;
                else { if (cur_len > 0) blk[nb++] = cur_len; cur_len = 0; }
            }
            if (cur_len > 0) blk[nb++] = cur_len;
            double mb = 0.0, sb = 0.0;
            if (nb > 0) {
                for (int i = 0; i < nb; i++) bsum += blk[i];
                double bm = bsum / nb, bv = 0;
                for (int i = 0; i < nb; i++) bv += (blk[i] - bm) * (blk[i] - bm);
                mb = ((bm - 4.0) / 4.0) / 100.0; sb = sqrt(bv / nb) / 100.0;
            }
            f[0] = (float)((double)cnt / S);
            f[1] = (float)(mean_s / (S - 1)); f[2] = (float)(std_s / (S - 1));
            double adj = (n - 5.5) / 3.5; f[3] = (float)(adj > 0.0 ? adj : 0.0);
            f[4] = (float)(2.0 * (tot - 0.5 * S) / S);
            f[5] = (float)mb; f[6] = (float)sb;
            f[7] = (float)best; f[8] = (float)om; f[9] = (float)ov;
            f[10] = (float)(2.0 * ((tot / S) - 0.5));
            f[11] = (float)(max_s / (S - 1));
        }
    }
    mask[(size_t)K * M * S] = 1;                                       /* :766 */
    free(starts); free(vals);
}

/* ---- the JOCN loop (examples/JOCN_Benchmark_2024/graph_load.py:157-164) for one replica ------------------------ */
/* nsteps iterations of {policy, step}; auto-reset after a terminal step.  out: [nsteps] or NULL. */
int orc_run_policy(orc_env *e, int policy, int nsteps, ongym_step_rec *out) {
    for (int i = 0; i < nsteps; i++) {
        int bres, bosnr;
        int a = orc_policy(e, policy, &bres, &bosnr);
        ongym_step_rec r;
        int rc = orc_step(e, a, &r);
        if (rc == ONGYM_E_STATE && policy >= ONGYM_POLICY_LOWEST_FRAGMENTATION && (r.flags & ONGYM_F_QOT_ERROR)) {
            /* the reference raises ValueError here (qrmsa.pyx:925-929; lowest fragmentation asked for the GN model at
             * slots + 1): the fused loop of the product rejects the request and flags it - the checker does the same */
            rc = orc_step(e, orc_reject_action(e), &r);
            r.flags |= ONGYM_F_QOT_ERROR | ONGYM_F_BLOCKED_OSNR;
            bres = bosnr = 0;
        }
        if (rc) return rc;
        if (bres) r.flags |= ONGYM_F_BLOCKED_RESOURCES;
        if (bosnr) r.flags |= ONGYM_F_BLOCKED_OSNR;
        if (out) out[i] = r;
        if (r.terminated && e->cfg.auto_reset) orc_reset(e);
    }
    return 0;
}

int orc_run_first_fit(orc_env *e, int nsteps, ongym_step_rec *out) {
    return orc_run_policy(e, ONGYM_POLICY_FIRST_FIT, nsteps, out);
}

/* B independent replicas, OpenMP over replicas (the reference's own fan-out is one process per simulation,
 * graph_load.py:361-363).  Used as bench.py's cpu_baseline ("port").  Returns total steps executed. */
int64_t orc_batch_run_first_fit(orc_env **envs, int batch, int nsteps, int threads) {
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : total)
    for (int b = 0; b < batch; b++) {
        if (orc_run_first_fit(envs[b], nsteps, 0) == 0) total += nsteps;
    }
    return total;
}

/* The same for any policy id (bench.py --policy, the bench-shape parity tests of the lean policy kernels). */
int64_t orc_batch_run_policy(orc_env **envs, int batch, int policy, int nsteps, int threads) {
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : total)
    for (int b = 0; b < batch; b++) {
        if (orc_run_policy(envs[b], policy, nsteps, 0) == 0) total += nsteps;
    }
    return total;
}
