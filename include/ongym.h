/*
 * ongym.h — C ABI of the MI355X-native batched QRMSA environment (libongym_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of LEA-UFPA/optical-networking-gym: the per-request loop of
 * optical_networking_gym/envs/qrmsa.pyx (first-fit policy + step + traffic/departure bookkeeping) with the GN model of
 * optical_networking_gym/core/osnr.pyx.  The reference has no C interface of its own (its Cython modules only export
 * the CPython module init), so each entry point below names the reference Python/Cython interface it replaces.
 * Plain pointers and sizes only; no torch / numpy types.  Host code stays Python (ctypes), see INTEGRATION.md.
 *
 * Ownership : the library owns all device state; callers own every buffer they pass in.  Input tables given to
 *             ongym_create are copied.  Output buffers are host pointers, or device pointers when cfg.io_device = 1
 *             (e.g. torch.Tensor.data_ptr() of a PyTorch-ROCm tensor on the same device).
 * Errors    : every call returns 0 on success, <0 on error (ONGYM_E_*); ongym_last_error() gives text. Nothing throws.
 *             The reference's ValueError on a QoT-infeasible action (qrmsa.pyx:925-929) is the per-replica
 *             ONGYM_F_QOT_ERROR flag of ongym_step_rec.flags (the Python shim re-raises it in single-env mode).
 * Threading : one ongym_env = one HIP device + one HIP stream; not thread-safe; launches are asynchronous on that
 *             stream, results are complete after ongym_sync() (calls that copy to host buffers sync themselves).
 * Multi-GPU : one process per GPU, one ongym_env each; replicas are independent so there is no data-path collective.
 */
#ifndef ONGYM_H
#define ONGYM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ONGYM_ABI_VERSION 3

enum {
    ONGYM_OK = 0,
    ONGYM_E_ARG = -1,      /* bad argument / inconsistent tables */
    ONGYM_E_HIP = -2,      /* HIP runtime error (no device, launch failure, ...) */
    ONGYM_E_STATE = -3,    /* call not valid in the current state (e.g. no request source set) */
    ONGYM_E_CAPACITY = -4, /* a replica overflowed its service table (cfg.capacity too small) */
    ONGYM_E_LIMIT = -5     /* configuration exceeds a compile-time limit of the kernels */
};

/* policies fused on device; replaces optical_networking_gym/heuristics/heuristics.py:923-966 */
enum {
    ONGYM_POLICY_FIRST_FIT = 0,      /* heuristic_shortest_available_path_first_fit_best_modulation, heuristics.py:923-966 */
    ONGYM_POLICY_LOAD_BALANCING = 1, /* load_balancing_best_modulation, heuristics.py:547-627 (graph_load.py heuristic 4) */
    ONGYM_POLICY_HIGHEST_SNR = 2,    /* heuristic_highest_snr, heuristics.py:272-328 (graph_load.py heuristic 2) */
    /* the cheaper remaining policies of heuristics.py, one shared kernel instantiation (graph_launch_power.py 2,3,6,7,9): */
    ONGYM_POLICY_LOWEST_SPECTRUM = 3, /* shortest_available_path_lowest_spectrum_best_modulation, :431-490 */
    ONGYM_POLICY_LB_FIRST_FIT = 4,    /* heuristic_load_balancing_first_fit, :202-269 */
    ONGYM_POLICY_BEST_MOD_LB = 5,     /* best_modulation_load_balancing, :491-545 */
    ONGYM_POLICY_MSCL_SIMPLIFIED = 6, /* heuristic_mscl_simplified, :765-839 */
    ONGYM_POLICY_MSCL_SEQUENTIAL = 7, /* heuristic_mscl_sequential_simplified, :841-921 */
    ONGYM_POLICY_PSR = 8,             /* heuristic_psr with its default coefficients, :1019-1119 */
    ONGYM_POLICY_EXACT_FIT = 9,       /* heuristic_exact_fit, :1121-1227 (asks for no guard slot: the step may answer
                                         with the occupied-slots penalty, retry = 1) */
    /* the two policies that score every candidate (one more instantiation, csrc/ongym_scored.hpp): */
    ONGYM_POLICY_LOWEST_FRAGMENTATION = 10, /* heuristic_lowest_fragmentation, :330-414 (its request is sized slots + 1; if the
                                               step's own GSNR check at `slots` then fails - the reference raises ValueError,
                                               qrmsa.pyx:925-929 - the fused loop rejects the request with
                                               ONGYM_F_QOT_ERROR | ONGYM_F_BLOCKED_OSNR and goes on) */
    ONGYM_POLICY_MSCL = 11,                 /* heuristic_mscl, :647-749 (discrete bit rates only: the loss is summed over them) */
    ONGYM_POLICY_COUNT = 12
};

/* ongym_step_rec.flags */
enum {
    ONGYM_F_BLOCKED_RESOURCES = 1, /* 2nd element of the heuristic's return tuple, heuristics.py:966 */
    ONGYM_F_BLOCKED_OSNR = 2,      /* 3rd element */
    ONGYM_F_QOT_ERROR = 4,         /* action decoded to free slots whose GSNR < threshold+margin: qrmsa.pyx:925-929 */
    ONGYM_F_OVERFLOW = 8,          /* service table full: request was rejected artificially, results invalid */
    ONGYM_F_NO_REQUEST = 16        /* replay trace exhausted: step was a no-op */
};

/*
 * Static description of the network + traffic.  Replaces the arguments of QRMSAEnv.__init__ (qrmsa.pyx:206-237) and
 * the data that optical_networking_gym/topology.pyx:244-369 (get_topology) attaches to the graph, flattened:
 *   pair_paths[(src*n_nodes+dst)*k_paths + k] = path id or -1      (k_shortest_paths[src,dst][k], qrmsa.pyx:277)
 *   path_links[p*max_hops + h], h < path_hops[p]                    (Path.links, topology.pyx:72-95; link "index")
 *   link_*[e]                                                       (Link.spans: all spans of a link are equal,
 *                                                                    topology.pyx:288-299; alpha in 1/m, nf linear,
 *                                                                    Span, topology.pyx:11-34)
 *   mod_se / mod_min_osnr [m]                                       (Modulation, topology.pyx:53-70), m ascending
 */
typedef struct ongym_config {
    int32_t struct_size; /* = sizeof(ongym_config) */
    int32_t abi_version; /* = ONGYM_ABI_VERSION */
    int32_t n_nodes, n_links, n_paths, k_paths, max_hops, n_mods, n_slots;
    int32_t batch;          /* number of independent replicas B */
    int32_t capacity;       /* max simultaneously running services per replica (multiple of 64) */
    int32_t episode_length; /* qrmsa.pyx:210; an episode is episode_length-1 steps */
    int32_t auto_reset;     /* 1: a replica that terminates is reset inside the same launch (graph_load.py:157-158) */
    int32_t bit_rate_mode;  /* 0 = "discrete" (bit_rates/bit_rate_cum), 1 = "continuous" (randint(lo,hi)) */
    int32_t n_bit_rates;
    int32_t bit_rate_lo, bit_rate_hi;
    int32_t device;         /* HIP device ordinal */
    int32_t io_device;      /* 1: in/out buffers of step/set_requests calls are device pointers */
    int32_t measure_disruptions; /* qrmsa.pyx:224, 937-952: after every accept re-evaluate the GSNR of the services that share a
                                    link with the new one and count those that fall below their modulation's threshold */
    int32_t defragmentation;     /* qrmsa.pyx:233, 1117-1119: after a departure, try to move running services to lower slots */
    int32_t n_defrag_services;   /* qrmsa.pyx:234: 0 = after EVERY departure, no limit on moves; N > 0 = only when the
                                    episode's request count is a multiple of N, at most N moves */
    double frequency_start;       /* Hz,  qrmsa.pyx:221 */
    double slot_bandwidth;        /* Hz,  qrmsa.pyx:222 */
    double channel_width;         /* GHz, qrmsa.pyx:228 (get_number_slots, qrmsa.pyx:1198-1205) */
    double launch_power_w;        /* 10**((dBm-30)/10), qrmsa.pyx:288 */
    double margin;                /* dB, qrmsa.pyx:223 */
    double load;                  /* Erlang, qrmsa.pyx:211 */
    double mean_holding_time;     /* s, qrmsa.pyx:212 */
    const int32_t *pair_paths;    /* [n_nodes*n_nodes*k_paths] */
    const int32_t *path_hops;     /* [n_paths] */
    const int32_t *path_links;    /* [n_paths*max_hops] */
    const int32_t *link_nspans;   /* [n_links] */
    const double *link_span_km;   /* [n_links] */
    const double *link_alpha;     /* [n_links] 1/m */
    const double *link_nf;        /* [n_links] linear */
    const int32_t *mod_se;        /* [n_mods] spectral efficiency (1..6) */
    const double *mod_min_osnr;   /* [n_mods] dB */
    const double *bit_rates;      /* [n_bit_rates] Gb/s */
    const double *bit_rate_cum;   /* [n_bit_rates] cumulative probabilities, last = 1 */
    const double *node_cum;       /* [n_nodes] cumulative node request probabilities, last = 1 (qrmsa.pyx:278-286) */
    /* optional per-replica overrides (NULL = use the scalar above): the JOCN sweeps over launch power / load / margin
     * (graph_launch_power.py, graph_load.py, graph_margin.py) become a batch dimension */
    const double *replica_launch_power_w; /* [batch] */
    const double *replica_load;           /* [batch] */
    const double *replica_margin;         /* [batch] */
    /* observation() only (qrmsa.pyx:583-781): route lengths normalised by the min/max LINK length (:692-705) */
    const double *path_len_norm;          /* [n_paths] or NULL (ongym_observe then fails) */
    double max_bit_rate;                  /* max(bit_rates), qrmsa.pyx:679 */
    /* 1: keep Service.service_id (qrmsa.pyx:1092) per running service, as cfg.defragmentation does.  calculate_osnr skips
     * the running services whose id equals the evaluated service's (core/osnr.pyx:65, "quirk Q12"); ids are unique inside
     * an episode, so this only shows after ongym_reset_episode_counters restarted them under services that keep running.
     * Needed by ongym_reset_episode_counters; selects the id-tracking (slower) kernels. */
    int32_t track_service_ids;
    /* modulations_to_consider (qrmsa.pyx:313): 0 or >= n_mods = all of them.  Below n_mods the action space shrinks to
     * k_paths * n_mods_consider * n_slots + 1 and addresses the n_mods_consider formats at and below max_modulation_idx
     * (action codec qrmsa.pyx:801-834, heuristics.py:36-54); ongym_observe then first finds max_modulation_idx like
     * get_max_modulation_index (qrmsa.pyx:543-581) and reports that window (:712-717). */
    int32_t n_mods_consider;
    /* Width that get_number_slots divides by (qrmsa.pyx:1198-1205), in the unit of channel_width; 0 = channel_width itself.
     * The reference's `bands` argument (qrmsa.pyx:417-425; passed by graph_launch_power.py:102) makes get_number_slots use the
     * C band's width in Hz - every service then needs ONE slot (quirk Q9) - while observation() and step() keep computing
     * frequencies with channel_width (qrmsa.pyx:606-610, 678): the two widths are separate fields here. */
    double nslots_channel_width;
} ongym_config;

/* One service request; replaces the fields drawn in QRMSAEnv._next_service (qrmsa.pyx:1079-1101). */
typedef struct ongym_request {
    float arrival_time; /* absolute, already rounded to float32 like the reference's `cdef float at` */
    float holding_time;
    float bit_rate;
    int16_t source;     /* node index */
    int16_t destination;
} ongym_request;

/* Per-replica result of one step; replaces the (obs, reward, terminated, truncated, info) tuple of QRMSAEnv.step
 * (qrmsa.pyx:838-1065) for gen_observation=False, plus what the heuristic returned. 56 bytes. */
typedef struct ongym_step_rec {
    int32_t action;      /* action index applied (p*M*S + (max_mod-m)*S + slot, reject = k*M*S; heuristics.py:36-54) */
    int16_t route;       /* info["chosen_path_index"], -1 on reject */
    int16_t modulation;  /* absolute modulation index, -1 on reject */
    int16_t slot;        /* info["chosen_slot"], -1 on reject */
    int16_t nslots;
    uint8_t accepted;
    uint8_t terminated;
    uint8_t retry;       /* 1: slots were not free, request stays current (qrmsa.pyx:886-897) */
    uint8_t flags;       /* ONGYM_F_* */
    int32_t active;      /* running services after the step (len(topology.graph["running_services"])) */
    double osnr, ase, nli; /* dB, of the accepted service (info["osnr"]); 0 on reject */
    double reward;
} ongym_step_rec;

/* A running service as seen by the compatibility view (Service, qrmsa.pyx:29-53). */
typedef struct ongym_service {
    int32_t path_id;
    int16_t slot, nslots;
    int16_t modulation;
    int16_t reserved;   /* 1: member of disrupted_services_list (measure_disruptions) */
    float release_time; /* float32(arrival+holding), the heap key after rounding (qrmsa.pyx:1114-1115,1329) */
    int32_t service_id; /* Service.service_id (qrmsa.pyx:1092); kept only when cfg.defragmentation, else -1 */
    int32_t pad_;
    double osnr;        /* Service.OSNR as last written (provisioning or defragment(), qrmsa.pyx:1630); only when
                           cfg.defragmentation, else 0 */
} ongym_service;

/* One reallocation done by defragment() (qrmsa.pyx:1590-1635) during the LAST step of a replica: what the compatibility
 * view needs to update the moved Service object (initial_slot, center_frequency, OSNR, ASE, NLI). */
#define ONGYM_MOVE_LOG 64
typedef struct ongym_move {
    int32_t service_id;
    int32_t slot;        /* new initial_slot */
    double osnr, ase, nli; /* dB, as rewritten by defragment() */
} ongym_move;

/* Counters behind the info dict (qrmsa.pyx:996-1060) and the JOCN per-episode CSV row (graph_load.py:169-186). */
typedef struct ongym_stats {
    int64_t services_processed, services_accepted;                 /* never reset (quirk Q3) */
    int64_t episode_services_processed, episode_services_accepted; /* current episode */
    double bit_rate_requested, bit_rate_provisioned;               /* reset by reset(), qrmsa.pyx:466-467 */
    double episode_bit_rate_requested, episode_bit_rate_provisioned;
    int64_t rejected;                                              /* bl_reject of the current episode */
    int64_t episode_modulation_hist[8];
    double episode_osnr_sum;                                       /* sum of Service.OSNR over the episode's services */
    int64_t episodes_completed;
    int64_t disrupted_services, episode_disrupted_services;       /* qrmsa.pyx:948-952 (both zeroed by reset()) */
    /* defragment() (qrmsa.pyx:1545-1639): counters of the current episode, their value when the last step built its info
     * dict (i.e. before that step's _next_service ran, :1008-1009) */
    int64_t episode_defrag_cycles, episode_service_reallocations;
    int64_t step_defrag_cycles, step_service_reallocations;
    /* totals over all completed steps since create (for throughput accounting and the RCCL stats reduction) */
    int64_t total_steps, total_accepted, total_gn_evals, total_interferer_terms;
    int64_t total_paths_tried, total_path_hops; /* candidate paths whose slot rows were read, and their hops */
    int64_t total_gn_shortcuts;                 /* GN evaluations decided by the ASE-only bound (device only) */
    int64_t total_active_sum;                   /* sum over steps of the running-service count after the step */
    double current_time;
    int32_t active, flags;
    int32_t max_modulation_idx; /* QRMSAEnv.max_modulation_idx: n_mods-1 after reset (qrmsa.pyx:437), set by observation()'s
                                   get_max_modulation_index (:543-581, 680); the action codec is relative to it */
    int32_t reserved0_;
    /* snapshot taken at the last terminal step (what graph_load.py writes per episode). Kept LAST: the kernels hold only
     * the fields above in LDS and write these straight to memory. */
    int64_t last_episode_processed, last_episode_accepted, last_rejected;
    double last_service_blocking_rate, last_episode_service_blocking_rate;
    double last_bit_rate_blocking_rate, last_episode_bit_rate_blocking_rate;
    int64_t last_modulation_hist[8];
    double last_mean_gsnr;
    int64_t last_episode_disrupted;
    int64_t last_episode_defrag_cycles, last_episode_service_reallocations;   /* :1008-1009 at the terminal step */
} ongym_stats;

typedef struct ongym_env ongym_env;

/* QRMSAEnv.__init__ (qrmsa.pyx:206-425) for B replicas; does NOT generate the first request: call
 * ongym_seed or ongym_set_requests, then ongym_reset. */
int ongym_create(const ongym_config *cfg, ongym_env **out);
void ongym_destroy(ongym_env *env);

/* Request source A — device generator: replica r draws from the counter-based stream (seed, r) defined in
 * ongym_traffic.h.  Stands in for `self.rng = random.Random()` (qrmsa.pyx:241; unseeded in the reference, quirk Q2). */
int ongym_seed(ongym_env *env, uint64_t seed);
/* The same with a replica offset: local replica r draws stream (seed, replica_base + r).  A batch sharded over several
 * environments / GPUs (shard k owning the global replicas [base_k, base_k + batch_k)) then simulates exactly the
 * replicas of the unsharded batch — the fan-out of graph_load.py:361-363 (one simulation per Pool task) with
 * reproducible streams.  ongym_seed(env, seed) == ongym_seed_base(env, seed, 0). */
int ongym_seed_base(ongym_env *env, uint64_t seed, uint64_t replica_base);
/* Request source B — trace replay: reqs[r*n_per_replica + i] is the i-th request replica r will draw.
 * Used for parity against captured reference traces (each _next_service call consumes one entry). */
int ongym_set_requests(ongym_env *env, const ongym_request *reqs, int64_t n_per_replica);

/* QRMSAEnv.reset (qrmsa.pyx:427-504) on the replicas with mask[r] != 0 (NULL = all). */
int ongym_reset(ongym_env *env, const uint8_t *mask);

/* QRMSAEnv.reset(options={"only_episode_counters": True}) (qrmsa.pyx:427-464) on the replicas with mask[r] != 0 (NULL =
 * all): episode counters and histograms to zero, and — like the reference's `self._events = []` — the departure heap is
 * dropped, so the services running at that moment never leave.  Grid, running services, totals, clock and the current
 * request stay; no request is drawn.  Needs cfg.track_service_ids (ONGYM_E_STATE otherwise): service ids restart at 0 under
 * services that keep running, and calculate_osnr identifies "self" by service id (core/osnr.pyx:65). */
int ongym_reset_episode_counters(ongym_env *env, const uint8_t *mask);

/* nsteps iterations of `action,_,_ = heuristic(env); env.step(action)` (graph_load.py:161-163) fused on device.
 * out: [nsteps][batch] records or NULL. */
int ongym_step_policy(ongym_env *env, int32_t policy, int32_t nsteps, ongym_step_rec *out);
/* QRMSAEnv.step(action) (qrmsa.pyx:838-1065) with caller-supplied actions[batch]. out: [batch] or NULL. */
int ongym_step_actions(ongym_env *env, const int32_t *actions, ongym_step_rec *out);
/* The loop body of the reference's drivers (graph_load.py:157-164: `action = heuristic(env); env.step(action)`) for callers that
 * hold ONE environment and need every result on the host: ongym_step_actions(actions), then - on the same stream, with no host
 * round trip in between - fused policy `next_policy` evaluated on the NEW current request (as ongym_policy_actions; a negative
 * id skips it), and step records [batch], the new requests [batch], the statistics [batch] and the next actions / flags [batch]
 * come back together behind ONE synchronisation (host buffers only; next_actions / next_flags may be NULL when next_policy < 0). */
int ongym_step_actions_bundle(ongym_env *env, const int32_t *actions, int32_t next_policy, ongym_step_rec *rec_out,
                              ongym_request *request_out, ongym_stats *stats_out, int32_t *next_actions, uint8_t *next_flags);
/* The heuristic alone, without stepping: actions[batch], flags[batch] (ONGYM_F_BLOCKED_*) (heuristics.py:923-966). */
int ongym_policy_actions(ongym_env *env, int32_t policy, int32_t *actions, uint8_t *flags);

/* QRMSAEnv.observation() (qrmsa.pyx:583-781, gen_observation=True) for the CURRENT request of every replica:
 * obs  float32 [batch][1 + 2 + k_paths + k_paths*Mc*12]  (bit rate, src, dst, k route lengths, 12 features per (path,
 *      modulation) pair incl. the normalised GSNR of calculate_osnr_observation, core/osnr.pyx:259-369)
 * mask uint8   [batch][k_paths*Mc*n_slots + 1]            (info['mask'], last entry = reject = 1)
 * Mc = cfg.n_mods_consider.  For Mc < n_mods the call first sets ongym_stats.max_modulation_idx like
 * get_max_modulation_index (qrmsa.pyx:543-581) and describes the Mc formats at and below it.
 * Needs slot_bandwidth == channel_width*1e9 (the reference's observation mixes the two, core/osnr.pyx:259-369). With per-link
 * attenuation the interferer field is evaluated term by term (no pair table): same results, about ten times slower. */
int ongym_observe(ongym_env *env, float *obs, uint8_t *mask);

/* One uniformly random VALID action per replica from an action mask [batch][k_paths*Mc*n_slots + 1] (as ongym_observe
 * writes it): what gymnasium's `action_space.sample(mask=info["mask"])` does on the reference's Discrete action space
 * (qrmsa.pyx:319-321; wrappers/qrmsa_gym.py:74-75 hands the mask out) - the masked random policy that exercises the
 * observation path.  Deterministic in (seed, draw_index, global replica index); the stream is this library's counter-based
 * generator (include/ongym_traffic.h), not NumPy's.  The reject action is always valid, so a choice always exists.
 * mask / actions: host buffers, or device buffers with cfg.io_device (then nothing synchronises). */
int ongym_sample_actions(ongym_env *env, const uint8_t *mask, uint64_t seed, uint64_t draw_index, int32_t *actions);

/* Plugin-API queries on one replica (host buffers always): */
/* QRMSAEnv.get_available_slots(path) (qrmsa.pyx:1482-1512): out[n_slots], 1 = free on every link of the path */
int ongym_query_available(ongym_env *env, int32_t replica, int32_t path_id, int32_t *out);
/* calculate_osnr(env, service) (core/osnr.pyx:21-142) for a candidate (path, slot, nslots): out = gsnr, ase, nli dB */
int ongym_query_gsnr(ongym_env *env, int32_t replica, int32_t path_id, int32_t slot, int32_t nslots, double out[3]);
/* The same for `count` candidates of one replica in ONE launch (one wavefront per candidate) — what a plugin heuristic
 * that scores every feasible start needs (heuristics.py:272-328, 330-416, 647-749): cands int32 [count][3] =
 * {path_id, slot, nslots}; out double [count][3] = {gsnr, ase, nli} dB.  Host buffers. */
int ongym_query_gsnr_many(ongym_env *env, int32_t replica, int32_t count, const int32_t *cands, double *out);
/* QRMSAEnv._get_candidates(available_slots, n, total_slots) (qrmsa.pyx:515-541) on an ARBITRARY row (1 = free):
 * starts_out[total_slots] receives the feasible start slots in ascending order, *count their number.
 * total_slots <= 1023. State-independent (no replica argument). */
int ongym_query_candidates(ongym_env *env, const int32_t *row, int32_t total_slots, int32_t nslots,
                           int32_t *starts_out, int32_t *count);
/* QRMSAEnv.is_path_free(path, initial_slot, number_slots) (qrmsa.pyx:1248-1264): *out = 1 if free */
int ongym_query_path_free(ongym_env *env, int32_t replica, int32_t path_id, int32_t slot, int32_t nslots,
                          int32_t *out);
/* The reallocations defragment() made while the last step of `replica` processed its departures, in order:
 * out[min(*count, ONGYM_MOVE_LOG)]; *count is the total (entries beyond ONGYM_MOVE_LOG are not kept). */
int ongym_query_moves(ongym_env *env, int32_t replica, ongym_move *out, int32_t *count);
/* topology.graph["available_slots"] (qrmsa.pyx:306-309): out[n_links*n_slots] */
int ongym_query_grid(ongym_env *env, int32_t replica, int32_t *out);
/* topology.graph["running_services"]: out[capacity], *n = count */
int ongym_query_services(ongym_env *env, int32_t replica, ongym_service *out, int32_t *n);
/* QRMSAEnv.current_service */
int ongym_query_request(ongym_env *env, int32_t replica, ongym_request *out);

/* per-replica counters: out[batch] (host buffer) */
int ongym_stats_get(ongym_env *env, ongym_stats *out);

/* Diagnostic: resident workgroups (one wavefront = one replica each) per compute unit of the kernel that
 * ongym_step_policy(ONGYM_POLICY_FIRST_FIT) launches on this environment, its dynamic LDS bytes per replica, and whether it
 * is the lean kernel (1) or the generic one (0). */
int ongym_query_occupancy(ongym_env *env, int32_t *blocks_per_cu, int32_t *lds_bytes, int32_t *lean_kernel);
/* The same for the kernel ongym_step_policy(policy) launches.  Lean kernels (csrc/ongym_fast.hpp) exist for first fit, load
 * balancing, highest SNR and lowest fragmentation - the four heuristics the reference benchmark selects among
 * (examples/JOCN_Benchmark_2024/graph_load.py:116-125). */
int ongym_query_occupancy_policy(ongym_env *env, int32_t policy, int32_t *blocks_per_cu, int32_t *lds_bytes, int32_t *lean_kernel);
int ongym_sync(ongym_env *env);
/* Run every later call of this environment on the CALLER's HIP stream (a hipStream_t passed as void *, e.g. PyTorch-ROCm's
 * torch.cuda.current_stream().cuda_stream) instead of the environment's own: the environment's launches are then ordered with
 * the caller's kernels on that stream and an RL loop (observe -> policy network -> step) needs no host synchronisation
 * between them.  use_own != 0 returns to the environment's own stream (hip_stream is then ignored); with use_own == 0 a NULL
 * hip_stream is HIP's default (null) stream, which is what PyTorch's default current stream is.  The call first drains the
 * stream used so far.  The caller keeps ownership of its stream and must keep it alive while it is set.  (The reference's
 * counterpart is the implicit ordering of a single Python thread: wrappers/qrmsa_gym.py:45-59.) */
int ongym_set_stream(ongym_env *env, void *hip_stream, int32_t use_own);
/* Device time (ms, HIP events on the env's stream) of the most recent step launch; <0 if none. */
double ongym_last_kernel_ms(ongym_env *env);
const char *ongym_last_error(ongym_env *env);
/* sizes the host side needs to allocate buffers / check the build */
int32_t ongym_abi_version(void);
int32_t ongym_sizeof(int32_t what); /* 0 config, 1 request, 2 step_rec, 3 service, 4 stats */

#ifdef __cplusplus
}
#endif
#endif /* ONGYM_H */
