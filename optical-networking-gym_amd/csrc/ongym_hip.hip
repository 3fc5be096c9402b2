// ongym_hip.hip — kernels + C ABI (include/ongym.h) of the MI355X-native batched QRMSA environment.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o libongym_hip.so ongym_hip.hip   (see __graft_entry__.build)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <map>
#include <mutex>
#include <utility>

#include "ongym_device.hpp"
#include "ongym_host.hpp"
#include "ongym_fast.hpp"      // fast_lds_bytes, PathRec (the kernels themselves: ongym_fast.hip)
#include "ongym_scored.hpp"

using namespace ongym;

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
// WAVES = waves per SIMD the register allocation is bounded for: 5 when the replica's LDS block is <= 8 KiB (20 replicas
// per CU), else 4
constexpr int kPolicyMisc = 3;   // template value of the shared instantiation for policy ids >= ONGYM_POLICY_LOWEST_SPECTRUM

template <bool UA, bool R32, int WAVES, int POLICY, bool DEFRAG = false>
__global__ __launch_bounds__(64, WAVES) void k_run(const Params *__restrict__ Pp, int run_mode, int nsteps, const int32_t *actions, int32_t *act_out,
                                            uint8_t *flag_out, ongym_step_rec *out, int policy_id) {
    extern __shared__ __align__(16) unsigned char smem[];
    const Params &P = *Pp;
    Ctx c(P);
    c.lane = threadIdx.x;
    c.replica = blockIdx.x;
    c.lane_terms = 0;
    c.gn_evals = 0;
    c.gn_skips = 0;
    c.paths_tried = 0; c.path_hops = 0; c.active_sum = 0;
    ctx_bind(c, smem);
#ifdef ONGYM_STAMPS
    for (int i = 0; i < ONGYM_NSTAMPS; i++) c.stamp_acc[i] = 0;
    c.stamp_last = __builtin_amdgcn_s_memtime();
#endif
    if (POLICY == ONGYM_POLICY_HIGHEST_SNR) {   // extra LDS: Fx f64[2S+2] | Vw u64[8*16] | xlist u16[2S+2] | needx u8[2S+2]
        c.fl.Fx = reinterpret_cast<double *>(smem + lds_bytes(P));
        c.fl.Vw = reinterpret_cast<uint64_t *>(c.fl.Fx + 2 * P.n_slots + 2);
        c.fl.xlist = reinterpret_cast<uint16_t *>(c.fl.Vw + kMaxMods * kMaxRowWords);
        c.fl.needx = reinterpret_cast<uint8_t *>(c.fl.xlist + 2 * P.n_slots + 2);
        c.fl.vs = kMaxRowWords;
    }
    load_state(c);
    STAMP(c, 8);
    for (int it = 0; it < nsteps; ++it) {
        DevEnv *e = c.e;
        // kModeActionThenPolicy (the single-environment surface, ongym_step_actions_bundle): iteration 0 applies the action,
        // iteration 1 evaluates the policy on the request that step drew - the same code, no second launch
        const int mode = run_mode == kModeActionThenPolicy ? (it == 0 ? kModeActionStep : kModePolicyOnly) : run_mode;
        ongym_step_rec *rec = (out && mode != kModePolicyOnly) ? out + (size_t)it * P.batch + c.replica : nullptr;
        if (!e->have_request) {   // no request source / trace exhausted: the step is a no-op
            if (c.lane == 0) {
                e->st.flags |= ONGYM_F_NO_REQUEST;
                if (rec) { memset(rec, 0, sizeof(*rec)); rec->action = -1; rec->route = rec->modulation = rec->slot = -1;
                           rec->flags = ONGYM_F_NO_REQUEST; rec->active = c.active; }
                if (mode == kModePolicyOnly) { act_out[c.replica] = -1; if (flag_out) flag_out[c.replica] = ONGYM_F_NO_REQUEST; }
            }
            wave_sync();
            continue;
        }
        int src = e->cur_src, dst = e->cur_dst;
        float br = e->cur_br;
        double lp = e->launch_power, mg = e->margin;
        (void)br;
        STAMP(c, 0);
        Choice ch;
        int outcome;
        if (mode == kModeActionStep) {
            outcome = evaluate_action<UA, R32>(c, src, dst, lp, mg, actions[c.replica], ch);
        } else {
            if (POLICY == ONGYM_POLICY_HIGHEST_SNR) policy_highest_snr<UA, R32>(c, src, dst, lp, mg, ch);
            else if (POLICY == ONGYM_POLICY_LOAD_BALANCING) policy_load_balancing<UA, R32>(c, src, dst, lp, mg, ch);
            else if (POLICY == kPolicyMisc) policy_misc<UA, R32>(c, policy_id, src, dst, mg, ch);
            else if (POLICY == kPolicyScored)
                policy_scored<UA, R32>(c, policy_id, src, dst, mg, ch, reinterpret_cast<int32_t *>(smem + ((lds_bytes(P) + 15) & ~(size_t)15)));
            else policy_first_fit<UA, R32>(c, src, dst, lp, mg, ch);
            outcome = ch.route >= 0 ? (ch.busy ? 2 : 0) : 1;
            if (POLICY == kPolicyScored && mode == kModePolicyStep && ch.route >= 0) {
                // `env.step(action)` evaluates the GN model itself, at the width the format needs (lowest fragmentation asked
                // for one slot more): the reference raises ValueError if that fails (envs/qrmsa.pyx:925-929) — a fused
                // episode rejects the request instead and flags it
                outcome = evaluate_action<UA, R32>(c, src, dst, lp, mg, ch.action, ch);
                if (outcome == 3) {
                    outcome = 1;
                    ch.action = P.k_paths * P.n_mods * P.n_slots; ch.route = -1; ch.mod = -1; ch.slot = -1; ch.n = 0;
                    ch.flags = ONGYM_F_QOT_ERROR | ONGYM_F_BLOCKED_OSNR;
                }
            }
            // modulations_to_consider < n_mods: `env.step(action)` decodes the heuristic's action index with the window codec
            // (envs/qrmsa.pyx:801-834), which need not give back what the heuristic meant (heuristics.py:36-54 encodes formats
            // outside the window past the codec's range): do what the reference's loop does
            if (P.n_mods_consider < P.n_mods && mode == kModePolicyStep && ch.route >= 0)
                outcome = evaluate_action<UA, R32>(c, src, dst, lp, mg, ch.action, ch);
        }
        if (mode == kModePolicyOnly) {
            if (c.lane == 0) { act_out[c.replica] = ch.action; if (flag_out) flag_out[c.replica] = (uint8_t)ch.flags; }
            continue;
        }
        apply_step<R32, DEFRAG>(c, ch, outcome, rec);
    }
    if (run_mode != kModePolicyOnly) store_state(c);
    STAMP(c, 9);
#ifdef ONGYM_STAMPS
    if (c.lane == 0 && P.dbg)
        for (int i = 0; i < ONGYM_NSTAMPS; i++) atomicAdd(&P.dbg[i], c.stamp_acc[i]);
#endif
}

__global__ __launch_bounds__(64) void k_reset(const Params *__restrict__ Pp, const uint8_t *mask) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (mask && !mask[blockIdx.x]) return;
    const Params &P = *Pp;
    Ctx c(P);
    c.lane = threadIdx.x;
    c.replica = blockIdx.x;
    c.lane_terms = 0;
    c.gn_evals = 0;
    c.gn_skips = 0;
    c.paths_tried = 0; c.path_hops = 0; c.active_sum = 0;
    ctx_bind(c, smem);
    load_state(c);
    reset_env(c);
    store_state(c);
}

#ifndef ONGYM_OBS_WAVES
#define ONGYM_OBS_WAVES 3
#endif
template <bool UA, bool R32>
__global__ __launch_bounds__(64, ONGYM_OBS_WAVES) void k_observe(const Params *__restrict__ Pp, float *obs, uint8_t *mask) {
    extern __shared__ __align__(16) unsigned char smem[];
    const Params &P = *Pp;
    Ctx c(P);
    c.lane = threadIdx.x;
    c.replica = blockIdx.x;
    c.lane_terms = 0;
    c.gn_evals = 0;
    c.gn_skips = 0;
    c.paths_tried = 0; c.path_hops = 0; c.active_sum = 0;
    ctx_bind(c, smem);
#ifdef ONGYM_STAMPS
    for (int i = 0; i < ONGYM_NSTAMPS; i++) c.stamp_acc[i] = 0;
    c.stamp_last = __builtin_amdgcn_s_memtime();
#endif
    load_state(c);
    STAMPW(c, 0);
    const size_t obs_dim = 3 + P.k_paths + (size_t)P.k_paths * P.n_mods_consider * 12;
    const size_t nact = (size_t)P.k_paths * P.n_mods_consider * P.n_slots + 1;
    const ObsLayout lay = obs_layout(P);          // observe_lds on the host sizes the block with the same function
    double *Fx = reinterpret_cast<double *>(smem + lay.fx);
    uint16_t *xlist = reinterpret_cast<uint16_t *>(smem + lay.xlist);
    uint64_t *Vw = reinterpret_cast<uint64_t *>(smem + lay.vw);
    uint8_t *needx = smem + lay.needx;
    c.list = reinterpret_cast<uint16_t *>(smem + lay.list);
    wave_sync();
    c.fl = FieldLds{Fx, Vw, xlist, needx, lay.vs};
    observe_env<UA, R32>(c, Fx, Vw, xlist, needx, lay.vs, obs + (size_t)c.replica * obs_dim, mask + (size_t)c.replica * nact);
#ifdef ONGYM_STAMPS
    STAMPW(c, 15);
    if (c.lane == 0 && P.dbg)
        for (int i = 0; i < ONGYM_NSTAMPS; i++) atomicAdd(&P.dbg[i], c.stamp_acc[i]);
#endif
}

__global__ void k_seed(Params P, uint64_t seed, uint64_t replica_base) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.batch) return;
    P.env[r].rng_key = ongym_stream_key(seed, replica_base + (uint64_t)r);
    P.env[r].req_index = 0;
}

// reset(options={"only_episode_counters": True}) (qrmsa.pyx:427-464): the episode counters and histograms go to zero and the
// departure heap is DROPPED (self._events = []): the services that are running stay in the network for good (release time
// +inf; the 'disrupted' sign bit is kept).  Grid, running services, totals, clock and the current request are untouched.
__global__ void k_reset_counters(Params P, const uint8_t *mask) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.batch || (mask && !mask[r])) return;
    DevEnv &e = P.env[r];
    ongym_stats &s = e.st;
    e.svc_list_extra += s.episode_services_processed;   // len(graph["services"]) keeps growing from where it was
    s.episode_bit_rate_requested = 0.0; s.episode_bit_rate_provisioned = 0.0;
    s.episode_services_processed = 0; s.episode_services_accepted = 0;
    s.episode_disrupted_services = 0; s.rejected = 0;
    s.episode_defrag_cycles = 0; s.episode_service_reallocations = 0;
    for (int m = 0; m < 8; m++) s.episode_modulation_hist[m] = 0;
    s.max_modulation_idx = P.n_mods - 1;                          // :437
    float *rr = P.svc_r + (size_t)r * P.capacity;
    for (int i = 0; i < s.active; i++) rr[i] = copysignf(INFINITY, rr[i]);
}

// ongym_sample_actions: one wave per replica.  The mask bytes are 0/1: the row is read as aligned dwords (lane-contiguous,
// popcount of w & 0x01010101) plus at most six head / tail bytes; any fixed order of the entries gives a uniform choice, so
// the order is "by lane, then by the lane's strided dwords": pass 1 counts per lane, a wave scan picks the lane that holds the
// r-th valid entry, pass 2 (the same loads, L2-resident) finds it.
__global__ __launch_bounds__(64) void k_sample_mask(const uint8_t *__restrict__ mask, long long nact, uint64_t seed, uint64_t replica_base,
                                                    uint64_t draw, int32_t *__restrict__ actions) {
    const int lane = threadIdx.x;
    const long long replica = blockIdx.x;
    const uint8_t *row = mask + replica * nact;
    const uintptr_t a0 = ((uintptr_t)row + 3) & ~(uintptr_t)3, a1 = ((uintptr_t)(row + nact)) & ~(uintptr_t)3;
    const long long head = (long long)(a0 - (uintptr_t)row);                      // bytes before the aligned middle (0..3)
    const long long ndw = a1 > a0 ? (long long)((a1 - a0) >> 2) : 0;               // aligned dwords
    const long long tail0 = head + 4 * ndw;                                        // first tail byte (element index)
    const uint32_t *mid = reinterpret_cast<const uint32_t *>(a0);
    // lanes 0..2: one head byte each, lanes 3..5: one tail byte each
    long long extra = -1;
    if (lane < 3 && lane < head) extra = lane;
    if (lane >= 3 && lane < 6 && tail0 + (lane - 3) < nact) extra = tail0 + (lane - 3);
    int cnt = (extra >= 0 && row[extra]) ? 1 : 0;
    for (long long j = lane; j < ndw; j += 64) cnt += __popc(mid[j] & 0x01010101u);
    // inclusive scan of the per-lane counts
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
    const int total = __shfl(incl, 63);
    int choice = (int)(nact - 1);                                                  // the reject action (always valid)
    if (total > 0) {
        const double u = ongym_uniform(ongym_stream_key(seed ^ 0x9E3779B97F4A7C15ull, replica_base + (uint64_t)replica), draw);
        int r = (int)(u * (double)total);
        r = r >= total ? total - 1 : r;
        const uint64_t at = __ballot(incl > r);                                    // first lane whose inclusive count exceeds r
        const int owner = __builtin_ctzll(at);
        if (lane == owner) {
            int k = r - (incl - cnt);                                              // k-th valid entry of this lane (0-based)
            long long found = -1;
            if (extra >= 0 && row[extra]) { if (k == 0) found = extra; k--; }
            for (long long j = lane; j < ndw && found < 0; j += 64) {
                uint32_t w = mid[j] & 0x01010101u;
                const int c = __popc(w);
                if (k < c) {
                    for (int b = 0; b < 4; b++)
                        if ((w >> (8 * b)) & 1u) { if (k == 0) { found = head + 4 * j + b; break; } k--; }
                } else k -= c;
            }
            if (found >= 0) choice = (int)found;
            actions[replica] = choice;
        }
    } else if (lane == 0) actions[replica] = choice;
}

__global__ void k_rewind(Params P) {   // new trace: cursor back to 0
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.batch) return;
    P.env[r].req_index = 0;
}

// calculate_osnr for MANY candidates of one replica: one wavefront per candidate (cand = {path_id, slot, nslots}).
template <bool UA, bool R32>
__global__ __launch_bounds__(64) void k_query_gsnr_many(const Params *__restrict__ Pp, int replica, int count,
                                                        const int32_t *__restrict__ cand, double *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const Params &P = *Pp;
    if ((int)blockIdx.x >= count) return;
    Ctx c(P);
    c.lane = threadIdx.x;
    c.replica = replica;
    c.lane_terms = 0;
    c.gn_evals = 0;
    c.gn_skips = 0;
    c.paths_tried = 0; c.path_hops = 0; c.active_sum = 0;
    ctx_bind(c, smem);
    load_state(c);
    const int path = cand[3 * blockIdx.x], slot = cand[3 * blockIdx.x + 1], n = cand[3 * blockIdx.x + 2];
    PathRef p = load_path(c, path);
    int L = gn_build_list<R32>(c, p.m0, p.m1);
    GnLin lin = gn_eval<UA, R32>(c, p, L, slot, n);
    double g[3];
    gn_to_db(lin, g);
    if (c.lane == 0) { out[3 * blockIdx.x] = g[0]; out[3 * blockIdx.x + 1] = g[1]; out[3 * blockIdx.x + 2] = g[2]; }
}

enum { kQAvailable = 0, kQGsnr = 1, kQGrid = 2, kQServices = 3, kQRequest = 4, kQCandidates = 5, kQPathFree = 6 };

template <bool UA, bool R32>
__global__ __launch_bounds__(64) void k_query(const Params *__restrict__ Pp, int what, int replica, int path, int slot, int n,
                                              int32_t *out_i, double *out_d) {
    extern __shared__ __align__(16) unsigned char smem[];
    const Params &P = *Pp;
    Ctx c(P);
    c.lane = threadIdx.x;
    c.replica = replica;
    c.lane_terms = 0;
    c.gn_evals = 0;
    c.gn_skips = 0;
    c.paths_tried = 0; c.path_hops = 0; c.active_sum = 0;
    ctx_bind(c, smem);
    load_state(c);
    if (what == kQAvailable) {          // get_available_slots(path), envs/qrmsa.pyx:1482-1512
        PathRef p = load_path(c, path);
        uint64_t x = path_free_ext(c, p);
        for (int j = 0; j < P.n_slots; j++) {
            uint64_t w = __shfl((unsigned long long)x, j >> 6);
            if (c.lane == 0) out_i[j] = (int32_t)((w >> (j & 63)) & 1ull);
        }
    } else if (what == kQGsnr) {        // calculate_osnr(env, candidate), core/osnr.pyx:21-142
        PathRef p = load_path(c, path);
        int L = gn_build_list<R32>(c, p.m0, p.m1);
        GnLin lin = gn_eval<UA, R32>(c, p, L, slot, n);
        double g[3];
        gn_to_db(lin, g);
        if (c.lane == 0) { out_d[0] = g[0]; out_d[1] = g[1]; out_d[2] = g[2]; }
    } else if (what == kQGrid) {        // topology.graph["available_slots"]
        for (int i = c.lane; i < P.n_links * P.n_slots; i += kWave) {
            int l = i / P.n_slots, j = i % P.n_slots;
            out_i[i] = (int32_t)((c.occ[l * P.row_words + (j >> 6)] >> (j & 63)) & 1ull);
        }
    } else if (what == kQServices) {    // running services
        ongym_service *o = reinterpret_cast<ongym_service *>(out_i + 2);
        if (c.lane == 0) out_i[0] = c.active;
        for (int i = c.lane; i < c.active; i += kWave) {
            uint32_t a = c.sa[i], b = c.sb[i];
            o[i].path_id = rec_path<R32>(a, b); o[i].slot = (int16_t)rec_slot<R32>(a, b);
            o[i].nslots = (int16_t)rec_n<R32>(a, b); o[i].modulation = (int16_t)rec_mod<R32>(a, b);
            o[i].reserved = c.sr[i] < 0.f ? 1 : 0;   // 1: in the disrupted list (measure_disruptions)
            o[i].release_time = fabsf(c.sr[i]);
            o[i].service_id = P.track_ids ? (int32_t)c.sq[i] : -1; o[i].pad_ = 0;
            o[i].osnr = P.track_ids ? c.so[i] : 0.0;
        }
    } else if (what == kQCandidates) {  // _get_candidates on a caller-supplied row: path = total_slots, n = nslots
        const int total = path;
        const int32_t *row = out_i;         // input row [total], output flags at out_i[1024 .. 1024+total)
        uint64_t x = 0;
        for (int j = 0; j < 64; j++) {
            int sl = c.lane * 64 + j;
            if (sl < total && row[sl] != 0) x |= 1ull << j;
        }
        if (c.lane == (total >> 6)) x |= 1ull << (total & 63);
        int rr = 1;
        x = run_and(x, rr, n + 1);
        for (int j = 0; j < 64; j++) {
            int sl = c.lane * 64 + j;
            if (sl < total) out_i[1024 + sl] = (int32_t)((x >> j) & 1ull);
        }
    } else if (what == kQPathFree) {    // is_path_free, envs/qrmsa.pyx:1248-1264
        PathRef p = load_path(c, path);
        int rr = 1;
        uint64_t ok = run_and(path_free_ext(c, p), rr, n + 1);
        uint64_t w = __shfl((unsigned long long)ok, slot >> 6);
        if (c.lane == 0) out_i[0] = (int32_t)((w >> (slot & 63)) & 1ull);
    } else if (what == kQRequest) {
        if (c.lane == 0) {
            ongym_request *q = reinterpret_cast<ongym_request *>(out_i);
            q->arrival_time = c.e->cur_at; q->holding_time = c.e->cur_ht; q->bit_rate = c.e->cur_br;
            q->source = (int16_t)c.e->cur_src; q->destination = (int16_t)c.e->cur_dst;
        }
    }
}

static std::string g_create_error;

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static std::mutex g_lds_mutex;
hipError_t raise_lds_limit(int device, const void *kernel, size_t bytes) {
    static std::map<std::pair<int, const void *>, size_t> limit;
    if (bytes <= 64 * 1024) return hipSuccess;        // the default limit covers it
    std::lock_guard<std::mutex> lock(g_lds_mutex);
    size_t &cur = limit[std::make_pair(device, kernel)];
    if (bytes <= cur) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}

static int push_params(ongym_env *env) {
    HIP_TRY(env, hipMemcpyAsync(env->d_P, &env->P, sizeof(Params), hipMemcpyHostToDevice, env->stream));
    return 0;
}

template <typename T>
static int upload(ongym_env *env, const T *src, size_t n, const T **dst) {
    void *d = nullptr;
    HIP_TRY(env, hipMalloc(&d, n ? n * sizeof(T) : sizeof(T)));
    env->allocs.push_back(d);
    if (n) HIP_TRY(env, hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(d);
    return 0;
}

template <typename T>
static int dev_alloc(ongym_env *env, size_t n, T **dst, bool zero) {
    void *d = nullptr;
    HIP_TRY(env, hipMalloc(&d, n ? n * sizeof(T) : sizeof(T)));
    env->allocs.push_back(d);
    if (zero && n) HIP_TRY(env, hipMemset(d, 0, n * sizeof(T)));
    *dst = static_cast<T *>(d);
    return 0;
}

static int fail_arg(ongym_env *env, const char *msg, int code = ONGYM_E_ARG) {
    env->err = msg;
    return code;
}

static int build(ongym_env *env, const ongym_config *c) {
    Params &P = env->P;
    if (c->n_nodes <= 1 || c->n_links <= 0 || c->n_paths <= 0 || c->k_paths <= 0 || c->max_hops <= 0 ||
        c->n_mods <= 0 || c->n_slots <= 0 || c->batch <= 0 || c->episode_length <= 1)
        return fail_arg(env, "non-positive size in ongym_config");
    if (c->n_mods > kMaxMods) return fail_arg(env, "n_mods > 8", ONGYM_E_LIMIT);
    if (c->n_links > kMaxLinks) return fail_arg(env, "n_links > 128", ONGYM_E_LIMIT);
    if (c->max_hops > kMaxHops) return fail_arg(env, "max_hops > 64", ONGYM_E_LIMIT);
    if (c->n_slots > 1023) return fail_arg(env, "n_slots > 1023", ONGYM_E_LIMIT);
    if (c->n_paths > 65535) return fail_arg(env, "n_paths > 65535", ONGYM_E_LIMIT);
    if (c->capacity <= 0 || c->capacity % 64 || c->capacity > 65535)
        return fail_arg(env, "capacity must be a multiple of 64 in (0, 65535)");
    if (!c->pair_paths || !c->path_hops || !c->path_links || !c->link_nspans || !c->link_span_km || !c->link_alpha ||
        !c->link_nf || !c->mod_se || !c->mod_min_osnr || !c->node_cum)
        return fail_arg(env, "null table pointer in ongym_config");
    if (c->bit_rate_mode == 0 && (!c->bit_rates || !c->bit_rate_cum || c->n_bit_rates <= 0))
        return fail_arg(env, "discrete bit-rate mode needs bit_rates/bit_rate_cum");
    if (c->load <= 0 || c->mean_holding_time <= 0) return fail_arg(env, "load and mean_holding_time must be positive");

    const int N = c->n_nodes, E = c->n_links, NP = c->n_paths, K = c->k_paths, H = c->max_hops, M = c->n_mods;
    // validate tables on the host before any kernel indexes with them
    for (int i = 0; i < N * N * K; i++)
        if (c->pair_paths[i] < -1 || c->pair_paths[i] >= NP) return fail_arg(env, "pair_paths entry out of range");
    for (int p = 0; p < NP; p++) {
        if (c->path_hops[p] <= 0 || c->path_hops[p] > H) return fail_arg(env, "path_hops entry out of range");
        for (int h = 0; h < c->path_hops[p]; h++)
            if (c->path_links[p * H + h] < 0 || c->path_links[p * H + h] >= E)
                return fail_arg(env, "path_links entry out of range");
    }
    for (int m = 0; m < M; m++)
        if (c->mod_se[m] < 1 || c->mod_se[m] > 6) return fail_arg(env, "spectral efficiency must be 1..6");
    for (int e = 0; e < E; e++)
        if (!(c->link_alpha[e] > 0) || !(c->link_span_km[e] > 0) || c->link_nspans[e] <= 0)
            return fail_arg(env, "link span parameters must be positive");

    P.n_nodes = N; P.n_links = E; P.n_paths = NP; P.k_paths = K; P.max_hops = H; P.n_mods = M; P.n_slots = c->n_slots;
    P.n_mods_consider = (c->n_mods_consider <= 0 || c->n_mods_consider > M) ? M : c->n_mods_consider;   // qrmsa.pyx:313
    P.row_words = (c->n_slots + 63) / 64;
    P.ext_words = c->n_slots / 64 + 1;
    P.batch = c->batch; P.capacity = c->capacity; P.episode_length = c->episode_length; P.auto_reset = c->auto_reset;
    P.bit_rate_mode = c->bit_rate_mode; P.n_bit_rates = c->n_bit_rates; P.br_lo = c->bit_rate_lo; P.br_hi = c->bit_rate_hi;
    P.req_mode = kReqNone;
    P.measure_disruptions = c->measure_disruptions ? 1 : 0;
    P.defragmentation = c->defragmentation ? 1 : 0;
    P.track_ids = (c->defragmentation || c->track_service_ids) ? 1 : 0;
    P.n_defrag_services = c->n_defrag_services;
    if (c->defragmentation && c->n_defrag_services < 0) return fail_arg(env, "n_defrag_services must be >= 0");
    P.f0 = c->frequency_start; P.slot_bw = c->slot_bandwidth; P.channel_width = c->channel_width;
    P.nslots_width = c->nslots_channel_width > 0 ? c->nslots_channel_width : c->channel_width;
    P.mean_holding = c->mean_holding_time;
    P.max_bit_rate = c->max_bit_rate;
    if (c->bit_rate_mode == 0) env->cfg_bit_rates.assign(c->bit_rates, c->bit_rates + c->n_bit_rates);
    P.path_len_norm = nullptr;
    P.mean_holding_f = (float)c->mean_holding_time;

    // derived GN tables in fp64 (core/osnr.pyx:22-24, 52-55, 58-61, 109-125)
    const double pi = 3.14159265358979323846, beta2 = 21.3e-27, h_planck = 6.626e-34;
    std::vector<double> w1(E), w2(E), cl(E), selfc(E), ase_link(E);
    bool uniform = true;
    for (int e = 0; e < E; e++) {
        double a = c->link_alpha[e], L = c->link_span_km[e];
        double l_eff = (1.0 - std::exp(-2.0 * a * L * 1e3)) / (2.0 * a);
        w1[e] = (double)c->link_nspans[e] * l_eff;
        w2[e] = (double)c->link_nspans[e] * l_eff * (l_eff / (L * 1e3));
        cl[e] = pi * pi * beta2 * (1.0 / (2.0 * a));
        selfc[e] = pi * pi * beta2 / (4.0 * a);
        ase_link[e] = (double)c->link_nspans[e] * h_planck * (std::exp(2.0 * a * L * 1e3) - 1.0) * c->link_nf[e];
        if (a != c->link_alpha[0]) uniform = false;
    }
    P.uniform_alpha = uniform ? 1 : 0;
    if (c->measure_disruptions && !uniform) return fail_arg(env, "measure_disruptions needs uniform attenuation", ONGYM_E_LIMIT);
    if (c->defragmentation && !uniform) return fail_arg(env, "defragmentation needs uniform attenuation", ONGYM_E_LIMIT);
    if (c->track_service_ids && !uniform) return fail_arg(env, "track_service_ids needs uniform attenuation", ONGYM_E_LIMIT);
    P.rec32 = (E <= 32 && NP <= 512 && c->n_slots <= 1023) ? 1 : 0;
    P.alpha0_cl = cl[0];
    std::vector<uint64_t> mask((size_t)NP * 2, 0);
    std::vector<double> path_ase(NP, 0.0), path_w1(NP, 0.0);
    for (int p = 0; p < NP; p++)
        for (int h = 0; h < c->path_hops[p]; h++) {
            int l = c->path_links[p * H + h];
            mask[2 * p + (l >> 6)] |= 1ull << (l & 63);
            path_ase[p] += ase_link[l];
            path_w1[p] += w1[l];
        }
    std::vector<double> nli_coef((size_t)c->n_slots + 2, 0.0);
    for (int n = 1; n <= c->n_slots + 1; n++) {
        const double gamma = 1.3e-3, bwn = c->slot_bandwidth * n;
        nli_coef[n] = (8.0 / (27.0 * pi * beta2)) * (gamma * gamma) / (bwn * bwn);
    }
    std::vector<double> self_asinh((size_t)c->n_slots + 2, 0.0);   // uniform alpha only
    for (int n = 0; n <= c->n_slots + 1; n++) {
        double bwn = c->slot_bandwidth * n;
        self_asinh[n] = std::asinh(selfc[0] * (bwn * bwn));
    }
    static const double phi_mod[6] = {1.0, 1.0, 2.0 / 3.0, 17.0 / 25.0, 69.0 / 100.0, 13.0 / 21.0};
    for (int m = 0; m < M; m++) {
        P.mod_se[m] = c->mod_se[m];
        P.mod_thr[m] = c->mod_min_osnr[m];
        P.mod_phi53[m] = phi_mod[c->mod_se[m] - 1] * (5.0 / 3.0);
    }
    // slots needed per (discrete bit rate, modulation): get_number_slots, envs/qrmsa.pyx:1198-1205
    std::vector<int32_t> nreq_tab((size_t)std::max(c->n_bit_rates, 1) * kMaxMods, 0);
    if (c->bit_rate_mode == 0)
        for (int b = 0; b < c->n_bit_rates; b++)
            for (int m = 0; m < M; m++)
                nreq_tab[(size_t)b * kMaxMods + m] =
                    (int32_t)std::ceil((double)(float)c->bit_rates[b] / ((double)c->mod_se[m] * P.nslots_width));
    // ASE-only rejection is exact iff no interferer term asinh(u)-asinh(v) - Phi*(5/3)*(Bk/|df|)*(l_eff/L) of
    // core/osnr.pyx:68-93 can be negative: scan the whole discrete domain (slot counts x centre distances in half
    // slots x modulation formats x distinct link classes) once.
    {
        bool all_nonneg = true;
        std::vector<std::pair<double, double>> classes;   // (cl, l_eff/(L*1e3))
        for (int e = 0; e < E; e++) {
            std::pair<double, double> k(cl[e], w2[e] / w1[e]);
            bool seen = false;
            for (auto &q : classes) if (q == k) seen = true;
            if (!seen) classes.push_back(k);
        }
        const int S = c->n_slots;
        for (auto &cls : classes)
            for (int m = 0; m < M && all_nonneg; m++) {
                const double K = P.mod_phi53[m] * cls.second;
                for (int nk = 1; nk <= S && all_nonneg; nk++) {
                    const double bk = c->slot_bandwidth * nk, ck = cls.first * bk;
                    for (int dfi = nk + 1; dfi <= 2 * S; dfi++) {
                        const double adf = 0.5 * c->slot_bandwidth * dfi;
                        const double t = (std::asinh(ck * (adf + 0.5 * bk)) - std::asinh(ck * (adf - 0.5 * bk))) - K * (bk / adf);
                        if (!(t > 0.0)) { all_nonneg = false; break; }
                    }
                }
            }
        P.ase_shortcut = all_nonneg ? 1 : 0;
    }
    int rc;
    // (asinh difference, Bk/|df|) table of core/osnr.pyx:68-93 for uniform attenuation. Both factors depend only on the
    // interferer's slot count nk and the centre distance in half slots: precompute them in fp64 with the device's own
    // expressions. nk range: the largest slot count the configured traffic can produce (anything larger, e.g. from a
    // replayed trace, is computed on the fly by the kernel).
    P.pair_tab = nullptr; P.tab_nmax = 0; P.tab_stride = 2 * c->n_slots + 1;
    P.path_rec = nullptr; P.pair_tab2k = nullptr; P.pair_tabp = nullptr;
    std::vector<double2> host_tab;
    if (uniform) {
        double max_rate = 0.0;
        if (c->bit_rate_mode == 0) for (int b = 0; b < c->n_bit_rates; b++) max_rate = std::max(max_rate, c->bit_rates[b]);
        else max_rate = (double)c->bit_rate_hi;
        int min_se = 6;
        for (int m = 0; m < M; m++) min_se = std::min(min_se, (int)c->mod_se[m]);
        int nmax = (int)std::ceil(max_rate / ((double)min_se * P.nslots_width));
        nmax = std::max(1, std::min(nmax, c->n_slots));
        size_t entries = (size_t)nmax * P.tab_stride;
        if (entries * sizeof(double2) <= (size_t)16 << 20) {
            std::vector<double2> tab(entries);
            for (int nk = 1; nk <= nmax; nk++) {
                const double bk = c->slot_bandwidth * nk, ck = cl[0] * bk;
                for (int d = 0; d < P.tab_stride; d++) {
                    double2 v; v.x = 0.0; v.y = 0.0;
                    if (d > nk) {   // allocations never overlap: |df| > Bk/2
                        const double adf = (0.5 * c->slot_bandwidth) * (double)d;
                        const double U = ck * (adf + 0.5 * bk), V = ck * (adf - 0.5 * bk);
                        v.x = std::log((U + std::sqrt(std::fma(U, U, 1.0))) / (V + std::sqrt(std::fma(V, V, 1.0))));
                        v.y = bk / adf;
                    }
                    tab[(size_t)(nk - 1) * P.tab_stride + d] = v;
                }
            }
            if ((rc = upload(env, tab.data(), tab.size(), &P.pair_tab))) return rc;
            P.tab_nmax = nmax;
            host_tab.swap(tab);
        }
    }
    if ((rc = upload(env, nreq_tab.data(), nreq_tab.size(), &P.nreq_tab))) return rc;
    if ((rc = upload(env, c->pair_paths, (size_t)N * N * K, &P.pair_paths))) return rc;
    if ((rc = upload(env, c->path_hops, (size_t)NP, &P.path_hops))) return rc;
    if ((rc = upload(env, c->path_links, (size_t)NP * H, &P.path_links))) return rc;
    if ((rc = upload(env, mask.data(), mask.size(), &P.path_mask))) return rc;
    if ((rc = upload(env, path_ase.data(), path_ase.size(), &P.path_ase))) return rc;
    if ((rc = upload(env, path_w1.data(), path_w1.size(), &P.path_w1))) return rc;
    if ((rc = upload(env, self_asinh.data(), self_asinh.size(), &P.self_asinh))) return rc;
    if ((rc = upload(env, nli_coef.data(), nli_coef.size(), &P.nli_coef))) return rc;
    {   // (nli_coef[n], self_asinh[n]) of the slot count of every (discrete bit rate, modulation): lets the request draw
        // fetch them with the SAME index as nreq_tab instead of a second, dependent round trip through n
        std::vector<double> rc2(nreq_tab.size() * 2, 0.0);
        for (size_t i = 0; i < nreq_tab.size(); i++) {
            const int32_t n = nreq_tab[i];
            if (n >= 1 && n <= c->n_slots) { rc2[2 * i] = nli_coef[(size_t)n]; rc2[2 * i + 1] = self_asinh[(size_t)n]; }
        }
        if ((rc = upload(env, rc2.data(), rc2.size(), &P.req_coef))) return rc;
    }
    if ((rc = upload(env, w1.data(), w1.size(), &P.link_w1))) return rc;
    if ((rc = upload(env, w2.data(), w2.size(), &P.link_w2))) return rc;
    if ((rc = upload(env, cl.data(), cl.size(), &P.link_cl))) return rc;
    if ((rc = upload(env, selfc.data(), selfc.size(), &P.link_selfc))) return rc;
    double one = 1.0, zero = 0.0;
    if ((rc = upload(env, c->bit_rate_mode == 0 ? c->bit_rates : &zero, c->bit_rate_mode == 0 ? (size_t)c->n_bit_rates : 1, &P.bit_rates))) return rc;
    if ((rc = upload(env, c->bit_rate_mode == 0 ? c->bit_rate_cum : &one, c->bit_rate_mode == 0 ? (size_t)c->n_bit_rates : 1, &P.bit_rate_cum))) return rc;
    if (c->bit_rate_mode != 0) P.n_bit_rates = 1;
    if ((rc = upload(env, c->node_cum, (size_t)N, &P.node_cum))) return rc;
    if (c->path_len_norm && (rc = upload(env, c->path_len_norm, (size_t)NP, &P.path_len_norm))) return rc;
    {   // link_shannon_entropy_ (utils.pyx:61-79): p = block / total_slots; p * math.log(p) — CPython's math.log is this log()
        std::vector<double> plogp((size_t)c->n_slots + 1, 0.0);
        for (int n = 1; n <= c->n_slots; n++) { const double pr = (double)n / (double)c->n_slots; plogp[n] = pr * std::log(pr); }
        if ((rc = upload(env, plogp.data(), plogp.size(), &P.plogp))) return rc;
    }

    // the pair table once more with a row pitch of 2048 entries (index = row << 11 | distance: one address instruction);
    // used by the lean first-fit kernel and by the observation field builder
    if (!host_tab.empty() && P.tab_stride < kTabPitch && (size_t)P.tab_nmax * kTabPitch * 16 <= ((size_t)64 << 20)) {
        std::vector<double> t2((size_t)P.tab_nmax * kTabPitch * 2, 0.0);
        for (int nk = 0; nk < P.tab_nmax; nk++)
            for (int d = 0; d < P.tab_stride; d++) {
                t2[((size_t)nk * kTabPitch + d) * 2] = host_tab[(size_t)nk * P.tab_stride + d].x;
                t2[((size_t)nk * kTabPitch + d) * 2 + 1] = host_tab[(size_t)nk * P.tab_stride + d].y;
            }
        if ((rc = upload(env, t2.data(), t2.size(), &P.pair_tab2k))) return rc;
        std::vector<double> t3((size_t)P.tab_nmax * kTabPitch * 2, 0.0);
        for (int nk = 0; nk < P.tab_nmax; nk++)
            for (int d = 0; d < P.tab_stride; d++) {
                const size_t e = (size_t)nk * kTabPitch + (size_t)(d & 1) * (kTabPitch / 2) + (size_t)(d >> 1);
                t3[e * 2] = host_tab[(size_t)nk * P.tab_stride + d].x;
                t3[e * 2 + 1] = host_tab[(size_t)nk * P.tab_stride + d].y;
            }
        if ((rc = upload(env, t3.data(), t3.size(), &P.pair_tabp))) return rc;
    }
    // ---- lean first-fit kernel (ongym_fast.hpp): eligibility and its path table ----
    {
        bool ok = uniform && P.ase_shortcut && !P.track_ids && !P.measure_disruptions && c->bit_rate_mode == 0 &&
                  P.n_mods_consider == M &&
                  c->n_bit_rates <= 8 && E <= 52 && N <= 64 && P.pair_tab2k != nullptr;
        const char *force = std::getenv("ONGYM_FORCE_GENERIC");
        if (force && force[0] == '1') ok = false;
        int max_n = 0;
        if (ok) {
            for (int b = 0; b < c->n_bit_rates; b++) {
                const double r = c->bit_rates[b];
                if (!(r > 0) || r != std::floor(r) || r > 16777216.0) ok = false;   // integer-valued, exact as float32
                for (int m = 0; m < M; m++) max_n = std::max(max_n, std::min((int)nreq_tab[(size_t)b * kMaxMods + m], (int)c->n_slots));
            }
            if (max_n < 1 || max_n > 512 || max_n > P.tab_nmax) ok = false;
        }
        const bool m64 = !P.rec32;
        const size_t flds = fast_lds_bytes(E, P.row_words, c->capacity, m64);
        if (flds > 160 * 1024) ok = false;
        P.path_hash_keys = nullptr; P.path_hash_vals = nullptr; P.path_hash_bits = 0; P.pad_hash = 0;
        if (ok && m64) {
            // the M64 record keeps the link set (<= 41 bits) instead of the path id: it must identify the route
            if (E > 32 + (int)kM64HiBits) ok = false;
            int bits = 4;
            while ((1 << bits) < 4 * NP) bits++;
            std::vector<uint64_t> keys((size_t)1 << bits, ~0ull);
            std::vector<int32_t> vals((size_t)1 << bits, -1);
            for (int p = 0; p < NP && ok; p++) {
                const uint64_t key = mask[2 * p];
                uint32_t h = path_hash_slot(key, bits);
                while (keys[h] != ~0ull && keys[h] != key) h = (h + 1u) & ((1u << bits) - 1u);
                if (keys[h] == key) ok = false;          // two path ids with the same link set: keep the generic kernel
                keys[h] = key; vals[h] = p;
            }
            if (ok) {
                if ((rc = upload(env, keys.data(), keys.size(), &P.path_hash_keys))) return rc;
                if ((rc = upload(env, vals.data(), vals.size(), &P.path_hash_vals))) return rc;
                P.path_hash_bits = bits;
            }
        }
        if (ok) {
            std::vector<PathRec> recs((size_t)NP);
            for (int p = 0; p < NP; p++) {
                PathRec &r = recs[(size_t)p];
                r.hops = (uint32_t)c->path_hops[p]; r.mask_lo = (uint32_t)mask[2 * p];
                r.mask_hi = (uint32_t)(mask[2 * p] >> 32);
                r.id = (uint32_t)p; r.ase = path_ase[(size_t)p]; r.w1 = path_w1[(size_t)p];
            }
            const PathRec *d_recs = nullptr;
            if ((rc = upload(env, recs.data(), recs.size(), &d_recs))) return rc;
            P.path_rec = d_recs;
            env->fast_ok = true; env->fast_m64 = m64; env->fast_lds = flds;
            // every record the lean kernels ever see carries a slot count of the traffic table (records written by the other
            // entry points included: eligibility demands discrete bit rates): none above 32 -> the narrow build
            const char *fw = std::getenv("ONGYM_FORCE_WIDE");
            env->fast_wide = max_n > 32 || (fw && fw[0] == '1');
        }
    }

    // mutable state
    const size_t B = (size_t)c->batch;
    if ((rc = dev_alloc(env, B * E * P.row_words, &P.occ, true))) return rc;
    if ((rc = dev_alloc(env, B * c->capacity, &P.svc_a, true))) return rc;
    if ((rc = dev_alloc(env, B * c->capacity, &P.svc_b, true))) return rc;
    if ((rc = dev_alloc(env, B * c->capacity, &P.svc_r, true))) return rc;
    P.svc_q = nullptr; P.svc_o = nullptr; P.move_log = nullptr; P.move_n = nullptr;
    if (P.track_ids) {
        if ((rc = dev_alloc(env, B * c->capacity, &P.svc_q, true))) return rc;
        if ((rc = dev_alloc(env, B * c->capacity, &P.svc_o, true))) return rc;
        if ((rc = dev_alloc(env, B * ONGYM_MOVE_LOG, &P.move_log, true))) return rc;
        if ((rc = dev_alloc(env, B, &P.move_n, true))) return rc;
    }
    if ((rc = dev_alloc(env, B, &P.env, false))) return rc;
    std::vector<DevEnv> host(B);
    memset(host.data(), 0, B * sizeof(DevEnv));
    for (size_t r = 0; r < B; r++) {
        DevEnv &d = host[r];
        d.launch_power = c->replica_launch_power_w ? c->replica_launch_power_w[r] : c->launch_power_w;
        d.margin = c->replica_margin ? c->replica_margin[r] : c->margin;
        double load = c->replica_load ? c->replica_load[r] : c->load;
        if (!(load > 0) || !(d.launch_power > 0)) return fail_arg(env, "per-replica load / launch power must be positive");
        d.mean_iat = 1 / (load / c->mean_holding_time);   // set_load, envs/qrmsa.pyx:1124-1132
        d.mean_iat_f = (float)d.mean_iat;
    }
    HIP_TRY(env, hipMemcpy(P.env, host.data(), B * sizeof(DevEnv), hipMemcpyHostToDevice));
    // the bitmaps start "all free" so that queries before the first reset see an empty network
    {
        std::vector<uint64_t> row(P.row_words);
        for (int w = 0; w < P.row_words; w++) {
            int a = 0, b = std::min(c->n_slots - 64 * w, 64);
            row[w] = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
            (void)a;
        }
        std::vector<uint64_t> all(B * E * P.row_words);
        for (size_t i = 0; i < all.size(); i++) all[i] = row[i % P.row_words];
        HIP_TRY(env, hipMemcpy(P.occ, all.data(), all.size() * 8, hipMemcpyHostToDevice));
    }
    env->lds = lds_bytes(P);
    if (env->lds > 64 * 1024) {
        if (env->lds > 160 * 1024) return fail_arg(env, "state does not fit the 160 KiB LDS: lower capacity", ONGYM_E_LIMIT);
#define ONGYM_SET_LDS(K) HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&K), env->lds))
        ONGYM_SET_LDS((k_run<true, true, 4, 0>)); ONGYM_SET_LDS((k_run<true, false, 4, 0>));
        ONGYM_SET_LDS((k_run<false, true, 4, 0>)); ONGYM_SET_LDS((k_run<false, false, 4, 0>));
        ONGYM_SET_LDS((k_run<true, true, 4, 1>)); ONGYM_SET_LDS((k_run<true, false, 4, 1>));
        ONGYM_SET_LDS((k_run<false, true, 4, 1>)); ONGYM_SET_LDS((k_run<false, false, 4, 1>));
        ONGYM_SET_LDS((k_run<true, true, 4, kPolicyMisc>)); ONGYM_SET_LDS((k_run<true, false, 4, kPolicyMisc>));
        ONGYM_SET_LDS((k_run<false, true, 4, kPolicyMisc>)); ONGYM_SET_LDS((k_run<false, false, 4, kPolicyMisc>));
        if (P.track_ids) {
            ONGYM_SET_LDS((k_run<true, true, 4, 0, true>)); ONGYM_SET_LDS((k_run<true, false, 4, 0, true>));
            ONGYM_SET_LDS((k_run<true, true, 4, 1, true>)); ONGYM_SET_LDS((k_run<true, false, 4, 1, true>));
            ONGYM_SET_LDS((k_run<true, true, 4, kPolicyMisc, true>)); ONGYM_SET_LDS((k_run<true, false, 4, kPolicyMisc, true>));
        }
        ONGYM_SET_LDS((k_query<true, true>)); ONGYM_SET_LDS((k_query<true, false>));
        ONGYM_SET_LDS((k_query<false, true>)); ONGYM_SET_LDS((k_query<false, false>));
        ONGYM_SET_LDS((k_query_gsnr_many<true, true>)); ONGYM_SET_LDS((k_query_gsnr_many<true, false>));
        ONGYM_SET_LDS((k_query_gsnr_many<false, true>)); ONGYM_SET_LDS((k_query_gsnr_many<false, false>));
        ONGYM_SET_LDS(k_reset);
#undef ONGYM_SET_LDS
    }
    if (env->fast_ok) {   // the lean kernels of every policy that has one (ongym_fast.hip): LDS limits; a policy whose block
                          // does not fit the CU keeps the generic kernel
        if (ONGYM_FAST_CALL(fast_prepare, 0, env)) env->fast_ok = false;
        env->fast_lb_ok = env->fast_ok && ONGYM_FAST_CALL(fast_prepare, 1, env) == 0;
        env->fast_hsnr_ok = env->fast_ok && ONGYM_FAST_CALL(fast_prepare, 2, env) == 0;
        env->fast_lf_ok = env->fast_ok && ONGYM_FAST_CALL(fast_prepare, 10, env) == 0;
        env->err.clear();
    }
    // scratch for queries / host-buffer I/O
    env->scratch_i_bytes = std::max(std::max((size_t)E * c->n_slots * 4, (size_t)c->capacity * sizeof(ongym_service) + 16), (size_t)2048 * 4);
    if ((rc = dev_alloc(env, env->scratch_i_bytes / 4 + 4, &env->d_scratch_i, true))) return rc;
    if ((rc = dev_alloc(env, 4, &env->d_scratch_d, true))) return rc;
    if ((rc = dev_alloc(env, B, &env->d_actions, true))) return rc;
    if ((rc = dev_alloc(env, B, &env->d_act_out, true))) return rc;
    if ((rc = dev_alloc(env, B, &env->d_flag_out, true))) return rc;
    if ((rc = dev_alloc(env, B, &env->d_mask, true))) return rc;
#ifdef ONGYM_STAMPS
    if ((rc = dev_alloc(env, 16, &P.dbg, true))) return rc;
#endif
    if ((rc = dev_alloc(env, 1, &env->d_P, false))) return rc;
    return push_params(env);
}

extern "C" {

int32_t ongym_abi_version(void) { return ONGYM_ABI_VERSION; }

int32_t ongym_sizeof(int32_t what) {
    switch (what) {
        case 0: return (int32_t)sizeof(ongym_config);
        case 1: return (int32_t)sizeof(ongym_request);
        case 2: return (int32_t)sizeof(ongym_step_rec);
        case 3: return (int32_t)sizeof(ongym_service);
        case 4: return (int32_t)sizeof(ongym_stats);
        case 5: return (int32_t)sizeof(ongym_move);
        default: return -1;
    }
}

const char *ongym_last_error(ongym_env *env) { return env ? env->err.c_str() : g_create_error.c_str(); }

int ongym_create(const ongym_config *cfg, ongym_env **out) {
    if (!cfg || !out) { g_create_error = "null argument"; return ONGYM_E_ARG; }
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ongym_config) || cfg->abi_version != ONGYM_ABI_VERSION) {
        g_create_error = "ongym_config struct_size / abi_version mismatch";
        return ONGYM_E_ARG;
    }
    ongym_env *env = new (std::nothrow) ongym_env();
    if (!env) { g_create_error = "out of memory"; return ONGYM_E_ARG; }
    env->cfg = *cfg;
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        g_create_error = "no usable HIP device (hipGetDeviceCount: " + std::string(hipGetErrorString(he)) + ")";
        delete env;
        return ONGYM_E_HIP;
    }
    int rc = 0;
    do {
        if (hipSetDevice(cfg->device) != hipSuccess) { env->err = "hipSetDevice failed"; rc = ONGYM_E_HIP; break; }
        if (hipStreamCreateWithFlags(&env->own_stream, hipStreamNonBlocking) != hipSuccess) { env->err = "hipStreamCreate failed"; rc = ONGYM_E_HIP; break; }
        env->stream = env->own_stream;
        if (hipEventCreate(&env->ev0) != hipSuccess || hipEventCreate(&env->ev1) != hipSuccess) { env->err = "hipEventCreate failed"; rc = ONGYM_E_HIP; break; }
        rc = build(env, cfg);
    } while (0);
    if (rc) {
        g_create_error = env->err;
        ongym_destroy(env);
        return rc;
    }
    *out = env;
    return ONGYM_OK;
}

void ongym_destroy(ongym_env *env) {
    if (!env) return;
    (void)hipSetDevice(env->cfg.device);
    if (env->stream) (void)hipStreamSynchronize(env->stream);
    for (void *p : env->allocs) (void)hipFree(p);
    if (env->d_trace) (void)hipFree(env->d_trace);
    if (env->d_out) (void)hipFree(env->d_out);
    if (env->h_pinned) (void)hipHostFree(env->h_pinned);
    if (env->ev0) (void)hipEventDestroy(env->ev0);
    if (env->ev1) (void)hipEventDestroy(env->ev1);
    if (env->own_stream) (void)hipStreamDestroy(env->own_stream);      // a caller's stream (ongym_set_stream) is the caller's
    delete env;
}

/* Resident workgroups (= replicas = wavefronts) per compute unit of the kernel that ongym_step_policy(first fit) launches,
 * as the HIP occupancy query reports it for this environment's LDS size. Diagnostic. */
static bool lean_policy(const ongym_env *env, int policy) {      // does ongym_step_policy(policy) run a lean kernel?
    if (!env->fast_ok || env->trace_used) return false;
    if (!(env->P.req_mode == kReqRng || (env->P.req_mode == kReqTrace && env->trace_fast_ok))) return false;
    switch (policy) {
        case ONGYM_POLICY_FIRST_FIT: return true;
        case ONGYM_POLICY_LOAD_BALANCING: return env->fast_lb_ok;
        case ONGYM_POLICY_HIGHEST_SNR: return env->fast_hsnr_ok;
        case ONGYM_POLICY_LOWEST_FRAGMENTATION: return env->fast_lf_ok;
        default: return false;
    }
}

int ongym_query_occupancy(ongym_env *env, int32_t *blocks_per_cu, int32_t *lds_bytes, int32_t *lean_kernel) {
    return ongym_query_occupancy_policy(env, ONGYM_POLICY_FIRST_FIT, blocks_per_cu, lds_bytes, lean_kernel);
}

int ongym_query_occupancy_policy(ongym_env *env, int32_t policy, int32_t *blocks_per_cu, int32_t *lds_bytes, int32_t *lean_kernel) {
    if (!env || !blocks_per_cu || !lds_bytes || !lean_kernel) return ONGYM_E_ARG;
    if (policy < ONGYM_POLICY_FIRST_FIT || policy >= ONGYM_POLICY_COUNT) return fail_arg(env, "unknown policy id");
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    int nb = 0, lds = 0;
    const bool lean = lean_policy(env, policy);
    if (lean) {
        int rc;
        switch (policy) {
            case ONGYM_POLICY_LOAD_BALANCING: rc = ONGYM_FAST_CALL(fast_occupancy, 1, env, &nb, &lds); break;
            case ONGYM_POLICY_HIGHEST_SNR: rc = ONGYM_FAST_CALL(fast_occupancy, 2, env, &nb, &lds); break;
            case ONGYM_POLICY_LOWEST_FRAGMENTATION: rc = ONGYM_FAST_CALL(fast_occupancy, 10, env, &nb, &lds); break;
            default: rc = ONGYM_FAST_CALL(fast_occupancy, 0, env, &nb, &lds); break;
        }
        if (rc) return rc;
        *lds_bytes = lds;
    } else {
        // the generic kernel's state block (the policies with scratch ask for more at launch time)
        if (env->lds <= 8192) HIP_TRY(env, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_run<true, true, 5, 0>, 64, env->lds));
        else HIP_TRY(env, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_run<true, true, 4, 0>, 64, env->lds));
        *lds_bytes = (int32_t)env->lds;
    }
    *blocks_per_cu = nb;
    *lean_kernel = lean ? 1 : 0;
    return ONGYM_OK;
}

int ongym_sync(ongym_env *env) {
    if (!env) return ONGYM_E_ARG;
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    return ONGYM_OK;
}

int ongym_set_stream(ongym_env *env, void *hip_stream, int32_t use_own) {
    if (!env) return ONGYM_E_ARG;
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    HIP_TRY(env, hipStreamSynchronize(env->stream));        // nothing of this environment is in flight on the old stream
    env->stream = use_own ? env->own_stream : static_cast<hipStream_t>(hip_stream);
    env->timed = false;                                     // the events were recorded on the old stream
    return ONGYM_OK;
}

double ongym_last_kernel_ms(ongym_env *env) {
    if (!env || !env->timed) return -1.0;
    if (hipEventSynchronize(env->ev1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, env->ev0, env->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

int ongym_seed(ongym_env *env, uint64_t seed) { return ongym_seed_base(env, seed, 0); }

int ongym_seed_base(ongym_env *env, uint64_t seed, uint64_t replica_base) {
    if (!env) return ONGYM_E_ARG;
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    env->P.req_mode = kReqRng;
    env->has_source = true;
    env->replica_base = replica_base;
    { int rc = push_params(env); if (rc) return rc; }
    int threads = 256, blocks = (env->P.batch + threads - 1) / threads;
    hipLaunchKernelGGL(k_seed, dim3(blocks), dim3(threads), 0, env->stream, env->P, seed, replica_base);
    HIP_TRY(env, hipGetLastError());
    return ONGYM_OK;
}

int ongym_set_requests(ongym_env *env, const ongym_request *reqs, int64_t n_per_replica) {
    if (!env || !reqs || n_per_replica <= 0) return env ? fail_arg(env, "bad trace") : ONGYM_E_ARG;
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    size_t n = (size_t)env->P.batch * (size_t)n_per_replica;
    // validate node indices on the host when the trace is a host buffer (device traces are the caller's contract)
    if (!env->cfg.io_device) {
        for (size_t i = 0; i < n; i++) {
            const ongym_request &q = reqs[i];
            if (q.source < 0 || q.source >= env->P.n_nodes || q.destination < 0 || q.destination >= env->P.n_nodes ||
                q.source == q.destination || !(q.bit_rate > 0))
                return fail_arg(env, "trace entry with invalid node pair / bit rate");
        }
        if (env->d_trace) { (void)hipFree(env->d_trace); env->d_trace = nullptr; }
        HIP_TRY(env, hipMalloc(&env->d_trace, n * sizeof(ongym_request)));
        HIP_TRY(env, hipMemcpy(env->d_trace, reqs, n * sizeof(ongym_request), hipMemcpyHostToDevice));
        env->P.trace = static_cast<const ongym_request *>(env->d_trace);
    } else {
        env->P.trace = reqs;
    }
    env->P.trace_n = n_per_replica;
    env->P.req_mode = kReqTrace;
    // the lean kernel replays a trace whose bit rates all come from the configured discrete table (it addresses per-request
    // constants by bit-rate index); anything else is the generic kernel's, for good (records may exceed the lean codec)
    env->trace_fast_ok = false;
    if (!env->cfg.io_device && env->fast_ok) {
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++) {
            bool hit = false;
            for (int b = 0; b < env->P.n_bit_rates; b++) hit |= (float)env->cfg_bit_rates[(size_t)b] == reqs[i].bit_rate;
            ok = hit;
        }
        env->trace_fast_ok = ok;
    }
    if (!env->trace_fast_ok) env->trace_used = true;
    env->has_source = true;
    { int rc = push_params(env); if (rc) return rc; }
    int threads = 256, blocks = (env->P.batch + threads - 1) / threads;
    hipLaunchKernelGGL(k_rewind, dim3(blocks), dim3(threads), 0, env->stream, env->P);
    HIP_TRY(env, hipGetLastError());
    return ONGYM_OK;
}

int ongym_reset(ongym_env *env, const uint8_t *mask) {
    if (!env) return ONGYM_E_ARG;
    if (!env->has_source) { env->err = "no request source: call ongym_seed or ongym_set_requests first"; return ONGYM_E_STATE; }
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    const uint8_t *dmask = nullptr;
    if (mask) {
        if (env->cfg.io_device) dmask = mask;
        else {
            HIP_TRY(env, hipMemcpyAsync(env->d_mask, mask, (size_t)env->P.batch, hipMemcpyHostToDevice, env->stream));
            dmask = env->d_mask;
        }
    }
    hipLaunchKernelGGL(k_reset, dim3(env->P.batch), dim3(64), env->lds, env->stream, env->d_P, dmask);
    HIP_TRY(env, hipGetLastError());
    return ONGYM_OK;
}

int ongym_reset_episode_counters(ongym_env *env, const uint8_t *mask) {
    if (!env) return ONGYM_E_ARG;
    if (!env->P.track_ids) {
        env->err = "ongym_reset_episode_counters needs cfg.track_service_ids (service ids restart under running services, "
                   "and calculate_osnr skips interferers by service id)";
        return ONGYM_E_STATE;
    }
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    const uint8_t *dmask = nullptr;
    if (mask) {
        if (env->cfg.io_device) dmask = mask;
        else {
            HIP_TRY(env, hipMemcpyAsync(env->d_mask, mask, (size_t)env->P.batch, hipMemcpyHostToDevice, env->stream));
            dmask = env->d_mask;
        }
    }
    int threads = 64, blocks = (env->P.batch + threads - 1) / threads;
    hipLaunchKernelGGL(k_reset_counters, dim3(blocks), dim3(threads), 0, env->stream, env->P, dmask);
    HIP_TRY(env, hipGetLastError());
    return ONGYM_OK;
}

static size_t observe_lds(const ongym_env *env) { return obs_layout(env->P).total; }   // k_observe's block (ongym_device.hpp)

static size_t field_lds(const ongym_env *env) {   // k_observe / highest-SNR k_run: state block + Fx, Vw, xlist, needx
    const Params &P = env->P;
    return ((env->lds + ((size_t)2 * P.n_slots + 2) * (sizeof(double) + 2 + 1) + kMaxMods * kMaxRowWords * 8) + 15) & ~(size_t)15;
}

// lowest fragmentation needs no scratch; MSCL the block of scored_lds_bytes
static size_t scored_lds(const ongym_env *env, int policy) {
    return policy == ONGYM_POLICY_MSCL ? ((env->lds + 15) & ~(size_t)15) + scored_lds_bytes(env->P.row_words) : env->lds;
}

static int launch_run(ongym_env *env, int mode, int policy, int nsteps, const int32_t *d_actions, int32_t *d_act_out,
                      uint8_t *d_flag_out, ongym_step_rec *d_out) {
    HIP_TRY(env, hipEventRecord(env->ev0, env->stream));
    const dim3 grid(env->P.batch), block(64);
    if (mode == kModePolicyStep && lean_policy(env, policy)) {
        // the lean kernels: same results, half the issued instructions (ongym_fast.hpp)
        int rc;
        switch (policy) {
            case ONGYM_POLICY_LOAD_BALANCING: rc = ONGYM_FAST_CALL(fast_launch, 1, env, nsteps, d_out); break;
            case ONGYM_POLICY_HIGHEST_SNR: rc = ONGYM_FAST_CALL(fast_launch, 2, env, nsteps, d_out); break;
            case ONGYM_POLICY_LOWEST_FRAGMENTATION: rc = ONGYM_FAST_CALL(fast_launch, 10, env, nsteps, d_out); break;
            default: rc = ONGYM_FAST_CALL(fast_launch, 0, env, nsteps, d_out); break;
        }
        if (rc) return rc;
        HIP_TRY(env, hipEventRecord(env->ev1, env->stream));
        env->timed = true;
        return 0;
    }
    if (policy == ONGYM_POLICY_HIGHEST_SNR && field_lds(env) > 64 * 1024) {
        HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<true, true, 4, ONGYM_POLICY_HIGHEST_SNR>), field_lds(env)));
        HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<true, false, 4, ONGYM_POLICY_HIGHEST_SNR>), field_lds(env)));
        HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<false, true, 4, ONGYM_POLICY_HIGHEST_SNR>), field_lds(env)));
        HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<false, false, 4, ONGYM_POLICY_HIGHEST_SNR>), field_lds(env)));
    }
#define ONGYM_LAUNCH_DEFRAG(R, POL, LDS)                                                                           \
    hipLaunchKernelGGL((k_run<true, R, 4, POL, true>), grid, block, LDS, env->stream, env->d_P, mode, nsteps,      \
                       d_actions, d_act_out, d_flag_out, d_out, policy)
    if (env->P.track_ids) {   // defragmentation / service-id tracking (uniform attenuation, checked at create): own instantiations
        if (policy == ONGYM_POLICY_HIGHEST_SNR) {
            if (field_lds(env) > 64 * 1024) {
                HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<true, true, 4, ONGYM_POLICY_HIGHEST_SNR, true>), field_lds(env)));
                HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<true, false, 4, ONGYM_POLICY_HIGHEST_SNR, true>), field_lds(env)));
            }
            if (env->P.rec32) ONGYM_LAUNCH_DEFRAG(true, ONGYM_POLICY_HIGHEST_SNR, field_lds(env));
            else ONGYM_LAUNCH_DEFRAG(false, ONGYM_POLICY_HIGHEST_SNR, field_lds(env));
        } else if (policy == ONGYM_POLICY_LOAD_BALANCING) {
            if (env->P.rec32) ONGYM_LAUNCH_DEFRAG(true, ONGYM_POLICY_LOAD_BALANCING, env->lds);
            else ONGYM_LAUNCH_DEFRAG(false, ONGYM_POLICY_LOAD_BALANCING, env->lds);
        } else if (policy >= ONGYM_POLICY_LOWEST_FRAGMENTATION) {
            if (scored_lds(env, policy) > 64 * 1024) {
                HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<true, true, 4, kPolicyScored, true>), scored_lds(env, policy)));
                HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<true, false, 4, kPolicyScored, true>), scored_lds(env, policy)));
            }
            if (env->P.rec32) ONGYM_LAUNCH_DEFRAG(true, kPolicyScored, scored_lds(env, policy));
            else ONGYM_LAUNCH_DEFRAG(false, kPolicyScored, scored_lds(env, policy));
        } else if (policy >= ONGYM_POLICY_LOWEST_SPECTRUM) {
            if (env->P.rec32) ONGYM_LAUNCH_DEFRAG(true, kPolicyMisc, env->lds);
            else ONGYM_LAUNCH_DEFRAG(false, kPolicyMisc, env->lds);
        } else {
            if (env->P.rec32) ONGYM_LAUNCH_DEFRAG(true, ONGYM_POLICY_FIRST_FIT, env->lds);
            else ONGYM_LAUNCH_DEFRAG(false, ONGYM_POLICY_FIRST_FIT, env->lds);
        }
        HIP_TRY(env, hipGetLastError());
        HIP_TRY(env, hipEventRecord(env->ev1, env->stream));
        env->timed = true;
        return 0;
    }
#undef ONGYM_LAUNCH_DEFRAG
#define ONGYM_LAUNCH_RUN(UA, R)                                                                                    \
    do {                                                                                                           \
        if (policy == ONGYM_POLICY_HIGHEST_SNR)                                                                    \
            hipLaunchKernelGGL((k_run<UA, R, 4, ONGYM_POLICY_HIGHEST_SNR>), grid, block, field_lds(env),            \
                               env->stream, env->d_P, mode, nsteps, d_actions, d_act_out, d_flag_out, d_out, policy);      \
        else if (policy == ONGYM_POLICY_LOAD_BALANCING)                                                            \
            hipLaunchKernelGGL((k_run<UA, R, 4, ONGYM_POLICY_LOAD_BALANCING>), grid, block, env->lds, env->stream,  \
                               env->d_P, mode, nsteps, d_actions, d_act_out, d_flag_out, d_out, policy);                   \
        else if (policy >= ONGYM_POLICY_LOWEST_FRAGMENTATION) {                                                    \
            if (scored_lds(env, policy) > 64 * 1024)                                                                       \
                HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_run<UA, R, 4, kPolicyScored>), scored_lds(env, policy))); \
            hipLaunchKernelGGL((k_run<UA, R, 4, kPolicyScored>), grid, block, scored_lds(env, policy), env->stream,         \
                               env->d_P, mode, nsteps, d_actions, d_act_out, d_flag_out, d_out, policy);                   \
        } else if (policy >= ONGYM_POLICY_LOWEST_SPECTRUM)                                                         \
            hipLaunchKernelGGL((k_run<UA, R, 4, kPolicyMisc>), grid, block, env->lds, env->stream,                  \
                               env->d_P, mode, nsteps, d_actions, d_act_out, d_flag_out, d_out, policy);                   \
        else if (env->lds <= 8192)                                                                                 \
            hipLaunchKernelGGL((k_run<UA, R, 5, ONGYM_POLICY_FIRST_FIT>), grid, block, env->lds, env->stream,       \
                               env->d_P, mode, nsteps, d_actions, d_act_out, d_flag_out, d_out, policy);                   \
        else                                                                                                       \
            hipLaunchKernelGGL((k_run<UA, R, 4, ONGYM_POLICY_FIRST_FIT>), grid, block, env->lds, env->stream,       \
                               env->d_P, mode, nsteps, d_actions, d_act_out, d_flag_out, d_out, policy);                   \
    } while (0)
    if (env->P.uniform_alpha) { if (env->P.rec32) ONGYM_LAUNCH_RUN(true, true); else ONGYM_LAUNCH_RUN(true, false); }
    else { if (env->P.rec32) ONGYM_LAUNCH_RUN(false, true); else ONGYM_LAUNCH_RUN(false, false); }
#undef ONGYM_LAUNCH_RUN
    HIP_TRY(env, hipGetLastError());
    HIP_TRY(env, hipEventRecord(env->ev1, env->stream));
    env->timed = true;
    return 0;
}

static int ensure_out(ongym_env *env, size_t n) {
    if (env->d_out_n >= n) return 0;
    if (env->d_out) { (void)hipFree(env->d_out); env->d_out = nullptr; env->d_out_n = 0; }
    HIP_TRY(env, hipMalloc(reinterpret_cast<void **>(&env->d_out), n * sizeof(ongym_step_rec)));
    env->d_out_n = n;
    return 0;
}

int ongym_step_policy(ongym_env *env, int32_t policy, int32_t nsteps, ongym_step_rec *out) {
    if (!env) return ONGYM_E_ARG;
    if (policy < ONGYM_POLICY_FIRST_FIT || policy >= ONGYM_POLICY_COUNT) return fail_arg(env, "unknown policy id");
    if (policy >= ONGYM_POLICY_LOWEST_SPECTRUM && env->P.k_paths > 8)
        return fail_arg(env, "this policy supports at most 8 candidate routes", ONGYM_E_LIMIT);
    if (policy != ONGYM_POLICY_FIRST_FIT && env->P.n_mods_consider < env->P.n_mods)
        return fail_arg(env, "only the first-fit policy is fused for modulations_to_consider < n_mods", ONGYM_E_LIMIT);
    if (policy == ONGYM_POLICY_MSCL && (env->P.bit_rate_mode != 0 || env->P.n_bit_rates <= 0))
        return fail_arg(env, "the MSCL policy sums its capacity loss over the discrete bit rates: bit_rate_mode must be discrete", ONGYM_E_LIMIT);
    if (nsteps <= 0) return fail_arg(env, "nsteps must be positive");
    if (!env->has_source) { env->err = "no request source: call ongym_seed or ongym_set_requests first"; return ONGYM_E_STATE; }
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    int rc;
    if (out && !env->cfg.io_device) {
        size_t n = (size_t)nsteps * env->P.batch;
        if ((rc = ensure_out(env, n))) return rc;
        if ((rc = launch_run(env, kModePolicyStep, policy, nsteps, nullptr, nullptr, nullptr, env->d_out))) return rc;
        HIP_TRY(env, hipMemcpyAsync(out, env->d_out, n * sizeof(ongym_step_rec), hipMemcpyDeviceToHost, env->stream));
        HIP_TRY(env, hipStreamSynchronize(env->stream));
        return ONGYM_OK;
    }
    return launch_run(env, kModePolicyStep, policy, nsteps, nullptr, nullptr, nullptr, out);
}

int ongym_step_actions(ongym_env *env, const int32_t *actions, ongym_step_rec *out) {
    if (!env || !actions) return env ? fail_arg(env, "null actions") : ONGYM_E_ARG;
    if (!env->has_source) { env->err = "no request source: call ongym_seed or ongym_set_requests first"; return ONGYM_E_STATE; }
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    int rc;
    if (env->cfg.io_device) return launch_run(env, kModeActionStep, ONGYM_POLICY_FIRST_FIT, 1, actions, nullptr, nullptr, out);
    HIP_TRY(env, hipMemcpyAsync(env->d_actions, actions, (size_t)env->P.batch * 4, hipMemcpyHostToDevice, env->stream));
    if (out) {
        if ((rc = ensure_out(env, (size_t)env->P.batch))) return rc;
        if ((rc = launch_run(env, kModeActionStep, ONGYM_POLICY_FIRST_FIT, 1, env->d_actions, nullptr, nullptr, env->d_out))) return rc;
        HIP_TRY(env, hipMemcpyAsync(out, env->d_out, (size_t)env->P.batch * sizeof(ongym_step_rec), hipMemcpyDeviceToHost, env->stream));
    } else if ((rc = launch_run(env, kModeActionStep, ONGYM_POLICY_FIRST_FIT, 1, env->d_actions, nullptr, nullptr, nullptr))) return rc;
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    return ONGYM_OK;
}

static int check_policy(ongym_env *env, int32_t policy) {
    if (policy < ONGYM_POLICY_FIRST_FIT || policy >= ONGYM_POLICY_COUNT) return fail_arg(env, "unknown policy id");
    if (policy >= ONGYM_POLICY_LOWEST_SPECTRUM && env->P.k_paths > 8)
        return fail_arg(env, "this policy supports at most 8 candidate routes", ONGYM_E_LIMIT);
    if (policy != ONGYM_POLICY_FIRST_FIT && env->P.n_mods_consider < env->P.n_mods)
        return fail_arg(env, "only the first-fit policy is fused for modulations_to_consider < n_mods", ONGYM_E_LIMIT);
    if (policy == ONGYM_POLICY_MSCL && (env->P.bit_rate_mode != 0 || env->P.n_bit_rates <= 0))
        return fail_arg(env, "the MSCL policy sums its capacity loss over the discrete bit rates: bit_rate_mode must be discrete", ONGYM_E_LIMIT);
    return 0;
}

int ongym_step_actions_bundle(ongym_env *env, const int32_t *actions, int32_t next_policy, ongym_step_rec *rec_out,
                              ongym_request *request_out, ongym_stats *stats_out, int32_t *next_actions, uint8_t *next_flags) {
    if (!env || !actions || !rec_out || !request_out || !stats_out) return env ? fail_arg(env, "null buffer") : ONGYM_E_ARG;
    if (env->cfg.io_device) return fail_arg(env, "ongym_step_actions_bundle returns host buffers: not with io_device", ONGYM_E_STATE);
    if (next_policy >= 0 && (!next_actions || !next_flags)) return fail_arg(env, "null next_actions / next_flags");
    if (!env->has_source) { env->err = "no request source: call ongym_seed or ongym_set_requests first"; return ONGYM_E_STATE; }
    int rc;
    if (next_policy >= 0 && (rc = check_policy(env, next_policy))) return rc;
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    const size_t B = (size_t)env->P.batch;
    // pinned staging: records | DevEnv (requests + statistics) | next actions | next flags | actions in
    const size_t o_rec = 0, o_env = o_rec + B * sizeof(ongym_step_rec), o_act = o_env + B * sizeof(DevEnv),
                 o_flag = o_act + B * sizeof(int32_t), o_in = o_flag + ((B + 15) & ~(size_t)15), total = o_in + B * sizeof(int32_t);
    if (env->h_pinned_bytes < total) {
        if (env->h_pinned) { (void)hipHostFree(env->h_pinned); env->h_pinned = nullptr; env->h_pinned_bytes = 0; }
        HIP_TRY(env, hipHostMalloc(&env->h_pinned, total, hipHostMallocDefault));
        env->h_pinned_bytes = total;
    }
    char *hp = static_cast<char *>(env->h_pinned);
    if (B <= 256) {
        // few replicas (the single-environment surface): the kernels read the actions from and write their results to the
        // pinned host buffer directly (it is device-accessible): launches and one synchronisation, no copy calls for them
        int32_t *h_act_in = reinterpret_cast<int32_t *>(hp + o_in);
        memcpy(h_act_in, actions, B * 4);
        if (next_policy >= 0)
            rc = launch_run(env, kModeActionThenPolicy, next_policy, 2, h_act_in, reinterpret_cast<int32_t *>(hp + o_act),
                            reinterpret_cast<uint8_t *>(hp + o_flag), reinterpret_cast<ongym_step_rec *>(hp + o_rec));
        else
            rc = launch_run(env, kModeActionStep, ONGYM_POLICY_FIRST_FIT, 1, h_act_in, nullptr, nullptr,
                            reinterpret_cast<ongym_step_rec *>(hp + o_rec));
        if (rc) return rc;
        HIP_TRY(env, hipMemcpyAsync(hp + o_env, env->P.env, B * sizeof(DevEnv), hipMemcpyDeviceToHost, env->stream));
    } else {
        if ((rc = ensure_out(env, B))) return rc;
        HIP_TRY(env, hipMemcpyAsync(env->d_actions, actions, B * 4, hipMemcpyHostToDevice, env->stream));
        if ((rc = launch_run(env, kModeActionStep, ONGYM_POLICY_FIRST_FIT, 1, env->d_actions, nullptr, nullptr, env->d_out))) return rc;
        if (next_policy >= 0 && (rc = launch_run(env, kModePolicyOnly, next_policy, 1, nullptr, env->d_act_out, env->d_flag_out, nullptr))) return rc;
        HIP_TRY(env, hipMemcpyAsync(hp + o_rec, env->d_out, B * sizeof(ongym_step_rec), hipMemcpyDeviceToHost, env->stream));
        HIP_TRY(env, hipMemcpyAsync(hp + o_env, env->P.env, B * sizeof(DevEnv), hipMemcpyDeviceToHost, env->stream));
        if (next_policy >= 0) {
            HIP_TRY(env, hipMemcpyAsync(hp + o_act, env->d_act_out, B * 4, hipMemcpyDeviceToHost, env->stream));
            HIP_TRY(env, hipMemcpyAsync(hp + o_flag, env->d_flag_out, B, hipMemcpyDeviceToHost, env->stream));
        }
    }
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    memcpy(rec_out, hp + o_rec, B * sizeof(ongym_step_rec));
    const DevEnv *de = reinterpret_cast<const DevEnv *>(hp + o_env);
    int flags = 0;
    for (size_t r = 0; r < B; r++) {
        stats_out[r] = de[r].st; flags |= de[r].st.flags;
        ongym_request &q = request_out[r];
        memset(&q, 0, sizeof(q));
        q.arrival_time = de[r].cur_at; q.holding_time = de[r].cur_ht; q.bit_rate = de[r].cur_br;
        q.source = (int16_t)de[r].cur_src; q.destination = (int16_t)de[r].cur_dst;
    }
    if (next_policy >= 0) { memcpy(next_actions, hp + o_act, B * 4); memcpy(next_flags, hp + o_flag, B); }
    if (flags & ONGYM_F_OVERFLOW) { env->err = "a replica overflowed its service table (raise capacity)"; return ONGYM_E_CAPACITY; }
    return ONGYM_OK;
}

int ongym_observe(ongym_env *env, float *obs, uint8_t *mask) {
    if (!env || !obs || !mask) return env ? fail_arg(env, "null obs/mask") : ONGYM_E_ARG;
    const Params &P = env->P;
    if (!P.path_len_norm || !(P.max_bit_rate > 0)) return fail_arg(env, "observation needs path_len_norm and max_bit_rate = max(bit_rates)");
    if (std::fabs(P.slot_bw - P.channel_width * 1e9) > 1e-6 * P.slot_bw) return fail_arg(env, "observation needs slot_bandwidth == channel_width*1e9");
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    const size_t obs_dim = 3 + P.k_paths + (size_t)P.k_paths * P.n_mods_consider * 12;
    const size_t nact = (size_t)P.k_paths * P.n_mods_consider * P.n_slots + 1;
    const size_t B = (size_t)P.batch;
    const size_t lds = observe_lds(env);
    if (lds > 64 * 1024) {
        HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_observe<true, true>), lds));
        HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_observe<true, false>), lds));
        HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_observe<false, true>), lds));
        HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(&k_observe<false, false>), lds));
    }
    float *d_obs = obs; uint8_t *d_mask = mask;
    if (!env->cfg.io_device) {
        if (!env->d_obs) {
            HIP_TRY(env, hipMalloc(reinterpret_cast<void **>(&env->d_obs), B * obs_dim * sizeof(float)));
            env->allocs.push_back(env->d_obs);
            HIP_TRY(env, hipMalloc(reinterpret_cast<void **>(&env->d_obsmask), B * nact));
            env->allocs.push_back(env->d_obsmask);
        }
        d_obs = env->d_obs; d_mask = env->d_obsmask;
    }
    HIP_TRY(env, hipMemsetAsync(d_mask, 0, B * nact, env->stream));     // k_observe only sets the ones
    HIP_TRY(env, hipEventRecord(env->ev0, env->stream));
#define ONGYM_LAUNCH_OBS(UA, R) hipLaunchKernelGGL((k_observe<UA, R>), dim3(P.batch), dim3(64), lds, env->stream, env->d_P, d_obs, d_mask)
    if (P.uniform_alpha) { if (P.rec32) ONGYM_LAUNCH_OBS(true, true); else ONGYM_LAUNCH_OBS(true, false); }
    else { if (P.rec32) ONGYM_LAUNCH_OBS(false, true); else ONGYM_LAUNCH_OBS(false, false); }
#undef ONGYM_LAUNCH_OBS
    HIP_TRY(env, hipGetLastError());
    HIP_TRY(env, hipEventRecord(env->ev1, env->stream));
    env->timed = true;
    if (!env->cfg.io_device) {
        HIP_TRY(env, hipMemcpyAsync(obs, d_obs, B * obs_dim * sizeof(float), hipMemcpyDeviceToHost, env->stream));
        HIP_TRY(env, hipMemcpyAsync(mask, d_mask, B * nact, hipMemcpyDeviceToHost, env->stream));
        HIP_TRY(env, hipStreamSynchronize(env->stream));
    }
    return ONGYM_OK;
}

int ongym_sample_actions(ongym_env *env, const uint8_t *mask, uint64_t seed, uint64_t draw_index, int32_t *actions) {
    if (!env || !mask || !actions) return env ? fail_arg(env, "null mask/actions") : ONGYM_E_ARG;
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    const Params &P = env->P;
    const size_t nact = (size_t)P.k_paths * P.n_mods_consider * P.n_slots + 1, B = (size_t)P.batch;
    const uint8_t *d_mask = mask;
    int32_t *d_act = actions;
    if (!env->cfg.io_device) {
        if (!env->d_obsmask) {
            HIP_TRY(env, hipMalloc(reinterpret_cast<void **>(&env->d_obsmask), B * nact));
            env->allocs.push_back(env->d_obsmask);
        }
        HIP_TRY(env, hipMemcpyAsync(env->d_obsmask, mask, B * nact, hipMemcpyHostToDevice, env->stream));
        d_mask = env->d_obsmask; d_act = env->d_act_out;
    }
    hipLaunchKernelGGL(k_sample_mask, dim3(P.batch), dim3(64), 0, env->stream, d_mask, (long long)nact, seed, env->replica_base,
                       draw_index, d_act);
    HIP_TRY(env, hipGetLastError());
    if (!env->cfg.io_device) {
        HIP_TRY(env, hipMemcpyAsync(actions, d_act, B * 4, hipMemcpyDeviceToHost, env->stream));
        HIP_TRY(env, hipStreamSynchronize(env->stream));
    }
    return ONGYM_OK;
}

int ongym_policy_actions(ongym_env *env, int32_t policy, int32_t *actions, uint8_t *flags) {
    if (!env || !actions) return env ? fail_arg(env, "null actions") : ONGYM_E_ARG;
    if (policy < ONGYM_POLICY_FIRST_FIT || policy >= ONGYM_POLICY_COUNT) return fail_arg(env, "unknown policy id");
    if (policy >= ONGYM_POLICY_LOWEST_SPECTRUM && env->P.k_paths > 8)
        return fail_arg(env, "this policy supports at most 8 candidate routes", ONGYM_E_LIMIT);
    if (policy != ONGYM_POLICY_FIRST_FIT && env->P.n_mods_consider < env->P.n_mods)
        return fail_arg(env, "only the first-fit policy is fused for modulations_to_consider < n_mods", ONGYM_E_LIMIT);
    if (policy == ONGYM_POLICY_MSCL && (env->P.bit_rate_mode != 0 || env->P.n_bit_rates <= 0))
        return fail_arg(env, "the MSCL policy sums its capacity loss over the discrete bit rates: bit_rate_mode must be discrete", ONGYM_E_LIMIT);
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    int rc;
    if (env->cfg.io_device) return launch_run(env, kModePolicyOnly, policy, 1, nullptr, actions, flags, nullptr);
    if ((rc = launch_run(env, kModePolicyOnly, policy, 1, nullptr, env->d_act_out, env->d_flag_out, nullptr))) return rc;
    HIP_TRY(env, hipMemcpyAsync(actions, env->d_act_out, (size_t)env->P.batch * 4, hipMemcpyDeviceToHost, env->stream));
    if (flags) HIP_TRY(env, hipMemcpyAsync(flags, env->d_flag_out, (size_t)env->P.batch, hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    return ONGYM_OK;
}

static int query(ongym_env *env, int what, int replica, int path, int slot, int n) {
    if (replica < 0 || replica >= env->P.batch) return fail_arg(env, "replica out of range");
    if ((what == kQAvailable || what == kQGsnr || what == kQPathFree) && (path < 0 || path >= env->P.n_paths)) return fail_arg(env, "path id out of range");
    if (what == kQGsnr && (slot < 0 || n <= 0 || slot + n > env->P.n_slots)) return fail_arg(env, "slot range out of the grid");
    if (what == kQPathFree && (slot < 0 || n <= 0 || slot >= env->P.n_slots || n > 1023)) return fail_arg(env, "slot / nslots out of range");
    if (what == kQCandidates && (path <= 0 || path > 1023 || n <= 0 || n > 1023)) return fail_arg(env, "total_slots / nslots out of range");
    HIP_TRY(env, hipSetDevice(env->cfg.device));
#define ONGYM_LAUNCH_Q(UA, R)                                                                                       \
    hipLaunchKernelGGL((k_query<UA, R>), dim3(1), dim3(64), env->lds, env->stream, env->d_P, what, replica, path,   \
                       slot, n, env->d_scratch_i, env->d_scratch_d)
    if (env->P.uniform_alpha) { if (env->P.rec32) ONGYM_LAUNCH_Q(true, true); else ONGYM_LAUNCH_Q(true, false); }
    else { if (env->P.rec32) ONGYM_LAUNCH_Q(false, true); else ONGYM_LAUNCH_Q(false, false); }
#undef ONGYM_LAUNCH_Q
    HIP_TRY(env, hipGetLastError());
    return 0;
}

int ongym_query_available(ongym_env *env, int32_t replica, int32_t path_id, int32_t *out) {
    if (!env || !out) return ONGYM_E_ARG;
    int rc = query(env, kQAvailable, replica, path_id, 0, 0);
    if (rc) return rc;
    HIP_TRY(env, hipMemcpyAsync(out, env->d_scratch_i, (size_t)env->P.n_slots * 4, hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    return ONGYM_OK;
}

int ongym_query_gsnr(ongym_env *env, int32_t replica, int32_t path_id, int32_t slot, int32_t nslots, double out[3]) {
    if (!env || !out) return ONGYM_E_ARG;
    int rc = query(env, kQGsnr, replica, path_id, slot, nslots);
    if (rc) return rc;
    HIP_TRY(env, hipMemcpyAsync(out, env->d_scratch_d, 3 * sizeof(double), hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    return ONGYM_OK;
}

int ongym_query_gsnr_many(ongym_env *env, int32_t replica, int32_t count, const int32_t *cands, double *out) {
    if (!env || (count > 0 && (!cands || !out))) return ONGYM_E_ARG;
    if (count < 0) return fail_arg(env, "negative candidate count");
    if (count == 0) return ONGYM_OK;
    if (replica < 0 || replica >= env->P.batch) return fail_arg(env, "replica out of range");
    for (int32_t i = 0; i < count; i++) {    // every operand is checked on the host before the launch
        const int32_t path = cands[3 * i], slot = cands[3 * i + 1], n = cands[3 * i + 2];
        if (path < 0 || path >= env->P.n_paths) return fail_arg(env, "path id out of range");
        if (slot < 0 || n <= 0 || slot + n > env->P.n_slots) return fail_arg(env, "slot range out of the grid");
    }
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    int32_t *d_c = nullptr;
    double *d_o = nullptr;
    HIP_TRY(env, hipMalloc(reinterpret_cast<void **>(&d_c), (size_t)count * 3 * sizeof(int32_t)));
    if (hipMalloc(reinterpret_cast<void **>(&d_o), (size_t)count * 3 * sizeof(double)) != hipSuccess) {
        (void)hipFree(d_c);
        return fail_arg(env, "hipMalloc failed", ONGYM_E_HIP);
    }
    hipError_t e = hipMemcpyAsync(d_c, cands, (size_t)count * 3 * sizeof(int32_t), hipMemcpyHostToDevice, env->stream);
#define ONGYM_LAUNCH_QM(UA, R)                                                                                      \
    hipLaunchKernelGGL((k_query_gsnr_many<UA, R>), dim3(count), dim3(64), env->lds, env->stream, env->d_P, replica, \
                       count, d_c, d_o)
    if (e == hipSuccess) {
        if (env->P.uniform_alpha) { if (env->P.rec32) ONGYM_LAUNCH_QM(true, true); else ONGYM_LAUNCH_QM(true, false); }
        else { if (env->P.rec32) ONGYM_LAUNCH_QM(false, true); else ONGYM_LAUNCH_QM(false, false); }
        e = hipGetLastError();
    }
#undef ONGYM_LAUNCH_QM
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, (size_t)count * 3 * sizeof(double), hipMemcpyDeviceToHost, env->stream);
    hipError_t e2 = hipStreamSynchronize(env->stream);
    (void)hipFree(d_c);
    (void)hipFree(d_o);
    if (e != hipSuccess || e2 != hipSuccess) {
        env->err = std::string("ongym_query_gsnr_many: ") + hipGetErrorString(e != hipSuccess ? e : e2);
        return ONGYM_E_HIP;
    }
    return ONGYM_OK;
}

int ongym_query_candidates(ongym_env *env, const int32_t *row, int32_t total_slots, int32_t nslots,
                           int32_t *starts_out, int32_t *count) {
    if (!env || !row || !starts_out || !count) return ONGYM_E_ARG;
    if (total_slots <= 0 || total_slots > 1023) return fail_arg(env, "total_slots out of range");
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    HIP_TRY(env, hipMemcpyAsync(env->d_scratch_i, row, (size_t)total_slots * 4, hipMemcpyHostToDevice, env->stream));
    int rc = query(env, kQCandidates, 0, total_slots, 0, nslots);
    if (rc) return rc;
    std::vector<int32_t> flags((size_t)total_slots);
    HIP_TRY(env, hipMemcpyAsync(flags.data(), env->d_scratch_i + 1024, (size_t)total_slots * 4, hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    int32_t k = 0;
    for (int32_t s = 0; s < total_slots; s++) if (flags[s]) starts_out[k++] = s;   // flag -> list, no arithmetic
    *count = k;
    return ONGYM_OK;
}

int ongym_query_path_free(ongym_env *env, int32_t replica, int32_t path_id, int32_t slot, int32_t nslots, int32_t *out) {
    if (!env || !out) return ONGYM_E_ARG;
    int rc = query(env, kQPathFree, replica, path_id, slot, nslots);
    if (rc) return rc;
    HIP_TRY(env, hipMemcpyAsync(out, env->d_scratch_i, 4, hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    return ONGYM_OK;
}

int ongym_query_moves(ongym_env *env, int32_t replica, ongym_move *out, int32_t *count) {
    if (!env || !out || !count) return ONGYM_E_ARG;
    if (replica < 0 || replica >= env->P.batch) return fail_arg(env, "replica out of range");
    *count = 0;
    if (!env->P.defragmentation) return ONGYM_OK;
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    int32_t n = 0;
    HIP_TRY(env, hipMemcpy(&n, env->P.move_n + replica, sizeof(n), hipMemcpyDeviceToHost));
    const int32_t kept = n < ONGYM_MOVE_LOG ? n : ONGYM_MOVE_LOG;
    if (kept > 0)
        HIP_TRY(env, hipMemcpy(out, env->P.move_log + (size_t)replica * ONGYM_MOVE_LOG, (size_t)kept * sizeof(ongym_move),
                               hipMemcpyDeviceToHost));
    *count = n;
    return ONGYM_OK;
}

int ongym_query_grid(ongym_env *env, int32_t replica, int32_t *out) {
    if (!env || !out) return ONGYM_E_ARG;
    int rc = query(env, kQGrid, replica, 0, 0, 0);
    if (rc) return rc;
    HIP_TRY(env, hipMemcpyAsync(out, env->d_scratch_i, (size_t)env->P.n_links * env->P.n_slots * 4, hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    return ONGYM_OK;
}

int ongym_query_services(ongym_env *env, int32_t replica, ongym_service *out, int32_t *n) {
    if (!env || !out || !n) return ONGYM_E_ARG;
    int rc = query(env, kQServices, replica, 0, 0, 0);
    if (rc) return rc;
    int32_t head[2];
    HIP_TRY(env, hipMemcpyAsync(head, env->d_scratch_i, 8, hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    *n = head[0];
    if (head[0] > 0) HIP_TRY(env, hipMemcpy(out, env->d_scratch_i + 2, (size_t)head[0] * sizeof(ongym_service), hipMemcpyDeviceToHost));
    return ONGYM_OK;
}

int ongym_query_request(ongym_env *env, int32_t replica, ongym_request *out) {
    if (!env || !out) return ONGYM_E_ARG;
    int rc = query(env, kQRequest, replica, 0, 0, 0);
    if (rc) return rc;
    HIP_TRY(env, hipMemcpyAsync(out, env->d_scratch_i, sizeof(ongym_request), hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    return ONGYM_OK;
}

int ongym_stats_get(ongym_env *env, ongym_stats *out) {
    if (!env || !out) return ONGYM_E_ARG;
    HIP_TRY(env, hipSetDevice(env->cfg.device));
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    std::vector<DevEnv> host((size_t)env->P.batch);
    HIP_TRY(env, hipMemcpy(host.data(), env->P.env, host.size() * sizeof(DevEnv), hipMemcpyDeviceToHost));
    int flags = 0;
    for (size_t r = 0; r < host.size(); r++) { out[r] = host[r].st; flags |= host[r].st.flags; }
    if (flags & ONGYM_F_OVERFLOW) { env->err = "a replica overflowed its service table (raise capacity)"; return ONGYM_E_CAPACITY; }
    return ONGYM_OK;
}

#ifdef ONGYM_STAMPS
// diagnostic build only: per-phase shader-cycle sums (not part of the ABI)
int ongym_debug_stamps(ongym_env *env, unsigned long long *out16) {
    HIP_TRY(env, hipStreamSynchronize(env->stream));
    HIP_TRY(env, hipMemcpy(out16, env->P.dbg, 16 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(env, hipMemset(env->P.dbg, 0, 16 * 8));
    return 0;
}
#endif

}  // extern "C"
