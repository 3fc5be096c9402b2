// ongym_device.hpp — device-side data layout and kernels of the batched QRMSA environment (gfx950 / CDNA4).
//
// Mapping: ONE WAVEFRONT (64 lanes) PER REPLICA, one 64-thread workgroup each, so there is no inter-wave
// synchronisation anywhere. A launch loads the replica's mutable state from HBM into LDS once, runs `nsteps` requests
// against it and writes it back; with nsteps = episode length the state never leaves the CU for a whole episode.
//
// Per-replica state in HBM (coalesced: lane i touches word i of a contiguous per-replica block):
//   occ   u64 [B][E][W]   free-slot bitmap, bit j of word w = slot 64w+j is FREE (the reference keeps int32[E][S] with
//                         1 = free, envs/qrmsa.pyx:306-309); bits >= S are 0
//   svc_a u32 [B][C]      running services, unordered: path_id | slot<<16
//   svc_b u32 [B][C]                                   nslots | modulation<<16
//   svc_r f32 [B][C]      release time = float32(arrival+holding) (heap key as compared at envs/qrmsa.pyx:1114-1115)
//   env   DevEnv [B]      clocks, counters, current request, RNG position
// Constant tables (shared by all replicas, L2/L1 resident): pair->path ids, path->links, path link-masks, GN weights.
//
// Wave-uniform bookkeeping (counters, clocks, request draw) is done by lane 0 on the LDS copy of DevEnv; everything
// that touches the slot grid or the service table is wave-parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ongym.h"
#include "../../include/ongym_traffic.h"

namespace ongym {

constexpr int kWave = 64;
constexpr int kMaxMods = 8;
constexpr int kMaxRowWords = 16;   // n_slots <= 1023 (+1 virtual guard bit)
constexpr int kMaxLinks = 128;     // two 64-bit link-mask words per path
constexpr int kMaxHops = 64;       // one lane per hop

enum RunMode { kModePolicyStep = 0, kModeActionStep = 1, kModePolicyOnly = 2,
               kModeActionThenPolicy = 3 };   // one launch: step(actions), then the policy alone on the new request (nsteps = 2)
enum ReqMode { kReqNone = 0, kReqRng = 1, kReqTrace = 2 };

struct DevEnv {
    double launch_power, margin, mean_iat;   // per-replica parameters (sweeps as a batch dimension)
    float mean_iat_f, pad1;                  // (float)mean_iat: the sampler works in float32
    uint64_t rng_key;
    uint64_t req_index;                      // requests drawn so far (rng counter / trace cursor)
    float cur_at, cur_ht, cur_br;            // current_service (C floats in the reference, envs/qrmsa.pyx:35-37)
    int32_t cur_src, cur_dst, cur_id;
    int32_t have_request;                    // _new_service (envs/qrmsa.pyx:1077-1078,1102)
    int32_t pad0;
    // sum of Service.OSNR of the running episode = osnr_flushed - 10 log10(osnr_prod): the product takes one multiply per
    // accepted service and one log10 per ~60 of them; both survive across launches so results do not depend on how
    // the steps are partitioned into launches. st.episode_osnr_sum is refreshed from them at every store.
    double osnr_flushed, osnr_prod;
    // len(topology.graph["services"]) - episode_services_processed: 0 unless reset(options={"only_episode_counters": True})
    // (qrmsa.pyx:461-464) zeroed the episode counter while the list of the episode's services kept growing; the JOCN
    // drivers average Service.OSNR over that list (graph_load.py:181-185)
    int64_t svc_list_extra;
    ongym_stats st;                          // LAST: its tail (the terminal-step snapshot) never enters LDS
};
// The kernels keep DevEnv up to (not including) st.last_episode_processed in LDS; the snapshot fields behind it are
// written to memory directly by snapshot_terminal.
constexpr size_t kEnvHotBytes = (offsetof(DevEnv, st) + offsetof(ongym_stats, last_episode_processed) + 7) & ~(size_t)7;
static_assert(offsetof(DevEnv, st) % 8 == 0, "DevEnv.st must be 8-byte aligned");

struct Params {
    int n_nodes, n_links, n_paths, k_paths, max_hops, n_mods, n_slots;
    int n_mods_consider;   // modulations_to_consider (envs/qrmsa.pyx:313), 1..n_mods: width of the action codec's format window
    int row_words;   // W  = ceil(S/64): words per link row in occ
    int ext_words;   // Wx = S/64 + 1  : words of the row extended by the virtual free slot S
    int batch, capacity, episode_length, auto_reset;
    int bit_rate_mode, n_bit_rates, br_lo, br_hi;
    int measure_disruptions;
    int defragmentation, n_defrag_services;   // envs/qrmsa.pyx:233-234
    int track_ids;      // Service.service_id (and OSNR) kept per record: defragmentation or cfg.track_service_ids
    int uniform_alpha;
    int rec32;          // record codec R32 in use (n_links <= 32, n_paths <= 512)
    int ase_shortcut;   // 1: every interferer term of the NLI sum is provably >= 0 (checked on the host at create)
    int req_mode;
    double f0, slot_bw, channel_width, mean_holding;
    double nslots_width;            // what get_number_slots divides by: channel_width, or the band's width when `bands` is set (quirk Q9)
    float mean_holding_f, pad_f;    // (float)mean_holding
    const int32_t *nreq_tab;        // [n_bit_rates*8] slots needed per (discrete bit rate, modulation)
    const double *req_coef;         // [n_bit_rates*8][2] (nli_coef[n], self_asinh[n]) of that slot count (0 if n is not in [1, S])
    double alpha0_cl;               // pi^2 |beta2| / (2 alpha) when alpha is uniform
    // constant tables
    const int32_t *pair_paths;      // [N*N*K]
    const int32_t *path_hops;       // [P]
    const int32_t *path_links;      // [P*max_hops]
    const uint64_t *path_mask;      // [P*2]  bit e set iff link e on path
    const double *path_ase;         // [P]    h * sum_l nspans_l (exp(2 a_l L_l) - 1) nf_l
    const double *link_w1;          // [E]    nspans * l_eff
    const double *link_w2;          // [E]    nspans * l_eff * l_eff / (L*1e3)
    const double *link_cl;          // [E]    pi^2 |beta2| * l_eff_a = pi^2 |beta2| / (2 alpha)
    const double *link_selfc;       // [E]    pi^2 |beta2| / (4 alpha)
    const double *path_w1;          // [P]    sum_l w1_l over the path's links (uniform-alpha self term)
    const double *self_asinh;       // [S+1]  asinh(pi^2 |b2| (slot_bw n)^2 / (4 alpha)) when alpha is uniform
    const double *nli_coef;         // [S+2]  8/(27 pi |b2|) gamma^2 / (slot_bw n)^2   (P_nli/P = nli_coef[n] * P^2 * sum)
    const double2 *pair_tab;        // [tab_nmax][2S+1] (asinh difference, Bk/|df|) by (interferer slots, centre distance in
                                    //  half slots), uniform alpha only; NULL = always compute
    int tab_nmax, tab_stride;
    // lean first-fit kernel (ongym_fast.hpp); NULL when the configuration is not eligible
    const void *path_rec;           // PathRec [P]
    const uint64_t *path_hash_keys; // lean M64 record codec: link mask (bits 0..40) -> path id, open addressing, 2^path_hash_bits slots
    const int32_t *path_hash_vals;
    int path_hash_bits, pad_hash;
    const double *pair_tab2k;       // the pair table with a row pitch of 2048 entries: [tab_nmax][2048][2]
    const double *pair_tabp;        // the same rows split by the parity of the distance: [tab_nmax][2][1024][2], entry (d & 1, d >> 1).
                                    // Candidate centres of one format are two half slots apart: consecutive candidates then
                                    // read consecutive 16-byte entries (full cache lines) instead of every other one
    const double *bit_rates, *bit_rate_cum, *node_cum;
    const double *path_len_norm;    // [P] observation(): (length - min link) / (max link - min link)
    const double *plogp;            // [S+1] (n/S)*log(n/S), n = 1..S: the Shannon-entropy terms of utils.pyx:61-79 (ongym_scored.hpp)
    double max_bit_rate;
    int mod_se[kMaxMods];
    double mod_thr[kMaxMods];
    double mod_phi53[kMaxMods];     // Phi_mod[se-1] * 5/3  (core/osnr.pyx:38-41,86-92)
    // mutable state
    uint64_t *occ;
    uint32_t *svc_a, *svc_b;
    float *svc_r;
    uint32_t *svc_q;   // Service.service_id per record  } only when track_ids
    double *svc_o;     // Service.OSNR per record        }
    ongym_move *move_log;   // [batch][ONGYM_MOVE_LOG] reallocations of the last step   } only when track_ids
    int32_t *move_n;        // [batch] their count                                       }
    DevEnv *env;
    // request trace (req_mode == kReqTrace)
    const ongym_request *trace;
    long long trace_n;
    unsigned long long *dbg;   // diagnostic build only (-DONGYM_STAMPS): per-phase cycle sums
};

// ---------------------------------------------------------------------------------------------------------------
// LDS carve-up (dynamic shared memory), 8-byte aligned pieces first
//   DevEnv | lim f64[8] | rp f64[2] | phi f64[8] | nlic f64[8] | selfa f64[8] | nreq i32[8] |      (fixed size, constant addresses)
//   occ u64[E*W] | lw f64[2E] (w1, w2 interleaved) | lcl f64[E] | lsc f64[E] | sa u32[C] | sb u32[C] | sr f32[C] | list u16[C] |
//   (lim0 f64[8] when measure_disruptions or track_ids) | (so f64[C] | sq u32[C] when track_ids)
// (DevEnv = its first kEnvHotBytes)
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t lds_bytes(int n_links, int row_words, int capacity, int uniform_alpha,
                                            int measure_disruptions, int track_ids) {
    size_t b = (size_t)n_links * row_words * 8 + (size_t)n_links * (uniform_alpha ? 16 : 32) + kEnvHotBytes;
    b += (size_t)capacity * 12 + 64 + 16 + 64 + 128 + 32 + (size_t)capacity * 2;
    b += (measure_disruptions || track_ids) ? 64 : 0;
    b += track_ids ? (size_t)capacity * 12 : 0;
    return (b + 15) & ~(size_t)15;
}

__host__ __device__ inline size_t lds_bytes(const Params &P) {
    return lds_bytes(P.n_links, P.row_words, P.capacity, P.uniform_alpha, P.measure_disruptions, P.track_ids);
}

#ifdef ONGYM_STAMPS
#define ONGYM_NSTAMPS 16
#define STAMP(c, idx)                                                         \
    do {                                                                      \
        unsigned long long _t = __builtin_amdgcn_s_memtime();                 \
        (c).stamp_acc[idx] += _t - (c).stamp_last;                            \
        (c).stamp_last = _t;                                                  \
    } while (0)
#define STAMPW(c, idx) do { __builtin_amdgcn_s_waitcnt(0); STAMP(c, idx); } while (0)   /* drain, then stamp */
#else
#define STAMP(c, idx) do { } while (0)
#define STAMPW(c, idx) do { } while (0)
#endif

// extra LDS of the kernels that evaluate EVERY candidate of a path (observation, highest-SNR policy)
//   Fx f64[2S+2] interferer field at half-slot centres | xlist u16[2S+2] needed centres, then candidate starts |
//   Vw u64[n_mods][vs] valid-start words per format (word per lane) | needx u8[2S+2] "centre needed" flags
struct FieldLds { double *Fx; uint64_t *Vw; uint16_t *xlist; uint8_t *needx; int vs; };

// LDS block of k_observe (byte offsets from the start of the workgroup's LDS; host and device use the same function).
// The observation never reads the release times `sr` nor, outside build_field, the interferer index list `list`:
//   compact (list is the LAST array of the state block, i.e. no id / disruption arrays, and `sr` is large enough):
//       Vw (stride ext_words) and needx live in `sr`; `list` OVERLAYS needx - build_field compacts the needed centres before it
//       builds the list - and Fx, xlist start where `list` was: 14 636 -> 13 740 B for NSFNET-320 / C = 448, i.e. 11 instead of
//       10 resident blocks per CU (LDS is allocated in 1 280-byte granules)
//   alias: Vw (stride 16) and needx in `sr`, `list` stays, Fx and xlist after the state block
//   plain: everything after the state block
struct ObsLayout { size_t fx, xlist, vw, needx, list, total; int vs; };
__host__ __device__ inline size_t lds_list_offset(const Params &P) {
    return (size_t)P.n_links * P.row_words * 8 + (size_t)P.n_links * (P.uniform_alpha ? 16 : 32) + kEnvHotBytes +
           (64 + 16 + 64 + 128 + 32) + (size_t)P.capacity * 12;
}
__host__ __device__ inline ObsLayout obs_layout(const Params &P) {
    ObsLayout o;
    const size_t nxb = (size_t)2 * P.n_slots + 2;
    const size_t sr = lds_list_offset(P) - (size_t)P.capacity * 4, cap4 = (size_t)P.capacity * 4;
    const size_t vw_c = (size_t)P.n_mods * P.ext_words * 8, needx_b = (nxb + 7) & ~(size_t)7, list_b = (size_t)P.capacity * 2;
    const bool list_last = !P.track_ids && !P.measure_disruptions;
    if (list_last && cap4 >= vw_c + (needx_b > list_b ? needx_b : list_b)) {
        o.vs = P.ext_words; o.vw = sr; o.needx = sr + vw_c; o.list = o.needx;
        o.fx = (lds_list_offset(P) + 15) & ~(size_t)15;
        o.xlist = o.fx + nxb * 8;
        o.total = o.xlist + nxb * 2;
    } else if (cap4 >= (size_t)kMaxMods * kMaxRowWords * 8 + nxb + 8) {
        o.vs = kMaxRowWords; o.vw = (sr + 7) & ~(size_t)7; o.needx = o.vw + (size_t)kMaxMods * kMaxRowWords * 8; o.list = lds_list_offset(P);
        o.fx = lds_bytes(P); o.xlist = o.fx + nxb * 8;
        o.total = o.xlist + nxb * 2;
    } else {
        o.vs = kMaxRowWords; o.list = lds_list_offset(P);
        o.fx = lds_bytes(P); o.xlist = o.fx + nxb * 8;
        o.vw = (o.xlist + nxb * 2 + 7) & ~(size_t)7; o.needx = o.vw + (size_t)kMaxMods * kMaxRowWords * 8;
        o.total = o.needx + nxb;
    }
    o.total = (o.total + 15) & ~(size_t)15;
    return o;
}

struct Ctx {
#ifdef ONGYM_STAMPS
    unsigned long long stamp_acc[ONGYM_NSTAMPS];
    unsigned long long stamp_last;
#endif
    const Params &P;
    int lane;
    int replica;
    uint64_t *occ;
    double *lw, *lcl, *lsc;   // lw[2l] = w1_l, lw[2l+1] = w2_l (interleaved: one ds_read2_b64 per link)
    DevEnv *e;
    uint32_t *sa, *sb;
    float *sr;
    FieldLds fl;       // valid only in k_observe and the highest-SNR variant of k_run
    int *nreq;
    double *lim;       // LDS [8] linear-domain acceptance limits 10^(-(thr_m+margin)/10) of this replica
    double *rp;        // LDS [2] 1/launch_power, launch_power^2 of this replica
    double *lim0;      // LDS [8] 10^(-thr_m/10): the disruption check ignores the margin (envs/qrmsa.pyx:947)
    double *so;        // LDS [C] Service.OSNR per record        } defragmentation only
    uint32_t *sq;      // LDS [C] Service.service_id per record  }
    double *phi;       // LDS [8] Phi_mod * 5/3 per modulation
    double *nlic;      // LDS [8] nli_coef[nreq[m]] * launch_power^2 of the CURRENT request } refreshed with nreq: the policy
    double *selfa;     // LDS [8] self_asinh[nreq[m]]                                        } reads them at LDS latency
    uint16_t *list;
    // first candidate path of the CURRENT request, loaded right after the request was drawn so that the table
    // round trips overlap the departures scan (id < 0: no such path)
    int pre_id, pre_hops, pre_mylink;
    uint64_t pre_m0, pre_m1;
    double pre_ase, pre_w1;
    int active;        // running services (wave-uniform, mirrored to e->st.active at store time)
    int skip_id;       // track_ids: service_id whose running namesakes the next GN evaluation skips (core/osnr.pyx:65); -1 none
    int lane_terms;    // per-lane interferer-link term counter (reduced once per launch)
    // sum of Service.OSNR = -10 log10(acc) over accepted services, kept as -10 log10 of a running PRODUCT of the acc's
    // (one multiply per step instead of one log10; folded into episode_osnr_sum by flush_osnr)
    double osnr_prod;
    int gn_evals;      // wave-uniform
    int gn_skips;      // wave-uniform: evaluations decided by the ASE-only bound
    int paths_tried, path_hops;   // wave-uniform statistics
    long long active_sum;
    __device__ Ctx(const Params &p) : P(p) {}
};

__device__ __forceinline__ void ctx_bind(Ctx &c, unsigned char *smem) {
    const Params &P = c.P;
    // fixed-size pieces first: their LDS addresses are link-time constants (immediate offsets, no scalar registers)
    c.e = reinterpret_cast<DevEnv *>(smem);
    c.lim = reinterpret_cast<double *>(smem + kEnvHotBytes);
    c.rp = c.lim + 8;
    c.phi = c.rp + 2;
    c.nlic = c.phi + 8;
    c.selfa = c.nlic + 8;
    c.nreq = reinterpret_cast<int *>(c.selfa + 8);
    // then the pieces whose size depends on the configuration
    c.occ = reinterpret_cast<uint64_t *>(c.nreq + 8);
    c.lw = reinterpret_cast<double *>(c.occ + (size_t)P.n_links * P.row_words);
    // the per-link alpha tables exist only when the attenuation is not uniform
    c.lcl = c.lw + 2 * P.n_links;
    c.lsc = c.lcl + (P.uniform_alpha ? 0 : P.n_links);
    c.sa = reinterpret_cast<uint32_t *>(c.lsc + (P.uniform_alpha ? 0 : P.n_links));
    c.sb = c.sa + P.capacity;
    c.sr = reinterpret_cast<float *>(c.sb + P.capacity);
    c.list = reinterpret_cast<uint16_t *>(c.sr + P.capacity);
    c.lim0 = reinterpret_cast<double *>(c.list + P.capacity);   // capacity % 64 == 0 -> 8-byte aligned; only if enabled
    c.so = c.lim0 + 8;                                          // only with defragmentation
    c.sq = reinterpret_cast<uint32_t *>(c.so + P.capacity);
}

// Table pointers live in a Params object read from memory, so the compiler cannot know they are global and would emit
// flat_load (which also ties up the LDS counter). G() states the address space.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T *G(const T *p) {
    return (const __attribute__((address_space(1))) T *)p;
}

// One wavefront per workgroup: the LDS executes a wave's DS instructions in issue order, so a value written by one lane
// is visible to every lane's later read without draining the memory counters. What is needed is only that the compiler
// keeps the program order across the point: a wavefront-scope fence (emits no s_waitcnt) around a wave barrier.
// (__syncthreads() would emit `s_waitcnt vmcnt(0) lgkmcnt(0)` and so also wait for every global load in flight.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------------
// small wave helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniform_f32(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ double uniform_f64(double v) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {   // DPP lane permutation of a 64-bit value (two 32-bit movs)
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xF, 0xF, true);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xF, 0xF, true);
    return b.d;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    union { double d; int i[2]; } a;
    a.d = v;
    a.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    a.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return a.d;
}
// x as an opaque value: the operation that produced it cannot be fused with the one that consumes it.  HIP compiles device code
// with -ffp-contract=fast and __dmul_rn / __dadd_rn are plain operators there, so `a * b + c` written with them still becomes ONE
// fma — one rounding where the reference (Python floats) makes two.  Found by the round-3 soak: two routes whose lowest-
// fragmentation scores tie with two roundings (the lower index wins, heuristics.py:404-406) and differ by an ulp with one.
__device__ __forceinline__ double fp_barrier(double x) {
    asm volatile("" : "+v"(x));
    return x;
}
// Wave-wide fp64 sum, identical bits in every lane: symmetric DPP exchanges inside each row of 16 lanes
// (quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror), then the 4 row sums through v_readlane.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {   // lane must be wave-uniform
    union { uint64_t u; int i[2]; } a;
    a.u = v;
    a.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    a.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return a.u;
}
// word of lane+1 within the first row of 16 lanes (0 past the row): DPP row_shl:1
__device__ __forceinline__ uint64_t next_word(uint64_t x) {
    union { uint64_t u; int i[2]; } a, b;
    a.u = x;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x101, 0xF, 0xF, true);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x101, 0xF, 0xF, true);
    return b.u;
}
__device__ __forceinline__ uint64_t lanes_below(int lane) { return (1ull << lane) - 1ull; }

// Bit s of the result is set iff bits [s, s+m) of the multi-word bitmap are all set. Word w lives in lane w (< 16;
// the other lanes hold 0). `x` enters as the run-AND for length r (r = 1: the bitmap itself) and leaves as the one for
// m >= r: x_m = x_r & (x_r >> (m - r)) while m - r <= r, so consecutive modulations (non-decreasing slot counts) cost
// one shift-AND each instead of a fresh log-step ladder.
__device__ __forceinline__ uint64_t run_and(uint64_t x, int &r, int m) {
    while (r < m) {
        int s = min(r, m - r);
        uint64_t y;
        if (s < 64) {
            y = (x >> s) | (next_word(x) << (64 - s));   // s >= 1
        } else {
            int q = s >> 6, t = s & 63;
            uint64_t a = __shfl_down((unsigned long long)x, q);
            uint64_t b = __shfl_down((unsigned long long)x, q + 1);
            y = t ? ((a >> t) | (b << (64 - t))) : a;
        }
        x &= y;
        r += s;
    }
    return x;
}

// lowest set position of a lane-distributed bitmap, or -1
__device__ __forceinline__ int first_set(uint64_t x) {
    uint64_t bal = __ballot(x != 0);
    if (!bal) return -1;
    int fl = __ffsll((unsigned long long)bal) - 1;
    uint64_t w = readlane_u64(x, fl);
    return fl * 64 + (__ffsll((unsigned long long)w) - 1);
}

// mask with bits [lo, hi) of word w set (global slot coordinates)
__device__ __forceinline__ uint64_t word_range(int w, int lo, int hi) {
    int a = max(lo - 64 * w, 0), b = min(hi - 64 * w, 64);
    if (b <= a) return 0;
    uint64_t m = (b - a == 64) ? ~0ull : ((1ull << (b - a)) - 1ull);
    return m << a;
}

// ---- running-service record codec -------------------------------------------------------------------------------------
// generic : a = path_id | slot<<16            b = nslots | modulation<<16          (link mask gathered from path_mask)
// R32     : a = 32-bit link mask of the path   b = slot | nslots<<10 | modulation<<20 | path_id<<23
//           (n_links <= 32, n_paths <= 512, n_slots <= 1023: NSFNET, COST239, rings) — the GN scans and the departures
//           then need no table lookup per record at all.
template <bool R32> __device__ __forceinline__ void rec_pack(int path, uint64_t mask0, int slot, int n, int mod,
                                                             uint32_t &a, uint32_t &b) {
    if (R32) { a = (uint32_t)mask0; b = (uint32_t)slot | ((uint32_t)n << 10) | ((uint32_t)mod << 20) | ((uint32_t)path << 23); }
    else { a = (uint32_t)path | ((uint32_t)slot << 16); b = (uint32_t)n | ((uint32_t)mod << 16); }
}
template <bool R32> __device__ __forceinline__ int rec_path(uint32_t a, uint32_t b) { return R32 ? (int)(b >> 23) : (int)(a & 0xFFFF); }
template <bool R32> __device__ __forceinline__ int rec_slot(uint32_t a, uint32_t b) { return R32 ? (int)(b & 0x3FF) : (int)(a >> 16); }
template <bool R32> __device__ __forceinline__ int rec_n(uint32_t a, uint32_t b) { return R32 ? (int)((b >> 10) & 0x3FF) : (int)(b & 0xFFFF); }
template <bool R32> __device__ __forceinline__ int rec_mod(uint32_t a, uint32_t b) { return R32 ? (int)((b >> 20) & 0x7) : (int)((b >> 16) & 0xFF); }

// A path as the wave sees it: lane h holds link h.
struct PathRef {
    int id, hops;
    int mylink;          // link index of hop `lane` (undefined for lane >= hops)
    uint64_t m0, m1;     // link mask
    double ase, w1;      // path_ase[id], path_w1[id] (wave-uniform)
};

__device__ __forceinline__ void prefetch_first_path(Ctx &c, int src, int dst) {
    const Params &P = c.P;
    int path = uniform_i32(G(P.pair_paths)[(src * P.n_nodes + dst) * P.k_paths]);
    c.pre_id = path;
    if (path >= 0) {
        c.pre_hops = uniform_i32(G(P.path_hops)[path]);
        c.pre_mylink = (c.lane < c.pre_hops) ? G(P.path_links)[path * P.max_hops + c.lane] : 0;
        c.pre_m0 = G(P.path_mask)[2 * path];
        c.pre_m1 = G(P.path_mask)[2 * path + 1];
        c.pre_ase = G(P.path_ase)[path];
        c.pre_w1 = G(P.path_w1)[path];
    }
}

__device__ __forceinline__ PathRef load_path(const Ctx &c, int path) {
    const Params &P = c.P;
    PathRef r;
    r.id = path;
    r.hops = G(P.path_hops)[path];
    r.mylink = (c.lane < r.hops) ? G(P.path_links)[path * P.max_hops + c.lane] : 0;
    r.m0 = G(P.path_mask)[2 * path];
    r.m1 = G(P.path_mask)[2 * path + 1];
    r.ase = G(P.path_ase)[path];
    r.w1 = G(P.path_w1)[path];
    return r;
}

// AND of the free bitmaps of a path's links (get_available_slots, envs/qrmsa.pyx:1482-1512), extended by the virtual
// free slot S: a run of n+1 set bits starting at s in this bitmap <=> `_get_candidates` accepts s
// (n slots + right guard, guard waived iff the run ends at S; envs/qrmsa.pyx:515-541, is_path_free :1248-1264).
__device__ __forceinline__ uint64_t path_free_ext(const Ctx &c, const PathRef &p) {
    const Params &P = c.P;
    uint64_t x = (c.lane < P.row_words) ? ~0ull : 0ull;
    for (int h = 0; h < p.hops; h++) {
        int l = __builtin_amdgcn_readlane(p.mylink, h);
        if (c.lane < P.row_words) x &= c.occ[l * P.row_words + c.lane];
    }
    if (c.lane == (P.n_slots >> 6)) x |= 1ull << (P.n_slots & 63);
    return x;
}

// set (free_=true) or clear the slots [lo, hi) on every link of the path; lane h owns link h.
template <bool SYNC = true>
__device__ __forceinline__ void mark_links(Ctx &c, int hops, int mylink, int lo, int hi, bool free_) {
    const Params &P = c.P;
    if (hi > P.n_slots) hi = P.n_slots;
    if (c.lane < hops && hi > lo) {
        for (int w = lo >> 6; w <= (hi - 1) >> 6; w++) {
            uint64_t m = word_range(w, lo, hi);
            uint64_t v = c.occ[mylink * P.row_words + w];
            c.occ[mylink * P.row_words + w] = free_ ? (v | m) : (v & ~m);
        }
    }
    if (SYNC) wave_sync();   // SYNC = false: the caller orders a later wave_sync before anything reads the bitmap again
}

// same, for a path given as a link mask (links < 32): lane l owns link l.
__device__ __forceinline__ void mark_mask(Ctx &c, uint32_t mask, int lo, int hi, bool free_) {
    const Params &P = c.P;
    if (hi > P.n_slots) hi = P.n_slots;
    if (c.lane < 32 && ((mask >> c.lane) & 1u) && hi > lo) {
        for (int w = lo >> 6; w <= (hi - 1) >> 6; w++) {
            uint64_t m = word_range(w, lo, hi);
            uint64_t v = c.occ[c.lane * P.row_words + w];
            c.occ[c.lane * P.row_words + w] = free_ ? (v | m) : (v & ~m);
        }
    }
    wave_sync();
}

// ---- GN model (core/osnr.pyx:21-142) ----------------------------------------------------------------------------
// pass 1: compact the indices of the running services that share >= 1 link with the candidate path.
template <bool R32>
__device__ __forceinline__ int gn_build_list(Ctx &c, uint64_t cm0, uint64_t cm1) {
    const Params &P = c.P;
    int L = 0;
    // two chunks of 64 records per iteration: both LDS reads (and, in the generic codec, both mask gathers) are in flight
    // before the first ballot
    for (int base = 0; base < c.active; base += 2 * kWave) {
        const int i0 = base + c.lane, i1 = i0 + kWave;
        const uint32_t a0 = i0 < c.active ? c.sa[i0] : 0u, a1 = i1 < c.active ? c.sa[i1] : 0u;
        bool ov0 = false, ov1 = false;
        if (R32) { ov0 = (a0 & (uint32_t)cm0) != 0; ov1 = (a1 & (uint32_t)cm0) != 0; }
        else {
            if (i0 < c.active) { const int pk = a0 & 0xFFFF; ov0 = ((G(P.path_mask)[2 * pk] & cm0) | (G(P.path_mask)[2 * pk + 1] & cm1)) != 0; }
            if (i1 < c.active) { const int pk = a1 & 0xFFFF; ov1 = ((G(P.path_mask)[2 * pk] & cm0) | (G(P.path_mask)[2 * pk + 1] & cm1)) != 0; }
        }
        if (P.track_ids && c.skip_id >= 0) {    // running services with the evaluated service's id are not interferers (quirk Q12)
            if (i0 < c.active && c.sq[i0] == (uint32_t)c.skip_id) ov0 = false;
            if (i1 < c.active && c.sq[i1] == (uint32_t)c.skip_id) ov1 = false;
        }
        const uint64_t bal0 = __ballot(ov0), bal1 = __ballot(ov1);
        const int n0 = __popcll((unsigned long long)bal0);
        if (ov0) c.list[L + __popcll((unsigned long long)(bal0 & lanes_below(c.lane)))] = (uint16_t)i0;
        if (ov1) c.list[L + n0 + __popcll((unsigned long long)(bal1 & lanes_below(c.lane)))] = (uint16_t)i1;
        L += n0 + __popcll((unsigned long long)bal1);
    }
    wave_sync();
    return L;
}

// asinh(U) - asinh(V), 0 < V < U, as one logarithm
__device__ __forceinline__ double asinh_diff(double U, double V) {
    return log((U + sqrt(fma(U, U, 1.0))) / (V + sqrt(fma(V, V, 1.0))));
}

struct GnLin {          // noise-to-signal ratios in the linear domain (wave-uniform)
    double ase, nli;    // acc_ase, acc_nli of core/osnr.pyx:133-135 summed over all spans
};

// pass 2: 1/SNR_ase and 1/SNR_nli of a candidate lightpath (path, slot s, n slots) against the compacted interferers.
// Span-hoisted: every span of a link is identical (topology.pyx:288-299), so the per-span sums of core/osnr.pyx:50-135
// collapse to per-link weights w1 = nspans*l_eff, w2 = nspans*l_eff*l_eff/(L*1e3) (quirk Q11).
struct GnCoef {         // the candidate's slot-count dependent factors (wave-uniform)
    double nlic;        // nli_coef[n] * launch_power^2
    double selfa;       // self_asinh[n] (uniform attenuation)
};
__device__ __forceinline__ GnCoef coef_for_slots(const Ctx &c, int n) {      // table loads (queries, arbitrary n)
    GnCoef k;
    k.nlic = G(c.P.nli_coef)[n] * c.rp[1];
    k.selfa = G(c.P.self_asinh)[n];
    return k;
}
__device__ __forceinline__ GnCoef coef_for_mod(const Ctx &c, int m) {        // the current request's, from LDS
    GnCoef k;
    k.nlic = c.nlic[m];
    k.selfa = c.selfa[m];
    return k;
}
// refresh nlic / selfa after nreq changed (lane m -> modulation m); n outside [1, S] never reaches the GN model
__device__ __forceinline__ void set_request_coefs(Ctx &c, int n, double lp2) {
    const Params &P = c.P;
    if (c.lane < P.n_mods) {
        const bool ok = n >= 1 && n <= P.n_slots;
        c.nlic[c.lane] = ok ? G(P.nli_coef)[n] * lp2 : 0.0;
        c.selfa[c.lane] = ok ? G(P.self_asinh)[n] : 0.0;
    }
}

template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ GnLin gn_eval(Ctx &c, const PathRef &p, int L, int s, int n, const GnCoef &kf) {
    const Params &P = c.P;
    const double bw = P.slot_bw * n;
    const int c2 = 2 * s + n;  // candidate centre in half-slots
    double part = 0.0;
    // self-channel term asinh(pi^2 |b2| B^2 / (4 alpha)) per link (core/osnr.pyx:58-61)
    if (UNIFORM_ALPHA) {
        if (c.lane == 0) part = p.w1 * kf.selfa;
    } else {
        if (c.lane < p.hops) part = c.lw[2 * (p.mylink)] * asinh(c.lsc[p.mylink] * (bw * bw));
    }
    int terms = 0;
    STAMPW(c, 10);
    for (int j = c.lane; j < L; j += kWave) {
        int idx = c.list[j];
        uint32_t a = c.sa[idx], b = c.sb[idx];
        int sk = rec_slot<R32>(a, b), nk = rec_n<R32>(a, b), mk = rec_mod<R32>(a, b);
        int adi = abs((2 * sk + nk) - c2);               // centre distance in half-slots (exact)
        uint64_t m0, m1;
        if (R32) { m0 = a & (uint32_t)p.m0; m1 = 0; }
        else { int pk = a & 0xFFFF; m0 = G(P.path_mask)[2 * pk] & p.m0; m1 = G(P.path_mask)[2 * pk + 1] & p.m1; }
        STAMPW(c, 11);
        if (UNIFORM_ALPHA) {
            double A, corr;
            if (nk <= P.tab_nmax) {   // (asinh difference, Bk/|df|) depend on two small integers only: one 16-byte gather
                const auto *t = G(reinterpret_cast<const double *>(P.pair_tab)) + 2 * ((nk - 1) * P.tab_stride + adi);
                A = t[0];
                corr = c.phi[mk] * t[1];
                STAMPW(c, 12);
            } else {
                double bk = P.slot_bw * nk, adf = (0.5 * P.slot_bw) * (double)adi, ck = P.alpha0_cl * bk;
                A = asinh_diff(ck * (adf + 0.5 * bk), ck * (adf - 0.5 * bk));   // adf > bk/2: allocations never overlap
                corr = c.phi[mk] * (bk / adf);
            }
            double w1 = 0.0, w2 = 0.0;
            if (R32) {
                // every listed interferer shares at least one link; the weights of the first TWO shared links are read
                // together (one LDS round trip instead of two), further ones - rare - in the loop
                uint32_t m = (uint32_t)m0;
                terms += __popc(m);
                const int l0 = __ffs(m) - 1;
                m &= m - 1;
                const bool two = m != 0;
                const int l1 = two ? __ffs(m) - 1 : l0;
                const double a0 = c.lw[2 * l0], b0 = c.lw[2 * l0 + 1], a1 = c.lw[2 * l1], b1 = c.lw[2 * l1 + 1];
                m &= m - 1;
                w1 = a0 + (two ? a1 : 0.0);
                w2 = b0 + (two ? b1 : 0.0);
                while (m) { int l = __ffs(m) - 1; m &= m - 1; w1 += c.lw[2 * (l)]; w2 += c.lw[2 * (l) + 1]; }
            } else {
                terms += __popcll((unsigned long long)m0) + __popcll((unsigned long long)m1);
                while (m0) { int l = __ffsll((unsigned long long)m0) - 1; m0 &= m0 - 1; w1 += c.lw[2 * (l)]; w2 += c.lw[2 * (l) + 1]; }
                while (m1) { int l = 64 + __ffsll((unsigned long long)m1) - 1; m1 &= m1 - 1; w1 += c.lw[2 * (l)]; w2 += c.lw[2 * (l) + 1]; }
            }
            part += A * w1 - corr * w2;
        } else {
            double bk = P.slot_bw * nk, adf = (0.5 * P.slot_bw) * (double)adi;
            double hi = adf + 0.5 * bk, lo = adf - 0.5 * bk, corr = c.phi[mk] * (bk / adf);
            terms += __popcll((unsigned long long)m0) + __popcll((unsigned long long)m1);
            while (m0 | m1) {
                int l;
                if (m0) { l = __ffsll((unsigned long long)m0) - 1; m0 &= m0 - 1; }
                else { l = 64 + __ffsll((unsigned long long)m1) - 1; m1 &= m1 - 1; }
                double ck = c.lcl[l] * bk;
                part += asinh_diff(ck * hi, ck * lo) * c.lw[2 * (l)] - corr * c.lw[2 * (l) + 1];
            }
        }
    }
    c.lane_terms += terms;
    c.gn_evals++;
    STAMPW(c, 13);
    double total = wave_sum(part);
    STAMPW(c, 14);
    // P_nli/P = (P/B)^3 * 8/(27 pi |b2|) * gamma^2 * B / P * sum = nli_coef[n] * P^2 * sum   (core/osnr.pyx:109-116,135)
    // P_ase/P = B * h * fc * sum_l n_l (exp(2 a L) - 1) NF / P                                 (core/osnr.pyx:119-125,134)
    double fc = P.f0 + (P.slot_bw * s) + (P.slot_bw * (n / 2.0));  // envs/qrmsa.pyx:901-905
    GnLin g;
    g.nli = kf.nlic * total;
    g.ase = (bw * fc * p.ase) * c.rp[0];
    STAMPW(c, 15);
    return g;
}
template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ GnLin gn_eval(Ctx &c, const PathRef &p, int L, int s, int n) {
    return gn_eval<UNIFORM_ALPHA, R32>(c, p, L, s, n, coef_for_slots(c, n));
}

__device__ __forceinline__ void gn_to_db(const GnLin &g, double out[3]) {   // core/osnr.pyx:138-140
    out[0] = -10.0 * log10(g.ase + g.nli);   // = 10 log10(1/acc)
    out[1] = -10.0 * log10(g.ase);
    out[2] = -10.0 * log10(g.nli);
}

// GSNR >= minimum_osnr + margin ?  (heuristics.py:957-958, envs/qrmsa.pyx:911). The comparison is made in the linear
// domain against lim = 10^(-(thr+margin)/10); inside a 1e-9 relative band around the limit it falls back to the
// reference's own expression 10*log10(1/acc) >= thr + margin, so the decision is the dB-domain one everywhere.
__device__ __forceinline__ int qot_ok(const Ctx &c, const GnLin &g, int m, double margin) {
    double acc = g.ase + g.nli, lim = c.lim[m];
    int ok;
    if (acc <= lim * (1.0 - 1e-9)) ok = 1;
    else if (acc >= lim * (1.0 + 1e-9)) ok = 0;
    else ok = 10.0 * log10(1.0 / acc) >= c.P.mod_thr[m] + margin;
    return uniform_i32(ok);
}

struct Choice {
    int action, route, mod, slot, n, path;
    int flags;          // ONGYM_F_BLOCKED_*
    int hops, mylink;   // chosen path as lanes see it
    uint64_t m0;        // its link mask (low word)
    GnLin g;
    int busy = 0;       // exact fit only: the proposed slots lack the guard slot -> the step answers with the penalty
};

// ---- heuristic_shortest_available_path_first_fit_best_modulation (heuristics/heuristics.py:923-966) -----------
template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ void policy_first_fit(Ctx &c, int src, int dst, double launch_power, double margin,
                                                 Choice &ch) {
    const Params &P = c.P;
    // formats from max_modulation_idx down (heuristics.py:930); the action index is relative to it and uses the codec's
    // window width (get_action_index, :36-54) — with modulations_to_consider < n_mods it can leave the window, exactly like
    // the reference's (the step then decodes what the codec says, see k_run)
    const int M = P.n_mods_consider, S = P.n_slots, max_mod = uniform_i32(c.e->st.max_modulation_idx);
    ch.action = P.k_paths * M * S; ch.route = -1; ch.mod = -1; ch.slot = -1; ch.n = 0; ch.hops = 0; ch.mylink = 0;
    ch.path = -1; ch.m0 = 0; ch.g.ase = ch.g.nli = 0.0;
    int bres = 0, bosnr = 0;
    for (int k = 0; k < P.k_paths; k++) {
        int path = k == 0 ? c.pre_id : G(P.pair_paths)[(src * P.n_nodes + dst) * P.k_paths + k];   // id looked up at draw time
        if (path < 0) break;
        PathRef p;
        p = load_path(c, path);
        c.paths_tried++; c.path_hops += p.hops;
        // run-AND of length r, extended modulation by modulation (the bitmap itself is not kept alive across the GN
        // evaluation: the rare restart below recomputes it)
        uint64_t runs = path_free_ext(c, p);
        STAMP(c, 1);
        int r = 1, L = -1;
        for (int m = max_mod; m >= 0; m--) {
            int n = uniform_i32(c.nreq[m]);
            if (n <= 0) continue;
            if (n + 1 < r) { runs = path_free_ext(c, p); r = 1; }   // slot counts normally grow as the modulation index falls
            runs = run_and(runs, r, n + 1);
            int first = first_set(runs);
            STAMP(c, 2);
            if (first < 0) { bres = 1; continue; }
            if (P.ase_shortcut) {
                // No interferer term of the NLI sum is negative (checked at create), so
                //   1/SNR >= 1/SNR_ase + 1/SNR_nli(self-channel term only):
                // a candidate that fails on this O(1) bound fails the full evaluation too (same expressions as
                // gn_eval, same 1e-9 guard band as qot_ok).
                double bw = P.slot_bw * n;
                double fc = P.f0 + (P.slot_bw * first) + (P.slot_bw * (n / 2.0));
                double lb = (bw * fc * p.ase) * c.rp[0];
                if (UNIFORM_ALPHA) lb += c.nlic[m] * (p.w1 * c.selfa[m]);
                if (uniform_i32(lb >= c.lim[m] * (1.0 + 1e-9))) { bosnr = 1; bres = 0; c.gn_skips++; continue; }
            }
            if (L < 0) { L = gn_build_list<R32>(c, p.m0, p.m1); STAMP(c, 3); }
            GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, first, n, coef_for_mod(c, m));
            int ok = qot_ok(c, g, m, margin);
            STAMP(c, 4);
            if (ok) {
                ch.action = k * M * S + (max_mod - m) * S + first;   // get_action_index, heuristics.py:36-54
                ch.route = k; ch.mod = m; ch.slot = first; ch.n = n; ch.hops = p.hops; ch.mylink = p.mylink;
                ch.path = path; ch.m0 = p.m0;
                ch.g.ase = uniform_f64(g.ase); ch.g.nli = uniform_f64(g.nli);
                ch.flags = 0;
                return;
            }
            bosnr = 1;
            bres = 0;
        }
    }
    ch.flags = (bres ? ONGYM_F_BLOCKED_RESOURCES : 0) | (bosnr ? ONGYM_F_BLOCKED_OSNR : 0);
}


// ---- load_balancing_best_modulation (heuristics/heuristics.py:547-627) ------------------------------------------
// Same per-path search as first fit (best modulation, lowest feasible slot, GN check), but every path is examined and the
// one with the fewest busy slots per hop wins; a path is skipped as soon as its load is not below the best so far.
template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ void policy_load_balancing(Ctx &c, int src, int dst, double launch_power, double margin,
                                                      Choice &ch) {
    const Params &P = c.P;
    const int M = P.n_mods, S = P.n_slots, max_mod = M - 1;
    ch.action = P.k_paths * M * S; ch.route = -1; ch.mod = -1; ch.slot = -1; ch.n = 0; ch.hops = 0; ch.mylink = 0;
    ch.path = -1; ch.m0 = 0; ch.g.ase = ch.g.nli = 0.0; ch.flags = 0;
    int any_res = 0, any_osnr = 0;
    double lowest_load = INFINITY;
    for (int k = 0; k < P.k_paths; k++) {
        int path = k == 0 ? c.pre_id : G(P.pair_paths)[(src * P.n_nodes + dst) * P.k_paths + k];
        if (path < 0) break;
        PathRef p;
        if (k == 0) { p.id = path; p.hops = c.pre_hops; p.mylink = c.pre_mylink; p.m0 = c.pre_m0; p.m1 = c.pre_m1; p.ase = c.pre_ase; p.w1 = c.pre_w1; }
        else p = load_path(c, path);
        c.paths_tried++; c.path_hops += p.hops;
        const uint64_t free_ext = path_free_ext(c, p);
        // np.sum(available_slots == 0) / len(path.links): busy slots of the path row (virtual guard bit excluded)
        const uint64_t aw = (c.lane == (S >> 6)) ? (free_ext & ~(1ull << (S & 63))) : free_ext;
        int freecnt = c.lane < P.row_words ? __popcll((unsigned long long)aw) : 0;
#pragma unroll
        for (int mm = 8; mm >= 1; mm >>= 1) freecnt += __shfl_xor(freecnt, mm);     // words live in lanes 0..15
        freecnt = uniform_i32(freecnt);
        const double current_load = (double)(S - freecnt) / (double)p.hops;
        if (current_load >= lowest_load) continue;
        uint64_t runs = free_ext;
        int r = 1, L = -1;
        for (int m = max_mod; m >= 0; m--) {
            int n = uniform_i32(c.nreq[m]);
            if (n <= 0) continue;
            if (n + 1 < r) { runs = free_ext; r = 1; }
            runs = run_and(runs, r, n + 1);
            int first = first_set(runs);
            if (first < 0) { any_res = 1; continue; }
            if (P.ase_shortcut) {   // exact lower bound, see policy_first_fit
                double bw = P.slot_bw * n;
                double fc = P.f0 + (P.slot_bw * first) + (P.slot_bw * (n / 2.0));
                double lb = (bw * fc * p.ase) * c.rp[0];
                if (UNIFORM_ALPHA) lb += c.nlic[m] * (p.w1 * c.selfa[m]);
                if (uniform_i32(lb >= c.lim[m] * (1.0 + 1e-9))) { any_osnr = 1; c.gn_skips++; continue; }
            }
            if (L < 0) L = gn_build_list<R32>(c, p.m0, p.m1);
            GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, first, n, coef_for_mod(c, m));
            if (qot_ok(c, g, m, margin)) {
                lowest_load = current_load;
                ch.action = k * M * S + (max_mod - m) * S + first;
                ch.route = k; ch.mod = m; ch.slot = first; ch.n = n; ch.hops = p.hops; ch.mylink = p.mylink;
                ch.path = path; ch.m0 = p.m0;
                ch.g.ase = uniform_f64(g.ase); ch.g.nli = uniform_f64(g.nli);
                break;
            }
            any_osnr = 1;
        }
    }
    if (ch.route >= 0) { ch.flags = 0; return; }
    if (any_osnr) any_res = 0;
    ch.flags = (any_res ? ONGYM_F_BLOCKED_RESOURCES : 0) | (any_osnr ? ONGYM_F_BLOCKED_OSNR : 0);
}


// ---- decode + validate an external action (envs/qrmsa.pyx:801-834, 867-909) ------------------------------------
// returns 0 accept (GN evaluated, passes), 1 reject action, 2 slots not free (retry), 3 QoT infeasible
template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ int evaluate_action(Ctx &c, int src, int dst, double launch_power, double margin,
                                               int action, Choice &ch) {
    const Params &P = c.P;
    const int M = P.n_mods_consider, S = P.n_slots, max_mod = uniform_i32(c.e->st.max_modulation_idx);
    ch.action = action; ch.route = -1; ch.mod = -1; ch.slot = -1; ch.n = 0; ch.flags = 0; ch.hops = 0; ch.mylink = 0;
    ch.path = -1; ch.m0 = 0; ch.g.ase = ch.g.nli = 0.0;
    if (action == P.k_paths * M * S) return 1;
    if (action < 0) return 2;
    // an index ABOVE the reject action is decoded like any other: encoded_decimal_to_array (qrmsa.pyx:801-834) takes the digits
    // modulo (slots, formats, routes), so it aliases a (route, format, slot).  The heuristics produce such indices under a
    // narrow codec (formats below the window, heuristics.py:36-54 with modulations_to_consider < len(modulations)).
    int slot = action % S; int t = action / S;
    int r = t % M; t /= M;
    int route = t % P.k_paths;
    // allowed_mods = range(max_idx, max_idx - M, -1) if max_idx > 1 else reversed(range(M))   (envs/qrmsa.pyx:821-825)
    int m = max_mod > 1 ? max_mod - r : (M - 1) - r;
    if (m < 0 || m >= P.n_mods) return 2;
    ch.route = route; ch.mod = m; ch.slot = slot;
    int path = G(P.pair_paths)[(src * P.n_nodes + dst) * P.k_paths + route];
    int n = c.nreq[m];
    ch.n = n; ch.path = path;
    if (path < 0 || n <= 0) return 2;
    PathRef p = load_path(c, path);
    ch.hops = p.hops; ch.mylink = p.mylink; ch.m0 = p.m0;
    int rr = 1;
    uint64_t ok_starts = run_and(path_free_ext(c, p), rr, n + 1);     // is_path_free, envs/qrmsa.pyx:1248-1264
    uint64_t w = readlane_u64(ok_starts, uniform_i32(slot >> 6));
    if (!((w >> (slot & 63)) & 1ull)) return 2;
    int L = gn_build_list<R32>(c, p.m0, p.m1);
    GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, slot, n, coef_for_mod(c, m));
    ch.g.ase = uniform_f64(g.ase); ch.g.nli = uniform_f64(g.nli);
    return qot_ok(c, g, m, margin) ? 0 : 3;
}

// ---- the cheaper remaining policies of heuristics/heuristics.py, one function (policy id is wave-uniform) -------------
// lowest spectrum (:431-490), load-balancing first fit (:202-269), best-modulation load balancing (:491-545), simplified
// MSCL (:765-839) and its sequential variant (:841-921), PSR with default coefficients (:1019-1119), exact fit
// (:1121-1227). All reuse the bitmap scan and the GN evaluation of the first-fit path; none is on the benchmark path, so
// they take the full GN evaluation for every candidate (no lower-bound shortcut).
__device__ __forceinline__ bool any_bits(uint64_t x) { return __ballot(x != 0) != 0; }

// multiword logical shift right by s >= 1 of a lane-distributed bitmap (word w in lane w < 16, zeros elsewhere)
__device__ __forceinline__ uint64_t shift_right(uint64_t x, int s) {
    if (s < 64) return (x >> s) | (next_word(x) << (64 - s));
    const int q = s >> 6, t = s & 63;
    const uint64_t a = __shfl_down((unsigned long long)x, q), b = __shfl_down((unsigned long long)x, q + 1);
    return t ? ((a >> t) | (b << (64 - t))) : a;
}
// multiword shift left by one
__device__ __forceinline__ uint64_t shift_left1(uint64_t x) {
    const uint64_t prev = __shfl_up((unsigned long long)x, 1);
    return (x << 1) | (threadIdx.x > 0 ? prev >> 63 : 0ull);
}
// longest run of set bits (_get_largest_contiguous_block, :751-763): largest m with a window of m ones, by doubling
// then binary refinement of the run-AND
__device__ __forceinline__ int longest_run(uint64_t x) {
    if (!any_bits(x)) return 0;
    uint64_t y = x;
    int r = 1;
    for (;;) {
        int r2 = r;
        const uint64_t y2 = run_and(y, r2, 2 * r);
        if (!any_bits(y2)) break;
        y = y2; r = r2;
    }
    for (int step = r >> 1; step >= 1; step >>= 1) {
        int r2 = r;
        const uint64_t y2 = run_and(y, r2, r + step);
        if (any_bits(y2)) { y = y2; r = r2; }
    }
    return r;   // doubling found the largest power of two with a run, the halving steps the exact maximum
}

__device__ __forceinline__ int path_free_count(const Ctx &c, uint64_t row) {   // free slots of a row (no virtual bit)
    int cnt = c.lane < c.P.row_words ? __popcll((unsigned long long)row) : 0;
#pragma unroll
    for (int mm = 8; mm >= 1; mm >>= 1) cnt += __shfl_xor(cnt, mm);
    return uniform_i32(cnt);
}

template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ void policy_misc(Ctx &c, int policy, int src, int dst, double margin, Choice &ch) {
    const Params &P = c.P;
    const int M = P.n_mods, S = P.n_slots, max_mod = M - 1, K = P.k_paths;
    ch.action = K * M * S; ch.route = -1; ch.mod = -1; ch.slot = -1; ch.n = 0; ch.hops = 0; ch.mylink = 0;
    ch.path = -1; ch.m0 = 0; ch.g.ase = ch.g.nli = 0.0; ch.flags = 0;
    int bres = 0, bosnr = 0;
    int best_score = -1;
    auto take = [&](int k, int m, int slot, int n, const PathRef &p, const GnLin &g) {
        ch.action = k * M * S + (max_mod - m) * S + slot;
        ch.route = k; ch.mod = m; ch.slot = slot; ch.n = n; ch.hops = p.hops; ch.mylink = p.mylink;
        ch.path = p.id; ch.m0 = p.m0;
        ch.g.ase = uniform_f64(g.ase); ch.g.nli = uniform_f64(g.nli);
    };
    auto row_of = [&](uint64_t free_ext) {   // the path row without the virtual free slot S
        return (c.lane == (S >> 6)) ? (free_ext & ~(1ull << (S & 63))) : free_ext;
    };
    auto path_at = [&](int k, PathRef &p) -> bool {
        const int path = uniform_i32(G(P.pair_paths)[(src * P.n_nodes + dst) * K + k]);
        if (path < 0) return false;
        p = load_path(c, path);
        return true;
    };

    if (policy == ONGYM_POLICY_BEST_MOD_LB) {
        // modulations outer (all of them), routes inner; first free run of >= n+1 slots INSIDE the row (no edge waiver)
        for (int m = M - 1; m >= 0; m--) {
            const int n = uniform_i32(c.nreq[m]);
            if (n <= 0) continue;
            for (int k = 0; k < K; k++) {
                PathRef p;
                if (!path_at(k, p)) break;
                int rr = 1;
                const int slot = first_set(run_and(row_of(path_free_ext(c, p)), rr, n + 1));
                if (slot < 0) continue;
                const int L = gn_build_list<R32>(c, p.m0, p.m1);
                const GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, slot, n);
                if (qot_ok(c, g, m, margin)) { take(k, m, slot, n, p, g); return; }
            }
        }
        return;   // (reject, False, False)
    }

    // route order: natural, or by (busy slots, index) for load-balancing first fit
    int order[8];
    int nk = 0;
    for (int k = 0; k < K && k < 8; k++) order[k] = k;
    if (policy == ONGYM_POLICY_LB_FIRST_FIT) {
        int busy[8];
        for (int k = 0; k < K && k < 8; k++) {
            PathRef p;
            if (!path_at(k, p)) break;
            busy[k] = S - path_free_count(c, row_of(path_free_ext(c, p)));
            nk++;
        }
        for (int i = 1; i < nk; i++)
            for (int j = i; j > 0 && busy[order[j]] < busy[order[j - 1]]; j--) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    } else {
        nk = K < 8 ? K : 8;
    }

    for (int i = 0; i < nk; i++) {
        const int k = order[i];
        PathRef p;
        if (!path_at(k, p)) break;
        const uint64_t free_ext = path_free_ext(c, p);
        const uint64_t row = row_of(free_ext);
        int L = -1;
        if (policy == ONGYM_POLICY_MSCL_SEQUENTIAL) best_score = -1;
        const bool row_empty = !any_bits(row);
        for (int m = max_mod; m >= 0; m--) {
            const int n = uniform_i32(c.nreq[m]);
            if (n <= 0) continue;
            if (policy == ONGYM_POLICY_EXACT_FIT) {
                if (row_empty) { bres = 1; continue; }
                // walk the free runs left to right: first of length == n, else the smallest >= n (first among equals)
                uint64_t y = row;
                int exact = -1, fit = -1, fit_len = 0x7fffffff;
                for (;;) {
                    const int s0 = first_set(y);
                    if (s0 < 0) break;
                    int e0 = first_set(~row & word_range(c.lane, s0, S));
                    if (e0 < 0) e0 = S;
                    const int len = e0 - s0;
                    if (len == n) { exact = s0; break; }
                    if (len >= n && len < fit_len) { fit_len = len; fit = s0; }
                    y &= ~word_range(c.lane, s0, e0);
                }
                const int slot = exact >= 0 ? exact : fit;
                if (slot < 0) { bres = 1; continue; }
                if (L < 0) L = gn_build_list<R32>(c, p.m0, p.m1);
                const GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, slot, n);
                if (qot_ok(c, g, m, margin)) {
                    take(k, m, slot, n, p, g);
                    int rr = 1;                                    // is_path_free (envs/qrmsa.pyx:1248-1264) wants the guard
                    const uint64_t okb = run_and(free_ext, rr, n + 1);
                    const uint64_t w = readlane_u64(okb, uniform_i32(slot >> 6));
                    ch.busy = !((w >> (slot & 63)) & 1ull);
                    return;
                }
                bosnr = 1;
                continue;
            }
            int rr = 1;
            uint64_t v = run_and(free_ext, rr, n + 1);             // _get_candidates
            if (policy == ONGYM_POLICY_PSR) {
                for (;;) {                                         // every start of this modulation, lowest first
                    const int s0 = first_set(v);
                    if (s0 < 0) break;
                    if (L < 0) L = gn_build_list<R32>(c, p.m0, p.m1);
                    const GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, s0, n);
                    if (qot_ok(c, g, m, margin)) { take(k, m, s0, n, p, g); return; }
                    if (c.lane == (s0 >> 6)) v &= ~(1ull << (s0 & 63));
                }
                continue;
            }
            const int first = first_set(v);
            if (first < 0) { bres = 1; continue; }
            if (L < 0) L = gn_build_list<R32>(c, p.m0, p.m1);
            const GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, first, n);
            const int ok = qot_ok(c, g, m, margin);
            if (policy == ONGYM_POLICY_MSCL_SIMPLIFIED || policy == ONGYM_POLICY_MSCL_SEQUENTIAL) {
                if (ok) {   // score: the longest free run left on the route once [first, first+n) is taken (no guard)
                    const int score = longest_run(row & ~word_range(c.lane, first, first + n));
                    if (score > best_score) { best_score = score; take(k, m, first, n, p, g); }
                } else bosnr = 1;
                continue;
            }
            if (ok) { take(k, m, first, n, p, g); return; }        // lowest spectrum / load-balancing first fit
            bosnr = 1;
        }
        if (policy == ONGYM_POLICY_MSCL_SEQUENTIAL && ch.route >= 0) return;
    }
    if (ch.route >= 0) return;                                     // simplified MSCL: best over all routes
    if (policy == ONGYM_POLICY_PSR || policy == ONGYM_POLICY_LB_FIRST_FIT) { ch.flags = ONGYM_F_BLOCKED_RESOURCES; return; }
    if ((policy == ONGYM_POLICY_LOWEST_SPECTRUM || policy == ONGYM_POLICY_EXACT_FIT) && bosnr) bres = 0;
    ch.flags = (bres ? ONGYM_F_BLOCKED_RESOURCES : 0) | (bosnr ? ONGYM_F_BLOCKED_OSNR : 0);
}

// ---- departures: release every running service with float32 key <= now (envs/qrmsa.pyx:1113-1122, 1332-1350) ---
// Set semantics are identical to the heap loop because float32 rounding is monotone. Removal = move the last record
// into the hole; processing holes from the highest index down keeps every record above the hole a keeper.
template <bool R32>
__device__ __forceinline__ void release_due(Ctx &c, float now) {
    const Params &P = c.P;
    int nchunks = (c.active + kWave - 1) / kWave;
    for (int ch = nchunks - 1; ch >= 0; ch--) {
        int i = ch * kWave + c.lane;
        float r = (i < c.active) ? fabsf(c.sr[i]) : INFINITY;   // sign bit = 'disrupted' flag (measure_disruptions)
        uint64_t bal = __ballot(r <= now);
        while (bal) {
            int ln = 63 - __clzll((unsigned long long)bal);
            bal &= ~(1ull << ln);
            int idx = ch * kWave + ln;
            uint32_t a = c.sa[idx], b = c.sb[idx];
            int sk = rec_slot<R32>(a, b), nk = rec_n<R32>(a, b);
            if (R32) mark_mask(c, a, sk, sk + nk + 1, true);       // frees n+1 slots, clamped at S (quirk Q7)
            else {
                int pk = a & 0xFFFF;
                int hops = G(P.path_hops)[pk];
                int mylink = (c.lane < hops) ? G(P.path_links)[pk * P.max_hops + c.lane] : 0;
                mark_links(c, hops, mylink, sk, sk + nk + 1, true);
            }
            int last = c.active - 1;
            if (c.lane == 0 && idx != last) { c.sa[idx] = c.sa[last]; c.sb[idx] = c.sb[last]; c.sr[idx] = c.sr[last]; }
            c.active = last;
            wave_sync();
        }
    }
}

// ---- _next_service, request half (envs/qrmsa.pyx:1067-1111): draw/replay the next request, advance the clock ---
// Wave-cooperative evaluation of ongym_draw_request (include/ongym_traffic.h) — the SAME operations, so the result is
// bit-identical to the header's scalar definition: the two logarithms run in lanes 0 and 1, the three cumulative-table
// searches are one __ballot each (first i with x < cum[i]  ==  number of cum[i] <= x among i < n-1).
__device__ __forceinline__ int cum_search(const Ctx &c, const double *cum, double cum_reg, int n, double x) {
    if (n <= kWave) return __popcll((unsigned long long)__ballot(c.lane < n - 1 && cum_reg <= x));
    int cnt = 0;
    for (int base = 0; base < n - 1; base += kWave) {
        int i = base + c.lane;
        cnt += __popcll((unsigned long long)__ballot(i < n - 1 && cum[i] <= x));
    }
    return cnt;
}

__device__ __forceinline__ void draw_next(Ctx &c) {
    ONGYM_NO_CONTRACT
    const Params &P = c.P;
    DevEnv *e = c.e;
    if (e->have_request) return;
    float at, ht, br; int src, dst, bi = -1;
    if (P.req_mode == kReqRng) {
        // traffic tables, one entry per lane (ballot searches below): loaded here rather than kept in registers across the
        // whole step — they hit L1 and the latency hides behind the hash and the logarithm
        const double node_cum_reg = (c.lane < P.n_nodes && P.n_nodes <= kWave) ? G(P.node_cum)[c.lane] : INFINITY;
        const double br_cum_reg = (c.lane < P.n_bit_rates && P.n_bit_rates <= kWave) ? G(P.bit_rate_cum)[c.lane] : INFINITY;
        const float br_reg = (c.lane < P.n_bit_rates && P.n_bit_rates <= kWave) ? (float)G(P.bit_rates)[c.lane] : 0.f;
        // the five uniforms of this request in lanes 0..4, the two logarithms in lanes 0 and 1
        const uint64_t key = e->rng_key, ctr = e->req_index * ONGYM_DRAWS_PER_REQUEST;
        const double ul = ongym_uniform(key, ctr + (uint64_t)min(c.lane, ONGYM_DRAWS_PER_REQUEST - 1));
        const double u0 = readlane_f64(ul, 0), u1 = readlane_f64(ul, 1), u2 = readlane_f64(ul, 2),
                     u3 = readlane_f64(ul, 3), u4 = readlane_f64(ul, 4);
        float lg = ongym_logf_det(1.0 - (c.lane == 1 ? u1 : u0));
        float l0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lg), 0));
        float l1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lg), 1));
        at = (float)e->st.current_time + (-l0 * e->mean_iat_f);     // ongym_expovariate_f
        ht = -l1 * P.mean_holding_f;
        const int n = P.n_nodes;
        const bool small = n <= kWave;
        double total = small ? readlane_f64(node_cum_reg, n - 1) : P.node_cum[n - 1];
        src = cum_search(c, P.node_cum, node_cum_reg, n, u2 * total);
        double hi_s = small ? readlane_f64(node_cum_reg, src) : P.node_cum[src];
        double lo_s = src > 0 ? (small ? readlane_f64(node_cum_reg, src - 1) : P.node_cum[src - 1]) : 0.0;
        double w_s = hi_s - lo_s;
        double x = u3 * (total - w_s);
        if (x >= lo_s) x += w_s;
        dst = cum_search(c, P.node_cum, node_cum_reg, n, x);
        if (dst == src) dst = (src + 1 < n) ? src + 1 : src - 1;
        if (P.bit_rate_mode == 0) {
            const int nb = P.n_bit_rates;
            double tb = nb <= kWave ? readlane_f64(br_cum_reg, nb - 1) : P.bit_rate_cum[nb - 1];
            bi = cum_search(c, P.bit_rate_cum, br_cum_reg, nb, u4 * tb);
            br = nb <= kWave ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(br_reg), bi)) : (float)P.bit_rates[bi];
        } else {
            int span = P.br_hi - P.br_lo + 1;
            int k = (int)(u4 * (double)span);
            if (k >= span) k = span - 1;
            br = (float)(P.br_lo + k);
        }
    } else if (P.req_mode == kReqTrace && (long long)e->req_index < P.trace_n) {
        const ongym_request q = P.trace[(long long)c.replica * P.trace_n + (long long)e->req_index];
        at = q.arrival_time; ht = q.holding_time; br = q.bit_rate; src = q.source; dst = q.destination;
    } else {
        if (c.lane == 0) e->st.flags |= ONGYM_F_NO_REQUEST;
        wave_sync();
        return;
    }
    // slots needed per modulation: get_number_slots (envs/qrmsa.pyx:1198-1205), lane m -> nreq[m]
    prefetch_first_path(c, src, dst);
    if (bi >= 0) {   // discrete bit rates: slot count and its GN factors by (bit rate, modulation), independent loads
        if (c.lane < P.n_mods) {
            const int at_ = bi * kMaxMods + c.lane;
            const int nr = G(P.nreq_tab)[at_];
            const double k0 = G(P.req_coef)[2 * at_], k1 = G(P.req_coef)[2 * at_ + 1];
            c.nreq[c.lane] = nr;
            c.nlic[c.lane] = k0 * c.rp[1];
            c.selfa[c.lane] = k1;
        }
    } else {
        int nr = 0;
        if (c.lane < P.n_mods) {
            nr = (int)ceil((double)br / ((double)P.mod_se[c.lane] * P.nslots_width));
            c.nreq[c.lane] = nr;
        }
        set_request_coefs(c, nr, c.rp[1]);
    }
    if (c.lane == 0) {
        e->req_index++;
        e->st.current_time = (double)at;
        e->cur_at = at; e->cur_ht = ht; e->cur_br = br; e->cur_src = src; e->cur_dst = dst;
        e->cur_id = (int32_t)e->st.episode_services_processed;
        e->have_request = 1;
        e->st.services_processed += 1;
        e->st.episode_services_processed += 1;
        e->st.bit_rate_requested += (double)br;
        e->st.episode_bit_rate_requested += (double)br;
    }
    wave_sync();
    if (P.track_ids) c.skip_id = uniform_i32(e->cur_id);
}

// fold the running product into DevEnv.st.episode_osnr_sum (lane 0 only)
__device__ __forceinline__ void flush_osnr(Ctx &c) {
    if (c.osnr_prod != 1.0) { c.e->osnr_flushed += -10.0 * log10(c.osnr_prod); c.osnr_prod = 1.0; }
}

// ---- reset (envs/qrmsa.pyx:427-504) ----------------------------------------------------------------------------
__device__ __forceinline__ void reset_env(Ctx &c) {
    const Params &P = c.P;
    DevEnv *e = c.e;
    c.active = 0;
    int words = P.n_links * P.row_words;
    for (int i = c.lane; i < words; i += kWave) c.occ[i] = word_range(i % P.row_words, 0, P.n_slots);
    if (c.lane == 0) {
        ongym_stats &s = e->st;
        s.episode_bit_rate_requested = 0.0; s.episode_bit_rate_provisioned = 0.0;
        s.episode_services_processed = 0; s.episode_services_accepted = 0;
        s.rejected = 0;
        for (int m = 0; m < 8; m++) s.episode_modulation_hist[m] = 0;
        s.bit_rate_requested = 0.0; s.bit_rate_provisioned = 0.0;   // :466-467
        s.disrupted_services = 0; s.episode_disrupted_services = 0;   // :432, 468-469
        s.episode_defrag_cycles = 0; s.episode_service_reallocations = 0;   // :438-439
        s.max_modulation_idx = P.n_mods - 1;                                // :437
        s.episode_osnr_sum = 0.0;
        e->osnr_flushed = 0.0; c.osnr_prod = 1.0;
        e->svc_list_extra = 0;
        e->have_request = 0;
    }
    wave_sync();
    draw_next(c);   // no departures possible: the network is empty
}

// info of the terminal step (envs/qrmsa.pyx:996-1060), lane 0 only. Source: the LDS copy; destination: the snapshot
// fields of the replica's DevEnv in memory (they are not part of the LDS image).
__device__ __forceinline__ void snapshot_terminal(Ctx &c) {
    const DevEnv *e = c.e;
    const ongym_stats &s = e->st;
    ongym_stats &o = c.P.env[c.replica].st;
    o.last_episode_processed = s.episode_services_processed; o.last_episode_accepted = s.episode_services_accepted;
    o.last_rejected = s.rejected;
    o.last_service_blocking_rate = s.services_processed > 0
        ? (double)(s.services_processed - s.services_accepted) / (double)s.services_processed : 0.0;
    o.last_episode_service_blocking_rate = s.episode_services_processed > 0
        ? (double)(s.episode_services_processed - s.episode_services_accepted) / (double)s.episode_services_processed : 0.0;
    o.last_bit_rate_blocking_rate = s.bit_rate_requested > 0
        ? (s.bit_rate_requested - s.bit_rate_provisioned) / s.bit_rate_requested : 0.0;
    o.last_episode_bit_rate_blocking_rate = s.episode_bit_rate_requested > 0
        ? (s.episode_bit_rate_requested - s.episode_bit_rate_provisioned) / s.episode_bit_rate_requested : 0.0;
    for (int m = 0; m < 8; m++) o.last_modulation_hist[m] = s.episode_modulation_hist[m];
    // graph_load.py:181-185: mean of Service.OSNR over topology.graph["services"] (one entry per completed step)
    o.last_mean_gsnr = s.episode_services_processed + e->svc_list_extra > 0
        ? e->osnr_flushed / (double)(s.episode_services_processed + e->svc_list_extra) : 0.0;
    o.last_episode_disrupted = s.episode_disrupted_services;
    o.last_episode_defrag_cycles = s.episode_defrag_cycles;
    o.last_episode_service_reallocations = s.episode_service_reallocations;
}


// ---- measure_disruptions (envs/qrmsa.pyx:937-952) ---------------------------------------------------------------------
// After an accept: every running service that shares a link with the new one (the new one included: it is already in the
// running lists) and is not yet in the disrupted list gets its GSNR re-evaluated against all other running services; below
// its modulation's minimum_osnr (no margin) it joins the list. The list membership is the sign bit of the record's
// release time. Per measured service one pass over the whole service table (lanes over records).
// 1/GSNR (linear) of the RUNNING service in record iy, evaluated as if it sat at slot `sy` (its own slot, or a slot
// defragment() wants to move it to), against every other running service: itself is skipped like the service_id test
// of core/osnr.pyx:65. One pass over the whole service table, lanes over records. Uniform attenuation only.
template <bool R32>
__device__ __forceinline__ GnLin gn_service_acc(Ctx &c, int iy, int py, int sy, int ny) {
    const Params &P = c.P;
    uint64_t ym0, ym1;
    if (R32) { ym0 = c.sa[iy]; ym1 = 0; } else { ym0 = G(P.path_mask)[2 * py]; ym1 = G(P.path_mask)[2 * py + 1]; }
    const int c2 = 2 * sy + ny;
    double part = 0.0;
    for (int base = 0; base < c.active; base += kWave) {
        const int iz = base + c.lane;
        if (iz >= c.active || iz == iy) continue;
        if (P.track_ids && c.sq[iz] == c.sq[iy]) continue;      // `rs.service_id != service.service_id`, core/osnr.pyx:65
        const uint32_t a = c.sa[iz], b = c.sb[iz];
        uint64_t m0, m1;
        if (R32) { m0 = a & (uint32_t)ym0; m1 = 0; }
        else { int pk = a & 0xFFFF; m0 = G(P.path_mask)[2 * pk] & ym0; m1 = G(P.path_mask)[2 * pk + 1] & ym1; }
        if (!(m0 | m1)) continue;
        const int sk = rec_slot<R32>(a, b), nk = rec_n<R32>(a, b), mk = rec_mod<R32>(a, b);
        const int adi = abs((2 * sk + nk) - c2);
        double A, corr;
        if (nk <= P.tab_nmax) {
            const auto *t = G(reinterpret_cast<const double *>(P.pair_tab)) + 2 * ((nk - 1) * P.tab_stride + adi);
            A = t[0]; corr = c.phi[mk] * t[1];
        } else {
            double bk = P.slot_bw * nk, adf = (0.5 * P.slot_bw) * (double)adi, ck = P.alpha0_cl * bk;
            A = asinh_diff(ck * (adf + 0.5 * bk), ck * (adf - 0.5 * bk));
            corr = c.phi[mk] * (bk / adf);
        }
        double w1 = 0.0, w2 = 0.0;
        while (m0) { int l = __ffsll((unsigned long long)m0) - 1; m0 &= m0 - 1; w1 += c.lw[2 * (l)]; w2 += c.lw[2 * (l) + 1]; }
        while (m1) { int l = 64 + __ffsll((unsigned long long)m1) - 1; m1 &= m1 - 1; w1 += c.lw[2 * (l)]; w2 += c.lw[2 * (l) + 1]; }
        part += A * w1 - corr * w2;
    }
    if (c.lane == 0) part += G(P.path_w1)[py] * G(P.self_asinh)[ny];
    const double total = wave_sum(part);
    const double bw = P.slot_bw * ny;
    const double fc = P.f0 + (P.slot_bw * sy) + (P.slot_bw * (ny / 2.0));
    GnLin g;
    g.ase = (bw * fc * G(P.path_ase)[py]) * c.rp[0];
    g.nli = (G(P.nli_coef)[ny] * c.rp[1]) * total;
    return g;
}

// GSNR < minimum_osnr of modulation `my` (no margin)?  Linear-domain compare with the dB fallback band of qot_ok.
__device__ __forceinline__ int below_minimum_osnr(const Ctx &c, double acc, int my) {
    const double lim = c.lim0[my];
    int below;
    if (acc >= lim * (1.0 + 1e-9)) below = 1;
    else if (acc <= lim * (1.0 - 1e-9)) below = 0;
    else below = 10.0 * log10(1.0 / acc) < c.P.mod_thr[my];
    return uniform_i32(below);
}

template <bool R32>
__device__ __forceinline__ int measure_disruptions(Ctx &c, uint64_t nm0, uint64_t nm1) {
    const int keep_skip = c.skip_id;
    c.skip_id = -1;                                          // a listing, not a GN sum
    const int L = gn_build_list<R32>(c, nm0, nm1);          // services on the new service's links
    c.skip_id = keep_skip;
    int newly = 0;
    for (int j = 0; j < L; j++) {
        const int iy = c.list[j];
        const float ry = c.sr[iy];
        if (ry < 0.f) continue;                              // already in disrupted_services_list
        const uint32_t ay = c.sa[iy], by = c.sb[iy];
        const int py = uniform_i32(rec_path<R32>(ay, by)), sy = uniform_i32(rec_slot<R32>(ay, by));
        const int ny = uniform_i32(rec_n<R32>(ay, by)), my = uniform_i32(rec_mod<R32>(ay, by));
        const GnLin g = gn_service_acc<R32>(c, iy, py, sy, ny);
        if (below_minimum_osnr(c, g.ase + g.nli, my)) {      // osnr_svc < minimum_osnr (envs/qrmsa.pyx:947)
            if (c.lane == 0) c.sr[iy] = -ry;
            newly++;
        }
    }
    wave_sync();
    return newly;
}

// ---- defragment (envs/qrmsa.pyx:1545-1639) ----------------------------------------------------------------------------
// Running services in the order of topology.graph["running_services"] (= ascending service_id: appended when
// provisioned, never reordered). Each is offered the valid starts of its own path for its own slot count, lowest first,
// below its present slot (its own slots still count as occupied); the first whose GSNR - service moved there, itself
// excluded - is not below minimum_osnr (no margin) is taken: old [slot, slot+n+1) freed (clamped at S), new
// [start, start+n (+1 unless it ends at S)) occupied, Service.OSNR rewritten (it feeds the episode's mean GSNR).
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        uint64_t o = __shfl_xor((unsigned long long)v, m);
        v = o < v ? o : v;
    }
    return v;
}

template <bool R32>
__device__ __forceinline__ void defragment(Ctx &c, int num_services) {
    const Params &P = c.P;
    DevEnv *e = c.e;
    if (c.lane == 0) e->st.episode_defrag_cycles += 1;
    if (num_services == 0) num_services = 1000000;
    int moved = 0;
    long long last = -1;
    for (;;) {
        if (moved >= num_services) break;
        uint64_t key = ~0ull;                                 // (service_id << 16) | record index, smallest id above `last`
        for (int i = c.lane; i < c.active; i += kWave) {
            const uint64_t q = c.sq[i];
            if ((long long)q > last) { const uint64_t k = (q << 16) | (uint64_t)i; key = k < key ? k : key; }
        }
        key = wave_min_u64(key);
        if (key == ~0ull) break;
        const int iy = (int)(key & 0xFFFF);
        last = (long long)(key >> 16);
        const uint32_t ay = c.sa[iy], by = c.sb[iy];
        const int py = uniform_i32(rec_path<R32>(ay, by)), sy = uniform_i32(rec_slot<R32>(ay, by));
        const int ny = uniform_i32(rec_n<R32>(ay, by)), my = uniform_i32(rec_mod<R32>(ay, by));
        if (sy == 0) continue;
        const PathRef p = load_path(c, py);
        int rr = 1;
        uint64_t v = run_and(path_free_ext(c, p), rr, ny + 1);   // _get_candidates(available, n, S)
        v &= word_range(c.lane, 0, sy);                          // candidate >= initial_slot: skipped (:1583-1584)
        int target = -1;
        GnLin g;
        g.ase = g.nli = 0.0;
        for (;;) {
            const int s0 = first_set(v);
            if (s0 < 0) break;
            g = gn_service_acc<R32>(c, iy, py, s0, ny);
            if (!below_minimum_osnr(c, g.ase + g.nli, my)) { target = s0; break; }
            if (c.lane == (s0 >> 6)) v &= ~(1ull << (s0 & 63));
        }
        if (target < 0) continue;
        wave_sync();
        mark_links(c, p.hops, p.mylink, sy, sy + ny + 1, true);
        int end = target + ny; if (end < P.n_slots) end += 1;
        mark_links(c, p.hops, p.mylink, target, end, false);
        if (c.lane == 0) {
            uint32_t ra, rb;
            rec_pack<R32>(py, p.m0, target, ny, my, ra, rb);
            c.sa[iy] = ra; c.sb[iy] = rb;
            const double osnr = -10.0 * log10(g.ase + g.nli);
            e->osnr_flushed += osnr - c.so[iy];              // the Service object in topology.graph["services"] is updated
            c.so[iy] = osnr;
            e->st.episode_service_reallocations += 1;
            const int nlog = P.move_n[c.replica];            // what the compatibility view replays onto its Service objects
            if (nlog < ONGYM_MOVE_LOG) {
                ongym_move mv;
                mv.service_id = (int32_t)c.sq[iy]; mv.slot = target;
                mv.osnr = osnr; mv.ase = -10.0 * log10(g.ase); mv.nli = -10.0 * log10(g.nli);
                P.move_log[(size_t)c.replica * ONGYM_MOVE_LOG + nlog] = mv;
            }
            P.move_n[c.replica] = nlog + 1;
        }
        moved++;
        wave_sync();
    }
}

// Departures with defragmentation (envs/qrmsa.pyx:1113-1122): the due services leave one at a time in heap order
// (release time, then service_id), and defragment() may run after each of them. Release times are kept as float32
// (the reference compares them as `cdef float`); two due services whose double keys differ but round to the same float32
// are ordered by service_id here.
template <bool R32>
__device__ __forceinline__ void release_due_defrag(Ctx &c, float now) {
    const Params &P = c.P;
    for (;;) {
        uint64_t key = ~0ull;                                 // (float bits of the release time << 32) | service_id
        int mine = -1;
        for (int i = c.lane; i < c.active; i += kWave) {
            const float r = fabsf(c.sr[i]);
            if (r <= now) {
                const uint64_t k = ((uint64_t)__float_as_uint(r) << 32) | (uint64_t)c.sq[i];
                if (k < key) { key = k; mine = i; }
            }
        }
        const uint64_t best = wave_min_u64(key);
        if (best == ~0ull) break;
        const uint64_t bal = __ballot(key == best);
        const int idx = __builtin_amdgcn_readlane(mine, __ffsll((unsigned long long)bal) - 1);
        const uint32_t a = c.sa[idx], b = c.sb[idx];
        const int sk = rec_slot<R32>(a, b), nk = rec_n<R32>(a, b);
        if (R32) mark_mask(c, a, sk, sk + nk + 1, true);
        else {
            const int pk = a & 0xFFFF;
            const int hops = G(P.path_hops)[pk];
            const int mylink = (c.lane < hops) ? G(P.path_links)[pk * P.max_hops + c.lane] : 0;
            mark_links(c, hops, mylink, sk, sk + nk + 1, true);
        }
        const int lastrec = c.active - 1;
        if (c.lane == 0 && idx != lastrec) {
            c.sa[idx] = c.sa[lastrec]; c.sb[idx] = c.sb[lastrec]; c.sr[idx] = c.sr[lastrec];
            c.sq[idx] = c.sq[lastrec]; c.so[idx] = c.so[lastrec];
        }
        c.active = lastrec;
        wave_sync();
        if (P.defragmentation && (P.n_defrag_services == 0 || c.e->st.episode_services_processed % P.n_defrag_services == 0))
            defragment<R32>(c, P.n_defrag_services);
    }
}

// ---- one request: apply the choice (envs/qrmsa.pyx:838-1065) ----------------------------------------------------
// outcome: 0 = accept & provision, 1 = reject action, 2 = retry (slots busy), 3 = QoT error
template <bool R32, bool DEFRAG>
__device__ __forceinline__ void apply_step(Ctx &c, const Choice &ch, int outcome, ongym_step_rec *rec) {
    const Params &P = c.P;
    DevEnv *e = c.e;
    if (outcome == 2 || outcome == 3) {
        // slots not free: penalty, same request stays current (quirk Q5, :886-897); QoT error: ValueError (:925-929)
        if (c.lane == 0 && rec) {
            ongym_step_rec r;
            r.action = ch.action; r.route = -1; r.modulation = -1; r.slot = -1; r.nslots = 0;
            r.accepted = 0; r.terminated = 0; r.retry = 0; r.flags = (uint8_t)ch.flags;
            r.osnr = 0.0; r.ase = 0.0; r.nli = 0.0; r.reward = 0.0; r.active = c.active;
            if (outcome == 2) {
                double failed = (double)(e->st.episode_services_processed - e->st.episode_services_accepted) /
                                (double)e->st.episode_services_processed;
                r.reward = -3.0 * (1.0 + failed);   // reward(), :1266-1271
                r.retry = 1; r.flags |= ONGYM_F_BLOCKED_RESOURCES;
            } else {
                r.flags |= ONGYM_F_QOT_ERROR; r.osnr = -10.0 * log10(ch.g.ase + ch.g.nli);
                r.route = (int16_t)ch.route; r.slot = (int16_t)ch.slot;
            }
            *rec = r;
        }
        return;
    }
    int overflow = 0;
    if (outcome == 0 && c.active >= P.capacity) { outcome = 1; overflow = 1; }
    float rel = 0.f;
    if (outcome == 0) {
        // _provision_path (:1288-1325): occupy n slots + one guard slot unless the allocation ends at S
        int end = ch.slot + ch.n; if (end < P.n_slots) end += 1;
        mark_links<false>(c, ch.hops, ch.mylink, ch.slot, end, false);   // ordered by the wave_sync after the bookkeeping
    }
    STAMP(c, 5);
    if (c.lane == 0) {
        ongym_stats &s = e->st;
        ongym_step_rec r;
        r.action = ch.action; r.route = -1; r.modulation = -1; r.slot = -1; r.nslots = 0;
        r.accepted = 0; r.terminated = 0; r.retry = 0; r.flags = (uint8_t)ch.flags;
        r.osnr = 0.0; r.ase = 0.0; r.nli = 0.0; r.reward = 0.0; r.active = 0;
        double osnr = 0.0;
        if (outcome == 0) {
            // Service.OSNR = 10 log10(1/acc): per step only when a record is requested; the episode sum (mean_gsnr of
            // graph_load.py:181-185) accumulates the product of the acc's and takes one log10 per ~60 services
            double g[3] = {0.0, 0.0, 0.0};
            if (rec) { g[0] = -10.0 * log10(ch.g.ase + ch.g.nli); g[1] = -10.0 * log10(ch.g.ase); g[2] = -10.0 * log10(ch.g.nli); }
            if (DEFRAG) {   // defragment() rewrites Service.OSNR of the services it moves: keep it per record, sum it directly
                if (!rec) g[0] = -10.0 * log10(ch.g.ase + ch.g.nli);
                e->osnr_flushed += g[0];
                c.so[c.active] = g[0];
                c.sq[c.active] = (uint32_t)e->cur_id;   // Service.service_id (:1092), fixed when the request was drawn
            } else {
                c.osnr_prod *= (ch.g.ase + ch.g.nli);
                if (c.osnr_prod < 1e-250) flush_osnr(c);
            }
            rel = e->cur_at + e->cur_ht;   // float + float (:1329); compared as float32 (:1114-1115)
            uint32_t ra, rb;
            rec_pack<R32>(ch.path, ch.m0, ch.slot, ch.n, ch.mod, ra, rb);
            c.sa[c.active] = ra; c.sb[c.active] = rb;
            c.sr[c.active] = rel;
            s.services_accepted += 1; s.episode_services_accepted += 1;
            s.bit_rate_provisioned += (double)e->cur_br;
            s.episode_bit_rate_provisioned =
                (double)(int64_t)(s.episode_bit_rate_provisioned + (double)e->cur_br);   // :1319-1321
            s.episode_modulation_hist[ch.mod] += 1;
            s.total_accepted += 1;
            r.accepted = 1; r.route = (int16_t)ch.route; r.modulation = (int16_t)ch.mod; r.slot = (int16_t)ch.slot;
            r.nslots = (int16_t)ch.n; r.osnr = g[0]; r.ase = g[1]; r.nli = g[2];
            osnr = g[0];
            r.reward = 0.0;                       // reward() falls off the end when accepted (quirk Q1)
        } else {
            s.rejected += 1;                      // bl_reject (:865)
            r.reward = -6.0;                      // :992-995
            if (overflow) { r.flags |= ONGYM_F_OVERFLOW; s.flags |= ONGYM_F_OVERFLOW; }
        }
        s.total_steps += 1;
        e->have_request = 0;
        if (DEFRAG) {   // info["episode_defrag_cicles"], info["episode_service_realocations"] (:1008-1009)
            s.step_defrag_cycles = s.episode_defrag_cycles; s.step_service_reallocations = s.episode_service_reallocations;
            P.move_n[c.replica] = 0;
        }
        // the info dict is computed before the next request is drawn (:996-1050); the step terminates the episode
        // iff that draw makes episode_services_processed reach episode_length (:1056)
        if (s.episode_services_processed + 1 == P.episode_length) { flush_osnr(c); snapshot_terminal(c); }
        if (rec) *rec = r;
    }
    if (outcome == 0) {
        c.active++;
    }
    wave_sync();
    if (outcome == 0 && P.measure_disruptions) {
        const int newly = measure_disruptions<R32>(c, ch.m0, R32 ? 0 : G(P.path_mask)[2 * ch.path + 1]);
        if (newly && c.lane == 0) { e->st.disrupted_services += newly; e->st.episode_disrupted_services += newly; }
        wave_sync();
    }
    draw_next(c);                                 // first half of _next_service (:1079-1111)
    STAMP(c, 6);
    if (DEFRAG) release_due_defrag<R32>(c, e->cur_at);
    else release_due<R32>(c, e->cur_at);          // second half of _next_service (:1113-1122)
    STAMP(c, 7);
    int terminated = e->st.episode_services_processed == P.episode_length;
    if (c.lane == 0) {
        if (terminated) e->st.episodes_completed += 1;
        // graph_load.py:181-185 reads Service.OSNR after the loop, i.e. after the departures (and moves) of this
        // _next_service
        if (DEFRAG && terminated)
            P.env[c.replica].st.last_mean_gsnr = e->osnr_flushed / (double)(e->st.episode_services_processed - 1 + e->svc_list_extra);
        if (rec) { rec->active = c.active; rec->terminated = (uint8_t)terminated; }
    }
    c.active_sum += c.active;
    if (terminated && P.auto_reset) { wave_sync(); reset_env(c); }
}


// ---- observation() + action mask (envs/qrmsa.pyx:583-781; calculate_osnr_observation core/osnr.pyx:259-369) -----------
// For one path the interferer part of the NLI sum of a candidate centred at half-slot x is
//     F(x) = sum_k [ A(n_k, |x - c_k|) * w1_k - Phi_k * R(n_k, |x - c_k|) * w2_k ],
// a superposition of per-interferer profiles that does not depend on the candidate's own width: F is built ONCE per
// path over all 2S+1 centres (lanes over x -> the pair-table reads of one interferer are contiguous), then each of the
// M x (valid starts) candidates costs one LDS read + the O(1) ASE/self terms + one log10.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// Wave-wide reductions without the LDS crossbar: symmetric DPP exchanges inside each row of 16 lanes (as wave_sum), then
// the four row results through v_readlane; the result is wave-uniform.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += dpp_i32<0xB1>(v); v += dpp_i32<0x4E>(v); v += dpp_i32<0x141>(v); v += dpp_i32<0x140>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, dpp_i32<0xB1>(v)); v = max(v, dpp_i32<0x4E>(v)); v = max(v, dpp_i32<0x141>(v)); v = max(v, dpp_i32<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_f64<0xB1>(v)); v = fmax(v, dpp_f64<0x4E>(v)); v = fmax(v, dpp_f64<0x141>(v)); v = fmax(v, dpp_f64<0x140>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}


// One tile of <= 64 interferers (lane t holds interferer t: centre c2k, slots nk, summed link weights w1, Phi-weighted
// pw2) added to the field at up to NA x 64 needed centres whose partial sums stay in registers: the interferer's
// parameters are broadcast ONCE (6 v_readlane), then each 64-centre chunk costs |x - c|, one address op, one 16-byte
// gather from the pitch-2048 pair table and two FMAs.  No validity test is needed: a needed centre x = 2s + n comes from
// a VALID start, whose slots do not overlap any running service on a shared link, so |x - c_k| > n_k always; lanes
// beyond the list hold x = 0 and are not stored.
template <int NA>
__device__ __forceinline__ void field_tile(const Params &P, int lane, double *Fx, const uint16_t *xlist, int nxl, int g0,
                                           int c2k, int nk, double w1, double pw2, int tile_n) {
    const char __attribute__((address_space(1))) *tab = (const char __attribute__((address_space(1))) *)P.pair_tab2k;
    uint32_t xs[NA];
    double f[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) {
        const int i = g0 + a * kWave + lane;
        xs[a] = i < nxl ? xlist[i] : 0;
        f[a] = 0.0;
    }
    // TU interferers per round: TU x NA independent gathers are in flight before the first FMA needs one (the loop is bound by
    // the L2 round trip otherwise); a round past the end of the tile repeats the last interferer with zero weights
#ifndef ONGYM_OBS_TU
#define ONGYM_OBS_TU 2
#endif
    constexpr int TU = NA <= 4 ? 2 * ONGYM_OBS_TU : ONGYM_OBS_TU;
    for (int t = 0; t < tile_n; t += TU) {
        uint32_t cc[TU], key4[TU];
        double w1t[TU], pw2t[TU];
#pragma unroll
        for (int u = 0; u < TU; u++) {
            const int tt = min(t + u, tile_n - 1);
            const bool real = t + u < tile_n;
            cc[u] = (uint32_t)__builtin_amdgcn_readlane(c2k, tt);
            key4[u] = (uint32_t)(__builtin_amdgcn_readlane(nk, tt) - 1) << 15;      // row * 2048 entries * 16 bytes
            w1t[u] = real ? readlane_f64(w1, tt) : 0.0;
            pw2t[u] = real ? readlane_f64(pw2, tt) : 0.0;
        }
        double qa[TU][NA], qr[TU][NA];
#pragma unroll
        for (int u = 0; u < TU; u++)
#pragma unroll
            for (int a = 0; a < NA; a++) {
                const uint32_t adi = __builtin_amdgcn_sad_u16(xs[a], cc[u], 0);
                const double __attribute__((address_space(1))) *q =
                    (const double __attribute__((address_space(1))) *)(tab + (size_t)(key4[u] | (adi << 4)));
                qa[u][a] = q[0]; qr[u][a] = q[1];
            }
#pragma unroll
        for (int u = 0; u < TU; u++)
#pragma unroll
            for (int a = 0; a < NA; a++) f[a] += qa[u][a] * w1t[u] - qr[u][a] * pw2t[u];
    }
#pragma unroll
    for (int a = 0; a < NA; a++) {
        const int i = g0 + a * kWave + lane;
        if (i < nxl) Fx[xs[a]] += f[a];
    }
}

// self-channel term of a candidate of n slots on a path: sum over its links of w1_l * asinh(pi^2 |b2| B^2 / (4 alpha_l))
// (core/osnr.pyx:58-61); one table product when the attenuation is uniform
template <bool UNIFORM_ALPHA>
__device__ __forceinline__ double path_self_term(const Ctx &c, const PathRef &p, int n) {
    const Params &P = c.P;
    if (UNIFORM_ALPHA) return G(P.path_w1)[p.id] * G(P.self_asinh)[n];
    const double bw = P.slot_bw * n;
    return wave_sum(c.lane < p.hops ? c.lw[2 * p.mylink] * asinh(c.lsc[p.mylink] * (bw * bw)) : 0.0);
}

template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ void build_field(Ctx &c, const PathRef &p, uint64_t free_ext, const FieldLds &fl) {
    const Params &P = c.P;
    const int M = P.n_mods, S = P.n_slots, nx = 2 * S + 1, W = P.row_words;
    double *Fx = fl.Fx; uint64_t *Vw = fl.Vw; uint16_t *xlist = fl.xlist; uint8_t *needx = fl.needx;
    const int vs = fl.vs;
    STAMPW(c, 1);
    // valid starts of every modulation (run-AND words, lane w = word w) and the candidate centres x = 2s + n they
    // produce: the field is only needed there (a loaded network has few valid starts)
    for (int x = c.lane; x < nx + 1; x += kWave) { Fx[x] = 0.0; needx[x] = 0; }
    {
        uint64_t runs0 = free_ext;
        int r0 = 1;
        for (int mi = 0; mi < M; mi++) {
            const int n = uniform_i32(c.nreq[M - 1 - mi]);
            uint64_t v = 0;
            if (n > 0 && n <= S) {
                if (n + 1 < r0) { runs0 = free_ext; r0 = 1; }
                runs0 = run_and(runs0, r0, n + 1);
                v = runs0;
            }
            if (c.lane < vs) Vw[mi * vs + c.lane] = v;
        }
    }
    wave_sync();
    STAMPW(c, 3);
    for (int mi = 0; mi < M; mi++) {
        const int n = uniform_i32(c.nreq[M - 1 - mi]);
        const uint64_t vrow = Vw[mi * vs + min(c.lane, vs - 1)];        // lane i = word i
        for (int i = 0; i < W; i++) {
            const uint64_t w = readlane_u64(vrow, i);
            if (!w) continue;
            const int sl = i * 64 + c.lane;
            if (((w >> c.lane) & 1ull) && sl < S) needx[2 * sl + n] = 1;
        }
    }
    wave_sync();
    STAMPW(c, 4);
    int nxl = 0;
    for (int x0 = 0; x0 < nx; x0 += kWave) {
        const int x = x0 + c.lane;
        const bool need = x < nx && needx[x];
        const uint64_t bal = __ballot(need);
        if (need) xlist[nxl + __popcll((unsigned long long)(bal & lanes_below(c.lane)))] = (uint16_t)x;
        nxl += __popcll((unsigned long long)bal);
    }
    wave_sync();
    STAMPW(c, 5);
    // the interferer list is built only now: in k_observe's compact layout it overlays needx, which is dead from here on
    const int L = gn_build_list<R32>(c, p.m0, p.m1);
    STAMPW(c, 2);
    for (int base = 0; base < L; base += kWave) {
        const int j = base + c.lane;
        int c2k = 0, nk = 0;
        double w1 = 0.0, pw2 = 0.0;
        uint64_t km0 = 0, km1 = 0;      // shared links of interferer j (kept for the per-link attenuation form)
        double kphi = 0.0;
        if (j < L) {
            const int idx = c.list[j];
            const uint32_t a = c.sa[idx], b = c.sb[idx];
            const int sk = rec_slot<R32>(a, b);
            nk = rec_n<R32>(a, b);
            c2k = 2 * sk + nk;
            uint64_t m0, m1;
            if (R32) { m0 = a & (uint32_t)p.m0; m1 = 0; }
            else { int pk = a & 0xFFFF; m0 = G(P.path_mask)[2 * pk] & p.m0; m1 = G(P.path_mask)[2 * pk + 1] & p.m1; }
            if (!UNIFORM_ALPHA) { km0 = m0; km1 = m1; kphi = c.phi[rec_mod<R32>(a, b)]; }
            double w2 = 0.0;
            while (m0) { int l = __ffsll((unsigned long long)m0) - 1; m0 &= m0 - 1; w1 += c.lw[2 * (l)]; w2 += c.lw[2 * (l) + 1]; }
            while (m1) { int l = 64 + __ffsll((unsigned long long)m1) - 1; m1 &= m1 - 1; w1 += c.lw[2 * (l)]; w2 += c.lw[2 * (l) + 1]; }
            pw2 = c.phi[rec_mod<R32>(a, b)] * w2;
        }
        const int tile_n = min(kWave, L - base);
        if (!UNIFORM_ALPHA) {
            // per-link attenuation: the asinh difference depends on the link (core/osnr.pyx:68-84), no table and no summed
            // weights - every (centre, interferer, shared link) term is evaluated like gn_eval's generic branch does
            for (int x0 = 0; x0 < nxl; x0 += kWave) {
                const bool live = x0 + c.lane < nxl;
                const int x = live ? xlist[x0 + c.lane] : 0;
                double f = 0.0;
                for (int t = 0; t < tile_n; t++) {
                    const int cc = __builtin_amdgcn_readlane(c2k, t), nn = __builtin_amdgcn_readlane(nk, t);
                    uint64_t m0 = readlane_u64(km0, t), m1 = readlane_u64(km1, t);
                    const double ph = readlane_f64(kphi, t);
                    const int adi = abs(x - cc);
                    const bool on = adi > nn;              // |df| > Bk/2; centres overlapping the interferer are never needed
                    const double bk = P.slot_bw * nn, adf = (0.5 * P.slot_bw) * (double)(on ? adi : nn + 1);
                    const double hi = adf + 0.5 * bk, lo = adf - 0.5 * bk, corr = ph * (bk / adf);
                    double ft = 0.0;
                    while (m0 | m1) {
                        int l;
                        if (m0) { l = __ffsll((unsigned long long)m0) - 1; m0 &= m0 - 1; }
                        else { l = 64 + __ffsll((unsigned long long)m1) - 1; m1 &= m1 - 1; }
                        const double ck = c.lcl[l] * bk;
                        ft += asinh_diff(ck * hi, ck * lo) * c.lw[2 * l] - corr * c.lw[2 * l + 1];
                    }
                    if (on) f += ft;
                }
                if (live) Fx[x] += f;
            }
            STAMPW(c, 7);
            continue;
        }
        // all interferers of the tile inside the pair table? (always, unless a replayed trace carries a bit rate
        // beyond the configured ones) -> branch-free inner loop with 4 gathers in flight
        const bool all_tab = __ballot(j < L && nk > P.tab_nmax) == 0;
        const auto *tab = G(reinterpret_cast<const double *>(P.pair_tab));
        STAMPW(c, 6);
        if (all_tab && P.pair_tab2k) {      // the fast form: partial sums of up to 8 chunks of centres in registers
            for (int g0 = 0; g0 < nxl; g0 += 8 * kWave) {
                const int chunks = (nxl - g0 + kWave - 1) / kWave;
                if (chunks <= 4) field_tile<4>(P, c.lane, Fx, xlist, nxl, g0, c2k, nk, w1, pw2, tile_n);
                else field_tile<8>(P, c.lane, Fx, xlist, nxl, g0, c2k, nk, w1, pw2, tile_n);
            }
            STAMPW(c, 7);
            continue;
        }
        for (int x0 = 0; x0 < nxl; x0 += kWave) {
            const bool live = x0 + c.lane < nxl;
            const int x = live ? xlist[x0 + c.lane] : 0;
            double f = 0.0;
            if (all_tab) {
                for (int t = 0; t < tile_n; t += 4) {
                    int off[4]; double w1t[4], pw2t[4]; bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int tt = min(t + u, tile_n - 1);
                        const int cc = __builtin_amdgcn_readlane(c2k, tt), nn = __builtin_amdgcn_readlane(nk, tt);
                        w1t[u] = readlane_f64(w1, tt); pw2t[u] = readlane_f64(pw2, tt);
                        const int adi = abs(x - cc);
                        ok[u] = (t + u < tile_n) && adi > nn && adi < P.tab_stride;   // |df| > Bk/2, inside the band
                        off[u] = ok[u] ? 2 * ((nn - 1) * P.tab_stride + adi) : 0;
                    }
                    double A[4], R[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { A[u] = tab[off[u]]; R[u] = tab[off[u] + 1]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) if (ok[u]) f += A[u] * w1t[u] - R[u] * pw2t[u];
                }
            } else {
                for (int t = 0; t < tile_n; t++) {
                    const int cc = __builtin_amdgcn_readlane(c2k, t), nn = __builtin_amdgcn_readlane(nk, t);
                    const double w1t = readlane_f64(w1, t), pw2t = readlane_f64(pw2, t);
                    const int adi = abs(x - cc);
                    if (adi > nn) {   // |df| > Bk/2; positions overlapping the interferer are never valid starts
                        double A, R;
                        if (nn <= P.tab_nmax) { A = tab[2 * ((nn - 1) * P.tab_stride + adi)]; R = tab[2 * ((nn - 1) * P.tab_stride + adi) + 1]; }
                        else {
                            double bk = P.slot_bw * nn, adf = (0.5 * P.slot_bw) * (double)adi, ck = P.alpha0_cl * bk;
                            A = asinh_diff(ck * (adf + 0.5 * bk), ck * (adf - 0.5 * bk));
                            R = bk / adf;
                        }
                        f += A * w1t - R * pw2t;
                    }
                }
            }
            if (live) Fx[x] += f;
        }
    }
    wave_sync();
    wave_sync();
}

template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ void observe_env(Ctx &c, double *Fx, uint64_t *Vw, uint16_t *xlist, uint8_t *needx, int vs,
                                            float *obs, uint8_t *mask) {
    const Params &P = c.P;
    const int K = P.k_paths, Mall = P.n_mods, M = P.n_mods_consider, S = P.n_slots, N = P.n_nodes;
    const int nx = 2 * S + 1, W = P.row_words;
    const int obs_dim = 3 + K + K * M * 12;
    const long long nact = (long long)K * M * S + 1;
    DevEnv *e = c.e;
    if (!e->have_request) {
        for (int i = c.lane; i < obs_dim; i += kWave) obs[i] = 0.f;
        return;                                            // the mask is zero-filled by the host before the launch
    }
    const int src = e->cur_src, dst = e->cur_dst;
    const double br = (double)e->cur_br;
    if (c.lane == 0) {
        obs[0] = (float)(br / P.max_bit_rate);                       // :688
        obs[1] = (float)((double)src / (double)(N - 1));             // :682-686
        obs[2] = (float)((double)dst / (double)(N - 1));
        mask[nact - 1] = 1;                                          // :766
    }
    // ---- get_max_modulation_index (:543-581, called at :680): path-major, best format first; the first candidate whose
    // calculate_osnr reaches minimum_osnr + margin fixes max_modulation_idx = max(its format, modulations_to_consider - 1).
    // With modulations_to_consider == n_mods that is n_mods - 1 whatever the network holds, so the scan is skipped.
    int max_idx = Mall - 1;
    if (M < Mall) {
        max_idx = M - 1;
        bool found = false;
        for (int k = 0; k < K && !found; k++) {
            const int path = uniform_i32(G(P.pair_paths)[(src * N + dst) * K + k]);
            if (path < 0) break;
            PathRef p = load_path(c, path);
            const uint64_t free_ext = path_free_ext(c, p);
            build_field<UNIFORM_ALPHA, R32>(c, p, free_ext, FieldLds{Fx, Vw, xlist, needx, vs});
            const double pase = G(P.path_ase)[path];
            for (int mi = 0; mi < Mall && !found; mi++) {
                const int m = Mall - 1 - mi;
                const int n = uniform_i32(c.nreq[m]);
                if (n <= 0 || n > S) continue;
                const double bw = P.slot_bw * n, lim = c.lim[m];
                const double self = path_self_term<UNIFORM_ALPHA>(c, p, n), nlic = G(P.nli_coef)[n] * c.rp[1];
                bool pass = false;
                for (int i = 0; i < W; i++) {
                    const uint64_t w = Vw[mi * vs + i];
                    const int s0 = i * 64 + c.lane;
                    if (((w >> c.lane) & 1ull) && s0 < S) {
                        const double fc = P.f0 + (P.slot_bw * s0) + (P.slot_bw * (n / 2.0));
                        const double acc = (bw * fc * pase) * c.rp[0] + nlic * (self + Fx[2 * s0 + n]);
                        if (acc <= lim * (1.0 - 1e-9)) pass = true;                       // same decision as qot_ok
                        else if (acc < lim * (1.0 + 1e-9)) pass |= 10.0 * log10(1.0 / acc) >= P.mod_thr[m] + e->margin;
                    }
                }
                if (__ballot(pass)) { max_idx = max(m, M - 1); found = true; }
            }
        }
    }
    if (c.lane == 0) P.env[c.replica].st.max_modulation_idx = max_idx;     // the codec of the next step() is relative to it
    // the window of formats the observation describes: modulations[start : start + M], best first (:712-717)
    const int mod_start = max_idx <= 1 ? 0 : max(0, max_idx - (M - 1));
    for (int k = 0; k < K; k++) {
        const int path = G(P.pair_paths)[(src * N + dst) * K + k];
        float *frow = obs + 3 + K + k * M * 12;
        uint8_t *mrow = mask + (long long)k * M * S;
        if (c.lane == 0) obs[3 + k] = path >= 0 ? (float)P.path_len_norm[path] : 0.f;   // :701-705
        if (path < 0) {
            for (int i = c.lane; i < M * 12; i += kWave) frow[i] = -1.f;
            continue;                                      // the mask rows stay zero (filled by the host)
        }
        PathRef p = load_path(c, path);
        const uint64_t free_ext = path_free_ext(c, p);
        // ---- free slots on the path and its free blocks (:632-652): number of blocks, sum of squared lengths
        const uint64_t aw = (c.lane == (S >> 6)) ? (free_ext & ~(1ull << (S & 63))) : free_ext;
        const int tot = wave_sum_i32(c.lane < W ? __popcll((unsigned long long)aw) : 0);
        int nb_l = 0, len2_l = 0, prev_zero = -1;
        for (int i = 0; i < W; i++) {
            const uint64_t w = readlane_u64(aw, i);
            const uint64_t nxt = (i + 1 < W) ? (readlane_u64(aw, i + 1) & 1ull) : 0ull;
            const uint64_t ends = w & ~((w >> 1) | (nxt << 63));      // last slot of every free block
            if ((ends >> c.lane) & 1ull) {
                uint64_t below = ~w & lanes_below(c.lane);
                int pz = below ? (i * 64 + 63 - __clzll((unsigned long long)below)) : prev_zero;
                int len = (i * 64 + c.lane) - pz;
                nb_l += 1; len2_l += len * len;
            }
            if (~w) prev_zero = i * 64 + 63 - __clzll((unsigned long long)~w);
        }
        const int nb = wave_sum_i32(nb_l);
        const double len2 = (double)wave_sum_i32(len2_l);
        STAMPW(c, 8);
        // ---- interferer field F(x) at the needed centres + valid starts per modulation (shared builder)
        build_field<UNIFORM_ALPHA, R32>(c, p, free_ext, FieldLds{Fx, Vw, xlist, needx, vs});
        STAMPW(c, 9);
        // ---- per format of the window, best first (mod_list = reversed(modulations[start : start + M]), :716-717); the field
        // builder numbers its valid-start rows from the best of ALL formats: row fi = n_mods - 1 - m
        const double pase = G(P.path_ase)[path];
        // features that depend on the path only (:632-652, 660-663)
        const double Sd = (double)S, S1 = (double)(S - 1), inv_S = 1.0 / Sd, inv_S1 = 1.0 / S1;
        double mb = 0.0, sb = 0.0;
        if (nb > 0) {
            const double bm = (double)tot / nb;
            mb = ((bm - 4.0) / 4.0) / 100.0;                             // :646-648
            sb = sqrt(fmax(len2 / nb - bm * bm, 0.0)) / 100.0;
        }
        const float f_free = (float)(2.0 * ((double)tot - 0.5 * Sd) / Sd), f_free2 = (float)(2.0 * (((double)tot / Sd) - 0.5));
        // the wave-uniform sums of format mi are parked in lane mi; the 12 features of all formats are then finished in ONE
        // pass (lane = format) instead of once per format on 64 identical lanes
        int r_cnt = 0, r_smax = 0, r_n = -1;
        double r_ssum = 0.0, r_ssum2 = 0.0, r_osum = 0.0, r_osum2 = 0.0, r_omax = 0.0;
        for (int mi = 0; mi < M; mi++) {
            const int m = mod_start + M - 1 - mi;
            const int fi = Mall - 1 - m;
            const int n = uniform_i32(c.nreq[m]);
            uint8_t *mm = mrow + (long long)mi * S;        // zero-filled by the host before the launch
            if (n <= 0 || n > S) continue;                 // lane mi keeps r_n = -1: only the two path-level features are set
            const double thr = P.mod_thr[m], bw = P.slot_bw * n, inv_thr = 1.0 / fabs(thr);
            const double self = path_self_term<UNIFORM_ALPHA>(c, p, n), nlic = G(P.nli_coef)[n] * c.rp[1];
            // the valid starts of _get_candidates (:590), compacted: ascending slot indices in xlist (free again after the
            // field was built), so that the per-candidate arithmetic runs on dense lanes
            int cnt = 0;
            const uint64_t vrow = Vw[fi * vs + min(c.lane, vs - 1)];    // lane i = word i
            for (int i = 0; i < W; i++) {
                uint64_t w = readlane_u64(vrow, i);
                if (i == (S >> 6)) w &= ~(1ull << (S & 63));              // the virtual slot S is not a start
                if (i * 64 >= S) w = 0;
                if (!w) continue;
                const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(w >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)w, 0));
                if ((w >> c.lane) & 1ull) xlist[cnt + pre] = (uint16_t)(i * 64 + c.lane);
                cnt += __popcll((unsigned long long)w);
            }
            wave_sync();
            STAMPW(c, 10);
            int sum_l = 0, sum2_l = 0;
            double os_l = 0.0, os2_l = 0.0, omax_l = -1e300;
            for (int j0 = 0; j0 < cnt; j0 += kWave) {
                if (j0 + c.lane < cnt) {
                    const int s0 = xlist[j0 + c.lane];
                    sum_l += s0; sum2_l += s0 * s0;
                    const double fc = P.f0 + (P.slot_bw * s0) + (P.slot_bw * (n / 2.0));
                    const double acc = (bw * fc * pase) * c.rp[0] + nlic * (self + Fx[2 * s0 + n]);
                    const double gsnr = -10.0 * log10(acc);
                    // np.round((gsnr - thr) / abs(thr), 10), osnr.pyx:368 (reciprocals instead of the two divisions: the
                    // value is reported as float32, and the sign test below could only differ within 1e-16 of -0.5e-10)
                    const double nv = rint(((gsnr - thr) * inv_thr) * 1e10) * 1e-10;
                    os_l += nv; os2_l += nv * nv; omax_l = fmax(omax_l, nv);
                    if (nv >= 0.0) mm[s0] = 1;                                           // :759-763
                }
            }
            const int smax = cnt > 0 ? (int)xlist[cnt - 1] : 0;          // ascending: the last one
            wave_sync();                                                 // xlist is rewritten for the next format
            STAMPW(c, 11);
            const double ssum = (double)wave_sum_i32(sum_l), ssum2 = (double)wave_sum_i32(sum2_l);
            const double osum = wave_sum(os_l), osum2 = wave_sum(os2_l), omax = wave_max_f64(omax_l);
            STAMPW(c, 12);
            if (c.lane == mi) {
                r_cnt = cnt; r_smax = smax; r_n = n;
                r_ssum = ssum; r_ssum2 = ssum2; r_osum = osum; r_osum2 = osum2; r_omax = omax;
            }
        }
        {
            double mean_s = 0.0, std_s = 0.0, om = 0.0, ov = 0.0, best = 0.0;
            if (r_cnt > 0) {
                const double inv_cnt = 1.0 / (double)r_cnt;
                mean_s = r_ssum * inv_cnt;
                std_s = sqrt(fmax(r_ssum2 * inv_cnt - mean_s * mean_s, 0.0));
                om = r_osum * inv_cnt;
                ov = fmax(r_osum2 * inv_cnt - om * om, 0.0);
                best = fmax(r_omax, 0.0);                                // osnr_best starts at 0.0 (:604,622)
            }
            const double adj = ((double)r_n - 5.5) / 3.5;
            const bool real = r_n > 0;                                   // a format whose slot count does not fit: zeros but 4, 10
            const float fv[12] = {real ? (float)((double)r_cnt * inv_S) : 0.f, real ? (float)(mean_s * inv_S1) : 0.f,
                                  real ? (float)(std_s * inv_S1) : 0.f, real ? (float)(adj > 0.0 ? adj : 0.0) : 0.f,
                                  real ? f_free : f_free2, real ? (float)mb : 0.f, real ? (float)sb : 0.f,
                                  real ? (float)best : 0.f, real ? (float)om : 0.f, real ? (float)ov : 0.f, f_free2,
                                  real ? (float)((double)r_smax * inv_S1) : 0.f};
            if (c.lane < M) {
#pragma unroll
                for (int q = 0; q < 12; q++) frow[c.lane * 12 + q] = fv[q];
            }
            STAMPW(c, 13);
        }
    }
}

// ---- heuristic_highest_snr (heuristics/heuristics.py:272-328) ---------------------------------------------------------
// Every valid start of every (path, modulation) pair is evaluated (one LDS read of the path's interferer field per
// candidate); among those that clear threshold + margin the highest GSNR wins, first one in (path, modulation best
// first, slot ascending) order on ties — the reference's strict `osnr > best_osnr`.
template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ void policy_highest_snr(Ctx &c, int src, int dst, double launch_power, double margin, Choice &ch) {
    const Params &P = c.P;
    const int M = P.n_mods, S = P.n_slots, max_mod = M - 1, W = P.row_words;
    ch.action = P.k_paths * M * S; ch.route = -1; ch.mod = -1; ch.slot = -1; ch.n = 0; ch.hops = 0; ch.mylink = 0;
    ch.path = -1; ch.m0 = 0; ch.g.ase = ch.g.nli = 0.0; ch.flags = 0;
    int any_res = 0, any_osnr = 0;
    double best_osnr = -INFINITY;
    for (int k = 0; k < P.k_paths; k++) {
        int path = k == 0 ? c.pre_id : G(P.pair_paths)[(src * P.n_nodes + dst) * P.k_paths + k];
        if (path < 0) break;
        PathRef p;
        if (k == 0) { p.id = path; p.hops = c.pre_hops; p.mylink = c.pre_mylink; p.m0 = c.pre_m0; p.m1 = c.pre_m1; p.ase = c.pre_ase; p.w1 = c.pre_w1; }
        else p = load_path(c, path);
        c.paths_tried++; c.path_hops += p.hops;
        const uint64_t free_ext = path_free_ext(c, p);
        build_field<UNIFORM_ALPHA, R32>(c, p, free_ext, c.fl);
        const double pase = G(P.path_ase)[path];
        for (int m = max_mod; m >= 0; m--) {
            const int mi = max_mod - m;
            const int n = uniform_i32(c.nreq[m]);
            if (n <= 0) continue;
            const double thr = P.mod_thr[m] + margin, lim = c.lim[m], bw = P.slot_bw * n;
            const double self = path_self_term<UNIFORM_ALPHA>(c, p, n), nlic = G(P.nli_coef)[n] * c.rp[1];
            int nvalid = 0;
            for (int i = 0; i < W; i++) {
                const uint64_t w = c.fl.Vw[mi * c.fl.vs + i];
                if (!w) continue;
                const int s = i * 64 + c.lane;
                const bool valid = ((w >> c.lane) & 1ull) && s < S;
                nvalid += __popcll((unsigned long long)w);
                double v = -INFINITY;
                bool fail = false;
                if (valid) {
                    const double fc = P.f0 + (P.slot_bw * s) + (P.slot_bw * (n / 2.0));
                    const double acc = (bw * fc * pase) * c.rp[0] + nlic * (self + c.fl.Fx[2 * s + n]);
                    const double osnr = -10.0 * log10(acc);
                    bool ok;
                    if (acc <= lim * (1.0 - 1e-9)) ok = true;
                    else if (acc >= lim * (1.0 + 1e-9)) ok = false;
                    else ok = 10.0 * log10(1.0 / acc) >= thr;
                    if (ok) v = osnr; else fail = true;
                }
                if (__ballot(fail)) any_osnr = 1;
                const double vmax = wave_max_f64(v);
                if (vmax > best_osnr + 1e-9) {                            // strict, beyond the rounding noise (exact ties of the reference: equal routes
                                                                          // on an empty network; see eval_cands in ongym_fast.hpp): the first maximum wins
                    const uint64_t bal = __ballot(v == vmax);
                    const int ln = __ffsll((unsigned long long)bal) - 1;
                    best_osnr = vmax;
                    ch.route = k; ch.mod = m; ch.slot = i * 64 + ln; ch.n = n; ch.path = path;
                }
            }
            if (nvalid == 0) any_res = 1;
            c.gn_evals += nvalid;
        }
    }
    if (ch.route < 0) {
        if (any_osnr) any_res = 0;
        ch.flags = (any_res ? ONGYM_F_BLOCKED_RESOURCES : 0) | (any_osnr ? ONGYM_F_BLOCKED_OSNR : 0);
        return;
    }
    // the winner's ASE / NLI split and path registers (same evaluation as the other policies)
    PathRef p = load_path(c, ch.path);
    ch.hops = p.hops; ch.mylink = p.mylink; ch.m0 = p.m0;
    ch.action = ch.route * M * S + (max_mod - ch.mod) * S + ch.slot;
    const int L = gn_build_list<R32>(c, p.m0, p.m1);
    GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, ch.slot, ch.n);
    c.gn_evals--;   // not a candidate evaluation of the heuristic
    ch.g.ase = uniform_f64(g.ase); ch.g.nli = uniform_f64(g.nli);
    (void)launch_power;
}

__device__ __forceinline__ void load_state(Ctx &c) {
    const Params &P = c.P;
    // DevEnv: 8-byte words by lanes
    const uint64_t *ge = reinterpret_cast<const uint64_t *>(P.env + c.replica);
    uint64_t *le = reinterpret_cast<uint64_t *>(c.e);
    for (int i = c.lane; i < (int)(kEnvHotBytes / 8); i += kWave) le[i] = ge[i];
    for (int i = c.lane; i < P.n_links; i += kWave) {
        c.lw[2 * (i)] = P.link_w1[i]; c.lw[2 * (i) + 1] = P.link_w2[i];
        if (!P.uniform_alpha) { c.lcl[i] = P.link_cl[i]; c.lsc[i] = P.link_selfc[i]; }
    }
    int words = P.n_links * P.row_words;
    const uint64_t *g = P.occ + (size_t)c.replica * words;
    for (int i = c.lane; i < words; i += kWave) c.occ[i] = g[i];
    wave_sync();
    c.active = c.e->st.active;
    c.osnr_prod = c.e->osnr_prod > 0.0 ? c.e->osnr_prod : 1.0;
    c.pre_id = -1;
    if (c.e->have_request) prefetch_first_path(c, c.e->cur_src, c.e->cur_dst);
    // per-replica acceptance limits in the linear domain (see qot_ok)
    if (c.lane < P.n_mods) c.lim[c.lane] = pow(10.0, -(P.mod_thr[c.lane] + c.e->margin) / 10.0);
    if ((P.measure_disruptions || P.track_ids) && c.lane < P.n_mods) c.lim0[c.lane] = pow(10.0, -P.mod_thr[c.lane] / 10.0);
    if (c.lane < kMaxMods) c.phi[c.lane] = c.lane < P.n_mods ? P.mod_phi53[c.lane] : 0.0;
    if (c.lane == 0) { c.rp[0] = 1.0 / c.e->launch_power; c.rp[1] = c.e->launch_power * c.e->launch_power; }
    // slots needed by the current request (kept in LDS between requests, recomputed on load)
    {
        int nr = 0;
        if (c.lane < P.n_mods) {
            nr = (int)ceil((double)c.e->cur_br / ((double)P.mod_se[c.lane] * P.nslots_width));
            c.nreq[c.lane] = nr;
        }
        set_request_coefs(c, nr, c.e->launch_power * c.e->launch_power);
    }
    size_t off = (size_t)c.replica * P.capacity;
    for (int i = c.lane; i < c.active; i += kWave) { c.sa[i] = P.svc_a[off + i]; c.sb[i] = P.svc_b[off + i]; c.sr[i] = P.svc_r[off + i]; }
    if (P.track_ids)
        for (int i = c.lane; i < c.active; i += kWave) { c.sq[i] = P.svc_q[off + i]; c.so[i] = P.svc_o[off + i]; }
    c.skip_id = (P.track_ids && c.e->have_request) ? c.e->cur_id : -1;
    wave_sync();
}

__device__ __forceinline__ void store_state(Ctx &c) {
    const Params &P = c.P;
    int terms = c.lane_terms;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) terms += __shfl_xor(terms, m);
    if (c.lane == 0) {
        c.e->osnr_prod = c.osnr_prod;
        c.e->st.episode_osnr_sum = c.e->osnr_flushed + (c.osnr_prod != 1.0 ? -10.0 * log10(c.osnr_prod) : 0.0);
        c.e->st.active = c.active;
        c.e->st.total_gn_evals += c.gn_evals;
        c.e->st.total_gn_shortcuts += c.gn_skips;
        c.e->st.total_interferer_terms += terms;
        c.e->st.total_paths_tried += c.paths_tried;
        c.e->st.total_path_hops += c.path_hops;
        c.e->st.total_active_sum += c.active_sum;
    }
    wave_sync();
    uint64_t *ge = reinterpret_cast<uint64_t *>(P.env + c.replica);
    const uint64_t *le = reinterpret_cast<const uint64_t *>(c.e);
    for (int i = c.lane; i < (int)(kEnvHotBytes / 8); i += kWave) ge[i] = le[i];
    int words = P.n_links * P.row_words;
    uint64_t *g = P.occ + (size_t)c.replica * words;
    for (int i = c.lane; i < words; i += kWave) g[i] = c.occ[i];
    size_t off = (size_t)c.replica * P.capacity;
    for (int i = c.lane; i < c.active; i += kWave) { P.svc_a[off + i] = c.sa[i]; P.svc_b[off + i] = c.sb[i]; P.svc_r[off + i] = c.sr[i]; }
    if (P.track_ids)
        for (int i = c.lane; i < c.active; i += kWave) { P.svc_q[off + i] = c.sq[i]; P.svc_o[off + i] = c.so[i]; }
}

}  // namespace ongym
