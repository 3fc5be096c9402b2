// ongym_fast.hpp — the lean fused policy + step kernels (k_fast<M64, REC, ENT, WAVES, TRACE, POL, WIDE>): same path and same results
// as the generic k_run, built around ISSUE-SLOT economy.  tools/ubench/issue_rates.hip measured on MI355X (cycles per
// wave-instruction per SIMD, >= 2 waves): 32-bit VALU 2.5, fp64 / 64-bit shift / DPP mov / v_readlane / v_alignbit 4.2,
// SALU 4.3, ds_read_b32 ~9.  The round-1 kernel spent 726 VALU + 625 SALU per request; at 5 waves/SIMD both pipes were
// ~2/3 busy, i.e. it was issue-bound on both.  What these kernels do differently (reference semantics unchanged:
// envs/qrmsa.pyx:838-1122, core/osnr.pyx:21-142, and per policy heuristics/heuristics.py:923-966 first fit, :547-627 load
// balancing, :272-328 highest SNR, :330-414 lowest fragmentation - the four heuristics of graph_load.py:116-125):
//   * requests are drawn 64 at a time (lane i = request index base+i of the same counter-based stream,
//     include/ongym_traffic.h) or read 64 at a time from a replayed trace, and kept in three VGPRs; a step pops one with
//     v_readlane.  The float32 clock chain stays serial (one v_add_f32 per request).
//   * no DevEnv in LDS and no lane-0 read-modify-write chains: wave-uniform counters are plain (scalar) variables, the
//     per-modulation / per-bit-rate histograms one lane-distributed VGPR (v_cmp + v_addc); everything is folded into
//     DevEnv in memory at the end of the launch and around the (rare) terminal step.
//   * the slot bitmap is handled as 32-bit words (lane w = word w): a run-AND step is DPP wave_shl:1 + v_alignbit_b32 +
//     v_and_b32; bits are set / cleared by ds_or_b32 / ds_and_b32 with EXEC = the path's link mask (lane l = link l).
//   * per (request, path) one fp64 FMA + compare over lanes (lane 8*bitrate + m = modulation m) removes every modulation
//     whose ASE + self-channel lower bound already fails at slot 0 (exact: the bound only grows with the slot index, and
//     no interferer term is negative — Params.ase_shortcut); the modulation loop walks the surviving bits.  The blocking
//     FLAGS of a rejected request (1 % of the steps) are recomputed exactly by a second, plain pass.
//   * the GN model caches the candidate path's interferers in registers once (centre, table row, link weights summed,
//     Phi folded in), so each further modulation of the same path costs |c - c_k|, one 16-byte table gather and two
//     fp64 FMAs per interferer; the acceptance test runs lane-parallel against per-lane limits (no readlane of doubles).
//   * records are 8 bytes in LDS (link mask | centre, modulation, slots-1, path) + the float32 release time; unused
//     entries are neutral (mask 0, release +inf), so no scan needs a bounds test.
//   * policies that look at every route fetch the routes' ids and records together (lane k = k-th route) and examine the
//     routes in the order that makes the FIRST route that serves the request the answer (ascending load / score);
//     policies that evaluate many starts of a (route, format) put the candidates on the lanes (eval_cands).
//   * WIDE = false (the narrow units, chosen on the host when no slot count of the traffic table exceeds 32): a run-AND step is
//     one shift of 1..31 bits, a mark touches two words, and the > 32-slot release path does not exist.
//   * the units are compiled with -mllvm -disable-machine-licm (__graft_entry__.FAST_UNIT_FLAGS): hoisting the constants of cold
//     code out of the step loop cost 53 spilled SGPRs + 13 spilled VGPRs; and wave-uniform values that must stay in a scalar
//     register across the loop are pinned by hand (ep_len).  Round 3 measured that this kernel pays per ISSUED INSTRUCTION
//     (~0.3 % per scalar instruction on the per-request path) and not per hidden latency: hence s_bitset0 / s_bfm_b64 /
//     v_mad_u32_u24 / ds_write2_b32 in place of the generic sequences, counters derived at store time, and vz.
// The state in HBM is the generic kernels' (same arrays, same record codec), converted on load / store: every other entry
// point keeps working on the same environment, and a launch may be split anywhere.
//
// Eligibility (checked on the host in build(), ongym_hip.hip; lean_policy() there decides per launch): policy-step mode,
// device request generator or a host trace whose bit rates all come from the configured table, discrete bit rates (<= 8,
// integer-valued), uniform attenuation, ase_shortcut, no defragmentation / disruptions / id tracking,
// modulations_to_consider == n_mods, n_links <= 52, n_nodes <= 64, every slot count of the traffic table <= min(512,
// tab_nmax), 2S+1 < 2048, and the policy's LDS block within 160 KiB.
#pragma once
#include "ongym_device.hpp"

namespace ongym {

constexpr int kTabPitch = 2048;      // pair-table row pitch (2S+1 <= 2047)
constexpr int kNearHalfSlots = 32;   // candidate-lane evaluation: interferers this close to the pass's candidates are staged first

// ---- path record (8 dwords, 32-byte aligned: one s_load_dwordx8) -------------------------------------------------------
struct PathRec {
    uint32_t hops, mask_lo, mask_hi, id;   // mask = links of the path (bit l = link l), n_links <= 52
    double ase, w1;                        // path_ase[id], path_w1[id]
};
static_assert(sizeof(PathRec) == 32, "PathRec must be 32 bytes");
__device__ __forceinline__ PathRec load_path_rec(const __attribute__((address_space(4))) PathRec *t, int path) {
    PathRec r;      // field by field: the constant address space has no copy constructor; still one s_load_dwordx8
    // (an unsigned 32-bit byte offset: base + offset is formed by the scalar load itself, not by a 64-bit add before it)
    const __attribute__((address_space(4))) PathRec *e =
        (const __attribute__((address_space(4))) PathRec *)((const __attribute__((address_space(4))) char *)t + ((uint32_t)path << 5));
    r.hops = e->hops; r.mask_lo = e->mask_lo; r.mask_hi = e->mask_hi; r.id = e->id;
    r.ase = e->ase; r.w1 = e->w1;
    return r;
}
struct TabPair { double x, y; };
__device__ __forceinline__ TabPair load_pair(const char __attribute__((address_space(1))) *tab, uint32_t byte_off) {
    const double __attribute__((address_space(1))) *p = (const double __attribute__((address_space(1))) *)(tab + (size_t)byte_off);
    TabPair t;
    t.x = p[0]; t.y = p[1];
    return t;
}

// Read-only tables are addressed through the constant address space: a wave-uniform index then always becomes a scalar
// load (the compiler cannot prove that the kernel's own global stores leave a plain global pointer's target alone).
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T *KC(const T *p) {
    return (const __attribute__((address_space(4))) T *)p;
}

__host__ __device__ inline bool fast_policy_has_xlist(int policy) {
    return policy == ONGYM_POLICY_HIGHEST_SNR || policy == ONGYM_POLICY_LOWEST_FRAGMENTATION;
}
__host__ __device__ inline size_t fast_lds_bytes(int n_links, int row_words, int capacity, bool m64, int n_slots = 0,
                                                 int policy = ONGYM_POLICY_FIRST_FIT) {
    size_t b = ((size_t)n_links * row_words * 8 + 15) & ~(size_t)15;   // occ  u32 [E][2W]
    b += (size_t)(n_links + 1) * 16 + 64;                               // lw (w1,w2) [E + a zero entry] | phi [8]
    b += (size_t)(capacity + 1) * 8;                                    // rec {a,b}, + one neutral entry
    b += (size_t)capacity * (4 + 2);                                    // rr | list
    b += 48;                                                            // cold wave-uniform state (5 doubles)
    b = (b + 15) & ~(size_t)15;
    if (fast_policy_has_xlist(policy)) {
        b += ((size_t)(n_slots + 1) * 2 + 15) & ~(size_t)15;            // xlist: candidate starts, compacted
        b += 64 * (8 + 16);                                             // stage: 64 interferers {centre, row} | {w1, Phi w2}
    }
    if (policy == ONGYM_POLICY_LOWEST_FRAGMENTATION) b += ((size_t)(n_slots + 1) * 8 + 15) & ~(size_t)15;   // plogp table
    return b;
}

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ float rlf(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// word of lane+1 (0 for lane 63): DPP wave_shl:1
__device__ __forceinline__ uint32_t next_word32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, true);
}

// run-AND on 32-bit words (see run_and): bit s of word w set iff slots [32w+s, 32w+s+m) are all free.  WIDE = false: the caller
// guarantees m <= 63 (every slot count of the configuration is <= 32), so every shift min(r, m - r) is in 1..31: one DPP move +
// v_alignbit + v_and and four scalar instructions per step.
template <bool WIDE>
__device__ __forceinline__ uint32_t run_and32(uint32_t x, int &r, int m) {
    if (!WIDE || m <= 63) {
        while (r < m) {
            const int s = min(r, m - r);
            x &= __builtin_amdgcn_alignbit(next_word32(x), x, s);
            r += s;
        }
        return x;
    }
    while (r < m) {
        const int s = min(r, m - r);
        uint32_t y = x;
        int t = s;
        while (t >= 32) { y = next_word32(y); t -= 32; }      // whole-word part of the shift (slot counts > 32 only)
        if (t) y = __builtin_amdgcn_alignbit(next_word32(y), y, t);
        x &= y;
        r += s;
    }
    return x;
}

__device__ __forceinline__ double wave_min_f64(double v) {     // as wave_max_f64 (ongym_device.hpp); wave-uniform result
    v = fmin(v, dpp_f64<0xB1>(v)); v = fmin(v, dpp_f64<0x4E>(v)); v = fmin(v, dpp_f64<0x141>(v)); v = fmin(v, dpp_f64<0x140>(v));
    return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// Wave-wide fp64 sum for eval_one (every lane ends up with a usable copy of the total).
//   default            : 4 symmetric DPP exchanges inside the rows of 16 lanes, then the rows combined with DPP row_bcast:15 /
//                        row_bcast:31 (total in lane 63, one v_readlane pair): 22 vector instructions, +1.7 % (first fit) over
//   ONGYM_X_ROW_SUM    : wave_sum (ongym_device.hpp): the 4 row sums through 8 v_readlane + 3 adds (31 vector instructions)
//   ONGYM_X_MFMA_SUM   : two v_mfma_f64_16x16x4_f64 against a matrix of ones + three adds (5 issue slots instead of 31).
//                        Measured on MI355X in round 3: 0.89x (first fit) - an fp64 16x16x4 MFMA occupies the matrix pipe for
//                        64 clocks on gfx950 (fp64 matrix rate = vector rate), two dependent ones cost more than the 31 VALU.
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double wave_sum_eval(double v) {
#if defined(ONGYM_X_MFMA_SUM)
    // A[i][k] = lane 16k + i, B = 1: D[i][j] = sum of lanes {i, i+16, i+32, i+48}; a lane's four accumulators hold four different
    // rows i and the four 16-lane groups hold all sixteen, so s = d0+d1+d2+d3 per group, and a second MFMA adds the four groups
    const double4_t z = {0.0, 0.0, 0.0, 0.0};
    const double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, z, 0, 0, 0);
    const double s = (d[0] + d[1]) + (d[2] + d[3]);
    const double4_t t = __builtin_amdgcn_mfma_f64_16x16x4f64(s, 1.0, z, 0, 0, 0);
    return t[0];
#elif !defined(ONGYM_X_ROW_SUM)
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);                 // every lane of a row holds the row's sum
    union { double d; int i[2]; } a, b;
    a.d = v;                                // rows 1 and 3 += lane 15 of the row before (row_bcast:15, row_mask 0b1010)
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x142, 0xA, 0xF, false);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x142, 0xA, 0xF, false);
    v += b.d;
    a.d = v;                                // rows 2 and 3 += lane 31 (row_bcast:31, row_mask 0b1100): row 3 holds the total
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x143, 0xC, 0xF, false);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x143, 0xC, 0xF, false);
    v += b.d;
    return readlane_f64(v, 63);
#else
    return wave_sum(v);
#endif
}

__device__ __forceinline__ float wave_min_f32(float v) {       // wave-uniform result
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
    // rows combined as in wave_sum_eval: rows 1, 3 with lane 15 of the row before, rows 2, 3 with lane 31 (the other rows keep v)
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xA, 0xF, false)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xC, 0xF, false)));
    return rlf(v, 63);
}

__device__ __forceinline__ int first_set32(uint32_t x) {
    const uint64_t bal = __ballot(x != 0);
    if (!bal) return -1;
    const int fl = __builtin_ctzll(bal);
    return fl * 32 + __builtin_ctz(rl(x, fl));
}

// EXEC-masked LDS operations: the lanes of `mask` (lane l = link l) apply `val` to the word at `addr`.  One asm statement
// each; EXEC is all ones before and after (the kernel's control flow is wave-uniform around every call site).
// (s_and_saveexec keeps the caller's EXEC: two scalar instructions around the DS operation, no VALU compare.)
__device__ __forceinline__ void lds_and_lanes(uint64_t mask, uint32_t addr, uint32_t val) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_and_b32 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "s"(mask), "v"(addr), "v"(val) : "memory", "scc");
}
__device__ __forceinline__ void lds_or_lanes(uint64_t mask, uint32_t addr, uint32_t val) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_or_b32 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "s"(mask), "v"(addr), "v"(val) : "memory", "scc");
}
__device__ __forceinline__ void lds_and2_lanes(uint64_t mask, uint32_t addr, uint32_t v0, uint32_t v1) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_and_b32 %2, %3\n\tds_and_b32 %2, %4 offset:4\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "s"(mask), "v"(addr), "v"(v0), "v"(v1) : "memory", "scc");
}
__device__ __forceinline__ void lds_or2_lanes(uint64_t mask, uint32_t addr, uint32_t v0, uint32_t v1) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_or_b32 %2, %3\n\tds_or_b32 %2, %4 offset:4\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "s"(mask), "v"(addr), "v"(v0), "v"(v1) : "memory", "scc");
}
// (the 8-byte record as two dwords: ds_write2_b32 takes them from two VGPRs, no 64-bit value has to be assembled first)
__device__ __forceinline__ void lds_write_lane0(uint32_t addr64, uint32_t lo, uint32_t hi, uint32_t addr32, uint32_t v32) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, 1\n\tds_write2_b32 %1, %2, %3 offset1:1\n\tds_write_b32 %4, %5\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(addr64), "v"(lo), "v"(hi), "v"(addr32), "v"(v32) : "memory", "scc");
}
__device__ __forceinline__ void lds_write_lanes01(uint32_t addr64, uint32_t lo, uint32_t hi, uint32_t addr32, uint32_t v32) {   // lanes 0 and 1, per-lane operands
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, 3\n\tds_write2_b32 %1, %2, %3 offset1:1\n\tds_write_b32 %4, %5\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(addr64), "v"(lo), "v"(hi), "v"(addr32), "v"(v32) : "memory", "scc");
}
__device__ __forceinline__ void lds_write_lane0_b32(uint32_t addr32, uint32_t v32) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(addr32), "v"(v32) : "memory", "scc");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) {   // byte address inside the workgroup's LDS
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}
__device__ __forceinline__ uint32_t word_mask32(int w, int lo, int hi) {     // bits [lo, hi) that fall into 32-bit word w
    const int a = max(lo - 32 * w, 0), b = min(hi - 32 * w, 32);
    if (b <= a) return 0u;
    return (b - a == 32) ? ~0u : (((1u << (b - a)) - 1u) << a);
}

// record word b:   centre 2*slot+n (11 bits) | modulation << 11 | (n-1) << 14 (9 bits) | path << 23 (9 bits)
// with M64 (33..41 links): centre (11) | modulation << 11 | (n-1) << 14 (9 bits) | link-mask bits 32..40 << 23.
// The path id is not in the M64 record: a route is identified by its link set (the two directions of a node pair share their
// Path objects, topology.pyx:310-355, i.e. their ids), and the store at the end of a launch looks the id up in a hash table
// built at create (Params.path_hash_*; a table with two ids for one link set keeps the generic kernel).
constexpr uint32_t kM64HiBits = 9;                   // links 32..40 of the mask live in the record word
__device__ __forceinline__ uint32_t fast_pack_b(int slot, int n, int mod, int path) {
    return (uint32_t)(2 * slot + n) | ((uint32_t)mod << 11) | ((uint32_t)(n - 1) << 14) | ((uint32_t)(path & 0x1FF) << 23);
}
__device__ __forceinline__ uint32_t fast_pack_b64(int slot, int n, int mod, uint32_t mask_hi) {
    return (uint32_t)(2 * slot + n) | ((uint32_t)mod << 11) | ((uint32_t)(n - 1) << 14) | (mask_hi << 23);
}
template <bool M64> __device__ __forceinline__ uint32_t rec_nm1(uint32_t y) { return (y >> 14) & 0x1FFu; }   // slots - 1
__host__ __device__ inline uint32_t path_hash_slot(uint64_t key, int bits) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - bits)); }


// 64 requests at once: lane i evaluates ongym_draw_request (include/ongym_traffic.h) for request index base + i, except the
// float32 clock add, which needs the previous arrival and is done when the request is popped.
__device__ __forceinline__ void fast_refill(const Params &P, const DevEnv *ge, uint64_t base, int lane, float &rq_iat, float &rq_ht,
                                         uint32_t &rq_pk) {
    ONGYM_NO_CONTRACT
    const int N = P.n_nodes, K = P.k_paths;
    const uint64_t key = ge->rng_key;
    const float mean_iat_f = ge->mean_iat_f;
    const uint64_t ctr = (base + (uint64_t)lane) * ONGYM_DRAWS_PER_REQUEST;
    const double u0 = ongym_uniform(key, ctr + 0), u1 = ongym_uniform(key, ctr + 1), u2 = ongym_uniform(key, ctr + 2),
                 u3 = ongym_uniform(key, ctr + 3), u4 = ongym_uniform(key, ctr + 4);
    rq_iat = -ongym_logf_det(1.0 - u0) * mean_iat_f;          // ongym_expovariate_f
    rq_ht = -ongym_logf_det(1.0 - u1) * P.mean_holding_f;
    const double total = G(P.node_cum)[N - 1];
    const double xs = u2 * total;
    int src = 0;
    for (int i = 0; i < N - 1; i++) src += (G(P.node_cum)[i] <= xs) ? 1 : 0;      // == ongym_bisect (cum is non-decreasing)
    const double hi_s = G(P.node_cum)[src];
    const double lo_s = src > 0 ? G(P.node_cum)[src - 1] : 0.0;
    const double w_s = hi_s - lo_s;
    double x = u3 * (total - w_s);
    if (x >= lo_s) x += w_s;
    int dst = 0;
    for (int i = 0; i < N - 1; i++) dst += (G(P.node_cum)[i] <= x) ? 1 : 0;
    if (dst == src) dst = (src + 1 < N) ? src + 1 : src - 1;
    const int nb = P.n_bit_rates;
    const double xb = u4 * G(P.bit_rate_cum)[nb - 1];
    int bi = 0;
    for (int i = 0; i < nb - 1; i++) bi += (G(P.bit_rate_cum)[i] <= xb) ? 1 : 0;
    const int p0 = G(P.pair_paths)[(src * N + dst) * K];
    rq_pk = (uint32_t)src | ((uint32_t)dst << 6) | ((uint32_t)bi << 12) | ((uint32_t)p0 << 15);
}

// The same ring from a replayed trace (ongym_set_requests with bit rates of the configured table, checked on the host):
// lane i holds trace entry base + i; `rq_iat` then carries the ABSOLUTE arrival time.  Entries past the end are never popped.
__device__ __forceinline__ void fast_refill_trace(const Params &P, int replica, uint64_t base, int lane, float &rq_at, float &rq_ht,
                                                  uint32_t &rq_pk) {
    const long long idx = (long long)base + lane;
    rq_at = 0.f; rq_ht = 0.f; rq_pk = 0;
    if (idx < P.trace_n) {
        const ongym_request q = P.trace[(long long)replica * P.trace_n + idx];
        rq_at = q.arrival_time; rq_ht = q.holding_time;
        int bi = 0;
        for (int b = 0; b < P.n_bit_rates; b++) if ((float)G(P.bit_rates)[b] == q.bit_rate) bi = b;
        const int p0 = G(P.pair_paths)[((int)q.source * P.n_nodes + (int)q.destination) * P.k_paths];
        rq_pk = (uint32_t)q.source | ((uint32_t)q.destination << 6) | ((uint32_t)bi << 12) | ((uint32_t)p0 << 15);
    }
}

template <bool M64, bool REC, int ENT, bool TRACE, int POL = ONGYM_POLICY_FIRST_FIT, bool WIDE = true>
__device__ __forceinline__ void fast_run(const Params &P, int nsteps, ongym_step_rec *out, unsigned char *smem) {
    ONGYM_NO_CONTRACT
#ifdef ONGYM_STAMPS
    unsigned long long stamp_acc[ONGYM_NSTAMPS] = {0}, stamp_last = __builtin_amdgcn_s_memtime();
#define FSTAMP(idx) do { __builtin_amdgcn_s_waitcnt(0); unsigned long long _t = __builtin_amdgcn_s_memtime(); stamp_acc[idx] += _t - stamp_last; stamp_last = _t; } while (0)
#else
#define FSTAMP(idx) do { } while (0)
#endif
    const int lane = threadIdx.x, replica = blockIdx.x;
    const int E = P.n_links, RW = P.row_words * 2, C = P.capacity, S = P.n_slots, M = P.n_mods, K = P.k_paths, N = P.n_nodes;
    DevEnv *const ge = P.env + replica;

    // ---- LDS carve-up ----
    uint32_t *const occ = reinterpret_cast<uint32_t *>(smem);
    size_t o = ((size_t)E * RW * 4 + 15) & ~(size_t)15;
    double *const lw = reinterpret_cast<double *>(smem + o); o += (size_t)(E + 1) * 16;      // lw[2E], lw[2E+1] = 0
    double *const phi = reinterpret_cast<double *>(smem + o); o += 64;
    uint2 *const rec = reinterpret_cast<uint2 *>(smem + o); o += (size_t)(C + 1) * 8;           // rec[C]: neutral entry
    float *const rr = reinterpret_cast<float *>(smem + o); o += (size_t)C * 4;
    uint16_t *const list = reinterpret_cast<uint16_t *>(smem + o); o += (size_t)C * 2;
    // wave-uniform state that is touched once in ~60 steps or less lives in LDS, not in registers:
    // [0] bit_rate_requested [1] bit_rate_provisioned [2] episode_bit_rate_requested [3] episode_bit_rate_provisioned at the
    // start of the launch / episode (the launch adds counts x bit rate), [4] osnr_flushed
    double *const cold = reinterpret_cast<double *>(smem + ((o + 7) & ~(size_t)7));
    // only with fast_policy_has_xlist: xlist u16[S+1] | st_ck uint2[64] | st_w double2[64] | (LOWEST_FRAGMENTATION) plt f64[S+1]
    const size_t o_x = fast_lds_bytes(E, P.row_words, C, M64);
    uint16_t *const xlist = reinterpret_cast<uint16_t *>(smem + o_x);
    uint2 *const st_ck = reinterpret_cast<uint2 *>(smem + o_x + (((size_t)(S + 1) * 2 + 15) & ~(size_t)15));
    double2 *const st_w = reinterpret_cast<double2 *>(st_ck + 64);
    double *const plt = reinterpret_cast<double *>(st_w + 64);
    const uint32_t occ_base = lds_addr(occ), rec_base = lds_addr(rec), rr_base = lds_addr(rr);
    // A zero the compiler cannot see through.  Wave-uniform integer arithmetic that only feeds LDS addresses and data is cheaper
    // on the vector pipe (2.5 cycles per instruction, ~60 % busy) than on the scalar pipe (4.3 cycles, ~70 % busy, and its
    // results would need v_mov to become DS operands anyway): adding `vz` to one operand keeps such a chain in VGPRs.
    uint32_t vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));

    // ---- load: constants, bitmap, records (generic codec -> lean codec) ----
    for (int i = lane; i <= E; i += kWave) { lw[2 * i] = i < E ? G(P.link_w1)[i] : 0.0; lw[2 * i + 1] = i < E ? G(P.link_w2)[i] : 0.0; }
    if (lane < kMaxMods) phi[lane] = lane < M ? P.mod_phi53[lane] : 0.0;
    if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION)
        for (int i = lane; i <= S; i += kWave) plt[i] = G(P.plogp)[i];
    {
        const uint32_t *g = reinterpret_cast<const uint32_t *>(P.occ + (size_t)replica * E * P.row_words);
        for (int i = lane; i < E * RW; i += kWave) occ[i] = g[i];
    }
    int active = uniform_i32(ge->st.active);
    {
        const size_t off = (size_t)replica * C;
        for (int i = lane; i <= C; i += kWave) {
            uint2 ab = make_uint2(0u, 0u);
            float r = INFINITY;
            uint32_t hi = 0;
            if (i < active) {
                const uint32_t a = P.svc_a[off + i], b = P.svc_b[off + i];
                r = P.svc_r[off + i];
                int path, slot, n, mod;
                if (P.rec32) { ab.x = a; slot = rec_slot<true>(a, b); n = rec_n<true>(a, b); mod = rec_mod<true>(a, b); path = rec_path<true>(a, b); }
                else {
                    slot = rec_slot<false>(a, b); n = rec_n<false>(a, b); mod = rec_mod<false>(a, b); path = rec_path<false>(a, b);
                    const uint64_t m0 = G(P.path_mask)[2 * path];
                    ab.x = (uint32_t)m0;
                    hi = (uint32_t)(m0 >> 32);
                }
                ab.y = M64 ? fast_pack_b64(slot, n, mod, hi) : fast_pack_b(slot, n, mod, path);
            }
            rec[i] = ab;
            if (i < C) rr[i] = r;
        }
    }

    // ---- per-lane tables: lane q = 8*bit_rate_index + modulation ----
    const double lp = uniform_f64(ge->launch_power), margin = uniform_f64(ge->margin);
    const double rp0 = 1.0 / lp, lp2 = lp * lp;
    int t_n, t1_n = 0;
    double t_nlic, t_selfa, t_lim_lo, t_lim_hi;
    double t1_nlic = 0.0, t1_selfa = 0.0, t1_lim_lo = -1.0, t1_lim_hi = -1.0;
    {
        const int qb = lane >> 3, qm = lane & 7;
        const bool valid = qb < P.n_bit_rates && qm < M;
        const int at_ = qb * kMaxMods + qm;
        t_n = valid ? G(P.nreq_tab)[at_] : 0;
        t_nlic = valid ? G(P.req_coef)[2 * at_] * lp2 : 0.0;
        t_selfa = valid ? G(P.req_coef)[2 * at_ + 1] : 0.0;
        const double lim = pow(10.0, -(P.mod_thr[qm < M ? qm : 0] + margin) / 10.0);      // same expression as load_state
        // band of qot_ok; lanes without a usable slot count can never pass (their limits are negative)
        const bool usable = valid && t_n >= 1 && t_n <= S;
        t_lim_lo = usable ? lim * (1.0 - 1e-9) : -1.0;
        t_lim_hi = usable ? lim * (1.0 + 1e-9) : -1.0;
        // LOWEST_FRAGMENTATION searches (and evaluates) at slots + 1 (heuristics.py:357): the same tables one slot wider
        if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION) {
            const bool usable1 = valid && t_n >= 1 && t_n + 1 <= S;
            t1_n = valid ? t_n + 1 : 0;
            t1_nlic = usable1 ? G(P.nli_coef)[t_n + 1] * lp2 : 0.0;
            t1_selfa = usable1 ? G(P.self_asinh)[t_n + 1] : 0.0;
            t1_lim_lo = usable1 ? lim * (1.0 - 1e-9) : -1.0;
            t1_lim_hi = usable1 ? lim * (1.0 + 1e-9) : -1.0;
        }
    }
    const uint32_t t_pk = t_n >= 1 ? (uint32_t)t_n | ((uint32_t)(lane & 7) << 11) | ((uint32_t)(t_n - 1) << 14) : 0u;
    // the tables the policy's search uses
    constexpr bool kLF = POL == ONGYM_POLICY_LOWEST_FRAGMENTATION;
    const int &w_n = kLF ? t1_n : t_n;
    const double &w_nlic = kLF ? t1_nlic : t_nlic, &w_selfa = kLF ? t1_selfa : t_selfa;
    const double &w_lim_lo = kLF ? t1_lim_lo : t_lim_lo, &w_lim_hi = kLF ? t1_lim_hi : t_lim_hi;
    // 1/GSNR of a candidate (start s) on a route is  (w_c1 + w_cb * s) * path_ase  +  w_nlic * (interferer sum)  +  w_c2 * path_w1
    // for the lane's (bit rate, format): the launch-invariant factors are kept per lane, so that a route costs three multiplications
    // (lane_fac) and an evaluation two FMAs after its sum.  w_c1 and w_c2 are computed in the order the bound always used; the
    // ASE term of an evaluation is now associated as (bw (f0 + h)/P + bw slot_bw/P * s) * ase instead of (bw (f0 + slot_bw s + h) ase)/P
    // (envs/qrmsa.pyx:901-905): a difference of a few ulp, far inside qot_ok's 1e-9 band and the 1e-9 the statistics are held to.
    struct LaneFac { double c1, cb, c2; };
    auto lane_fac = [&](int n_, double nlic_, double selfa_) -> LaneFac {
        const double bw = P.slot_bw * n_, h = P.slot_bw * (n_ / 2.0);
        LaneFac f;
        f.c1 = (bw * (P.f0 + h)) * rp0;
        f.cb = (bw * P.slot_bw) * rp0;
        f.c2 = nlic_ * selfa_;
        return f;
    };
    const LaneFac w_fac = lane_fac(w_n, w_nlic, w_selfa);

    // ---- wave-uniform state from DevEnv ----
    uint64_t req_base = readlane_u64(ge->req_index, 0);        // ring lane i = request req_base + i
    float v_at = uniform_f32((float)ge->st.current_time);  // current_time is always a float32 value ((double)at, qrmsa.pyx:1081)
    int epp = uniform_i32((int)ge->st.episode_services_processed);
    int erej = uniform_i32((int)ge->st.rejected);
    if (lane == 0) {
        cold[0] = ge->st.bit_rate_requested; cold[1] = ge->st.bit_rate_provisioned;
        cold[2] = ge->st.episode_bit_rate_requested; cold[3] = ge->st.episode_bit_rate_provisioned;
        cold[4] = ge->osnr_flushed;
    }
    double osnr_prod = uniform_f64(ge->osnr_prod > 0.0 ? ge->osnr_prod : 1.0);
    int cnt = (lane >= 8 && lane < 16) ? (int)ge->st.episode_modulation_hist[lane - 8] : 0;
    int lane_terms = 0;
    // current request (drawn by whoever ran before: k_reset, k_run or a previous k_fast launch)
    float cur_ht = uniform_f32(ge->cur_ht);
    // source | destination << 6, unpacked where a route beyond the first is looked up (9 % of the requests with first fit)
    uint32_t cur_sd = (uint32_t)uniform_i32(ge->cur_src) | ((uint32_t)uniform_i32(ge->cur_dst) << 6);
    auto pair_base = [&]() -> int {      // index of the pair's first route (behind an optimisation barrier: computed where it is used)
        uint32_t sd = cur_sd;
        asm volatile("" : "+s"(sd));
        return ((int)(sd & 63u) * N + (int)(sd >> 6)) * K;
    };
    int cur_bi;
    {
        const float br = uniform_f32(ge->cur_br);
        const uint64_t hit = __ballot(lane < P.n_bit_rates && (float)G(P.bit_rates)[min(lane, P.n_bit_rates - 1)] == br);
        cur_bi = hit ? __builtin_ctzll(hit) : 0;
    }
    int cur_p0 = uniform_i32(KC(P.pair_paths)[pair_base()]);
    const __attribute__((address_space(4))) PathRec *const path_recs = KC(reinterpret_cast<const PathRec *>(P.path_rec));
    // launch deltas
    // (not counted per step: requests popped = the advance of the request index; steps = the advance of the step loop, except for a
    //  replayed trace, whose exhausted steps do not count; modulations settled by the bound = M x routes examined - d_feas with first fit)
    int d_acc = 0, d_rej = 0, d_steps = 0, d_evals = 0, d_skips = 0, d_feas = 0, d_paths = 0, d_hops = 0, d_flags = 0, d_episodes = 0;
    int it = 0, it_base = 0;                // step loop counter; steps before it_base are already in DevEnv
    uint32_t d_active_sum = 0;              // < 2^32: the host splits a launch so that steps x capacity stays below (fast_launch)

    float rq_iat, rq_ht;
    uint32_t rq_pk;
    int rq_pos = 0;
    auto refill = [&]() {
        if (TRACE) fast_refill_trace(P, replica, req_base, lane, rq_iat, rq_ht, rq_pk);
        else fast_refill(P, ge, req_base, lane, rq_iat, rq_ht, rq_pk);
        rq_pos = 0;
    };
    bool have = true;           // a current request exists (a replayed trace can run out: every further step is a no-op)
    refill();
    wave_sync();

    // ---- fold the launch state into DevEnv (memory); idempotent: deltas are zeroed once added ----
    auto store_env = [&](int it_upto) {
        int terms = lane_terms;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) terms += __shfl_xor(terms, m);
        lane_terms = 0;
        int hist[8], nreq_b[8], nacc_b[8];
        float brs[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { hist[i] = (int)rl((uint32_t)cnt, 8 + i); nreq_b[i] = (int)rl((uint32_t)cnt, 16 + i); nacc_b[i] = (int)rl((uint32_t)cnt, 24 + i); brs[i] = i < P.n_bit_rates ? (float)KC(P.bit_rates)[i] : 0.f; }
        if (lane == 0) {
            ongym_stats &s = ge->st;
            const uint64_t req_now = req_base + (uint64_t)rq_pos;
            s.services_processed += (long long)(req_now - ge->req_index);
            ge->req_index = req_now;
            s.current_time = (double)v_at;
            ge->cur_at = v_at; ge->cur_ht = cur_ht; ge->cur_br = brs[cur_bi]; ge->cur_src = (int)(cur_sd & 63u); ge->cur_dst = (int)(cur_sd >> 6);
            ge->cur_id = epp - 1; ge->have_request = have ? 1 : 0;
            s.episode_services_processed = epp;
            const int acc_ = TRACE ? d_acc : it_upto - it_base - d_rej;     // every step of the device generator accepts or rejects
            s.services_accepted += acc_; s.total_accepted += acc_;
            s.rejected = erej;
            double rq = 0.0, pv = 0.0;       // exact: bit rates are integer-valued (eligibility), counts are small integers
            int eacc = 0;
            for (int i = 0; i < 8; i++) { s.episode_modulation_hist[i] = hist[i]; eacc += hist[i]; rq += (double)nreq_b[i] * (double)brs[i]; pv += (double)nacc_b[i] * (double)brs[i]; }
            s.episode_services_accepted = eacc;
            s.bit_rate_requested = cold[0] + rq; s.episode_bit_rate_requested = cold[2] + rq;
            s.bit_rate_provisioned = cold[1] + pv; s.episode_bit_rate_provisioned = cold[3] + pv;
            s.episode_osnr_sum = cold[4] + (osnr_prod != 1.0 ? -10.0 * log10(osnr_prod) : 0.0);
            ge->osnr_flushed = cold[4]; ge->osnr_prod = osnr_prod;
            s.episodes_completed += d_episodes;
            s.total_steps += TRACE ? d_steps : it_upto - it_base;
            s.total_gn_evals += d_evals;
            s.total_gn_shortcuts += POL == ONGYM_POLICY_FIRST_FIT ? M * d_paths - d_feas : d_skips;
            s.total_interferer_terms += terms; s.total_paths_tried += d_paths; s.total_path_hops += d_hops;
            s.total_active_sum += (long long)d_active_sum;
            s.active = active; s.flags |= d_flags;
        }
        d_acc = d_rej = d_steps = d_evals = d_skips = d_feas = d_paths = d_hops = d_flags = d_episodes = 0;
        d_active_sum = 0;
        it_base = it_upto;
    };

    // next request: pop the ring, advance the float32 clock (envs/qrmsa.pyx:1079-1111)
    auto pop_request = [&]() {
        if (rq_pos == kWave) { req_base += kWave; refill(); }
        if (TRACE && (long long)(req_base + (uint64_t)rq_pos) >= P.trace_n) {      // trace exhausted (see draw_next)
            have = false; d_flags |= ONGYM_F_NO_REQUEST;
            return;
        }
        const uint32_t pk = rl(rq_pk, rq_pos);
        if (TRACE) v_at = rlf(rq_iat, rq_pos);          // the trace carries absolute float32 arrival times
        else v_at = v_at + rlf(rq_iat, rq_pos);         // at = float32(current_time + expovariate)
        cur_ht = rlf(rq_ht, rq_pos);
        rq_pos++;
        cur_sd = pk & 0xFFFu; cur_bi = (pk >> 12) & 7; cur_p0 = (int)pk >> 15;            // the first route's id, sign-extended (-1: the pair has no route)
        epp++;
        cnt += (lane == 16 + cur_bi) ? 1 : 0;           // bit_rate_requested, by bit rate
    };

    // AND of the free bitmaps of a path's links, extended by the virtual free slot S (see path_free_ext); the links come
    // from the path's mask (their order does not matter)
    auto path_and = [&](uint64_t links64) -> uint32_t {
        const int wl = min(lane, RW - 1);
        uint32_t x = lane < RW ? ~0u : 0u;
        if (M64) {
            uint64_t links = links64;
            const uint32_t v_word = occ_base + (uint32_t)wl * 4u, v_rw4 = vz + (uint32_t)RW * 4u;
            while (links) {
                const int l = __builtin_ctzll(links);
                asm("s_bitset0_b64 %0, %1" : "+s"(links) : "s"(l));
                const uint32_t a = __umul24((uint32_t)l, v_rw4) + v_word;
                x &= *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)a;
            }
        } else {
            // per link: s_ff1, s_bitset0 (one instruction instead of the add/and pair of links &= links - 1), and the row address
            // lane offset + l * row bytes as one v_mad_u32_u24 on the vector pipe
            uint32_t links = (uint32_t)links64;
            const uint32_t v_word = occ_base + (uint32_t)wl * 4u, v_rw4 = vz + (uint32_t)RW * 4u;
            while (links) {
                const int l = __builtin_ctz(links);
                asm("s_bitset0_b32 %0, %1" : "+s"(links) : "s"(l));
                const uint32_t a = __umul24((uint32_t)l, v_rw4) + v_word;
                x &= *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)a;
            }
        }
        if (lane == (S >> 5)) x |= 1u << (S & 31);
        return x;
    };

    // set (free) or clear the slots [lo, hi) on the links of `mask`.  Up to 33 slots (n <= 32) touch at most two words:
    // both are updated at once (the second one by a neutral operand when the range does not reach it; it may then be the
    // word after the row, which is left unchanged).  Longer ranges take the word loop.
    auto mark = [&](uint64_t mask, int lo, int hi, bool free_) {
        const uint32_t v_rowaddr = occ_base + (uint32_t)lane * (uint32_t)RW * 4u;    // lane l = link l: its bitmap row
        const int len = hi - lo;
#ifndef ONGYM_X_MARK_LOOP
        if (!WIDE || len <= 33) {
            uint64_t m;                                      // ((1 << len) - 1) << (lo & 31) is one scalar instruction (len <= 33)
            asm("s_bfm_b64 %0, %1, %2" : "=s"(m) : "s"(len), "s"(lo & 31));
            const uint32_t a = v_rowaddr + (uint32_t)(lo >> 5) * 4u;
            if (free_) lds_or2_lanes(mask, a, (uint32_t)m, (uint32_t)(m >> 32));
            else lds_and2_lanes(mask, a, ~(uint32_t)m, ~(uint32_t)(m >> 32));
            return;
        }
#endif
        for (int w = lo >> 5; w <= (hi - 1) >> 5; w++) {
            const uint32_t m = word_mask32(w, lo, hi);
            if (free_) lds_or_lanes(mask, v_rowaddr + (uint32_t)w * 4u, m);
            else lds_and_lanes(mask, v_rowaddr + (uint32_t)w * 4u, ~m);
        }
    };

    // the same for slots [lo, lo+len), len <= 33, with `lo` and `len` living in VGPRs (wave-uniform values): two words per link,
    // EXEC from the link mask held in a VGPR pair (departures) — no scalar arithmetic at all
    auto mark_v = [&](uint32_t m_lo, uint32_t m_hi, uint32_t lo, uint32_t len, bool free_) {
        const uint64_t m = ((1ull << len) - 1ull) << (lo & 31u);
        const bool mine = lane < 32 ? ((m_lo >> lane) & 1u) : (M64 && ((m_hi >> (lane - 32)) & 1u));
        if (mine) {
            uint32_t *w = occ + (uint32_t)lane * (uint32_t)RW + (lo >> 5);          // lane l = link l: its row
            if (free_) { __hip_atomic_fetch_or(w, (uint32_t)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                         __hip_atomic_fetch_or(w + 1, (uint32_t)(m >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
            else { __hip_atomic_fetch_and(w, ~(uint32_t)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                   __hip_atomic_fetch_and(w + 1, ~(uint32_t)(m >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
        }
    };

    int ep_len = P.episode_length;          // held in a scalar register (the lean units are built without loop-invariant code motion)
    asm volatile("" : "+s"(ep_len));
    const char __attribute__((address_space(1))) *const tab = (const char __attribute__((address_space(1))) *)P.pair_tab2k;
    const char __attribute__((address_space(1))) *const tabp = (const char __attribute__((address_space(1))) *)P.pair_tabp;

    // LOWEST_FRAGMENTATION: score of a route = 0.33 * mean link entropy + 0.33 * cuts + 0.34 * rss over the route's link rows
    // (heuristics.py:375-384, utils.pyx:61-107), bit for bit — see lf_route_score (ongym_scored.hpp) for why that is possible.
    // The trial allocation paints free slots free, so the score depends on the rows of the route's links only, and a link's
    // share of it (entropy of its OCCUPIED runs, their number, the sums of their lengths and squared lengths) only changes when
    // a service is provisioned on or released from that link: lane l keeps link l's four numbers and walks its row again (the
    // same left-to-right walk on 32-bit words, p*log(p) from the LDS copy of Params.plogp) only when the row changed since —
    // about five links per request instead of the rows of all k routes.  A route's score then adds the cached entropies in
    // the order of path.links (ds_bpermute by link index) and the three integer sums over the lanes of its link mask.
    double ls_ent = 0.0;
    int ls_cuts = 0, ls_sl = 0, ls_sq = 0;
    bool ls_dirty = POL == ONGYM_POLICY_LOWEST_FRAGMENTATION && lane < E;
    auto link_stats_update = [&]() {
        if (!__ballot(ls_dirty)) return;
        if (ls_dirty) {
            double ent = 0.0;
            int cuts = 0, sl = 0, sq = 0;
            const uint32_t *row = occ + (size_t)lane * RW;
            int carry = 0;                                   // length of the run that is open at the current position
            auto close = [&](int len) {
                ent += plt[len];                             // entropy += p * math.log(p)
                cuts++;
                sq += len * len;
                sl += len;
            };
            for (int w = 0; w < RW; w++) {
                const int nb = min(32, S - 32 * w);
                if (nb <= 0) break;
                uint32_t z = ~row[w];
                if (nb < 32) z &= (1u << nb) - 1u;
                int pos = 0;
                while (pos < nb) {
                    const uint32_t rest = z >> pos;
                    if (carry > 0 || (rest & 1u)) {
                        const uint32_t inv = ~rest;
                        int len = inv ? __builtin_ctz(inv) : 32;
                        len = min(len, nb - pos);
                        carry += len; pos += len;
                        if (pos < nb) { if (carry > 0) close(carry); carry = 0; }
                    } else {
                        pos += rest ? __builtin_ctz(rest) : 32;
                    }
                }
            }
            if (carry > 0) close(carry);
            ls_ent = ent != 0.0 ? -ent : 0.0;                // utils.pyx:79
            ls_cuts = cuts; ls_sl = sl; ls_sq = sq;
            ls_dirty = false;
        }
    };
    auto bperm_f64 = [&](double v, int src_lane) -> double {       // v of lane src_lane (every lane active)
        union { double d; int i[2]; } a_, b_;
        a_.d = v;
        b_.i[0] = __builtin_amdgcn_ds_bpermute(src_lane << 2, a_.i[0]);
        b_.i[1] = __builtin_amdgcn_ds_bpermute(src_lane << 2, a_.i[1]);
        return b_.d;
    };
    auto route_score = [&](int path, int hops, uint64_t pmask) -> double {
        const int lh = lane < hops ? G(P.path_links)[path * P.max_hops + lane] : 0;    // lane h: the route's h-th link
        const double ent_h = bperm_f64(ls_ent, lh);
        double se = 0.0;                                     // sum(entropies): left to right, starting from int 0
        for (int h = 0; h < hops; h++) se = __dadd_rn(se, readlane_f64(ent_h, h));
        se = se / (double)hops;
        const bool in = (pmask >> lane) & 1ull;
        const int tc = wave_sum_i32(in ? ls_cuts : 0), tsl = wave_sum_i32(in ? ls_sl : 0), tsq = wave_sum_i32(in ? ls_sq : 0);
        const double rss = tsl == 0 ? 0.0 : sqrt((double)tsq) / (double)tsl;
        // every product and sum rounded on its own (fp_barrier: no fma), as Python computes 0.33 * se + 0.33 * cuts + 0.34 * rss
        const double t1 = fp_barrier(0.33 * se), t2 = fp_barrier(0.33 * (double)tc), t3 = fp_barrier(0.34 * rss);
        return uniform_f64(fp_barrier(t1 + t2) + t3);
    };

    if (!uniform_i32(ge->have_request)) {            // never reset: every step is a no-op (same as k_run)
        if (lane == 0) {
            ge->st.flags |= ONGYM_F_NO_REQUEST;
            if (REC) for (int it = 0; it < nsteps; ++it) {
                ongym_step_rec r;
                memset(&r, 0, sizeof(r));
                r.action = -1; r.route = r.modulation = r.slot = -1; r.flags = ONGYM_F_NO_REQUEST; r.active = active;
                out[(size_t)it * P.batch + replica] = r;
            }
        }
        return;
    }

    float next_rel;
    {
        float m = INFINITY;
        for (int i = lane; i < active; i += kWave) m = fminf(m, rr[i]);
        next_rel = wave_min_f32(m);
    }
    FSTAMP(11);
    for (it = 0; it < nsteps; ++it) {
        if (TRACE && !have) {           // no request source left: the step is a no-op (as in k_run)
            if (REC && lane == 0) {
                ongym_step_rec r;
                memset(&r, 0, sizeof(r));
                r.action = -1; r.route = r.modulation = r.slot = -1; r.flags = ONGYM_F_NO_REQUEST; r.active = active;
                out[(size_t)it * P.batch + replica] = r;
            }
            continue;
        }
        // ================= policy (heuristics/heuristics.py) ==========================================================
        //   FIRST_FIT            :923-966  first (route, best format, lowest start) whose GSNR passes
        //   LOAD_BALANCING       :547-627  the same search on every route whose load (busy slots per hop) is below the best so far
        //   HIGHEST_SNR          :272-328  every start of every (route, format): the highest GSNR among those that pass
        //   LOWEST_FRAGMENTATION :330-414  routes in order of a fragmentation score that is one number per route (quirk, see
        //                                  ongym_scored.hpp); per route the first start (format high to low, start low to high)
        //                                  whose GSNR passes at slots + 1
        // (only ch_k is initialised: the others are read when a candidate was chosen, and eight scalar moves per step are saved)
        int ch_k = -1, ch_m, ch_slot, ch_n, ch_path;
        uint64_t ch_mask;
        double ch_acc, ch_ase, ch_nli;
        double best_acc = INFINITY;                 // HIGHEST_SNR: 1/GSNR of the best candidate so far
        bool lf_qot = false;
        // The routes of the request.  FIRST_FIT walks them in order and stops at the first that serves the request; the other
        // policies look at all of them, so their ids and records are fetched together (lane k = k-th route).
        int v_path = -1, v_rank = lane, nk = K;
        uint32_t v_hops = 1, v_mlo = 0, v_mhi = 0;
        double v_ase = 0.0, v_w1 = 0.0;
        if (POL != ONGYM_POLICY_FIRST_FIT) {
            if (lane < K) {
                v_path = G(P.pair_paths)[pair_base() + lane];
                if (v_path >= 0) {
                    const auto *g = G(reinterpret_cast<const PathRec *>(P.path_rec)) + v_path;
                    v_hops = g->hops; v_mlo = g->mask_lo; v_mhi = g->mask_hi; v_ase = g->ase; v_w1 = g->w1;
                }
            }
            nk = __popcll((unsigned long long)__ballot(v_path >= 0));    // a pair's routes come first, -1 entries after them
            d_paths += nk;
        }
        auto route_rec = [&](int k) -> PathRec {
            PathRec r;
            r.id = rl((uint32_t)v_path, k); r.hops = rl(v_hops, k); r.mask_lo = rl(v_mlo, k); r.mask_hi = M64 ? rl(v_mhi, k) : 0u;
            r.ase = readlane_f64(v_ase, k); r.w1 = readlane_f64(v_w1, k);
            return r;
        };
        auto mask_of = [&](const PathRec &pr) -> uint64_t {
            return M64 ? ((uint64_t)pr.mask_lo | ((uint64_t)(pr.mask_hi & 0x1FFu) << 32)) : (uint64_t)pr.mask_lo;
        };
        // LOAD_BALANCING keeps the route of lowest load among those that serve the request, the first of them on ties (a route
        // only replaces the solution when its load is strictly lower, :571-575, :612-616); LOWEST_FRAGMENTATION the same with the
        // route's score (:404-406).  Whether a route serves the request does not depend on the order the routes are examined
        // in, so they are examined in ascending (load | score, index) order and the first one that serves is the answer.
        if (POL == ONGYM_POLICY_LOAD_BALANCING) {
            // load = np.sum(available_slots == 0) / len(path.links) (:568): compared as exact fractions (numerators <= 1023,
            // denominators <= 64: two distinct fractions differ by far more than an fp64 rounding)
            int v_busy = 0;
            for (int k = 0; k < nk; k++) {
                const PathRec pr = route_rec(k);
                d_hops += pr.hops;
                const int busy = S - (wave_sum_i32(__popc(path_and(mask_of(pr)))) - 1);     // the virtual slot S is the only bit beyond the row
                if (lane == k) v_busy = busy;
            }
            v_rank = 0;
            for (int j = 0; j < nk; j++) {
                const int a = (int)rl((uint32_t)v_busy, j) * (int)v_hops, b = v_busy * (int)rl(v_hops, j);
                v_rank += (a < b || (a == b && j < lane)) ? 1 : 0;
            }
            FSTAMP(13);
        }
        if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION) {
            double v_score = 0.0;
            link_stats_update();
            for (int k = 0; k < nk; k++) {
                const PathRec prk = route_rec(k);
                const double sc = route_score((int)prk.id, (int)prk.hops, mask_of(prk));
                d_hops += (int)prk.hops;
                if (lane == k) v_score = sc;
            }
            v_rank = 0;
            for (int j = 0; j < nk; j++) {
                const double sj = readlane_f64(v_score, j);
                v_rank += (sj < v_score || (sj == v_score && j < lane)) ? 1 : 0;
            }
            FSTAMP(13);
        }

        for (int r = 0; r < nk; r++) {
            int k = r, path;
            PathRec pr;
            if (POL == ONGYM_POLICY_FIRST_FIT) {
                path = k == 0 ? cur_p0 : uniform_i32(KC(P.pair_paths)[pair_base() + k]);
                if (path < 0) break;
                // (fetching the first route's record already when the request is popped was measured 2.4 % SLOWER: eight more
                //  live SGPRs across the departures scan cost more than the scalar-load latency they hide)
                pr = load_path_rec(path_recs, path);
                d_paths++; d_hops += pr.hops;
            } else {
                if (POL != ONGYM_POLICY_HIGHEST_SNR) k = __builtin_ctzll(__ballot(lane < nk && v_rank == r));
                pr = route_rec(k);
                path = (int)pr.id;
                if (POL == ONGYM_POLICY_HIGHEST_SNR) d_hops += pr.hops;
            }
            // modulations whose lower bound at slot 0 already fails cannot pass at any slot
            // lower bound of 1/GSNR at slot 0 = ASE(slot 0) + self-channel NLI (a few fp64 operations per lane: cheaper than
            // keeping their factors in registers for the whole launch)
            const double r_a = w_fac.c1 * pr.ase, r_b = w_fac.cb * pr.ase, r_d = w_fac.c2 * pr.w1;    // this route's ASE at slot 0, ASE per slot, self-channel NLI
            const double lb = r_a + r_d;
            // (lb >= lim*(1+1e-9) => the full sum is at least that large up to rounding, and inside the 1e-9 band the
            //  dB-domain test rejects anything above lim: the skipped evaluation would have failed)
            uint32_t feas = (uint32_t)(__ballot(lb < w_lim_hi) >> (8 * cur_bi)) & 0xFFu;
            // HIGHEST_SNR: nor can a format beat the best candidate so far when its bound is not below that one's 1/GSNR
            if (POL == ONGYM_POLICY_HIGHEST_SNR) feas &= (uint32_t)(__ballot(lb < best_acc) >> (8 * cur_bi));
            if (POL == ONGYM_POLICY_FIRST_FIT) d_feas += __popc(feas);      // modulations settled by the bound (statistics only): see store_env
            else d_skips += M - __popc(feas);
            FSTAMP(0);
            if (!feas) continue;
            const uint64_t pmask = mask_of(pr);
            uint32_t runs = path_and(pmask);
            FSTAMP(1);
            int r_len = 1;
            // (declared per route: nothing of the cache is live across routes or steps)
            // the interferers of ONE route, cached in registers (lane j + 64 e = entry j of the list); `L` = their number
            uint32_t e_c2k[ENT], e_key4[ENT];
            double e_w1[ENT], e_pw2[ENT];
            int e_terms = 0, L = -1, cache_terms = 0;
            double ev_acc = 0.0, ev_ase = 0.0, ev_nli = 0.0;      // results of eval_one for the lane that counts

            // centre, table row, summed link weights (Phi folded in) of running service `idx` as an interferer of the route
            auto intf_of = [&](int idx, uint32_t mask_lo, uint32_t mask_hi, uint32_t &c2k, uint32_t &key4, double &w1o, double &pw2o) -> int {
                const uint2 ab = rec[idx];
                c2k = ab.y & 0x7FFu;
                key4 = rec_nm1<M64>(ab.y) << 15;                // (n-1) * kTabPitch entries * 16 bytes
                uint32_t mm = ab.x & mask_lo;
                double w1 = 0.0, w2 = 0.0;
                int terms = __popc(mm);
                while (mm) { const int l = __ffs(mm) - 1; mm &= mm - 1; w1 += lw[2 * l]; w2 += lw[2 * l + 1]; }
                if (M64) {
                    uint32_t mh = (ab.y >> 23) & mask_hi;
                    terms += __popc(mh);
                    while (mh) { const int l = 32 + __ffs(mh) - 1; mh &= mh - 1; w1 += lw[2 * l]; w2 += lw[2 * l + 1]; }
                }
                w1o = w1;
                pw2o = phi[(ab.y >> 11) & 7u] * w2;
                return terms;
            };
            // pass 1 of the GN model: interferers of the route -> LDS list -> registers (the first 64*ENT of them)
            auto build_cache = [&](uint32_t mask_lo, uint32_t mask_hi) {
                L = 0;
                for (int base = 0; base < active; base += 2 * kWave) {
                    const int i0 = base + lane, i1 = i0 + kWave;
                    const int i1c = min(i1, C);          // beyond the table: the neutral entry
                    bool ov0 = (rec[i0].x & mask_lo) != 0, ov1 = (rec[i1c].x & mask_lo) != 0;    // unused entries: mask 0
                    if (M64) { ov0 |= ((rec[i0].y >> 23) & mask_hi) != 0; ov1 |= ((rec[i1c].y >> 23) & mask_hi) != 0; }
                    const uint64_t bal0 = __ballot(ov0), bal1 = __ballot(ov1);
                    const int n0 = __popcll((unsigned long long)bal0);
                    const int p0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal0, 0));
                    const int p1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal1, 0));
                    if (ov0) list[L + p0] = (uint16_t)i0;
                    if (ov1) list[L + n0 + p1] = (uint16_t)i1;
                    L += n0 + __popcll((unsigned long long)bal1);
                }
                wave_sync();
                FSTAMP(3);
                e_terms = 0;
    #pragma unroll
                for (int e = 0; e < ENT; e++) {
                    const int j = lane + kWave * e;
                    e_c2k[e] = 0; e_key4[e] = 0; e_w1[e] = 0.0; e_pw2[e] = 0.0;
                    if (j < L) e_terms += intf_of(list[j], mask_lo, mask_hi, e_c2k[e], e_key4[e], e_w1[e], e_pw2[e]);
                }
                if (POL == ONGYM_POLICY_HIGHEST_SNR || POL == ONGYM_POLICY_LOWEST_FRAGMENTATION) {
                    int tt = e_terms;       // interferer-link terms of one evaluation on this route (statistics)
                    for (int base = kWave * ENT; base < L; base += kWave)
                        if (base + lane < L) { uint32_t a_, b_; double c_, d_; tt += intf_of(list[base + lane], mask_lo, mask_hi, a_, b_, c_, d_); }
                    cache_terms = wave_sum_i32(tt);
                }
            };
            // pass 2: 1/GSNR of ONE candidate (start `first`, width nn) with lanes over the interferers.  Every lane finishes the
            // evaluation for ITS (bit rate, format) entry of the per-lane tables l_*; lane q is the one that counts.  Returns
            // the decision of qot_ok; ev_acc (and ev_ase / ev_nli when records are written) hold lane q's values when it passes.
            auto eval_one = [&](int first, int nn, int q, const PathRec &pr, double l_a, double l_b, double l_d, double l_nlic,
                                double l_lo, double l_hi) -> int {
                const uint32_t c2 = (uint32_t)(2 * first + nn);
                double part = 0.0;
                {
                    TabPair t[ENT];
    #pragma unroll
                    for (int e = 0; e < ENT; e++) {
                        t[e].x = 0.0; t[e].y = 0.0;
                        if (e == 0 || L > kWave * e) {          // wave-uniform: a cache group without interferers costs nothing
                            const uint32_t adi = __builtin_amdgcn_sad_u16(e_c2k[e], c2, 0);
                            t[e] = load_pair(tab, e_key4[e] | (adi << 4));
                        }
                    }
    #pragma unroll              // (a group without interferers has zero weights and zero table values: no select needed)
                    for (int e = 0; e < ENT; e++) part = fma(t[e].x, e_w1[e], fma(-t[e].y, e_pw2[e], part));
                }
                lane_terms += e_terms;
                for (int base = kWave * ENT; base < L; base += kWave) {       // interferers beyond the register cache
                    const int j = base + lane;
                    if (j < L) {
                        uint32_t c2k, key4;
                        double w1, pw2;
                        lane_terms += intf_of(list[j], pr.mask_lo, pr.mask_hi, c2k, key4, w1, pw2);
                        const uint32_t adi = __builtin_amdgcn_sad_u16(c2k, c2, 0);
                        const TabPair t = load_pair(tab, key4 | (adi << 4));
                        part += t.x * w1 - t.y * pw2;
                    }
                }
                d_evals++;
                const double tot = wave_sum_eval(part);
                const double g_nli = fma(l_nlic, tot, l_d);
                const double g_ase = fma(l_b, (double)first, l_a);                  // envs/qrmsa.pyx:901-905, see lane_fac
                const double acc = g_ase + g_nli;
                int ok;
                {
                    const uint64_t yes = __ballot(acc <= l_lo), no = __ballot(acc >= l_hi);
                    if ((yes >> q) & 1ull) ok = 1;
                    else if ((no >> q) & 1ull) ok = 0;
                    else {      // inside the 1e-9 band: the reference's own dB-domain expression (see qot_ok)
                        const uint64_t db = __ballot(10.0 * log10(1.0 / acc) >= P.mod_thr[lane & 7] + *KC(&ge->margin));
                        ok = (int)((db >> q) & 1ull);
                    }
                }
                if (ok) {
                    ev_acc = readlane_f64(acc, q);
                    if (REC) { ev_ase = readlane_f64(g_ase, q); ev_nli = readlane_f64(g_nli, q); }
                }
                return ok;
            };
            // HIGHEST_SNR / LOWEST_FRAGMENTATION: the set bits of `v` (run-AND words, lane w = word w) as ascending slot indices in
            // dense lanes (xlist); returns their number
            auto compact_starts = [&](uint32_t v) -> int {
                int cnt = 0;
                for (int i = 0; 2 * i < RW; i++) {
                    const uint32_t w0 = rl(v, 2 * i), w1 = rl(v, 2 * i + 1);
                    if (!(w0 | w1)) continue;
                    const int pre = __builtin_amdgcn_mbcnt_hi(w1, __builtin_amdgcn_mbcnt_lo(w0, 0));
                    const uint32_t mine = lane < 32 ? (w0 >> lane) : (w1 >> (lane - 32));
                    if (mine & 1u) xlist[cnt + pre] = (uint16_t)(64 * i + lane);
                    cnt += __popc(w0) + __popc(w1);
                }
                wave_sync();
                return cnt;
            };
            // ... and their GN evaluation with LANES OVER THE CANDIDATES (two chunks of 64 per pass).  The route's interferers are
            // staged in LDS 64 at a time (centre, table row, summed link weights); every lane then reads the same entry (a
            // broadcast read, no v_readlane and no scalar registers) and each (interferer, chunk) costs |x - c_k|, one address
            // op, one 16-byte gather and two FMAs, four interferers (eight gathers) in flight.  No validity test: a valid start
            // never overlaps a running service on a shared link, so |x - c_k| > n_k; dead lanes read table entries that exist.
            // Partial sums are lower bounds of 1/GSNR (no interferer term is negative, Params.ase_shortcut): a pass stops as soon
            // as no candidate can stay below the acceptance limit — or, for HIGHEST_SNR, below the best candidate so far.
            // HIGHEST_SNR keeps the best passing candidate (strictly smaller 1/GSNR: the first maximum wins, heuristics.py:316);
            // LOWEST_FRAGMENTATION stops at the first that passes.
            auto eval_cands = [&](int cnt, int nn, int m, int q, const PathRec &pr, int k, int path, uint64_t pmask, double l_nlic,
                                  double l_selfa, double l_lo, double l_hi) -> bool {
#ifndef ONGYM_CAND_NA
#define ONGYM_CAND_NA 2
#endif
#ifndef ONGYM_CAND_TU
#define ONGYM_CAND_TU 4
#endif
                constexpr int NA = ONGYM_CAND_NA, TU = ONGYM_CAND_TU;
                const double c_bw = P.slot_bw * nn, c_h = P.slot_bw * (nn / 2.0);
                const double c_nlic = readlane_f64(l_nlic, q), c_self = pr.w1 * readlane_f64(l_selfa, q);
                const double c_lo = readlane_f64(l_lo, q), c_hi = readlane_f64(l_hi, q);
                for (int j0 = 0; j0 < cnt; j0 += NA * kWave) {
                    const bool two = NA > 1 && j0 + kWave < cnt;          // the second chunk holds candidates (wave-uniform)
                    uint32_t ss[NA], xs[NA];
                    double f[NA], gase[NA];
                    bool live[NA];
    #pragma unroll
                    for (int a = 0; a < NA; a++) {
                        const int j = j0 + a * kWave + lane;
                        live[a] = j < cnt;
                        ss[a] = live[a] ? (uint32_t)xlist[j] : 0u;
                        xs[a] = 2u * ss[a] + (uint32_t)nn;
                        f[a] = 0.0;
                        const double fc = P.f0 + (P.slot_bw * (int)ss[a]) + c_h;           // envs/qrmsa.pyx:901-905
                        gase[a] = (c_bw * fc * pr.ase) * rp0;
                    }
                    const double thr = POL == ONGYM_POLICY_HIGHEST_SNR ? fmin(c_hi, best_acc) : c_hi;
                    const int x_lo = 2 * (int)xlist[j0] + nn, x_hi = 2 * (int)xlist[min(j0 + NA * kWave, cnt) - 1] + nn;     // span of the pass's centres
                    bool cut = false;
                    for (int base = 0; base < L && !cut; base += kWave) {
                        {   // stage: lane j = interferer base + j (zero weights beyond the list)
                            uint32_t c2k = 0, key4 = 0;
                            double w1 = 0.0, pw2 = 0.0;
                            if (base + lane < L) intf_of((int)list[base + lane], pr.mask_lo, pr.mask_hi, c2k, key4, w1, pw2);
                            key4 |= ((c2k + (uint32_t)nn) & 1u) << 14;       // the parity half of the row: every candidate of the pass has x = 2s + nn
                            // interferers within kNearHalfSlots half slots of the pass's candidates first (stable partition): their terms
                            // are the large ones, so the partial sums reach the threshold sooner (+7 % with the exit test every four
                            // interferers; the window size hardly matters: 8..96 half slots give +6.4..+7.7 %)
                            const bool real_ = base + lane < L;
                            const bool near_ = real_ && (int)c2k >= x_lo - kNearHalfSlots && (int)c2k <= x_hi + kNearHalfSlots;
                            const uint64_t bn = __ballot(near_);
                            const int pn = __builtin_amdgcn_mbcnt_hi((uint32_t)(bn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bn, 0));
                            const int pos = near_ ? pn : __popcll((unsigned long long)bn) + (lane - pn);
                            st_ck[pos] = make_uint2(c2k, key4);
                            st_w[pos] = make_double2(w1, pw2);
                        }
                        wave_sync();
                        const int ne = min(kWave, L - base);
                        for (int t = 0; t < ne; t += TU) {
                            uint2 ck[TU];
                            double2 w[TU];
    #pragma unroll
                            for (int u = 0; u < TU; u++) { ck[u] = st_ck[t + u]; w[u] = st_w[t + u]; }
                            TabPair tp[TU][NA];
    #pragma unroll
                            for (int u = 0; u < TU; u++)
    #pragma unroll
                                for (int a = 0; a < NA; a++)
                                    if (a == 0 || two)      // entry (d & 1, d >> 1) of the row, d = |x - c_k|
                                        tp[u][a] = load_pair(tabp, ck[u].y | ((__builtin_amdgcn_sad_u16(xs[a], ck[u].x, 0) << 3) & 0xFFF0u));
    #pragma unroll
                            for (int u = 0; u < TU; u++)
    #pragma unroll
                                for (int a = 0; a < NA; a++)
                                    if (a == 0 || two) f[a] = fma(tp[u][a].x, w[u].x, fma(-tp[u][a].y, w[u].y, f[a]));
    #ifndef ONGYM_X_NO_EARLY_EXIT
#ifndef ONGYM_EXIT_EVERY
#define ONGYM_EXIT_EVERY 4
#endif
                            if (((t + TU) & (ONGYM_EXIT_EVERY - 1)) == 0 && base + t + TU < L) {      // every ONGYM_EXIT_EVERY interferers: can any candidate still matter?
                                bool alive = false;
#pragma unroll
                                for (int a = 0; a < NA; a++)
                                    if (a == 0 || two) alive |= live[a] && gase[a] + c_nlic * (f[a] + c_self) < thr;
                                if (!__ballot(alive)) { cut = true; break; }
                            }
#endif
                        }
                        wave_sync();                            // the stage is rewritten by the next block
                    }
    #pragma unroll
                    for (int a = 0; a < NA; a++) {
                        if (a > 0 && !two) break;
                        d_evals += __popcll((unsigned long long)__ballot(live[a]));
                        lane_terms += live[a] ? cache_terms : 0;
                        if (cut) continue;
                        const double g_nli = c_nlic * (f[a] + c_self);
                        const double acc = gase[a] + g_nli;
                        bool ok = live[a] && acc <= c_lo;
                        if (live[a] && !ok && acc < c_hi) ok = 10.0 * log10(1.0 / acc) >= P.mod_thr[m] + margin;   // the 1e-9 band (qot_ok)
                        const uint64_t okb = __ballot(ok);
                        if (!okb) continue;
                        if (POL == ONGYM_POLICY_HIGHEST_SNR) {
                            const double v = ok ? acc : INFINITY;
                            const double vmin = wave_min_f64(v);
                            // a new best must beat the old one by more than the rounding noise between this sum and the
                            // reference's (different association: ~1e-13): candidates whose 1/GSNR agree to 1e-10 are exact ties
                            // in the reference (equal routes on an empty network) and the first one keeps the lead there
                            if (vmin < best_acc * (1.0 - 1e-10)) {
                                const int ln = __builtin_ctzll(__ballot(v == vmin));
                                best_acc = vmin; ch_acc = vmin;
                                ch_k = k; ch_m = m; ch_n = nn; ch_path = path; ch_mask = pmask; ch_slot = (int)rl(ss[a], ln);
                                if (REC) { ch_ase = readlane_f64(gase[a], ln); ch_nli = readlane_f64(g_nli, ln); }
                            }
                        } else {
                            const int ln = __builtin_ctzll(okb);
                            ch_k = k; ch_m = m; ch_n = nn - 1; ch_path = path; ch_mask = pmask; ch_slot = (int)rl(ss[a], ln);
                            return true;
                        }
                    }
                }
                return false;
            };

            while (feas) {
                const int m = 31 - __builtin_clz(feas);            // best modulation first
                asm("s_bitset0_b32 %0, %1" : "+s"(feas) : "s"(m));
                const int q = 8 * cur_bi + m;
                const int n = (int)rl((uint32_t)t_n, q);
                const int nn = POL == ONGYM_POLICY_LOWEST_FRAGMENTATION ? n + 1 : n;      // quirk: the request is sized slots + 1 (:357)
                if (nn + 1 < r_len) { runs = path_and(pmask); r_len = 1; }     // slot counts normally grow as the modulation index falls
                runs = run_and32<WIDE>(runs, r_len, nn + 1);
                const uint64_t has = __ballot(runs != 0);
                FSTAMP(2);
                if (!has) continue;
                const int first_w = __builtin_ctzll(has);
                const int first = first_w * 32 + __builtin_ctz(rl(runs, first_w));       // == first_set32(runs)
                if (POL == ONGYM_POLICY_HIGHEST_SNR && !((__ballot(lb < best_acc) >> q) & 1ull)) continue;   // best_acc may have improved
                if (L < 0) build_cache(pr.mask_lo, pr.mask_hi);
                FSTAMP(4);
                if (POL == ONGYM_POLICY_HIGHEST_SNR) {
                    const int cnt = compact_starts(runs);
                    FSTAMP(14);
                    eval_cands(cnt, nn, m, q, pr, k, path, pmask, w_nlic, w_selfa, w_lim_lo, w_lim_hi);
                    FSTAMP(15);
                    continue;
                }
                const int ok = eval_one(first, nn, q, pr, r_a, r_b, r_d, w_nlic, w_lim_lo, w_lim_hi);
                FSTAMP(5);
                if (ok) {
                    ch_k = k; ch_m = m; ch_slot = first; ch_n = n; ch_path = path;
                    ch_mask = pmask;
                    ch_acc = ev_acc; ch_ase = ev_ase; ch_nli = ev_nli;
                    break;
                }
                if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION) {
                    // the other starts of this format, 128 at a time, until one passes
                    const uint32_t rest = lane == (first >> 5) ? runs & ~(1u << (first & 31)) : runs;
                    const int cnt = compact_starts(rest);
                    FSTAMP(14);
                    const bool hit = cnt > 0 && eval_cands(cnt, nn, m, q, pr, k, path, pmask, w_nlic, w_selfa, w_lim_lo, w_lim_hi);
                    FSTAMP(15);
                    if (hit) break;
                }
            }
            if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION && ch_k >= 0) {
                // `env.step(action)` evaluates the GN model itself at the width the format needs (the heuristic asked for one slot
                // more): the reference raises ValueError if that fails (envs/qrmsa.pyx:925-929) — a fused episode rejects the
                // request and flags it (as k_run does).  Same route: the interferer cache is the one just used.
                const LaneFac sf = lane_fac(t_n, t_nlic, t_selfa);
                if (eval_one(ch_slot, ch_n, 8 * cur_bi + ch_m, pr, sf.c1 * pr.ase, sf.cb * pr.ase, sf.c2 * pr.w1, t_nlic, t_lim_lo, t_lim_hi)) {
                    ch_acc = ev_acc; ch_ase = ev_ase; ch_nli = ev_nli;
                } else { ch_k = -1; lf_qot = true; }
                break;
            }
            if (POL != ONGYM_POLICY_HIGHEST_SNR && ch_k >= 0) break;
        }

        // ================= step (envs/qrmsa.pyx:838-1065) ==============================================================
        int accepted = ch_k >= 0;
        int rflags = 0;
        if (accepted && active >= C) { accepted = 0; rflags |= ONGYM_F_OVERFLOW; d_flags |= ONGYM_F_OVERFLOW; }
        if (accepted) {
            // _provision_path (:1288-1325): n slots + one guard slot unless the allocation ends at S
            int end = ch_slot + ch_n; if (end < S) end += 1;
#ifdef ONGYM_X_ACCEPT_MARKV
            if (end - ch_slot <= 33) mark_v((uint32_t)ch_mask, M64 ? (uint32_t)(ch_mask >> 32) : 0u, (uint32_t)ch_slot + vz, (uint32_t)(end - ch_slot) + vz, false);
            else
#endif
            mark(ch_mask, ch_slot, end, false);
            if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION) ls_dirty |= ((ch_mask >> lane) & 1ull) != 0;      // these rows changed
            const float rel = v_at + cur_ht;                  // float + float (:1329); compared as float32 (:1114-1115)
            next_rel = fminf(next_rel, rel);
            const uint32_t ra = (uint32_t)ch_mask;
            // record word b (fast_pack_b / fast_pack_b64): the format's part (n | modulation << 11 | (n - 1) << 14) comes from the
            // per-lane table, the rest is the start and the path id (or the mask's high bits); ch_n == t_n of the chosen lane
            const uint32_t rb = rl(t_pk, 8 * cur_bi + ch_m) + 2u * (uint32_t)ch_slot +
                                ((M64 ? (uint32_t)(ch_mask >> 32) : (uint32_t)(ch_path & 0x1FF)) << 23);
            const uint32_t v_act = vz + (uint32_t)active;        // (address arithmetic on the vector pipe, see vz)
            lds_write_lane0(rec_base + v_act * 8u, ra, rb, rr_base + v_act * 4u, __float_as_uint(rel));
            active++;
            if (TRACE) d_acc++;
            cnt += (lane == 8 + ch_m || lane == 24 + cur_bi) ? 1 : 0;
            osnr_prod *= ch_acc;
            if (osnr_prod < 1e-250) { if (lane == 0) cold[4] += -10.0 * log10(osnr_prod); osnr_prod = 1.0; wave_sync(); }
        } else {
            erej++; d_rej++;
            if (ch_k < 0 && POL == ONGYM_POLICY_LOWEST_FRAGMENTATION && lf_qot) {
                rflags |= ONGYM_F_QOT_ERROR | ONGYM_F_BLOCKED_OSNR;
            } else if (ch_k < 0 && POL != ONGYM_POLICY_FIRST_FIT) {
                // blocking flags of the heuristic's return tuple, exactly: nothing passed, so every (route, format) pair with a
                // candidate start set any_blocked_osnr (its candidates failed) and every pair without one any_blocked_resources,
                // which the return statement clears when any_blocked_osnr is set (heuristics.py:322-328, 410-414, 620-627).
                // Candidates shrink as the slot count grows: a pair with candidates exists iff some route has a start for the
                // smallest slot count.
                int n_small = 0x7fffffff;
                for (int m = M - 1; m >= 0; m--) {
                    const int n = (int)rl((uint32_t)t_n, 8 * cur_bi + m);
                    if (n > 0) n_small = min(n_small, n);
                }
                if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION) n_small += 1;
                int bosnr = 0;
                for (int k = 0; k < K && !bosnr && n_small <= S; k++) {
                    const int path = uniform_i32(KC(P.pair_paths)[pair_base() + k]);
                    if (path < 0) break;
                    const PathRec pr = load_path_rec(path_recs, path);
                    const uint32_t x = path_and(M64 ? ((uint64_t)pr.mask_lo | ((uint64_t)(pr.mask_hi & 0x1FFu) << 32)) : (uint64_t)pr.mask_lo);
                    int r1 = 1;
                    if (first_set32(run_and32<WIDE>(x, r1, n_small + 1)) >= 0) bosnr = 1;
                }
                rflags |= bosnr ? ONGYM_F_BLOCKED_OSNR : ONGYM_F_BLOCKED_RESOURCES;
            } else if (ch_k < 0) {
                // blocking flags of the heuristic's return tuple, exactly: a (path, modulation) pair without candidates sets
                // blocked_resources, one with candidates clears it and sets blocked_osnr (none passed, or we would not be
                // here).  Candidates shrink as the slot count grows, so only the smallest and, on the last path, the
                // largest slot count matter.
                int bres = 0, bosnr = 0;
                int n_small = 0x7fffffff, n_big = 0, n_last = 0;
                for (int m = M - 1; m >= 0; m--) {
                    const int n = (int)rl((uint32_t)t_n, 8 * cur_bi + m);
                    if (n <= 0) continue;
                    n_small = min(n_small, n); n_big = max(n_big, n); n_last = n;
                }
                for (int k = 0; k < K && n_big > 0; k++) {
                    const int path = uniform_i32(KC(P.pair_paths)[pair_base() + k]);
                    if (path < 0) break;
                    const PathRec pr = load_path_rec(path_recs, path);
                    const uint32_t x = path_and(M64 ? ((uint64_t)pr.mask_lo | ((uint64_t)(pr.mask_hi & 0x1FFu) << 32)) : (uint64_t)pr.mask_lo);
                    int r1 = 1;
                    const bool any = first_set32(run_and32<WIDE>(x, r1, n_small + 1)) >= 0;
                    if (any) { bosnr = 1; bres = 0; }
                    // the LAST pair visited decides blocked_resources: modulation 0's slot count on this path
                    int r2 = 1;
                    const bool last_has = first_set32(run_and32<WIDE>(x, r2, n_last + 1)) >= 0;
                    bres = last_has ? 0 : 1;
                    (void)n_big;
                }
                rflags |= (bres ? ONGYM_F_BLOCKED_RESOURCES : 0) | (bosnr ? ONGYM_F_BLOCKED_OSNR : 0);
            }
        }
        if (TRACE) d_steps++;
        FSTAMP(6);

        ongym_step_rec *const recp = REC ? out + (size_t)it * P.batch + replica : nullptr;
        if (REC) {
            ongym_step_rec r;
            r.action = K * M * S; r.route = -1; r.modulation = -1; r.slot = -1; r.nslots = 0;
            r.accepted = 0; r.terminated = 0; r.retry = 0; r.flags = (uint8_t)rflags;
            r.osnr = 0.0; r.ase = 0.0; r.nli = 0.0; r.reward = -6.0; r.active = 0;
            if (accepted) {
                r.action = ch_k * M * S + (M - 1 - ch_m) * S + ch_slot;      // get_action_index, heuristics.py:36-54
                r.route = (int16_t)ch_k; r.modulation = (int16_t)ch_m; r.slot = (int16_t)ch_slot; r.nslots = (int16_t)ch_n;
                r.accepted = 1; r.reward = 0.0;                                  // quirk Q1
                // the three dB values of the record in ONE log10 evaluation: lanes 0, 1, 2 take GSNR, ASE, NLI (a log10 costs ~50
                // vector instructions whatever the lanes hold)
                const double v3 = lane == 1 ? ch_ase : lane == 2 ? ch_nli : ch_ase + ch_nli;
                const double l3 = -10.0 * log10(v3);
                r.osnr = readlane_f64(l3, 0); r.ase = readlane_f64(l3, 1); r.nli = readlane_f64(l3, 2);
            } else if (ch_k >= 0) {     // overflow: the policy had chosen, the table is full
                r.action = ch_k * M * S + (M - 1 - ch_m) * S + ch_slot;
            }
            if (lane == 0) *recp = r;
        }

        // the info dict of the terminal step is computed before the next request is drawn (:996-1060)
        const bool term = (epp + 1 == ep_len);
        if (term) {
            if (lane == 0) cold[4] += (osnr_prod != 1.0) ? -10.0 * log10(osnr_prod) : 0.0;      // flush_osnr
            osnr_prod = 1.0;
            wave_sync();
            store_env(it + 1);
            __threadfence_block();
            if (lane == 0) {
                const ongym_stats &s = ge->st;
                ongym_stats &so = ge->st;
                so.last_episode_processed = s.episode_services_processed; so.last_episode_accepted = s.episode_services_accepted;
                so.last_rejected = s.rejected;
                so.last_service_blocking_rate = s.services_processed > 0
                    ? (double)(s.services_processed - s.services_accepted) / (double)s.services_processed : 0.0;
                so.last_episode_service_blocking_rate = s.episode_services_processed > 0
                    ? (double)(s.episode_services_processed - s.episode_services_accepted) / (double)s.episode_services_processed : 0.0;
                so.last_bit_rate_blocking_rate = s.bit_rate_requested > 0
                    ? (s.bit_rate_requested - s.bit_rate_provisioned) / s.bit_rate_requested : 0.0;
                so.last_episode_bit_rate_blocking_rate = s.episode_bit_rate_requested > 0
                    ? (s.episode_bit_rate_requested - s.episode_bit_rate_provisioned) / s.episode_bit_rate_requested : 0.0;
                for (int m = 0; m < 8; m++) so.last_modulation_hist[m] = s.episode_modulation_hist[m];
                so.last_mean_gsnr = s.episode_services_processed + ge->svc_list_extra > 0
                    ? cold[4] / (double)(s.episode_services_processed + ge->svc_list_extra) : 0.0;
                so.last_episode_disrupted = s.episode_disrupted_services;
                so.last_episode_defrag_cycles = s.episode_defrag_cycles;
                so.last_episode_service_reallocations = s.episode_service_reallocations;
            }
        }

        // ================= _next_service (:1067-1122): next request, then the departures it triggers ==================
        FSTAMP(10);
        pop_request();
        FSTAMP(7);
        // Departures: release times of 4 chunks of 64 records are read together (one LDS round trip), highest chunk first.
        // Processing a chunk only rewrites entries at or above the hole being filled, so the comparisons made ahead of time
        // for the lower chunks stay valid.
        auto depart = [&](int ch, uint64_t bal) {
            while (bal) {
                const int ln = 63 - __builtin_clzll(bal);                   // highest index first: the hole is filled by a keeper
                asm("s_bitset0_b64 %0, %1" : "+s"(bal) : "s"(ln));
                const int idx = ch * kWave + ln;
                const uint2 ab = rec[idx + vz];                             // same address in every lane: broadcast read
                const uint32_t nk = rec_nm1<M64>(ab.y) + 1u, sk = ((ab.y & 0x7FFu) - nk) >> 1;
                const uint32_t hi = min(sk + nk + 1u, (uint32_t)S);         // frees n+1 slots, clamped at S (quirk Q7)
                if (!WIDE || __builtin_amdgcn_readfirstlane(nk) <= 32u)
                    mark_v(ab.x, M64 ? (ab.y >> 23) : 0u, sk, hi - sk, true);
                else {
                    // (unsigned: readfirstlane returns int, and link 31 / link 40 are the sign bits of the two words)
                    uint64_t mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)ab.x);
                    if (M64) mask |= (uint64_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)ab.y) >> 23) << 32;
                    mark(mask, (int)__builtin_amdgcn_readfirstlane(sk), (int)__builtin_amdgcn_readfirstlane(hi), true);
                }
                if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION)          // the departed service's links: their rows changed
                    ls_dirty |= lane < 32 ? ((ab.x >> lane) & 1u) != 0 : (M64 && (((ab.y >> 23) >> (lane - 32)) & 1u) != 0);
                const int last = active - 1;
                // lane 0 moves the last record into the hole, lane 1 neutralises the vacated entry (both write the neutral entry when the
                // hole IS the last record): one masked pair of writes, addresses and data formed on the vector pipe
                const uint32_t v_last = vz + (uint32_t)last;
                const uint2 lab = rec[v_last];
                const float lr = rr[v_last];
                const bool mv = lane == 0 && idx != last;
                const uint32_t e = lane == 0 ? (uint32_t)idx + vz : v_last;
                lds_write_lanes01(rec_base + e * 8u, mv ? lab.x : 0u, mv ? lab.y : 0u, rr_base + e * 4u, mv ? __float_as_uint(lr) : 0x7F800000u);
                active = last;
                wave_sync();
                FSTAMP(9);
            }
        };
        // the earliest release time among the running services is kept between steps: no service leaves before it, and about a
        // third of the requests arrive before it (departures per inter-arrival time are Poisson with mean ~1): no scan then
        if (v_at >= next_rel) {
            float keep_min = INFINITY;
        for (int top = ((active + kWave - 1) / kWave) - 1; top >= 0; top -= 4) {
            const float r0 = rr[top * kWave + lane], r1 = rr[max(top - 1, 0) * kWave + lane];
            const float r2 = rr[max(top - 2, 0) * kWave + lane], r3 = rr[max(top - 3, 0) * kWave + lane];   // unused entries hold +inf
            const uint64_t b0 = __ballot(r0 <= v_at), b1 = top >= 1 ? __ballot(r1 <= v_at) : 0ull;
            const uint64_t b2 = top >= 2 ? __ballot(r2 <= v_at) : 0ull, b3 = top >= 3 ? __ballot(r3 <= v_at) : 0ull;
            // release times that stay (the moves below do not change the multiset of the remaining ones)
            keep_min = fminf(fminf(keep_min, r0 <= v_at ? INFINITY : r0), fminf(r1 <= v_at ? INFINITY : r1, fminf(r2 <= v_at ? INFINITY : r2, r3 <= v_at ? INFINITY : r3)));
            FSTAMP(8);
            if (!(b0 | b1 | b2 | b3)) continue;
            depart(top, b0);
            if (b1) depart(top - 1, b1);
            if (b2) depart(top - 2, b2);
            if (b3) depart(top - 3, b3);
        }
            next_rel = wave_min_f32(keep_min);
        }
        d_active_sum += (uint32_t)active;
        const bool terminated = epp == ep_len;
        if (terminated) d_episodes++;
        if (REC && lane == 0) { recp->active = active; recp->terminated = (uint8_t)terminated; }
        if (terminated && P.auto_reset) {
            // reset (:427-504): empty network, episode counters to zero, draw request #0 of the next episode
            for (int i = lane; i < E * RW; i += kWave) occ[i] = word_mask32(i % RW, 0, S);
            for (int i = lane; i < active; i += kWave) { rec[i] = make_uint2(0u, 0u); rr[i] = INFINITY; }
            active = 0;
            next_rel = INFINITY;
            epp = 0; erej = 0; cnt = 0;
            if (POL == ONGYM_POLICY_LOWEST_FRAGMENTATION) { ls_ent = 0.0; ls_cuts = ls_sl = ls_sq = 0; ls_dirty = false; }   // empty rows
            if (lane < 5) cold[lane] = 0.0;
            if (lane == 0) ge->svc_list_extra = 0;
            osnr_prod = 1.0;
            wave_sync();
            pop_request();
        }
    }

    // ---- store: DevEnv, bitmap, records (lean codec -> generic codec) ----
    store_env(nsteps);
    wave_sync();
    {
        uint32_t *g = reinterpret_cast<uint32_t *>(P.occ + (size_t)replica * E * P.row_words);
        for (int i = lane; i < E * RW; i += kWave) g[i] = occ[i];
        const size_t off = (size_t)replica * C;
        for (int i = lane; i < active; i += kWave) {
            const uint2 ab = rec[i];
            const int n = (int)rec_nm1<M64>(ab.y) + 1, slot = ((int)(ab.y & 0x7FFu) - n) >> 1, mod = (int)((ab.y >> 11) & 7u);
            int path = (int)(ab.y >> 23);
            if (M64) {      // link set -> path id: open addressing; every route's link set is in the table (built at create), and the
                            // probe is bounded by the table size all the same
                const uint64_t key = (uint64_t)ab.x | ((uint64_t)(ab.y >> 23) << 32);
                const uint32_t hm = (1u << P.path_hash_bits) - 1u;
                uint32_t h = path_hash_slot(key, P.path_hash_bits);
                path = 0;
                for (uint32_t tries = 0; tries <= hm; tries++, h = (h + 1u) & hm)
                    if (G(P.path_hash_keys)[h] == key) { path = G(P.path_hash_vals)[h]; break; }
            }
            uint32_t ga, gb;
            if (P.rec32) rec_pack<true>(path, ab.x, slot, n, mod, ga, gb);
            else rec_pack<false>(path, 0, slot, n, mod, ga, gb);
            P.svc_a[off + i] = ga; P.svc_b[off + i] = gb; P.svc_r[off + i] = rr[i];
        }
    }
    FSTAMP(12);
#ifdef ONGYM_STAMPS
    if (lane == 0 && P.dbg)
        for (int i = 0; i < ONGYM_NSTAMPS; i++) atomicAdd(&P.dbg[i], stamp_acc[i]);
#endif
#undef FSTAMP
}

}  // namespace ongym
