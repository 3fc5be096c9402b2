// The two score-everything policies of heuristics/heuristics.py, fused on device (policy ids 10 and 11):
//
//   heuristic_lowest_fragmentation  (:330-414)   score of a route = 0.33*mean link entropy + 0.33*cuts + 0.34*rss
//   heuristic_mscl                  (:647-749)   capacity loss of a (route, format, start) over every route of the network
//
// Both evaluate the GN model for EVERY candidate start of every format of every route; neither is on the benchmark path.
// They share k_run's state block and helpers (ongym_device.hpp) and are one extra instantiation (kPolicyScored).
//
// Why the floating-point score of lowest fragmentation is reproducible bit for bit: the reference computes, per link row,
//   entropy = 0.0; for every run of value 0, left to right: p = len/S; entropy += p*math.log(p)       (utils.pyx:61-79)
// p*log(p) takes S distinct values: the host tabulates them with the C library's log (what CPython's math.log calls) in
// Params.plogp, and the device adds the table entries in the same order. Python's sum() (3.10: plain left-to-right
// addition) over the links, the two exact integer sums of rss (:92-107), one correctly rounded sqrt and division, and the
// final 0.33*a + 0.33*b + 0.34*c without contraction complete it.
#pragma once
#include "ongym_device.hpp"

namespace ongym {

constexpr int kPolicyScored = ONGYM_POLICY_LOWEST_FRAGMENTATION;   // template value of the shared instantiation

// extra dynamic LDS of the kPolicyScored instantiation: H, LOSS int32[64*W] and PH int32[64*W + 1] (capacity-loss search)
__host__ __device__ inline size_t scored_lds_bytes(int row_words) { return ((size_t)(3 * 64 * row_words + 1) * 4 + 15) & ~(size_t)15; }

// Quirk kept (documented in heuristics.py of this package): the trial allocation paints 1 over slots that are 1 already, so
// the score depends on the route only; the "free blocks" of utils.pyx:61-107 are the runs of value 0 = OCCUPIED slots.
// Lane h walks the occupied runs of the route's h-th link.
__device__ __forceinline__ double lf_route_score(const Ctx &c, const PathRef &p) {
#pragma clang fp contract(off)
    const Params &P = c.P;
    const int S = P.n_slots, RW = P.row_words;
    double ent = 0.0, sq = 0.0, sl = 0.0;
    int cuts = 0;
    if (c.lane < p.hops) {
        const uint64_t *row = c.occ + (size_t)p.mylink * RW;
        int carry = 0;                                   // length of the run that is open at the current position
        auto close = [&](int len) {
            ent += G(P.plogp)[len];                      // entropy += p * math.log(p)
            cuts++;
            sq += (double)len * (double)len;
            sl += (double)len;
        };
        for (int w = 0; w < RW; w++) {
            const int nb = min(64, S - 64 * w);
            uint64_t z = ~row[w];
            if (nb < 64) z &= (1ull << nb) - 1ull;
            int pos = 0;
            while (pos < nb) {
                const uint64_t rest = z >> pos;
                if (carry > 0 || (rest & 1ull)) {
                    const uint64_t inv = ~rest;
                    int len = inv ? __builtin_ctzll(inv) : 64;
                    len = min(len, nb - pos);
                    carry += len; pos += len;
                    if (pos < nb) { if (carry > 0) close(carry); carry = 0; }
                } else {
                    pos += rest ? __builtin_ctzll(rest) : 64;
                }
            }
        }
        if (carry > 0) close(carry);
        ent = ent != 0.0 ? -ent : 0.0;                   // utils.pyx:79
    }
    double se = 0.0;                                     // sum(entropies): left to right, starting from int 0
    for (int h = 0; h < p.hops; h++) se = __dadd_rn(se, readlane_f64(ent, h));
    se = se / (double)p.hops;
    int tc = cuts;
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) tc += __shfl_xor(tc, mm);
    const double tsq = wave_sum(sq), tsl = wave_sum(sl);   // integers < 2^53: exact in any order
    const double rss = tsl == 0.0 ? 0.0 : sqrt(tsq) / tsl;
    const double score = __dadd_rn(__dadd_rn(__dmul_rn(0.33, se), __dmul_rn(0.33, (double)uniform_i32(tc))), __dmul_rn(0.34, rss));
    return uniform_f64(score);
}

__device__ __forceinline__ int wave_incl_scan_i32(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d);
        if (lane >= d) v += u;
    }
    return v;
}
__device__ __forceinline__ long long wave_min_i64(long long v) {
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) {
        const long long u = __shfl_xor(v, mm);
        v = u < v ? u : v;
    }
    return v;
}

// scr: H int32[64*W] | LOSS int32[64*W] | PH int32[64*W + 1]
template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ void policy_scored(Ctx &c, int policy, int src, int dst, double margin, Choice &ch, int32_t *scr) {
    const Params &P = c.P;
    const int M = P.n_mods, S = P.n_slots, max_mod = M - 1, K = P.k_paths, RW = P.row_words;
    ch.action = K * M * S; ch.route = -1; ch.mod = -1; ch.slot = -1; ch.n = 0; ch.hops = 0; ch.mylink = 0;
    ch.path = -1; ch.m0 = 0; ch.g.ase = ch.g.nli = 0.0; ch.flags = 0;
    int bres = 0, bosnr = 0;
    double best_score = INFINITY;
    long long best_loss = 0x7fffffffffffffffll;
    int32_t *H = scr, *LOSS = scr + 64 * RW, *PH = scr + 128 * RW;
    auto take = [&](int k, int m, int slot, int n, const PathRef &p) {
        ch.action = k * M * S + (max_mod - m) * S + slot;          // get_action_index, heuristics.py:36-54
        ch.route = k; ch.mod = m; ch.slot = slot; ch.n = n; ch.hops = p.hops; ch.mylink = p.mylink;
        ch.path = p.id; ch.m0 = p.m0;
    };
    auto row_of = [&](uint64_t free_ext) {   // the path row without the virtual free slot S
        return (c.lane == (S >> 6)) ? (free_ext & ~(1ull << (S & 63))) : free_ext;
    };
    for (int k = 0; k < K; k++) {
        const int path = uniform_i32(G(P.pair_paths)[(src * P.n_nodes + dst) * K + k]);
        if (path < 0) break;
        const PathRef p = load_path(c, path);
        const uint64_t free_ext = path_free_ext(c, p);
        int L = -1;
        if (policy == ONGYM_POLICY_LOWEST_FRAGMENTATION) {
            // every candidate of the route has the route's score: the first one (format high to low, start low to high)
            // whose GSNR passes is the route's, and it wins iff the score is strictly below the best so far; what the
            // reference evaluates after that changes neither the choice nor the flags (they are dropped once a choice exists)
            const double score = lf_route_score(c, p);
            if (ch.route >= 0 && !(score < best_score)) continue;
            bool taken = false;
            for (int m = max_mod; m >= 0 && !taken; m--) {
                const int n1 = uniform_i32(c.nreq[m]) + 1;         // quirk: the request is sized slots + 1 (:357)
                if (n1 <= 0) continue;
                int rr = 1;
                uint64_t v = run_and(free_ext, rr, n1 + 1);        // _get_candidates(available, n1, S)
                if (!any_bits(v)) { bres = 1; continue; }
                for (;;) {
                    const int s0 = first_set(v);
                    if (s0 < 0) break;
                    if (L < 0) L = gn_build_list<R32>(c, p.m0, p.m1);
                    const GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, s0, n1);   // ... and evaluated at that width (:388-399)
                    if (qot_ok(c, g, m, margin)) { take(k, m, s0, n1 - 1, p); best_score = score; taken = true; break; }
                    bosnr = 1;
                    if (c.lane == (s0 >> 6)) v &= ~(1ull << (s0 & 63));
                }
            }
            continue;
        }
        // ---- heuristic_mscl ------------------------------------------------------------------------------------------
        for (int m = max_mod; m >= 0; m--) {
            const int n = uniform_i32(c.nreq[m]);
            if (n <= 0) continue;
            int rr = 1;
            uint64_t v = run_and(free_ext, rr, n + 1);
            if (!any_bits(v)) { bres = 1; continue; }
            uint64_t okb = 0;                                      // starts whose GSNR passes (lane-distributed like v)
            for (;;) {
                const int s0 = first_set(v);
                if (s0 < 0) break;
                if (c.lane == (s0 >> 6)) v &= ~(1ull << (s0 & 63));
                if (L < 0) L = gn_build_list<R32>(c, p.m0, p.m1);
                const GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, s0, n, coef_for_mod(c, m));
                if (qot_ok(c, g, m, margin)) { if (c.lane == (s0 >> 6)) okb |= 1ull << (s0 & 63); }
                else bosnr = 1;
            }
            if (!any_bits(okb)) continue;
            // capacity loss of taking [start, start+n) (no guard, :719-720): per configured bit rate (width w at THIS format)
            // and per route q of the network sharing a link with the candidate route, the free windows of width w of q's
            // row that overlap the block, i.e. the window starts t in (start - w, start + n). H[t] = number of routes with
            // a free window at t; PH its prefix sum; LOSS[start] += PH[min(start+n, S)] - PH[max(0, start-w+1)].
            // (The reference's route list holds every route twice, once per direction: a uniform factor 2 on every loss.)
            for (int j = 0; j < RW; j++) LOSS[64 * j + c.lane] = 0;
            for (int b = 0; b < P.n_bit_rates; b++) {
                const int w = uniform_i32(G(P.nreq_tab)[b * kMaxMods + m]);
                if (w <= 0) continue;
                for (int j = 0; j < RW; j++) H[64 * j + c.lane] = 0;
                for (int q = 0; q < P.n_paths; q++) {
                    const uint64_t q0 = G(P.path_mask)[2 * q], q1 = G(P.path_mask)[2 * q + 1];
                    if (!uniform_i32(((q0 & p.m0) | (q1 & p.m1)) != 0)) continue;
                    const PathRef pq = load_path(c, q);
                    int r2 = 1;
                    const uint64_t Wq = run_and(row_of(path_free_ext(c, pq)), r2, w);
                    for (int j = 0; j < RW; j++) {
                        const uint64_t word = readlane_u64(Wq, j);
                        H[64 * j + c.lane] += (int32_t)((word >> c.lane) & 1ull);
                    }
                }
                int carry = 0;
                if (c.lane == 0) PH[0] = 0;
                for (int j = 0; j < RW; j++) {
                    const int incl = wave_incl_scan_i32(H[64 * j + c.lane], c.lane);
                    PH[64 * j + c.lane + 1] = carry + incl;
                    carry += __builtin_amdgcn_readlane(incl, 63);
                }
                wave_sync();
                for (int j = 0; j < RW; j++) {
                    const int t = 64 * j + c.lane;
                    if (t < S) LOSS[t] += PH[min(t + n, S)] - PH[max(0, t - w + 1)];
                }
                wave_sync();
            }
            long long key = 0x7fffffffffffffffll;
            for (int j = 0; j < RW; j++) {
                const uint64_t word = readlane_u64(okb, j);
                const int t = 64 * j + c.lane;
                if ((word >> c.lane) & 1ull) { const long long kk = ((long long)LOSS[t] << 16) | t; key = kk < key ? kk : key; }
            }
            key = wave_min_i64(key);
            const long long loss = key >> 16;
            if (loss < best_loss) { best_loss = loss; take(k, m, (int)(key & 0xFFFF), n, p); }   // strict <: first of the minima
            wave_sync();
        }
    }
    if (ch.route >= 0) return;                                     // (action, False, False)
    if (policy == ONGYM_POLICY_LOWEST_FRAGMENTATION && bosnr) bres = 0;
    ch.flags = (bres ? ONGYM_F_BLOCKED_RESOURCES : 0) | (bosnr ? ONGYM_F_BLOCKED_OSNR : 0);
}

}  // namespace ongym
