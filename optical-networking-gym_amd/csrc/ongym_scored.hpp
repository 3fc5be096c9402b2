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

// extra dynamic LDS of the kPolicyScored instantiation (capacity-loss search of MSCL):
//   OKB u64[8][16] | LOSSM int32[8][64*W] | PH int32[64*W + 1 (+1 pad)] | H int32[kScoredWidths][64*W]
constexpr int kScoredWidths = 12;      // window widths whose route counts are accumulated in one pass over the routes
__host__ __device__ inline size_t scored_lds_bytes(int row_words) {
    return (size_t)kMaxMods * kMaxRowWords * 8 + (size_t)kMaxMods * 64 * row_words * 4 + ((size_t)64 * row_words + 2) * 4 +
           (size_t)kScoredWidths * 64 * row_words * 4 + 16;
}

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
    // 0.33 * se + 0.33 * cuts + 0.34 * rss with every product and sum rounded on its own (fp_barrier: no fma)
    const double t1 = fp_barrier(0.33 * se), t2 = fp_barrier(0.33 * (double)uniform_i32(tc)), t3 = fp_barrier(0.34 * rss);
    const double score = fp_barrier(t1 + t2) + t3;
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

// scr: the block described at scored_lds_bytes (16-byte aligned)
template <bool UNIFORM_ALPHA, bool R32>
__device__ __forceinline__ void policy_scored(Ctx &c, int policy, int src, int dst, double margin, Choice &ch, int32_t *scr) {
    const Params &P = c.P;
    const int M = P.n_mods, S = P.n_slots, max_mod = M - 1, K = P.k_paths, RW = P.row_words;
    ch.action = K * M * S; ch.route = -1; ch.mod = -1; ch.slot = -1; ch.n = 0; ch.hops = 0; ch.mylink = 0;
    ch.path = -1; ch.m0 = 0; ch.g.ase = ch.g.nli = 0.0; ch.flags = 0;
    int bres = 0, bosnr = 0;
    double best_score = INFINITY;
    long long best_loss = 0x7fffffffffffffffll;
    uint64_t *OKB = reinterpret_cast<uint64_t *>(scr);
    int32_t *LOSSM = scr + kMaxMods * kMaxRowWords * 2;
    int32_t *PH = LOSSM + kMaxMods * 64 * RW;
    int32_t *H = PH + 64 * RW + 2;
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
        // 1. the GN model for every candidate start of every format: OKB[m] = the starts whose GSNR passes
        bool any_ok = false;
        for (int m = max_mod; m >= 0; m--) {
            if (c.lane < kMaxRowWords) OKB[m * kMaxRowWords + c.lane] = 0ull;
            const int n = uniform_i32(c.nreq[m]);
            if (n <= 0) continue;
            int rr = 1;
            uint64_t v = run_and(free_ext, rr, n + 1);
            if (!any_bits(v)) { bres = 1; continue; }
            uint64_t okb = 0;                                      // lane-distributed like v
            for (;;) {
                const int s0 = first_set(v);
                if (s0 < 0) break;
                if (c.lane == (s0 >> 6)) v &= ~(1ull << (s0 & 63));
                if (UNIFORM_ALPHA && P.ase_shortcut) {
                    // exact lower bound (see policy_first_fit): ASE + self-channel term alone already above the limit. It
                    // only grows with the slot index, so every later start of this format fails too
                    const double bw = P.slot_bw * n, fc = P.f0 + (P.slot_bw * s0) + (P.slot_bw * (n / 2.0));
                    const double lb = (bw * fc * p.ase) * c.rp[0] + c.nlic[m] * (p.w1 * c.selfa[m]);
                    if (uniform_i32(lb >= c.lim[m] * (1.0 + 1e-9))) { bosnr = 1; c.gn_skips++; break; }
                }
                if (L < 0) L = gn_build_list<R32>(c, p.m0, p.m1);
                const GnLin g = gn_eval<UNIFORM_ALPHA, R32>(c, p, L, s0, n, coef_for_mod(c, m));
                if (qot_ok(c, g, m, margin)) { if (c.lane == (s0 >> 6)) okb |= 1ull << (s0 & 63); }
                else bosnr = 1;
            }
            if (c.lane < kMaxRowWords) OKB[m * kMaxRowWords + c.lane] = okb;
            if (any_bits(okb)) {
                any_ok = true;
                for (int j = 0; j < RW; j++) LOSSM[m * 64 * RW + 64 * j + c.lane] = 0;
            }
        }
        wave_sync();
        STAMPW(c, 1);
        if (!any_ok) continue;
        // 2. capacity loss of taking [start, start+n) (no guard, :719-720): per configured bit rate (width w at the candidate's
        //    format) and per route q of the network sharing a link with the candidate route, the free windows of width w of
        //    q's row that overlap the block, i.e. the window starts t in (start - w, start + n).  H_w[t] = number of such
        //    routes with a free window of width w at t; PH_w its prefix sum;
        //        LOSS[m][start] = sum over bit rates of PH_w[min(start+n, S)] - PH_w[max(0, start-w+1)],  w = slots(bit rate, m).
        //    The distinct widths of all (format, bit rate) pairs (a bitmap WB, word per lane) are served by ONE pass over the
        //    routes: a route's row is loaded once and the run-AND ladder extended width by width (ascending).
        //    (The reference's route list holds every route twice, once per direction: a uniform factor 2 on every loss;
        //    the device visits the (a, d) entries with a < d only.)
        uint64_t WB = 0;
        for (int m = max_mod; m >= 0; m--) {
            if (!any_bits(c.lane < kMaxRowWords ? OKB[m * kMaxRowWords + c.lane] : 0ull)) continue;
            for (int b = 0; b < P.n_bit_rates; b++) {
                const int w = uniform_i32(G(P.nreq_tab)[b * kMaxMods + m]);
                if (w > 0 && w <= S && c.lane == (w >> 6)) WB |= 1ull << (w & 63);   // wider than the row: no window anywhere
            }
        }
        auto rank_of = [&](int w) {      // index of width w among the set bits of WB (ascending)
            int below = c.lane < (w >> 6) ? __popcll((unsigned long long)WB)
                      : c.lane == (w >> 6) ? __popcll((unsigned long long)(WB & ((1ull << (w & 63)) - 1ull))) : 0;
#pragma unroll
            for (int mm = 8; mm >= 1; mm >>= 1) below += __shfl_xor(below, mm);      // words live in lanes 0..15
            return uniform_i32(below);
        };
        int D = c.lane < kMaxRowWords ? __popcll((unsigned long long)WB) : 0;
#pragma unroll
        for (int mm = 8; mm >= 1; mm >>= 1) D += __shfl_xor(D, mm);
        D = uniform_i32(D);
        for (int i0 = 0; i0 < D; i0 += kScoredWidths) {            // kScoredWidths widths of the ascending list per pass
            const int nw = min(D - i0, kScoredWidths);
            for (int i = c.lane; i < nw * 64 * RW; i += kWave) H[i] = 0;
            wave_sync();
            STAMPW(c, 2);
            // the widths of this pass, ascending: lane i keeps the i-th
            int wv = 0;
            {
                uint64_t WBp = WB;
                for (int i = 0; i < i0 + nw; i++) {
                    const int w = first_set(WBp);
                    if (c.lane == (w >> 6)) WBp &= ~(1ull << (w & 63));
                    if (c.lane == i - i0) wv = w;
                }
            }
            // the routes of the network = the entries of the (pair, k) table, as the reference's dict of k routes per node pair
            // lists them (:671-674; with fewer routes per pair than the topology holds, only those); one direction per pair
            const int NK = P.n_nodes * K, n_entries = P.n_nodes * NK;
            for (int cbase = 0; cbase < n_entries; cbase += kWave) {             // 64 table entries per ballot
                const int t = cbase + c.lane;
                bool touch = false;
                int myq = -1;
                if (t < n_entries) {
                    const int a = t / NK, d = (t / K) % P.n_nodes;
                    if (a < d) myq = G(P.pair_paths)[t];
                    if (myq >= 0) touch = ((G(P.path_mask)[2 * myq] & p.m0) | (G(P.path_mask)[2 * myq + 1] & p.m1)) != 0;
                }
                uint64_t bal = __ballot(touch);
                if (!bal) continue;
                int q = __builtin_amdgcn_readlane(myq, __ffsll((unsigned long long)bal) - 1);
                bal &= bal - 1;
                PathRef pq = load_path(c, q);
                for (;;) {
                    const bool more = bal != 0;
                    const int qn = more ? __builtin_amdgcn_readlane(myq, __ffsll((unsigned long long)bal) - 1) : q;
                    bal &= bal - 1;
                    const PathRef pn = load_path(c, qn);                          // in flight while this route is processed
                    uint64_t x = row_of(path_free_ext(c, pq));
                    int r2 = 1;
                    for (int i = 0; i < nw; i++) {
                        const int w = __builtin_amdgcn_readlane(wv, i);
                        x = run_and(x, r2, w);
                        int32_t *Hi = H + i * 64 * RW;
                        for (int j = 0; j < RW; j++) {                            // ds_add_u32 without return: no round trip
                            const uint64_t word = readlane_u64(x, j);
                            if (word) __hip_atomic_fetch_add(&Hi[64 * j + c.lane], (int32_t)((word >> c.lane) & 1ull),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                    if (!more) break;
                    q = qn; pq = pn;
                }
            }
            wave_sync();
            STAMPW(c, 3);
            for (int m = max_mod; m >= 0; m--) {
                if (!any_bits(c.lane < kMaxRowWords ? OKB[m * kMaxRowWords + c.lane] : 0ull)) continue;
                const int n = uniform_i32(c.nreq[m]);
                for (int b = 0; b < P.n_bit_rates; b++) {
                    const int w = uniform_i32(G(P.nreq_tab)[b * kMaxMods + m]);
                    if (w <= 0 || w > S) continue;
                    const int i = rank_of(w) - i0;
                    if (i < 0 || i >= nw) continue;
                    const int32_t *Hi = H + i * 64 * RW;
                    int carry = 0;
                    if (c.lane == 0) PH[0] = 0;
                    for (int j = 0; j < RW; j++) {
                        const int incl = wave_incl_scan_i32((int)Hi[64 * j + c.lane], c.lane);
                        PH[64 * j + c.lane + 1] = carry + incl;
                        carry += __builtin_amdgcn_readlane(incl, 63);
                    }
                    wave_sync();
                    for (int j = 0; j < RW; j++) {
                        const int t = 64 * j + c.lane;
                        if (t < S) LOSSM[m * 64 * RW + t] += PH[min(t + n, S)] - PH[max(0, t - w + 1)];
                    }
                    wave_sync();
                }
            }
        }
        STAMPW(c, 4);
        // 3. the passing start of least loss, formats high to low, strict `<`: the first of the minima in the reference's order
        for (int m = max_mod; m >= 0; m--) {
            const uint64_t okb = c.lane < kMaxRowWords ? OKB[m * kMaxRowWords + c.lane] : 0ull;
            if (!any_bits(okb)) continue;
            long long key = 0x7fffffffffffffffll;
            for (int j = 0; j < RW; j++) {
                const uint64_t word = readlane_u64(okb, j);
                const int t = 64 * j + c.lane;
                if ((word >> c.lane) & 1ull) { const long long kk = ((long long)LOSSM[m * 64 * RW + t] << 16) | t; key = kk < key ? kk : key; }
            }
            key = wave_min_i64(key);
            const long long loss = key >> 16;
            if (loss < best_loss) { best_loss = loss; take(k, m, (int)(key & 0xFFFF), uniform_i32(c.nreq[m]), p); }
        }
        wave_sync();
        STAMPW(c, 5);
    }
    if (ch.route >= 0) return;                                     // (action, False, False)
    if (policy == ONGYM_POLICY_LOWEST_FRAGMENTATION && bosnr) bres = 0;
    ch.flags = (bres ? ONGYM_F_BLOCKED_RESOURCES : 0) | (bosnr ? ONGYM_F_BLOCKED_OSNR : 0);
}

}  // namespace ongym
