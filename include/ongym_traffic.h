/*
 * ongym_traffic.h — definition of the device request stream ("request source A" of ongym.h).
 *
 * The reference draws traffic from an UNSEEDED CPython random.Random() (qrmsa.pyx:241, quirk Q2), so there is no
 * reference stream to reproduce bit-for-bit; what defines parity is the DISTRIBUTION and the DRAW ORDER of
 * QRMSAEnv._next_service / _get_node_pair (qrmsa.pyx:1079-1089, 1134-1148):
 *     1. at  = float32(current_time + expovariate(1/mean_inter_arrival))      mean_iat = 1/(load/holding) (:1130)
 *     2. ht  = float32(expovariate(1/mean_holding))                            (both sampled in float32 here)
 *     3. src ~ node_request_probabilities
 *     4. dst ~ the same weights with src zeroed and renormalised
 *     5. bit_rate ~ choices(bit_rates, probs)  |  randint(lo, hi)
 * This header fixes one counter-based generator with exactly that order (5 draws per request) so that the HIP kernels
 * and any host-side checker produce the same requests from (seed, replica, request index).  Every operation below is an
 * IEEE-754 correctly rounded one (+, *, fma, conversions) or an integer one, so host and device agree bit for bit; the
 * natural logarithm is therefore spelled out instead of calling libm / ocml.
 *
 * Usable from C (gcc, compile with -ffp-contract=off), C++ and HIP device code.
 */
#ifndef ONGYM_TRAFFIC_H
#define ONGYM_TRAFFIC_H
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define ONGYM_HD __host__ __device__ static inline
#else
#define ONGYM_HD static inline
#endif

#define ONGYM_DRAWS_PER_REQUEST 5

/* hipcc defaults to -ffp-contract=fast, which would fuse a*b+c across statements on the device only */
#if defined(__clang__)
#define ONGYM_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define ONGYM_NO_CONTRACT
#endif

ONGYM_HD uint64_t ongym_mix64(uint64_t z) { /* splitmix64 finaliser */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* key of replica r's stream */
ONGYM_HD uint64_t ongym_stream_key(uint64_t seed, uint64_t replica) {
    return ongym_mix64(seed ^ ongym_mix64(replica + 0x9E3779B97F4A7C15ull));
}

/* the counter-th 64-bit word of a stream, as a double in [0,1) with 53 random bits */
ONGYM_HD double ongym_uniform(uint64_t key, uint64_t counter) {
    uint64_t x = ongym_mix64(key + (counter + 1) * 0x9E3779B97F4A7C15ull);
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/*
 * float32 natural logarithm of a double x in (0, 1] (x = 1 - u is exact in double). Built from exactly rounded
 * float operations only (conversion, fmaf, multiply) so host and device agree bit for bit; no division.
 * log(m), m in (sqrt(1/2), sqrt(2)], is a degree-9 polynomial in f = m - 1 (max abs error 3.4e-8).
 */
ONGYM_HD float ongym_logf_det(double x) {
    ONGYM_NO_CONTRACT
    union { double d; uint64_t u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7FF) - 1023;
    v.u = (v.u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull; /* m in [1,2) */
    float m = (float)v.d;                                         /* may round up to 2.0f: handled by the halving */
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float p = -0.07638592272996902f;
    p = fmaf(p, f, 0.129141703248024f);
    p = fmaf(p, f, -0.13240842521190643f);
    p = fmaf(p, f, 0.14180079102516174f);
    p = fmaf(p, f, -0.16609126329421997f);
    p = fmaf(p, f, 0.2000207155942917f);
    p = fmaf(p, f, -0.25001585483551025f);
    p = fmaf(p, f, 0.33333325386047363f);
    p = fmaf(p, f, -0.49999988079071045f);
    p = fmaf(p, f, 1.0f);
    return fmaf((float)e, 0.693147182464599609375f, p * f);
}

/* random.expovariate(1/mean) = -log(1 - random()) / lambd, evaluated in float32 as -logf(1-u) * mean: the request
 * clocks are C floats in the reference anyway (envs/qrmsa.pyx:1068-1075), the sampler's relative error is ~1e-7 */
ONGYM_HD float ongym_expovariate_f(double u, float mean) {
    ONGYM_NO_CONTRACT
    return -ongym_logf_det(1.0 - u) * mean;
}

/* first index i with x < cum[i] (CPython choices: bisect(cum_weights, u*total, 0, n-1)); cum[n-1] is the total */
ONGYM_HD int ongym_bisect(const double *cum, int n, double x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (x < cum[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

typedef struct ongym_traffic_params {
    double mean_inter_arrival; /* 1/(load/mean_holding), qrmsa.pyx:1130 */
    double mean_holding;
    const double *node_cum;    /* [n_nodes] */
    int n_nodes;
    int bit_rate_mode;         /* 0 discrete, 1 continuous */
    const double *bit_rates;   /* [n_bit_rates] */
    const double *bit_rate_cum;
    int n_bit_rates;
    int bit_rate_lo, bit_rate_hi;
} ongym_traffic_params;

typedef struct ongym_drawn_request {
    float arrival_time, holding_time, bit_rate;
    int source, destination;
} ongym_drawn_request;

/*
 * Request number `index` (0-based, counted over the life of the replica) given the clock before it.
 * `current_time` is the double the reference keeps in self.current_time (= (double)(float) of the last arrival).
 */
ONGYM_HD ongym_drawn_request ongym_draw_request(uint64_t key, uint64_t index, double current_time,
                                                const ongym_traffic_params *tp) {
    ONGYM_NO_CONTRACT
    ongym_drawn_request r;
    uint64_t c = index * ONGYM_DRAWS_PER_REQUEST;
    double u0 = ongym_uniform(key, c + 0), u1 = ongym_uniform(key, c + 1), u2 = ongym_uniform(key, c + 2),
           u3 = ongym_uniform(key, c + 3), u4 = ongym_uniform(key, c + 4);
    r.arrival_time = (float)current_time + ongym_expovariate_f(u0, (float)tp->mean_inter_arrival);
    r.holding_time = ongym_expovariate_f(u1, (float)tp->mean_holding);
    int n = tp->n_nodes;
    double total = tp->node_cum[n - 1];
    int src = ongym_bisect(tp->node_cum, n, u2 * total);
    /* destination: weights with src zeroed (renormalisation only rescales u): skip src's interval */
    double lo_s = src > 0 ? tp->node_cum[src - 1] : 0.0;
    double w_s = tp->node_cum[src] - lo_s;
    double x = u3 * (total - w_s);
    if (x >= lo_s) x += w_s;
    int dst = ongym_bisect(tp->node_cum, n, x);
    if (dst == src) dst = (src + 1 < n) ? src + 1 : src - 1; /* rounding guard; never equal to src */
    r.source = src;
    r.destination = dst;
    if (tp->bit_rate_mode == 0) {
        int b = ongym_bisect(tp->bit_rate_cum, tp->n_bit_rates, u4 * tp->bit_rate_cum[tp->n_bit_rates - 1]);
        r.bit_rate = (float)tp->bit_rates[b];
    } else {
        int span = tp->bit_rate_hi - tp->bit_rate_lo + 1;
        int k = (int)(u4 * (double)span);
        if (k >= span) k = span - 1;
        r.bit_rate = (float)(tp->bit_rate_lo + k);
    }
    return r;
}

#endif /* ONGYM_TRAFFIC_H */
