// ongym_fast.hip — the lean fused policy+step kernels (ongym_fast.hpp), two translation units per policy:
//   hipcc -c -DONGYM_FAST_POLICY=<ONGYM_POLICY_* id> -DONGYM_FAST_WIDE=<0|1> ongym_fast.hip -o ongym_fast_p<id>[w].o
// (WIDE = 0: every slot count of the configuration is <= 32, see ongym_env::fast_wide)
// (the units compile in parallel; see __graft_entry__.build).  Each exports fast_launch / fast_occupancy / fast_prepare
// for its policy; ongym_hip.hip dispatches on the policy id.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "ongym_host.hpp"
#include "ongym_fast.hpp"

#ifndef ONGYM_FAST_POLICY
#error "compile with -DONGYM_FAST_POLICY=<policy id>"
#endif

namespace ongym {

constexpr int kPol = ONGYM_FAST_POLICY;

// M64: link masks need two words (32 < n_links <= 52); ENT: interferers per lane cached in registers; WAVES: waves per
// SIMD the register allocation is bounded for; POL: the policy (part of the kernel's name: one set of kernels per unit)
#ifndef ONGYM_FAST_WIDE
#define ONGYM_FAST_WIDE 1
#endif
constexpr bool kWide = ONGYM_FAST_WIDE != 0;
// (WIDE is part of the kernel's name: the narrow and the wide unit of a policy are linked into one library)
template <bool M64, bool REC, int ENT, int WAVES, bool TRACE, int POL, bool WIDE>
__global__ __launch_bounds__(64, WAVES) void k_fast(const Params *__restrict__ Pp, int nsteps, ongym_step_rec *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    fast_run<M64, REC, ENT, TRACE, POL, WIDE>(*Pp, nsteps, out, smem);
}

static size_t lds_of(const ongym_env *env) {
    return fast_lds_bytes(env->P.n_links, env->P.row_words, env->P.capacity, env->fast_m64, env->P.n_slots, kPol);
}

// Register budget (waves per SIMD) of the non-M64 instantiation that runs.  gfx950 hands out LDS in granules (measured: 8160 B
// per workgroup gave 18 workgroups per CU, 7648 B gave 20): first fit and load balancing take 5 waves per SIMD (96 VGPRs, a few
// spills in cold code) when the LDS block admits that many replicas, else 4.  Measured (tools/ab_policies.sh, NSFNET-320,
// B = 65 536): load balancing +10.5 % at 5 over 4; highest SNR is gather-bound and loses 5 % at 3 (the fourth wave per SIMD
// matters more than the spills it causes); lowest fragmentation gains 11 % at 3 (its route-score walk spills 122 VGPRs at 4).
#ifdef ONGYM_POLICY_WAVES
constexpr int kPolicyWaves = ONGYM_POLICY_WAVES;            // experiment builds
#else
constexpr int kPolicyWaves = kPol == ONGYM_POLICY_HIGHEST_SNR ? 4 : kPol == ONGYM_POLICY_LOWEST_FRAGMENTATION ? 3 : 5;
#endif
static int waves_of(const ongym_env *env) {
    if (env->fast_m64) return 3;
    if (kPolicyWaves != 5) return kPolicyWaves;
    const size_t granule = 1280, per_cu = (160 * 1024) / (((lds_of(env) + granule - 1) / granule) * granule);
    return per_cu >= 17 ? 5 : 4;
}

template <class F>
static int with_kernel(const ongym_env *env, bool rec, bool trace, F &&f) {
    const int waves = waves_of(env);
#define ONGYM_TRY_VARIANT(M64, ENT, WAVES)                                                                         \
    if (env->fast_m64 == M64 && waves == WAVES) {                                                                  \
        if (trace && rec) return f(k_fast<M64, true, ENT, WAVES, true, kPol, kWide>);                                     \
        if (trace) return f(k_fast<M64, false, ENT, WAVES, true, kPol, kWide>);                                           \
        if (rec) return f(k_fast<M64, true, ENT, WAVES, false, kPol, kWide>);                                             \
        return f(k_fast<M64, false, ENT, WAVES, false, kPol, kWide>);                                                     \
    }
    ONGYM_TRY_VARIANT(true, 4, 3)
    if constexpr (kPolicyWaves == 2) { ONGYM_TRY_VARIANT(false, 2, 2) }
    if constexpr (kPolicyWaves == 3) { ONGYM_TRY_VARIANT(false, 2, 3) }
    if constexpr (kPolicyWaves >= 4) { ONGYM_TRY_VARIANT(false, 2, 4) }
    if constexpr (kPolicyWaves == 5) { ONGYM_TRY_VARIANT(false, 2, 5) }
#undef ONGYM_TRY_VARIANT
    return ONGYM_E_LIMIT;
}

#define FAST_CAT2(a, b) a##b
#define FAST_CAT(a, b) FAST_CAT2(a, b)
#if ONGYM_FAST_WIDE
#define FAST_FN(name) FAST_CAT(FAST_CAT(name##_p, ONGYM_FAST_POLICY), w)
#else
#define FAST_FN(name) FAST_CAT(name##_p, ONGYM_FAST_POLICY)
#endif

int FAST_FN(fast_prepare)(ongym_env *env) {
    const size_t lds = lds_of(env);
    if (lds > 160 * 1024) return ONGYM_E_LIMIT;
    if (lds <= 64 * 1024) return 0;
    for (int rec = 0; rec < 2; rec++)
        for (int tr = 0; tr < 2; tr++) {
            const int rc = with_kernel(env, rec != 0, tr != 0, [&](auto kernel) -> int {
                HIP_TRY(env, raise_lds_limit(env->cfg.device, reinterpret_cast<const void *>(kernel), lds));
                return 0;
            });
            if (rc) return rc;
        }
    return 0;
}

int FAST_FN(fast_launch)(ongym_env *env, int nsteps, ongym_step_rec *d_out) {
    const bool tr = env->P.req_mode == kReqTrace;
    const size_t lds = lds_of(env);
    // the kernel sums the running services of its steps in 32 bits (one scalar add per step): a launch is split so that
    // steps x capacity stays below 2^32
    const long long chunk = std::max(1LL, 0xFFFFFFFFLL / std::max(1, env->P.capacity));
    return with_kernel(env, d_out != nullptr, tr, [&](auto kernel) -> int {
        for (long long done = 0; done < nsteps; done += chunk) {
            const int n = (int)std::min<long long>(chunk, nsteps - done);
            hipLaunchKernelGGL(kernel, dim3(env->P.batch), dim3(64), lds, env->stream, env->d_P, n,
                               d_out ? d_out + (size_t)done * env->P.batch : nullptr);
            HIP_TRY(env, hipGetLastError());
        }
        return 0;
    });
}

int FAST_FN(fast_occupancy)(ongym_env *env, int *blocks_per_cu, int *lds_bytes) {
    const size_t lds = lds_of(env);
    *lds_bytes = (int)lds;
    return with_kernel(env, false, false, [&](auto kernel) -> int {
        HIP_TRY(env, hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kernel, 64, lds));
        return 0;
    });
}

}  // namespace ongym
