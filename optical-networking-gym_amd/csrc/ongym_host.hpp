// ongym_host.hpp — host-side state of one environment, shared by the translation units of libongym_hip.so
// (ongym_hip.hip: generic kernels + C ABI; ongym_fast.hip: the lean policy kernels).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "ongym_device.hpp"

using ongym::Params;

struct ongym_env {
    ongym_config cfg{};
    Params P{};
    Params *d_P = nullptr;          // device copy read by the kernels (scalar loads); refreshed by push_params
    hipStream_t stream = nullptr;   // the stream every call runs on: own_stream, or the caller's (ongym_set_stream)
    hipStream_t own_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    size_t lds = 0;
    std::vector<void *> allocs;
    std::string err;
    void *d_trace = nullptr;        // owned copy of a host trace
    ongym_step_rec *d_out = nullptr; size_t d_out_n = 0;
    int32_t *d_actions = nullptr; int32_t *d_act_out = nullptr; uint8_t *d_flag_out = nullptr; uint8_t *d_mask = nullptr;
    float *d_obs = nullptr; uint8_t *d_obsmask = nullptr;   // lazily allocated staging for ongym_observe with host buffers
    int32_t *d_scratch_i = nullptr; size_t scratch_i_bytes = 0; double *d_scratch_d = nullptr;
    void *h_pinned = nullptr; size_t h_pinned_bytes = 0;     // pinned staging of ongym_step_actions_bundle
    bool has_source = false;
    uint64_t replica_base = 0;      // global index of this environment's first replica (ongym_seed_base)
    std::vector<double> cfg_bit_rates;     // host copy of the discrete bit rates
    bool fast_ok = false;           // the configuration is eligible for k_fast (see fast_eligible)
    bool fast_m64 = false;
    bool fast_wide = true;          // some slot count of the traffic table exceeds 32: the general (`w`) lean kernels run
    bool fast_lb_ok = false, fast_hsnr_ok = false, fast_lf_ok = false;   // ... and for the lean kernels of the other policies
    bool trace_used = false;        // a trace the lean kernel cannot replay was installed: its records may not fit the lean codec
    bool trace_fast_ok = false;     // the installed (host) trace only carries bit rates of the configured table
    size_t fast_lds = 0;
};

#define HIP_TRY(env, expr)                                                                               \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            (env)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                              \
            return ONGYM_E_HIP;                                                                          \
        }                                                                                                \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is set on the CURRENT device's function object: keep, per (device, kernel), the
// largest request ever made (an environment with a smaller LDS block must not lower the limit under one created earlier)
// and only ever raise it.  Defined in ongym_hip.hip; serialised by a mutex (environments may be created from several threads).
hipError_t raise_lds_limit(int device, const void *kernel, size_t bytes);

// ongym_fast.hip, two units per policy id p: the lean kernels k_fast<..., p> — LDS limit, launch, occupancy query.  The `w`
// units are the general build (slot counts up to 512); the plain ones assume every slot count of the configuration is <= 32
// (ongym_env::fast_wide == false: one-step run-AND shifts, two-word marks, no wide-release path: +6 % on NSFNET-320).
namespace ongym {
#define ONGYM_FAST_DECL(p)                                                            \
    int fast_prepare_p##p(ongym_env *env);                                            \
    int fast_launch_p##p(ongym_env *env, int nsteps, ongym_step_rec *d_out);          \
    int fast_occupancy_p##p(ongym_env *env, int *blocks_per_cu, int *lds_bytes);
ONGYM_FAST_DECL(0) ONGYM_FAST_DECL(1) ONGYM_FAST_DECL(2) ONGYM_FAST_DECL(10)
ONGYM_FAST_DECL(0w) ONGYM_FAST_DECL(1w) ONGYM_FAST_DECL(2w) ONGYM_FAST_DECL(10w)
#undef ONGYM_FAST_DECL
// fn = fast_prepare / fast_launch / fast_occupancy of policy unit p, narrow or wide build as the environment needs
#define ONGYM_FAST_CALL(fn, p, env, ...) ((env)->fast_wide ? fn##_p##p##w((env), ##__VA_ARGS__) : fn##_p##p((env), ##__VA_ARGS__))
// policies with a lean kernel (the ids above)
inline bool fast_policy_supported(int policy) {
    return policy == ONGYM_POLICY_FIRST_FIT || policy == ONGYM_POLICY_LOAD_BALANCING || policy == ONGYM_POLICY_HIGHEST_SNR ||
           policy == ONGYM_POLICY_LOWEST_FRAGMENTATION;
}
}
