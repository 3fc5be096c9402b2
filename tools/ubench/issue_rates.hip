// issue_rates.hip — diagnostic microbenchmark (not part of the product): wave-instruction issue rates on gfx950 as a function
// of waves per SIMD, for the instruction classes the QRMSA kernel is made of.  Build: hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 64
#define ITERS 2000

// each kernel: ITERS iterations of REP instructions of one class; out keeps the compiler honest
#define KERNEL(name, decl, body)                                                           \
    __global__ __launch_bounds__(64) void name(unsigned long long *out, int iters) {                            \
        decl;                                                                               \
        for (int it = 0; it < iters; ++it) {                                                \
            _Pragma("unroll") for (int r = 0; r < REP / 4; ++r) { body; }                   \
        }                                                                                   \
        FIN;                                                                                \
    }

// ---- VALU 32-bit add: 4 independent chains
#define FIN out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(a0 + a1 + a2 + a3)
KERNEL(k_valu32, unsigned a0 = threadIdx.x; unsigned a1 = 1; unsigned a2 = 2; unsigned a3 = 3,
       asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)))
// dependent chain (one register)
KERNEL(k_valu32_dep, unsigned a0 = threadIdx.x; unsigned a1 = 1; unsigned a2 = 2; unsigned a3 = 3,
       asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0"
                    : "+v"(a0)))
#undef FIN
#define FIN out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(a0 + a1 + a2 + a3)
KERNEL(k_shl64, unsigned long long a0 = threadIdx.x; unsigned long long a1 = 1; unsigned long long a2 = 2; unsigned long long a3 = 3,
       asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)))
#undef FIN
#define FIN out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(a0 + a1 + a2 + a3)
KERNEL(k_fma64, double a0 = threadIdx.x; double a1 = 1; double a2 = 2; double a3 = 3,
       asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)))
KERNEL(k_add64f, double a0 = threadIdx.x; double a1 = 1; double a2 = 2; double a3 = 3,
       asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)))
#undef FIN
#define FIN out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(a0 + a1 + a2 + a3)
KERNEL(k_mullo, unsigned a0 = threadIdx.x; unsigned a1 = 1; unsigned a2 = 2; unsigned a3 = 3,
       asm volatile("v_mul_lo_u32 %0, %0, %0\n v_mul_lo_u32 %1, %1, %1\n v_mul_lo_u32 %2, %2, %2\n v_mul_lo_u32 %3, %3, %3"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)))
KERNEL(k_dppmov, unsigned a0 = threadIdx.x; unsigned a1 = 1; unsigned a2 = 2; unsigned a3 = 3,
       asm volatile("v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %2, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shl:1 row_mask:0xf bank_mask:0xf"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)))
KERNEL(k_alignbit, unsigned a0 = threadIdx.x; unsigned a1 = 1; unsigned a2 = 2; unsigned a3 = 3,
       asm volatile("v_alignbit_b32 %0, %0, %1, 3\n v_alignbit_b32 %1, %1, %2, 3\n v_alignbit_b32 %2, %2, %3, 3\n v_alignbit_b32 %3, %3, %0, 3"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)))
#undef FIN
// ---- SALU: 4 independent s_add
#define FIN out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(s0 + s1 + s2 + s3)
KERNEL(k_salu, unsigned s0 = blockIdx.x; unsigned s1 = 1; unsigned s2 = 2; unsigned s3 = 3,
       asm volatile("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3"
                    : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc"))
#undef FIN
// ---- mixed: 2 VALU + 2 SALU interleaved
#define FIN out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(a0 + a1 + s0 + s1)
KERNEL(k_mixed, unsigned a0 = threadIdx.x; unsigned a1 = 1; unsigned s0 = blockIdx.x; unsigned s1 = 3,
       asm volatile("v_add_u32 %0, %0, %0\n s_add_u32 %2, %2, %2\n v_add_u32 %1, %1, %1\n s_add_u32 %3, %3, %3"
                    : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : : "scc"))
#undef FIN
// ---- readlane (VALU->SGPR) chains
#define FIN out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(a0 + s0 + s1 + s2 + s3)
KERNEL(k_readlane, unsigned a0 = threadIdx.x; unsigned s0 = 0; unsigned s1 = 0; unsigned s2 = 0; unsigned s3 = 0,
       asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9"
                    : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0)))
#undef FIN
// ---- LDS read b32 / b64 (independent, no wait inside the group)
#define FIN out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(a0 + a1 + a2 + a3)
__global__ __launch_bounds__(64) void k_ldsread(unsigned long long *out, int iters) {
    __shared__ unsigned buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = i;
    __syncthreads();
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    unsigned addr = threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 4; ++r) {
            unsigned t0, t1, t2, t3;
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                         : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(addr));
            a0 += t0; a1 += t1; a2 += t2; a3 += t3;
        }
    }
    FIN;
}
// dependent LDS read chain: latency
__global__ __launch_bounds__(64) void k_ldslat(unsigned long long *out, int iters) {
    __shared__ unsigned buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned addr = threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = addr;
}
#undef FIN

typedef void (*kern_t)(unsigned long long *, int);
struct Entry { const char *name; kern_t k; int insts_per_iter; };

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned long long *out;
    hipMalloc(&out, (size_t)cus * 64 * 64 * 8);
    Entry es[] = {{"v_add_u32 (4 indep)", k_valu32, REP}, {"v_add_u32 (dependent)", k_valu32_dep, REP}, {"v_lshlrev_b64", k_shl64, REP},
                  {"v_fma_f64", k_fma64, REP}, {"v_add_f64", k_add64f, REP}, {"v_mul_lo_u32", k_mullo, REP}, {"v_mov_b32_dpp", k_dppmov, REP},
                  {"v_alignbit_b32", k_alignbit, REP}, {"s_add_u32", k_salu, REP}, {"mixed v_add/s_add", k_mixed, REP},
                  {"v_readlane_b32", k_readlane, REP}, {"ds_read_b32 x4 + wait", k_ldsread, REP}, {"ds_read_b32 dependent", k_ldslat, REP}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs %d; clock reported %d kHz. cycles per wave-instruction PER SIMD at 2.4 GHz (lower = faster); waves/SIMD = blocks per CU / 4\n", cus, prop.clockRate);
    printf("%-26s", "instruction");
    const int wps[] = {1, 2, 4, 5, 8};
    for (int w : wps) printf("  %4dw/SIMD", w);
    printf("\n");
    for (auto &e : es) {
        printf("%-26s", e.name);
        for (int w : wps) {
            const int blocks = cus * 4 * w;   // one wave per block; the dispatcher spreads them over CUs/SIMDs
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, out, ITERS);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: w waves x ITERS x insts ; cycles = ms*2.4e6
            const double cyc = (double)ms * 2.4e6 / ((double)w * ITERS * e.insts_per_iter);
            printf("  %9.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
