// group_replicas.hip — diagnostic prototype (not part of the product): the SPECTRUM phase of one QRMSA request
// (AND of the route's link rows, run-AND ladder for n+1 slots, first fit, mark the block on the route's links, release an
// older block) in two layouts, at the product's LDS footprint per replica (7 648 B, NSFNET-320: 21 links x 10 words):
//
//   A  one replica per wavefront (the product's layout): lane w < 10 = bitmap word w, request parameters wave-uniform
//      (scalar registers, scalar branches), 20 workgroups per CU = 5 waves per SIMD.
//   B  four replicas per wavefront: 16-lane groups (DPP rows), every request parameter a vector register, loops run to the
//      maximum over the four groups with predication; one workgroup = one wave = 4 x 7 648 B of LDS -> 5 waves per CU.
//
// Both process the same synthetic request stream per replica (hash of (replica, step): 2-4 links, 2-9 slots) and must leave
// identical bitmaps (checked).  Output: requests/s of either layout.  Build: hipcc --offload-arch=gfx950 -O3 -o group_replicas group_replicas.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int E = 21, NW = 10, S = 320, RING = 8;
constexpr int LDS_PER_REPLICA = 7648;   // the product's block (records, release times, list ... are only reserved here)

__host__ __device__ inline unsigned hash32(unsigned a, unsigned b) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x;
}
// request of (replica, step): hops in 2..4, links (distinct by construction: l, l+5, l+10, l+15 mod 21), slots 2..9
struct Req { int hops, l0, l1, l2, l3, n; };
__host__ __device__ inline Req draw(unsigned replica, unsigned step) {
    const unsigned h = hash32(replica, step);
    Req q;
    q.hops = 2 + (h & 3) % 3;
    q.l0 = (h >> 4) % E; q.l1 = (q.l0 + 5) % E; q.l2 = (q.l0 + 10) % E; q.l3 = (q.l0 + 15) % E;
    q.n = 2 + ((h >> 12) & 7);
    return q;
}

__device__ __forceinline__ unsigned dpp_wave_shl1(unsigned x) {   // lane i <- lane i+1 (whole wave), 0 into lane 63
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned dpp_row_shl1(unsigned x) {    // lane i <- lane i+1 inside each row of 16, 0 into the row's last lane
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned range_mask(int w, int lo, int hi) {   // bits [lo, hi) of word w (slot coordinates)
    const int a = max(lo - 32 * w, 0), b = min(hi - 32 * w, 32);
    if (b <= a) return 0u;
    return (b - a == 32) ? ~0u : (((1u << (b - a)) - 1u) << a);
}

// ---- layout A ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, 5) void k_one(unsigned *grid, int nsteps, unsigned long long *found_out) {
    extern __shared__ unsigned smem[];
    unsigned *occ = smem;                       // [E][NW]
    unsigned *ring = smem + E * NW;             // [RING][2]: (start | n << 16 | hops << 24, l0 | l1<<8 | l2<<16 | l3<<24)
    const int lane = threadIdx.x, replica = blockIdx.x;
    for (int i = lane; i < E * NW; i += 64) occ[i] = grid[(size_t)replica * E * NW + i];
    if (lane < 2 * RING) ring[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    unsigned long long found = 0;
    for (int it = 0; it < nsteps; ++it) {
        // release the block taken RING steps ago
        const unsigned ra = ring[2 * (it % RING)], rb = ring[2 * (it % RING) + 1];
        if (ra) {
            const int s0 = ra & 0xFFFF, n0 = (ra >> 16) & 0xFF, h0 = ra >> 24;
            if (lane < h0) {
                const int l = (rb >> (8 * lane)) & 0xFF, hi = min(s0 + n0 + 1, S);
                const int w0 = s0 >> 5;                                           // a block of <= 10 slots spans <= 2 words
                atomicOr(&occ[l * NW + w0], range_mask(w0, s0, hi));
                if (((hi - 1) >> 5) != w0) atomicOr(&occ[l * NW + w0 + 1], range_mask(w0 + 1, s0, hi));
            }
        }
        __builtin_amdgcn_wave_barrier();
        const Req q = draw(replica, it);
        unsigned x = lane < NW ? ~0u : 0u;
        if (lane < NW) {
            x &= occ[q.l0 * NW + lane] & occ[q.l1 * NW + lane];
            if (q.hops > 2) x &= occ[q.l2 * NW + lane];
            if (q.hops > 3) x &= occ[q.l3 * NW + lane];
        }
        if (lane == NW) x = 1u;                                  // the virtual free slot S
        const int m = q.n + 1;
        for (int r = 1; r < m;) {                                // run-AND ladder (scalar loop)
            const int s = min(r, m - r);
            const unsigned nx = dpp_wave_shl1(x);
            x &= __builtin_amdgcn_alignbit(nx, x, s);
            r += s;
        }
        const unsigned long long bal = __ballot(x != 0);
        unsigned ra2 = 0, rb2 = 0;
        if (bal) {
            const int fl = __ffsll((long long)bal) - 1;
            const unsigned wd = __builtin_amdgcn_readlane(x, fl);
            const int start = 32 * fl + __ffs(wd) - 1;
            if (start + q.n <= S) {
                found++;
                const int hi = min(start + q.n + 1, S);
                if (lane < q.hops) {
                    const int l = lane == 0 ? q.l0 : lane == 1 ? q.l1 : lane == 2 ? q.l2 : q.l3;
                    const int w0 = start >> 5;
                    atomicAnd(&occ[l * NW + w0], ~range_mask(w0, start, hi));
                    if (((hi - 1) >> 5) != w0) atomicAnd(&occ[l * NW + w0 + 1], ~range_mask(w0 + 1, start, hi));
                }
                ra2 = start | (q.n << 16) | (q.hops << 24);
                rb2 = q.l0 | (q.l1 << 8) | (q.l2 << 16) | (q.l3 << 24);
            }
        }
        if (lane == 0) { ring[2 * (it % RING)] = ra2; ring[2 * (it % RING) + 1] = rb2; }
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = lane; i < E * NW; i += 64) grid[(size_t)replica * E * NW + i] = occ[i];
    if (lane == 0) found_out[replica] = found;
}

// ---- layout B ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, 2) void k_four(unsigned *grid, int nsteps, unsigned long long *found_out) {
    extern __shared__ unsigned smem[];
    const int lane = threadIdx.x, g = lane >> 4, w = lane & 15;
    const int replica = blockIdx.x * 4 + g;
    unsigned *occ = smem + g * (LDS_PER_REPLICA / 4);            // this group's block
    unsigned *ring = occ + E * NW;
    for (int i = w; i < E * NW; i += 16) occ[i] = grid[(size_t)replica * E * NW + i];
    if (w < 2 * RING) ring[w] = 0;
    __builtin_amdgcn_wave_barrier();
    unsigned found = 0;
    for (int it = 0; it < nsteps; ++it) {
        const unsigned ra = ring[2 * (it % RING)], rb = ring[2 * (it % RING) + 1];   // per group (vector values)
        {
            const int s0 = ra & 0xFFFF, n0 = (ra >> 16) & 0xFF, h0 = ra >> 24;
            if (ra && w < h0) {
                const int l = (rb >> (8 * w)) & 0xFF, hi = min(s0 + n0 + 1, S);
                const int w0 = s0 >> 5;                                           // a block of <= 10 slots spans <= 2 words
                atomicOr(&occ[l * NW + w0], range_mask(w0, s0, hi));
                if (((hi - 1) >> 5) != w0) atomicOr(&occ[l * NW + w0 + 1], range_mask(w0 + 1, s0, hi));
            }
        }
        __builtin_amdgcn_wave_barrier();
        const Req q = draw(replica, it);                                          // vector registers
        unsigned x = w < NW ? ~0u : 0u;
        if (w < NW) {
            x &= occ[q.l0 * NW + w] & occ[q.l1 * NW + w];
            if (q.hops > 2) x &= occ[q.l2 * NW + w];                              // predicated per group
            if (q.hops > 3) x &= occ[q.l3 * NW + w];
        }
        if (w == NW) x = 1u;
        const int m = q.n + 1;
        int r = 1;
        while (__ballot(r < m)) {                                                 // ladder to the longest of the four
            const int s = min(r, m - r);
            const unsigned nx = dpp_row_shl1(x);
            const unsigned y = x & __builtin_amdgcn_alignbit(nx, x, s & 31);
            if (r < m) { x = y; r += s; }
        }
        const unsigned long long bal = __ballot(x != 0);
        const unsigned gb = (unsigned)(bal >> (16 * g)) & 0xFFFFu;                // this group's lanes
        unsigned ra2 = 0, rb2 = 0;
        if (gb) {
            const int fl = __ffs(gb) - 1;
            const unsigned wd = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (16 * g + fl), (int)x);
            const int start = 32 * fl + __ffs(wd) - 1;
            if (start + q.n <= S) {
                found++;
                const int hi = min(start + q.n + 1, S);
                if (w < q.hops) {
                    const int l = w == 0 ? q.l0 : w == 1 ? q.l1 : w == 2 ? q.l2 : q.l3;
                    const int w0 = start >> 5;
                    atomicAnd(&occ[l * NW + w0], ~range_mask(w0, start, hi));
                    if (((hi - 1) >> 5) != w0) atomicAnd(&occ[l * NW + w0 + 1], ~range_mask(w0 + 1, start, hi));
                }
                ra2 = start | (q.n << 16) | (q.hops << 24);
                rb2 = q.l0 | (q.l1 << 8) | (q.l2 << 16) | (q.l3 << 24);
            }
        }
        if (w == 0) { ring[2 * (it % RING)] = ra2; ring[2 * (it % RING) + 1] = rb2; }
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = w; i < E * NW; i += 16) grid[(size_t)replica * E * NW + i] = occ[i];
    if (w == 0) found_out[replica] = found;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 65536, steps = argc > 2 ? atoi(argv[2]) : 1000;
    std::vector<unsigned> init((size_t)B * E * NW);
    for (size_t i = 0; i < init.size(); i++) init[i] = ~0u;                       // empty network: every slot free
    unsigned *ga, *gb;
    unsigned long long *fa, *fb;
    hipMalloc(&ga, init.size() * 4); hipMalloc(&gb, init.size() * 4);
    hipMalloc(&fa, (size_t)B * 8); hipMalloc(&fb, (size_t)B * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_four), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * LDS_PER_REPLICA);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(ga, init.data(), init.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(gb, init.data(), init.size() * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_one, dim3(B), dim3(64), LDS_PER_REPLICA, 0, ga, steps, fa);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[0], e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_four, dim3(B / 4), dim3(64), 4 * LDS_PER_REPLICA, 0, gb, steps, fb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[1], e0, e1);
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
    }
    std::vector<unsigned> ha(init.size()), hb(init.size());
    std::vector<unsigned long long> ca(B), cb(B);
    hipMemcpy(ha.data(), ga, ha.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), gb, hb.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ca.data(), fa, (size_t)B * 8, hipMemcpyDeviceToHost); hipMemcpy(cb.data(), fb, (size_t)B * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; unsigned long long tot = 0;
    for (size_t i = 0; i < ha.size(); i++) bad += ha[i] != hb[i];
    for (int i = 0; i < B; i++) { bad += ca[i] != cb[i]; tot += ca[i]; }
    int na = 0, nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&na, k_one, 64, LDS_PER_REPLICA);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_four, 64, 4 * LDS_PER_REPLICA);
    printf("spectrum phase only (route AND, run-AND ladder, first fit, mark, release), %d replicas x %d requests, %.1f%% placed\n",
           B, steps, 100.0 * (double)tot / ((double)B * steps));
    printf("A one replica per wave   : %8.3f ms  %.3e requests/s  (%d workgroups = replicas per CU)\n", ms[0], (double)B * steps / ms[0] * 1e3, na);
    printf("B four replicas per wave : %8.3f ms  %.3e requests/s  (%d workgroups = %d replicas per CU)\n", ms[1], (double)B * steps / ms[1] * 1e3, nb, 4 * nb);
    printf("B / A = %.2f; final bitmaps and placement counts %s\n", ms[0] / ms[1], bad ? "DIFFER" : "identical");
    return bad != 0;
}
