// Developer microbenchmark (round 4): what a "last finisher" hand-over between workgroups on different XCDs costs.
//   every wave of a 512-thread workgroup: 16-byte agent-scope store (sc1) -> s_waitcnt vmcnt(0) -> agent-scope atomic add returning
//   -> workgroup barrier -> 16-byte agent-scope load (sc1) of a row another workgroup stored
// timed with s_memrealtime (100 MHz) per phase, 480 workgroups in flight (the MCTS's trunk launch shape), averaged over workgroups.
// Build: hipcc -O3 --offload-arch=gfx950 sync_cost.hip -o bin/sync_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: every wave its own atomic (128 increments per 16-board group);  MODE 1: workgroup barrier, ONE atomic per workgroup (16 per
// group), result broadcast through LDS behind a second barrier
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(float* rows, unsigned int* cnt, unsigned long long* out, int nwg) {
    __shared__ unsigned int flag[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rows, 0, nwg * 512, 0x00020000);
    // some work first so that the workgroups are out of phase
    float x = threadIdx.x;
    for (int i = 0; i < 200 + 37 * (blockIdx.x & 7); ++i) x = x * 1.0001f + 0.5f;
    const u32x4 v = {__builtin_bit_cast(unsigned int, x), 2u, 3u, (unsigned int)blockIdx.x};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (lane < 4) __builtin_amdgcn_raw_buffer_store_b128(v, rs, wave * 64 + lane * 16, blockIdx.x * 512, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    unsigned int old = 0;
    if (MODE == 0) {
        if (lane == 0) old = __hip_atomic_fetch_add(cnt + (blockIdx.x >> 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
    } else {
        __syncthreads();
        if (threadIdx.x == 0) flag[0] = __hip_atomic_fetch_add(cnt + 64 + (blockIdx.x >> 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0 && lane == 0) flag[wave] = old;
    __syncthreads();
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    const int other = (blockIdx.x ^ 1) < nwg ? (blockIdx.x ^ 1) : blockIdx.x;      // a row of a workgroup on another XCD
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 8, other * 512, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned long long* o = out + 4ull * blockIdx.x;
        o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = (t4 - t3) + (w[0] == 0x12345u ? 1 : 0) + (flag[3] == 0xFFFFFFFFu ? 1 : 0);
    }
}

int main() {
    const int nwg = 480;
    float* rows; unsigned int* cnt; unsigned long long* out;
    hipMalloc(&rows, nwg * 512); hipMalloc(&cnt, 4096); hipMalloc(&out, nwg * 32);
    hipMemset(cnt, 0, 4096);
    std::vector<unsigned long long> h(4 * nwg);
    for (int rep = 0; rep < 6; ++rep) {
        if (rep & 1) hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(512), 0, 0, rows, cnt, out, nwg);
        else hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(512), 0, 0, rows, cnt, out, nwg);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, nwg * 32, hipMemcpyDeviceToHost);
        double s[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
        for (int i = 0; i < nwg; ++i) for (int j = 0; j < 4; ++j) { s[j] += h[4 * i + j]; if (h[4 * i + j] > mx[j]) mx[j] = h[4 * i + j]; }
        printf("rep %d mode %d (wave 0 of each workgroup, us, mean / max over %d workgroups): store+ack %.2f / %.2f   atomic %.2f / %.2f   barrier %.2f / %.2f   load %.2f / %.2f\n",
               rep, rep & 1, nwg, s[0] / nwg / 100, mx[0] / 100, s[1] / nwg / 100, mx[1] / 100, s[2] / nwg / 100, mx[2] / 100, s[3] / nwg / 100, mx[3] / 100);
    }
    return 0;
}
