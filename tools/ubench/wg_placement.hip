// Developer probe: which workgroups of a 512-block launch (58 KB LDS, 512 threads: 2 per CU) share a CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(512, 4) void k(unsigned int* out, unsigned long long* t) {
    __shared__ unsigned int lds[14000];
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = hw;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);     // stay resident so the grid fills the chip
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = lds[0]; out[2 * blockIdx.x + 1] = xcc; t[blockIdx.x] = t0; }
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned int* d; unsigned long long* dt; hipMalloc(&d, 512 * 8); hipMalloc(&dt, 512 * 8);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d, dt);
    hipDeviceSynchronize();
    std::vector<unsigned int> h(1024); hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    std::map<unsigned int, std::vector<int>> cu;
    for (int b = 0; b < 512; ++b) {
        const unsigned int hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned int cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("distinct CUs: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) { if (shown++ < 24) { printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
    std::map<int, int> diffs;
    for (auto& kv : cu) if (kv.second.size() == 2) diffs[kv.second[1] - kv.second[0]]++;
    for (auto& kv : diffs) printf("block index difference %d: %d CUs\n", kv.first, kv.second);
    return 0;
}
