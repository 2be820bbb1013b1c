// Round-4 probe: does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let two kernels of ONE stream overlap on gfx950 (AQL packet without
// the barrier bit), and does the property survive hipStreamBeginCapture / hipGraphLaunch?
// Build: hipcc -O2 --offload-arch=gfx950 tools/ubench/anyorder.hip -o tools/ubench/bin/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long cycles, unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter();   // s_memtime: constant 100 MHz clock
    unsigned long long t = t0;
    while (t - t0 < cycles) { __builtin_amdgcn_s_sleep(10); t = __builtin_readcyclecounter(); }
    if (out && threadIdx.x == 0) out[blockIdx.x] = t - t0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long* out;
    CK(hipMalloc(&out, 1024));
    unsigned long long cyc = 200000;            // 2 ms at 100 MHz
    void* args[2] = {&cyc, &out};
    auto launch = [&](int flags) { return hipExtLaunchKernel((const void*)spin, dim3(4), dim3(64), args, 0, st, nullptr, nullptr, flags); };
    CK(launch(0)); CK(hipStreamSynchronize(st));
    for (int mode = 0; mode < 2; ++mode) {
        double t0 = now();
        for (int i = 0; i < 4; ++i) CK(launch(mode ? hipExtAnyOrderLaunch : 0));
        CK(hipStreamSynchronize(st));
        printf("plain stream, 4 x 2 ms kernels, flags=%d: %.2f ms\n", mode, (now() - t0) * 1e3);
    }
    // mixed: [ordered, any, ordered, any]: pairs should overlap -> ~4 ms
    {
        double t0 = now();
        for (int i = 0; i < 4; ++i) CK(launch((i & 1) ? hipExtAnyOrderLaunch : 0));
        CK(hipStreamSynchronize(st));
        printf("plain stream, ordered/any alternating: %.2f ms (pairs overlap -> ~4)\n", (now() - t0) * 1e3);
    }
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 4; ++i) CK(launch((mode && (i & 1)) ? hipExtAnyOrderLaunch : 0));
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        double t0 = now();
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        printf("captured graph, %s: %.2f ms\n", mode ? "ordered/any alternating" : "all ordered", (now() - t0) * 1e3);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
