// Round-4 probe: does a captured hipGraph with two parallel branches (stream A: a chain of kernels; stream B: kernels forked off A by
// events and joined back two steps later) run the branches CONCURRENTLY on replay, and what does a replay of many short nodes cost?
// Build: hipcc -O2 --offload-arch=gfx950 tools/ubench/graph_fork.hip -o tools/ubench/bin/graph_fork
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long cycles) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t evS, evH[3];
    CK(hipEventCreateWithFlags(&evS, hipEventDisableTiming));
    for (auto& e : evH) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int R = 400;
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: serial chain on A: S T H per round (15 + 10 + 6 us).  mode 1: S on A, (T, H) on B, S_{j+2} waits for H_j.  mode 2: as 1, empty kernels (0 us)
        const unsigned long long cs = mode == 2 ? 0 : 1500, ct = mode == 2 ? 0 : 1000, ch = mode == 2 ? 0 : 600;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
        for (int j = 0; j < R; ++j) {
            if (mode == 0) {
                hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, A, cs);
                hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, A, ct);
                hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, A, ch);
            } else {
                if (j >= 2) CK(hipStreamWaitEvent(A, evH[(j - 2) % 3], 0));
                hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, A, cs);
                CK(hipEventRecord(evS, A));
                CK(hipStreamWaitEvent(B, evS, 0));
                hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, B, ct);
                hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, B, ch);
                CK(hipEventRecord(evH[j % 3], B));
            }
        }
        if (mode != 0) { CK(hipStreamWaitEvent(A, evH[(R - 1) % 3], 0)); CK(hipStreamWaitEvent(A, evH[(R - 2) % 3], 0)); }
        CK(hipStreamEndCapture(A, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, A)); CK(hipStreamSynchronize(A));
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipGraphLaunch(ge, A)); CK(hipStreamSynchronize(A));
            best = best < now() - t0 ? best : now() - t0;
        }
        printf("mode %d: %d rounds: %.2f ms = %.2f us per round  (serial sum 31 us + gaps; forked ideal ~16 us)\n", mode, R, best * 1e3, best * 1e6 / R);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
