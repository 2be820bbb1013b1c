// aqg_common.hpp -- launch/error plumbing shared by the .hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

namespace aqg {

extern thread_local char g_err[512];

inline int fail(const char* what, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), "%s%s%s", what, detail[0] ? ": " : "", detail);
    return -1;
}

inline int check_launch(const char* kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(kernel, hipGetErrorString(e));
    return 0;
}

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// state loader: fmt 0 = state72 record, 1 = packed 24-byte QState
}  // namespace aqg

#include "quoridor_core.hpp"

namespace aqg {
__device__ __forceinline__ QState load_state(const void* base, int fmt, size_t b) {
    if (fmt == 0) return unpack72(reinterpret_cast<const uint8_t*>(base) + b * STATE72);
    const uint64_t* q = reinterpret_cast<const uint64_t*>(base) + b * 3;
    QState s;
    s.hw = q[0]; s.vw = q[1];
    uint64_t m = q[2];
    s.ppos = (uint8_t)(m & 0xff); s.pwl = (uint8_t)((m >> 8) & 0xff);
    s.epos = (uint8_t)((m >> 16) & 0xff); s.ewl = (uint8_t)((m >> 24) & 0xff);
    s.plies = (uint16_t)((m >> 32) & 0xffff); s.pad = 0;
    return s;
}
__device__ __forceinline__ void store_state(void* base, size_t b, const QState& s) {
    uint64_t* q = reinterpret_cast<uint64_t*>(base) + b * 3;
    q[0] = s.hw; q[1] = s.vw;
    q[2] = (uint64_t)s.ppos | ((uint64_t)s.pwl << 8) | ((uint64_t)s.epos << 16) | ((uint64_t)s.ewl << 24) |
           ((uint64_t)s.plies << 32);
}
}  // namespace aqg
