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

// Diagnostic build only (-DAQG_TRACE, tools/trace_overlap.py; never shipped): every workgroup of the three per-simulation
// kernels logs {kernel id | where it ran, tag, blockIdx, start, end} (s_memrealtime, 100 MHz) into a caller-supplied buffer whose first
// word is the entry counter -- the only way to see which kernels of different game sets REALLY run side by side (rocprofv3's
// kernel trace serialises the dispatches it intercepts).
#ifdef AQG_TRACE
#ifndef AQG_TRACE_TU
#define AQG_TRACE_TU other
#endif
// (one copy per translation unit, under a per-file NAME: without relocatable device code every file is its own code object, and
//  the runtime registers device variables by name -- two statics of the same name end up sharing one registration)
#define AQG_CAT2(a, b) a##b
#define AQG_CAT(a, b) AQG_CAT2(a, b)
#define g_trace_buf AQG_CAT(g_trace_buf_, AQG_TRACE_TU)
#define g_trace_cap AQG_CAT(g_trace_cap_, AQG_TRACE_TU)
static __device__ unsigned long long* g_trace_buf = nullptr;
static __device__ unsigned int g_trace_cap = 0;
#define AQG_TRACE_BEGIN unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime(); unsigned int tr_hw, tr_xcc; \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(tr_hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(tr_xcc));
#define AQG_TRACE_END(kid, tag) { __syncthreads(); if (threadIdx.x == 0 && g_trace_buf) { const unsigned int i = atomicAdd(reinterpret_cast<unsigned int*>(g_trace_buf), 1u); \
    if (i < g_trace_cap) { unsigned long long* r = g_trace_buf + 1 + 4ull * i; r[0] = (unsigned long long)(kid) | ((unsigned long long)(tr_hw & 0xFFFFu) << 8) | ((unsigned long long)(tr_xcc & 0xFu) << 24);   /* bits 8..23: HW_ID (wave, SIMD, pipe, CU, SH, SE), 24..27: XCC */ r[1] = (unsigned long long)(tag); r[2] = tr_t0; r[3] = __builtin_amdgcn_s_memrealtime() | ((unsigned long long)blockIdx.x << 48); } } }
#define AQG_TRACE_SETTER(name) int name(void* buf, unsigned int cap) { unsigned long long* b = (unsigned long long*)buf; \
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &b, sizeof(b)) != hipSuccess) return -1; \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cap), &cap, sizeof(cap)) == hipSuccess ? 0 : -1; }
#else
#define AQG_TRACE_BEGIN
#define AQG_TRACE_END(kid, tag)
#define AQG_TRACE_SETTER(name) int name(void*, unsigned int) { return 0; }
#endif

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
