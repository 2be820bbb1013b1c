// Developer microbenchmark: issue cost (cycles per wave-instruction) of the vector instructions the trunk epilogue
// uses, one wave per SIMD (256 threads, 1 workgroup per CU) and two (512 threads).  s_memtime around 64 x 16
// independent copies.  Build: hipcc -O3 --offload-arch=gfx950 issue_cost.hip -o /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP4(x) x x x x

template <int KIND>
__global__ void k(unsigned long long* out, float* sink, float seed) {
    __shared__ float lds[4096];
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    f2 d0 = {a0, a1}, d1 = {a2, a3}, d2 = {a1, a0}, d3 = {a3, a0}, d4 = {a2, a1};
    f4 q0 = {a0, a1, a2, a3}, q1 = {a3, a2, a1, a0}, q2 = {a1, a1, a2, a2}, q3 = {a0, a0, a3, a3}, q4 = {a2, a0, a3, a1};
    f16v w0; for (int i = 0; i < 16; ++i) w0[i] = a0 + i;
    unsigned int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    const unsigned int addr = (threadIdx.x & 63) * 16;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a0) : "v"(a1));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d0) : "v"(d1));) }
        if (KIND == 2) { REP16(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(a1), "v"(a2));) }
        if (KIND == 3) { REP16(asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(a3) : "v"(p1));) }
        if (KIND == 4) { REP16(asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a3) : "v"(p1));) }
        if (KIND == 5) { REP16(asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a0) : "v"(a1));) }
        if (KIND == 6) { REP16(asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(d2) : "v"(d1));) }
        if (KIND == 7) { REP16(asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(d1) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (KIND == 8) { REP16(asm volatile("ds_write2_b64 %0, %1, %2 offset1:4" :: "v"(addr), "v"(d1), "v"(d2) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (KIND == 9) { REP16(asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(q0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (KIND == 10) { REP16(asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a0) : "v"(a1));) }
        if (KIND == 11) { REP16(asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(a3) : "v"(a1), "v"(a2));) }
        if (KIND == 12) { REP16(asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(q0) : "v"(q1));) }
        if (KIND == 13) { REP16(asm volatile("v_mfma_f32_16x16x32_f16 %0, %3, %3, %0\n v_fma_f32 %1, %2, %2, %1\n v_fma_f32 %2, %1, %1, %2" : "+v"(q0), "+v"(a0), "+v"(a1) : "v"(q1));) }
        if (KIND == 14) { REP16(asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(w0) : "v"(q1));) }
        if (KIND == 15) { REP16(asm volatile("s_nop 0");) }
        if (KIND == 20) { REP4(asm volatile("v_fma_f32 %0, %4, %4, %0\n v_fma_f32 %1, %4, %4, %1\n v_fma_f32 %2, %4, %4, %2\n v_fma_f32 %3, %4, %4, %3" : "+v"(q0[0]), "+v"(q0[1]), "+v"(q0[2]), "+v"(q0[3]) : "v"(a1));) }
        if (KIND == 21) { REP4(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %4\n v_cvt_pk_f16_f32 %2, %4, %4\n v_cvt_pk_f16_f32 %3, %5, %5" : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(a1), "v"(a2));) }
        if (KIND == 22) { REP4(asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4));) }
        if (KIND == 23) { REP4(asm volatile("v_cvt_f32_f16_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_e32 %1, %4\n v_cvt_f32_f16_sdwa %2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_e32 %3, %5" : "=v"(q0[0]), "=v"(q0[1]), "=v"(q0[2]), "=v"(q0[3]) : "v"(p1), "v"(p2));) }
        if (KIND == 24) { REP4(asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %4, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %4, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %4, %2\n v_mfma_f32_16x16x32_f16 %3, %4, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q4));) }
        if (KIND == 25) { REP4(asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %4, %0\n v_fma_f32 %5, %7, %7, %5\n v_fma_f32 %6, %7, %7, %6\n v_mfma_f32_16x16x32_f16 %1, %4, %4, %1\n v_fma_f32 %5, %7, %7, %5\n v_fma_f32 %6, %7, %7, %6\n v_mfma_f32_16x16x32_f16 %2, %4, %4, %2\n v_fma_f32 %5, %7, %7, %5\n v_fma_f32 %6, %7, %7, %6\n v_mfma_f32_16x16x32_f16 %3, %4, %4, %3\n v_fma_f32 %5, %7, %7, %5\n v_fma_f32 %6, %7, %7, %6" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q4), "v"(a0), "v"(a1), "v"(a2));) }
        if (KIND == 26) { REP4(asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %4, %0\n v_fma_f32 %5, %7, %7, %5\n v_fma_f32 %6, %7, %7, %6\n v_fma_f32 %8, %7, %7, %8\n v_mfma_f32_16x16x32_f16 %1, %4, %4, %1\n v_fma_f32 %5, %7, %7, %5\n v_fma_f32 %6, %7, %7, %6\n v_fma_f32 %8, %7, %7, %8\n v_mfma_f32_16x16x32_f16 %2, %4, %4, %2\n v_fma_f32 %5, %7, %7, %5\n v_fma_f32 %6, %7, %7, %6\n v_fma_f32 %8, %7, %7, %8\n v_mfma_f32_16x16x32_f16 %3, %4, %4, %3\n v_fma_f32 %5, %7, %7, %5\n v_fma_f32 %6, %7, %7, %6\n v_fma_f32 %8, %7, %7, %8" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q4), "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (KIND == 27) { REP4(asm volatile("s_and_b32 s20, s21, s22\n s_and_b32 s23, s21, s22\n s_and_b32 s24, s21, s22\n s_and_b32 s25, s21, s22" ::: "s20", "s23", "s24", "s25");) }
        if (KIND == 16) { REP16(asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(d0));) }
        if (KIND == 17) { REP16(asm volatile("v_mov_b32 %0, %1" : "=v"(a3) : "v"(a1));) }
        if (KIND == 18) { REP16(asm volatile("s_and_b32 s20, s20, s21" ::: "s20");) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + d0[0] + d1[1] + d2[0] + d3[0] + d4[1] + q2[0] + q3[1] + q4[2] + q0[0] + q0[1] + q0[2] + q0[3] + q1[1] + w0[0] + w0[15] + p0 + p1 + p2 + p3 + lds[threadIdx.x];
}

template <int KIND> void run(const char* name) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, sink, 1.0f);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, sink, 1.0f);
        hipDeviceSynchronize();
        unsigned long long h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-28s waves/CU %2d: %6.1f cycles per instruction (wave 0)\n", name, threads / 64, (double)h[0] / (64.0 * 16.0));
    }
    hipFree(out); hipFree(sink);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("start\n");
    run<0>("v_fma_f32 dependent"); run<20>("v_fma_f32 x4 independent"); run<21>("v_cvt_pk_f16_f32 x4 indep"); run<22>("v_pk_fma_f32 x4 indep"); run<23>("v_cvt_f32_f16 x4 indep");
    run<27>("s_and_b32 x4 indep"); run<12>("mfma16x16x32 dependent"); run<24>("mfma16x16x32 x4 indep"); run<25>("mfma16x16x32 + 2 v_fma"); run<26>("mfma16x16x32 + 3 v_fma"); run<14>("mfma32x32x16 dependent");
    return 0;
}
