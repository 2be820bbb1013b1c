// gcn_forward.hip -- K1/K2: GraphPolicyValueNetwork.forward (pv_network_gnn.py:53-64) for gfx950, fp32.
//
// trunk kernel (boards): one 256-thread workgroup walks boards; per board the whole 3-layer GCN trunk runs
// out of ONE in-place LDS image H[81][132] f32 (HBM traffic: 72/24 B in, 512 B out per board):
//   setup   : wall masks -> per-node degree / sym-norm coefficients + the 6 node features (pv_network_cnn.py:88-114)
//   layer 1 : aggregate the 6-wide features over the <=5-point wall-cut stencil, then 6->128 on VALU
//   layer 2,3: dense 128x128 contraction on f32-input MFMA (v_mfma_f32_16x16x4_f32; rows 0..79 as five
//             16-row tiles, row 80 on VALU -> no padded MFMA work), accumulators staged in registers and
//             written back in place; then the normalised neighbour gather (= PyG's scatter-add on this
//             fixed-degree graph) + bias + ReLU, again register-staged in place
//   pool    : global_mean_pool fused into the layer-3 gather
// Each wave owns 32 output columns and keeps its slice of W2^T and W3^T in registers for the whole kernel
// (128 VGPRs), so weights cost no LDS/L2 traffic per board.  K is permuted (lane quarter q covers
// k in [kbase[q], kbase[q]+32)) so every A fragment is 8 contiguous ds_read_b128 and the padded row stride
// (132 floats) keeps each 16-lane ds_read_b128 group on 16 distinct 16-byte bank slots.
//
// heads kernel: policy MLP 128->64->209 (+Softmax) and value MLP 128->64->1 (+Tanh), 16 boards per workgroup.
//
// graph kernels: the same network on an arbitrary (x, CSR, graph_ptr) batch -- generic boundary path.
#include "aqg_common.hpp"
#include <vector>

namespace aqg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HID = 128;          // HIDDEN_DIM pv_network_gnn.py:18
constexpr int FPAD = 8;           // NUM_FEATURES (6) padded
constexpr int LD = 132;           // LDS row stride in floats (528 B: 33 x 16-B slots, odd -> conflict-free)
constexpr int APAD = 256;         // policy outputs padded

// packed weight offsets (floats) -- documented in include/aqgnn.h
struct PackedLayout {
    static constexpr size_t W1 = 0;                       // [HID][FPAD]
    static constexpr size_t B1 = W1 + HID * FPAD;         // [HID]
    static constexpr size_t W2T = B1 + HID;               // [HID k][HID n]
    static constexpr size_t B2 = W2T + HID * HID;
    static constexpr size_t W3T = B2 + HID;
    static constexpr size_t B3 = W3T + HID * HID;
    static constexpr size_t HW1T = B3 + HID;              // [HID k][HID unit]
    static constexpr size_t HB1 = HW1T + HID * HID;
    static constexpr size_t PW2T = HB1 + HID;             // [HID/2 k][APAD]
    static constexpr size_t PB2 = PW2T + (HID / 2) * APAD;
    static constexpr size_t VW2 = PB2 + APAD;             // [HID/2]
    static constexpr size_t VB2 = VW2 + HID / 2;          // [4]
    // MFMA B-fragment order of W2^T / W3^T: [wave 4][ntile 2][s4 8][lane 64][4]  (see load_wfrag)
    static constexpr size_t WF2 = VB2 + 4;
    static constexpr size_t WF3 = WF2 + HID * HID;
    // bf16 3-way split (hi, mid, lo) of W2^T / W3^T in 16x16x32 MFMA B-fragment order, stored as raw dwords:
    // [plane 3][wave 4][ntile 2][kblock 4][lane 64][4 dwords = 8 bf16]   (see load_bfrag)
    static constexpr size_t WB2 = WF3 + HID * HID;
    static constexpr size_t WB3 = WB2 + 3 * HID * HID / 2;
    // fp16 2-way split (hi, lo) of W2^T / W3^T, same fragment order: [plane 2][wave 4][ntile 2][kblock 4][lane 64][4 dwords]
    static constexpr size_t WH2 = WB3 + 3 * HID * HID / 2;
    static constexpr size_t WH3 = WH2 + 2 * HID * HID / 2;
    static constexpr size_t TOTAL = WH3 + 2 * HID * HID / 2;
};

size_t packed_floats() { return PackedLayout::TOTAL; }

// round-to-nearest-even f32 -> bf16 (host side of the split; the device uses v_cvt_pk_bf16_f32, also RNE)
static inline uint16_t host_bf16_rn(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    return (uint16_t)u;
}
static inline float host_bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline void host_split3(float x, uint16_t (&pl)[3]) {
    pl[0] = host_bf16_rn(x); float r = x - host_bf16_to_f32(pl[0]);
    pl[1] = host_bf16_rn(r); r = r - host_bf16_to_f32(pl[1]);
    pl[2] = host_bf16_rn(r);
}

// tensors (host fp32), state_dict order: gcn0.w[H,F] gcn0.b gcn1.w[H,H] gcn1.b gcn2.w gcn2.b
// pol0.w[H/2,H] pol0.b pol2.w[A,H/2] pol2.b val0.w[H/2,H] val0.b val2.w[1,H/2] val2.b
int pack_weights_host(int N, const float* const* t, float* out) {
    const int F = 6, A = N * N + 2 * (N - 1) * (N - 1);
    if (A > APAD) return fail("policy size exceeds APAD");
    memset(out, 0, sizeof(float) * PackedLayout::TOTAL);
    for (int n = 0; n < HID; ++n)
        for (int f = 0; f < F; ++f) out[PackedLayout::W1 + n * FPAD + f] = t[0][n * F + f];
    memcpy(out + PackedLayout::B1, t[1], sizeof(float) * HID);
    for (int n = 0; n < HID; ++n)
        for (int k = 0; k < HID; ++k) {
            out[PackedLayout::W2T + k * HID + n] = t[2][n * HID + k];
            out[PackedLayout::W3T + k * HID + n] = t[4][n * HID + k];
        }
    memcpy(out + PackedLayout::B2, t[3], sizeof(float) * HID);
    memcpy(out + PackedLayout::B3, t[5], sizeof(float) * HID);
    for (int w = 0; w < 4; ++w)
        for (int j = 0; j < 2; ++j)
            for (int s4 = 0; s4 < 8; ++s4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i) {
                        const int c = lane & 15, q = lane >> 4;
                        const int k = (q & 1) * 64 + (q >> 1) * 32 + 4 * s4 + i, n = 32 * w + 16 * j + c;
                        const size_t o = ((((size_t)w * 2 + j) * 8 + s4) * 64 + lane) * 4 + i;
                        out[PackedLayout::WF2 + o] = t[2][n * HID + k];
                        out[PackedLayout::WF3 + o] = t[4][n * HID + k];
                    }
    for (int L = 0; L < 2; ++L) {
        const float* W = t[L == 0 ? 2 : 4];                       // [n][k]
        uint32_t* dst = reinterpret_cast<uint32_t*>(out + (L == 0 ? PackedLayout::WB2 : PackedLayout::WB3));
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < 2; ++j)
                for (int kb = 0; kb < 4; ++kb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int d = 0; d < 4; ++d) {
                            const int c = lane & 15, q = lane >> 4, n = 32 * w + 16 * j + c;
                            uint16_t a[3], b[3];
                            host_split3(W[n * HID + 32 * kb + 8 * q + 2 * d], a);
                            host_split3(W[n * HID + 32 * kb + 8 * q + 2 * d + 1], b);
                            for (int pl = 0; pl < 3; ++pl) {
                                const size_t o = (((((size_t)pl * 4 + w) * 2 + j) * 4 + kb) * 64 + lane) * 4 + d;
                                dst[o] = (uint32_t)a[pl] | ((uint32_t)b[pl] << 16);
                            }
                        }
    }
    for (int L = 0; L < 2; ++L) {                                 // fp16 split planes
        const float* W = t[L == 0 ? 2 : 4];
        uint32_t* dst = reinterpret_cast<uint32_t*>(out + (L == 0 ? PackedLayout::WH2 : PackedLayout::WH3));
        auto split2 = [](float x, uint16_t (&pl)[2]) {
            const _Float16 h = (_Float16)x;                       // RNE
            const _Float16 l = (_Float16)(x - (float)h);
            memcpy(&pl[0], &h, 2); memcpy(&pl[1], &l, 2);
        };
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < 2; ++j)
                for (int kb = 0; kb < 4; ++kb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int d = 0; d < 4; ++d) {
                            const int c = lane & 15, q = lane >> 4, n = 32 * w + 16 * j + c;
                            uint16_t a[2], b[2];
                            split2(W[n * HID + 32 * kb + 8 * q + 2 * d], a);
                            split2(W[n * HID + 32 * kb + 8 * q + 2 * d + 1], b);
                            for (int pl = 0; pl < 2; ++pl) {
                                const size_t o = (((((size_t)pl * 4 + w) * 2 + j) * 4 + kb) * 64 + lane) * 4 + d;
                                dst[o] = (uint32_t)a[pl] | ((uint32_t)b[pl] << 16);
                            }
                        }
    }
    for (int u = 0; u < HID / 2; ++u)
        for (int k = 0; k < HID; ++k) {
            out[PackedLayout::HW1T + k * HID + u] = t[6][u * HID + k];
            out[PackedLayout::HW1T + k * HID + HID / 2 + u] = t[10][u * HID + k];
        }
    memcpy(out + PackedLayout::HB1, t[7], sizeof(float) * (HID / 2));
    memcpy(out + PackedLayout::HB1 + HID / 2, t[11], sizeof(float) * (HID / 2));
    for (int a = 0; a < A; ++a)
        for (int k = 0; k < HID / 2; ++k) out[PackedLayout::PW2T + k * APAD + a] = t[8][a * (HID / 2) + k];
    memcpy(out + PackedLayout::PB2, t[9], sizeof(float) * A);
    memcpy(out + PackedLayout::VW2, t[12], sizeof(float) * (HID / 2));
    out[PackedLayout::VB2] = t[13][0];
    return 0;
}

// deg^-1/2 for deg 1..5 (self loop + <=4 open neighbours), correctly rounded f32
__device__ __forceinline__ float dinv_of(int deg) {
    switch (deg) {
        case 1: return 1.0f;
        case 2: return 0.70710678118654752f;
        case 3: return 0.57735026918962576f;
        case 4: return 0.5f;
        default: return 0.44721359549995794f;
    }
}

__device__ __forceinline__ float dinv_of_bits(int bits) { return dinv_of(1 + __popc(bits)); }

// Diagnostic build only (-DAQG_STAMP, never shipped): per-phase s_memtime sums of workgroup 0 / wave 0 are
// written behind the pooled rows (pooled + B*128, as 16 x u64).  In the real kernel no stamp executes.
#ifdef AQG_STAMP
#define AQG_STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int st_n = 0;
#define AQG_STAMP_AT(i) { unsigned long long st_now = __builtin_readcyclecounter(); st_sum[i] += st_now - st_prev; st_prev = st_now; }
#else
#define AQG_STAMP_DECL
#define AQG_STAMP_AT(i)
#endif

struct alignas(16) TrunkSmem {
    alignas(16) float H[81 * LD];      // in-place activation image, [node][channel]
    alignas(16) float X0[81 * FPAD];   // node features
    alignas(16) float AX[81 * FPAD];   // A_hat * X0 (layer-1 input after the 6-wide gather)
    alignas(16) float coef[96][8];     // per node: self, U, D, L, R gather coefficients (0 when the edge is cut), 3 pad
    int obits[96];                     // per node: open-edge bits (U,D,L,R) -- setup step 1 -> step 2
};

// Stripe row schedule shared by layer 1 and the stripe gathers: lane -> 4 columns (cg = lane & 7) of one row
// per iteration (rs = lane >> 3).  Iterations 0..7 take rows it + 8*rs (0..63): with the 132-float row stride
// the 16-lane ds_read_b128 groups then hit 16 distinct 16-byte bank slots.  Iterations 8..10 take rows
// 64 + 8*(it-8) + rs (64..80; the last one only row 80).
__device__ __forceinline__ int stripe_row(int it, int rs) { return it < 8 ? it + 8 * rs : 64 + 8 * (it - 8) + rs; }
constexpr int STRIPE_ITERS = 11;

// B fragments of W^T for this wave from the fragment-ordered copy: Wf[j][s] = W^T[kb + s][32*wave + 16*j + c].
// 16 fully coalesced dwordx4 loads (1 KiB per wave-instruction) off one scalar base + one lane offset.
__device__ __forceinline__ void load_wfrag(float (&Wf)[2][32], const float* __restrict__ WF, int wave, int lane) {
    const float* base = WF + (size_t)__builtin_amdgcn_readfirstlane(wave) * (2 * 8 * 256) + lane * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (j * 8 + s4) * 256);
            Wf[j][4 * s4 + 0] = v[0]; Wf[j][4 * s4 + 1] = v[1]; Wf[j][4 * s4 + 2] = v[2]; Wf[j][4 * s4 + 3] = v[3];
        }
}

// ---- MFMA phase of one GCN layer: XW[0..79][stripe] = H[0..79][:] x W[:, stripe]  (PyG order: linear first).
// Wave `wave` owns the 32-column stripe [32*wave, 32*wave+32) for ALL rows; its A operand is the whole image,
// read as plain ds_read_b128 (no VALU in the MFMA stream).  Node 80 is done on VALU by the same lanes.
// Half tiles (16 k-steps = 4 x ds_read_b128) are double-buffered; sched_barrier pins the read/MFMA order.
__device__ __forceinline__ void stripe_matmul(const float* __restrict__ H, const float (&Wf)[2][32], int lane,
                                              f32x4 (&acc)[5][2], float& r0, float& r1) {
    const int c = lane & 15, q = lane >> 4;
    const int kb = (q & 1) * 64 + (q >> 1) * 32;  // kbase = {0, 64, 32, 96}
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float* arow = H + c * LD + kb;
    f32x4 cur[4], nxt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[j] = *reinterpret_cast<const f32x4*>(arow + 4 * j);
#pragma unroll
    for (int ht = 0; ht < 10; ++ht) {          // half tile ht: rows 16*(ht/2).., k-steps 16*(ht&1)..
        const int m = ht >> 1, h = ht & 1;
        const float* nsrc = (ht < 9) ? arow + 16 * ((ht + 1) >> 1) * LD + 16 * ((ht + 1) & 1) : H + 80 * LD + kb;
#pragma unroll
        for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const f32x4*>(nsrc + 4 * j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float av = cur[s >> 2][s & 3];
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Wf[0][16 * h + s], acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Wf[1][16 * h + s], acc[m][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
    // node 80: each lane covers its quarter of K for its two columns, quarters combined by xor-shuffles
    r0 = 0.f; r1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const f32x4*>(H + 80 * LD + kb + 16 + 4 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r0 = fmaf(cur[j][i], Wf[0][4 * j + i], r0);
            r1 = fmaf(cur[j][i], Wf[1][4 * j + i], r1);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r0 = fmaf(nxt[j][i], Wf[0][16 + 4 * j + i], r0);
            r1 = fmaf(nxt[j][i], Wf[1][16 + 4 * j + i], r1);
        }
    r0 += __shfl_xor(r0, 16); r0 += __shfl_xor(r0, 32);
    r1 += __shfl_xor(r1, 16); r1 += __shfl_xor(r1, 32);
}

// ---- stripe epilogue: the wave parks its XW stripe in its own columns of H (the image is dead after the
// barrier), then gathers it back with the normalised neighbour coefficients (= PyG's scatter-add on this
// fixed-degree graph), + bias, ReLU.  Only this wave touches these columns, so ordering is wave-local.
// LAST: mean-pool instead of writing back.
template <bool LAST>
__device__ __forceinline__ void stripe_gather(TrunkSmem& sm, const f32x4 (&acc)[5][2], float r0, float r1,
                                              const float* __restrict__ bias_g, int wave, int lane,
                                              float* __restrict__ pooled_out) {
    {
        const int c = lane & 15, q = lane >> 4;
        float* w = sm.H + (4 * q) * LD + 32 * wave + c;
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[(16 * m + i) * LD] = acc[m][0][i];
                w[(16 * m + i) * LD + 16] = acc[m][1][i];
            }
        if (q == 0) {
            sm.H[80 * LD + 32 * wave + c] = r0;
            sm.H[80 * LD + 32 * wave + 16 + c] = r1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int cg = lane & 7, rs = lane >> 3;
    const int colb = 32 * wave + 4 * cg;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_g + colb);
    f32x4 out[STRIPE_ITERS];
    // Row r's neighbours sit at fixed row offsets (-9, +9, -1, +1); off-board ones are clamped to r itself (their
    // coefficient is 0 -- but the clamp is REQUIRED: 0 x NaN garbage is NaN, which the ReLU would silently turn into 0).  With the schedule of stripe_row() every clamp is known at compile time except
    // "up" in iterations 0..7 (only the rs == 0 lanes have r < 9), which is one per-lane offset.
    const float* p1 = sm.H + colb + rs * 8 * LD;          // iteration it < 8: row it + 8*rs
    const float* p2 = sm.H + colb + (64 + rs) * LD;       // iteration 8, 9: rows 64+rs, 72+rs
    const float* k1 = &sm.coef[8 * rs][0];
    const float* k2 = &sm.coef[64 + rs][0];
    const int offU1 = rs == 0 ? 0 : -9 * LD;
#pragma unroll
    for (int it = 0; it < STRIPE_ITERS; ++it) {
        const float *ps, *pu, *pd, *pl, *pr, *pk;
        if (it < 8) {
            ps = p1 + it * LD; pk = k1 + it * 8;
            pu = (it == 0) ? (rs <= 1 ? ps : ps - 9 * LD) : ps + offU1;   // rows it + 8*rs < 9: rs == 0, and row 8 (it 0, rs 1)
            pd = ps + 9 * LD; pr = ps + LD;
            pl = (it == 0) ? (rs == 0 ? ps : ps - LD) : ps - LD;
        } else if (it == 8) {
            ps = p2; pk = k2; pu = ps - 9 * LD; pd = ps + 9 * LD; pl = ps - LD; pr = ps + LD;
        } else if (it == 9) {
            ps = p2 + 8 * LD; pk = k2 + 64; pu = ps - 9 * LD; pd = ps; pl = ps - LD; pr = ps + LD;
        } else {                                             // row 80 (every lane computes it; only rs == 0 is used)
            ps = sm.H + colb + 80 * LD; pk = &sm.coef[80][0]; pu = ps - 9 * LD; pd = ps; pl = ps - LD; pr = ps;
        }
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(pk);
        const float kr = pk[4];
        const f32x4 hs = *reinterpret_cast<const f32x4*>(ps);
        const f32x4 hu = *reinterpret_cast<const f32x4*>(pu);
        const f32x4 hd = *reinterpret_cast<const f32x4*>(pd);
        const f32x4 hl = *reinterpret_cast<const f32x4*>(pl);
        const f32x4 hr = *reinterpret_cast<const f32x4*>(pr);
        f32x4 v = bias + k4[0] * hs + k4[1] * hu + k4[2] * hd + k4[3] * hl + kr * hr;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        out[it] = v;
        if (it & 1) __builtin_amdgcn_sched_barrier(0);   // two iterations' reads (14) in flight, not all 77
    }
    if (!LAST) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < STRIPE_ITERS; ++it) {
            const int r = stripe_row(it, rs);
            if (r < 81) *reinterpret_cast<f32x4*>(sm.H + r * LD + colb) = out[it];
        }
    } else {
        f32x4 sum = out[0];
#pragma unroll
        for (int it = 1; it < STRIPE_ITERS - 1; ++it) sum += out[it];
        if (rs == 0) sum += out[STRIPE_ITERS - 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = sum[e];
            x += __shfl_xor(x, 8); x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
            sum[e] = x * (1.0f / 81.0f);
        }
        if (rs == 0) *reinterpret_cast<f32x4*>(pooled_out + colb) = sum;
    }
}

// RESIDENT = true : one workgroup per CU (512-VGPR budget), both layers' fragments live in registers for the
//                   whole kernel -> zero per-board weight traffic, but no cross-workgroup phase overlap.
// RESIDENT = false: two workgroups per CU (256 VGPRs); each layer's fragments are re-fetched per board from
//                   L2 (128 KB per board per workgroup), issued one phase ahead of use.
template <bool RESIDENT, int WGS_PER_CU>
__global__ __launch_bounds__(256, WGS_PER_CU) void gcn_trunk_boards_kernel(const void* __restrict__ states, int fmt,
                                                                                   int B, const float* __restrict__ pk,
                                                                                   float* __restrict__ pooled,
                                                                                   const uint8_t* __restrict__ active) {
    constexpr int N = 9, V = 81, S = 8;
    __shared__ TrunkSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    float W2f[2][32], W3f[2][32];
    if (RESIDENT) {
        load_wfrag(W2f, pk + PackedLayout::WF2, wave, lane);
        load_wfrag(W3f, pk + PackedLayout::WF3, wave, lane);
    }
    // Raw state prefetch: the record's dwords are loaded one board ahead and only unpacked at setup time, so
    // the global-load latency hides under the previous board's layers (18 dwords for state72, 6 for QState).
    const int ndw = fmt == 0 ? 18 : 6;
    auto fetch_raw = [&](uint32_t (&raw)[18], int bb) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(states) + (size_t)bb * ndw;
#pragma unroll
        for (int i = 0; i < 18; ++i) raw[i] = (i < ndw) ? src[i] : 0u;
    };
    auto unpack_raw = [&](const uint32_t (&raw)[18]) -> QState {
        QState s;
        if (fmt == 0) {
            uint64_t h = 0, v = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                h |= (uint64_t)gather_bit0_x4(raw[1 + i]) << (4 * i);
                v |= (uint64_t)gather_bit0_x4(raw[1 + i] >> 1) << (4 * i);
            }
            s.hw = h; s.vw = v;
            s.ppos = (uint8_t)(raw[0] & 0xff); s.pwl = (uint8_t)((raw[0] >> 8) & 0xff);
            s.epos = (uint8_t)((raw[0] >> 16) & 0xff); s.ewl = (uint8_t)(raw[0] >> 24);
            s.plies = (uint16_t)(raw[17] & 0xffff);
        } else {
            s.hw = (uint64_t)raw[0] | ((uint64_t)raw[1] << 32);
            s.vw = (uint64_t)raw[2] | ((uint64_t)raw[3] << 32);
            s.ppos = (uint8_t)(raw[4] & 0xff); s.pwl = (uint8_t)((raw[4] >> 8) & 0xff);
            s.epos = (uint8_t)((raw[4] >> 16) & 0xff); s.ewl = (uint8_t)(raw[4] >> 24);
            s.plies = (uint16_t)(raw[5] & 0xffff);
        }
        s.pad = 0;
        return s;
    };
    int b = blockIdx.x;
    while (b < B && active && !active[b]) b += gridDim.x;
    uint32_t raw[18];
    if (b < B && tid < V) fetch_raw(raw, b);

    AQG_STAMP_DECL
    while (b < B) {
        AQG_STAMP_AT(7)
        // ---- setup step 1: node features + this tile's open-edge bits
        if (tid < V) {
            const QState s = unpack_raw(raw);
            const int t = tid, x = t / N, y = t % N;
            sm.obits[t] = tile_open_bits<N>(s.hw, s.vw, t);
            const bool slot_ok = (x < S) && (y < S);
            const int slot = x * S + y;
            f32x4 xa, xb;
            xa[0] = (t == s.ppos) ? 1.f : 0.f;
            xa[1] = (float)s.pwl;
            xa[2] = (t == s.epos) ? 1.f : 0.f;      // enemy's own frame (pv_network_cnn.py:101)
            xa[3] = (float)s.ewl;
            xb[0] = (slot_ok && ((s.hw >> slot) & 1)) ? 1.f : 0.f;
            xb[1] = (slot_ok && ((s.vw >> slot) & 1)) ? 1.f : 0.f;
            xb[2] = 0.f; xb[3] = 0.f;
            *reinterpret_cast<f32x4*>(sm.X0 + t * FPAD) = xa;
            *reinterpret_cast<f32x4*>(sm.X0 + t * FPAD + 4) = xb;
        }
        // prefetch the next board's raw record (lands under this board's layers)
        int bn = b + gridDim.x;
        while (bn < B && active && !active[bn]) bn += gridDim.x;
        if (bn < B && tid < V) fetch_raw(raw, bn);
        if (!RESIDENT) load_wfrag(W2f, pk + PackedLayout::WF2, wave, lane);   // lands under layer 1
        __syncthreads();
        AQG_STAMP_AT(0)
        // ---- setup step 2 + layer 1a: sym-norm coefficients from the neighbours' degrees, AX = A_hat * X0
        if (tid < V) {
            const int t = tid;
            const int ob = sm.obits[t];
            const int tu = t >= 9 ? t - 9 : t, td = t < 72 ? t + 9 : t, tl = t > 0 ? t - 1 : t, tr = t < 80 ? t + 1 : t;
            const float di = dinv_of_bits(ob);
            f32x4 k4;
            k4[0] = di * di;
            k4[1] = (ob & 1) ? di * dinv_of_bits(sm.obits[tu]) : 0.f;
            k4[2] = (ob & 2) ? di * dinv_of_bits(sm.obits[td]) : 0.f;
            k4[3] = (ob & 4) ? di * dinv_of_bits(sm.obits[tl]) : 0.f;
            const float kr = (ob & 8) ? di * dinv_of_bits(sm.obits[tr]) : 0.f;
            *reinterpret_cast<f32x4*>(&sm.coef[t][0]) = k4;
            sm.coef[t][4] = kr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = k4[0] * *reinterpret_cast<const f32x4*>(sm.X0 + t * FPAD + 4 * h) +
                                k4[1] * *reinterpret_cast<const f32x4*>(sm.X0 + tu * FPAD + 4 * h) +
                                k4[2] * *reinterpret_cast<const f32x4*>(sm.X0 + td * FPAD + 4 * h) +
                                k4[3] * *reinterpret_cast<const f32x4*>(sm.X0 + tl * FPAD + 4 * h) +
                                kr * *reinterpret_cast<const f32x4*>(sm.X0 + tr * FPAD + 4 * h);
                *reinterpret_cast<f32x4*>(sm.AX + t * FPAD + 4 * h) = v;
            }
        }
        __syncthreads();
        AQG_STAMP_AT(1)
        // ---- layer 1b: H1[r][cols] = ReLU(b1 + AX[r] . W1[cols]) on this wave's stripe (4 columns x 1 row per lane-iteration)
        {
            const int cg = lane & 7, rs = lane >> 3;
            const int colb = 32 * wave + 4 * cg;
            float w1[4][6];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(pk + PackedLayout::W1 + (colb + e) * FPAD);
                const float2 hi = *reinterpret_cast<const float2*>(pk + PackedLayout::W1 + (colb + e) * FPAD + 4);
                w1[e][0] = lo[0]; w1[e][1] = lo[1]; w1[e][2] = lo[2]; w1[e][3] = lo[3]; w1[e][4] = hi.x; w1[e][5] = hi.y;
            }
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(pk + PackedLayout::B1 + colb);
#pragma unroll
            for (int it = 0; it < STRIPE_ITERS; ++it) {
                const int r = stripe_row(it, rs);
                if (r < V) {
                    const f32x4 xa = *reinterpret_cast<const f32x4*>(sm.AX + r * FPAD);
                    const float2 xb = *reinterpret_cast<const float2*>(sm.AX + r * FPAD + 4);
                    f32x4 v = b1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = v[e];
                        a = fmaf(xa[0], w1[e][0], a); a = fmaf(xa[1], w1[e][1], a); a = fmaf(xa[2], w1[e][2], a);
                        a = fmaf(xa[3], w1[e][3], a); a = fmaf(xb.x, w1[e][4], a); a = fmaf(xb.y, w1[e][5], a);
                        v[e] = fmaxf(a, 0.f);
                    }
                    *reinterpret_cast<f32x4*>(sm.H + r * LD + colb) = v;
                }
                if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        AQG_STAMP_AT(2)
        // ---- layer 2
        f32x4 acc[5][2];
        float r0, r1;
        stripe_matmul(sm.H, W2f, lane, acc, r0, r1);
        if (!RESIDENT) load_wfrag(W3f, pk + PackedLayout::WF3, wave, lane);   // lands under the layer-2 epilogue
        __syncthreads();                       // every wave has finished reading the image
        AQG_STAMP_AT(3)
        stripe_gather<false>(sm, acc, r0, r1, pk + PackedLayout::B2, wave, lane, nullptr);
        __syncthreads();
        AQG_STAMP_AT(4)
        // ---- layer 3 + mean pool
        stripe_matmul(sm.H, W3f, lane, acc, r0, r1);
        __syncthreads();
        AQG_STAMP_AT(5)
        stripe_gather<true>(sm, acc, r0, r1, pk + PackedLayout::B3, wave, lane, pooled + (size_t)b * HID);
        __syncthreads();                       // coef / image are rewritten by the next board's setup
        AQG_STAMP_AT(6)
#ifdef AQG_STAMP
        ++st_n;
#endif
        b = bn;
    }
#ifdef AQG_STAMP
    if (blockIdx.x == 0 && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(pooled + (size_t)B * HID);
        for (int i = 0; i < 8; ++i) o[i] = st_sum[i];
        o[8] = (unsigned long long)st_n;
    }
#endif
}

// =============================================================================================
// split-precision trunk: the same network with the two 128x128 contractions on the 16-bit matrix pipe.
// Every f32 operand x is split into 16-bit planes (each the RNE of the running remainder) and the product is
// rebuilt from partial products accumulated in f32 by v_mfma_f32_16x16x32_{f16,bf16} (8 passes per 16x16x32 tile =
// 16x the f32-input MFMA rate):
//   fp16 planes, x = hi + lo (11 + 11 mantissa bits):  a*b ~= hi*hi + hi*lo + lo*hi      [default, "f16x3"]
//       the dropped lo*lo term is ~2^-22 |ab|: fp32-equivalent for this network (logits within 1.1e-7 of the exact
//       f32 path offline, i.e. the same distance the exact f32 path has from the fp64 oracle).  Activations here are
//       O(1) after ReLU/normalised aggregation and |W| < 1, far inside fp16 range; lo underflows to fp16 subnormals
//       only below |x| ~ 2^-14 * 2^-11, where the absolute error (< 2^-25) is irrelevant for O(1) sums.
//   bf16 planes, x = hi + lo' (8 + 8 bits):            a*b ~= hi*hi + hi*lo' + lo'*hi    ["bf16x3", ~2^-16 |ab|]
//   bf16 planes, x = hi + mid + lo: six terms (NT = 6), fp32-equivalent, three planes -- compiled only in
//       diagnostic builds, see the note at g_trunk_variant.
// All variants stay inside the stated 1e-5 / 1e-4 tolerance against the fp64 oracle (tests/test_gpu_parity.py).
// The image lives in LDS as 16-bit planes [plane][81][136]; the split is done ONCE per element in the stripe
// epilogue (non-redundant), the MFMA phase reads ready-made fragments with ds_read_b128.
// =============================================================================================
// Diagnostic builds only (never shipped; DESIGN.md 3 "LDS above 128 KB"): -DAQG_PAD_X3_FRONT pushes the two-plane
// image up by one plane so the second-resident workgroup's plane 1 lies above 128 KB of the CU's LDS (reproduces the
// cold-launch stale reads); -DAQG_DIAG_SITES=<mask> adds a fence + double barrier at the marked hand-off sites.
#ifdef AQG_DIAG_SITES
#define AQG_DIAG_FENCE_AT(bit) do { if ((AQG_DIAG_SITES) & (bit)) { __threadfence_block(); __syncthreads(); __builtin_amdgcn_s_waitcnt(0); __syncthreads(); } } while (0)
#else
#define AQG_DIAG_FENCE_AT(bit) do {} while (0)
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int PROW = 272;                    // plane row stride in bytes: 128 bf16 + 16 B pad (17 slots of 16 B: odd)
constexpr int PPLANE = 81 * PROW + 48;       // plane stride (22,080 B): == 4 (mod 16) slots -> conflict-free stripe reads

// Row permutation for the small per-row tables (coef, AX): rows handled by one instruction are r = it + 8*rs, i.e.
// 8 apart; with 32-byte rows that is a 256-byte stride = every lane group on the same banks (4-way conflicts).
// Class-major order puts the rows of one iteration next to each other instead.
__device__ __forceinline__ int prow(int r) { return (r & 7) * 11 + (r >> 3); }   // 0..87 for r in 0..80

template <int NPL>
struct alignas(16) TrunkSmemB {
#ifdef AQG_PAD_X3_FRONT
    alignas(16) unsigned char diag_front[NPL == 2 ? 22080 : 16];   // diagnostic: push the x3 planes up so P[1] of the
                                                                    // second-resident workgroup straddles 128 KB
#endif
    alignas(16) unsigned char P[NPL][PPLANE];   // bf16 planes of the activation image; a wave's stripe bytes of
                                                // planes 0/1 double as its parked f32 XW stripe
    alignas(16) float X0[81 * FPAD];
    alignas(16) float AX[88 * FPAD];            // indexed by prow(r)
    alignas(16) float coef[88][8];              // indexed by prow(r)
    int obits[96];
    unsigned int raw[20];                       // next board's record, staged by <= 18 lanes (1 VGPR instead of 18)
};

__device__ __forceinline__ unsigned int pack_bf16x2(__bf16 a, __bf16 b) {
    return (unsigned int)__builtin_bit_cast(unsigned short, a) | ((unsigned int)__builtin_bit_cast(unsigned short, b) << 16);
}

// split 4 f32 values into NPL bf16 planes and store them as 8 bytes per plane at byte offset `off` of each plane.
// Written on packed pairs so it compiles to v_cvt_pk_bf16_f32 / v_pk_add_f32: ~18 vector instructions per 4 values.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cvt_pk_bf16(float a, float b) {          // RNE, low half = a
    return __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ f32x4 bf16_pairs_to_f32(unsigned int p01, unsigned int p23) {
    return (f32x4){__builtin_bit_cast(float, p01 << 16), __builtin_bit_cast(float, p01 & 0xFFFF0000u),
                   __builtin_bit_cast(float, p23 << 16), __builtin_bit_cast(float, p23 & 0xFFFF0000u)};
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int cvt_pk_f16(float a, float b) {            // RNE, low half = a
    return __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){a, b}, f16x2));
}
__device__ __forceinline__ f32x4 f16_pairs_to_f32(unsigned int p01, unsigned int p23) {
    const f32x2 a = __builtin_convertvector(__builtin_bit_cast(f16x2, p01), f32x2);
    const f32x2 b = __builtin_convertvector(__builtin_bit_cast(f16x2, p23), f32x2);
    return (f32x4){a[0], a[1], b[0], b[1]};
}
template <int NPL, bool F16, typename SM>
__device__ __forceinline__ void store_split4(SM& sm, int off, const f32x4 v) {
    if (F16) {     // fp16 planes: hi (11 bits) + lo (next 11 bits) = 22 mantissa bits in two planes
        const unsigned int h01 = cvt_pk_f16(v[0], v[1]), h23 = cvt_pk_f16(v[2], v[3]);
        const f32x4 r1 = v - f16_pairs_to_f32(h01, h23);
        *reinterpret_cast<u32x2*>(&sm.P[0][off]) = (u32x2){h01, h23};
        *reinterpret_cast<u32x2*>(&sm.P[1][off]) = (u32x2){cvt_pk_f16(r1[0], r1[1]), cvt_pk_f16(r1[2], r1[3])};
        return;
    }
    const unsigned int h01 = cvt_pk_bf16(v[0], v[1]), h23 = cvt_pk_bf16(v[2], v[3]);
    const f32x4 r1 = v - bf16_pairs_to_f32(h01, h23);
    const unsigned int m01 = cvt_pk_bf16(r1[0], r1[1]), m23 = cvt_pk_bf16(r1[2], r1[3]);
    *reinterpret_cast<u32x2*>(&sm.P[0][off]) = (u32x2){h01, h23};
    *reinterpret_cast<u32x2*>(&sm.P[1][off]) = (u32x2){m01, m23};
    if (NPL == 3) {
        const f32x4 r2 = r1 - bf16_pairs_to_f32(m01, m23);
        *reinterpret_cast<u32x2*>(&sm.P[2][off]) = (u32x2){cvt_pk_bf16(r2[0], r2[1]), cvt_pk_bf16(r2[2], r2[3])};
    }
}

// B fragments: Bf[pl][j][kb] = 8 bf16 of plane pl: W[k = 32*kb + 8*q + 0..7][n = 32*wave + 16*j + c]
template <int NPL>
__device__ __forceinline__ void load_bfrag(u32x4 (&Bf)[NPL][2][4], const float* __restrict__ WB, int wave, int lane) {
    const u32x4* base = reinterpret_cast<const u32x4*>(WB) + (size_t)__builtin_amdgcn_readfirstlane(wave) * (2 * 4 * 64) + lane;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) Bf[pl][j][kb] = base[(size_t)pl * (4 * 2 * 4 * 64) + (j * 4 + kb) * 64];
}

// MFMA phase: six 16-row tiles (rows >= 81 of the last tile are clamped to row 80 and discarded) x four 32-deep
// k blocks; per (tile, block) NPL ds_read_b128 feed 2*NT MFMAs, smallest terms first.
template <bool F16>
__device__ __forceinline__ f32x4 mfma_split(const u32x4 a, const u32x4 b, const f32x4 c) {
    if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int NPL, bool F16, typename SM>
__device__ __forceinline__ void stripe_matmul_bf16(const SM& sm, const u32x4 (&Bf)[NPL][2][4], int lane, f32x4 (&acc)[6][2]) {
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    u32x4 cur[NPL], nxt[NPL];
    const int row5 = (80 + c) > 80 ? 80 : 80 + c;           // tile 5: rows 80 + c clamped to 80
    auto frag_off = [&](int step) -> int {                  // step = m*4 + kb
        const int m = step >> 2, kb = step & 3;
        const int row = (m < 5) ? 16 * m + c : row5;
        return row * PROW + 64 * kb + 16 * q;
    };
    {
        const int o = frag_off(0);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) cur[pl] = *reinterpret_cast<const u32x4*>(&sm.P[pl][o]);
    }
#pragma unroll
    for (int step = 0; step < 24; ++step) {
        const int m = step >> 2, kb = step & 3;
        if (step < 23) {
            const int o = frag_off(step + 1);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) nxt[pl] = *reinterpret_cast<const u32x4*>(&sm.P[pl][o]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 a = acc[m][j];
            if (NPL == 3) {
                a = mfma_split<F16>(cur[2], Bf[0][j][kb], a);
                a = mfma_split<F16>(cur[0], Bf[NPL - 1][j][kb], a);
                a = mfma_split<F16>(cur[1], Bf[1][j][kb], a);
            }
            a = mfma_split<F16>(cur[1], Bf[0][j][kb], a);      // smallest terms first
            a = mfma_split<F16>(cur[0], Bf[1][j][kb], a);
            a = mfma_split<F16>(cur[0], Bf[0][j][kb], a);
            acc[m][j] = a;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (step < 23) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) cur[pl] = nxt[pl];
        }
    }
}

// stripe epilogue on the bf16 image: park XW (f32) in this wave's bytes of planes 0/1, gather + bias + ReLU,
// then write the result back as split planes (or mean-pool when LAST).
template <int NPL, bool F16, bool LAST, typename SM>
__device__ __forceinline__ void stripe_gather_bf16(SM& sm, const f32x4 (&acc)[6][2], const f32x4 bias,
                                                   int wave, int lane, float* __restrict__ pooled_out) {
    {
        const int c = lane & 15, q = lane >> 4;
        const int o = (4 * q) * PROW + 64 * wave + 4 * c;
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float*>(&sm.P[0][o + (16 * m + i) * PROW]) = acc[m][0][i];
                *reinterpret_cast<float*>(&sm.P[1][o + (16 * m + i) * PROW]) = acc[m][1][i];
            }
        if (q == 0) {
            *reinterpret_cast<float*>(&sm.P[0][80 * PROW + 64 * wave + 4 * c]) = acc[5][0][0];
            *reinterpret_cast<float*>(&sm.P[1][80 * PROW + 64 * wave + 4 * c]) = acc[5][1][0];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int cg = lane & 7, rs = lane >> 3;
    const int colb = 32 * wave + 4 * cg;
    f32x4 out[STRIPE_ITERS];
    const unsigned char* pbase = &sm.P[cg >> 2][0] + 64 * wave + 16 * (cg & 3);   // parked f32 x4 of columns colb..colb+3
    const unsigned char* p1 = pbase + rs * 8 * PROW;
    const unsigned char* p2 = pbase + (64 + rs) * PROW;
    const float* k1 = &sm.coef[rs][0];                  // iteration it < 8: prow(it + 8*rs) = it*11 + rs
    const float* k2 = &sm.coef[rs * 11 + 8][0];         // iteration 8: prow(64 + rs) = rs*11 + 8; iteration 9: +1
    const int offU1 = rs == 0 ? 0 : -9 * PROW;
#pragma unroll
    for (int it = 0; it < STRIPE_ITERS; ++it) {
        const unsigned char *ps, *pu, *pd, *pl, *pr;
        const float* pk;
        if (it < 8) {
            ps = p1 + it * PROW; pk = k1 + it * 11 * 8;
            pu = (it == 0) ? (rs <= 1 ? ps : ps - 9 * PROW) : ps + offU1;   // rows it + 8*rs < 9: rs == 0, and row 8 (it 0, rs 1)
            pd = ps + 9 * PROW; pr = ps + PROW;
            pl = (it == 0) ? (rs == 0 ? ps : ps - PROW) : ps - PROW;
        } else if (it == 8) {
            ps = p2; pk = k2; pu = ps - 9 * PROW; pd = ps + 9 * PROW; pl = ps - PROW; pr = ps + PROW;
        } else if (it == 9) {
            ps = p2 + 8 * PROW; pk = k2 + 8; pu = ps - 9 * PROW; pd = ps; pl = ps - PROW; pr = ps + PROW;
        } else {
            ps = pbase + 80 * PROW; pk = &sm.coef[prow(80)][0]; pu = ps - 9 * PROW; pd = ps; pl = ps - PROW; pr = ps;
        }
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(pk);
        const float kr = pk[4];
        const f32x4 hs = *reinterpret_cast<const f32x4*>(ps);
        const f32x4 hu = *reinterpret_cast<const f32x4*>(pu);
        const f32x4 hd = *reinterpret_cast<const f32x4*>(pd);
        const f32x4 hl = *reinterpret_cast<const f32x4*>(pl);
        const f32x4 hr = *reinterpret_cast<const f32x4*>(pr);
        f32x4 v = bias + k4[0] * hs + k4[1] * hu + k4[2] * hd + k4[3] * hl + kr * hr;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        out[it] = v;
        if (it & 1) __builtin_amdgcn_sched_barrier(0);
    }
    if (!LAST) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < STRIPE_ITERS; ++it) {
            const int r = stripe_row(it, rs);
            if (r < 81) store_split4<NPL, F16>(sm, r * PROW + 2 * colb, out[it]);
        }
    } else {
        f32x4 sum = out[0];
#pragma unroll
        for (int it = 1; it < STRIPE_ITERS - 1; ++it) sum += out[it];
        if (rs == 0) sum += out[STRIPE_ITERS - 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = sum[e];
            x += __shfl_xor(x, 8); x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
            sum[e] = x * (1.0f / 81.0f);
        }
        if (rs == 0) *reinterpret_cast<f32x4*>(pooled_out + colb) = sum;
    }
}

// NT = MFMA terms per product block: 6 (bf16 hi/mid/lo, 3 planes) or 3 (2 planes).  F16 selects fp16 planes instead of
// bf16: hi + lo then carry 22 mantissa bits, so NT = 3 with fp16 is fp32-equivalent with only TWO planes.
template <int NT, int WGS_PER_CU, bool F16>
__global__ __launch_bounds__(256, WGS_PER_CU) void gcn_trunk_boards_bf16_kernel(const void* __restrict__ states, int fmt, int B,
                                                                                 const float* __restrict__ pk,
                                                                                 float* __restrict__ pooled,
                                                                                 const uint8_t* __restrict__ active) {
    constexpr int N = 9, V = 81, S = 8;
    constexpr int NPL = (NT == 6) ? 3 : 2;
    __shared__ TrunkSmemB<NPL> sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform: wave-derived offsets stay in SGPRs

    const int ndw = fmt == 0 ? 18 : 6;
    int b = blockIdx.x;
    while (b < B && active && !active[b]) b += gridDim.x;
    if (b < B && tid < ndw) sm.raw[tid] = reinterpret_cast<const uint32_t*>(states)[(size_t)b * ndw + tid];
    __syncthreads();
    u32x4 Bf[NPL][2][4];

    AQG_STAMP_DECL
    while (b < B) {
        AQG_STAMP_AT(7)
        // Global loads are issued right AFTER a barrier, never just before one: __syncthreads() waits vmcnt(0), so a
        // load issued ahead of it exposes its whole L2 latency, while one issued behind it lands under the next phase.
        // vmcnt retires in order: a small load issued AFTER a prefetch can only be waited for by draining the whole
        // prefetch, so the per-layer biases are fetched first and handed to the gathers by value.
        const f32x4 bias2 = *reinterpret_cast<const f32x4*>(pk + PackedLayout::B2 + 32 * wave + 4 * (lane & 7));
        const f32x4 bias3 = *reinterpret_cast<const f32x4*>(pk + PackedLayout::B3 + 32 * wave + 4 * (lane & 7));
        load_bfrag<NPL>(Bf, pk + (F16 ? PackedLayout::WH2 : PackedLayout::WB2), wave, lane);            // lands under setup + layer 1
        __builtin_amdgcn_sched_barrier(0);                                  // keep the loads at the phase start
        // ---- setup step 1: node features + this tile's open-edge bits
        if (tid < V) {
            QState s;
            if (fmt == 0) {
                uint64_t h = 0, v = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t x = sm.raw[1 + i];
                    h |= (uint64_t)gather_bit0_x4(x) << (4 * i);
                    v |= (uint64_t)gather_bit0_x4(x >> 1) << (4 * i);
                }
                s.hw = h; s.vw = v;
            } else {
                s.hw = (uint64_t)sm.raw[0] | ((uint64_t)sm.raw[1] << 32);
                s.vw = (uint64_t)sm.raw[2] | ((uint64_t)sm.raw[3] << 32);
            }
            const uint32_t hd = sm.raw[fmt == 0 ? 0 : 4];
            s.ppos = (uint8_t)(hd & 0xff); s.pwl = (uint8_t)((hd >> 8) & 0xff);
            s.epos = (uint8_t)((hd >> 16) & 0xff); s.ewl = (uint8_t)(hd >> 24);
            s.plies = 0; s.pad = 0;
            int t = tid;
            asm volatile("" : "+v"(t));   // opaque per iteration: keeps hipcc from hoisting (and then spilling) the
                                          // per-tile masks / addresses out of the board loop
            const int x = t / N, y = t % N;
            sm.obits[t] = tile_open_bits<N>(s.hw, s.vw, t);
            const bool slot_ok = (x < S) && (y < S);
            const int slot = x * S + y;
            f32x4 xa, xb;
            xa[0] = (t == s.ppos) ? 1.f : 0.f;
            xa[1] = (float)s.pwl;
            xa[2] = (t == s.epos) ? 1.f : 0.f;
            xa[3] = (float)s.ewl;
            xb[0] = (slot_ok && ((s.hw >> slot) & 1)) ? 1.f : 0.f;
            xb[1] = (slot_ok && ((s.vw >> slot) & 1)) ? 1.f : 0.f;
            xb[2] = 0.f; xb[3] = 0.f;
            *reinterpret_cast<f32x4*>(sm.X0 + t * FPAD) = xa;
            *reinterpret_cast<f32x4*>(sm.X0 + t * FPAD + 4) = xb;
        }
        int bn = b + gridDim.x;
        while (bn < B && active && !active[bn]) bn += gridDim.x;
        __syncthreads();
        AQG_STAMP_AT(0)
        uint32_t rawreg = 0;                                                // next board's record: one dword per lane,
        if (bn < B && tid < ndw) rawreg = reinterpret_cast<const uint32_t*>(states)[(size_t)bn * ndw + tid];   // parked in LDS below
        // ---- setup step 2 + layer 1a
        if (tid < V) {
            int t = tid;
            asm volatile("" : "+v"(t));
            const int ob = sm.obits[t];
            const int tu = t >= 9 ? t - 9 : t, td = t < 72 ? t + 9 : t, tl = t > 0 ? t - 1 : t, tr = t < 80 ? t + 1 : t;
            const float di = dinv_of_bits(ob);
            f32x4 k4;
            k4[0] = di * di;
            k4[1] = (ob & 1) ? di * dinv_of_bits(sm.obits[tu]) : 0.f;
            k4[2] = (ob & 2) ? di * dinv_of_bits(sm.obits[td]) : 0.f;
            k4[3] = (ob & 4) ? di * dinv_of_bits(sm.obits[tl]) : 0.f;
            const float kr = (ob & 8) ? di * dinv_of_bits(sm.obits[tr]) : 0.f;
            *reinterpret_cast<f32x4*>(&sm.coef[prow(t)][0]) = k4;
            sm.coef[prow(t)][4] = kr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = k4[0] * *reinterpret_cast<const f32x4*>(sm.X0 + t * FPAD + 4 * h) +
                                k4[1] * *reinterpret_cast<const f32x4*>(sm.X0 + tu * FPAD + 4 * h) +
                                k4[2] * *reinterpret_cast<const f32x4*>(sm.X0 + td * FPAD + 4 * h) +
                                k4[3] * *reinterpret_cast<const f32x4*>(sm.X0 + tl * FPAD + 4 * h) +
                                kr * *reinterpret_cast<const f32x4*>(sm.X0 + tr * FPAD + 4 * h);
                *reinterpret_cast<f32x4*>(sm.AX + prow(t) * FPAD + 4 * h) = v;
            }
        }
        __syncthreads();
        AQG_STAMP_AT(1)
        // ---- layer 1b (f32 VALU, K = 6) -> split planes
        {
            const int cg = lane & 7, rs = lane >> 3;
            const int colb = 32 * wave + 4 * cg;
            // W1 for this lane's 4 columns as six 4-wide vectors (one per input feature): the K = 6 contraction is then
            // six packed FMAs per row instead of 24 scalar ones
            f32x4 wc[6];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(pk + PackedLayout::W1 + (colb + e) * FPAD);
                const float2 hi = *reinterpret_cast<const float2*>(pk + PackedLayout::W1 + (colb + e) * FPAD + 4);
                wc[0][e] = lo[0]; wc[1][e] = lo[1]; wc[2][e] = lo[2]; wc[3][e] = lo[3]; wc[4][e] = hi.x; wc[5][e] = hi.y;
            }
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(pk + PackedLayout::B1 + colb);
#pragma unroll
            for (int it = 0; it < STRIPE_ITERS; ++it) {
                const int r = stripe_row(it, rs);
                if (r < V) {
                    const f32x4 xa = *reinterpret_cast<const f32x4*>(sm.AX + prow(r) * FPAD);
                    const float2 xb = *reinterpret_cast<const float2*>(sm.AX + prow(r) * FPAD + 4);
                    f32x4 v = b1 + xa[0] * wc[0] + xa[1] * wc[1] + xa[2] * wc[2] + xa[3] * wc[3] + xb.x * wc[4] + xb.y * wc[5];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    store_split4<NPL, F16>(sm, r * PROW + 2 * colb, v);
                }
                if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        AQG_STAMP_AT(2)
        // ---- layer 2
        f32x4 acc[6][2];
        AQG_DIAG_FENCE_AT(1);
        stripe_matmul_bf16<NPL, F16>(sm, Bf, lane, acc);
        AQG_DIAG_FENCE_AT(2);
        __syncthreads();
        AQG_STAMP_AT(3)
        load_bfrag<NPL>(Bf, pk + (F16 ? PackedLayout::WH3 : PackedLayout::WB3), wave, lane);            // lands under the layer-2 epilogue
        __builtin_amdgcn_sched_barrier(0);
        stripe_gather_bf16<NPL, F16, false>(sm, acc, bias2, wave, lane, nullptr);
        AQG_DIAG_FENCE_AT(4);
        __syncthreads();
        AQG_STAMP_AT(4)
        // ---- layer 3 + mean pool
        stripe_matmul_bf16<NPL, F16>(sm, Bf, lane, acc);
        AQG_DIAG_FENCE_AT(8);
        __syncthreads();
        AQG_STAMP_AT(5)
        stripe_gather_bf16<NPL, F16, true>(sm, acc, bias3, wave, lane, pooled + (size_t)b * HID);
        if (tid < ndw) sm.raw[tid] = rawreg;                                // setup (its only reader) is long done
        __syncthreads();
        AQG_STAMP_AT(6)
#ifdef AQG_STAMP
        ++st_n;
#endif
        b = bn;
    }
#ifdef AQG_STAMP
    if (blockIdx.x == 0 && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(pooled + (size_t)B * HID);
        for (int i = 0; i < 8; ++i) o[i] = st_sum[i];
        o[8] = (unsigned long long)st_n;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// heads: 16 boards per workgroup
// ---------------------------------------------------------------------------------------------
constexpr int HB = 8;   // boards per workgroup: B = 2048 -> 256 workgroups, one per CU

// Latency-bound small GEMMs: the weights stream from L2 (118 KB, shared by every workgroup), so the loop is built
// for memory-level parallelism -- 16 independent coalesced weight loads in flight per thread -- and the 8 boards
// of a workgroup sit transposed in LDS ([k][board]) so one k-step reads them with two broadcast ds_read_b128.
__global__ __launch_bounds__(256) void gcn_heads_kernel(const float* __restrict__ pooled, int B, int A,
                                                        const float* __restrict__ pk, float* __restrict__ logits,
                                                        float* __restrict__ policy, float* __restrict__ value_pre,
                                                        float* __restrict__ value, const uint8_t* __restrict__ active) {
    __shared__ alignas(16) float gT[HID][HB];          // pooled features, transposed
    __shared__ alignas(16) float part[2][HID][HB];     // hidden-layer partial sums of the two k halves
    __shared__ alignas(16) float hidT[HID][HB];        // hidden activations (0..63 policy, 64..127 value), transposed
    __shared__ float lg[HB][APAD];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * HB;
    const int nb = min(HB, B - b0);
    for (int i = tid; i < HB * HID; i += 256) {
        const int r = i / HID, k = i % HID;
        gT[k][r] = (r < nb) ? pooled[(size_t)(b0 + r) * HID + k] : 0.f;
    }
    __syncthreads();
    {   // hidden layer of both heads: unit u, k half kh (64 k each), all 8 boards
        const int u = tid & 127, kh = tid >> 7;
        float acc[HB];
#pragma unroll
        for (int i = 0; i < HB; ++i) acc[i] = 0.f;
        const float* w = pk + PackedLayout::HW1T + (size_t)(64 * kh) * HID + u;
#pragma unroll
        for (int k0 = 0; k0 < 64; k0 += 16) {
            float wk[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wk[j] = w[(size_t)(k0 + j) * HID];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(&gT[64 * kh + k0 + j][0]);
                const f32x4 gb = *reinterpret_cast<const f32x4*>(&gT[64 * kh + k0 + j][4]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i] = fmaf(ga[i], wk[j], acc[i]); acc[4 + i] = fmaf(gb[i], wk[j], acc[4 + i]); }
            }
        }
        *reinterpret_cast<f32x4*>(&part[kh][u][0]) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(&part[kh][u][4]) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
    }
    __syncthreads();
    for (int i = tid; i < HID * HB; i += 256) {
        const int u = i / HB, r = i % HB;
        hidT[u][r] = fmaxf(part[0][u][r] + part[1][u][r] + pk[PackedLayout::HB1 + u], 0.f);
    }
    __syncthreads();
    if (tid < A) {
        float acc[HB];
        const float bias = pk[PackedLayout::PB2 + tid];
#pragma unroll
        for (int i = 0; i < HB; ++i) acc[i] = bias;
        const float* w = pk + PackedLayout::PW2T + tid;
#pragma unroll
        for (int k0 = 0; k0 < HID / 2; k0 += 16) {
            float wk[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wk[j] = w[(size_t)(k0 + j) * APAD];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const f32x4 ha = *reinterpret_cast<const f32x4*>(&hidT[k0 + j][0]);
                const f32x4 hb = *reinterpret_cast<const f32x4*>(&hidT[k0 + j][4]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i] = fmaf(ha[i], wk[j], acc[i]); acc[4 + i] = fmaf(hb[i], wk[j], acc[4 + i]); }
            }
        }
#pragma unroll
        for (int i = 0; i < HB; ++i) lg[i][tid] = acc[i];
    } else if (tid >= 248) {   // value head: one thread per board
        const int i = tid - 248;
        float acc = pk[PackedLayout::VB2];
        for (int k = 0; k < HID / 2; ++k) acc = fmaf(hidT[HID / 2 + k][i], pk[PackedLayout::VW2 + k], acc);
        if (i < nb && !(active && !active[b0 + i])) {
            if (value_pre) value_pre[b0 + i] = acc;
            if (value) value[b0 + i] = tanhf(acc);
        }
    }
    __syncthreads();
    // softmax: wave w handles boards 2w, 2w+1
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = 2 * wave; r < 2 * wave + 2; ++r) {
        if (r >= nb) break;
        if (active && !active[b0 + r]) continue;
        float m = -INFINITY;
        for (int a = lane; a < A; a += 64) m = fmaxf(m, lg[r][a]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float e[4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = lane + 64 * j;
            e[j] = (a < A) ? expf(lg[r][a] - m) : 0.f;
            s += e[j];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = lane + 64 * j;
            if (a < A) {
                if (logits) logits[(size_t)(b0 + r) * A + a] = lg[r][a];
                if (policy) policy[(size_t)(b0 + r) * A + a] = e[j] / s;
            }
        }
    }
}

// Trunk variants (aqg_set_option("trunk_variant", v)):
//   0 exact f32 MFMA, weights resident, 1 workgroup/CU      1 exact f32 MFMA, 2 workgroups/CU
//   3 fp16x3 split MFMA (hi+lo fp16 planes = 22 mantissa bits: fp32-equivalent), 2/CU  [default]
//   4 bf16x3 split MFMA (~2^-16 relative per product), 2/CU
// A bf16x6 (three bf16 planes) instantiation of the same template was the default for a while and is NOT shipped: its
// 74.9 KB image puts the second-resident workgroup's third plane above 128 KB of the CU's LDS, where cross-wave
// hand-offs through s_waitcnt lgkmcnt(0) + s_barrier were observed to read stale data on cold launches.
// Measured and dropped this round (slower): 3 workgroups/CU under a 168-VGPR cap (spills), and an 8-wave /
// 16-column-stripe form at 4 waves per SIMD (spills + doubled A-operand LDS reads): 15.6 M vs 23.3 M boards/s.
int g_trunk_variant = 3;
int g_trunk_grid = 0;      // 0 = default persistent grid; otherwise override (diagnostics)

// Diagnostic: fill every CU's LDS with NaN bit patterns so that any read-before-write in a later kernel shows up
// deterministically (used by the parity tests; LDS contents are otherwise whatever the previous kernel left).
__global__ __launch_bounds__(256, 2) void poison_lds_kernel(unsigned int* sink) {
    __shared__ unsigned int buf[20000];     // 80,000 B: two workgroups per CU cover 160 KB
    for (int i = threadIdx.x; i < 20000; i += 256) buf[i] = 0xFFFFFFFFu;
    __syncthreads();
    if (sink && buf[(threadIdx.x * 77) % 20000] == 0x12345678u) sink[0] = 1;   // keep the stores alive
}
int launch_poison_lds(hipStream_t st) {
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(256), 0, st, (unsigned int*)nullptr);
    return check_launch("poison_lds_kernel");
}

// Optional launch profiling of the dominant kernel (aqg_set_option("profile_trunk", 1)): a HIP event pair is
// recorded around every trunk launch on the launch stream; aqg_profile_collect() reads them back.
int g_profile_trunk = 0;
static std::vector<hipEvent_t> g_prof_events;
static size_t g_prof_used = 0;
static double g_prof_ms = 0.0;
static long long g_prof_launches = 0;
static long long g_prof_boards = 0;

static hipEvent_t prof_event() {
    if (g_prof_used == g_prof_events.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        g_prof_events.push_back(e);
    }
    return g_prof_events[g_prof_used++];
}

int profile_collect(double* total_ms, long long* launches, long long* boards, int reset) {
    if (g_prof_used) {
        if (hipEventSynchronize(g_prof_events[g_prof_used - 1]) != hipSuccess) return fail("hipEventSynchronize");
        for (size_t i = 0; i + 1 < g_prof_used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, g_prof_events[i], g_prof_events[i + 1]) != hipSuccess) return fail("hipEventElapsedTime");
            g_prof_ms += ms;
            ++g_prof_launches;
        }
        g_prof_used = 0;
    }
    if (total_ms) *total_ms = g_prof_ms;
    if (launches) *launches = g_prof_launches;
    if (boards) *boards = g_prof_boards;
    if (reset) { g_prof_ms = 0.0; g_prof_launches = 0; g_prof_boards = 0; }
    return 0;
}

int launch_gcn_forward_boards(int N, const void* states, int fmt, int B, const float* packed, float* pooled,
                              float* logits, float* policy, float* value_pre, float* value, const uint8_t* active,
                              hipStream_t st) {
    if (N != 9) return fail("fused board trunk is built for 9x9; use aqg_gcn_forward_graph for other sizes");
    if (B <= 0) return 0;
    if (!pooled) return fail("pooled workspace is required");
    if (N * N + 2 * (N - 1) * (N - 1) > 248) return fail("policy size exceeds 248");
    const int A = N * N + 2 * (N - 1) * (N - 1);
    // persistent grid: 256 CUs x resident workgroups per CU, grid-stride over boards
    if (g_profile_trunk) { hipEventRecord(prof_event(), st); g_prof_boards += B; }
    if (g_trunk_variant == 0) {
        int grid = B < 256 ? B : 256;
        hipLaunchKernelGGL((gcn_trunk_boards_kernel<true, 1>), dim3(grid), dim3(256), 0, st, states, fmt, B, packed, pooled, active);
    } else if (g_trunk_variant == 1) {
        int grid = B < 512 ? B : 512;
        hipLaunchKernelGGL((gcn_trunk_boards_kernel<false, 2>), dim3(grid), dim3(256), 0, st, states, fmt, B, packed, pooled, active);
    } else if (g_trunk_variant == 3) {
        int grid = B < 512 ? B : 512;
        if (g_trunk_grid > 0 && g_trunk_grid < grid) grid = g_trunk_grid;
        hipLaunchKernelGGL((gcn_trunk_boards_bf16_kernel<3, 2, true>), dim3(grid), dim3(256), 0, st, states, fmt, B, packed, pooled, active);
    } else {
        int grid = B < 512 ? B : 512;
        if (g_trunk_grid > 0 && g_trunk_grid < grid) grid = g_trunk_grid;
        hipLaunchKernelGGL((gcn_trunk_boards_bf16_kernel<3, 2, false>), dim3(grid), dim3(256), 0, st, states, fmt, B, packed, pooled, active);
    }
    if (g_profile_trunk) hipEventRecord(prof_event(), st);
    if (int r = check_launch("gcn_trunk_boards_kernel")) return r;
    if (!logits && !policy && !value_pre && !value) return 0;   // trunk only (bench: time the dominant kernel alone)
    hipLaunchKernelGGL(gcn_heads_kernel, dim3((B + HB - 1) / HB), dim3(256), 0, st, (const float*)pooled, B, A, packed,
                       logits, policy, value_pre, value, active);
    return check_launch("gcn_heads_kernel");
}

// ---------------------------------------------------------------------------------------------
// generic graph path (forward(x, edge_index, batch)): linear -> CSR gather -> pool; correctness-first
// ---------------------------------------------------------------------------------------------
// Y[n][HID] = X[n][K] * WT[K][HID]   (WT row stride ldw; K = 6 (padded rows of W1 read as [n][f]) or 128)
template <bool W_IS_NF>
__global__ __launch_bounds__(256) void graph_linear_kernel(const float* __restrict__ X, int K, int num_nodes,
                                                           const float* __restrict__ W, float* __restrict__ Y) {
    __shared__ float xs[32][HID + 1];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * 32;
    for (int i = tid; i < 32 * K; i += 256) {
        const int r = i / K, k = i % K;
        xs[r][k] = (n0 + r < num_nodes) ? X[(size_t)(n0 + r) * K + k] : 0.f;
    }
    __syncthreads();
    const int col = tid & 127, half = tid >> 7;   // 16 nodes per thread
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float w = W_IS_NF ? W[col * FPAD + k] : W[(size_t)k * HID + col];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(xs[16 * half + i][k], w, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = n0 + 16 * half + i;
        if (n < num_nodes) Y[(size_t)n * HID + col] = acc[i];
    }
}

// out[i] = relu(sum_{e in csr[i]} w_e * Y[src_e] + bias): one wave per node, lane = 2 columns
__global__ __launch_bounds__(256) void graph_gather_kernel(const float* __restrict__ Y, int num_nodes,
                                                           const int32_t* __restrict__ ptr, const int32_t* __restrict__ src,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= num_nodes) return;
    float a0 = bias[2 * lane], a1 = bias[2 * lane + 1];
    for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
        const float we = w[e];
        const float2 y = *reinterpret_cast<const float2*>(Y + (size_t)src[e] * HID + 2 * lane);
        a0 = fmaf(we, y.x, a0);
        a1 = fmaf(we, y.y, a1);
    }
    *reinterpret_cast<float2*>(out + (size_t)i * HID + 2 * lane) = make_float2(fmaxf(a0, 0.f), fmaxf(a1, 0.f));
}

__global__ __launch_bounds__(128) void graph_pool_kernel(const float* __restrict__ Hn, const int32_t* __restrict__ gptr,
                                                         int num_graphs, float* __restrict__ pooled) {
    const int g = blockIdx.x;
    if (g >= num_graphs) return;
    const int a = gptr[g], b = gptr[g + 1];
    float s = 0.f;
    for (int i = a; i < b; ++i) s += Hn[(size_t)i * HID + threadIdx.x];
    pooled[(size_t)g * HID + threadIdx.x] = (b > a) ? s / (float)(b - a) : 0.f;
}

int launch_gcn_forward_graph(int F, int A, const float* x, int num_nodes, const int32_t* csr_ptr, const int32_t* csr_src,
                             const float* csr_w, const int32_t* graph_ptr, int num_graphs, const float* packed,
                             float* work0, float* work1, float* pooled, float* logits, float* policy, float* value_pre,
                             float* value, hipStream_t st) {
    if (F != 6) return fail("num_features must be 6 (NUM_FEATURES pv_network_gnn.py:17)");
    if (A > 248) return fail("policy size exceeds 248");
    if (num_nodes <= 0 || num_graphs <= 0) return 0;
    dim3 lg((num_nodes + 31) / 32), gg((num_nodes + 3) / 4);
    hipLaunchKernelGGL(graph_linear_kernel<true>, lg, dim3(256), 0, st, x, F, num_nodes, packed + PackedLayout::W1, work0);
    hipLaunchKernelGGL(graph_gather_kernel, gg, dim3(256), 0, st, (const float*)work0, num_nodes, csr_ptr, csr_src, csr_w,
                       packed + PackedLayout::B1, work1);
    hipLaunchKernelGGL(graph_linear_kernel<false>, lg, dim3(256), 0, st, (const float*)work1, HID, num_nodes,
                       packed + PackedLayout::W2T, work0);
    hipLaunchKernelGGL(graph_gather_kernel, gg, dim3(256), 0, st, (const float*)work0, num_nodes, csr_ptr, csr_src, csr_w,
                       packed + PackedLayout::B2, work1);
    hipLaunchKernelGGL(graph_linear_kernel<false>, lg, dim3(256), 0, st, (const float*)work1, HID, num_nodes,
                       packed + PackedLayout::W3T, work0);
    hipLaunchKernelGGL(graph_gather_kernel, gg, dim3(256), 0, st, (const float*)work0, num_nodes, csr_ptr, csr_src, csr_w,
                       packed + PackedLayout::B3, work1);
    hipLaunchKernelGGL(graph_pool_kernel, dim3(num_graphs), dim3(128), 0, st, (const float*)work1, graph_ptr, num_graphs, pooled);
    if (int r = check_launch("graph kernels")) return r;
    hipLaunchKernelGGL(gcn_heads_kernel, dim3((num_graphs + HB - 1) / HB), dim3(256), 0, st, (const float*)pooled, num_graphs, A,
                       packed, logits, policy, value_pre, value, (const uint8_t*)nullptr);
    return check_launch("gcn_heads_kernel");
}

}  // namespace aqg
