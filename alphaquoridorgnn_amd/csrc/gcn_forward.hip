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
    static constexpr size_t TOTAL = WF3 + HID * HID;
};

size_t packed_floats() { return PackedLayout::TOTAL; }

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

struct TrunkSmem {
    float H[81 * LD];          // in-place activation image
    float X0[81 * FPAD];       // node features
    float coef[5][96];         // self, U, D, L, R gather coefficients per node (0 when the edge is cut)
    float part[8][HID];        // pooling partials
};

// ---- one dense layer: acc = H[0..79] x W (MFMA), row 80 on VALU; then write back in place ------------
// A fragments are software-pipelined in half tiles (16 k-steps = 4 x ds_read_b128): the next half tile's
// reads are issued before the current half tile's 32 MFMAs; sched_barrier keeps hipcc from hoisting every
// read to the top (which spills at the 256-VGPR budget that two workgroups per CU allow).
__device__ __forceinline__ void dense_layer(float* __restrict__ H, const float (&Wf)[2][32], int wave, int lane) {
    const int c = lane & 15, q = lane >> 4;
    const int kb = (q & 1) * 64 + (q >> 1) * 32;  // kbase = {0, 64, 32, 96}
    f32x4 acc[5][2];
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
        if (ht < 9) {
            const float* nsrc = arow + 16 * ((ht + 1) >> 1) * LD + 16 * ((ht + 1) & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const f32x4*>(nsrc + 4 * j);
        } else {                                 // last stage prefetches node 80's first half for the VALU row
            const float* nsrc = H + 80 * LD + kb;
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const f32x4*>(nsrc + 4 * j);
        }
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
    float r0 = 0.f, r1 = 0.f;
    {
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
    __syncthreads();  // every wave has finished reading H
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) H[(16 * m + 4 * q + i) * LD + 32 * wave + 16 * j + c] = acc[m][j][i];
    if (q == 0) {
        H[80 * LD + 32 * wave + c] = r0;
        H[80 * LD + 32 * wave + 16 + c] = r1;
    }
    __syncthreads();
}

// ---- normalised neighbour gather + bias + ReLU, register-staged in place (POOL: reduce instead of write) ----
template <bool POOL>
__device__ __forceinline__ void gather_layer(TrunkSmem& sm, const float* __restrict__ bias_g, int tid,
                                             float* __restrict__ pooled_out) {
    const int cg = tid & 31, ng = tid >> 5;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_g + 4 * cg);
    f32x4 out[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int t = ng + 8 * i;
        if (t < 81) {
            const float cs = sm.coef[0][t], cu = sm.coef[1][t], cd = sm.coef[2][t], cl = sm.coef[3][t], cr = sm.coef[4][t];
            const int tu = t >= 9 ? t - 9 : t, td = t < 72 ? t + 9 : t, tl = t > 0 ? t - 1 : t, tr = t < 80 ? t + 1 : t;
            const f32x4 hs = *reinterpret_cast<const f32x4*>(sm.H + t * LD + 4 * cg);
            const f32x4 hu = *reinterpret_cast<const f32x4*>(sm.H + tu * LD + 4 * cg);
            const f32x4 hd = *reinterpret_cast<const f32x4*>(sm.H + td * LD + 4 * cg);
            const f32x4 hl = *reinterpret_cast<const f32x4*>(sm.H + tl * LD + 4 * cg);
            const f32x4 hr = *reinterpret_cast<const f32x4*>(sm.H + tr * LD + 4 * cg);
            f32x4 v = bias + cs * hs + cu * hu + cd * hd + cl * hl + cr * hr;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            out[i] = v;
        } else {
            out[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (POOL) {
        f32x4 sum = out[0];
#pragma unroll
        for (int i = 1; i < 11; ++i) sum += out[i];
        *reinterpret_cast<f32x4*>(&sm.part[ng][4 * cg]) = sum;
        __syncthreads();
        if (tid < HID) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += sm.part[g][tid];
            pooled_out[tid] = s * (1.0f / 81.0f);
        }
    } else {
        __syncthreads();  // all reads of the old image done
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int t = ng + 8 * i;
            if (t < 81) *reinterpret_cast<f32x4*>(sm.H + t * LD + 4 * cg) = out[i];
        }
        __syncthreads();
    }
}

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

// RESIDENT = true : one workgroup per CU (512-VGPR budget), both layers' fragments live in registers for the
//                   whole kernel -> zero per-board weight traffic, but no cross-workgroup phase overlap.
// RESIDENT = false: two workgroups per CU (256 VGPRs); each layer's fragments are re-fetched per board from
//                   L2 (128 KB per board per workgroup), issued one phase ahead of use.
template <bool RESIDENT>
__global__ __launch_bounds__(256, RESIDENT ? 1 : 2) void gcn_trunk_boards_kernel(const void* __restrict__ states, int fmt,
                                                                                   int B, const float* __restrict__ pk,
                                                                                   float* __restrict__ pooled,
                                                                                   const uint8_t* __restrict__ active) {
    constexpr int N = 9, V = 81, S = 8;
    __shared__ TrunkSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int kb = (q & 1) * 64 + (q >> 1) * 32;
    const int cg = tid & 31, ng = tid >> 5;

    float W2f[2][32], W3f[2][32];
    if (RESIDENT) {
        load_wfrag(W2f, pk + PackedLayout::WF2, wave, lane);
        load_wfrag(W3f, pk + PackedLayout::WF3, wave, lane);
    }

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        if (active && !active[b]) continue;   // uniform per workgroup
        // ---- setup: features + gather coefficients
        if (tid < V) {
            const QState s = load_state(states, fmt, b);
            const int t = tid, x = t / N, y = t % N;
            const int ob = tile_open_bits<N>(s.hw, s.vw, t);
            const float di = dinv_of_bits(ob);
            sm.coef[0][t] = di * di;
            sm.coef[1][t] = (ob & 1) ? di * dinv_of_bits(tile_open_bits<N>(s.hw, s.vw, t - N)) : 0.f;
            sm.coef[2][t] = (ob & 2) ? di * dinv_of_bits(tile_open_bits<N>(s.hw, s.vw, t + N)) : 0.f;
            sm.coef[3][t] = (ob & 4) ? di * dinv_of_bits(tile_open_bits<N>(s.hw, s.vw, t - 1)) : 0.f;
            sm.coef[4][t] = (ob & 8) ? di * dinv_of_bits(tile_open_bits<N>(s.hw, s.vw, t + 1)) : 0.f;
            float* xr = sm.X0 + t * FPAD;
            const bool slot_ok = (x < S) && (y < S);
            const int slot = x * S + y;
            xr[0] = (t == s.ppos) ? 1.f : 0.f;
            xr[1] = (float)s.pwl;
            xr[2] = (t == s.epos) ? 1.f : 0.f;      // enemy's own frame (pv_network_cnn.py:101)
            xr[3] = (float)s.ewl;
            xr[4] = (slot_ok && ((s.hw >> slot) & 1)) ? 1.f : 0.f;
            xr[5] = (slot_ok && ((s.vw >> slot) & 1)) ? 1.f : 0.f;
            xr[6] = 0.f; xr[7] = 0.f;
        }
        if (!RESIDENT) load_wfrag(W2f, pk + PackedLayout::WF2, wave, lane);   // lands under layer 1
        __syncthreads();
        // ---- layer 1: gather the 6 features, then 6 -> 128, bias, ReLU (weights re-read per board: L1/L2 hits)
        float w1[4][6];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(pk + PackedLayout::W1 + (4 * cg + e) * FPAD);
            const float2 hi = *reinterpret_cast<const float2*>(pk + PackedLayout::W1 + (4 * cg + e) * FPAD + 4);
            w1[e][0] = lo[0]; w1[e][1] = lo[1]; w1[e][2] = lo[2]; w1[e][3] = lo[3]; w1[e][4] = hi.x; w1[e][5] = hi.y;
        }
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(pk + PackedLayout::B1 + 4 * cg);
#pragma unroll 1
        for (int i = 0; i < 11; ++i) {
            const int t = ng + 8 * i;
            if (t < V) {
                const float cs = sm.coef[0][t], cu = sm.coef[1][t], cd = sm.coef[2][t], cl = sm.coef[3][t], cr = sm.coef[4][t];
                const int tu = t >= 9 ? t - 9 : t, td = t < 72 ? t + 9 : t, tl = t > 0 ? t - 1 : t, tr = t < 80 ? t + 1 : t;
                float ax[6];
#pragma unroll
                for (int f = 0; f < 6; ++f)
                    ax[f] = cs * sm.X0[t * FPAD + f] + cu * sm.X0[tu * FPAD + f] + cd * sm.X0[td * FPAD + f] +
                            cl * sm.X0[tl * FPAD + f] + cr * sm.X0[tr * FPAD + f];
                f32x4 v = b1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) v[e] = fmaf(ax[f], w1[e][f], v[e]);
                    v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<f32x4*>(sm.H + t * LD + 4 * cg) = v;
            }
        }
        __syncthreads();
        // ---- layer 2
        dense_layer(sm.H, W2f, wave, lane);
        if (!RESIDENT) load_wfrag(W3f, pk + PackedLayout::WF3, wave, lane);   // lands under the layer-2 gather
        gather_layer<false>(sm, pk + PackedLayout::B2, tid, nullptr);
        // ---- layer 3 + mean pool
        dense_layer(sm.H, W3f, wave, lane);
        gather_layer<true>(sm, pk + PackedLayout::B3, tid, pooled + (size_t)b * HID);
        __syncthreads();  // part[] / coef / X0 reused by the next board
    }
}

// ---------------------------------------------------------------------------------------------
// heads: 16 boards per workgroup
// ---------------------------------------------------------------------------------------------
constexpr int HB = 16;

__global__ __launch_bounds__(256) void gcn_heads_kernel(const float* __restrict__ pooled, int B, int A,
                                                        const float* __restrict__ pk, float* __restrict__ logits,
                                                        float* __restrict__ policy, float* __restrict__ value_pre,
                                                        float* __restrict__ value, const uint8_t* __restrict__ active) {
    __shared__ float g[HB][HID];
    __shared__ float hid[HB][HID];
    __shared__ float lg[HB][APAD];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * HB;
    const int nb = min(HB, B - b0);
    for (int i = tid; i < HB * HID; i += 256) {
        const int r = i / HID, k = i % HID;
        g[r][k] = (r < nb) ? pooled[(size_t)(b0 + r) * HID + k] : 0.f;
    }
    __syncthreads();
    {   // hidden layer of both heads: unit u (0..63 policy, 64..127 value), 8 boards per thread
        const int u = tid & 127, bh = tid >> 7;
        float acc[8];
        const float bias = pk[PackedLayout::HB1 + u];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = bias;
        const float* w = pk + PackedLayout::HW1T + u;
#pragma unroll 4
        for (int k = 0; k < HID; ++k) {
            const float wk = w[(size_t)k * HID];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(g[8 * bh + i][k], wk, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) hid[8 * bh + i][u] = fmaxf(acc[i], 0.f);
    }
    __syncthreads();
    if (tid < A) {
        float acc[HB];
        const float bias = pk[PackedLayout::PB2 + tid];
#pragma unroll
        for (int i = 0; i < HB; ++i) acc[i] = bias;
        const float* w = pk + PackedLayout::PW2T + tid;
#pragma unroll 4
        for (int k = 0; k < HID / 2; ++k) {
            const float wk = w[(size_t)k * APAD];
#pragma unroll
            for (int i = 0; i < HB; ++i) acc[i] = fmaf(hid[i][k], wk, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < HB; ++i) lg[i][tid] = acc[i];
    } else if (tid >= 240) {   // value head: one thread per board
        const int i = tid - 240;
        float acc = pk[PackedLayout::VB2];
        for (int k = 0; k < HID / 2; ++k) acc = fmaf(hid[i][HID / 2 + k], pk[PackedLayout::VW2 + k], acc);
        if (i < nb && !(active && !active[b0 + i])) {
            if (value_pre) value_pre[b0 + i] = acc;
            if (value) value[b0 + i] = tanhf(acc);
        }
    }
    __syncthreads();
    // softmax: wave w handles boards 4w..4w+3
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = 4 * wave; r < 4 * wave + 4; ++r) {
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

int g_trunk_variant = 1;  // set by aqg_set_option("trunk_variant", v)

int launch_gcn_forward_boards(int N, const void* states, int fmt, int B, const float* packed, float* pooled,
                              float* logits, float* policy, float* value_pre, float* value, const uint8_t* active,
                              hipStream_t st) {
    if (N != 9) return fail("fused board trunk is built for 9x9; use aqg_gcn_forward_graph for other sizes");
    if (B <= 0) return 0;
    if (!pooled) return fail("pooled workspace is required");
    if (N * N + 2 * (N - 1) * (N - 1) > 240) return fail("policy size exceeds 240");
    const int A = N * N + 2 * (N - 1) * (N - 1);
    // persistent grid: 256 CUs x resident workgroups per CU, grid-stride over boards
    if (g_trunk_variant == 0) {
        int grid = B < 256 ? B : 256;
        hipLaunchKernelGGL(gcn_trunk_boards_kernel<true>, dim3(grid), dim3(256), 0, st, states, fmt, B, packed, pooled, active);
    } else {
        int grid = B < 512 ? B : 512;
        hipLaunchKernelGGL(gcn_trunk_boards_kernel<false>, dim3(grid), dim3(256), 0, st, states, fmt, B, packed, pooled, active);
    }
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
    if (A > APAD) return fail("policy size exceeds 256");
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
