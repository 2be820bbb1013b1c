// gcn_train.hip -- one optimisation step of the reference's training loop on the GNN, for gfx950.
//
// train_network.py:68-95 (forward, CrossEntropyLoss on the ALREADY-softmaxed policy + MSELoss on the
// tanh value, backward, Adam) applied to GraphPolicyValueNetwork (pv_network_gnn.py:23-64, GCNConv = PyG defaults).
//
//   forward   Z = H W^T, P = A_hat Z + b, H' = relu(P)   x3;  g = mean_nodes H3;  heads;  pol = softmax, val = tanh
//   loss      Lp = mean_b -sum_a t_a log_softmax(pol)_a   (the reference's double softmax, kept on purpose)
//             Lv = mean_b (val - z)^2
//   backward  dP = dH' (.) [H' > 0];  db = colsum dP;  dZ = A_hat dP (A_hat symmetric);  dW = dZ^T H;  dH = dZ W
//   update    torch.optim.Adam (lr, betas, eps; bias-corrected; no weight decay, no amsgrad)
//
// The batch is 128 positions (train_network.py:15) = 10,368 graph nodes and 2.2 GFLOP per step.  Every contraction over the
// node rows runs on the f32 matrix pipe (v_mfma_f32_16x16x4_f32: exact f32 products, which the 2e-5 gradient parity against
// fp64 autograd needs).  Two forms of the step (aqg_set_option("train_fused")):
//
//   fused (default): train_board_kernel -- ONE 8-wave workgroup per position does forward, heads + losses and backward; only
//     the per-board partial gradients leave the CU -- then train_final_kernel.
//   six launches + final: two workgroups per board split the 128 feature columns (all 256 CUs busy at batch 128) and exchange
//     full rows through memory between launches:
//       fwd12   (board, column half)  features + graph from the record; layer 1 (K = 6, both halves redundantly), layer 2 half
//       fwd3    (board, column half)  layer 3 half + the mean pool of that half
//       heads   (board)               both heads, the two losses, and the head gradients back to dg
//       bwd<3>  (board, column half)  dP3, db3, dZ3 = A_hat dP3, dW3 = dZ3^T H2 (per-board partial)
//       bwd<2>  (board, column half)  dH2 = dZ3 W3 (half of the columns), dP2, db2, dZ2, dW2 partial
//       bwd<1>  (board, column half)  dH1 = dZ2 W2, dP1, db1, dZ1, dW1 partial
//     The column split is consistent through the chain: the aggregation is per column, and each contraction takes FULL rows of
//     its input (written by the previous launch) and produces one column half.
//   final   (parameter element)   sums the per-board partials in a fixed order, forms the head weight gradients as
//                                 batch dot products, writes the gradient and applies Adam to that element
//
// A board's 81 node rows never leave its workgroup (the aggregation needs all of them).  No atomics anywhere: results are
// run-to-run identical.
#include "aqg_common.hpp"
#include "../../include/aqgnn.h"

namespace aqg {

constexpr int TH = 128;    // HIDDEN_DIM
constexpr int TF = 6;      // NUM_FEATURES
constexpr int HH = 64;     // columns per workgroup
constexpr int SA = 132;    // LDS row stride of a [rows][128] A operand  (132 = 4 mod 64: 16 rows x 4 k-lanes hit 64 banks)
constexpr int SZ = 68;     // LDS row stride of a [rows][64] accumulator image (same property for the accumulator layout)
constexpr int SD = 80;     // LDS row stride of dZ [nodes][64] as the A operand of the weight gradient (80 = 16 mod 64)
constexpr int SB = 144;    // LDS row stride of H [nodes][128] as its B operand (144 = 16 mod 64)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DAQG_STAMP, tools/stamp_train.py; never shipped): thread 0 of workgroup 0 adds the cycles between
// consecutive phase marks of kernel k to g_train_stamp[k][phase].
#ifdef AQG_STAMP
__device__ unsigned long long g_train_stamp[10][16];
#define TS_DECL unsigned long long ts_prev = __builtin_readcyclecounter();
#define TS(k, i) { const unsigned long long ts_now = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) g_train_stamp[k][i] += ts_now - ts_prev; ts_prev = ts_now; }
#else
#define TS_DECL
#define TS(k, i)
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }

// One board's graph in LDS: PyG gcn_norm weights in ELL form (self, U, D, L, R; a closed side has weight 0 and points at
// the node itself) and the six node features (pv_network_cnn.py:88-114; edges = open tile adjacencies, game_logic.py:145-167).
struct BoardGraph {
    float w[96 * 5];
    float x0[96 * 8];          // features, zero-padded to 8 columns and to whole row tiles
    unsigned char nb[96 * 4];
};

template <int N>
__device__ __forceinline__ void board_graph(BoardGraph& gr, const uint8_t* __restrict__ rec, int t) {
    constexpr int V = N * N, S = N - 1, VP = (V + 15) / 16 * 16;
    if (t < VP) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        st4(gr.x0 + t * 8, z);
        st4(gr.x0 + t * 8 + 4, z);
    }
    if (t < V) {
        const QState s = unpack72(rec);
        const int x = t / N, y = t % N;
        const bool slot_ok = x < S && y < S;
        const int slot = x * S + y;
        float* f = gr.x0 + t * 8;
        f[0] = (t == s.ppos) ? 1.f : 0.f;
        f[1] = (float)s.pwl;
        f[2] = (t == s.epos) ? 1.f : 0.f;
        f[3] = (float)s.ewl;
        f[4] = (slot_ok && ((s.hw >> slot) & 1)) ? 1.f : 0.f;
        f[5] = (slot_ok && ((s.vw >> slot) & 1)) ? 1.f : 0.f;
        const int ob = tile_open_bits<N>(s.hw, s.vw, t);
        const float di = 1.0f / sqrtf((float)(1 + __popc(ob)));
        const int nbr[4] = {t - N, t + N, t - 1, t + 1};
        gr.w[t * 5] = di * di;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const bool open = (ob >> d) & 1;
            float w = 0.f;
            int j = t;
            if (open) {
                const int obn = tile_open_bits<N>(s.hw, s.vw, nbr[d]);
                w = di * (1.0f / sqrtf((float)(1 + __popc(obn))));
                j = nbr[d];
            }
            gr.w[t * 5 + 1 + d] = w;
            gr.nb[t * 4 + d] = (unsigned char)j;
        }
    }
}

// (A_hat Z)[n][c4 .. c4+3] from an LDS image of Z with row stride ZS
template <int ZS>
__device__ __forceinline__ f32x4 agg_row(const float* Zs, const BoardGraph& gr, int n, int c4) {
    const float* w = gr.w + n * 5;
    const unsigned char* nb = gr.nb + n * 4;
    f32x4 a = w[0] * ld4(Zs + n * ZS + c4);
#pragma unroll
    for (int d = 0; d < 4; ++d) a += w[1 + d] * ld4(Zs + (int)nb[d] * ZS + c4);
    return a;
}

// The B operand of a 128-deep contraction for 64 output columns, staged in LDS by coalesced 16-byte loads (the weights
// were rewritten by the previous step's Adam update: every launch finds them cold in its XCD's L2).
//   forward   B[k][c] = W[c0 + c][k]      -> image [c][k], stride SA: lane (c, k = 4 ks + q) reads conflict-free
//   dgrad     B[k][c] = W[k][c0 + c]      -> image [k][c], stride SD: idem
struct WTile {
    f32x4 v[8];
    template <bool FWD> __device__ __forceinline__ void issue(const float* __restrict__ W, int c0, int t) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = t + 256 * k;
            v[k] = FWD ? ld4(W + (size_t)(c0 + (i >> 5)) * TH + (i & 31) * 4) : ld4(W + (size_t)(i >> 4) * TH + c0 + (i & 15) * 4);
        }
    }
    template <bool FWD> __device__ __forceinline__ void land(float* Ws, int t) const {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = t + 256 * k;
            if (FWD) st4(Ws + (i >> 5) * SA + (i & 31) * 4, v[k]);
            else st4(Ws + (i >> 4) * SD + (i & 15) * 4, v[k]);
        }
    }
};

// acc[rt] += A[16 rt + r16][k] * B[k][16 wave + r16]  over k = 0..127; A an LDS image with stride SA, B a WTile image
template <int RT, bool FWD>
__device__ __forceinline__ void mfma_rows(f32x4 (&acc)[RT], const float* As, const float* Ws, int wave, int r16, int q) {
    const float* bp = FWD ? Ws + (16 * wave + r16) * SA + q : Ws + q * SD + 16 * wave + r16;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        const float bv = bp[FWD ? 4 * ks : 4 * ks * SD];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma4(As[(16 * rt + r16) * SA + 4 * ks + q], bv, acc[rt]);
    }
}

// The same contraction with the B operand held in registers (the fused per-board kernel: its LDS is full of activations):
// bw[ks] = W[(4 ks + q) * sk + col * sc], 32 strided dwords per lane, requested one phase ahead of their use.
__device__ __forceinline__ void load_bfrag(float (&bw)[32], const float* __restrict__ W, int sk, int sc, int col, int q) {
    const float* p = W + (size_t)col * sc + (size_t)q * sk;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) bw[ks] = p[(size_t)4 * ks * sk];
}
// Forward form (B[k][c] = W[c][k], a row of W per output column): 16-byte loads, lane (c, q) holds W[c][16 j + 4 q .. + 3], so the
// contraction index is enumerated as k = 16 j + 4 q + e and the A operand comes by ds_read_b128 (4 k values per lane; with
// the 132-float row stride the 16 rows of a quarter wave fall into 16 different 16-byte bank groups).  A quarter of the
// address-unit work of the dword form (16 segments per instruction either way, 8 instructions instead of 32).
__device__ __forceinline__ void load_bfrag4(f32x4 (&bv)[8], const float* __restrict__ W, int col, int q) {
    const float* p = W + (size_t)col * TH + 4 * q;
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = ld4(p + 16 * j);
}
template <int RT>
__device__ __forceinline__ void mfma_rows_reg4(f32x4 (&acc)[RT], const float* As, const f32x4 (&bv)[8], int r16, int q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f32x4 a[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = ld4(As + (16 * rt + r16) * SA + 16 * j + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma4(a[rt][e], bv[j][e], acc[rt]);
        }
    }
}
template <int RT>
__device__ __forceinline__ void mfma_rows_reg(f32x4 (&acc)[RT], const float* As, const float (&bw)[32], int r16, int q) {
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma4(As[(16 * rt + r16) * SA + 4 * ks + q], bw[ks], acc[rt]);
    }
}

// accumulator tiles -> LDS image [rows][64] (stride SZ); this wave's 16 columns start at 16 * wave
template <int RT>
__device__ __forceinline__ void store_acc(float* Zs, const f32x4 (&acc)[RT], int wave, int r16, int q) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Zs[(16 * rt + 4 * q + i) * SZ + 16 * wave + r16] = acc[rt][i];
    }
}

// rows [0, V) x 128 floats of a global [.][128] array -> LDS image with row stride S, rows [V, VZ) zero-filled.  Two
// halves so that a caller can put other work between the issue of the loads and the LDS writes.
template <int V, int VZ, int NT = 256> struct RowTile {
    static constexpr int IT = (VZ * 32 + NT - 1) / NT;
    f32x4 v[IT];
    __device__ __forceinline__ void issue(const float* __restrict__ src, int t) {
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = t + NT * k, n = i >> 5, c4 = (i & 31) * 4;
            v[k] = n < V ? ld4(src + (size_t)n * TH + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    template <int S> __device__ __forceinline__ void land(float* dst, int t) const {
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = t + NT * k, n = i >> 5, c4 = (i & 31) * 4;
            if (n < VZ) st4(dst + n * S + c4, v[k]);
        }
    }
};

__device__ __forceinline__ size_t record_of(const int64_t* __restrict__ order, int first, int b) {
    return order ? (size_t)order[first + b] : (size_t)(first + b);
}

// ---------------------------------------------------------------------------------------------
// forward, layers 1 + 2.   grid = 2 * B: workgroup = (board, column half)
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void train_fwd12_kernel(const uint8_t* __restrict__ states72, const int64_t* __restrict__ order, int first,
                                                          const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ W2, const float* __restrict__ b2,
                                                          float* __restrict__ h1, float* __restrict__ h2) {
    constexpr int V = N * N, RT = (V + 15) / 16;
    __shared__ float Zs[96 * SA];
    __shared__ float Hs[96 * SA];
    __shared__ float Ws[64 * SA];
    __shared__ BoardGraph gr;
    const int b = blockIdx.x >> 1, half = blockIdx.x & 1, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, q = lane >> 4, r16 = lane & 15;
    TS_DECL
    WTile wt;
    wt.issue<true>(W2, HH * half, t);
    board_graph<N>(gr, states72 + record_of(order, first, b) * STATE72, t);
    __syncthreads();
    TS(0, 0)
    // layer 1: Z1 = X0 W1^T, K = 6 padded to 8; this wave's column tiles are 2 wave, 2 wave + 1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 16 * (2 * wave + j) + r16;
        const float w_lo = W1[col * TF + q];                         // k = q      (0..3)
        const float w_hi = (q < 2) ? W1[col * TF + 4 + q] : 0.f;     // k = 4 + q  (4, 5; 6 and 7 are padding)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = mfma4(gr.x0[(16 * rt + r16) * 8 + q], w_lo, acc);
            acc = mfma4(gr.x0[(16 * rt + r16) * 8 + 4 + q], w_hi, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) Zs[(16 * rt + 4 * q + i) * SA + col] = acc[i];
        }
    }
    __syncthreads();
    TS(0, 1)
    {   // H1 = relu(A_hat Z1 + b1): all 128 columns into LDS (layer 2 contracts over them), this half to memory
        const int c4 = (t & 31) * 4;
        const f32x4 bias = ld4(b1 + c4);
        const bool mine = (c4 >> 6) == half;
#pragma unroll
        for (int it = 0; it < (V + 7) / 8; ++it) {
            const int n = (t >> 5) + 8 * it;
            if (n < V) {
                const f32x4 a = relu4(agg_row<SA>(Zs, gr, n, c4) + bias);
                st4(Hs + n * SA + c4, a);
                if (mine) st4(h1 + ((size_t)b * V + n) * TH + c4, a);
            }
        }
    }
    wt.land<true>(Ws, t);
    __syncthreads();
    TS(0, 2)
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    mfma_rows<RT, true>(acc, Hs, Ws, wave, r16, q);
    TS(0, 3)
    store_acc<RT>(Zs, acc, wave, r16, q);        // Z1 is dead since the barrier above
    __syncthreads();
    TS(0, 4)
    {
        const int c4 = (t & 15) * 4;
        const f32x4 bias = ld4(b2 + HH * half + c4);
#pragma unroll
        for (int it = 0; it < (V + 15) / 16; ++it) {
            const int n = (t >> 4) + 16 * it;
            if (n < V) st4(h2 + ((size_t)b * V + n) * TH + HH * half + c4, relu4(agg_row<SZ>(Zs, gr, n, c4) + bias));
        }
    }
    TS(0, 5)
}

// ---------------------------------------------------------------------------------------------
// forward, layer 3 + mean pool.   grid = 2 * B
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void train_fwd3_kernel(const uint8_t* __restrict__ states72, const int64_t* __restrict__ order, int first,
                                                         const float* __restrict__ W3, const float* __restrict__ b3,
                                                         const float* __restrict__ h2, float* __restrict__ h3, float* __restrict__ g) {
    constexpr int V = N * N, RT = (V + 15) / 16;
    __shared__ float Hs[96 * SA];
    __shared__ float Zs[96 * SZ];
    __shared__ float Ws[64 * SA];
    __shared__ BoardGraph gr;
    const int b = blockIdx.x >> 1, half = blockIdx.x & 1, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, q = lane >> 4, r16 = lane & 15;
    TS_DECL
    RowTile<V, V> hin;
    hin.issue(h2 + (size_t)b * V * TH, t);
    WTile wt;
    wt.issue<true>(W3, HH * half, t);
    board_graph<N>(gr, states72 + record_of(order, first, b) * STATE72, t);
    hin.template land<SA>(Hs, t);
    wt.land<true>(Ws, t);
    __syncthreads();
    TS(1, 0)
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    mfma_rows<RT, true>(acc, Hs, Ws, wave, r16, q);
    TS(1, 1)
    store_acc<RT>(Zs, acc, wave, r16, q);
    __syncthreads();
    TS(1, 2)
    float* cs = Hs;                              // H2 is dead: every wave finished its MFMAs before the barrier
    {
        const int c4 = (t & 15) * 4;
        const f32x4 bias = ld4(b3 + HH * half + c4);
        f32x4 colsum = {0.f, 0.f, 0.f, 0.f};     // this thread's rows (t >> 4, + 16, ...) in ascending order
#pragma unroll
        for (int it = 0; it < (V + 15) / 16; ++it) {
            const int n = (t >> 4) + 16 * it;
            if (n < V) {
                const f32x4 a = relu4(agg_row<SZ>(Zs, gr, n, c4) + bias);
                st4(h3 + ((size_t)b * V + n) * TH + HH * half + c4, a);
                colsum += a;
            }
        }
        st4(cs + (t >> 4) * HH + c4, colsum);
    }
    __syncthreads();
    TS(1, 3)
    if (t < HH) {                                // global_mean_pool: the 16 row-group sums in fixed order
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += cs[r * HH + t];
        g[(size_t)b * TH + HH * half + t] = s / (float)V;
    }
    TS(1, 4)
}

// ---------------------------------------------------------------------------------------------
// heads, losses and the way back to dg.   grid = B, one workgroup per position.
//   train_network.py:54,85: CrossEntropyLoss(policy_pred, policy_target) with policy_pred ALREADY softmaxed
//   (pv_network_gnn.py:42,62) and probability targets: l_b = -sum_a t_a log_softmax(pol)_a, mean over the batch
//   train_network.py:55,86: MSELoss(value_pred.squeeze(), value_target), mean over the batch
// Leaves: pol, val, loss terms; hp, hv (hidden layers); lg = d loss / d logits, vp = d loss / d pre-tanh value;
// dhp, dhv (gradients at the hidden layers, ReLU applied); dg.
// ---------------------------------------------------------------------------------------------
struct HeadParams { const float* p[8]; };        // state_dict tensors 6..13
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f(float old, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float x) {              // fixed order: quads, 8, 16, 32, 64 lanes
    x += dpp_f<0xB1, 0xf>(0.f, x);     // quad_perm [1,0,3,2]
    x += dpp_f<0x4E, 0xf>(0.f, x);     // quad_perm [2,3,0,1]
    x += dpp_f<0x141, 0xf>(0.f, x);    // row_half_mirror
    x += dpp_f<0x140, 0xf>(0.f, x);    // row_mirror: 16 lanes agree
    x += dpp_f<0x142, 0xa>(0.f, x);    // row_bcast15 -> rows 1, 3
    x += dpp_f<0x143, 0xc>(0.f, x);    // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
struct HeadsSmem { float gs[TH], hs[TH], dhs[TH], dl[256], red[2][8][2], part[4 * TH]; };
// One position's heads, losses and head gradients by a workgroup of NW wavefronts (the first four do the work; all take part
// in the barriers).  `sm.gs` = the pooled features if g == nullptr (the fused kernel has them in LDS already); on return
// sm.part[0..127] + sm.part[128..255] = dg, also stored to dg_out if that is not null.
template <int NW>
__device__ __forceinline__ void heads_board(HeadsSmem& sm, int b, const float* __restrict__ g, const HeadParams& Pm,
                                            const float* __restrict__ pi_all, const float* __restrict__ z_all,
                                            const int64_t* __restrict__ order, int first, int A, int B,
                                            float* __restrict__ hp, float* __restrict__ hv, float* __restrict__ lg,
                                            float* __restrict__ pol, float* __restrict__ vp, float* __restrict__ val,
                                            float* __restrict__ loss, float* __restrict__ dhp, float* __restrict__ dhv,
                                            float* __restrict__ dg) {
    float (&gs)[TH] = sm.gs; float (&hs)[TH] = sm.hs; float (&dhs)[TH] = sm.dhs; float (&dl)[256] = sm.dl;
    float (&red)[2][8][2] = sm.red; float (&part)[4 * TH] = sm.part;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool worker = t < 256;
    const float *Wp1 = Pm.p[0], *bp1 = Pm.p[1], *Wp2 = Pm.p[2], *bp2 = Pm.p[3], *Wv1 = Pm.p[4], *bv1 = Pm.p[5], *Wv2 = Pm.p[6], *bv2 = Pm.p[7];
    const size_t rec = record_of(order, first, b);
    int flip = 0;                                 // two exchange rows, used alternately: one barrier per reduction
    auto block_sum2 = [&](float& x, float& y) {
        x = wave_sum(x); y = wave_sum(y);
        if (lane == 0 && worker) { red[flip][wave][0] = x; red[flip][wave][1] = y; }
        __syncthreads();
        x = (red[flip][0][0] + red[flip][1][0]) + (red[flip][2][0] + red[flip][3][0]);      // waves 4.. hold zeros / -inf: not read
        y = (red[flip][0][1] + red[flip][1][1]) + (red[flip][2][1] + red[flip][3][1]);
        flip ^= 1;
    };
    auto block_max = [&](float v) { v = wave_max(v); if (lane == 0 && worker) red[flip][wave][0] = v; __syncthreads(); const float r = fmaxf(fmaxf(red[flip][0][0], red[flip][1][0]), fmaxf(red[flip][2][0], red[flip][3][0])); flip ^= 1; return r; };
    TS_DECL
    constexpr int HEADS_TS = NW == 4 ? 2 : 7;
    (void)HEADS_TS;
    // Every weight this workgroup multiplies by is requested up front, and by COALESCED loads: a wave takes whole rows of the
    // two first-layer matrices (512 B: two columns per lane) and of policy_head.2 (256 B: one column per lane) and reduces
    // each row's products with a DPP wave sum.  (One thread per output row -- 64 different rows per load instruction -- spent
    // 14 k cycles in the address unit before the first multiply.)
    const bool on = t < A;
    constexpr int HROWS = TH / NW;                                  // hidden units per wave (rows w, w + NW, ...)
    constexpr int LROWS = (256 + NW - 1) / NW;                      // logits per wave
    float wh0[HROWS], wh1[HROWS], wl[LROWS];
#pragma unroll
    for (int i = 0; i < HROWS; ++i) {
        const int o = wave + NW * i;
        const float* wr = (o < HH ? Wp1 + (size_t)o * TH : Wv1 + (size_t)(o - HH) * TH) + 2 * lane;
        wh0[i] = wr[0]; wh1[i] = wr[1];
    }
#pragma unroll
    for (int i = 0; i < LROWS; ++i) {
        const int a = wave + NW * i;
        wl[i] = a < A ? Wp2[(size_t)a * HH + lane] : 0.f;
    }
    constexpr int DROWS = 256 / NW;                                    // d loss / d policy hidden layer: the A terms dealt over the waves
    const int per = (A + NW - 1) / NW, a0 = wave * per;
    float wd[DROWS];
#pragma unroll
    for (int u = 0; u < DROWS; ++u) wd[u] = (u < per && a0 + u < A) ? Wp2[(size_t)(a0 + u) * HH + lane] : 0.f;
    const float tgt = on ? pi_all[rec * A + t] : 0.f;
    const float zt = z_all[rec];
    const float lb = on ? bp2[t] : 0.f, wv2 = Wv2[lane], bv = bv2[0];
    const float hbias = lane < HROWS ? ((wave + NW * lane) < HH ? bp1[wave + NW * lane] : bv1[wave + NW * lane - HH]) : 0.f;
    if (g && t < TH) gs[t] = g[(size_t)b * TH + t];
    __syncthreads();
    TS(HEADS_TS, 0)
    {   // hidden layers: hs[0..63] policy, hs[64..127] value
        const float g0 = gs[2 * lane], g1 = gs[2 * lane + 1];
#pragma unroll
        for (int i = 0; i < HROWS; ++i) {
            const int o = wave + NW * i;
            float s = wave_sum(fmaf(wh0[i], g0, wh1[i] * g1));
            s = fmaxf(s + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hbias), i)), 0.f);
            if (lane == 0) {
                hs[o] = s;
                (o < HH ? hp : hv)[(size_t)b * HH + (o & 63)] = s;
            }
        }
    }
    __syncthreads();
    TS(HEADS_TS, 1)
    {
        const float h = hs[lane];
#pragma unroll
        for (int i = 0; i < LROWS; ++i) {
            const int a = wave + NW * i;
            const float s = wave_sum(wl[i] * h);
            if (lane == 0 && a < A) dl[a] = s;                       // (dl is reused for d loss / d logits below)
        }
    }
    __syncthreads();
    const float l = on ? dl[t] + lb : -INFINITY;
    __syncthreads();                                                 // everybody holds its logit before dl is overwritten
    TS(HEADS_TS, 2)
    const float m = block_max(l);
    const float e = on ? expf(l - m) : 0.f;
    float se = e, tsum = tgt;
    block_sum2(se, tsum);
    const float p = e / se;                                      // first softmax (the network's own, pv_network_gnn.py:42)
    const float e2 = on ? expf(p) : 0.f;                         // second softmax inside CrossEntropyLoss; p in [0,1]: no shift needed
    float s2 = e2, vsum = wave == 0 ? wv2 * hs[HH + lane] : 0.f; // (the value head's 64-term dot product rides along)
    block_sum2(s2, vsum);
    const float qq = e2 / s2;
    const float dpol = on ? (qq * tsum - tgt) / (float)B : 0.f;  // d(mean_b l_b) / d pol
    float lp = on ? -tgt * (p - logf(s2)) : 0.f, dot = dpol * p;
    block_sum2(lp, dot);
    const float dlogit = on ? p * (dpol - dot) : 0.f;            // back through the first softmax
    if (worker) dl[t] = dlogit;
    if (on) {
        pol[(size_t)b * A + t] = p;
        lg[(size_t)b * A + t] = dlogit;
    }
    const float v = tanhf(vsum + bv);
    const float dv = v - zt;
    const float dvp = (2.f * dv / (float)B) * (1.f - v * v);
    if (t == 0) {
        val[b] = v;
        vp[b] = dvp;
        loss[2 * b] = lp;
        loss[2 * b + 1] = dv * dv;
    }
    __syncthreads();
    TS(HEADS_TS, 3)
    {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < DROWS; ++u) s = fmaf(dl[min(a0 + u, 255)], wd[u], s);
        part[wave * HH + lane] = s;
    }
    __syncthreads();
    TS(HEADS_TS, 4)
    if (t < TH) {
        const int j = t & 63;
        float s;
        if (t < HH) {
            s = (part[j] + part[HH + j]) + (part[2 * HH + j] + part[3 * HH + j]);
            if (NW == 8) s += (part[4 * HH + j] + part[5 * HH + j]) + (part[6 * HH + j] + part[7 * HH + j]);
        } else s = dvp * Wv2[j];
        if (!(hs[t] > 0.f)) s = 0.f;
        dhs[t] = s;
        (t < HH ? dhp : dhv)[(size_t)b * HH + j] = s;
    }
    __syncthreads();
    TS(HEADS_TS, 5)
    {   // dg = dhp W_p1 + dhv W_v1: threads 0..127 the policy part, 128..255 the value part
        const int k = t & 127, hsel = (t >> 7) & 1;
        const float* W = hsel ? Wv1 : Wp1;
        float s = 0.f;
#pragma unroll 16
        for (int j = 0; j < HH; ++j) s = fmaf(dhs[hsel * HH + j], W[(size_t)j * TH + k], s);
        if (worker) part[hsel * TH + k] = s;
    }
    __syncthreads();
    TS(HEADS_TS, 6)
    if (dg && t < TH) dg[(size_t)b * TH + t] = part[t] + part[TH + t];
}

__global__ __launch_bounds__(256) void train_heads_kernel(const float* __restrict__ g, HeadParams Pm,
                                                          const float* __restrict__ pi_all, const float* __restrict__ z_all,
                                                          const int64_t* __restrict__ order, int first, int A, int B,
                                                          float* __restrict__ hp, float* __restrict__ hv, float* __restrict__ lg,
                                                          float* __restrict__ pol, float* __restrict__ vp, float* __restrict__ val,
                                                          float* __restrict__ loss, float* __restrict__ dhp, float* __restrict__ dhv,
                                                          float* __restrict__ dg) {
    __shared__ HeadsSmem sm;
    heads_board<4>(sm, blockIdx.x, g, Pm, pi_all, z_all, order, first, A, B, hp, hv, lg, pol, vp, val, loss, dhp, dhv, dg);
}

// ---------------------------------------------------------------------------------------------
// backward through GCN layer L (3, 2, 1).   grid = 2 * B: workgroup = (board, column half of layer L's output)
//   L == 3: dH = dg / V on every node.        L < 3: dH = dZ_{L+1} W_{L+1}  (this half of the columns)
//   dP = dH (.) [H_L > 0];  db partial = column sums;  dZ = A_hat dP -> dZout (L > 1: the next launch contracts over it)
//   dW partial [64 rows of W_L][K] = dZ^T H_{L-1}   (K = 128 on the matrix pipe; K = 6 for layer 1 on the VALU)
// ---------------------------------------------------------------------------------------------
template <int N, int L>
__global__ __launch_bounds__(256) void train_bwd_kernel(const uint8_t* __restrict__ states72, const int64_t* __restrict__ order, int first,
                                                        const float* __restrict__ dg, const float* __restrict__ dZin,
                                                        const float* __restrict__ Wnext, const float* __restrict__ Hout,
                                                        const float* __restrict__ Hin, float* __restrict__ dZout,
                                                        float* __restrict__ part_dW, float* __restrict__ part_db) {
    constexpr int V = N * N, RT = (V + 15) / 16, VK = (V + 3) / 4 * 4;
    constexpr int UN = 96 * SZ + 84 * SD > 96 * SA ? 96 * SZ + 84 * SD : 96 * SA;
    __shared__ float U[UN];                      // first dZin as an A operand, then dP (stride SZ) and dZ (stride SD)
    __shared__ float Hb[L > 1 ? 84 * SB : 4];    // H_{L-1}, the B operand of the weight gradient
    __shared__ float Ws[L < 3 ? 128 * SD : 4];   // W_{L+1}, the B operand of the data gradient
    __shared__ float cs[4 * HH];
    __shared__ BoardGraph gr;
    float* dPs = U;
    float* dZs = U + 96 * SZ;
    const int b = blockIdx.x >> 1, half = blockIdx.x & 1, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, q = lane >> 4, r16 = lane & 15;
    const int col = HH * half + 16 * wave + r16;
    TS_DECL
    f32x4 acc[RT];
    float hm[RT][4];                             // H_L in the accumulator layout (lanes = 16 consecutive columns): the ReLU mask
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 16 * rt + 4 * q + i;
            hm[rt][i] = n < V ? Hout[((size_t)b * V + n) * TH + col] : 0.f;
        }
    }
    RowTile<V, VK> hin;                          // H_{L-1}: only the weight gradient at the very end needs it -- requested
    if (L < 3) {                                 // last, landed after the data-gradient MFMAs
        RowTile<V, V> zin;
        zin.issue(dZin + (size_t)b * V * TH, t);
        WTile wt;
        wt.issue<false>(Wnext, HH * half, t);
        if (L > 1) hin.issue(Hin + (size_t)b * V * TH, t);
        board_graph<N>(gr, states72 + record_of(order, first, b) * STATE72, t);
        zin.template land<SA>(U, t);
        wt.land<false>(Ws, t);
        __syncthreads();
        TS(6 - L, 0)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        mfma_rows<RT, false>(acc, U, Ws, wave, r16, q);
        if (L > 1) hin.template land<SB>(Hb, t);
        TS(6 - L, 1)
    } else {
        hin.issue(Hin + (size_t)b * V * TH, t);
        board_graph<N>(gr, states72 + record_of(order, first, b) * STATE72, t);
        const float v = dg[(size_t)b * TH + col] / (float)V;      // global_mean_pool backward
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{v, v, v, v};
    }
    float dbp = 0.f;                             // this lane's share of the bias gradient: its rows in ascending order
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!(hm[rt][i] > 0.f)) acc[rt][i] = 0.f;             // rows >= V carry hm = 0
            dbp += acc[rt][i];
        }
    }
    __syncthreads();                             // every wave is done with the A operand in U
    TS(6 - L, 2)
    store_acc<RT>(dPs, acc, wave, r16, q);
    cs[q * HH + 16 * wave + r16] = dbp;
    for (int i = t; i < (VK - V) * 16; i += 256) st4(dZs + (V + (i >> 4)) * SD + (i & 15) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    __syncthreads();
    TS(6 - L, 3)
    {
        const int c4 = (t & 15) * 4;
#pragma unroll
        for (int it = 0; it < (V + 15) / 16; ++it) {
            const int n = (t >> 4) + 16 * it;
            if (n < V) {
                const f32x4 a = agg_row<SZ>(dPs, gr, n, c4);       // dZ = A_hat dP (A_hat is symmetric)
                st4(dZs + n * SD + c4, a);
                if (L > 1) st4(dZout + ((size_t)b * V + n) * TH + HH * half + c4, a);
            }
        }
        if (t < HH) part_db[(size_t)b * TH + HH * half + t] = (cs[t] + cs[HH + t]) + (cs[2 * HH + t] + cs[3 * HH + t]);
    }
    if (L == 3) hin.template land<SB>(Hb, t);
    __syncthreads();
    TS(6 - L, 4)
    // dW[64 half + 16 wave + ..][k] = sum_n dZ[n][j] H_{L-1}[n][k]: this wave's 16 rows of W_L
    if (L > 1) {
        f32x4 wacc[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) wacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < VK / 4; ++ks) {
            const float a = dZs[(4 * ks + q) * SD + 16 * wave + r16];
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) wacc[ct] = mfma4(a, Hb[(4 * ks + q) * SB + 16 * ct + r16], wacc[ct]);
        }
        float* dst = part_dW + (size_t)b * TH * TH + (size_t)(HH * half + 16 * wave + 4 * q) * TH + r16;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(size_t)i * TH + 16 * ct] = wacc[ct][i];
        }
    } else {
        // layer 1: H_0 = the six features (columns 6..15 of the tile are padding)
        f32x4 wacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < VK / 4; ++ks)
            wacc = mfma4(dZs[(4 * ks + q) * SD + 16 * wave + r16], r16 < 8 ? gr.x0[(4 * ks + q) * 8 + r16] : 0.f, wacc);
        if (r16 < TF) {
            float* dst = part_dW + (size_t)b * TH * TF + (size_t)(HH * half + 16 * wave + 4 * q) * TF + r16;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i * TF] = wacc[i];
        }
    }
    TS(6 - L, 5)
}

// ---------------------------------------------------------------------------------------------
// The whole forward + backward of ONE position in one workgroup (8 wavefronts; wave w owns feature columns 16 w .. 16 w + 15,
// and rows 16 w .. of the weight gradients).  Nothing but the per-board partial gradients leaves the CU: the activations H1,
// H2 go to memory once and come back through the L2 of the same XCD, H3 never leaves LDS.  grid = B.
// LDS: Hs (A operand: H_l, then dZ_l), Zs (accumulator images; the heads' scratch), Hb (H_{l-1} as the B operand of the
// weight gradient and as the ReLU mask of the next layer down).
// ---------------------------------------------------------------------------------------------
struct TrunkParams { const float* p[6]; };       // state_dict tensors 0..5
template <int N>
__global__ __launch_bounds__(512) void train_board_kernel(const uint8_t* __restrict__ states72, const int64_t* __restrict__ order, int first,
                                                          TrunkParams tp, HeadParams hpm, const float* __restrict__ pi_all,
                                                          const float* __restrict__ z_all, int A, int B,
                                                          float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ g_out,
                                                          float* __restrict__ hp, float* __restrict__ hv, float* __restrict__ lg,
                                                          float* __restrict__ pol, float* __restrict__ vp, float* __restrict__ val,
                                                          float* __restrict__ loss, float* __restrict__ dhp, float* __restrict__ dhv,
                                                          float* __restrict__ part_dW3, float* __restrict__ part_dW2,
                                                          float* __restrict__ part_dW1, float* __restrict__ part_db) {
    constexpr int V = N * N, RT = (V + 15) / 16, VK = (V + 3) / 4 * 4, NIT = (V + 15) / 16;
    __shared__ float Hs[96 * SA];
    __shared__ float Zs[96 * SA];
    __shared__ float Hb[84 * SB];
    __shared__ float cs[4 * TH];
    __shared__ BoardGraph gr;
    HeadsSmem& hsm = *reinterpret_cast<HeadsSmem*>(Zs);
    static_assert(sizeof(HeadsSmem) <= sizeof(float) * 96 * SA && 16 * TH <= 84 * SB, "scratch aliases");
    const int b = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, q = lane >> 4, r16 = lane & 15;
    const int col = 16 * wave + r16;
    const int c4 = (t & 31) * 4, rg = t >> 5;                       // aggregation mapping: 32 float4 per row x 16 row groups
    const float *W1 = tp.p[0], *b1 = tp.p[1], *W2 = tp.p[2], *b2 = tp.p[3], *W3 = tp.p[4], *b3 = tp.p[5];
    float bw[32];
    f32x4 bv4[8];
    f32x4 acc[RT];
    const f32x4 bias1 = ld4(b1 + c4), bias2 = ld4(b2 + c4), bias3 = ld4(b3 + c4);   // (ahead of the weight fragments in the load queue)
    auto zero_acc = [&]() {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto acc_to_Zs = [&]() {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Zs[(16 * rt + 4 * q + i) * SA + col] = acc[rt][i];
        }
    };
    TS_DECL
    board_graph<N>(gr, states72 + record_of(order, first, b) * STATE72, t);
    load_bfrag4(bv4, W2, col, q);
    // Every weight this kernel will read was rewritten by the previous step's Adam update and is cold in this XCD's L2.  One
    // load per 64-byte line pulls W3 and the heads' matrices in now, under the graph setup and layer 1, instead of in front
    // of the phases that need them (the values are summed into `warm_sink`, which is never equal to its magic number).
    float warm[6];
    {
        const int l16 = t * 16;
        warm[0] = W3[l16]; warm[1] = W3[l16 + 512 * 16];
        warm[2] = hpm.p[0][l16]; warm[3] = hpm.p[4][l16];
        warm[4] = l16 < A * HH ? hpm.p[2][l16] : 0.f; warm[5] = l16 + 512 * 16 < A * HH ? hpm.p[2][l16 + 512 * 16] : 0.f;
    }
    for (int i = t; i < (96 - V) * 32; i += 512) st4(Hs + (V + (i >> 5)) * SA + (i & 31) * 4, f32x4{0.f, 0.f, 0.f, 0.f});   // rows V..95: zero for good
                                                                     // (the padding rows of every contraction over the nodes)
    __syncthreads();
    TS(8, 0)
    // ---- forward, layer 1 (K = 6 padded to 8)
    {
        const float w_lo = W1[col * TF + q];
        const float w_hi = (q < 2) ? W1[col * TF + 4 + q] : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mfma4(gr.x0[(16 * rt + r16) * 8 + q], w_lo, a);
            a = mfma4(gr.x0[(16 * rt + r16) * 8 + 4 + q], w_hi, a);
            acc[rt] = a;
        }
        acc_to_Zs();
    }
    __syncthreads();
    auto aggregate_relu = [&](const f32x4 bv, float* __restrict__ hglob) {          // Zs -> Hs (+ memory)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = rg + 16 * it;
            if (n < V) {
                const f32x4 a = relu4(agg_row<SA>(Zs, gr, n, c4) + bv);
                st4(Hs + n * SA + c4, a);
                if (hglob) st4(hglob + ((size_t)b * V + n) * TH + c4, a);
            }
        }
    };
    aggregate_relu(bias1, h1);
    const float warm_sink = ((warm[0] + warm[1]) + (warm[2] + warm[3])) + (warm[4] + warm[5]);
    __syncthreads();
    TS(8, 1)
    // ---- layer 2
    zero_acc();
    mfma_rows_reg4<RT>(acc, Hs, bv4, r16, q);
    TS(8, 2)
    load_bfrag4(bv4, W3, col, q);
    acc_to_Zs();
    __syncthreads();
    aggregate_relu(bias2, h2);
    __syncthreads();
    TS(8, 3)
    // ---- layer 3 + mean pool (H3 stays in LDS: the backward needs only its sign)
    zero_acc();
    mfma_rows_reg4<RT>(acc, Hs, bv4, r16, q);
    TS(8, 4)
    acc_to_Zs();
    __syncthreads();
    {
        const f32x4 bv = bias3;
        f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = rg + 16 * it;
            if (n < V) {
                const f32x4 a = relu4(agg_row<SA>(Zs, gr, n, c4) + bv);
                st4(Hs + n * SA + c4, a);
                colsum += a;
            }
        }
        st4(Hb + rg * TH + c4, colsum);                              // (Hb is free until the backward loads H2 into it)
    }
    __syncthreads();
    if (t < TH) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += Hb[r * TH + t];
        s /= (float)V;                                               // global_mean_pool
        hsm.gs[t] = s;
        g_out[(size_t)b * TH + t] = s;                               // (the head weight gradients are batch dot products with it)
    }
    TS(8, 5)
    // ---- heads, losses, head gradients (its first barrier publishes gs)
    heads_board<8>(hsm, b, nullptr, hpm, pi_all, z_all, order, first, A, B, hp, hv, lg, pol, vp, val, loss, dhp, dhv, nullptr);
    TS(8, 6)
    load_bfrag(bw, W3, TH, 1, col, q);                               // the data gradient's fragments of W3 (B[j][k] = W3[j][k]): land under layer 3's backward
    // ---- backward.  One layer: dP (accumulator layout) -> Zs;  dZ = A_hat dP -> Hs;  dW partial = dZ^T H_{l-1} (Hb)
    RowTile<V, VK, 512> hin;
    auto mask_and_bias_grad = [&](const float* M, int stride) -> float {   // acc (.)= [M > 0]; returns this lane's column sum
        float dbp = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = 16 * rt + 4 * q + i;
                if (!(n < V && M[n * stride + col] > 0.f)) acc[rt][i] = 0.f;
                dbp += acc[rt][i];
            }
        }
        return dbp;
    };
    auto finish_layer = [&](float dbp, const float* __restrict__ hprev, float* __restrict__ pdb) {
        // callers have passed a barrier since the last read of Zs / of Hs as an A operand / of Hb as a mask
        acc_to_Zs();
        cs[q * TH + col] = dbp;
        if (hprev) hin.issue(hprev + (size_t)b * V * TH, t);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = rg + 16 * it;
            if (n < V) st4(Hs + n * SA + c4, agg_row<SA>(Zs, gr, n, c4));                 // dZ = A_hat dP (A_hat is symmetric)
        }
        if (t < TH) pdb[(size_t)b * TH + t] = (cs[t] + cs[TH + t]) + (cs[2 * TH + t] + cs[3 * TH + t]);
        if (hprev) hin.template land<SB>(Hb, t);
        __syncthreads();
    };
    auto weight_grad = [&](float* __restrict__ pdW) {                  // rows 16 wave .. of W_l, all 128 columns
        f32x4 wacc[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) wacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < VK / 4; ++ks) {
            const float a = Hs[(4 * ks + q) * SA + 16 * wave + r16];
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) wacc[ct] = mfma4(a, Hb[(4 * ks + q) * SB + 16 * ct + r16], wacc[ct]);
        }
        float* dst = pdW + (size_t)b * TH * TH + (size_t)(16 * wave + 4 * q) * TH + r16;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(size_t)i * TH + 16 * ct] = wacc[ct][i];
        }
    };
    // layer 3: dH3 = dg / V on every node (global_mean_pool backward); the mask is H3, still in Hs
    {
        const float v = (hsm.part[col] + hsm.part[TH + col]) / (float)V;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{v, v, v, v};
        const float dbp = mask_and_bias_grad(Hs, SA);
        __syncthreads();                                             // everybody has read dg (in Zs) and H3 (in Hs)
        finish_layer(dbp, h2, part_db + (size_t)2 * B * TH);
        TS(8, 7)
        weight_grad(part_dW3);
        TS(8, 8)
    }
    // layer 2: dH2 = dZ3 W3, mask H2 (in Hb)
    {
        zero_acc();
        mfma_rows_reg<RT>(acc, Hs, bw, r16, q);
        TS(8, 9)
        load_bfrag(bw, W2, TH, 1, col, q);
        const float dbp = mask_and_bias_grad(Hb, SB);
        __syncthreads();                                             // dZ3 (Hs) and H2 (Hb) are dead
        finish_layer(dbp, h1, part_db + (size_t)B * TH);
        TS(8, 10)
        weight_grad(part_dW2);
        TS(8, 11)
    }
    // layer 1: dH1 = dZ2 W2, mask H1 (in Hb); dW1 = dZ1^T X0 (six feature columns of one padded tile)
    {
        zero_acc();
        mfma_rows_reg<RT>(acc, Hs, bw, r16, q);
        TS(8, 12)
        const float dbp = mask_and_bias_grad(Hb, SB);
        __syncthreads();
        finish_layer(dbp, nullptr, part_db);
        TS(8, 13)
        f32x4 wacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < VK / 4; ++ks)
            wacc = mfma4(Hs[(4 * ks + q) * SA + 16 * wave + r16], r16 < 8 ? gr.x0[(4 * ks + q) * 8 + r16] : 0.f, wacc);
        if (r16 < TF) {
            float* dst = part_dW1 + (size_t)b * TH * TF + (size_t)(16 * wave + 4 * q) * TF + r16;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i * TF] = wacc[i];
        }
        TS(8, 14)
    }
    if (warm_sink == -1.2345678e-31f) part_db[0] = warm_sink;          // (keeps the warm-up loads alive; never taken)
}

// ---------------------------------------------------------------------------------------------
// gradient of every parameter element + its Adam update.   One thread per element of the 14 tensors.
// parameter order = state_dict order (KEYS in INTEGRATION.md):
//  0 gcn0.w [H,F]  1 gcn0.b  2 gcn1.w [H,H]  3 gcn1.b  4 gcn2.w  5 gcn2.b
//  6 pol0.w [H/2,H]  7 pol0.b  8 pol2.w [A,H/2]  9 pol2.b  10 val0.w [H/2,H]  11 val0.b  12 val2.w [1,H/2]  13 val2.b
// torch.optim.Adam.step() (no weight decay, no amsgrad); bias corrections computed on the host in f64.
// ---------------------------------------------------------------------------------------------
struct FinalJobs {
    float* p[14]; float* g[14]; float* m[14]; float* v[14];
    unsigned int end[14];                        // running element count after tensor i
    const float* part_dW[3]; const float* part_db[3];
    const float *dlg, *dvp, *hp, *hv, *dhp, *dhv, *gp, *loss;
    float* loss_sums;                            // optional: += the two batch-mean losses (elements end[13], end[13] + 1)
    int B, A, compute, update;
    float lr, beta1, beta2, eps, bc1, bc2_sqrt;
};
__device__ __forceinline__ void adam_update(const FinalJobs& jb, int i, unsigned int e, float gr) {
    const float mi = jb.beta1 * jb.m[i][e] + (1.f - jb.beta1) * gr;          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = jb.beta2 * jb.v[i][e] + (1.f - jb.beta2) * gr * gr;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    jb.m[i][e] = mi; jb.v[i][e] = vi;
    const float denom = sqrtf(vi) / jb.bc2_sqrt + jb.eps;
    jb.p[i][e] -= (jb.lr / jb.bc1) * (mi / denom);
}
// A workgroup = 32 lanes x 8 board groups: a thread sums its group's boards in order, the 8 group sums are added in group
// order -- a fixed summation order with 8x the loads in flight of one thread per element.  The first FINAL_BIG_BLOCKS
// workgroups take the two [128,128] trunk weights four elements per lane (16-byte loads of the 16 MB of per-board
// partials); the rest take every other tensor one element per lane (end[] counts those tensors only).
constexpr int FINAL_BIG_BLOCKS = 2 * TH * TH / 128;
__global__ __launch_bounds__(256) void train_final_kernel(FinalJobs jb) {
    __shared__ f32x4 red4[8][33];
    TS_DECL
    const int le = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int B = jb.B, A = jb.A;
    const int per = (B + 7) / 8, b0 = grp * per, b1 = min(B, b0 + per);
    if (blockIdx.x < FINAL_BIG_BLOCKS) {
        const unsigned int q4 = blockIdx.x * 32 + le;             // float4 index over gcn1.w then gcn2.w
        const int i = q4 < TH * TH / 4 ? 2 : 4;
        const unsigned int e = (q4 & (TH * TH / 4 - 1)) * 4;
        f32x4 gr4;
        if (jb.compute) {
            const float* src = jb.part_dW[i >> 1] + e;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int b = b0; b < b1; ++b) s += ld4(src + (size_t)b * TH * TH);
            red4[grp][le] = s;
            __syncthreads();
            TS(6, 1)
            if (grp != 0) return;
            gr4 = ((red4[0][le] + red4[1][le]) + (red4[2][le] + red4[3][le])) + ((red4[4][le] + red4[5][le]) + (red4[6][le] + red4[7][le]));
            st4(jb.g[i] + e, gr4);
        } else {
            if (grp != 0) return;
            gr4 = ld4(jb.g[i] + e);
        }
        if (jb.update) {
#pragma unroll
            for (int k = 0; k < 4; ++k) adam_update(jb, i, e + k, gr4[k]);
        }
        return;
    }
    float (*red)[33] = reinterpret_cast<float (*)[33]>(&red4[0][0]);
    const unsigned int e0 = (blockIdx.x - FINAL_BIG_BLOCKS) * 32 + le;
    const unsigned int total = jb.end[13] + (jb.loss_sums ? 2u : 0u);
    const bool live = e0 < total;
    int i = 0;
    if (live) while (i < 14 && e0 >= jb.end[i]) ++i;
    const unsigned int e = e0 - (i ? jb.end[i - 1] : 0u);
    TS(6, 0)
    if (jb.compute) {
        float s = 0.f;
        if (!live) {
        } else if (i < 6) {
            if (i & 1) { const float* src = jb.part_db[i >> 1] + e; for (int b = b0; b < b1; ++b) s += src[(size_t)b * TH]; }
            else { const float* src = jb.part_dW[0] + e; for (int b = b0; b < b1; ++b) s += src[(size_t)b * TH * TF]; }       // gcn0.w
        } else if (i == 6 || i == 10) {
            const int j = e / TH, k = e % TH;
            const float* d = (i == 6 ? jb.dhp : jb.dhv) + j;
            const float* x = jb.gp + k;
#pragma unroll 8
            for (int b = b0; b < b1; ++b) s = fmaf(d[(size_t)b * HH], x[(size_t)b * TH], s);
        } else if (i == 7 || i == 11) {
            const float* d = (i == 7 ? jb.dhp : jb.dhv) + e;
            for (int b = b0; b < b1; ++b) s += d[(size_t)b * HH];
        } else if (i == 8) {
            const int a = e / HH, j = e % HH;
#pragma unroll 8
            for (int b = b0; b < b1; ++b) s = fmaf(jb.dlg[(size_t)b * A + a], jb.hp[(size_t)b * HH + j], s);
        } else if (i == 9) {
            for (int b = b0; b < b1; ++b) s += jb.dlg[(size_t)b * A + e];
        } else if (i == 12) {
            for (int b = b0; b < b1; ++b) s = fmaf(jb.dvp[b], jb.hv[(size_t)b * HH + e], s);
        } else if (i == 13) {
            for (int b = b0; b < b1; ++b) s += jb.dvp[b];
        } else {
            for (int b = b0; b < b1; ++b) s += jb.loss[2 * b + e];
        }
        red[grp][le] = s;
    }
    __syncthreads();
    if (grp != 0 || !live) return;
    float gr;
    if (jb.compute) {
        gr = ((red[0][le] + red[1][le]) + (red[2][le] + red[3][le])) + ((red[4][le] + red[5][le]) + (red[6][le] + red[7][le]));
        if (i == 14) { jb.loss_sums[e] += gr / (float)B; return; }
        jb.g[i][e] = gr;
    } else {
        if (i == 14) return;
        gr = jb.g[i][e];
    }
    if (jb.update) adam_update(jb, i, e, gr);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int g_train_fused = 1;    // aqg_set_option("train_fused"): 1 = one workgroup per position for the whole forward + backward, 0 = six launches

template <int N>
static void launch_forward_backward(const aqg_train& t, const uint8_t* states72, const float* pi, const float* z, const int64_t* order,
                                    int first, int B, hipStream_t st) {
    const int A = t.policy_size;
    float* const* P = t.params;
    if (g_train_fused) {
        float* pdW3 = t.part;
        float* pdW2 = pdW3 + (size_t)B * TH * TH;
        float* pdW1 = pdW2 + (size_t)B * TH * TH;
        float* pdb = pdW1 + (size_t)B * TH * TF;
        TrunkParams tpm;
        for (int i = 0; i < 6; ++i) tpm.p[i] = P[i];
        HeadParams hpm;
        for (int i = 0; i < 8; ++i) hpm.p[i] = P[6 + i];
        hipLaunchKernelGGL(train_board_kernel<N>, dim3(B), dim3(512), 0, st, states72, order, first, tpm, hpm, pi, z, A, B, t.h1, t.h2, t.g,
                           t.hp, t.hv, t.lg, t.pol, t.vp, t.val, t.loss, t.dhp, t.dhv, pdW3, pdW2, pdW1, pdb);
        return;
    }
    const dim3 grid(2 * B), block(256);
    float* pdW3 = t.part;
    float* pdW2 = pdW3 + (size_t)B * TH * TH;
    float* pdW1 = pdW2 + (size_t)B * TH * TH;
    float* pdb = pdW1 + (size_t)B * TH * TF;                        // [3][B][128]: layer 1, 2, 3
    hipLaunchKernelGGL(train_fwd12_kernel<N>, grid, block, 0, st, states72, order, first, (const float*)P[0], (const float*)P[1],
                       (const float*)P[2], (const float*)P[3], t.h1, t.h2);
    hipLaunchKernelGGL(train_fwd3_kernel<N>, grid, block, 0, st, states72, order, first, (const float*)P[4], (const float*)P[5],
                       (const float*)t.h2, t.h3, t.g);
    HeadParams hpm;
    for (int i = 0; i < 8; ++i) hpm.p[i] = P[6 + i];
    hipLaunchKernelGGL(train_heads_kernel, dim3(B), block, 0, st, (const float*)t.g, hpm, pi, z, order, first, A, B,
                       t.hp, t.hv, t.lg, t.pol, t.vp, t.val, t.loss, t.dhp, t.dhv, t.dg);
    hipLaunchKernelGGL((train_bwd_kernel<N, 3>), grid, block, 0, st, states72, order, first, (const float*)t.dg, (const float*)nullptr,
                       (const float*)nullptr, (const float*)t.h3, (const float*)t.h2, t.zbuf, pdW3, pdb + (size_t)2 * B * TH);
    hipLaunchKernelGGL((train_bwd_kernel<N, 2>), grid, block, 0, st, states72, order, first, (const float*)nullptr, (const float*)t.zbuf,
                       (const float*)P[4], (const float*)t.h2, (const float*)t.h1, t.dh, pdW2, pdb + (size_t)B * TH);
    hipLaunchKernelGGL((train_bwd_kernel<N, 1>), grid, block, 0, st, states72, order, first, (const float*)nullptr, (const float*)t.dh,
                       (const float*)P[2], (const float*)t.h1, (const float*)nullptr, (float*)nullptr, pdW1, pdb);
}

static int launch_final(const aqg_train& t, int B, bool compute, bool update, int step, float* loss_sums, hipStream_t st) {
    const int A = t.policy_size;
    const size_t sizes[14] = {(size_t)TH * TF, TH, (size_t)TH * TH, TH, (size_t)TH * TH, TH, (size_t)HH * TH, (size_t)HH, (size_t)A * HH, (size_t)A,
                              (size_t)HH * TH, (size_t)HH, (size_t)HH, 1};
    FinalJobs jb{};
    unsigned int run = 0;
    for (int i = 0; i < 14; ++i) {
        jb.p[i] = t.params[i]; jb.g[i] = t.grads[i]; jb.m[i] = t.adam_m[i]; jb.v[i] = t.adam_v[i];
        if (i != 2 && i != 4) run += (unsigned int)sizes[i];      // the two big trunk weights have their own workgroups
        jb.end[i] = run;
    }
    const float* pdW3 = t.part;
    const float* pdW2 = pdW3 + (size_t)B * TH * TH;
    const float* pdW1 = pdW2 + (size_t)B * TH * TH;
    const float* pdb = pdW1 + (size_t)B * TH * TF;
    jb.part_dW[0] = pdW1; jb.part_dW[1] = pdW2; jb.part_dW[2] = pdW3;
    jb.part_db[0] = pdb; jb.part_db[1] = pdb + (size_t)B * TH; jb.part_db[2] = pdb + (size_t)2 * B * TH;
    jb.dlg = t.lg; jb.dvp = t.vp; jb.hp = t.hp; jb.hv = t.hv; jb.dhp = t.dhp; jb.dhv = t.dhv; jb.gp = t.g; jb.loss = t.loss;
    jb.loss_sums = compute ? loss_sums : nullptr;
    jb.B = B; jb.A = A; jb.compute = compute; jb.update = update;
    const double bc1 = 1.0 - pow((double)t.beta1, (double)step), bc2 = 1.0 - pow((double)t.beta2, (double)step);
    jb.lr = t.lr; jb.beta1 = t.beta1; jb.beta2 = t.beta2; jb.eps = t.eps; jb.bc1 = (float)bc1; jb.bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(train_final_kernel, dim3(FINAL_BIG_BLOCKS + (run + 2 + 31) / 32), dim3(256), 0, st, jb);
    return check_launch("train_final_kernel");
}

static int forward_backward(const aqg_train& t, const uint8_t* states72, const float* pi, const float* z, const int64_t* order, int first,
                            int B, hipStream_t st) {
    switch (t.board_size) {
        case 3: launch_forward_backward<3>(t, states72, pi, z, order, first, B, st); break;
        case 5: launch_forward_backward<5>(t, states72, pi, z, order, first, B, st); break;
        case 7: launch_forward_backward<7>(t, states72, pi, z, order, first, B, st); break;
        default: launch_forward_backward<9>(t, states72, pi, z, order, first, B, st); break;
    }
    return check_launch("training forward/backward kernels");
}

static int validate(const aqg_train& t) {
    const int N = t.board_size, A = t.policy_size;
    if (!(N == 3 || N == 5 || N == 7 || N == 9)) return fail("board_size must be 3, 5, 7 or 9");
    if (A != N * N + 2 * (N - 1) * (N - 1) || A > 256) return fail("policy_size does not match the board");
    return 0;
}

// mode 0 = gradients only, 1 = gradients + Adam, 2 = Adam only (data-parallel: local gradients, all-reduce, update)
int train_step(const aqg_train& t, const uint8_t* states72, const float* pi, const float* z, int mode, hipStream_t st) {
    if (int r = validate(t)) return r;
    const int B = t.batch;
    if (mode != 2 && B > 0) {
        if (int r = forward_backward(t, states72, pi, z, nullptr, 0, B, st)) return r;
        return launch_final(t, B, true, mode == 1, t.step, nullptr, st);
    }
    if (mode >= 1) return launch_final(t, B, false, true, t.step, nullptr, st);
    return 0;
}

// A run of consecutive single-process steps over a shuffled data set, no host work in between: step i takes the positions
// order[i * batch .. (i + 1) * batch) (the last batch may be short, train_network.py's DataLoader keeps it) of the
// resident arrays, t.step counts up from its entry value, and each step's loss terms are added to loss_sums[2]
// (policy, value: the per-step batch means, what train_network.py:89-90 accumulates per epoch).
int train_steps(const aqg_train& t, const uint8_t* states72, const float* pi, const float* z, const int64_t* order, long long positions,
                float* loss_sums, hipStream_t st) {
    if (int r = validate(t)) return r;
    if (t.batch < 1) return fail("aqg_gcn_train_steps: batch must be >= 1");
    int step = t.step;
    for (long long first = 0; first < positions; first += t.batch, ++step) {
        const int B = (int)(positions - first < t.batch ? positions - first : t.batch);
        if (int r = forward_backward(t, states72, pi, z, order, (int)first, B, st)) return r;
        if (int r = launch_final(t, B, true, true, step, loss_sums, st)) return r;
    }
    return 0;
}

#ifdef AQG_STAMP
extern "C" int aqg_debug_train_stamps(unsigned long long* out_host, int reset) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_train_stamp), sizeof(unsigned long long) * 160) != hipSuccess) return -1;
    if (reset) { unsigned long long z[160] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_train_stamp), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

}  // namespace aqg
