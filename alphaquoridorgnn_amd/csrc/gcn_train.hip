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
// The batch is 128 positions (train_network.py:15) = 10,368 graph nodes and 2.2 GFLOP per step.  Three forms of the step
// (aqg_set_option("train_fused")), all within 2e-5 max|g| of fp64 autograd:
//
//   2, split (default on 9x9): train_board_split_kernel -- ONE 8-wave workgroup per position does forward, heads + losses and
//     backward with every contraction on the 16-bit matrix pipe in fp16 hi/lo split precision (split_mfma.hpp; three fp16 products
//     per f32 product, f32 accumulation: fp32-equivalent), the neighbourhood aggregation included (banded A_hat blocks as MFMA
//     operands).  A position whose values leave fp16 range is redone by the f32 body in the same launch.  Then train_final_kernel.
//   1, fused f32: train_board_kernel -- the same one-workgroup-per-position structure on the f32-input matrix pipe
//     (v_mfma_f32_16x16x4_f32: exact f32 products), aggregation as a VALU gather over LDS; the default on 3x3 / 5x5 / 7x7.
//   0, six launches + final: two workgroups per board split the 128 feature columns (all 256 CUs busy at batch 128) and exchange
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
#include "split_mfma.hpp"
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

#ifdef AQG_TRAIN_DEBUG      // developer build only (tools/train_debug.py): dense dumps of intermediate gradients, [slot][b][96][128]
__device__ float* g_train_dbg = nullptr;
#define DBG_PUT(slot, B_, b_, n_, col_, v_) { if (g_train_dbg) g_train_dbg[(((size_t)(slot) * (B_) + (b_)) * 96 + (n_)) * 128 + (col_)] = (v_); }
#else
#define DBG_PUT(slot, B_, b_, n_, col_, v_)
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
struct HeadsSmem {
    float gs[TH], hs[TH], dhs[TH], dl[256], red[2][8][2];
    float dgv[TH];                       // d loss / d pooled features: the result the backward pass starts from
    alignas(16) float part[32][TH];      // per (wave, row group): partial sums over that group's weight rows
};
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]))); }
__device__ __forceinline__ float row16_total(float x) {          // sum over the 16 lanes of a DPP row, in every lane of the row
    x += dpp_f<0x128, 0xf>(0.f, x);    // row_ror:8
    x += dpp_f<0x124, 0xf>(0.f, x);    // row_ror:4
    x += dpp_f<0x122, 0xf>(0.f, x);    // row_ror:2
    x += dpp_f<0x121, 0xf>(0.f, x);    // row_ror:1
    return x;
}
// One position's heads, losses and head gradients by a workgroup of NW wavefronts (all of them load and multiply; the softmax /
// loss reductions run on the first four).  `sm.gs` = the pooled features if g == nullptr (the fused kernels have them in LDS
// already); on return sm.dgv = dg, also stored to dg_out if that is not null.
//
// Weight access.  Every matrix is read ONCE, by 16-byte loads, into registers that serve its forward product AND its transposed
// product on the way back: a quarter wave (16 lanes = one DPP row) owns a weight row, lane l of it holds columns 4 l .. 4 l + 3 (and
// 64 + 4 l .. for the 128-wide first layers), so
//   forward    y[row]  = sum_k W[row][k] x[k]     = 4 or 8 FMAs per lane + four DPP row rotations
//   backward   dx[k]  += dy[row] W[row][k]          = FMAs into the lane's own columns, no reduction until the rows of the 4 NW
//                                                     quarter waves are added up through LDS in a fixed order.
// (Before: one row per wave instruction with a 64-lane reduction per row, policy_head.2 and both first layers read twice, the
//  second time with 4-byte strided loads -- 180 vector-memory instructions and ~100 weight registers per lane; now 15 and 60.)
template <int NW>
__device__ __forceinline__ void heads_board(HeadsSmem& sm, int b, const float* __restrict__ g, const HeadParams& Pm,
                                            const float* __restrict__ pi_all, const float* __restrict__ z_all,
                                            const int64_t* __restrict__ order, int first, int A, int B,
                                            float* __restrict__ hp, float* __restrict__ hv, float* __restrict__ lg,
                                            float* __restrict__ pol, float* __restrict__ vp, float* __restrict__ val,
                                            float* __restrict__ loss, float* __restrict__ dhp, float* __restrict__ dhv,
                                            float* __restrict__ dg) {
    float (&gs)[TH] = sm.gs; float (&hs)[TH] = sm.hs; float (&dhs)[TH] = sm.dhs; float (&dl)[256] = sm.dl;
    float (&red)[2][8][2] = sm.red;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, rg = lane >> 4, l = lane & 15;
    const float *Wp1 = Pm.p[0], *bp1 = Pm.p[1], *Wp2 = Pm.p[2], *bp2 = Pm.p[3], *Wv1 = Pm.p[4], *bv1 = Pm.p[5], *Wv2 = Pm.p[6], *bv2 = Pm.p[7];
    const size_t rec = record_of(order, first, b);
    TS_DECL
    constexpr int HEADS_TS = NW == 4 ? 2 : 7;
    (void)HEADS_TS;
    constexpr int HG = TH / (4 * NW);                               // first-layer row groups per wave (4 rows each)
    constexpr int LG = 64 / NW;                                     // policy_head.2 row groups per wave: 64 groups = 256 rows >= A
    f32x4 w1a[HG], w1b[HG], w2[LG];
    float hb[HG];
#pragma unroll
    for (int i = 0; i < HG; ++i) {
        const int o = 4 * (wave + NW * i) + rg;                     // hidden unit: 0..63 policy head, 64..127 value head
        const float* wr = (o < HH ? Wp1 + (size_t)o * TH : Wv1 + (size_t)(o - HH) * TH) + 4 * l;
        w1a[i] = ld4(wr); w1b[i] = ld4(wr + 64);
        hb[i] = o < HH ? bp1[o] : bv1[o - HH];
    }
#pragma unroll
    for (int i = 0; i < LG; ++i) {
        const int a = 4 * (wave + NW * i) + rg;
        w2[i] = a < A ? ld4(Wp2 + (size_t)a * HH + 4 * l) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // wave 0's inputs of the softmax / loss section, requested up front with everything else: targets and logit biases of its four
    // logits per lane, the value head's second layer
    float tgq[4], lbq[4], wv2q = 0.f, bvq = 0.f, ztq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int a = lane + 64 * j;
        const bool ok = wave == 0 && a < A;
        tgq[j] = ok ? pi_all[rec * A + a] : 0.f;
        lbq[j] = ok ? bp2[a] : 0.f;
    }
    if (wave == 0) { wv2q = Wv2[lane]; bvq = bv2[0]; ztq = z_all[rec]; }
    if (g && t < TH) gs[t] = g[(size_t)b * TH + t];
    __syncthreads();
    TS(HEADS_TS, 0)
    {   // hidden layers: hs[0..63] policy, hs[64..127] value
        const f32x4 g0 = ld4(gs + 4 * l), g1 = ld4(gs + 64 + 4 * l);
#pragma unroll
        for (int i = 0; i < HG; ++i) {
            const int o = 4 * (wave + NW * i) + rg;
            const float s = fmaxf(row16_total(dot4(w1a[i], g0) + dot4(w1b[i], g1)) + hb[i], 0.f);
            if (l == 0) {
                hs[o] = s;
                (o < HH ? hp : hv)[(size_t)b * HH + (o & 63)] = s;
            }
        }
    }
    __syncthreads();
    TS(HEADS_TS, 1)
    {
        const f32x4 h4 = ld4(hs + 4 * l);
#pragma unroll
        for (int i = 0; i < LG; ++i) {
            const int a = 4 * (wave + NW * i) + rg;
            const float s = row16_total(dot4(w2[i], h4));
            if (l == 0 && a < A) dl[a] = s;                          // (dl is reused for d loss / d logits below)
        }
    }
    __syncthreads();
    TS(HEADS_TS, 2)
    // softmax, the reference's second softmax inside CrossEntropyLoss, both losses and the way back to the logits: ONE wavefront
    // holds all A <= 256 logits (four per lane) and every reduction is a wave reduction -- no barrier until the results are out
    // (four workgroup-wide reductions with a barrier each took 3.9 k cycles of the 12 k the heads need).
    if (wave == 0) {
        float lgv[4], tg[4], pv[4], dp[4];
        float mxl = -INFINITY, ts = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = lane + 64 * j;
            const bool ok = a < A;
            lgv[j] = ok ? dl[a] + lbq[j] : -INFINITY;
            tg[j] = tgq[j];
            mxl = fmaxf(mxl, lgv[j]);
            ts += tg[j];
        }
        const float m = wave_max(mxl);
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pv[j] = lane + 64 * j < A ? expf(lgv[j] - m) : 0.f; se += pv[j]; }
        se = wave_sum(se);
        const float tsum = wave_sum(ts);
        float s2 = 0.f, e2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pv[j] = pv[j] / se;                                  // first softmax (the network's own, pv_network_gnn.py:42)
            e2[j] = lane + 64 * j < A ? expf(pv[j]) : 0.f;       // second softmax inside CrossEntropyLoss; p in [0,1]: no shift needed
            s2 += e2[j];
        }
        s2 = wave_sum(s2);
        const float ls2 = logf(s2);
        float lp = 0.f, dot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = lane + 64 * j < A;
            dp[j] = ok ? ((e2[j] / s2) * tsum - tg[j]) / (float)B : 0.f;   // d(mean_b l_b) / d pol
            lp += ok ? -tg[j] * (pv[j] - ls2) : 0.f;
            dot += dp[j] * pv[j];
        }
        lp = wave_sum(lp);
        dot = wave_sum(dot);
        const float vsum = wave_sum(wv2q * hs[HH + lane]);       // the value head's 64-term dot product
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = lane + 64 * j;
            const float dlogit = a < A ? pv[j] * (dp[j] - dot) : 0.f;   // back through the first softmax
            dl[a] = dlogit;                                             // (zero for the rows A..255 of the padded row groups)
            if (a < A) {
                pol[(size_t)b * A + a] = pv[j];
                lg[(size_t)b * A + a] = dlogit;
            }
        }
        const float v = tanhf(vsum + bvq);
        const float dv = v - ztq;
        const float dvp0 = (2.f * dv / (float)B) * (1.f - v * v);
        if (lane == 0) {
            red[0][0][0] = dvp0;
            val[b] = v;
            vp[b] = dvp0;
            loss[2 * b] = lp;
            loss[2 * b + 1] = dv * dv;
        }
    }
    __syncthreads();
    const float dvp = red[0][0][0];
    TS(HEADS_TS, 3)
    {   // d loss / d policy hidden layer: this quarter wave's rows of policy_head.2, transposed product
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < LG; ++i) acc += dl[4 * (wave + NW * i) + rg] * w2[i];
        st4(&sm.part[4 * wave + rg][4 * l], acc);
    }
    __syncthreads();
    TS(HEADS_TS, 4)
    if (t < TH) {
        const int j = t & 63;
        float s;
        if (t < HH) {
            s = 0.f;
#pragma unroll
            for (int r = 0; r < 4 * NW; ++r) s += sm.part[r][j];
        } else s = dvp * Wv2[j];
        if (!(hs[t] > 0.f)) s = 0.f;
        dhs[t] = s;
        (t < HH ? dhp : dhv)[(size_t)b * HH + j] = s;
    }
    __syncthreads();
    TS(HEADS_TS, 5)
    {   // dg = dhp W_p1 + dhv W_v1: this quarter wave's rows of the two first layers, transposed product
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < HG; ++i) {
            const float d = dhs[4 * (wave + NW * i) + rg];
            a0 += d * w1a[i]; a1 += d * w1b[i];
        }
        st4(&sm.part[4 * wave + rg][4 * l], a0);
        st4(&sm.part[4 * wave + rg][64 + 4 * l], a1);
    }
    __syncthreads();
    TS(HEADS_TS, 6)
    if (t < TH) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4 * NW; ++r) s += sm.part[r][t];
        sm.dgv[t] = s;
        if (dg) dg[(size_t)b * TH + t] = s;
    }
    __syncthreads();
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
// (the body is a device function over ONE raw LDS block so that the split-precision kernel below can fall back to it for a board
//  whose values leave fp16 range without owning two sets of static LDS arrays; `hrows` = rows per board of the h1 / h2 buffers)
constexpr int F32_BODY_SMEM = (int)(sizeof(float) * (2 * 96 * SA + 84 * SB + 4 * TH) + sizeof(BoardGraph));
template <int N>
__device__ __forceinline__ void train_board_f32_body(unsigned char* __restrict__ smem, const uint8_t* __restrict__ states72,
                                                     const int64_t* __restrict__ order, int first,
                                                     const TrunkParams& tp, const HeadParams& hpm, const float* __restrict__ pi_all,
                                                     const float* __restrict__ z_all, int A, int B, int hrows,
                                                     float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ g_out,
                                                     float* __restrict__ hp, float* __restrict__ hv, float* __restrict__ lg,
                                                     float* __restrict__ pol, float* __restrict__ vp, float* __restrict__ val,
                                                     float* __restrict__ loss, float* __restrict__ dhp, float* __restrict__ dhv,
                                                     float* __restrict__ part_dW3, float* __restrict__ part_dW2,
                                                     float* __restrict__ part_dW1, float* __restrict__ part_db) {
    constexpr int V = N * N, RT = (V + 15) / 16, VK = (V + 3) / 4 * 4, NIT = (V + 15) / 16;
    float* const Hs = reinterpret_cast<float*>(smem);
    float* const Zs = Hs + 96 * SA;
    float* const Hb = Zs + 96 * SA;
    float* const cs = Hb + 84 * SB;
    BoardGraph& gr = *reinterpret_cast<BoardGraph*>(cs + 4 * TH);
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
                if (hglob) st4(hglob + ((size_t)b * hrows + n) * TH + c4, a);
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
        if (hprev) hin.issue(hprev + (size_t)b * hrows * TH, t);
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
        const float v = hsm.dgv[col] / (float)V;
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
#ifdef AQG_TRAIN_DEBUG
        for (int rt = 0; rt < RT; ++rt) for (int i = 0; i < 4; ++i) if (16 * rt + 4 * q + i < V) DBG_PUT(1, B, b, 16 * rt + 4 * q + i, col, acc[rt][i])
        for (int i = t; i < V * TH; i += 512) DBG_PUT(2, B, b, i / TH, i % TH, Hs[(i / TH) * SA + (i % TH)])
#endif
        const float dbp = mask_and_bias_grad(Hb, SB);
#ifdef AQG_TRAIN_DEBUG
        for (int rt = 0; rt < RT; ++rt) for (int i = 0; i < 4; ++i) if (16 * rt + 4 * q + i < V) DBG_PUT(0, B, b, 16 * rt + 4 * q + i, col, acc[rt][i])
#endif
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
template <int N>
__global__ __launch_bounds__(512) void train_board_kernel(const uint8_t* __restrict__ states72, const int64_t* __restrict__ order, int first,
                                                          TrunkParams tp, HeadParams hpm, const float* __restrict__ pi_all,
                                                          const float* __restrict__ z_all, int A, int B, int hrows,
                                                          float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ g_out,
                                                          float* __restrict__ hp, float* __restrict__ hv, float* __restrict__ lg,
                                                          float* __restrict__ pol, float* __restrict__ vp, float* __restrict__ val,
                                                          float* __restrict__ loss, float* __restrict__ dhp, float* __restrict__ dhv,
                                                          float* __restrict__ part_dW3, float* __restrict__ part_dW2,
                                                          float* __restrict__ part_dW1, float* __restrict__ part_db) {
    __shared__ __align__(16) unsigned char smem[F32_BODY_SMEM];
    train_board_f32_body<N>(smem, states72, order, first, tp, hpm, pi_all, z_all, A, B, hrows, h1, h2, g_out, hp, hv, lg, pol, vp, val, loss,
                            dhp, dhv, part_dW3, part_dW2, part_dW1, part_db);
}

// ---------------------------------------------------------------------------------------------
// The same step of ONE 9x9 position with every contraction on the 16-bit matrix pipe in split precision (split_mfma.hpp): the
// default on the 9x9 board.  v_mfma_f32_16x16x32_f16 runs at 16x the rate of the f32-input MFMA the body above uses; with three
// fp16 terms per f32 product the contractions cost a fifth, and the neighbourhood aggregation -- a VALU gather over LDS above,
// 38 % of that kernel -- becomes 30 MFMAs on the board's banded A_hat (ten 32x16 blocks, f32 entries split hi / lo).
//
// Layouts.  Wave w owns feature columns 16 w .. 16 w + 15 everywhere.  A 16x16 accumulator tile has lane = column c (lane & 15)
// and rows 4 q + r (q = lane >> 4) in its four registers; two consecutive row tiles of NODES are therefore an operand fragment of
// any product contracted over the nodes (k-slot order of split_mfma.hpp), with no data movement:
//   linear map      U = X W^T        A = fp16 planes of X in LDS [node][feature] (ds_read_b128), B = this wave's rows of W, split on
//                                    the fly from the f32 master weights  ->  U: lane = feature, registers = nodes
//   aggregation, T  V^T = U^T A_hat  A = U (registers), B = A_hat block  ->  lane = node, registers = 4 consecutive features:
//                                    relu, split, 8-byte plane stores: the next linear map's A operand
//   aggregation, R  V = A_hat U      A = A_hat block (the SAME fragment: A_hat is symmetric), B = U (registers)  ->  lane = feature,
//                                    registers = nodes: an operand of the weight gradient dW = dZ^T H, which contracts over nodes
//   weight gradient dW[all j][k in the wave's 16] = sum_n dZ[n][j] H[n][k]: B = the wave's OWN H fragments (form R of the forward
//                                    pass, parked in memory lane-linearly and read back), A = the dZ fragments of all eight waves
//                                    through 48 KB of LDS, lane-linear (form R of the backward pass)
// so a layer costs 72 (linear) + 30 + 30 (both forms) MFMAs per wave going forward, and 72 (data gradient) + 60 + 72 (weight
// gradient) going back; nothing is ever transposed.  ReLU masks are 24 bits per lane and layer, kept in registers.
// Range: forward values are O(1); the backward pass is scaled per board by a power of two that puts max |dg| at 128..256 (the
// gradients of a mean loss over 128 positions would otherwise sit in fp16's subnormals) and unscaled, exactly, at the stores of
// the partial sums.  Every f32 value is range-checked before it is split; a board that meets |x| > 65504 anywhere is redone
// by the exact-f32 body above in the same launch (counted in g_train_fallbacks) -- the reference's fp32 has no such cliff.
// ---------------------------------------------------------------------------------------------
struct alignas(16) SplitSmem {
    alignas(16) unsigned char P[2][PPLANE];                 // fp16 hi / lo planes [node][feature]: H_l going forward, dZ_l going back
    alignas(16) unsigned int AF[2][AF_BLOCKS][64][4];       // hi / lo fragments of the ten non-zero blocks of A_hat
    alignas(16) unsigned int FR[8][3][2][64][4];            // [wave][k block][hi / lo]: dZ_l as A fragments of the weight gradient (the heads' scratch before)
    alignas(16) unsigned short X0A[96][8];                  // the six input features per node (fp16, exact), rows of the layer-1 A operand
    alignas(16) unsigned short X0T[16][96];                 // ... and feature-major: B operand of layer 1's weight gradient
    alignas(16) float dinv[96];                             // deg^-1/2 (self loop included), 0 for the padding nodes
    unsigned char ob[96];                                   // open sides of a tile: bit 0 up (n - 9), 1 down, 2 left, 3 right
};
static_assert(sizeof(HeadsSmem) <= sizeof(unsigned int) * 8 * 3 * 2 * 64 * 4, "heads scratch aliases FR");
constexpr int SPLIT_KERNEL_SMEM = (int)sizeof(SplitSmem) > F32_BODY_SMEM ? (int)sizeof(SplitSmem) : F32_BODY_SMEM;
static_assert(SPLIT_KERNEL_SMEM <= 160 * 1024, "one workgroup per CU");
__device__ unsigned int g_train_fallbacks = 0;
constexpr int BWD_SCALE_LOG2 = 7;       // max |dg| s in [128, 256): dP3 = dg s / 81 <= 3.2, 2^14 of headroom, every lo half a normal fp16

// Range guard: the largest |x| this lane has split.  (The bit-pattern form of the inference trunk -- one signed and one unsigned
// integer maximum, one v_max3 per two values each, no canonicalising v_max per operand -- saves 400 of this body's 3,300 vector
// instructions and is 7 % SLOWER here, 0.0610 against 0.0571 ms per step in a same-box A/B: -DAQG_INT_TRK, tools/ab_train.sh.)
#ifdef AQG_INT_TRK
struct Rng { int i = 0; unsigned int u = 0u; };
__device__ __forceinline__ void trk(Rng& m, float a, float b) {
    const int ia = __builtin_bit_cast(int, a), ib = __builtin_bit_cast(int, b);
    m.i = max(max(ia, ib), m.i);
    m.u = max(max((unsigned int)ia, (unsigned int)ib), m.u);
}
__device__ __forceinline__ bool out_of_fp16_range(const Rng& m) { return m.i > 0x477FE000 || m.u > 0xC77FE000u; }   // 65504.0f / -65504.0f
// (through a scalar parameter: __builtin_bit_cast applied to a vector ELEMENT expression read element 0 for all four -- hipcc 7.2)
__device__ __forceinline__ float relu1i(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }
#else
struct Rng { float m = 0.f; };
__device__ __forceinline__ void trk(Rng& m, float a, float b) { m.m = fmaxf(m.m, fmaxf(fabsf(a), fabsf(b))); }
__device__ __forceinline__ bool out_of_fp16_range(const Rng& m) { return !(m.m <= 65504.0f); }
#ifdef AQG_ABL_FLOAT_RELU
__device__ __forceinline__ float relu1i(float x) { return fmaxf(x, 0.f); }
#else   // relu on the bit pattern: one v_max_i32, no canonicalising v_max on top (through a scalar parameter: see above)
__device__ __forceinline__ float relu1i(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }
#endif
#endif
// 1 if x > 0 else 0, on the bit pattern (a positive float is a positive integer): one v_med3_i32
__device__ __forceinline__ unsigned int positive_bit(float x) { return (unsigned int)min(max(__builtin_bit_cast(int, x), 0), 1); }
__device__ __forceinline__ void trk(Rng& m, const f32x4 v) { trk(m, v[0], v[1]); trk(m, v[2], v[3]); }
__device__ __forceinline__ f32x4 relu4i(const f32x4 v) { return f32x4{relu1i(v[0]), relu1i(v[1]), relu1i(v[2]), relu1i(v[3])}; }
__device__ __forceinline__ void mfma_fence(u32x4& a) { asm volatile("s_nop 3" : "+v"(a)); }
// tile m of a [nodes][16] accumulator image -> dwords 2 (m & 1), + 1 of k block m >> 1 of its hi / lo node-contraction fragments
// The range check of two values that are being split: ONE v_max3_f32 with |.| modifiers (fmaxf(|a|, |b|) costs the compiler a
// canonicalising v_max per operand on top).  As an asm statement it must not be the first reader of a matrix-pipe result (hipcc pads
// nothing for asm): `dep` is the packed fp16 pair the compiler-visible v_cvt_pk has just made of the same two values, so the check
// sits behind that instruction, the way lo_pair() does.
__device__ __forceinline__ void trk_after(Rng& m, unsigned int dep, float a, float b) {
#ifdef AQG_INT_TRK
    (void)dep; trk(m, a, b);
#else
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m.m) : "v"(a), "v"(b), "v"(dep));
#endif
}
__device__ __forceinline__ void split_tile(const f32x4 z, int m, u32x4 (&zh)[3], u32x4 (&zl)[3], Rng* rng = nullptr) {
    const int kb = m >> 1, d = 2 * (m & 1);
    zh[kb][d] = cvt_pk_f16(z[0], z[1]); zh[kb][d + 1] = cvt_pk_f16(z[2], z[3]);
    if (rng) { trk_after(*rng, zh[kb][d], z[0], z[1]); trk_after(*rng, zh[kb][d + 1], z[2], z[3]); }
    zl[kb][d] = lo_pair(zh[kb][d], z[0], z[1]); zl[kb][d + 1] = lo_pair(zh[kb][d + 1], z[2], z[3]);
}
__device__ __forceinline__ void plane_store4(unsigned char (&P)[2][PPLANE], int off, const f32x4 v, Rng* rng = nullptr) {
    const unsigned int h01 = cvt_pk_f16(v[0], v[1]), h23 = cvt_pk_f16(v[2], v[3]);
    if (rng) { trk_after(*rng, h01, v[0], v[1]); trk_after(*rng, h23, v[2], v[3]); }
    *reinterpret_cast<u32x2*>(&P[0][off]) = (u32x2){h01, h23};
    *reinterpret_cast<u32x2*>(&P[1][off]) = (u32x2){lo_pair(h01, v[0], v[1]), lo_pair(h23, v[2], v[3])};
}
__device__ __forceinline__ bool live_row(int nt, int q, int r) { return nt < 5 || (q == 0 && r == 0); }      // node 16 nt + 4 q + r < 81

// U = X W^T for this wave's 16 columns from the planes (six 16-row tiles, tile 5 = row 80 repeated, x four 32-deep k blocks, three
// fp16 terms, smallest first); post(m, tile) sees every finished tile before it is split into the node-contraction fragments.
template <class Post>
__device__ __forceinline__ void linear_split_post(const unsigned char (&P)[2][PPLANE], const u32x4 (&Bh)[4], const u32x4 (&Bl)[4], int lane,
                                                  u32x4 (&zh)[3], u32x4 (&zl)[3], Rng& rng, Post post) {
    const int c = lane & 15, q = lane >> 4;
    // The fragments of step s + AQG_TRAIN_FRAG_AHEAD are requested while step s multiplies (a ring of that many register pairs).
    // One step ahead is enough: 2 / 3 / 5 steps measured 0.0568 / 0.0572 / 0.0613 ms per step against 0.0565 (tools/ab_train.sh) --
    // the phase is not waiting for LDS.
#ifndef AQG_TRAIN_FRAG_AHEAD
#define AQG_TRAIN_FRAG_AHEAD 1
#endif
    constexpr int D = AQG_TRAIN_FRAG_AHEAD;
    u32x4 ring[D + 1][2];
    auto frag_off = [&](int step) -> int {
        const int m = step >> 2, kb = step & 3;
        return plane_off(m < 5 ? 16 * m + c : 80, 4 * kb + q);
    };
    auto request = [&](int step) {
        const int o = frag_off(step);
        ring[step % (D + 1)][0] = *reinterpret_cast<const u32x4*>(&P[0][o]);
        ring[step % (D + 1)][1] = *reinterpret_cast<const u32x4*>(&P[1][o]);
    };
#pragma unroll
    for (int i = 0; i < D; ++i) request(i);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, done = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int step = 0; step < 24; ++step) {
        const int m = step >> 2, kb = step & 3;
        if (step + D < 24) request(step + D);
        __builtin_amdgcn_sched_barrier(0);                              // (keeps the 48 fragment reads from being hoisted in a body: 192 registers)
        const u32x4 hi = ring[step % (D + 1)][0], lo = ring[step % (D + 1)][1];
        f32x4 a = kb == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc;
        a = mfma_f16(lo, Bh[kb], a);
        a = mfma_f16(hi, Bl[kb], a);
        a = mfma_f16(hi, Bh[kb], a);
        acc = a;
        // the finished tile m - 1 is masked / checked and split under tile m's first MFMA group
        if (m > 0 && kb == 0) { post(m - 1, done); split_tile(done, m - 1, zh, zl, &rng); }
        __builtin_amdgcn_sched_barrier(0);
        if (kb == 3) done = acc;
    }
    post(5, done);
    split_tile(done, 5, zh, zl, &rng);
}

// Both forms of the aggregation over the ten blocks (header of this section), node tile by node tile: the blocks of a tile are
// consecutive, and epi(nt, oT, oR) gets the finished tile (started from the presets pT / pR: bias rows or zero) while the next
// tile's MFMAs are issued -- only one tile's accumulators are alive at a time.
template <bool DO_T, bool DO_R, class Epi>
__device__ __forceinline__ void aggregate_tr(const unsigned int (&AF)[2][AF_BLOCKS][64][4], const u32x4 (&zh)[3], const u32x4 (&zl)[3],
                                             const f32x4 pT, const f32x4 pR, int lane, Epi epi) {
    f32x4 oT = pT, oR = pR;
#pragma unroll
    for (int blk = 0; blk < AF_BLOCKS; ++blk) {
        const int kb = af_kb(blk), nt = af_nt(blk);
        const u32x4 ah = *reinterpret_cast<const u32x4*>(&AF[0][blk][lane][0]);
        const u32x4 al = *reinterpret_cast<const u32x4*>(&AF[1][blk][lane][0]);
        if (DO_T) {
            oT = mfma_f16(zl[kb], ah, oT);
            oT = mfma_f16(zh[kb], al, oT);
            oT = mfma_f16(zh[kb], ah, oT);
        }
        if (DO_R) {
            oR = mfma_f16(ah, zl[kb], oR);
            oR = mfma_f16(al, zh[kb], oR);
            oR = mfma_f16(ah, zh[kb], oR);
        }
        if (blk + 1 == AF_BLOCKS || af_nt(blk + 1) != nt) {
            epi(nt, oT, oR);
            oT = pT; oR = pR;
        }
    }
}

// (a real call: inlined into the split body, the heads' ~130 registers on top of the trunk's state spill -- 300 registers, and the
//  heads alone then take 124 k cycles instead of 25 k; as a callee they get a register allocation of their own)
//  (the scratch travels as its LDS byte offset and is cast back from the LDS address space inside, so that the callee's accesses are
//  ds_ instructions, not flat ones)
typedef __attribute__((address_space(3))) HeadsSmem HeadsSmemLds;
typedef const __attribute__((address_space(1))) float* gcf;           // pointer arguments in the global address space: global_, not flat_
typedef __attribute__((address_space(1))) float* gf;
__device__ __attribute__((noinline)) void heads_board_call(unsigned int sm_lds, int b, gcf w0, gcf w1, gcf w2, gcf w3, gcf w4, gcf w5, gcf w6, gcf w7,
                                                           gcf pi_all, gcf z_all, const __attribute__((address_space(1))) int64_t* order, int first,
                                                           int A, int B, gf hp, gf hv, gf lg, gf pol, gf vp, gf val, gf loss, gf dhp, gf dhv) {
    HeadsSmem& sm = *(HeadsSmem*)reinterpret_cast<HeadsSmemLds*>((size_t)sm_lds);
    HeadParams Pg;
    Pg.p[0] = (const float*)w0; Pg.p[1] = (const float*)w1; Pg.p[2] = (const float*)w2; Pg.p[3] = (const float*)w3;
    Pg.p[4] = (const float*)w4; Pg.p[5] = (const float*)w5; Pg.p[6] = (const float*)w6; Pg.p[7] = (const float*)w7;
    heads_board<8>(sm, b, nullptr, Pg, (const float*)pi_all, (const float*)z_all, (const int64_t*)order, first, A, B, (float*)hp, (float*)hv, (float*)lg,
                   (float*)pol, (float*)vp, (float*)val, (float*)loss, (float*)dhp, (float*)dhv, nullptr);
}

// returns false (to every thread of the workgroup) if a value left fp16 range: the caller redoes the board with the f32 body
__device__ __forceinline__ bool train_board_split_body(unsigned char* __restrict__ smem, const uint8_t* __restrict__ states72,
                                                       const int64_t* __restrict__ order, int first,
                                                       const TrunkParams& tp, const HeadParams& hpm, const float* __restrict__ pi_all,
                                                       const float* __restrict__ z_all, int A, int B,
                                                       float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ g_out,
                                                       float* __restrict__ hp, float* __restrict__ hv, float* __restrict__ lg,
                                                       float* __restrict__ pol, float* __restrict__ vp, float* __restrict__ val,
                                                       float* __restrict__ loss, float* __restrict__ dhp, float* __restrict__ dhv,
                                                       float* __restrict__ part_dW3, float* __restrict__ part_dW2,
                                                       float* __restrict__ part_dW1, float* __restrict__ part_db) {
    constexpr int N = 9, V = 81;
    SplitSmem& sm = *reinterpret_cast<SplitSmem*>(smem);
    HeadsSmem& hsm = *reinterpret_cast<HeadsSmem*>(&sm.FR[0][0][0][0][0]);
    const int b = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), q = lane >> 4, c = lane & 15;
    const int col = 16 * wave + c;
    const float *W1 = tp.p[0], *b1 = tp.p[1], *W2 = tp.p[2], *b2 = tp.p[3], *W3 = tp.p[4], *b3 = tp.p[5];
    Rng mx;                                                            // range guard over everything this lane splits
    TS_DECL
    // ---- loads that do not depend on the board: biases (both layouts), W1, W2 rows of this wave's columns
    const f32x4 bT1 = ld4(b1 + 16 * wave + 4 * q), bT2 = ld4(b2 + 16 * wave + 4 * q);
    const float bR1 = b1[col], bR2 = b2[col], bR3 = b3[col];
    float w1v[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) w1v[e] = q == 0 ? W1[col * TF + e] : 0.f;
    f32x4 wf[8];                                                       // W_l[col][32 kb + 8 q + 0..7]: B fragments of the forward linear maps
    auto request_w = [&](const float* __restrict__ W) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { wf[2 * kb] = ld4(W + (size_t)col * TH + 32 * kb + 8 * q); wf[2 * kb + 1] = ld4(W + (size_t)col * TH + 32 * kb + 8 * q + 4); }
    };
    u32x4 Bh[4], Bl[4];
    auto split_w = [&]() {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x4 a = wf[2 * kb], bb = wf[2 * kb + 1];
            Bh[kb] = (u32x4){cvt_pk_f16(a[0], a[1]), cvt_pk_f16(a[2], a[3]), cvt_pk_f16(bb[0], bb[1]), cvt_pk_f16(bb[2], bb[3])};
            trk_after(mx, Bh[kb][0], a[0], a[1]); trk_after(mx, Bh[kb][1], a[2], a[3]);
            trk_after(mx, Bh[kb][2], bb[0], bb[1]); trk_after(mx, Bh[kb][3], bb[2], bb[3]);
            Bl[kb] = (u32x4){lo_pair(Bh[kb][0], a[0], a[1]), lo_pair(Bh[kb][1], a[2], a[3]), lo_pair(Bh[kb][2], bb[0], bb[1]), lo_pair(Bh[kb][3], bb[2], bb[3])};
            mfma_fence(Bl[kb]);
        }
    };
    request_w(W2);
    // (warming the heads' matrices into this XCD's L2 from here -- one load per 64-byte line, as the f32 body does -- measured 6 % SLOWER
    //  for this body: 0.0607 against 0.0572 ms per step, tools/ab_train.sh)
    // ---- the board: features, open sides, deg^-1/2
    const uint8_t* rec = states72 + record_of(order, first, b) * STATE72;
    if (t < 96) {
        unsigned short xa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float di = 0.f;
        int ob = 0;
        if (t < V) {
            const QState s = unpack72(rec);
            const int x = t / N, y = t % N;
            const bool slot_ok = x < N - 1 && y < N - 1;
            const int slot = x * (N - 1) + y;
            const unsigned short one = 0x3C00;
            xa[0] = t == s.ppos ? one : 0;
            xa[1] = __builtin_bit_cast(unsigned short, (_Float16)(float)s.pwl);
            xa[2] = t == s.epos ? one : 0;
            xa[3] = __builtin_bit_cast(unsigned short, (_Float16)(float)s.ewl);
            xa[4] = (slot_ok && ((s.hw >> slot) & 1)) ? one : 0;
            xa[5] = (slot_ok && ((s.vw >> slot) & 1)) ? one : 0;
            ob = tile_open_bits<N>(s.hw, s.vw, t);
            di = 1.0f / sqrtf((float)(1 + __popc(ob)));
        }
        *reinterpret_cast<u32x4*>(&sm.X0A[t][0]) = (u32x4){xa[0] | ((unsigned)xa[1] << 16), xa[2] | ((unsigned)xa[3] << 16), xa[4] | ((unsigned)xa[5] << 16), 0u};
#pragma unroll
        for (int k = 0; k < 8; ++k) sm.X0T[k][t] = xa[k];
        sm.dinv[t] = di;
        sm.ob[t] = (unsigned char)ob;
    } else if (t < 96 + 8 * 96 / 2) {
        reinterpret_cast<unsigned int*>(&sm.X0T[8][0])[t - 96] = 0u;                     // feature rows 8..15 of the padded tile
    }
    __syncthreads();
    TS(9, 0)
    // ---- A_hat fragments: entry (k-slot e of lane (c, q), block (kb, nt)) = dinv[n] dinv[k] where k is in the closed neighbourhood of n = 16 nt + c
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int blk = wave + 8 * it;                                                   // wave-uniform
        if (blk < AF_BLOCKS) {
            const int kb = (AF_KB_PACK >> (2 * blk)) & 3, nt = (AF_NT_PACK >> (3 * blk)) & 7;
            const int n = 16 * nt + c;
            const int obn = sm.ob[n];
            const float dn = sm.dinv[n];
            float v[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k0 = 32 * kb + 16 * h + 4 * q;
                const f32x4 dk = *reinterpret_cast<const f32x4*>(&sm.dinv[k0]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = k0 + i - n;
                    const bool adj = d == 0 || (d == -N && (obn & 1)) || (d == N && (obn & 2)) || (d == -1 && (obn & 4)) || (d == 1 && (obn & 8));
                    v[4 * h + i] = adj ? dn * dk[i] : 0.f;
                }
            }
            u32x4 fh, fl;
#pragma unroll
            for (int p = 0; p < 4; ++p) { fh[p] = cvt_pk_f16(v[2 * p], v[2 * p + 1]); fl[p] = lo_pair(fh[p], v[2 * p], v[2 * p + 1]); }
            *reinterpret_cast<u32x4*>(&sm.AF[0][blk][lane][0]) = fh;
            *reinterpret_cast<u32x4*>(&sm.AF[1][blk][lane][0]) = fl;
        }
    }
    // ---- layer 1, linear: Z1 = X0 W1^T (K = 6 in one 32-deep block; X0 is exact in fp16: two terms)
    u32x4 zh[3], zl[3];
    {
        u32x4 w1h = {cvt_pk_f16(w1v[0], w1v[1]), cvt_pk_f16(w1v[2], w1v[3]), cvt_pk_f16(w1v[4], w1v[5]), 0u};
        u32x4 w1l = {lo_pair(w1h[0], w1v[0], w1v[1]), lo_pair(w1h[1], w1v[2], w1v[3]), lo_pair(w1h[2], w1v[4], w1v[5]), 0u};
        mfma_fence(w1l);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            u32x4 xa = *reinterpret_cast<const u32x4*>(&sm.X0A[m < 5 ? 16 * m + c : 80][0]);
            if (q != 0) xa = (u32x4){0u, 0u, 0u, 0u};
            f32x4 z = mfma_f16(xa, w1l, (f32x4){0.f, 0.f, 0.f, 0.f});
            z = mfma_f16(xa, w1h, z);
            split_tile(z, m, zh, zl, &mx);
        }
    }
    __syncthreads();                                                   // A_hat fragments complete
    TS(9, 1)
    // ---- forward epilogues
    const int poff = (2 * wave + (q >> 1)) /* 16-byte slot of features 16 w + 4 q .. */, pbyte = 8 * (q & 1);
    auto store_plane_tile = [&](int nt, f32x4 v, bool relu) {          // T form: lane = node c of tile nt, features 16 w + 4 q + r
        if (relu) v = relu4i(v);                                        // (what is split is range-checked: a pre-activation below -65504 is a zero)
        if (nt < 5 || c == 0) plane_store4(sm.P, plane_off(16 * nt + c, poff) + pbyte, v, &mx);
    };
    unsigned int msk[3] = {0u, 0u, 0u};                                 // ReLU masks of the three layers: bit 4 nt + r, R layout
    u32x4 hh[3], hl[3];                                                 // R form of H_l (lane = feature c, nodes 16 nt + 4 q + r) as fragments
    auto park_tile = [&](int nt, f32x4 v, unsigned int& m) {       // (the same values as the T form, which store_plane_tile has range-checked)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                  // (a NaN would count as positive: the range guard has long fired then)
            const float x = v[r];
            if (nt < 5) m |= positive_bit(x) << (4 * nt + r);
            else if (live_row(nt, q, r)) m |= positive_bit(x) << (4 * nt + r);
        }
        v = relu4i(v);
        split_tile(v, nt, hh, hl);
    };
    auto park_store = [&](float* __restrict__ hpark) {
        u32x4* dst = reinterpret_cast<u32x4*>(hpark + (size_t)b * 96 * TH) + (size_t)wave * 6 * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) { dst[(2 * kb) * 64] = hh[kb]; dst[(2 * kb + 1) * 64] = hl[kb]; }
    };
    // ---- layer 1: aggregation, planes of H1, parked fragments of H1
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) mfma_fence(zl[kb]);
    aggregate_tr<true, true>(sm.AF, zh, zl, bT1, (f32x4){bR1, bR1, bR1, bR1}, lane, [&](int nt, const f32x4& oT, const f32x4& oR) {
        store_plane_tile(nt, oT, true);
        park_tile(nt, oR, msk[0]);
    });
    TS(9, 14)
    park_store(h1);
    split_w();                                                          // W2 fragments
    request_w(W3);
    TS(9, 15)
    __syncthreads();                                                    // planes of H1 complete
    TS(9, 2)
    // ---- layer 2
    linear_split_post(sm.P, Bh, Bl, lane, zh, zl, mx, [&](int, f32x4&) {});
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) mfma_fence(zl[kb]);
    __syncthreads();                                                    // everybody has read the planes of H1
    TS(9, 3)
    aggregate_tr<true, true>(sm.AF, zh, zl, bT2, (f32x4){bR2, bR2, bR2, bR2}, lane, [&](int nt, const f32x4& oT, const f32x4& oR) {
        store_plane_tile(nt, oT, true);
        park_tile(nt, oR, msk[1]);
    });
    park_store(h2);
    split_w();                                                          // W3 fragments
    __syncthreads();                                                    // planes of H2 complete
    TS(9, 4)
    // ---- layer 3 (R form only: its ReLU mask and the mean pool; H3 itself is not needed again)
    linear_split_post(sm.P, Bh, Bl, lane, zh, zl, mx, [&](int, f32x4&) {});
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) mfma_fence(zl[kb]);
    {
        float s = 0.f;
        aggregate_tr<false, true>(sm.AF, zh, zl, bT2, (f32x4){bR3, bR3, bR3, bR3}, lane, [&](int nt, const f32x4&, const f32x4& oR) {
            // (H3 is not split: only its signs and its f32 column sums are used)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (oR[r] > 0.f && live_row(nt, q, r)) { msk[2] |= 1u << (4 * nt + r); s += oR[r]; }
        });
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        s /= (float)V;                                                  // global_mean_pool
        // (FR, which the heads' scratch aliases, is first written in the backward pass)
        if (q == 0) { hsm.gs[col] = s; g_out[(size_t)b * TH + col] = s; }
    }
    // the backward pass's own operands: W_{l+1}^T fragments of the data gradients and this wave's parked H_{l-1} (requested a phase ahead)
    float wt[32];                                                       // W_l[32 kb + 8 q + e][col]: B fragments of the data gradients
    auto request_wt = [&](const float* __restrict__ W) {
#pragma unroll
        for (int i = 0; i < 32; ++i) wt[i] = W[(size_t)(32 * (i >> 3) + 8 * q + (i & 7)) * TH + col];
    };
    auto split_wt = [&]() {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float a0 = wt[8 * kb + 2 * p], a1 = wt[8 * kb + 2 * p + 1];
                Bh[kb][p] = cvt_pk_f16(a0, a1);                 // (range-checked as W_l's rows by split_w: the eight waves' rows are the whole matrix)
                Bl[kb][p] = lo_pair(Bh[kb][p], a0, a1);
            }
            mfma_fence(Bl[kb]);
        }
    };
    auto request_h = [&](const float* __restrict__ hpark) {
        const u32x4* src = reinterpret_cast<const u32x4*>(hpark + (size_t)b * 96 * TH) + (size_t)wave * 6 * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) { hh[kb] = src[(2 * kb) * 64]; hl[kb] = src[(2 * kb + 1) * 64]; }
    };
    TS(9, 5)
    // ---- heads, losses, head gradients (its first barrier publishes gs)
#ifdef AQG_HEADS_INLINE      // developer A/B (tools/ab_train.sh): the heads inlined into this body
    heads_board<8>(hsm, b, nullptr, hpm, pi_all, z_all, order, first, A, B, hp, hv, lg, pol, vp, val, loss, dhp, dhv, nullptr);
#else
    heads_board_call((unsigned int)(size_t)(HeadsSmemLds*)&hsm, b, (gcf)hpm.p[0], (gcf)hpm.p[1], (gcf)hpm.p[2], (gcf)hpm.p[3], (gcf)hpm.p[4], (gcf)hpm.p[5],
                     (gcf)hpm.p[6], (gcf)hpm.p[7], (gcf)pi_all, (gcf)z_all, (const __attribute__((address_space(1))) int64_t*)order, first, A, B,
                     (gf)hp, (gf)hv, (gf)lg, (gf)pol, (gf)vp, (gf)val, (gf)loss, (gf)dhp, (gf)dhv);
#endif
    TS(9, 6)
    {
        // (requested BEHIND the call: hoisted above it -- which the compiler does unless the base pointers pass through this empty
        //  asm statement -- the 56 registers would be loaded, waited for, spilled around the call and reloaded)
        const float* W3b = W3;
        const float* h2b = h2;
#ifndef AQG_ABL_LAUNDER
        asm volatile("" : "+s"(W3b), "+s"(h2b));
#endif
        request_wt(W3b);
        request_h(h2b);
    }
    // ---- backward.  dg = hsm.dgv; scaled by a power of two s with max |dg| s in [128, 256)
    float dgs, inv_s;
    {
        const float d0 = hsm.dgv[lane], d1 = hsm.dgv[64 + lane];
        const float m = wave_max(fmaxf(fabsf(d0), fabsf(d1)));
        const int e = (__builtin_bit_cast(int, m) >> 23) & 0xFF;
        int es = 254 + BWD_SCALE_LOG2 - e;                              // biased exponent of s = 2^(BWD_SCALE_LOG2 - (e - 127))
        es = (e == 0 || e == 255) ? 127 : min(max(es, 1), 254);
        const float s = __builtin_bit_cast(float, es << 23);
        inv_s = 1.0f / s;
        dgs = hsm.dgv[col] * s / (float)V;                              // global_mean_pool backward, this lane's column
    }
    __syncthreads();                                                    // the heads' scratch is dead: FR may be written
    auto store_db = [&](float sdb, int layer) {
        sdb += __shfl_xor(sdb, 16);
        sdb += __shfl_xor(sdb, 32);
        if (q == 0) part_db[((size_t)layer * B + b) * TH + col] = sdb * inv_s;
    };
    u32x4 ah[3], al[3];                                                 // R form of dZ_l: A fragments of dW_l
    auto publish_dz = [&]() {
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            *reinterpret_cast<u32x4*>(&sm.FR[wave][kb][0][lane][0]) = ah[kb];
            *reinterpret_cast<u32x4*>(&sm.FR[wave][kb][1][lane][0]) = al[kb];
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto aggregate_back = [&](bool planes) {
        if (planes)
            aggregate_tr<true, true>(sm.AF, zh, zl, zero4, zero4, lane, [&](int nt, const f32x4& oT, const f32x4& oR) {
                store_plane_tile(nt, oT, false);                        // dZ_l: A operand of the next data gradient (range-checked there; oR repeats it)
                split_tile(oR, nt, ah, al);
            });
        else
            aggregate_tr<false, true>(sm.AF, zh, zl, zero4, zero4, lane, [&](int nt, const f32x4&, const f32x4& oR) {
                split_tile(oR, nt, ah, al, &mx);
            });
    };
    auto weight_grad = [&](float* __restrict__ pdW) {                   // dW_l[all j][this wave's k]: A = parked H_{l-1}, B = FR (all waves)
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) mfma_fence(hl[kb]);              // (loaded, not computed: harmless)
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            // the TRANSPOSED tile dW^T[k][j] = sum_n H[n][k] dZ[n][j]: lane = row j = 16 jt + c of dW, registers = 4 consecutive
            // columns k = 16 w + 4 q + r -- one 16-byte store per tile and lane (the other operand order needs four 4-byte ones)
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                const u32x4 zh_ = *reinterpret_cast<const u32x4*>(&sm.FR[jt][kb][0][lane][0]);
                const u32x4 zl_ = *reinterpret_cast<const u32x4*>(&sm.FR[jt][kb][1][lane][0]);
                o = mfma_f16(hl[kb], zh_, o);
                o = mfma_f16(hh[kb], zl_, o);
                o = mfma_f16(hh[kb], zh_, o);
            }
            st4(pdW + (size_t)b * TH * TH + (size_t)(16 * jt + c) * TH + 16 * wave + 4 * q, o * inv_s);
        }
    };
    // layer 3: dP3 = dg / V on the nodes whose H3 is positive
    {
        float sdb = 0.f;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = ((msk[2] >> (4 * nt + r)) & 1u) ? dgs : 0.f; sdb += v[r]; }
            split_tile(v, nt, zh, zl);
        }
        trk(mx, dgs, dgs);
        store_db(sdb, 2);
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) mfma_fence(zl[kb]);
        aggregate_back(true);                                           // (every wave left the planes of H2 long ago)
        publish_dz();
        split_wt();                                                     // W3^T fragments
        request_wt(W2);
        __syncthreads();                                                // planes and fragments of dZ3 complete
        TS(9, 7)
        weight_grad(part_dW3);
        request_h(h1);
        TS(9, 8)
    }
    // layers 2 and 1: dH_l = dZ_{l+1} W_{l+1}, masked by H_l > 0
    auto masked_linear = [&](unsigned int m, int layer) {
        float sdb = 0.f;
#ifdef AQG_TRAIN_DEBUG
        if (layer == 1)            // the planes hold dZ3 / s: dense dump of hi + lo
            for (int i = t; i < V * TH; i += 512) {
                const int n = i / TH, f = i % TH, o = plane_off(n, f >> 3) + 2 * (f & 7);
                DBG_PUT(2, B, b, n, f, ((float)*reinterpret_cast<const _Float16*>(&sm.P[0][o]) + (float)*reinterpret_cast<const _Float16*>(&sm.P[1][o])) * inv_s)
            }
#endif
        linear_split_post(sm.P, Bh, Bl, lane, zh, zl, mx, [&](int mt, f32x4& z) {
#ifdef AQG_TRAIN_DEBUG
            if (layer == 1) for (int r = 0; r < 4; ++r) if (live_row(mt, q, r)) DBG_PUT(1, B, b, 16 * mt + 4 * q + r, col, z[r] * inv_s)
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) { if (!((m >> (4 * mt + r)) & 1u)) z[r] = 0.f; sdb += z[r]; }
#ifdef AQG_TRAIN_DEBUG
            if (layer == 1) for (int r = 0; r < 4; ++r) if (live_row(mt, q, r)) DBG_PUT(0, B, b, 16 * mt + 4 * q + r, col, z[r] * inv_s)
#endif
        });
        store_db(sdb, layer);
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) mfma_fence(zl[kb]);
    };
    {
        masked_linear(msk[1], 1);
        TS(9, 9)
        __syncthreads();                                                // everybody has read the planes of dZ3 (and FR: weight_grad is behind)
        aggregate_back(true);
        publish_dz();
        split_wt();                                                     // W2^T fragments
        __syncthreads();
        TS(9, 10)
        weight_grad(part_dW2);
        TS(9, 11)
    }
    {
        masked_linear(msk[0], 0);
        TS(9, 12)
        aggregate_back(false);
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) mfma_fence(al[kb]);
        // dW1[this wave's j][k < 6] = sum_n dZ1[n][j] X0[n][k]: A = own fragments, B = the feature-major X0 image (exact: two terms)
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            const u32x2 x0 = *reinterpret_cast<const u32x2*>(&sm.X0T[c][32 * kb + 4 * q]);
            const u32x2 x1 = *reinterpret_cast<const u32x2*>(&sm.X0T[c][32 * kb + 16 + 4 * q]);
            const u32x4 xb = {x0[0], x0[1], x1[0], x1[1]};
            o = mfma_f16(al[kb], xb, o);
            o = mfma_f16(ah[kb], xb, o);
        }
        if (c < TF) {
            float* dst = part_dW1 + (size_t)b * TH * TF + (size_t)(16 * wave + 4 * q) * TF + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r * TF] = o[r] * inv_s;
        }
        TS(9, 13)
    }
    return !__syncthreads_or(out_of_fp16_range(mx));
}

// option "train_fused" = 3 forces the fallback for every board (tests)
__global__ __launch_bounds__(512) void train_board_split_kernel(const uint8_t* __restrict__ states72, const int64_t* __restrict__ order, int first,
                                                                TrunkParams tp, HeadParams hpm, const float* __restrict__ pi_all,
                                                                const float* __restrict__ z_all, int A, int B, int force_fallback,
                                                                float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ g_out,
                                                                float* __restrict__ hp, float* __restrict__ hv, float* __restrict__ lg,
                                                                float* __restrict__ pol, float* __restrict__ vp, float* __restrict__ val,
                                                                float* __restrict__ loss, float* __restrict__ dhp, float* __restrict__ dhv,
                                                                float* __restrict__ part_dW3, float* __restrict__ part_dW2,
                                                                float* __restrict__ part_dW1, float* __restrict__ part_db) {
    __shared__ __align__(16) unsigned char smem[SPLIT_KERNEL_SMEM];
    const bool ok = train_board_split_body(smem, states72, order, first, tp, hpm, pi_all, z_all, A, B, h1, h2, g_out, hp, hv, lg, pol, vp, val,
                                           loss, dhp, dhv, part_dW3, part_dW2, part_dW1, part_db);
    if (ok && !force_fallback) return;
    if (threadIdx.x == 0) atomicAdd(&g_train_fallbacks, 1u);
    __syncthreads();
    train_board_f32_body<9>(smem, states72, order, first, tp, hpm, pi_all, z_all, A, B, 96, h1, h2, g_out, hp, hv, lg, pol, vp, val, loss,
                            dhp, dhv, part_dW3, part_dW2, part_dW1, part_db);
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
// (the old state is fetched by adam_fetch() BEFORE the gradient's own loads: one memory round trip per workgroup instead of two)
struct AdamOld { float m, v, p; };
__device__ __forceinline__ AdamOld adam_fetch(const FinalJobs& jb, int i, unsigned int e) { return AdamOld{jb.m[i][e], jb.v[i][e], jb.p[i][e]}; }
__device__ __forceinline__ void adam_update(const FinalJobs& jb, int i, unsigned int e, float gr, const AdamOld& o) {
    const float mi = jb.beta1 * o.m + (1.f - jb.beta1) * gr;          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = jb.beta2 * o.v + (1.f - jb.beta2) * gr * gr;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    jb.m[i][e] = mi; jb.v[i][e] = vi;
    const float denom = sqrtf(vi) / jb.bc2_sqrt + jb.eps;
    jb.p[i][e] = o.p - (jb.lr / jb.bc1) * (mi / denom);
}
// A team = 32 lanes x FINAL_GROUPS board groups: a thread sums its group's boards in order, the group sums are added pairwise in
// group order -- a fixed summation order with FINAL_GROUPS x the loads in flight of one thread per element.  The first
// FINAL_BIG_BLOCKS teams ("rows") take the two [128,128] trunk weights four elements per lane (16-byte loads of the 16 MB of
// per-board partials); the rest take every other tensor one element per lane (end[] counts those tensors only).  A workgroup =
// FINAL_TEAMS teams = 1,024 threads.
constexpr int FINAL_BIG_BLOCKS = 2 * TH * TH / 128;
#ifndef AQG_FINAL_GROUPS
#define AQG_FINAL_GROUPS 4          // board groups per element: a thread sums B / groups boards (4 / 8 / 16 groups: 0.0467 / 0.0473 / 0.0527 ms per step)
#endif
constexpr int FINAL_GROUPS = AQG_FINAL_GROUPS, FINAL_TEAM_THREADS = 32 * FINAL_GROUPS;
constexpr int FINAL_TEAMS = 1024 / FINAL_TEAM_THREADS;
__global__ __launch_bounds__(FINAL_TEAM_THREADS * FINAL_TEAMS) void train_final_kernel(FinalJobs jb) {
    __shared__ f32x4 red4s[FINAL_TEAMS][FINAL_GROUPS][33];
    TS_DECL
    const int team = threadIdx.x / FINAL_TEAM_THREADS, tt = threadIdx.x % FINAL_TEAM_THREADS;
    const unsigned int row = blockIdx.x * FINAL_TEAMS + team;
    f32x4 (*red4)[33] = red4s[team];
    const int le = tt & 31, grp = tt >> 5;
    const int B = jb.B, A = jb.A;
    const int per = (B + FINAL_GROUPS - 1) / FINAL_GROUPS, b0 = grp * per, b1 = min(B, b0 + per);
    if (row < FINAL_BIG_BLOCKS) {
        const unsigned int q4 = row * 32 + le;                    // float4 index over gcn1.w then gcn2.w
        const int i = q4 < TH * TH / 4 ? 2 : 4;
        const unsigned int e = (q4 & (TH * TH / 4 - 1)) * 4;
        f32x4 gr4;
        AdamOld old[4];
        if (jb.update && grp == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) old[k] = adam_fetch(jb, i, e + k);
        }
        if (jb.compute) {
            const float* src = jb.part_dW[i >> 1] + e;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
            for (int b = b0; b < b1; ++b) s += ld4(src + (size_t)b * TH * TH);
            red4[grp][le] = s;
            __syncthreads();
            TS(6, 1)
            if (grp != 0) return;
            gr4 = (red4[0][le] + red4[1][le]) + (red4[2][le] + red4[3][le]);
            if (FINAL_GROUPS >= 8) gr4 += (red4[4][le] + red4[5][le]) + (red4[6][le] + red4[7][le]);
            if (FINAL_GROUPS == 16) gr4 += ((red4[8][le] + red4[9][le]) + (red4[10][le] + red4[11][le])) + ((red4[12][le] + red4[13][le]) + (red4[14][le] + red4[15][le]));
            st4(jb.g[i] + e, gr4);
        } else {
            if (grp != 0) return;
            gr4 = ld4(jb.g[i] + e);
        }
        if (jb.update) {
#pragma unroll
            for (int k = 0; k < 4; ++k) adam_update(jb, i, e + k, gr4[k], old[k]);
        }
        return;
    }
    float (*red)[33] = reinterpret_cast<float (*)[33]>(&red4[0][0]);
    const unsigned int e0 = (row - FINAL_BIG_BLOCKS) * 32 + le;
    const unsigned int total = jb.end[13] + (jb.loss_sums ? 2u : 0u);
    const bool live = e0 < total;
    int i = 0;
    if (live) while (i < 14 && e0 >= jb.end[i]) ++i;
    const unsigned int e = e0 - (i ? jb.end[i - 1] : 0u);
    TS(6, 0)
    AdamOld old{0.f, 0.f, 0.f};
    if (jb.update && grp == 0 && live && i < 14) old = adam_fetch(jb, i, e);
    if (jb.compute) {
        float s = 0.f;
        if (!live) {
        } else if (i < 6) {
            if (i & 1) {
                const float* src = jb.part_db[i >> 1] + e;
#pragma unroll 16
                for (int b = b0; b < b1; ++b) s += src[(size_t)b * TH];
            } else {
                const float* src = jb.part_dW[0] + e;                                                                      // gcn0.w
#pragma unroll 16
                for (int b = b0; b < b1; ++b) s += src[(size_t)b * TH * TF];
            }
        } else if (i == 6 || i == 10) {
            const int j = e / TH, k = e % TH;
            const float* d = (i == 6 ? jb.dhp : jb.dhv) + j;
            const float* x = jb.gp + k;
#pragma unroll 16
            for (int b = b0; b < b1; ++b) s = fmaf(d[(size_t)b * HH], x[(size_t)b * TH], s);
        } else if (i == 7 || i == 11) {
            const float* d = (i == 7 ? jb.dhp : jb.dhv) + e;
#pragma unroll 16
            for (int b = b0; b < b1; ++b) s += d[(size_t)b * HH];
        } else if (i == 8) {
            const int a = e / HH, j = e % HH;
#pragma unroll 16
            for (int b = b0; b < b1; ++b) s = fmaf(jb.dlg[(size_t)b * A + a], jb.hp[(size_t)b * HH + j], s);
        } else if (i == 9) {
#pragma unroll 16
            for (int b = b0; b < b1; ++b) s += jb.dlg[(size_t)b * A + e];
        } else if (i == 12) {
#pragma unroll 16
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
        gr = (red[0][le] + red[1][le]) + (red[2][le] + red[3][le]);
        if (FINAL_GROUPS >= 8) gr += (red[4][le] + red[5][le]) + (red[6][le] + red[7][le]);
        if (FINAL_GROUPS == 16) gr += ((red[8][le] + red[9][le]) + (red[10][le] + red[11][le])) + ((red[12][le] + red[13][le]) + (red[14][le] + red[15][le]));
        if (i == 14) { jb.loss_sums[e] += gr / (float)B; return; }
        jb.g[i][e] = gr;
    } else {
        if (i == 14) return;
        gr = jb.g[i][e];
    }
    if (jb.update) adam_update(jb, i, e, gr, old);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// aqg_set_option("train_fused"): 2 (default) = one workgroup per position, contractions in fp16 split precision on the 9x9 board (other
// boards: as 1);  1 = one workgroup per position, f32-input MFMA;  0 = six launches;  3 = as 2 with every board sent through the
// f32 fallback (tests)
int g_train_fused = 2;
long long train_fallbacks(int reset) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_train_fallbacks), sizeof(v)) != hipSuccess) return -1;
    if (reset) { const unsigned int z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_train_fallbacks), &z, sizeof(z)) != hipSuccess) return -1; }
    return (long long)v;
}

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
        if (N == 9 && g_train_fused >= 2)      // h1 / h2 hold 96 rows per board here: the parked fragments (and the fallback's rows)
            hipLaunchKernelGGL(train_board_split_kernel, dim3(B), dim3(512), 0, st, states72, order, first, tpm, hpm, pi, z, A, B, g_train_fused == 3 ? 1 : 0,
                               t.h1, t.h2, t.g, t.hp, t.hv, t.lg, t.pol, t.vp, t.val, t.loss, t.dhp, t.dhv, pdW3, pdW2, pdW1, pdb);
        else
            hipLaunchKernelGGL(train_board_kernel<N>, dim3(B), dim3(512), 0, st, states72, order, first, tpm, hpm, pi, z, A, B, N * N, t.h1, t.h2, t.g,
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
    const unsigned int rows = FINAL_BIG_BLOCKS + (run + 2 + 31) / 32;
    hipLaunchKernelGGL(train_final_kernel, dim3((rows + FINAL_TEAMS - 1) / FINAL_TEAMS), dim3(FINAL_TEAM_THREADS * FINAL_TEAMS), 0, st, jb);
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

#ifdef AQG_TRAIN_DEBUG
extern "C" int aqg_debug_train_buf(float* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_train_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#endif
#ifdef AQG_STAMP
extern "C" int aqg_debug_train_stamps(unsigned long long* out_host, int reset) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_train_stamp), sizeof(unsigned long long) * 160) != hipSuccess) return -1;
    if (reset) { unsigned long long z[160] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_train_stamp), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

}  // namespace aqg
