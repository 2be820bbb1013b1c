// gcn_train.hip -- one optimisation step of the reference's training loop on the GNN, fp32, for gfx950.
//
// train_network.py:68-95 (forward, CrossEntropyLoss on the ALREADY-softmaxed policy + MSELoss on the
// tanh value, backward, Adam) applied to GraphPolicyValueNetwork (pv_network_gnn.py:23-64, GCNConv = PyG defaults).
// The batch is 128 positions (train_network.py:15), i.e. 10,368 graph nodes: a launch-bound, L2-resident workload, so
// this file is deliberately a chain of small, plain kernels (LDS-tiled f32 GEMMs, ELL aggregation with the fixed
// <= 5-regular board graph, deterministic column reductions -- no atomics anywhere, results are run-to-run identical)
// working directly on the state_dict tensors in their PyTorch layouts.
//
//   forward   Z = H W^T, P = A_hat Z + b, H' = relu(P)   x3;  g = mean_nodes H3;  heads;  pol = softmax, val = tanh
//   loss      Lp = mean_b -sum_a t_a log_softmax(pol)_a   (the reference's double softmax, kept on purpose)
//             Lv = mean_b (val - z)^2
//   backward  dP = dH' (.) [H' > 0];  db = colsum dP;  dZ = A_hat dP (A_hat symmetric);  dW = dZ^T H;  dH = dZ W
//   update    torch.optim.Adam (lr, betas, eps; bias-corrected; no weight decay, no amsgrad)
#include "aqg_common.hpp"
#include "../../include/aqgnn.h"
#include <rocblas/rocblas.h>

namespace aqg {

constexpr int TH = 128;    // HIDDEN_DIM
constexpr int TF = 6;      // NUM_FEATURES
constexpr int ELL = 5;     // self, U, D, L, R

// ---------------------------------------------------------------------------------------------
// board records -> node features [B*V][6] and the normalised adjacency in ELL form [B*V][5]
// (pv_network_cnn.py:88-114 features; edges = open tile adjacencies, game_logic.py:145-167; PyG gcn_norm weights)
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void train_prep_kernel(const uint8_t* __restrict__ states72, int B, float* __restrict__ x0,
                                                         int32_t* __restrict__ ell_idx, float* __restrict__ ell_w) {
    constexpr int V = N * N, S = N - 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * V) return;
    const int b = i / V, t = i % V;
    const QState s = unpack72(states72 + (size_t)b * STATE72);
    const int x = t / N, y = t % N;
    const bool slot_ok = x < S && y < S;
    const int slot = x * S + y;
    float* f = x0 + (size_t)i * TF;
    f[0] = (t == s.ppos) ? 1.f : 0.f;
    f[1] = (float)s.pwl;
    f[2] = (t == s.epos) ? 1.f : 0.f;
    f[3] = (float)s.ewl;
    f[4] = (slot_ok && ((s.hw >> slot) & 1)) ? 1.f : 0.f;
    f[5] = (slot_ok && ((s.vw >> slot) & 1)) ? 1.f : 0.f;
    const int ob = tile_open_bits<N>(s.hw, s.vw, t);
    const float di = 1.0f / sqrtf((float)(1 + __popc(ob)));
    const int nb[4] = {t - N, t + N, t - 1, t + 1};
    ell_idx[(size_t)i * ELL] = i;
    ell_w[(size_t)i * ELL] = di * di;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const bool open = (ob >> d) & 1;
        float w = 0.f;
        int j = -1;
        if (open) {
            const int obn = tile_open_bits<N>(s.hw, s.vw, nb[d]);
            w = di * (1.0f / sqrtf((float)(1 + __popc(obn))));
            j = b * V + nb[d];
        }
        ell_idx[(size_t)i * ELL + 1 + d] = j;
        ell_w[(size_t)i * ELL + 1 + d] = w;
    }
}

// ---------------------------------------------------------------------------------------------
// Y[r][c] (+)= sum_s X[r][s] * W[c*sc + s*ss]  (+ bias[c]) (relu)      X: [R][S] row-major, Y: [R][C] row-major
//   forward linear:  W = weight [C][S]  -> sc = S, ss = 1;    data gradient: W = weight [S][C] -> sc = 1, ss = C
// ---------------------------------------------------------------------------------------------
template <bool RELU, bool ACC>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ X, int R, int S, const float* __restrict__ W, int sc,
                                                   int ss, const float* __restrict__ bias, int C, float* __restrict__ Y) {
    __shared__ float xs[32][33], ws[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // ty 0..7: rows ty, ty+8, ty+16, ty+24
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = ty + 8 * i;
            xs[rr][tx] = (r0 + rr < R && s0 + tx < S) ? X[(size_t)(r0 + rr) * S + s0 + tx] : 0.f;
            ws[rr][tx] = (c0 + rr < C && s0 + tx < S) ? W[(size_t)(c0 + rr) * sc + (size_t)(s0 + tx) * ss] : 0.f;   // ws[c][s]
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float w = ws[tx][s];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(xs[ty + 8 * i][s], w, acc[i]);
        }
        __syncthreads();
    }
    const int c = c0 + tx;
    if (c < C) {
        const float bv = bias ? bias[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty + 8 * i;
            if (r < R) {
                float v = acc[i] + bv;
                if (ACC) v += Y[(size_t)r * C + c];
                if (RELU) v = fmaxf(v, 0.f);
                Y[(size_t)r * C + c] = v;
            }
        }
    }
}

// weight gradient: G[a][b] = sum_r A[r][a] * Bm[r][b]     A: [R][Ja], Bm: [R][Kb], G: [Ja][Kb].
// The row range is cut into gridDim.z slices (the 10,368-row trunk gradients would otherwise run on 16 workgroups);
// slice z writes its partial sum to G + z * Ja * Kb and reduce_partials_kernel adds the slices in fixed order.
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ A, int Ja, const float* __restrict__ Bm, int Kb, int R,
                                                    int rows_per_slice, float* __restrict__ G) {
    __shared__ float as[32][33], bs[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int rbeg = blockIdx.z * rows_per_slice, rend = min(R, rbeg + rows_per_slice);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = rbeg; r0 < rend; r0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = ty + 8 * i;
            as[rr][tx] = (r0 + rr < rend && a0 + tx < Ja) ? A[(size_t)(r0 + rr) * Ja + a0 + tx] : 0.f;
            bs[rr][tx] = (r0 + rr < rend && b0 + tx < Kb) ? Bm[(size_t)(r0 + rr) * Kb + b0 + tx] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const float bv = bs[r][tx];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(as[r][ty + 8 * i], bv, acc[i]);
        }
        __syncthreads();
    }
    float* Gz = G + (size_t)blockIdx.z * Ja * Kb;
    if (b0 + tx < Kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (a0 + ty + 8 * i < Ja) Gz[(size_t)(a0 + ty + 8 * i) * Kb + b0 + tx] = acc[i];
    }
}

// out[c] = sum_r A[r][c]    (bias gradients); 32 columns x 8 row lanes per workgroup, gridDim.y row slices
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int R, int C, int rows_per_slice, float* __restrict__ out) {
    __shared__ float part[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const int rbeg = blockIdx.y * rows_per_slice, rend = min(R, rbeg + rows_per_slice);
    float s = 0.f;
    if (c < C) for (int r = rbeg + ty; r < rend; r += 8) s += A[(size_t)r * C + c];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += part[i][tx];
        out[(size_t)blockIdx.y * C + c] = t;
    }
}

// All sliced reductions of a step are finished by ONE launch: job j sums `slices` partial arrays of n floats starting
// at part + off into out, slices added in index order (deterministic).  blockIdx.y = job.
struct ReduceJobs {
    int count;
    unsigned long long off[14];
    unsigned int n[14];
    int slices[14];
    float* out[14];
};
__global__ void reduce_jobs_kernel(const float* __restrict__ part, ReduceJobs jobs) {
    const int j = blockIdx.y;
    const unsigned int n = jobs.n[j];
    const float* src = part + jobs.off[j];
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < jobs.slices[j]; ++z) s += src[(size_t)z * n + i];
        jobs.out[j][i] = s;
    }
}

// ELL aggregation, one wavefront per node (2 columns per lane): out[n] = sum_s w[n][s] * Z[idx[n][s]] (+ bias) (relu)
template <bool RELU>
__global__ __launch_bounds__(256) void agg_kernel(const float* __restrict__ Z, int num_nodes, const int32_t* __restrict__ idx,
                                                  const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= num_nodes) return;
    float a0 = bias ? bias[2 * lane] : 0.f, a1 = bias ? bias[2 * lane + 1] : 0.f;
#pragma unroll
    for (int s = 0; s < ELL; ++s) {
        const int j = idx[(size_t)n * ELL + s];
        if (j >= 0) {
            const float we = w[(size_t)n * ELL + s];
            const float2 z = *reinterpret_cast<const float2*>(Z + (size_t)j * TH + 2 * lane);
            a0 = fmaf(we, z.x, a0);
            a1 = fmaf(we, z.y, a1);
        }
    }
    if (RELU) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); }
    *reinterpret_cast<float2*>(out + (size_t)n * TH + 2 * lane) = make_float2(a0, a1);
}

// d[i] = (h[i] > 0) ? d[i] : 0     (ReLU backward, in place)
__global__ void relu_mask_kernel(float* __restrict__ d, const float* __restrict__ h, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(h[i] > 0.f)) d[i] = 0.f;
}

// g[b][j] = mean over the V nodes of board b          /          dh[b*V + n][j] = dg[b][j] / V
__global__ __launch_bounds__(128) void pool_fwd_kernel(const float* __restrict__ h, int V, float* __restrict__ g) {
    const int b = blockIdx.x, j = threadIdx.x;
    float s = 0.f;
    for (int n = 0; n < V; ++n) s += h[((size_t)b * V + n) * TH + j];
    g[(size_t)b * TH + j] = s / (float)V;
}
__global__ __launch_bounds__(128) void pool_bwd_kernel(const float* __restrict__ dg, int V, float* __restrict__ dh) {
    const int b = blockIdx.x, j = threadIdx.x;
    const float v = dg[(size_t)b * TH + j] / (float)V;
    for (int n = 0; n < V; ++n) dh[((size_t)b * V + n) * TH + j] = v;
}

// per board: pol = softmax(logits); value = tanh(vp); loss terms; gradients wrt logits / pre-tanh value (in place)
//   train_network.py:54,85: CrossEntropyLoss(policy_pred, policy_target) with policy_pred ALREADY softmaxed
//   (pv_network_gnn.py:42,62) and probability targets: l_b = -sum_a t_a log_softmax(pol)_a, mean over the batch
//   train_network.py:55,86: MSELoss(value_pred.squeeze(), value_target), mean over the batch
__global__ __launch_bounds__(256) void loss_kernel(float* __restrict__ lg, float* __restrict__ pol, float* __restrict__ vp,
                                                   float* __restrict__ val, const float* __restrict__ pi, const float* __restrict__ z,
                                                   int A, int B, float* __restrict__ loss) {
    __shared__ float red[256];
    const int b = blockIdx.x, t = threadIdx.x;
    auto block_max = [&](float v) { red[t] = v; __syncthreads(); for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmaxf(red[t], red[t + o]); __syncthreads(); } const float r = red[0]; __syncthreads(); return r; };
    auto block_sum = [&](float v) { red[t] = v; __syncthreads(); for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); } const float r = red[0]; __syncthreads(); return r; };
    const bool on = t < A;
    const float l = on ? lg[(size_t)b * A + t] : -INFINITY;
    const float m = block_max(l);
    const float e = on ? expf(l - m) : 0.f;
    const float p = e / block_sum(e);                            // first softmax (the network's own, pv_network_gnn.py:42)
    const float e2 = on ? expf(p) : 0.f;                         // second softmax inside CrossEntropyLoss; p in [0,1]: no shift needed
    const float s2 = block_sum(e2);
    const float q = e2 / s2;
    const float tgt = on ? pi[(size_t)b * A + t] : 0.f;
    const float tsum = block_sum(tgt);
    const float lp = block_sum(on ? -tgt * (p - logf(s2)) : 0.f);
    const float dpol = on ? (q * tsum - tgt) / (float)B : 0.f;   // d(mean_b l_b) / d pol
    const float dot = block_sum(dpol * p);
    if (on) {
        pol[(size_t)b * A + t] = p;
        lg[(size_t)b * A + t] = p * (dpol - dot);                // back through the first softmax
    }
    if (t == 0) {
        const float v = tanhf(vp[b]);
        const float d = v - z[b];
        val[b] = v;
        vp[b] = (2.f * d / (float)B) * (1.f - v * v);
        loss[2 * b] = lp;
        loss[2 * b + 1] = d * d;
    }
}

// torch.optim.Adam.step() for all 14 tensors in one launch (no weight decay, no amsgrad); bias corrections computed on
// the host in f64.  blockIdx.y = tensor.
struct AdamJobs {
    float* p[14]; const float* g[14]; float* m[14]; float* v[14];
    unsigned int n[14];
};
__global__ void adam_kernel(AdamJobs jobs, float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt) {
    const int j = blockIdx.y;
    const unsigned int n = jobs.n[j];
    float* __restrict__ p = jobs.p[j];
    const float* __restrict__ g = jobs.g[j];
    float* __restrict__ m = jobs.m[j];
    float* __restrict__ v = jobs.v[j];
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static void gemm(hipStream_t st, bool relu, bool acc, const float* X, int R, int S, const float* W, int sc, int ss, const float* bias,
                 int C, float* Y) {
    const dim3 grid((R + 31) / 32, (C + 31) / 32), block(256);
    if (relu && !acc) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, st, X, R, S, W, sc, ss, bias, C, Y);
    else if (!relu && acc) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, st, X, R, S, W, sc, ss, bias, C, Y);
    else hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, st, X, R, S, W, sc, ss, bias, C, Y);
}
constexpr int TRAIN_SLICE_ROWS = 256;           // rows per partial sum
constexpr int TRAIN_MAX_SLICES = 64;
constexpr size_t TRAIN_PART_FLOATS = 64 * (2 * 128 * 128 + 128 * 8 + 3 * 128);   // aqg_train.part: every partial array of one step at 64 slices

struct PartialSums {                            // bump allocator over aqg_train.part + the list of pending reductions
    float* part;
    size_t used = 0;
    ReduceJobs jobs{};
    float* take(size_t n, int slices, float* out) {
        float* dst = part + used;
        jobs.off[jobs.count] = used; jobs.n[jobs.count] = (unsigned int)n; jobs.slices[jobs.count] = slices; jobs.out[jobs.count] = out;
        ++jobs.count;
        used += n * slices;
        return dst;
    }
};

static void wgrad(hipStream_t st, PartialSums& ps, const float* A, int Ja, const float* Bm, int Kb, int R, float* G) {
    int slices = (R + TRAIN_SLICE_ROWS - 1) / TRAIN_SLICE_ROWS;
    if (slices > TRAIN_MAX_SLICES) slices = TRAIN_MAX_SLICES;
    const int rows = ((R + slices - 1) / slices + 31) / 32 * 32;
    const dim3 grid((Ja + 31) / 32, (Kb + 31) / 32, slices);
    float* dst = slices == 1 ? G : ps.take((size_t)Ja * Kb, slices, G);
    hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, st, A, Ja, Bm, Kb, R, rows, dst);
}
static void colsum(hipStream_t st, PartialSums& ps, const float* A, int R, int C, float* out) {
    int slices = (R + TRAIN_SLICE_ROWS - 1) / TRAIN_SLICE_ROWS;
    if (slices > TRAIN_MAX_SLICES) slices = TRAIN_MAX_SLICES;
    const int rows = ((R + slices - 1) / slices + 7) / 8 * 8;
    const dim3 grid((C + 31) / 32, slices);
    float* dst = slices == 1 ? out : ps.take((size_t)C, slices, out);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, A, R, C, rows, dst);
}

// The three GEMM shapes over all B*V node rows -- Z = H W^T, dH = dZ W, dW = dZ^T H -- are plain dense f32 GEMMs with
// nothing to fuse: they go to rocBLAS (f32 MFMA kernels; atomics off, so results stay run-to-run identical).  Row-major
// operands are handed over as their column-major transposes.
static rocblas_handle blas_handle(hipStream_t st) {
    static rocblas_handle h = nullptr;
    if (!h) {
        if (rocblas_create_handle(&h) != rocblas_status_success) { h = nullptr; return nullptr; }
        rocblas_set_atomics_mode(h, rocblas_atomics_not_allowed);
        rocblas_set_pointer_mode(h, rocblas_pointer_mode_host);
    }
    rocblas_set_stream(h, st);
    return h;
}
// Z[R][C] = X[R][S] * W[C][S]^T
static int blas_forward(hipStream_t st, const float* X, int R, int S, const float* W, int C, float* Z) {
    rocblas_handle h = blas_handle(st);
    const float one = 1.f, zero = 0.f;
    if (!h || rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, C, R, S, &one, W, S, X, S, &zero, Z, C) != rocblas_status_success)
        return fail("rocblas_sgemm (forward)");
    return 0;
}
// dX[R][K] = dY[R][J] * W[J][K]
static int blas_dgrad(hipStream_t st, const float* dY, int R, int J, const float* W, int K, float* dX) {
    rocblas_handle h = blas_handle(st);
    const float one = 1.f, zero = 0.f;
    if (!h || rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, K, R, J, &one, W, K, dY, J, &zero, dX, K) != rocblas_status_success)
        return fail("rocblas_sgemm (data gradient)");
    return 0;
}
// dW[J][K] = dY[R][J]^T * X[R][K]
static int blas_wgrad(hipStream_t st, const float* dY, int J, const float* X, int K, int R, float* dW) {
    rocblas_handle h = blas_handle(st);
    const float one = 1.f, zero = 0.f;
    if (!h || rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, K, J, R, &one, X, K, dY, J, &zero, dW, K) != rocblas_status_success)
        return fail("rocblas_sgemm (weight gradient)");
    return 0;
}

// parameter order = state_dict order (KEYS in INTEGRATION.md):
//  0 gcn0.w [H,F]  1 gcn0.b  2 gcn1.w [H,H]  3 gcn1.b  4 gcn2.w  5 gcn2.b
//  6 pol0.w [H/2,H]  7 pol0.b  8 pol2.w [A,H/2]  9 pol2.b  10 val0.w [H/2,H]  11 val0.b  12 val2.w [1,H/2]  13 val2.b
static int train_gradients(const aqg_train& t, const uint8_t* states72, const float* pi, const float* z, hipStream_t st) {
    const int N = t.board_size, B = t.batch, A = t.policy_size;
    const int V = N * N, R = B * V, H2 = TH / 2;
    float* const* P = t.params;
    float* const* G = t.grads;
    // ---- forward
#define CALL_PREP(n) hipLaunchKernelGGL(train_prep_kernel<n>, dim3((R + 255) / 256), dim3(256), 0, st, states72, B, t.x0, t.ell_idx, t.ell_w)
    switch (N) { case 3: CALL_PREP(3); break; case 5: CALL_PREP(5); break; case 7: CALL_PREP(7); break; default: CALL_PREP(9); break; }
    const dim3 ag((R + 3) / 4), ab(256);
    if (int r = blas_forward(st, t.x0, R, TF, P[0], TH, t.zbuf)) return r;
    hipLaunchKernelGGL(agg_kernel<true>, ag, ab, 0, st, (const float*)t.zbuf, R, (const int32_t*)t.ell_idx, (const float*)t.ell_w, (const float*)P[1], t.h1);
    if (int r = blas_forward(st, t.h1, R, TH, P[2], TH, t.zbuf)) return r;
    hipLaunchKernelGGL(agg_kernel<true>, ag, ab, 0, st, (const float*)t.zbuf, R, (const int32_t*)t.ell_idx, (const float*)t.ell_w, (const float*)P[3], t.h2);
    if (int r = blas_forward(st, t.h2, R, TH, P[4], TH, t.zbuf)) return r;
    hipLaunchKernelGGL(agg_kernel<true>, ag, ab, 0, st, (const float*)t.zbuf, R, (const int32_t*)t.ell_idx, (const float*)t.ell_w, (const float*)P[5], t.h3);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(B), dim3(128), 0, st, (const float*)t.h3, V, t.g);
    gemm(st, true, false, t.g, B, TH, P[6], TH, 1, P[7], H2, t.hp);
    gemm(st, false, false, t.hp, B, H2, P[8], H2, 1, P[9], A, t.lg);
    gemm(st, true, false, t.g, B, TH, P[10], TH, 1, P[11], H2, t.hv);
    gemm(st, false, false, t.hv, B, H2, P[12], H2, 1, P[13], 1, t.vp);
    // ---- loss and its gradient wrt logits (t.lg) / pre-tanh value (t.vp), in place
    hipLaunchKernelGGL(loss_kernel, dim3(B), dim3(256), 0, st, t.lg, t.pol, t.vp, t.val, pi, z, A, B, t.loss);
    if (int r = check_launch("training forward kernels")) return r;
    // ---- backward: heads (every sliced reduction parks its partial sums in t.part; one launch finishes them all)
    PartialSums ps;
    ps.part = t.part;
    wgrad(st, ps, t.lg, A, t.hp, H2, B, G[8]);
    colsum(st, ps, t.lg, B, A, G[9]);
    gemm(st, false, false, t.lg, B, A, P[8], 1, H2, nullptr, H2, t.dhp);           // dhp = dlogits W_p2
    hipLaunchKernelGGL(relu_mask_kernel, dim3((B * H2 + 255) / 256), dim3(256), 0, st, t.dhp, (const float*)t.hp, (size_t)B * H2);
    wgrad(st, ps, t.dhp, H2, t.g, TH, B, G[6]);
    colsum(st, ps, t.dhp, B, H2, G[7]);
    gemm(st, false, false, t.dhp, B, H2, P[6], 1, TH, nullptr, TH, t.dg);           // dg = dhp W_p1
    wgrad(st, ps, t.vp, 1, t.hv, H2, B, G[12]);
    colsum(st, ps, t.vp, B, 1, G[13]);
    gemm(st, false, false, t.vp, B, 1, P[12], 1, H2, nullptr, H2, t.dhv);
    hipLaunchKernelGGL(relu_mask_kernel, dim3((B * H2 + 255) / 256), dim3(256), 0, st, t.dhv, (const float*)t.hv, (size_t)B * H2);
    wgrad(st, ps, t.dhv, H2, t.g, TH, B, G[10]);
    colsum(st, ps, t.dhv, B, H2, G[11]);
    gemm(st, false, true, t.dhv, B, H2, P[10], 1, TH, nullptr, TH, t.dg);           // dg += dhv W_v1
    // ---- backward: trunk
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(B), dim3(128), 0, st, (const float*)t.dg, V, t.dh);
    const size_t nel = (size_t)R * TH;
    const dim3 mg((unsigned)((nel + 255) / 256)), mb(256);
    const float* hin[3] = {t.x0, t.h1, t.h2};
    float* hout[3] = {t.h1, t.h2, t.h3};
    for (int L = 2; L >= 0; --L) {
        hipLaunchKernelGGL(relu_mask_kernel, mg, mb, 0, st, t.dh, (const float*)hout[L], nel);      // dP
        colsum(st, ps, t.dh, R, TH, G[2 * L + 1]);
        hipLaunchKernelGGL(agg_kernel<false>, ag, ab, 0, st, (const float*)t.dh, R, (const int32_t*)t.ell_idx, (const float*)t.ell_w,
                           (const float*)nullptr, t.zbuf);                                          // dZ = A_hat dP
        const int K = L == 0 ? TF : TH;
        if (int r = blas_wgrad(st, t.zbuf, TH, hin[L], K, R, G[2 * L])) return r;
        if (L > 0) { if (int r = blas_dgrad(st, t.zbuf, R, TH, P[2 * L], TH, t.dh)) return r; }   // dH_{L-1} = dZ W_L
    }
    if (ps.used > TRAIN_PART_FLOATS) return fail("training: partial-sum workspace too small for this batch");
    if (ps.jobs.count) hipLaunchKernelGGL(reduce_jobs_kernel, dim3(16, ps.jobs.count), dim3(256), 0, st, (const float*)t.part, ps.jobs);
    if (int r = check_launch("training backward kernels")) return r;
    return 0;
}

static int train_update(const aqg_train& t, hipStream_t st) {
    const int A = t.policy_size, H2 = TH / 2;
    const double bc1 = 1.0 - pow((double)t.beta1, (double)t.step), bc2 = 1.0 - pow((double)t.beta2, (double)t.step);
    const size_t sizes[14] = {(size_t)TH * TF, TH, (size_t)TH * TH, TH, (size_t)TH * TH, TH, (size_t)H2 * TH, (size_t)H2, (size_t)A * H2, (size_t)A,
                              (size_t)H2 * TH, (size_t)H2, (size_t)H2, 1};
    AdamJobs aj;
    for (int i = 0; i < 14; ++i) { aj.p[i] = t.params[i]; aj.g[i] = t.grads[i]; aj.m[i] = t.adam_m[i]; aj.v[i] = t.adam_v[i]; aj.n[i] = (unsigned int)sizes[i]; }
    hipLaunchKernelGGL(adam_kernel, dim3(16, 14), dim3(256), 0, st, aj, t.lr, t.beta1, t.beta2, t.eps, (float)bc1, (float)sqrt(bc2));
    return check_launch("adam_kernel");
}

// mode 0 = gradients only, 1 = gradients + Adam, 2 = Adam only (data-parallel: local gradients, all-reduce, update)
int train_step(const aqg_train& t, const uint8_t* states72, const float* pi, const float* z, int mode, hipStream_t st) {
    const int N = t.board_size, A = t.policy_size;
    if (!(N == 3 || N == 5 || N == 7 || N == 9)) return fail("board_size must be 3, 5, 7 or 9");
    if (A != N * N + 2 * (N - 1) * (N - 1) || A > 256) return fail("policy_size does not match the board");
    if (mode != 2 && t.batch > 0) {
        if (int r = train_gradients(t, states72, pi, z, st)) return r;
    }
    if (mode >= 1 && (t.batch > 0 || mode == 2)) {
        if (int r = train_update(t, st)) return r;
    }
    return 0;
}

}  // namespace aqg
