// mcts.hip -- K4: batched lock-step PV-MCTS + self-play move loop for gfx950.
//
// Semantics reproduced per game (exactly, given the same evaluator outputs and uniforms):
//   pv_mcts.py:20-95   Node/evaluate/next_child_node/pv_mcts_policy   (C_PUCT 1.25, first-max argmax,
//                      float32 PUCT arithmetic under NumPy-2 promotion, float64 w accumulators)
//   self_play.py:40-68 play(): record (state, visit distribution), sample like np.random.choice, next(), z.
//
// Layout: one 64-lane wavefront per game.  A game's tree is a flat pool of 32-byte reference-"Node" records
// (w, p, n, first-child|count, action); children of a node are contiguous, in State.legal_actions() order,
// so the PUCT arg-max is a strided wave reduction and "first maximum wins" is (max score, min index).
// Child STATES are never stored: the descent re-applies next() from the root's 24-byte packed state, so
// a node costs 32 bytes instead of the reference's full State copy.
// One simulation = fused step kernel (expand/backup of the previous leaf, select, legal actions of the new leaf)
// -> GNN trunk -> GNN heads on the leaf batch;
// every game has exactly one leaf in flight, so no virtual loss is needed and per-game semantics equal the
// sequential reference.
#define AQG_TRACE_TU mcts
#include "aqg_common.hpp"
#include <vector>
#include <cstring>
#include "legal_wave.hpp"
#include "../../include/aqgnn.h"

// PUCT scores must be evaluated exactly as written (no fma contraction, IEEE divide/sqrt).
#pragma clang fp contract(off)

namespace aqg {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

int launch_legal_actions(int N, const void* states, int fmt, int B, uint8_t* mask, uint8_t* order, int32_t* count,
                         const uint8_t* active, hipStream_t st);
int launch_gcn_forward_boards(int N, const void* states, int fmt, int B, const float* packed, float* pooled,
                              float* logits, float* policy, float* value_pre, float* value, const uint8_t* active,
                              int flags, int32_t* saturated, hipStream_t st, const int32_t* list = nullptr, const int32_t* list_count = nullptr);
size_t boards_any_workspace_floats(int N, int B);
int launch_gcn_forward_boards_any(int N, const void* states, int fmt, int B, const float* packed, float* workspace,
                                  size_t workspace_floats, float* pooled, float* logits, float* policy, float* value_pre,
                                  float* value, const uint8_t* active, int flags, int32_t* saturated, hipStream_t st, const int32_t* list = nullptr, const int32_t* list_count = nullptr);
extern int g_trunk_variant, g_trunk_grid, g_trunk_phase_delay, g_trunk_delay_min_boards, g_profile_trunk, g_trunk_prio, g_heads_prio;
void profile_mark(hipStream_t st, long long units);
int g_use_graph = 1;       // aqg_set_option("use_graph", 0) forces plain launches

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Wave-wide reductions on DPP (row operations inside the SIMD) instead of ds_bpermute shuffles through the LDS crossbar: the
// step kernel is one wavefront's dependent chain, and a six-round bpermute reduction costs it more than the tree level's
// arithmetic.  Result in an SGPR (lane 63 holds the total after the row_bcast steps).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWMASK, 0xf, false); }
__device__ __forceinline__ float wave_max_dpp(float x) {       // max over the 64 lanes (NaN entries are ignored, like v_max_f32)
    auto step = [](float v, int moved) { return fmaxf(v, __builtin_bit_cast(float, moved)); };
    x = step(x, dpp_i<0xB1, 0xf>(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x)));     // quad_perm [1,0,3,2]
    x = step(x, dpp_i<0x4E, 0xf>(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x)));     // quad_perm [2,3,0,1]
    x = step(x, dpp_i<0x141, 0xf>(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x)));    // row_half_mirror
    x = step(x, dpp_i<0x140, 0xf>(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x)));    // row_mirror: 16 lanes agree
    x = step(x, dpp_i<0x142, 0xa>(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x)));    // row_bcast15 -> rows 1, 3
    x = step(x, dpp_i<0x143, 0xc>(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x)));    // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
// The same maximum as six v_max_f32 with DPP operands (hipcc makes v_mov_dpp + a canonicalising v_max + v_max of each builtin step: 24
// instructions and their wait states on the step kernel's per-level chain).  A DPP operand needs two wait states behind the VALU write
// of its source: s_nop 1 between the steps (nothing is padded inside an asm statement).
__device__ __forceinline__ float wave_max_dpp_asm(float x) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ int wave_sum_dpp(int x) {
    x += dpp_i<0xB1, 0xf>(0, x);
    x += dpp_i<0x4E, 0xf>(0, x);
    x += dpp_i<0x141, 0xf>(0, x);
    x += dpp_i<0x140, 0xf>(0, x);
    x += dpp_i<0x142, 0xa>(0, x);
    x += dpp_i<0x143, 0xc>(0, x);
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
// a game's state is the same in every lane of its wavefront: as scalars, next() / is_lose() / is_draw() run on the scalar unit
__device__ __forceinline__ QState uniform_state(const QState& v) {
    QState s;
    s.hw = rfl64(v.hw); s.vw = rfl64(v.vw);
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)v.ppos | ((uint32_t)v.pwl << 8) | ((uint32_t)v.epos << 16) | ((uint32_t)v.ewl << 24)));
    s.ppos = (uint8_t)(m & 0xff); s.pwl = (uint8_t)((m >> 8) & 0xff); s.epos = (uint8_t)((m >> 16) & 0xff); s.ewl = (uint8_t)(m >> 24);
    s.plies = (uint16_t)__builtin_amdgcn_readfirstlane((int)v.plies); s.pad = 0;
    return s;
}

// One reference-"Node" (pv_mcts.py:24-31) per 32-byte record: the statistics, the prior, the action that led here and
// the child range sit in one cache sector, so a descent level is ONE dependent load round (the chosen child's
// `kids` and `action` arrive together with its w/n/p) of two aligned 16-byte loads per child.
struct alignas(32) NodeRec {
    // cold half (bytes 0..15): what a descent needs only of the child it CHOSE
    double w;          // cumulative value (python float in the reference)
    float p;           // prior
    uint32_t action;   // action that led to this node (0xFF for the root)
    // hot half (bytes 16..31): what PUCT scores every child with -- one aligned 16-byte load per child
    int32_t n;         // visit count
    uint32_t kids;     // first child (24 bits) | child count << 24 ; 0 = unexpanded
    float q;           // f32(-w / n) as PUCT adds it (pv_mcts.py:74), 0 while n == 0: maintained by every writer of (w, n), so the
                       // descent reads it with the record instead of doing a float64 division per tree level on its critical path
    float cp;          // f32(C_PUCT * p), the first product of PUCT's exploration term (pv_mcts.py:75, evaluated left to right in f32):
                       // written with p, so the descent's per-level chain starts one multiply later
};
static_assert(sizeof(NodeRec) == 32, "NodeRec must be 32 bytes");

// The exploitation term exactly as the reference forms it: python float division of the float64 sums, rounded to float32 where it
// meets the float32 exploration term (pv_mcts.py:74 under NumPy-2 promotion; pinned by the reference traces).
__device__ __forceinline__ float q_of(double w, int n) { return n ? (float)(-w / (double)n) : 0.0f; }

__device__ __forceinline__ NodeRec* game_nodes(const aqg_engine& e, int g) {
    return reinterpret_cast<NodeRec*>(e.node_rec) + (size_t)g * e.node_cap;
}

// Backup (pv_mcts.py:36-42,:49-50,:62-64): every node on the path gets w += value, n += 1 with the sign flipping
// per ply.  The path nodes are distinct, so lane d updates path[d] independently (one parallel step instead of a
// serial chain of dependent global read-modify-writes); the sums are the same float64 additions.
__device__ __forceinline__ void backup_path(NodeRec* __restrict__ nodes, const int* __restrict__ path, int depth,
                                            double leaf_value, int lane) {
    for (int d = lane; d <= depth; d += 64) {
        NodeRec& r = nodes[path[d]];
        r.w += ((depth - d) & 1) ? -leaf_value : leaf_value;
        r.n += 1;
        r.q = q_of(r.w, r.n);
    }
}

// ------------------------------------------------------------------------------------------------
// reset: every slot -> initial position (game_logic.py:25-40), active
// ------------------------------------------------------------------------------------------------
__global__ void engine_reset_kernel(aqg_engine e) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) {
        e.counters[0] = e.num_games;          // active slots
        e.counters[1] = 0;                    // finished games
        e.counters[2] = 0;                    // dead-end aborts
        e.counters[3] = e.num_games;          // next game index to hand out (slot refill)
        for (int i = 4; i < 8; ++i) e.counters[i] = 0;
    }
    if (g < e.quota) {                        // per-game records (quota >= num_games)
        e.game_plies[g] = 0;
        e.game_result[g] = 0;
        e.game_done[g] = 0;
        e.game_slot[g] = g < e.num_games ? g : -1;
        e.game_first_move[g] = 0;
    }
    if (g >= e.num_games) return;
    const int N = e.board_size;
    QState s;
    s.hw = 0; s.vw = 0;
    s.ppos = (uint8_t)(N * (N - 1) + N / 2); s.pwl = (uint8_t)e.num_walls;
    s.epos = s.ppos; s.ewl = s.pwl;
    s.plies = 0; s.pad = 0;
    store_state(e.root_state, g, s);
    e.game_active[g] = 1;
    e.slot_game[g] = g;
    e.node_count[g] = 0;
    e.leaf_flag[g] = 0;
    e.stat_leaf_evals[g] = 0;
    e.stat_terminal_sims[g] = 0;
    if (e.eval_cache_keys) { e.stat_cache_hits[g] = 0; e.eval_cache_slot[g] = -1; e.eval_mask[g] = 0; }
}

__global__ void engine_set_roots_kernel(aqg_engine e, const uint8_t* __restrict__ roots72) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.num_games) return;
    store_state(e.root_state, g, unpack72(roots72 + (size_t)g * STATE72));
    e.game_active[g] = 1;
    e.slot_game[g] = g;
    e.game_plies[g] = 0;
}

// ------------------------------------------------------------------------------------------------
// begin move: fresh tree per move (pv_mcts.py:81: no tree reuse)
// ------------------------------------------------------------------------------------------------
__global__ void engine_begin_move_kernel(aqg_engine e) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (e.eval_count) for (int i = g; i <= e.sims; i += gridDim.x * blockDim.x) e.eval_count[i] = 0;      // evaluation cache: entries of each simulation's list
    if (g >= e.num_games || !e.game_active[g]) return;
    NodeRec root;
    root.w = 0.0; root.p = 0.f; root.n = 0; root.kids = 0; root.action = 0xFF; root.q = 0.f; root.cp = 0.f;
    game_nodes(e, g)[0] = root;
    e.node_count[g] = 1;
    const int k = e.slot_game[g];                  // the game this slot is playing
    const int ply = e.game_plies[k];
    if (e.hist_visits && ply < e.max_plies) {      // clear this ply's dense visit row (filled by finish_move)
        const int A = e.board_size * e.board_size + 2 * (e.board_size - 1) * (e.board_size - 1);
        uint16_t* hv = e.hist_visits + ((size_t)k * e.max_plies + ply) * A;
        for (int a = 0; a < A; ++a) hv[a] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// select: descend by PUCT to a terminal node (back up at once) or to an unexpanded leaf (emit its state)
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void game_select(const aqg_engine& e, int g, int lane, int active, QState s) {
    if (!active) { if (lane == 0) e.leaf_flag[g] = 0; return; }
    if (lane == 0) e.leaf_flag[g] = 0;
    NodeRec* __restrict__ nodes = game_nodes(e, g);
    int* path = e.path + (size_t)g * (e.sims + 2);
    int node = 0, depth = 0;
    int mynode = 0;                      // lane d keeps the path node at depth d (d < 64) in a register
    if (lane == 0) path[0] = 0;
    int terminal = 0;
    double value = 0.0;
    uint32_t kids = nodes[0].kids;       // child range of the current node (wave-uniform)
    for (;;) {
        const bool lose = is_lose<N>(s), draw = is_draw(s, e.plies_for_draw);
        if (lose || draw) {                                    // pv_mcts.py:35-42
            value = lose ? -1.0 : 0.0;
            terminal = 1;
            break;
        }
        const int cnt = (int)(kids >> 24), first = (int)(kids & 0xFFFFFF);
        if (cnt == 0) break;                                   // pv_mcts.py:45 unexpanded leaf
        // pv_mcts.py:69-78 next_child_node: each lane reads up to 3 whole child records
        NodeRec rec[3];
        int t = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = lane + 64 * r;
            if (i < cnt) {
                rec[r] = nodes[first + i];
                t += rec[r].n;
            } else { rec[r].w = 0.0; rec[r].p = 0.f; rec[r].n = 0; rec[r].kids = 0; rec[r].action = 0; }
        }
        t = wave_sum_i(t);
        // f32(math.sqrt(t)): t < 2^24 is exact in f32 and the compiler's f32 square root is correctly rounded
        // (-fhip-fp32-correctly-rounded-divide-sqrt, the default), and rounding sqrt to 53 bits first never changes the 24-bit
        // result (a binary64 square root cannot land within half an ulp of a binary32 midpoint unless it IS one: 53 >= 2*24 + 2)
        // -- so the f64 Newton chain (14 dependent double-rate instructions per level) is not needed.  The reference traces pin it.
        const float st = sqrtf((float)t);
        float best = -INFINITY; int besti = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = lane + 64 * r;
            if (i < cnt) {
                const float u = ((e.c_puct * rec[r].p) * st) / (float)(1 + rec[r].n);
                const float q = rec[r].n ? (float)(-rec[r].w / (double)rec[r].n) : 0.0f;
                const float sc = q + u;
                if (sc > best) { best = sc; besti = i; }       // strict > keeps the lowest index within a lane
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {               // (max score, min index) across the wave
            const float ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(besti, off);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (besti == 0x7fffffff) besti = 0;                    // all-NaN guard (np.argmax would return 0)
        // the winner's kids / action live in lane (besti & 63), slot (besti >> 6)
        const int slot = besti >> 6, src = besti & 63;
        const uint32_t k_sel = slot == 0 ? rec[0].kids : (slot == 1 ? rec[1].kids : rec[2].kids);
        const uint32_t a_sel = slot == 0 ? rec[0].action : (slot == 1 ? rec[1].action : rec[2].action);
        kids = (uint32_t)__shfl((int)k_sel, src);
        const int action = __shfl((int)a_sel, src);
        node = first + besti;
        s = next_state<N>(s, action);
        ++depth;
        if (lane == 0) path[depth] = node;
        if (lane == (depth & 63) && depth < 64) mynode = node;
    }
    if (terminal) {
        // backup (pv_mcts.py:36-42): lane d updates the node at depth d from its register copy; the (practically
        // unreachable) part of a path deeper than 63 is finished by lane 0 from its own path[] stores
        if (lane <= depth) {
            NodeRec& r = nodes[mynode];
            r.w += ((depth - lane) & 1) ? -value : value;
            r.n += 1;
            r.q = q_of(r.w, r.n);
        }
        if (lane == 0) {
            e.stat_terminal_sims[g] += 1;
            for (int d = 64; d <= depth; ++d) {
                NodeRec& r = nodes[path[d]];
                r.w += ((depth - d) & 1) ? -value : value;
                r.n += 1;
                r.q = q_of(r.w, r.n);
            }
        }
    } else {
        // unexpanded leaf: its legal actions are computed right here by the same wavefront (one lane per wall slot)
        const int total = wave_legal_actions<N>(s, lane, nullptr, e.legal_order + (size_t)g * MAX_LEGAL);
        if (lane == 0) {
            store_state(e.leaf_state, g, s);
            e.legal_count[g] = total;
            e.path_len[g] = depth;
            e.leaf_flag[g] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// `fake` evaluator (tests): oracle/mcts.py FakeModel, exact integer hash -> f32 priors (written over the
// first `count` entries of policy[g]) and value.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fnv1a_state72(const uint8_t* r68, int plies) {
    uint32_t h = 0x811C9DC5u;
    for (int i = 0; i < 68; ++i) { h ^= r68[i]; h *= 0x01000193u; }
    h ^= (uint32_t)(plies & 0xFF); h *= 0x01000193u;
    h ^= (uint32_t)((plies >> 8) & 0xFF); h *= 0x01000193u;
    return h;
}

template <int N>
__global__ __launch_bounds__(256) void engine_fake_eval_kernel(aqg_engine e) {
    constexpr int V = N * N, A = Geo<N>::A;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= e.num_games || e.leaf_flag[g] != 1) return;
    const QState s = load_state(e.leaf_state, 1, g);
    uint8_t rec[STATE72];
    pack72(s, N, rec);
    const uint32_t h = fnv1a_state72(rec, s.plies);
    const int cnt = e.legal_count[g];
    const uint8_t* ord = e.legal_order + (size_t)g * MAX_LEGAL;
    const int prow = s.ppos / N;
    int rl[3]; int tot = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int i = lane + 64 * r;
        rl[r] = 0;
        if (i < cnt) {
            const int a = ord[i];
            uint32_t x = ((h ^ ((uint32_t)(a + 1) * 0x9E3779B1u)) * 0x85EBCA6Bu) >> 22;
            int rr = (int)x + 1;
            if (a < V && (a / N) < prow) rr *= 1 + e.fake_bias;
            rl[r] = rr; tot += rr;
        }
    }
    tot = wave_sum_i(tot);
    float* pol = e.policy + (size_t)g * A;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int i = lane + 64 * r;
        if (i < cnt) pol[i] = (float)rl[r] / (float)tot;
    }
    if (lane == 0) e.value[g] = (float)((int)((h * 0xC2B2AE35u) >> 16) - 32768) / 32768.0f;
}

// ------------------------------------------------------------------------------------------------
// expand + backup (pv_mcts.py:47-57, :60-66)
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void game_expand_backup(const aqg_engine& e, int g, int lane, int flag, int depth, int cnt, int first,
                                                   float leaf_value) {
    constexpr int A = Geo<N>::A;
    // Round 1 of loads: everything whose address depends on nothing but g and the lane -- the caller's five scalars, this
    // lane's legal-action bytes and its path entry -- is requested before the first branch, so the wave pays ONE memory
    // round trip here instead of one per dependent step (the whole kernel is a latency chain).
    NodeRec* __restrict__ nodes = game_nodes(e, g);
    const int* path = e.path + (size_t)g * (e.sims + 2);
    const uint8_t* ord = e.legal_order + (size_t)g * MAX_LEGAL;
    const float* pol = e.policy + (size_t)g * A;
    uint8_t oa[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { const int i = lane + 64 * r; oa[r] = (i < MAX_LEGAL) ? ord[i] : (uint8_t)0; }
    const int pnode = (lane < e.sims + 2) ? path[lane] : 0;           // path node at depth `lane`
    if (flag != 1) return;
    const int leaf = depth < 64 ? __shfl(pnode, depth) : path[depth];
    float pl[3];
    if (e.prior_mode == 0) {       // P0: gather at legal actions, divide by the sum unless 0 (pv_network_cnn.py:129-132)
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = lane + 64 * r;
            pl[r] = (i < cnt) ? pol[oa[r]] : 0.f;
            sum += pl[r];
        }
        sum = wave_sum_f(sum);
        const float den = (sum != 0.f) ? sum : 1.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) pl[r] = pl[r] / den;
    } else {                       // fake evaluator already wrote legal-ordered normalised priors
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = lane + 64 * r;
            pl[r] = (i < cnt) ? pol[i] : 0.f;
        }
    }
    if (first + cnt <= e.node_cap && cnt > 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = lane + 64 * r;
            if (i < cnt) {
                NodeRec c;
                c.w = 0.0; c.p = pl[r]; c.n = 0; c.kids = 0; c.action = oa[r]; c.q = 0.f; c.cp = e.c_puct * pl[r];
                nodes[first + i] = c;
            }
        }
    }
    if (lane == 0 && first + cnt <= e.node_cap && cnt > 0) {
        nodes[leaf].kids = (uint32_t)first | ((uint32_t)cnt << 24);
        e.node_count[g] = first + cnt;
    }
    // backup (pv_mcts.py:60-66): lane d updates the path node at depth d from its register copy; deeper parts of a path
    // (practically unreachable) go through memory
    {
        const double v = (double)leaf_value;                       // value.item() -> python float
        if (lane <= depth) {
            NodeRec& r = nodes[pnode];
            r.w += ((depth - lane) & 1) ? -v : v;
            r.n += 1;
            r.q = q_of(r.w, r.n);
        }
        for (int d = lane + 64; d <= depth; d += 64) {
            NodeRec& r = nodes[path[d]];
            r.w += ((depth - d) & 1) ? -v : v;
            r.n += 1;
            r.q = q_of(r.w, r.n);
        }
    }
    if (lane == 0) e.stat_leaf_evals[g] += 1;   // per-game slot: a shared counter would serialise 2048 atomics per step
}

// ------------------------------------------------------------------------------------------------
// The same step with the dependent memory rounds cut to the minimum (default; aqg_set_option("step_variant", 0) = the
// two functions above behind one workgroup-scope fence).  The step is a latency chain: above, every phase waits for
// its own loads -- scalars, legal list, policy gather, path, read-modify-write of the path nodes, root children, one
// round per tree level -- about ten dependent round trips to L2/HBM per simulation.  Here:
//   round 1   everything whose address follows from (game, lane) alone is requested at once: scalars, root state,
//             legal list, old path, the whole policy row, the root record AND the root's children (node 1 ...: the root
//             is expanded first in every move, so its children always start at node 1);
//   no store -> load dependency inside the kernel: the previous simulation's backup and expansion are APPLIED IN
//             REGISTERS to whatever the descent loads (a child on the old path gets w += +-v, n += 1 -- the same
//             float64 addition the store performs; the old leaf's children are the records just built), and written to
//             memory behind the descent.  Every load therefore sees the state the previous launch left, whatever the
//             timing, and the descent's only dependent rounds are the child blocks of levels >= 2;
//   the policy gather at the legal actions goes through 1 KB of LDS instead of a second global round.
// Identical arithmetic, identical visit order: bit-exact with the reference traces like the variant above.  Paths
// deeper than `fast_depth` (61; never seen) fall back to that variant mid-flight: pending updates are flushed, fenced,
// and the descent continues on memory (the tests run the goldens with fast_depth 1 and 2 to exercise every hand-over).
// ------------------------------------------------------------------------------------------------
// Diagnostic build only (-DAQG_STAMP, tools/stamp_step.py; never shipped): lane 0 of every game adds the cycles spent in each
// phase of the step to pooled[g][2 i .. 2 i + 1] as u64 (the fake-evaluator runs the tool uses never touch `pooled`).
#ifdef AQG_STAMP
#define STEP_STAMP_DECL unsigned long long sp_prev = __builtin_readcyclecounter(), sp_loc[5] = {0, 0, 0, 0, 0};
#define STEP_STAMP(i) { const unsigned long long sp_now = __builtin_readcyclecounter(); if (lane == 0) reinterpret_cast<unsigned long long*>(e.pooled + (size_t)g * 128)[i] += sp_now - sp_prev; sp_loc[i] = sp_now - sp_prev; sp_prev = sp_now; }
#define LEVEL_STAMP(i) { const unsigned long long lv_now = __builtin_readcyclecounter(); if (lane == 0) reinterpret_cast<unsigned long long*>(e.pooled + (size_t)g * 128)[i] += lv_now - lv_prev; lv_prev = lv_now; }
#else
#define STEP_STAMP_DECL
#define STEP_STAMP(i)
#endif
constexpr int EVAL_CACHE_ROW = 704;          // f32 priors[MAX_LEGAL] + u8 actions[MAX_LEGAL], padded to 64 bytes (aqgnn.h)
static_assert(MAX_LEGAL * 5 <= EVAL_CACHE_ROW && MAX_LEGAL % 4 == 0, "evaluation cache row");
__device__ __forceinline__ uint32_t eval_cache_misc(const QState& s) {
    return (uint32_t)s.ppos | ((uint32_t)s.pwl << 8) | ((uint32_t)s.epos << 16) | ((uint32_t)s.ewl << 24);
}

int g_step_prio = 1;               // wave priority of the fast step kernel (0..3)
int g_step_waves = 8;              // games (wavefronts) per workgroup of the fast step kernel (1, 2, 4 or 8).  Round 4: 8 -- at 96 registers two step
                                   // waves per SIMD fit beside one trunk workgroup, half as many workgroups: +0.5-0.8 % games/s at 2,048 and 16,384 games
int g_step_variant = 1;
int g_step_fast_depth = 61;

// CACHE: the evaluation cache's code is compiled in (its own kernel instantiation: the cache-less kernel carries none of it)
template <int N, bool CACHE>
__device__ __forceinline__ void game_step_fast(const aqg_engine& e, int g, int lane, int do_expand, int do_select, int fast_depth,
                                               float* __restrict__ polbuf /* this wave's 256 floats of LDS */, int list_sim) {
    constexpr int A = Geo<N>::A;
    NodeRec* __restrict__ nodes = game_nodes(e, g);
    int* path = e.path + (size_t)g * (e.sims + 2);
    const uint8_t* ord = e.legal_order + (size_t)g * MAX_LEGAL;
    const float* pol = e.policy + (size_t)g * A;
    STEP_STAMP_DECL
    // ---------------- round 1
    const int active = e.game_active[g];
    const QState s_loaded = load_state(e.root_state, 1, g);
    int flag = 0, depth_old = 0, cnt_new = 0, first_new = 0;
    float value = 0.f;
    uint8_t oa[3] = {0, 0, 0};
    int pnode = 0;
    float polr[4] = {0.f, 0.f, 0.f, 0.f};
    // evaluation cache (aqgnn.h, ABI 10): a per-slot table of the positions this slot's games have already sent through the network
    constexpr bool cache_on = CACHE;
    int cslot = -1;
    QState leaf_prev = s_loaded;      // the previous simulation's leaf (its key, when its evaluation goes into the table)
    if (do_expand) {
        flag = e.leaf_flag[g]; depth_old = e.path_len[g]; cnt_new = e.legal_count[g]; first_new = e.node_count[g]; value = e.value[g];
        if (cache_on) { cslot = e.eval_cache_slot[g]; leaf_prev = load_state(e.leaf_state, 1, g); }
#pragma unroll
        for (int r = 0; r < 3; ++r) { const int i = lane + 64 * r; oa[r] = (i < MAX_LEGAL) ? ord[i] : (uint8_t)0; }
        pnode = (lane < e.sims + 2) ? path[lane] : 0;
        if (e.prior_mode == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int a = lane + 64 * r; polr[r] = (a < A) ? pol[a] : 0.f; }
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) { const int i = lane + 64 * r; polr[r] = (i < MAX_LEGAL && i < A) ? pol[i] : 0.f; }
        }
    }
    const NodeRec rootrec = nodes[0];
    // (children travel as the record's two aligned 16-byte halves -- [2 i] = {w.lo, w.hi, p, action}, [2 i + 1] = {n, kids, q, cp} -- and
    //  stay vectors: as separate scalars their loop-carried copies were made behind an s_waitcnt at the descent loop's back edge)
    const u32x4* __restrict__ nhalf = reinterpret_cast<const u32x4*>(nodes);
    u32x4 hot[3], cold[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int i = min(1 + lane + 64 * r, e.node_cap - 1);
        cold[r] = nhalf[2 * i]; hot[r] = nhalf[2 * i + 1];
    }
    if (!do_expand) flag = 0;
    // (wave-uniform values the compiler cannot know to be uniform: as scalars they steer branches and v_readlane)
    flag = __builtin_amdgcn_readfirstlane(flag); depth_old = __builtin_amdgcn_readfirstlane(depth_old);
    cnt_new = __builtin_amdgcn_readfirstlane(cnt_new); first_new = __builtin_amdgcn_readfirstlane(first_new);
    // leaf_flag 2: the leaf was served from the evaluation cache -- policy[g][0 .. cnt) already holds the renormalised priors over its
    // legal actions in order (the layout of the other evaluator modes), legal_order / legal_count / value came with them
    const bool hit_old = flag == 2;
    if (hit_old) flag = 1;
    if (flag != 1 && !do_select) return;
#ifdef AQG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STEP_STAMP(0)

    // ---------------- previous simulation: priors, new children, backup deltas (registers; stores issued, nothing re-read)
    const bool expanded = flag == 1 && cnt_new > 0 && first_new + cnt_new <= e.node_cap;
    const int leaf_old = flag == 1 ? (depth_old < 64 ? __builtin_amdgcn_readlane(pnode, depth_old & 63) : path[depth_old]) : -1;
    float pl[3] = {0.f, 0.f, 0.f};
    if (flag == 1) {
        if (e.prior_mode == 0 && !hit_old) {     // P0: gather at the legal actions, divide by the sum unless 0 (pv_network_cnn.py:129-132)
#pragma unroll
            for (int r = 0; r < 4; ++r) polbuf[lane + 64 * r] = polr[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = lane + 64 * r;
                pl[r] = (i < cnt_new) ? polbuf[oa[r]] : 0.f;
                sum += pl[r];
            }
            sum = wave_sum_f(sum);
            const float den = (sum != 0.f) ? sum : 1.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) pl[r] = pl[r] / den;
        } else {                     // fake / external evaluator, or a leaf served from the evaluation cache: legal-ordered normalised priors
#pragma unroll
            for (int r = 0; r < 3; ++r) { const int i = lane + 64 * r; pl[r] = (i < cnt_new) ? polr[r] : 0.f; }
        }
        if (cache_on && !hit_old) {
            // this evaluation goes into the entry the select step reserved: the row first (priors + actions, defined over all
            // MAX_LEGAL places), then the key record that makes it findable.  Only this wave ever touches this slot's table.
            cslot = __builtin_amdgcn_readfirstlane(cslot);
            if (cslot >= 0) {
                const size_t ent = ((size_t)g << e.eval_cache_log2) + (size_t)cslot;
                unsigned char* row = reinterpret_cast<unsigned char*>(e.eval_cache_rows) + ent * EVAL_CACHE_ROW;
                float* rp = reinterpret_cast<float*>(row);
                uint8_t* ro = row + MAX_LEGAL * sizeof(float);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int i = lane + 64 * r;
                    if (i < MAX_LEGAL) { rp[i] = pl[r]; ro[i] = (i < cnt_new) ? oa[r] : (uint8_t)0xFF; }
                }
                if (lane == 0) {
                    const QState k = uniform_state(leaf_prev);
                    u32x4* kr = reinterpret_cast<u32x4*>(e.eval_cache_keys) + 2 * ent;
                    kr[0] = (u32x4){(uint32_t)k.hw, (uint32_t)(k.hw >> 32), (uint32_t)k.vw, (uint32_t)(k.vw >> 32)};
                    kr[1] = (u32x4){eval_cache_misc(k), 2u, (uint32_t)cnt_new, __builtin_bit_cast(uint32_t, value)};
                }
            }
        }
        if (expanded) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = lane + 64 * r;
                if (i < cnt_new) {
                    NodeRec c;
                    c.w = 0.0; c.p = pl[r]; c.n = 0; c.kids = 0; c.action = oa[r]; c.q = 0.f; c.cp = e.c_puct * pl[r];
                    nodes[first_new + i] = c;
                }
            }
            if (lane == 0) {
                nodes[leaf_old].kids = (uint32_t)first_new | ((uint32_t)cnt_new << 24);
                e.node_count[g] = first_new + cnt_new;
            }
        }
        if (lane == 0) e.stat_leaf_evals[g] += 1;
    }
    const uint32_t kids_new = expanded ? ((uint32_t)first_new | ((uint32_t)cnt_new << 24)) : 0u;
    const double v_old = (double)value;                              // value.item() -> python float
    // lane d <= depth_old holds the old path node at depth d: its record after the backup (pv_mcts.py:60-66), store pending
    double bw = 0.0; int bn = 0;
    float bq = 0.f;                   // ... and its exploitation term after the backup: ONE float64 division per step, off the
                                      // descent's per-level chain (the levels below the root read it by v_readlane)
    const bool fast_old = flag == 1 && depth_old <= fast_depth && depth_old < 63;
    if (flag == 1 && fast_old) {
        if (lane <= depth_old) {
            const NodeRec& r = nodes[pnode];
            bw = r.w + (((depth_old - lane) & 1) ? -v_old : v_old);
            bn = r.n + 1;
            bq = q_of(bw, bn);
        }
    }
    bool pending = fast_old;          // the old path's updated (w, n) are in registers, not in memory
    auto flush_old = [&]() {
        if (pending && lane <= depth_old) { NodeRec& r = nodes[pnode]; r.w = bw; r.n = bn; r.q = bq; }
        pending = false;
    };
    if (flag == 1 && !fast_old) {     // deep old path: plain read-modify-write through memory, then everything below reads memory
        if (lane <= depth_old && lane < 64) {
            NodeRec& r = nodes[pnode];
            r.w += ((depth_old - lane) & 1) ? -v_old : v_old;
            r.n += 1;
            r.q = q_of(r.w, r.n);
        }
        for (int d = lane + 64; d <= depth_old; d += 64) {
            NodeRec& r = nodes[path[d]];
            r.w += ((depth_old - d) & 1) ? -v_old : v_old;
            r.n += 1;
            r.q = q_of(r.w, r.n);
        }
    }
    if (!do_select) { flush_old(); return; }
    if (!active) { flush_old(); if (lane == 0) { e.leaf_flag[g] = 0; if (cache_on) e.eval_mask[g] = 0; } return; }
    if (lane == 0) { e.leaf_flag[g] = 0; if (cache_on) e.eval_mask[g] = 0; }
    STEP_STAMP(1)

    // ---------------- descent (pv_mcts.py:33-66 via :69-78)
    // A tree level is one dependent chain -- children arrive -> scores -> arg-max -> the chosen child's range -> next fetch -- and the
    // step kernel is one wave per SIMD, so everything that does NOT depend on the children is moved off that chain:
    //   * t = sum of the children's visit counts (pv_mcts.py:71) is the parent's own n minus one -- a node is visited once when it is
    //     expanded and once more for every descent into a child (pv_mcts.py:49-50, :62-64) -- so sqrt(t) is formed from the parent's
    //     record while the children's loads are in flight (no wave reduction, no square root behind the loads);
    //   * C_PUCT * p comes with the record (NodeRec::cp);
    //   * the pending backup patches the one child that lies on the old path with n + 1 and the q its own lane already holds; its w
    //     is never needed here: if the new path stays on the old one, lane d already owns that node's updated (w, n) -- bw, bn;
    //   * the next level's children are requested as soon as the chosen child's range is known; next() of the game state, the path
    //     bookkeeping and the chosen child's statistics follow behind the loads;
    //   * validity is a scalar mask, slots beyond the node's child count are skipped by scalar branches (no exec-mask regions), the
    //     wave maximum is six v_max_f32 with DPP operands.
    QState s = uniform_state(s_loaded);
    bool regs = true;                 // round-1 / register copies are current (false after a fall-back to memory)
    if (flag == 1 && !fast_old) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        regs = false;
    }
    int node = 0, depth = 0;
    int mynode = 0;                   // lane d: new path node at depth d ...
    double nw = 0.0; int nn = 0;      // ... and its current (w, n), pending updates included
    const bool pend0 = flag == 1 && fast_old;                        // (wave-uniform) the old path's backup is pending in registers
    if (lane == 0 && regs) { nw = pend0 ? bw : rootrec.w; nn = pend0 ? bn : rootrec.n; }
    bool onpath = pend0;              // the current node IS the old path's node at this depth
    int terminal = 0;
    double tvalue = 0.0;
    uint32_t kids = regs ? ((onpath && depth_old == 0) ? kids_new : rootrec.kids) : nodes[0].kids;
    kids = (uint32_t)__builtin_amdgcn_readfirstlane((int)kids);
    // n of the current node with the pending backup applied (lane 0 holds the root's)
    int npar = __builtin_amdgcn_readfirstlane(regs ? (pend0 ? bn : rootrec.n) : nodes[0].n);
    // One level's selection (pv_mcts.py:69-78) from the children's records `hot` / `cold`; results in the scalars below.  The records
    // are never modified in registers: the pending backup's patch goes into temporaries.
    uint32_t kids_n = 0u; int action = 0, cn = 0, besti = 0; double cw = 0.0;
    auto select_level = [&](const u32x4 (&hot)[3], const u32x4 (&cold)[3]) {
        const int cnt = (int)(kids >> 24), first = (int)(kids & 0xFFFFFF);
        // the old path's child of this node: its index among these children, and its exploitation term after the pending backup --
        // lane depth + 1 computed it from that node's own record (one division per step, started before the descent); at the root
        // it is formed below from the round-1 copy, so that level 0 does not wait for the second load round
        const bool patch = regs && onpath && depth < depth_old;
        const int pidx = patch ? __builtin_amdgcn_readlane(pnode, (depth + 1) & 63) - first : -1;
        const float pq = (patch && depth > 0) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bq), (depth + 1) & 63)) : 0.f;
        const float st = sqrtf((float)(npar - 1));                // == f32(math.sqrt(t)), see game_select; t = npar - 1
        // At the root the patched child's q cannot come from lane 1 (that lane's record is the second load round): it is formed from
        // the round-1 copy of the child itself -- one float64 division per step, under a scalar branch, in front of the scores
        float q0fix = 0.f;
        if (patch && depth == 0) {
            const int ps = pidx >> 6;
            const u32x4 cc = ps == 0 ? cold[0] : (ps == 1 ? cold[1] : cold[2]);
            const u32x4 hh = ps == 0 ? hot[0] : (ps == 1 ? hot[1] : hot[2]);
            const uint32_t w0 = cc[0], w1 = cc[1], nb = hh[0];
            const double wr = __builtin_bit_cast(double, ((uint64_t)w1 << 32) | w0);
            q0fix = q_of(wr + (((depth_old - 1) & 1) ? -v_old : v_old), (int)nb + 1);   // (only the lane of the patched child uses it)
        }
        const float pqv = depth == 0 ? q0fix : pq;
        const bool leafnext = patch && depth + 1 == depth_old;     // the patched child is the old leaf: it has children now
        float sc[3];
        int neff[3];
        uint32_t keff[3];
        // straight-line scores: the slots' chains (patch by selects, int -> float, multiply, IEEE division, add) are independent, so
        // that the in-order issue of a lone wave interleaves them; nodes with at most 64 children (every node once the walls are
        // placed) take the one-slot copy of the same code
        auto score = [&](int r) {
            // (elements go through scalars: __builtin_bit_cast of a vector ELEMENT expression reads element 0 with hipcc 7.2)
            const uint32_t nb = hot[r][0], kb = hot[r][1], qb = hot[r][2], cb = hot[r][3];
            const bool me = patch && (lane + 64 * r == pidx);
            neff[r] = (int)nb + (me ? 1 : 0);
            keff[r] = (me && leafnext) ? kids_new : kb;
            const float q = me ? pqv : __builtin_bit_cast(float, qb);          // q = f32(-w / n) travels with the record (NodeRec::q)
            const float u = (__builtin_bit_cast(float, cb) * st) / (float)(1 + neff[r]);
            sc[r] = (lane + 64 * r < cnt) ? q + u : -INFINITY;
        };
        if (cnt <= 64) {
            score(0);
            sc[1] = sc[2] = -INFINITY; neff[1] = neff[2] = 0; keff[1] = keff[2] = 0u;
        } else {
            score(0); score(1); score(2);
        }
        // np.argmax: the first index of the maximum.  Wave maximum by DPP, then the lowest child index holding it from up to three
        // ballots (children lane, lane + 64, lane + 128 in that order), masked to the node's children.  NaN scores never equal the
        // maximum; if nothing matches (all NaN) child 0 is taken, as before.
        const float best = wave_max_dpp_asm(fmaxf(fmaxf(sc[0], sc[1]), sc[2]));
        const uint64_t v0 = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
        const uint64_t m0 = __ballot(sc[0] == best) & v0;
        int bi = 0;
        if (m0) bi = __builtin_ctzll(m0);
        else if (cnt > 64) {
            const uint64_t v1 = cnt >= 128 ? ~0ull : ((1ull << (cnt - 64)) - 1ull);
            const uint64_t m1 = __ballot(sc[1] == best) & v1;
            if (m1) bi = 64 + __builtin_ctzll(m1);
            else if (cnt > 128) {
                const uint64_t m2 = __ballot(sc[2] == best) & ((1ull << (cnt - 128)) - 1ull);
                if (m2) bi = 128 + __builtin_ctzll(m2);
            }
        }
        besti = __builtin_amdgcn_readfirstlane(bi);
        const int slot = besti >> 6, src = besti & 63;          // wave-uniform: the winner's fields come by v_readlane
        uint32_t wlo, whi;
        auto pick = [&](const u32x4 c, uint32_t k, int n) {
            const uint32_t c0 = c[0], c1 = c[1], c3 = c[3];
            kids_n = (uint32_t)__builtin_amdgcn_readlane((int)k, src);
            action = __builtin_amdgcn_readlane((int)c3, src);
            cn = __builtin_amdgcn_readlane(n, src);
            wlo = (uint32_t)__builtin_amdgcn_readlane((int)c0, src);
            whi = (uint32_t)__builtin_amdgcn_readlane((int)c1, src);
        };
        if (slot == 0) pick(cold[0], keff[0], neff[0]); else if (slot == 1) pick(cold[1], keff[1], neff[1]); else pick(cold[2], keff[2], neff[2]);
        cw = __builtin_bit_cast(double, ((uint64_t)whi << 32) | wlo);   // (used only if the path ends on a terminal node off the old path)
        node = first + besti;
        onpath = patch && besti == pidx;                        // the new path follows the old one a level further
        // (every element of the six vectors stays allocated up to here: the record's p is never read, and the allocator handed the
        //  register of that dead element of an IN-FLIGHT load to the next temporary -- a write-after-write hazard it then covered with an
        //  s_waitcnt vmcnt(0) right behind the request)
#pragma unroll
        for (int r = 0; r < 3; ++r) asm volatile("" :: "v"(hot[r]), "v"(cold[r]));
    };
    // what stops the descent at the current node: 1 terminal, 2 unexpanded leaf, 3 the old leaf (expanded a moment ago: its children are
    // the records built above -- handled behind the loop, no record is needed there), 0 go on
    auto stop_here = [&]() -> int {
        const bool lose = is_lose<N>(s), draw = is_draw(s, e.plies_for_draw);
        if (lose || draw) { tvalue = lose ? -1.0 : 0.0; terminal = 1; return 1; }   // pv_mcts.py:35-42
        if ((kids >> 24) == 0) return 2;                                            // pv_mcts.py:45 unexpanded leaf
        if (regs && onpath && depth == depth_old) return 3;
        return 0;
    };
    // Children travel as the record's two aligned 16-byte halves and are requested for all three slots whatever the child count (lanes /
    // slots beyond it read the last child, or node 0 for an unexpanded child: same cache lines, no divergent region around the loads).
    // They are loaded and consumed inside ONE loop iteration -- a loop-carried record cost a copy of every register behind an
    // s_waitcnt at the back edge -- and what the previous level's choice still owes (next() of the game state, the path, the chosen
    // child's statistics for its lane) is done between the request and the first use: behind the loads, off the level's chain.
    bool at_old_leaf = false;
#ifdef AQG_STAMP_LEVELS
    unsigned long long lv_prev = __builtin_readcyclecounter();
#endif
    int stop = stop_here();
    if (stop == 0) {
        if (!regs) {                      // (deep old path, written through memory above: the round-1 copies are stale)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = (int)(kids & 0xFFFFFF) + max(min(lane + 64 * r, (int)(kids >> 24) - 1), 0);
                cold[r] = nhalf[2 * i]; hot[r] = nhalf[2 * i + 1];
            }
        }
        select_level(hot, cold);          // level 0: the root's children came with round 1
#ifdef AQG_STAMP_LEVELS
        lv_prev = __builtin_readcyclecounter();
#endif
        for (;;) {
            // next level's children (hand-over to memory first: flush what is pending, fence, go on reading memory)
            if (regs && depth + 1 >= fast_depth) {
                flush_old();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                regs = false;
                onpath = false;
            }
            // (Requesting the old path's next child block speculatively, before the scores are computed, was tried: the level
            //  got 14 % SLOWER -- a wrong guess costs a second round.)
            u32x4 h[3], c[3];
#ifdef AQG_STAMP_LEVELS
            LEVEL_STAMP(11)                                  // child chosen -> next request (hand-over test, addresses)
#endif
            {
                const int cnt = (int)(kids_n >> 24), first = (int)(kids_n & 0xFFFFFF);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int i = first + max(min(lane + 64 * r, cnt - 1), 0);
                    c[r] = nhalf[2 * i]; h[r] = nhalf[2 * i + 1];
                }
            }
            // ... and behind the loads: the chosen child becomes the current node
            ++depth;
            kids = kids_n;
            npar = cn;
            s = next_state<N>(s, action);
            // (the path stays in registers -- lane d owns depth d -- and is written once behind the descent: a store per level sat in
            //  the same in-order counter as the next level's loads.  Depths beyond 63, never seen, go through memory at once.)
            if (depth >= 64 && lane == 0) path[depth] = node;
            if (lane == (depth & 63) && depth < 64) { mynode = node; nw = onpath ? bw : cw; nn = onpath ? bn : cn; }
#ifdef AQG_STAMP
            if (lane == 0) reinterpret_cast<unsigned long long*>(e.pooled + (size_t)g * 128)[6] += 1;     // levels descended
#endif
            stop = stop_here();
            if (stop) break;
#ifdef AQG_STAMP_LEVELS
            LEVEL_STAMP(8)                                   // request -> state advanced, stop test done (work behind the loads)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LEVEL_STAMP(9)                                   // ... -> children arrived (what is left of the load latency)
#endif
            select_level(h, c);
#ifdef AQG_STAMP_LEVELS
            LEVEL_STAMP(10)                                  // ... -> child chosen (scores, arg-max, the winner's fields)
#endif
        }
    }
    at_old_leaf = stop == 3;
    if (at_old_leaf) {
        // The descent has followed the old path down to the old leaf, whose children are the records built above (n = 0, q = 0).  Their
        // visit counts sum to t = 0, so every score is 0 + (cp * 0) / 1 = 0 (or NaN for a NaN prior: never the maximum) and np.argmax
        // takes the FIRST child (pv_mcts.py:72-78) -- no record is needed to know that, and the child is a fresh leaf
        // (or a terminal position): the descent ends one level below.
        node = (int)(kids & 0xFFFFFF);
        s = next_state<N>(s, __builtin_amdgcn_readlane((int)oa[0], 0));
        ++depth;
        if (depth >= 64 && lane == 0) path[depth] = node;
        if (lane == (depth & 63) && depth < 64) { mynode = node; nw = 0.0; nn = 0; }
        const bool lose = is_lose<N>(s), draw = is_draw(s, e.plies_for_draw);
        if (lose || draw) { tvalue = lose ? -1.0 : 0.0; terminal = 1; }
#ifdef AQG_STAMP
        if (lane == 0) reinterpret_cast<unsigned long long*>(e.pooled + (size_t)g * 128)[6] += 1;
#endif
    }
    if (lane <= min(depth, 63)) path[lane] = mynode;             // the new path, depths 0..63 (lane 0: the root, node 0)
#ifdef AQG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STEP_STAMP(2)
    if (terminal) {
        // backup of THIS simulation (pv_mcts.py:36-42).  Pending old-path stores go first; the new path's stores carry both
        // updates for the nodes the two paths share (same wavefront, same address: stores keep their order).
        if (regs) {
            flush_old();
            if (lane <= depth && lane < 64) {
                NodeRec& r = nodes[mynode];
                r.w = nw + (((depth - lane) & 1) ? -tvalue : tvalue);
                r.n = nn + 1;
                r.q = q_of(r.w, r.n);
            }
        } else {
            if (lane <= depth && lane < 64) {
                NodeRec& r = nodes[mynode];
                r.w += ((depth - lane) & 1) ? -tvalue : tvalue;
                r.n += 1;
                r.q = q_of(r.w, r.n);
            }
        }
        if (lane == 0) {
            e.stat_terminal_sims[g] += 1;
            for (int d = 64; d <= depth; ++d) {
                NodeRec& r = nodes[path[d]];
                r.w += ((depth - d) & 1) ? -tvalue : tvalue;
                r.n += 1;
                r.q = q_of(r.w, r.n);
            }
        }
    } else {
        flush_old();
        // Evaluation cache: has this slot asked the network for this position before?  One probe round -- lane i compares the key
        // record of table entry (home + i) -- decides; a hit copies the entry's priors, actions, count and value to where the
        // evaluator and wave_legal_actions would have put them, and the leaf is sent neither through the legal-move search nor
        // through the network (eval_mask 0).  A miss reserves the first empty entry of the window (or replaces one) for the
        // evaluation that the next step's expansion will see.
        bool hit = false;
        int newslot = -1;
        LegalPrep prep;
        if (!cache_on) prep = wave_legal_prepare<N>(s, lane);
        if (cache_on) {
            const uint32_t misc = eval_cache_misc(s);
            uint64_t h = s.hw * 0x9E3779B97F4A7C15ull ^ s.vw * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)misc * 0x165667B19E3779F9ull;
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            const uint32_t cmask = (1u << e.eval_cache_log2) - 1u, home = (uint32_t)h & cmask;
            const size_t base = (size_t)g << e.eval_cache_log2;
            const u32x4* keys = reinterpret_cast<const u32x4*>(e.eval_cache_keys) + 2 * base;
            const uint32_t idx = (home + (uint32_t)lane) & cmask;
            const u32x4 k0 = keys[2 * idx], k1 = keys[2 * idx + 1];
            // ... and while the probe is in flight: the part of legal_actions() that needs no memory (placement masks, touch-count
            // prefilter: scalar mask algebra) -- a miss has it ready, a hit has lost nothing
            prep = wave_legal_prepare<N>(s, lane);
            // (elements through scalars: __builtin_bit_cast / readlane of a vector ELEMENT expression reads element 0 with hipcc 7.2)
            const uint32_t a0 = k0[0], a1 = k0[1], a2 = k0[2], a3 = k0[3], b0 = k1[0], b1 = k1[1], b2 = k1[2], b3 = k1[3];
            const bool match = a0 == (uint32_t)s.hw && a1 == (uint32_t)(s.hw >> 32) && a2 == (uint32_t)s.vw && a3 == (uint32_t)(s.vw >> 32) &&
                               b0 == misc && b1 == 2u;
            const uint64_t mb = __ballot(match);
            if (mb) {
                hit = true;
                const int src = __builtin_ctzll(mb);
                const uint32_t hs = (home + (uint32_t)src) & cmask;
                const int cnt = __builtin_amdgcn_readlane((int)b2, src);
                const uint32_t vbits = (uint32_t)__builtin_amdgcn_readlane((int)b3, src);
                const unsigned char* row = reinterpret_cast<const unsigned char*>(e.eval_cache_rows) + (base + hs) * EVAL_CACHE_ROW;
                const float* rp = reinterpret_cast<const float*>(row);
                float* pdst = e.policy + (size_t)g * A;
#pragma unroll
                for (int r = 0; r < 3; ++r) { const int i = lane + 64 * r; if (i < cnt) pdst[i] = rp[i]; }
                if (lane < MAX_LEGAL / 4)
                    reinterpret_cast<uint32_t*>(e.legal_order + (size_t)g * MAX_LEGAL)[lane] = reinterpret_cast<const uint32_t*>(row + MAX_LEGAL * sizeof(float))[lane];
                STEP_STAMP(3)
                if (lane == 0) {
                    store_state(e.leaf_state, g, s);
                    e.legal_count[g] = cnt;
                    e.path_len[g] = depth;
                    e.value[g] = __builtin_bit_cast(float, vbits);
                    e.leaf_flag[g] = 2;
                    e.stat_cache_hits[g] += 1;
                }
            } else {
                const uint64_t eb = __ballot(b1 == 0u);
                newslot = (int)((home + (uint32_t)(eb ? __builtin_ctzll(eb) : (int)((h >> 40) & 63u))) & cmask);
            }
        }
        if (!hit) {
            const int total = wave_legal_finish<N>(s, prep, lane, nullptr, e.legal_order + (size_t)g * MAX_LEGAL);
            STEP_STAMP(3)
            if (lane == 0) {
                store_state(e.leaf_state, g, s);
                e.legal_count[g] = total;
                e.path_len[g] = depth;
                e.leaf_flag[g] = 1;
                if (cache_on) {
                    e.eval_mask[g] = 1; e.eval_cache_slot[g] = newslot;
                    // large sets: the leaves the network must evaluate, as a compact list for the trunk launch of this simulation (the
                    // order of the entries is whatever order the waves arrive in -- every board's evaluation is independent of it)
                    if (list_sim >= 0) e.eval_list[atomicAdd(e.eval_count + list_sim, 1)] = g;
                }
            }
        }
    }
#ifdef AQG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STEP_STAMP(4)
    if (lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(e.pooled + (size_t)g * 128);
        o[7] += 1;                                                                                    // steps
        // the tail: this game's LONGEST step (a launch lasts as long as the slowest game of its set) with its phases and depth, and a
        // histogram of step lengths in 2,048-cycle buckets
        const unsigned long long tot = sp_loc[0] + sp_loc[1] + sp_loc[2] + sp_loc[3] + sp_loc[4];
        if (tot > o[16]) { o[16] = tot; for (int i = 0; i < 5; ++i) o[17 + i] = sp_loc[i]; o[22] = (unsigned long long)depth; o[23] = (unsigned long long)terminal; }
        const unsigned long long bk = tot >> 11;
        o[24 + (bk < 39 ? bk : 39)] += 1;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// fused simulation step, one wavefront per game:
//   [expand + backup of the PREVIOUS simulation's leaf]  ->  [select the next leaf + its legal actions]
// Both halves touch only this game's pools, and the wave that wrote the children is the wave that reads them, so
// a workgroup-scope fence is all the ordering needed.  Per simulation the engine then launches
// step -> GNN trunk -> GNN heads (3 kernels instead of select / legal / trunk / heads / expand).
// ------------------------------------------------------------------------------------------------
template <int N, bool CACHE>
__global__ __launch_bounds__(512) void engine_step_fast_kernel(aqg_engine e, int do_expand, int do_select, int fast_depth, int list_sim) {
    __shared__ float polbuf[8][256];
    AQG_TRACE_BEGIN
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // 1, 2, 4 or 8 games per workgroup (option "step_waves")
    // a game's wave is a latency-bound chain that issues little: at priority 1 it wins the arbitration against a co-resident trunk
    // workgroup's vector work, finishes sooner and gives its CU's second trunk slot back sooner (option "step_prio")
    { const int pr = (fast_depth >> 8) & 3; if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 3) __builtin_amdgcn_s_setprio(3); }
    fast_depth &= 0xFF;
    if (g < e.num_games) game_step_fast<N, CACHE>(e, g, lane, do_expand, do_select, fast_depth, polbuf[threadIdx.x >> 6], list_sim);
    AQG_TRACE_END(1, (unsigned long long)(uintptr_t)e.pooled)
}
AQG_TRACE_SETTER(set_trace_mcts)

template <int N>
__global__ __launch_bounds__(256) void engine_step_kernel(aqg_engine e, int do_expand, int do_select) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= e.num_games) return;
    // loads that depend on nothing but g: one round for both halves of the step
    const int active = e.game_active[g];
    const QState root = load_state(e.root_state, 1, g);
    int flag = 0, depth = 0, cnt = 0, first = 0;
    float value = 0.f;
    if (do_expand) { flag = e.leaf_flag[g]; depth = e.path_len[g]; cnt = e.legal_count[g]; first = e.node_count[g]; value = e.value[g]; }
    if (do_expand) game_expand_backup<N>(e, g, lane, flag, depth, cnt, first, value);
    if (do_select) {
        if (do_expand) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        game_select<N>(e, g, lane, active, root);
    }
}

template <int N>
static void launch_step(const aqg_engine& e, int do_expand, int do_select, hipStream_t st, int list_sim = -1) {
    const dim3 grid((e.num_games + 3) / 4), block(256);
    if (g_profile_trunk == 2) profile_mark(st, e.num_games);       // measurement mode 2: the event pairs bracket the step launches
    if (g_step_variant == 1) {
        const int wpb = (g_step_waves == 1 || g_step_waves == 2 || g_step_waves == 8) ? g_step_waves : 4;
        const int fd = g_step_fast_depth | ((g_step_prio & 3) << 8);
        const dim3 sg((e.num_games + wpb - 1) / wpb), sb(64 * wpb);
        if (e.eval_cache_keys && e.prior_mode == 0) hipLaunchKernelGGL((engine_step_fast_kernel<N, true>), sg, sb, 0, st, e, do_expand, do_select, fd, list_sim);
        else hipLaunchKernelGGL((engine_step_fast_kernel<N, false>), sg, sb, 0, st, e, do_expand, do_select, fd, -1);
    }
    else hipLaunchKernelGGL(engine_step_kernel<N>, grid, block, 0, st, e, do_expand, do_select);
    if (g_profile_trunk == 2) profile_mark(st, -1);
}

// ------------------------------------------------------------------------------------------------
// finish move: visits -> policy (pv_mcts.py:88-95), record, np.random.choice, next(), terminal handling
// ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void engine_finish_move_kernel(aqg_engine e, const double* __restrict__ uniforms) {
    constexpr int A = Geo<N>::A;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= e.num_games || !e.game_active[g]) return;
    const NodeRec* __restrict__ nodes = game_nodes(e, g);
    const uint32_t kids = nodes[0].kids;
    const int cnt = (int)(kids >> 24), first = (int)(kids & 0xFFFFFF);
    const int k = e.slot_game[g];                  // history, plies and result are kept per GAME: a slot plays several
    const int ply = e.game_plies[k];
    QState s = load_state(e.root_state, 1, g);

    // history row: state72 + dense visit counts
    if (ply < e.max_plies) {
        uint8_t* hs = e.hist_state72 + ((size_t)k * e.max_plies + ply) * STATE72;
        if (lane == 0) pack72(s, N, hs);
        uint16_t* hv = e.hist_visits + ((size_t)k * e.max_plies + ply) * A;
        for (int i = lane; i < cnt; i += 64) hv[nodes[first + i].action] = (uint16_t)nodes[first + i].n;
    }
    if (lane != 0) return;

    int chosen = -1;
    if (cnt > 0) {
        int idx = 0;
        if (e.temperature == 0.f) {                            // one-hot at the first maximum, then choice(p=one-hot)
            int bestn = -1;
            for (int i = 0; i < cnt; ++i) { const int n = nodes[first + i].n; if (n > bestn) { bestn = n; idx = i; } }
        } else {
            // boltzman (pv_mcts.py:106-109): xs = n ** (1/T); p = x / sum(xs).  T == 1 is exact (n ** 1.0 == float(n)).
            const double invT = 1.0 / (double)e.temperature;
            double tot = 0.0;
            for (int i = 0; i < cnt; ++i) {
                const double x = (double)nodes[first + i].n;
                tot += (e.temperature == 1.f) ? x : pow(x, invT);
            }
            // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, side='right')
            double last = 0.0;
            for (int i = 0; i < cnt; ++i) {
                const double x = (double)nodes[first + i].n;
                last += ((e.temperature == 1.f) ? x : pow(x, invT)) / tot;
            }
            const double u = uniforms[g];
            double acc = 0.0;
            idx = 0;
            for (int i = 0; i < cnt; ++i) {
                const double x = (double)nodes[first + i].n;
                acc += ((e.temperature == 1.f) ? x : pow(x, invT)) / tot;
                if (acc / last <= u) idx = i + 1;
            }
            if (idx >= cnt) idx = cnt - 1;
        }
        chosen = (int)nodes[first + idx].action;
    }
    if (chosen < 0) {
        // Dead end: legal_actions() is empty.  The reference would re-predict forever-leaf and np.random.choice([])
        // raises; we abort the game as a draw and count it.
        e.game_active[g] = 0; e.game_result[k] = 0; e.game_done[k] = 1;
        atomicAdd(&e.counters[2], 1); atomicAdd(&e.counters[1], 1); atomicSub(&e.counters[0], 1);
        return;
    }
    if (ply < e.max_plies) e.hist_action[(size_t)k * e.max_plies + ply] = (uint8_t)chosen;
    const QState t = next_state<N>(s, chosen);
    store_state(e.root_state, g, t);
    e.game_plies[k] = ply + 1;
    const bool lose = is_lose<N>(t), draw = is_draw(t, e.plies_for_draw);
    if (lose || draw) {
        // first_player_value (self_play.py:22-27): ended state's mover lost; z of ply 0, alternating afterwards (:63-66)
        int z = 0;
        if (lose) z = ((t.plies % 2) == 0) ? -1 : 1;
        e.game_result[k] = (int8_t)z;
        e.game_done[k] = 1;
        e.game_active[g] = 0;                  // engine_refill_kernel may hand the slot its next game
        atomicAdd(&e.counters[1], 1); atomicSub(&e.counters[0], 1);
    }
}

// ------------------------------------------------------------------------------------------------
// slot refill: a rank plays a QUOTA of games on its G slots (the reference's plain loop over games, self_play.py:81-84).
// After every move the idle slots -- in slot order, so that the assignment is deterministic -- take the next game
// indices not yet handed out and start from the initial position; once the quota is exhausted a finished slot stays
// idle.  One workgroup: a block-wide exclusive scan over the slots' "idle" flags.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void engine_refill_kernel(aqg_engine e) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base = e.counters[3];
    __syncthreads();
    const int move_index = e.counters[4] + 1;               // counters[4] = moves finished before this one; new games join the next
    __syncthreads();
    for (int g0 = 0; g0 < e.num_games; g0 += 1024) {
        const int g = g0 + tid;
        const int idle = (g < e.num_games && !e.game_active[g] && e.slot_game[g] >= 0) ? 1 : 0;
        int x = idle;                                        // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int before = 0, total = 0;
        for (int i = 0; i < 16; ++i) { if (i < w) before += wsum[i]; total += wsum[i]; }
        const int k = base + before + x - idle;              // this slot's next game, if any is left
        if (idle) {
            if (k < e.quota) {
                const int N = e.board_size;
                QState s;
                s.hw = 0; s.vw = 0;
                s.ppos = (uint8_t)(N * (N - 1) + N / 2); s.pwl = (uint8_t)e.num_walls;
                s.epos = s.ppos; s.ewl = s.pwl;
                s.plies = 0; s.pad = 0;
                store_state(e.root_state, g, s);
                e.slot_game[g] = k;
                e.game_slot[k] = g;
                e.game_first_move[k] = move_index;
                e.game_active[g] = 1;
                e.leaf_flag[g] = 0;
            } else {
                e.slot_game[g] = -1;                         // retired
            }
        }
        __syncthreads();
        if (tid == 0) {
            const int handed = min(total, max(e.quota - base, 0));
            base += total;
            if (handed) atomicAdd(&e.counters[0], handed);
        }
        __syncthreads();
    }
    if (tid == 0) { e.counters[3] = min(base, e.quota); e.counters[4] = move_index; }
}

template <int N>
__global__ __launch_bounds__(256) void engine_root_visits_kernel(aqg_engine e, int32_t* __restrict__ visits,
                                                                 uint8_t* __restrict__ actions, int32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= e.num_games) return;
    const NodeRec* __restrict__ nodes = game_nodes(e, g);
    const uint32_t kids = nodes[0].kids;
    const int cnt = (int)(kids >> 24), first = (int)(kids & 0xFFFFFF);
    for (int i = lane; i < MAX_LEGAL; i += 64) {
        visits[(size_t)g * MAX_LEGAL + i] = (i < cnt) ? nodes[first + i].n : 0;
        actions[(size_t)g * MAX_LEGAL + i] = (i < cnt) ? (uint8_t)nodes[first + i].action : 0xFF;
    }
    if (lane == 0) count[g] = cnt;
}

// ------------------------------------------------------------------------------------------------
// host-side enqueue (no sync, no allocation)
// ------------------------------------------------------------------------------------------------
static int validate(const aqg_engine& e) {
    const int N = e.board_size;
    if (!(N == 3 || N == 5 || N == 7 || N == 9)) return fail("unsupported board_size");
    if (e.num_games <= 0 || e.sims <= 0) return fail("num_games and sims must be positive");
    if ((long long)e.node_cap >= (1 << 24)) return fail("node_cap must be < 2^24");
    if (e.node_cap < 1 + MAX_LEGAL) return fail("node_cap too small");
    if (e.prior_mode == 0 && N != 9 && !e.gnn_workspace) return fail("boards other than 9x9 need gnn_workspace for the GNN evaluator");
    if (e.prior_mode < 0 || e.prior_mode > 2) return fail("prior_mode must be 0, 1 or 2");
    if (e.quota < e.num_games) return fail("quota must be >= num_games");
    if (!e.slot_game || !e.game_done || !e.game_slot || !e.game_first_move) return fail("slot_game / game_done / game_slot / game_first_move are required");
    if (e.eval_cache_keys) {
        if (e.prior_mode != 0) return fail("the evaluation cache serves the GNN evaluator only (prior_mode 0)");
        if (!e.eval_cache_rows || !e.eval_cache_slot || !e.eval_mask || !e.stat_cache_hits) return fail("eval_cache_rows / eval_cache_slot / eval_mask / stat_cache_hits are required with eval_cache_keys");
        if ((e.eval_list == nullptr) != (e.eval_count == nullptr)) return fail("eval_list and eval_count come together");
        if (e.eval_cache_log2 < 6 || e.eval_cache_log2 > 20) return fail("eval_cache_log2 must be 6..20");
        if (g_step_variant != 1) return fail("the evaluation cache needs step_variant 1");
    }
    return 0;
}

template <int N>
static int enqueue_sims(const aqg_engine& e, hipStream_t st) {
    if (e.prior_mode == 2) return fail("prior_mode 2 (external evaluator): drive the move with aqg_engine_begin_move / _step / _finish_move");
    const dim3 grid((e.num_games + 3) / 4), block(256);
    hipLaunchKernelGGL(engine_begin_move_kernel, dim3((e.num_games + 255) / 256), dim3(256), 0, st, e);
    // evaluation cache on a set larger than the trunk's grid: the leaves that miss the cache go to the trunk as a compact list
    const bool use_list = e.eval_cache_keys && e.eval_list && N == 9 && e.num_games > 512 && g_trunk_variant >= 3 && g_step_variant == 1 && !(e.gnn_flags & AQG_GNN_EXACT_F32);
    for (int sim = 0; sim < e.sims; ++sim) {
        launch_step<N>(e, sim > 0 ? 1 : 0, 1, st, use_list ? sim : -1);
        if (e.prior_mode == 0) {
            // 9x9: the fused trunk; smaller boards: plain kernels over e.gnn_workspace
            if (int r = launch_gcn_forward_boards_any(N, e.leaf_state, 1, e.num_games, e.packed_weights, e.gnn_workspace,
                                                      e.gnn_workspace ? boards_any_workspace_floats(N, e.num_games) : 0, e.pooled, nullptr,
                                                      e.policy, nullptr, e.value, e.eval_cache_keys ? e.eval_mask : e.leaf_flag, e.gnn_flags, e.counters + 5, st,
                                                      use_list ? e.eval_list : nullptr, use_list ? e.eval_count + sim : nullptr))
                return r;
        } else {
            hipLaunchKernelGGL(engine_fake_eval_kernel<N>, grid, block, 0, st, e);
        }
    }
    launch_step<N>(e, 1, 0, st);   // expand + backup of the last simulation
    return check_launch("engine simulation kernels");
}

// One move's search is 3 * sims + 2 launches with constant arguments: on a capturable (non-default) stream it is
// captured once into a hipGraph and replayed per move, so the host cost per move is one graph launch instead of ~600
// kernel launches (with several game sets on several streams the host is otherwise the bottleneck).  The cache key is
// the engine struct itself plus the trunk options the launches read.
// (Host entry points are called from one host thread per process, like the reference's single-threaded loop; the cache
// below and the library's other host-side globals are not synchronised.)
struct SimGraph {
    aqg_engine e;
    int opts[8];
    hipGraphExec_t exec;
    hipEvent_t last;          // recorded behind every replay: eviction waits for THIS graph's last replay, not for the device
};
static std::vector<SimGraph> g_sim_graphs;
constexpr size_t SIM_GRAPH_CACHE = 64;   // engines x game sets that can alternate without re-capturing (4 sets per engine: 16 engines)

static int replay(SimGraph& g, hipStream_t st) {
    if (hipGraphLaunch(g.exec, st) != hipSuccess) return fail("hipGraphLaunch");
    if (hipEventRecord(g.last, st) != hipSuccess) return fail("hipEventRecord");
    return 0;
}

template <int N>
static int run_sims(const aqg_engine& e, hipStream_t st) {
    if (!g_use_graph || g_profile_trunk || st == nullptr || e.sims < 4) return enqueue_sims<N>(e, st);
    // every option a captured launch bakes in is part of the key: a changed option must never replay a stale graph
    const int opts[8] = {g_trunk_variant, g_trunk_grid, g_trunk_phase_delay, g_trunk_delay_min_boards, N, g_step_variant, g_step_fast_depth, ((g_trunk_prio & 0xff) << 8) | (g_step_waves << 16) | (g_step_prio << 24) | (g_heads_prio << 28)};
    for (SimGraph& g : g_sim_graphs)
        if (!memcmp(&g.e, &e, sizeof(aqg_engine)) && !memcmp(g.opts, opts, sizeof(opts))) return replay(g, st);
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return enqueue_sims<N>(e, st);                       // stream not capturable: plain launches
    }
    const int r = enqueue_sims<N>(e, st);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(st, &graph);
    if (r) { if (graph) (void)hipGraphDestroy(graph); return r; }
    if (ec != hipSuccess || !graph) return fail("hipStreamEndCapture");
    SimGraph g;
    memcpy(&g.e, &e, sizeof(aqg_engine));
    memcpy(g.opts, opts, sizeof(opts));
    const hipError_t ei = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) return fail("hipGraphInstantiate");
    if (hipEventCreateWithFlags(&g.last, hipEventDisableTiming) != hipSuccess) { (void)hipGraphExecDestroy(g.exec); return fail("hipEventCreate"); }
    if (g_sim_graphs.size() >= SIM_GRAPH_CACHE) {            // FIFO: engines come and go
        SimGraph& old = g_sim_graphs.front();
        (void)hipEventSynchronize(old.last);                 // its last replay may still be running on ANOTHER stream: wait for that
        (void)hipGraphExecDestroy(old.exec);                 // replay only -- the other game sets' streams keep running
        (void)hipEventDestroy(old.last);
        g_sim_graphs.erase(g_sim_graphs.begin());
    }
    g_sim_graphs.push_back(g);
    return replay(g_sim_graphs.back(), st);
}

template <int N>
static int do_move(const aqg_engine& e, const double* uniforms, hipStream_t st) {
    if (int r = run_sims<N>(e, st)) return r;
    hipLaunchKernelGGL(engine_finish_move_kernel<N>, dim3((e.num_games + 3) / 4), dim3(256), 0, st, e, uniforms);
    if (e.quota > e.num_games) hipLaunchKernelGGL(engine_refill_kernel, dim3(1), dim3(1024), 0, st, e);
    return check_launch("engine_finish_move_kernel");
}

// External-evaluator mode (prior_mode 2): the caller runs the simulation loop itself -- begin, then per simulation
// step(expand the previous leaf, select the next) -> its OWN evaluator fills policy[g][0 .. legal_count[g]) (a PMF over
// legal_actions() in order, the BaseNetwork.predict contract BaseNetwork.py:36-40) and value[g] for every game with
// leaf_flag[g] == 1 -> ... -> step(expand, no select) -> finish.  Same kernels, same per-game semantics; the library
// merely launches no evaluator between the steps.
int engine_begin_move(const aqg_engine& e, hipStream_t st) {
    if (int r = validate(e)) return r;
    hipLaunchKernelGGL(engine_begin_move_kernel, dim3((e.num_games + 255) / 256), dim3(256), 0, st, e);
    return check_launch("engine_begin_move_kernel");
}

int engine_step(const aqg_engine& e, int do_expand, int do_select, hipStream_t st) {
    if (int r = validate(e)) return r;
    switch (e.board_size) {
        case 3: launch_step<3>(e, do_expand, do_select, st); break;
        case 5: launch_step<5>(e, do_expand, do_select, st); break;
        case 7: launch_step<7>(e, do_expand, do_select, st); break;
        default: launch_step<9>(e, do_expand, do_select, st); break;
    }
    return check_launch("engine_step_kernel");
}

int engine_finish_move(const aqg_engine& e, const double* uniforms, hipStream_t st) {
    if (int r = validate(e)) return r;
    const dim3 grid((e.num_games + 3) / 4), block(256);
    switch (e.board_size) {
        case 3: hipLaunchKernelGGL(engine_finish_move_kernel<3>, grid, block, 0, st, e, uniforms); break;
        case 5: hipLaunchKernelGGL(engine_finish_move_kernel<5>, grid, block, 0, st, e, uniforms); break;
        case 7: hipLaunchKernelGGL(engine_finish_move_kernel<7>, grid, block, 0, st, e, uniforms); break;
        default: hipLaunchKernelGGL(engine_finish_move_kernel<9>, grid, block, 0, st, e, uniforms); break;
    }
    if (e.quota > e.num_games) hipLaunchKernelGGL(engine_refill_kernel, dim3(1), dim3(1024), 0, st, e);
    return check_launch("engine_finish_move_kernel");
}

int engine_clear_eval_cache(const aqg_engine& e, hipStream_t st) {
    if (int r = validate(e)) return r;
    if (!e.eval_cache_keys) return 0;
    if (hipMemsetAsync(e.eval_cache_keys, 0, ((size_t)e.num_games << e.eval_cache_log2) * 32, st) != hipSuccess) return fail("hipMemsetAsync(eval_cache_keys)");
    return 0;
}

int engine_reset(const aqg_engine& e, hipStream_t st) {
    if (int r = validate(e)) return r;
    if (int r = engine_clear_eval_cache(e, st)) return r;
    hipLaunchKernelGGL(engine_reset_kernel, dim3((max(e.num_games, e.quota) + 255) / 256), dim3(256), 0, st, e);
    return check_launch("engine_reset_kernel");
}

int engine_move(const aqg_engine& e, const double* uniforms, hipStream_t st) {
    if (int r = validate(e)) return r;
    switch (e.board_size) {
        case 3: return do_move<3>(e, uniforms, st);
        case 5: return do_move<5>(e, uniforms, st);
        case 7: return do_move<7>(e, uniforms, st);
        default: return do_move<9>(e, uniforms, st);
    }
}

int engine_set_roots(const aqg_engine& e, const uint8_t* roots72, hipStream_t st) {
    if (int r = validate(e)) return r;
    hipLaunchKernelGGL(engine_set_roots_kernel, dim3((e.num_games + 255) / 256), dim3(256), 0, st, e, roots72);
    return check_launch("engine_set_roots_kernel");
}

int engine_search(const aqg_engine& e, const uint8_t* roots72, hipStream_t st) {
    if (int r = validate(e)) return r;
    hipLaunchKernelGGL(engine_set_roots_kernel, dim3((e.num_games + 255) / 256), dim3(256), 0, st, e, roots72);
    switch (e.board_size) {
        case 3: return run_sims<3>(e, st);
        case 5: return run_sims<5>(e, st);
        case 7: return run_sims<7>(e, st);
        default: return run_sims<9>(e, st);
    }
}

int engine_root_visits(const aqg_engine& e, int32_t* visits, uint8_t* actions, int32_t* count, hipStream_t st) {
    const dim3 grid((e.num_games + 3) / 4), block(256);
    switch (e.board_size) {
        case 3: hipLaunchKernelGGL(engine_root_visits_kernel<3>, grid, block, 0, st, e, visits, actions, count); break;
        case 5: hipLaunchKernelGGL(engine_root_visits_kernel<5>, grid, block, 0, st, e, visits, actions, count); break;
        case 7: hipLaunchKernelGGL(engine_root_visits_kernel<7>, grid, block, 0, st, e, visits, actions, count); break;
        default: hipLaunchKernelGGL(engine_root_visits_kernel<9>, grid, block, 0, st, e, visits, actions, count); break;
    }
    return check_launch("engine_root_visits_kernel");
}

}  // namespace aqg
