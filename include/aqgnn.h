/*
 * include/aqgnn.h -- C ABI of libaqgnn_hip.so (MI355X / gfx950 only).
 *
 * The reference (ApproximateCaesar/AlphaQuoridorGNN) is pure Python and has no FFI layer; its boundary is
 * the Python module surface.  This library is the native layer *beneath* that surface: each
 * entry point is what a ctypes binding inside the reference's own modules would call for the hot path.  The
 * reference interface each one replaces is cited as file:line (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the parameter name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is stream-ordered and
 *     no entry point synchronises, allocates or frees device memory;
 *   - return value 0 = success, negative = error; aqg_last_error() returns a thread-local message;
 *   - `board_size` N in {3,5,7,9}; V = N*N tiles, NW = (N-1)^2 wall slots, A = V + 2*NW actions (209 at 9x9).
 *
 * state72 record (72 bytes) == State.to_array() (game_logic.py:96-100) flattened, plus plies and N:
 *   [0] player pos  [1] player walls left  [2] enemy pos (ENEMY's frame)  [3] enemy walls left
 *   [4..67] walls[64] (0 none / 1 horizontal / 2 vertical; first NW used)  [68..69] plies_played u16 LE
 *   [70] N  [71] 0
 */
#ifndef AQGNN_H
#define AQGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AQG_MAX_LEGAL 136 /* >= 5 pawn moves + 128 wall placements */
#define AQG_ABI_VERSION 10

int aqg_abi_version(void);
const char* aqg_last_error(void);
/* tuning knobs: "trunk_variant" 0/1 = exact f32-input MFMA with 1/2 workgroups per CU, 3 = all-MFMA fp16 split trunk
 * (hi*hi + hi*lo + lo*hi: fp32-equivalent products; default = the 8-wave x 2-per-CU form, also selected by 6); "step_variant" 1/0 = one-load-round MCTS step / round-1 step, "step_fast_depth" = tree depth at which
 * the fast step hands over to memory mode; "trunk_prio" = static wave priorities in the trunk (-1 = by launch size, default); "step_prio" / "heads_prio" 0..3 = wave
 * priority of the MCTS step / heads kernels (defaults 1 / 3); "step_waves" = games per step workgroup (1, 2, 4 or 8; default 8);
 * "train_fused" = form of the training step
 * (csrc/gcn_train.hip): 2 (default) one workgroup per position with every contraction in fp16 split precision on the 16-bit
 * matrix pipe (9x9 board; a position whose values leave fp16 range is redone in f32 inside the same launch, counted by
 * aqg_gcn_train_fallbacks), 1 one workgroup per position with f32-input MFMA, 0 the six-launch column-split chain, 3 = 2 with
 * every position sent through the f32 fallback (tests);
 * "trunk_phase_delay" = start offset of the second- / third-resident workgroups in units of 64 cycles, applied to
 * launches of at least "trunk_delay_min_boards" boards; "trunk_grid" = workgroups of a trunk launch (0 = default: min(boards, 512); diagnostics);
 * "use_graph" 0/1 = replay
 * a move's 3*sims+2 launches as one captured hipGraph when the stream is capturable (default 1); "profile_trunk" 0/1/2 = no event pairs / around trunk launches / around MCTS step launches */
int aqg_set_option(const char* name, int value);
/* Measurement aid (bench.py): with option "profile_trunk" = 1 a HIP event pair is recorded around every launch of the
 * dominant kernel (the GCN trunk) on its launch stream; = 2 brackets the MCTS step kernel's launches instead (`boards` then
 * counts games).  This call waits for the last recorded event, accumulates
 * the elapsed times and returns the running totals (HOST pointers; `boards` = sum of launch batch sizes incl.
 * masked-out rows); reset != 0 clears the totals.  It is the only entry point that blocks the host. */
int aqg_profile_collect(double* total_ms_host, long long* launches_host, long long* boards_host, int reset);
/* Test aid: overwrite the LDS of every CU with NaN bit patterns (stream-ordered), so a kernel that reads LDS it
 * has not written fails deterministically instead of depending on what the previous kernel left behind. */
int aqg_debug_poison_lds(void* stream);
/* Diagnostic builds (-DAQG_TRACE) only; a no-op otherwise: every workgroup of the per-simulation kernels appends
 * {kernel id, tag, start, end | blockIdx << 48} (u64 x 4, 100 MHz timestamps) to `buffer` (device memory, first u64 = entry
 * counter, zeroed by the caller, room for `capacity` entries).  tools/trace_overlap.py. */
int aqg_debug_trace(void* buffer, unsigned int capacity);

/* ------------------------------------------------------------------ game rules (game_logic.py) */

/* State.legal_actions()  game_logic.py:103-117 (+ :120-192 pawn moves, :195-357 wall legality incl. the
 * touch-count prefilter :227-307 and the jump-aware BFS :309-348), batched.
 *   mask  [B, A]  u8, 1 = legal (may be NULL)
 *   order [B, AQG_MAX_LEGAL] u8 action ids in the reference's list order (pawn U,D,L,R/jumps, then per slot
 *         H,V interleaved), entries >= count are 0xFF (may be NULL)
 *   count [B] i32 number of legal actions (may be NULL) */
int aqg_legal_actions(int board_size, const uint8_t* states72, int B, uint8_t* mask, uint8_t* order,
                      int32_t* count, void* stream);

/* State.next(action)  game_logic.py:366-391 (incl. rotate_walls :359-364), batched. */
int aqg_state_next(int board_size, const uint8_t* states72, const int32_t* actions, int B, uint8_t* out72,
                   void* stream);

/* is_lose / is_draw  game_logic.py:43-54 : flags[b] = (lose ? 1 : 0) | (draw ? 2 : 0). */
int aqg_state_status(int board_size, const uint8_t* states72, int B, int plies_for_draw, uint8_t* flags,
                     void* stream);

/* ------------------------------------------------------------------ GNN (pv_network_gnn.py) */

/* Packed weights: one float32 device buffer holding the state_dict of GraphPolicyValueNetwork
 * (pv_network_gnn.py:23-51) in kernel order.  Offsets in floats (9x9: F=6, H=128, A=209):
 *   W1  [H][8]   gcn_layers.0.lin.weight [H,F] rows padded to 8     b1 [H]
 *   W2T [H][H]   gcn_layers.1.lin.weight transposed ([k][n])        b2 [H]
 *   W3T [H][H]   gcn_layers.2.lin.weight transposed                 b3 [H]
 *   HW1T[H][H]   hidden layer of both heads, [k][unit]: unit<H/2 policy_head.0, else value_head.0    hb1 [H]
 *   PW2T[H/2][256] policy_head.2.weight transposed ([k][a], a padded to 256)   pb2 [256]
 *   VW2 [H/2]    value_head.2.weight                                vb2 [1] (+3 pad)
 *   WF2, WF3     W2^T / W3^T again in f32 MFMA B-fragment order (exact-f32 trunk variants 0/1)
 *   WH2, WH3     fp16 hi/lo planes of W2^T / W3^T in 16x16x32 MFMA B-fragment order (default trunk)
 *   WH1          fp16 hi/lo of (15/16) gcn_layers.0.lin.weight as A fragments, rows = output features, the split folded into one
 *                32-deep k block (default trunk: layer 1 runs aggregate-first, (A_hat X) W1)
 *   WHH1, WHP2   the heads' matrices as fp16 hi/lo fragments;  TB  bias rows (15/16) b sqrt(deg) per layer and degree;
 *   GUARD [4]    thresholds of the trunk's fp16-range guard on the linear maps' outputs: [0] = (65504 - max |TB of layer 2|) / 2.07
 *                (below it layer 2's aggregate provably stays under 65504), [1] = 65504, [2..3] spare (ABI 8: the last four floats);
 *                both are -1 (no maximum passes: every board is reported) when a trunk weight's fp16 hi half is not finite
 *                (|W| (16/15) >= 65504, inf or NaN)
 * aqg_gcn_packed_floats() returns the total; aqg_gcn_pack_weights_host() fills a HOST buffer from the 14
 * state_dict tensors given as HOST float32 pointers in the key order of KEYS in INTEGRATION.md. */
size_t aqg_gcn_packed_floats(int board_size);
int aqg_gcn_pack_weights_host(int board_size, const float* const* tensors_host, float* packed_host);

/* GraphPolicyValueNetwork.forward  pv_network_gnn.py:53-64 on B boards given as state72 records: node
 * features = pv_network_cnn.py:88-114 read as [V,6]; graph = wall-cut 4-neighbour grid;
 * 3 x (GCNConv + ReLU) -> global_mean_pool -> heads.  f32 data and accumulation throughout; the two 128x128
 * contractions and the neighbourhood aggregation run on the matrix cores as an fp16 hi/lo split with f32 accumulation
 * (default, fp32-equivalent) or as exact f32-input MFMA + VALU aggregation (trunk_variant 0/1, same tolerance).
 *   pooled    [B,128]  workspace/out: mean-pooled trunk features
 *   logits    [B,A]    pre-softmax policy (may be NULL)
 *   policy    [B,A]    Softmax output == module output (may be NULL)
 *   value_pre [B]      pre-tanh value (may be NULL)
 *   value     [B]      Tanh output == module output (may be NULL)
 * state_fmt: 0 = state72 records, 1 = 24-byte packed QState (engine-internal).
 * flags: AQG_GNN_EXACT_F32 forces the exact f32-input MFMA trunk and the f32 heads for this call whatever "trunk_variant" says.
 *   The fp16-split kernels hold every activation as two fp16 numbers: they are fp32-equivalent while all activations stay
 *   inside fp16 range (|x| < 65504 -- true for any sanely scaled network, saturating instead of overflowing beyond), and the
 *   host wrapper checks each weight set once against the exact kernels on calibration boards and sets this flag when they
 *   disagree (pv_network_gnn.GraphPolicyValueNetwork.packed_weights). */
#define AQG_GNN_EXACT_F32 1
/* AQG_GNN_RANGE_PROVEN (with a non-NULL `saturated` word, ignored together with AQG_GNN_EXACT_F32): the caller has PROVEN -- by a
 * bound over ALL inputs whose two walls-in-hand counts are at most AQG_GNN_PROVEN_MAX_WALLS, not by sampling -- that no value the
 * split trunk holds as an fp16 pair can leave fp16 range for this weight set.  The trunk then skips its per-value range tracking
 * (one vector instruction per stored value, 4 % of the kernel) and checks each record's two wall counts against that maximum
 * instead: a record beyond it raises the word exactly as an out-of-range value would.  The bound GraphPolicyValueNetwork uses
 * (pv_network_gnn._range_proven): with R = 0.2 + 4/sqrt(10) = 1.465 >= every row sum of A_hat on a wall-cut grid and x_max =
 * (1, W, 1, W, 1, 1) for W = AQG_GNN_PROVEN_MAX_WALLS,  z_1 = |W_1| x_max,  h_l = R z_l + |b_l|,  z_{l+1} = |W_{l+1}| h_l;  proven
 * iff 2.3 max(z_l, h_l, |W_l|) < 65504 (2.3 covers the kernel's internal scales: planes c sqrt(deg) H <= 2.10 h, linear-map
 * outputs sqrt(deg) Z <= 2.24 z, weights W / c). */
#define AQG_GNN_RANGE_PROVEN 2
#define AQG_GNN_PROVEN_MAX_WALLS 16
int aqg_gcn_forward_boards(int board_size, const void* states, int state_fmt, int B, const float* packed,
                           float* pooled, float* logits, float* policy, float* value_pre, float* value,
                           int flags, void* stream);
/* The same call with the runtime fp16-range guard: `saturated` (device int32, may be NULL) is OR-ed with 1 by any fp16-split
 * kernel of the call that met a value it cannot hold as an fp16 pair -- a linear-map output or a post-ReLU activation beyond
 * 65504 (the activation is clamped there, never inf / NaN), a pooled feature or head hidden unit beyond it -- or could not rule one
 * out (the trunk bounds layer 2's aggregate by its linear map's output: a linear-map output beyond GUARD[0], about 31,600, is reported).  The results of such
 * a call are finite but not the network's: the caller repeats it with AQG_GNN_EXACT_F32 and keeps that flag for the weight set
 * (GraphPolicyValueNetwork.forward_states / predict do; the engine reports the same event in counters[5]).  The word is never
 * cleared by the library.  Exact-f32 calls never set it.  Reference behaviour: fp32 throughout, no cliff (pv_network_gnn.py:53-64). */
int aqg_gcn_forward_boards_guarded(int board_size, const void* states, int state_fmt, int B, const float* packed,
                                   float* pooled, float* logits, float* policy, float* value_pre, float* value,
                                   int flags, int32_t* saturated, void* stream);
/* Same network on an arbitrary batched graph: forward(x, edge_index, batch)  pv_network_gnn.py:53.
 *   x [num_nodes, F] f32;  csr_ptr [num_nodes+1] i32 / csr_src [E'] i32 / csr_w [E'] f32 : incoming edges of
 *   each node INCLUDING self loops with the gcn_norm weights already attached (built by the host wrapper
 *   from edge_index with torch ops);  graph_ptr [num_graphs+1] i32 node ranges (batch must be sorted);
 *   work0/work1 [num_nodes,128] f32 scratch. */
/* The same forward for ANY board size 3/5/7/9: 9x9 dispatches to the fused kernels above (workspace unused), the smaller
 * boards of the reference's constants.py:5-20 run on plain kernels (features + ELL adjacency -> linear / gather x3 -> pool
 * -> heads) and need `workspace` of aqg_gcn_boards_any_workspace_floats(board_size, B) floats. */
size_t aqg_gcn_boards_any_workspace_floats(int board_size, int B);
int aqg_gcn_forward_boards_any(int board_size, const void* states, int state_fmt, int B, const float* packed, float* workspace,
                               size_t workspace_floats, float* pooled, float* logits, float* policy, float* value_pre, float* value,
                               int flags, void* stream);

int aqg_gcn_forward_graph(int num_features, int num_actions, const float* x, int num_nodes,
                          const int32_t* csr_ptr, const int32_t* csr_src, const float* csr_w,
                          const int32_t* graph_ptr, int num_graphs, const float* packed, float* work0,
                          float* work1, float* pooled, float* logits, float* policy, float* value_pre,
                          float* value, void* stream);

/* ------------------------------------------------------------------ batched PV-MCTS self-play (pv_mcts.py, self_play.py) */

/* All engine memory is owned by the caller (the Python host allocates torch tensors); this struct only
 * carries device pointers + sizes.  G concurrent games, one wavefront per game, lock-step simulations. */
typedef struct aqg_engine {
    int32_t board_size, num_walls, plies_for_draw;
    int32_t num_games;        /* G: concurrent game slots */
    int32_t quota;            /* games to play on those slots in all, >= G: a finished slot takes the next game not yet handed out
                                 (self_play.py:81-84 is a plain loop over games).  quota == G: one lock-step generation */
    int32_t sims;             /* PV_EVALUATE_COUNT  pv_mcts.py:18 */
    int32_t node_cap;         /* nodes per game tree >= 1 + sims * AQG_MAX_LEGAL */
    int32_t max_plies;        /* history rows per game slot (>= plies_for_draw) */
    int32_t prior_mode;       /* 0: network policy gathered at legal actions + renormalised (pv_network_cnn.py:129-132)
                                 1: `fake` integer-hash evaluator (tests; oracle/mcts.py FakeModel)
                                 2: external evaluator -- the caller's own model.predict (BaseNetwork.py:36-40) fills policy / value
                                    between aqg_engine_step calls (see below) */
    int32_t fake_bias;
    int32_t gnn_flags;        /* flags of the GNN forward for prior_mode 0 (AQG_GNN_EXACT_F32 or 0) */
    float c_puct;             /* 1.25  pv_mcts.py:71 */
    float temperature;        /* SP_TEMPERATURE self_play.py:20; 1.0 exact, 0 = argmax */
    /* tree pool: [G * node_cap] 32-byte node records, two aligned 16-byte halves (ABI 9): {f64 w, f32 p, u32 action} -- read only of the
     * child a descent chooses -- and {i32 n, u32 first_child|count<<24, f32 q = f32(-w/n), f32 cp = f32(c_puct * p)} -- what PUCT
     * scores every child with (pv_mcts.py:69-78) */
    void* node_rec;
    /* per game, [G] */
    int32_t* node_count; uint8_t* root_state /* [G,24] */; int32_t* path /* [G, sims+2] */; int32_t* path_len;
    uint8_t* leaf_flag /* [G] 1 = this simulation's leaf needs an evaluation, 2 = it was served from the evaluation cache (below), 0 = no leaf (terminal / idle) */; uint8_t* leaf_state /* [G,24] */;
    uint8_t* game_active /* [G] slot is playing */; int32_t* slot_game /* [G] index of the game the slot is playing, -1 = retired */;
    /* per game, [quota] */
    int32_t* game_plies /* rows recorded */; int8_t* game_result /* z of ply 0 */; uint8_t* game_done /* 1 = finished */;
    int32_t* game_slot /* slot that played / plays it, -1 = not started */; int32_t* game_first_move /* index of its first aqg_engine_move */;
    /* evaluation buffers, [G,...] */
    uint8_t* legal_order /* [G,AQG_MAX_LEGAL] */; int32_t* legal_count; float* pooled /* [G,128] */;
    float* policy /* [G,A] */; float* value /* [G] */;
    /* history, per game: [quota, max_plies, ...] */
    uint8_t* hist_state72; uint16_t* hist_visits /* [G,max_plies,A] root child visit counts, dense by action */;
    uint8_t* hist_action /* [G,max_plies] */;
    /* counters [8] i32: 0 active slots, 1 finished games, 2 dead-end aborts, 3 next game index to hand out, 4 moves made
     * (updated once per move), 5 fp16-range guard: set to 1 by a GNN evaluation of this engine that met a value outside fp16 range
     * (see aqg_gcn_forward_boards_guarded; cleared by aqg_engine_reset only) */
    int32_t* counters;
    /* per-game statistics [G] i32 (summed by the host): network evaluations, simulations that ended on a terminal node */
    int32_t* stat_leaf_evals; int32_t* stat_terminal_sims;
    const float* packed_weights;
    /* boards other than 9x9 with prior_mode 0: workspace of the any-size forward, aqg_gcn_boards_any_workspace_floats(N, G)
     * floats (may be NULL for 9x9 and for prior_mode 1) */
    float* gnn_workspace;
    /* Evaluation cache (ABI 10; prior_mode 0 only; eval_cache_keys == NULL: off).  The reference builds a new tree for every move
     * (pv_mcts.py:84) and keeps no transposition table, so a game asks model.predict (pv_mcts.py:47) for the same position again and
     * again: transpositions inside a search, and the sub-tree of the move that was played in the next search.  The network's output and
     * legal_actions() are pure functions of (walls, pawns, walls in hand) -- not of the ply counter -- and the fused kernels compute every
     * board independently of its launch, so a per-slot table keyed by those 20 bytes returns bit-identical priors, values and legal
     * lists: the search, the visit counts and the game records do not change, only the leaf is not sent through the network (nor through
     * the legal-move kernel) again.  Per slot 1 << eval_cache_log2 entries: a 32-byte key record {u64 hw, u64 vw, u32 ppos|pwl<<8|
     * epos<<16|ewl<<24, u32 state (0 empty, 2 filled), i32 legal count, f32 value} + a 704-byte row {f32 priors[AQG_MAX_LEGAL] over
     * legal_actions() in order, renormalised (pv_network_cnn.py:129-132), u8 actions[AQG_MAX_LEGAL]}.  A slot keeps its table over its
     * games (positions stay valid); aqg_engine_reset clears it, and so must the caller when the weights change
     * (aqg_engine_clear_eval_cache).  leaf_flag[g] == 2 marks a leaf that was served from the table; eval_mask[g] == 1 the leaves the
     * network evaluates (the `active` mask of the GNN launches). */
    void* eval_cache_keys; void* eval_cache_rows;
    int32_t* eval_cache_slot /* [G] entry reserved for the pending evaluation, -1 none */; uint8_t* eval_mask /* [G] */;
    int32_t* stat_cache_hits /* [G] */;
    /* optional (both or neither): for sets of more than 512 slots aqg_engine_move hands the leaves that miss the table to the trunk
     * launch of simulation s as a compact list -- eval_list[0 .. eval_count[s]) -- instead of a mask to walk */
    int32_t* eval_list /* [G] */; int32_t* eval_count /* [sims + 1] */;
    int32_t eval_cache_log2;
} aqg_engine;

/* Reset all G slots to the initial position (State() game_logic.py:25-40) and mark them active; clears the evaluation cache. */
int aqg_engine_reset(const aqg_engine* e_host, void* stream);
/* Empty the evaluation cache of every slot (after a weight update). */
int aqg_engine_clear_eval_cache(const aqg_engine* e_host, void* stream);
/* One self-play move for every active game == pv_mcts_policy (pv_mcts.py:20-95, `sims` lock-step simulations:
 * select :69-78, terminal/leaf evaluate :33-57, backup) + the body of play() (self_play.py:45-60): record
 * (state, visit counts), sample the action with uniforms[g] exactly like np.random.choice (self_play.py:57),
 * apply next(); finished games get z (self_play.py:22-27,:63-66) and go inactive.
 *   uniforms [G] f64 in [0,1). */
int aqg_engine_move(const aqg_engine* e_host, const double* uniforms, void* stream);
/* The same move in pieces, for an evaluator the library does not own (prior_mode 2; pv_mcts.py:47 calls model.predict on
 * whatever BaseNetwork it is given):
 *     aqg_engine_begin_move
 *     for sim in 0 .. sims-1:  aqg_engine_step(do_expand = sim > 0, do_select = 1)
 *                              -> for every game with leaf_flag[g] == 1 the caller writes policy[g][0 .. legal_count[g]) = PMF over
 *                                 the leaf's legal_actions() IN ORDER (legal_order[g]) and value[g]; leaf_state[g] is the 24-byte leaf
 *     aqg_engine_step(1, 0); aqg_engine_finish_move(uniforms)        (or aqg_engine_root_visits for a search only)
 * Per game exactly the reference's loop, one predict per simulation.  aqg_engine_set_roots installs caller-supplied roots. */
int aqg_engine_begin_move(const aqg_engine* e_host, void* stream);
int aqg_engine_step(const aqg_engine* e_host, int do_expand, int do_select, void* stream);
int aqg_engine_finish_move(const aqg_engine* e_host, const double* uniforms, void* stream);
int aqg_engine_set_roots(const aqg_engine* e_host, const uint8_t* root_states72, void* stream);
/* pv_mcts_policy only, for caller-supplied root states (no history, no transition): after the call
 * node_n of the root's children holds the visit counts; `root_states72` [G,72]. */
int aqg_engine_search(const aqg_engine* e_host, const uint8_t* root_states72, void* stream);
/* Read back the root's children after a search: visits [G,AQG_MAX_LEGAL] i32, actions [G,AQG_MAX_LEGAL] u8, count [G]. */
int aqg_engine_root_visits(const aqg_engine* e_host, int32_t* visits, uint8_t* actions, int32_t* count, void* stream);

/* ------------------------------------------------------------------ training step (train_network.py:68-95 on the GNN) */

/* One optimisation step on a batch of positions: forward, the reference's losses (CrossEntropyLoss applied to the
 * already-softmaxed policy with probability targets train_network.py:54,85 + MSELoss on the tanh value :55,86),
 * backward, and torch.optim.Adam's update (:56,:90-92).  f32 data and f32 accumulation; two launches per step: one workgroup per
 * position for forward + backward, one fixed-order reduction + Adam (csrc/gcn_train.hip).  On the 9x9 board (option "train_fused" 2,
 * the default) every contraction runs on the 16-bit matrix pipe in fp16 hi/lo split precision (three fp16 products per f32
 * product, the arithmetic of the inference trunk; a position whose values leave fp16 range is redone by the f32-input MFMA body inside
 * the same launch, aqg_gcn_train_fallbacks counts them); smaller boards and "train_fused" 1 use f32-input MFMA.  mode 0 = gradients only (into grads),
 * 1 = gradients + update, 2 = update only from whatever grads holds -- data-parallel training computes local gradients
 * (mode 0), all-reduces them over RCCL, and applies them (mode 2).  The 14 parameter tensors
 * are the state_dict tensors themselves in their PyTorch layouts and in the key order of KEYS in INTEGRATION.md; grads,
 * adam_m, adam_v have the same shapes.  All memory is the caller's (device pointers); nothing allocates or synchronises.
 * B = batch (the capacity the workspace was sized for is the caller's business), V = board_size^2, A = policy size. */
typedef struct aqg_train {
    int32_t board_size, batch, policy_size;
    int32_t step;                 /* Adam step count of THIS update, >= 1 */
    float lr, beta1, beta2, eps;  /* 1e-3 * LambdaLR factor (train_network.py:56-66), 0.9, 0.999, 1e-8 */
    float* params[14]; float* grads[14]; float* adam_m[14]; float* adam_v[14];
    /* workspace */
    float* h1; float* h2;         /* [B*96, 128] (96 rows per position on every board size): post-ReLU activations of layers 1, 2 -- node
                                   * rows in the f32 forms, this round's parked fp16 hi / lo fragments in the split form */
    float* h3;                    /* [B*V, 128] layer 3 (six-launch form only) */
    float* zbuf; float* dh;       /* [B*V, 128] dZ = A_hat dP of layer 3 / layer 2 (the next launch contracts over full rows) */
    float* g; float* dg;          /* [B, 128]   pooled features and their gradient */
    float* hp; float* hv; float* dhp; float* dhv;   /* [B, 64] head hidden layers and gradients */
    float* lg;                    /* [B, A]     d loss / d logits */
    float* pol;                   /* [B, A]     softmax policy (the network output) */
    float* vp; float* val;        /* [B]        d loss / d pre-tanh value; tanh value */
    float* loss;                  /* [B, 2]     per-position policy / value loss terms (their means are the two losses) */
    float* part;                  /* [B * AQG_TRAIN_PART_FLOATS] per-board partial sums of the trunk's weight and bias gradients */
} aqg_train;
#define AQG_TRAIN_PART_FLOATS (2 * 128 * 128 + 128 * 6 + 3 * 128)
int aqg_gcn_train_step(const aqg_train* t_host, const uint8_t* states72, const float* pi_target, const float* z_target,
                       int mode, void* stream);
/* A run of consecutive single-process steps (mode 1) with no host work in between -- one epoch of train_network.py:72-95:
 * step i takes positions order[i*batch .. (i+1)*batch) (int64 indices into the resident arrays states72 [n,72],
 * pi_target [n,A], z_target [n]; order = NULL means 0..positions-1; the last batch may be short, as the reference's
 * DataLoader keeps it), t->step is the Adam count of the FIRST step, and every step adds its two batch-mean losses to
 * loss_sums[2] (device, may be NULL) -- the per-epoch sums train_network.py:89-90 prints. */
int aqg_gcn_train_steps(const aqg_train* t_host, const uint8_t* states72, const float* pi_target, const float* z_target,
                        const int64_t* order, long long positions, float* loss_sums, void* stream);
/* Positions that the split-precision step ("train_fused" 2) has redone in f32 since the last reset (synchronises the device);
 * -1 on error.  A diagnostic: results do not depend on it. */
long long aqg_gcn_train_fallbacks(int reset);

/* ------------------------------------------------------------------ CPU baseline agents (agents.py) -- HOST pointers, host code */

/* The reference's baseline opponents (agents.py:14-214) are CPU code; so are these: the host instantiation of the rule header
 * the kernels compile, exported from the same library.  `rec72_host` = one state72 record in HOST memory.
 *   aqg_host_legal_actions   State.legal_actions() game_logic.py:103-117 -> ordered ids in out136_host[AQG_MAX_LEGAL], returns the count
 *   aqg_host_next            State.next(action) game_logic.py:366-391
 *   aqg_host_shortest_path   shortest_path_bfs agents.py:27-41: plies to the goal row over legal_actions_pos (jumps, frozen enemy), -1 = none
 *   aqg_host_heuristic_eval  heuristic_eval agents.py:22-54: (enemy's path - mover's path) / max_dist_from_goal
 *   aqg_host_alpha_beta_action  alpha_beta_action agents.py:60-108: depth-limited negamax with that heuristic, first best action; -1 = none */
int aqg_host_legal_actions(int board_size, const uint8_t* rec72_host, uint8_t* out136_host);
int aqg_host_next(int board_size, const uint8_t* rec72_host, int action, uint8_t* out72_host);
int aqg_host_shortest_path(int board_size, const uint8_t* rec72_host);
double aqg_host_heuristic_eval(int board_size, const uint8_t* rec72_host, int max_dist_from_goal);
int aqg_host_alpha_beta_action(int board_size, const uint8_t* rec72_host, int plies_for_draw, int max_dist_from_goal, int max_depth);

#ifdef __cplusplus
}
#endif
#endif /* AQGNN_H */
