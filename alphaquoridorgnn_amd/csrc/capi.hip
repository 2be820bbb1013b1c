// capi.hip -- extern "C" entry points of libaqgnn_hip.so (declared in include/aqgnn.h).
#include "aqg_common.hpp"
#include "../../include/aqgnn.h"

namespace aqg {
thread_local char g_err[512] = "";
extern int g_trunk_variant;
extern int g_profile_trunk;
extern int g_trunk_prio;
extern int g_heads_prio;
extern int g_train_fused;
extern int g_trunk_grid;
extern int g_trunk_phase_delay;
extern int g_trunk_delay_min_boards;
extern int g_use_graph;
extern int g_step_variant;
extern int g_step_waves;
extern int g_step_prio;
extern int g_step_fast_depth;
int profile_collect(double* total_ms, long long* launches, long long* boards, int reset);
int launch_poison_lds(hipStream_t st);
int set_trace_gcn(void* buf, unsigned int cap);
int set_trace_mcts(void* buf, unsigned int cap);

size_t packed_floats();
int pack_weights_host(int N, const float* const* t, float* out);
int launch_legal_actions(int N, const void* states, int fmt, int B, uint8_t* mask, uint8_t* order, int32_t* count,
                         const uint8_t* active, hipStream_t st);
int launch_state_next(int N, const uint8_t* in, const int32_t* actions, int B, uint8_t* out, hipStream_t st);
int launch_state_status(int N, const uint8_t* in, int B, int draw, uint8_t* flags, hipStream_t st);
int launch_gcn_forward_boards(int N, const void* states, int fmt, int B, const float* packed, float* pooled,
                              float* logits, float* policy, float* value_pre, float* value, const uint8_t* active,
                              int flags, int32_t* saturated, hipStream_t st, const int32_t* list = nullptr, const int32_t* list_count = nullptr);
size_t boards_any_workspace_floats(int N, int B);
int launch_gcn_forward_boards_any(int N, const void* states, int fmt, int B, const float* packed, float* workspace,
                                  size_t workspace_floats, float* pooled, float* logits, float* policy, float* value_pre,
                                  float* value, const uint8_t* active, int flags, int32_t* saturated, hipStream_t st, const int32_t* list = nullptr, const int32_t* list_count = nullptr);
int launch_gcn_forward_graph(int F, int A, const float* x, int num_nodes, const int32_t* csr_ptr, const int32_t* csr_src,
                             const float* csr_w, const int32_t* graph_ptr, int num_graphs, const float* packed,
                             float* work0, float* work1, float* pooled, float* logits, float* policy, float* value_pre,
                             float* value, hipStream_t st);
int engine_reset(const aqg_engine& e, hipStream_t st);
int engine_clear_eval_cache(const aqg_engine& e, hipStream_t st);
int engine_begin_move(const aqg_engine& e, hipStream_t st);
int engine_step(const aqg_engine& e, int do_expand, int do_select, hipStream_t st);
int engine_finish_move(const aqg_engine& e, const double* uniforms, hipStream_t st);
int engine_set_roots(const aqg_engine& e, const uint8_t* roots72, hipStream_t st);
int engine_move(const aqg_engine& e, const double* uniforms, hipStream_t st);
int engine_search(const aqg_engine& e, const uint8_t* roots72, hipStream_t st);
int engine_root_visits(const aqg_engine& e, int32_t* visits, uint8_t* actions, int32_t* count, hipStream_t st);
int train_step(const aqg_train& t, const uint8_t* states72, const float* pi, const float* z, int mode, hipStream_t st);
long long train_fallbacks(int reset);
int train_steps(const aqg_train& t, const uint8_t* states72, const float* pi, const float* z, const int64_t* order, long long positions,
                float* loss_sums, hipStream_t st);
}  // namespace aqg

using namespace aqg;

extern "C" {

int aqg_abi_version(void) { return AQG_ABI_VERSION; }
const char* aqg_last_error(void) { return g_err; }

int aqg_set_option(const char* name, int value) {
    if (name && !strcmp(name, "trunk_variant")) { if (!(value == 0 || value == 1 || value == 3 || value == 6)) return fail("trunk_variant must be 0, 1, 3 or 6"); g_trunk_variant = value; return 0; }
    if (name && !strcmp(name, "heads_prio")) { g_heads_prio = value < 0 ? 0 : (value > 3 ? 3 : value); return 0; }
    if (name && !strcmp(name, "trunk_prio")) { g_trunk_prio = value < 0 ? -1 : (value & 15); return 0; }
    if (name && !strcmp(name, "trunk_grid")) { g_trunk_grid = value; return 0; }
    if (name && !strcmp(name, "trunk_phase_delay")) { if (value < 0 || value > 4096) return fail("trunk_phase_delay out of range"); g_trunk_phase_delay = value; return 0; }
    if (name && !strcmp(name, "trunk_delay_min_boards")) { g_trunk_delay_min_boards = value; return 0; }
    if (name && !strcmp(name, "step_prio")) { g_step_prio = value & 3; return 0; }
    if (name && !strcmp(name, "step_waves")) { g_step_waves = value; return 0; }
    if (name && !strcmp(name, "step_variant")) { g_step_variant = value ? 1 : 0; return 0; }
    if (name && !strcmp(name, "step_fast_depth")) { if (value < 0 || value > 61) return fail("step_fast_depth must be 0..61"); g_step_fast_depth = value; return 0; }
    if (name && !strcmp(name, "train_fused")) { if (value < 0 || value > 3) return fail("train_fused: 0..3"); g_train_fused = value; return 0; }
    if (name && !strcmp(name, "use_graph")) { g_use_graph = value ? 1 : 0; return 0; }
    if (name && !strcmp(name, "profile_trunk")) { g_profile_trunk = (value == 1 || value == 2) ? value : 0; return 0; }   // 1 = trunk launches, 2 = step launches
    return fail("unknown option", name ? name : "(null)");
}

int aqg_debug_poison_lds(void* stream) { return launch_poison_lds((hipStream_t)stream); }

int aqg_debug_trace(void* buffer, unsigned int capacity) {
    if (set_trace_gcn(buffer, capacity) || set_trace_mcts(buffer, capacity)) return fail("aqg_debug_trace: hipMemcpyToSymbol");
    return 0;
}

int aqg_profile_collect(double* total_ms_host, long long* launches_host, long long* boards_host, int reset) {
    return profile_collect(total_ms_host, launches_host, boards_host, reset);
}

int aqg_legal_actions(int board_size, const uint8_t* states72, int B, uint8_t* mask, uint8_t* order, int32_t* count,
                      void* stream) {
    if (B < 0 || (B > 0 && !states72)) return fail("aqg_legal_actions: bad arguments");
    return launch_legal_actions(board_size, states72, 0, B, mask, order, count, nullptr, (hipStream_t)stream);
}

int aqg_state_next(int board_size, const uint8_t* states72, const int32_t* actions, int B, uint8_t* out72, void* stream) {
    if (B < 0 || (B > 0 && (!states72 || !actions || !out72))) return fail("aqg_state_next: bad arguments");
    return launch_state_next(board_size, states72, actions, B, out72, (hipStream_t)stream);
}

int aqg_state_status(int board_size, const uint8_t* states72, int B, int plies_for_draw, uint8_t* flags, void* stream) {
    if (B < 0 || (B > 0 && (!states72 || !flags))) return fail("aqg_state_status: bad arguments");
    return launch_state_status(board_size, states72, B, plies_for_draw, flags, (hipStream_t)stream);
}

size_t aqg_gcn_packed_floats(int board_size) { (void)board_size; return packed_floats(); }

int aqg_gcn_pack_weights_host(int board_size, const float* const* tensors_host, float* packed_host) {
    if (!tensors_host || !packed_host) return fail("aqg_gcn_pack_weights_host: null argument");
    for (int i = 0; i < 14; ++i) if (!tensors_host[i]) return fail("aqg_gcn_pack_weights_host: null tensor");
    return pack_weights_host(board_size, tensors_host, packed_host);
}

int aqg_gcn_forward_boards(int board_size, const void* states, int state_fmt, int B, const float* packed, float* pooled,
                           float* logits, float* policy, float* value_pre, float* value, int flags, void* stream) {
    if (B < 0 || (B > 0 && (!states || !packed))) return fail("aqg_gcn_forward_boards: bad arguments");
    if (state_fmt != 0 && state_fmt != 1) return fail("aqg_gcn_forward_boards: state_fmt must be 0 or 1");
    return launch_gcn_forward_boards(board_size, states, state_fmt, B, packed, pooled, logits, policy, value_pre, value, nullptr,
                                     flags, nullptr, (hipStream_t)stream);
}

int aqg_gcn_forward_boards_guarded(int board_size, const void* states, int state_fmt, int B, const float* packed, float* pooled,
                                   float* logits, float* policy, float* value_pre, float* value, int flags, int32_t* saturated,
                                   void* stream) {
    if (B < 0 || (B > 0 && (!states || !packed))) return fail("aqg_gcn_forward_boards_guarded: bad arguments");
    if (state_fmt != 0 && state_fmt != 1) return fail("aqg_gcn_forward_boards_guarded: state_fmt must be 0 or 1");
    return launch_gcn_forward_boards(board_size, states, state_fmt, B, packed, pooled, logits, policy, value_pre, value, nullptr,
                                     flags, saturated, (hipStream_t)stream);
}

size_t aqg_gcn_boards_any_workspace_floats(int board_size, int B) { return B > 0 ? boards_any_workspace_floats(board_size, B) : 0; }

int aqg_gcn_forward_boards_any(int board_size, const void* states, int state_fmt, int B, const float* packed, float* workspace,
                               size_t workspace_floats, float* pooled, float* logits, float* policy, float* value_pre, float* value,
                               int flags, void* stream) {
    if (B < 0 || (B > 0 && (!states || !packed))) return fail("aqg_gcn_forward_boards_any: bad arguments");
    if (state_fmt != 0 && state_fmt != 1) return fail("aqg_gcn_forward_boards_any: state_fmt must be 0 or 1");
    return launch_gcn_forward_boards_any(board_size, states, state_fmt, B, packed, workspace, workspace_floats, pooled, logits, policy,
                                         value_pre, value, nullptr, flags, nullptr, (hipStream_t)stream);
}

int aqg_gcn_forward_graph(int num_features, int num_actions, const float* x, int num_nodes, const int32_t* csr_ptr,
                          const int32_t* csr_src, const float* csr_w, const int32_t* graph_ptr, int num_graphs,
                          const float* packed, float* work0, float* work1, float* pooled, float* logits, float* policy,
                          float* value_pre, float* value, void* stream) {
    if (!x || !csr_ptr || !csr_src || !csr_w || !graph_ptr || !packed || !work0 || !work1 || !pooled)
        return fail("aqg_gcn_forward_graph: null argument");
    return launch_gcn_forward_graph(num_features, num_actions, x, num_nodes, csr_ptr, csr_src, csr_w, graph_ptr, num_graphs,
                                    packed, work0, work1, pooled, logits, policy, value_pre, value, (hipStream_t)stream);
}

int aqg_engine_reset(const aqg_engine* e, void* stream) {
    if (!e) return fail("aqg_engine_reset: null engine");
    return engine_reset(*e, (hipStream_t)stream);
}
int aqg_engine_clear_eval_cache(const aqg_engine* e, void* stream) {
    if (!e) return fail("aqg_engine_clear_eval_cache: null engine");
    return engine_clear_eval_cache(*e, (hipStream_t)stream);
}
int aqg_engine_move(const aqg_engine* e, const double* uniforms, void* stream) {
    if (!e || !uniforms) return fail("aqg_engine_move: null argument");
    return engine_move(*e, uniforms, (hipStream_t)stream);
}
int aqg_engine_begin_move(const aqg_engine* e, void* stream) {
    if (!e) return fail("aqg_engine_begin_move: null engine");
    return engine_begin_move(*e, (hipStream_t)stream);
}
int aqg_engine_step(const aqg_engine* e, int do_expand, int do_select, void* stream) {
    if (!e) return fail("aqg_engine_step: null engine");
    return engine_step(*e, do_expand ? 1 : 0, do_select ? 1 : 0, (hipStream_t)stream);
}
int aqg_engine_finish_move(const aqg_engine* e, const double* uniforms, void* stream) {
    if (!e || !uniforms) return fail("aqg_engine_finish_move: null argument");
    return engine_finish_move(*e, uniforms, (hipStream_t)stream);
}
int aqg_engine_set_roots(const aqg_engine* e, const uint8_t* root_states72, void* stream) {
    if (!e || !root_states72) return fail("aqg_engine_set_roots: null argument");
    return engine_set_roots(*e, root_states72, (hipStream_t)stream);
}
int aqg_engine_search(const aqg_engine* e, const uint8_t* root_states72, void* stream) {
    if (!e || !root_states72) return fail("aqg_engine_search: null argument");
    return engine_search(*e, root_states72, (hipStream_t)stream);
}
int aqg_engine_root_visits(const aqg_engine* e, int32_t* visits, uint8_t* actions, int32_t* count, void* stream) {
    if (!e || !visits || !actions || !count) return fail("aqg_engine_root_visits: null argument");
    return engine_root_visits(*e, visits, actions, count, (hipStream_t)stream);
}

int aqg_gcn_train_step(const aqg_train* t, const uint8_t* states72, const float* pi_target, const float* z_target, int mode,
                       void* stream) {
    if (!t || mode < 0 || mode > 2) return fail("aqg_gcn_train_step: bad argument");
    if (mode != 2 && (!states72 || !pi_target || !z_target)) return fail("aqg_gcn_train_step: null argument");
    for (int i = 0; i < 14; ++i)
        if (!t->params[i] || !t->grads[i] || (mode >= 1 && (!t->adam_m[i] || !t->adam_v[i]))) return fail("aqg_gcn_train_step: null parameter tensor");
    if (mode >= 1 && t->step < 1) return fail("aqg_gcn_train_step: step must be >= 1");
    return train_step(*t, states72, pi_target, z_target, mode, (hipStream_t)stream);
}
long long aqg_gcn_train_fallbacks(int reset) { return train_fallbacks(reset); }
int aqg_gcn_train_steps(const aqg_train* t, const uint8_t* states72, const float* pi_target, const float* z_target, const int64_t* order,
                        long long positions, float* loss_sums, void* stream) {
    if (!t || !states72 || !pi_target || !z_target || positions < 0 || positions > 0x7fffffffLL) return fail("aqg_gcn_train_steps: bad argument");
    for (int i = 0; i < 14; ++i)
        if (!t->params[i] || !t->grads[i] || !t->adam_m[i] || !t->adam_v[i]) return fail("aqg_gcn_train_steps: null parameter tensor");
    if (t->step < 1) return fail("aqg_gcn_train_steps: step must be >= 1");
    return train_steps(*t, states72, pi_target, z_target, order, positions, loss_sums, (hipStream_t)stream);
}

}  // extern "C"
