// host_agents.cpp -- CPU baseline opponents of the reference's agents.py (random / alpha-beta / rollout MCTS), native.
//
// The reference's agents ARE CPU code (agents.py:14-214; used by evaluate_agents.py:62-89 for strength tracking): this is
// their host-side counterpart, not a fallback of the GPU path.  It runs the SAME rule header the kernels compile
// (quoridor_core.hpp, host instantiation), so a baseline opponent and the engine can never disagree about the rules.
//   aqg_host_legal_actions / aqg_host_next   State.legal_actions() / State.next()   game_logic.py:103-117, :366-391
//   aqg_host_shortest_path                   shortest_path_bfs       agents.py:27-41  (BFS over legal_actions_pos: jumps, static enemy)
//   aqg_host_heuristic_eval                  heuristic_eval          agents.py:22-54  ((enemy's path - mover's path) / MAX_DIST_FROM_GOAL)
//   aqg_host_alpha_beta_action               alpha_beta_action       agents.py:60-108 (depth-limited negamax, first best action wins)
#include <limits>
#include <cstdint>
#include <cstring>
#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>      // (diagnostic one-command builds push this file through hipcc too; build.sh uses g++)
#endif
#include "quoridor_core.hpp"
#include "../../include/aqgnn.h"

namespace {
using namespace aqg;

template <int N> int shortest_path(const QState& s) {
    constexpr int V = N * N;
    const Open o = make_open<N>(s.hw, s.vw);
    const int enemy = V - 1 - s.epos;                  // the other pawn in the mover's frame (game_logic.py:136)
    int depth_of[V];
    for (int i = 0; i < V; ++i) depth_of[i] = -1;
    int queue[V], head = 0, tail = 0;
    queue[tail++] = s.ppos; depth_of[s.ppos] = 0;
    while (head < tail) {                              // FIFO, children in legal_actions_pos order: the depth of the first goal tile popped
        const int p = queue[head++];
        if (p / N == 0) return depth_of[p];
        uint8_t nxt[8];
        const int c = legal_pos_list<N>(o, p, enemy, nxt);
        for (int i = 0; i < c; ++i)
            if (depth_of[nxt[i]] < 0) { depth_of[nxt[i]] = depth_of[p] + 1; queue[tail++] = nxt[i]; }
    }
    return -1;
}

// the position seen by the other side WITHOUT a move: rotate the walls, swap the pawns (agents.py:46-47)
template <int N> QState flipped(const QState& s) {
    constexpr int NW = (N - 1) * (N - 1);
    QState r = s;
    r.hw = brev64(s.hw) >> (64 - NW); r.vw = brev64(s.vw) >> (64 - NW);
    r.ppos = s.epos; r.pwl = s.ewl; r.epos = s.ppos; r.ewl = s.pwl;
    return r;
}

template <int N> double heuristic(const QState& s, int max_dist) {
    const int sp = shortest_path<N>(s), se = shortest_path<N>(flipped<N>(s));
    return (double)(se - sp) / (double)max_dist;
}

template <int N> double alpha_beta(const QState& s, double alpha, double beta, int depth, int draw, int max_dist) {
    const bool lose = is_lose<N>(s), dr = is_draw(s, draw);
    if (depth == 0 || lose || dr) {                    // agents.py:71-76
        if (lose) return -1.0;
        if (dr) return 0.0;
        return heuristic<N>(s, max_dist);
    }
    uint8_t acts[MAX_LEGAL];
    const int n = legal_actions_serial<N>(s, acts);
    for (int i = 0; i < n; ++i) {
        const double score = -alpha_beta<N>(next_state<N>(s, acts[i]), -beta, -alpha, depth - 1, draw, max_dist);
        if (score > alpha) alpha = score;
        if (alpha >= beta) return alpha;               // beta cutoff
    }
    return alpha;
}

template <int N> int alpha_beta_action(const QState& s, int draw, int max_dist, int max_depth) {
    uint8_t acts[MAX_LEGAL];
    const int n = legal_actions_serial<N>(s, acts);
    int best = -1;
    double alpha = -1.0 / 0.0;
    for (int i = 0; i < n; ++i) {
        const double score = -alpha_beta<N>(next_state<N>(s, acts[i]), -1.0 / 0.0, -alpha, max_depth, draw, max_dist);
        if (score > alpha) { best = acts[i]; alpha = score; }
    }
    return best;
}
}  // namespace

#define AQG_HOST_DISPATCH(N, EXPR3, EXPR5, EXPR7, EXPR9) \
    switch (N) { case 3: return EXPR3; case 5: return EXPR5; case 7: return EXPR7; case 9: return EXPR9; default: return -1; }

extern "C" {

int aqg_host_legal_actions(int board_size, const uint8_t* rec72, uint8_t* out136) {
    if (!rec72 || !out136) return -1;
    const QState s = unpack72(rec72);
    AQG_HOST_DISPATCH(board_size, legal_actions_serial<3>(s, out136), legal_actions_serial<5>(s, out136), legal_actions_serial<7>(s, out136),
                      legal_actions_serial<9>(s, out136))
}

int aqg_host_next(int board_size, const uint8_t* rec72, int action, uint8_t* out72) {
    if (!rec72 || !out72) return -1;
    const QState s = unpack72(rec72);
    QState t;
    switch (board_size) {
        case 3: t = next_state<3>(s, action); break;
        case 5: t = next_state<5>(s, action); break;
        case 7: t = next_state<7>(s, action); break;
        case 9: t = next_state<9>(s, action); break;
        default: return -1;
    }
    pack72(t, board_size, out72);
    return 0;
}

int aqg_host_shortest_path(int board_size, const uint8_t* rec72) {
    if (!rec72) return -2;
    const QState s = unpack72(rec72);
    switch (board_size) {
        case 3: return shortest_path<3>(s);
        case 5: return shortest_path<5>(s);
        case 7: return shortest_path<7>(s);
        case 9: return shortest_path<9>(s);
        default: return -2;
    }
}

double aqg_host_heuristic_eval(int board_size, const uint8_t* rec72, int max_dist_from_goal) {
    if (!rec72 || max_dist_from_goal == 0) return 0.0;
    const QState s = unpack72(rec72);
    switch (board_size) {
        case 3: return heuristic<3>(s, max_dist_from_goal);
        case 5: return heuristic<5>(s, max_dist_from_goal);
        case 7: return heuristic<7>(s, max_dist_from_goal);
        case 9: return heuristic<9>(s, max_dist_from_goal);
        default: return std::numeric_limits<double>::quiet_NaN();      // unsupported board size: an error value, like the siblings' -1 / -2
    }
}

int aqg_host_alpha_beta_action(int board_size, const uint8_t* rec72, int plies_for_draw, int max_dist_from_goal, int max_depth) {
    if (!rec72 || max_dist_from_goal == 0 || max_depth < 0) return -1;
    const QState s = unpack72(rec72);
    AQG_HOST_DISPATCH(board_size, alpha_beta_action<3>(s, plies_for_draw, max_dist_from_goal, max_depth),
                      alpha_beta_action<5>(s, plies_for_draw, max_dist_from_goal, max_depth),
                      alpha_beta_action<7>(s, plies_for_draw, max_dist_from_goal, max_depth),
                      alpha_beta_action<9>(s, plies_for_draw, max_dist_from_goal, max_depth))
}

}  // extern "C"
