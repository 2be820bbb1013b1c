// legal_wave.hpp -- State.legal_actions() (game_logic.py:103-117) computed by ONE 64-lane wavefront for one state.
// Shared by the batched legal_actions kernel and the fused MCTS step kernel.
//
// One lane per wall slot (64 slots at 9x9).  Each lane tests both orientations of its slot: geometric placement
// (bit ops on the slot masks), the reference's touch-count prefilter, and -- only when that says "possibly
// blocking" -- the two jump-aware flood fills.  The reference's list order (pawn moves, then H,V interleaved per
// slot) is rebuilt with two wave ballots and popcounts: no atomics, no sorting.
#pragma once
#include "quoridor_core.hpp"

namespace aqg {

// mask: [A] u8 or nullptr; order: [MAX_LEGAL] u8 or nullptr (entries >= count are 0xFF).  Returns the count
// (uniform across the wave).  All 64 lanes of the wave must call it.
template <int N>
__device__ __forceinline__ int wave_legal_actions(const QState& s, int lane, uint8_t* __restrict__ mask,
                                                  uint8_t* __restrict__ order) {
    constexpr int V = Geo<N>::V, NW = Geo<N>::NW, A = Geo<N>::A;
    const Open base = make_open<N>(s.hw, s.vw);
    bool legH = false, legV = false;
    if (s.pwl > 0 && lane < NW) {
        uint64_t hp, vp;
        placeable_masks<N>(s.hw, s.vw, hp, vp);
        if ((hp >> lane) & 1) legH = wall_keeps_paths<N>(s, base, 1, lane);
        if ((vp >> lane) & 1) legV = wall_keeps_paths<N>(s, base, 2, lane);
    }
    const uint64_t mH = __ballot(legH), mV = __ballot(legV);

    uint8_t pawn[8];
    const int npawn = legal_pos_list<N>(base, s.ppos, V - 1 - s.epos, pawn);
    const int total = npawn + __popcll(mH) + __popcll(mV);

    if (mask) {
        for (int a = lane; a < A; a += 64) {
            bool on;
            if (a < V) {
                on = false;
                for (int i = 0; i < npawn; ++i) on |= (pawn[i] == a);
            } else if (a < V + NW) on = (mH >> (a - V)) & 1;
            else on = (mV >> (a - V - NW)) & 1;
            mask[a] = on ? 1 : 0;
        }
    }
    if (order) {
        const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int prefix = npawn + __popcll(mH & below) + __popcll(mV & below);
        if (legH) order[prefix] = (uint8_t)(V + lane);
        if (legV) order[prefix + (legH ? 1 : 0)] = (uint8_t)(V + NW + lane);
        if (lane < npawn) order[lane] = pawn[lane];
        for (int i = total + lane; i < MAX_LEGAL; i += 64) order[i] = 0xFF;
    }
    return total;
}

}  // namespace aqg
