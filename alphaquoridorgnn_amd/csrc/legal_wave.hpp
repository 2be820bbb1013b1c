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

// Step 1 of legal_actions() for one state, separable from the searches: the open-edge boards and, per wall slot, geometric placement
// and the reference's touch-count prefilter -- wave-uniform mask algebra, no memory access (the MCTS step kernel runs it in the shadow of
// its evaluation-cache probe).  All 64 lanes of the wave must call it.
struct LegalPrep {
    Open base;
    uint64_t pH, pV, nH, nV;      // placeable H / V slots; those that also need the two path searches
    bool needH, needV;            // this lane's slot
};
template <int N>
__device__ __forceinline__ LegalPrep wave_legal_prepare(const QState& s, int lane) {
    constexpr int NW = Geo<N>::NW;
    LegalPrep p;
    // The wall masks are the same in every lane: as wave-uniform scalars, the open-edge bitboards, the placement masks
    // and the touch-count prefilter of ALL slots are computed once per wave on the scalar unit; lanes only test bits.
    const uint64_t hw = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(s.hw >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s.hw);
    const uint64_t vw = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(s.vw >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s.vw);
    p.base = make_open<N>(hw, vw);
    // Step 1 (lane = wall slot): geometric placement and the touch-count prefilter.  A placeable candidate that
    // the prefilter clears is legal outright (game_logic.py:327-328); the others need the two path searches.
    bool placeH = false, placeV = false;
    p.needH = false; p.needV = false;
    if (s.pwl > 0 && lane < NW) {
        uint64_t hp, vp, hb, vb;
        placeable_masks<N>(hw, vw, hp, vp);
        possibly_blocking_masks<N>(hw, vw, hb, vb);
        placeH = (hp >> lane) & 1;
        placeV = (vp >> lane) & 1;
        p.needH = placeH && ((hb >> lane) & 1);
        p.needV = placeV && ((vb >> lane) & 1);
    }
    p.pH = __ballot(placeH); p.pV = __ballot(placeV); p.nH = __ballot(p.needH); p.nV = __ballot(p.needV);
    return p;
}

// mask: [A] u8 or nullptr; order: [MAX_LEGAL] u8 or nullptr (entries >= count are 0xFF).  Returns the count
// (uniform across the wave).  All 64 lanes of the wave must call it.
template <int N>
__device__ __forceinline__ int wave_legal_finish(const QState& s, const LegalPrep& prep, int lane, uint8_t* __restrict__ mask,
                                                 uint8_t* __restrict__ order) {
    constexpr int V = Geo<N>::V, NW = Geo<N>::NW, A = Geo<N>::A;
    const Open& base = prep.base;
    const bool needH = prep.needH, needV = prep.needV;
    const uint64_t pH = prep.pH, pV = prep.pV, nH = prep.nH, nV = prep.nV;
    // Step 2 (lane = task): the k-th candidate that needs the searches (H candidates in slot order, then V) goes to
    // lane k, which runs the mover's and the enemy's flood fill interleaved (can_reach2: two independent dependency
    // chains keep a lone wavefront's VALU busy; one fill per lane and twice the rounds measured slower).
    const int cH = __popcll(nH), cV = __popcll(nV);
    const int ntask = cH + cV;
    // Results go back without any cross-lane reduction: the task of candidate (H, slot L) ran in lane rank = number of
    // needed H candidates below L (V candidates follow the H ones), so ONE ballot of "my task failed" per round of 64 tasks
    // is all the exchange there is -- lane L (= slot L) looks its own two bits up.  (Six rounds of four ds_bpermute each, a
    // dependent chain through the LDS crossbar, were a quarter of this function's latency.)
    bool myfailH = false, myfailV = false;
    const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int rankH = __popcll(nH & below), rankV = cH + __popcll(nV & below);
    for (int tbase = 0; tbase < ntask; tbase += 64) {
        const int task = tbase + lane;
        bool failed = false;
        if (task < ntask) {
            const int k = task;
            const int orient = k < cH ? 1 : 2;
            uint64_t m = orient == 1 ? nH : nV;
            int rank = orient == 1 ? k : k - cH;
            int slot = 0;                                   // position of the rank-th set bit of m
#pragma unroll
            for (int w = 32; w >= 1; w >>= 1) {
                const uint64_t lowmask = (1ull << w) - 1;
                const int c = __popcll((m >> slot) & lowmask);
                if (rank >= c) { rank -= c; slot += w; }
            }
            // (the pawns are the same in every lane: as scalars, the jump-source positions of the searches are scalars too)
            const int me = __builtin_amdgcn_readfirstlane((int)s.ppos), other = V - 1 - __builtin_amdgcn_readfirstlane((int)s.epos);
            const int ok = can_reach2_w3<N>(base, orient, slot, me, other, mask_row<N>(0), other, me, mask_row<N>(N - 1));
            failed = ok != 3;
        }
        const uint64_t fm = __ballot(failed);               // bit t = task tbase + t failed
        if (needH && rankH >= tbase && rankH < tbase + 64) myfailH = (fm >> (rankH - tbase)) & 1;
        if (needV && rankV >= tbase && rankV < tbase + 64) myfailV = (fm >> (rankV - tbase)) & 1;
    }
    const uint64_t failH = __ballot(myfailH), failV = __ballot(myfailV);
    const uint64_t mH = pH & ~failH, mV = pV & ~failV;
    const bool legH = (mH >> lane) & 1, legV = (mV >> lane) & 1;

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


template <int N>
__device__ __forceinline__ int wave_legal_actions(const QState& s, int lane, uint8_t* __restrict__ mask,
                                                  uint8_t* __restrict__ order) {
    const LegalPrep prep = wave_legal_prepare<N>(s, lane);
    return wave_legal_finish<N>(s, prep, lane, mask, order);
}

}  // namespace aqg
