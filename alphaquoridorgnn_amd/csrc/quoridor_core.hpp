// quoridor_core.hpp -- bitboard Quoridor rules, shared by every HIP kernel (and by the host-side
// self-check build under tests/hostcheck, so the logic tested on CPU is the logic run on gfx950).
//
// Contract reproduced (bit-exact): /root/reference/game_logic.py
//   State fields            :25-40     is_lose/is_draw :43-54    legal_actions      :103-117
//   legal_actions_pos       :120-192   legal_actions_wall :195-357 (can_place_wall :199-223,
//   is_goal_possibly_blocked :227-307, bfs :309-324, can_reach_goal :325-348)
//   rotate_walls/next       :359-391
//
// Formulation (NOT the reference's): tiles are an N*N-bit board (two u64), wall slots a 64-bit
// mask per orientation.  "Blocked" masks per direction are built once per state by spreading the
// slot masks (stride N-1) onto the tile grid (stride N); reachability is an iterated 4-direction
// shift-and-mask flood fill.  The reference's BFS expands with legal_actions_pos, i.e. the other
// pawn is an obstacle that can only be jumped (game_logic.py:174-188, :319); that is folded in as
// (a) the obstacle tile is never entered, (b) for each of its <=4 neighbours p with an open edge
// p->obstacle, reaching p also reaches the jump targets J_d.  The enemy's search runs in the
// mover's frame (start/goal/obstacle rotated by 180 degrees instead of rotating the walls,
// which is reachability-equivalent to game_logic.py:344-345).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define QHD __host__ __device__ __forceinline__
#else
#define QHD inline
#endif

namespace aqg {

// ---------------------------------------------------------------------------------------------
// state72: the byte record at the C-ABI boundary == State.to_array() flattened + plies + N.
//   [0] player pos [1] player walls [2] enemy pos [3] enemy walls [4..67] walls[64]
//   [68..69] plies (u16 LE) [70] N [71] 0
// QState: the 24-byte register/HBM form used inside kernels and tree pools.
// ---------------------------------------------------------------------------------------------
constexpr int STATE72 = 72;
constexpr int MAX_LEGAL = 136;  // >= 5 pawn moves + 128 walls, multiple of 8

struct QState {
    uint64_t hw;     // bit i: walls[i] == 1 (horizontal), mover's frame
    uint64_t vw;     // bit i: walls[i] == 2 (vertical)
    uint8_t ppos;    // player[0], player's own frame
    uint8_t pwl;     // player[1]
    uint8_t epos;    // enemy[0], ENEMY's own frame (game_logic.py:20-21)
    uint8_t ewl;     // enemy[1]
    uint16_t plies;  // plies_played
    uint16_t pad;
};
static_assert(sizeof(QState) == 24, "QState must be 24 bytes");

// 4 wall bytes (one u32) -> 4 H bits and 4 V bits: bit0 / bit1 of every byte gathered by a multiply
QHD uint32_t gather_bit0_x4(uint32_t x) { return (((x & 0x01010101u) * 0x01020408u) >> 24) & 0xFu; }

QHD QState unpack72(const uint8_t* r) {
    // 18 aligned dword loads instead of 72 byte loads (records are 72-byte strided, 4-byte aligned)
    const uint32_t* w = reinterpret_cast<const uint32_t*>(r);
    QState s;
    uint64_t h = 0, v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 16; ++i) {
        const uint32_t x = w[1 + i];
        h |= (uint64_t)gather_bit0_x4(x) << (4 * i);
        v |= (uint64_t)gather_bit0_x4(x >> 1) << (4 * i);
    }
    s.hw = h; s.vw = v;
    const uint32_t hd = w[0], tl = w[17];
    s.ppos = (uint8_t)(hd & 0xff); s.pwl = (uint8_t)((hd >> 8) & 0xff);
    s.epos = (uint8_t)((hd >> 16) & 0xff); s.ewl = (uint8_t)(hd >> 24);
    s.plies = (uint16_t)(tl & 0xffff);
    s.pad = 0;
    return s;
}

QHD void pack72(const QState& s, int N, uint8_t* r) {
    r[0] = s.ppos; r[1] = s.pwl; r[2] = s.epos; r[3] = s.ewl;
    for (int i = 0; i < 64; ++i) r[4 + i] = (uint8_t)(((s.hw >> i) & 1) | (((s.vw >> i) & 1) << 1));
    r[68] = (uint8_t)(s.plies & 0xff); r[69] = (uint8_t)(s.plies >> 8);
    r[70] = (uint8_t)N; r[71] = 0;
}

QHD uint64_t brev64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
    x = ((x >> 8) & 0x00ff00ff00ff00ffull) | ((x & 0x00ff00ff00ff00ffull) << 8);
    x = ((x >> 16) & 0x0000ffff0000ffffull) | ((x & 0x0000ffff0000ffffull) << 16);
    return (x >> 32) | (x << 32);
#endif
}

// ---------------------------------------------------------------------------------------------
// 128-bit tile board
// ---------------------------------------------------------------------------------------------
struct BB {
    uint64_t lo, hi;
};
QHD BB bb(uint64_t lo, uint64_t hi) { BB r; r.lo = lo; r.hi = hi; return r; }
QHD BB bb_bit(int t) { return t < 64 ? bb(1ull << t, 0) : bb(0, 1ull << (t - 64)); }
QHD BB operator|(BB a, BB b) { return bb(a.lo | b.lo, a.hi | b.hi); }
QHD BB operator&(BB a, BB b) { return bb(a.lo & b.lo, a.hi & b.hi); }
QHD BB operator~(BB a) { return bb(~a.lo, ~a.hi); }
QHD bool bb_any(BB a) { return (a.lo | a.hi) != 0; }
QHD bool bb_eq(BB a, BB b) { return a.lo == b.lo && a.hi == b.hi; }
QHD bool bb_test(BB a, int t) { return t < 64 ? ((a.lo >> t) & 1) : ((a.hi >> (t - 64)) & 1); }
template <int K> QHD BB bb_shl(BB a) {  // 0 < K < 64
    return bb(a.lo << K, (a.hi << K) | (a.lo >> (64 - K)));
}
template <int K> QHD BB bb_shr(BB a) {
    return bb((a.lo >> K) | (a.hi << (64 - K)), a.hi >> K);
}

template <int N> struct Geo {
    static constexpr int V = N * N;          // tiles
    static constexpr int S = N - 1;          // wall slots per row
    static constexpr int NW = S * S;         // wall slots
    static constexpr int A = V + 2 * NW;     // actions (209 at 9x9)
    static_assert(N % 2 == 1 && N >= 3 && N <= 9, "odd board size 3..9 (game_logic.py:28-29)");
};

// constant tile masks ------------------------------------------------------------------------
template <int N> QHD BB mask_all() {
    constexpr int V = N * N;
    if constexpr (V > 64) return bb(~0ull, (1ull << (V - 64)) - 1);
    else return bb((1ull << V) - 1, 0);
}
template <int N> QHD BB mask_row(int x) {  // all tiles of row x
    BB r = bb(0, 0);
    for (int y = 0; y < N; ++y) r = r | bb_bit(x * N + y);
    return r;
}
template <int N> QHD BB mask_col(int y) {
    BB r = bb(0, 0);
    for (int x = 0; x < N; ++x) r = r | bb_bit(x * N + y);
    return r;
}

// per-direction "a move from this tile in direction d stays in-board and crosses no wall"
struct Open {
    BB U, D, L, R;
};

// spread a 64-bit slot mask (row stride S) onto the tile grid (row stride N): slot (sx,sy) -> tile (sx,sy)
template <int N> QHD BB spread_slots(uint64_t m) {
    constexpr int S = N - 1;
    BB r = bb(0, 0);
#pragma unroll
    for (int sx = 0; sx < S; ++sx) {
        uint64_t row = (m >> (S * sx)) & ((1ull << S) - 1);
        int off = N * sx;
        if (off < 64) {
            r.lo |= row << off;
            if (off + S > 64) r.hi |= row >> (64 - off);
        } else {
            r.hi |= row << (off - 64);
        }
    }
    return r;
}

// game_logic.py:145-167 (is_wall_blocking) for every tile at once.
//   H wall at slot (sx,sy): blocks DOWN from tiles (sx,sy),(sx,sy+1) and UP from (sx+1,sy),(sx+1,sy+1)
//   V wall at slot (sx,sy): blocks RIGHT from (sx,sy),(sx+1,sy) and LEFT from (sx,sy+1),(sx+1,sy+1)
template <int N> QHD Open make_open(uint64_t hw, uint64_t vw) {
    BB all = mask_all<N>();
    BB sh = spread_slots<N>(hw), sv = spread_slots<N>(vw);
    BB blockD = sh | bb_shl<1>(sh);
    BB blockU = bb_shl<N>(blockD);
    BB blockR = sv | bb_shl<N>(sv);
    BB blockL = bb_shl<1>(blockR);
    Open o;
    o.D = all & ~mask_row<N>(N - 1) & ~blockD;
    o.U = all & ~mask_row<N>(0) & ~blockU;
    o.R = all & ~mask_col<N>(N - 1) & ~blockR;
    o.L = all & ~mask_col<N>(0) & ~blockL;
    return o;
}

// The same predicate for ONE tile straight from the slot masks (few live registers; used by the GNN trunk
// to derive node degrees).  bit0 = U, bit1 = D, bit2 = L, bit3 = R open.  game_logic.py:145-167.
template <int N> QHD int tile_open_bits(uint64_t hw, uint64_t vw, int t) {
    constexpr int S = N - 1;
    const int x = t / N, y = t % N;
    auto H = [&](int sx, int sy) -> bool { return (hw >> (sx * S + sy)) & 1; };
    auto Vw = [&](int sx, int sy) -> bool { return (vw >> (sx * S + sy)) & 1; };
    const bool u = x > 0 && !((y < S && H(x - 1, y)) || (y > 0 && H(x - 1, y - 1)));
    const bool d = x < N - 1 && !((y < S && H(x, y)) || (y > 0 && H(x, y - 1)));
    const bool l = y > 0 && !((x < S && Vw(x, y - 1)) || (x > 0 && Vw(x - 1, y - 1)));
    const bool r = y < N - 1 && !((x < S && Vw(x, y)) || (x > 0 && Vw(x - 1, y)));
    return (int)u | ((int)d << 1) | ((int)l << 2) | ((int)r << 3);
}

// add one candidate wall to the open masks (orientation 1 = H, 2 = V; slot index i)
template <int N> QHD Open add_wall(Open o, int orient, int slot) {
    constexpr int S = N - 1;
    int t = slot + slot / S;  // top-left tile of the 2x2 block: (slot/S)*N + slot%S
    if (orient == 1) {
        o.D = o.D & ~(bb_bit(t) | bb_bit(t + 1));
        o.U = o.U & ~(bb_bit(t + N) | bb_bit(t + N + 1));
    } else {
        o.R = o.R & ~(bb_bit(t) | bb_bit(t + N));
        o.L = o.L & ~(bb_bit(t + 1) | bb_bit(t + N + 1));
    }
    return o;
}

// Jump rule of game_logic.py:174-188 as "extra edges": for the direction d in which a pawn would step
// ONTO the obstacle, the set of tiles it lands on instead.
struct Jumps {
    int p[4];   // tile from which direction d (0=U,1=D,2=L,3=R) leads onto the obstacle, or -1
    BB J[4];    // landing tiles for that direction
};

template <int N> QHD Jumps make_jumps(const Open& o, int e) {
    Jumps j;
    const int ex = e / N, ey = e % N;
    // d = U: mover below the obstacle (p = e + N) moving up
    j.p[0] = (ex + 1 < N && bb_test(o.U, e + N)) ? e + N : -1;
    j.J[0] = bb_test(o.U, e) ? bb_bit(e - N)
                             : ((bb_test(o.L, e) ? bb_bit(e - 1) : bb(0, 0)) | (bb_test(o.R, e) ? bb_bit(e + 1) : bb(0, 0)));
    // d = D: mover above (p = e - N) moving down
    j.p[1] = (ex - 1 >= 0 && bb_test(o.D, e - N)) ? e - N : -1;
    j.J[1] = bb_test(o.D, e) ? bb_bit(e + N)
                             : ((bb_test(o.L, e) ? bb_bit(e - 1) : bb(0, 0)) | (bb_test(o.R, e) ? bb_bit(e + 1) : bb(0, 0)));
    // d = L: mover to the right (p = e + 1) moving left
    j.p[2] = (ey + 1 < N && bb_test(o.L, e + 1)) ? e + 1 : -1;
    j.J[2] = bb_test(o.L, e) ? bb_bit(e - 1)
                             : ((bb_test(o.U, e) ? bb_bit(e - N) : bb(0, 0)) | (bb_test(o.D, e) ? bb_bit(e + N) : bb(0, 0)));
    // d = R: mover to the left (p = e - 1) moving right
    j.p[3] = (ey - 1 >= 0 && bb_test(o.R, e - 1)) ? e - 1 : -1;
    j.J[3] = bb_test(o.R, e) ? bb_bit(e + 1)
                             : ((bb_test(o.U, e) ? bb_bit(e - N) : bb(0, 0)) | (bb_test(o.D, e) ? bb_bit(e + N) : bb(0, 0)));
    return j;
}

// game_logic.py:309-324 (bfs) as a flood fill: can a pawn starting on `start` reach any tile of `goal`
// when the other pawn stands on `obst`?
template <int N> QHD bool can_reach(const Open& o, int start, int obst, BB goal) {
    const Jumps j = make_jumps<N>(o, obst);
    const BB notobst = ~bb_bit(obst);
    BB reach = bb_bit(start);
    for (int it = 0; it < N * N; ++it) {
        if (bb_any(reach & goal)) return true;
        BB nr = reach | bb_shl<N>(reach & o.D) | bb_shr<N>(reach & o.U) | bb_shl<1>(reach & o.R) | bb_shr<1>(reach & o.L);
        nr = nr & notobst;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (j.p[d] >= 0 && bb_test(reach, j.p[d])) nr = nr | j.J[d];
        if (bb_eq(nr, reach)) return false;
        reach = nr;
    }
    return bb_any(reach & goal);
}

// The two searches a wall candidate needs (mover -> its goal row, enemy -> its goal row) advanced in lock step.
// Same decisions as two can_reach() calls -- each search is decided at the iteration at which it would be alone --
// but the two dependency chains are independent, so a wavefront that has its SIMD to itself issues them
// back to back instead of waiting out each instruction's latency.  Returns bit0 = A reaches, bit1 = B reaches.
template <int N> QHD int can_reach2(const Open& o, int sA, int obA, BB goalA, int sB, int obB, BB goalB) {
    const Jumps jA = make_jumps<N>(o, obA), jB = make_jumps<N>(o, obB);
    const BB nA = ~bb_bit(obA), nB = ~bb_bit(obB);
    BB rA = bb_bit(sA), rB = bb_bit(sB);
    int res = 0, done = 0;
    for (int it = 0; it < N * N; ++it) {
        if (!(done & 1) && bb_any(rA & goalA)) { res |= 1; done |= 1; }
        if (!(done & 2) && bb_any(rB & goalB)) { res |= 2; done |= 2; }
        if (done == 3) break;
        BB a = rA | bb_shl<N>(rA & o.D) | bb_shr<N>(rA & o.U) | bb_shl<1>(rA & o.R) | bb_shr<1>(rA & o.L);
        BB b = rB | bb_shl<N>(rB & o.D) | bb_shr<N>(rB & o.U) | bb_shl<1>(rB & o.R) | bb_shr<1>(rB & o.L);
        a = a & nA; b = b & nB;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (jA.p[d] >= 0 && bb_test(rA, jA.p[d])) a = a | jA.J[d];
            if (jB.p[d] >= 0 && bb_test(rB, jB.p[d])) b = b | jB.J[d];
        }
        if (bb_eq(a, rA)) done |= 1;                       // fixpoint without the goal: this search has failed
        if (bb_eq(b, rB)) done |= 2;
        if (done == 3) break;
        rA = a; rB = b;
    }
    if (!(done & 1) && bb_any(rA & goalA)) res |= 1;
    if (!(done & 2) && bb_any(rB & goalB)) res |= 2;
    return res;
}

// ---------------------------------------------------------------------------------------------
// The same pair of searches on three 32-bit words per board (V <= 81 < 96 tiles) -- what the GPU runs.  One wavefront per
// state means one LANE per candidate wall, alone on its SIMD: the search is a chain of dependent vector instructions and
// its length is the latency of legal_actions().  Against the two-u64 form above: a shift by N is three v_alignbit_b32
// instead of two 64-bit shifts + OR per half (64-bit shifts are slow), the jump landings are OR-ed in under an
// all-ones / zero mask taken from the source tile's bit (no branches: the u64 form spends 60 scalar instructions per
// round on exec-mask bookkeeping), and a round is ~115 vector instructions instead of ~235.  Same decisions as two
// can_reach() calls (checked against it on every golden fixture by the host build, like can_reach2).
// ---------------------------------------------------------------------------------------------
struct W3 {
    uint32_t a, b, c;        // tiles 0..31, 32..63, 64..95
};
QHD W3 w3(const BB& x) { W3 r; r.a = (uint32_t)x.lo; r.b = (uint32_t)(x.lo >> 32); r.c = (uint32_t)x.hi; return r; }
QHD W3 w3_bit(int t) { W3 r; r.a = t < 32 ? 1u << t : 0u; r.b = (t >= 32 && t < 64) ? 1u << (t - 32) : 0u; r.c = t >= 64 ? 1u << (t - 64) : 0u; return r; }
QHD W3 operator&(W3 x, W3 y) { W3 r; r.a = x.a & y.a; r.b = x.b & y.b; r.c = x.c & y.c; return r; }
QHD W3 operator|(W3 x, W3 y) { W3 r; r.a = x.a | y.a; r.b = x.b | y.b; r.c = x.c | y.c; return r; }
QHD W3 operator~(W3 x) { W3 r; r.a = ~x.a; r.b = ~x.b; r.c = ~x.c; return r; }
QHD bool w3_any(W3 x) { return (x.a | x.b | x.c) != 0; }
QHD bool w3_eq(W3 x, W3 y) { return ((x.a ^ y.a) | (x.b ^ y.b) | (x.c ^ y.c)) == 0; }
template <int K> QHD W3 w3_shl(W3 x) {   // 0 < K < 32
    W3 r; r.a = x.a << K; r.b = (x.b << K) | (x.a >> (32 - K)); r.c = (x.c << K) | (x.b >> (32 - K)); return r;
}
template <int K> QHD W3 w3_shr(W3 x) {
    W3 r; r.a = (x.a >> K) | (x.b << (32 - K)); r.b = (x.b >> K) | (x.c << (32 - K)); r.c = x.c >> K; return r;
}
// all-ones if tile t is set in x, else zero
QHD uint32_t w3_mask_of(W3 x, int t) {          // 0 <= t < 96
    const uint32_t w = t < 32 ? x.a : (t < 64 ? x.b : x.c);
    return 0u - ((w >> (t & 31)) & 1u);
}

// landing masks of the jump rule (game_logic.py:174-188) around obstacle e, branch-free: for direction d (0=U,1=D,2=L,3=R)
// the tiles a pawn stepping onto e lands on instead, already AND-ed with "the source tile exists and its edge to e is open".
// e is the same in every lane on the GPU (scalar tile masks); the open-edge boards differ per candidate wall.
template <int N> QHD void jump_landings_w3(const W3& oU, const W3& oD, const W3& oL, const W3& oR, int e, W3 (&J)[4], int (&src)[4]) {
    const int ex = e / N, ey = e % N;
    auto clampt = [](int t) { return t < 0 ? 0 : (t > 95 ? 95 : t); };
    const uint32_t tU = w3_mask_of(oU, e), tD = w3_mask_of(oD, e), tL = w3_mask_of(oL, e), tR = w3_mask_of(oR, e);
    // (an open edge at e guarantees that the neighbour exists, so the neighbour masks need no range checks of their own)
    const W3 bU = w3_bit(clampt(e - N)), bD = w3_bit(clampt(e + N)), bL = w3_bit(clampt(e - 1)), bR = w3_bit(clampt(e + 1));
    auto sel = [](uint32_t m, const W3& x) { W3 r; r.a = x.a & m; r.b = x.b & m; r.c = x.c & m; return r; };
    const W3 sideLR = sel(tL, bL) | sel(tR, bR), sideUD = sel(tU, bU) | sel(tD, bD);
    const W3 jU = sel(tU, bU) | sel(~tU, sideLR), jD = sel(tD, bD) | sel(~tD, sideLR);
    const W3 jL = sel(tL, bL) | sel(~tL, sideUD), jR = sel(tR, bR) | sel(~tR, sideUD);
    src[0] = clampt(e + N); src[1] = clampt(e - N); src[2] = clampt(e + 1); src[3] = clampt(e - 1);
    const uint32_t vU = (ex + 1 < N) ? w3_mask_of(oU, src[0]) : 0u;      // mover below the obstacle, moving up
    const uint32_t vD = (ex - 1 >= 0) ? w3_mask_of(oD, src[1]) : 0u;     // mover above, moving down
    const uint32_t vL = (ey + 1 < N) ? w3_mask_of(oL, src[2]) : 0u;      // mover to the right, moving left
    const uint32_t vR = (ey - 1 >= 0) ? w3_mask_of(oR, src[3]) : 0u;     // mover to the left, moving right
    J[0] = sel(vU, jU); J[1] = sel(vD, jD); J[2] = sel(vL, jL); J[3] = sel(vR, jR);
}

// `base` = open-edge boards of the position (wave-uniform on the GPU), plus ONE candidate wall (orient 1 = H / 2 = V at `slot`,
// per lane; orient 0 = none).  Returns bit0 = A reaches its goal, bit1 = B reaches its goal.
template <int N> QHD int can_reach2_w3(const Open& base, int orient, int slot, int sA, int obA, BB goalA, int sB, int obB, BB goalB) {
    constexpr int S = N - 1;
    W3 oU = w3(base.U), oD = w3(base.D), oL = w3(base.L), oR = w3(base.R);
    {   // add_wall, branch-free: the wall's two blocked edges, both directions
        const int t = slot + slot / S;                                  // top-left tile of the 2x2 block
        const uint32_t mh = orient == 1 ? ~0u : 0u, mv = orient == 2 ? ~0u : 0u;
        const W3 b0 = w3_bit(t), b1 = w3_bit(t + 1), bN = w3_bit(t + N), bN1 = w3_bit(t + N + 1);
        auto clear = [](W3& x, const W3& m, uint32_t on) { x.a &= ~(m.a & on); x.b &= ~(m.b & on); x.c &= ~(m.c & on); };
        clear(oD, b0 | b1, mh); clear(oU, bN | bN1, mh);
        clear(oR, b0 | bN, mv); clear(oL, b1 | bN1, mv);
    }
    const W3 nA = ~w3_bit(obA), nB = ~w3_bit(obB);
    const W3 gA = w3(goalA), gB = w3(goalB);
    W3 JA[4], JB[4];
    int qA[4], qB[4];
    jump_landings_w3<N>(oU, oD, oL, oR, obA, JA, qA);
    jump_landings_w3<N>(oU, oD, oL, oR, obB, JB, qB);
    W3 rA = w3_bit(sA), rB = w3_bit(sB);
    int res = 0, done = 0;
    for (int it = 0; it < N * N; ++it) {
        if (!(done & 1) && w3_any(rA & gA)) { res |= 1; done |= 1; }
        if (!(done & 2) && w3_any(rB & gB)) { res |= 2; done |= 2; }
        if (done == 3) break;
        W3 a = (rA | w3_shl<N>(rA & oD) | w3_shr<N>(rA & oU) | w3_shl<1>(rA & oR) | w3_shr<1>(rA & oL)) & nA;
        W3 b = (rB | w3_shl<N>(rB & oD) | w3_shr<N>(rB & oU) | w3_shl<1>(rB & oR) | w3_shr<1>(rB & oL)) & nB;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int d = 0; d < 4; ++d) {
            const uint32_t mA = w3_mask_of(rA, qA[d]), mB = w3_mask_of(rB, qB[d]);
            a.a |= JA[d].a & mA; a.b |= JA[d].b & mA; a.c |= JA[d].c & mA;
            b.a |= JB[d].a & mB; b.b |= JB[d].b & mB; b.c |= JB[d].c & mB;
        }
        if (w3_eq(a, rA)) done |= 1;                       // fixpoint without the goal: this search has failed
        if (w3_eq(b, rB)) done |= 2;
        if (done == 3) break;
        rA = a; rB = b;
    }
    if (!(done & 1) && w3_any(rA & gA)) res |= 1;
    if (!(done & 2) && w3_any(rB & gB)) res |= 2;
    return res;
}

// game_logic.py:120-192 legal_actions_pos(pos) with the enemy pawn on tile `e` (mover's frame).
// Ordered: U, D, L, R; a jump contributes the straight landing, else (left,right) / (up,down).
template <int N> QHD int legal_pos_list(const Open& o, int pos, int e, uint8_t* out) {
    int c = 0;
    const int dstep[4] = {-N, N, -1, 1};
    const BB* od[4] = {&o.U, &o.D, &o.L, &o.R};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (!bb_test(*od[d], pos)) continue;
        int n = pos + dstep[d];
        if (n != e) { out[c++] = (uint8_t)n; continue; }
        if (bb_test(*od[d], n)) { out[c++] = (uint8_t)(n + dstep[d]); continue; }
        if (d < 2) {
            if (bb_test(o.L, n)) out[c++] = (uint8_t)(n - 1);
            if (bb_test(o.R, n)) out[c++] = (uint8_t)(n + 1);
        } else {
            if (bb_test(o.U, n)) out[c++] = (uint8_t)(n - N);
            if (bb_test(o.D, n)) out[c++] = (uint8_t)(n + N);
        }
    }
    return c;
}

// game_logic.py:199-223 can_place_wall, for all slots at once -> bit i set = placement geometrically allowed
template <int N> QHD void placeable_masks(uint64_t hw, uint64_t vw, uint64_t& hplace, uint64_t& vplace) {
    constexpr int S = N - 1;
    constexpr int NW = S * S;
    const uint64_t ALL = NW == 64 ? ~0ull : ((1ull << (NW % 64)) - 1);
    uint64_t col0 = 0, colL = 0;
    for (int sx = 0; sx < S; ++sx) { col0 |= 1ull << (sx * S); colL |= 1ull << (sx * S + S - 1); }
    uint64_t occ = hw | vw;
    uint64_t hleft = (hw << 1) & ~col0;    // slot i has an H wall at i-1 in the same row
    uint64_t hright = (hw >> 1) & ~colL;   // ... at i+1
    hplace = ALL & ~occ & ~hleft & ~hright;
    uint64_t vup = vw << S, vdown = vw >> S;
    vplace = ALL & ~occ & ~vup & ~vdown;
}

// game_logic.py:227-307 is_goal_possibly_blocked(orientation, pos): >= 2 of {end A, middle, end B} touched
template <int N> QHD bool possibly_blocking(uint64_t hw, uint64_t vw, int orient, int pos) {
    constexpr int S = N - 1;
    const int x = pos / S, y = pos % S;
    auto H = [&](int i) -> bool { return (hw >> i) & 1; };
    auto Vv = [&](int i) -> bool { return (vw >> i) & 1; };
    int cnt;
    if (orient == 1) {
        bool left = (y == 0) ||
                    (y > 0 && (Vv(pos - 1) || (x > 0 && Vv(pos - S - 1)) || (x < S - 1 && Vv(pos + S - 1)))) ||
                    (y > 1 && H(pos - 2));
        bool mid = (x > 0 && Vv(pos - S)) || (x < S - 1 && Vv(pos + S));
        bool right = (y == S - 1) ||
                     (y < S - 1 && (Vv(pos + 1) || (x > 0 && Vv(pos - S + 1)) || (x < S - 1 && Vv(pos + S + 1)))) ||
                     (y < S - 2 && H(pos + 2));
        cnt = (int)left + (int)mid + (int)right;
    } else {
        bool top = (x == 0) ||
                   (x > 0 && (H(pos - S) || (y > 0 && H(pos - S - 1)) || (y < S - 1 && H(pos - S + 1)))) ||
                   (x > 1 && Vv(pos - 2 * S));
        bool mid = (y > 0 && H(pos - 1)) || (y < S - 1 && H(pos + 1));
        bool bot = (x == S - 1) ||
                   (x < S - 1 && (H(pos + S) || (y > 0 && H(pos + S - 1)) || (y < S - 1 && H(pos + S + 1)))) ||
                   (x < S - 2 && Vv(pos + 2 * S));
        cnt = (int)top + (int)mid + (int)bot;
    }
    return cnt >= 2;
}

// The same prefilter for ALL slots at once as two 64-bit masks (bit i = slot i is "possibly blocking" as an H / V
// candidate): shifts of the two wall masks with column masks, ~60 scalar operations per state instead of ~100 vector
// operations per lane.  S x S slots, bit = x * S + y (row x, column y); rows never wrap because every shifted term that
// could cross a row end is masked by the column it must not come from.
template <int N> QHD void possibly_blocking_masks(uint64_t hw, uint64_t vw, uint64_t& hmask, uint64_t& vmask) {
    constexpr int S = N - 1, NW = S * S;
    const uint64_t ALL = NW == 64 ? ~0ull : ((1ull << (NW % 64)) - 1);
    uint64_t col0 = 0, col1 = 0, colL = 0, colL1 = 0, row0 = 0, rowL = 0;
    for (int i = 0; i < S; ++i) {
        col0 |= 1ull << (i * S); colL |= 1ull << (i * S + S - 1);
        if (S > 1) { col1 |= 1ull << (i * S + 1); colL1 |= 1ull << (i * S + S - 2); }
        row0 |= 1ull << i; rowL |= 1ull << ((S - 1) * S + i);
    }
    // H candidate at (x, y): left end, middle, right end (game_logic.py:251-274)
    const uint64_t hl = col0 | (((vw << 1) | (vw << (S + 1)) | (vw >> (S - 1))) & ~col0) | ((hw << 2) & ~col0 & ~col1);
    const uint64_t hm = (vw << S) | (vw >> S);
    const uint64_t hr = colL | (((vw >> 1) | (vw << (S - 1)) | (vw >> (S + 1))) & ~colL) | ((hw >> 2) & ~colL & ~colL1);
    hmask = ((hl & hm) | (hl & hr) | (hm & hr)) & ALL;
    // V candidate: top end, middle, bottom end (game_logic.py:281-304)
    const uint64_t vt = row0 | (hw << S) | ((hw << (S + 1)) & ~col0) | ((hw << (S - 1)) & ~colL) | (vw << (2 * S));
    const uint64_t vm = ((hw << 1) & ~col0) | ((hw >> 1) & ~colL);
    const uint64_t vb = rowL | (hw >> S) | ((hw >> (S - 1)) & ~col0) | ((hw >> (S + 1)) & ~colL) | (vw >> (2 * S));
    vmask = ((vt & vm) | (vt & vb) | (vm & vb)) & ALL;
}

// game_logic.py:325-348 can_reach_goal for a geometrically placeable candidate
template <int N> QHD bool wall_keeps_paths(const QState& s, const Open& base, int orient, int pos) {
    const bool pb = possibly_blocking<N>(s.hw, s.vw, orient, pos);
    {   // the all-slots mask form the GPU wavefront uses must agree (host tests run this against every fixture)
        uint64_t hm, vm;
        possibly_blocking_masks<N>(s.hw, s.vw, hm, vm);
        if ((((orient == 1 ? hm : vm) >> pos) & 1) != (pb ? 1u : 0u)) return pb;   // a disagreement flips the answer below
    }
    if (!pb) return true;                                              // :327-328 prefilter
    constexpr int V = N * N;
    const Open o = add_wall<N>(base, orient, pos);
    const int me = s.ppos, other = V - 1 - s.epos;                     // :136 enemy in mover's frame
    const bool rp = can_reach<N>(o, me, other, mask_row<N>(0));        // :335
    const bool re = can_reach<N>(o, other, me, mask_row<N>(N - 1));    // :344-345, un-rotated
    // the lock-step form the GPU wavefront uses must decide exactly the same (host tests run this against every fixture)
    const int both = can_reach2<N>(o, me, other, mask_row<N>(0), other, me, mask_row<N>(N - 1));
    if (both != ((rp ? 1 : 0) | (re ? 2 : 0))) return !(rp && re);    // a disagreement flips the answer and fails the golden tests
    const int both3 = can_reach2_w3<N>(base, orient, pos, me, other, mask_row<N>(0), other, me, mask_row<N>(N - 1));   // the three-word form the GPU runs
    if (both3 != both) return !(rp && re);
    return rp && re;
}

// game_logic.py:366-391 next(action): apply, rotate walls by 180 degrees (= bit reversal of the slot masks), swap
template <int N> QHD QState next_state(const QState& s, int action) {
    constexpr int V = N * N, NW = (N - 1) * (N - 1);
    QState t = s;
    if (action < V) t.ppos = (uint8_t)action;
    else if (action < V + NW) { t.hw |= 1ull << (action - V); t.pwl -= 1; }
    else { t.vw |= 1ull << (action - V - NW); t.pwl -= 1; }
    QState r;
    r.hw = brev64(t.hw) >> (64 - NW);
    r.vw = brev64(t.vw) >> (64 - NW);
    r.ppos = t.epos; r.pwl = t.ewl;
    r.epos = t.ppos; r.ewl = t.pwl;
    r.plies = (uint16_t)(s.plies + 1);
    r.pad = 0;
    return r;
}

template <int N> QHD bool is_lose(const QState& s) { return s.epos / N == 0; }          // :43-46
QHD bool is_draw(const QState& s, int plies_for_draw) { return s.plies >= plies_for_draw; }  // :49-50

// Serial (one thread per state) legal_actions() -- used by the MCTS kernels for small jobs and by the
// host self-check.  Returns the count; `out` receives the reference-ordered action ids (:111-115, :352-355).
template <int N> QHD int legal_actions_serial(const QState& s, uint8_t* out) {
    constexpr int V = N * N, NW = (N - 1) * (N - 1);
    const Open base = make_open<N>(s.hw, s.vw);
    int c = legal_pos_list<N>(base, s.ppos, V - 1 - s.epos, out);
    if (s.pwl > 0) {
        uint64_t hp, vp;
        placeable_masks<N>(s.hw, s.vw, hp, vp);
        for (int pos = 0; pos < NW; ++pos) {
            if (((hp >> pos) & 1) && wall_keeps_paths<N>(s, base, 1, pos)) out[c++] = (uint8_t)(V + pos);
            if (((vp >> pos) & 1) && wall_keeps_paths<N>(s, base, 2, pos)) out[c++] = (uint8_t)(V + NW + pos);
        }
    }
    return c;
}

}  // namespace aqg
