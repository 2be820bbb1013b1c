// tests/hostcheck/hostcheck.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the SAME header the HIP kernels use (alphaquoridorgnn_amd/csrc/quoridor_core.hpp) for the
// host, so the bitboard rules can be checked against the oracle and the golden vectors in a container
// without a GPU.  Never imported by the package; the product path is the HIP library only.
#include "../../alphaquoridorgnn_amd/csrc/quoridor_core.hpp"
#include <cstring>
using namespace aqg;

template <int N>
static void legal_batch(const uint8_t* recs, int B, int16_t* out, int32_t* counts) {
    for (int b = 0; b < B; ++b) {
        QState s = unpack72(recs + (size_t)STATE72 * b);
        uint8_t acts[MAX_LEGAL];
        int c = legal_actions_serial<N>(s, acts);
        counts[b] = c;
        for (int i = 0; i < MAX_LEGAL; ++i) out[(size_t)b * MAX_LEGAL + i] = i < c ? (int16_t)acts[i] : (int16_t)-1;
    }
}
template <int N>
static void next_batch(const uint8_t* recs, const int32_t* actions, int B, uint8_t* out) {
    for (int b = 0; b < B; ++b) {
        QState s = unpack72(recs + (size_t)STATE72 * b);
        QState t = next_state<N>(s, actions[b]);
        pack72(t, N, out + (size_t)STATE72 * b);
    }
}
template <int N>
static void status_batch(const uint8_t* recs, int B, int draw, uint8_t* out) {
    for (int b = 0; b < B; ++b) {
        QState s = unpack72(recs + (size_t)STATE72 * b);
        out[b] = (uint8_t)((is_lose<N>(s) ? 1 : 0) | (is_draw(s, draw) ? 2 : 0));
    }
}

#define DISPATCH(N, CALL) \
    switch (N) { case 3: CALL(3); break; case 5: CALL(5); break; case 7: CALL(7); break; case 9: CALL(9); break; default: return -1; }

extern "C" {
int hc_legal_actions_batch(int N, const uint8_t* recs, int B, int16_t* out, int32_t* counts) {
#define C1(n) legal_batch<n>(recs, B, out, counts)
    DISPATCH(N, C1)
    return 0;
}
int hc_next_batch(int N, const uint8_t* recs, const int32_t* actions, int B, uint8_t* out) {
#define C2(n) next_batch<n>(recs, actions, B, out)
    DISPATCH(N, C2)
    return 0;
}
int hc_status_batch(int N, const uint8_t* recs, int B, int draw, uint8_t* out) {
#define C3(n) status_batch<n>(recs, B, draw, out)
    DISPATCH(N, C3)
    return 0;
}
int hc_pack_roundtrip(int N, const uint8_t* rec, uint8_t* out) {
    QState s = unpack72(rec);
    pack72(s, N, out);
    return 0;
}
}
