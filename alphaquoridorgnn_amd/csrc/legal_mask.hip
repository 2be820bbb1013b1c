// legal_mask.hip -- K3: batched State.legal_actions() (game_logic.py:103-117) for gfx950.
//
// One 64-lane wavefront per state, one lane per wall slot (64 slots at 9x9).  Each lane tests both
// orientations of its slot: geometric placement (bit ops on the slot masks), the reference's touch-count
// prefilter, and -- only when that says "possibly blocking" -- the two jump-aware flood fills.  The
// reference's list order (pawn moves, then H,V interleaved per slot) is rebuilt with two wave ballots and
// popcounts, so no atomics and no sorting.  Integer/bit work only: the bound is VALU issue + divergence,
// not HBM (100 algorithmic bytes per state).
#include "aqg_common.hpp"
#include "legal_wave.hpp"

namespace aqg {

template <int N>
__global__ __launch_bounds__(256) void legal_actions_kernel(const void* __restrict__ states, int fmt, int B,
                                                            uint8_t* __restrict__ mask, uint8_t* __restrict__ order,
                                                            int32_t* __restrict__ count,
                                                            const uint8_t* __restrict__ active) {
    constexpr int A = Geo<N>::A;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    if (active && !active[b]) return;
    const QState s = load_state(states, fmt, b);
    const int total = wave_legal_actions<N>(s, lane, mask ? mask + (size_t)b * A : nullptr,
                                            order ? order + (size_t)b * MAX_LEGAL : nullptr);
    if (count && lane == 0) count[b] = total;
}

template <int N>
__global__ void state_next_kernel(const uint8_t* __restrict__ in, const int32_t* __restrict__ actions, int B,
                                  uint8_t* __restrict__ out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    QState s = unpack72(in + (size_t)b * STATE72);
    QState t = next_state<N>(s, actions[b]);
    pack72(t, N, out + (size_t)b * STATE72);
}

template <int N>
__global__ void state_status_kernel(const uint8_t* __restrict__ in, int B, int draw, uint8_t* __restrict__ flags) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    QState s = unpack72(in + (size_t)b * STATE72);
    flags[b] = (uint8_t)((is_lose<N>(s) ? 1 : 0) | (is_draw(s, draw) ? 2 : 0));
}

#define AQG_DISPATCH_N(N, CALL)                  \
    switch (N) {                                 \
        case 3: CALL(3); break;                  \
        case 5: CALL(5); break;                  \
        case 7: CALL(7); break;                  \
        case 9: CALL(9); break;                  \
        default: return fail("unsupported board_size (odd 3..9)"); \
    }

int launch_legal_actions(int N, const void* states, int fmt, int B, uint8_t* mask, uint8_t* order, int32_t* count,
                         const uint8_t* active, hipStream_t st) {
    if (B <= 0) return 0;
    dim3 grid((B + 3) / 4), block(256);
#define CALL_LA(n) hipLaunchKernelGGL(legal_actions_kernel<n>, grid, block, 0, st, states, fmt, B, mask, order, count, active)
    AQG_DISPATCH_N(N, CALL_LA)
    return check_launch("legal_actions_kernel");
}

int launch_state_next(int N, const uint8_t* in, const int32_t* actions, int B, uint8_t* out, hipStream_t st) {
    if (B <= 0) return 0;
    dim3 grid((B + 255) / 256), block(256);
#define CALL_SN(n) hipLaunchKernelGGL(state_next_kernel<n>, grid, block, 0, st, in, actions, B, out)
    AQG_DISPATCH_N(N, CALL_SN)
    return check_launch("state_next_kernel");
}

int launch_state_status(int N, const uint8_t* in, int B, int draw, uint8_t* flags, hipStream_t st) {
    if (B <= 0) return 0;
    dim3 grid((B + 255) / 256), block(256);
#define CALL_SS(n) hipLaunchKernelGGL(state_status_kernel<n>, grid, block, 0, st, in, B, draw, flags)
    AQG_DISPATCH_N(N, CALL_SS)
    return check_launch("state_status_kernel");
}

}  // namespace aqg
