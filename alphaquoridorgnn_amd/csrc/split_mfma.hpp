// split_mfma.hpp -- the fp16 hi/lo split arithmetic shared by the inference trunk (gcn_forward.hip) and the training step
// (gcn_train.hip): every f32 operand x travels as two fp16 numbers hi = RNE16(x), lo = RNE16(x - hi) (11 + 11 mantissa bits), a
// product is rebuilt as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with f32 accumulation (the dropped lo*lo term is
// ~2^-22 |ab|), and the board's banded adjacency is enumerated in the k-slot order the accumulators hold the nodes in.
#pragma once
#include <hip/hip_runtime.h>

namespace aqg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int PROW = 256;                    // plane row = 128 fp16, unpadded: 16 slots of 16 B, XOR-swizzled by the row
constexpr int PPLANE = 81 * PROW;            // plane stride (20,736 B)
// Byte offset of 16-byte slot `slot` (0..15) of plane row `row`: slot ^ (row & 15).  The 16 rows an A-fragment
// ds_read_b128 lane group touches (same slot, rows 16m + 0..15) land on 16 distinct bank slots, and so do the 16 rows of
// one 8-byte store group -- what the 272-byte padded rows did before, without the 2.6 KB of padding per board.
__device__ __forceinline__ int plane_off(int row, int slot) { return row * PROW + ((slot ^ (row & 15)) << 4); }

__device__ __forceinline__ unsigned int cvt_pk_f16(float a, float b) {            // RNE, low half = a
    return __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){a, b}, f16x2));
}
__device__ __forceinline__ f32x4 f16_pairs_to_f32(unsigned int p01, unsigned int p23) {
    const f32x2 a = __builtin_convertvector(__builtin_bit_cast(f16x2, p01), f32x2);
    const f32x2 b = __builtin_convertvector(__builtin_bit_cast(f16x2, p23), f32x2);
    return (f32x4){a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ f32x4 mfma_f16(const u32x4 a, const u32x4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Low part of the fp16 split of two f32 values whose high parts are the halves of h01: f16(x0 - h01.lo) | f16(x1 - h01.hi) << 16.
// v_fma_mix{lo,hi}_f16 reads the fp16 half directly and rounds the f32 difference (exact: h is x rounded to 11 bits) to
// fp16 in ONE instruction per value, where convert-back / subtract / convert-pair took 2.5.  hipcc does not select the
// mix forms for this pattern, hence the asm; its result must not feed an MFMA without the wait states of split_fence().
__device__ __forceinline__ unsigned int lo_pair(unsigned int h01, float x0, float x1) {
    unsigned int d;
    // (-1 comes in a scalar register: how a floating-point inline constant is widened for a source whose op_sel_hi bit says
    // "f32" is not something to depend on)
    asm("v_fma_mixlo_f16 %0, %1, %4, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(d) : "v"(h01), "v"(x0), "v"(x1), "s"(-1.0f));
    return d;
}
// VALU write inside an asm statement -> MFMA operand: hipcc pads nothing for asm (guide 5.7 item 2); the fragments pass
// through this statement (two wait states) before their first MFMA.
__device__ __forceinline__ void split_fence(u32x4& a, u32x4& b, u32x4& c) {
    asm volatile("s_nop 3" : "+v"(a), "+v"(b), "+v"(c));
}

// Banded adjacency blocks of the 9x9 board: 81 nodes = 6 node tiles of 16 (tile 5 = node 80) x 3 k blocks of 32 nodes; a node's
// closed neighbourhood is |k - n| in {0, 1, 9}, so 10 of the 18 (k-block, node-tile) blocks are non-zero.  k-slot e of lane
// (c = lane & 15, q = lane >> 4) of k block kb is node 32 kb + 16 (e >> 2) + 4 q + (e & 3): the order in which the accumulators of
// two consecutive 16-row tiles (lane = column, 4 consecutive rows per lane) hold the nodes, so that such accumulators ARE
// operand fragments of a product contracted over the nodes.
constexpr int AF_BLOCKS = 10;
// non-zero (k-block, node-tile) blocks of the banded adjacency, 2 / 3 bits per entry, block 0 in the low bits:
//   kb = {0,0,1,0,1,1,2,1,2,2}   nt = {0,1,1,2,2,3,3,4,4,5}
constexpr unsigned int AF_KB_PACK = 0u | (0u << 2) | (1u << 4) | (0u << 6) | (1u << 8) | (1u << 10) | (2u << 12) | (1u << 14) | (2u << 16) | (2u << 18);
constexpr unsigned int AF_NT_PACK = 0u | (1u << 3) | (1u << 6) | (2u << 9) | (2u << 12) | (3u << 15) | (3u << 18) | (4u << 21) | (4u << 24) | (5u << 27);
constexpr int af_kb(int blk) { return (AF_KB_PACK >> (2 * blk)) & 3; }
constexpr int af_nt(int blk) { return (AF_NT_PACK >> (3 * blk)) & 7; }


}  // namespace aqg
