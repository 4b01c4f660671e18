// tile_kernels.hip -- the dense-tile half of libflex_spmm.so: C[rows,:] += Atile[32x32] * B[cols,:] on the matrix cores.
//
// north_star: "MFMA only where RCM/Gorder reordering yields dense block-sparse tiles".  The reference only has a stub
// there (flex_spmm.cu:1164-1168 prints "tensor core"); the orderings that would feed it are order_gorder.cu:35-143 and
// DataLoader.cu:789-857.  Here the planner (dense_tiles.cpp, detect_dense_tiles) looks at the matrix in SCHEDULE coordinates
// after the chosen ordering: a 32 x 32 tile (32 consecutive rows of the schedule x 32 consecutive column positions) whose
// fill reaches the threshold leaves the record stream and is stored as a dense fp32 block; everything else stays with
// the vector kernel.  This kernel then adds the dense part: one wave per row tile, all dense tiles of that row tile in
// column order into 32 x 64 accumulators (v_mfma_f32_32x32x2_f32, two 32-column output tiles per wave), and ONE read-modify-
// write of the 32 C rows at the end -- it runs after the vector kernel on the same stream, so the sum order is fixed
// (vector part first, tiles in column order) and the result is reproducible.
//
// What the matrix core buys: the 32 gathered B rows of a tile are used by all 32 rows of A (the vector kernel gathers
// a B row once per nonzero), and v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf chain at the fp32 vector peak
// (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD); a tile does 1/fill times the useful flops, which is why only tiles above a
// fill threshold are routed here.
//   A operand: lane l holds A[row l&31][k = 2*kk + (l>>5)]      (stored by the planner in exactly this order)
//   B operand: lane l holds B[colrow(2*kk + (l>>5))][c0 + (l&31)] (4-byte loads: 32 lanes = one 128-byte line of a B row)
//   C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
#include "internal.h"

namespace flex {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef FLEX_TILE_WAVES
#define FLEX_TILE_WAVES 3  // waves per SIMD the register allocator aims for (two whole tiles of operands are live in the loop)
#endif
constexpr int kTileNT = 2;  // 32-column output tiles per wave: 64 columns of C per wave, grid.y = ceil(k / 64)

// One wave = one row tile x 64 columns of C.  The loop is software-pipelined by hand, a whole tile deep: the 32 B values per lane
// and the A block of the NEXT tile, and the column offsets of the tile after it, are in flight while the MFMAs of the current tile
// issue (the matrix cores need 21 us for the 12 500-tile probe, the fabric 69 us for its 415 MB).
template <bool OFF32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FLEX_TILE_WAVES))) void spmm_tile_kernel(TileView tv, const float *__restrict__ B, float *__restrict__ C, int k,
                                                        int ldb, int ldc) {
    constexpr int NT = kTileNT;
    const int lane = threadIdx.x & 63;
    // 64-column groups of C: for 1, 2 or 4 of them (k <= 256) the waves of a row tile's groups sit side by side in one workgroup
    // (1-D grid), so that the A blocks they all read come from HBM once; otherwise the group is the slow grid dimension.
    const uint32_t wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t groups = (static_cast<uint32_t>(k) + 32 * NT - 1) / (32 * NT);
    const bool side_by_side = gridDim.y == 1;
    const uint32_t rt = side_by_side ? wid / groups : wid;
    if (rt >= tv.n_row_tiles) return;
    const int half = lane >> 5, j = lane & 31;
    const int n_base = static_cast<int>(side_by_side ? wid % groups : blockIdx.y) * (32 * NT);
    const uint32_t t0 = tv.rt_ptr[rt], t1 = tv.rt_ptr[rt + 1];
    const uint32_t row_l = tv.rt_rows[static_cast<uint64_t>(rt) * 32 + j];  // used by the epilogue only: issued early
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    // columns this lane reads/writes; past k the address is clamped (a valid column of the same row) and the value zeroed
    int col[NT];
    bool col_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        col_ok[nt] = n_base + nt * 32 + j < k;
        col[nt] = col_ok[nt] ? n_base + nt * 32 + j : 0;
    }
    auto load_a = [&](uint32_t t, f32x4 (&a)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4 *>(tv.a + (static_cast<uint64_t>(t) * 4 + q) * 256 + lane * 4);
    };
    // OFF32: SGPR base + one 32-bit VGPR offset per load (a 64-bit address per load would hold two registers each while 32 loads are in flight)
    const char *const Bb = reinterpret_cast<const char *>(B);
    uint32_t loff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) loff[nt] = static_cast<uint32_t>(col[nt]) * 4u;
    auto load_b = [&](uint32_t boff_l, int q, float (&b)[4][NT]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t off = __shfl(boff_l, 2 * (4 * q + e) + half);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (OFF32) b[e][nt] = *reinterpret_cast<const float *>(Bb + static_cast<uint32_t>(off + loff[nt]));
                else b[e][nt] = (B + static_cast<uint64_t>(off) * ldb)[col[nt]];
            }
        }
    };
    uint32_t boff_cur = tv.boff[static_cast<uint64_t>(t0) * 32 + j];
    uint32_t boff_nxt = tv.boff[static_cast<uint64_t>(min(t0 + 1, t1 - 1)) * 32 + j];
    f32x4 a_cur[4];
    load_a(t0, a_cur);
    float bq[4][4][NT];  // the B operands of one whole tile: [k-group][k-step][output tile]
#pragma unroll
    for (int q = 0; q < 4; ++q) load_b(boff_cur, q, bq[q]);
    for (uint32_t t = t0; t < t1; ++t) {
        bool any_bad = false;  // a non-finite B value among this tile's operands (per lane)
        // The NEXT tile's operands -- its 32 B values per lane and its A block -- and the column offsets of the tile AFTER it (the B
        // addresses of the next round of loads) are all in flight before this tile's first MFMA: a tile is 2 048 MFMA cycles (~1 us)
        // per wave, a load under traffic comes back in 1-3 us, and with one k-group of look-ahead (rounds 2-3) the wave stalled in
        // every k-group (118 us for the 12 500-tile probe's 415 MB: 3.5 TB/s).  The last tiles prefetch the last tile again:
        // harmless, keeps every load unconditional.
        const uint32_t boff_nn = tv.boff[static_cast<uint64_t>(min(t + 2, t1 - 1)) * 32 + j];
        f32x4 a_nxt[4];
        load_a(min(t + 1, t1 - 1), a_nxt);
        float bn[4][4][NT];
#pragma unroll
        for (int q = 0; q < 4; ++q) load_b(boff_nxt, q, bn[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // A tile is stored dense: its absent cells are zeros of the A operand, and 0 x inf = NaN would reach rows that do not
            // reference that B row (the vector kernel, the oracle and the reference never touch it).  Non-finite B values therefore
            // enter the MFMA as 0; their exact contribution -- to the rows whose cell is present, explicit zeros included -- is added
            // after the tile by the masked pass below.
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bv = col_ok[nt] ? bq[q][e][nt] : 0.f;
                    const bool bad = (__float_as_uint(bv) & 0x7F800000u) == 0x7F800000u;
                    any_bad |= bad;
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q][e], bad ? 0.f : bv, acc[nt], 0, 0, 0);
                }
        }
        if (__builtin_amdgcn_ballot_w64(any_bad) != 0) {  // wave-uniform, rare: the exact share of this tile's non-finite B values
            const uint32_t mask_l = tv.mask[static_cast<uint64_t>(t) * 32 + j];  // lane j holds row j's cell mask
            for (int kc = 0; kc < 32; ++kc) {
                const uint32_t off = __shfl(boff_cur, kc);
                const float *brow = OFF32 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + off)
                                          : B + static_cast<uint64_t>(off) * ldb;
                // A[i][kc] sits at (q, lane, e) = (kc >> 3, i + 32 (kc & 1), (kc >> 1) & 3) of the operand-ordered block
                const float *acol = tv.a + (static_cast<uint64_t>(t) * 4 + (kc >> 3)) * 256 + ((kc >> 1) & 3) + 128 * (kc & 1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bv = col_ok[nt] ? brow[col[nt]] : 0.f;
                    const bool bad = (__float_as_uint(bv) & 0x7F800000u) == 0x7F800000u;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // the C row this register holds
                        const uint32_t mi = __shfl(mask_l, i);
                        if (bad && ((mi >> kc) & 1u)) acc[nt][reg] = fmaf(acol[i * 4], bv, acc[nt][reg]);
                    }
                }
            }
        }
        boff_cur = boff_nxt;
        boff_nxt = boff_nn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a_cur[q] = a_nxt[q];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq[q][e][nt] = bn[q][e][nt];
        }
    }
    // C[rows of this row tile, :] += acc   (the vector kernel wrote those rows earlier on this stream).  All loads of BOTH
    // output tiles go out before the first add: a load -> add -> store chain per element is sixty-four dependent round trips
    // (the compiler cannot move a load of one C row above the store to another), and one chain per output tile is still two.
    float *base[16];  // column n_base + j of each of the sixteen C rows this lane holds; output tile nt sits 32 columns further
    float old[NT][16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const uint32_t dst = __shfl(row_l, (reg & 3) + 8 * (reg >> 2) + 4 * half);
        // a row tile that hangs over the end of the schedule names no row there: never loaded, never stored
        base[reg] = dst == 0xFFFFFFFFu ? nullptr : C + static_cast<uint64_t>(dst) * ldc + n_base + j;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) old[nt][reg] = (base[reg] && col_ok[nt]) ? base[reg][32 * nt] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            if (base[reg] && col_ok[nt]) base[reg][32 * nt] = old[nt][reg] + acc[nt][reg];
}

}  // namespace

int launch_tiles(const TileView &tv, bool off32, const float *dB, float *dC, int k, int ldb, int ldc, hipStream_t s) {
    if (tv.n_row_tiles == 0) return FLEX_OK;
    const uint32_t groups = (static_cast<uint32_t>(k) + 32 * kTileNT - 1) / (32 * kTileNT);
    const bool side_by_side = groups == 1 || groups == 2 || groups == 4;  // a workgroup's 4 waves hold whole row tiles
    const dim3 grid = side_by_side ? dim3((tv.n_row_tiles * groups + 3) / 4, 1) : dim3((tv.n_row_tiles + 3) / 4, groups);
    const dim3 block(256);
    if (off32)
        hipLaunchKernelGGL(spmm_tile_kernel<true>, grid, block, 0, s, tv, dB, dC, k, ldb, ldc);
    else
        hipLaunchKernelGGL(spmm_tile_kernel<false>, grid, block, 0, s, tv, dB, dC, k, ldb, ldc);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

}  // namespace flex
