// tile_kernels.hip -- the dense-tile half of libflex_spmm.so: C[rows,:] += Atile[32x32] * B[cols,:] on the matrix cores.
//
// north_star: "MFMA only where RCM/Gorder reordering yields dense block-sparse tiles".  The reference only has a stub
// there (flex_spmm.cu:1164-1168 prints "tensor core"); the orderings that would feed it are order_gorder.cu:35-143 and
// DataLoader.cu:789-857.  Here the planner (dense_tiles.cpp, detect_dense_tiles) looks at the matrix in SCHEDULE coordinates
// after the chosen ordering: a 32 x 32 tile (32 consecutive rows of the schedule x 32 consecutive column positions) whose
// fill reaches the threshold leaves the record stream and is stored as a dense fp32 block; everything else stays with
// the vector kernel.  This kernel then adds the dense part: one wave per GROUP of two vertically adjacent row tiles, all dense
// tiles of the group in column order into 2 x 32 x 128 accumulators (v_mfma_f32_32x32x2_f32), and ONE read-modify-
// write of the 64 C rows at the end -- it runs after the vector kernel on the same stream, so the sum order is fixed
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

// One wave = one GROUP of two vertically adjacent row tiles (64 rows) x NT 32-column output tiles (NT = 4: 128 columns of C; 2 for
// k <= 64).  Round 4: (a) the B rows of a column tile feed BOTH row tiles' MFMAs -- a 64-row diagonal block fetches its B rows once,
// not twice; (b) a wave covers up to 128 columns, so an A block is read once per launch at k = 128, not once per 64-column slab.
// What is left is the read-modify-write of the C rows (the vector kernel wrote its share of them earlier on this stream): per tile
// 2 KB of A + 4 KB of B (shared) + 8 KB of C at k = 128 where round 3 moved 4 + 8 + 8.
// The loop is software-pipelined by hand: the B values of the next k-group (four k-steps x NT output tiles) and the next A quarters
// are in flight while the MFMAs of the current group issue -- a wave otherwise spends six dependent memory round trips per tile.
template <bool OFF32, int NT>
__global__ __launch_bounds__(256) void spmm_tile_kernel(TileView tv, const float *__restrict__ B, float *__restrict__ C, int k,
                                                        int ldb, int ldc) {
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= tv.n_groups) return;
    const int half = lane >> 5, j = lane & 31;
    const int n_base = blockIdx.y * (32 * NT);
    const uint32_t e0 = tv.gp_ptr[g], e1 = tv.gp_ptr[g + 1];
    // used by the epilogue only: issued early.  Lane j holds the C row of row j of the upper (0) and of the lower (1) row tile.
    const uint32_t row_l[2] = {tv.gp_rows[static_cast<uint64_t>(g) * 64 + j], tv.gp_rows[static_cast<uint64_t>(g) * 64 + 32 + j]};
    f32x16 acc[2][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[h][nt][i] = 0.f;
    // columns this lane reads/writes; past k the address is clamped (a valid column of the same row) and the value zeroed
    int col[NT];
    bool col_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        col_ok[nt] = n_base + nt * 32 + j < k;
        col[nt] = col_ok[nt] ? n_base + nt * 32 + j : 0;
    }
    auto load_a = [&](uint32_t t, int q) -> f32x4 { return *reinterpret_cast<const f32x4 *>(tv.a + (static_cast<uint64_t>(t) * 4 + q) * 256 + lane * 4); };
    auto load_b = [&](uint32_t boff_l, int q, float (&b)[4][NT]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t off = __shfl(boff_l, 2 * (4 * q + e) + half);
            const float *brow = OFF32 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + off)
                                      : B + static_cast<uint64_t>(off) * ldb;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[e][nt] = brow[col[nt]];
        }
    };
    // an entry names the tile of the upper and / or the lower row tile; an absent one is read as a copy of the present one (every
    // load stays unconditional) and its MFMAs are skipped (wave-uniform)
    auto tiles_of = [&](uint32_t e, uint32_t (&t)[2], bool (&has)[2]) {
        const uint2 en = tv.gp_ent[e];
        has[0] = en.x != kNoTile;
        has[1] = en.y != kNoTile;
        t[0] = has[0] ? en.x : en.y;
        t[1] = has[1] ? en.y : en.x;
    };
    uint32_t t_cur[2];
    bool has_cur[2];
    tiles_of(e0, t_cur, has_cur);
    uint32_t boff_cur = tv.boff[static_cast<uint64_t>(t_cur[0]) * 32 + j];
    f32x4 a_q[2] = {load_a(t_cur[0], 0), load_a(t_cur[1], 0)};
    float bq[4][NT];
    load_b(boff_cur, 0, bq);
    for (uint32_t e = e0; e < e1; ++e) {
        bool any_bad = false;  // a non-finite B value among this column tile's operands (per lane)
        const uint32_t en = min(e + 1, e1 - 1);  // the last entry prefetches itself: harmless, keeps every load unconditional
        uint32_t t_nxt[2];
        bool has_nxt[2];
        tiles_of(en, t_nxt, has_nxt);
        const uint32_t boff_nxt = tv.boff[static_cast<uint64_t>(t_nxt[0]) * 32 + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float bn[4][NT];
            f32x4 a_n[2];
            if (q < 3) {
                load_b(boff_cur, q + 1, bn);
                a_n[0] = load_a(t_cur[0], q + 1);
                a_n[1] = load_a(t_cur[1], q + 1);
            } else {
                load_b(boff_nxt, 0, bn);
                a_n[0] = load_a(t_nxt[0], 0);
                a_n[1] = load_a(t_nxt[1], 0);
            }
            // A tile is stored dense: its absent cells are zeros of the A operand, and 0 x inf = NaN would reach rows that do not
            // reference that B row (the vector kernel, the oracle and the reference never touch it).  Non-finite B values therefore
            // enter the MFMA as 0; their exact contribution -- to the rows whose cell is present, explicit zeros included -- is added
            // after the column tile by the masked pass below.
            float bv[4][NT];
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float x = col_ok[nt] ? bq[e4][nt] : 0.f;
                    const bool bad = (__float_as_uint(x) & 0x7F800000u) == 0x7F800000u;
                    any_bad |= bad;
                    bv[e4][nt] = bad ? 0.f : x;
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (has_cur[h]) {  // wave-uniform
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[h][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q[h][e4], bv[e4][nt], acc[h][nt], 0, 0, 0);
                }
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq[e4][nt] = bn[e4][nt];
            a_q[0] = a_n[0];
            a_q[1] = a_n[1];
        }
        if (__builtin_amdgcn_ballot_w64(any_bad) != 0) {  // wave-uniform, rare: the exact share of this column tile's non-finite B values
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!has_cur[h]) continue;
                const uint32_t t = t_cur[h];
                const uint32_t mask_l = tv.mask[static_cast<uint64_t>(t) * 32 + j];  // lane j holds row j's cell mask
                for (int kc = 0; kc < 32; ++kc) {
                    const uint32_t off = __shfl(boff_cur, kc);
                    const float *brow = OFF32 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + off)
                                              : B + static_cast<uint64_t>(off) * ldb;
                    // A[i][kc] sits at (q, lane, e) = (kc >> 3, i + 32 (kc & 1), (kc >> 1) & 3) of the operand-ordered block
                    const float *acol = tv.a + (static_cast<uint64_t>(t) * 4 + (kc >> 3)) * 256 + ((kc >> 1) & 3) + 128 * (kc & 1);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float x = col_ok[nt] ? brow[col[nt]] : 0.f;
                        const bool bad = (__float_as_uint(x) & 0x7F800000u) == 0x7F800000u;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // the C row this register holds
                            const uint32_t mi = __shfl(mask_l, i);
                            if (bad && ((mi >> kc) & 1u)) acc[h][nt][reg] = fmaf(acol[i * 4], x, acc[h][nt][reg]);
                        }
                    }
                }
            }
        }
        boff_cur = boff_nxt;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            t_cur[h] = t_nxt[h];
            has_cur[h] = has_nxt[h];
        }
    }
    // C[rows of this group, :] += acc   (the vector kernel wrote those rows earlier on this stream).  All sixteen
    // loads of an output tile go out before the first add: a load -> add -> store chain per element is sixty-four
    // dependent round trips (the compiler cannot move a load of one C row above the store to another).
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float *ptr[16];
            float old[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const uint32_t dst = __shfl(row_l[h], (reg & 3) + 8 * (reg >> 2) + 4 * half);
                // a group that hangs over the end of the schedule: never load or store such a row
                ptr[reg] = dst == 0xFFFFFFFFu ? nullptr : C + static_cast<uint64_t>(dst) * ldc + col[nt];
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) old[reg] = (ptr[reg] && col_ok[nt]) ? *ptr[reg] : 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (ptr[reg] && col_ok[nt]) *ptr[reg] = old[reg] + acc[h][nt][reg];
        }
}

}  // namespace

int launch_tiles(const TileView &tv, bool off32, const float *dB, float *dC, int k, int ldb, int ldc, hipStream_t s, int cols_per_wave) {
    if (tv.n_groups == 0) return FLEX_OK;
    const dim3 block(256);
    const uint32_t gx = (tv.n_groups + 3) / 4;
    if (k > 64 && cols_per_wave != 64) {
        const dim3 grid(gx, (k + 127) / 128);
        if (off32) hipLaunchKernelGGL((spmm_tile_kernel<true, 4>), grid, block, 0, s, tv, dB, dC, k, ldb, ldc);
        else hipLaunchKernelGGL((spmm_tile_kernel<false, 4>), grid, block, 0, s, tv, dB, dC, k, ldb, ldc);
    } else {
        const dim3 grid(gx, (k + 63) / 64);
        if (off32) hipLaunchKernelGGL((spmm_tile_kernel<true, 2>), grid, block, 0, s, tv, dB, dC, k, ldb, ldc);
        else hipLaunchKernelGGL((spmm_tile_kernel<false, 2>), grid, block, 0, s, tv, dB, dC, k, ldb, ldc);
    }
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

}  // namespace flex
