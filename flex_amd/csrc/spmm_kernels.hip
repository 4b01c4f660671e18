// spmm_kernels.hip -- gfx950 (CDNA4, wave64) SpMM kernels of libflex_spmm.so.
//
// Replaces the reference's kernel family flex.cu:80-4124 (selected kernel
// alpha_w_atomic_spmm_v36, flex.cu:4008-4124; closest in structure: spmm_test2,
// flex.cu:190-273).  None of that code is reused.  Design:
//
//  * A *task* is one C row (or one chunk of a long row); a *wave* owns a short
//    contiguous run of tasks chosen by the host planner so that every wave has
//    about the same number of nonzeros.  No inter-wave communication, no
//    atomics, no barriers, no LDS: the kernel is a pure gather stream.
//  * k/4 lanes (G) cooperate on one nonzero, each lane owning 4 consecutive
//    columns of C, so one wave-instruction gathers 64/G different B rows with
//    16-byte loads (k=128: two 512-B rows = 1 KiB per global_load_dwordx4;
//    k=32: eight 128-B rows).  The 64/G slots of a wave walk the SAME row with
//    stride 64/G and are combined by a log2(64/G)-step xor-shuffle at row end.
//  * U independent gathers are issued per lane before the first FMA so that
//    every wave keeps U KiB in flight; the tail of a row is one branch-free
//    block of exactly the leftover step count, so short rows (GNN graphs:
//    median degree < 10) still issue all their gathers back to back.
//  * Workgroup ids are remapped so that each XCD (private 4 MiB L2) walks its
//    own contiguous eighth of the schedule: rows that are neighbours in the
//    (RCM) schedule share B rows through that XCD's L2.
//  * C is written once with non-temporal 16-byte stores; rows split over
//    several waves go to a k-wide partial slot instead and are summed, in a
//    fixed order, by a second tiny kernel (deterministic; the reference uses
//    atomicAdd for its split rows, mat.cu:816-824).
#include "internal.h"

namespace flex {
namespace {

__device__ __forceinline__ float as_f32(uint32_t u) { return __uint_as_float(u); }

template <bool OFF32>
__device__ __forceinline__ float4 gather4(const char *__restrict__ Bb, uint32_t recx, uint32_t lane_off,
                                          uint64_t row_bytes) {
    if constexpr (OFF32) {
        // base (SGPR pair) + 32-bit VGPR offset: global_load_dwordx4 v, v_off, s[base]
        return *reinterpret_cast<const float4 *>(Bb + static_cast<uint32_t>(recx + lane_off));
    } else {
        return *reinterpret_cast<const float4 *>(Bb + (static_cast<uint64_t>(recx) * row_bytes + lane_off));
    }
}

__device__ __forceinline__ void fma4(float4 &acc, float v, const float4 &b) {
    acc.x = fmaf(v, b.x, acc.x);
    acc.y = fmaf(v, b.y, acc.y);
    acc.z = fmaf(v, b.z, acc.z);
    acc.w = fmaf(v, b.w, acc.w);
}

// N steps of S nonzeros starting at z.  With PRED, steps whose record index falls at or
// past `ze` are neutralised without branches: the index is clamped to the row's last
// record (so every address stays valid and the N gathers still issue back to back) and
// both the value and the gathered B entries are zeroed (0*inf must not leak in).
template <int N, int S, bool OFF32, bool PRED>
__device__ __forceinline__ void steps(float4 &acc, const uint2 *__restrict__ rec, const char *__restrict__ Bb,
                                      uint32_t z, uint32_t ze, int slot, uint32_t lane_off,
                                      uint64_t row_bytes) {
    uint2 r[N];
    float4 b[N];
    bool ok[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        uint32_t zi = z + u * S + slot;
        ok[u] = PRED ? (zi < ze) : true;
        if (PRED) zi = min(zi, ze - 1);
        r[u] = rec[zi];
    }
#pragma unroll
    for (int u = 0; u < N; ++u) b[u] = gather4<OFF32>(Bb, r[u].x, lane_off, row_bytes);
#pragma unroll
    for (int u = 0; u < N; ++u) {
        float v = as_f32(r[u].y);
        if (PRED) {
            v = ok[u] ? v : 0.f;
            b[u].x = ok[u] ? b[u].x : 0.f;
            b[u].y = ok[u] ? b[u].y : 0.f;
            b[u].z = ok[u] ? b[u].z : 0.f;
            b[u].w = ok[u] ? b[u].w : 0.f;
        }
        fma4(acc, v, b[u]);
    }
}

// the 1..U steps left after the unrolled loop, as ONE block of exactly that many
// gathers (wave-uniform switch), only the last step predicated per lane
template <int U, int S, bool OFF32>
__device__ __forceinline__ void tail_steps(uint32_t rem, float4 &acc, const uint2 *__restrict__ rec,
                                           const char *__restrict__ Bb, uint32_t z, uint32_t ze, int slot,
                                           uint32_t lane_off, uint64_t row_bytes) {
#define FLEX_TAIL_CASE(N)                                                                  \
    case N:                                                                                \
        if constexpr (N <= U) steps<N, S, OFF32, true>(acc, rec, Bb, z, ze, slot, lane_off, row_bytes); \
        break;
    switch (rem) {
        FLEX_TAIL_CASE(1)
        FLEX_TAIL_CASE(2)
        FLEX_TAIL_CASE(3)
        FLEX_TAIL_CASE(4)
        FLEX_TAIL_CASE(5)
        FLEX_TAIL_CASE(6)
        FLEX_TAIL_CASE(7)
        FLEX_TAIL_CASE(8)
        default: break;
    }
#undef FLEX_TAIL_CASE
}

typedef float v4f __attribute__((ext_vector_type(4)));

// G lanes per nonzero, 4 columns per lane; covers k <= 4*G per blockIdx.y tile.
template <int G, bool OFF32, int U>
__global__ __launch_bounds__(256) void spmm_v4_kernel(PlanView p, const float *__restrict__ B,
                                                      float *__restrict__ C) {
    constexpr int S = 64 / G;
    static_assert(U >= 2 && U <= 8, "tail_steps covers 1..7 leftover steps");
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware remap: hardware deals consecutive workgroup ids round-robin over the 8
    // XCDs; give XCD x the contiguous schedule slice [x*cpx, (x+1)*cpx). gridDim.x % 8 == 0.
    const uint32_t cpx = gridDim.x / kXcds;
    const uint32_t bid = p.xcd_remap ? (blockIdx.x % kXcds) * cpx + (blockIdx.x / kXcds) : blockIdx.x;
    const uint32_t w = bid * kWavesPerBlock + wib;
    if (w >= p.n_waves) return;

    const int slot = lane / G;
    const int sub = lane % G;
    const int k = p.k;
    const int c0 = blockIdx.y * (4 * G) + sub * 4;  // first of this lane's 4 columns
    const bool col_ok = c0 < k;                      // k % 4 == 0 on this path
    const uint32_t lane_off = (col_ok ? c0 : 0) * 4u;
    const char *Bb = reinterpret_cast<const char *>(B);
    const uint64_t row_bytes = static_cast<uint64_t>(k) * 4u;
    const uint2 *__restrict__ rec = p.rec;

    // A wave owns at most 63 tasks (planner invariant): fetch all its descriptors with one
    // coalesced load per array and hand them out with v_readlane, so the per-row critical
    // path has no dependent descriptor load in it.
    const uint32_t t0 = p.w_task[w], t1 = p.w_task[w + 1];
    const uint32_t nt = t1 - t0;
    const uint32_t my_beg = (static_cast<uint32_t>(lane) <= nt) ? p.t_beg[t0 + lane] : 0u;
    const uint32_t my_dst = (static_cast<uint32_t>(lane) < nt) ? p.t_dst[t0 + lane] : 0u;
    for (uint32_t i = 0; i < nt; ++i) {
        const uint32_t zb = __builtin_amdgcn_readlane(my_beg, i);
        const uint32_t ze = __builtin_amdgcn_readlane(my_beg, i + 1);
        const uint32_t dst = __builtin_amdgcn_readlane(my_dst, i);
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        uint32_t z = zb;
        for (; z + S * U <= ze; z += S * U)  // every record of the block exists
            steps<U, S, OFF32, false>(acc, rec, Bb, z, ze, slot, lane_off, row_bytes);
        // 0..U steps left (U when the last one is partial), the last one predicated per lane
        tail_steps<U, S, OFF32>((ze - z + S - 1) / S, acc, rec, Bb, z, ze, slot, lane_off, row_bytes);
        // combine the S slots (they hold disjoint nonzeros of the same row)
#pragma unroll
        for (int off = G; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off);
            acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off);
            acc.w += __shfl_xor(acc.w, off);
        }
        if (slot == 0 && col_ok) {
            if (dst & kPartialFlag) {
                float4 *o = reinterpret_cast<float4 *>(p.partial + static_cast<uint64_t>(dst & ~kPartialFlag) * k + c0);
                *o = acc;
            } else {
                v4f *o = reinterpret_cast<v4f *>(C + static_cast<uint64_t>(dst) * k + c0);
                const v4f val = {acc.x, acc.y, acc.z, acc.w};
                __builtin_nontemporal_store(val, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Flat-stream kernel (the default).  Rows of GNN graphs are short (median degree
// < 10), so a per-row loop restarts its load -> gather -> reduce dependency chain
// every few nonzeros and leaves a wave with 2-3 loads in flight.  Here a wave
// treats ALL its records as one stream of steps (S records per step; the planner
// pads every row to a multiple of S with zero-valued records): the records of a
// window are fetched once, coalesced, into a wave-private LDS slice, then blocks
// of U gathers are issued back to back regardless of row boundaries, and a
// wave-uniform scalar check after each step flushes the accumulator when a row
// ends.  No barriers: the LDS slice is private to the wave.
// ---------------------------------------------------------------------------
constexpr int kWindowRecs = 256;  // records staged per wave and window (2 KiB of LDS)

template <int G, bool OFF32, int U>
__global__ __launch_bounds__(256) void spmm_flat_kernel(PlanView p, const float *__restrict__ B,
                                                        float *__restrict__ C) {
    constexpr int S = 64 / G;
    __shared__ uint2 lds_rec[kWavesPerBlock][kWindowRecs];
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t cpx = gridDim.x / kXcds;
    const uint32_t bid = p.xcd_remap ? (blockIdx.x % kXcds) * cpx + (blockIdx.x / kXcds) : blockIdx.x;
    const uint32_t w = bid * kWavesPerBlock + wib;
    if (w >= p.n_waves) return;

    const int slot = lane / G;
    const int sub = lane % G;
    const int k = p.k;
    const int c0 = blockIdx.y * (4 * G) + sub * 4;
    const bool col_ok = c0 < k;
    const uint32_t lane_off = (col_ok ? c0 : 0) * 4u;
    const char *Bb = reinterpret_cast<const char *>(B);
    const uint64_t row_bytes = static_cast<uint64_t>(k) * 4u;
    const uint2 *__restrict__ rec = p.rec;
    uint2 *my_lds = lds_rec[wib];

    const uint32_t t0 = p.w_task[w], t1 = p.w_task[w + 1];
    const uint32_t nt = t1 - t0;  // <= 63 (planner invariant)
    const uint32_t my_beg = (static_cast<uint32_t>(lane) <= nt) ? p.t_beg[t0 + lane] : 0u;
    const uint32_t my_dst = (static_cast<uint32_t>(lane) < nt) ? p.t_dst[t0 + lane] : 0u;
    const uint32_t zb = __builtin_amdgcn_readlane(my_beg, 0);
    const uint32_t ze = __builtin_amdgcn_readlane(my_beg, nt);

    uint32_t ti = 0;                                          // current task
    uint32_t row_end = __builtin_amdgcn_readlane(my_beg, 1);  // where it ends in the record stream
    float4 acc = {0.f, 0.f, 0.f, 0.f};

    // write out every task that ends at stream position `pos` (several when rows are empty)
    auto drain = [&](uint32_t pos) {
        while (ti < nt && row_end == pos) {
            float4 r = acc;
#pragma unroll
            for (int off = G; off < 64; off <<= 1) {
                r.x += __shfl_xor(r.x, off);
                r.y += __shfl_xor(r.y, off);
                r.z += __shfl_xor(r.z, off);
                r.w += __shfl_xor(r.w, off);
            }
            const uint32_t dst = __builtin_amdgcn_readlane(my_dst, ti);
            if (slot == 0 && col_ok) {
                if (dst & kPartialFlag) {
                    *reinterpret_cast<float4 *>(p.partial + static_cast<uint64_t>(dst & ~kPartialFlag) * k + c0) = r;
                } else {
                    const v4f val = {r.x, r.y, r.z, r.w};
                    __builtin_nontemporal_store(val, reinterpret_cast<v4f *>(C + static_cast<uint64_t>(dst) * k + c0));
                }
            }
            acc = {0.f, 0.f, 0.f, 0.f};
            ++ti;
            row_end = __builtin_amdgcn_readlane(my_beg, ti + 1 <= nt ? ti + 1 : nt);
        }
    };
    drain(zb);  // leading empty rows

    for (uint32_t wz = zb; wz < ze; wz += kWindowRecs) {
        const uint32_t wn = min(static_cast<uint32_t>(kWindowRecs), ze - wz);
        // stage this window's records: coalesced 512-B loads, one ds_write_b64 per lane and load
#pragma unroll
        for (int i = 0; i < kWindowRecs / 64; ++i) {
            const uint32_t idx = i * 64 + lane;
            if (idx < wn) my_lds[idx] = rec[wz + idx];
        }
        const uint32_t nsteps = wn / S;  // rows are padded to multiples of S
        for (uint32_t j = 0; j < nsteps; j += U) {
            uint2 r[U];
            float4 b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = my_lds[min(j + u, nsteps - 1) * S + slot];
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = gather4<OFF32>(Bb, r[u].x, lane_off, row_bytes);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < nsteps) {  // wave-uniform
                    fma4(acc, as_f32(r[u].y), b[u]);
                    drain(wz + (j + u + 1) * S);
                }
            }
        }
    }
}

// Any k (k % 4 != 0 or unaligned B/C): one wave per task, lane owns columns
// lane, lane+64, lane+128, lane+192 of the blockIdx.y-th 256-column tile.
template <bool OFF32>
__global__ __launch_bounds__(256) void spmm_generic_kernel(PlanView p, const float *__restrict__ B,
                                                           float *__restrict__ C) {
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t cpx = gridDim.x / kXcds;
    const uint32_t bid = p.xcd_remap ? (blockIdx.x % kXcds) * cpx + (blockIdx.x / kXcds) : blockIdx.x;
    const uint32_t w = bid * kWavesPerBlock + wib;
    if (w >= p.n_waves) return;
    const int k = p.k;
    const int cb = blockIdx.y * 256 + lane;
    const uint2 *__restrict__ rec = p.rec;
    const uint32_t t0 = p.w_task[w], t1 = p.w_task[w + 1];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t zb = p.t_beg[t], ze = p.t_beg[t + 1];
        const uint32_t dst = p.t_dst[t];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t z = zb; z < ze; ++z) {
            const uint2 r = rec[z];
            const float v = as_f32(r.y);
            const float *brow = OFF32 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + r.x)
                                      : B + static_cast<uint64_t>(r.x) * k;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = cb + 64 * i;
                if (c < k) acc[i] = fmaf(v, brow[c], acc[i]);
            }
        }
        float *orow = (dst & kPartialFlag) ? p.partial + static_cast<uint64_t>(dst & ~kPartialFlag) * k
                                           : C + static_cast<uint64_t>(dst) * k;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cb + 64 * i;
            if (c < k) orow[c] = acc[i];
        }
    }
}

// C[row,:] = partial[first,:] + partial[first+1,:] + ... in that fixed order.
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const float *__restrict__ partial,
                                                         const SplitRow *__restrict__ rows, uint32_t n_rows,
                                                         int k, float *__restrict__ C) {
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t i = blockIdx.x * kWavesPerBlock + wib;
    if (i >= n_rows) return;
    const SplitRow sr = rows[i];
    for (int c = lane; c < k; c += 64) {
        float s = 0.f;
        for (uint32_t j = 0; j < sr.count; ++j) s += partial[static_cast<uint64_t>(sr.first + j) * k + c];
        C[static_cast<uint64_t>(sr.row) * k + c] = s;
    }
}

// dst[r,:] = src[idx[r],:]  (≙ flexspmm_v9_permuteX, flex.cu:276-289)
template <bool VEC4>
__global__ __launch_bounds__(256) void gather_rows_kernel(float *__restrict__ dst, const float *__restrict__ src,
                                                          const int32_t *__restrict__ idx, int64_t n, int k) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6); r < n; r += nwaves) {
        const int64_t s = idx[r];
        if constexpr (VEC4) {
            const float4 *sp = reinterpret_cast<const float4 *>(src + s * k);
            float4 *dp = reinterpret_cast<float4 *>(dst + r * k);
            for (int c = lane; c < k / 4; c += 64) dp[c] = sp[c];
        } else {
            for (int c = lane; c < k; c += 64) dst[r * k + c] = src[s * k + c];
        }
    }
}

template <int G, bool OFF32, int U>
int launch_v4(const PlanView &v, const float *dB, float *dC, hipStream_t s) {
    uint32_t nblk = (v.n_waves + kWavesPerBlock - 1) / kWavesPerBlock;
    nblk = (nblk + kXcds - 1) / kXcds * kXcds;
    const uint32_t ktiles = (v.k + 4 * G - 1) / (4 * G);
    // lds_extra: unused dynamic LDS that only lowers the number of resident workgroups per CU
    if (v.variant == 1)
        hipLaunchKernelGGL((spmm_v4_kernel<G, OFF32, (U > 4 ? 8 : 4)>), dim3(nblk, ktiles), dim3(256), v.lds_extra, s, v, dB, dC);
    else
        hipLaunchKernelGGL((spmm_flat_kernel<G, OFF32, U>), dim3(nblk, ktiles), dim3(256), v.lds_extra, s, v, dB, dC);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

template <int G, int U>
int launch_v4_off(const PlanView &v, bool off32, const float *dB, float *dC, hipStream_t s) {
    return off32 ? launch_v4<G, true, U>(v, dB, dC, s) : launch_v4<G, false, U>(v, dB, dC, s);
}

}  // namespace

int launch_spmm(const PlanView &v, int lanes_per_nz, bool off32, bool vec4, const float *dB, float *dC,
                hipStream_t s) {
    if (v.n_waves == 0) return FLEX_OK;
    if (!vec4) {
        uint32_t nblk = (v.n_waves + kWavesPerBlock - 1) / kWavesPerBlock;
        nblk = (nblk + kXcds - 1) / kXcds * kXcds;
        const uint32_t ktiles = (v.k + 255) / 256;
        if (off32)
            hipLaunchKernelGGL((spmm_generic_kernel<true>), dim3(nblk, ktiles), dim3(256), 0, s, v, dB, dC);
        else
            hipLaunchKernelGGL((spmm_generic_kernel<false>), dim3(nblk, ktiles), dim3(256), 0, s, v, dB, dC);
        FLEX_HIP_TRY(hipGetLastError());
        return FLEX_OK;
    }
    switch (lanes_per_nz) {
        case 8: return launch_v4_off<8, 4>(v, off32, dB, dC, s);
        case 16: return launch_v4_off<16, 4>(v, off32, dB, dC, s);
        case 32: return launch_v4_off<32, 8>(v, off32, dB, dC, s);
        case 64: return launch_v4_off<64, 8>(v, off32, dB, dC, s);
        default: return FLEX_ERR_UNSUPPORTED;
    }
}

int launch_fixup(const float *partial, const SplitRow *rows, uint32_t n_rows, int k, float *dC, hipStream_t s) {
    if (n_rows == 0) return FLEX_OK;
    const uint32_t nblk = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(spmm_fixup_kernel, dim3(nblk), dim3(256), 0, s, partial, rows, n_rows, k, dC);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int launch_gather_rows(float *dst, const float *src, const int32_t *idx, int64_t n, int k, hipStream_t s) {
    if (n <= 0) return FLEX_OK;
    const int64_t want = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint32_t nblk = static_cast<uint32_t>(want < 2048 ? want : 2048);  // grid-stride beyond 8 blocks/CU
    const bool vec4 = (k % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0);
    if (vec4)
        hipLaunchKernelGGL((gather_rows_kernel<true>), dim3(nblk), dim3(256), 0, s, dst, src, idx, n, k);
    else
        hipLaunchKernelGGL((gather_rows_kernel<false>), dim3(nblk), dim3(256), 0, s, dst, src, idx, n, k);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

}  // namespace flex
