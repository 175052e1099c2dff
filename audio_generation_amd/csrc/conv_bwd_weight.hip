// Weight / bias gradients of a conv layer for gfx950.
//
// In the polyphase weight space the gradient is one more GEMM,
//   dWp[n = ci*J + j][m = co*q + p] = sum_{b,t} dy[b, co, q t + p] * x[b, ci, t s + j d - P],
// with the long dimension (batch x time) as the contraction: M = q*Cout rows (A = dy),
// N = Cin*J columns (B = shifted views of x), K = B * Lt.  fp32-input MFMA again (exact fp32).
// The contraction is cut into `n_slices` slices that different workgroups accumulate in
// registers; every slice writes its partial tile to the workspace and a second kernel adds the
// slices in a fixed order (deterministic -- no float atomics), folds the polyphase space back to
// the torch weight layout and applies the weight-norm chain rule (w = g v / |v|, utils.py:34-42):
//   dg[r] = <dw_r, v_r> / |v_r|,   dv_r = (g_r / |v_r|) (dw_r - v_r <dw_r, v_r> / |v_r|^2).
#include "mfma_tile.hpp"

namespace agx {

// XCD-aware block order.  The hardware hands consecutive workgroups of the launch order (x fastest, then y, z) to the 8 XCDs in
// turn, so the n-tiles of one contraction slice -- which read the SAME dy rows -- land on 8 different L2s and each fetches
// them from HBM again (9 x for a 128 -> 128 3 x 3 layer: ~5 GB per launch).  Renumbering gives every XCD a CONTIGUOUS range of
// the virtual order: all tiles of a slice share one L2.  Bijective for any grid size.
struct BlockId { int x, y, z; };
__device__ __forceinline__ BlockId xcd_block_id(bool on) {
    BlockId b{int(blockIdx.x), int(blockIdx.y), int(blockIdx.z)};
    if (!on) return b;
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * int(gridDim.z);
    const int L = b.x + gx * (b.y + gy * b.z);
    const int per = total >> 3, rem = total & 7, xcd = L & 7, idx = L >> 3;
    const int v = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
    b.x = v % gx;
    const int t = v / gx;
    b.y = t % gy;
    b.z = t / gy;
    return b;
}


constexpr int BW_T = 64;        // time positions per LDS stage
constexpr int BW_TS = BW_T + 1; // dyS row stride (odd: conflict-free column reads)

// Workgroup tile = (32*MW*WM) rows of m x (32*NW*WN) columns of n; 4 waves as WM x WN.
// The n-tile-0 workgroups also accumulate the bias gradient (row sums of the staged dy tile).
template <int MW, int NW, int WM, int WN, int PREC = 0>   // PREC 1: bf16x3 contraction
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void conv_bwd_weight_kernel(ConvPlan p, int span, int n_chan, int n_slices,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ dy,
                                                              float *__restrict__ part,
                                                              float *__restrict__ bias_part) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = 32 * MW * WM, BN = 32 * NW * WN;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *dys = sm;                 // [BM][BW_TS]
    float *xs = sm + BM * BW_TS;     // [n_chan][span]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int NK = p.Cin * p.J;
    const int n_base = blockIdx.x * BN, m_base = blockIdx.y * BM, slice = blockIdx.z;
    const int ci_first = n_base / p.J;

    int boff[NW];
    bool nvalid[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = n_base + (wn * NW + k) * 32 + li;
        nvalid[k] = n < NK;
        const int nc = min(n, NK - 1);
        const int ci = nc / p.J, j = nc - ci * p.J;
        boff[k] = (ci - ci_first) * span + j * p.d;
    }
    int arow[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) arow[i] = ((wm * MW + i) * 32 + li) * BW_TS;

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
    // bias gradient: TPR threads per row of the dy tile, each summing BW_T / TPR columns
    constexpr int TPR = 256 / BM, CPT = BW_T / TPR;
    float bsum = 0.f;
    const bool do_bias = bias_part != nullptr && blockIdx.x == 0;

    const int chunks = (p.Lt + BW_T - 1) / BW_T;
    const int items = p.B * chunks;
    constexpr int NB = BM * BW_T / 256;       // dy-tile elements per thread (rows tid / 64 + 4 u, column tid % 64)
    const int tt = tid & 63;
    const int total = n_chan * span;
    const float inv_span = 1.f / float(span);
    for (int item = slice; item < items; item += n_slices) {
        const int b = item / chunks, t0 = (item - b * chunks) * BW_T;
        __syncthreads();
        // Staging: all loads of a batch are issued on clamped addresses before any is used (a conditional
        // load is waited for one by one), zeros are selected afterwards.
        // dy tile: row r <-> m = m_base + r = co*q + ph, column tt <-> dy[b, co, q*(t0+tt) + ph];
        // a thread's column tt = tid % 64 is the same for all its elements
        {
            const int t = t0 + tt;
#pragma unroll
            for (int u0 = 0; u0 < NB; u0 += 8) {
                float v[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = min(m_base + (tid >> 6) + 4 * (u0 + u), p.M - 1);
                    const int co = p.q == 1 ? m : m / p.q, ph = m - co * p.q;
                    const int uu = p.q * t + ph;
                    ok[u] = t < p.Lt && uu < p.Lout && m_base + (tid >> 6) + 4 * (u0 + u) < p.M;
                    v[u] = dy[(size_t(b) * p.Cout + co) * p.Lout + min(uu, p.Lout - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) dys[((tid >> 6) + 4 * (u0 + u)) * BW_TS + tt] = ok[u] ? v[u] : 0.f;
            }
        }
        // x tile: channels ci_first .. ci_first + n_chan - 1, positions t0*s - P + [0, span)
        {
            const int in0 = t0 * p.s - p.P;
            const float *xb = x + size_t(b) * p.Cin * p.Lin;
            for (int e0 = tid; e0 < total; e0 += 256 * 8) {
                float v[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * 256, total - 1);
                    const int c = int((float(e) + 0.5f) * inv_span), i = e - c * span;   // exact: e < 2^20
                    const int ch = ci_first + c, pos = in0 + i;
                    ok[u] = ch < p.Cin && pos >= 0 && pos < p.Lvalid;
                    v[u] = xb[size_t(min(ch, p.Cin - 1)) * p.Lin + min(max(pos, 0), p.Lvalid - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * 256 < total) xs[e0 + u * 256] = ok[u] ? v[u] : 0.f;
            }
        }
        __syncthreads();
        if (do_bias) {
            const float *row = dys + (tid / TPR) * BW_TS + (tid % TPR) * CPT;
#pragma unroll
            for (int c = 0; c < CPT; ++c) bsum += row[c];
        }
        if (PREC == 1) {
#pragma unroll
            for (int kb = 0; kb < BW_T / 16; ++kb) {     // K = 16 blocks: this lane's positions 16 kb + 8 lh + 0..7
                const int k0 = 16 * kb + 8 * lh;
                bf16x8 aq[3][MW], bq[3][NW];
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    float xq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xq[e] = dys[arow[i] + k0 + e];
                    split3(xq, aq[0][i], aq[1][i], aq[2][i]);
                }
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    float xq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xq[e] = nvalid[k] ? xs[boff[k] + (k0 + e) * p.s] : 0.f;
                    split3(xq, bq[0][k], bq[1][k], bq[2][k]);
                }
                mfma_block_bf<MW, NW>(acc, aq, bq);
            }
        } else
#pragma unroll 4
        for (int ks = 0; ks < BW_T / 2; ++ks) {
            const int tt = 2 * ks + lh;
            float a[MW], bv[NW];
#pragma unroll
            for (int i = 0; i < MW; ++i) a[i] = dys[arow[i] + tt];
#pragma unroll
            for (int k = 0; k < NW; ++k) bv[k] = nvalid[k] ? xs[boff[k] + tt * p.s] : 0.f;
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int k = 0; k < NW; ++k)
                    acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[k], acc[i][k], 0, 0, 0);
        }
    }
    // partial tiles: column = lane & 31 <-> n, rows (registers) <-> m
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = n_base + (wn * NW + k) * 32 + li;
        if (n >= NK) continue;
        float *dst = part + (size_t(slice) * NK + n) * p.M;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + (wm * MW + i) * 32 + acc_row(r, lh);
                if (m < p.M) dst[m] = acc[i][k][r];
            }
    }
    if (do_bias) {   // block-uniform branch
#pragma unroll
        for (int off = 1; off < TPR; off <<= 1) bsum += __shfl_xor(bsum, off);
        const int m = m_base + tid / TPR;
        if (tid % TPR == 0 && m < p.M) bias_part[size_t(slice) * p.M + m] = bsum;
    }
}

// ------------------------------------------------------------------------------------------------
// Stride-1 layers (every residual-block conv: 48 of the 60 weight gradients of config S): no staging phase and no
// workgroup barrier.  The contraction order of an MFMA is free as long as both operands agree, so lane (li, lh)
// takes the 16 CONSECUTIVE positions t0 + 16 lh + 0..15 of "its" row (A: dy[co = row li]; B: x[ci] shifted by the
// tap) and k-step ks pairs position 16 lh + ks of both.  Every wave runs its own stream over a contiguous range of
// 32-position items: the 32 rows x 128 bytes of an operand block travel global -> LDS by LDS-DMA (dwordx4, 8 lanes
// per row = whole cache lines; loading "row per lane" straight into registers costs one TA cycle per lane and was
// TA-bound at 62-82 TFLOP/s), into a wave-private buffer whose 16-byte chunks are XOR-swizzled (chunk c of row r
// at slot c ^ (r & 7)) so that the four ds_read_b128 per block that fetch the MFMA operands are conflict-free.
// The DMA of item n+1 is in flight during the 16 MW NW MFMAs of item n (operands already in registers).
// Waves left over by a small tile (WK = 4 / (WM WN)) take different slices of the contraction and are added
// through LDS at the end.
// Strided and transposed layers reach the same kernel through phase-split copies of one operand (one pass over it):
//   stride s:  xs[b, ci s + rho, u] = x[b, ci, u s + rho]   -- with j d - P = a s + rho the window of B row (ci, j) is
//              row ci s + rho of xs shifted by a: contiguous again (`sp` = s below, rows of length Lin' = ceil(Lin / s));
//   q phases:  dys[b, co q + ph, t] = dy[b, co, q t + ph]    -- simply the (B, M, Lt) matrix of the polyphase GEMM.
// The host passes a ConvPlan whose Lin / Lvalid / Lout / Cout describe these copies.
__device__ __forceinline__ int floordiv_bw(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

__global__ __launch_bounds__(256) void phase_split_rows_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                               int64_t rows, int L, int valid, int Lp, int s) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;      // one element of a phase row, all phases
    if (e >= rows * Lp) return;
    const int64_t row = e / Lp;
    const int u = int(e - row * Lp);
    for (int rho = 0; rho < s; ++rho) {
        const int xi = u * s + rho;
        dst[(row * s + rho) * Lp + u] = xi < valid ? src[row * L + xi] : 0.f;
    }
}

// ---- pieces shared by the 1-D and the 2-D barrier-free kernels -----------------------------------------------
// MFMA operands of one item out of the wave's swizzled DMA buffer: row li of every block, chunks 4 lh .. 4 lh + 3
template <int MW, int NW>
__device__ __forceinline__ void direct_read_lds(const float *wbuf, int li, int lh, f32x4 (&A)[MW][4], f32x4 (&Bv)[NW][4]) {
    const int rd_off = (li >> 3) * 256 + (li & 7) * 32;          // floats: this lane's row in a block
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            A[i][e] = *reinterpret_cast<const f32x4 *>(wbuf + i * 1024 + rd_off + 4 * ((4 * lh + e) ^ (li & 7)));
#pragma unroll
    for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            Bv[k][e] = *reinterpret_cast<const f32x4 *>(wbuf + (MW + k) * 1024 + rd_off + 4 * ((4 * lh + e) ^ (li & 7)));
}

// the 16 MW NW MFMAs of one item (k-step = position 16 lh + 4 e + c of both operands) + the bias row sums
template <int MW, int NW>
__device__ __forceinline__ void direct_compute(f32x16 (&acc)[MW][NW], float (&bsum)[MW], const f32x4 (&A)[MW][4],
                                               const f32x4 (&Bv)[NW][4], bool do_bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int k = 0; k < NW; ++k)
                    acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i][e][c], Bv[k][e][c], acc[i][k], 0, 0, 0);
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[i] += (A[i][e][0] + A[i][e][1]) + (A[i][e][2] + A[i][e][3]);
    }
}

// End of a workgroup: the WK waves that shared a tile add their partials through LDS in a fixed order (wave 1, 2, 3
// onto wave 0; the exchange reuses the DMA buffers), then the tile and the bias row sums go to the workspace.
template <int MW, int NW, int WM, int WN>
__device__ __forceinline__ void direct_finish(f32x16 (&acc)[MW][NW], float (&bsum)[MW], float *dma_buf, int wk, int wr,
                                              int lane, int n_base, int m_base, int NK, int M, float *__restrict__ part,
                                              float *__restrict__ bias_part, bool do_bias, int oslice) {
    constexpr int WK = 4 / (WM * WN);
    const int li = lane & 31, lh = lane >> 5, wm = wr / WN, wn = wr % WN;
    if constexpr (WK > 1) {
        constexpr int PER = (MW * NW * 16 + MW) * 64;
        static_assert((WK - 1) * WM * WN * PER <= 4 * (MW + NW) * 1024, "the exchange reuses the DMA buffers");
        float *red = dma_buf;
        __syncthreads();      // every wave is done with its DMA buffer
        float *mine = red + ((wk > 0 ? wk - 1 : 0) * WM * WN + wr) * PER + lane;
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < MW; ++i) {
#pragma unroll
                for (int k = 0; k < NW; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[((i * NW + k) * 16 + r) * 64] = acc[i][k][r];
                mine[(MW * NW * 16 + i) * 64] = bsum[i];
            }
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int w = 1; w < WK; ++w) {
            const float *src = red + ((w - 1) * WM * WN + wr) * PER + lane;
#pragma unroll
            for (int i = 0; i < MW; ++i) {
#pragma unroll
                for (int k = 0; k < NW; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][k][r] += src[((i * NW + k) * 16 + r) * 64];
                bsum[i] += src[(MW * NW * 16 + i) * 64];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = n_base + (wn * NW + k) * 32 + li;
        if (n >= NK) continue;
        float *dst = part + (size_t(oslice) * NK + n) * M;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + (wm * MW + i) * 32 + acc_row(r, lh);
                if (m < M) dst[m] = acc[i][k][r];
            }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
            const int m = m_base + (wm * MW + i) * 32 + li;
            if (lh == 0 && m < M) bias_part[size_t(oslice) * M + m] = tot;
        }
    }
}

template <int MW, int NW, int WM, int WN, bool PHASES = false>   // PHASES: x is a phase-split copy (sp_arg > 1)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void conv_bwd_weight_direct_kernel(ConvPlan p, int sp_arg, const float *__restrict__ x,
                                                                     const float *__restrict__ dy,
                                                                     float *__restrict__ part,
                                                                     float *__restrict__ bias_part) {
    constexpr int WK = 4 / (WM * WN), BM = 32 * MW * WM, BN = 32 * NW * WN, T = 32;
    static_assert(WK * WM * WN == 4, "4 waves");
    const int sp = PHASES ? sp_arg : 1;   // (a compile-time 1 keeps the stride-1 instantiation's code as it was)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the item loop's branches are wave-uniform
    const int li = lane & 31, lh = lane >> 5;
    const int wk = wave / (WM * WN), wr = wave % (WM * WN), wm = wr / WN, wn = wr % WN;
    const int NK = p.Cin * p.J;
    const int n_base = blockIdx.x * BN, m_base = blockIdx.y * BM;
    const int slice = blockIdx.z * WK + wk, n_slices = gridDim.z * WK;   // contraction slices (waves); one partial tile per workgroup

    int aoff[MW], brow[NW], bshift[NW];
#pragma unroll
    for (int i = 0; i < MW; ++i) aoff[i] = min(m_base + (wm * MW + i) * 32 + li, p.M - 1) * p.Lout;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = min(n_base + (wn * NW + k) * 32 + li, NK - 1);
        const int ci = n / p.J, j = n - ci * p.J, a = floordiv_bw(j * p.d - p.P, sp);
        brow[k] = (ci * sp + (j * p.d - p.P - a * sp)) * p.Lin;
        bshift[k] = a;
    }
    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
    float bsum[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) bsum[i] = 0.f;
    const bool do_bias = bias_part != nullptr && blockIdx.x == 0 && wn == 0;

    const int chunks = (p.Lt + T - 1) / T;
    const int items = p.B * chunks;
    const int tmin_shift = floordiv_bw(-p.P, sp), tmax_shift = floordiv_bw((p.J - 1) * p.d - p.P, sp);   // extreme tap shifts

    // interior chunks (all 32 positions and all taps inside the row) go through the wave's LDS buffer:
    // DMA instruction i of a block carries rows 8 i .. 8 i + 7, lane l -> row 8 i + (l >> 3), slot l & 7
    extern __shared__ __attribute__((aligned(16))) float dma_buf[];
    float *wbuf = dma_buf + wave * ((MW + NW) * 1024);          // (MW + NW) blocks of 32 rows x 32 floats
    const int dr = lane >> 3, dchunk = (lane & 7) ^ dr;          // this lane's row within an instruction, global chunk
    int adma[MW][4], bdma[NW][4];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            adma[i][v] = min(m_base + (wm * MW + i) * 32 + 8 * v + dr, p.M - 1) * p.Lout + 4 * dchunk;
#pragma unroll
    for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = min(n_base + (wn * NW + k) * 32 + 8 * v + dr, NK - 1);
            const int ci = n / p.J, j = n - ci * p.J, a = floordiv_bw(j * p.d - p.P, sp);
            bdma[k][v] = (ci * sp + (j * p.d - p.P - a * sp)) * p.Lin + a + 4 * dchunk;
        }
    auto dma = [&](int item) {
        const int b = item / chunks, tw = (item - b * chunks) * T;
        const float *dyb = dy + size_t(b) * p.Cout * p.Lout + tw;
        const float *xb = x + size_t(b) * p.Cin * sp * p.Lin + tw;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dyb + adma[i][v]),
                                                 (__attribute__((address_space(3))) void *)(wbuf + i * 1024 + v * 256), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xb + bdma[k][v]),
                                                 (__attribute__((address_space(3))) void *)(wbuf + (MW + k) * 1024 + v * 256), 16, 0, 0);
    };
    // first / last chunks of a row: element by element on clamped addresses, masked to zero afterwards
    // (bitwise AND: a select would be turned back into predicated loads, each with its own wait)
    auto load_edge = [&](f32x4 (&A)[MW][4], f32x4 (&Bv)[NW][4], int item) {
        const int b = item / chunks, t0 = (item - b * chunks) * T + 16 * lh;
        const float *dyb = dy + size_t(b) * p.Cout * p.Lout;
        const float *xb = x + size_t(b) * p.Cin * sp * p.Lin;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = t0 + e;
                const unsigned v = __float_as_uint(dyb[aoff[i] + min(t, p.Lt - 1)]);
                A[i][e >> 2][e & 3] = __uint_as_float(v & (t < p.Lt ? ~0u : 0u));
            }
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xi = t0 + e + bshift[k];
                const unsigned v = __float_as_uint(xb[brow[k] + min(max(xi, 0), p.Lvalid - 1)]);
                Bv[k][e >> 2][e & 3] = __uint_as_float(v & ((xi >= 0 && xi < p.Lvalid) ? ~0u : 0u));
            }
    };
    auto compute = [&](const f32x4 (&A)[MW][4], const f32x4 (&Bv)[NW][4]) { direct_compute<MW, NW>(acc, bsum, A, Bv, do_bias); };

    // A slice owns a CONTIGUOUS range of items (every row is streamed front to back).  Runs of interior chunks:
    // wait for the DMA of item n, pull its operands into registers, start the DMA of item n+1 (unconditional, the
    // index is clamped to the run), then the MFMAs of item n.  (Sending the edge chunks through the DMA as well and
    // masking them in registers costs more in the common path -- registers, a branch in the loop -- than it saves.)
    f32x4 A0[MW][4], B0[NW][4];
    const int per = (items + n_slices - 1) / n_slices;
    int item = slice * per;
    const int end = min(items, item + per);
    const int c_lo = (max(-tmin_shift, 0) + T - 1) / T;                                    // first interior chunk of a row
    const int c_hi = max(0, min(p.Lt / T, (p.Lvalid - max(tmax_shift, 0)) / T));            // one past the last
    while (item < end) {
        const int c = item % chunks;
        if (c < c_lo || c >= c_hi) {
            load_edge(A0, B0, item);
            compute(A0, B0);
            ++item;
            continue;
        }
        const int run = min(end - item, c_hi - c), last = item + run - 1;
        dma(item);
        for (int n = 0; n < run; ++n) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA has landed (wave-private buffer: no barrier)
            direct_read_lds<MW, NW>(wbuf, li, lh, A0, B0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // operands are in registers: the buffer may be overwritten
            dma(min(item + n + 1, last));
            compute(A0, B0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the clamped extra DMA of the last step)
        item += run;
    }
    direct_finish<MW, NW, WM, WN>(acc, bsum, dma_buf, wk, wr, lane, n_base, m_base, NK, p.M, part, bias_part, do_bias, blockIdx.z);
}

// out[e] = sum over slices of part[slice][e], in a fixed order: 4 slice groups (the 4 waves of a
// block) sum every 4th slice, then the groups are added 0+1+2+3 -- deterministic, and 4x shorter
// dependent load chains than one thread per element.
__global__ __launch_bounds__(256) void bwd_slice_reduce_kernel(const float *__restrict__ part, int n_slices,
                                                               int64_t n, float *__restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t e = int64_t(blockIdx.x) * 64 + lane;
    float s = 0.f;
    if (e < n)
        for (int k = grp; k < n_slices; k += 4) s += part[size_t(k) * n + e];
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && e < n) out[e] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// The same sum for short outputs (bias gradients: n <= a few hundred elements, thousands of slices): one thread per
// element leaves a handful of waves walking a chain of n_slices / 4 dependent loads (43 us for a 256-element bias).
// Here a block takes 16 elements x 16 slice groups (group g sums slices g, g + 16, ...; the groups are added as a
// fixed binary tree), so the chains are 4x shorter and there are 4x the blocks.
__global__ __launch_bounds__(256) void bwd_slice_reduce_short_kernel(const float *__restrict__ part, int n_slices,
                                                                     int64_t n, float *__restrict__ out) {
    __shared__ float red[16][17];
    const int le = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t e = int64_t(blockIdx.x) * 16 + le;
    float s0 = 0.f, s1 = 0.f;
    if (e < n) {
        int k = grp;
        for (; k + 16 < n_slices; k += 32) { s0 += part[size_t(k) * n + e]; s1 += part[size_t(k + 16) * n + e]; }
        if (k < n_slices) s0 += part[size_t(k) * n + e];
    }
    red[grp][le] = s0 + s1;
    __syncthreads();
    for (int w = 8; w >= 1; w >>= 1) {
        if (grp < w) red[grp][le] += red[grp + w][le];
        __syncthreads();
    }
    if (grp == 0 && e < n) out[e] = red[0][le];
}

static void launch_slice_reduce(const float *part, int n_slices, int64_t n, float *out, hipStream_t st) {
    if (n <= 4096 && n_slices >= 64)
        hipLaunchKernelGGL(bwd_slice_reduce_short_kernel, dim3((unsigned)ceil_div64(n, 16)), dim3(256), 0, st, part, n_slices, n, out);
    else
        hipLaunchKernelGGL(bwd_slice_reduce_kernel, dim3((unsigned)ceil_div64(n, 64)), dim3(256), 0, st, part, n_slices, n, out);
}

// One block per dim-0 row r of the torch weight: fold dWp back to dw_r, then the weight-norm chain rule.
__global__ __launch_bounds__(256) void bwd_weight_unpack_kernel(const float *__restrict__ dwp,
                                                                const float *__restrict__ v,
                                                                const float *__restrict__ g, float *__restrict__ dv,
                                                                float *__restrict__ dg, int kind, int Cin, int Cout,
                                                                int K, int q, int J, int P, int up) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const bool transposed = kind == AGX_CONV_TRANSPOSED;
    const int inner = (transposed ? Cout : Cin) * K;
    const int M = q * Cout;
    const float *vr = v + size_t(r) * inner;
    float *dvr = dv + size_t(r) * inner;
    auto dw_at = [&](int e) -> float {  // gradient w.r.t. the folded weight element (r, e)
        const int o = e / K, k = e - o * K;  // o = ci (normal layouts) or co (transposed layout)
        if (kind == AGX_CONV_CAUSAL || kind == AGX_CONV_SAME || kind == AGX_CONV_PADDED)
            return dwp[(size_t(o) * J + k) * M + r];
        if (kind == AGX_CONV_UPSAMPLE) {
            const int pl = (K - 1) / 2;
            float s = 0.f;
            for (int ph = 0; ph < q; ++ph) {
                const int j = floordiv_bw(ph + k - pl, up) + P;
                s += dwp[(size_t(o) * J + j) * M + r * q + ph];
            }
            return s;
        }
        // transposed: r = ci, o = co; tap k = ph + up*(J-1-j)
        const int ph = k % up, j = J - 1 - k / up;
        return dwp[(size_t(r) * J + j) * M + o * q + ph];
    };
    float dot = 0.f, nrm = 0.f;
    for (int e = threadIdx.x; e < inner; e += 256) {
        const float w = dw_at(e), vv = vr[e];
        dot = fmaf(w, vv, dot);
        nrm = fmaf(vv, vv, nrm);
    }
    for (int pass = 0; pass < 2; ++pass) {
        float val = pass == 0 ? dot : nrm;
        for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
        __syncthreads();
        val = (red[0] + red[1]) + (red[2] + red[3]);
        if (pass == 0) dot = val; else nrm = val;
    }
    if (!g) {  // plain weight: dv is the weight gradient itself
        for (int e = threadIdx.x; e < inner; e += 256) dvr[e] = dw_at(e);
        return;
    }
    const float norm = sqrtf(nrm), scale = g[r] / norm;
    if (threadIdx.x == 0) dg[r] = dot / norm;
    for (int e = threadIdx.x; e < inner; e += 256) dvr[e] = scale * (dw_at(e) - vr[e] * (dot / nrm));
}

// db[co] = sum over the output phases of the slice-reduced row sums
__global__ __launch_bounds__(256) void bwd_bias_fold_kernel(const float *__restrict__ rowsum, int q, int Cout,
                                                            float *__restrict__ db) {
    const int co = blockIdx.x * 256 + threadIdx.x;
    if (co >= Cout) return;
    float s = 0.f;
    for (int ph = 0; ph < q; ++ph) s += rowsum[co * q + ph];
    db[co] = s;
}

// ------------------------------------------------------------------------------------------------
// Conv2d weight gradient (discriminator.py:101-114, 150-167):
//   dW[co][n = (ci, dh, dw)] = sum_{b, t, f} dy[b, co, t, f] x[b, ci, t sh + dh - ph, f sw + dw - pw].
// Same GEMM as above with the contraction tile of 64 positions laid out as R output rows x WF output
// columns (a whole row when it fits), the x operand staged as one (R-1) sh + kh by (WF-1) sw + kw patch per
// channel -- so the tile stays full when the frequency axis is short.
struct Bw2dGeom {
    int B, Cin, Cout, Hin, Win, Hout, Wout, kh, kw, sh, sw, ph, pw;
    int R, WF, RH, SW, span, n_chan, n_slices;
    int prec;   // 1: bf16x3 contraction (AGX_IMPL_MFMA_BF16X3)
    int Wp;     // direct kernel: width of a column-phase plane of x (= Win when sw == 1)
    int xcd;    // 1: XCD-aware block order (xcd_block_id)
    // narrow feature maps (Wout < 32) on the shared kernel: x and dy are first copied into ZERO-PADDED, phase-split planes whose
    // flattened rows make every item 32 consecutive positions again (prepad_x_kernel / prepad_dy_kernel below):
    //   XP[rho_h sw + rho_w][b, ci][rr Wp + cc] = x[b, ci, sh (rr + amin) + rho_h, sw (cc + bmin) + rho_w]   (0 outside the image)
    //   DYP[b, co][t Wp + cc]                  = dy[b, co, t, cc]  (0 for cc >= Wout and behind the last row, up to pp_lpr)
    // with dh - ph = sh ah + rho_h, dw - pw = sw aw + rho_w, amin / bmin the smallest ah / aw, Wp = Wout + (bmax - bmin): the
    // window of B row (ci, dh, dw) is XP[phase][ci] shifted by (ah - amin) Wp + (aw - bmin) -- no masks, no padding rows, no edges.
    int prepad;          // 1: the kernel reads such planes (one "row" of pp_lpr positions per image)
    int pp_lpr, pp_hwi;  // positions per DYP plane (a multiple of 32) / floats per XP plane
    int pp_amin, pp_bmin;
};

template <int MW, int NW, int WM, int WN, int PREC = 0>   // PREC 1: bf16x3 contraction (mfma_tile.hpp), both operands split in registers
__global__ __launch_bounds__(256) void conv2d_bwd_weight_kernel(Bw2dGeom g, const float *__restrict__ x,
                                                                const float *__restrict__ dy,
                                                                float *__restrict__ part,
                                                                float *__restrict__ bias_part) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = 32 * MW * WM, BN = 32 * NW * WN;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *dys = sm;                          // [BM][BW_TS]
    float *xs = sm + BM * BW_TS;              // [n_chan][span]
    int *kofft = reinterpret_cast<int *>(xs + g.n_chan * g.span);  // [BW_T] patch offset of contraction index k
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int KK = g.kh * g.kw, NK = g.Cin * KK, M = g.Cout;
    const BlockId bid = xcd_block_id(g.xcd != 0);
    const int n_base = bid.x * BN, m_base = bid.y * BM, slice = bid.z;
    const int ci_first = n_base / KK;
    const int npos = g.R * g.WF;              // contraction positions per tile (<= BW_T)
    if (tid < BW_T) {
        const int r = tid / g.WF, fc = tid - r * g.WF;
        kofft[tid] = tid < npos ? (r * g.sh) * g.SW + fc * g.sw : 0;
    }
    const int my_k = tid & 63, my_rr = my_k / g.WF, my_fc = my_k - my_rr * g.WF;   // this thread's dy-tile column
    const float inv_span = 1.f / float(g.span), inv_sw = 1.f / float(g.SW);
    int boff[NW];
    bool nvalid[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = n_base + (wn * NW + k) * 32 + li;
        nvalid[k] = n < NK;
        const int nc = min(n, NK - 1);
        const int ci = nc / KK, rem = nc - ci * KK;
        const int dh = rem / g.kw, dw = rem - dh * g.kw;
        boff[k] = (ci - ci_first) * g.span + dh * g.SW + dw;
    }
    int arow[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) arow[i] = ((wm * MW + i) * 32 + li) * BW_TS;

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
    constexpr int TPR = 256 / BM, CPT = BW_T / TPR;
    float bsum = 0.f;
    const bool do_bias = bias_part != nullptr && bid.x == 0;

    const int nft = (g.Wout + g.WF - 1) / g.WF, nrg = (g.Hout + g.R - 1) / g.R;
    const int items = g.B * nrg * nft;
    for (int item = slice; item < items; item += g.n_slices) {
        int it = item;
        const int ft = it % nft;
        it /= nft;
        const int rg = it % nrg, b = it / nrg;
        const int trow0 = rg * g.R, f0 = ft * g.WF;
        __syncthreads();
        // Staging: all loads of a batch are issued on clamped addresses before any is used (a conditional
        // load is waited for one by one), zeros are selected afterwards.
        // dy tile: row r <-> co = m_base + r, column k = rr * WF + fc <-> dy[b, co, trow0 + rr, f0 + fc];
        // a thread's column k = tid % 64 is the same for all its elements (256 % 64 == 0)
        {
            const int t = trow0 + my_rr, f = f0 + my_fc;
            const bool pos_ok = my_k < npos && t < g.Hout && f < g.Wout;
            const size_t pos_off = size_t(min(t, g.Hout - 1)) * g.Wout + min(f, g.Wout - 1);
            const float *dyb = dy + size_t(b) * g.Cout * g.Hout * g.Wout + pos_off;
            constexpr int NB = BM * BW_T / 256;   // elements per thread (rows tid / 64 + 4 u)
#pragma unroll
            for (int u0 = 0; u0 < NB; u0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int co = m_base + (tid >> 6) + 4 * (u0 + u);
                    v[u] = dyb[size_t(min(co, M - 1)) * g.Hout * g.Wout];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = (tid >> 6) + 4 * (u0 + u);
                    dys[r * BW_TS + my_k] = (pos_ok && m_base + r < M) ? v[u] : 0.f;
                }
            }
        }
        // x patches of channels ci_first .. ci_first + n_chan - 1
        {
            const int row0 = trow0 * g.sh - g.ph, col0 = f0 * g.sw - g.pw;
            const float *xb = x + size_t(b) * g.Cin * g.Hin * g.Win;
            const int total = g.n_chan * g.span;
            for (int e0 = tid; e0 < total; e0 += 256 * 8) {
                float v[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * 256, total - 1);
                    const int c = int((float(e) + 0.5f) * inv_span), i = e - c * g.span;   // exact: e < 2^20
                    const int rr = int((float(i) + 0.5f) * inv_sw), cc = i - rr * g.SW;
                    const int ch = ci_first + c, gr = row0 + rr, gc = col0 + cc;
                    ok[u] = ch < g.Cin && gr >= 0 && gr < g.Hin && gc >= 0 && gc < g.Win;
                    v[u] = xb[(size_t(min(ch, g.Cin - 1)) * g.Hin + min(max(gr, 0), g.Hin - 1)) * g.Win +
                              min(max(gc, 0), g.Win - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * 256 < total) xs[e0 + u * 256] = ok[u] ? v[u] : 0.f;
            }
        }
        __syncthreads();
        if (do_bias) {
            const float *row = dys + (tid / TPR) * BW_TS + (tid % TPR) * CPT;
#pragma unroll
            for (int c = 0; c < CPT; ++c) bsum += row[c];
        }
        if (PREC == 1) {
#pragma unroll
            for (int kb = 0; kb < BW_T / 16; ++kb) {     // K = 16 blocks: this lane's positions 16 kb + 8 lh + 0..7
                const int k0 = 16 * kb + 8 * lh;
                int ko[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ko[e] = kofft[k0 + e];
                bf16x8 aq[3][MW], bq[3][NW];
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    float xq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xq[e] = dys[arow[i] + k0 + e];
                    split3(xq, aq[0][i], aq[1][i], aq[2][i]);
                }
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    float xq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xq[e] = nvalid[k] ? xs[boff[k] + ko[e]] : 0.f;
                    split3(xq, bq[0][k], bq[1][k], bq[2][k]);
                }
                mfma_block_bf<MW, NW>(acc, aq, bq);
            }
        } else
#pragma unroll 4
        for (int ks = 0; ks < BW_T / 2; ++ks) {
            const int tt = 2 * ks + lh;
            const int ko = kofft[tt];
            float a[MW], bv[NW];
#pragma unroll
            for (int i = 0; i < MW; ++i) a[i] = dys[arow[i] + tt];
#pragma unroll
            for (int k = 0; k < NW; ++k) bv[k] = nvalid[k] ? xs[boff[k] + ko] : 0.f;
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int k = 0; k < NW; ++k)
                    acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[k], acc[i][k], 0, 0, 0);
        }
    }
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = n_base + (wn * NW + k) * 32 + li;
        if (n >= NK) continue;
        float *dst = part + (size_t(slice) * NK + n) * M;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + (wm * MW + i) * 32 + acc_row(r, lh);
                if (m < M) dst[m] = acc[i][k][r];
            }
    }
    if (do_bias) {
#pragma unroll
        for (int off = 1; off < TPR; off <<= 1) bsum += __shfl_xor(bsum, off);
        const int m = m_base + tid / TPR;
        if (tid % TPR == 0 && m < M) bias_part[size_t(slice) * M + m] = bsum;
    }
}

// Stride-1 "same" layers (the 3x3 convs and the 7x7 first conv of the STFT discriminators): the 2-D form of
// conv_bwd_weight_direct_kernel -- no staging phase, no workgroup barrier, wave-private LDS-DMA buffers with
// XOR-swizzled 16-byte chunks.  An item is 32 consecutive columns of one output row; B row n = (ci, dh, dw) is
// the window x[ci, t + dh - ph, f0 + dw - pw + 0..31].  Rows in the vertical padding are DMA'd from a page of
// zeros (an address select, no branch); a window may run up to pw elements over the end of its image row, into
// the neighbouring row of the same tensor -- those elements are masked in registers, and only where the DMA could
// leave the tensor itself (first rows of batch element 0, last rows of element B - 1) the operands are loaded
// element by element.
// Column-strided layers (sw = 2: the (3,4)/(4,4) convs) read x through its sw column-phase planes
// xs[rho][b, ci, r, f] = x[b, ci, r, f sw + rho] (deinterleave_cols_kernel, one pass over x): with dw - pw = a sw + rho the
// window of B row (ci, dh, dw) is xs[rho][ci, t sh + dh - ph, f0 + a + 0..31] -- contiguous again; a row stride sh only
// enters the row address.
__device__ float bw_zero_page[64];

__global__ __launch_bounds__(256) void deinterleave_cols_kernel(const float *__restrict__ x, float *__restrict__ xs,
                                                                int64_t rows, int Win, int Wp, int sw) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;      // one element of a plane row, all phases
    if (e >= rows * Wp) return;
    const int64_t row = e / Wp;
    const int f = int(e - row * Wp);
    for (int rho = 0; rho < sw; ++rho) {
        const int xi = f * sw + rho;
        xs[(int64_t(rho) * rows + row) * Wp + f] = xi < Win ? x[row * Win + xi] : 0.f;
    }
}

template <int MW, int NW, int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void conv2d_bwd_weight_direct_kernel(
    Bw2dGeom g, const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part,
    float *__restrict__ bias_part) {
    constexpr int WK = 4 / (WM * WN), BM = 32 * MW * WM, BN = 32 * NW * WN, T = 32;
    static_assert(WK * WM * WN == 4, "4 waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wk = wave / (WM * WN), wr = wave % (WM * WN), wm = wr / WN, wn = wr % WN;
    const int KK = g.kh * g.kw, NK = g.Cin * KK, M = g.Cout;
    const int Wp = g.Wp, HWo = g.Hout * g.Wout, HWi = g.Hin * Wp, FC = g.Wout / T;
    const int plane = g.B * g.Cin * HWi;          // floats per column-phase plane
    const BlockId bid = xcd_block_id(g.xcd != 0);
    const int n_base = bid.x * BN, m_base = bid.y * BM;
    const int slice = bid.z * WK + wk, n_slices = gridDim.z * WK;

    // the MFMA rows of this lane (edge path, masks)
    int aoff[MW], brow[NW], bdh[NW], bdw[NW];
#pragma unroll
    for (int i = 0; i < MW; ++i) aoff[i] = min(m_base + (wm * MW + i) * 32 + li, M - 1) * HWo;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = min(n_base + (wn * NW + k) * 32 + li, NK - 1);
        const int ci = n / KK, rem = n - ci * KK, dh = rem / g.kw;
        const int dwp = rem - dh * g.kw - g.pw, a = floordiv_bw(dwp, g.sw);
        brow[k] = (dwp - a * g.sw) * plane + ci * HWi;      // phase plane + channel
        bdh[k] = dh - g.ph;
        bdw[k] = a;                                           // column shift inside the plane
    }
    // the DMA rows of this lane: instruction v of a block carries rows 8 v .. 8 v + 7, lane l -> row 8 v + (l >> 3)
    extern __shared__ __attribute__((aligned(16))) float dma_buf[];
    float *wbuf = dma_buf + wave * ((MW + NW) * 1024);
    const int dr = lane >> 3, dchunk = (lane & 7) ^ dr;
    int adma[MW][4], bdma[NW][4], bdmah[NW];        // bdmah: the four dh - ph of a block, one byte each (+ 64)
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) adma[i][v] = min(m_base + (wm * MW + i) * 32 + 8 * v + dr, M - 1) * HWo + 4 * dchunk;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        bdmah[k] = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = min(n_base + (wn * NW + k) * 32 + 8 * v + dr, NK - 1);
            const int ci = n / KK, rem = n - ci * KK, dh = rem / g.kw, dwp = rem - dh * g.kw - g.pw;
            const int a = floordiv_bw(dwp, g.sw);
            bdma[k][v] = (dwp - a * g.sw) * plane + ci * HWi + (dh - g.ph) * Wp + a + 4 * dchunk;
            bdmah[k] |= (dh - g.ph + 64) << (8 * v);
        }
    }
    const float *zsrc = bw_zero_page + 4 * (lane & 7);

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
    float bsum[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) bsum[i] = 0.f;
    const bool do_bias = bias_part != nullptr && bid.x == 0 && wn == 0;

    const int rows = g.B * g.Hout, items = rows * FC;
    auto dma = [&](int item) {
        const int row = item / FC, fc = item - row * FC, b = row / g.Hout, t = row - b * g.Hout;
        const float *dyb = dy + size_t(b) * M * HWo + t * g.Wout + fc * T;
        const float *xb = x + size_t(b) * g.Cin * HWi + t * g.sh * Wp + fc * T;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dyb + adma[i][v]),
                                                 (__attribute__((address_space(3))) void *)(wbuf + i * 1024 + v * 256), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = t * g.sh + ((bdmah[k] >> (8 * v)) & 255) - 64;
                const float *src = (r >= 0 && r < g.Hin) ? xb + bdma[k][v] : zsrc;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(wbuf + (MW + k) * 1024 + v * 256), 16, 0, 0);
            }
    };
    // first / last chunk of an image row: the (at most 4) window elements in the horizontal padding
    auto mask_cols = [&](f32x4 (&Bv)[NW][4], int fc) {
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int x0 = fc * T + 16 * lh + c + bdw[k], x1 = x0 + 12;
                Bv[k][0][c] = __uint_as_float(__float_as_uint(Bv[k][0][c]) & ((x0 >= 0 && x0 < Wp) ? ~0u : 0u));
                Bv[k][3][c] = __uint_as_float(__float_as_uint(Bv[k][3][c]) & ((x1 >= 0 && x1 < Wp) ? ~0u : 0u));
            }
    };
    auto load_edge = [&](f32x4 (&A)[MW][4], f32x4 (&Bv)[NW][4], int item) {
        const int row = item / FC, fc = item - row * FC, b = row / g.Hout, t = row - b * g.Hout;
        const float *dyb = dy + size_t(b) * M * HWo + t * g.Wout + fc * T + 16 * lh;
        const float *xb = x + size_t(b) * g.Cin * HWi;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) A[i][e] = *reinterpret_cast<const f32x4 *>(dyb + aoff[i] + 4 * e);
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int r = t * g.sh + bdh[k];
            const bool rok = r >= 0 && r < g.Hin;
            const float *xr = xb + brow[k] + min(max(r, 0), g.Hin - 1) * Wp;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xi = fc * T + 16 * lh + e + bdw[k];
                const unsigned v = __float_as_uint(xr[min(max(xi, 0), Wp - 1)]);
                Bv[k][e >> 2][e & 3] = __uint_as_float(v & ((rok && xi >= 0 && xi < Wp) ? ~0u : 0u));
            }
        }
    };
    auto compute = [&](const f32x4 (&A)[MW][4], const f32x4 (&Bv)[NW][4]) { direct_compute<MW, NW>(acc, bsum, A, Bv, do_bias); };

    f32x4 A0[MW][4], B0[NW][4];
    const int per = (items + n_slices - 1) / n_slices;
    int item = slice * per;
    const int end = min(items, item + per);
    // where a DMA window could leave the tensor: chunk 0 of the first ph + 1 rows of batch element 0 (it starts pw
    // elements before its row), the last chunk of the last kh - ph rows of element B - 1
    const int head_rows = min(g.ph + 1, rows), tail_row0 = max(rows - (g.kh - g.ph), 0);   // (supersets when sh > 1)
    while (item < end) {
        const int row = item / FC, fc = item - row * FC;
        const bool head = row < head_rows, tail = row >= tail_row0;
        const bool unsafe = (head && fc == 0) || (tail && fc == FC - 1);
        const int run_end = tail ? row * FC + FC - 1 : (head ? (row + 1) * FC : tail_row0 * FC);
        const int run = unsafe ? 0 : min(end, run_end) - item;
        if (run <= 0) {
            load_edge(A0, B0, item);
            compute(A0, B0);
            ++item;
            continue;
        }
        const int last = item + run - 1;
        dma(item);
        for (int n = 0; n < run; ++n) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            direct_read_lds<MW, NW>(wbuf, li, lh, A0, B0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma(min(item + n + 1, last));
            const int fcn = (item + n) % FC;
            if (fcn == 0 || fcn == FC - 1) mask_cols(B0, fcn);
            compute(A0, B0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        item += run;
    }
    direct_finish<MW, NW, WM, WN>(acc, bsum, dma_buf, wk, wr, lane, n_base, m_base, NK, M, part, bias_part, do_bias, bid.z);
}

// The same item stream with the operands SHARED by the workgroup (knob dw2_shared, default on).  In the kernel above every
// wave DMAs its own A and B blocks: on a 2 x 2 wave grid each block is fetched twice, 64 KB of L2 -> LDS traffic per item
// of 64 MFMAs per wave -- ~10 TB/s at the full MFMA rate, which is what holds that kernel at 80-93 TFLOP/s.  Here the
// (BM + BN) / 32 blocks of an item are fetched once (each wave issues a quarter of the DMA instructions) into one of two
// LDS slots, and one barrier per item hands the slot over: item n + 1 streams in while item n is multiplied, exactly the
// hand-over of conv_p.hip's two-slot ring.  Items whose DMA window could leave the tensor (head / tail rows) take the
// element-by-element path, workgroup-uniformly.
// probe build (-DAGX_STAMPS, tools/dw_stamps.py): per wave, the cycles of the item loop by segment, summed over the wave's items
#ifdef AGX_STAMPS
#define DW_STAMP_START() do { __builtin_amdgcn_sched_barrier(0); dws_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DW_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); dws[k] += tn_ - dws_prev; dws_prev = tn_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define DW_STAMP_COUNT(n) do { dws[4] += (unsigned long long)(n); } while (0)
#define DW_STAMP_WRITE() do { if (lane == 0) { const int L_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
        if ((L_ * 4 + wave) * 8 + 8 <= (1 << 16)) for (int k_ = 0; k_ < 5; ++k_) g_stamps[(L_ * 4 + wave) * 8 + k_] = dws[k_]; } } while (0)
#else
#define DW_STAMP_START() ((void)0)
#define DW_STAMP(k) ((void)0)
#define DW_STAMP_COUNT(n) ((void)0)
#define DW_STAMP_WRITE() ((void)0)
#endif
// PREC 1: the bf16x3 contraction (mfma_tile.hpp): both operands are activations, so each lane splits the 32 + 32 values of an item
// into three bf16 pieces in registers -- ~500 vector instructions against 48 bf16 MFMAs (1 536 cycles; the fp32 item: 64 MFMAs = 4 096
// cycles): bound by the split, still ~1.4 x the fp32 kernel.  The contraction order of an MFMA is free as long as both operands
// agree: k-block kb of lane half lh = positions 16 lh + 8 kb + (0 .. 7) -- the registers the fp32 path already holds.
template <int MW, int NW, int WM, int WN, int PREC = 0>
__global__ __launch_bounds__(256, 2) void conv2d_bwd_weight_shared_kernel(Bw2dGeom g, const float *__restrict__ x,
                                                                          const float *__restrict__ dy,
                                                                          float *__restrict__ part,
                                                                          float *__restrict__ bias_part) {
    static_assert(WM * WN == 4, "4 waves, no contraction split");
    constexpr int BM = 32 * MW * WM, BN = 32 * NW * WN, T = 32;
    constexpr int NBA = BM / 32, NBB = BN / 32, NBLK = NBA + NBB;   // 4 KB operand blocks of an item
    constexpr int NI = NBLK;                                        // DMA instructions per wave and item (4 per block / 4 waves)
    constexpr int SLOTF = NBLK * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int KK = g.kh * g.kw, NK = g.Cin * KK, M = g.Cout;
    const bool pp = g.prepad != 0;
    const int Wp = g.Wp, HWo = pp ? g.pp_lpr : g.Hout * g.Wout, HWi = pp ? g.pp_hwi : g.Hin * Wp, FC = (pp ? g.pp_lpr : g.Wout) / T;
    const int Hk = pp ? 1 : g.Hout;                 // item rows per image (prepadded planes: one flattened row)
    const int plane = g.B * g.Cin * HWi;
    const BlockId bid = xcd_block_id(g.xcd != 0);
    const int n_base = bid.x * BN, m_base = bid.y * BM;
    const int slice = bid.z, n_slices = gridDim.z;

    int aoff[MW], brow[NW], bdh[NW], bdw[NW];   // this lane's MFMA rows (edge path, masks)
#pragma unroll
    for (int i = 0; i < MW; ++i) aoff[i] = min(m_base + (wm * MW + i) * 32 + li, M - 1) * HWo;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = min(n_base + (wn * NW + k) * 32 + li, NK - 1);
        const int ci = n / KK, rem = n - ci * KK, dh = rem / g.kw;
        const int dwp = rem - dh * g.kw - g.pw, a = floordiv_bw(dwp, g.sw);
        brow[k] = (dwp - a * g.sw) * plane + ci * HWi;
        bdh[k] = dh - g.ph;
        bdw[k] = a;
    }
    // DMA instruction q = wave + 4 r of an item: block q / 4 (A blocks first), rows 8 (q % 4) .. + 7 of it, lane l -> row
    // 8 (q % 4) + (l >> 3), 16-byte chunk (l & 7) ^ (l >> 3) of the row (the swizzle direct_read_lds undoes)
    extern __shared__ __attribute__((aligned(16))) float dma_buf[];
    const int dr = lane >> 3, dchunk = (lane & 7) ^ dr;
    // this lane's source offset of instruction r as an unsigned BYTE offset from (dy item base) resp. (x item base - BIAS floats):
    // BIAS makes every B offset non-negative (a >= -4, dh - ph >= -ph); the kernel rows dh of the B instructions, a byte each
    const int BIAS = g.ph * Wp + 8;
    unsigned du[NI], dhpack[(NI + 3) / 4];
#pragma unroll
    for (int r = 0; r < (NI + 3) / 4; ++r) dhpack[r] = 0;
#pragma unroll
    for (int r = 0; r < NI; ++r) {
        const int q = wave + 4 * r, blk = q >> 2, v = q & 3;
        if (blk < NBA) {
            du[r] = unsigned(min(m_base + blk * 32 + 8 * v + dr, M - 1) * HWo + 4 * dchunk) * 4u;
        } else {
            const int n = min(n_base + (blk - NBA) * 32 + 8 * v + dr, NK - 1);
            const int ci = n / KK, rem = n - ci * KK, dh = rem / g.kw, dwp = rem - dh * g.kw - g.pw;
            const int a = floordiv_bw(dwp, g.sw);
            if (pp) {      // phase plane (row phase, column phase) + the tap's shift inside the padded plane
                const int ah = floordiv_bw(dh - g.ph, g.sh), rh = (dh - g.ph) - ah * g.sh;
                du[r] = unsigned((rh * g.sw + (dwp - a * g.sw)) * plane + ci * HWi + (ah - g.pp_amin) * Wp + (a - g.pp_bmin) +
                                 4 * dchunk + BIAS) * 4u;
            } else {
                du[r] = unsigned((dwp - a * g.sw) * plane + ci * HWi + (dh - g.ph) * Wp + a + 4 * dchunk + BIAS) * 4u;
            }
            dhpack[r >> 2] |= unsigned(dh) << (8 * (r & 3));
        }
    }
    const float *zsrc = bw_zero_page + 4 * (lane & 7);

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
    float bsum[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) bsum[i] = 0.f;
    const bool do_bias = bias_part != nullptr && bid.x == 0 && wn == 0;

    const int rows = g.B * Hk, items = rows * FC;
    // Fast path (every item whose kh input rows are all inside the image -- all but the first / last rows of an image): the
    // instruction's address is a wave-uniform base (SGPR pair) + this lane's precomputed unsigned 32-bit offset, so an item costs two
    // scalar base updates and NI (m0, global_load_lds saddr) pairs.  The general form below -- a select between the row and a page
    // of zeros per instruction, 64-bit lane addresses -- took ~250 mostly dependent instructions per item: measured 22 % of the
    // kernel (128 -> 128 3 x 3: 101 -> 129 TFLOP/s with the DMA issue removed).
    auto dma_fast = [&](int b, int t, int fc, float *slot) {
        const char *dyb = reinterpret_cast<const char *>(dy + size_t(b) * M * HWo + t * g.Wout + fc * T);
        const char *xb = reinterpret_cast<const char *>(x + size_t(b) * g.Cin * HWi + t * g.sh * Wp + fc * T) - size_t(BIAS) * 4;
#pragma unroll
        for (int r = 0; r < NI; ++r)      // instruction q = wave + 4 r lies in block r: A blocks first
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((r < NBA ? dyb : xb) + du[r]),
                                             (__attribute__((address_space(3))) void *)(slot + (wave + 4 * r) * 256), 16, 0, 0);
    };
    auto dma = [&](int b, int t, int fc, float *slot) {
        if (pp || (t * g.sh - g.ph >= 0 && t * g.sh + g.kh - 1 - g.ph < g.Hin)) return dma_fast(b, t, fc, slot);
        // first / last rows of an image: B rows in the vertical padding come from a page of zeros (an address select per instruction)
        const char *dyb = reinterpret_cast<const char *>(dy + size_t(b) * M * HWo + t * g.Wout + fc * T);
        const char *xb = reinterpret_cast<const char *>(x + size_t(b) * g.Cin * HWi + t * g.sh * Wp + fc * T) - size_t(BIAS) * 4;
#pragma unroll
        for (int r = 0; r < NI; ++r) {
            const char *src;
            if (r < NBA) {
                src = dyb + du[r];
            } else {
                const int rr = t * g.sh + int((dhpack[r >> 2] >> (8 * (r & 3))) & 255u) - g.ph;
                src = (rr >= 0 && rr < g.Hin) ? xb + du[r] : reinterpret_cast<const char *>(zsrc);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + (wave + 4 * r) * 256), 16, 0, 0);
        }
    };
    // operands of this wave's fragments out of a slot: A blocks wm MW + i, B blocks NBA + wn NW + k
    auto read_slot = [&](const float *slot, f32x4 (&A)[MW][4], f32x4 (&Bv)[NW][4]) {
        // position-major: the operands of the item's first MFMAs arrive first, the rest while those MFMAs issue (the DMA goes
        // to the OTHER slot, so nothing has to be in registers before it starts)
        const int rd_off = (li >> 3) * 256 + (li & 7) * 32, rsw = li & 7;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int i = 0; i < MW; ++i)
                A[i][e] = *reinterpret_cast<const f32x4 *>(slot + (wm * MW + i) * 1024 + rd_off + 4 * ((4 * lh + e) ^ rsw));
#pragma unroll
            for (int k = 0; k < NW; ++k)
                Bv[k][e] = *reinterpret_cast<const f32x4 *>(slot + (NBA + wn * NW + k) * 1024 + rd_off +
                                                            4 * ((4 * lh + e) ^ rsw));
        }
    };
    auto mask_cols = [&](f32x4 (&Bv)[NW][4], int fc) {
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int x0 = fc * T + 16 * lh + c + bdw[k], x1 = x0 + 12;
                Bv[k][0][c] = __uint_as_float(__float_as_uint(Bv[k][0][c]) & ((x0 >= 0 && x0 < Wp) ? ~0u : 0u));
                Bv[k][3][c] = __uint_as_float(__float_as_uint(Bv[k][3][c]) & ((x1 >= 0 && x1 < Wp) ? ~0u : 0u));
            }
    };
    auto load_edge = [&](f32x4 (&A)[MW][4], f32x4 (&Bv)[NW][4], int item) {
        const int row = item / FC, fc = item - row * FC, b = row / g.Hout, t = row - b * g.Hout;
        const float *dyb = dy + size_t(b) * M * HWo + t * g.Wout + fc * T + 16 * lh;
        const float *xb = x + size_t(b) * g.Cin * HWi;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) A[i][e] = *reinterpret_cast<const f32x4 *>(dyb + aoff[i] + 4 * e);
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int r = t * g.sh + bdh[k];
            const bool rok = r >= 0 && r < g.Hin;
            const float *xr = xb + brow[k] + min(max(r, 0), g.Hin - 1) * Wp;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xi = fc * T + 16 * lh + e + bdw[k];
                const unsigned v = __float_as_uint(xr[min(max(xi, 0), Wp - 1)]);
                Bv[k][e >> 2][e & 3] = __uint_as_float(v & ((rok && xi >= 0 && xi < Wp) ? ~0u : 0u));
            }
        }
    };
    auto compute = [&](const f32x4 (&A)[MW][4], const f32x4 (&Bv)[NW][4]) {
        if constexpr (PREC == 0) {
            direct_compute<MW, NW>(acc, bsum, A, Bv, do_bias);
        } else {
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};   // mm hl lh hm mh hh (small terms first)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 a3[3][MW];
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    const float xq[8] = {A[i][2 * kb][0], A[i][2 * kb][1], A[i][2 * kb][2], A[i][2 * kb][3],
                                         A[i][2 * kb + 1][0], A[i][2 * kb + 1][1], A[i][2 * kb + 1][2], A[i][2 * kb + 1][3]};
                    split3(xq, a3[0][i], a3[1][i], a3[2][i]);
                }
#pragma unroll
                for (int k = 0; k < NW; ++k) {      // one column block's pieces at a time (12 live registers instead of 12 NW)
                    bf16x8 b3[3];
                    const float xq[8] = {Bv[k][2 * kb][0], Bv[k][2 * kb][1], Bv[k][2 * kb][2], Bv[k][2 * kb][3],
                                         Bv[k][2 * kb + 1][0], Bv[k][2 * kb + 1][1], Bv[k][2 * kb + 1][2], Bv[k][2 * kb + 1][3]};
                    split3(xq, b3[0], b3[1], b3[2]);
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int i = 0; i < MW; ++i)
                            acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[PA[t]][i], b3[PB[t]], acc[i][k], 0, 0, 0);
                }
            }
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsum[i] += (A[i][e][0] + A[i][e][1]) + (A[i][e][2] + A[i][e][3]);
            }
        }
    };

    f32x4 A0[MW][4], B0[NW][4];
#ifdef AGX_STAMPS
    unsigned long long dws[5] = {0, 0, 0, 0, 0}, dws_prev = 0;
#endif
    const int per = (items + n_slices - 1) / n_slices;
    int item = slice * per;
    const int end = min(items, item + per);
    // (prepadded planes are surrounded by zeros and a guard: no row is unsafe)
    const int head_rows = pp ? 0 : min(g.ph + 1, rows), tail_row0 = pp ? rows : max(rows - (g.kh - g.ph), 0);
    while (item < end) {   // (every branch below is workgroup-uniform)
        const int row = item / FC, fc = item - row * FC;
        const bool head = row < head_rows, tail = row >= tail_row0;
        const bool unsafe = (head && fc == 0) || (tail && fc == FC - 1);
        const int run_end = tail ? row * FC + FC - 1 : (head ? (row + 1) * FC : tail_row0 * FC);
        const int run = unsafe ? 0 : min(end, run_end) - item;
        if (run <= 0) {
            load_edge(A0, B0, item);
            compute(A0, B0);
            ++item;
            continue;
        }
        // (b, t, fc) of the item being multiplied and of the one being fetched, advanced without divisions
        int cb = row / Hk, ct = row - cb * Hk, cfc = fc;
        dma(cb, ct, cfc, dma_buf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // item 0 of the run is complete in slot 0
        DW_STAMP_START();
        for (int n = 0; n < run; ++n) {
            float *cur = dma_buf + (n & 1) * SLOTF, *nxt = dma_buf + ((n + 1) & 1) * SLOTF;
            const int fcn = cfc;
            if (n + 1 < run) {                                 // (the last iteration fetches its own item again: harmless)
                if (++cfc == FC) {
                    cfc = 0;
                    if (++ct == Hk) ct = 0, ++cb;
                }
            }
            read_slot(cur, A0, B0);
            dma(cb, ct, cfc, nxt);                             // the other slot: everyone left it at the last barrier
            DW_STAMP(0);
            if (!pp && (fcn == 0 || fcn == FC - 1)) mask_cols(B0, fcn);
            compute(A0, B0);
            DW_STAMP(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the next item has landed
            DW_STAMP(2);
            __syncthreads();                                   // everyone's has, and everyone has read the current slot
            DW_STAMP(3);
        }
        DW_STAMP_COUNT(run);
        item += run;
    }
    DW_STAMP_WRITE();
    direct_finish<MW, NW, WM, WN>(acc, bsum, dma_buf, 0, wave, lane, n_base, m_base, NK, M, part, bias_part, do_bias, bid.z);
}

// Gradient w.r.t. the normalised weight G[co][n] = dwp[n][co] -> dw (torch layout), and per-row <G, W>.
__global__ __launch_bounds__(256) void bwd2d_unpack_kernel(const float *__restrict__ dwp, const float *__restrict__ w,
                                                           float *__restrict__ dw, float *__restrict__ rowdot, int NK,
                                                           int M) {
    __shared__ float red[4];
    const int co = blockIdx.x;
    float dot = 0.f;
    for (int n = threadIdx.x; n < NK; n += 256) {
        const float gv = dwp[size_t(n) * M + co];
        dw[size_t(co) * NK + n] = gv;
        if (w) dot = fmaf(gv, w[size_t(co) * NK + n], dot);
    }
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) rowdot[co] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Spectral-norm chain rule (W_n = W / sigma, sigma = u^T W v with u, v constants):
//   dW = G / sigma - (<G, W> / sigma^2) u v^T
__global__ __launch_bounds__(256) void bwd2d_spectral_kernel(float *__restrict__ dw, const float *__restrict__ rowdot,
                                                             const float *__restrict__ sigma,
                                                             const float *__restrict__ u, const float *__restrict__ v,
                                                             int NK, int M) {
    __shared__ float tot_s;
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < M; ++i) t += rowdot[i];
        tot_s = t;
    }
    __syncthreads();
    const int co = blockIdx.x;
    const float sg = sigma[0], coef = tot_s / (sg * sg) * u[co], inv = 1.f / sg;
    for (int n = threadIdx.x; n < NK; n += 256) {
        const size_t e = size_t(co) * NK + n;
        dw[e] = dw[e] * inv - coef * v[n];
    }
}

// Zero-padded, phase-split copies for narrow feature maps (Bw2dGeom::prepad): one thread per element of the copy.
__global__ __launch_bounds__(256) void prepad_x_kernel(const float *__restrict__ x, float *__restrict__ xp, int64_t nplanes, int Hin,
                                                       int Win, int Hp, int Wp, int hwi, int sh, int sw, int amin, int bmin) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t per_phase = nplanes * hwi;
    if (e >= per_phase * sh * sw) return;
    const int phase = int(e / per_phase);
    const int64_t rem = e - phase * per_phase;
    const int64_t pl = rem / hwi;
    const int idx = int(rem - pl * hwi), rr = idx / Wp, cc = idx - rr * Wp;
    const int r = sh * (rr + amin) + phase / sw, c = sw * (cc + bmin) + phase % sw;
    const float v = x[(pl * Hin + min(max(r, 0), Hin - 1)) * Win + min(max(c, 0), Win - 1)];
    xp[e] = (rr < Hp && r >= 0 && r < Hin && c >= 0 && c < Win) ? v : 0.f;
}
__global__ __launch_bounds__(256) void prepad_dy_kernel(const float *__restrict__ dy, float *__restrict__ dyp, int64_t nplanes, int Hout,
                                                        int Wout, int Wp, int lpr) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= nplanes * lpr) return;
    const int64_t pl = e / lpr;
    const int idx = int(e - pl * lpr), t = idx / Wp, cc = idx - t * Wp;
    const float v = dy[(pl * Hout + min(t, Hout - 1)) * Wout + min(cc, Wout - 1)];
    dyp[e] = (t < Hout && cc < Wout) ? v : 0.f;
}

static int bw2d_geometry(const agx_conv2d_desc *d, Bw2dGeom *g, int *cfg, int *bm, dim3 *grid, size_t *lds) {
    ConvPlan f;
    int rc = lower_conv2d(d, &f);
    if (rc != AGX_OK) return rc;
    g->B = d->batch; g->Cin = d->c_in; g->Cout = d->c_out; g->Hin = d->h_in; g->Win = d->w_in;
    g->Hout = f.Tout; g->Wout = f.Lout; g->kh = d->kh; g->kw = d->kw; g->sh = d->stride_h; g->sw = d->stride_w;
    g->ph = d->pad_h; g->pw = d->pad_w;
    g->WF = g->Wout < BW_T ? g->Wout : BW_T;
    g->R = BW_T / g->WF;
    if (g->R > g->Hout) g->R = g->Hout;
    g->RH = (g->R - 1) * g->sh + g->kh;
    g->SW = (g->WF - 1) * g->sw + g->kw;
    g->span = g->RH * g->SW;
    const int KK = g->kh * g->kw;
    g->n_chan = 127 / KK + 2;
    g->prec = d->impl == AGX_IMPL_MFMA_BF16X3 ? 1 : 0;
    g->xcd = tuning().dw_xcd;
    g->prepad = 0; g->pp_lpr = 0; g->pp_hwi = 0; g->pp_amin = 0; g->pp_bmin = 0;
    *cfg = g->Cout >= 128 ? 0 : (g->Cout >= 64 ? 1 : 2);
    *bm = *cfg == 0 ? 128 : (*cfg == 1 ? 64 : 32);
    const int nt = ceil_div(g->Cin * KK, 128), mt = ceil_div(g->Cout, *bm);
    const int64_t items = int64_t(g->B) * ceil_div(g->Hout, g->R) * ceil_div(g->Wout, g->WF);
    int64_t ns = ceil_div(tuning().dw_wgs, nt * mt);   // default 1536: +5..15 % over 768 on the 3x3 layers
    if (ns > items) ns = items;
    if (ns < 1) ns = 1;
    if (ns > 65535) ns = 65535;
    g->n_slices = int(ns);
    *grid = dim3(nt, mt, g->n_slices);
    *lds = (size_t(*bm) * BW_TS + size_t(g->n_chan) * g->span + BW_T) * sizeof(float);
    // conv2d_bwd_weight_direct_kernel: stride-1 "same" layers whose rows are whole 32-column items
    // (also for AGX_IMPL_MFMA_BF16X3 descriptors: fp32 on this kernel is faster than bf16x3 on the staged one, and exact)
    // or column-strided ones whose output row is as wide as a column-phase plane of x
    g->Wp = ceil_div(g->Win, g->sw);
    const int a_lo = -ceil_div(g->pw, g->sw), a_hi = (g->kw - 1 - g->pw) / g->sw;      // column shifts inside a plane
    const bool same = g->sh == 1 && g->sw == 1 && g->Hout == g->Hin && g->Wout == g->Win;
    const bool strided = tuning().dw2_direct >= 2 && g->sw > 1 && g->sw <= 4 && g->Win % g->sw == 0 && g->Wout == g->Wp &&
                         int64_t(g->sw) * g->B * g->Cin * g->Hin * g->Wp < (int64_t(1) << 31);   // 32-bit plane offsets
    if (tuning().dw2_direct && (same || strided) &&
        g->Wout % 32 == 0 && -a_lo <= 4 && a_hi <= 4 && g->ph < 32 && g->kh - g->ph < 32 &&
        (g->Cout > 32 || g->Cin * KK <= 32 || g->Cin * KK >= 192)) {   // (32 rows x 98 columns, the 7x7 first conv: the 256-wide tile loses to the staged kernel)
        const int NK = g->Cin * KK;
        int dbm, dbn, wk;
        if (g->Cout > 64)      { *cfg = 10; dbm = 128; dbn = 128; wk = 1; }   // <2,2,2,2>
        else if (g->Cout > 32) { if (NK > 64) { *cfg = 11; dbm = 64; dbn = 128; wk = 2; }    // <2,2,1,2>
                                 else         { *cfg = 12; dbm = 64; dbn = 64; wk = 4; } }   // <2,2,1,1>
        else                   { if (NK > 32) { *cfg = 13; dbm = 32; dbn = 256; wk = 1; }    // <1,2,1,4>
                                 else         { *cfg = 14; dbm = 32; dbn = 32; wk = 4; } }   // <1,1,1,1>
        // 32 rows: 96-column tiles (the four waves split the contraction) when they waste fewer columns than 256-column
        // ones -- 32 -> 32 3x3: NK = 288 = 3 x 96, against 2 x 256
        if (*cfg == 13 && ceil_div(NK, 96) * 96 < ceil_div(NK, 256) * 256) { *cfg = 17; dbm = 32; dbn = 96; wk = 4; }   // <1,3,1,1>
        // (64 rows x 96 columns, <2,3,1,1>, for NK = 576 = 6 x 96: 73.8 TFLOP/s against 81 on 5 tiles of 128 -- not kept)
        if (tuning().dw2_shared >= 2 && *cfg == 11) { *cfg = 15; dbm = 64; dbn = 128; wk = 1; }    // shared <2,1,1,4>
        if (tuning().dw2_shared >= 2 && *cfg == 13) { *cfg = 16; dbm = 32; dbn = 256; wk = 1; }    // shared <1,2,1,4>
        const int dnt = ceil_div(NK, dbn), dmt = ceil_div(g->Cout, dbm);
        const int64_t ditems = int64_t(g->B) * g->Hout * (g->Wout / 32);
        int64_t gz = ceil_div(tuning().dw_wgs, dnt * dmt);
        if (gz * wk > ditems) gz = ceil_div64(ditems, wk);
        if (gz < 1) gz = 1;
        if (gz > 65535) gz = 65535;
        g->n_slices = int(gz);
        *bm = dbm;
        *grid = dim3(dnt, dmt, g->n_slices);
        *lds = 0;
    } else if (tuning().dw2_prepad && tuning().dw2_shared && g->Wout < 32 && g->Wout >= 4 && g->Cout >= 64 && g->sh <= 2 && g->sw <= 2 &&
               g->kh <= 8 && g->kw <= 8) {
        // narrow maps: the shared kernel on zero-padded, phase-split, flattened copies of x and dy (Bw2dGeom::prepad)
        const int amin = -ceil_div(g->ph, g->sh), amax = (g->kh - 1 - g->ph) >= 0 ? (g->kh - 1 - g->ph) / g->sh : -ceil_div(g->ph - g->kh + 1, g->sh);
        const int bmin = -ceil_div(g->pw, g->sw), bmax = (g->kw - 1 - g->pw) >= 0 ? (g->kw - 1 - g->pw) / g->sw : -ceil_div(g->pw - g->kw + 1, g->sw);
        const int Wpp = g->Wout + (bmax - bmin), Hp = g->Hout + (amax - amin);
        const int64_t lpr = ceil_div64(int64_t(g->Hout) * Wpp, 32) * 32;
        const int64_t hwi = (int64_t(Hp) * Wpp + 64 + 3) / 4 * 4;
        const bool fits = int64_t(g->sh) * g->sw * g->B * g->Cin * hwi < (int64_t(1) << 30) && int64_t(g->B) * g->Cout * lpr < (int64_t(1) << 30);
        if (fits && lpr * 10 <= int64_t(g->Hout) * g->Wout * 13) {      // padded positions <= 1.3 x the real ones: 8 columns and more
                                                                         // (measured: 16 columns 64 -> 79 TFLOP/s, 8 columns 69 -> 76, 4 columns 67 -> 61)
            const int NK = g->Cin * KK;
            int dbm, dbn;
            if (g->Cout > 64) { *cfg = 10; dbm = 128; dbn = 128; }
            else              { *cfg = 15; dbm = 64; dbn = 128; }
            g->prepad = 1; g->Wp = Wpp; g->pp_lpr = int(lpr); g->pp_hwi = int(hwi); g->pp_amin = amin; g->pp_bmin = bmin;
            const int dnt = ceil_div(NK, dbn), dmt = ceil_div(g->Cout, dbm);
            const int64_t ditems = int64_t(g->B) * (lpr / 32);
            int64_t gz = ceil_div(tuning().dw_wgs, dnt * dmt);
            if (gz > ditems) gz = ditems;
            if (gz < 1) gz = 1;
            if (gz > 65535) gz = 65535;
            g->n_slices = int(gz);
            *bm = dbm;
            *grid = dim3(dnt, dmt, g->n_slices);
            *lds = 0;
        }
    }
    return AGX_OK;
}

struct BwGeom {
    int cfg;  // 0: 128x128 tile, 1: 64x128, 2: 32x128;  direct kernel: 10..14 (see bw_geometry)
    int bm, span, n_chan, n_slices;
    bool direct;
    dim3 grid;
    size_t lds;
};

static BwGeom bw_geometry(const ConvPlan &p, bool bf16x3) {
    BwGeom g;
    // (also for AGX_IMPL_MFMA_BF16X3 descriptors: fp32 on this kernel is faster than bf16x3 on the staged one, and exact)
    // dw_direct: 1 = the k = 1 layers, 2 = + every stride-1 layer, 3 (default) = + strided / transposed layers through a
    // phase-split copy of x / dy
    const int dd = tuning().dw_direct;
    g.direct = p.G == 1 && ((p.s == 1 && p.q == 1 && (dd >= 2 || (dd == 1 && p.J == 1))) ||
                            (dd >= 3 && (p.s == 1 || p.q == 1) && p.s <= 16 && p.q <= 16));
    (void)bf16x3;
    if (g.direct) {   // conv_bwd_weight_direct_kernel: tile and the waves left for the contraction (WK)
        const int NK = p.Cin * p.J;
        int bn, wk;
        if (p.M > 64)      { g.cfg = 10; g.bm = 128; bn = 128; wk = 1; }   // <2,2,2,2>
        else if (p.M > 32) { if (NK > 64) { g.cfg = 11; g.bm = 64; bn = 128; wk = 2; }    // <2,2,1,2>
                             else         { g.cfg = 12; g.bm = 64; bn = 64; wk = 4; } }   // <2,2,1,1>
        else               { if (NK > 32) { g.cfg = 13; g.bm = 32; bn = 256; wk = 1; }    // <1,2,1,4>
                             else         { g.cfg = 14; g.bm = 32; bn = 32; wk = 4; } }   // <1,1,1,1>
        const int nt = ceil_div(NK, bn), mt = ceil_div(p.M, g.bm);
        const int items = p.B * ceil_div(p.Lt, 32);
        int gz = ceil_div(tuning().dw1_wgs, nt * mt);
        if (gz * wk > items) gz = ceil_div(items, wk);
        if (gz < 1) gz = 1;
        if (gz > 65535) gz = 65535;
        g.n_slices = gz;     // partial tiles in the workspace (the WK waves of a workgroup are added in LDS)
        g.grid = dim3(nt, mt, gz);
        g.span = g.n_chan = 0;
        g.lds = 0;
        return g;
    }
    g.cfg = p.M >= 128 ? 0 : (p.M >= 64 ? 1 : 2);
    g.bm = g.cfg == 0 ? 128 : (g.cfg == 1 ? 64 : 32);
    g.span = (BW_T - 1) * p.s + (p.J - 1) * p.d + 1;
    g.n_chan = 127 / p.J + 2;  // channels a 128-column tile of n = ci*J + j can touch
    const int nt = ceil_div(p.Cin * p.J, 128), mt = ceil_div(p.M, g.bm);
    const int items = p.B * ceil_div(p.Lt, BW_T);
    int ns = ceil_div(768, nt * mt);  // ~3 workgroups per CU in total
    if (ns > items) ns = items;
    if (ns < 1) ns = 1;
    if (ns > 65535) ns = 65535;
    g.n_slices = ns;
    g.grid = dim3(nt, mt, ns);
    g.lds = (size_t(g.bm) * BW_TS + size_t(g.n_chan) * g.span) * sizeof(float);
    return g;
}

}  // namespace agx

extern "C" {

size_t agx_conv_bwd_weight_workspace_bytes(const agx_conv_desc *d) {
    using namespace agx;
    ConvPlan p;
    if (lower_conv(d, &p) != AGX_OK) return 0;
    const BwGeom g = bw_geometry(p, d->impl == AGX_IMPL_MFMA_BF16X3);
    // slices of dWp + the reduced dWp + slices of the bias row sums (+ the phase-split copy of x or dy)
    size_t floats = (size_t(g.n_slices) + 1) * p.Cin * p.J * p.M + (size_t(g.n_slices) + 1) * p.M;
    if (g.direct && p.s > 1) floats += size_t(p.B) * p.Cin * p.s * ceil_div(p.Lin, p.s) + 128;
    if (g.direct && p.q > 1) floats += size_t(p.B) * p.M * p.Lt + 128;
    return floats * sizeof(float);
}

int agx_conv_bwd_weight(const agx_conv_desc *d, const float *x, const float *dy, const float *v, const float *g,
                        float *dv, float *dg, float *dbias, void *workspace, size_t workspace_bytes,
                        void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    if (!x || !dy || !v || !dv || (g && !dg)) return fail(AGX_ERR_NULL_POINTER, "agx_conv_bwd_weight: NULL pointer");
    if (!workspace || workspace_bytes < agx_conv_bwd_weight_workspace_bytes(d))
        return fail(AGX_ERR_WORKSPACE, "agx_conv_bwd_weight: workspace too small (%zu < %zu)", workspace_bytes,
                    agx_conv_bwd_weight_workspace_bytes(d));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const BwGeom geo = bw_geometry(p, d->impl == AGX_IMPL_MFMA_BF16X3);
    if (geo.lds > 160 * 1024) return fail(AGX_ERR_UNSUPPORTED, "agx_conv_bwd_weight: tile needs %zu B of LDS", geo.lds);
    float *part = static_cast<float *>(workspace);
    const int64_t nw = int64_t(p.Cin) * p.J * p.M;
    float *dwp = part + size_t(geo.n_slices) * nw;
    float *bias_part = dbias ? dwp + nw : nullptr;
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, geo.grid, dim3(256), geo.lds, st, p, geo.span, geo.n_chan, geo.n_slices, x, dy, part,
                           bias_part);
        return AGX_OK;
    };
    ConvPlan pd = p;            // the plan as the direct kernel sees it (phase-split operands: see the kernel's header)
    const float *xd = x, *dyd = dy;
    int sp = 1;
    if (geo.direct && (p.s > 1 || p.q > 1)) {
        float *extra = part + (size_t(geo.n_slices) + 1) * nw + (size_t(geo.n_slices) + 1) * p.M + 64;
        if (p.s > 1) {
            const int Lp = ceil_div(p.Lin, p.s);
            const int64_t rows = int64_t(p.B) * p.Cin;
            hipLaunchKernelGGL(phase_split_rows_kernel, dim3((unsigned)ceil_div64(rows * Lp, 256)), dim3(256), 0, st, x, extra,
                               rows, p.Lin, p.Lvalid, Lp, p.s);
            xd = extra;
            sp = p.s;
            pd.Lin = pd.Lvalid = Lp;
        } else {
            const int64_t rows = int64_t(p.B) * p.Cout;
            hipLaunchKernelGGL(phase_split_rows_kernel, dim3((unsigned)ceil_div64(rows * p.Lt, 256)), dim3(256), 0, st, dy, extra,
                               rows, p.Lout, p.Lout, p.Lt, p.q);
            dyd = extra;
            pd.Lout = p.Lt;
            pd.Cout = p.M;
        }
    }
    auto launch_direct = [&](auto kern, int blocks_per_wave) -> int {   // 4 waves x (MW + NW) operand blocks of 4 KB
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, geo.grid, dim3(256), size_t(4) * blocks_per_wave * 4096, st, pd, sp, xd, dyd, part, bias_part);
        return AGX_OK;
    };
    if (geo.direct && sp > 1)
        rc = geo.cfg == 10 ? launch_direct(conv_bwd_weight_direct_kernel<2, 2, 2, 2, true>, 4)
           : geo.cfg == 11 ? launch_direct(conv_bwd_weight_direct_kernel<2, 2, 1, 2, true>, 4)
           : geo.cfg == 12 ? launch_direct(conv_bwd_weight_direct_kernel<2, 2, 1, 1, true>, 4)
           : geo.cfg == 13 ? launch_direct(conv_bwd_weight_direct_kernel<1, 2, 1, 4, true>, 3)
                           : launch_direct(conv_bwd_weight_direct_kernel<1, 1, 1, 1, true>, 2);
    else if (geo.direct)
        rc = geo.cfg == 10 ? launch_direct(conv_bwd_weight_direct_kernel<2, 2, 2, 2>, 4)
           : geo.cfg == 11 ? launch_direct(conv_bwd_weight_direct_kernel<2, 2, 1, 2>, 4)
           : geo.cfg == 12 ? launch_direct(conv_bwd_weight_direct_kernel<2, 2, 1, 1>, 4)
           : geo.cfg == 13 ? launch_direct(conv_bwd_weight_direct_kernel<1, 2, 1, 4>, 3)
                           : launch_direct(conv_bwd_weight_direct_kernel<1, 1, 1, 1>, 2);
    else if (d->impl == AGX_IMPL_MFMA_BF16X3)
        rc = geo.cfg == 0 ? launch(conv_bwd_weight_kernel<2, 2, 2, 2, 1>)
           : geo.cfg == 1 ? launch(conv_bwd_weight_kernel<1, 2, 2, 2, 1>)
                          : launch(conv_bwd_weight_kernel<1, 1, 1, 4, 1>);
    else
        rc = geo.cfg == 0 ? launch(conv_bwd_weight_kernel<2, 2, 2, 2>)
           : geo.cfg == 1 ? launch(conv_bwd_weight_kernel<1, 2, 2, 2>)
                          : launch(conv_bwd_weight_kernel<1, 1, 1, 4>);
    if (rc != AGX_OK) return rc;
    launch_slice_reduce(part, geo.n_slices, nw, dwp, st);
    const bool transposed = d->kind == AGX_CONV_TRANSPOSED;
    const int dim0 = transposed ? d->c_in : d->c_out;
    hipLaunchKernelGGL(bwd_weight_unpack_kernel, dim3(dim0), dim3(256), 0, st, dwp, v, g, dv, dg, d->kind, p.Cin,
                       p.Cout, d->kernel, p.q, p.J, p.P, d->stride);
    if (dbias) {
        float *rowsum = bias_part + size_t(geo.n_slices) * p.M;
        launch_slice_reduce(bias_part, geo.n_slices, int64_t(p.M), rowsum, st);
        hipLaunchKernelGGL(bwd_bias_fold_kernel, dim3(ceil_div(p.Cout, 256)), dim3(256), 0, st, rowsum, p.q, p.Cout,
                           dbias);
    }
    return check_launch("agx_conv_bwd_weight");
}

size_t agx_conv2d_bwd_weight_workspace_bytes(const agx_conv2d_desc *d) {
    using namespace agx;
    Bw2dGeom g;
    int cfg, bm;
    dim3 grid;
    size_t lds;
    if (bw2d_geometry(d, &g, &cfg, &bm, &grid, &lds) != AGX_OK) return 0;
    const size_t nw = size_t(g.Cin) * g.kh * g.kw * g.Cout;
    size_t floats = (size_t(g.n_slices) + 1) * nw + (size_t(g.n_slices) + 2) * g.Cout;
    if (cfg >= 10 && g.prepad) floats += size_t(g.sh) * g.sw * g.B * g.Cin * g.pp_hwi + size_t(g.B) * g.Cout * g.pp_lpr + 256;   // padded copies
    else if (cfg >= 10 && g.sw > 1) floats += size_t(g.sw) * g.B * g.Cin * g.Hin * g.Wp + 128;   // column-phase planes of x (+ slack)
    return floats * sizeof(float);
}

int agx_conv2d_bwd_weight(const agx_conv2d_desc *d, const float *x, const float *dy, const float *w,
                          const float *sigma, const float *u, const float *v, float *dw, float *dbias,
                          void *workspace, size_t workspace_bytes, void *stream) {
    using namespace agx;
    Bw2dGeom g;
    int cfg, bm;
    dim3 grid;
    size_t lds;
    int rc = bw2d_geometry(d, &g, &cfg, &bm, &grid, &lds);
    if (rc != AGX_OK) return rc;
    if (!x || !dy || !dw || (sigma && (!w || !u || !v)))
        return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_bwd_weight: NULL pointer");
    if (!workspace || workspace_bytes < agx_conv2d_bwd_weight_workspace_bytes(d))
        return fail(AGX_ERR_WORKSPACE, "agx_conv2d_bwd_weight: workspace too small");
    if (lds > 160 * 1024) return fail(AGX_ERR_UNSUPPORTED, "agx_conv2d_bwd_weight: tile needs %zu B of LDS", lds);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int NK = g.Cin * g.kh * g.kw, M = g.Cout;
    const int64_t nw = int64_t(NK) * M;
    float *part = static_cast<float *>(workspace);
    float *dwp = part + size_t(g.n_slices) * nw;
    float *bias_part = dwp + nw;                       // [n_slices][M], then rowsum [M], then rowdot [M]
    float *rowsum = bias_part + size_t(g.n_slices) * M;
    float *rowdot = rowsum + M;
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g, x, dy, part, dbias ? bias_part : nullptr);
        return AGX_OK;
    };
    auto launch_direct = [&](auto kern, int blocks_per_wave) -> int {   // 4 waves x (MW + NW) operand blocks of 4 KB
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, grid, dim3(256), size_t(4) * blocks_per_wave * 4096, st, g, x, dy, part,
                           dbias ? bias_part : nullptr);
        return AGX_OK;
    };
    if (cfg >= 10) {
        if (g.prepad) {   // narrow map: zero-padded, phase-split, flattened copies of both operands
            float *xp = rowdot + M + 64;
            xp += (4 - (reinterpret_cast<uintptr_t>(xp) / 4) % 4) % 4;          // 16-byte aligned planes
            const int64_t nx = int64_t(g.sh) * g.sw * g.B * g.Cin * g.pp_hwi, ny = int64_t(g.B) * M * g.pp_lpr;
            float *dyp = xp + nx;
            const int amax_rows = g.Hout + ((g.kh - 1 - g.ph) >= 0 ? (g.kh - 1 - g.ph) / g.sh : 0) - g.pp_amin;
            hipLaunchKernelGGL(prepad_x_kernel, dim3((unsigned)ceil_div64(nx, 256)), dim3(256), 0, st, x, xp, int64_t(g.B) * g.Cin, g.Hin,
                               g.Win, amax_rows, g.Wp, g.pp_hwi, g.sh, g.sw, g.pp_amin, g.pp_bmin);
            hipLaunchKernelGGL(prepad_dy_kernel, dim3((unsigned)ceil_div64(ny, 256)), dim3(256), 0, st, dy, dyp, int64_t(g.B) * M, g.Hout,
                               g.Wout, g.Wp, g.pp_lpr);
            x = xp;
            dy = dyp;
        } else if (g.sw > 1) {   // column-strided layer: x through its sw column-phase planes
            float *xs = rowdot + M + 64;
            const int64_t xrows = int64_t(g.B) * g.Cin * g.Hin;
            hipLaunchKernelGGL(deinterleave_cols_kernel, dim3((unsigned)ceil_div64(xrows * g.Wp, 256)), dim3(256), 0, st, x,
                               xs, xrows, g.Win, g.Wp, g.sw);
            x = xs;
        }
        auto launch_shared = [&](auto kern, int blocks) -> int {   // two slots of (BM + BN) / 32 operand blocks of 4 KB
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(kern, grid, dim3(256), size_t(2) * blocks * 4096, st, g, x, dy, part, dbias ? bias_part : nullptr);
            return AGX_OK;
        };
        const bool bf = g.prec == 1 && tuning().dw2_bf;   // bf16x3 descriptors: the shared kernel's bf16x3 contraction
        rc = (cfg == 10 && tuning().dw2_shared) ? (bf ? launch_shared(conv2d_bwd_weight_shared_kernel<2, 2, 2, 2, 1>, 8)
                                                      : launch_shared(conv2d_bwd_weight_shared_kernel<2, 2, 2, 2>, 8))
           : (cfg == 15) ? (bf ? launch_shared(conv2d_bwd_weight_shared_kernel<2, 1, 1, 4, 1>, 6)
                               : launch_shared(conv2d_bwd_weight_shared_kernel<2, 1, 1, 4>, 6))
           : (cfg == 16) ? launch_shared(conv2d_bwd_weight_shared_kernel<1, 2, 1, 4>, 9)
           : cfg == 10 ? launch_direct(conv2d_bwd_weight_direct_kernel<2, 2, 2, 2>, 4)
           : cfg == 11 ? launch_direct(conv2d_bwd_weight_direct_kernel<2, 2, 1, 2>, 4)
           : cfg == 12 ? launch_direct(conv2d_bwd_weight_direct_kernel<2, 2, 1, 1>, 4)
           : cfg == 13 ? launch_direct(conv2d_bwd_weight_direct_kernel<1, 2, 1, 4>, 3)
           : cfg == 17 ? launch_direct(conv2d_bwd_weight_direct_kernel<1, 3, 1, 1>, 4)
                       : launch_direct(conv2d_bwd_weight_direct_kernel<1, 1, 1, 1>, 2);
    } else if (g.prec) {
        rc = cfg == 0 ? launch(conv2d_bwd_weight_kernel<2, 2, 2, 2, 1>)
           : cfg == 1 ? launch(conv2d_bwd_weight_kernel<1, 2, 2, 2, 1>)
                      : launch(conv2d_bwd_weight_kernel<1, 1, 1, 4, 1>);
    } else {
        rc = cfg == 0 ? launch(conv2d_bwd_weight_kernel<2, 2, 2, 2>)
           : cfg == 1 ? launch(conv2d_bwd_weight_kernel<1, 2, 2, 2>)
                      : launch(conv2d_bwd_weight_kernel<1, 1, 1, 4>);
    }
    if (rc != AGX_OK) return rc;
    launch_slice_reduce(part, g.n_slices, nw, dwp, st);
    hipLaunchKernelGGL(bwd2d_unpack_kernel, dim3(M), dim3(256), 0, st, dwp, sigma ? w : nullptr, dw, rowdot, NK, M);
    if (sigma) hipLaunchKernelGGL(bwd2d_spectral_kernel, dim3(M), dim3(256), 0, st, dw, rowdot, sigma, u, v, NK, M);
    if (dbias)
        launch_slice_reduce(bias_part, g.n_slices, int64_t(M), dbias, st);
    return check_launch("agx_conv2d_bwd_weight");
}

#ifdef AGX_STAMPS
int agx_debug_read_stamps(unsigned long long *host, int n) {   // probe build only: copy out and clear the stamp array
    if (n > (1 << 16)) n = 1 << 16;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(agx::g_stamps), size_t(n) * 8) != hipSuccess) return AGX_ERR_LAUNCH;
    static unsigned long long zeros[1 << 16];
    if (hipMemcpyToSymbol(HIP_SYMBOL(agx::g_stamps), zeros, sizeof(zeros)) != hipSuccess) return AGX_ERR_LAUNCH;
    return AGX_OK;
}
#endif

}  // extern "C"
