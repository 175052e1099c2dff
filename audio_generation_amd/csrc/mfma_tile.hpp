// Device-side building blocks shared by the fp32-MFMA convolution kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"

namespace agx {

// In-kernel stamps for the diagnostic probe build only (tools/rb_probe.hip defines AGX_STAMPS);
// a no-op in libagx.
#ifdef AGX_STAMPS
__device__ unsigned long long g_stamps[1 << 16];
#define AGX_STAMP(slot)                                                                          \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)                              \
            g_stamps[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime();                   \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#define AGX_STAMP_ADD(slot, t0)                                                                  \
    do {                                                                                         \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)                              \
            g_stamps[blockIdx.x * 16 + (slot)] += __builtin_amdgcn_s_memtime() - (t0);           \
    } while (0)
#else
#define AGX_STAMP(slot) ((void)0)
#define AGX_STAMP_ADD(slot, t0) ((void)0)
#endif

// Phase scheduling of conv_gemm (template parameter SCHED):
//   0  operand requests as one block ahead of the MFMAs (sched_barrier between them)
//   1  weights block ahead, B-fragment ds_reads threaded between the MFMAs
//   2  weights (global loads) and B fragments both threaded between the MFMAs
constexpr int kSchedDefault = 1;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Row of accumulator register r in lane half lh (C/D layout of v_mfma_f32_32x32x2_f32:
// column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)).
__device__ __forceinline__ constexpr int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// torch.nn.GELU() default (approximate='none'): 0.5 x (1 + erf(x / sqrt 2))
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

// d/dx of the exact GELU
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// k-step mapping inside a CC-channel chunk: MFMA step ks multiplies channel
// c0 + ks (lane half 0) and channel c0 + CC/2 + ks (lane half 1), so a lane's
// operands for consecutive steps are consecutive channels = contiguous floats of
// the packed image (common.hpp): one 16-byte load feeds 4 MFMAs.
//
// A fragments (weights) of one (channel chunk, tap) phase: a[ks][i] =
// W[row arow[i]][channel c0 + lh*CC/2 + ks][tap j].
template <int MW, int CC>
__device__ __forceinline__ void load_a_phase(float (&a)[CC / 2][MW], const float *__restrict__ wp, int c0,
                                             int j, int J, int M, int lh, const int (&arow)[MW]) {
    const int ch0 = c0 + lh * (CC / 2);  // first channel of this lane half
    const float *slab = wp + (size_t(ch0 / kWG) * J + j) * M * kWG + (ch0 % kWG);
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const float *row = slab + size_t(arow[i]) * kWG;
#pragma unroll
        for (int v = 0; v < CC / 8; ++v) {
            // CC = 32: the half spans a whole 16-channel group = 4 vectors; CC = 16: 2; CC = 8: 1
            const f32x4 q = *reinterpret_cast<const f32x4 *>(row + 4 * v);
            a[4 * v + 0][i] = q[0];
            a[4 * v + 1][i] = q[1];
            a[4 * v + 2][i] = q[2];
            a[4 * v + 3][i] = q[3];
        }
    }
}

// Asynchronous global -> LDS copy of one dword per lane (LDS-DMA): the LDS
// destination is wave-uniform base + lane * 4, the global source is per lane,
// EXEC-masked lanes write nothing.  No VGPR is involved, so a whole input chunk
// can be in flight while the MFMAs of the previous chunk run.
__device__ __forceinline__ void glds_dword(const float *gsrc_lane, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}

// Where the input rows of the implicit GEMM live.  1-D layers: row c = channel c of one batch item.
struct RowMap1D {
    const float *xb;  // x + b * Cin * Lin
    int Lin;
    __device__ __forceinline__ const float *row(int c, bool &ok) const {
        ok = true;
        return xb + size_t(c) * Lin;
    }
};
// 2-D layers run as 1-D convs along the last axis over virtual channels c' = ci * kh + dh (input row
// t*sh - ph + dh of channel ci; rows outside [0, Tin) and channels >= ncv read as zero).
struct RowMap2D {
    const float *xb;  // x + b * Cin * cstride
    int64_t cstride;  // elements between channels (Tin * Lin)
    int Lin, kh, row0, Tin, ncv;
    __device__ __forceinline__ const float *row(int c, bool &ok) const {
        const int ci = c / kh, dh = c - ci * kh, r = row0 + dh;
        ok = c < ncv && r >= 0 && r < Tin;
        return xb + ci * cstride + int64_t(ok ? r : 0) * Lin;
    }
};

// ---- input staging policies of conv_gemm_rows ----------------------------------------------------
// A stager owns the global -> LDS traffic of one workgroup tile: the one-off zero fill of the positions
// the DMA never writes, the per-chunk DMA issue, and the LDS offset of tap j relative to tap 0.

// Rows: LDS row c = positions [in0, in0 + span) of input row rm.row(c).  Out-of-range positions are never
// written (zero-filled once: the in-range set does not depend on the channel chunk); rows the map reports
// absent (row-folded 2-D only) are zeroed at issue time.
template <class RowMap>
struct StagerRows {
    RowMap rm;
    int Lvalid, in0, d;

    __device__ __forceinline__ int tapoff(int j) const { return j * d; }

    __device__ __forceinline__ void zero_fill(float *xs, int nrows, int span, int tid) const {
        // interior tiles have nothing to fill: skipped instead of costing nrows*span/256 LDS stores per thread
        const int lo = min(max(-in0, 0), span);             // first in-range tile position
        const int hi = max(min(Lvalid - in0, span), lo);    // one past the last
        if (lo > 0 || hi < span) {
            const int nz = lo + (span - hi);                // out-of-range positions per row
            for (int e = tid; e < nrows * nz; e += 256) {
                const int row = e / nz, i = e - row * nz;
                xs[row * span + (i < lo ? i : hi + (i - lo))] = 0.f;
            }
        }
    }

    template <int CC>
    __device__ __forceinline__ void issue(float *__restrict__ buf, int c0, int span, int wave, int lane) const {
#pragma unroll
        for (int rr = 0; rr < CC / 4; ++rr) {
            const int c = wave + 4 * rr;
            bool ok;
            const float *src = rm.row(c0 + c, ok) + in0 + lane;
            float *dst = buf + c * span;
            if (ok) {
                for (int i0 = 0; i0 < span; i0 += 64) {
                    const int i = i0 + lane, pos = in0 + i;
                    if (i < span && pos >= 0 && pos < Lvalid) glds_dword(src + i0, dst + i0);
                }
            } else {
                for (int i = lane; i < span; i += 64) dst[i] = 0.f;
            }
        }
    }
};

// Patches (2-D layers, conv2d.hip): LDS row c = the RH x SW input patch of REAL channel c that the tile's
// R output rows x WF output columns need, row-major with pitch SW; tap j = dh * kw + dw sits at
// dh * SW + dw.  Out-of-image positions are zero-filled once (same set for every channel).
struct StagerPatch {
    const float *xb;   // x + b * Cin * cstride
    int64_t cstride;   // elements between channels (Tin * Fin)
    int Fin, Tin, row0, col0, SW, kw, ncr;  // ncr: real channels (the packed image rounds up to 16)
    float inv_sw;      // 1 / SW for the exact small-integer division below

    __device__ __forceinline__ int tapoff(int j) const {
        const int dh = j / kw;
        return dh * SW + (j - dh * kw);
    }
    __device__ __forceinline__ bool inside(int i, int &goff) const {
        const int r = int((float(i) + 0.5f) * inv_sw);  // exact: i < 2^16, quotient < 2^8
        const int c = i - r * SW;
        const int gr = row0 + r, gc = col0 + c;
        goff = gr * Fin + gc;
        return gr >= 0 && gr < Tin && gc >= 0 && gc < Fin;
    }
    __device__ __forceinline__ void zero_fill(float *xs, int nrows, int span, int tid) const {
        const int RH = span / SW;
        const bool interior = row0 >= 0 && row0 + RH <= Tin && col0 >= 0 && col0 + SW <= Fin;
        if (interior) return;
        for (int i = tid; i < span; i += 256) {
            int goff;
            if (!inside(i, goff))
                for (int row = 0; row < nrows; ++row) xs[row * span + i] = 0.f;
        }
    }
    template <int CC>
    __device__ __forceinline__ void issue(float *__restrict__ buf, int c0, int span, int wave, int lane) const {
#pragma unroll
        for (int rr = 0; rr < CC / 4; ++rr) {
            const int c = wave + 4 * rr;
            float *dst = buf + c * span;
            if (c0 + c < ncr) {
                const float *src = xb + (c0 + c) * cstride;
                for (int i0 = 0; i0 < span; i0 += 64) {
                    int goff;
                    const int i = i0 + lane;
                    if (inside(i, goff) && i < span) glds_dword(src + goff, dst + i0);
                }
            } else {
                for (int i = lane; i < span; i += 64) dst[i] = 0.f;
            }
        }
    }
};

// The implicit-GEMM main loop over all (channel chunk, tap) phases.
//   acc[i][k] += sum_{c, j} Wp[c*J + j][arow[i]] * x[c][bcol0[k] + j*d]   (x via the LDS tile)
//
// Pipeline: the input chunk c+1 streams into the second LDS buffer by LDS-DMA
// while the MFMAs of chunk c run (one barrier per chunk); A fragments (weights,
// L2-resident) are prefetched one (chunk, tap) phase ahead into registers; B
// fragments are single ds_read_b32, batched per phase.  xs holds 2 * CC * span floats.
template <int NW, int CC>
__device__ __forceinline__ void load_b_phase(float (&bf)[CC / 2][NW], const float *xj, int span,
                                             const int (&bcol)[NW]) {
#pragma unroll
    for (int ks = 0; ks < CC / 2; ++ks)
#pragma unroll
        for (int k = 0; k < NW; ++k) bf[ks][k] = xj[ks * span + bcol[k]];
}

// CC = channels per LDS chunk (one DMA hand-over + barrier per chunk); the register pipeline
// works in phases of PC = min(CC, 16) channels x one tap (operand arrays sized for PC).
template <int MW, int NW, int CC, int SCHED, class Stager>
__device__ __forceinline__ void conv_gemm_rows(f32x16 (&acc)[MW][NW], float *__restrict__ xs,
                                               const Stager &stg, const float *__restrict__ wp,
                                               const ConvPlan &p, int M, int span,
                                               const int (&arow)[MW], const int (&bcol)[NW], int wave, int lane) {
    constexpr int PC = CC < 16 ? CC : 16;
    constexpr int NH = CC / PC;  // register phases groups per chunk
    const int lh = lane >> 5;
    const int tid = wave * 64 + lane;
    float *buf0 = xs, *buf1 = xs + CC * span;
    stg.zero_fill(xs, 2 * CC, span, tid);
    float a_cur[PC / 2][MW], a_nxt[PC / 2][MW];
    float b_cur[PC / 2][NW], b_nxt[PC / 2][NW];
    load_a_phase<MW, PC>(a_cur, wp, 0, 0, p.J, M, lh, arow);
    __syncthreads();
    stg.template issue<CC>(buf0, 0, span, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    AGX_STAMP(1);
    int it = 0;
    for (int c0 = 0; c0 < p.Cin; c0 += CC, ++it) {
        float *cur = (it & 1) ? buf1 : buf0;
        float *nxt = (it & 1) ? buf0 : buf1;
        // first phase of the chunk: its B fragments can only be read now (the tile just landed)
        load_b_phase<NW, PC>(b_cur, cur, span, bcol);
        // retire these reads HERE (explicit lgkmcnt(0)): otherwise the loop header inherits them as
        // pending and hipcc puts its s_waitcnt in front of every phase's MFMAs instead of in front of
        // the end-of-phase register copies
        __builtin_amdgcn_s_waitcnt(0xC07F);
        for (int h = 0; h < NH; ++h) {
            for (int j = 0; j < p.J; ++j) {
                // next phase: (h, j+1) | (h+1, 0) | (next chunk, 0, 0)
                int nj = j + 1, nh = h, nc0 = c0;
                if (nj == p.J) {
                    nj = 0;
                    if (++nh == NH) {
                        nh = 0;
                        nc0 += CC;
                    }
                }
                // The next chunk's input DMA goes out first (once per chunk) ...
                if (h == 0 && j == 0 && c0 + CC < p.Cin)
                    stg.template issue<CC>(nxt, c0 + CC, span, wave, lane);
                // ... then the operands of the NEXT phase are requested, in the same basic block as this
                // phase's MFMAs so that the scheduler can thread them between the MFMAs: neither the L2
                // latency of the weights nor the LDS latency of the input sits in front of an MFMA (a
                // bare MFMA loop with its ds_read directly ahead of it loses 13-22 %: tools/mfma_peak.hip).
                // Branch-free on purpose (behind a conditional load hipcc puts a full s_waitcnt in front
                // of the MFMAs): after the very last phase the prefetch re-reads phase 0, and across a
                // chunk boundary the B prefetch is discarded (b_cur is re-read after the barrier).
                const bool last = nc0 >= p.Cin;
                load_a_phase<MW, PC>(a_nxt, wp, last ? 0 : nc0 + nh * PC, last ? 0 : nj, p.J, M, lh, arow);
                if (SCHED == 1) __builtin_amdgcn_sched_barrier(0);
                load_b_phase<NW, PC>(b_nxt, cur + (nc0 == c0 ? nh * PC * span + stg.tapoff(nj) : 0), span, bcol);
                if (SCHED == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < PC / 2; ++ks)
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int k = 0; k < NW; ++k)
                            acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ks][i], b_cur[ks][k], acc[i][k], 0, 0, 0);
                if (SCHED != 0) {
                    // Thread the next phase's operand requests between this phase's MFMAs (2 MFMA : 1 VMEM
                    // read : 1 DS read ...): an MFMA occupies the issue port for a fraction of its 64
                    // cycles, so the requests ride in its shadow.
#pragma unroll
                    for (int gidx = 0; gidx < 2 * MW * (PC / 8) + NW * (PC / 2); ++gidx) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
                        if (SCHED == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef AGX_STAMPS
                {   // how long does the end-of-phase wait for the prefetched weights take?
                    const unsigned long long tq = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (h == 0 && j == 0) AGX_STAMP_ADD(9, tq); else AGX_STAMP_ADD(10, tq);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
#pragma unroll
                for (int ks = 0; ks < PC / 2; ++ks) {
#pragma unroll
                    for (int i = 0; i < MW; ++i) a_cur[ks][i] = a_nxt[ks][i];
#pragma unroll
                    for (int k = 0; k < NW; ++k) b_cur[ks][k] = b_nxt[ks][k];
                }
            }
        }
#ifdef AGX_STAMPS
        const unsigned long long tw = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of the next chunk has landed
        __syncthreads();                                   // everyone's has, and everyone is done reading cur
        AGX_STAMP_ADD(8, tw);
    }
    AGX_STAMP(2);
}

// ---------------------------------------------------------------------------------------------
// bf16x3 form of the main loop (AGX_IMPL_MFMA_BF16X3).  Every fp32 operand is split into three bf16
// pieces x = h + m + l (24 significant bits); a product block is the six bf16 MFMAs hh + hm + mh + hl + lh + mm
// (fp32 accumulation, small terms first): 6 x 32 cycles for a 32 x 32 x 16 block against 8 x 64 on the fp32
// MFMA, with fp32-class accuracy (tools/gemm_bf16x3.hip: 6.8e-7 vs 3.9e-7 of the fp32 chain over K = 1024) but
// NOT the bitwise fp32 FMA chain -- used where only a float tolerance applies (the decoder), never where an
// integer is decided (encoder -> RVQ indices).  Weights are split once by the pack kernel (three planes per
// 16-channel group: 96 B per (group, tap, row)); the input is split in registers after its LDS read, one phase
// ahead, in the shadow of the previous phase's MFMAs.  A K = 16 block = the 16 channels of a phase: lane half
// lh holds channels 8 lh .. 8 lh + 7 -- the same channel assignment as the fp32 loop, so staging is shared.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 hh = (__bf16)x[i];
        const float r1 = x[i] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        h[i] = hh;
        m[i] = mm;
        l[i] = (__bf16)(r1 - (float)mm);
    }
}

template <int MW>
__device__ __forceinline__ void load_a_phase_bf(bf16x8 (&a)[3][MW], const __bf16 *__restrict__ wpb, int c0, int j, int J,
                                                int M, int lh, const int (&arow)[MW]) {
    const __bf16 *slab = wpb + (size_t(c0 / kWG) * J + j) * M * 48 + lh * 8;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const __bf16 *row = slab + size_t(arow[i]) * 48;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[pl][i] = *reinterpret_cast<const bf16x8 *>(row + pl * 16);
    }
}

// The six products of one K = 16 block, small terms first.  Product-major order: consecutive MFMAs go to
// DIFFERENT accumulators (MW * NW independent chains), so none waits for its predecessor's result.
template <int MW, int NW>
__device__ __forceinline__ void mfma_block_bf(f32x16 (&acc)[MW][NW], const bf16x8 (&a)[3][MW], const bf16x8 (&b)[3][NW]) {
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};   // (plane of a, plane of b): mm hl lh hm mh hh
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int k = 0; k < NW; ++k)
                acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]][k], acc[i][k], 0, 0, 0);
}

// Same pipeline as conv_gemm_rows (LDS-DMA double buffer per chunk, weights one phase ahead in registers,
// input fragments read one phase ahead), CC a multiple of 16.
template <int MW, int NW, int CC, class Stager, int BSCHED = 0>
__device__ __forceinline__ void conv_gemm_rows_bf(f32x16 (&acc)[MW][NW], float *__restrict__ xs, const Stager &stg,
                                                  const __bf16 *__restrict__ wpb, const ConvPlan &p, int M, int span,
                                                  const int (&arow)[MW], const int (&bcol)[NW], int wave, int lane) {
    static_assert(CC % 16 == 0, "bf16x3 phases are 16 channels deep");
    constexpr int NH = CC / 16;
    const int lh = lane >> 5;
    const int tid = wave * 64 + lane;
    float *buf0 = xs, *buf1 = xs + CC * span;
    stg.zero_fill(xs, 2 * CC, span, tid);
    bf16x8 a_cur[3][MW], a_nxt[3][MW], b_cur[3][NW];
    float b_raw[8][NW];
    load_a_phase_bf<MW>(a_cur, wpb, 0, 0, p.J, M, lh, arow);
    __syncthreads();
    stg.template issue<CC>(buf0, 0, span, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int it = 0;
    for (int c0 = 0; c0 < p.Cin; c0 += CC, ++it) {
        float *cur = (it & 1) ? buf1 : buf0;
        float *nxt = (it & 1) ? buf0 : buf1;
        load_b_phase<NW, 16>(b_raw, cur, span, bcol);
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = b_raw[q][k];
            split3(x, b_cur[0][k], b_cur[1][k], b_cur[2][k]);
        }
        for (int h = 0; h < NH; ++h) {
            for (int j = 0; j < p.J; ++j) {
                int nj = j + 1, nh = h, nc0 = c0;
                if (nj == p.J) {
                    nj = 0;
                    if (++nh == NH) {
                        nh = 0;
                        nc0 += CC;
                    }
                }
                if (h == 0 && j == 0 && c0 + CC < p.Cin) stg.template issue<CC>(nxt, c0 + CC, span, wave, lane);
                const bool last = nc0 >= p.Cin;
                load_a_phase_bf<MW>(a_nxt, wpb, last ? 0 : nc0 + nh * 16, last ? 0 : nj, p.J, M, lh, arow);
                load_b_phase<NW, 16>(b_raw, cur + (nc0 == c0 ? nh * 16 * span + stg.tapoff(nj) : 0), span, bcol);
                bf16x8 b_new[3][NW];
                if (BSCHED == 2) {   // split the next phase's input BEFORE this phase's MFMAs (it then only waits for LDS)
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
                        float x[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) x[q] = b_raw[q][k];
                        split3(x, b_new[0][k], b_new[1][k], b_new[2][k]);
                    }
                }
                mfma_block_bf<MW, NW>(acc, a_cur, b_cur);
                if (BSCHED != 2) {   // ... or while this phase's MFMAs drain
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
                        float x[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) x[q] = b_raw[q][k];
                        split3(x, b_new[0][k], b_new[1][k], b_new[2][k]);
                    }
                }
                if (BSCHED == 1) {   // thread the VALU work between the MFMAs: 1 MFMA : 3 VALU
#pragma unroll
                    for (int gidx = 0; gidx < 6 * MW * NW; ++gidx) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    }
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int k = 0; k < NW; ++k) b_cur[pl][k] = b_new[pl][k];
#pragma unroll
                    for (int i = 0; i < MW; ++i) a_cur[pl][i] = a_nxt[pl][i];
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// 1-D entry point (rows = channels of one batch item at xb).
template <int MW, int NW, int CC, int SCHED = kSchedDefault>
__device__ __forceinline__ void conv_gemm(f32x16 (&acc)[MW][NW], float *__restrict__ xs,
                                          const float *__restrict__ xb, const float *__restrict__ wp,
                                          const ConvPlan &p, int M, int span, int in0,
                                          const int (&arow)[MW], const int (&bcol)[NW], int wave, int lane) {
    const StagerRows<RowMap1D> stg{RowMap1D{xb, p.Lin}, p.Lvalid, in0, p.d};
    conv_gemm_rows<MW, NW, CC, SCHED>(acc, xs, stg, wp, p, M, span, arow, bcol, wave, lane);
}

}  // namespace agx
