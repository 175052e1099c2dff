// Device-side building blocks shared by the fp32-MFMA convolution kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"

namespace agx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Row of accumulator register r in lane half lh (C/D layout of v_mfma_f32_32x32x2_f32:
// column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)).
__device__ __forceinline__ constexpr int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// torch.nn.GELU() default (approximate='none'): 0.5 x (1 + erf(x / sqrt 2))
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

// Stage CC channel rows of the input, positions [in0, in0 + span), into LDS
// (xs[c * span + i]); positions outside [0, Lvalid) read as zero.  Loads are
// issued unconditionally on clamped addresses so that RPW * U of them are in
// flight per wave before the first LDS write (a conditional load makes hipcc
// wait for every element separately).
template <int CC, int U = 2>
__device__ __forceinline__ void stage_rows(float *__restrict__ xs, const float *__restrict__ xb, int Lin,
                                           int Lvalid, int in0, int span, int wave, int lane) {
    constexpr int RPW = CC / 4;  // rows per wave (4 waves)
    for (int i0 = lane; i0 < span; i0 += 64 * U) {
        float v[RPW][U];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const float *src = xb + size_t(wave + 4 * rr) * Lin;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pos = in0 + i0 + u * 64;
                const float t = src[min(max(pos, 0), Lvalid - 1)];
                v[rr][u] = (pos >= 0 && pos < Lvalid) ? t : 0.f;
            }
        }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 64;
                if (i < span) xs[(wave + 4 * rr) * span + i] = v[rr][u];
            }
    }
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

// Issue the LDS-DMA of CC channel rows, positions [in0, in0 + span), into buf
// (row stride span).  Out-of-range positions are never written: the caller
// zero-fills both buffers once and they stay zero (the in-range set does not
// depend on the channel chunk).
template <int CC>
__device__ __forceinline__ void issue_rows_dma(float *__restrict__ buf, const float *__restrict__ xb, int Lin,
                                               int Lvalid, int in0, int span, int wave, int lane) {
#pragma unroll
    for (int rr = 0; rr < CC / 4; ++rr) {
        const int c = wave + 4 * rr;
        const float *src = xb + size_t(c) * Lin + in0 + lane;
        float *dst = buf + c * span;
        for (int i0 = 0; i0 < span; i0 += 64) {
            const int i = i0 + lane, pos = in0 + i;
            if (i < span && pos >= 0 && pos < Lvalid) glds_dword(src + i0, dst + i0);
        }
    }
}

// The implicit-GEMM main loop over all (channel chunk, tap) phases.
//   acc[i][k] += sum_{c, j} Wp[c*J + j][arow[i]] * x[c][bcol0[k] + j*d]   (x via the LDS tile)
//
// Pipeline: the input chunk c+1 streams into the second LDS buffer by LDS-DMA
// while the MFMAs of chunk c run (one barrier per chunk); A fragments (weights,
// L2-resident) are prefetched one (chunk, tap) phase ahead into registers; B
// fragments are single ds_read_b32, batched per phase.  xs holds 2 * CC * span floats.
// ABL (ablation, timing-only diagnostic builds; 0 in every shipped launch):
//   bit 0: skip the per-chunk input DMA, bit 1: skip the weight prefetch loads,
//   bit 2: skip the per-chunk barrier.  Results are wrong by construction.
struct NoChunkHook {
    __device__ __forceinline__ void operator()(int, const float *) const {}
};

// on_chunk(c0, cur) is called once per channel chunk while its LDS tile `cur` is valid
// (used by the fused residual block to pick the residual operand out of the staged input).
template <int MW, int NW, int CC, int ABL = 0, typename Hook = NoChunkHook>
__device__ __forceinline__ void conv_gemm(f32x16 (&acc)[MW][NW], float *__restrict__ xs,
                                          const float *__restrict__ xb, const float *__restrict__ wp,
                                          const ConvPlan &p, int M, int span, int in0,
                                          const int (&arow)[MW], const int (&bcol)[NW], int wave, int lane,
                                          Hook on_chunk = Hook()) {
    const int lh = lane >> 5;
    const int tid = wave * 64 + lane;
    float *buf0 = xs, *buf1 = xs + CC * span;
    for (int e = tid; e < 2 * CC * span; e += 256) xs[e] = 0.f;
    float a_cur[CC / 2][MW], a_nxt[CC / 2][MW];
    load_a_phase<MW, CC>(a_cur, wp, 0, 0, p.J, M, lh, arow);
    __syncthreads();
    issue_rows_dma<CC>(buf0, xb, p.Lin, p.Lvalid, in0, span, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int it = 0;
    for (int c0 = 0; c0 < p.Cin; c0 += CC, ++it) {
        float *cur = (it & 1) ? buf1 : buf0;
        float *nxt = (it & 1) ? buf0 : buf1;
        on_chunk(c0, cur);
        for (int j = 0; j < p.J; ++j) {
            int nj = j + 1, nc0 = c0;
            if (nj == p.J) {
                nj = 0;
                nc0 += CC;
            }
            if (!(ABL & 2) && nc0 < p.Cin) load_a_phase<MW, CC>(a_nxt, wp, nc0, nj, p.J, M, lh, arow);
            if (ABL & 2) {
#pragma unroll
                for (int ks = 0; ks < CC / 2; ++ks)
#pragma unroll
                    for (int i = 0; i < MW; ++i) a_nxt[ks][i] = a_cur[ks][i] * 1.0001f;
            }
            // The next chunk's DMA goes out behind this phase's weight prefetch: vmcnt retires in
            // order, and hipcc waits vmcnt(0) at the top of the next phase for the prefetched
            // weights, so the DMA gets a full phase of MFMAs to land instead of none.
            if (!(ABL & 1) && j == 0 && c0 + CC < p.Cin)
                issue_rows_dma<CC>(nxt, xb + size_t(c0 + CC) * p.Lin, p.Lin, p.Lvalid, in0, span, wave, lane);
            // pin the prefetch at the head of the phase: without the barrier hipcc sinks these
            // loads to the end of the phase and waits vmcnt(0) for them at the top of the next one
            __builtin_amdgcn_sched_barrier(0);
            const float *xj = cur + j * p.d;
            float bf[CC / 2][NW];
#pragma unroll
            for (int ks = 0; ks < CC / 2; ++ks)
#pragma unroll
                for (int k = 0; k < NW; ++k) bf[ks][k] = xj[ks * span + bcol[k]];
#pragma unroll
            for (int ks = 0; ks < CC / 2; ++ks)
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int k = 0; k < NW; ++k)
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ks][i], bf[ks][k], acc[i][k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < CC / 2; ++ks)
#pragma unroll
                for (int i = 0; i < MW; ++i) a_cur[ks][i] = a_nxt[ks][i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of the next chunk has landed
        if (!(ABL & 4)) __syncthreads();                   // everyone's has, and everyone is done reading cur
    }
}

}  // namespace agx
