// Fused causal residual block for gfx950 -- one launch, one read and one write
// of the activation tensor:
//
//     y = leaky( x + W2 . leaky( W1 (*)_dil x + b1 ) + b2 )
//
// (networks/vae.py:113-117 plus the activation that follows the block in the
// enclosing Sequential, vae.py:130-135 / 193-198.)
//
// A wave owns ALL C channels of its 32*NW time columns.  GEMM1 (the dilated
// k-tap conv) is the implicit GEMM of conv_mfma.hip.  Its 32x32 accumulator has
// the time column on the lane and the channel in the registers, which is exactly
// the B-operand shape of v_mfma_f32_32x32x2_f32 for a product that sums over
// the accumulator's ROW index: register s of lane half h holds hidden channel
// (s&3) + 8*(s>>2) + 4*h, so GEMM2 (the k=1 conv) feeds the activated
// accumulator registers straight back as B operands, k-step s <-> register s,
// with the A fragment W2[co][that channel].  The hidden activation never leaves
// the register file: no LDS round trip, no second kernel, and the memory-bound
// k=1 conv of the unfused path disappears.
#include <cstdlib>

#include "mfma_tile.hpp"

namespace agx {

// Residual operand for accumulator registers of channel chunk IC, read from the
// staged LDS tile (the same bytes GEMM1 consumes) instead of a second pass over HBM.
// Chunk IC covers channels [IC*16, IC*16+16) = subtile IC/2, registers 8*(IC&1) .. +7.
template <int IC, int MW, int NW>
__device__ __forceinline__ void grab_residual(f32x16 (&xres)[MW][NW], const float *cur, int span,
                                              const int (&rcol)[NW], int lh) {
    if constexpr (IC * 16 < 32 * MW) {
        constexpr int i = (IC * 16) / 32, r0 = 8 * (IC & 1);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int chl = (rr & 3) + 8 * (rr >> 2) + 4 * lh;  // acc_row(r0 + rr, lh) - 16*(IC&1)
#pragma unroll
            for (int k = 0; k < NW; ++k) xres[i][k][r0 + rr] = cur[chl * span + rcol[k]];
        }
    }
}

template <int MW, int NW, int CC, int ABL = 0, bool RESL = true>
__global__ __launch_bounds__(256, (MW <= 4 ? 2 : 1)) void resblock_mfma_kernel(ConvPlan p, int span, int post_act,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ w1,
                                                            const float *__restrict__ b1,
                                                            const float *__restrict__ w2,
                                                            const float *__restrict__ b2,
                                                            float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [2][CC][span]
    constexpr int C = 32 * MW, BN = 32 * NW * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int t0 = blockIdx.x * BN;
    const int n0 = wave * (32 * NW);
    const int b = blockIdx.y;
    const int in0 = t0 - p.P;  // stride 1

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    int arow[MW], bcol[NW];
#pragma unroll
    for (int i = 0; i < MW; ++i) arow[i] = i * 32 + li;
#pragma unroll
    for (int k = 0; k < NW; ++k) bcol[k] = n0 + k * 32 + li + lh * (CC / 2) * span;

    const float *xb = x + size_t(b) * C * p.Lin;

    // ---- GEMM1: h = W1 (*) x ------------------------------------------------------
    // RES_LDS: keep the residual operand x[co][t] (this lane's output elements) in registers,
    // picked from each staged chunk; costs 16*MW*NW VGPRs, so only while 2 waves/SIMD still fit.
    constexpr bool RES_LDS = RESL && (MW <= 4) && CC == 16;
    f32x16 xres[RES_LDS ? MW : 1][NW];
    int rcol[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) rcol[k] = n0 + k * 32 + li + p.P;
    auto hook = [&](int c0, const float *cur) {
        if constexpr (RES_LDS) {
            switch (c0 / 16) {
                case 0: grab_residual<0, MW, NW>(xres, cur, span, rcol, lh); break;
                case 1: grab_residual<1, MW, NW>(xres, cur, span, rcol, lh); break;
                case 2: grab_residual<2, MW, NW>(xres, cur, span, rcol, lh); break;
                case 3: grab_residual<3, MW, NW>(xres, cur, span, rcol, lh); break;
                case 4: grab_residual<4, MW, NW>(xres, cur, span, rcol, lh); break;
                case 5: grab_residual<5, MW, NW>(xres, cur, span, rcol, lh); break;
                case 6: grab_residual<6, MW, NW>(xres, cur, span, rcol, lh); break;
                default: grab_residual<7, MW, NW>(xres, cur, span, rcol, lh); break;
            }
        }
    };
    conv_gemm<MW, NW, CC, ABL>(acc, xs, xb, w1, p, C, span, in0, arow, bcol, wave, lane, hook);

    // ---- hidden activation, in registers --------------------------------------------
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = b1 ? b1[i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] : 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const float v = acc[i][k][r] + bv;
                acc[i][k][r] = v > 0.f ? v : v * p.slope;
            }
        }

    // ---- GEMM2: out = W2 . h, B operand = the accumulator registers -------------------
    f32x16 out[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[i][k][r] = 0.f;

#pragma unroll
    for (int i = 0; i < MW; ++i) {       // hidden-channel subtile
#pragma unroll
        for (int g = 0; g < 4; ++g) {    // register group: k-steps 4g..4g+3 <-> 4 consecutive hidden channels
            const int kch = i * 32 + 8 * g + 4 * lh;  // == i*32 + acc_row(4g, lh)
            const float *w2k = w2 + size_t(kch / kWG) * C * kWG + (kch % kWG) + size_t(li) * kWG;
            f32x4 a[MW];
#pragma unroll
            for (int io = 0; io < MW; ++io) a[io] = *reinterpret_cast<const f32x4 *>(w2k + size_t(io) * 32 * kWG);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int io = 0; io < MW; ++io)
#pragma unroll
                    for (int k = 0; k < NW; ++k)
                        out[io][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[io][s4], acc[i][k][4 * g + s4],
                                                                          out[io][k], 0, 0, 0);
        }
    }

    // ---- epilogue: + b2 + x, trailing activation (loads hoisted, stores predicated) ----
    float *yb = y + size_t(b) * C * p.Lin;
#pragma unroll
    for (int io = 0; io < MW; ++io) {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = b2 ? b2[io * 32 + acc_row(r, lh)] : 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int t = t0 + n0 + k * 32 + li;
            const int tc = min(t, p.Lin - 1);
            float xv[16];
            if constexpr (RES_LDS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xv[r] = xres[io][k][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) xv[r] = xb[size_t(io * 32 + acc_row(r, lh)) * p.Lin + tc];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = out[io][k][r] + bv[r] + xv[r];
                if (post_act) v = leaky(v, p.slope);
                if (t < p.Lin) yb[size_t(io * 32 + acc_row(r, lh)) * p.Lin + t] = v;
            }
        }
    }
}

template <int MW, int NW, int CC, int ABL = 0, bool RESL = true>
static int launch_rb(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2,
                     const float *b2, float *y, int post_act, hipStream_t st) {
    constexpr int BN = 32 * NW * 4;
    const int span = (BN - 1) + (p.J - 1) * p.d + 1;
    const size_t lds = size_t(2) * CC * span * sizeof(float);  // double-buffered input tile
    auto kern = resblock_mfma_kernel<MW, NW, CC, ABL, RESL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(ceil_div(p.Lin, BN), p.B), block(256);
    if (grid.y > 65535) return fail(AGX_ERR_BAD_SHAPE, "resblock: batch too large for one launch");
    hipLaunchKernelGGL(kern, grid, block, lds, st, p, span, post_act, x, w1, b1, w2, b2, y);
    return check_launch("resblock_mfma");
}

bool resblock_fused_supported(const ConvPlan &p) {
    if (p.Cin != p.Cout || p.s != 1 || p.q != 1 || p.Lvalid != p.Lin || p.Lt != p.Lin) return false;
    if (p.Cin != 32 && p.Cin != 64 && p.Cin != 128 && p.Cin != 256) return false;
    const int bn = p.Cin == 32 ? 512 : (p.Cin == 64 ? 256 : 128);
    const size_t span = size_t(bn - 1) + size_t(p.J - 1) * p.d + 1;
    return 2 * 16 * span * sizeof(float) <= 160 * 1024;
}

const char *resblock_variant(const ConvPlan &p) {
    switch (p.Cin) {
        case 32: return "resblock_mfma<1,4,16>";
        case 64: return "resblock_mfma<2,2,16>";
        case 128: return "resblock_mfma<4,1,16>";
        default: return "resblock_mfma<8,1,16>";
    }
}

int launch_resblock_fused(const ConvPlan &p, const float *x, const float *w1, const float *b1,
                          const float *w2, const float *b2, float *y, int post_act, hipStream_t st) {
    if (!resblock_fused_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "resblock: no fused kernel for C=%d", p.Cin);
    const bool resl = tuning().resblock_res_lds != 0;
#define AGX_RB(MW, NW, ABL)                                                                   \
    (resl ? launch_rb<MW, NW, 16, ABL, true>(p, x, w1, b1, w2, b2, y, post_act, st)          \
          : launch_rb<MW, NW, 16, ABL, false>(p, x, w1, b1, w2, b2, y, post_act, st))
    switch (p.Cin) {
        case 32: return AGX_RB(1, 4, 0);
        case 64: return AGX_RB(2, 2, 0);
        case 128:
            switch (tuning().ablate) {  // timing-only diagnostic builds of this one shape
                case 1: return AGX_RB(4, 1, 1);
                case 2: return AGX_RB(4, 1, 2);
                case 3: return AGX_RB(4, 1, 3);
                case 7: return AGX_RB(4, 1, 7);
                default: return AGX_RB(4, 1, 0);
            }
        default: return launch_rb<8, 1, 16, 0, false>(p, x, w1, b1, w2, b2, y, post_act, st);
    }
#undef AGX_RB
}

}  // namespace agx
