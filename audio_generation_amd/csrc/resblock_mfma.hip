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
#include "mfma_tile.hpp"

namespace agx {

template <int MW, int NW, int CC, int SCHED = kSchedDefault, int OCC = (MW <= 4 ? 2 : 1), int PREC = 0>   // PREC 1: bf16x3 (SCHED = its schedule)
__global__ __launch_bounds__(256, OCC) void resblock_mfma_kernel(ConvPlan p, int span, int post_act,
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

    AGX_STAMP(0);
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
    for (int k = 0; k < NW; ++k) bcol[k] = n0 + k * 32 + li + lh * ((CC < 16 ? CC : 16) / 2) * span;

    const float *xb = x + size_t(b) * C * p.Lin;

    // ---- GEMM1: h = W1 (*) x ------------------------------------------------------
    if (PREC == 1) {
        const StagerRows<RowMap1D> stg{RowMap1D{xb, p.Lin}, p.Lvalid, in0, p.d};
        conv_gemm_rows_bf<MW, NW, (CC < 16 ? 16 : CC), StagerRows<RowMap1D>, SCHED>(
            acc, xs, stg, reinterpret_cast<const __bf16 *>(w1), p, C, span, arow, bcol, wave, lane);
    } else {
        conv_gemm<MW, NW, CC, SCHED>(acc, xs, xb, w1, p, C, span, in0, arow, bcol, wave, lane);
    }

    // ---- hidden activation, in registers --------------------------------------------
    // Bias loads are unconditional (a NULL bias reads the weight image instead and is masked by a select):
    // behind `b1 ? b1[..] : 0` hipcc emits one branch + one waited load per element.
    const bool has_b1 = b1 != nullptr, has_b2 = b2 != nullptr;
    const float *b1p = has_b1 ? b1 : w1, *b2p = has_b2 ? b2 : w1;
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bl = b1p[i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
            const float bv = has_b1 ? bl : 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const float v = acc[i][k][r] + bv;
                acc[i][k][r] = v > 0.f ? v : v * p.slope;
            }
        }

    AGX_STAMP(3);
    // ---- GEMM2: out = W2 . h, B operand = the accumulator registers -------------------
    f32x16 out[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[i][k][r] = 0.f;

    if (PREC == 1) {
        // bf16x3 GEMM2: a K = 16 block = accumulator registers 8 kb .. 8 kb + 7 of hidden subtile i, i.e. hidden
        // channels 16 (2i + kb) + {4 lh + 0..3, 8 + 4 lh + 0..3}: the matching W2 pieces are two 8-byte runs of the
        // standard bf16x3 image per plane (group 2i + kb, tap 0).
        const __bf16 *w2b = reinterpret_cast<const __bf16 *>(w2);
#pragma unroll
        for (int i = 0; i < MW; ++i) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 bq[3][NW];
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    float xq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xq[e] = acc[i][k][8 * kb + e];
                    split3(xq, bq[0][k], bq[1][k], bq[2][k]);
                }
                bf16x8 aq[3][MW];
                const __bf16 *grp = w2b + size_t(2 * i + kb) * C * 48 + 4 * lh;
#pragma unroll
                for (int io = 0; io < MW; ++io) {
                    const __bf16 *row = grp + size_t(io * 32 + li) * 48;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(row + pl * 16);
                        const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(row + pl * 16 + 8);
                        aq[pl][io] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                }
                mfma_block_bf<MW, NW>(out, aq, bq);
            }
        }
    } else
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

    AGX_STAMP(4);
    // ---- epilogue: + b2 + x, trailing activation.  ALL residual loads go out first (the GEMM1
    // accumulators are dead, so their registers hold the 16*MW*NW operands): one exposed memory
    // latency per workgroup instead of one per 32x32 block.
    float *yb = y + size_t(b) * C * p.Lin;
    constexpr int EG = MW <= 2 ? MW : 2;  // row blocks per load batch (register budget at 2 waves/SIMD)
#pragma unroll
    for (int g0 = 0; g0 < MW; g0 += EG) {
        float xv[EG][NW][16];
        __builtin_amdgcn_sched_barrier(0);  // keep each batch of loads where it is written
#pragma unroll
        for (int ig = 0; ig < EG; ++ig)
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int tc = min(t0 + n0 + k * 32 + li, p.Lin - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    xv[ig][k][r] = xb[((g0 + ig) * 32 + acc_row(r, lh)) * p.Lin + tc];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ig = 0; ig < EG; ++ig) {
            const int io = g0 + ig;
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bl = b2p[io * 32 + acc_row(r, lh)];
                bv[r] = has_b2 ? bl : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int t = t0 + n0 + k * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = out[io][k][r] + bv[r] + xv[ig][k][r];
                    if (post_act) v = leaky(v, p.slope);
                    if (t < p.Lin) yb[(io * 32 + acc_row(r, lh)) * p.Lin + t] = v;
                }
            }
        }
    }
    AGX_STAMP(5);
}

template <int MW, int NW, int CC, int SCHED = kSchedDefault, int OCC = (MW <= 4 ? 2 : 1), int PREC = 0>
static int launch_rb(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2,
                     const float *b2, float *y, int post_act, hipStream_t st) {
    constexpr int BN = 32 * NW * 4;
    const int span = (BN - 1) + (p.J - 1) * p.d + 1;
    size_t lds = size_t(2) * CC * span * sizeof(float);  // double-buffered input tile
    if (lds > 160 * 1024) return fail(AGX_ERR_UNSUPPORTED, "resblock: tile needs %zu B of LDS", lds);
    const int wgs = tuning().rb_wgs;  // diagnostic: cap workgroups per CU by requesting more LDS
    if (wgs >= 1 && wgs <= 3 && lds < size_t(160 * 1024) / wgs) lds = size_t(160 * 1024) / wgs;
    auto kern = resblock_mfma_kernel<MW, NW, CC, SCHED, OCC, PREC>;
    static DeviceOnce once;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, nullptr, "resblock_mfma")) return rc;
    dim3 grid(ceil_div(p.Lin, BN), p.B), block(256);
    if (grid.y > 65535) return fail(AGX_ERR_BAD_SHAPE, "resblock: batch too large for one launch");
    hipLaunchKernelGGL(kern, grid, block, lds, st, p, span, post_act, x, w1, b1, w2, b2, y);
    return check_launch("resblock_mfma");
}

static int rb_bn(int c) { return c == 32 ? 512 : (c == 64 ? 256 : 128); }

bool resblock_fused_supported(const ConvPlan &p) {
    if (p.Cin != p.Cout || p.s != 1 || p.q != 1 || p.Lvalid != p.Lin || p.Lt != p.Lin) return false;
    if (p.Cin != 32 && p.Cin != 64 && p.Cin != 128 && p.Cin != 256) return false;
    const size_t span = size_t(rb_bn(p.Cin) - 1) + size_t(p.J - 1) * p.d + 1;
    return 2 * 16 * span * sizeof(float) <= 160 * 1024;
}

static bool rb_use_cc32(const ConvPlan &p) {
    const size_t span = size_t(rb_bn(p.Cin) - 1) + size_t(p.J - 1) * p.d + 1;
    return tuning().rb_cc == 32 && 2 * 32 * span * sizeof(float) <= 160 * 1024;
}

const char *resblock_variant(const ConvPlan &p) {
    const bool c32 = rb_use_cc32(p) && !p.prec;
    switch (p.Cin) {
        case 32: return c32 ? "resblock_mfma<1,4,32>" : "resblock_mfma<1,4,16>";
        case 64: return c32 ? "resblock_mfma<2,2,32>" : "resblock_mfma<2,2,16>";
        case 128: return c32 ? "resblock_mfma<4,1,32>" : "resblock_mfma<4,1,16>";
        default: return "resblock_mfma<8,1,16>";
    }
}

int launch_resblock_fused(const ConvPlan &p, const float *x, const float *w1, const float *b1,
                          const float *w2, const float *b2, float *y, int post_act, hipStream_t st) {
    if (!resblock_fused_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "resblock: no fused kernel for C=%d", p.Cin);
    const bool c32 = rb_use_cc32(p);
    // phase scheduling: measured best per shape (tools/ab_bench.py rb_sched 0 1 2, in-process A/B):
    // C=32: 1 (-12 % vs 0), C=64: 2 (-6 %), C=128: 1 (-11 %), C=256: 2 (-2 %); knob -1 = this table
    int sched = tuning().rb_sched;
    if (sched < 0) sched = (p.Cin == 64 || p.Cin == 256) ? 2 : 1;
#define AGX_RB(MW, NW)                                                                                   \
    (c32 ? launch_rb<MW, NW, 32>(p, x, w1, b1, w2, b2, y, post_act, st)                                 \
         : sched == 0 ? launch_rb<MW, NW, 16, 0>(p, x, w1, b1, w2, b2, y, post_act, st)                 \
         : sched == 2 ? launch_rb<MW, NW, 16, 2>(p, x, w1, b1, w2, b2, y, post_act, st)                 \
                      : launch_rb<MW, NW, 16, 1>(p, x, w1, b1, w2, b2, y, post_act, st))
    if (p.prec) {   // bf16x3 (AGX_IMPL_MFMA_BF16X3): both packed images are bf16x3 images
        // measured per shape (AGX_BF16X3=1 tools/ab_bench.py bf_sched 0 1 2): C=32: 0, C=64: 2 (-5 %), C=128: 1 (-12 %),
        // C=256: 1 (-6 %); knob -1 = this table
        int bs = tuning().bf_sched;
        if (bs < 0) bs = p.Cin == 64 ? 2 : (p.Cin >= 128 ? 1 : 0);
#define AGX_RBF(MW, NW, OCC)                                                                        \
    (bs == 1 ? launch_rb<MW, NW, 16, 1, OCC, 1>(p, x, w1, b1, w2, b2, y, post_act, st)             \
     : bs == 2 ? launch_rb<MW, NW, 16, 2, OCC, 1>(p, x, w1, b1, w2, b2, y, post_act, st)           \
               : launch_rb<MW, NW, 16, 0, OCC, 1>(p, x, w1, b1, w2, b2, y, post_act, st))
        switch (p.Cin) {
            case 32: return AGX_RBF(1, 4, 2);
            case 64: return AGX_RBF(2, 2, 2);
            case 128: return AGX_RBF(4, 1, 2);
            default: return AGX_RBF(8, 1, 1);
        }
#undef AGX_RBF
    }
    if (tuning().rb_occ == 3 && !c32) {  // diagnostic: cap VGPRs at 168 so that 3 waves/SIMD fit
        switch (p.Cin) {
            case 32: return launch_rb<1, 4, 16, 1, 3>(p, x, w1, b1, w2, b2, y, post_act, st);
            case 64: return launch_rb<2, 2, 16, 2, 3>(p, x, w1, b1, w2, b2, y, post_act, st);
            case 128: return launch_rb<4, 1, 16, 1, 3>(p, x, w1, b1, w2, b2, y, post_act, st);
            default: break;
        }
    }
    switch (p.Cin) {
        case 32: return AGX_RB(1, 4);
        case 64: return AGX_RB(2, 2);
        case 128: return AGX_RB(4, 1);
        default: return AGX_RB(8, 1);
    }
#undef AGX_RB
}

}  // namespace agx
