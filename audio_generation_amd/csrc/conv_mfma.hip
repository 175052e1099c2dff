// fp32-input MFMA implicit-GEMM polyphase convolution for gfx950.
//
// GEMM view of   y[b, co, q*t+p] = sum_{ci,j} Wp[ci*J+j][co*q+p] * x[b, ci, t*s + j*d - P]:
//   M = q*Cout rows (A = packed weights, K-major so a wave's 32 rows are one
//       128-byte line), N = base positions t of one batch item (B = the input
//       tile staged in LDS, time contiguous so the 32 columns of a fragment are
//       32 consecutive LDS dwords), K = Cin*J.
// v_mfma_f32_32x32x2_f32 is an exact binary32 FMA chain in k order at the f32
// vector peak rate (MI355X_MICROARCH: 64 FLOP/clk/SIMD), so results match an
// fp32 reference to rounding order only -- no reduced precision anywhere.
//
// Work decomposition: 256 threads = 4 waves arranged WM x WN; each wave owns
// MW x NW accumulator tiles of 32x32.  The channel loop stages CC input
// channels x (tile + halo) per step; A fragments stream from L2 straight into
// registers (every workgroup reads the same few MB of weights), B fragments
// are one ds_read_b32 each.  The f32 MFMA takes 64 cycles per issue, so operand
// delivery is far off the critical path; the tile shapes below are chosen for
// grid size and halo re-read, not LDS bandwidth.
//
// Replaces: F.pad + F.conv1d (networks/vae.py:34-37), conv_transpose1d + crop
// (vae.py:61-64), interpolate + conv1d (vae.py:86-89) and the elementwise
// LeakyReLU / residual add around them (vae.py:113-117, 130-141, 186-198).
#include "mfma_tile.hpp"

namespace agx {

#ifndef AGX_BF_CONV_SCHED
#define AGX_BF_CONV_SCHED 0
#endif
constexpr int kBfConvSched = AGX_BF_CONV_SCHED;   // schedule of the bf16x3 loop in the plain 1-D convs (mfma_tile.hpp)

template <int MW, int NW, int WM, int WN, int CC, int MODE = 0, int PREC = 0>  // MODE 0: 1-D  1: 2-D row-folded  2: 2-D patches
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvPlan p, int span,
                                                        const float *__restrict__ x,
                                                        const float *__restrict__ wp,
                                                        const float *__restrict__ bias,
                                                        const float *__restrict__ res,
                                                        float *__restrict__ y) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [2][CC][span]
    constexpr int BM = 32 * MW * WM, BN = 32 * NW * WN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    constexpr bool TWO_D = MODE != 0;
    // 2-D layers put the (b, output row / tile) index on grid.x (it can exceed the 65535 limit of grid.z)
    int t0, b, trow = 0, f0 = 0;
    if (MODE == 0) {
        t0 = blockIdx.x * BN;
        b = blockIdx.z;
    } else if (MODE == 1) {                      // blockIdx.x = b * Tout + output row
        t0 = blockIdx.z * BN;
        b = blockIdx.x / p.Tout;
        trow = blockIdx.x - b * p.Tout;
    } else {                                     // blockIdx.x = (b * row groups + rg) * column tiles + ft
        const int nft = (p.Lt + p.pm_WF - 1) / p.pm_WF, nrg = (p.Tt + p.pm_R - 1) / p.pm_R;
        int bx = blockIdx.x;
        const int ft = bx % nft;
        bx /= nft;
        const int rg = bx % nrg;
        b = bx / nrg;
        trow = rg * p.pm_R;
        f0 = ft * p.pm_WF;
        t0 = 0;
    }
    const int m0 = blockIdx.y * BM + wm * (32 * MW);
    const int n0 = wn * (32 * NW);
    const int in0 = t0 * p.s - p.P;

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    // per-lane A row offsets (clamped: rows >= M are computed but never stored)
    int arow[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) arow[i] = min(m0 + i * 32 + li, p.M - 1);
    // per-lane B column offsets inside the staged tile
    int bcol[NW];
    int prow[NW], pcol[NW];  // patch mode: output row / column (inside the tile) of this lane's column k
    const int SW = MODE == 2 ? (p.pm_WF - 1) * p.s + p.J / p.kh : 0;  // patch pitch
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = n0 + k * 32 + li;
        if (MODE == 2) {
            const int r = n / p.pm_WF;
            prow[k] = r;
            pcol[k] = n - r * p.pm_WF;
            bcol[k] = (min(r, p.pm_R - 1) * p.sh) * SW + pcol[k] * p.s + lh * ((CC < 16 ? CC : 16) / 2) * span;
        } else {
            prow[k] = pcol[k] = 0;
            bcol[k] = n * p.s + lh * ((CC < 16 ? CC : 16) / 2) * span;
        }
    }

    if (MODE == 2) {
        const StagerPatch stg{x + size_t(b) * p.cin_real * p.x_cstride, p.x_cstride, p.Lin, p.Tin,
                              trow * p.sh - p.ph, f0 * p.s - p.P, SW, p.J / p.kh, p.cin_real, 1.f / float(SW)};
        if (PREC == 1)
            conv_gemm_rows_bf<MW, NW, (CC < 16 ? 16 : CC)>(acc, xs, stg, reinterpret_cast<const __bf16 *>(wp), p, p.M, span,
                                                           arow, bcol, wave, lane);
        else
            conv_gemm_rows<MW, NW, CC, kSchedDefault>(acc, xs, stg, wp, p, p.M, span, arow, bcol, wave, lane);
    } else if (MODE == 1) {
        const StagerRows<RowMap2D> stg{RowMap2D{x + size_t(b) * p.cin_real * p.x_cstride, p.x_cstride, p.Lin, p.kh,
                                                trow * p.sh - p.ph, p.Tin, p.ncv},
                                       p.Lvalid, in0, p.d};
        conv_gemm_rows<MW, NW, CC, kSchedDefault>(acc, xs, stg, wp, p, p.M, span, arow, bcol, wave, lane);
    } else if (PREC == 1) {
        const StagerRows<RowMap1D> stg{RowMap1D{x + size_t(b) * p.Cin * p.Lin, p.Lin}, p.Lvalid, in0, p.d};
        conv_gemm_rows_bf<MW, NW, (CC < 16 ? 16 : CC), StagerRows<RowMap1D>, kBfConvSched>(
            acc, xs, stg, reinterpret_cast<const __bf16 *>(wp), p, p.M, span, arow, bcol, wave, lane);
    } else {
        const float *xb = x + size_t(b) * p.Cin * p.Lin;
        conv_gemm<MW, NW, CC>(acc, xs, xb, wp, p, p.M, span, in0, arow, bcol, wave, lane);
    }

    // ---- epilogue.  All loads (bias, residual, mask) are issued on clamped addresses before
    // any use so they overlap; only the stores are predicated.
    const bool has_res = (p.epilogue & AGX_EPI_RESIDUAL) != 0, has_mask = (p.epilogue & AGX_EPI_MASK) != 0;
    const bool pre = (p.epilogue & AGX_EPI_LEAKY_PRE) != 0, post = (p.epilogue & AGX_EPI_LEAKY_POST) != 0;
    const bool gelu = (p.epilogue & AGX_EPI_GELU_PRE) != 0;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        int co[16], ph[16], pa[16];  // output channel, column phase, row phase (patch backward-data only)
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + i * 32 + acc_row(r, lh), p.M - 1);
            const int mq = (p.q == 1) ? m : m / p.q;
            ph[r] = m - mq * p.q;
            if (MODE == 2 && p.qh > 1) {
                co[r] = mq / p.qh;
                pa[r] = mq - co[r] * p.qh;
            } else {
                co[r] = mq;
                pa[r] = 0;
            }
            const float bl = (bias ? bias : wp)[co[r]];   // unconditional load, masked below (no branch per element)
            bv[r] = bias ? bl : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            // output position of this lane's column: 1-D / row-folded (row trow, position t) or patch
            // (row trow + prow, column f0 + pcol)
            const int t = MODE == 2 ? f0 + pcol[k] : t0 + n0 + k * 32 + li;
            const int orow = MODE == 2 ? trow + prow[k] : trow;
            const bool col_ok = MODE == 2 ? (prow[k] < p.pm_R && orow < p.Tt && t < p.Lt) : t < p.Lt;
            const int tc = min(t, p.Lt - 1);
            // elements in groups of EG: 16 for the 1-D kernels (all loads in flight at once), 8 for the 2-D
            // modes, whose row / phase bookkeeping would otherwise spill
            constexpr int EG = MODE == 0 ? 16 : 8;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += EG) {
                size_t off[EG];
                float rv[EG], mv[EG];
                bool row_ok[EG];
#pragma unroll
                for (int e = 0; e < EG; ++e) {
                    const int r = r0 + e;
                    const int u = min(max(tc * p.q + ph[r] - p.oshift, 0), p.Lout - 1);
                    const int orr = MODE == 2 ? orow * p.qh + pa[r] - p.oshift_h : orow;  // output row of this element
                    row_ok[e] = !TWO_D || (orr >= 0 && orr < p.Tout);
                    const int orc = TWO_D ? min(max(orr, 0), p.Tout - 1) : 0;
                    off[e] = TWO_D ? (size_t(b) * p.Cout + co[r]) * p.y_cstride + size_t(orc) * p.Lout + u
                                   : (size_t(b) * p.Cout + co[r]) * p.Lout + u;
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < EG; ++e) rv[e] = res[off[e]];
                }
                if (has_mask) {
#pragma unroll
                    for (int e = 0; e < EG; ++e) mv[e] = p.mask[off[e]];
                }
#pragma unroll
                for (int e = 0; e < EG; ++e) {
                    const int r = r0 + e;
                    float v = acc[i][k][r] + bv[r];
                    if (pre) v = leaky(v, p.slope);
                    if (gelu) v = gelu_erf(v);
                    if (has_res) v += rv[e];
                    if (post) v = leaky(v, p.slope);
                    if (has_mask) v = mv[e] > 0.f ? v : v * p.slope;
                    const int u = t * p.q + ph[r] - p.oshift;
                    const bool ok = col_ok && row_ok[e] && (m0 + i * 32 + acc_row(r, lh)) < p.M && u >= 0 && u < p.Lout;
                    if (ok) y[off[e]] = v;
                }
            }
        }
    }
}

// LDS floats per staged channel for a BN-column tile (patch mode also fixes the tile's rows x columns).
static int tile_span(const ConvPlan &p, int BN, int *R, int *WF) {
    if (!p.pm_R) return (BN - 1) * p.s + (p.J - 1) * p.d + 1;
    // tile = R output rows x WF output columns (R * WF <= BN); backward-data of strided layers has 2^k + 1 columns,
    // so the widest tile is often not the best.
    // candidates: the whole row (when it fits) and the power-of-two splits of BN; the cost of a split is the number
    // of tiles (each costs BN columns of MFMA work whatever it covers) -- a 65-column base grid on "whole row" tiles
    // of 128 was half empty (3.15 -> 2.0 ms on the 128 -> 256 strided layer's backward-data)
    int wf = p.Lt < BN ? (p.Lt < 1 ? 1 : p.Lt) : BN, r = BN / wf;
    if (r > p.Tt) r = p.Tt;
    if (r < 1) r = 1;
    long best = long(ceil_div(p.Lt, wf)) * ceil_div(p.Tt, r);
    long best_staged = best * ((r - 1) * p.sh + p.kh) * ((wf - 1) * p.s + p.J / p.kh);
    {
        const int kwt = p.J / p.kh;
        for (int cand = BN; cand >= 8; cand /= 2) {
            int rr = BN / cand;
            if (rr > p.Tt) rr = p.Tt;
            const long tiles = long(ceil_div(p.Lt, cand)) * ceil_div(p.Tt, rr);
            const long staged = tiles * ((rr - 1) * p.sh + p.kh) * ((cand - 1) * p.s + kwt);
            if (tiles < best || (tuning().patch_tie && tiles == best && cand >= 16 && staged < best_staged)) {
                best = tiles;
                best_staged = staged;
                wf = cand;
                r = rr;
            }
        }
    }
    if (r > p.Tt) r = p.Tt;
    if (r < 1) r = 1;
    if (R) *R = r;
    if (WF) *WF = wf;
    return ((r - 1) * p.sh + p.kh) * ((wf - 1) * p.s + p.J / p.kh);
}

template <int MW, int NW, int WM, int WN, int CC, int MODE = 0, int PREC = 0>
static int launch_variant(const ConvPlan &p0, const float *x, const float *wp, const float *bias,
                          const float *res, float *y, hipStream_t st) {
    constexpr int BM = 32 * MW * WM, BN = 32 * NW * WN;
    ConvPlan p = p0;
    const int span = tile_span(p0, BN, &p.pm_R, &p.pm_WF);
    const size_t lds = size_t(2) * CC * span * sizeof(float);  // double-buffered input tile
    if (lds > 160 * 1024) return fail(AGX_ERR_UNSUPPORTED, "conv_mfma: tile needs %zu B of LDS", lds);
    auto kern = conv_mfma_kernel<MW, NW, WM, WN, CC, MODE, PREC>;
    static DeviceOnce once;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, nullptr, "conv_mfma")) return rc;
    dim3 grid(ceil_div(p.Lt, BN), ceil_div(p.M, BM), p.B), block(256);
    if (MODE == 1) grid = dim3(p.B * p.Tout, ceil_div(p.M, BM), ceil_div(p.Lt, BN));
    if (MODE == 2)
        grid = dim3(unsigned(int64_t(p.B) * ceil_div(p.Tt, p.pm_R) * ceil_div(p.Lt, p.pm_WF)), ceil_div(p.M, BM), 1);
    if (grid.y > 65535 || grid.z > 65535) return fail(AGX_ERR_BAD_SHAPE, "conv_mfma: grid too large");
    hipLaunchKernelGGL(kern, grid, block, lds, st, p, span, x, wp, bias, res, y);
    return check_launch("conv_mfma");
}

// ---- tile variant table -----------------------------------------------------------
// Preference order per M class; a variant is eligible when its input tile fits LDS.
// <= 72 KB keeps two workgroups resident per CU, which is what hides the staging phase.
struct Variant {
    int mw, nw, wm, wn, cc;
    const char *name;
    int (*launch)(const ConvPlan &, const float *, const float *, const float *, const float *, float *,
                  hipStream_t);
    int (*launch2d)(const ConvPlan &, const float *, const float *, const float *, const float *, float *,
                    hipStream_t);  // row-folded 2-D; nullptr: no 2-D instantiation of this tile
    int (*launch_patch)(const ConvPlan &, const float *, const float *, const float *, const float *, float *,
                        hipStream_t);  // patch 2-D
    int (*launch_bf)(const ConvPlan &, const float *, const float *, const float *, const float *, float *,
                     hipStream_t);     // 1-D bf16x3 (16-channel chunks only)
    int (*launch_patch_bf)(const ConvPlan &, const float *, const float *, const float *, const float *, float *,
                           hipStream_t);   // patch 2-D bf16x3
};

#define AGX_VARIANT(MW, NW, WM, WN, CC) \
    { MW, NW, WM, WN, CC, "conv_mfma<" #MW "," #NW "," #WM "," #WN "," #CC ">", launch_variant<MW, NW, WM, WN, CC>, nullptr, nullptr, nullptr, nullptr }
#define AGX_VARIANT2(MW, NW, WM, WN, CC) \
    { MW, NW, WM, WN, CC, "conv_mfma<" #MW "," #NW "," #WM "," #WN "," #CC ">", launch_variant<MW, NW, WM, WN, CC>, \
      launch_variant<MW, NW, WM, WN, CC, 1>, launch_variant<MW, NW, WM, WN, CC, 2>, nullptr, nullptr }
#define AGX_VARIANT3(MW, NW, WM, WN, CC) \
    { MW, NW, WM, WN, CC, "conv_mfma<" #MW "," #NW "," #WM "," #WN "," #CC ">", launch_variant<MW, NW, WM, WN, CC>, \
      launch_variant<MW, NW, WM, WN, CC, 1>, launch_variant<MW, NW, WM, WN, CC, 2>, launch_variant<MW, NW, WM, WN, CC, 0, 1>, \
      launch_variant<MW, NW, WM, WN, CC, 2, 1> }

static const Variant kWide[] = {AGX_VARIANT3(2, 2, 2, 2, 16), AGX_VARIANT2(2, 2, 2, 2, 8), AGX_VARIANT(2, 2, 2, 2, 32)};
// 128 x 64 tiles for short signals: twice the workgroups when the 128 x 128 grid would leave
// a CU with a single resident workgroup (nothing to overlap staging / epilogue with).
static const Variant kWideShort[] = {AGX_VARIANT3(1, 2, 4, 1, 16), AGX_VARIANT2(1, 2, 4, 1, 8), AGX_VARIANT(1, 2, 4, 1, 32)};
static const Variant kWideAlt[] = {AGX_VARIANT(1, 4, 4, 1, 16), AGX_VARIANT(1, 4, 4, 1, 8), AGX_VARIANT(1, 4, 4, 1, 32)};
static const Variant kMid[] = {AGX_VARIANT3(2, 2, 1, 4, 16), AGX_VARIANT2(2, 2, 1, 4, 8), AGX_VARIANT3(2, 1, 1, 4, 16),
                               AGX_VARIANT2(2, 1, 1, 4, 8),  AGX_VARIANT(2, 2, 1, 4, 32)};
static const Variant kNarrow[] = {AGX_VARIANT3(1, 4, 1, 4, 16), AGX_VARIANT2(1, 4, 1, 4, 8), AGX_VARIANT3(1, 1, 1, 4, 16),
                                  AGX_VARIANT2(1, 1, 1, 4, 8),  AGX_VARIANT(1, 4, 1, 4, 32)};

static size_t variant_lds(const Variant &v, const ConvPlan &p) {
    const int bn = 32 * v.nw * v.wn;
    return size_t(2) * v.cc * size_t(tile_span(p, bn, nullptr, nullptr)) * sizeof(float);
}

static const Variant *pick(const Variant *list, int n, const ConvPlan &p, int want_cc) {
    if (want_cc) {
        for (int i = 0; i < n; ++i)
            if (list[i].cc == want_cc && (want_cc != 32 || p.Cin % 32 == 0) && variant_lds(list[i], p) <= 160 * 1024)
                return &list[i];
        return nullptr;
    }
    // (32-channel chunks measured no better than 16 on any config-S shape: kept as forced variants only)
    for (int i = 0; i < n; ++i)
        if (list[i].cc != 32 && variant_lds(list[i], p) <= 72 * 1024) return &list[i];
    for (int i = 0; i < n; ++i)
        if (list[i].cc != 32 && variant_lds(list[i], p) <= 160 * 1024) return &list[i];
    return nullptr;
}

static const Variant *select_variant(const ConvPlan &p) {
    if (p.Cin % 16 != 0 || p.M < (p.pm_R ? 8 : 32) || p.G != 1) return nullptr;   // patch tiles tolerate few rows (clamped)
    const Variant *list = p.M >= 128 ? kWide : (p.M >= 64 ? kMid : kNarrow);
    const int n = p.M >= 128 ? 3 : 5;
    if (p.M >= 128 && tuning().conv_shape == 1) list = kWideAlt;
    const Variant *v = pick(list, n, p, p.prec ? 16 : tuning().conv_cc);
    // row-folded 2-D layers with few rows (the 2-channel 7x7 first conv of the STFT discriminators) on narrow maps: a tile
    // is ONE output row, so the 512-column tile of kNarrow's first entry is 25-50 % full on 128 / 256-column maps
    // (4.4 ms against 1.1 ms for the same work on 1024 columns): 128-column tiles there
    if (v && list == kNarrow && !p.pm_R && (p.kh > 1 || p.Tout > 1) && p.Lt <= 256 && !p.prec && !tuning().conv_cc) {
        const Variant *vn = pick(kNarrow + 2, 2, p, 0);
        if (vn) v = vn;
    }
    if (v && p.M >= 128 && tuning().conv_short && !p.pm_R) {
        // Short signals: the same channel chunk (= the same summation order, so results do not depend on
        // the batch size or the signal length) on 128 x 64 tiles.
        const long wgs = long(ceil_div(p.Lt, 128)) * ceil_div(p.M, 128) * p.B * p.Tout;
        if (wgs < 2 * 256) {
            const Variant *vs = pick(kWideShort, 3, p, v->cc);
            if (vs) v = vs;
        }
    }
    return v;
}

bool conv_mfma_supported(const ConvPlan &p) { return select_variant(p) != nullptr; }

const char *conv_mfma_variant(const ConvPlan &p) {
    const Variant *v = select_variant(p);
    return v ? v->name : "conv_mfma<unsupported>";
}

int launch_conv_mfma(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                     const float *res, float *y, hipStream_t st) {
    const Variant *v = select_variant(p);
    if (!v)
        return fail(AGX_ERR_UNSUPPORTED,
                    "conv_mfma: needs Cin %% 16 == 0, q*Cout >= 32 and an input tile that fits LDS (Cin=%d M=%d s=%d J=%d d=%d)",
                    p.Cin, p.M, p.s, p.J, p.d);
    if (p.prec) {
        if (p.pm_R && v->launch_patch_bf) return v->launch_patch_bf(p, x, wp, bias, res, y, st);
        if (!v->launch_bf || p.pm_R || p.kh > 1 || p.Tout > 1)
            return fail(AGX_ERR_UNSUPPORTED, "conv_mfma: no bf16x3 instantiation for this layer (%s)", v->name);
        return v->launch_bf(p, x, wp, bias, res, y, st);
    }
    if (p.pm_R) {
        if (!v->launch_patch) return fail(AGX_ERR_UNSUPPORTED, "conv_mfma: no 2-D instantiation of %s", v->name);
        return v->launch_patch(p, x, wp, bias, res, y, st);
    }
    if (p.kh > 1 || p.Tout > 1 || p.ncv != p.Cin) {
        if (!v->launch2d) return fail(AGX_ERR_UNSUPPORTED, "conv_mfma: no 2-D instantiation of %s", v->name);
        return v->launch2d(p, x, wp, bias, res, y, st);
    }
    return v->launch(p, x, wp, bias, res, y, st);
}

}  // namespace agx
