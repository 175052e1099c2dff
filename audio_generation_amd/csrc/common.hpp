// Shared host-side helpers of libagx (error state, launch checks, lowered conv plan).
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "agx.h"

namespace agx {

// Thread-local last-error message (the only mutable global state of the library besides the measurement knobs below -- relaxed atomics --
// and, in the PROBE build only, the RVQ stamp buffer).
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
const char *last_error();

// A measurement knob: a process-wide int read at launch time and written by agx_set_tuning.  Relaxed atomic (round 4): a host
// thread changing a knob while another thread launches is a defined -- if pointless -- thing to do; no ordering is implied.
struct Knob {
    std::atomic<int> v;
    explicit Knob(int x) : v(x) {}
    operator int() const { return v.load(std::memory_order_relaxed); }
    Knob &operator=(int x) {
        v.store(x, std::memory_order_relaxed);
        return *this;
    }
};

// diagnostic knobs (agx_set_tuning)
struct Tuning {
    Knob rb_cc{16};   // channels per LDS chunk of the fused residual block (16 or 32)
    Knob rb_wgs{0};   // 1..3: cap resident workgroups per CU of the fused residual block (0 = natural)
    Knob patch_tie{1};  // 2-D patch tiles: 1 = among the R x WF splits with the same padded area take the one that stages the fewest
                        // input elements (tall tiles share the row halo), 0 = always the widest
    Knob bf_sched{-1};  // (-1 = per-shape table) schedule of the bf16x3 main loop in the fused residual block: 0 split after the MFMAs,
                        // 1 the same with MFMA / VALU interleave hints, 2 split before the MFMAs
    Knob dw2_shared{2}; // conv2d weight gradient: workgroup-shared operand slots + one barrier per item for 1 = the 128-row tiles, 2 = also the
                        // 64- / 32-row tiles (64 -> 64 3 x 3: 81 -> 90 TFLOP/s since the DMA issue is cheap), 0 = wave-private buffers
    Knob c2b3_sl{0};    // conv2d_b3 tile split R x 2^SL: 0 = cost model (padded area x (matrix time + staging rounds)), -1 = least padded area
                        // with ties to the widest rows (the first rule), 3..7 = forced where the tile has it
    Knob dw2_prepad{1}; // conv2d weight gradient of maps narrower than 32 columns: 1 = the shared kernel on zero-padded flattened copies, 0 = staged kernel
    Knob dw2_bf{1};     // conv2d weight gradient of bf16x3 descriptors on the shared kernel: 1 = bf16x3 contraction, 0 = fp32 (exact)
    Knob dw2_direct{2}; // conv2d weight gradient on the barrier-free LDS-DMA kernel: 1 = stride-1 "same" layers, 2 = also the column-strided
                        // layers (x read through its column-phase planes), 0 = the staged kernel everywhere
    Knob dw_direct{3};  // 1-D weight gradient on the barrier-free kernel: 3 = every dense layer (strided / transposed ones through a
                        // phase-split copy of x / dy), 2 = the stride-1 layers, 1 = the k = 1 layers only, 0 = none
    Knob dw1_wgs{768};  // workgroups the 1-D LDS-free weight-gradient kernel aims for
    Knob dw_wgs{1536};  // workgroups the conv2d weight-gradient kernel aims for (slices = dw_wgs / tiles)
    Knob dw_xcd{0};     // conv2d weight gradient: 1 = XCD-aware block order (all tiles of a contraction slice on one XCD's L2); measured
                        // WORSE (128 -> 128 3 x 3: 106.6 -> 95.1 TFLOP/s): the slice's operands are better spread over the eight L2s
    Knob conv_cc{0};    // diagnostic: force the LDS chunk (8/16/32 channels) of the MFMA conv; 0 = table
    Knob conv_shape{0}; // diagnostic: 1 = 128x128 conv tiles as 4 row-waves x (1x4) fragments
    Knob conv_short{1}; // 1: 128x64 conv tiles when the 128x128 grid is under two workgroups per CU
    Knob rb_occ{2};    // 3: build of the fused residual block capped at 168 VGPRs (3 waves/SIMD)
    Knob conv_impl{1};  // resampling / stride-1 1-D layers: 1 = persistent ring kernel (conv_p.hip) where it applies, 0 = conv_mfma.hip
    Knob rb_impl{1};    // fused residual block: 1 = persistent ring kernel (resblock_p.hip) where it applies, 0 = resblock_mfma.hip
    Knob b3_dbg{0};    // DIAGNOSTIC switchboard of round 3 (default 0 everywhere): 1 = resblock_b3 alternates (C = 64 as one 64 x 512 workgroup
                       // per CU, C = 128 as 128 x 256 with double-buffered planes), 2 = resblock_b3 without the GEMM1 priority.  Values
                       // 7 / 8 / 9 (RVQ score bound: accumulation term x 4 / x 0 / candidate counts in the squared-error output) act in
                       // the PROBE build of rvq.hip only (-DAGX_RVQ_PROBE); agx_set_tuning refuses them otherwise
    Knob rvq_verify{0}; // DEBUG: 1 = rvq_forward is followed by the checker kernel (full defining search of every (frame, stage), agx_rvq_verify_counts)
    Knob rb_sched{-1}; // phase scheduling of the fused residual block (mfma_tile.hpp: 0 / 1 / 2; -1 = per-shape table)
};
Tuning &tuning();

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return AGX_OK;
}

// Per-device launch preparation of one kernel.  The dynamic-LDS attribute (> 64 KB) and the CU count are properties of a
// DEVICE: round 3 cached them in function-local statics once per PROCESS, so a process that later used a second device never
// raised the limit there (launches needing > 64 KB of LDS failed) and sized its persistent grid with the first device's CU
// count; two threads making the first call also raced.  `once` is one static object per kernel instantiation: bit d = done on
// device d (atomic; devices >= 64 are prepared on every launch).  Returns AGX_OK or a failure code; *n_cu may be NULL.
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
};
int device_cu_count(int dev, int *n_cu);     // cached per device (core.hip)
inline int prepare_kernel(const void *kern, DeviceOnce &once, int lds_limit_bytes, int *n_cu, const char *what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(AGX_ERR_LAUNCH, "%s: cannot query the device", what);
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (lds_limit_bytes > 0 && (bit == 0 || !(once.done.load(std::memory_order_acquire) & bit))) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit_bytes);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
        once.done.fetch_or(bit, std::memory_order_release);
    }
    return n_cu ? device_cu_count(dev, n_cu) : AGX_OK;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Every layer kind of agx_conv_desc lowers to one "polyphase convolution":
//   y[b, co, q*t + p] = epi( bias[co] + sum_{ci, j<J} Wp[ci*J + j][co*q + p] * x[b, ci, t*s + j*d - P] )
// for t in [0, Lt), p in [0, q), output positions >= Lout dropped and input
// positions outside [0, Lvalid) read as zero.
struct ConvPlan {
    int B, Cin, Cout, Lin;
    int Lt;      // base positions
    int q;       // output phases per base position
    int J;       // taps per phase
    int s, d;    // input step per base position / per tap
    int P;       // left offset
    int Lout;    // output length
    int Lvalid;  // input positions >= Lvalid are zero (crop by a negative right pad)
    int M;       // q * Cout  (rows of the implicit GEMM)
    int epilogue;
    float slope;
    int oshift;         // output index = q*t + p - oshift (backward-data of strided convs), 0 otherwise
    const float *mask;  // AGX_EPI_MASK: v *= (mask[o] > 0 ? 1 : slope)  (LeakyReLU gradient), else unused
    // grouped layers (direct kernel only): Cin / Cout are the layer totals, the packed image has Cin / G
    // channels and output row m reads input channels (m / (Cout / G)) * (Cin / G) + [0, Cin / G)
    int G;
    // 2-D layers (conv2d.hip) as 1-D convs along the last axis: virtual channel c' = ci * kh + dh (Cin counts
    // those), virtual batch z = b * Tout + t; input row of (z, c') = t * sh - ph + dh (zero outside [0, Tin)).
    // 1-D layers: kh = 1, Tin = Tout = 1, x_cstride = Lin, y_cstride = Lout.
    int kh, sh, ph, Tin, Tout;
    int ncv, cin_real;             // virtual channels actually present (Cin rounds them up to 16) / real input channels
    int64_t x_cstride, y_cstride;  // elements between consecutive real channels of x / y
    // patch mode (conv2d.hip; 0 = off): Cin counts REAL channels, tap j = dh * kw + dw (J = kh * kw); a tile is
    // pm_R output rows x pm_WF output columns, chosen by the launcher
    int pm_R, pm_WF;
    // patch mode, backward-data of strided layers: qh output-row phases per base row (rows m = (co*qh + a)*q + c),
    // output row = qh * base row + a - oshift_h; Tt base rows (forward: qh = 1, oshift_h = 0, Tt = Tout)
    int qh, oshift_h, Tt;
    int64_t tile_off;   // floats from the packed image to the layer's tile image (below), -1: the layer has none
    int prec;   // 0: fp32 MFMA (exact fp32 FMA chain); 1: bf16x3 on the bf16 MFMA (mfma_tile.hpp), 1-D MFMA kernels only
};

// Packed weight image ("group-K-major"): channels in groups of 16,
//   Wp[((ci / 16) * J + j) * M + m][ci % 16]
// so that (a) a lane's A operands for consecutive k-steps are contiguous (one 16-byte
// load feeds 4 MFMAs) and (b) one (16-channel group, tap) slab is a contiguous M*64 B block.
constexpr int kWG = 16;  // channels per weight group
__host__ __device__ inline int64_t packed_weight_floats(int Cin, int J, int M) {
    return int64_t((Cin + kWG - 1) / kWG) * J * M * kWG;
}
// bf16x3 image (mfma_tile.hpp): three bf16 planes per 16-channel group = 48 bf16 = 24 floats per (group, tap, row)
__host__ __device__ inline int64_t packed_weight_floats_bf(int Cin, int J, int M) {
    return int64_t((Cin + kWG - 1) / kWG) * J * M * 24;
}
__host__ __device__ inline size_t packed_weight_index(int ci, int j, int m, int J, int M) {
    return (size_t(ci / kWG) * J + j) * M * kWG + size_t(m) * kWG + (ci % kWG);
}

// "Tile image" (resblock_p.hip): a second copy of a dense layer's folded fp32 weights, laid out so that a chunk of
// channels is ONE contiguous block that LDS-DMA drops into LDS as it stands and the MFMA A fragments come out of
// with conflict-free 16-byte reads:   Wt[((ci / 4) * J + j) * M + m][ci % 4]
// It follows the standard image and the dim0 scale scratch inside the same packed buffer
// (agx_conv_packed_floats accounts for it), for the layers tile_image_eligible() names.
__host__ __device__ inline int64_t tile_image_floats(int Cin, int J, int M) {
    return int64_t((Cin + 3) / 4) * J * M * 4;
}
__host__ __device__ inline size_t tile_image_index(int ci, int j, int m, int J, int M) {
    return ((size_t(ci / 4) * J + j) * M + m) * 4 + (ci % 4);
}
// Geometry class of a layer the persistent conv kernel is instantiated for (conv_p.hip), 0 = none.
int conv_p_geometry(const ConvPlan &p);
int conv_p2d_geometry(const ConvPlan &p);   // conv_p.hip: ring form of a patch-mode Conv2d plan (0 = none)
// Layers that get a tile image: the stride-1 causal layers of the fused residual block (the dilated k = 7 conv and the
// k = 1 conv, C in {32,64,128,256}) and the resampling / stride-1 layers of conv_p.hip.
inline bool tile_image_eligible(const ConvPlan &p, int kind) {
    if (p.prec != 0 || p.G != 1) return false;
    if (kind == AGX_CONV_CAUSAL && p.s == 1 && p.q == 1 && p.Cin == p.Cout &&
        (p.Cin == 32 || p.Cin == 64 || p.Cin == 128 || p.Cin == 256) && (p.J == 7 || p.J == 1))
        return true;
    return conv_p_geometry(p) != 0;
}

// "B3 tile image" (resblock_b3.hip): the bf16x3 weights of a dense layer in the layout the bf16x3 ring kernels DMA and read:
//   Wb[((g * J + j) * 3 + plane) * 2 + lh][m][8]   bf16,  g = ci / 16, slot e of lane half lh = channel 16 g + 8 lh + e
// -- one (group, tap) phase = one contiguous 96 M-byte block, an A fragment = one conflict-free 16-byte LDS read.  k = 1
// layers (the second conv of the fused block) are stored in GEMM2 order instead: slot e of lane half lh = channel
// 16 g + 4 lh + e (e < 4), 16 g + 8 + 4 lh + (e - 4) (e >= 4) -- the hidden channels a lane holds in accumulator
// registers 8 (g % 2) .. + 7.  Same size as the standard bf16x3 image; follows it and the dim0 scale scratch.
int conv_b3_geometry(const ConvPlan &p);   // conv_b3.hip: bf16x3 ring form of a stride-1 polyphase layer (0 = none)
int conv2d_b3_geometry(const ConvPlan &p); // conv_b3.hip: bf16x3 ring form of a patch-mode Conv2d plan (3 x 3, stride 1, "same"; 0 = none)
inline bool b3_image_eligible(const ConvPlan &p, int kind) {
    if (p.prec != 1 || p.G != 1) return false;
    if (kind == AGX_CONV_CAUSAL && p.s == 1 && p.q == 1 && p.Cin == p.Cout &&
        (p.Cin == 32 || p.Cin == 64 || p.Cin == 128 || p.Cin == 256) && (p.J == 7 || p.J == 1))
        return true;
    return conv_b3_geometry(p) != 0;
}

// Lower a descriptor; returns AGX_OK or an error (message set).
int lower_conv(const agx_conv_desc *d, ConvPlan *p);
// Plan of the backward-data op of layer `d` (gradient w.r.t. the layer input, given the gradient
// w.r.t. its output): the same polyphase form with input/output channels swapped.
int lower_conv_bwd_data(const agx_conv_desc *d, ConvPlan *p);
// 2-D layer (torch.nn.Conv2d, zero padding) as a 1-D conv over virtual channels / batch (see ConvPlan).
int lower_conv2d(const agx_conv2d_desc *d, ConvPlan *p);
int lower_conv2d_bwd_data(const agx_conv2d_desc *d, ConvPlan *p);

}  // namespace agx
