/* agx.h -- flat C ABI of the MI355X-native neural-audio-codec forward path.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own: its hot
 * path is a chain of ATen calls made from torch.nn.Modules.  Each entry point
 * below replaces the ATen call sequence of one reference function (cited per
 * declaration, paths relative to the reference tree); INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory the caller owns; the library
 *     never allocates, frees or retains a pointer past the call;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     launches are asynchronous on it and the library never synchronises;
 *   - return value 0 = success, negative = error (AGX_ERR_*); the message of the
 *     last error of the calling thread is returned by agx_last_error();
 *   - tensors are contiguous fp32, "NCL" (batch, channel, time) unless said
 *     otherwise; indices are int64 to match torch.argmin.
 */
#ifndef AGX_H
#define AGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_VERSION 120 /* 120: activation planes (agx_planes_bytes / agx_planes_split / agx_conv_forward_planes / agx_conv_planes_supported): pre-split bf16x3 input of the decoder's resampling convs; conv_p one-phase geometries (k = 1, "same" k = 11 / 3) with GELU / residual epilogues; 119: agx_rvq_verify_counts + knob rvq_verify (debug: the full defining search beside the fast path); agx_rvq_debug_stamps and the b3_dbg 7/8/9 bound knobs exist in the probe build only; 118: agx_feature_means(_backward) (the feature-matching pair in one pass); 117: agx_rvq_debug_stamps (diagnostic); 116: agx_multires_backward, agx_layernorm_ct one-pass kernel (same signature); agx_rvq_forward (legacy form) needs the workspace of agx_rvq_workspace_bytes since 114; 115: agx_rvq_ema_stats, agx_conv2d_bwd_data_kernel_name; 114: agx_attention_alibi_backward_ex (any T), agx_rvq_forward_ex; 113: tile images (resblock_p / conv_p), agx_attention_alibi_ex, agx_sizeof_*; 0.1.1: agx_conv_desc gained groups / padding (zero = old behaviour); 111: resample, conv2d column split */

#define AGX_OK 0
#define AGX_ERR_BAD_SHAPE (-1)
#define AGX_ERR_NULL_POINTER (-2)
#define AGX_ERR_WORKSPACE (-3)
#define AGX_ERR_LAUNCH (-4)
#define AGX_ERR_UNSUPPORTED (-5)

int agx_version(void);
const char *agx_last_error(void);

/* sizeof() of the two descriptor structs as THIS library was compiled: a binding (ctypes / cgo / JNI stub) asserts
 * its own struct size against these once at load time, so that a descriptor that gained a field cannot be read
 * past the end of a caller's shorter struct. */
int32_t agx_sizeof_conv_desc(void);
int32_t agx_sizeof_conv2d_desc(void);

/* Diagnostic tuning knobs (A/B experiments from one process; defaults are the
 * shipped configuration).  Unknown names return AGX_ERR_BAD_SHAPE.  Process-wide relaxed atomics read at launch
 * time: changing one while another thread launches is defined but pointless -- set them before the first launch.
 *   "rb_cc"  16 | 32   channels per LDS chunk of the fused residual block
 *   "rb_wgs" 0 | 1..3  cap on resident workgroups per CU of the fused residual block (0 = natural)
 *   "rb_sched" 0|1|2   phase scheduling of the fused residual block (csrc/mfma_tile.hpp)
 *   "rb_occ" 2|3       3: fused residual block built under a 168-VGPR cap (3 waves/SIMD); no gain measured
 *   "conv_short" 0|1   128x64 MFMA conv tiles when the 128x128 grid is under two workgroups per CU (default 1)
 *   "conv_cc" 0|8|16|32  force the LDS channel chunk of the MFMA conv (0 = table)
 *   "patch_tie" 0|1      2-D patch tiles: tie between equally padded R x WF splits goes to the fewest staged elements (1) or the widest (0)
 *   "bf_sched" -1|0|1|2  schedule of the bf16x3 main loop of the fused residual block (-1 = per-shape table)
 *   "dw_direct" 0..3   1-D weight gradient on the barrier-free LDS-DMA kernel: 3 every dense layer (default; strided / transposed
 *                      ones through a phase-split copy of x / dy), 2 the stride-1 layers, 1 the k = 1 layers, 0 none
 *   "dw2_direct" 0|1|2 conv2d weight gradient on the barrier-free LDS-DMA kernel: 1 stride-1 "same" layers, 2 (default) also the
 *                      column-strided layers (x through its column-phase planes), 0 the staged kernel everywhere
 *   "dw1_wgs" N        workgroups the 1-D barrier-free weight-gradient kernel aims for (default 768)
 *   "dw_wgs" n         workgroups the conv2d weight-gradient kernel aims for (default 1536)
 *   "dw2_prepad" 0|1   conv2d weight gradient of feature maps narrower than 32 columns: 1 (default) the shared kernel on zero-padded,
 *                      phase-split, flattened copies of x and dy; 0 the staged kernel
 *   "c2b3_sl" -1|0|3..7  bf16x3 Conv2d ring kernel, split of a tile into R rows of 2^SL columns: 0 (default) the split with the least
 *                      padded area x (matrix time + input staging rounds), -1 the least padded area alone, 3..7 forced
 *   "dw2_bf" 0|1       conv2d weight gradient of AGX_IMPL_MFMA_BF16X3 descriptors on the shared kernel: 1 (default) bf16x3
 *                      contraction (both operands split in registers), 0 the fp32 contraction
 *   "dw_xcd" 0|1       conv2d weight gradient: 0 (default); 1 = XCD-aware block order -- the tiles of one contraction slice share
 *                      an L2 (measured SLOWER: 106.6 -> 95.1 TFLOP/s on the 128 -> 128 3 x 3 layer)
 *   "rvq_verify" 0|1   DEBUG: 1 = every agx_rvq_forward(_ex) launch is followed by a checker kernel that runs the full DEFINING
 *                      search (binary64, oracle/rvq_exact.c) of every (frame, stage) on the residual the fast path searched and
 *                      counts the indices that differ (agx_rvq_verify_counts); tens of milliseconds per call at config S
 *   "conv_shape" 0|1   1: 128x128 conv tiles as four row-waves of 1x4 fragments
 *   "rb_impl" 0|1      fused residual block: 1 (default) the persistent ring kernel (csrc/resblock_p.hip) where it applies,
 *                      0 the first kernel (csrc/resblock_mfma.hip) everywhere
 *   "conv_impl" 0|1    resampling / stride-1 1-D layers and the Conv2d layers the ring kernel has a geometry for (forward and
 *                      backward-data): 1 (default) the persistent ring kernel (csrc/conv_p.hip), 0 conv_mfma.hip
 *   "dw2_shared" 0|1|2 conv2d weight gradient: the tiles fetch their operands once per workgroup (two LDS slots, one barrier per
 *                      item): 1 the 128 x 128 tiles, 2 (default) also the 64- and 32-row tiles, 0 wave-private buffers
 *   (the diagnostics of the experiments DESIGN 4.11 lists as dropped -- a DMA-only wave, start staggers, deferred stores --
 *    were removed together with their code)                                                                */
int agx_set_tuning(const char *name, int32_t value);
int agx_get_tuning(const char *name);

/* ------------------------------------------------------------------------- *
 * Convolutions                                                               *
 * ------------------------------------------------------------------------- */

/* Which reference layer a convolution descriptor stands for. */
#define AGX_CONV_CAUSAL 0   /* CausalConv1d,          networks/vae.py:14-43  */
#define AGX_CONV_TRANSPOSED 1 /* CausalConvT1d,       networks/vae.py:45-64  */
#define AGX_CONV_UPSAMPLE 2 /* CausalUpsampleConv1d,  networks/vae.py:66-89  */
#define AGX_CONV_SAME 3     /* Conv1d(padding="same"), networks/wavelets.py:193-201 */
#define AGX_CONV_PADDED 4 /* torch.nn.Conv1d(padding=p, stride, dilation, groups): discriminator.py:33-41 */

/* Which kernel family executes it (AGX_IMPL_AUTO picks by shape). */
#define AGX_IMPL_AUTO 0
#define AGX_IMPL_DIRECT 1 /* fp32 VALU, any shape                            */
#define AGX_IMPL_MFMA 2   /* fp32-input MFMA implicit GEMM (exact fp32 FMA chain) */
#define AGX_IMPL_MFMA_BF16X3 3 /* operands split into 3 bf16 pieces, 6 bf16 MFMAs per product block: fp32-class
                               * accuracy (~1e-7 rel.), not the bitwise fp32 chain; dense 1-D layers with
                               * Cin % 16 == 0 and q*Cout >= 32; its own packed image (agx_conv_pack with this impl) */

/* Epilogue flags (fused into the conv kernel; all optional). */
#define AGX_EPI_LEAKY_PRE 1  /* LeakyReLU(slope) on (acc + bias)   vae.py:99,125,156 */
#define AGX_EPI_RESIDUAL 2   /* += res[b,co,t]                     vae.py:117        */
#define AGX_EPI_LEAKY_POST 4 /* LeakyReLU(slope) after the add     vae.py:131-134    */
#define AGX_EPI_GELU_PRE 8   /* exact (erf) GELU on (acc + bias)   transformers.py:216, wavelets.py:96 */
#define AGX_EPI_MASK 16      /* backward only: v *= (mask[o] > 0 ? 1 : slope), the LeakyReLU gradient     */

typedef struct agx_conv_desc {
    int32_t kind;      /* AGX_CONV_*                                              */
    int32_t batch;     /* B                                                       */
    int32_t c_in;      /* input channels                                          */
    int32_t c_out;     /* output channels                                         */
    int32_t l_in;      /* input length                                            */
    int32_t kernel;    /* K (reference kernel_size)                               */
    int32_t stride;    /* conv stride (CAUSAL) or up-factor (TRANSPOSED/UPSAMPLE) */
    int32_t dilation;  /* CAUSAL only, else 1                                     */
    int32_t epilogue;  /* OR of AGX_EPI_*                                         */
    float slope;       /* LeakyReLU negative slope (reference: 0.1)               */
    int32_t impl;      /* AGX_IMPL_*                                              */
    int32_t groups;    /* PADDED only: conv groups (0 or 1 = dense); grouped layers run on the direct kernel */
    int32_t padding;   /* PADDED only: zeros on both sides (torch Conv1d padding=)  */
} agx_conv_desc;

/* Output length of the layer exactly as the reference computes it
 * (vae.py:32-43 incl. _calc_extra_pad; vae.py:58-64; vae.py:86-89).  <0 on error. */
int64_t agx_conv_out_len(const agx_conv_desc *d);

/* Number of floats of the packed weight image of this layer. */
int64_t agx_conv_packed_floats(const agx_conv_desc *d);

/* Fold weight-norm and repack one conv layer's weight for the kernels.
 *   v : the reference parameter `weight_v` (or the plain `weight` when g == NULL),
 *       torch layout: (c_out, c_in, K) for CAUSAL/UPSAMPLE/SAME, (c_in, c_out, K)
 *       for TRANSPOSED;
 *   g : `weight_g` (dim0,1,1) or NULL.  w = g * v / ||v||, norm over all dims but
 *       0 -- utils.py:34-42 (torch.nn.utils.weight_norm, dim=0);
 *   packed : agx_conv_packed_floats(d) floats.  UPSAMPLE layers are stored as the
 *       `stride` polyphase 3-tap filters of the nearest-upsample + conv pair. */
int agx_conv_pack(const agx_conv_desc *d, const float *v, const float *g, float *packed,
                  void *stream);

/* y = epilogue(conv(x) + bias).  x (B,c_in,l_in); y,res (B,c_out,l_out);
 * bias (c_out) or NULL; res only read when AGX_EPI_RESIDUAL is set.
 * Replaces F.pad + conv1d (vae.py:34-37), conv_transpose1d + crop (vae.py:61-64),
 * interpolate + conv1d (vae.py:86-89). */
int agx_conv_forward(const agx_conv_desc *d, const float *x, const float *packed,
                     const float *bias, const float *res, float *y, void *stream);

/* ---- backward of a conv layer (training.py:380 loss.backward()) ------------
 * The gradient w.r.t. the layer INPUT is the same polyphase convolution with the channel roles
 * swapped and the taps re-indexed (stride-1 convs: flipped kernel; strided convs: transposed-conv
 * phase form; polyphase up-convs: a strided conv), so it runs on the forward kernels with its own
 * packed image:
 *   dx = [mask-gradient]( [add] + conv_bwd(dy) )
 * d is the FORWARD descriptor of the layer (its epilogue field is ignored).
 *   dy   (B, c_out, l_out)  gradient w.r.t. the layer's linear output (pre-activation)
 *   add  (B, c_in, l_in) or NULL: accumulated first (residual branch of vae.py:117)
 *   mask (B, c_in, l_in) or NULL: the layer input as saved by the forward (= the previous layer's
 *        post-LeakyReLU output); the result is multiplied by its activation gradient (1 or slope)
 *   dx   (B, c_in, l_in) */
int64_t agx_conv_bwd_packed_floats(const agx_conv_desc *d);
int agx_conv_pack_bwd(const agx_conv_desc *d, const float *v, const float *g, float *packed, void *stream);
int agx_conv_bwd_data(const agx_conv_desc *d, const float *dy, const float *packed_bwd, const float *add,
                      const float *mask, float slope, float *dx, void *stream);

/* Gradients w.r.t. the layer's parameters.  x (B,c_in,l_in) is the saved layer input, dy the gradient
 * w.r.t. the linear output; v/g are the weight-norm parameters as in agx_conv_pack (g NULL = plain weight).
 * Writes dv (shape of v), dg (dim0) when g != NULL, dbias (c_out) when dbias != NULL.  The time x batch
 * contraction runs on the fp32 MFMA in slices reduced in a fixed order (deterministic). */
size_t agx_conv_bwd_weight_workspace_bytes(const agx_conv_desc *d);
int agx_conv_bwd_weight(const agx_conv_desc *d, const float *x, const float *dy, const float *v, const float *g,
                        float *dv, float *dg, float *dbias, void *workspace, size_t workspace_bytes,
                        void *stream);

/* Name of the kernel family/tile variant agx_conv_forward would launch for this
 * descriptor (e.g. "conv_mfma<2,2,2,2,16>"), for profilers and bench.py; matches
 * the template arguments in the rocprofv3 kernel names.  Returns AGX_OK. */
int agx_conv_kernel_name(const agx_conv_desc *d, char *buf, size_t buf_len);

/* Fused CausalResidualBlock1d + trailing activation (vae.py:113-117 wrapped by
 * the Sequential(..., activation) of vae.py:130-135 / 193-198):
 *   y = leaky( x + conv_k1( leaky( conv_kK,dil(x) + b1 ) ) + b2 )
 * d describes conv1 (kind CAUSAL, stride 1, c_in == c_out; its epilogue field is
 * ignored); packed1/packed2 are the packed images of conv1 and of the k=1 conv2.
 * post_act = 0 skips the trailing activation.  Shapes without a single-kernel
 * implementation run as two fused-epilogue conv launches through `workspace`
 * (agx_resblock_workspace_bytes(d) bytes; the hidden activation). */
size_t agx_resblock_workspace_bytes(const agx_conv_desc *d);
/* Kernel name agx_resblock_forward would launch ("resblock_mfma<..>" when fused,
 * "2x:<conv kernel>" for the two-launch form). */
int agx_resblock_kernel_name(const agx_conv_desc *d, char *buf, size_t buf_len);
int agx_resblock_forward(const agx_conv_desc *d, const float *x, const float *packed1,
                         const float *bias1, const float *packed2, const float *bias2,
                         float *y, int32_t post_act, void *workspace, size_t workspace_bytes,
                         void *stream);

/* ------------------------------------------------------------------------- *
 * Residual vector quantiser (external `som_quantizer.ResidualQuantizer`;      *
 * call sites vae.py:245-251, 315-318, 333)                                   *
 * ------------------------------------------------------------------------- */

/* Packed codebook image: per stage the transposed codebook (D,K), the squared
 * norms (K) and max norm; floats needed for Q stages. */
int64_t agx_rvq_packed_floats(int32_t n_q, int32_t k, int32_t dim);
int agx_rvq_pack(const float *codebooks /* (Q,K,D) */, int32_t n_q, int32_t k, int32_t dim,
                 float *packed, void *stream);
/* The same for stages with different codebook sizes (the reference takes one size per quantizer, vae.py:233):
 * codebooks is (Q, K, D) with K = the largest stage, sizes[q] <= K (HOST array, NULL = all K) the number of real
 * codewords of stage q; rows beyond it are padding that agx_rvq_forward can never select.  Q <= 64. */
int agx_rvq_pack_sized(const float *codebooks, const int32_t *sizes, int32_t n_q, int32_t k, int32_t dim, float *packed,
                       void *stream);

/* Nearest-codeword search over q_used residual stages.
 *   x, xq   : frames, element (b,t,d) at  b*stride_b + t*stride_t + d*stride_d
 *             (so both "b l c" and "b c l" tensors are accepted without a copy);
 *   index   : (B,T,q_used) int64, contiguous (utils.py:249);
 *   sq_err  : q_used doubles, = sum over all elements of the squared residual
 *             left after each stage (written, not accumulated; commit loss =
 *             sum(sq_err)/(B*T*D)), reduced over the workgroups in a fixed order
 *             by a second tiny kernel -- deterministic, no float atomics;
 *   workspace: agx_rvq_workspace_bytes() bytes (the per-workgroup partial sums).
 * The arg-min is exact: scores on the bf16 matrix pipe (two pieces per operand) only select candidates under an
 * error bound (fp32 terms: rigorous; the bf16 MFMA's accumulation term: 35 x 2^-24 per instruction, what the aligned-truncation model of the
 * instruction proves -- the model is an empirical characterisation, see the knob "rvq_verify"); ties and near ties are decided by the defining
 * binary64 arithmetic (oracle/rvq_exact.c).
 * One workgroup keeps the fp32 residuals and the bf16 pieces of 32 frames in LDS: D <= 560 (AGX_ERR_UNSUPPORTED beyond). */
size_t agx_rvq_workspace_bytes(int32_t batch, int32_t t, int32_t dim, int32_t k, int32_t q_used);
int agx_rvq_forward(const float *x, int64_t x_sb, int64_t x_st, int64_t x_sd,
                    const float *codebooks /* (Q,K,D) */, const float *packed,
                    int32_t batch, int32_t t, int32_t dim, int32_t k, int32_t q_used,
                    float *xq, int64_t q_sb, int64_t q_st, int64_t q_sd,
                    int64_t *index, double *sq_err, void *workspace, size_t workspace_bytes,
                    void *stream);
/* The same; additionally writes the commit loss sum(sq_err) / (B*T*D) as one float to `commit_loss` (may be NULL). */
int agx_rvq_forward_ex(const float *x, int64_t x_sb, int64_t x_st, int64_t x_sd,
                    const float *codebooks /* (Q,K,D) */, const float *packed,
                    int32_t batch, int32_t t, int32_t dim, int32_t k, int32_t q_used,
                    float *xq, int64_t q_sb, int64_t q_st, int64_t q_sd,
                    int64_t *index, double *sq_err, float *commit_loss, void *workspace, size_t workspace_bytes,
                    void *stream);

/* Assignment statistics of the EMA codebook update (`update_codebook=True`, vae.py:315-318; the update rule itself is
 * build-defined): stats (q_used, K, D+1): [q][k][0] = number of frames whose stage-q index is k, [q][k][1+d] = the
 * sum of their stage-q residuals, added in a fixed order (deterministic; no atomics).  frames (N, D) contiguous, index
 * (N, q_used) int64, codebooks (Q >= q_used, K, D) BEFORE the update.  D <= 1024.  workspace: the per-stage residuals,
 * agx_rvq_ema_workspace_bytes() bytes. */
size_t agx_rvq_ema_workspace_bytes(int64_t n_frames, int32_t dim, int32_t q_used);
int agx_rvq_ema_stats(const float *frames, const float *codebooks, const int64_t *index, float *stats, int64_t n_frames,
                      int32_t dim, int32_t k, int32_t q_used, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Activation planes (bf16x3 arithmetic, round 4)                              *
 * ------------------------------------------------------------------------- *
 * A bf16x3 layer multiplies three bf16 pieces x = h + m + l of every fp32 operand.  Splitting the INPUT is per-element
 * vector work that every consumer tile repeats (a polyphase up-conv with M = q Cout rows re-splits the same input tile
 * M / 128 times).  "Activation planes" are that split done ONCE, by the producer: for an activation (B, C, L), C % 8 == 0,
 *     planes[b][C / 8][piece 3][L][8]   bf16   (piece 0 = h, 1 = m, 2 = l;  h + m + l == x exactly)
 * -- 6 bytes per element (1.5 x fp32), a cell of 8 channels x 1 time step = 16 bytes, so a consumer stages a 16-channel
 * chunk of its input tile as six contiguous row pieces by LDS-DMA: no register staging, no vector work.
 *   agx_planes_bytes           size of the planes of a (batch, channels, length) activation (0: bad shape)
 *   agx_planes_split           fp32 (B, C, L) contiguous -> planes (a bandwidth-bound pass; producers with a planes
 *                              output write them from their epilogue instead)
 *   agx_conv_planes_supported  0: this descriptor cannot take planes; 1: it can READ planes (AGX_IMPL_MFMA_BF16X3 layers
 *                              with the ring form: CausalUpsampleConv1d x2 / x4 / x5 / x8, CausalConvT1d k7 s1,
 *                              vae.py:45-89); 2: it can also WRITE its output as planes (one output phase)
 *   agx_conv_forward_planes    agx_conv_forward with x given as planes; y_planes != NULL (code 2 layers): the output is
 *                              written as planes [B][Cout / 8][3][Lout][8] INSTEAD of fp32 (y may then be NULL).
 *                              Results are bit-identical to agx_conv_forward on the fp32 input (same pieces, same order). */
size_t agx_planes_bytes(int32_t batch, int32_t channels, int32_t length);
int agx_planes_split(const float *x, void *planes, int32_t batch, int32_t channels, int32_t length, void *stream);
int agx_conv_planes_supported(const agx_conv_desc *d);
int agx_conv_forward_planes(const agx_conv_desc *d, const void *x_planes, const float *packed, const float *bias, float *y,
                            void *y_planes, void *stream);

/* Diagnostic (not part of the reference surface), PROBE BUILD ONLY (-DAGX_RVQ_PROBE: `python tools/rvq_stamps.py build` writes
 * lib/libagx_rvq_probe.so; the product library returns AGX_ERR_UNSUPPORTED and keeps no pointer between calls): the next
 * agx_rvq_forward launches write s_memtime stamps of workgroup w into
 * device_buffer[w * 16 + slot] (64-bit each; slots 0 / 1 / 15 = kernel start / end of the stage loop / kernel end, 2..13 = the
 * phase boundaries of residual stage `stage`, 14 = the (frame, candidate) pairs that went to the binary64 distance).  NULL = off. */
int agx_rvq_debug_stamps(void *device_buffer, int32_t stage);

/* Verify mode (knob "rvq_verify" = 1; debug).  out3[0] = codes where the fast path's index differs from the full defining search
 * run on the same residual, out3[1] = frames with at least one such code, out3[2] = codes checked -- accumulated over every
 * agx_rvq_forward(_ex) launch since the last reset.  Synchronises the device (a blocking copy).  out3 may be NULL (reset only). */
int agx_rvq_verify_counts(int64_t *out3, int32_t reset);

/* quantizers[i].dequantize(idx) (vae.py:333): out[n,:] (+)= codebook[idx[n],:].
 * out element (n,d) at n*stride_n + d*stride_d. */
int agx_rvq_dequantize(const float *codebook /* (K,D) */, const int64_t *idx, int64_t n,
                       int32_t k, int32_t dim, float *out, int64_t o_sn, int64_t o_sd,
                       int32_t accumulate, void *stream);

/* ------------------------------------------------------------------------- *
 * Attention bottleneck (networks/transformers.py:7-279), channel-major layout *
 * ------------------------------------------------------------------------- *
 * The block runs on (B, C, T) tensors -- the layout the encoder produces -- so
 * every Linear is an agx_conv_forward with kernel 1 (AGX_EPI_GELU_PRE /
 * AGX_EPI_RESIDUAL fuse the FFN activation and both residual adds). */

/* torch.nn.LayerNorm over the channel dim of a (B, C, T) tensor
 * (transformers.py:159 and :214): y = (x - mean) / sqrt(var + eps) * w + b. */
int agx_layernorm_ct(const float *x, const float *weight, const float *bias, float *y,
                     int32_t batch, int32_t channels, int32_t t, float eps, void *stream);

/* softmax(Q K^T / scale_div + M) V per (batch, head) with the ALiBi bias
 * M[h,i,j] = -slopes[h] * |i - j| computed in the kernel (transformers.py:175-188;
 * Alibi :38-39, 62-75).  qkv is (B, 3*H*Dh, T): q rows [0,H*Dh), k rows
 * [H*Dh, 2*H*Dh), v rows [2*H*Dh, 3*H*Dh), head-major inside each.  out is
 * (B, H*Dh, T).  fp32-input MFMA for both contractions (exact fp32), any T: a single-pass kernel for T <= 256, the
 * online-softmax (flash) form over key blocks beyond.  Dh <= 128. */
int agx_attention_alibi(const float *qkv, const float *slopes, float *out, int32_t batch,
                        int32_t heads, int32_t head_dim, int32_t t, float scale_div, void *stream);
/* The same with a choice of arithmetic: AGX_ATTN_FP32 (as above) or AGX_ATTN_BF16 -- operands rounded to bf16, both
 * contractions on the bf16 MFMA with fp32 accumulation, fp32 softmax (BASELINE config 3; flash form at every T).
 * `flash` != 0 forces the flash form for fp32 at T <= 256 too (tests). */
#define AGX_ATTN_FP32 0
#define AGX_ATTN_BF16 1
int agx_attention_alibi_ex(const float *qkv, const float *slopes, float *out, int32_t batch, int32_t heads,
                           int32_t head_dim, int32_t t, float scale_div, int32_t precision, int32_t flash, void *stream);

/* ------------------------------------------------------------------------- *
 * Wavelet / multiresolution layers (networks/wavelets.py)                     *
 * ------------------------------------------------------------------------- */

/* CausalMultiresConv1d.forward (wavelets.py:79-96): depth-level cascade of two
 * depthwise causal filters h0/h1 (C,1,K) with dilation 1,2,4,..., per-channel
 * mixing w (C, depth+2), exact GELU.  x, y (B, C, L).  One launch. */
int agx_multires_forward(const float *x, const float *h0, const float *h1, const float *w, float *y,
                         int32_t batch, int32_t channels, int32_t length, int32_t kernel,
                         int32_t depth, void *stream);

/* Backward of agx_multires_forward: dx (B, C, L), dh0 / dh1 (C, 1, K), dw (C, depth + 2) from dout (B, C, L); the
 * cascade is re-formed per tile in LDS (nothing is kept from the forward).  Parameter sums are per-tile partials in
 * `workspace` reduced in a fixed order (deterministic).  Needs 2 K + depth + 2 <= 64. */
size_t agx_multires_backward_workspace_bytes(int32_t batch, int32_t channels, int32_t length, int32_t kernel,
                                             int32_t depth);
int agx_multires_backward(const float *x, const float *dout, const float *h0, const float *h1, const float *w,
                          float *dx, float *dh0, float *dh1, float *dw, void *workspace, size_t workspace_bytes,
                          int32_t batch, int32_t channels, int32_t length, int32_t kernel, int32_t depth,
                          void *stream);

/* Adjoint of a nearest-neighbour upsample by `group` (MultiresScaleBlock, wavelets.py:112-119): out[i] = sum of
 * g[i * group .. i * group + group), i < n_out; with `gelu_pre` != NULL multiplied by the exact-GELU derivative at
 * gelu_pre[i] (the activation of the k = 1 conv that precedes the upsample). */
int agx_group_sum(const float *g, const float *gelu_pre, float *out, int64_t n_out, int32_t group, void *stream);

/* The fold in the middle of WaveletLayer.forward (wavelets.py:221-231):
 * every input step emits an n_points-long wavelet cos(t)exp(-t^2/sigma_c) * h laid
 * end to end, the output is the sliding-window sum (window n_points, hop
 * n_points/scale) plus the reference's raw-sample tail.  h (B, C, L) ->
 * y (B, C, L*scale).  sigma has `sigma_len` entries (C, or 1 when shared). */
int agx_wavelet_fold(const float *h, const float *space, const float *sigma, int32_t sigma_len,
                     float *y, int32_t batch, int32_t channels, int32_t length, int32_t n_points,
                     int32_t scale, void *stream);

/* Backward of agx_wavelet_fold: dh (B,C,L) and dsigma (sigma_len) from dout (B,C,L*scale).
 * workspace: B*C floats (per-row partials of dsigma, reduced in a fixed order). */
int agx_wavelet_fold_backward(const float *h, const float *dout, const float *space, const float *sigma,
                              int32_t sigma_len, float *dh, float *dsigma, float *workspace, int32_t batch,
                              int32_t channels, int32_t length, int32_t n_points, int32_t scale, void *stream);

/* Backward of agx_layernorm_ct: dx = LN'(x)^T (dy * weight) [+ add], dweight[c] = sum dy * xhat, dbias[c] = sum dy.
 * workspace: 2 * batch * ceil(t / 64) * channels floats. */
int agx_layernorm_ct_backward(const float *x, const float *weight, const float *dy, const float *add, float *dx,
                              float *dweight, float *dbias, float *workspace, int32_t batch, int32_t channels,
                              int32_t t, float eps, void *stream);
/* Backward of agx_attention_alibi: dqkv (B, 3*H*Dh, T) from qkv and dout (B, H*Dh, T).  head_dim <= 64, T <= 256. */
int agx_attention_alibi_backward(const float *qkv, const float *slopes, const float *dout, float *dqkv, int32_t batch,
                                 int32_t heads, int32_t head_dim, int32_t t, float scale_div, void *stream);
/* The same for ANY t and head_dim <= 128 (flash-style split into three deterministic kernels: row statistics, dQ per query block,
 * dK / dV per key block; csrc/attention_flash.hip).  `out` = the forward's output (B, H*Dh, T); workspace:
 * agx_attention_backward_workspace_bytes() bytes. */
size_t agx_attention_backward_workspace_bytes(int32_t batch, int32_t heads, int32_t t);
int agx_attention_alibi_backward_ex(const float *qkv, const float *slopes, const float *out, const float *dout, float *dqkv,
                                    float *workspace, size_t workspace_bytes, int32_t batch, int32_t heads, int32_t head_dim,
                                    int32_t t, float scale_div, void *stream);
/* agx_conv_bwd_data followed by the exact-GELU gradient: dx = (W^T dy [+ add]) * gelu'(pre)  (two launches). */
int agx_conv_bwd_data_gelu(const agx_conv_desc *d, const float *dy, const float *packed_bwd, const float *add,
                           const float *pre, float *dx, void *stream);

/* ------------------------------------------------------------------------- *
 * Discriminators: networks/discriminator.py
 * ------------------------------------------------------------------------- */

/* Spectral norm (torch.nn.utils.spectral_norm, utils.py:34-42 with norm="spectral"): W is the weight
 * viewed as (rows = dim 0, cols = the rest).  power_iterations > 0 (training mode) first updates the
 * buffers in place, v = normalize(W^T u), u = normalize(W v) (eps as F.normalize); then
 * sigma[0] = u . (W v).  sigma stays on the device (consumed by agx_conv_pack_sigma / agx_conv2d_pack).
 * workspace: rows + cols floats. */
int agx_spectral_sigma(const float *w, int32_t rows, int32_t cols, float *u, float *v,
                       int32_t power_iterations, float eps, float *sigma, float *workspace, void *stream);

/* agx_conv_pack for a spectrally normalised layer: every row scaled by 1 / sigma[0] (device scalar). */
int agx_conv_pack_sigma(const agx_conv_desc *d, const float *w, const float *sigma, float *packed,
                        void *stream);

/* Backward of grouped AGX_CONV_PADDED layers (dilation 1), VALU kernels on the torch weight layout
 * w (c_out, c_in / groups, K); sigma (device scalar, may be NULL) divides w.  add / mask / slope as in
 * agx_conv_bwd_data.  bwd_weight returns the PLAIN weight gradient (apply agx_spectral_grad afterwards). */
int agx_conv_grouped_bwd_data(const agx_conv_desc *d, const float *dz, const float *w, const float *sigma,
                              const float *add, const float *mask, float slope, float *dx, void *stream);
size_t agx_conv_grouped_bwd_weight_workspace_bytes(const agx_conv_desc *d);
int agx_conv_grouped_bwd_weight(const agx_conv_desc *d, const float *x, const float *dz, float *dw, float *dbias,
                                void *workspace, size_t workspace_bytes, void *stream);
/* agx_conv_pack_bwd for a spectrally normalised dense layer (rows scaled by 1 / sigma[0]). */
int agx_conv_pack_bwd_sigma(const agx_conv_desc *d, const float *w, const float *sigma, float *packed, void *stream);

/* torch.nn.AvgPool1d(kernel, stride, padding) with count_include_pad=True (discriminator.py:32) over
 * `rows` independent rows of length l_in; returns the output length via agx_avgpool1d_out_len. */
int64_t agx_avgpool1d_out_len(int32_t l_in, int32_t kernel, int32_t stride, int32_t padding);
int agx_avgpool1d(const float *x, float *y, int64_t rows, int32_t l_in, int32_t kernel, int32_t stride,
                  int32_t padding, void *stream);

/* torch.nn.Conv2d(c_in, c_out, (kh, kw), stride, padding) + optional fused LeakyReLU, NCHW fp32
 * (discriminator.py:101-114, 150-167).  Runs on the 1-D MFMA / direct conv kernels with the kernel rows
 * folded into virtual input channels. */
typedef struct agx_conv2d_desc {
    int32_t batch, c_in, c_out, h_in, w_in;
    int32_t kh, kw, stride_h, stride_w, pad_h, pad_w;
    int32_t epilogue; /* AGX_EPI_LEAKY_PRE or 0 */
    float slope;
    int32_t impl;     /* AGX_IMPL_* */
} agx_conv2d_desc;
int agx_conv2d_out_shape(const agx_conv2d_desc *d, int32_t *h_out, int32_t *w_out);
int64_t agx_conv2d_packed_floats(const agx_conv2d_desc *d);
/* w (c_out, c_in, kh, kw); sigma: device scalar of agx_spectral_sigma or NULL (no normalisation). */
int agx_conv2d_pack(const agx_conv2d_desc *d, const float *w, const float *sigma, float *packed, void *stream);
int agx_conv2d_forward(const agx_conv2d_desc *d, const float *x, const float *packed, const float *bias,
                       float *y, void *stream);
int agx_conv2d_kernel_name(const agx_conv2d_desc *d, char *buf, size_t buf_len);
int agx_conv2d_bwd_data_kernel_name(const agx_conv2d_desc *d, char *buf, size_t buf_len);   /* kernel agx_conv2d_bwd_data runs */
/* Backward of the Conv2d layer `d` describes (forward descriptor): gradient w.r.t. its input from the
 * gradient w.r.t. its output, on the same kernels (strided layers as a 2-D polyphase conv over dy);
 * add: optional tensor added to dx (a gradient arriving from another consumer of the input, e.g. the
 * feature-matching loss); mask/slope: optional fused LeakyReLU gradient of the layer that produced the
 * input (mask = its output), applied after the add as in agx_conv_bwd_data. */
int64_t agx_conv2d_bwd_packed_floats(const agx_conv2d_desc *d);
int agx_conv2d_pack_bwd(const agx_conv2d_desc *d, const float *w, const float *sigma, float *packed, void *stream);
int agx_conv2d_bwd_data(const agx_conv2d_desc *d, const float *dy, const float *packed_bwd, const float *add,
                        const float *mask, float slope, float *dx, void *stream);
/* Backward-data of a stride-1 layer with very few input channels (the 2-channel first conv), in two steps:
 * agx_conv2d_colsplit_weights writes the weights of the auxiliary (kh x 1) layer with c_in*kw output rows
 * (wp: c_in*kw x c_out x kh floats; run it with agx_conv2d_forward on dy, padding (kh-1-pad_h, 0)); its output
 * P (B, c_in*kw, h_in, w_out) is folded by agx_conv2d_colsum: dx[c][i][j] = sum_dw P[c*kw+dw][i][j-dw+pad_w] (+ add). */
int agx_conv2d_colsplit_weights(const agx_conv2d_desc *d, const float *w, const float *sigma, float *wp, void *stream);
int agx_conv2d_colsum(const agx_conv2d_desc *d, const float *pbuf, const float *add, float *dx, void *stream);
/* dW (c_out, c_in, kh, kw) and dbias (c_out, may be NULL) of the layer.  With sigma != NULL the layer is
 * spectrally normalised: w is weight_orig, u / v the vectors sigma was computed with, and dw is the
 * gradient w.r.t. weight_orig:  G / sigma - (<G, W> / sigma^2) u v^T  (G = gradient w.r.t. W / sigma). */
size_t agx_conv2d_bwd_weight_workspace_bytes(const agx_conv2d_desc *d);
int agx_conv2d_bwd_weight(const agx_conv2d_desc *d, const float *x, const float *dy, const float *w,
                          const float *sigma, const float *u, const float *v, float *dw, float *dbias,
                          void *workspace, size_t workspace_bytes, void *stream);

/* The torch.stft call of STFTDiscriminator.forward (discriminator.py:181-187): rectangular window,
 * center=True (reflect padding), two-sided, optionally normalised by n_fft^-1/2, hop = n_fft / 4.
 * x (B, L) -> y (B, 2, T, n_fft) with T = 1 + L / hop, channel 0 real / 1 imaginary (the layout after the
 * reference's rearrange "b f t c -> b c t f").  The DFT runs as a conv on the MFMA kernels:
 * agx_stft_pack builds its weight image once per n_fft. */
int64_t agx_stft_frames(int32_t length, int32_t n_fft);
int64_t agx_stft_packed_floats(int32_t n_fft);
int agx_stft_pack(int32_t n_fft, int32_t normalized, float *packed, void *stream);
int64_t agx_stft_workspace_bytes(int32_t batch, int32_t length, int32_t n_fft);
int agx_stft_forward(const float *x, const float *packed, float *y, void *workspace, int32_t batch,
                     int32_t length, int32_t n_fft, void *stream);

/* Adjoint of agx_stft_forward: dx (B, L) from dy (B, 2, T, n_fft).  workspace as agx_stft_workspace_bytes;
 * packed_bwd from agx_stft_pack_bwd (agx_stft_packed_floats floats as well). */
int agx_stft_pack_bwd(int32_t n_fft, int32_t normalized, float *packed_bwd, void *stream);
int agx_stft_backward(const float *dy, const float *packed_bwd, float *dx, void *workspace, int32_t batch,
                      int32_t length, int32_t n_fft, void *stream);
/* AvgPool1d backward (count_include_pad): dx (rows, l_in) from dy (rows, l_out); add may be NULL. */
int agx_avgpool1d_backward(const float *dy, const float *add, float *dx, int64_t rows, int32_t l_in, int32_t kernel,
                           int32_t stride, int32_t padding, void *stream);
/* dz = dy * s * (1 - s) with s = the sigmoid OUTPUT. */
int agx_sigmoid_backward(const float *dy, const float *s, float *dz, int64_t n, void *stream);
/* Spectral-norm chain rule in place on a plain weight gradient G (rows x cols, torch layout):
 * G <- G / sigma - (<G, W> / sigma^2) u v^T.  workspace: rows floats. */
int agx_spectral_grad(float *g, const float *w, const float *sigma, const float *u, const float *v, int32_t rows,
                      int32_t cols, float *workspace, void *stream);

/* Reductions of discriminator_generator_loss (discriminator.py:204-246), one launch each, result in
 * out[0] (device):  mode 0 mean(x) | 1 mean(min(x - 1, 0)) | 2 mean(min(-x - 1, 0)) |
 * 3 mean|x - y| | 4 mean|x + 1e-3| | 5 mean (log(x + 1e-8) - log(y + 1e-8))^2 (training.py:74).
 * y only for modes 3 and 5.  workspace: 1024 floats. */
int agx_reduce_mean(const float *x, const float *y, int64_t n, int32_t mode, float *out, float *workspace,
                    void *stream);
/* Gradient of the above: dx = grad[0] * d mean(term) / dx (and dy = -dx for the L1 term; dy may be NULL). */
int agx_reduce_mean_backward(const float *x, const float *y, int64_t n, int32_t mode, const float *grad, float *dx,
                             float *dy, void *stream);
/* The two means of one feature-matching term (discriminator.py:236-243: mean|x - y| and the scale mean|x + 1e-3| of the
 * real feature map) in ONE pass over the pair: out[0] = mean|x - y|, out[1] = mean|x + 1e-3| -- the same values as modes 3
 * and 4 above give.  workspace: 2048 floats.  backward: dx = grad[0] d out[0] / dx + grad[1] d out[1] / dx,
 * dy = grad[0] d out[0] / dy (either may be NULL) -- bit for bit the sum of the two separate gradients. */
int agx_feature_means(const float *x, const float *y, int64_t n, float *out, float *workspace, void *stream);
int agx_feature_means_backward(const float *x, const float *y, int64_t n, const float *grad, float *dx, float *dy,
                               void *stream);
/* final_activation of the discriminators (torch.nn.Sigmoid, discriminator.py:46, 173). */
int agx_sigmoid(const float *x, float *y, int64_t n, void *stream);

/* ------------------------------------------------------------------------- *
 * Training-loop signal ops: torchaudio in the reference -- parity unpinned (torchaudio is not
 * installed in the build container; restated from its documented behaviour, oracle/signal.py)
 * ------------------------------------------------------------------------- */

/* Framed DFT: windowed STFT with hop | n_fft as a polyphase conv on the MFMA conv kernel.
 *   window_kind 0 rectangular | 1 periodic Hann (win_length <= n_fft, centred as torch.stft does)
 *   norm_kind   0 none | 1 n_fft^-1/2 | 2 (sum window^2)^-1/2  (torchaudio Spectrogram(normalized=True))
 * x (B, L) -> y (B, 2 R, T): R = agx_fdft_rows() real rows then R imaginary rows (bins >= F are zero padding),
 * T = 1 + L / hop, reflect-centred.  backward: the adjoint.  workspace: agx_fdft_workspace_bytes. */
int64_t agx_fdft_frames(int32_t length, int32_t n_fft, int32_t hop);
int64_t agx_fdft_rows(int32_t n_fft, int32_t onesided);
int64_t agx_fdft_packed_floats(int32_t n_fft, int32_t win_length, int32_t hop, int32_t onesided, int32_t backward);
int agx_fdft_pack(int32_t n_fft, int32_t win_length, int32_t hop, int32_t onesided, int32_t window_kind,
                  int32_t norm_kind, int32_t backward, float *packed, void *stream);
int64_t agx_fdft_workspace_bytes(int32_t batch, int32_t length, int32_t n_fft, int32_t hop);
int agx_fdft_forward(const float *x, const float *packed, float *y, void *workspace, int32_t batch, int32_t length,
                     int32_t n_fft, int32_t win_length, int32_t hop, int32_t onesided, void *stream);
int agx_fdft_backward(const float *dy, const float *packed_bwd, float *dx, void *workspace, int32_t batch,
                      int32_t length, int32_t n_fft, int32_t win_length, int32_t hop, int32_t onesided, void *stream);
/* mel[b, m, t] = sum_{f < bins} fb[f, m] (re^2 + im^2) on the fdft layout; fb (bins, n_mels) row-major. */
int agx_melpower(const float *cv, const float *fb, float *mel, int32_t batch, int32_t bins, int32_t frames,
                 int32_t n_mels, void *stream);
int agx_melpower_backward(const float *cv, const float *fb, const float *dmel, float *dcv, int32_t batch, int32_t bins,
                          int32_t frames, int32_t n_mels, void *stream);
/* torchaudio.functional.preemphasis (training.py:333-334): y[n] = x[n] - coeff x[n-1]; adjoint != 0: its transpose. */
int agx_preemphasis(const float *x, float *y, int64_t rows, int32_t length, float coeff, int32_t adjoint,
                    void *stream);
/* torchaudio.functional.lowpass_biquad (training.py:316-318): RBJ low-pass, direct form I, output clamped to [-1, 1]. */
int agx_lowpass_biquad(const float *x, float *y, int64_t rows, int32_t length, float sample_rate, float cutoff_freq,
                       float q, void *stream);
/* torchaudio.transforms.Resample (training.py:554, applied per clip by utils.collator utils.py:157-158): y (rows,
 * agx_resample_out_len(length)) = polyphase sinc interpolation of x (rows, length) with the caller's table
 * (new_freq, 2 width + orig_freq) -- orig_freq / new_freq already divided by their gcd. */
int64_t agx_resample_out_len(int64_t length, int32_t orig_freq, int32_t new_freq);
int agx_resample(const float *x, const float *table, float *y, int64_t rows, int32_t length, int32_t orig_freq,
                 int32_t new_freq, int32_t width, void *stream);

/* ------------------------------------------------------------------------- *
 * Codec bitstream (wire size per utils.py:137-147)               *
 * ------------------------------------------------------------------------- */

/* Dense little-endian packing of n_codes indices at `bits` (1..16) bits each: code e occupies
 * bits [e*bits, (e+1)*bits) of the stream.  packed bytes = ceil(n_codes*bits/8). */
int64_t agx_codes_packed_bytes(int64_t n_codes, int32_t bits);
int agx_codes_pack(const int64_t *codes, int64_t n_codes, int32_t bits, uint8_t *out, void *stream);
int agx_codes_unpack(const uint8_t *in, int64_t n_codes, int32_t bits, int64_t *codes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AGX_H */
