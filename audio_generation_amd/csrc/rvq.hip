// Residual vector quantiser for gfx950: one launch for all residual stages.
//
// Stands in for the external `som_quantizer.ResidualQuantizer` the reference
// calls at networks/vae.py:315-318 (source absent from the reference tree; the
// arithmetic reproduced bit-for-bit is the one fixed in oracle/rvq_exact.c).
//
// One workgroup (8 waves) owns 32 frames for the whole stage loop.  The fp32 residual tile R[frame][d] (frame-major) and
// the two bf16 planes of the centred residual live in LDS, so latents are read once and x_q / indices written once
// (x_q is rebuilt at the end from the chosen codewords, added in stage order).  Per stage:
//   A. scores s[k] = |c'_k|^2 - 2 r'.c'_k for all K codewords on the bf16 matrix pipe (three products of two bf16 pieces
//      per operand; rows = codewords streamed from the stage image in L2 -- the phase is bound by the CU's L2 port: 2 MB per
//      stage and workgroup --, columns = the 32 frames read from the planes), per-frame minimum by in-lane min over the
//      accumulator registers + one lane shuffle + an 8-entry LDS exchange;
//   B. every codeword whose score is within the error margin of the minimum (kernel header of the error unit: what is proved, what is modelled) is a candidate; a frame with one
//      candidate is decided, the rest are decided by the defining binary64 distance (sequential, unfused): squares by all
//      lanes, the d-ordered sums of up to 16 pairs side by side, one lane each;
//   C. r -= c, index written, squared residual accumulated, and the next stage's r' = fl(r - mu) split into the planes.
#include "common.hpp"

namespace agx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 rvq_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int FT = 32;    // frames per workgroup
constexpr int NWV = 8;    // waves per workgroup (2 per SIMD: one computes while the other waits on L2)
constexpr int NT = 64 * NWV;
constexpr int RS = 33;    // LDS row stride of R / O (conflict-free in both access patterns)
constexpr int CAND = 8;   // candidate slots per frame

__host__ __device__ inline int rvq_dp(int dim) { return (dim + 15) & ~15; }  // D rounded up to the 16-deep k block of the bf16 MFMA
// stage image, all on CENTRED codewords c' = fl(c - mu), mu = the stage's mean codeword:
//   CbH[Dp/16][plane 2][lane half 2][K][8] bf16 -- c' = h + m + (error <= 2^-16 |c'_d|), the A operand of
//   v_mfma_f32_32x32x16_bf16 as it stands: lane (code, half) reads its 8 dims 16 kq + 8 half + (0..7) of one plane with one
//   16-byte load; same number of bytes as an fp32 image -- | c2 = |c'|^2 (K) | cn = |c'| (K) | cmax2 + pad(3) | mu (Dp)
// Distances are translation invariant, and scoring |c'|^2 - 2 r'.c' with r' = fl(r - mu) keeps the fp32
// error bound proportional to the SPREAD of the data instead of its offset from the origin: without the
// centring, latents that share a large common component (any encoder with a bias) put most codewords
// inside the error margin of the minimum.
__host__ __device__ inline int64_t rvq_stage_floats(int k, int dim) {
    return int64_t(rvq_dp(dim)) * k + 2 * int64_t(k) + 4 + rvq_dp(dim);
}

// ------------------------------------------------------------------------- pack
// Stages may have fewer than K codewords (the reference takes one codebook size per quantizer, vae.py:233): rows
// kq .. K-1 of such a stage are padding -- excluded from the mean and from cmax2, |c'|^2 = +inf (never a candidate),
// zero in the score image; the stage's count rides in the image's tail for the kernel's full-search fallback.
struct RvqSizes { int n[64]; };

// mu[d] = mean_k c[k][d].  One wave per dimension d: lane L adds codewords L, L + 64, ... in order, the 64 partial sums are
// combined by a fixed shuffle tree (deterministic).  (Round 2: one THREAD per d walking all K rows with a stride of D
// floats -- 384 us per repack at 8 x 1024 x 512.)
__global__ __launch_bounds__(256) void rvq_mean_kernel(const float *__restrict__ cb, int k, int dim,
                                                       float *__restrict__ packed, RvqSizes sizes) {
    const int q = blockIdx.y;
    const int kq = sizes.n[q];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dp = rvq_dp(dim);
    float *mu = packed + q * rvq_stage_floats(k, dim) + size_t(dp) * k + 2 * size_t(k) + 4;
    // 4 waves x 16 dims per block: lane = (codeword phase c0 = lane / 16, dim d0 + lane % 16) so that a wave's loads are
    // 64-byte row segments; each lane walks codewords c0, c0 + 4, ... and the four phases are combined at the end
    const int d = blockIdx.x * 64 + wave * 16 + (lane & 15);
    const int c0 = lane >> 4;
    float acc = 0.f;
    if (d < dim)
        for (int c = c0; c < kq; c += 4) acc += cb[(size_t(q) * k + c) * dim + d];
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (d < dp && c0 == 0) mu[d] = d < dim ? acc / float(kq) : 0.f;
}

__global__ __launch_bounds__(256) void rvq_pack_kernel(const float *__restrict__ cb, int n_q, int k,
                                                       int dim, float *__restrict__ packed, RvqSizes sizes) {
    const int q = blockIdx.y;
    const bool pad = blockIdx.x * 256 + threadIdx.x >= sizes.n[q];
    const int code = blockIdx.x * 256 + threadIdx.x;
    float *img = packed + q * rvq_stage_floats(k, dim);
    if (code >= k) return;
    const float *row = cb + (size_t(q) * k + code) * dim;
    const int dp = rvq_dp(dim);
    const float *mu = img + size_t(dp) * k + 2 * size_t(k) + 4;
    float acc = 0.f;
    __bf16 *imgb = reinterpret_cast<__bf16 *>(img);
    for (int d = 0; d < dp; ++d) {
        const float v = (d < dim && !pad) ? row[d] - mu[d] : 0.f;
        const __bf16 h = (__bf16)v;
        const __bf16 m = (__bf16)(v - (float)h);
        const size_t base = ((size_t(d >> 4) * 4 + ((d >> 3) & 1)) * k + code) * 8 + (d & 7);   // plane 0 (h), lane half (d/8)%2
        imgb[base] = h;
        imgb[base + size_t(2) * k * 8] = m;                                                      // plane 1 (m)
        acc = fmaf(v, v, acc);
    }
    img[size_t(dp) * k + code] = pad ? INFINITY : acc;
    img[size_t(dp) * k + k + code] = pad ? 0.f : sqrtf(acc);
}

__global__ __launch_bounds__(256) void rvq_cmax_kernel(int k, int dim, float *__restrict__ packed, RvqSizes sizes) {
    float *img = packed + blockIdx.x * rvq_stage_floats(k, dim);
    const float *c2 = img + size_t(rvq_dp(dim)) * k;
    const int kq = sizes.n[blockIdx.x];
    float m = 0.f;
    for (int i = threadIdx.x; i < kq; i += 256) m = fmaxf(m, c2[i]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float *tail = img + size_t(rvq_dp(dim)) * k + 2 * size_t(k);
        tail[0] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        tail[1] = __int_as_float(kq);      // codewords of this stage (an integer's bits)
        tail[2] = tail[3] = 0.f;
    }
}

// ------------------------------------------------------------------------ forward
struct RvqArgs {
    const float *x;
    int64_t x_sb, x_st, x_sd;
    const float *cb;      // (Q,K,D)
    const float *packed;  // stage images
    int B, T, D, K, Q;
    float *xq;
    int64_t q_sb, q_st, q_sd;
    int64_t *index;  // (B*T, Q)
    double *sq_err;  // (Q) [unused by the kernel since the partials moved to `part`]
    double *part;    // (workgroups, Q) squared error of each workgroup's 32 frames per stage
    // PROBE BUILD ONLY (-DAGX_RVQ_PROBE, tools/rvq_stamps.py build): the product build fixes acc_scale = 1 and compiles no stamp.
    float acc_scale; // 1; probe knob b3_dbg = 7: 4 (widens the accumulation term of the score error bound), 8: 0 (NOT rigorous:
                     // timing only), 9: -1 (the stage's "squared error" output carries the largest candidate count)
    unsigned long long *stamps;   // agx_rvq_debug_stamps: [workgroup][16] s_memtime at the phase boundaries of stage stamp_q
    int stamp_q;
};
#ifdef AGX_RVQ_PROBE
// thread 0 of every workgroup: kernel-level slots (0, 1, 15) and the phase boundaries of ONE stage (a null pointer costs a scalar branch)
#define RVQ_STAMP(slot, cond)                                                                                      \
    do {                                                                                                           \
        if (a.stamps != nullptr && tid == 0 && (cond)) a.stamps[size_t(blockIdx.x) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define RVQ_ACC_SCALE(a) fabsf((a).acc_scale)
#define RVQ_DIAG(a) ((a).acc_scale < 0.f)
#else
#define RVQ_STAMP(slot, cond) do { } while (0)
#define RVQ_ACC_SCALE(a) 1.f
#define RVQ_DIAG(a) false
#endif

// LDS map (byte offsets).  R is FRAME-major (a frame's residual is one contiguous row: the update, the centring / split and the
// binary64 distances walk it with 16-byte accesses); the score GEMM reads its B operand from the two bf16 PLANES
//   PL[piece 2][Dp / 8][frame][8 bf16]      r' = fl(r - mu) = h + m + O(2^-16 r'),
// written ONCE per stage by the update phase (round 3, first form: every wave re-read R, centred and split its B fragments
// itself -- 16 x redundant, ~50 vector instructions per 6 MFMAs).  The plane region is dead between a stage's score passes and
// its update: the binary64 distances use it for their squares.  x_q is not accumulated in LDS any more: it is rebuilt at the end
// from the chosen codewords, added in stage order (the same fp32 sums).
struct RvqLds {
    int R, PL, wmin, rn2, sqf, mus, cnt, state, best, ccode, cscore, cdist, work, flags, hist, tails, total;
};
constexpr int SQ_PAIRS = 16;   // (frame, candidate) pairs whose squares fit the plane region at a time: 128 Dp / (8 Dp)
constexpr int PROW = (FT + 1) * 16;   // bytes per 8-dim group of a plane: 32 frames x 16 B + 16 B, so that the update's writes (lane = group)
                                      // land 4 banks apart (4-way instead of 64-way conflicts); the GEMM's reads stay contiguous
__host__ __device__ inline int rvq_pad64(int n) { return (n + 63) & ~63; }
__host__ __device__ inline int rvq_rsf(int Dp) { return Dp + 4; }   // floats per row of R (rows stay 16-byte aligned, 4 banks apart)
__host__ __device__ inline RvqLds rvq_lds_map(int Dp, int K, int Q, bool tail) {
    RvqLds m;
    int o = 0;
    m.R = o;      o += FT * rvq_rsf(Dp) * 4;
    m.PL = o;     o += 2 * (Dp / 8) * PROW;      // >= SQ_PAIRS x Dp doubles
    m.cdist = o;  o += FT * CAND * 8;            // (8-byte aligned: everything above is a multiple of 16)
    m.wmin = o;   o += 2 * NWV * FT * 4;      // one buffer per score-pass parity
    m.rn2 = o;    o += FT * 4;
    m.sqf = o;    o += FT * 4;
    m.mus = o;    o += 2 * rvq_pad64(Dp) * 4;   // two buffers: the next stage's mean streams in (LDS-DMA) during the search
    m.cnt = o;    o += FT * 4;
    m.state = o;  o += FT * 4;
    m.best = o;   o += FT * 4;
    m.ccode = o;  o += FT * CAND * 4;
    m.cscore = o; o += FT * CAND * 4;
    m.work = o;   o += FT * CAND * 4;
    m.flags = o;  o += 16;
    m.hist = o;   o += Q * FT * 4;
    m.tails = o;  o += tail ? 2 * rvq_pad64(2 * K) * 4 : 0;   // two buffers, as mus
    m.total = o;
    return m;
}

// sum over the 64 lanes, the same value in every lane: four DPP steps inside the 16-lane rows (quad swaps, half-row and row mirrors
// -- no LDS crossbar), then the four row sums through v_readlane.  (Six ds_bpermute rounds per sum before: 2.9 k cycles for the
// eight sums of an update phase.)
#define RVQ_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float rvq_wave_sum(float v) {
    v += RVQ_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
    v += RVQ_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
    v += RVQ_DPP(v, 0x141);   // row_half_mirror
    v += RVQ_DPP(v, 0x140);   // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ---------------------------------------------------------------- exact distance
// The DEFINING arithmetic (oracle/rvq_exact.c): binary64, d ascending, one subtract + one multiply + one add per term, never fused.
#pragma clang fp contract(off)
// squares of one (frame, codeword) pair in the DEFINING arithmetic (binary64, never fused), all lanes in parallel: lane L forms
// dims 8 L .. 8 L + 7 (+ 512 per round) and leaves them in sq[d]; dims >= D give +0.0.
__device__ __forceinline__ void rvq_pair_squares(double *__restrict__ sq, const float *__restrict__ r_row,
                                                 const float *__restrict__ c, int D, int Dp, int lane) {
    for (int d0 = 8 * lane; d0 < Dp; d0 += 512) {
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(r_row + d0), r1 = *reinterpret_cast<const f32x4 *>(r_row + d0 + 4);
        float cv[8];
        if ((D & 3) == 0) {   // rows are 16-byte aligned
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 c0 = d0 < D ? *reinterpret_cast<const f32x4 *>(c + d0) : z;
            const f32x4 c1 = d0 + 4 < D ? *reinterpret_cast<const f32x4 *>(c + d0 + 4) : z;
#pragma unroll
            for (int e = 0; e < 4; ++e) cv[e] = c0[e], cv[4 + e] = c1[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) cv[e] = d0 + e < D ? c[d0 + e] : 0.f;
        }
        double s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double diff = double(e < 4 ? r0[e] : r1[e - 4]) - double(cv[e]);
            s[e] = diff * diff;
        }
        typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f64x2 v = {s[e], s[e + 1]};
            *reinterpret_cast<f64x2 *>(sq + d0 + e) = v;
        }
    }
}
// ... and their sum in d order, ONE lane per pair: acc = (..((0 + sq_0) + sq_1) + ..) + sq_{D-1}
__device__ __forceinline__ double rvq_chain_sum(const double *__restrict__ sq, int D) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    double acc = 0.0;
    int d = 0;
    for (; d + 16 <= D; d += 16) {
        f64x2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f64x2 *>(sq + d + 2 * j);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc = acc + v[j][0];
            acc = acc + v[j][1];
        }
    }
    for (; d < D; ++d) acc = acc + sq[d];
    return acc;
}
// The full defining search of one frame (candidate overflow), ONE codeword per lane: every lane walks its codeword's row in d
// order -- the defining arithmetic and its summation order per codeword, 64 codewords side by side (the wave-wide evaluation above
// hands the running sum from lane to lane: 21 k cycles per distance, 1.5 ms per overflowing frame at K = 1024).  Returns this
// lane's distance; the caller takes the minimum (lowest index on ties).
__device__ __forceinline__ double rvq_dist_lane(const float *__restrict__ r_row /* LDS, the same for all lanes */,
                                                const float *__restrict__ c /* this lane's codeword */, int D) {
    double acc = 0.0;
    int d = 0;
    if ((D & 3) == 0) {
        for (; d + 16 <= D; d += 16) {
            f32x4 cv[4], rv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cv[j] = *reinterpret_cast<const f32x4 *>(c + d + 4 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[j] = *reinterpret_cast<const f32x4 *>(r_row + d + 4 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double diff = double(rv[j][e]) - double(cv[j][e]);
                    const double sq = diff * diff;
                    acc = acc + sq;
                }
        }
    }
    for (; d < D; ++d) {
        const double diff = double(r_row[d]) - double(c[d]);
        const double sq = diff * diff;
        acc = acc + sq;
    }
    return acc;
}
#pragma clang fp contract(fast)

template <int MT, bool TAIL_LDS>   // TAIL_LDS: the stage's |c'|^2 and |c'| tables are copied to LDS (they fit beside R / the planes)
__global__ __launch_bounds__(NT) void rvq_forward_kernel(RvqArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int Dp = rvq_dp(a.D), RSF = rvq_rsf(Dp), NG = Dp / 8;
    const RvqLds L = rvq_lds_map(Dp, a.K, a.Q, TAIL_LDS);
    float *R = reinterpret_cast<float *>(smem_b + L.R);           // [FT][RSF]
    char *PL = smem_b + L.PL;                                     // [2][NG][PROW]: frame f of group g at g PROW + 16 f
    double *SQ = reinterpret_cast<double *>(smem_b + L.PL);       // [SQ_PAIRS][Dp]  (between the score passes and the update)
    float *wmin = reinterpret_cast<float *>(smem_b + L.wmin);     // [2][NWV][FT]
    float *rn2 = reinterpret_cast<float *>(smem_b + L.rn2);       // [FT]   ||r - mu||^2 of the current stage
    float *sqf = reinterpret_cast<float *>(smem_b + L.sqf);       // [FT]   ||r||^2 after the stage's update
    float *musb = reinterpret_cast<float *>(smem_b + L.mus);      // [2][MP] mean codeword of stage q in buffer q & 1
    int *cnt = reinterpret_cast<int *>(smem_b + L.cnt);           // [FT]
    int *state = reinterpret_cast<int *>(smem_b + L.state);       // [FT] 0 decided / 1 exact among cands / 2 full
    int *best = reinterpret_cast<int *>(smem_b + L.best);         // [FT]
    int *ccode = reinterpret_cast<int *>(smem_b + L.ccode);       // [FT][CAND]
    float *cscore = reinterpret_cast<float *>(smem_b + L.cscore); // [FT][CAND]
    double *cdist = reinterpret_cast<double *>(smem_b + L.cdist); // [FT][CAND]
    int *work = reinterpret_cast<int *>(smem_b + L.work);         // [FT * CAND] (frame, candidate) pairs that need the exact distance
    int *flags = reinterpret_cast<int *>(smem_b + L.flags);       // [0] number of pairs, [1] any frame in candidate overflow
    int *hist = reinterpret_cast<int *>(smem_b + L.hist);         // [Q][FT] the chosen codes (x_q is rebuilt from them)
    float *tailsb = reinterpret_cast<float *>(smem_b + L.tails);  // [2][TP] c2 | cn of stage q in buffer q & 1 (TAIL_LDS)
    const int MP = rvq_pad64(Dp), TP = rvq_pad64(2 * a.K);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t N = int64_t(a.B) * a.T;
    const int64_t n0 = int64_t(blockIdx.x) * FT;
    const int D = a.D, K = a.K;
    constexpr int FPW = FT / NWV;

    RVQ_STAMP(0, true);
    // ---- stage the latents: R[f][d] = x[n0+f][d] (zero for d >= D and for frames past the end); pick the coalesced order ----
    if (a.x_st == 1 || a.x_sd != 1) {  // time-contiguous ("b c l"): frames fastest
        for (int e = tid; e < RSF * FT; e += NT) {
            const int d = e >> 5, f = e & 31;
            const int64_t n = n0 + f;
            float v = 0.f;
            if (d < D && n < N) {
                const int64_t b = n / a.T, t = n - b * a.T;
                v = a.x[b * a.x_sb + t * a.x_st + d * a.x_sd];
            }
            R[f * RSF + d] = v;
        }
    } else {  // channel-contiguous ("b l c"): d fastest
        for (int f = wave; f < FT; f += NWV) {
            const int64_t n = n0 + f;
            const int64_t b = n < N ? n / a.T : 0, t = n < N ? n - b * a.T : 0;
            const float *src = a.x + b * a.x_sb + t * a.x_st;
            for (int d = lane; d < RSF; d += 64) R[f * RSF + d] = (d < D && n < N) ? src[d] : 0.f;
        }
    }
    // the 8 dims d0 .. d0 + 7 of a codeword row: the LOADS only, from clamped addresses (a select on the loaded value would make
    // the wave wait for each row in turn -- four serial L2 round trips per update phase, measured); the elements past D are
    // zeroed by mask_code8 where the row is used
    auto load_code8 = [&](float (&cv)[8], const float *__restrict__ c, int d0) {
        if ((D & 3) == 0) {
            const f32x4 c0 = *reinterpret_cast<const f32x4 *>(c + (d0 < D ? d0 : 0));
            const f32x4 c1 = *reinterpret_cast<const f32x4 *>(c + (d0 + 4 < D ? d0 + 4 : 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) cv[e] = c0[e], cv[4 + e] = c1[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) cv[e] = c[min(d0 + e, D - 1)];
        }
    };
    auto mask_code8 = [&](float (&cv)[8], int d0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cv[e] = d0 + e < D ? cv[e] : 0.f;
    };
    // The pass between two stages over a wave's FPW frames (w, w + NWV, ...), lane L on dims d0 = 8 L .. 8 L + 7 (+ 512 per round):
    //   r -= c (c = the chosen codeword; not before stage 0), squared residual, r' = fl(r - mu) with the NEXT stage's mean,
    //   ||r'||^2, and the two bf16 pieces of r' into the planes.  All frames' loads first, then the arithmetic, then the stores:
    //   the frames' latencies overlap instead of adding up.  red[2 j] += sum r^2, red[2 j + 1] += sum r'^2 of frame j (this lane's part).
    auto frames_pass = [&](float (&cv)[FPW][8], bool sub, const float *mus, int d0, float (&red)[2 * FPW]) {
        f32x4 r0[FPW], r1[FPW];
#pragma unroll
        for (int j = 0; j < FPW; ++j) {
            const float *rr = R + (wave + j * NWV) * RSF + d0;
            r0[j] = *reinterpret_cast<const f32x4 *>(rr), r1[j] = *reinterpret_cast<const f32x4 *>(rr + 4);
        }
        f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0;
        if (mus != nullptr) m0 = *reinterpret_cast<const f32x4 *>(mus + d0), m1 = *reinterpret_cast<const f32x4 *>(mus + d0 + 4);
#pragma unroll
        for (int j = 0; j < FPW; ++j) {
            const int f = wave + j * NWV;
            if (sub) {
                mask_code8(cv[j], d0);
#pragma unroll
                for (int e = 0; e < 4; ++e) r0[j][e] -= cv[j][e], r1[j][e] -= cv[j][4 + e];
                float *rr = R + f * RSF + d0;
                *reinterpret_cast<f32x4 *>(rr) = r0[j];
                *reinterpret_cast<f32x4 *>(rr + 4) = r1[j];
            }
            rvq_bf16x8 h8, m8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float rv = e < 4 ? r0[j][e] : r1[j][e - 4];
                red[2 * j] = fmaf(rv, rv, red[2 * j]);
                const float v = rv - (e < 4 ? m0[e] : m1[e - 4]);
                red[2 * j + 1] = fmaf(v, v, red[2 * j + 1]);
                const __bf16 hh = (__bf16)v;
                h8[e] = hh;
                m8[e] = (__bf16)(v - (float)hh);
            }
            if (mus != nullptr) {   // (after the last stage only the squared residual is wanted)
                const int g = d0 >> 3;
                *reinterpret_cast<rvq_bf16x8 *>(PL + g * PROW + f * 16) = h8;
                *reinterpret_cast<rvq_bf16x8 *>(PL + (NG + g) * PROW + f * 16) = m8;
            }
        }
    };
    auto wave_sums = [&](float (&red)[2 * FPW]) {      // the 2 FPW sums over the lanes (uniform results), chains side by side
#pragma unroll
        for (int i = 0; i < 2 * FPW; ++i) red[i] = rvq_wave_sum(red[i]);
    };
    // |c'|^2 | |c'| (adjacent in the image) and the mean codeword of stage qn -> buffer qn & 1, by LDS-DMA (a dword per lane, 256 B
    // per instruction, issued by all waves in turn): no registers are held across the search and nothing is copied afterwards.
    // Lanes past the end re-read the last element into the buffer's padding.
    auto dma_tables = [&](int qn) {
        const float *c2n = a.packed + qn * rvq_stage_floats(K, D) + size_t(Dp) * K;
        const float *mun = c2n + 2 * size_t(K) + 4;
        if (TAIL_LDS)
            for (int i = wave; i * 64 < 2 * K; i += NWV)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(c2n + min(i * 64 + lane, 2 * K - 1)),
                                                 (__attribute__((address_space(3))) void *)(tailsb + (qn & 1) * TP + i * 64), 4, 0, 0);
        for (int i = wave; i * 64 < Dp; i += NWV)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(mun + min(i * 64 + lane, Dp - 1)),
                                             (__attribute__((address_space(3))) void *)(musb + (qn & 1) * MP + i * 64), 4, 0, 0);
    };
    // Error of one computed score s_k = |c'_k|^2 - 2 dot_k against the exact |r - c_k|^2 - |r'|^2, in units of (|r'| + |c'_k|)^2:
    //   * |c'|^2 table and the two centring roundings: (D + 8) 2^-24 x 1.25 (as for the fp32 score GEMM of rounds 1-2);
    //   * the dot product on two bf16 pieces per operand: each piece pair drops <= 2^-16 of its element and the m.m product
    //     (<= 2^-16) is not formed: <= 3 x 2^-16 |r'||c'| in all;
    //   * its accumulation.  v_mfma_f32_32x32x16_bf16 forms c + sum of 16 exact products.  MODEL of the instruction: the 17 addends
    //     are aligned to the largest one and each is TRUNCATED at the 24th bit of that alignment (<= 2^-23 of the largest addend each),
    //     then the sum is rounded once (<= 2^-24 of the result): |error| <= (17 x 2 + 1) 2^-24 (|c| + sum |a_k b_k|) = 35 x 2^-24 (...)
    //     per instruction -- the constant this model PROVES, and the one used (round 3 used 18, i.e. 2.6 x the worst case measured
    //     but not what the model proves).  The model itself is an empirical characterisation of the hardware
    //     (tools/mfma_bf16_err.hip: equal magnitudes, exponents spread over 2^20, heavy cancellation, large accumulators: correctly
    //     rounded when the addends are of similar size, otherwise off by at most 7.0 x 2^-24 (|c| + sum |a_k b_k|) -- 5 x inside the
    //     constant); tests/test_gpu_rvq_adversarial.py feeds the kernel the operand patterns that would break a tighter model and the
    //     knob rvq_verify re-runs the full defining search beside the fast path.  Worst-case linear growth over the
    //     chain of 3 Dp / 16 instructions with |c| <= the sum of all |products| <= (1 + 2^-7) |r'||c'|;
    //   with |r'||c'| <= (|r'| + |c'|)^2 / 4 and the factor 2 of the score: x 1/2.
    // Far above the typical error (which grows like the square root of the chain length); only the candidate selection
    // depends on it -- a wider margin means more frames decided by the defining binary64 distance, never a different index.
    const float acc_err = 35.f * float(3 * Dp / 16 + 1) * 5.9604645e-8f * (1.f + 0.0078125f);
    const float err_unit = float(D + 8) * 5.9604645e-8f * 1.25f + 0.5f * (RVQ_ACC_SCALE(a) * acc_err + 3.f * 1.5258789e-5f);
    constexpr int CHUNK = NWV * 32 * MT;  // codewords scored per pass
    const int n_chunks = (K + CHUNK - 1) / CHUNK;

    for (int q = 0; q < a.Q; ++q) {
        const float *img = a.packed + q * rvq_stage_floats(K, D);
        const float *c2 = img + size_t(Dp) * K;
        const float *cn = c2 + K;
        const float *cbq = a.cb + size_t(q) * K * D;

        if (q == 0) {   // later stages: all of this is prepared by the previous stage's update phase (C)
            if (tid < FT) cnt[tid] = 0;
            if (tid < 2) flags[tid] = 0;
            dma_tables(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // r' = fl(r - mu), ||r'||^2 and the planes of stage 0 (wave w owns frames w, w + NWV, ...)
            {
                float red[2 * FPW], none[FPW][8];
#pragma unroll
                for (int i = 0; i < 2 * FPW; ++i) red[i] = 0.f;
                for (int d0 = 8 * lane; d0 < Dp; d0 += 512) frames_pass(none, false, musb, d0, red);
                wave_sums(red);
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < FPW; ++j) rn2[wave + j * NWV] = red[2 * j + 1];
                }
            }
            __syncthreads();
        }
        // The next stage's tables stream into the other buffers during this stage's search (everyone left them at the barrier
        // that ended stage q - 1's search).
        const bool more = q + 1 < a.Q;
        if (more) dma_tables(q + 1);
        const float *tails = tailsb + (q & 1) * TP;
        // Error bound of one computed score: |s_k + |r'|^2 - |r - c_k|^2| <= E_k = eu (|r'| + |c'_k|)^2
        // (fp32 MFMA chain and |c'|^2: (D+2) u (c'^2 + 2 r'c'); the two centring roundings: 2 u (r'+c')^2).
        // Per CODEWORD, not per codebook: one far-away outlier codeword must not widen the margin of
        // the near ones.  k is a candidate iff its lower bound s_k - E_k does not exceed
        // U = min_j (s_j + E_j), the smallest upper bound -- the true arg-min always qualifies.
        const float rnorm = sqrtf(rn2[li]);
        const float eu = 1.01f * err_unit;
        float running = INFINITY;
        const bool stq = q == a.stamp_q;
        RVQ_STAMP(2, stq);

        for (int ch = 0; ch < n_chunks; ++ch) {
            const int code0 = ch * CHUNK + wave * (32 * MT);
            f32x16 acc[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            int acol[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) acol[i] = min(code0 + i * 32 + li, K - 1);

            // ---- A: scores on the bf16 matrix pipe.  A[i=code][k=d], B[k=d][j=frame].  Both operands are two bf16 pieces
            // (x = h + m + O(2^-16 x)); a 16-deep block of d is three v_mfma_f32_32x32x16_bf16 per subtile -- m.h, h.m, h.h
            // (small terms first; m.m is below 2^-16 of the product and goes into the error bound) -- 96 cycles against 512
            // for the same block on the fp32-input MFMA.  The scores only SELECT candidates: every decision is still made
            // on the bound below and, among several candidates, by the defining binary64 distance, so the indices
            // do not depend on this arithmetic.  Lane (code, half) holds dims 16 kq + 8 half + (0..7): the codeword pieces
            // are two 16-byte loads of the CbH image straight from L2, requested NB - 1 blocks ahead into a ring of register
            // sets (statically addressed: the loop is unrolled by NB); the residual pieces are two conflict-free
            // ds_read_b128 of the planes, one block ahead.
            const char *pb = PL + lh * PROW + li * 16;
            const char *ab = reinterpret_cast<const char *>(img) + size_t(lh) * K * 16;
            constexpr int NB = 4;
            rvq_bf16x8 a_r[NB][MT][2];
            rvq_bf16x8 bh_cur, bm_cur, bh_nxt, bm_nxt;
            const int nblk = Dp / 16;
            auto load_a = [&](rvq_bf16x8 (&dst)[MT][2], int kq) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        dst[i][pl] = *reinterpret_cast<const rvq_bf16x8 *>(ab + (size_t(kq * 2 + pl) * 2 * K + acol[i]) * 16);
            };
            auto load_b = [&](rvq_bf16x8 &bh, rvq_bf16x8 &bm, int kq) {
                bh = *reinterpret_cast<const rvq_bf16x8 *>(pb + 2 * kq * PROW);
                bm = *reinterpret_cast<const rvq_bf16x8 *>(pb + (NG + 2 * kq) * PROW);
            };
#pragma unroll
            for (int pb0 = 0; pb0 < NB - 1; ++pb0) load_a(a_r[pb0], pb0 < nblk ? pb0 : 0);
            load_b(bh_cur, bm_cur, 0);
            for (int k0 = 0; k0 < nblk; k0 += NB) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int kq = k0 + u;
                    if (kq < nblk) {      // wave-uniform
                        const int kf = kq + NB - 1 < nblk ? kq + NB - 1 : 0;      // past the end the prefetch re-reads block 0
                        const int kn = kq + 1 < nblk ? kq + 1 : 0;
                        load_a(a_r[(u + NB - 1) % NB], kf);
                        load_b(bh_nxt, bm_nxt, kn);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_r[u][i][1], bh_cur, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_r[u][i][0], bm_cur, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_r[u][i][0], bh_cur, acc[i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        bh_cur = bh_nxt;
                        bm_cur = bm_nxt;
                    }
                }
            }
            RVQ_STAMP(3 + 3 * min(ch, 1), stq);
            // lower bounds in place; rows of register r: (r&3) + 8*(r>>2) + 4*lh.  One subtile at a
            // time (the sched_barrier keeps hipcc from hoisting all 2*16*MT table loads at once,
            // which spills at 2 waves/SIMD).
            float m = INFINITY;
            if (TAIL_LDS && (K & 3) == 0) {   // the four codes of a register group are consecutive: one 16-byte read per table
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cb0 = code0 + i * 32 + 8 * g + 4 * lh;
                        const int cc = min(cb0, K - 4);
                        const f32x4 t2 = *reinterpret_cast<const f32x4 *>(tails + cc), tn = *reinterpret_cast<const f32x4 *>(tails + K + cc);
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const int r = 4 * g + e4;
                            const float s = (cb0 < K) ? (t2[e4] - 2.f * acc[i][r]) : INFINITY;
                            const float rc = rnorm + tn[e4];
                            const float e = eu * rc * rc;
                            acc[i][r] = s - e;          // keep the lower bound
                            m = fminf(m, s + e);        // reduce the upper bound
                        }
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int code = code0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int cc = min(code, K - 1);
                    const float s = (code < K) ? ((TAIL_LDS ? tails[cc] : c2[cc]) - 2.f * acc[i][r]) : INFINITY;
                    const float rc = rnorm + (TAIL_LDS ? tails[K + cc] : cn[cc]);
                    const float e = eu * rc * rc;
                    acc[i][r] = s - e;          // keep the lower bound
                    m = fminf(m, s + e);        // reduce the upper bound
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            m = fminf(m, __shfl_xor(m, 32));
            // (one buffer per pass parity: a wave may post pass ch + 1's minima while another still reads pass ch's -- the only
            // barrier of a pass is the one below)
            float *wm = wmin + (ch & 1) * (NWV * FT);
            if (lh == 0) wm[wave * FT + li] = m;
            __syncthreads();
            RVQ_STAMP(4 + 3 * min(ch, 1), stq);
            float cm = wm[li];
#pragma unroll
            for (int w = 1; w < NWV; ++w) cm = fminf(cm, wm[w * FT + li]);
            running = fminf(running, cm);
            const float thr = running;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = acc[i][r];
                    if (s <= thr) {
                        const int code = code0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int slot = atomicAdd(&cnt[li], 1);
                        if (slot < CAND) {
                            ccode[li * CAND + slot] = code;
                            cscore[li * CAND + slot] = s;
                        }
                    }
                }
            RVQ_STAMP(5 + 3 * min(ch, 1), stq);
        }
        __syncthreads();
        // ---- B: decide ----
        if (tid < FT) {  // wave 0, lane == li == frame
            const int f = tid;
            const int n = cnt[f];
            if (n0 + f >= N) {          // padding frame of the last workgroup (its residual is not a latent): nothing to decide
                best[f] = 0;
                state[f] = 0;
                cnt[f] = 0;
            } else if (n > CAND) {
                state[f] = 2;
                flags[1] = 1;
            } else {
                const float thr = running;
                int kept = 0;
                for (int c = 0; c < n; ++c) {
                    const float s = cscore[f * CAND + c];
                    const int code = ccode[f * CAND + c];
                    if (s <= thr) {
                        ccode[f * CAND + kept] = code;
                        ++kept;
                    }
                }
                cnt[f] = kept;
                if (kept == 0) {  // only reachable with NaN scores: stay in bounds
                    best[f] = 0;
                    state[f] = 0;
                } else if (kept == 1) {
                    best[f] = ccode[f * CAND];
                    state[f] = 0;
                } else {
                    state[f] = 1;
                    const int slot = atomicAdd(&flags[0], kept);
                    for (int c = 0; c < kept; ++c) work[slot + c] = f * CAND + c;
                }
            }
        }
        __syncthreads();
        RVQ_STAMP(9, stq);
        if (a.stamps != nullptr && tid == 0 && stq) a.stamps[size_t(blockIdx.x) * 16 + 14] = (unsigned long long)flags[0];
        // The update phase (C) needs the chosen codeword rows: those of the frames decided by the bounds alone are requested NOW,
        // an L2 round trip (~4 k cycles with every other CU streaming codebooks) that then hides behind the binary64 phase.
        // (before that: this wave's pieces of the next stage's tables, DMA'd at the start of the stage, have long landed -- the wait
        // is free here, and the pick barrier below then publishes them without waiting for the loads issued now)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float cv[FPW][8];
        bool early[FPW];
#pragma unroll
        for (int j = 0; j < FPW; ++j) {
            const int f = wave + j * NWV;
            early[j] = state[f] == 0;                  // (wave-uniform)
            if (early[j]) load_code8(cv[j], cbq + size_t(best[f]) * D, 8 * lane);
        }
        // The listed (frame, candidate) pairs, SQ_PAIRS at a time: every wave forms the squares of its pairs with all lanes
        // (coalesced codeword loads) into the plane region, then ONE lane per pair adds them in d order -- the pairs' 512-long
        // dependent chains run side by side (lane p / 8 of wave p % 8) instead of one after the other.
        {
            const int n_pairs = flags[0];
            for (int base = 0; base < n_pairs; base += SQ_PAIRS) {   // (uniform)
                const int np = min(SQ_PAIRS, n_pairs - base);
                for (int pp = wave; pp < np; pp += NWV) {
                    const int fc = work[base + pp], f = fc >> 3;
                    rvq_pair_squares(SQ + size_t(pp) * Dp, R + f * RSF, cbq + size_t(ccode[fc]) * D, D, Dp, lane);
                }
                __syncthreads();
                const int pp = lane * NWV + wave;
                if (lane < SQ_PAIRS / NWV && pp < np) cdist[work[base + pp]] = rvq_chain_sum(SQ + size_t(pp) * Dp, D);
                __syncthreads();
            }
        }
        RVQ_STAMP(10, stq);
        if (tid < FT && state[tid] == 1) {
            const int f = tid;
            double bd = cdist[f * CAND];
            int bc = ccode[f * CAND];
            for (int c = 1; c < cnt[f]; ++c) {
                const double dc = cdist[f * CAND + c];
                const int code = ccode[f * CAND + c];
                if (dc < bd || (dc == bd && code < bc)) {
                    bd = dc;
                    bc = code;
                }
            }
            best[f] = bc;
        }
        __syncthreads();
        RVQ_STAMP(11, stq);
        int diag_mx = 0;
        if (RVQ_DIAG(a) && tid == 0)     // DIAGNOSTIC, probe build only (knob b3_dbg = 9): the stage's "squared error" output = largest candidate count
            for (int f = 0; f < FT; ++f) diag_mx = max(diag_mx, state[f] == 2 ? 1000 + cnt[f] : cnt[f]);
        // candidate overflow (degenerate codebooks): full defining search, whole block per frame
        const int any_overflow = flags[1];
        const int kq_bits = __float_as_int(c2[2 * size_t(K) + 1]);
        const int Kq = kq_bits > 0 && kq_bits <= K ? kq_bits : K;     // this stage's codewords (rest of the K rows: padding)
        for (int f = 0; f < (any_overflow ? FT : 0); ++f) {
            if (state[f] != 2) continue;  // uniform across the block (LDS value)
            double bd = INFINITY;
            int bc = 0x7fffffff;
            for (int base = wave * 64; base < Kq; base += NWV * 64) {      // a codeword per lane (rvq_dist_lane)
                const int code = base + lane;
                const double dc = rvq_dist_lane(R + f * RSF, cbq + size_t(min(code, Kq - 1)) * D, D);
                if (code < Kq && (dc < bd || (dc == bd && code < bc))) {
                    bd = dc;
                    bc = code;
                }
            }
            for (int off = 32; off > 0; off >>= 1) {                        // the wave's minimum, lowest index on ties
                const double od = __shfl_xor(bd, off);
                const int oc = __shfl_xor(bc, off);
                if (od < bd || (od == bd && oc < bc)) {
                    bd = od;
                    bc = oc;
                }
            }
            __syncthreads();  // cdist / ccode row 0 are free to reuse as exchange
            if (lane == 0) {
                cdist[wave] = bd;
                ccode[wave] = bc;
            }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NWV; ++w)
                    if (cdist[w] < bd || (cdist[w] == bd && ccode[w] < bc)) {
                        bd = cdist[w];
                        bc = ccode[w];
                    }
                best[f] = bc;
            }
            __syncthreads();
        }
        RVQ_STAMP(12, stq);
        // ---- C: r -= c, index, squared residual; the next stage's r' = fl(r - mu), its norm and its planes ----
        {
            float red[2 * FPW];
#pragma unroll
            for (int i = 0; i < 2 * FPW; ++i) red[i] = 0.f;
            for (int db = 0; db < Dp; db += 512) {
                const int d0 = db + 8 * lane;
#pragma unroll
                for (int j = 0; j < FPW; ++j)
                    if (db > 0 || !early[j]) load_code8(cv[j], cbq + size_t(best[wave + j * NWV]) * D, d0);
                if (d0 < Dp) frames_pass(cv, true, more ? musb + ((q + 1) & 1) * MP : nullptr, d0, red);
            }
            wave_sums(red);
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < FPW; ++j) {
                    const int f = wave + j * NWV;
                    sqf[f] = red[2 * j];       // the squared residual of this frame
                    rn2[f] = red[2 * j + 1];
                    cnt[f] = 0;                // (read last by the pick step, before the barrier above)
                    hist[q * FT + f] = best[f];
                    if (n0 + f < N) a.index[(n0 + f) * a.Q + q] = best[f];
                }
            }
            if (tid == 0) flags[0] = 0;        // (read last at the start of the binary64 phase)
        }
        __syncthreads();                       // the ONE barrier between the update and the next stage's search
        if (tid == 0) {
            flags[1] = 0;                      // (everyone read it right after the pick barrier)
            if (RVQ_DIAG(a)) {
                sqf[0] = float(diag_mx);
                for (int f = 1; f < FT; ++f) sqf[f] = 0.f;
            }
        }
        if (wave == 0) {                  // the stage's squared error: 32 frames, pairwise in double
            double s = (lane < FT && n0 + lane < N) ? double(sqf[lane]) : 0.0;
            for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) a.part[size_t(blockIdx.x) * a.Q + q] = s;   // reduced in a fixed order by rvq_sqerr_kernel
        }
        RVQ_STAMP(13, stq);
    }
    RVQ_STAMP(1, true);
    // ---- x_q = ((0 + c_0) + c_1) + ... rebuilt from the chosen codewords, stage order ----
    const bool q_rows = !(a.q_st == 1 || a.q_sd != 1);     // channel-contiguous output: a frame's row goes out as it is formed
    float *Ot = reinterpret_cast<float *>(smem_b);         // [Dp][RS] transposition tile over R / the planes (both dead now)
    __syncthreads();
    for (int db = 0; db < Dp; db += 512) {
        const int d0 = db + 8 * lane;
#pragma unroll
        for (int j = 0; j < FPW; ++j) {
            const int f = wave + j * NWV;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;
            for (int q0 = 0; q0 < a.Q; q0 += 4) {          // four stages' rows in flight
                float cv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = min(q0 + u, a.Q - 1);
                    load_code8(cv[u], a.cb + (size_t(q) * K + hist[q * FT + f]) * D, d0 < Dp ? d0 : 0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (q0 + u < a.Q) {
                        mask_code8(cv[u], d0);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = o[e] + cv[u][e];
                    }
            }
            if (d0 < Dp) {
                if (q_rows) {
                    const int64_t n = n0 + f;
                    if (n < N) {
                        const int64_t b = n / a.T, t = n - b * a.T;
                        float *dst = a.xq + b * a.q_sb + t * a.q_st;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (d0 + e < D) dst[d0 + e] = o[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) Ot[(d0 + e) * RS + f] = o[e];
                }
            }
        }
    }
    if (!q_rows) {
        __syncthreads();
        for (int e = tid; e < D * FT; e += NT) {
            const int d = e >> 5, f = e & 31;
            const int64_t n = n0 + f;
            if (n < N) {
                const int64_t b = n / a.T, t = n - b * a.T;
                a.xq[b * a.q_sb + t * a.q_st + d * a.q_sd] = Ot[d * RS + f];
            }
        }
    }
    RVQ_STAMP(15, true);
}

// ------------------------------------------------------------------------ verify mode (debug knob rvq_verify)
// The DEFINING search of every (frame, stage) next to the fast path's answer: one wave per frame, a codeword per lane
// (rvq_dist_lane: binary64, d ascending, never fused -- oracle/rvq_exact.c), the minimum with the lowest index on ties, compared
// with index[frame][stage]; the residual then follows the FAST path's choice (r <- r - c[index], binary32), so every stage is
// checked on exactly the residual the fast path searched.  Counters (g_rvq_verify): [0] codes that differ, [1] frames with at
// least one, [2] codes checked.  ~30 G binary64 terms on the bench workload: tens of milliseconds, a debug tool.
__device__ unsigned long long g_rvq_verify[4];
__global__ __launch_bounds__(256) void rvq_verify_kernel(const float *__restrict__ x, int64_t x_sb, int64_t x_st, int64_t x_sd,
                                                         const float *__restrict__ cb, const float *__restrict__ packed, int64_t N,
                                                         int T, int D, int K, int Q, const int64_t *__restrict__ index) {
    extern __shared__ __align__(16) float vr[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t f = int64_t(blockIdx.x) * 4 + wave;
    if (f >= N) return;       // (no workgroup barrier below: a wave only touches its own row)
    float *r = vr + size_t(wave) * (D + 4);
    const float *xf = x + (f / T) * x_sb + (f % T) * x_st;
    for (int d = lane; d < D; d += 64) r[d] = xf[d * x_sd];
    const int Dp = rvq_dp(D);
    int bad = 0;
    for (int q = 0; q < Q; ++q) {
        const float *cbq = cb + size_t(q) * K * D;
        const int kq_bits = __float_as_int(packed[q * rvq_stage_floats(K, D) + size_t(Dp) * K + 2 * size_t(K) + 1]);
        const int Kq = kq_bits > 0 && kq_bits <= K ? kq_bits : K;
        double best = 1.0 / 0.0;
        int arg = 0x7fffffff;
        for (int c0 = 0; c0 < Kq; c0 += 64) {
            const int c = c0 + lane;
            const double dc = rvq_dist_lane(r, cbq + size_t(min(c, Kq - 1)) * D, D);
            if (c < Kq && dc < best) best = dc, arg = c;       // ascending c per lane: a tie keeps the lower index
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int oa = __shfl_xor(arg, off);
            if (ob < best || (ob == best && oa < arg)) best = ob, arg = oa;
        }
        const int64_t got = index[f * Q + q];
        if (got != int64_t(arg)) ++bad;
        const float *cs = cbq + size_t(got >= 0 && got < Kq ? got : 0) * D;
        for (int d = lane; d < D; d += 64) r[d] = r[d] - cs[d];
    }
    if (lane == 0) {
        if (bad) {
            atomicAdd(&g_rvq_verify[0], (unsigned long long)bad);
            atomicAdd(&g_rvq_verify[1], 1ull);
        }
        atomicAdd(&g_rvq_verify[2], (unsigned long long)Q);
    }
}

static size_t rvq_lds_bytes(int dim, int k, int q, bool *tail_in_lds) {
    const int Dp = rvq_dp(dim);
    const size_t with = size_t(rvq_lds_map(Dp, k, q, true).total);
    const bool fits = with <= 160 * 1024;
    if (tail_in_lds) *tail_in_lds = fits;
    return fits ? with : size_t(rvq_lds_map(Dp, k, q, false).total);
}

// ------------------------------------------------------------------------ dequantize
__global__ __launch_bounds__(256) void rvq_dequant_kernel(const float *__restrict__ cb,
                                                          const int64_t *__restrict__ idx, int64_t n,
                                                          int k, int dim, float *__restrict__ out,
                                                          int64_t o_sn, int64_t o_sd, int accumulate) {
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    int64_t i = idx[row];
    i = i < 0 ? 0 : (i >= k ? k - 1 : i);
    const float *c = cb + i * dim;
    float *o = out + row * o_sn;
    for (int d = threadIdx.x & 63; d < dim; d += 64) {
        const float v = c[d];
        o[d * o_sd] = accumulate ? o[d * o_sd] + v : v;
    }
}

// sq_err[q] = sum over workgroups of part[wg][q] (pairwise per lane, then a fixed shuffle tree); commit = sum_q sq_err[q] * inv
__global__ __launch_bounds__(64) void rvq_sqerr_kernel(const double *__restrict__ part, int nwg, int Q, double *__restrict__ sq_err,
                                                       float *__restrict__ commit, double inv_numel) {
    const int lane = threadIdx.x;
    double total = 0.0;
    for (int q = 0; q < Q; ++q) {
        double s = 0.0;
        for (int w = lane; w < nwg; w += 64) s += part[size_t(w) * Q + q];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) sq_err[q] = s;
        total += s;
    }
    if (lane == 0 && commit) *commit = float(total * inv_numel);
}

}  // namespace agx

namespace agx {

// Assignment statistics of the EMA codebook update (quantizer.py:_ema_update): for stage q and code k
//     stats[q][k][0] = #{n : index[n][q] = k},   stats[q][k][1 + d] = sum over those n of r_q[n][d]
// with r_q[n] = frames[n] - c_0[index[n][0]] - ... - c_{q-1}[index[n][q-1]] (fp32, subtracted in stage order: the residual
// the search saw).  Two launches, no atomics, so the update is reproducible run to run (torch's index_add_ on the
// device is not):
//   rvq_residuals_kernel   R[q][n][:] for every stage (workspace, Q N D floats)
//   rvq_ema_stats_kernel   one workgroup per (code, stage): ordered compaction of the matching frames (ballot + prefix),
//                          wave w adds rows w, w+4, ... of the list in list order, the four partial sums are added in
//                          wave order.
constexpr int EMA_LIST = 1024;

__global__ __launch_bounds__(256) void rvq_residuals_kernel(const float *__restrict__ frames, const float *__restrict__ cb,
                                                            const int64_t *__restrict__ index, float *__restrict__ R,
                                                            int n, int dim, int K, int Q) {
    const int64_t total = int64_t(n) * dim;
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < total; e += int64_t(gridDim.x) * 256) {
        const int f = int(e / dim), d = int(e - int64_t(f) * dim);
        float r = frames[e];
        for (int q = 0; q < Q; ++q) {
            R[int64_t(q) * total + e] = r;
            const int64_t c = index[int64_t(f) * Q + q];
            if (c >= 0 && c < K) r -= cb[(int64_t(q) * K + c) * dim + d];
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void rvq_ema_stats_kernel(const float *__restrict__ R, const int64_t *__restrict__ index,
                                                            float *__restrict__ stats, int n, int dim, int K, int Q) {
    constexpr int NA = VEC ? 4 : 16;          // accumulators per lane: dim <= 1024
    __shared__ int list[EMA_LIST];
    __shared__ int wave_cnt[4];
    __shared__ float part[4][1024];
    const int k = blockIdx.x, q = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *Rq = R + int64_t(q) * n * dim;
    f32x4 av[VEC ? NA : 1];
    float as[VEC ? 1 : NA];
#pragma unroll
    for (int u = 0; u < (VEC ? NA : 1); ++u) av[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < (VEC ? 1 : NA); ++u) as[u] = 0.f;
    int held = 0, total = 0;
    auto drain = [&]() {
#pragma unroll 4
        for (int e = wave; e < held; e += 4) {
            const float *row = Rq + int64_t(list[e]) * dim;
            if (VEC) {
#pragma unroll
                for (int u = 0; u < NA; ++u) {
                    const int c = 4 * (lane + 64 * u);
                    if (c < dim) av[u] += *reinterpret_cast<const f32x4 *>(row + c);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NA; ++u) {
                    const int d = lane + 64 * u;
                    if (d < dim) as[u] += row[d];
                }
            }
        }
    };
    for (int base = 0; base < n; base += 256) {
        const int f = base + tid;
        const bool hit = f < n && index[size_t(f) * Q + q] == k;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = held;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = f;
        const int add = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        held += add;
        total += add;
        __syncthreads();
        if (held > EMA_LIST - 256) {   // uniform: the next block of 256 might not fit
            drain();
            held = 0;
            __syncthreads();
        }
    }
    drain();
    if (VEC) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int c = 4 * (lane + 64 * u);
            if (c < dim)
#pragma unroll
                for (int i = 0; i < 4; ++i) part[wave][c + i] = av[u][i];
        }
    } else {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int d = lane + 64 * u;
            if (d < dim) part[wave][d] = as[u];
        }
    }
    __syncthreads();
    float *o = stats + (size_t(q) * K + k) * (dim + 1);
    if (tid == 0) o[0] = float(total);
    for (int d = tid; d < dim; d += 256) o[1 + d] = ((part[0][d] + part[1][d]) + part[2][d]) + part[3][d];
}

}  // namespace agx

extern "C" {

int64_t agx_rvq_packed_floats(int32_t n_q, int32_t k, int32_t dim) {
    if (n_q <= 0 || k <= 0 || dim <= 0) return AGX_ERR_BAD_SHAPE;
    return int64_t(n_q) * agx::rvq_stage_floats(k, dim);
}

int agx_rvq_pack_sized(const float *codebooks, const int32_t *sizes, int32_t n_q, int32_t k, int32_t dim, float *packed,
                       void *stream) {
    using namespace agx;
    if (n_q <= 0 || k <= 0 || dim <= 0) return fail(AGX_ERR_BAD_SHAPE, "rvq_pack: bad shape Q=%d K=%d D=%d", n_q, k, dim);
    if (!codebooks || !packed) return fail(AGX_ERR_NULL_POINTER, "rvq_pack: NULL pointer");
    if (n_q > 64) return fail(AGX_ERR_UNSUPPORTED, "rvq_pack: at most 64 stages (Q=%d)", n_q);
    RvqSizes sz;
    for (int q = 0; q < 64; ++q) sz.n[q] = k;
    for (int q = 0; q < n_q && sizes; ++q) {
        if (sizes[q] < 1 || sizes[q] > k) return fail(AGX_ERR_BAD_SHAPE, "rvq_pack: stage %d has %d codewords (K=%d)", q, sizes[q], k);
        sz.n[q] = sizes[q];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(rvq_mean_kernel, dim3(ceil_div(rvq_dp(dim), 64), n_q), dim3(256), 0, st, codebooks, k, dim, packed, sz);
    hipLaunchKernelGGL(rvq_pack_kernel, dim3(ceil_div(k, 256), n_q), dim3(256), 0, st, codebooks, n_q, k, dim, packed, sz);
    hipLaunchKernelGGL(rvq_cmax_kernel, dim3(n_q), dim3(256), 0, st, k, dim, packed, sz);
    return check_launch("agx_rvq_pack");
}

int agx_rvq_pack(const float *codebooks, int32_t n_q, int32_t k, int32_t dim, float *packed, void *stream) {
    return agx_rvq_pack_sized(codebooks, nullptr, n_q, k, dim, packed, stream);
}

size_t agx_rvq_workspace_bytes(int32_t batch, int32_t t, int32_t, int32_t, int32_t q_used) {
    if (batch <= 0 || t <= 0 || q_used <= 0) return 0;
    return size_t(agx::ceil_div64(int64_t(batch) * t, agx::FT)) * q_used * sizeof(double);
}

// diagnostic: device buffer of (workgroups x 16) 64-bit stamps the next rvq_forward launches fill (NULL switches it off).
// PROBE BUILD ONLY (-DAGX_RVQ_PROBE; `tools/rvq_stamps.py build` writes lib/libagx_rvq_probe.so): the product library keeps no
// raw pointer between calls and refuses.
#ifdef AGX_RVQ_PROBE
static unsigned long long *g_rvq_stamps = nullptr;
static int g_rvq_stamp_q = 0;
int agx_rvq_debug_stamps(void *device_buffer, int32_t stage) {
    g_rvq_stamps = static_cast<unsigned long long *>(device_buffer);
    g_rvq_stamp_q = stage;
    return AGX_OK;
}
#else
int agx_rvq_debug_stamps(void *, int32_t) {
    return agx::fail(AGX_ERR_UNSUPPORTED, "agx_rvq_debug_stamps: probe build only (tools/rvq_stamps.py build)");
}
#endif

// Verify mode (knob rvq_verify = 1): the counters of the checker kernel, in the code object (nothing is allocated).
int agx_rvq_verify_counts(int64_t *out3, int32_t reset) {
    unsigned long long h[4] = {0, 0, 0, 0};
    if (out3) {
        hipError_t e = hipMemcpyFromSymbol(h, HIP_SYMBOL(agx::g_rvq_verify), sizeof(h), 0, hipMemcpyDeviceToHost);   // synchronises
        if (e != hipSuccess) return agx::fail(AGX_ERR_LAUNCH, "agx_rvq_verify_counts: %s", hipGetErrorString(e));
        for (int i = 0; i < 3; ++i) out3[i] = int64_t(h[i]);
    }
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(agx::g_rvq_verify), z, sizeof(z), 0, hipMemcpyHostToDevice);
        if (e != hipSuccess) return agx::fail(AGX_ERR_LAUNCH, "agx_rvq_verify_counts: %s", hipGetErrorString(e));
    }
    return AGX_OK;
}

int agx_rvq_forward(const float *x, int64_t x_sb, int64_t x_st, int64_t x_sd, const float *codebooks,
                    const float *packed, int32_t batch, int32_t t, int32_t dim, int32_t k, int32_t q_used,
                    float *xq, int64_t q_sb, int64_t q_st, int64_t q_sd, int64_t *index, double *sq_err,
                    void *workspace, size_t workspace_bytes, void *stream) {
    return agx_rvq_forward_ex(x, x_sb, x_st, x_sd, codebooks, packed, batch, t, dim, k, q_used, xq, q_sb, q_st, q_sd, index,
                              sq_err, nullptr, workspace, workspace_bytes, stream);
}

int agx_rvq_forward_ex(const float *x, int64_t x_sb, int64_t x_st, int64_t x_sd, const float *codebooks,
                       const float *packed, int32_t batch, int32_t t, int32_t dim, int32_t k, int32_t q_used,
                       float *xq, int64_t q_sb, int64_t q_st, int64_t q_sd, int64_t *index, double *sq_err,
                       float *commit_loss, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace agx;
    if (q_used > 0 && (!workspace || workspace_bytes < agx_rvq_workspace_bytes(batch, t, dim, k, q_used)))
        return fail(AGX_ERR_WORKSPACE, "rvq_forward: workspace too small (%zu < %zu)", workspace_bytes,
                    agx_rvq_workspace_bytes(batch, t, dim, k, q_used));
    if (batch <= 0 || t <= 0 || dim <= 0 || k <= 0 || q_used < 0)
        return fail(AGX_ERR_BAD_SHAPE, "rvq_forward: bad shape B=%d T=%d D=%d K=%d Q=%d", batch, t, dim, k, q_used);
    if (!x || !codebooks || !packed || !xq || (q_used > 0 && (!index || !sq_err)))
        return fail(AGX_ERR_NULL_POINTER, "rvq_forward: NULL pointer");
    bool tail_lds = false;
    const size_t lds = rvq_lds_bytes(dim, k, q_used, &tail_lds);
    if (lds > 160 * 1024) return fail(AGX_ERR_UNSUPPORTED, "rvq_forward: D=%d needs %zu B of LDS", dim, lds);
    RvqArgs a{x, x_sb, x_st, x_sd, codebooks, packed, batch, t, dim, k, q_used, xq, q_sb, q_st, q_sd, index, sq_err,
              static_cast<double *>(workspace),
#ifdef AGX_RVQ_PROBE
              tuning().b3_dbg == 7 ? 4.f : (tuning().b3_dbg == 8 ? 0.f : (tuning().b3_dbg == 9 ? -1.f : 1.f)), g_rvq_stamps, g_rvq_stamp_q};
#else
              1.f, nullptr, 0};
#endif
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n = int64_t(batch) * t;
    dim3 grid((unsigned)ceil_div64(n, FT)), block(NT);
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
        if (q_used > 0)   // per-stage sums and the commit loss, in a fixed order: deterministic, no atomics, no host arithmetic
            hipLaunchKernelGGL(rvq_sqerr_kernel, dim3(1), dim3(64), 0, st, a.part, int(grid.x), q_used, sq_err, commit_loss,
                               1.0 / (double(batch) * t * dim));
        if (tuning().rvq_verify && q_used > 0) {   // DEBUG: the full defining search of every (frame, stage) beside the fast path
            const size_t vlds = size_t(4) * (size_t(dim) + 4) * sizeof(float);
            if (vlds > 64 * 1024) return fail(AGX_ERR_UNSUPPORTED, "rvq_verify: D=%d too large for the checker", dim);
            hipLaunchKernelGGL(rvq_verify_kernel, dim3((unsigned)ceil_div64(n, 4)), dim3(256), vlds, st, x, x_sb, x_st, x_sd,
                               codebooks, packed, n, t, dim, k, q_used, index);
        }
        return check_launch("rvq_forward");
    };
    // codewords per pass = 8 waves x MT x 32.  MT = 4 (one pass for K = 1024) spills at the 256-VGPR cap
    // of 2 waves/SIMD (measured again with the operand ring: 903 us against 807), so K > 512 runs as passes of
    // 512 codewords with the running-bound candidate rule.
    if (k > 256) return tail_lds ? launch(rvq_forward_kernel<2, true>) : launch(rvq_forward_kernel<2, false>);
    return tail_lds ? launch(rvq_forward_kernel<1, true>) : launch(rvq_forward_kernel<1, false>);
}

int agx_rvq_dequantize(const float *codebook, const int64_t *idx, int64_t n, int32_t k, int32_t dim,
                       float *out, int64_t o_sn, int64_t o_sd, int32_t accumulate, void *stream) {
    using namespace agx;
    if (n <= 0 || k <= 0 || dim <= 0) return fail(AGX_ERR_BAD_SHAPE, "rvq_dequantize: bad shape");
    if (!codebook || !idx || !out) return fail(AGX_ERR_NULL_POINTER, "rvq_dequantize: NULL pointer");
    hipLaunchKernelGGL(rvq_dequant_kernel, dim3((unsigned)ceil_div64(n, 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), codebook, idx, n, k, dim, out, o_sn, o_sd, accumulate);
    return check_launch("rvq_dequantize");
}

size_t agx_rvq_ema_workspace_bytes(int64_t n_frames, int32_t dim, int32_t q_used) {
    if (n_frames <= 0 || dim <= 0 || q_used <= 0) return 0;
    return size_t(n_frames) * dim * q_used * sizeof(float);
}

int agx_rvq_ema_stats(const float *frames, const float *codebooks, const int64_t *index, float *stats, int64_t n_frames,
                      int32_t dim, int32_t k, int32_t q_used, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace agx;
    if (n_frames <= 0 || n_frames > INT32_MAX || dim <= 0 || k <= 0 || q_used <= 0 || q_used > 65535)
        return fail(AGX_ERR_BAD_SHAPE, "rvq_ema_stats: bad shape N=%lld D=%d K=%d Q=%d", (long long)n_frames, dim, k, q_used);
    if (dim > 1024) return fail(AGX_ERR_UNSUPPORTED, "rvq_ema_stats: D=%d > 1024", dim);
    if (!frames || !codebooks || !index || !stats || !workspace) return fail(AGX_ERR_NULL_POINTER, "rvq_ema_stats: NULL pointer");
    if (workspace_bytes < agx_rvq_ema_workspace_bytes(n_frames, dim, q_used))
        return fail(AGX_ERR_WORKSPACE, "rvq_ema_stats: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *R = static_cast<float *>(workspace);
    const int64_t total = n_frames * dim;
    hipLaunchKernelGGL(rvq_residuals_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(total, 256), 16384)), dim3(256), 0, st,
                       frames, codebooks, index, R, int(n_frames), dim, k, q_used);
    if (dim % 4 == 0)
        hipLaunchKernelGGL(rvq_ema_stats_kernel<true>, dim3(k, q_used), dim3(256), 0, st, R, index, stats, int(n_frames), dim, k,
                           q_used);
    else
        hipLaunchKernelGGL(rvq_ema_stats_kernel<false>, dim3(k, q_used), dim3(256), 0, st, R, index, stats, int(n_frames), dim,
                           k, q_used);
    return check_launch("rvq_ema_stats");
}

}  // extern "C"
