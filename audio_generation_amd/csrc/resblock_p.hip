// Fused causal residual block, persistent "ring" form for gfx950:
//
//     y = leaky( x + W2 . leaky( W1 (*)_dil x + b1 ) + b2 )        (networks/vae.py:113-117 + the activation that
//                                                                   follows the block, vae.py:130-135 / 193-198)
//
// Same arithmetic and register-resident GEMM1 -> GEMM2 hand-over as resblock_mfma.hip; what changes is how the
// operands reach the matrix pipe.  Measured on the first kernel (tools/rb_probe.hip, PMC): the main loop holds the
// pipe at ~80 %, the cost sits in the vector-memory path -- every wave fetched the SAME weights from L2 into
// registers (4x redundant, 64-byte-strided 16-byte gathers) and staged the input with dword LDS-DMA issued from a
// scalar loop (20 instructions + address arithmetic per wave and chunk).  Here:
//
//   * a workgroup is PERSISTENT: it walks over its tiles (time blocks of BN columns of one clip) and keeps ONE
//     software pipeline running across tile boundaries -- the operands of the next tile's first chunk are already in
//     LDS when the current tile's GEMM2 / epilogue run, and there is no per-tile prologue;
//   * BOTH operands live in LDS: a ring of three slots, each holding one chunk of CCH input channels = the
//     weights [CCH x 7 taps x C rows] (shared by the four waves: one L2 read per workgroup instead of four) and
//     the input rows [CCH x (BN + halo)].  Both are contiguous / 16-byte-cell copies issued as dwordx4 LDS-DMA
//     (1 KiB per instruction, addresses precomputed once per kernel): ~7 vector-memory instructions per wave and
//     chunk, all of them at the top of an interval, none between the MFMAs;
//   * the weights come from the layer's "tile image" (common.hpp: tile_image_index) whose chunk is one contiguous
//     block; an A fragment is one conflict-free ds_read_b128 (4 k-steps);
//   * chunk q+2 is requested right after the barrier that ends interval q-1 and is waited for before the barrier
//     that ends interval q, so chunk q+1 is complete one full interval before its first read: the operands of the
//     next chunk's first phase are read BEFORE the barrier, and the barrier only orders "slot q is free";
//   * every LDS offset (tap, k-step, column block) is a compile-time immediate: the kernel is instantiated per
//     (C, dilation); operand registers ping-pong between two sets over the fully unrolled 7 taps (no copies).
#include "mfma_tile.hpp"

namespace agx {

// 16 bytes per lane global -> LDS (1 KiB per wave instruction); LDS destination = wave-uniform base + lane * 16
__device__ __forceinline__ void glds_b128(const float *gsrc_lane, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MW, int NW, int CCH, int D, int NS = 3>
struct RbpGeom {
    static constexpr int C = 32 * MW, BN = 128 * NW, J = 7;
    static constexpr int P = (J - 1) * D;              // causal left pad of the stride-1 conv (vae.py:32)
    static constexpr int PA = (P + 3) / 4 * 4;         // the LDS row starts PA (a multiple of 4) before the tile: 16-byte cells
    static constexpr int SHIFT = PA - P;
    static constexpr int SPANP = BN + PA;              // floats per LDS input row
    static constexpr int NCELL = SPANP / 4;
    static constexpr int KS = CCH / 2;                 // MFMA k-steps per (chunk, tap) phase
    static constexpr int NCH = C / CCH;                // chunks per tile
    static constexpr int AFL = CCH * J * C;            // floats of weights per chunk
    static constexpr int BFL = CCH * SPANP;            // floats of input per chunk
    static constexpr int SLOT = AFL + BFL;
    static constexpr int NSLOT = NS;                   // ring slots: 3 = the next chunk is complete one interval early (its first
                                                       // operands are read before the barrier), 2 = it completes AT the barrier
    static constexpr int HS = MW > 4 ? MW / 2 : MW;    // output row blocks per GEMM2 pass (register budget: 16 * HS * NW accumulators)
    static constexpr int NPA = AFL / 256;              // 1 KiB DMA pieces of the weight chunk
    static constexpr int RA = (NPA + 3) / 4;           // ... per wave
    static constexpr int NCB = CCH * NCELL;            // 16-byte cells of the input chunk
    static constexpr int RB = (NCB + 255) / 256;       // ... DMA rounds (256 cells each)
    static constexpr int BIAS0 = NSLOT * SLOT;         // [2][C] floats behind the ring: b1, b2 (zeros when absent)
    static constexpr size_t LDS_BYTES = size_t(NSLOT * SLOT + 2 * C) * sizeof(float);
    static_assert(AFL % 256 == 0, "weight chunk must be whole 1 KiB pieces");
    static_assert(KS == 2 || KS == 4 || KS == 8, "chunk of 4, 8 or 16 channels");
};

// operand registers of one (chunk, tap) phase
template <int MW, int NW, int KS>
struct Frag {
    float a[KS][MW];
    float b[KS][NW];
};

template <int MW, int NW, int CCH, int D>
__device__ __forceinline__ void load_frag(Frag<MW, NW, CCH / 2> &f, const float *__restrict__ As,
                                          const float *__restrict__ Bs, int j) {
    using G = RbpGeom<MW, NW, CCH, D>;
    constexpr int KS = G::KS;
    // As / Bs already carry the lane-dependent part; everything below is an immediate offset
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        if (KS == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(As + (j * G::C + i * 32) * 4);
            f.a[0][i] = v[0], f.a[1][i] = v[1], f.a[2][i] = v[2], f.a[3][i] = v[3];
        } else if (KS == 2) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(As + (j * G::C + i * 32) * 4);
            f.a[0][i] = v[0], f.a[1][i] = v[1];
        } else {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(As + ((h2 * G::J + j) * G::C + i * 32) * 4);
                f.a[4 * h2 + 0][i] = v[0], f.a[4 * h2 + 1][i] = v[1], f.a[4 * h2 + 2][i] = v[2], f.a[4 * h2 + 3][i] = v[3];
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int k = 0; k < NW; ++k) f.b[ks][k] = Bs[ks * G::SPANP + k * 32 + j * D];
}

template <int MW, int NW, int KS>
__device__ __forceinline__ void mfma_frag(f32x16 (&acc)[MW][NW], const Frag<MW, NW, KS> &f) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int k = 0; k < NW; ++k)
                acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[ks][i], f.b[ks][k], acc[i][k], 0, 0, 0);
}

// One phase: request the operands of the NEXT phase, run this phase's MFMAs, with the LDS reads threaded
// between the MFMAs (an MFMA holds the issue port for a fraction of its 64 cycles).
template <int MW, int NW, int CCH, int D>
__device__ __forceinline__ void phase(f32x16 (&acc)[MW][NW], const Frag<MW, NW, CCH / 2> &cur,
                                      Frag<MW, NW, CCH / 2> &nxt, const float *__restrict__ As,
                                      const float *__restrict__ Bs, int jn) {
    constexpr int KS = CCH / 2;
    load_frag<MW, NW, CCH, D>(nxt, As, Bs, jn);
    mfma_frag<MW, NW, KS>(acc, cur);
    constexpr int NDS = MW * (KS == 8 ? 2 : 1) + KS * NW, NMF = KS * MW * NW;
    constexpr int PER = NMF / NDS > 0 ? NMF / NDS : 1;
#pragma unroll
    for (int g = 0; g < NDS; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MW, int NW, int CCH, int D, int NS>
__global__ __launch_bounds__(256, 2) void resblock_p_kernel(ConvPlan p, int tiles_per_clip, int ntiles, int step_b,
                                                            int step_t, int post_act, int stagger,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ wt1, const float *__restrict__ b1,
                                                            const float *__restrict__ wt2, const float *__restrict__ b2,
                                                            float *__restrict__ y) {
    using G = RbpGeom<MW, NW, CCH, D, NS>;
    constexpr int C = G::C, BN = G::BN, KS = G::KS, NCH = G::NCH, SLOT = G::SLOT, AFL = G::AFL, HS = G::HS;
    constexpr bool PRE3 = G::NSLOT == 3;
    constexpr bool EARLY_RES = MW <= 4;   // residual requested during the tile's last interval (C = 256: no registers to park it in)
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [3][AFL + BFL]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA destinations stay in SGPRs
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = wave * (32 * NW);
    const int Lin = p.Lin;

    if (stagger > 0) {   // knob rb_stagger: every other workgroup of the grid starts late (de-synchronises the two workgroups of a CU)
        if ((blockIdx.x / (gridDim.x / 2 > 0 ? gridDim.x / 2 : 1)) & 1)
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }

    // ---- per-lane constants of the DMA (the same for every chunk and tile) ------------------------------------
    unsigned boffB[G::RB];   // byte offset of this lane's 16-byte cell inside the chunk's rows
    int colB[G::RB];         // its first column relative to the LDS row start; < 0: no cell in that round
#pragma unroll
    for (int r = 0; r < G::RB; ++r) {
        const int e = wave * 64 + lane + 256 * r;
        const int row = e / G::NCELL, col = e - row * G::NCELL;
        colB[r] = e < G::NCB ? 4 * col : -(1 << 28);
        boffB[r] = unsigned(row * Lin + 4 * col) * 4u;
    }
    // consumer-side lane offsets (floats, relative to a slot)
    const int aLane = (KS == 4 ? lh * G::J * C * 4 : (KS == 8 ? 2 * lh * G::J * C * 4 : lh * 2)) + li * 4;
    const int bLane = AFL + lh * KS * G::SPANP + n0 + li + G::SHIFT;

    // this workgroup's tiles: blockIdx.x, + gridDim.x, ...; (clip, time block) advance by (step_b, step_t) with carry
    const int my_tiles = int(blockIdx.x) < ntiles ? (ntiles - 1 - int(blockIdx.x)) / int(gridDim.x) + 1 : 0;
    const int nq = my_tiles * NCH;  // chunks this workgroup consumes
    const int first_b = int(blockIdx.x) / tiles_per_clip, first_t = int(blockIdx.x) - first_b * tiles_per_clip;

    // DMA cursor: the next chunk to request (runs two chunks ahead of the MFMAs, across tile boundaries)
    int iq = 0, ic = 0, ib = first_b, it = first_t;
    auto issue = [&]() {
        if (iq >= nq) return;
        float *slot = lds + (iq % G::NSLOT) * SLOT;
        // weights: one contiguous block of the tile image, 1 KiB pieces dealt round-robin to the waves
        const char *wsrc = reinterpret_cast<const char *>(wt1 + size_t(ic) * AFL);
#pragma unroll
        for (int r = 0; r < G::RA; ++r) {
            const int pi = wave + 4 * r;
            if (pi < G::NPA) glds_b128(reinterpret_cast<const float *>(wsrc + unsigned(pi * 1024 + lane * 16)), slot + pi * 256);
        }
        // input rows: 16-byte cells; cells outside [0, Lvalid) are written as zeros by the lane that owns them
        const int in0a = it * BN - G::PA;
        const char *xsrc = reinterpret_cast<const char *>(x + (size_t(ib) * C + size_t(ic) * CCH) * Lin + in0a);
#pragma unroll
        for (int r = 0; r < G::RB; ++r) {
            float *dst = slot + AFL + (wave * 64 + 256 * r) * 4;
            const int pos = in0a + colB[r];
            const bool mine = colB[r] >= 0, ok = mine && pos >= 0 && pos < p.Lvalid;
            if (ok) glds_b128(reinterpret_cast<const float *>(xsrc + boffB[r]), dst);
            if (mine && !ok) *reinterpret_cast<f32x4 *>(dst + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ++iq;
        if (++ic == NCH) {
            ic = 0;
            ib += step_b;
            it += step_t;
            if (it >= tiles_per_clip) it -= tiles_per_clip, ++ib;
        }
    };

    if (nq == 0) return;
    for (int i = tid; i < 2 * C; i += 256)   // biases: read once per kernel, served from LDS afterwards
        lds[G::BIAS0 + i] = i < C ? (b1 ? b1[i] : 0.f) : (b2 ? b2[i - C] : 0.f);
    issue();
    if (PRE3) issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    Frag<MW, NW, KS> f0, f1;
    if (PRE3) load_frag<MW, NW, CCH, D>(f0, lds + aLane, lds + bLane, 0);

    int q = 0, cb = first_b, ct = first_t;
    for (int k = 0; k < my_tiles; ++k) {
        const int b = cb, t0 = ct * BN;
        cb += step_b;
        ct += step_t;
        if (ct >= tiles_per_clip) ct -= tiles_per_clip, ++cb;

        f32x16 acc[MW][NW];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int kk = 0; kk < NW; ++kk)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][kk][r] = 0.f;
        if (!PRE3)   // 2-slot ring: the tile's first operands are read here (its first chunk landed at the last barrier)
            load_frag<MW, NW, CCH, D>(f0, lds + (q % G::NSLOT) * SLOT + aLane, lds + (q % G::NSLOT) * SLOT + bLane, 0);

        // ---- GEMM1 over the tile's chunks, two per iteration (the operand sets swap roles every 7 phases) -------
        // One interval: request chunk q+2, run the 7 tap phases of chunk q (the last one already reads the first
        // operands of chunk q+1), wait for this wave's DMA, barrier.
#define AGX_RBP_CHUNK(FA, FB, PRE, TAILC)                                                                            \
    {                                                                                                                \
        issue();                                                                                                     \
        PRE;                                                                                                         \
        const float *As = lds + (q % G::NSLOT) * SLOT + aLane, *Bs = lds + (q % G::NSLOT) * SLOT + bLane;            \
        const float *An = lds + ((q + 1) % G::NSLOT) * SLOT + aLane, *Bn = lds + ((q + 1) % G::NSLOT) * SLOT + bLane; \
        phase<MW, NW, CCH, D>(acc, FA, FB, As, Bs, 1);                                                               \
        phase<MW, NW, CCH, D>(acc, FB, FA, As, Bs, 2);                                                               \
        phase<MW, NW, CCH, D>(acc, FA, FB, As, Bs, 3);                                                               \
        phase<MW, NW, CCH, D>(acc, FB, FA, As, Bs, 4);                                                               \
        phase<MW, NW, CCH, D>(acc, FA, FB, As, Bs, 5);                                                               \
        phase<MW, NW, CCH, D>(acc, FB, FA, As, Bs, 6);                                                               \
        if (PRE3) {                                                                                                  \
            phase<MW, NW, CCH, D>(acc, FA, FB, An, Bn, 0); /* next chunk's first phase: complete since the last barrier */ \
        } else {                                                                                                     \
            mfma_frag<MW, NW, KS>(acc, FA);                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's part of the requested chunk has landed */     \
        __syncthreads();                                 /* everyone's has; slot q is free */                         \
        if (!PRE3 && !(TAILC)) load_frag<MW, NW, CCH, D>(FB, An, Bn, 0);                                              \
        ++q;                                                                                                         \
    }
        const char *xb = reinterpret_cast<const char *>(x + size_t(b) * C * Lin);
        char *yb = reinterpret_cast<char *>(y + size_t(b) * C * Lin);
        int linv = Lin;                       // opaque per-tile copies: keep the row offsets / fragment addresses from being
        asm volatile("" : "+v"(linv));        // hoisted out of the tile loop (they would be live across the whole main loop)
        // GEMM2's accumulator starts from the residual: out = x (+ b2 below) + W2 . h.  Its 16*MW*NW loads go out at
        // the top of the tile's LAST interval and have the whole interval to arrive.
        f32x16 out[HS][NW];
        auto load_residual = [&](int pass) {
#pragma unroll
            for (int io = 0; io < HS; ++io)
#pragma unroll
                for (int kk = 0; kk < NW; ++kk) {
                    const int tc = min(t0 + n0 + kk * 32 + li, Lin - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        out[io][kk][r] = *reinterpret_cast<const float *>(
                            xb + unsigned(((pass * HS + io) * 32 + acc_row(r, lh)) * linv + tc) * 4u);
                }
        };
        for (int c = 0; c < NCH - 2; c += 2) {
            AGX_RBP_CHUNK(f0, f1, (void)0, false)
            AGX_RBP_CHUNK(f1, f0, (void)0, false)
        }
        AGX_RBP_CHUNK(f0, f1, (void)0, false)
        AGX_RBP_CHUNK(f1, f0, if (EARLY_RES) load_residual(0), true)
#undef AGX_RBP_CHUNK

        // ---- tail: the first GEMM2 weight block travels while the activation runs -------------------------------------
        unsigned w2off = unsigned(lh * C + li) * 16u;
        asm volatile("" : "+v"(w2off));
        const char *w2b = reinterpret_cast<const char *>(wt2);
        // GEMM2 weight block (i, g) for output row blocks pass*HS ..: hidden channels i*32 + 8g + 4lh + (0..3) = block
        // 8i + 2g + lh of the tile image
        auto load_w2 = [&](f32x4 (&a)[HS], int blk, int pass) {
#pragma unroll
            for (int io = 0; io < HS; ++io)
                a[io] = *reinterpret_cast<const f32x4 *>(w2b + (w2off + unsigned((2 * blk * C + (pass * HS + io) * 32) * 16)));
        };
        f32x4 wa[2][HS];
        load_w2(wa[0], 0, 0);
        __builtin_amdgcn_sched_barrier(0);

        // ---- hidden activation, in registers (bias from LDS) ---------------------------------------------------------
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bq = *reinterpret_cast<const f32x4 *>(lds + G::BIAS0 + i * 32 + 8 * g + 4 * lh);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const float v = acc[i][kk][4 * g + s4] + bq[s4];
                        acc[i][kk][4 * g + s4] = v > 0.f ? v : v * p.slope;
                    }
            }

#pragma unroll
        for (int pass = 0; pass < MW / HS; ++pass) {
            if (pass > 0 || !EARLY_RES) {   // (C = 256 only) this latency is exposed, twice per 8192-MFMA tile
                if (pass > 0) load_w2(wa[0], 0, pass);
                load_residual(pass);
            }
#pragma unroll
            for (int io = 0; io < HS; ++io)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b2q = *reinterpret_cast<const f32x4 *>(lds + G::BIAS0 + C + (pass * HS + io) * 32 + 8 * g + 4 * lh);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int kk = 0; kk < NW; ++kk) out[io][kk][4 * g + s4] += b2q[s4];
                }
            // ---- GEMM2: out += W2 . h, B operand = the accumulator registers (resblock_mfma.hip); weights one block ahead
#pragma unroll
            for (int blk = 0; blk < 4 * MW; ++blk) {
                const int i = blk >> 2, g = blk & 3;
                if (blk + 1 < 4 * MW) load_w2(wa[(blk + 1) & 1], blk + 1, pass);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int io = 0; io < HS; ++io)
#pragma unroll
                        for (int kk = 0; kk < NW; ++kk)
                            out[io][kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[blk & 1][io][s4], acc[i][kk][4 * g + s4],
                                                                              out[io][kk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue: trailing activation, store -------------------------------------------------------------------
#pragma unroll
            for (int io = 0; io < HS; ++io)
#pragma unroll
                for (int kk = 0; kk < NW; ++kk) {
                    const int t = t0 + n0 + kk * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = out[io][kk][r];
                        if (post_act) v = leaky(v, p.slope);
                        if (t < Lin)
                            *reinterpret_cast<float *>(yb + unsigned(((pass * HS + io) * 32 + acc_row(r, lh)) * linv + t) * 4u) = v;
                    }
                }
        }
    }
}

template <int MW, int NW, int CCH, int D, int NS>
static int launch_rbp(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2,
                      const float *b2, float *y, int post_act, hipStream_t st) {
    using G = RbpGeom<MW, NW, CCH, D, NS>;
    auto kern = resblock_p_kernel<MW, NW, CCH, D, NS>;
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return fail(AGX_ERR_LAUNCH, "resblock_p: cannot query the device");
        n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles_per_clip = ceil_div(p.Lin, G::BN);
    const int64_t ntiles64 = int64_t(tiles_per_clip) * p.B;
    if (ntiles64 > (1 << 30)) return fail(AGX_ERR_BAD_SHAPE, "resblock_p: too many tiles");
    const int ntiles = int(ntiles64);
    const int wg_per_cu = int((160 * 1024) / G::LDS_BYTES) >= 2 ? 2 : 1;
    int grid = n_cu * wg_per_cu;
    if (grid > ntiles) grid = ntiles;
    // the tile images follow the standard image and the dim0 scale scratch in the packed buffers (common.hpp)
    const float *wt1 = w1 + packed_weight_floats(G::C, G::J, G::C) + G::C;
    const float *wt2 = w2 + packed_weight_floats(G::C, 1, G::C) + G::C;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, st, p, tiles_per_clip, ntiles, grid / tiles_per_clip,
                       grid % tiles_per_clip, post_act, tuning().rb_stagger, x, wt1, b1, wt2, b2, y);
    return check_launch("resblock_p");
}

// shapes the persistent kernel is instantiated for: C in {32, 64, 128, 256}, k = 7, dilation in {1, 3, 9}, L % 4 == 0
bool resblock_p_supported(const ConvPlan &p) {
    if (p.prec != 0 || p.Cin != p.Cout || p.s != 1 || p.q != 1 || p.J != 7 || p.G != 1) return false;
    if (p.Lvalid != p.Lin || p.Lt != p.Lin || p.Lin % 4 != 0 || p.Lin < 4) return false;
    if (p.Cin != 32 && p.Cin != 64 && p.Cin != 128 && p.Cin != 256) return false;
    return p.d == 1 || p.d == 3 || p.d == 9;
}

const char *resblock_p_variant(const ConvPlan &p) {
    switch (p.Cin) {
        case 32: return "resblock_p<1,4,8>";
        case 64: return "resblock_p<2,2,8>";
        case 128: return "resblock_p<4,1,4>";
        default: return "resblock_p<8,1,4>";
    }
}

int launch_resblock_p(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2,
                      const float *b2, float *y, int post_act, hipStream_t st) {
    if (!resblock_p_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "resblock_p: unsupported shape");
#define AGX_RBP(MW, NW, CCH, NS)                                                                 \
    (p.d == 1 ? launch_rbp<MW, NW, CCH, 1, NS>(p, x, w1, b1, w2, b2, y, post_act, st)            \
     : p.d == 3 ? launch_rbp<MW, NW, CCH, 3, NS>(p, x, w1, b1, w2, b2, y, post_act, st)          \
                : launch_rbp<MW, NW, CCH, 9, NS>(p, x, w1, b1, w2, b2, y, post_act, st))
    switch (p.Cin) {
        case 32: return AGX_RBP(1, 4, 8, 3);
        case 64: return AGX_RBP(2, 2, 8, 3);
        case 128: return AGX_RBP(4, 1, 4, 3);
        default: return AGX_RBP(8, 1, 4, 2);   // 28.7 KB of weights per 4-channel chunk: two slots keep two workgroups per CU
    }
#undef AGX_RBP
}

}  // namespace agx
