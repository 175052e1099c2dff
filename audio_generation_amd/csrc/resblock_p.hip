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

// probe build (tools/rbp_probe.hip): s_memtime at the phase boundaries of ONE interval (the workgroup's stamp_q-th) per wave
#ifdef AGX_STAMPS
#define AGX_RSTAMP(slot)                                                                           \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (q == AGX_STAMP_Q && lane == 0) g_stamps[(blockIdx.x * 4 + wave) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define AGX_TSTAMP(slot)                                                                           \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (k == 2 && lane == 0) g_stamps[(blockIdx.x * 4 + wave) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define AGX_RSTAMP(slot) ((void)0)
#define AGX_TSTAMP(slot) ((void)0)
#endif

// 16 bytes per lane global -> LDS (1 KiB per wave instruction); LDS destination = wave-uniform base + lane * 16
__device__ __forceinline__ void glds_b128(const float *gsrc_lane, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 1 KiB of zeros in the code object: the DMA source of input cells outside [0, L) (causal left pad, ragged last tile) and
// of the instruction slots a wave has no piece for -- every DMA instruction is issued unconditionally, branch-free.
__device__ __attribute__((aligned(1024))) float g_rbp_zero_page[256] = {0.f};

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
    static constexpr int NCB = CCH * NCELL;            // 16-byte cells of the input chunk
    static constexpr int NIB = (NCB + 63) / 64;        // ... = DMA instructions (64 consecutive cells each); the last one may
    static constexpr int BFLP = NIB * 256;             //     overrun the rows by < 1 KiB: the slot's input region is padded to it
    static constexpr int SLOT = AFL + BFLP;
    static constexpr int NSLOT = NS;                   // ring slots: 3 = the next chunk is complete one interval early (its first
                                                       // operands are read before the barrier), 2 = it completes AT the barrier
    static constexpr int HS = MW > 4 ? MW / 2 : MW;    // output row blocks per GEMM2 pass (register budget: 16 * HS * NW accumulators)
    static constexpr int NPA = AFL / 256;              // 1 KiB DMA pieces (= instructions) of the weight chunk
    static constexpr int RA = (NPA + 3) / 4;           // ... per wave (instruction n belongs to wave n % 4)
    static constexpr int RB = (NIB + 3) / 4;
    static constexpr int NOPS = RA + RB;               // DMA instructions per wave and interval, spread over the 7 phases
    static constexpr int BIAS0 = NSLOT * SLOT;         // [2][C] floats behind the ring: b1, b2 (zeros when absent)
    static constexpr int SCR0 = BIAS0 + 2 * C;         // 1 KiB per wave: transposition scratch of the tile tail (8 rows x 32 columns)
    static constexpr size_t LDS_BYTES = size_t(NSLOT * SLOT + 2 * C + 4 * 256) * sizeof(float);
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
template <int MW, int NW, int CCH, int D, class Extra>
__device__ __forceinline__ void phase(f32x16 (&acc)[MW][NW], const Frag<MW, NW, CCH / 2> &cur,
                                      Frag<MW, NW, CCH / 2> &nxt, const float *__restrict__ As,
                                      const float *__restrict__ Bs, int jn, Extra &&extra) {
    constexpr int KS = CCH / 2;
    load_frag<MW, NW, CCH, D>(nxt, As, Bs, jn);
    extra();   // this phase's share of the interval's DMA instructions (address selects + 1-2 global_load_lds)
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
                                                            int step_t, int post_act,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ wt1, const float *__restrict__ b1,
                                                            const float *__restrict__ wt2, const float *__restrict__ b2,
                                                            float *__restrict__ y) {
    using G = RbpGeom<MW, NW, CCH, D, NS>;
    constexpr int C = G::C, BN = G::BN, KS = G::KS, NCH = G::NCH, SLOT = G::SLOT, AFL = G::AFL, HS = G::HS;
    constexpr bool PRE3 = G::NSLOT == 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [3][AFL + BFL]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA destinations stay in SGPRs
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = wave * (32 * NW);
    const int Lin = p.Lin;

    // ---- per-lane constants of the DMA (the same for every chunk and tile) ------------------------------------
    // Instruction slot (k, wave) of an interval moves piece n = wave + 4 k; a slot beyond the chunk's last piece repeats
    // piece n % count (same bytes to the same place: harmless), so that every slot issues unconditionally, branch-free.
    // input instruction n covers cells 64 n .. 64 n + 63 of the chunk (cell e = row e / NCELL, columns 4 (e % NCELL) ..)
    unsigned boffB[G::RB];   // byte offset of this lane's cell inside the chunk's rows
    int colB[G::RB];         // its first column relative to the LDS row start; hugely negative: beyond the chunk
    int nB[G::RB];
#pragma unroll
    for (int r = 0; r < G::RB; ++r) {
        nB[r] = (wave + 4 * r) % G::NIB;
        const int e = nB[r] * 64 + lane;
        const int row = e / G::NCELL, col = e - row * G::NCELL;
        colB[r] = e < G::NCB ? 4 * col : -(1 << 28);
        boffB[r] = unsigned(row * Lin + 4 * col) * 4u;
    }
    const char *zpage = reinterpret_cast<const char *>(g_rbp_zero_page) + lane * 16;
    // consumer-side lane offsets (floats, relative to a slot)
    const int aLane = (KS == 4 ? lh * G::J * C * 4 : (KS == 8 ? 2 * lh * G::J * C * 4 : lh * 2)) + li * 4;
    const int bLane = AFL + lh * KS * G::SPANP + n0 + li + G::SHIFT;

    // this workgroup's tiles: blockIdx.x, + gridDim.x, ...; (clip, time block) advance by (step_b, step_t) with carry
    const int my_tiles = int(blockIdx.x) < ntiles ? (ntiles - 1 - int(blockIdx.x)) / int(gridDim.x) + 1 : 0;
    const int nq = my_tiles * NCH;  // chunks this workgroup consumes
    const int first_b = int(blockIdx.x) / tiles_per_clip, first_t = int(blockIdx.x) - first_b * tiles_per_clip;

    // DMA cursor: the next chunk to request (runs ahead of the MFMAs, across tile boundaries).  begin_chunk() fixes the
    // interval's uniform bases (all of it incremental: no multiplication or division per interval); dma_op(k), k < NOPS,
    // issues ONE instruction.  A cell outside the signal reads the zero page; so does every request past the workgroup's
    // last chunk (into a ring slot nobody reads any more).
    int iq = 0, ic = 0, ib = first_b, it = first_t, isl = 0;   // chunk index in the stream / in the tile, tile coordinates, ring slot
    const char *w_next = reinterpret_cast<const char *>(wt1);
    const char *x_next = reinterpret_cast<const char *>(x + size_t(ib) * C * Lin + (it * BN - G::PA));
    float *d_slot = lds;
    const char *d_w = nullptr, *d_x = nullptr;
    int d_in0a = 0;
    bool d_live = false;
    auto begin_chunk = [&]() {
        d_live = iq < nq;
        d_slot = lds + isl * SLOT;
        d_w = w_next;
        d_x = x_next;
        d_in0a = it * BN - G::PA;
        ++iq;
        isl = isl + 1 == G::NSLOT ? 0 : isl + 1;
        w_next += AFL * sizeof(float);
        x_next += size_t(CCH) * Lin * sizeof(float);
        if (++ic == NCH) {   // next tile
            ic = 0;
            ib += step_b;
            it += step_t;
            if (it >= tiles_per_clip) it -= tiles_per_clip, ++ib;
            w_next = reinterpret_cast<const char *>(wt1);
            x_next = reinterpret_cast<const char *>(x + size_t(ib) * C * Lin + (it * BN - G::PA));
        }
    };
    auto dma_op = [&](int k) {
        if (k < G::RA) {              // weights: 1 KiB pieces of one contiguous block of the tile image
            const int n = (wave + 4 * k) % G::NPA;
            const char *src = d_live ? d_w + n * 1024 + lane * 16 : zpage;
            glds_b128(reinterpret_cast<const float *>(src), d_slot + n * 256);
        } else if (k < G::NOPS) {     // input rows, 64 cells per instruction
            const int r = k - G::RA;
            const int pos = d_in0a + colB[r];
            const bool ok = d_live && pos >= 0 && pos < p.Lvalid;
            const char *src = ok ? d_x + boffB[r] : zpage;
            glds_b128(reinterpret_cast<const float *>(src), d_slot + AFL + nB[r] * 256);
        }
    };
    auto issue = [&]() {
        begin_chunk();
#pragma unroll
        for (int k = 0; k < G::NOPS; ++k) dma_op(k);
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

    // ---- tile tail <-> memory: 16-byte pieces through a per-wave LDS scratch ----------------------------------------------
    // A piece = rows 8g .. 8g+7 of row block io x the 32 columns of column block kk: one instruction moves 8 rows x 128 B
    // (lane -> row l / 8, columns 4 (l % 8) ..).  In the accumulator layout lane (li, lh) holds rows 8g + 4lh + (0..3) of
    // column li in registers 4g .. 4g+3; the 1 KiB scratch (8 rows x 32 columns) converts between the two.
    // The residual tile is requested in pieces spread over the phases of the tile's LAST interval, parked in the `out`
    // registers, brought into the accumulator layout in the tail (GEMM2's accumulator starts from x + b2), and the finished
    // tile is stored right after GEMM2.  (Measured and dropped: deferring the stores into the next tile's first phases --
    // 127 vs 131 TFLOP/s at C = 64; a fifth, DMA-only wave; start-time staggers within a CU and across the chip.)
    constexpr bool EARLY = MW <= 4;              // residual requested during the tile's last interval (C = 256: no registers to park it in)
    constexpr int NPIECE = HS * 4 * NW;
    f32x16 out[HS][NW];
    float *scr = lds + G::SCR0 + wave * 256;
    const int prow = lane >> 3, pcol = (lane & 7) * 4;
    auto store_piece = [&](int pc, int pass, char *ybase, int tcol0, int linv) {
        const int io = pc / (4 * NW), g = (pc / NW) % 4, kk = pc % NW;
        // (all scratch accesses are float-typed and fenced for the compiler: a vector-typed store next to scalar loads of the
        //  same bytes was reordered under type-based alias analysis)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) scr[(4 * lh + s4) * 32 + li] = out[io][kk][4 * g + s4];
        asm volatile("" ::: "memory");
        f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = scr[lane * 4 + e];
        asm volatile("" ::: "memory");
        const int tq = tcol0 + n0 + kk * 32 + pcol;
        if (tq < Lin)   // L % 4 == 0: a piece is inside the clip or outside it
            *reinterpret_cast<f32x4 *>(ybase + unsigned(((pass * HS + io) * 32 + 8 * g + prow) * linv + tq) * 4u) = v4;
    };
    auto load_piece = [&](int pc, int pass, const char *xbase, int tcol0, int linv) {
        const int io = pc / (4 * NW), g = (pc / NW) % 4, kk = pc % NW;
        const int tq = min(tcol0 + n0 + kk * 32 + pcol, Lin - 4);   // columns beyond the clip are never stored
        const f32x4 v = *reinterpret_cast<const f32x4 *>(xbase + unsigned(((pass * HS + io) * 32 + 8 * g + prow) * linv + tq) * 4u);
        out[io][kk][4 * g + 0] = v[0], out[io][kk][4 * g + 1] = v[1], out[io][kk][4 * g + 2] = v[2], out[io][kk][4 * g + 3] = v[3];
    };

    int q = 0, qs = 0, cb = first_b, ct = first_t;   // consumed chunks, their ring slot, tile coordinates
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
            load_frag<MW, NW, CCH, D>(f0, lds + qs * SLOT + aLane, lds + qs * SLOT + bLane, 0);

        const char *xb = reinterpret_cast<const char *>(x + size_t(b) * C * Lin);
        char *yb = reinterpret_cast<char *>(y + size_t(b) * C * Lin);
        int linv = Lin;                       // opaque per-tile copies: keep the row offsets / fragment addresses from being
        asm volatile("" : "+v"(linv));        // hoisted out of the tile loop (they would be live across the whole main loop)

        // ---- GEMM1 over the tile's chunks (the operand sets swap roles every 7 phases) -----------------------------------
        // One interval: request the chunk NSLOT-1 ahead, run the 7 tap phases of chunk q (the last one already reads the
        // first operands of chunk q+1), wait for this wave's DMA, barrier.  Each phase also carries its share of the DMA
        // instructions and, in the tile's LAST chunk, of the residual loads.
#define AGX_RBP_OPS(j)                                                                                                \
    [&]() {                                                                                                           \
        dma_op(j);                                                                                                    \
        if (G::NOPS > 7) dma_op(j + 7);                                                                               \
        if (EARLY && LAST_) {                                                                                         \
            _Pragma("unroll") for (int pc = (j); pc < NPIECE; pc += 7) load_piece(pc, 0, xb, t0, linv);               \
        }                                                                                                             \
    }
#define AGX_RBP_CHUNK(FA, FB, LAST)                                                                                  \
    {                                                                                                                \
        AGX_RSTAMP(0);                                                                                               \
        constexpr bool LAST_ = LAST;                                                                                 \
        begin_chunk();                                                                                               \
        AGX_RSTAMP(1);                                                                                               \
        const int qsn = qs + 1 == G::NSLOT ? 0 : qs + 1;                                                             \
        const float *As = lds + qs * SLOT + aLane, *Bs = lds + qs * SLOT + bLane;                                    \
        const float *An = lds + qsn * SLOT + aLane, *Bn = lds + qsn * SLOT + bLane;                                  \
        phase<MW, NW, CCH, D>(acc, FA, FB, As, Bs, 1, AGX_RBP_OPS(0));                                               \
        AGX_RSTAMP(2);                                                                                               \
        phase<MW, NW, CCH, D>(acc, FB, FA, As, Bs, 2, AGX_RBP_OPS(1));                                               \
        AGX_RSTAMP(3);                                                                                               \
        phase<MW, NW, CCH, D>(acc, FA, FB, As, Bs, 3, AGX_RBP_OPS(2));                                               \
        AGX_RSTAMP(4);                                                                                               \
        phase<MW, NW, CCH, D>(acc, FB, FA, As, Bs, 4, AGX_RBP_OPS(3));                                               \
        AGX_RSTAMP(5);                                                                                               \
        phase<MW, NW, CCH, D>(acc, FA, FB, As, Bs, 5, AGX_RBP_OPS(4));                                               \
        AGX_RSTAMP(6);                                                                                               \
        phase<MW, NW, CCH, D>(acc, FB, FA, As, Bs, 6, AGX_RBP_OPS(5));                                               \
        AGX_RSTAMP(7);                                                                                               \
        if (PRE3) {                                                                                                  \
            phase<MW, NW, CCH, D>(acc, FA, FB, An, Bn, 0, AGX_RBP_OPS(6)); /* next chunk's first phase: complete since the last barrier */ \
        } else {                                                                                                     \
            AGX_RBP_OPS(6)();                                                                                        \
            mfma_frag<MW, NW, KS>(acc, FA);                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        AGX_RSTAMP(8);                                                                                               \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's part of the requested chunk has landed */     \
        AGX_RSTAMP(9);                                                                                               \
        __syncthreads();                                 /* everyone's has; slot q is free */                         \
        AGX_RSTAMP(10);                                                                                              \
        if (!PRE3 && !(LAST)) load_frag<MW, NW, CCH, D>(FB, An, Bn, 0);                                               \
        ++q; (void)q;                                                                                                \
        qs = qsn;                                                                                                    \
    }
        for (int c = 0; c < NCH - 2; c += 2) {
            AGX_RBP_CHUNK(f0, f1, false)
            AGX_RBP_CHUNK(f1, f0, false)
        }
        AGX_RBP_CHUNK(f0, f1, false)
        AGX_RBP_CHUNK(f1, f0, true)
#undef AGX_RBP_CHUNK
#undef AGX_RBP_OPS

        // ---- tail: the first GEMM2 weight block travels while the activation runs -------------------------------------
        AGX_TSTAMP(11);
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

        AGX_TSTAMP(12);
#pragma unroll
        for (int pass = 0; pass < MW / HS; ++pass) {
            if (pass > 0 || !EARLY) {   // (C = 256 only) the residual is requested here: exposed latency, twice per 8192-MFMA tile
                if (pass > 0) load_w2(wa[0], 0, pass);
#pragma unroll
                for (int pc = 0; pc < NPIECE; ++pc) load_piece(pc, pass, xb, t0, linv);
            }
            // residual pieces -> accumulator layout, + b2: GEMM2's accumulator starts from x + b2
#pragma unroll
            for (int io = 0; io < HS; ++io)
#pragma unroll
                for (int kk = 0; kk < NW; ++kk)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int e = 0; e < 4; ++e) scr[lane * 4 + e] = out[io][kk][4 * g + e];
                        asm volatile("" ::: "memory");
                        const f32x4 b2q = *reinterpret_cast<const f32x4 *>(lds + G::BIAS0 + C + (pass * HS + io) * 32 + 8 * g + 4 * lh);
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) out[io][kk][4 * g + s4] = scr[(4 * lh + s4) * 32 + li] + b2q[s4];
                        asm volatile("" ::: "memory");
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
            if (pass == 0) AGX_TSTAMP(13);
            // ---- trailing activation, store (16-byte pieces through the scratch)
            if (post_act) {
#pragma unroll
                for (int io = 0; io < HS; ++io)
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk)
#pragma unroll
                        for (int r = 0; r < 16; ++r) out[io][kk][r] = leaky(out[io][kk][r], p.slope);
            }
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) store_piece(pc, pass, yb, t0, linv);
            if (pass == 0) AGX_TSTAMP(14);
        }
    }
}

template <int MW, int NW, int CCH, int D, int NS>
static int launch_rbp(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2,
                      const float *b2, float *y, int post_act, hipStream_t st) {
    using G = RbpGeom<MW, NW, CCH, D, NS>;
    auto kern = resblock_p_kernel<MW, NW, CCH, D, NS>;
    static DeviceOnce once;
    int n_cu = 0;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, &n_cu, "resblock_p")) return rc;
    const int tiles_per_clip = ceil_div(p.Lin, G::BN);
    const int64_t ntiles64 = int64_t(tiles_per_clip) * p.B;
    if (ntiles64 > (1 << 30)) return fail(AGX_ERR_BAD_SHAPE, "resblock_p: too many tiles");
    const int ntiles = int(ntiles64);
    int wg_per_cu = int((160 * 1024) / G::LDS_BYTES) >= 2 ? 2 : 1;
    size_t lds_bytes = G::LDS_BYTES;
    if (tuning().rb_wgs == 1) wg_per_cu = 1, lds_bytes = 100 * 1024;   // diagnostic: one workgroup per CU
    int grid = n_cu * wg_per_cu;
    if (grid > ntiles) grid = ntiles;
    // the tile images follow the standard image and the dim0 scale scratch in the packed buffers (common.hpp)
    const float *wt1 = w1 + packed_weight_floats(G::C, G::J, G::C) + G::C;
    const float *wt2 = w2 + packed_weight_floats(G::C, 1, G::C) + G::C;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, st, p, tiles_per_clip, ntiles, grid / tiles_per_clip,
                       grid % tiles_per_clip, post_act, x, wt1, b1, wt2, b2, y);
    return check_launch("resblock_p");
}

// shapes the persistent kernel is instantiated for: C in {32, 64, 128, 256}, k = 7, dilation in {1, 3, 9}, L % 4 == 0
bool resblock_p_supported(const ConvPlan &p) {
    if (p.prec != 0 || p.Cin != p.Cout || p.s != 1 || p.q != 1 || p.J != 7 || p.G != 1) return false;
    if (p.Lvalid != p.Lin || p.Lt != p.Lin || p.Lin % 4 != 0 || p.Lin < 4) return false;
    if (p.Cin != 32 && p.Cin != 64 && p.Cin != 128 && p.Cin != 256) return false;
    if (int64_t(p.Cin) * p.Lin * 4 >= (int64_t(1) << 32)) return false;   // 32-bit byte offsets inside a clip
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
