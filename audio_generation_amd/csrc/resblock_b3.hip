// Fused causal residual block on the bf16 matrix pipe with fp32-class accuracy ("bf16x3"), persistent ring form:
//
//     y = leaky( x + W2 . leaky( W1 (*)_dil x + b1 ) + b2 )        (networks/vae.py:113-117 + the activation that
//                                                                   follows the block, vae.py:130-135 / 193-198)
//
// Arithmetic (mfma_tile.hpp): every fp32 operand is the sum of three bf16 pieces x = h + m + l (24
// significant bits); a K = 16 product block is the six bf16 MFMAs mm + hl + lh + hm + mh + hh with fp32 accumulation:
// 6 x 32 cycles against 8 x 64 on the fp32-input MFMA.  Not the bitwise fp32 FMA chain -- opt-in, never where an integer
// is decided.  Round 1 split the INPUT per tap in registers (44 vector instructions per 6 MFMAs: the kernel was bound by
// that); here every operand is split ONCE:
//
//   * weights: split by the pack kernel into the "B3 tile image" [16-channel group][tap][plane][lane half][row][8 bf16]:
//     one (group, tap) PHASE is one contiguous 96 C-byte block that LDS-DMA drops into a ring slot as it stands, and an
//     A fragment (one row block x K = 16 x one plane) is one conflict-free ds_read_b128;
//   * input: a chunk of 16 channels x (BN + halo) time steps travels global -> registers (coalesced dword loads issued
//     two phases before they are needed) -> three bf16 planes [plane][lane half][time][8 channels] in LDS, ONE split per
//     element; a B fragment of any tap is one conflict-free ds_read_b128 at an immediate offset (tap * dilation * 16 B);
//   * a workgroup (4 waves, one per SIMD, up to 512 registers per lane) is persistent and owns all C rows of a
//     128 NW-column tile: a wave holds its C x 32 NW accumulators for GEMM1 and feeds them back, split in registers, as
//     the B operand of GEMM2 (k-slot <-> accumulator-register assignment: the W2 image is packed in that order);
//   * ring: 4 one-phase weight slots (two sets of two), two plane buffers.  Phases of a chunk run in groups [0,1] [2,3]
//     [4,5] [6].  Operand reads run one MFMA step (phase, row half, column block) ahead of the MFMAs, so a group's LAST
//     operands are in registers before its last step: the group's barrier sits BEFORE that step ("early"), the DMA of the
//     group after next goes out right behind it into the set just released, and the next group's first operands are read
//     while the last step's MFMAs run.  The next chunk's loads are issued with phase 4 and split + written, one task per
//     step under that step's MFMAs, before the chunk's last barrier.  4 barriers per 7 phases (1 wave per SIMD).
#include "mfma_tile.hpp"

namespace agx {

typedef __bf16 b3x8 __attribute__((ext_vector_type(8)));

// probe build (tools/b3_probe.hip): s_memtime stamps of the workgroup's third tile, one lane per wave
#ifdef AGX_STAMPS
#define B3_TSTAMP(slot)                                                                                                \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (k == 2 && lane == 0) g_stamps[(blockIdx.x * 4 + wave) * 16 + (slot)] = __builtin_amdgcn_s_memtime();       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
#define B3_RSTAMP(slot)                                                                                                \
    do {                                                                                                               \
        if (lane == 0) g_stamps[(blockIdx.x * 4 + wave) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();             \
    } while (0)
// shader-clock stamp next to a realtime (100 MHz) one: the clock the chip holds over the kernel = d memtime / d memrealtime x 100 MHz
#define B3_CSTAMP(slot)                                                                                                \
    do {                                                                                                               \
        if (lane == 0) g_stamps[(blockIdx.x * 4 + wave) * 16 + (slot)] = __builtin_amdgcn_s_memtime();                 \
    } while (0)
#else
#define B3_TSTAMP(slot) ((void)0)
#define B3_RSTAMP(slot) ((void)0)
#define B3_CSTAMP(slot) ((void)0)
#endif

__device__ __attribute__((aligned(1024))) float g_b3_zero_page[256] = {0.f};

__device__ __forceinline__ void b3_glds_b128(const void *gsrc_lane, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int MW, int NW, int D, int NPB = 2>
struct B3Geom {
    static constexpr int C = 32 * MW, BN = 128 * NW, J = 7;
    static constexpr int P = (J - 1) * D;                  // causal left pad (vae.py:32)
    static constexpr int W = BN + P;                       // time steps per plane row (tile + halo)
    static constexpr int NCH = C / 16;                     // 16-channel chunks per tile
    static constexpr int PLANE_B = 6 * W * 16;             // one chunk: [plane 3][lane half 2][W][8 bf16]
    static constexpr int WSLOT_B = 96 * C;                 // one phase of weights: [plane 3][lane half 2][C][8 bf16]
    static constexpr int NPW = WSLOT_B / 1024;             // 1 KiB DMA pieces per phase
    static constexpr int RW = (NPW + 3) / 4;               // ... per wave
    static constexpr int NT = (2 * W + 255) / 256;         // conversion tasks (time step x 8 channels) per thread and chunk
    static constexpr int RH = MW > 4 ? 4 : MW;             // row blocks per MFMA sub-phase (operand register budget)
    static constexpr int HS = MW > 2 ? MW / 2 : MW;        // output row blocks per GEMM2 pass
    static constexpr int OFF_W = NPB * PLANE_B;            // byte offsets inside the dynamic LDS (NPB plane buffers)
    static constexpr int OFF_BIAS = OFF_W + 4 * WSLOT_B;
    static constexpr size_t LDS_BYTES = size_t(OFF_BIAS) + 2 * C * sizeof(float);
    static_assert(WSLOT_B % 1024 == 0, "a weight phase must be whole 1 KiB pieces");
};

// the six products of one K = 16 block for ONE column block, small terms first: mm hl lh hm mh hh  (products T0 .. T1-1)
template <int RH, int T0 = 0, int T1 = 6>
__device__ __forceinline__ void b3_products(f32x16 (&acc)[RH], const b3x8 (&a)[3][RH], const b3x8 (&b)[3]) {
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int i = 0; i < RH; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]], acc[i], 0, 0, 0);
}

__device__ __forceinline__ void b3_split8(const float (&x)[8], b3x8 &h, b3x8 &m, b3x8 &l) {
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

template <int MW, int NW, int D, int NPB>
__global__ __launch_bounds__(256, NPB == 1 ? 2 : 1) void resblock_b3_kernel(ConvPlan p, int tiles_per_clip, int ntiles, int step_b,
                                                             int step_t, int post_act, int prio, const float *__restrict__ x,
                                                             const char *__restrict__ wt1, const float *__restrict__ b1,
                                                             const char *__restrict__ wt2, const float *__restrict__ b2,
                                                             float *__restrict__ y) {
    using G = B3Geom<MW, NW, D, NPB>;
    constexpr int C = G::C, BN = G::BN, W = G::W, NCH = G::NCH, RH = G::RH, NT = G::NT, HS = G::HS;
    constexpr int NHALF = MW / RH;             // operand units (row halves) per phase
    constexpr int NU = 7 * NHALF;              // ... per chunk
    constexpr int NSTEP = NU * NW;             // MFMA steps per chunk: (phase, row half, column block)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = wave * (32 * NW);
    const int Lin = p.Lin;

    const int my_tiles = int(blockIdx.x) < ntiles ? (ntiles - 1 - int(blockIdx.x)) / int(gridDim.x) + 1 : 0;
    if (my_tiles == 0) return;
    const int first_b = int(blockIdx.x) / tiles_per_clip, first_t = int(blockIdx.x) - first_b * tiles_per_clip;

    B3_RSTAMP(14);
    if (NCH <= 4) B3_CSTAMP(5);      // (slots 5 .. 8 are chunk stamps only for C = 128: 8 chunks)
    float *bias_s = reinterpret_cast<float *>(lds + G::OFF_BIAS);
    for (int i = tid; i < 2 * C; i += 256) bias_s[i] = i < C ? (b1 ? b1[i] : 0.f) : (b2 ? b2[i - C] : 0.f);

    // ---- weight DMA: a linear stream of phases (tile after tile: chunk-major, tap-minor = the image's own order) ----
    // the slot of phase (chunk, j) is fixed by j alone (groups [0,1] [2,3] [4,5] [6] alternate between the two slot sets):
    // j -> {0, 1, 2, 3, 0, 1, 2}.  The DMA runs TWO groups ahead of the MFMAs.
    const char *zpage = reinterpret_cast<const char *>(g_b3_zero_page) + lane * 16;
    int w_tile = 0, w_chunk = 0;                       // DMA cursor: tile index (of this workgroup) and chunk
    auto dma_phase = [&](int jj) {                     // one phase of the cursor's chunk -> its slot
        const int slot = jj < 4 ? jj : jj - 4;
        const bool live = w_tile < my_tiles;
        const char *src0 = wt1 + (size_t(w_chunk) * G::J + jj) * G::WSLOT_B;
#pragma unroll
        for (int r = 0; r < G::RW; ++r) {
            const int n = (wave + 4 * r) % G::NPW;
            const char *src = live ? src0 + n * 1024 + lane * 16 : zpage;
            b3_glds_b128(src, lds + G::OFF_W + slot * G::WSLOT_B + n * 1024);
        }
    };
    auto dma_advance_chunk = [&]() {
        if (++w_chunk == NCH) w_chunk = 0, ++w_tile;
    };

    // ---- input stream: chunk after chunk, tile after tile ----------------------------------------------------------------
    int i_tile = 0, i_chunk = 0, i_b = first_b, i_t = first_t;     // the next chunk to load
    float st[NT][8];
    int st_t[NT];          // plane row index of the task (time step inside the tile row), -1: no task
    auto input_load = [&]() {
        const bool live = i_tile < my_tiles;
        const int in0 = i_t * BN - G::P;
        const char *xc = reinterpret_cast<const char *>(live ? x + (size_t(i_b) * C + i_chunk * 16) * Lin : x);   // uniform; past
                                                                                       // the last tile: any valid rows (zeroed below)
        unsigned lin4 = unsigned(Lin) * 4u;
        asm volatile("" : "+s"(lin4));       // opaque: keeps the 8 row offsets from being hoisted out of the chunk loop
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int u = tid + 256 * n;
            const int uh = u >= W ? 1 : 0;
            const int t = u - uh * W;
            const int pos = in0 + t;
            const bool task = u < 2 * W;
            const bool ok = live && task && pos >= 0 && pos < p.Lvalid;
            const int posc = min(max(pos, 0), Lin - 1);
            const unsigned off = unsigned(8 * uh) * lin4 + unsigned(posc) * 4u;
            st_t[n] = task ? (uh * W + t) : -1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = *reinterpret_cast<const float *>(xc + (off + unsigned(e) * lin4));
                st[n][e] = ok ? v : 0.f;
            }
        }
        if (++i_chunk == NCH) {
            i_chunk = 0;
            ++i_tile;
            i_b += step_b;
            i_t += step_t;
            if (i_t >= tiles_per_clip) i_t -= tiles_per_clip, ++i_b;
        }
    };
    auto input_store_task = [&](int n, int buf) {      // split + write ONE staged task into plane buffer `buf`
        char *pb = lds + (buf % NPB) * G::PLANE_B;
        b3x8 h, m, l;
        b3_split8(st[n], h, m, l);
        if (st_t[n] >= 0) {
            *reinterpret_cast<b3x8 *>(pb + (0 * 2 * W + st_t[n]) * 16) = h;
            *reinterpret_cast<b3x8 *>(pb + (1 * 2 * W + st_t[n]) * 16) = m;
            *reinterpret_cast<b3x8 *>(pb + (2 * 2 * W + st_t[n]) * 16) = l;
        }
    };

    // consumer-side lane offsets (bytes)
    const int aLane = (lh * C + li) * 16;                      // + (plane * 2 C + 32 i) * 16
    const int bLane = (lh * W + n0 + li) * 16;                 // + (plane * 2 W + 32 k + j D) * 16
    auto load_a = [&](b3x8 (&a)[3][RH], int jj, int half) {
        const char *wslot = lds + G::OFF_W + (jj < 4 ? jj : jj - 4) * G::WSLOT_B + aLane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < RH; ++i)
                a[pl][i] = *reinterpret_cast<const b3x8 *>(wslot + (pl * 2 * C + 32 * (half * RH + i)) * 16);
    };
    auto load_b = [&](b3x8 (&bf)[3], int buf, int jj, int kk) {
        const char *pbuf = lds + (buf % NPB) * G::PLANE_B + bLane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bf[pl] = *reinterpret_cast<const b3x8 *>(pbuf + (pl * 2 * W + 32 * kk + jj * D) * 16);
    };

    // ---- prologue: first chunk's planes, the first two groups' weights (phases 0 .. 3) ---------------------------------
    input_load();
    dma_phase(0);
    dma_phase(1);
    dma_phase(2);
    dma_phase(3);
#pragma unroll
    for (int n = 0; n < NT; ++n) input_store_task(n, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // operand registers: A of the current / next (phase, row half) unit, B of the current / next step
    b3x8 fa[2][3][RH], fb[2][3];
    load_a(fa[0], 0, 0);
    load_b(fb[0], 0, 0, 0);

    f32x16 acc[MW][NW];
    int cb = first_b, ct = first_t;
    for (int k = 0; k < my_tiles; ++k) {
        const int b = cb, t0 = ct * BN;
        cb += step_b;
        ct += step_t;
        if (ct >= tiles_per_clip) ct -= tiles_per_clip, ++cb;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int kk = 0; kk < NW; ++kk)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][kk][r] = 0.f;

        B3_TSTAMP(0);
        if (prio) __builtin_amdgcn_s_setprio(1);      // GEMM1 outranks the partner workgroup's tail on the shared SIMD (+1..3 %; knob b3_dbg = 2: off)
        // ---- GEMM1: two chunks per iteration (the unit count of a chunk may be odd: the A register sets swap roles) --------
        for (int c2 = 0; c2 < NCH; c2 += 2) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                // chunk c2 + cc reads plane buffer cc (NCH is even: the parity restarts with every tile)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int j = u / NHALF, half = u % NHALF;
                    const int ua = (cc * NU + u) & 1;                       // A register set of this unit
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const int step = u * NW + kk;
                        const int sb = (cc * NSTEP + step) & 1;             // B register set of this step
                        const bool last_of_unit = kk == NW - 1;
                        const bool last_of_phase = last_of_unit && half == NHALF - 1;
                        const bool group_end = last_of_phase && (j == 1 || j == 3 || j == 5 || j == 6);
                        // the next chunk's loads are issued behind the barrier that ends phase 1 (in flight over phases 2, 3),
                        // split + written one task per step from phase 4 on, all before the chunk's last barrier
                        const int first_store = 4 * NHALF * NW;
                        if (group_end) {
                            // EARLY barrier: this group's last operands are in registers already, so its slots (and, at the
                            // chunk's end, nothing of the plane buffer) are needed no more.  The next group's weights -- and
                            // the next chunk's planes -- have landed once every wave has waited for its own part.
                            // COUNTED wait (round 4).  vmcnt counts loads and LDS-DMA together, in issue order.  What this barrier
                            // needs is the weight DMA issued at the previous one; the next chunk's 8 NT input loads (HBM latency)
                            // were issued behind phases 4 / 5 at the j == 1 barrier and are only needed at the chunk's end: at the
                            // j == 3 barrier they may stay in flight (the vmcnt(0) of round 3 drained them two phases after their
                            // issue, every chunk).  The j == 5 barrier needs phase 6 (younger than the loads): vmcnt(0).
                            if (j == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * NT) : "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __syncthreads();
                            if (j == 1) {
                                dma_phase(4);
                                dma_phase(5);
                                asm volatile("" ::: "memory");          // the loads stay BEHIND the two DMA phases in issue order
                                __builtin_amdgcn_sched_barrier(0);
                                input_load();
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            else if (j == 3) { dma_phase(6); dma_advance_chunk(); }
                            else if (j == 5) { dma_phase(0); dma_phase(1); }
                            else { dma_phase(2); dma_phase(3); }
                        }
                        f32x16 part[RH];
#pragma unroll
                        for (int i = 0; i < RH; ++i) part[i] = acc[half * RH + i][kk];
                        const bool chunk_end = group_end && j == 6;
                        __builtin_amdgcn_sched_barrier(0);
                        b3_products<RH, 0, 3>(part, fa[ua], fb[sb]);
                        __builtin_amdgcn_sched_barrier(0);
                        // operands of the NEXT step, issued in the middle of this step's MFMAs: they have landed when the
                        // next step starts, whatever counter value the compiler waits for there.  (One plane buffer: the
                        // next chunk's planes only exist behind the second barrier below.)
                        int nu = u, nk = kk + 1, ncc = cc;
                        if (nk == NW) nk = 0, ++nu;
                        if (nu == NU) nu = 0, ncc ^= 1;
                        const int nj = nu / NHALF, nhalf = nu % NHALF;
                        if (!(NPB == 1 && chunk_end)) {
                            if (nk == 0) load_a(fa[ua ^ 1], nj, nhalf);
                            load_b(fb[sb ^ 1], ncc, nj, nk);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        b3_products<RH, 3, 6>(part, fa[ua], fb[sb]);
#pragma unroll
                        for (int i = 0; i < RH; ++i) acc[half * RH + i][kk] = part[i];
                        if (NPB == 2 && step >= first_store && step < first_store + NT) {
                            input_store_task(step - first_store, cc ^ 1);   // the buffer the previous chunk read: free since its last barrier
                            // thread the split's vector instructions between the MFMAs (3 RH MFMAs, ~48 VALU + 3 LDS writes)
#pragma unroll
                            for (int g = 0; g < 3 * RH; ++g) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, (48 + 3 * RH - 1) / (3 * RH), 0);
                            }
                        }
                        if (NPB == 1 && chunk_end) {
                            // ONE plane buffer (two workgroups per CU: the partner's MFMAs run meanwhile): every wave is past the
                            // chunk's barrier, i.e. holds the chunk's last operands in registers -- the buffer is free.  Split +
                            // write the next chunk under this step's last MFMAs, barrier, then read the next step's operands.
#pragma unroll
                            for (int n = 0; n < NT; ++n) input_store_task(n, 0);
#pragma unroll
                            for (int g = 0; g < 3 * RH; ++g) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, (48 * NT + 3 * RH - 1) / (3 * RH), 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            __syncthreads();
                            load_a(fa[ua ^ 1], nj, nhalf);
                            load_b(fb[sb ^ 1], ncc, nj, nk);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                B3_TSTAMP(1 + c2 + cc);
            }
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        static_assert(4 * NHALF * NW + NT <= NSTEP - 1, "the chunk's split must finish before its last (early) barrier");

        // ---- tile tail -------------------------------------------------------------------------------------------------
        const char *xb = reinterpret_cast<const char *>(x + size_t(b) * C * Lin);     // uniform bases + 32-bit lane offsets
        char *yb = reinterpret_cast<char *>(y + size_t(b) * C * Lin);
        unsigned linv = unsigned(Lin), w2off = unsigned(aLane);
        asm volatile("" : "+v"(linv), "+v"(w2off));   // opaque per tile: no address of the tail is hoisted over the main loop
        auto load_w2 = [&](b3x8 (&a2)[3][HS], int kb, int pass) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int io = 0; io < HS; ++io)
                    a2[pl][io] = *reinterpret_cast<const b3x8 *>(wt2 + (w2off + unsigned((kb * 6 * C + pl * 2 * C + 32 * (pass * HS + io)) * 16)));
        };
        b3x8 a2[2][3][HS];
        load_w2(a2[0], 0, 0);          // travels while the activation runs
        __builtin_amdgcn_sched_barrier(0);
        // hidden activation in registers (bias from LDS); rows of register r of row block i: 32 i + 8 (r / 4) + 4 lh + r % 4
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bq = *reinterpret_cast<const f32x4 *>(bias_s + i * 32 + 8 * g + 4 * lh);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const float v = acc[i][kk][4 * g + s4] + bq[s4];
                        acc[i][kk][4 * g + s4] = v > 0.f ? v : v * p.slope;
                    }
            }
        B3_TSTAMP(9);
        // GEMM2: out = b2 + W2 . h, in MW / HS row passes (register budget: acc + out + W2 fragments); k-block kb = hidden
        // channels 16 kb .. 16 kb + 15 = accumulator registers 8 (kb % 2) .. + 7 of row block kb / 2.  The residual is added
        // in the epilogue (the GEMM1 accumulators are dead by then: room to have a whole pass of x in flight).
#pragma unroll
        for (int pass = 0; pass < MW / HS; ++pass) {
            f32x16 out[HS][NW];
#pragma unroll
            for (int io = 0; io < HS; ++io)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(bias_s + C + (pass * HS + io) * 32 + 8 * g + 4 * lh);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int kk = 0; kk < NW; ++kk) out[io][kk][4 * g + s4] = bq[s4];
                }
            b3x8 hb[2][3];            // split hidden block of the current / next (kb, kk)
            {
                float hv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = acc[0][0][e];
                b3_split8(hv, hb[0][0], hb[0][1], hb[0][2]);
            }
#pragma unroll
            for (int kb = 0; kb < C / 16; ++kb) {
                // weights one k-block ahead (the last block of a pass fetches the next pass's first)
                if (kb + 1 < C / 16) load_w2(a2[(kb + 1) & 1], kb + 1, pass);
                else if (pass + 1 < MW / HS) load_w2(a2[0], 0, pass + 1);
#pragma unroll
                for (int kk = 0; kk < NW; ++kk) {
                    const int sidx = (kb * NW + kk) & 1;
                    // split the NEXT block while this one's MFMAs run
                    {
                        int nkb = kb, nkk = kk + 1;
                        if (nkk == NW) nkk = 0, ++nkb;
                        if (nkb < C / 16) {
                            float hv[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) hv[e] = acc[nkb >> 1][nkk][8 * (nkb & 1) + e];
                            b3_split8(hv, hb[sidx ^ 1][0], hb[sidx ^ 1][1], hb[sidx ^ 1][2]);
                        }
                    }
                    f32x16 part[HS];
#pragma unroll
                    for (int io = 0; io < HS; ++io) part[io] = out[io][kk];
                    b3_products<HS>(part, a2[kb & 1], hb[sidx]);
#pragma unroll
                    for (int io = 0; io < HS; ++io) out[io][kk] = part[io];
#pragma unroll
                    for (int g = 0; g < 6 * HS; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, (48 + 6 * HS - 1) / (6 * HS), 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (pass == 0) B3_TSTAMP(10);
            static_assert((C / 16 * NW) % 2 == 0, "the split register sets must line up across passes");
            // residual + trailing activation + store (accumulator layout: 128-byte row segments per half wave).  Round 4: ALL of the
            // pass's residual loads are issued before the first use (the GEMM1 accumulators are dead: their registers hold the
            // pass's x) -- one exposed memory round trip per pass instead of one per (row block, column block); round 3 loaded 16
            // values, waited, stored, and the next block's loads queued behind those stores (vmcnt counts both in order).
            // (C = 128 / 256 keep one block at a time: a whole pass of x in registers spilled 132 / 204 bytes per lane there.)
            constexpr int XB = MW <= 2 ? HS * NW : 1;      // (row block, column block) pairs whose residual is in flight together (C >= 128: register budget)
            static_assert((HS * NW) % XB == 0, "residual batches");
#pragma unroll
            for (int g0 = 0; g0 < HS * NW; g0 += XB) {
                float xr[XB][16];
#pragma unroll
                for (int gi = 0; gi < XB; ++gi) {
                    const int io = (g0 + gi) / NW, kk = (g0 + gi) % NW;
                    const int colc = min(t0 + n0 + 32 * kk + li, Lin - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * (pass * HS + io) + 8 * (r >> 2) + 4 * lh + (r & 3);
                        xr[gi][r] = *reinterpret_cast<const float *>(xb + (unsigned(row) * linv + unsigned(colc)) * 4u);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int gi = 0; gi < XB; ++gi) {
                    const int io = (g0 + gi) / NW, kk = (g0 + gi) % NW;
                    const int col = t0 + n0 + 32 * kk + li;
                    if (col < Lin) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = 32 * (pass * HS + io) + 8 * (r >> 2) + 4 * lh + (r & 3);
                            float v = out[io][kk][r] + xr[gi][r];
                            if (post_act) v = leaky(v, p.slope);
                            *reinterpret_cast<float *>(yb + (unsigned(row) * linv + unsigned(col)) * 4u) = v;
                        }
                    }
                }
            }
            if (pass == 0) B3_TSTAMP(11);
        }
        B3_TSTAMP(12);
    }
    if (NCH <= 4) B3_CSTAMP(6);
    B3_RSTAMP(15);
}

template <int MW, int NW, int D, int NPB>
static int launch_b3(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
                     float *y, int post_act, hipStream_t st) {
    using G = B3Geom<MW, NW, D, NPB>;
    auto kern = resblock_b3_kernel<MW, NW, D, NPB>;
    static DeviceOnce once;
    int n_cu = 0;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, &n_cu, "resblock_b3")) return rc;
    static_assert(G::LDS_BYTES * (NPB == 1 ? 2 : 1) <= 160 * 1024, "resblock_b3: LDS budget");
    const int tiles_per_clip = ceil_div(p.Lin, G::BN);
    const int64_t ntiles64 = int64_t(tiles_per_clip) * p.B;
    if (ntiles64 > (1 << 30)) return fail(AGX_ERR_BAD_SHAPE, "resblock_b3: too many tiles");
    const int ntiles = int(ntiles64);
    const int want = n_cu * (NPB == 1 ? 2 : 1);            // persistent: two workgroups per CU where the LDS / registers allow
    const int grid = ntiles < want ? ntiles : want;
    // B3 tile images: behind the bf16x3 standard image and the dim0 scale scratch (common.hpp: b3 images)
    const char *wt1 = reinterpret_cast<const char *>(w1 + packed_weight_floats_bf(G::C, G::J, G::C) + G::C);
    const char *wt2 = reinterpret_cast<const char *>(w2 + packed_weight_floats_bf(G::C, 1, G::C) + G::C);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, st, p, tiles_per_clip, ntiles, grid / tiles_per_clip,
                       grid % tiles_per_clip, post_act, tuning().b3_dbg != 2, x, wt1, b1, wt2, b2, y);
    return check_launch("resblock_b3");
}

// shapes the kernel is instantiated for: C in {32, 64, 128, 256}, k = 7, dilation in {1, 3, 9}, bf16x3 descriptors
bool resblock_b3_supported(const ConvPlan &p) {
    if (p.prec != 1 || p.tile_off < 0 || p.Cin != p.Cout || p.s != 1 || p.q != 1 || p.J != 7 || p.G != 1) return false;
    if (p.Lvalid != p.Lin || p.Lt != p.Lin || p.Lin < 1) return false;
    if (p.Cin != 32 && p.Cin != 64 && p.Cin != 128 && p.Cin != 256) return false;
    if (int64_t(p.Cin) * p.Lin * 4 >= (int64_t(1) << 32)) return false;
    return p.d == 1 || p.d == 3 || p.d == 9;
}

const char *resblock_b3_variant(const ConvPlan &p) {
    switch (p.Cin) {
        case 32: return "resblock_b3<1,4,x2>";
        case 64: return "resblock_b3<2,2,x2>";
        case 128: return "resblock_b3<4,1,x2>";
        default: return "resblock_b3<8,1>";
    }
}

int launch_resblock_b3(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
                       float *y, int post_act, hipStream_t st) {
    if (!resblock_b3_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "resblock_b3: unsupported shape");
#define AGX_B3(MW, NW, NPB)                                                                  \
    (p.d == 1 ? launch_b3<MW, NW, 1, NPB>(p, x, w1, b1, w2, b2, y, post_act, st)             \
     : p.d == 3 ? launch_b3<MW, NW, 3, NPB>(p, x, w1, b1, w2, b2, y, post_act, st)           \
                : launch_b3<MW, NW, 9, NPB>(p, x, w1, b1, w2, b2, y, post_act, st))
    // <.., 1>: one plane buffer, <= 256 registers: two workgroups per CU (one covers the other's vector work and barriers);
    // <.., 2>: double-buffered planes, one workgroup per CU with up to 512 registers (tiles too big for two)
    switch (p.Cin) {
        case 32: return AGX_B3(1, 4, 1);
        case 64: return tuning().b3_dbg == 1 ? AGX_B3(2, 4, 2) : AGX_B3(2, 2, 1);
        case 128: return tuning().b3_dbg == 1 ? AGX_B3(4, 2, 2) : AGX_B3(4, 1, 1);
        default: return AGX_B3(8, 1, 2);
    }
#undef AGX_B3
}

}  // namespace agx
