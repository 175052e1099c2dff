// Stride-1 polyphase convolutions on the bf16 matrix pipe with fp32-class accuracy ("bf16x3"), persistent ring form: the
// decoder's resampling layers -- CausalUpsampleConv1d x2 / x4 / x5 / x8 (networks/vae.py:66-89) as their q polyphase 3-tap
// filters on the low-rate signal, the stride-1 CausalConvT1d k = 7 (vae.py:45-64) -- with bias and LeakyReLU fused:
//
//     y[b, m / q, q t + m % q] = leaky( bias[m / q] + sum_{ci, j < J} Wp[ci, j][m] x[b, ci, t + j - P] ),   m < M = q Cout
//
// Same machinery as resblock_b3.hip (see there): the weights come from the layer's B3 tile image ([16-channel group][tap]
// [piece][lane half][M rows][8 bf16]; a phase's BM-row block is 6 contiguous BM x 16-byte pieces) through four one-phase
// LDS slots, the input chunk (16 channels x (BN + J - 1) time steps) is split ONCE per tile into three bf16 planes in LDS,
// operand reads run one MFMA step ahead, the group barrier sits before the group's last step, the DMA runs two groups
// ahead.  What differs: a tile is (clip, time block, ROW block) -- BM = 128 rows x 128 columns on 2 x 2 waves, 64 x 256 on
// 1 x 4 for the 64-row layer -- there is no second GEMM, and the epilogue writes the polyphase rows to their output
// samples (row m -> channel m / q, sample q t + m % q).  Two workgroups per CU (one plane buffer, <= 256 registers).
// A row block re-splits the input its neighbours split as well (x2 .. x16): the tiles of one (clip, time block) are
// consecutive, so those reads hit L2, and the split hides under the partner workgroup's MFMAs.
#include "mfma_tile.hpp"

#ifndef C2B3_GT3
#define C2B3_GT3 1   // conv2d_b3_kernel: weight groups of three taps on the 32- / 64-row tiles
#endif

namespace agx {

typedef __bf16 cb3x8 __attribute__((ext_vector_type(8)));
typedef float f32x2c __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(1024))) float g_cb3_zero_page[256] = {0.f};

__device__ __forceinline__ void cb3_glds_b128(const void *gsrc_lane, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// XP: the input arrives as ready-made bf16 planes (common.hpp: "activation planes", written by a producer's epilogue or by
// planes_split_kernel) and is staged by LDS-DMA into TWO plane buffers -- no register staging, no split, no second barrier.
// S (round 4): input step per output position -- the encoder's strided down-convs (CausalConv1d(K = 2 s + 1, stride s), vae.py:136-139).
// The planes of a chunk are PHASE-SPLIT: position w of the tile's input window sits in row w % S at cell w / S, so tap j of output t
// (w = S t + j) reads row j % S at cell t + j / S -- unit lane stride, conflict-free, an immediate offset per tap as for S = 1.
template <int MW, int NW, int WM, int J_, int Q_, int P_, bool XP = false, int S_ = 1>
struct Cb3Geom {
    static constexpr int WN = 4 / WM;                      // waves along the columns
    static constexpr int BM = 32 * MW * WM, BN = 32 * NW * WN, J = J_, Q = Q_, P = P_, S = S_;
    static constexpr int WP = BN - 1 + (J + S - 1) / S;    // cells per phase row (S = 1: tile + halo = BN + J - 1)
    static constexpr int W = S * WP;                       // cells per (piece, lane half): S phase rows
    static constexpr int PLANE_B = 6 * W * 16;             // one chunk: [piece 3][lane half 2][phase S][WP][8 bf16]
    static constexpr int WSLOT_B = 96 * BM;                // one phase of weights: [piece 3][lane half 2][BM][8 bf16]
    static constexpr int NPW = WSLOT_B / 1024;             // 1 KiB DMA pieces per phase (BM / 64 per (piece, half))
    static constexpr int RW = (NPW + 3) / 4;
    static constexpr int NT = (2 * W + 255) / 256;         // split tasks (time step x 8 channels) per thread and chunk
    static constexpr int NGRP = (J + 1) / 2;               // weight groups per chunk: [0,1] [2,3] .. (the last one single); an odd count
                                                           // swaps the two slot sets from chunk to chunk (running group parity)
    static constexpr int NIR = (W + 63) / 64;              // XP: 1 KiB DMA instructions per plane row (the last one partial)
    static constexpr int RX = (6 * NIR + 3) / 4;           // ... per wave and chunk
    static constexpr int OFF_W = (XP ? 2 : 1) * PLANE_B;
    static constexpr size_t LDS_BYTES = size_t(OFF_W) + 4 * WSLOT_B;
    static_assert(J % 2 == 1, "taps come in pairs plus one");
    static_assert(!(XP && S > 1), "plane input: stride-1 layers");
};

template <int N>
__device__ __forceinline__ void cb3_products(f32x16 (&acc)[N], const cb3x8 (&a)[3][N], const cb3x8 (&b)[3], int t0, int t1) {
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};      // m.m  h.l  l.h  h.m  m.h  h.h
#pragma unroll
    for (int t = 0; t < 6; ++t)
        if (t >= t0 && t < t1) {
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]], acc[i], 0, 0, 0);
        }
}

// four values of one accumulator row group -> three bf16 pieces each (8 bytes per piece: channels 4 lh .. 4 lh + 3 of a cell)
typedef __bf16 cb3x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cb3_split4(const float (&v)[4], cb3x4 &h, cb3x4 &m, cb3x4 &l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = (__bf16)v[e];
        const float r1 = v[e] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        h[e] = hh;
        m[e] = mm;
        l[e] = (__bf16)(r1 - (float)mm);
    }
}

// XP: x = activation planes (bf16 [B][Cin / 8][3][Lin][8]); YP (Q = 1): y = activation planes of the output instead of fp32
template <int MW, int NW, int WM, int J_, int Q_, int P_, bool XP = false, bool YP = false, int S_ = 1>
__global__ __launch_bounds__(256, 2) void conv_b3_kernel(ConvPlan p, int tb_per_clip, int mb_count, int ntiles,
                                                         const float *__restrict__ x, const char *__restrict__ wt,
                                                         const float *__restrict__ bias, float *__restrict__ y) {
    using G = Cb3Geom<MW, NW, WM, J_, Q_, P_, XP, S_>;
    static_assert(!YP || Q_ == 1, "plane output: one output phase");
    constexpr int BM = G::BM, BN = G::BN, W = G::W, J = G::J, Q = G::Q, NT = G::NT, S = G::S, WP = G::WP;
    constexpr int NSTEP = J * NW;                         // MFMA steps per chunk: (tap, column block); all MW row blocks per step
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int r0w = wm * (32 * MW), n0 = wn * (32 * NW);  // this wave's rows / columns inside the tile
    const int Lin = p.Lin, M = p.M, Lout = p.Lout;
    const int nch = p.Cin / 16;                           // chunks per tile (even: checked by the launcher)

    const int my_tiles = int(blockIdx.x) < ntiles ? (ntiles - 1 - int(blockIdx.x)) / int(gridDim.x) + 1 : 0;
    if (my_tiles == 0) return;
    // tile id -> (clip b, time block tb, row block mb): row blocks of one (clip, time block) are consecutive ids
    auto decode = [&](int tile, int &b, int &tb, int &mb) {
        mb = tile % mb_count;
        const int r = tile / mb_count;
        tb = r % tb_per_clip;
        b = r / tb_per_clip;
    };

    // ---- weight DMA: one stream of groups (tile after tile, chunk after chunk); every group-end barrier issues one ----
    const char *zpage = reinterpret_cast<const char *>(g_cb3_zero_page) + lane * 16;
    int w_k = 0, w_g = 0;                                 // DMA cursor: the workgroup's w_k-th tile, group w_g of that tile
    int w_m0 = 0;
    {
        int b_, tb_, mb_;
        decode(int(blockIdx.x), b_, tb_, mb_);
        w_m0 = mb_ * BM;
    }
    auto dma_next_group = [&]() {
        const bool live = w_k < my_tiles;
        const int chunk = w_g / G::NGRP, which = w_g % G::NGRP;
        const int set2 = (w_g & 1) * 2;          // running group parity (a tile has an even number of groups: nch is even)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = 2 * which + i;
            const bool valid = live && j < J;
            // phase (chunk, j): [(chunk J + j) 6 + plh][M][16 B]; this tile's rows start at w_m0
            const char *src0 = wt + (size_t(chunk) * J + (valid ? j : 0)) * 6 * size_t(M) * 16 + size_t(w_m0) * 16;
            char *dst0 = lds + G::OFF_W + (set2 + i) * G::WSLOT_B;
#pragma unroll
            for (int r = 0; r < G::RW; ++r) {
                const int n = (wave + 4 * r) % G::NPW;                       // piece n: (piece, half) = n / (BM / 64), part n % (BM / 64)
                const int plh = n / (BM / 64), part = n % (BM / 64);
                const char *src = valid ? src0 + (size_t(plh) * M + 64 * part) * 16 + lane * 16 : zpage;
                cb3_glds_b128(src, dst0 + n * 1024);
            }
        }
        if (++w_g == nch * G::NGRP) {
            w_g = 0;
            ++w_k;
            int b_, tb_, mb_;
            decode(min(int(blockIdx.x) + w_k * int(gridDim.x), ntiles - 1), b_, tb_, mb_);
            w_m0 = mb_ * BM;
        }
    };

    // ---- input stream ----
    // Two chunks of input are in flight in registers (two sets, addressed statically by the chunk's parity): with three taps a
    // chunk is 72 MFMAs per wave -- 2.3 k cycles of matrix pipe, 4.6 k next to the partner workgroup --, less than a global load
    // takes under load, and a request issued ONE chunk ahead was waited for at every chunk boundary.
    int i_k = 0, i_chunk = 0;                             // the next chunk to load: tile index, chunk
    // XP: the chunk's six plane rows ((piece, channel half) x W cells of 16 bytes) by LDS-DMA, instruction q = wave + 4 r:
    // row q / NIR, cells 64 (q % NIR) .. + 63; cells outside the signal come from the page of zeros, lanes past the row's
    // end are switched off (the only masked DMA: one instruction per row)
    auto planes_dma = [&](int buf) {
        const bool live = i_k < my_tiles;
        int b_, tb_, mb_;
        decode(min(int(blockIdx.x) + i_k * int(gridDim.x), ntiles - 1), b_, tb_, mb_);
        const int in0 = tb_ * BN - G::P;
        const char *xc = reinterpret_cast<const char *>(x) + (size_t(b_) * (p.Cin / 8) + size_t(i_chunk) * 2) * 3 * size_t(Lin) * 16;
#pragma unroll
        for (int r = 0; r < G::RX; ++r) {
            const int q = wave + 4 * r;
            const int row = q / G::NIR, part = q - row * G::NIR;       // row = piece * 2 + half
            const int cell = part * 64 + lane;
            const int pos = in0 + cell;
            const bool ok = live && pos >= 0 && pos < p.Lvalid;
            const char *src = ok ? xc + (size_t((row & 1) * 3 + (row >> 1)) * Lin + size_t(pos)) * 16 : zpage;
            // exactly RX instructions per wave (the counted waits rely on it): slots past the last row re-write this wave's last
            // real piece (same bytes, same wave: in order)
            const bool real = q < 6 * G::NIR;
            const int rowd = real ? row : 5, partd = real ? part : G::NIR - 1;
            const int celld = partd * 64 + lane;
            const int posd = in0 + celld;
            const char *srcd = real ? src : ((live && posd >= 0 && posd < p.Lvalid) ? xc + (size_t((rowd & 1) * 3 + (rowd >> 1)) * Lin + size_t(posd)) * 16 : zpage);
            if (celld < W) cb3_glds_b128(srcd, lds + buf * G::PLANE_B + (rowd * W + partd * 64) * 16);
        }
        if (++i_chunk == nch) i_chunk = 0, ++i_k;
    };
    float st2[XP ? 1 : 2][XP ? 1 : NT][8];
    int st_t2[XP ? 1 : 2][XP ? 1 : NT];
    auto input_load = [&](float (&st)[XP ? 1 : NT][8], int (&st_t)[XP ? 1 : NT]) {
        if (XP) return;
        const bool live = i_k < my_tiles;
        int b_, tb_, mb_;
        decode(min(int(blockIdx.x) + i_k * int(gridDim.x), ntiles - 1), b_, tb_, mb_);
        const int in0 = tb_ * BN * S - G::P;
        const char *xc = reinterpret_cast<const char *>(x + (size_t(b_) * p.Cin + i_chunk * 16) * Lin);
        unsigned lin4 = unsigned(Lin) * 4u;
        asm volatile("" : "+s"(lin4));
#pragma unroll
        for (int n = 0; n < (XP ? 1 : NT); ++n) {
            const int u = tid + 256 * n;
            const int uh = u >= W ? 1 : 0;
            const int t = u - uh * W;
            const int pos = in0 + t;
            const bool task = u < 2 * W;
            const bool ok = live && task && pos >= 0 && pos < p.Lvalid;
            const int posc = min(max(pos, 0), Lin - 1);
            const unsigned off = unsigned(8 * uh) * lin4 + unsigned(posc) * 4u;
            st_t[n] = task ? (uh * W + (S == 1 ? t : (t % S) * WP + t / S)) : -1;      // phase-split cell of window position t
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = *reinterpret_cast<const float *>(xc + (off + unsigned(e) * lin4));
                st[n][e] = ok ? v : 0.f;
            }
        }
        if (++i_chunk == nch) i_chunk = 0, ++i_k;
    };
    auto input_store_all = [&](const float (&st)[XP ? 1 : NT][8], const int (&st_t)[XP ? 1 : NT]) {
        if (XP) return;
#pragma unroll
        for (int n = 0; n < (XP ? 1 : NT); ++n) {
            cb3x8 h, m, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 hh = (__bf16)st[n][e];
                const float r1 = st[n][e] - (float)hh;
                const __bf16 mm = (__bf16)r1;
                h[e] = hh;
                m[e] = mm;
                l[e] = (__bf16)(r1 - (float)mm);
            }
            if (st_t[n] >= 0) {
                *reinterpret_cast<cb3x8 *>(lds + (0 * 2 * W + st_t[n]) * 16) = h;
                *reinterpret_cast<cb3x8 *>(lds + (1 * 2 * W + st_t[n]) * 16) = m;
                *reinterpret_cast<cb3x8 *>(lds + (2 * 2 * W + st_t[n]) * 16) = l;
            }
        }
    };

    const int aLane = (lh * BM + r0w + li) * 16;          // + (piece * 2 BM + 32 i) * 16
    const int bLane = (lh * W + n0 + li) * 16;            // + (piece * 2 W + 32 k + j) * 16
    auto load_a = [&](cb3x8 (&a)[3][MW], int j, int gpar) {      // gpar: parity of the group's running index (cc NGRP + j / 2)
        const char *ws = lds + G::OFF_W + ((gpar & 1) * 2 + (j & 1)) * G::WSLOT_B + aLane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < MW; ++i) a[pl][i] = *reinterpret_cast<const cb3x8 *>(ws + (pl * 2 * BM + 32 * i) * 16);
    };
    auto load_b = [&](cb3x8 (&bf)[3], int j, int kk, int buf = 0) {     // buf: XP only (two plane buffers, chunk parity)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bf[pl] = *reinterpret_cast<const cb3x8 *>(lds + (XP ? buf * G::PLANE_B : 0) + bLane +
                                                      (pl * 2 * W + (j % S) * WP + 32 * kk + j / S) * 16);
    };

    // ---- prologue: chunk 0 -> planes, chunks 1 and 2 on their way (XP: chunk 0 by DMA, chunk 1 requested) ----
    if (XP) planes_dma(0);
    input_load(st2[0], st_t2[0]);
    dma_next_group();
    dma_next_group();
    input_store_all(st2[0], st_t2[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (XP) planes_dma(1);
    input_load(st2[XP ? 0 : 1], st_t2[XP ? 0 : 1]);
    input_load(st2[0], st_t2[0]);

    cb3x8 fa[2][3][MW], fb[2][3];
    for (int k = 0; k < my_tiles; ++k) {
        int b, tb, mb;
        decode(int(blockIdx.x) + k * int(gridDim.x), b, tb, mb);
        const int t0 = tb * BN, m0 = mb * BM;
        f32x16 acc[MW][NW];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int kk = 0; kk < NW; ++kk)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][kk][r] = 0.f;
        load_a(fa[0], 0, 0);
        load_b(fb[0], 0, 0);

        for (int c2 = 0; c2 < nch; c2 += 2) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int ua = (cc * J + j) & 1;
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const int step = j * NW + kk;
                        const int sb = (cc * NSTEP + step) & 1;
                        const bool last_of_phase = kk == NW - 1;
                        const bool group_end = last_of_phase && ((j & 1) == 1 || j == J - 1);
                        const bool chunk_end = last_of_phase && j == J - 1;
                        if (group_end) {      // early barrier: this group's last operands are in registers
                            // COUNTED wait (round 4; vmcnt counts loads and LDS-DMA in issue order).  A group-end barrier needs the
                            // weight group issued at the previous one.  Behind the chunk's LAST barrier the next input went out as
                            // well -- the plane DMA of the chunk after next (XP: RX instructions per wave) or the register loads
                            // (8 NT) --, needed a whole chunk later: at the chunk's FIRST group end they stay in flight (round 3
                            // drained them with vmcnt(0) two taps after their issue, every chunk).  Later group ends of the chunk
                            // wait for everything (their weights are younger than that input).
                            if (j == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XP ? G::RX : 8 * NT) : "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __syncthreads();
                            dma_next_group();
                            // XP, the chunk's LAST group: every wave holds the chunk's last operands, so its plane buffer is free --
                            // the chunk after next goes there; the NEXT chunk's planes (requested a chunk ago) have landed behind
                            // this barrier, so the operand prefetch simply runs on across the chunk boundary
                            if (XP && j == J - 1) {
                                asm volatile("" ::: "memory");
                                __builtin_amdgcn_sched_barrier(0);
                                planes_dma(cc);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        f32x16 part[MW];
#pragma unroll
                        for (int i = 0; i < MW; ++i) part[i] = acc[i][kk];
                        __builtin_amdgcn_sched_barrier(0);
                        cb3_products<MW>(part, fa[ua], fb[sb], 0, 3);
                        __builtin_amdgcn_sched_barrier(0);
                        int nj = j, nk = kk + 1;
                        if (nk == NW) nk = 0, ++nj;
                        if (nj == J) nj = 0;
                        const int ngp0 = (cc ^ 1) * G::NGRP;      // group parity base of the NEXT chunk (a tile starts at 0: nch is even)
                        if (!chunk_end) {      // (the next chunk's planes only exist behind the second barrier below)
                            if (nk == 0) load_a(fa[ua ^ 1], nj, cc * G::NGRP + (nj >> 1));
                            load_b(fb[sb ^ 1], nj, nk, cc);
                        } else if (XP && !(c2 + 2 >= nch && cc == 1)) {      // (a tile's first operands are read at its top)
                            load_a(fa[ua ^ 1], 0, ngp0);
                            load_b(fb[sb ^ 1], 0, 0, cc ^ 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        cb3_products<MW>(part, fa[ua], fb[sb], 3, 6);
#pragma unroll
                        for (int i = 0; i < MW; ++i) acc[i][kk] = part[i];
                        if (chunk_end && !XP) {
                            // one plane buffer: every wave is past the chunk's barrier (holds its last operands in registers);
                            // split + write the next chunk (register set of its parity: nch is even, so the parity of a chunk
                            // within its tile is its parity in the stream), barrier, read the next step's operands
                            input_store_all(st2[cc ^ 1], st_t2[cc ^ 1]);
                            __builtin_amdgcn_sched_barrier(0);
                            __syncthreads();
                            load_a(fa[ua ^ 1], 0, ngp0);
                            load_b(fb[sb ^ 1], 0, 0);
                            input_load(st2[cc ^ 1], st_t2[cc ^ 1]);   // the chunk THREE ahead: two chunks of flight time
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }

        // ---- epilogue: bias, LeakyReLU, polyphase store.  Register r of row block i: row m0 + r0w + 32 i + 8 (r / 4) + 4 lh + r % 4 ----
        const bool pre = (p.epilogue & AGX_EPI_LEAKY_PRE) != 0;
        float *yb = y + size_t(b) * p.Cout * Lout;
        // the tile's bias values in one batch, before the first store (a load between two stores makes the second wait for the first:
        // vmcnt counts both in order -- see conv2d_b3_kernel)
        float bq[MW][16];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bq[i][r] = bias ? bias[(m0 + r0w + 32 * i + 8 * (r >> 2) + 4 * lh + (r & 3)) / Q] : 0.f;
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int kk = 0; kk < NW; ++kk) {
                const int t = t0 + n0 + 32 * kk + li;
                if (t < p.Lt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int mrow = m0 + r0w + 32 * i + 8 * g + 4 * lh;    // + s4; a multiple of 4
                        if (YP) {                 // activation planes for the next bf16x3 layer: channels mrow .. mrow + 3 = half of a cell
                            float v4[4];
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                const float a = acc[i][kk][4 * g + s4] + bq[i][4 * g + s4];
                                v4[s4] = pre ? leaky(a, p.slope) : a;
                            }
                            cb3x4 ph, pm, pl;
                            cb3_split4(v4, ph, pm, pl);
                            char *cell = reinterpret_cast<char *>(y) + ((size_t(b) * (p.Cout / 8) + size_t(mrow >> 3)) * 3 * size_t(Lout) + size_t(t)) * 16 +
                                         (mrow & 4) * 2;
                            *reinterpret_cast<cb3x4 *>(cell) = ph;
                            *reinterpret_cast<cb3x4 *>(cell + size_t(Lout) * 16) = pm;
                            *reinterpret_cast<cb3x4 *>(cell + size_t(Lout) * 32) = pl;
                        } else if (Q % 4 == 0) {         // the 4 rows are 4 consecutive output samples of one channel: one 16-byte store
                            const int co = mrow / Q, ph = mrow % Q;
                            const float bv = bq[i][4 * g];
                            f32x4 v;
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                float a = acc[i][kk][4 * g + s4] + bv;
                                v[s4] = pre ? leaky(a, p.slope) : a;
                            }
                            *reinterpret_cast<f32x4 *>(yb + size_t(co) * Lout + size_t(Q) * t + ph) = v;
                        } else if (Q == 2) {      // rows (0,1) and (2,3): two channels x two phases -> two 8-byte stores
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const int co = mrow / 2 + h2;
                                const float bv = bq[i][4 * g + 2 * h2];
                                f32x2c v;
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    float a = acc[i][kk][4 * g + 2 * h2 + e] + bv;
                                    v[e] = pre ? leaky(a, p.slope) : a;
                                }
                                *reinterpret_cast<f32x2c *>(yb + size_t(co) * Lout + size_t(2) * t) = v;
                            }
                        } else {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                const int m = mrow + s4, co = m / Q, ph = m % Q;
                                float a = acc[i][kk][4 * g + s4] + bq[i][4 * g + s4];
                                yb[size_t(co) * Lout + size_t(Q) * t + ph] = pre ? leaky(a, p.slope) : a;
                            }
                        }
                    }
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Conv2d 3 x 3, stride 1, "same" padding (the STFT / waveform discriminators' stride-1 layers and their backward-data op,
// discriminator.py:101-114, 150-167) on the same machinery.  A tile is R output rows x WF = 2^SL columns (the STFT maps are powers
// of two wide: no padded columns), flattened: output position n = r WF + f; the input planes hold the (R + 2) x (WF + 2) patch
// with row pitch SWP = WF + 2, so tap (dh, dw) of output position n is the plane position (n + 2 r) + dh SWP + dw -- a per-lane
// base (n + 2 r, fixed for the whole tile loop) plus an immediate offset, exactly the 1-D kernel's "j".  Nine taps = five weight
// groups per chunk: the slot set of a group is the parity of the tile's running group
// count (the 1-D kernel's geometries all have an even number of groups per chunk).  Epilogue: bias, LeakyReLU (forward);
// the arriving gradient added and the LeakyReLU-gradient mask applied (backward-data), as conv_p.hip's Conv2d epilogue.
template <int MW, int NW, int WM, int SL, int KH_ = 3, int KW_ = 3>
struct C2b3Geom {
    static constexpr int WN = 4 / WM, KH = KH_, KW = KW_, J = KH_ * KW_;
    static constexpr int BM = 32 * MW * WM, BN = 32 * NW * WN, WF = 1 << SL, SWP = WF + (KW - 1), R = BN >> SL;
    static constexpr int W = (R + KH - 1) * SWP;               // plane positions
    static constexpr int PLANE_B = 6 * W * 16;
    static constexpr int WSLOT_B = 96 * BM;
    static constexpr int NPW = WSLOT_B / 1024;
    static constexpr int RW = (NPW + 3) / 4;
    static constexpr int NT = (2 * W + 255) / 256;
    static constexpr int OFF_W = (PLANE_B + 1023) / 1024 * 1024;
    // taps per weight group (= per workgroup barrier): 2; 3 for the 32- / 64-row tiles, whose groups of two taps are only 0.6 / 1.1 k
    // MFMA cycles apart -- their waves spent 32-40 % of the time in barriers and waits (PMC), the 128-row tiles 18-27 %
    static constexpr int GT = (C2B3_GT3 && BM <= 64 && J >= 6 && OFF_W + 6 * WSLOT_B <= 80 * 1024) ? 3 : 2;
    static constexpr int NGRP = (J + GT - 1) / GT;
    static constexpr size_t LDS_BYTES = size_t(OFF_W) + 2 * GT * WSLOT_B;
    static_assert(R >= 1 && WF >= 8, "tile rows / columns");
    __host__ __device__ static constexpr int tap(int j) { return (j / KW) * SWP + (j % KW); }
};

// (KH, KW, Q, QH): (3, 3, 1, 1) the 3 x 3 layers and their backward-data; the backward-data of the strided layers is a stride-1
// conv over dy with the ceil(k / s) taps per axis whose M = Cin sh sw rows carry the output phases (conv2d.hip:
// lower_conv2d_bwd_data): (2, 2, 2, 2) for the 4 x 4 stride-(2, 2) layers, (3, 2, 2, 1) for the 3 x 4 stride-(1, 2) ones -- the tile
// walks the BASE grid Tt x Lt, row m = (ci QH + a) Q + c of base position (t', f') goes to output (ci, QH t' + a - oshift_h,
// Q f' + c - oshift).
// probe build (-DC2B3_STAMPS, tools/c2b3_stamps.py): per wave, the cycles of conv2d_b3_kernel by segment, summed over the kernel
#ifdef C2B3_STAMPS
__device__ unsigned long long g_c2b3_stamps[1 << 16];
#define C2S_START() do { __builtin_amdgcn_sched_barrier(0); c2s_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define C2S(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); c2s[k] += tn_ - c2s_prev; c2s_prev = tn_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define C2S_WRITE() do { if (lane == 0 && (int(blockIdx.x) * 4 + wave) * 8 + 8 <= (1 << 16)) for (int k_ = 0; k_ < 8; ++k_) g_c2b3_stamps[(int(blockIdx.x) * 4 + wave) * 8 + k_] = c2s[k_]; } while (0)
#else
#define C2S_START() ((void)0)
#define C2S(k) ((void)0)
#define C2S_WRITE() ((void)0)
#endif

template <int MW, int NW, int WM, int SL, int KH, int KW, int Q, int QH>
__global__ __launch_bounds__(256, 2) void conv2d_b3_kernel(ConvPlan p, int cb_count, int rb_count, int mb_count, int ntiles,
                                                           const float *__restrict__ x, const char *__restrict__ wt,
                                                           const float *__restrict__ bias, const float *__restrict__ add,
                                                           const float *__restrict__ mask, float *__restrict__ y) {
    using G = C2b3Geom<MW, NW, WM, SL, KH, KW>;
    constexpr int BM = G::BM, BN = G::BN, W = G::W, J = G::J, NT = G::NT, SWP = G::SWP;
    constexpr int NSTEP = J * NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int r0w = wm * (32 * MW), n0 = wn * (32 * NW);
    const int M = p.M, Hin = p.Tin, Win = p.Lin, Hout = p.Tout, Wout = p.Lout;
    const int nch = p.Cin / 16;

    const int my_tiles = int(blockIdx.x) < ntiles ? (ntiles - 1 - int(blockIdx.x)) / int(gridDim.x) + 1 : 0;
    if (my_tiles == 0) return;
    // tile id -> (clip b, row block rb, column block cb, output-channel block mb): the channel blocks of one patch are consecutive
    auto decode = [&](int tile, int &b, int &rb, int &cb, int &mb) {
        mb = tile % mb_count;
        int r = tile / mb_count;
        cb = r % cb_count;
        r /= cb_count;
        rb = r % rb_count;
        b = r / rb_count;
    };

    // ---- weight DMA (as conv_b3_kernel; the slot set is the parity of the tile's running group count) ----
    const char *zpage = reinterpret_cast<const char *>(g_cb3_zero_page) + lane * 16;
    int w_k = 0, w_g = 0, w_m0 = 0;
    {
        int b_, rb_, cb_, mb_;
        decode(int(blockIdx.x), b_, rb_, cb_, mb_);
        w_m0 = mb_ * BM;
    }
    auto dma_next_group = [&]() {
        const bool live = w_k < my_tiles;
        const int chunk = w_g / G::NGRP, which = w_g % G::NGRP;
        const int set2 = (w_g & 1) * G::GT;
#pragma unroll
        for (int i = 0; i < G::GT; ++i) {
            const int j = G::GT * which + i;
            const bool valid = live && j < J;
            const char *src0 = wt + (size_t(chunk) * J + (valid ? j : 0)) * 6 * size_t(M) * 16 + size_t(w_m0) * 16;
            char *dst0 = lds + G::OFF_W + (set2 + i) * G::WSLOT_B;
#pragma unroll
            for (int r = 0; r < G::RW; ++r) {
                const int n = (wave + 4 * r) % G::NPW;                  // 1 KiB piece n of the slot [piece, half][BM rows][16 B]
                const int o16 = n * 64 + lane, plh = o16 / BM, row = o16 % BM;
                const char *src = valid ? src0 + (size_t(plh) * M + row) * 16 : zpage;
                cb3_glds_b128(src, dst0 + n * 1024);
            }
        }
        if (++w_g == nch * G::NGRP) {
            w_g = 0;
            ++w_k;
            int b_, rb_, cb_, mb_;
            decode(min(int(blockIdx.x) + w_k * int(gridDim.x), ntiles - 1), b_, rb_, cb_, mb_);
            w_m0 = mb_ * BM;
        }
    };

    // ---- input stream: task u = (channel half, plane position); 8 channels of one position per task ----
    int i_k = 0, i_chunk = 0;
    float st[NT][8];
    int st_t[NT];
    auto input_load = [&]() {
        const bool live = i_k < my_tiles;
        int b_, rb_, cb_, mb_;
        decode(min(int(blockIdx.x) + i_k * int(gridDim.x), ntiles - 1), b_, rb_, cb_, mb_);
        // FORWARD of a strided layer (p.s / p.sh = its column / row stride, p.cin_real = its channels): the kernel runs the
        // space-to-depth form -- sh sw cin_real virtual channels (phase-major), ceil(k / s) taps, stride 1 -- and this is where the
        // virtual planes come from: chunk -> (phase, 16 real channels), plane position (pr, pc) -> input (sh pr' + rho_h - ph,
        // s pc' + rho_w - P).  Stride-1 plans: one phase, the identity.
        const int cv0 = i_chunk * 16, phase = cv0 / p.cin_real, ci0 = cv0 - phase * p.cin_real;
        const int rho_h = phase / p.s, rho_w = phase - rho_h * p.s;
        const int row0 = p.sh * (rb_ * G::R) + rho_h - p.ph, col0 = p.s * (cb_ * G::WF) + rho_w - p.P;
        const char *xc = reinterpret_cast<const char *>(x + (size_t(b_) * p.cin_real + ci0) * p.x_cstride);
        unsigned cs4 = unsigned(p.x_cstride) * 4u;
        asm volatile("" : "+s"(cs4));
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int u = tid + 256 * n;
            const int uh = u >= W ? 1 : 0;
            const int t = u - uh * W;
            const int pr = t / SWP, gr = row0 + p.sh * pr, gc = col0 + p.s * (t - pr * SWP);
            const bool task = u < 2 * W;
            const bool ok = live && task && gr >= 0 && gr < Hin && gc >= 0 && gc < Win;
            const unsigned off = unsigned(8 * uh) * cs4 + unsigned(min(max(gr, 0), Hin - 1) * Win + min(max(gc, 0), Win - 1)) * 4u;
            st_t[n] = task ? (uh * W + t) : -1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = *reinterpret_cast<const float *>(xc + (off + unsigned(e) * cs4));
                st[n][e] = ok ? v : 0.f;
            }
        }
        if (++i_chunk == nch) i_chunk = 0, ++i_k;
    };
    auto input_store_all = [&]() {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            cb3x8 h, m, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 hh = (__bf16)st[n][e];
                const float r1 = st[n][e] - (float)hh;
                const __bf16 mm = (__bf16)r1;
                h[e] = hh;
                m[e] = mm;
                l[e] = (__bf16)(r1 - (float)mm);
            }
            if (st_t[n] >= 0) {
                *reinterpret_cast<cb3x8 *>(lds + (0 * 2 * W + st_t[n]) * 16) = h;
                *reinterpret_cast<cb3x8 *>(lds + (1 * 2 * W + st_t[n]) * 16) = m;
                *reinterpret_cast<cb3x8 *>(lds + (2 * 2 * W + st_t[n]) * 16) = l;
            }
        }
    };

    const int aLane = (lh * BM + r0w + li) * 16;
    int bLane[NW];                                        // this lane's plane position of output column block kk, tap (0, 0)
#pragma unroll
    for (int kk = 0; kk < NW; ++kk) {
        const int n = n0 + 32 * kk + li;
        bLane[kk] = (lh * W + n + (G::KW - 1) * (n >> SL)) * 16;
    }
    auto load_a = [&](cb3x8 (&a)[3][MW], int slot) {
        const char *ws = lds + G::OFF_W + slot * G::WSLOT_B + aLane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < MW; ++i) a[pl][i] = *reinterpret_cast<const cb3x8 *>(ws + (pl * 2 * BM + 32 * i) * 16);
    };
    auto load_b = [&](cb3x8 (&bf)[3], int tapoff, int kk) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const cb3x8 *>(lds + bLane[kk] + (pl * 2 * W + tapoff) * 16);
    };

    // ---- prologue ----
    input_load();
    dma_next_group();
    dma_next_group();
    input_store_all();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    input_load();

    cb3x8 fa[2][3][MW], fb[2][3];
#ifdef C2B3_STAMPS
    unsigned long long c2s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c2s_prev = 0;
#endif
    C2S_START();
    for (int k = 0; k < my_tiles; ++k) {
        int b, rb, cb, mb;
        decode(int(blockIdx.x) + k * int(gridDim.x), b, rb, cb, mb);
        const int m0 = mb * BM;
        f32x16 acc[MW][NW];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int kk = 0; kk < NW; ++kk)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][kk][r] = 0.f;
        load_a(fa[0], 0);
        load_b(fb[0], G::tap(0), 0);

        for (int c2 = 0; c2 < nch; c2 += 2) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int ua = (cc * J + j) & 1;
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const int step = j * NW + kk;
                        const int sb = (cc * NSTEP + step) & 1;
                        const bool last_of_phase = kk == NW - 1;
                        const bool group_end = last_of_phase && (j % G::GT == G::GT - 1 || j == J - 1);
                        const bool chunk_end = last_of_phase && j == J - 1;
                        C2S(6);
                        if (group_end) {      // early barrier: this group's last operands are in registers
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __syncthreads();
                            C2S(1);
                            dma_next_group();
                            C2S(0);
                        }
                        f32x16 part[MW];
#pragma unroll
                        for (int i = 0; i < MW; ++i) part[i] = acc[i][kk];
                        __builtin_amdgcn_sched_barrier(0);
                        cb3_products<MW>(part, fa[ua], fb[sb], 0, 3);
                        __builtin_amdgcn_sched_barrier(0);
                        C2S(2);
                        int nj = j, nk = kk + 1, ncc = cc;
                        if (nk == NW) nk = 0, ++nj;
                        if (nj == J) nj = 0, ncc = cc ^ 1;      // (two chunks = ten groups: the parity is back where it started)
                        const int nslot = ((ncc * G::NGRP + nj / G::GT) & 1) * G::GT + nj % G::GT;
                        if (!chunk_end) {      // (the next chunk's planes only exist behind the second barrier below)
                            if (nk == 0) load_a(fa[ua ^ 1], nslot);
                            load_b(fb[sb ^ 1], G::tap(nj), nk);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        C2S(3);
                        cb3_products<MW>(part, fa[ua], fb[sb], 3, 6);
                        C2S(2);
#pragma unroll
                        for (int i = 0; i < MW; ++i) acc[i][kk] = part[i];
                        if (chunk_end) {
                            input_store_all();
                            __builtin_amdgcn_sched_barrier(0);
                            C2S(4);
                            __syncthreads();
                            C2S(1);
                            load_a(fa[ua ^ 1], nslot);
                            load_b(fb[sb ^ 1], G::tap(0), 0);
                            C2S(3);
                            input_load();
                            C2S(7);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }

        C2S(6);
        // ---- epilogue.  Register r of row block i: row m0 + r0w + 32 i + 8 (r / 4) + 4 lh + r % 4; column n -> base (row, column) of the tile.
        // No load may sit between two stores: vmcnt counts loads and stores in order, so a bias load per element (the first version)
        // made every store wait for the one before it -- 16 MW NW serial memory round trips per tile, a third of the 32-row layers'
        // time (tools/c2b3_stamps.py).  The bias values of the tile are fetched in one batch, the gradient-add / mask operands of a
        // block of 16 elements in one batch before its stores (one drain of the store queue per block instead of one per element).
        const bool pre = (p.epilogue & AGX_EPI_LEAKY_PRE) != 0;
        const size_t ybase = size_t(b) * p.Cout * p.y_cstride;
        float *yb = y + ybase;
        const float *ab = add ? add + ybase : nullptr, *kb = mask ? mask + ybase : nullptr;
        unsigned ycs = unsigned(p.y_cstride);
        asm volatile("" : "+s"(ycs));
        float bv[MW][16];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bv[i][r] = bias ? bias[(m0 + r0w + 32 * i + 8 * (r >> 2) + 4 * lh + (r & 3)) / (Q * QH)] : 0.f;
        constexpr int NB = MW * NW;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {      // block (kk, i): 16 elements per lane
            const int kk = blk / MW, i = blk % MW;
            const int n = n0 + 32 * kk + li;
            const int brow = rb * G::R + (n >> SL), bcol = cb * G::WF + (n & (G::WF - 1));
            const bool okb = brow < p.Tt && bcol < p.Lt;
            auto elem = [&](int r, bool &ok) -> unsigned {      // (recomputed where needed: 16 offsets per block do not fit the registers)
                const int m = m0 + r0w + 32 * i + 8 * (r >> 2) + 4 * lh + (r & 3);
                const int co = m / (Q * QH), a = (m / Q) % QH, c = m % Q;
                const int orow = QH * brow + a - p.oshift_h, ocol = Q * bcol + c - p.oshift;
                ok = okb && orow >= 0 && orow < Hout && ocol >= 0 && ocol < Wout;
                return ok ? unsigned(co) * ycs + unsigned(orow * Wout + ocol) : 0u;
            };
            float rv[16], mv[16];
            if (ab) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    bool ok;
                    rv[r] = ab[elem(r, ok)];
                }
            }
            if (kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    bool ok;
                    mv[r] = kb[elem(r, ok)];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bool ok;
                const unsigned o = elem(r, ok);
                float v = acc[i][kk][r] + bv[i][r];
                if (pre) v = leaky(v, p.slope);
                if (ab) v += rv[r];
                if (kb) v = mv[r] > 0.f ? v : v * p.slope;
#ifdef C2B3_NOSTORE
                if (ok && v == 12345.678f) yb[o] = v;
#else
                if (ok) yb[o] = v;
#endif
            }
        }
        C2S(5);
    }
    C2S_WRITE();
}

enum { CB3_NONE = 0, CB3_UP2, CB3_UP4, CB3_UP5, CB3_UP8, CB3_K7, CB3_K3, CB3_DOWN2, CB3_DOWN4, CB3_DOWN5, CB3_DOWN8 };

// shape-only test (also decides whether agx_conv_pack of a bf16x3 descriptor appends the B3 tile image: common.hpp)
int conv_b3_geometry(const ConvPlan &p) {
    if (p.prec != 1 || p.G != 1 || p.d != 1 || p.kh != 1 || p.Tout != 1 || p.pm_R != 0) return CB3_NONE;
    if (p.Cin % 32 != 0) return CB3_NONE;                      // whole 16-channel chunks, an even number of them
    if (p.s == 1) {
        if (p.q == 2 && p.J == 3 && p.P == 1 && p.M == 64) return CB3_UP2;
        if (p.q == 4 && p.J == 3 && p.P == 1 && p.M % 128 == 0) return CB3_UP4;
        if (p.q == 5 && p.J == 3 && p.P == 1 && p.M % 128 == 0) return CB3_UP5;
        if (p.q == 8 && p.J == 3 && p.P == 1 && p.M % 128 == 0) return CB3_UP8;
        if (p.q == 1 && p.J == 7 && p.P == 6 && p.M % 128 == 0) return CB3_K7;
        if (p.q == 1 && p.J == 3 && p.P == 2 && p.M % 128 == 0) return CB3_K3;      // causal k = 3 (the encoder's last conv, vae.py:266)
        return CB3_NONE;
    }
    // round 4: the encoder's strided down-convs CausalConv1d(K = 2 s + 1, stride s) (vae.py:136-139), P = K - s
    if (p.q != 1) return CB3_NONE;
    if (p.s == 2 && p.J == 5 && p.P == 3 && p.M == 64) return CB3_DOWN2;
    if (p.s == 4 && p.J == 9 && p.P == 5 && p.M % 128 == 0) return CB3_DOWN4;
    if (p.s == 5 && p.J == 11 && p.P == 6 && p.M % 128 == 0) return CB3_DOWN5;
    if (p.s == 8 && p.J == 17 && p.P == 9 && p.M % 128 == 0) return CB3_DOWN8;
    return CB3_NONE;
}

bool conv_b3_supported(const ConvPlan &p) {
    if (p.tile_off < 0 || conv_b3_geometry(p) == CB3_NONE) return false;
    if ((p.epilogue & ~AGX_EPI_LEAKY_PRE) != 0 || p.oshift != 0 || p.mask != nullptr) return false;
    if (p.Lvalid != p.Lin || p.Lin < 1 || p.Lout != p.q * p.Lt || (p.s == 1 && p.Lt != p.Lin)) return false;
    if (int64_t(p.Cin) * p.Lin * 4 >= (int64_t(1) << 32)) return false;     // 32-bit byte offsets of the input loads
    return true;
}

const char *conv_b3_variant(const ConvPlan &p) {
    switch (conv_b3_geometry(p)) {
        case CB3_UP2: return "conv_b3<up2,64x256>";
        case CB3_UP4: return "conv_b3<up4,128x128>";
        case CB3_UP5: return "conv_b3<up5,128x128>";
        case CB3_UP8: return "conv_b3<up8,128x128>";
        case CB3_K7: return "conv_b3<k7,128x128>";
        case CB3_K3: return "conv_b3<k3,128x128>";
        case CB3_DOWN2: return "conv_b3<down2,64x128>";
        case CB3_DOWN4: return "conv_b3<down4,128x64>";
        case CB3_DOWN5: return "conv_b3<down5,128x64>";
        case CB3_DOWN8: return "conv_b3<down8,128x32>";
        default: return "conv_b3<unsupported>";
    }
}

template <int MW, int NW, int WM, int J_, int Q_, int P_, bool XP = false, bool YP = false, int S_ = 1>
static int launch_cb3(const ConvPlan &p, const float *x, const float *wp, const float *bias, float *y, hipStream_t st) {
    using G = Cb3Geom<MW, NW, WM, J_, Q_, P_, XP, S_>;
    auto kern = conv_b3_kernel<MW, NW, WM, J_, Q_, P_, XP, YP, S_>;
    static DeviceOnce once;
    int n_cu = 0;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, &n_cu, "conv_b3")) return rc;
    static_assert(2 * G::LDS_BYTES <= 160 * 1024, "conv_b3: LDS budget of two workgroups per CU");
    const int tb = ceil_div(p.Lt, G::BN), mb = p.M / G::BM;
    const int64_t ntiles64 = int64_t(tb) * mb * p.B;
    if (ntiles64 > (1 << 30)) return fail(AGX_ERR_BAD_SHAPE, "conv_b3: too many tiles");
    const int ntiles = int(ntiles64);
    const int grid = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
    // B3 tile image: behind the bf16x3 standard image and the dim0 scale scratch
    const char *wt = reinterpret_cast<const char *>(wp + p.tile_off);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, st, p, tb, mb, ntiles, x, wt, bias, y);
    return check_launch("conv_b3");
}

int launch_conv_b3(const ConvPlan &p, const float *x, const float *wp, const float *bias, float *y, hipStream_t st) {
    if (!conv_b3_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "conv_b3: unsupported layer");
    switch (conv_b3_geometry(p)) {
        case CB3_UP2: return launch_cb3<2, 2, 1, 3, 2, 1>(p, x, wp, bias, y, st);
        case CB3_UP4: return launch_cb3<2, 2, 2, 3, 4, 1>(p, x, wp, bias, y, st);
        case CB3_UP5: return launch_cb3<2, 2, 2, 3, 5, 1>(p, x, wp, bias, y, st);
        case CB3_UP8: return launch_cb3<2, 2, 2, 3, 8, 1>(p, x, wp, bias, y, st);
        case CB3_K7: return launch_cb3<2, 2, 2, 7, 1, 6>(p, x, wp, bias, y, st);
        case CB3_K3: return launch_cb3<2, 2, 2, 3, 1, 2>(p, x, wp, bias, y, st);
        //                          MW NW WM  J  Q  P  XP     YP     S      tile (rows x output columns)
        case CB3_DOWN2: return launch_cb3<2, 1, 1, 5, 1, 3, false, false, 2>(p, x, wp, bias, y, st);     //  64 x 128
        case CB3_DOWN4: return launch_cb3<2, 1, 2, 9, 1, 5, false, false, 4>(p, x, wp, bias, y, st);     // 128 x 64
        case CB3_DOWN5: return launch_cb3<2, 1, 2, 11, 1, 6, false, false, 5>(p, x, wp, bias, y, st);    // 128 x 64
        case CB3_DOWN8: return launch_cb3<1, 1, 4, 17, 1, 9, false, false, 8>(p, x, wp, bias, y, st);    // 128 x 32
        default: return fail(AGX_ERR_UNSUPPORTED, "conv_b3: unsupported layer");
    }
}

// The same layers fed with activation planes (common.hpp) -- x_planes: bf16 [B][Cin / 8][3][Lin][8]; y_planes (k = 7 layer only, may
// be NULL): the output as planes [B][Cout / 8][3][Lout][8] instead of fp32.
int launch_conv_b3_planes(const ConvPlan &p, const void *x_planes, const float *wp, const float *bias, float *y, void *y_planes,
                          hipStream_t st) {
    if (!conv_b3_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "conv_b3: unsupported layer");
    if (int64_t(p.Cin / 8) * 3 * p.Lin * 16 >= (int64_t(1) << 40)) return fail(AGX_ERR_BAD_SHAPE, "conv_b3: clip too long");
    const float *xp = static_cast<const float *>(x_planes);
    const int g = conv_b3_geometry(p);
    if (p.s != 1 || g == CB3_K3) return fail(AGX_ERR_UNSUPPORTED, "conv_b3: plane input is for the decoder's stride-1 layers");
    if (y_planes && (g != CB3_K7 || p.Cout % 8 != 0)) return fail(AGX_ERR_UNSUPPORTED, "conv_b3: plane output is for the one-phase layer");
    switch (g) {
        case CB3_UP2: return launch_cb3<2, 2, 1, 3, 2, 1, true>(p, xp, wp, bias, y, st);
        case CB3_UP4: return launch_cb3<2, 2, 2, 3, 4, 1, true>(p, xp, wp, bias, y, st);
        case CB3_UP5: return launch_cb3<2, 2, 2, 3, 5, 1, true>(p, xp, wp, bias, y, st);
        case CB3_UP8: return launch_cb3<2, 2, 2, 3, 8, 1, true>(p, xp, wp, bias, y, st);
        case CB3_K7:
            return y_planes ? launch_cb3<2, 2, 2, 7, 1, 6, true, true>(p, xp, wp, bias, static_cast<float *>(y_planes), st)
                            : launch_cb3<2, 2, 2, 7, 1, 6, true, false>(p, xp, wp, bias, y, st);
        default: return fail(AGX_ERR_UNSUPPORTED, "conv_b3: unsupported layer");
    }
}

// fp32 (B, C, L) -> activation planes: one thread = one cell (8 channels of one time step), loads coalesced along time
__global__ __launch_bounds__(256) void planes_split_kernel(const float *__restrict__ x, char *__restrict__ planes, int C8, int L,
                                                           int64_t ncell) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= ncell) return;
    const int t = int(i % L);
    const int64_t bg = i / L;                       // b * C8 + g
    const float *src = x + bg * 8 * L + t;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[size_t(e) * L];
    cb3x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hh = (__bf16)v[e];
        const float r1 = v[e] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        h[e] = hh;
        m[e] = mm;
        l[e] = (__bf16)(r1 - (float)mm);
    }
    char *dst = planes + (bg * 3 * L + t) * 16;
    *reinterpret_cast<cb3x8 *>(dst) = h;
    *reinterpret_cast<cb3x8 *>(dst + size_t(L) * 16) = m;
    *reinterpret_cast<cb3x8 *>(dst + size_t(L) * 32) = l;
}
int launch_planes_split(const float *x, void *planes, int batch, int channels, int length, hipStream_t st) {
    const int64_t ncell = int64_t(batch) * (channels / 8) * length;
    if (ncell > (int64_t(1) << 38)) return fail(AGX_ERR_BAD_SHAPE, "planes_split: too many cells");
    hipLaunchKernelGGL(planes_split_kernel, dim3((unsigned)ceil_div64(ncell, 256)), dim3(256), 0, st, x, static_cast<char *>(planes),
                       channels / 8, length, ncell);
    return check_launch("planes_split");
}

// floats of the B3 tile image of a Conv2d plan with the ring form: the plan's own bf16x3 weights, or -- strided forward layers -- the
// space-to-depth weights (sh sw Cin virtual channels x ceil(kh / sh) ceil(kw / sw) taps: the same count when the strides divide the
// kernel, as they do for the two layer shapes covered)
int64_t conv2d_b3_tile_floats(const ConvPlan &p) {
    if (p.s == 1 && p.sh == 1) return packed_weight_floats_bf(p.ncv, p.J, p.M);
    const int khv = ceil_div(p.kh, p.sh), kwv = ceil_div(p.J / p.kh, p.s);
    return packed_weight_floats_bf(p.sh * p.s * p.Cin, khv * kwv, p.M);
}

// ---- Conv2d (conv2d_b3_kernel) ---------------------------------------------------------------------------------------
enum { C2B3_NONE = 0, C2B3_M128 = 1, C2B3_M64 = 2, C2B3_M32 = 3 };      // tile shape (low 4 bits of the geometry code)
enum { C2B3_T33 = 0, C2B3_T22 = 1, C2B3_T32 = 2, C2B3_T22F = 3, C2B3_T32F = 4 };   // tap shape (next 4 bits); F: strided FORWARD, space-to-depth

// shape-only test (also decides whether the conv2d pack functions append the B3 tile image: conv2d.hip)
int conv2d_b3_geometry(const ConvPlan &p) {
    if (p.prec != 1 || p.G != 1 || p.pm_R <= 0 || p.d != 1) return C2B3_NONE;
    if (p.cin_real != p.Cin || p.ncv != p.Cin) return C2B3_NONE;
    int taps;
    if (p.s == 2 && p.q == 1 && p.qh == 1 && p.P == 1 && p.ph == 1 && p.oshift == 0 && p.oshift_h == 0 && p.Cin % 16 == 0 &&
        ((p.sh == 2 && p.kh == 4 && p.J == 16) || (p.sh == 1 && p.kh == 3 && p.J == 12))) {
        // FORWARD of the (4,4)/(2,2) and (3,4)/(1,2) layers, pad (1,1): space-to-depth = sh sw Cin virtual channels (16-channel
        // chunks stay inside one phase), 2 x 2 resp. 3 x 2 taps, stride 1 (conv2d_b3_kernel's staging)
        if (p.Lt != p.Lout || p.Tt != p.Tout) return C2B3_NONE;
        taps = p.sh == 2 ? C2B3_T22F : C2B3_T32F;
    } else if (p.s != 1 || p.sh != 1 || p.Cin % 32 != 0) {             // (stride-1 plans: whole 16-channel chunks, an even number)
        return C2B3_NONE;
    } else if (p.kh == 3 && p.J == 9 && p.q == 1 && p.qh == 1) {      // 3 x 3, stride 1, "same": forward and backward-data
        if (p.P != 1 || p.ph != 1 || p.oshift != 0 || p.oshift_h != 0) return C2B3_NONE;
        if (p.Lt != p.Lout || p.Tt != p.Tout || p.Lout != p.Lin || p.Tout != p.Tin) return C2B3_NONE;
        taps = C2B3_T33;
    } else if (p.kh == 2 && p.J == 4 && p.q == 2 && p.qh == 2 && p.P == 1 && p.ph == 1) {
        taps = C2B3_T22;                                              // backward-data of a 4 x 4 stride-(2, 2) layer
    } else if (p.kh == 3 && p.J == 6 && p.q == 2 && p.qh == 1 && p.P == 1 && p.ph == 1 && p.oshift_h == 0 && p.Tt == p.Tout) {
        taps = C2B3_T32;                                              // backward-data of a 3 x 4 stride-(1, 2) layer
    } else {
        return C2B3_NONE;
    }
    const int tile = p.M % 128 == 0 ? C2B3_M128 : p.M == 64 ? C2B3_M64 : p.M == 32 ? C2B3_M32 : C2B3_NONE;
    return tile == C2B3_NONE ? C2B3_NONE : (tile | (taps << 4));
}

// tile rows R = BN >> SL of WF = 2^SL columns over the base grid.  A tile costs its matrix time (the same for every split) plus
// the staging of its input planes -- (R + kh - 1)(WF + kw - 1) positions x 2 halves of 8 channels in rounds of 256 threads, ~450
// cycles a round against 32 cycles per MFMA: two rows of 128 columns stage 1040 tasks = 5 rounds, eight rows of 32 columns 680 =
// 3 -- so the split minimises (padded area) x (matrix cycles + rounds x (450 + 3200 / WF)) per 16-channel chunk; measured on the
// window-1024 discriminator: 64 -> 64 3 x 3 127 -> 158 TFLOP/s, the 32-row layers 88 -> 97, the (3,4)/(1,2) forward 86 -> 123.  (c2b3_sl = -1: the first rule,
// least padded area with ties to the widest rows.)
// kh, kw, frags: the kernel's taps and MFMA tiles per wave (MW x NW); frags = 0: the padded area alone (the support gate)
static int c2b3_pick_sl(const ConvPlan &p, int BN, int sl_max, int kh, int kw, int frags) {
    const int forced = frags ? tuning().c2b3_sl : -1;
    if (forced >= 3) return forced < sl_max ? forced : sl_max;
    const int64_t mfma = int64_t(kh) * kw * frags * 6 * 32;      // per wave and chunk
    int best = 3;
    int64_t best_cost = -1;
    for (int sl = 3; sl <= sl_max; ++sl) {
        const int R = BN >> sl, WF = 1 << sl;
        if (R < 1) continue;
        const int64_t area = int64_t(ceil_div(p.Tt, R)) * R * ceil_div(p.Lt, WF) * WF;
        const int rounds = ceil_div(2 * (R + kh - 1) * (WF + kw - 1), 256);
        const int64_t cost = forced < 0 ? area : area * (mfma + rounds * (450 + 3200 / WF));   // (short rows: shorter runs in memory)
        if (best_cost < 0 || cost <= best_cost) best = sl, best_cost = cost;     // (ties: the wider rows -- longer runs in memory)
    }
    return best;
}

// Backward-data of a column-strided layer: the base grid of the phase GEMM has Lt = W / sw + 1 positions (513, 257, ...), one more
// than whole power-of-two blocks, and its first position only produces output column 0.  As on the fp32 ring (conv_p.hip:
// conv2d_bwd_first_cols_kernel) the ring kernel runs f' = 1 .. Lt - 1 and this kernel the first column, in fp32 on the weights
// h + m + l of the standard bf16x3 image (exact: the three pieces hold the 24 bits of the fp32 weight):
//   dx[ci, QH t' + a - oshift_h, 0] = sum_{co, jh} W[co][jh Jw + Jw - 1][m] dy[co, t' - ph + jh, 0],   m = (ci QH + a) Q + oshift.
// One workgroup per (clip, base row); the dy column sits in LDS; four lanes share a row m.
__global__ __launch_bounds__(256) void conv2d_b3_first_col_kernel(ConvPlan p, const float *__restrict__ dy,
                                                                  const __bf16 *__restrict__ img, const float *__restrict__ add,
                                                                  const float *__restrict__ mask, float *__restrict__ dx) {
    extern __shared__ float dcol[];   // [kh][Cin]
    const int bq = blockIdx.x / p.Tt, trow = blockIdx.x - bq * p.Tt;
    const int Jw = p.J / p.kh, nv = p.kh * p.Cin;
    for (int v = threadIdx.x; v < nv; v += 256) {
        const int jh = v / p.Cin, co = v - jh * p.Cin;
        const int r = trow - p.ph + jh;
        dcol[v] = (r >= 0 && r < p.Tin) ? dy[(size_t(bq) * p.Cin + co) * p.x_cstride + size_t(r) * p.Lin] : 0.f;
    }
    __syncthreads();
    const int rows = p.M / p.q, sub = threadIdx.x & 3;
    for (int mq = threadIdx.x >> 2; mq < rows; mq += 64) {
        const int m = mq * p.q + p.oshift;
        const int ci = mq / p.qh, a = mq - ci * p.qh;
        const int orow = p.qh * trow + a - p.oshift_h;
        float acc = 0.f;
        for (int v = sub; v < nv; v += 4) {       // (jh, co): 4 lanes x every 4th virtual channel, fixed order
            const int jh = v / p.Cin, co = v - jh * p.Cin;
            const __bf16 *w = img + ((size_t(co >> 4) * p.J + jh * Jw + (Jw - 1)) * p.M + m) * 48 + (co & 15);
            acc = fmaf(((float)w[0] + (float)w[16]) + (float)w[32], dcol[v], acc);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (sub != 0 || orow < 0 || orow >= p.Tout) continue;
        const size_t o = (size_t(bq) * p.Cout + ci) * p.y_cstride + size_t(orow) * p.Lout;
        if (add) acc += add[o];
        if (mask) acc = mask[o] > 0.f ? acc : acc * p.slope;
        dx[o] = acc;
    }
}

// the plan the ring kernel runs once the first base column is peeled off (above)
// (worth it on wide maps only: the first column is 1 / Lt of the layer's work on a scalar kernel -- base grids of 513 / 257 columns
// 2.07 -> 1.75 ms and 2.53 -> 2.24 ms, 129 and narrower lose)
static inline bool c2b3_peels_first_col(const ConvPlan &p) { return p.q == 2 && p.oshift == 1 && p.P == 1 && p.Lt > 192; }

bool conv2d_b3_supported(const ConvPlan &p) {
    const int geom = conv2d_b3_geometry(p), tile = geom & 15;
    if (p.tile_off < 0 || geom == C2B3_NONE) return false;
    {   // very narrow / ragged feature maps: beyond 1.4 x padded area the fp32 ring kernel wins
        ConvPlan pp = p;
        if (c2b3_peels_first_col(p)) pp.Lt = p.Lt - 1;
        const int BN = tile == C2B3_M128 ? 128 : 256, sl = c2b3_pick_sl(pp, BN, tile == C2B3_M128 ? 6 : 7, 1, 1, 0);
        const int R = BN >> sl, WF = 1 << sl;
        const int64_t area = int64_t(ceil_div(pp.Tt, R)) * R * ceil_div(pp.Lt, WF) * WF;
        if (area * 10 > int64_t(pp.Tt) * pp.Lt * 14) return false;
    }
    if ((p.epilogue & ~(AGX_EPI_LEAKY_PRE | AGX_EPI_RESIDUAL | AGX_EPI_MASK)) != 0) return false;
    if (p.x_cstride != int64_t(p.Tin) * p.Lin || p.y_cstride != int64_t(p.Tout) * p.Lout) return false;
    if (p.x_cstride * 16 * 4 >= (int64_t(1) << 32) || p.y_cstride * p.Cout >= (int64_t(1) << 31)) return false;   // 32-bit offsets
    return true;
}

const char *conv2d_b3_variant(const ConvPlan &p) {
    static const char *names[5][3] = {{"conv2d_b3<3x3,128x128>", "conv2d_b3<3x3,64x256>", "conv2d_b3<3x3,32x256>"},
                                      {"conv2d_b3<2x2 phases 2x2,128x128>", "conv2d_b3<2x2 phases 2x2,64x256>", "conv2d_b3<2x2 phases 2x2,32x256>"},
                                      {"conv2d_b3<3x2 phases 1x2,128x128>", "conv2d_b3<3x2 phases 1x2,64x256>", "conv2d_b3<3x2 phases 1x2,32x256>"},
                                      {"conv2d_b3<4x4 s2 as 2x2 s2d,128x128>", "conv2d_b3<4x4 s2 as 2x2 s2d,64x256>", "conv2d_b3<4x4 s2 as 2x2 s2d,32x256>"},
                                      {"conv2d_b3<3x4 s(1,2) as 3x2 s2d,128x128>", "conv2d_b3<3x4 s(1,2) as 3x2 s2d,64x256>", "conv2d_b3<3x4 s(1,2) as 3x2 s2d,32x256>"}};
    const int geom = conv2d_b3_geometry(p);
    return geom == C2B3_NONE ? "conv2d_b3<unsupported>" : names[geom >> 4][(geom & 15) - 1];
}

template <int MW, int NW, int WM, int SL, int KH, int KW, int Q, int QH>
static int launch_c2b3(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *add, const float *mask,
                       float *y, hipStream_t st) {
    using G = C2b3Geom<MW, NW, WM, SL, KH, KW>;
    auto kern = conv2d_b3_kernel<MW, NW, WM, SL, KH, KW, Q, QH>;
    static DeviceOnce once;
    int n_cu = 0;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, &n_cu, "conv2d_b3")) return rc;
    static_assert(2 * G::LDS_BYTES <= 160 * 1024, "conv2d_b3: LDS budget of two workgroups per CU");
    const int cb = ceil_div(p.Lt, G::WF), rb = ceil_div(p.Tt, G::R), mb = p.M / G::BM;
    const int64_t ntiles64 = int64_t(cb) * rb * mb * p.B;
    if (ntiles64 > (1 << 30)) return fail(AGX_ERR_BAD_SHAPE, "conv2d_b3: too many tiles");
    const int ntiles = int(ntiles64);
    const int grid = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
    const char *wt = reinterpret_cast<const char *>(wp + p.tile_off);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, st, p, cb, rb, mb, ntiles, x, wt, bias,
                       (p.epilogue & AGX_EPI_RESIDUAL) ? add : nullptr, (p.epilogue & AGX_EPI_MASK) ? mask : nullptr, y);
    return check_launch("conv2d_b3");
}

template <int MW, int NW, int WM, int SLMAX, int KH, int KW, int Q, int QH>
static int launch_c2b3_sl(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *add,
                          const float *mask, float *y, hipStream_t st) {
    const int sl = c2b3_pick_sl(p, 32 * NW * (4 / WM), SLMAX, KH, KW, MW * NW);
    if (sl == 3) return launch_c2b3<MW, NW, WM, 3, KH, KW, Q, QH>(p, x, wp, bias, add, mask, y, st);
    if (sl == 4) return launch_c2b3<MW, NW, WM, 4, KH, KW, Q, QH>(p, x, wp, bias, add, mask, y, st);
    if (sl == 5) return launch_c2b3<MW, NW, WM, 5, KH, KW, Q, QH>(p, x, wp, bias, add, mask, y, st);
    if (sl == 6 || SLMAX == 6) return launch_c2b3<MW, NW, WM, 6, KH, KW, Q, QH>(p, x, wp, bias, add, mask, y, st);
    return launch_c2b3<MW, NW, WM, SLMAX, KH, KW, Q, QH>(p, x, wp, bias, add, mask, y, st);
}

template <int KH, int KW, int Q, int QH>
static int launch_c2b3_tile(int tile, const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res,
                            float *y, hipStream_t st) {
    switch (tile) {
        case C2B3_M128: return launch_c2b3_sl<2, 2, 2, 6, KH, KW, Q, QH>(p, x, wp, bias, res, p.mask, y, st);   // (one row of 128 columns: the planes of two workgroups do not fit)
        case C2B3_M64: return launch_c2b3_sl<2, 2, 1, 7, KH, KW, Q, QH>(p, x, wp, bias, res, p.mask, y, st);
        case C2B3_M32: return launch_c2b3_sl<1, 2, 1, 7, KH, KW, Q, QH>(p, x, wp, bias, res, p.mask, y, st);
        default: return fail(AGX_ERR_UNSUPPORTED, "conv2d_b3: unsupported layer");
    }
}

// res = the tensor added in the epilogue (AGX_EPI_RESIDUAL: backward-data, the gradient arriving at this feature map); p.mask as conv_p2d
int launch_conv2d_b3(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res, float *y, hipStream_t st) {
    if (!conv2d_b3_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "conv2d_b3: unsupported layer");
    const int geom = conv2d_b3_geometry(p);
    if ((geom >> 4) == C2B3_T22F || (geom >> 4) == C2B3_T32F) {     // the virtual (space-to-depth) plan the kernel loops over
        ConvPlan v = p;
        v.Cin = p.sh * p.s * p.Cin;
        v.kh = (geom >> 4) == C2B3_T22F ? 2 : 3;
        v.J = v.kh * 2;
        return (geom >> 4) == C2B3_T22F ? launch_c2b3_tile<2, 2, 1, 1>(geom & 15, v, x, wp, bias, res, y, st)
                                        : launch_c2b3_tile<3, 2, 1, 1>(geom & 15, v, x, wp, bias, res, y, st);
    }
    if ((geom >> 4) == C2B3_T33) return launch_c2b3_tile<3, 3, 1, 1>(geom & 15, p, x, wp, bias, res, y, st);
    ConvPlan pp = p;
    if (c2b3_peels_first_col(p)) {      // column phases: base position 0 (output column 0) on its own kernel, whole blocks for the ring
        if (int64_t(p.B) * p.Tt > (int64_t(1) << 30)) return fail(AGX_ERR_BAD_SHAPE, "conv2d_b3: grid too large");
        hipLaunchKernelGGL(conv2d_b3_first_col_kernel, dim3(p.B * p.Tt), dim3(256), size_t(p.kh) * p.Cin * sizeof(float), st, p, x,
                           reinterpret_cast<const __bf16 *>(wp), res, p.mask, y);
        pp.Lt = p.Lt - 1;
        pp.oshift = p.oshift - 2;
        pp.P = p.P - 1;
    }
    return (geom >> 4) == C2B3_T22 ? launch_c2b3_tile<2, 2, 2, 2>(geom & 15, pp, x, wp, bias, res, y, st)
                                   : launch_c2b3_tile<3, 2, 2, 1>(geom & 15, pp, x, wp, bias, res, y, st);
}

}  // namespace agx

#ifdef C2B3_STAMPS
extern "C" int agx_debug_read_c2b3_stamps(unsigned long long *host, int n) {   // probe build only: copy out and clear
    if (n > (1 << 16)) n = 1 << 16;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(agx::g_c2b3_stamps), size_t(n) * 8) != hipSuccess) return AGX_ERR_LAUNCH;
    static unsigned long long zeros[1 << 16];
    if (hipMemcpyToSymbol(HIP_SYMBOL(agx::g_c2b3_stamps), zeros, sizeof(zeros)) != hipSuccess) return AGX_ERR_LAUNCH;
    return AGX_OK;
}
#endif
