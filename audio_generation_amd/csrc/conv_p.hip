// Persistent "ring" form of the dense 1-D polyphase convolution for gfx950 -- the resampling layers of the codec:
//
//   y[b, co, Q*t + ph] = act( bias[co] + sum_{ci, j < J} Wp[ci, j][co*Q + ph] * x[b, ci, t*S + j - P] )
//
//   strided down-convs      CausalConv1d(K = 2s+1, stride s)          networks/vae.py:136-139   Q = 1, J = K, S = s
//   polyphase up-convs      CausalUpsampleConv1d(K = 2s+1, x s)        networks/vae.py:176-179   Q = s, J = 3, S = 1
//   stride-1 (transposed)   CausalConv1d k3 / CausalConvT1d k7 s1      networks/vae.py:266, 269  Q = 1, S = 1
//
// Same operand path as resblock_p.hip (measured there: +15-20 % over the register-fed first kernel): a persistent
// workgroup walks over (clip, time block, row block) tiles; a ring of LDS slots holds, per chunk of CCH input channels,
// the weights of the tile's BM rows (contiguous pieces of the layer's tile image, shared by the four waves) and the
// input rows (16-byte cells, zero page outside the signal), both by dwordx4 LDS-DMA issued branch-free, spread over
// the MFMA phases; operand registers ping-pong over the unrolled taps; every LDS offset is an immediate (the kernel
// is instantiated per layer geometry).  Strided layers read the input tile with a lane stride of S dwords (an S-way
// bank conflict for S = 2, 4, 8): with NW fragments per MW x NW MFMAs the LDS has the cycles to spare.
#include <utility>

#include "mfma_tile.hpp"

namespace agx {

// 1 KiB of zeros in the code object: DMA source of input cells outside the signal and of unused instruction slots
__device__ __attribute__((aligned(1024))) float g_cp_zero_page[256] = {0.f};

__device__ __forceinline__ void cp_glds_b128(const float *gsrc_lane, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <class F, int... I>
__device__ __forceinline__ void cp_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cp_static_for(F &&f) {
    cp_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// MW x NW fragments of 32 x 32 per wave, WM x WN waves, CCH channels per chunk, J taps, input step S, Q output phases,
// PL = left offset P, NS ring slots (3: the next chunk is complete one interval early and its first operands are read
// before the barrier; 2: it completes AT the barrier)
// D2: Conv2d layers "row-folded" onto the same machinery (conv2d.hip): one tile = BN columns of ONE output row, the
// chunk sequence runs over (kernel row dh, CCH real channels) -- virtual channel v = dh * Cin + ci -- and the input rows
// of a chunk are row i * sh - ph + dh of CCH consecutive feature maps (the page of zeros when that row is padding).
// QH (D2 only): output-row phases -- backward-data of a strided Conv2d: rows m = (ci * QH + a) * Q + c of the GEMM are
// the (row, column) phases of input channel ci, written to row QH * t' + a - oshift_h, column Q * f' + c - oshift.
template <int MW_, int NW_, int WM_, int WN_, int CCH_, int J_, int S_, int Q_, int PL_, int NS_, bool D2_ = false, int QH_ = 1,
          int NB_ = 1024>
struct CpGeom {
    static constexpr int MW = MW_, NW = NW_, WM = WM_, WN = WN_, CCH = CCH_, J = J_, S = S_, Q = Q_, PL = PL_, NSLOT = NS_;
    static constexpr bool D2 = D2_;
    static constexpr int QH = QH_;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static constexpr int BM = 32 * MW * WM, BN = 32 * NW * WN;
    static constexpr int PA = (PL + 3) / 4 * 4, SHIFT = PA - PL;
    // D2 tiles may hold R = BN / WF output rows of WF columns each (narrow feature maps; WF a power of two >= MINWF,
    // chosen by the launcher): the LDS row of a channel is then R segments of seg_floats(WF) floats
    static constexpr int MINWF = 32;
    static constexpr int seg_floats(int wf) { return (SHIFT + (wf - 1) * S + J + 3) / 4 * 4; }
    static constexpr int SPAN1 = seg_floats(BN), SPANMR = (BN / MINWF) * seg_floats(MINWF);
    static constexpr int SPANP = (D2 && SPANMR > SPAN1) ? SPANMR : SPAN1;   // floats per LDS input row
    static constexpr int NCELL = SPANP / 4;
    static constexpr int KS = CCH / 2;                  // MFMA k-steps per (chunk, tap) phase
    static constexpr int NG4 = CCH / 4;                 // 4-channel blocks of the tile image per chunk
    static constexpr int AFL = NG4 * J * BM * 4;        // floats of weights per chunk: [g4][j][BM][4]
    static constexpr int NPA = AFL / 256;               // 1 KiB DMA instructions of the weight chunk
    static constexpr int PPB = BM / 64;                 // ... per (g4, j) piece
    static constexpr int NCB = CCH * NCELL, NIB = (NCB + 63) / 64, BFLP = NIB * 256;
    static constexpr int SLOT = AFL + BFLP;
    static constexpr int RA = (NPA + 3) / 4, RB = (NIB + 3) / 4, NOPS = RA + RB;
    static constexpr int OPP = (NOPS + J - 1) / J;      // DMA instructions per phase
    static constexpr int BIAS0 = NSLOT * SLOT;          // bias[Cout] staged once per kernel (Cout <= 1024)
    static constexpr int NBIAS = NB_;                  // most output channels a layer of this geometry may have (bias staged in LDS)
    static constexpr int DUMMY0 = BIAS0 + NBIAS;        // 1 KiB nobody reads: destination of the DMA slots a wave has no piece for
    static constexpr size_t LDS_BYTES = size_t(DUMMY0 + 256) * sizeof(float);
    static constexpr int NDS = MW * (KS == 8 ? 2 : 1) + KS * NW, NMF = KS * MW * NW;   // LDS reads / MFMAs per phase
    static_assert(KS == 2 || KS == 4 || KS == 8, "chunk of 4, 8 or 16 channels");
    // weight pieces (g4, j) are whole 1 KiB instructions; a 32-row tile needs M == 32: its chunk of the tile image is one
    // contiguous block and is copied flat
    static_assert(BM % 64 == 0 || (BM == 32 && (NG4 * J) % 2 == 0), "weight pieces");
};

template <int MW, int NW, int KS>
struct CpFrag {
    float a[KS][MW];
    float b[KS][NW];
};

template <class G>
__device__ __forceinline__ void cp_load_frag(CpFrag<G::MW, G::NW, G::KS> &f, const float *__restrict__ As,
                                             const float *__restrict__ Bs, int j, const int (&bk)[G::NW]) {
    constexpr int KS = G::KS;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < G::MW; ++i) {
        if (KS == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(As + (j * G::BM + i * 32) * 4);
            f.a[0][i] = v[0], f.a[1][i] = v[1], f.a[2][i] = v[2], f.a[3][i] = v[3];
        } else if (KS == 2) {
            const f32x2_t v = *reinterpret_cast<const f32x2_t *>(As + (j * G::BM + i * 32) * 4);
            f.a[0][i] = v[0], f.a[1][i] = v[1];
        } else {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(As + ((h2 * G::J + j) * G::BM + i * 32) * 4);
                f.a[4 * h2 + 0][i] = v[0], f.a[4 * h2 + 1][i] = v[1], f.a[4 * h2 + 2][i] = v[2], f.a[4 * h2 + 3][i] = v[3];
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int k = 0; k < G::NW; ++k)
            f.b[ks][k] = G::D2 ? Bs[bk[k] + ks * G::SPANP + j] : Bs[ks * G::SPANP + k * 32 * G::S + j];
}

template <class G>
__device__ __forceinline__ void cp_mfma_frag(f32x16 (&acc)[G::MW][G::NW], const CpFrag<G::MW, G::NW, G::KS> &f) {
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
        for (int i = 0; i < G::MW; ++i)
#pragma unroll
            for (int k = 0; k < G::NW; ++k)
                acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[ks][i], f.b[ks][k], acc[i][k], 0, 0, 0);
}

// thread the phase's LDS reads between its MFMAs
template <class G>
__device__ __forceinline__ void cp_interleave() {
    constexpr int PER = G::NMF / G::NDS > 0 ? G::NMF / G::NDS : 1;
#pragma unroll
    for (int g = 0; g < G::NDS; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
    }
    __builtin_amdgcn_sched_barrier(0);
}

// (clip, time block, row block) of a tile, advanced by the grid size with carries -- no division in the kernel
struct CpTileCur {
    int mb, nb, b;
    __device__ __forceinline__ void advance(int sm, int sn, int sb, int mblocks, int nblocks) {
        mb += sm;
        nb += sn;
        b += sb;
        if (mb >= mblocks) mb -= mblocks, ++nb;
        if (nb >= nblocks) nb -= nblocks, ++b;
    }
};

template <class G>
__global__ __launch_bounds__(256, 2) void conv_p_kernel(ConvPlan p, int mblocks, int nblocks, int ntiles, int step_m,
                                                        int step_n, int step_b, const float *__restrict__ x,
                                                        const float *__restrict__ timg, const float *__restrict__ bias,
                                                        float *__restrict__ y, const float *__restrict__ add2,
                                                        const float *__restrict__ mask2, int wfs) {
    constexpr int MW = G::MW, NW = G::NW, KS = G::KS, J = G::J, S = G::S, Q = G::Q, BM = G::BM, BN = G::BN;
    constexpr int SLOT = G::SLOT, AFL = G::AFL, CCH = G::CCH;
    // NSLOT >= 3: the next chunk is complete one interval early and its first operands are read before the barrier.  NSLOT > 3
    // (round 4, the one-phase k = 1 layers whose interval is only 16 / 32 MFMAs): DEPTH = NSLOT - 3 further chunks stay IN FLIGHT
    // across the interval's barrier -- the wait at the end of an interval is counted (vmcnt(DEPTH * NOPS): every wave issues
    // exactly NOPS DMA instructions per chunk, in order), so a chunk has DEPTH + 1 intervals to arrive instead of one.
    constexpr bool PRE3 = G::NSLOT >= 3;
    constexpr int DEPTH = G::NSLOT > 3 ? G::NSLOT - 3 : 0, QN = DEPTH + 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int Lin = p.Lin, M = p.M;
    constexpr bool D2 = G::D2;
    const int ncc = p.Cin / CCH;                   // channel chunks (per kernel row)
    const int nch = D2 ? ncc * p.kh : ncc;         // chunks per tile
    const int rstride = D2 ? int(p.x_cstride) : Lin;   // floats between the input rows of a chunk
    // D2: the tile is RT output rows x WF columns (WF = 1 << wfs); 1-D: one row of BN columns
    const int WF = D2 ? (1 << wfs) : BN, RT = D2 ? (BN >> wfs) : 1;
    const int segf = D2 ? (G::SHIFT + (WF - 1) * S + J + 3) / 4 * 4 : G::SPANP, segc = segf / 4;
    const int rbs = D2 ? (p.Tt + RT - 1) / RT : 1;   // row blocks per clip

    // ---- per-lane / per-wave constants of the DMA ---------------------------------------------------------------
    unsigned boffB[G::RB];
    int colB[G::RB];
    int rowB[D2 ? G::RB : 1];   // D2: input-row offset of the cell's segment
#pragma unroll
    for (int r = 0; r < G::RB; ++r) {
        const int e = (wave + 4 * r) * 64 + lane;
        const int row = e / G::NCELL, col = e - row * G::NCELL;
        if constexpr (D2) {
            const int rr = col / segc, cc = col - rr * segc;   // segment (output row of the tile), cell within it
            colB[r] = (e < G::NCB && rr < RT) ? 4 * cc : -(1 << 28);
            rowB[r] = rr * p.sh;
            boffB[r] = unsigned(row * rstride + rr * p.sh * Lin + 4 * cc) * 4u;
        } else {
            colB[r] = e < G::NCB ? 4 * col : -(1 << 28);
            boffB[r] = unsigned(row * rstride + 4 * col) * 4u;
        }
    }
    unsigned aoff[G::RA];   // weight instruction n = wave + 4 r: piece (g4, j) = n / PPB of the chunk, 1 KiB part n % PPB
#pragma unroll
    for (int r = 0; r < G::RA; ++r) {
        const int n = wave + 4 * r;
        if constexpr (G::PPB > 0) aoff[r] = unsigned(n / G::PPB) * unsigned(M) * 16u + unsigned(n % G::PPB) * 1024u;
        else aoff[r] = unsigned(n) * 1024u;   // BM = M = 32
    }
    const char *zpage = reinterpret_cast<const char *>(g_cp_zero_page) + lane * 16;
    // consumer-side lane offsets (floats, relative to a slot)
    const int aLane = (KS == 4 ? lh * J * BM * 4 : (KS == 8 ? 2 * lh * J * BM * 4 : lh * 2)) + (wm * 32 * MW + li) * 4;
    const int bLane = AFL + lh * KS * G::SPANP + (D2 ? 0 : (wn * 32 * NW + li) * S) + G::SHIFT;
    int bk[NW];   // D2: LDS offset of this lane's column in fragment k (row segment + column)
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int n = wn * 32 * NW + k * 32 + li;
        bk[k] = (n >> wfs) * segf + (n & (WF - 1)) * S;
    }

    const int my_tiles = int(blockIdx.x) < ntiles ? (ntiles - 1 - int(blockIdx.x)) / int(gridDim.x) + 1 : 0;
    const int nq = my_tiles * nch;
    if (nq == 0) return;
    CpTileCur first;
    {
        const int t = int(blockIdx.x);
        first.mb = t % mblocks;
        const int rest = t / mblocks;
        first.nb = rest % nblocks;
        first.b = rest / nblocks;
    }
    const size_t wchunk = size_t(G::NG4) * J * M * 16;   // bytes of the tile image per chunk (all rows)

    // DMA cursor (runs ahead of the MFMAs, across tile boundaries)
    CpTileCur ic = first;
    int iq = 0, icc = 0, isl = 0;
    int icr = 0, idh = 0, irow0 = 0;   // D2: channel chunk within the kernel row, kernel row, input row of dh = 0
    const char *w_next = reinterpret_cast<const char *>(timg) + size_t(ic.mb) * BM * 16;
    const char *x_tile = nullptr;      // D2: (clip, channel 0, row irow0, first column) of the tile
    auto tile_x = [&]() -> const char * {
        if (D2) {
            const int bq = ic.b / rbs, i = (ic.b - bq * rbs) * RT;   // (clip, first base row of the tile)
            irow0 = i * p.sh - p.ph;
            x_tile = reinterpret_cast<const char *>(x + size_t(bq) * p.cin_real * p.x_cstride + int64_t(irow0) * Lin +
                                                    (ic.nb * WF * S - G::PA));
            return x_tile;
        }
        return reinterpret_cast<const char *>(x + size_t(ic.b) * p.Cin * Lin + (ic.nb * BN * S - G::PA));
    };
    const char *x_next = tile_x();
    float *d_slot = lds;
    const char *d_w = nullptr, *d_x = nullptr;
    int d_in0a = 0;
    bool d_live = false;
    int d_irow = 0;   // D2: input row of the chunk's first segment (padding rows: cells from the zero page)
    auto begin_chunk = [&]() {
        d_live = iq < nq;
        if (D2) d_irow = irow0 + idh;
        d_slot = lds + isl * SLOT;
        d_w = w_next;
        d_x = x_next;
        d_in0a = ic.nb * WF * S - G::PA;
        ++iq;
        isl = isl + 1 == G::NSLOT ? 0 : isl + 1;
        w_next += wchunk;
        x_next += size_t(CCH) * rstride * sizeof(float);
        if (D2 && ++icr == ncc) {
            icr = 0;
            ++idh;
            x_next = x_tile + size_t(idh) * Lin * sizeof(float);
        }
        if (++icc == nch) {
            icc = 0;
            icr = 0;
            idh = 0;
            ic.advance(step_m, step_n, step_b, mblocks, nblocks);
            w_next = reinterpret_cast<const char *>(timg) + size_t(ic.mb) * BM * 16;
            x_next = tile_x();
        }
    };
    auto dma_op = [&](int k) {
        if (k < G::RA) {
            const int n = wave + 4 * k;
            const bool has = d_live && n < G::NPA;
            const char *src = has ? d_w + aoff[k] + lane * 16 : zpage;
            cp_glds_b128(reinterpret_cast<const float *>(src), has ? d_slot + n * 256 : lds + G::DUMMY0);
        } else if (k < G::NOPS) {
            const int r = k - G::RA, n = wave + 4 * r;
            const int pos = d_in0a + colB[r];
            const bool has = d_live && n < G::NIB;
            const bool rowok = !D2 || unsigned(d_irow + rowB[D2 ? r : 0]) < unsigned(p.Tin);
            const bool ok = has && rowok && pos >= 0 && pos < p.Lvalid;
            // a cell that straddles the end of the row (L % 4 != 0) is fetched from the row's LAST four elements
            // (in bounds) and put right by fix_ragged() once it has landed
            const int over = max(pos + 4 - p.Lvalid, 0);
            const char *src = ok ? d_x + boffB[r] - over * 4 : zpage;
            cp_glds_b128(reinterpret_cast<const float *>(src), has ? d_slot + AFL + n * 256 : lds + G::DUMMY0);
        }
    };
    // after this wave's DMA has landed: cells that straddle the end of a ragged row hold x[L-4 .. L-1]; shift them to
    // x[pos .. L-1] followed by zeros.  Only the last time block of a clip, and only when L % 4 != 0.
    const bool ragged = (p.Lvalid & 3) != 0;
    // states of the chunks whose DMA may still be in flight (newest last): what fix_ragged needs once a chunk has landed
    bool q_live[QN];
    int q_in0a[QN], q_irow[QN];
    float *q_slot[QN];
#pragma unroll
    for (int i = 0; i < QN; ++i) q_live[i] = false, q_in0a[i] = 0, q_irow[i] = 0, q_slot[i] = lds;
    auto push_state = [&]() {
#pragma unroll
        for (int i = 0; i + 1 < QN; ++i) q_live[i] = q_live[i + 1], q_in0a[i] = q_in0a[i + 1], q_irow[i] = q_irow[i + 1], q_slot[i] = q_slot[i + 1];
        q_live[QN - 1] = d_live, q_in0a[QN - 1] = d_in0a, q_irow[QN - 1] = d_irow, q_slot[QN - 1] = d_slot;
    };
    auto fix_ragged = [&](int qi) {
        if (!ragged || !q_live[qi] || q_in0a[qi] + segf <= p.Lvalid) return;
#pragma unroll
        for (int r = 0; r < G::RB; ++r) {
            const int n = wave + 4 * r;
            const int pos = q_in0a[qi] + colB[r];
            const int over = pos + 4 - p.Lvalid;
            const bool rowok = !D2 || unsigned(q_irow[qi] + rowB[D2 ? r : 0]) < unsigned(p.Tin);
            if (n < G::NIB && rowok && pos >= 0 && pos < p.Lvalid && over > 0) {
                f32x4 *cell = reinterpret_cast<f32x4 *>(q_slot[qi] + AFL + n * 256 + lane * 4);
                const f32x4 v = *cell;
                f32x4 o;
                o[0] = over == 1 ? v[1] : (over == 2 ? v[2] : v[3]);
                o[1] = over == 1 ? v[2] : (over == 2 ? v[3] : 0.f);
                o[2] = over == 1 ? v[3] : 0.f;
                o[3] = 0.f;
                *cell = o;
            }
        }
    };
    auto issue_all = [&]() {
        begin_chunk();
#pragma unroll
        for (int k = 0; k < G::NOPS; ++k) dma_op(k);
    };

    for (int i = tid; i < G::NBIAS; i += 256) lds[G::BIAS0 + i] = (bias && i < p.Cout) ? bias[i] : 0.f;
    // prologue: NSLOT - 1 chunks requested; chunk 0 (and 1: its first operands are read below) must have landed
    issue_all();
    push_state();
    if (PRE3) {
#pragma unroll
        for (int c = 1; c < G::NSLOT - 1; ++c) {
            issue_all();
            push_state();
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * G::NOPS) : "memory");
    if (PRE3) fix_ragged(QN - 2 - DEPTH);
    fix_ragged(QN - 1 - DEPTH);
    __syncthreads();

    CpFrag<MW, NW, KS> f[2];
    if (PRE3) cp_load_frag<G>(f[0], lds + aLane, lds + bLane, 0, bk);

    const bool pre_act = (p.epilogue & AGX_EPI_LEAKY_PRE) != 0;
    const bool gelu_act = (p.epilogue & AGX_EPI_GELU_PRE) != 0, post_act = (p.epilogue & AGX_EPI_LEAKY_POST) != 0;
    CpTileCur cc = first;
    int qs = 0;   // ring slot of the chunk being consumed
    f32x16 acc[MW][NW];

    // One interval = the J tap phases of one chunk; PAR = which operand set holds tap 0.
    auto chunk = [&](auto par_c, bool tail_chunk) {
        constexpr int PAR = decltype(par_c)::value;
        begin_chunk();
        const int qsn = qs + 1 == G::NSLOT ? 0 : qs + 1;
        const float *As = lds + qs * SLOT + aLane, *Bs = lds + qs * SLOT + bLane;
        const float *An = lds + qsn * SLOT + aLane, *Bn = lds + qsn * SLOT + bLane;
        cp_static_for<J>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            CpFrag<MW, NW, KS> &cur = f[(PAR + j) & 1], &nxt = f[(PAR + j + 1) & 1];
            if (j + 1 < J) cp_load_frag<G>(nxt, As, Bs, j + 1, bk);
            else if (PRE3) cp_load_frag<G>(nxt, An, Bn, 0, bk);   // next chunk's first phase: complete since the last barrier
#pragma unroll
            for (int o = 0; o < G::OPP; ++o) dma_op(j * G::OPP + o);
            cp_mfma_frag<G>(acc, cur);
            if (j + 1 < J || PRE3) cp_interleave<G>();
            else __builtin_amdgcn_sched_barrier(0);
        });
        push_state();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * G::NOPS) : "memory");   // this wave's part of the chunk requested DEPTH intervals ago has landed
        fix_ragged(QN - 1 - DEPTH);
        __syncthreads();                                   // everyone's has; the consumed slot is free
        if (!PRE3 && !tail_chunk) cp_load_frag<G>(f[(PAR + J) & 1], An, Bn, 0, bk);
        qs = qsn;
    };

    for (int k = 0; k < my_tiles; ++k) {
        const int mb = cc.mb, nb = cc.nb, b = cc.b;
        cc.advance(step_m, step_n, step_b, mblocks, nblocks);
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int kk = 0; kk < NW; ++kk)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][kk][r] = 0.f;
        if (!PRE3) cp_load_frag<G>(f[0], lds + qs * SLOT + aLane, lds + qs * SLOT + bLane, 0, bk);

        if (J % 2 == 0) {
            for (int c = 0; c < nch - 1; ++c) chunk(std::integral_constant<int, 0>{}, false);
            chunk(std::integral_constant<int, 0>{}, true);
        } else {   // odd tap count: the operand sets swap roles every chunk (nch is even for these layers)
            for (int c = 0; c < nch - 2; c += 2) {
                chunk(std::integral_constant<int, 0>{}, false);
                chunk(std::integral_constant<int, 1>{}, false);
            }
            chunk(std::integral_constant<int, 0>{}, false);
            chunk(std::integral_constant<int, 1>{}, true);
        }

        // ---- epilogue: bias (LDS), activation, store ------------------------------------------------------------------
        int loutv = D2 ? int(p.y_cstride) : p.Lout;   // opaque per-tile copy: the row offsets are formed here, not hoisted above the main loop
        asm volatile("" : "+v"(loutv));
        const int mrow0 = mb * BM + wm * 32 * MW;
        size_t ybase = size_t(b) * p.Cout * p.Lout;
        int trow = 0;   // D2: base row of the tile
        if (D2) {
            const int bq = b / rbs;
            trow = (b - bq * rbs) * RT;
            ybase = size_t(bq) * p.Cout * p.y_cstride;
        }
        char *yb = reinterpret_cast<char *>(y + ybase);
        const char *ab = reinterpret_cast<const char *>(add2 + ybase), *kb = reinterpret_cast<const char *>(mask2 + ybase);
#pragma unroll
        for (int i = 0; i < MW; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m4 = mrow0 + i * 32 + 8 * g + 4 * lh;   // first of this lane's 4 consecutive rows
                if constexpr (D2) {   // Conv2d: tile rows x columns; backward-data of a strided layer: phases scatter to rows / columns
                    // the loads of a pair of rows (gradient add, mask) are issued on clamped offsets before any use
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        unsigned off[2][NW];
                        bool okv[2][NW];
                        float rv[2][NW], mv[2][NW], bv[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int m = m4 + 2 * h2 + e;
                            const int co = m / (Q * G::QH), a = (m / Q) % G::QH, c = m % Q;
                            bv[e] = lds[G::BIAS0 + co];
#pragma unroll
                            for (int kk = 0; kk < NW; ++kk) {
                                const int n = wn * 32 * NW + kk * 32 + li;
                                const int brow = trow + (n >> wfs), t = nb * WF + (n & (WF - 1));   // base row / column
                                const int orow = G::QH * brow + a - p.oshift_h, ocol = Q * t + c - p.oshift;
                                okv[e][kk] = brow < p.Tt && t < p.Lt && orow >= 0 && orow < p.Tout && ocol >= 0 && ocol < p.Lout;
                                off[e][kk] = okv[e][kk] ? unsigned(co * loutv + orow * p.Lout + ocol) * 4u : 0u;
                            }
                        }
                        if (add2) {
#pragma unroll
                            for (int e = 0; e < 2; ++e)
#pragma unroll
                                for (int kk = 0; kk < NW; ++kk) rv[e][kk] = *reinterpret_cast<const float *>(ab + off[e][kk]);
                        }
                        if (mask2) {
#pragma unroll
                            for (int e = 0; e < 2; ++e)
#pragma unroll
                                for (int kk = 0; kk < NW; ++kk) mv[e][kk] = *reinterpret_cast<const float *>(kb + off[e][kk]);
                        }
#pragma unroll
                        for (int e = 0; e < 2; ++e)
#pragma unroll
                            for (int kk = 0; kk < NW; ++kk) {
                                float v = acc[i][kk][4 * g + 2 * h2 + e] + bv[e];
                                if (pre_act) v = leaky(v, p.slope);
                                if (add2) v += rv[e][kk];
                                if (mask2) v = mv[e][kk] > 0.f ? v : v * p.slope;
                                if (okv[e][kk]) *reinterpret_cast<float *>(yb + off[e][kk]) = v;
                            }
                    }
                } else if (Q == 1) {
                    // 1-D layers with one output phase also carry the transformer block's epilogues (transformers.py:157-223: exact
                    // GELU behind FFN-in, the residual add behind W_o / FFN-out) and the unfused residual block's (vae.py:113-117):
                    // v = post( res + gelu|leaky( acc + bias ) ).  The residual values of a row group are requested before its
                    // first store (vmcnt counts loads and stores in order: a load between two stores waits for the store).
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(lds + G::BIAS0 + m4);
                    float rv[NW][4];
                    if (add2) {
#pragma unroll
                        for (int kk = 0; kk < NW; ++kk) {
                            const int tc = min(nb * BN + wn * 32 * NW + kk * 32 + li, p.Lt - 1);
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                rv[kk][s4] = *reinterpret_cast<const float *>(ab + unsigned((m4 + s4) * loutv + tc) * 4u);
                        }
                    }
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const int t = nb * BN + wn * 32 * NW + kk * 32 + li;
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            float v = acc[i][kk][4 * g + s4] + bq[s4];
                            if (pre_act) v = leaky(v, p.slope);
                            if (gelu_act) v = gelu_erf(v);
                            if (add2) v += rv[kk][s4];
                            if (post_act) v = leaky(v, p.slope);
                            if (t < p.Lt) *reinterpret_cast<float *>(yb + unsigned((m4 + s4) * loutv + t) * 4u) = v;
                        }
                    }
                } else if (Q % 4 == 0) {   // the 4 rows are 4 consecutive output phases of one channel: one 16-byte store
                    const int co = m4 / Q, ph = m4 % Q;
                    const float bv = lds[G::BIAS0 + co];
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const int t = nb * BN + wn * 32 * NW + kk * 32 + li;
                        f32x4 v4;
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            float v = acc[i][kk][4 * g + s4] + bv;
                            if (pre_act) v = leaky(v, p.slope);
                            v4[s4] = v;
                        }
                        if (t < p.Lt) *reinterpret_cast<f32x4 *>(yb + unsigned(co * loutv + Q * t + ph) * 4u) = v4;
                    }
                } else if (Q == 2) {       // rows (0,1) and (2,3): two channels x two phases -> two 8-byte stores
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    const int co = m4 / 2;
                    const f32x2_t b2 = *reinterpret_cast<const f32x2_t *>(lds + G::BIAS0 + co);
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const int t = nb * BN + wn * 32 * NW + kk * 32 + li;
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) {
                            f32x2_t v2;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                float v = acc[i][kk][4 * g + 2 * hp + e] + b2[hp];
                                if (pre_act) v = leaky(v, p.slope);
                                v2[e] = v;
                            }
                            if (t < p.Lt) *reinterpret_cast<f32x2_t *>(yb + unsigned((co + hp) * loutv + 2 * t) * 4u) = v2;
                        }
                    }
                } else {                   // any Q: one dword per element
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int m = m4 + s4, co = m / Q, ph = m - co * Q;
                        const float bv = lds[G::BIAS0 + co];
#pragma unroll
                        for (int kk = 0; kk < NW; ++kk) {
                            const int t = nb * BN + wn * 32 * NW + kk * 32 + li;
                            float v = acc[i][kk][4 * g + s4] + bv;
                            if (pre_act) v = leaky(v, p.slope);
                            if (t < p.Lt) *reinterpret_cast<float *>(yb + unsigned(co * loutv + Q * t + ph) * 4u) = v;
                        }
                    }
                }
            }
        }
    }
}

template <class G>
static int launch_cp(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res, float *y,
                     hipStream_t st) {
    auto kern = conv_p_kernel<G>;
    static DeviceOnce once;
    int n_cu = 0;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, &n_cu, "conv_p")) return rc;
    static_assert(G::LDS_BYTES <= 160 * 1024, "ring does not fit LDS");
    const int mblocks = p.M / G::BM, nblocks = ceil_div(p.Lt, G::BN);
    const int64_t ntiles64 = int64_t(mblocks) * nblocks * p.B;
    if (ntiles64 > (1 << 30)) return fail(AGX_ERR_BAD_SHAPE, "conv_p: too many tiles");
    const int ntiles = int(ntiles64);
    const int wg_per_cu = int((160 * 1024) / G::LDS_BYTES) >= 2 ? 2 : 1;
    int grid = n_cu * wg_per_cu;
    if (grid > ntiles) grid = ntiles;
    // grid = sb * (mblocks * nblocks) + sn * mblocks + sm
    const int per_clip = mblocks * nblocks;
    const int sb = grid / per_clip, rem = grid % per_clip;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, st, p, mblocks, nblocks, ntiles, rem % mblocks,
                       rem / mblocks, sb, x, wp + p.tile_off, bias, y, (p.epilogue & AGX_EPI_RESIDUAL) ? res : nullptr,
                       static_cast<const float *>(nullptr), 0);
    return check_launch("conv_p");
}

// columns per tile row: the smallest power of two >= min(lt, BN), at least 32 (the rest of the BN columns are further rows)
static int cp2d_wf_shift(int lt, int bn) {
    int wf = bn;
    while (wf / 2 >= 32 && wf / 2 >= lt) wf /= 2;
    int s = 0;
    while ((1 << s) < wf) ++s;
    return s;
}

// Backward-data of a column-strided Conv2d on the ring: the base grid f' in [0, Lt) of the phase GEMM has one position
// more than fits whole blocks (Lt = W / sw + 1: 513, 257, ...), and its first position only produces the output columns
// [0, Q - oshift).  The ring runs f' = 1 .. Lt - 1 (exactly W / sw positions: whole blocks) and this kernel the first
// columns: dx[ci, QH t' + a - oshift_h, c - oshift] = sum_{co, jh} Wt[(jh, co)][Jw - 1][m] dy[co, t' - ph + jh, 0],
// m = (ci QH + a) Q + c, c >= oshift.  One workgroup per (clip, base row); the dy column sits in LDS.
__global__ __launch_bounds__(256) void conv2d_bwd_first_cols_kernel(ConvPlan p, const float *__restrict__ dy,
                                                                    const float *__restrict__ timg,
                                                                    const float *__restrict__ add,
                                                                    const float *__restrict__ mask, float *__restrict__ dx) {
    extern __shared__ float dcol[];   // [kh][Cin]
    const int bq = blockIdx.x / p.Tt, trow = blockIdx.x - bq * p.Tt;
    const int Jw = p.J / p.kh, nv = p.kh * p.Cin;
    for (int v = threadIdx.x; v < nv; v += 256) {
        const int jh = v / p.Cin, co = v - jh * p.Cin;
        const int r = trow * p.sh - p.ph + jh;
        dcol[v] = (r >= 0 && r < p.Tin) ? dy[(size_t(bq) * p.Cin + co) * p.x_cstride + size_t(r) * p.Lin] : 0.f;
    }
    __syncthreads();
    const int ncol = p.q - p.oshift;   // output columns 0 .. ncol - 1
    const int rows = (p.M / p.q) * ncol;
    for (int e = threadIdx.x; e < rows; e += 256) {
        const int mq = e / ncol, c = p.oshift + (e - mq * ncol);
        const int m = mq * p.q + c;
        const int ci = mq / p.qh, a = mq - ci * p.qh;
        const int orow = p.qh * trow + a - p.oshift_h;
        if (orow < 0 || orow >= p.Tout) continue;
        float acc = 0.f;
        for (int v4 = 0; v4 < nv / 4; ++v4) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(timg + (size_t(v4) * Jw + (Jw - 1)) * p.M * 4 + size_t(m) * 4);
            acc = fmaf(w[0], dcol[4 * v4], acc);
            acc = fmaf(w[1], dcol[4 * v4 + 1], acc);
            acc = fmaf(w[2], dcol[4 * v4 + 2], acc);
            acc = fmaf(w[3], dcol[4 * v4 + 3], acc);
        }
        const size_t o = (size_t(bq) * p.Cout + ci) * p.y_cstride + size_t(orow) * p.Lout + (c - p.oshift);
        if (add) acc += add[o];
        if (mask) acc = mask[o] > 0.f ? acc : acc * p.slope;
        dx[o] = acc;
    }
}

// Conv2d layers (D2 geometries): tiles = row blocks x column blocks x (clip, base row)
template <class G>
static int launch_cp2d(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *add,
                       float *y, hipStream_t st) {
    auto kern = conv_p_kernel<G>;
    static DeviceOnce once;
    int n_cu = 0;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, &n_cu, "conv_p")) return rc;
    static_assert(G::LDS_BYTES <= 160 * 1024, "ring does not fit LDS");
    const float *mask = (p.epilogue & AGX_EPI_MASK) ? p.mask : nullptr;
    ConvPlan pp = p;
    if (G::Q > 1) {   // column phases: the ring takes base positions 1 .. Lt - 1, conv2d_bwd_first_cols_kernel position 0
        pp.Lt = p.Lt - 1;
        pp.oshift = p.oshift - G::Q;
        if (int64_t(p.B) * p.Tt > (int64_t(1) << 30)) return fail(AGX_ERR_BAD_SHAPE, "conv_p2d: grid too large");
        hipLaunchKernelGGL(conv2d_bwd_first_cols_kernel, dim3(p.B * p.Tt), dim3(256), size_t(p.kh) * p.Cin * sizeof(float), st, p,
                           x, wp + p.tile_off, add, mask, y);
    }
    const int wfs = cp2d_wf_shift(pp.Lt, G::BN);   // tile = (BN >> wfs) rows x (1 << wfs) columns
    const int mblocks = pp.M / G::BM, nblocks = ceil_div(pp.Lt, 1 << wfs);
    const int64_t ntiles64 = int64_t(mblocks) * nblocks * pp.B * ceil_div(pp.Tt, G::BN >> wfs);
    if (ntiles64 > (1 << 30)) return fail(AGX_ERR_BAD_SHAPE, "conv_p: too many tiles");
    const int ntiles = int(ntiles64);
    const int wg_per_cu = int((160 * 1024) / G::LDS_BYTES) >= 2 ? 2 : 1;
    int grid = n_cu * wg_per_cu;
    if (grid > ntiles) grid = ntiles;
    const int per_row = mblocks * nblocks;
    const int sb = grid / per_row, rem = grid % per_row;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, st, pp, mblocks, nblocks, ntiles, rem % mblocks,
                       rem / mblocks, sb, x, wp + p.tile_off, bias, y, add, mask, wfs);
    return check_launch("conv_p2d");
}

// ---- layer geometries the kernel is instantiated for -------------------------------------------------------------------
//                      MW NW WM WN CCH  J  S  Q  P NS
typedef CpGeom<2, 2, 1, 4, 8, 5, 2, 1, 3, 2> CpDown2;     // Conv1d k5 s2, M = 64:      64 x 256 tiles
typedef CpGeom<2, 2, 2, 2, 4, 7, 3, 1, 4, 2> CpDown3;     // Conv1d k7 s3 (the class-default strides (2, 3, 4, 4, 5), vae.py:215): 128 x 128
typedef CpGeom<2, 2, 2, 2, 4, 9, 4, 1, 5, 2> CpDown4;     // Conv1d k9 s4:             128 x 128
typedef CpGeom<2, 2, 2, 2, 4, 11, 5, 1, 6, 2> CpDown5;    // Conv1d k11 s5:            128 x 128
typedef CpGeom<2, 1, 2, 2, 4, 17, 8, 1, 9, 3> CpDown8;    // Conv1d k17 s8:            128 x 64 (one workgroup per CU)
typedef CpGeom<2, 1, 2, 2, 16, 3, 1, 1, 2, 2> CpK3;       // Conv1d k3 s1:             128 x 64
typedef CpGeom<2, 1, 2, 2, 8, 7, 1, 1, 6, 2> CpK7;        // Conv1d k7 s1 / ConvT k7 s1 (flipped kernel): 128 x 64
typedef CpGeom<2, 1, 2, 2, 16, 3, 1, 8, 1, 2> CpUp8;      // upsample x8 (J = 3):      128 x 64
typedef CpGeom<2, 2, 2, 2, 8, 3, 1, 5, 1, 2> CpUp5;       // upsample x5:              128 x 128
typedef CpGeom<2, 2, 2, 2, 8, 3, 1, 4, 1, 2> CpUp4;       // upsample x4:              128 x 128
typedef CpGeom<2, 2, 1, 4, 8, 3, 1, 3, 1, 2> CpUp3;       // upsample x3 (M = 3 Cout = 192 for 128 -> 64): 64 x 256
typedef CpGeom<2, 2, 1, 4, 8, 3, 1, 2, 1, 2> CpUp2;       // upsample x2, M = 64:       64 x 256
// round 4 (configs 3 / 4 off the first-round kernels):
typedef CpGeom<2, 2, 2, 2, 16, 1, 1, 1, 0, 4, false, 1, 2048> CpK1;       // k = 1 (every Linear of the transformer block, transformers.py:157-223; the
                                                          // unfused block's second conv): 128 x 128 (12 KB of operands per 16-MFMA phase on 128 x 64 tiles ran
                                                          // no faster than 128 x 64: 75 TFLOP/s either way), one phase per chunk -> 4 slots, one chunk in flight across the barrier
typedef CpGeom<2, 2, 2, 2, 4, 11, 1, 1, 5, 2> CpSame11;   // Conv1d(K = 11, padding="same") -- WaveletLayer's first conv at stride 5 (wavelets.py:193-201): 128 x 128
typedef CpGeom<2, 1, 2, 2, 16, 3, 1, 1, 1, 2> CpSame3;    // Conv1d(K = 3, padding="same") -- WaveletLayer's last conv: 128 x 64

// Conv2d (row-folded; the kernel's row count / row stride / row padding are run-time):
//                      MW NW WM WN CCH  J  S  Q  P NS  D2
typedef CpGeom<2, 2, 2, 2, 8, 3, 1, 1, 1, 2, true> Cp2dK3;         // kh x 3 kernels, column stride 1, pad 1:  128 x 128
typedef CpGeom<2, 2, 1, 4, 8, 3, 1, 1, 1, 2, true> Cp2dK3M64;      // ... 64 output rows:                         64 x 256
typedef CpGeom<1, 4, 1, 4, 8, 3, 1, 1, 1, 2, true> Cp2dK3M32;      // ... 32 output rows (= M):                   32 x 512
typedef CpGeom<2, 2, 2, 2, 8, 4, 2, 1, 1, 2, true> Cp2dK4S2;       // kh x 4 kernels, column stride 2, pad 1:  128 x 128
typedef CpGeom<2, 2, 1, 4, 8, 4, 2, 1, 1, 2, true> Cp2dK4S2M64;    // ... 64 output rows:                         64 x 256

// backward-data of the column-stride-2 layers: 2 column taps over dy, Q = 2 column phases, QH row phases
typedef CpGeom<2, 2, 2, 2, 16, 2, 1, 2, 0, 2, true, 1> Cp2dB2;      // (kh x 4, stride (1, 2)):   128 x 128
typedef CpGeom<2, 2, 1, 4, 8, 2, 1, 2, 0, 2, true, 1> Cp2dB2M64;    // ... M = 64:                  64 x 256
typedef CpGeom<2, 2, 2, 2, 16, 2, 1, 2, 0, 2, true, 2> Cp2dB2H2;    // (4 x 4, stride (2, 2)):    128 x 128

enum { CP2D_NONE = 0, CP2D_K3, CP2D_K3M64, CP2D_K3M32, CP2D_K4S2, CP2D_K4S2M64, CP2D_B2, CP2D_B2M64, CP2D_B2H2 };

// weights-only part of the test: the packed image of a layer must not depend on the size of the feature map it is
// later applied to (discriminator.py packs with a nominal size)
template <class G>
static bool cp2d_fits(const ConvPlan &p) {
    const int ncc = p.Cin / G::CCH;
    const int nch = ncc * p.kh;
    return p.M % G::BM == 0 && p.Cin % G::CCH == 0 && p.Cin == p.cin_real && ncc >= 1 && nch >= 2 &&
           (G::J % 2 == 0 || nch % 2 == 0) && p.Cout <= G::NBIAS;
}

// patch-mode plan of conv2d.hip (forward, or backward-data of a stride-1 layer) -> ring geometry; depends on the
// layer (channels, kernel, strides, padding) only: decides whether the packed image carries a tile image
int conv_p2d_geometry(const ConvPlan &p) {
    if (p.pm_R == 0 || p.prec != 0 || p.G != 1 || p.d != 1 || p.kh <= 0 || p.J % p.kh != 0) return CP2D_NONE;
    const int kw = p.J / p.kh;
    if (p.q == 2 && kw == 2 && p.s == 1 && p.sh == 1 && p.P == 1 && p.oshift == 1) {   // backward-data, column stride 2, pad 1
        if (p.qh == 1 && p.oshift_h == 0) {
            if (p.M == 64 && cp2d_fits<Cp2dB2M64>(p)) return CP2D_B2M64;
            if (cp2d_fits<Cp2dB2>(p)) return CP2D_B2;
        }
        if (p.qh == 2 && cp2d_fits<Cp2dB2H2>(p)) return CP2D_B2H2;
        return CP2D_NONE;
    }
    if (p.q != 1 || p.qh != 1 || p.oshift != 0 || p.oshift_h != 0) return CP2D_NONE;
    if (kw == 3 && p.s == 1 && p.P == 1) {
        if (p.M == 32 && cp2d_fits<Cp2dK3M32>(p)) return CP2D_K3M32;
        if (p.M == 64 && cp2d_fits<Cp2dK3M64>(p)) return CP2D_K3M64;
        if (cp2d_fits<Cp2dK3>(p)) return CP2D_K3;
    }
    if (kw == 4 && p.s == 2 && p.P == 1) {
        if (p.M == 64 && cp2d_fits<Cp2dK4S2M64>(p)) return CP2D_K4S2M64;
        if (cp2d_fits<Cp2dK4S2>(p)) return CP2D_K4S2;
    }
    return CP2D_NONE;
}

// can THIS call run on the ring kernel?  (feature-map size: column blocks at least 70 % full -- narrow maps stay on the
// patch tiles --, 32-bit DMA / epilogue offsets; epilogue: bias, LeakyReLU, gradient add, LeakyReLU-gradient mask)
bool conv_p2d_supported(const ConvPlan &p) {
    const int g = conv_p2d_geometry(p);
    if (p.tile_off < 0 || g == CP2D_NONE) return false;
    if ((p.epilogue & ~(AGX_EPI_LEAKY_PRE | AGX_EPI_RESIDUAL | AGX_EPI_MASK)) != 0) return false;
    if (p.Lvalid != p.Lin || p.Lin < 4) return false;
    if (p.q == 1 && (p.Lt != p.Lout || p.Tt != p.Tout)) return false;
    if (int64_t(16) * p.x_cstride * 4 >= (int64_t(1) << 31)) return false;
    if (int64_t(p.Cout) * p.y_cstride * 4 >= (int64_t(1) << 32)) return false;
    const int bn = g == CP2D_K3M32 ? 512 : ((g == CP2D_K3M64 || g == CP2D_K4S2M64 || g == CP2D_B2M64) ? 256 : 128);
    const int lt = p.q == 1 ? p.Lt : p.Lt - 1;   // column phases: the ring runs base positions 1 .. Lt - 1
    if (lt < 1) return false;
    const int wf = 1 << cp2d_wf_shift(lt, bn);   // narrow maps: several rows per tile
    return 10 * int64_t(lt) >= 7 * int64_t(ceil_div(lt, wf)) * wf;
}

const char *conv_p2d_variant(const ConvPlan &p) {
    switch (conv_p2d_geometry(p)) {
        case CP2D_K3: return "conv_p2d<k3,128x128>";
        case CP2D_K3M64: return "conv_p2d<k3,64x256>";
        case CP2D_K3M32: return "conv_p2d<k3,32x512>";
        case CP2D_K4S2: return "conv_p2d<k4s2,128x128>";
        case CP2D_K4S2M64: return "conv_p2d<k4s2,64x256>";
        case CP2D_B2: return "conv_p2d<bwd s(1,2),128x128>";
        case CP2D_B2M64: return "conv_p2d<bwd s(1,2),64x256>";
        case CP2D_B2H2: return "conv_p2d<bwd s(2,2),128x128>";
        default: return "conv_p2d<unsupported>";
    }
}

// res = the tensor added in the epilogue (AGX_EPI_RESIDUAL; backward-data: the gradient arriving at this feature map)
int launch_conv_p2d(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res, float *y,
                    hipStream_t st) {
    if (!conv_p2d_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "conv_p2d: unsupported layer");
    const float *add = (p.epilogue & AGX_EPI_RESIDUAL) ? res : nullptr;
    switch (conv_p2d_geometry(p)) {
        case CP2D_K3: return launch_cp2d<Cp2dK3>(p, x, wp, bias, add, y, st);
        case CP2D_K3M64: return launch_cp2d<Cp2dK3M64>(p, x, wp, bias, add, y, st);
        case CP2D_K3M32: return launch_cp2d<Cp2dK3M32>(p, x, wp, bias, add, y, st);
        case CP2D_K4S2: return launch_cp2d<Cp2dK4S2>(p, x, wp, bias, add, y, st);
        case CP2D_K4S2M64: return launch_cp2d<Cp2dK4S2M64>(p, x, wp, bias, add, y, st);
        case CP2D_B2: return launch_cp2d<Cp2dB2>(p, x, wp, bias, add, y, st);
        case CP2D_B2M64: return launch_cp2d<Cp2dB2M64>(p, x, wp, bias, add, y, st);
        case CP2D_B2H2: return launch_cp2d<Cp2dB2H2>(p, x, wp, bias, add, y, st);
        default: return fail(AGX_ERR_UNSUPPORTED, "conv_p2d: unsupported layer");
    }
}

enum { CP_NONE = 0, CP_DOWN2, CP_DOWN4, CP_DOWN5, CP_DOWN8, CP_K3, CP_K7, CP_UP8, CP_UP5, CP_UP4, CP_UP2, CP_DOWN3, CP_UP3, CP_K1, CP_SAME11,
       CP_SAME3 };

template <class G>
static bool cp_fits(const ConvPlan &p) {
    return p.M % G::BM == 0 && p.Cin % G::CCH == 0 && (G::J % 2 == 0 || (p.Cin / G::CCH) % 2 == 0) && p.Cin / G::CCH >= 2 &&
           p.Cout <= G::NBIAS;
}

// shape-only test (also decides whether agx_conv_pack appends a tile image: common.hpp)
int conv_p_geometry(const ConvPlan &p) {
    if (p.prec != 0 || p.G != 1 || p.d != 1 || p.kh != 1 || p.Tout != 1 || p.pm_R != 0) return CP_NONE;
    const int J = p.J, S = p.s, Q = p.q, P = p.P;
    if (Q == 1 && J == 5 && S == 2 && P == 3 && p.M == 64 && cp_fits<CpDown2>(p)) return CP_DOWN2;
    if (Q == 1 && J == 7 && S == 3 && P == 4 && cp_fits<CpDown3>(p)) return CP_DOWN3;
    if (Q == 1 && J == 9 && S == 4 && P == 5 && cp_fits<CpDown4>(p)) return CP_DOWN4;
    if (Q == 1 && J == 11 && S == 5 && P == 6 && cp_fits<CpDown5>(p)) return CP_DOWN5;
    if (Q == 1 && J == 17 && S == 8 && P == 9 && cp_fits<CpDown8>(p)) return CP_DOWN8;
    if (Q == 1 && J == 3 && S == 1 && P == 2 && cp_fits<CpK3>(p)) return CP_K3;
    if (Q == 1 && J == 7 && S == 1 && P == 6 && cp_fits<CpK7>(p)) return CP_K7;
    if (Q == 8 && J == 3 && S == 1 && P == 1 && cp_fits<CpUp8>(p)) return CP_UP8;
    if (Q == 5 && J == 3 && S == 1 && P == 1 && cp_fits<CpUp5>(p)) return CP_UP5;
    if (Q == 4 && J == 3 && S == 1 && P == 1 && cp_fits<CpUp4>(p)) return CP_UP4;
    if (Q == 3 && J == 3 && S == 1 && P == 1 && cp_fits<CpUp3>(p)) return CP_UP3;
    if (Q == 2 && J == 3 && S == 1 && P == 1 && p.M == 64 && cp_fits<CpUp2>(p)) return CP_UP2;
    if (Q == 1 && J == 1 && S == 1 && P == 0 && cp_fits<CpK1>(p)) return CP_K1;
    if (Q == 1 && J == 11 && S == 1 && P == 5 && cp_fits<CpSame11>(p)) return CP_SAME11;
    if (Q == 1 && J == 3 && S == 1 && P == 1 && cp_fits<CpSame3>(p)) return CP_SAME3;
    return CP_NONE;
}

// can THIS call run on the ring kernel? (epilogue: bias + optional LeakyReLU; one-phase layers also GELU / residual / the
// activation behind the residual)
bool conv_p_supported(const ConvPlan &p) {
    if (p.tile_off < 0 || conv_p_geometry(p) == CP_NONE) return false;
    const int epi_ok = p.q == 1 ? (AGX_EPI_LEAKY_PRE | AGX_EPI_RESIDUAL | AGX_EPI_LEAKY_POST | AGX_EPI_GELU_PRE) : AGX_EPI_LEAKY_PRE;
    if ((p.epilogue & ~epi_ok) != 0 || p.oshift != 0 || p.mask != nullptr) return false;
    if (p.Lvalid != p.Lin || p.Lin < 4 || p.Lout != p.q * p.Lt) return false;   // (ragged L: fix_ragged)
    if (p.q > 1 && p.q % 4 == 0 && (p.Lout % 4 != 0)) return false;
    // 32-bit byte offsets inside a clip (DMA cells: row * Lin + 4 * col; stores: row * Lout + t): very long clips fall back
    // to conv_mfma, whose addressing is 64-bit
    if (int64_t(16) * p.Lin * 4 >= (int64_t(1) << 31)) return false;
    if (int64_t(p.Cout) * p.Lout * 4 >= (int64_t(1) << 32)) return false;
    return true;
}

const char *conv_p_variant(const ConvPlan &p) {
    switch (conv_p_geometry(p)) {
        case CP_DOWN2: return "conv_p<down2,64x256>";
        case CP_DOWN4: return "conv_p<down4,128x128>";
        case CP_DOWN5: return "conv_p<down5,128x128>";
        case CP_DOWN8: return "conv_p<down8,128x64>";
        case CP_K3: return "conv_p<k3,128x64>";
        case CP_K7: return "conv_p<k7,128x64>";
        case CP_UP8: return "conv_p<up8,128x64>";
        case CP_UP5: return "conv_p<up5,128x128>";
        case CP_UP4: return "conv_p<up4,128x128>";
        case CP_UP2: return "conv_p<up2,64x256>";
        case CP_DOWN3: return "conv_p<down3,128x128>";
        case CP_UP3: return "conv_p<up3,64x256>";
        case CP_K1: return "conv_p<k1,128x128>";
        case CP_SAME11: return "conv_p<same11,128x128>";
        case CP_SAME3: return "conv_p<same3,128x64>";
        default: return "conv_p<unsupported>";
    }
}

int launch_conv_p(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res, float *y,
                  hipStream_t st) {
    if (!conv_p_supported(p)) return fail(AGX_ERR_UNSUPPORTED, "conv_p: unsupported layer");
    switch (conv_p_geometry(p)) {
        case CP_DOWN2: return launch_cp<CpDown2>(p, x, wp, bias, res, y, st);
        case CP_DOWN4: return launch_cp<CpDown4>(p, x, wp, bias, res, y, st);
        case CP_DOWN5: return launch_cp<CpDown5>(p, x, wp, bias, res, y, st);
        case CP_DOWN8: return launch_cp<CpDown8>(p, x, wp, bias, res, y, st);
        case CP_K3: return launch_cp<CpK3>(p, x, wp, bias, res, y, st);
        case CP_K7: return launch_cp<CpK7>(p, x, wp, bias, res, y, st);
        case CP_UP8: return launch_cp<CpUp8>(p, x, wp, bias, res, y, st);
        case CP_UP5: return launch_cp<CpUp5>(p, x, wp, bias, res, y, st);
        case CP_UP4: return launch_cp<CpUp4>(p, x, wp, bias, res, y, st);
        case CP_UP2: return launch_cp<CpUp2>(p, x, wp, bias, res, y, st);
        case CP_DOWN3: return launch_cp<CpDown3>(p, x, wp, bias, res, y, st);
        case CP_UP3: return launch_cp<CpUp3>(p, x, wp, bias, res, y, st);
        case CP_K1: return launch_cp<CpK1>(p, x, wp, bias, res, y, st);
        case CP_SAME11: return launch_cp<CpSame11>(p, x, wp, bias, res, y, st);
        case CP_SAME3: return launch_cp<CpSame3>(p, x, wp, bias, res, y, st);
        default: return fail(AGX_ERR_UNSUPPORTED, "conv_p: unsupported layer");
    }
}

}  // namespace agx
