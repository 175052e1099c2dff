// Diagnostic (not part of libagx): what does the MFMA step of resblock_b3 -- 12 x v_mfma_f32_32x32x16_bf16 on two
// accumulators (six products, operands rotating over three planes) -- sustain on THIS device: bare, with the B fragments
// read from LDS one step ahead (3 x ds_read_b128 per step), and with the A fragments re-read every 4 steps as well?
//   hipcc --offload-arch=gfx950 -O3 tools/b3_peak.hip -o tools/b3_peak_bin && tools/b3_peak_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b3x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: registers only, 1: B from LDS (prefetched mid-step), 2: B and A from LDS
__global__ __launch_bounds__(256, 2) void step_loop(const float *in, float *out, unsigned long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[48 * 1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 12 * 1024; i += 256) reinterpret_cast<float *>(lds)[i] = in[(blockIdx.x * 4096 + i) & 0xffff] * 1e-3f;
    __syncthreads();
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    b3x8 fa[3][2], fb[2][3];
    const char *pa = lds + tid * 16, *pb = lds + 24 * 1024 + (tid & 63) * 16;
    for (int pl = 0; pl < 3; ++pl) {
        for (int i = 0; i < 2; ++i) fa[pl][i] = *reinterpret_cast<const b3x8 *>(pa + (pl * 2 + i) * 4096);
        fb[0][pl] = *reinterpret_cast<const b3x8 *>(pb + pl * 1024);
        fb[1][pl] = *reinterpret_cast<const b3x8 *>(pb + pl * 1024 + 4096);
    }
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sb = s & 1;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][i], fb[sb][PB[t]], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[sb ^ 1][pl] = *reinterpret_cast<const b3x8 *>(pb + pl * 1024 + ((it * 4 + s) & 7) * 512);
            }
            if (MODE >= 2 && s == 3) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[pl][i] = *reinterpret_cast<const b3x8 *>(pa + (pl * 2 + i) * 4096 + (it & 1) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 3; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][i], fb[sb][PB[t]], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// The same arithmetic on v_mfma_f32_16x16x32_bf16 (round 4; MI355X_MICROARCH "DVFS give-back" item 7: in bare bf16 loops the 16 x 16 x 32
// shape delivered ~1.15 x the FLOP/s of 32 x 32 x 16 at equal cycles per FLOP because the chip holds a higher clock for it).  One step =
// a 64-row x 32-column block at K = 32: 6 products x (4 row blocks x 2 column blocks) = 48 MFMAs of 16 cycles = the FLOPs of TWO steps of
// the loop above; A: 4 fragments per piece, B: 2 per piece (the same ds_read_b128 count per FLOP).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void step_loop16(const float *in, float *out, unsigned long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[48 * 1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 12 * 1024; i += 256) reinterpret_cast<float *>(lds)[i] = in[(blockIdx.x * 4096 + i) & 0xffff] * 1e-3f;
    __syncthreads();
    f32x4 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    b3x8 fa[3][4], fb[2][3][2];
    const char *pa = lds + tid * 16, *pb = lds + 24 * 1024 + (tid & 63) * 16;
    for (int pl = 0; pl < 3; ++pl) {
        for (int i = 0; i < 4; ++i) fa[pl][i] = *reinterpret_cast<const b3x8 *>(pa + ((pl * 4 + i) % 6) * 4096);
        for (int j = 0; j < 2; ++j) {
            fb[0][pl][j] = *reinterpret_cast<const b3x8 *>(pb + pl * 1024 + j * 2048);
            fb[1][pl][j] = *reinterpret_cast<const b3x8 *>(pb + pl * 1024 + 4096 + j * 2048);
        }
    }
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // two K = 32 steps = the four K = 16 steps of step_loop
            const int sb = s & 1;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA[t]][i], fb[sb][PB[t]][j], acc[i * 2 + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        fb[sb ^ 1][pl][j] = *reinterpret_cast<const b3x8 *>(pb + pl * 1024 + j * 2048 + ((it * 2 + s) & 7) * 512);
            }
            if (MODE >= 2 && s == 1) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[pl][i] = *reinterpret_cast<const b3x8 *>(pa + ((pl * 4 + i) % 6) * 4096 + (it & 1) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 3; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA[t]][i], fb[sb][PB[t]][j], acc[i * 2 + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 4; ++r) sum += acc[a][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// Composition (e) of DESIGN 4.13: K = 32 = TWO piece products of one 16-channel phase (lane quarter q >> 1 selects the piece through a per-lane
// plane offset, q & 1 the channel half) -- today's LDS images stay, but every piece is re-read per product pair: per 64 x 32 block and phase
// 3 pairs x (4 A + 2 B) = 18 ds_read_b128 for 24 MFMAs of 16 cycles (0.75 per MFMA against 0.25 in the chunk-pair composition above).
// Operands of the next pair are requested while the current pair's 8 MFMAs run (two register sets).
__global__ __launch_bounds__(256, 2) void step_loop16_pairs(const float *in, float *out, unsigned long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[48 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4;
    for (int i = tid; i < 12 * 1024; i += 256) reinterpret_cast<float *>(lds)[i] = in[(blockIdx.x * 4096 + i) & 0xffff] * 1e-3f;
    __syncthreads();
    f32x4 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    // planes of 2 KB: A image [plane 3][half 2][64 rows][16 B] = 6 planes, B image behind it; lane offsets: piece by q >> 1, half by q & 1
    const int PAa[3] = {1, 2, 1}, PAb[3] = {0, 0, 0}, PBa[3] = {1, 0, 0}, PBb[3] = {2, 1, 0};     // (m,h)(m,l) | (l,h)(h,m) | (m,h)(h,h)
    int offA[3], offB[3];
    for (int p = 0; p < 3; ++p) {
        offA[p] = (((q >> 1) ? PAb[p] : PAa[p]) * 2 + (q & 1)) * 1024 + (lane & 15) * 16;
        offB[p] = 24 * 1024 + (((q >> 1) ? PBb[p] : PBa[p]) * 2 + (q & 1)) * 512 + (lane & 15) * 16;
    }
    b3x8 fa[2][4], fb[2][2];
    auto load_pair = [&](int set, int p, int it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const b3x8 *>(lds + offA[p] + i * 256 + (it & 1) * 6144);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[set][j] = *reinterpret_cast<const b3x8 *>(lds + offB[p] + j * 256 + ((it & 7) * 3072));
    };
    load_pair(0, 0, 0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {          // four 16-channel phases = the FLOPs of the four K = 16 steps of step_loop
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int set = (s * 3 + p) & 1;
                load_pair(set ^ 1, (p + 1) % 3, it * 4 + s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][i], fb[set][j], acc[i * 2 + j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 4; ++r) sum += acc[a][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, bool S16 = false>
void run(const char *name, int wg_per_cu, const float *din, float *dout, unsigned long long *dclk) {
    const int grid = 256 * wg_per_cu, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = MODE == 3 ? step_loop16_pairs : (S16 ? step_loop16<MODE < 3 ? MODE : 2> : step_loop<MODE < 3 ? MODE : 2>);
    for (int warm = 0; warm < 2; ++warm) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, din, dout, dclk, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, din, dout, dclk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> clk(2 * grid);
    hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost);
    double ghz = 0, cyc = 0;
    for (int b = 0; b < grid; ++b) { ghz += double(clk[2 * b]) / double(clk[2 * b + 1]) * 0.1; cyc += double(clk[2 * b]); }
    ghz /= grid; cyc /= grid;
    const double n_mfma = double(iters) * (S16 ? 96 : 48);         // per wave (the same FLOPs either way: 16 x 16 x 32 is half a 32 x 32 x 16)
    const double flop = double(grid) * 4 * double(iters) * 48 * 2.0 * 32 * 32 * 16;
    printf("%-44s %s wg/cu %d: %8.3f ms  %7.1f bf16 TFLOP/s = %6.1f fp32-equivalent (x1/6)  %5.1f cycles per MFMA per wave  clock %.2f GHz\n",
           name, S16 ? "16x16x32" : "32x32x16", wg_per_cu, best, flop / best * 1e-9, flop / best * 1e-9 / 6, cyc / n_mfma, ghz);
}

int main() {
    float *din, *dout; unsigned long long *dclk;
    std::vector<float> h(65536);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 256 * 8 * 256 * 4); hipMalloc(&dclk, 256 * 8 * 16);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("registers only", 1, din, dout, dclk);
    run<0>("registers only", 2, din, dout, dclk);
    run<1>("B fragments from LDS, one step ahead", 1, din, dout, dclk);
    run<1>("B fragments from LDS, one step ahead", 2, din, dout, dclk);
    run<2>("B every step + A every 4 steps from LDS", 1, din, dout, dclk);
    run<2>("B every step + A every 4 steps from LDS", 2, din, dout, dclk);
    run<0, true>("registers only", 1, din, dout, dclk);
    run<0, true>("registers only", 2, din, dout, dclk);
    run<1, true>("B fragments from LDS, one step ahead", 1, din, dout, dclk);
    run<1, true>("B fragments from LDS, one step ahead", 2, din, dout, dclk);
    run<2, true>("B every step + A every 2 steps from LDS", 1, din, dout, dclk);
    run<2, true>("B every step + A every 2 steps from LDS", 2, din, dout, dclk);
    run<3, true>("product pairs: 18 ds_read_b128 per 24 MFMAs", 1, din, dout, dclk);
    run<3, true>("product pairs: 18 ds_read_b128 per 24 MFMAs", 2, din, dout, dclk);
    return 0;
}
