// Diagnostic (not part of libagx): what does a bare v_mfma_f32_32x32x2_f32 loop sustain on THIS
// device, and at what in-kernel clock?  Gives the measured ceiling the conv kernels are judged by
// (cdna guide rule 10: ceilings come from a known-good reference on the same hardware).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSB>
__global__ __launch_bounds__(256) void mfma_loop(const float *in, float *out, unsigned long long *clk, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = in[(blockIdx.x * 4096 + i) & 0xffff];
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av[NACC];
    for (int a = 0; a < NACC; ++a) av[a] = in[(tid * 7 + a * 131) & 0xffff];
    float bv = in[(tid * 13 + 5) & 0xffff];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LDSB) bv = lds[(tid + it * 64) & 4095];
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv, acc[a], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC, bool LDSB>
void run(const char *name, int wg_per_cu, const float *din, float *dout, unsigned long long *dclk) {
    const int grid = 256 * wg_per_cu, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int warm = 0; warm < 2; ++warm) hipLaunchKernelGGL((mfma_loop<NACC, LDSB>), dim3(grid), dim3(256), 0, 0, din, dout, dclk, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, LDSB>), dim3(grid), dim3(256), 0, 0, din, dout, dclk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> clk(2 * grid);
    hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int b = 0; b < grid; ++b) ghz += double(clk[2 * b]) / double(clk[2 * b + 1]) * 0.1;
    ghz /= grid;
    const double flop = double(grid) * 4 * iters * NACC * 4096.0;
    printf("%-34s wg/cu %d: %8.3f ms  %7.1f TFLOP/s  in-kernel clock %.2f GHz\n", name, wg_per_cu, best, flop / best * 1e-9, ghz);
}

int main() {
    float *din, *dout; unsigned long long *dclk;
    std::vector<float> h(65536);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 256 * 8 * 256 * 4); hipMalloc(&dclk, 256 * 8 * 16);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<4, false>("4 acc, register operands", 1, din, dout, dclk);
    run<4, false>("4 acc, register operands", 2, din, dout, dclk);
    run<4, false>("4 acc, register operands", 3, din, dout, dclk);
    run<1, false>("1 acc (dependent chain)", 1, din, dout, dclk);
    run<4, true>("4 acc, B operand from LDS", 1, din, dout, dclk);
    run<4, true>("4 acc, B operand from LDS", 2, din, dout, dclk);
    return 0;
}
