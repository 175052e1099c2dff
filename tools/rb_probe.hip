// Diagnostic probe (not part of libagx): where does a workgroup of the fused residual-block
// kernel spend its cycles?  Builds the kernel sources with AGX_STAMPS and prints the mean
// s_memtime deltas between the stamp points over the workgroups of batch item 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iaudio_generation_amd/csrc tools/rb_probe.hip -o tools/rb_probe_bin
#define AGX_STAMPS 1
#include "../audio_generation_amd/csrc/core.hip"
#include "../audio_generation_amd/csrc/pack.hip"
#include "../audio_generation_amd/csrc/resblock_mfma.hip"
// the 2-D lowering lives in conv2d.hip (not linked into the probe): pack.hip only needs the symbols to exist
namespace agx {
int lower_conv2d(const agx_conv2d_desc *, ConvPlan *) { return AGX_ERR_UNSUPPORTED; }
int lower_conv2d_bwd_data(const agx_conv2d_desc *, ConvPlan *) { return AGX_ERR_UNSUPPORTED; }
int conv_p_geometry(const ConvPlan &) { return 0; }   // conv_p.hip is not linked into the probe
}  // namespace agx

#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 128, L = argc > 2 ? atoi(argv[2]) : 9000, B = 32, dil = argc > 3 ? atoi(argv[3]) : 1;
    agx_conv_desc d1{AGX_CONV_CAUSAL, B, C, C, L, 7, 1, dil, AGX_EPI_LEAKY_PRE, 0.1f, 0};
    agx_conv_desc d2{AGX_CONV_CAUSAL, B, C, C, L, 1, 1, 1, 0, 0.1f, 0};
    std::vector<float> hx(size_t(B) * C * L), hw1(size_t(C) * C * 7), hw2(size_t(C) * C), hb(C, 0.01f);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return float(s >> 8) / 16777216.f - 0.5f; };
    for (auto &v : hx) v = rnd();
    for (auto &v : hw1) v = rnd() * 0.06f;
    for (auto &v : hw2) v = rnd() * 0.1f;
    float *x, *y, *w1, *w2, *p1, *p2, *b;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&y, hx.size() * 4);
    hipMalloc(&w1, hw1.size() * 4); hipMalloc(&w2, hw2.size() * 4); hipMalloc(&b, C * 4);
    hipMalloc(&p1, agx_conv_packed_floats(&d1) * 4); hipMalloc(&p2, agx_conv_packed_floats(&d2) * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
    agx_conv_pack(&d1, w1, nullptr, p1, nullptr);
    agx_conv_pack(&d2, w2, nullptr, p2, nullptr);
    agx::ConvPlan p;
    agx::lower_conv(&d1, &p);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) agx::launch_resblock_fused(p, x, p1, b, p2, b, y, 1, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> z(1 << 16, 0);
    hipMemcpyToSymbol(HIP_SYMBOL(agx::g_stamps), z.data(), z.size() * 8);
    hipEventRecord(e0);
    agx::launch_resblock_fused(p, x, p1, b, p2, b, y, 1, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(agx::g_stamps), z.size() * 8);
    const int nb = (L + 127) / 128 > 500 ? 500 : (L + (C == 32 ? 511 : C == 64 ? 255 : 127)) / (C == 32 ? 512 : C == 64 ? 256 : 128);
    double seg[6] = {0}, waitsum = 0, w9 = 0, w10 = 0; int n = 0;
    for (int blk = 0; blk < nb; ++blk) {
        unsigned long long *t = &z[blk * 16];
        if (!t[5]) continue;
        for (int i = 0; i < 5; ++i) seg[i] += double(t[i + 1] - t[i]);
        waitsum += double(t[8]); w9 += double(t[9]); w10 += double(t[10]); ++n;
    }
    printf("C=%d L=%d d=%d: kernel %.1f us; per-workgroup cycles (mean of %d WGs of batch 0):\n", C, L, dil, ms * 1e3, n);
    const char *name[5] = {"prologue (zero-fill, first DMA, first weights)", "GEMM1 main loop", "hidden activation", "GEMM2", "epilogue (bias + residual + store)"};
    double tot = 0; for (int i = 0; i < 5; ++i) tot += seg[i] / n;
    for (int i = 0; i < 5; ++i) printf("  %-50s %10.0f  (%4.1f %%)\n", name[i], seg[i] / n, 100 * seg[i] / n / tot);
    printf("  %-50s %10.0f  (%4.1f %% of the main loop)\n", "  of which chunk-end vmcnt(0)+barrier waits", waitsum / n, 100 * waitsum / seg[1]);
    printf("  %-50s %10.0f  (%4.1f %% of the main loop)\n", "  of which weight waits in the DMA-issuing phase", w9 / n, 100 * w9 / seg[1]);
    printf("  %-50s %10.0f  (%4.1f %% of the main loop)\n", "  of which weight waits in the other phases", w10 / n, 100 * w10 / seg[1]);
    printf("  total %10.0f cycles = %.1f us at 100 MHz memtime ticks?\n", tot, tot / 100.0);
    return 0;
}
