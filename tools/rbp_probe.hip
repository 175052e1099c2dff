// Diagnostic probe (not part of libagx): cycle stamps at the phase boundaries of ONE steady-state interval of the
// persistent residual-block kernel (csrc/resblock_p.hip), per wave, over all workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Iinclude -Iaudio_generation_amd/csrc tools/rbp_probe.hip -o tools/rbp_probe_bin
#define AGX_STAMPS 1
#ifndef AGX_STAMP_Q
#define AGX_STAMP_Q 5
#endif
#include "../audio_generation_amd/csrc/core.hip"
#include "../audio_generation_amd/csrc/pack.hip"
#include "../audio_generation_amd/csrc/resblock_p.hip"
namespace agx {
int lower_conv2d(const agx_conv2d_desc *, ConvPlan *) { return AGX_ERR_UNSUPPORTED; }
int lower_conv2d_bwd_data(const agx_conv2d_desc *, ConvPlan *) { return AGX_ERR_UNSUPPORTED; }
int conv_p_geometry(const ConvPlan &) { return 0; }   // conv_p.hip is not linked into the probe
}  // namespace agx

#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 36000, B = 32, dil = argc > 3 ? atoi(argv[3]) : 1;
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
    for (int i = 0; i < 3; ++i) agx::launch_resblock_p(p, x, p1, b, p2, b, y, 1, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> z(1 << 16, 0);
    hipMemcpyToSymbol(HIP_SYMBOL(agx::g_stamps), z.data(), z.size() * 8);
    hipEventRecord(e0);
    agx::launch_resblock_p(p, x, p1, b, p2, b, y, 1, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(agx::g_stamps), z.size() * 8);
    const char *name[10] = {"DMA issue (+ residual request)", "phase 0 (MFMAs of tap 0, reads of tap 1)", "phase 1", "phase 2", "phase 3",
                            "phase 4", "phase 5", "phase 6 (reads: next chunk tap 0)", "wait for own DMA (vmcnt 0)", "barrier"};
    std::vector<double> seg[10];
    for (int w = 0; w < 2048; ++w) {
        unsigned long long *t = &z[w * 16];
        if (!t[10] || !t[0]) continue;
        for (int i = 0; i < 10; ++i) seg[i].push_back(double(t[i + 1] - t[i]));
    }
    printf("C=%d L=%d d=%d: kernel %.1f us (stamped build); interval %d of every workgroup, %zu waves; cycles: median [p10 .. p90]\n", C, L,
           dil, ms * 1e3, AGX_STAMP_Q, seg[0].size());
    double tot = 0;
    for (int i = 0; i < 10; ++i) {
        auto &v = seg[i];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %-45s %8.0f [%8.0f .. %8.0f]\n", name[i], v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
        tot += v[v.size() / 2];
    }
    printf("  sum of medians %.0f cycles\n", tot);
    // tile tail of every workgroup's third tile: stamps 11 (after the last barrier) .. 14 (stores issued)
    const char *tname[3] = {"tail: W2 request + hidden activation", "tail: GEMM2 (first row pass)", "tail: activation + stores (first row pass)"};
    std::vector<double> ts[3];
    for (int w = 0; w < 2048; ++w) {
        unsigned long long *t = &z[w * 16];
        if (!t[14] || !t[11]) continue;
        for (int i = 0; i < 3; ++i) ts[i].push_back(double(t[12 + i] - t[11 + i]));
    }
    for (int i = 0; i < 3; ++i) {
        auto &v = ts[i];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %-45s %8.0f [%8.0f .. %8.0f]\n", tname[i], v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
    }
    return 0;
}
