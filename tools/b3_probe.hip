// Diagnostic probe (not part of libagx): cycle stamps of one tile (every workgroup's third) of the bf16x3 ring block
// (csrc/resblock_b3.hip), per wave, plus the workgroups' start / end times (are two per CU resident all the time?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Iinclude -Iaudio_generation_amd/csrc tools/b3_probe.hip -o tools/b3_probe_bin
#define AGX_STAMPS 1
#include "../audio_generation_amd/csrc/core.hip"
#include "../audio_generation_amd/csrc/pack.hip"
#include "../audio_generation_amd/csrc/resblock_b3.hip"
namespace agx {
int lower_conv2d(const agx_conv2d_desc *, ConvPlan *) { return AGX_ERR_UNSUPPORTED; }
int lower_conv2d_bwd_data(const agx_conv2d_desc *, ConvPlan *) { return AGX_ERR_UNSUPPORTED; }
int conv_p_geometry(const ConvPlan &) { return 0; }
int conv_b3_geometry(const ConvPlan &) { return 0; }
}  // namespace agx

#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 36000, B = 32, dil = argc > 3 ? atoi(argv[3]) : 1;
    agx_conv_desc d1{AGX_CONV_CAUSAL, B, C, C, L, 7, 1, dil, AGX_EPI_LEAKY_PRE, 0.1f, AGX_IMPL_MFMA_BF16X3};
    agx_conv_desc d2{AGX_CONV_CAUSAL, B, C, C, L, 1, 1, 1, 0, 0.1f, AGX_IMPL_MFMA_BF16X3};
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
    for (int i = 0; i < 3; ++i) agx::launch_resblock_b3(p, x, p1, b, p2, b, y, 1, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> z(1 << 16, 0);
    hipMemcpyToSymbol(HIP_SYMBOL(agx::g_stamps), z.data(), z.size() * 8);
    hipEventRecord(e0);
    agx::launch_resblock_b3(p, x, p1, b, p2, b, y, 1, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(agx::g_stamps), z.size() * 8);
    const int nch = C / 16;
    printf("C=%d L=%d d=%d: kernel %.1f us (stamped build); third tile of every workgroup; cycles: median [p10 .. p90]\n", C, L, dil, ms * 1e3);
    auto report = [&](const char *name, int a, int bb) {
        std::vector<double> v;
        for (int w = 0; w < 4096; ++w) {
            unsigned long long *t = &z[w * 16];
            if (!t[a] || !t[bb]) continue;
            v.push_back(double(t[bb] - t[a]));
        }
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        printf("  %-44s %8.0f [%8.0f .. %8.0f]  (%zu waves)\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], v.size());
    };
    report("tile start -> first chunk done", 0, 1);
    for (int c = 1; c < nch && c < 8; ++c) { char nm[64]; snprintf(nm, 64, "chunk %d", c); report(nm, c, c + 1); }
    report("hidden activation (+ first W2 request)", nch, 9);
    report("GEMM2, first row pass", 9, 10);
    report("epilogue, first row pass", 10, 11);
    report("remaining passes", 11, 12);
    report("whole tile", 0, 12);
    // residency: realtime (100 MHz) start / end of every wave
    unsigned long long tmin = ~0ull, tmax = 0;
    std::vector<double> life;
    for (int w = 0; w < 4096; ++w) {
        unsigned long long *t = &z[w * 16];
        if (!t[14] || !t[15]) continue;
        tmin = std::min(tmin, t[14]); tmax = std::max(tmax, t[15]);
    }
    std::vector<double> st, en;
    for (int w = 0; w < 4096; ++w) {
        unsigned long long *t = &z[w * 16];
        if (!t[14] || !t[15]) continue;
        st.push_back((t[14] - tmin) * 0.01); en.push_back((t[15] - tmin) * 0.01); life.push_back((t[15] - t[14]) * 0.01);
    }
    // in-kernel clock (C <= 64: slots 5 / 6 hold s_memtime at the kernel's start / end, next to the 100 MHz stamps 14 / 15)
    std::vector<double> ghz;
    for (int w = 0; w < 4096 && nch <= 4; ++w) {
        unsigned long long *t = &z[w * 16];
        if (!t[5] || !t[6] || !t[14] || !t[15] || t[15] <= t[14]) continue;
        ghz.push_back(double(t[6] - t[5]) / double(t[15] - t[14]) * 0.1);
    }
    if (!ghz.empty()) {
        std::sort(ghz.begin(), ghz.end());
        printf("  shader clock over the kernel (d s_memtime / d s_memrealtime): median %.3f GHz [p10 %.3f .. p90 %.3f]\n", ghz[ghz.size() / 2],
               ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10]);
    }
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); std::sort(life.begin(), life.end());
    if (!st.empty())
        printf("  waves: %zu; start us: median %.1f p90 %.1f max %.1f; end us: p10 %.1f median %.1f max %.1f; lifetime us: p10 %.1f median %.1f\n",
               st.size(), st[st.size() / 2], st[st.size() * 9 / 10], st.back(), en[en.size() / 10], en[en.size() / 2], en.back(),
               life[life.size() / 10], life[life.size() / 2]);
    return 0;
}
