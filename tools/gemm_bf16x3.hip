// Diagnostic (not part of libagx): is a 3-way bf16 split ("bf16x3") of fp32 operands on the bf16 MFMA a way past
// the fp32-MFMA ceiling (157 TFLOP/s) WITHOUT giving up fp32-level accuracy?
//   x = h + m + l (three bf16, 24 significant bits),  x*w ~= hh + hm + mh + hl + lh + mm   (6 bf16 MFMAs, K = 16 each,
//   32 cycles) against 8 fp32 MFMAs (K = 2 each, 64 cycles) for the same 32x32x16 block: 192 vs 512 cycles.
// Same tiled GEMM twice -- C[M x N] = A[M x K] B[K x N], 128 x 128 per workgroup, B through LDS as fp32 (split in
// registers, as a conv kernel would have to), A pre-split / pre-packed (as the weight pack kernel would) -- once
// on v_mfma_f32_32x32x2_f32, once on v_mfma_f32_32x32x16_bf16 x 6; prints rate and max error vs an fp64 host sum.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bf16x3.hip -o tools/gemm_bf16x3_bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 16;      // k per LDS chunk (one bf16 MFMA deep)
constexpr int BNT = 128;    // tile columns
constexpr int LDB = BNT + 4;

// ---- fp32 MFMA baseline: A stored [K][M]
__global__ __launch_bounds__(256, 2) void gemm_f32(const float *__restrict__ At, const float *__restrict__ B,
                                                   float *__restrict__ C, int M, int N, int K) {
    __shared__ float bs[KC * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64, n0 = blockIdx.x * 128, nw = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();
        for (int e = tid; e < KC * BNT; e += 256) bs[(e / BNT) * LDB + e % BNT] = B[size_t(k0 + e / BNT) * N + n0 + e % BNT];
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            const int k = 2 * ks + lh;
            float a[2], b[2];
            for (int i = 0; i < 2; ++i) a[i] = At[size_t(k0 + k) * M + m0 + i * 32 + li];
            for (int j = 0; j < 2; ++j) b[j] = bs[k * LDB + nw + j * 32 + li];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                C[size_t(row) * N + n0 + nw + j * 32 + li] = acc[i][j][r];
            }
}

// ---- bf16x3: A planes [3][K/8][M][8] bf16
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 hh = (__bf16)x[i];
        const float r1 = x[i] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        h[i] = hh; m[i] = mm; l[i] = (__bf16)r2;
    }
}

template <int NPROD>   // 6: full split; 3: hh + hm + mh (~2^-16)
__global__ __launch_bounds__(256, 2) void gemm_bf16x3(const __bf16 *__restrict__ Ap, const float *__restrict__ B,
                                                      float *__restrict__ C, int M, int N, int K) {
    __shared__ float bs[KC * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64, n0 = blockIdx.x * 128, nw = (wave & 1) * 64;
    const size_t plane = size_t(K / 8) * M * 8;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();
        for (int e = tid; e < KC * BNT; e += 256) bs[(e / BNT) * LDB + e % BNT] = B[size_t(k0 + e / BNT) * N + n0 + e % BNT];
        __syncthreads();
        bf16x8 a[3][2], b[3][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const __bf16 *src = Ap + (size_t(k0 / 8 + lh) * M + m0 + i * 32 + li) * 8;
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p][i] = *reinterpret_cast<const bf16x8 *>(src + p * plane);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = bs[(8 * lh + q) * LDB + nw + j * 32 + li];
            split3(x, b[0][j], b[1][j], b[2][j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                if (NPROD == 6) {   // small terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
                acc[i][j] = c;
            }
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                C[size_t(row) * N + n0 + nw + j * 32 + li] = acc[i][j][r];
            }
}

static unsigned short bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    const unsigned lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (unsigned short)(u >> 16);
}
static float bf16_to_f(unsigned short h) { unsigned u = unsigned(h) << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int M = 2048, N = 8192, K = 1024;
    std::vector<float> A(size_t(M) * K), Bm(size_t(K) * N), At(size_t(K) * M);
    unsigned s = 1234567u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float(s >> 8) / 16777216.f - 0.5f); };
    for (auto &v : A) v = rnd() * 0.2f;
    for (auto &v : Bm) v = rnd();
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) At[size_t(k) * M + m] = A[size_t(m) * K + k];
    std::vector<unsigned short> Ap(size_t(3) * K * M);
    const size_t plane = size_t(K / 8) * M * 8;
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            const float x = A[size_t(m) * K + k];
            const unsigned short h = bf16_rne(x); const float r1 = x - bf16_to_f(h);
            const unsigned short mm = bf16_rne(r1); const float r2 = r1 - bf16_to_f(mm);
            const unsigned short l = bf16_rne(r2);
            const size_t o = (size_t(k / 8) * M + m) * 8 + k % 8;
            Ap[o] = h; Ap[plane + o] = mm; Ap[2 * plane + o] = l;
        }
    float *dAt, *dB, *dC; void *dAp;
    hipMalloc(&dAt, At.size() * 4); hipMalloc(&dB, Bm.size() * 4); hipMalloc(&dC, size_t(M) * N * 4); hipMalloc(&dAp, Ap.size() * 2);
    hipMemcpy(dAt, At.data(), At.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bm.data(), Bm.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice);
    // fp64 reference on a sample of entries
    std::vector<int> rows = {0, 1, 37, 777, 2047}, cols = {0, 5, 4095, 8191};
    std::vector<double> ref;
    for (int r : rows) for (int c : cols) { double acc = 0; for (int k = 0; k < K; ++k) acc += double(A[size_t(r) * K + k]) * double(Bm[size_t(k) * N + c]); ref.push_back(acc); }
    std::vector<float> hC(size_t(M) * N);
    dim3 grid(N / 128, M / 128), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * M * N * K;
    auto report = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        double maxrel = 0, scale = 0; size_t q = 0;
        for (double v : ref) scale = fmax(scale, fabs(v));
        for (int r : rows) for (int c : cols) { maxrel = fmax(maxrel, fabs(double(hC[size_t(r) * N + c]) - ref[q]) / scale); ++q; }
        printf("%-44s %7.3f ms  %7.1f TFLOP/s (fp32-equivalent)  max |err| / max |C| = %.2e\n", name, best, flop / best * 1e-9, maxrel);
    };
    report("fp32 MFMA 32x32x2 (exact fp32 chain)", [&]() { hipLaunchKernelGGL(gemm_f32, grid, block, 0, 0, dAt, dB, dC, M, N, K); });
    report("bf16x3, 6 products on MFMA 32x32x16 bf16", [&]() { hipLaunchKernelGGL(gemm_bf16x3<6>, grid, block, 0, 0, (const __bf16 *)dAp, dB, dC, M, N, K); });
    report("bf16x3, 3 products (hh + hm + mh)", [&]() { hipLaunchKernelGGL(gemm_bf16x3<3>, grid, block, 0, 0, (const __bf16 *)dAp, dB, dC, M, N, K); });
    return 0;
}
