// Codec bitstream packing: the (B, T, Q) int64 code indices <-> a dense
// little-endian bitstream of `bits` bits per code (10 bits for the 1024-entry codebooks;
// utils.py:137-147 `bitrate_calculator` is the reference's only statement about the wire size:
// bits per frame = Q * log2(K)).  Byte/integer work, HBM-bound: 8 B read per 10 bits written.
// One thread owns 8 consecutive codes = exactly `bits` bytes, so no two threads share a byte.
#include "common.hpp"

namespace agx {

__global__ __launch_bounds__(256) void codes_pack_kernel(const int64_t *__restrict__ codes, int64_t n, int bits,
                                                         uint8_t *__restrict__ out) {
    const int64_t g = int64_t(blockIdx.x) * 256 + threadIdx.x;  // group of 8 codes
    if (g * 8 >= n) return;
    unsigned __int128 acc = 0;
    const uint64_t mask = (bits >= 64) ? ~0ull : ((1ull << bits) - 1);
    for (int i = 0; i < 8; ++i) {
        const int64_t e = g * 8 + i;
        const uint64_t v = e < n ? (uint64_t(codes[e]) & mask) : 0;
        acc |= (unsigned __int128)v << (i * bits);
    }
    uint8_t *dst = out + g * bits;
    const int64_t total = (n * bits + 7) / 8;
    for (int b = 0; b < bits; ++b)
        if (g * bits + b < total) dst[b] = uint8_t(acc >> (8 * b));
}

__global__ __launch_bounds__(256) void codes_unpack_kernel(const uint8_t *__restrict__ in, int64_t n, int bits,
                                                           int64_t *__restrict__ codes) {
    const int64_t g = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (g * 8 >= n) return;
    const int64_t total = (n * bits + 7) / 8;
    unsigned __int128 acc = 0;
    for (int b = 0; b < bits; ++b)
        if (g * bits + b < total) acc |= (unsigned __int128)in[g * bits + b] << (8 * b);
    const uint64_t mask = (1ull << bits) - 1;
    for (int i = 0; i < 8; ++i) {
        const int64_t e = g * 8 + i;
        if (e < n) codes[e] = int64_t(uint64_t(acc >> (i * bits)) & mask);
    }
}

}  // namespace agx

extern "C" {

int64_t agx_codes_packed_bytes(int64_t n_codes, int32_t bits) {
    if (n_codes < 0 || bits < 1 || bits > 16) return AGX_ERR_BAD_SHAPE;
    return (n_codes * bits + 7) / 8;
}

int agx_codes_pack(const int64_t *codes, int64_t n_codes, int32_t bits, uint8_t *out, void *stream) {
    using namespace agx;
    if (n_codes <= 0 || bits < 1 || bits > 16) return fail(AGX_ERR_BAD_SHAPE, "codes_pack: n=%lld bits=%d", (long long)n_codes, bits);
    if (!codes || !out) return fail(AGX_ERR_NULL_POINTER, "codes_pack: NULL pointer");
    hipLaunchKernelGGL(codes_pack_kernel, dim3((unsigned)ceil_div64(ceil_div64(n_codes, 8), 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), codes, n_codes, bits, out);
    return check_launch("codes_pack");
}

int agx_codes_unpack(const uint8_t *in, int64_t n_codes, int32_t bits, int64_t *codes, void *stream) {
    using namespace agx;
    if (n_codes <= 0 || bits < 1 || bits > 16) return fail(AGX_ERR_BAD_SHAPE, "codes_unpack: n=%lld bits=%d", (long long)n_codes, bits);
    if (!in || !codes) return fail(AGX_ERR_NULL_POINTER, "codes_unpack: NULL pointer");
    hipLaunchKernelGGL(codes_unpack_kernel, dim3((unsigned)ceil_div64(ceil_div64(n_codes, 8), 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, n_codes, bits, codes);
    return check_launch("codes_unpack");
}

}  // extern "C"
