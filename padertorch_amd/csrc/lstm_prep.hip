// Operand forms of one (bi)directional LSTM layer's parameters, produced by ONE launch after an optimizer step.
//
// torch.nn.LSTM keeps weight_ih / weight_hh / bias_ih / bias_hh per direction (the reference model:
// padertorch/contrib/examples/source_separation/pit/model.py:60-66); the kernels of this library consume
//   w_ih_cat [ndir * 4H][Ipad]   both directions' input weights stacked (one projection GEMM per layer), the
//                                reduction axis zero-padded to a multiple of 4 (float4 loads of csrc/gemm.hip)
//   bias     [ndir * 4H]         b_ih + b_hh
//   w_pad    [ndir][4H][KP]      recurrent weights, columns zero-padded to KP = roundup(H, 16)  (forward recurrence)
//   w_t      [ndir][H][4H]       their transpose                                                 (backward recurrence)
//   amax     [2]                 float bits of max |W_ih|, max |W_hh| over both directions (operand scales of the split
//                                16-bit products)
// Round 1/2 built them with ~14 torch kernels per layer and step (cat, add, cat, stack, pad, transpose, 2 x (memset +
// absmax) + maximum: ~95 us of launch-bound work per layer); this is one pass: every parameter is read once.
#include "common.h"

namespace ptmi {

struct PrepArgs {
    const float* w_ih[2];
    const float* w_hh[2];
    const float* b_ih[2];
    const float* b_hh[2];
    float* w_ih_cat;
    float* bias;
    float* w_pad;
    float* w_t;
    unsigned* amax;          // [2], zeroed by the host call
    int ndir, G, I, Ipad, H, KP;
    int blocks_ih;           // workgroups of the first job (the others follow in blockIdx order)
    int tiles_g, tiles_h;    // 32 x 32 tiles of W_hh (rows g, columns h up to KP)
};

__device__ __forceinline__ void publish_max(unsigned* word, float m, unsigned* red) {
    unsigned u = __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = u;
    __syncthreads();
    if (threadIdx.x == 0) {
        u = max(max(red[0], red[1]), max(red[2], red[3]));
        if (u >= 0x7f800000u) u = 0x7f7fffffu;
        // monotonic word: skip the atomic when it could not raise it (most workgroups; the atomics of a launch queue up
        // on one L2 address)
        if (u > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, u);
    }
}

__global__ __launch_bounds__(256) void lstm_weight_prep_kernel(const PrepArgs A) {
    __shared__ float tile[32][33];
    __shared__ unsigned red[4];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    if (blk < A.blocks_ih) {
        // job 1: rows of w_ih_cat, 4096 elements per workgroup (row-major over [ndir * G][Ipad])
        const long long total = (long long)A.ndir * A.G * A.Ipad;
        float m = 0.f;
        for (int e = 0; e < 16; ++e) {
            const long long idx = (long long)blk * 4096 + e * 256 + tid;
            if (idx < total) {
                const long long row = idx / A.Ipad;
                const int col = (int)(idx - row * A.Ipad);
                const int d = (int)(row / A.G);
                const long long g = row - (long long)d * A.G;
                const float v = col < A.I ? A.w_ih[d][g * A.I + col] : 0.f;
                A.w_ih_cat[idx] = v;
                m = fmaxf(m, fabsf(v));
            }
        }
        publish_max(A.amax, m, red);
        return;
    }
    blk -= A.blocks_ih;
    const int ntile = A.ndir * A.tiles_g * A.tiles_h;
    if (blk < ntile) {
        // job 2: one 32 x 32 tile of W_hh[d]: padded copy and transpose
        const int d = blk / (A.tiles_g * A.tiles_h), rem = blk - d * A.tiles_g * A.tiles_h;
        const int tg = rem / A.tiles_h, th = rem - tg * A.tiles_h;
        const int g0 = tg * 32, h0 = th * 32;
        const int tx = tid & 31, ty = tid >> 5;          // 32 x 8 threads, 4 rows each
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int g = g0 + ty + q * 8, h = h0 + tx;
            float v = 0.f;
            if (g < A.G && h < A.H) v = A.w_hh[d][(long long)g * A.H + h];
            if (g < A.G && h < A.KP) A.w_pad[((long long)d * A.G + g) * A.KP + h] = v;
            tile[ty + q * 8][tx] = v;
            m = fmaxf(m, fabsf(v));
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = h0 + ty + q * 8, g = g0 + tx;
            if (h < A.H && g < A.G) A.w_t[((long long)d * A.H + h) * A.G + g] = tile[tx][ty + q * 8];
        }
        publish_max(A.amax + 1, m, red);
        return;
    }
    blk -= ntile;
    // job 3: summed biases
    const int idx = blk * 256 + tid;
    if (idx < A.ndir * A.G) {
        const int d = idx / A.G, g = idx - d * A.G;
        A.bias[idx] = A.b_ih[d][g] + A.b_hh[d][g];
    }
}

}  // namespace ptmi

using namespace ptmi;

extern "C" int ptmi_lstm_weight_prep(const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                                     const float* const* b_hh, int32_t ndir, int32_t H, int32_t I, float* w_ih_cat, int32_t Ipad,
                                     float* bias, float* w_pad, int32_t KP, float* w_t, uint32_t* amax, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!w_ih || !w_hh || !b_ih || !b_hh || !w_ih_cat || !bias || !w_pad || !w_t || !amax, PTMI_E_INVALID);
    PTMI_RETURN_IF(ndir < 1 || ndir > 2 || H < 1 || I < 1 || Ipad < I || KP < H, PTMI_E_INVALID);
    PrepArgs A{};
    for (int d = 0; d < ndir; ++d) {
        PTMI_RETURN_IF(!w_ih[d] || !w_hh[d] || !b_ih[d] || !b_hh[d], PTMI_E_INVALID);
        A.w_ih[d] = w_ih[d];
        A.w_hh[d] = w_hh[d];
        A.b_ih[d] = b_ih[d];
        A.b_hh[d] = b_hh[d];
    }
    A.w_ih_cat = w_ih_cat;
    A.bias = bias;
    A.w_pad = w_pad;
    A.w_t = w_t;
    A.amax = amax;
    A.ndir = ndir;
    A.G = 4 * H;
    A.I = I;
    A.Ipad = Ipad;
    A.H = H;
    A.KP = KP;
    const long long total_ih = (long long)ndir * A.G * Ipad;
    A.blocks_ih = (int)((total_ih + 4095) / 4096);
    A.tiles_g = (A.G + 31) / 32;
    A.tiles_h = (KP + 31) / 32;
    const int blocks_bias = (ndir * A.G + 255) / 256;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = zero_words_async(amax, 2, st);
    if (e != hipSuccess) return (int)e;
    const unsigned grid = (unsigned)(A.blocks_ih + ndir * A.tiles_g * A.tiles_h + blocks_bias);
    hipLaunchKernelGGL(lstm_weight_prep_kernel, dim3(grid), dim3(256), 0, st, A);
    return launch_status();
}
