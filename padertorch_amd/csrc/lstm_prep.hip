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

constexpr int kIhPerWg = 16384;      // elements of w_ih_cat per workgroup (job 1)
constexpr int kStrip = 4;            // 32 x 32 tiles of W_hh per workgroup (job 2)

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
    int vec_ih;              // rows of weight_ih can be copied as float4
    int blocks_ih;           // workgroups of the first job (the others follow in blockIdx order)
    int tiles_g, strips_h;   // W_hh in strips of 32 rows (g) x kStrip * 32 columns (h up to KP)
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
        // on one L2 address).  Round 6, 1100 workgroups, c2 layer: 21.5 us per launch, 14.2 without this word, 22.5 with an
        // unconditional atomicMax (scripts/exp_prep.py under rocprofv3)
        if (u > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, u);
    }
}

__global__ __launch_bounds__(256) void lstm_weight_prep_kernel(const PrepArgs A) {
    __shared__ float tile[kStrip][32][33];
    __shared__ unsigned red[4];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    if (blk < A.blocks_ih) {
        // job 1: rows of w_ih_cat, kIhPerWg elements per workgroup (row-major over [ndir * G][Ipad]): 16 float4 per thread, all
        // requested before the first is stored.  32-bit index arithmetic and ONE division per thread (the 64-bit `idx / Ipad` per
        // element of round 3 was ~200 instructions each); few, large workgroups: every workgroup ends with an access to the ONE
        // maximum word, and those serialise (round 6: 4257 workgroups -> ~14 us of a 40 us launch was that word).
        const unsigned total = (unsigned)(A.ndir * A.G) * (unsigned)A.Ipad;        // < 2^31 (checked by the host call)
        const unsigned Ipad = (unsigned)A.Ipad;
        float m = 0.f;
        unsigned idx = (unsigned)blk * (unsigned)kIhPerWg + (unsigned)tid * 4u;
        unsigned row = idx / Ipad, col = idx - row * Ipad;
        const bool vec = A.vec_ih != 0;            // I == Ipad and 16-byte aligned parameters (host check)
        const unsigned skip_rows = 1024u / Ipad, skip_cols = 1024u - skip_rows * Ipad;
        float4 v[kIhPerWg / 1024];
#pragma unroll
        for (int e = 0; e < kIhPerWg / 1024; ++e) {
            v[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx + 1024u * e < total) {
                const unsigned d = row >= (unsigned)A.G ? 1u : 0u;
                const unsigned g = row - d * (unsigned)A.G;
                const float* src = A.w_ih[d] + (size_t)g * A.I;
                if (vec) {
                    v[e] = *reinterpret_cast<const float4*>(src + col);
                } else {
                    v[e].x = col + 0 < (unsigned)A.I ? src[col + 0] : 0.f;
                    v[e].y = col + 1 < (unsigned)A.I ? src[col + 1] : 0.f;
                    v[e].z = col + 2 < (unsigned)A.I ? src[col + 2] : 0.f;
                    v[e].w = col + 3 < (unsigned)A.I ? src[col + 3] : 0.f;
                }
            }
            col += skip_cols;
            row += skip_rows;
            if (col >= Ipad) {
                col -= Ipad;
                ++row;
            }
        }
#pragma unroll
        for (int e = 0; e < kIhPerWg / 1024; ++e) {
            if (idx + 1024u * e < total) {
                *reinterpret_cast<float4*>(A.w_ih_cat + idx + 1024u * e) = v[e];
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[e].x), fabsf(v[e].y))), fmaxf(fabsf(v[e].z), fabsf(v[e].w)));
            }
        }
        publish_max(A.amax, m, red);
        return;
    }
    blk -= A.blocks_ih;
    const int ntile = A.ndir * A.tiles_g * A.strips_h;
    if (blk < ntile) {
        // job 2: a strip of kStrip 32 x 32 tiles of W_hh[d] (32 rows g, kStrip * 32 columns h): padded copy and transpose; all of the
        // strip's values are requested before the first is stored
        const int per_d = A.tiles_g * A.strips_h;
        const int d = blk / per_d, rem = blk - d * per_d;
        const int tg = rem / A.strips_h, ts = rem - tg * A.strips_h;
        const int g0 = tg * 32, h00 = ts * (32 * kStrip);
        const int tx = tid & 31, ty = tid >> 5;          // 32 x 8 threads, 4 rows each per tile
        const float* src = A.w_hh[d];
        float v[kStrip][4];
#pragma unroll
        for (int s = 0; s < kStrip; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = g0 + ty + q * 8, h = h00 + s * 32 + tx;
                v[s][q] = (g < A.G && h < A.H) ? src[(size_t)g * A.H + h] : 0.f;
            }
        }
        float m = 0.f;
#pragma unroll
        for (int s = 0; s < kStrip; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = g0 + ty + q * 8, h = h00 + s * 32 + tx;
                if (g < A.G && h < A.KP) A.w_pad[((size_t)d * A.G + g) * A.KP + h] = v[s][q];
                tile[s][ty + q * 8][tx] = v[s][q];
                m = fmaxf(m, fabsf(v[s][q]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < kStrip; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int h = h00 + s * 32 + ty + q * 8, g = g0 + tx;
                if (h < A.H && g < A.G) A.w_t[((size_t)d * A.H + h) * A.G + g] = tile[s][tx][ty + q * 8];
            }
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
    PTMI_RETURN_IF(ndir < 1 || ndir > 2 || H < 1 || I < 1 || Ipad < I || (Ipad & 3) || KP < H, PTMI_E_INVALID);
    PTMI_RETURN_IF((long long)ndir * 4 * H * Ipad >= (1ll << 31), PTMI_E_INVALID);
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
    A.vec_ih = (I & 3) == 0 && Ipad == I;
    for (int d = 0; d < ndir; ++d) A.vec_ih = A.vec_ih && (reinterpret_cast<uintptr_t>(w_ih[d]) & 15) == 0;
    PTMI_RETURN_IF(reinterpret_cast<uintptr_t>(w_ih_cat) & 15, PTMI_E_INVALID);
    const long long total_ih = (long long)ndir * A.G * Ipad;
    A.blocks_ih = (int)((total_ih + kIhPerWg - 1) / kIhPerWg);
    A.tiles_g = (A.G + 31) / 32;
    A.strips_h = (KP + 32 * kStrip - 1) / (32 * kStrip);
    const int blocks_bias = (ndir * A.G + 255) / 256;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = zero_words_async(amax, 2, st);
    if (e != hipSuccess) return (int)e;
    const unsigned grid = (unsigned)(A.blocks_ih + ndir * A.tiles_g * A.strips_h + blocks_bias);
    hipLaunchKernelGGL(lstm_weight_prep_kernel, dim3(grid), dim3(256), 0, st, A);
    return launch_status();
}
