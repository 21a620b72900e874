// Packed-sequence (B)LSTM recurrence for gfx950 (MI355X): forward and backward-through-time.
//
// Replaces the recurrent part of torch.nn.LSTM on a PackedSequence as used by
// padertorch/contrib/examples/source_separation/pit/model.py:60-66,97 and contrib/tcl/dc.py:32-34,61
// (PyTorch semantics: gate order i,f,g,o; two bias vectors; zero initial state).
//
// The input projections X W_ih^T + b (60 % of the LSTM FLOPs) are ONE dense GEMM per layer for
// both directions and stay on the BLAS library.  What is hand-written here is the part that is
// sequential in time: per timestep ONE launch that covers both directions
//     gates = gx[t] + h_{t-1} W_hh^T   (exact fp32 on the matrix cores: v_mfma_f32_16x16x4_f32,
//                                        operands streamed L2 -> VGPR as K-contiguous float4,
//                                        K split over the 4 wavefronts of a workgroup, LDS reduce)
//     i,f,o = sigmoid, g = tanh, c_t = f c_{t-1} + i g, h_t = o tanh(c_t)      (fused epilogue)
// and the mirrored step of the backward pass (dh_rec = dgates_{t+1} W_hh, then the gate
// derivatives).  A per-timestep all-gather of h is an all-to-all seam, so the time loop is cut at
// kernel boundaries (MI355X guide: megakernel verdict) instead of a persistent kernel with grid
// barriers; MIOpen spends 4 launches (2 GEMMs + 2 pointwise) per step and direction here.
//
// Layouts (rows = packed time-major rows of the PackedSequence, row(t, b) = offs[t] + b):
//   gx / gates / dgates  [rows][ndir][4][H]   (pre-activations in, activations out: in place)
//   hy, c, dhy           [rows][ndir][H]
//   w_hh_pad             [ndir][4H][KP]       KP = H rounded up to 16, zero padded
//   w_hh_t               [ndir][H][4H]
#include "common.h"

namespace ptmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LstmArgs {
    float* gx;
    float* hy;
    float* c;
    const float* w;
    const int32_t* bs;
    const int64_t* offs;
    int T, H, KP, ndir, step;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// One forward timestep.  grid = (H / JT, ndir, ceil(maxB / 32)), 256 threads.
// Workgroup tile: 32 batch rows x (4 gates x JT hidden units) = 32 x 32 outputs (JT = 8).
template <int JT>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const LstmArgs A) {
    constexpr int NC = 4 * JT;          // gate columns per workgroup (32)
    static_assert(NC == 32, "tile is 2 x 2 MFMA tiles");
    const int dir = blockIdx.y;
    const int j0 = blockIdx.x * JT;
    const int m0 = blockIdx.z * 32;
    const int t = dir == 0 ? A.step : A.T - 1 - A.step;
    const int nb = A.bs[t];
    if (m0 >= nb) return;
    const long long row0 = A.offs[t];
    const int tp = dir == 0 ? t - 1 : t + 1;
    int nprev = 0;
    long long prow0 = 0;
    if (tp >= 0 && tp < A.T) {
        nprev = min(A.bs[tp], nb);
        prow0 = A.offs[tp];
    }
    const int H = A.H, G = 4 * H;
    const long long ld_g = (long long)A.ndir * G, ld_h = (long long)A.ndir * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, r = lane & 15;

    __shared__ float red[4][32][NC + 1];
    const bool has_rec = nprev > m0;

    // epilogue operands first: their latency hides behind the GEMM
    const int bl = tid / JT, u = tid - bl * JT;
    const int b = m0 + bl;
    const bool act = b < nb && j0 + u < H;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    float cprev = 0.f;
    float* gp = A.gx + (row0 + b) * ld_g + (long long)dir * G + j0 + u;
    if (act) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = gp[q * H];
        if (b < nprev) cprev = A.c[(prow0 + b) * ld_h + dir * H + j0 + u];
    }

    if (has_rec) {
        f32x4 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nblk = A.KP >> 4;
        const int per = (nblk + 3) >> 2;
        const int kb0 = wave * per;
        const int kb1 = min(nblk, kb0 + per);
        const float* ap[2];
        bool av[2];
        const float* bp[2];
        bool bv[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int i = m0 + mt * 16 + r;
            av[mt] = i < nprev;
            ap[mt] = A.hy + (prow0 + (av[mt] ? i : 0)) * ld_h + dir * H + 4 * g4;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int cidx = nt * 16 + r;
            const int gate = cidx / JT, uu = cidx - gate * JT;
            bv[nt] = j0 + uu < H;
            bp[nt] = A.w + ((long long)dir * G + gate * H + (bv[nt] ? j0 + uu : 0)) * A.KP + 4 * g4;
        }
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load_a = [&](int mt, int kb) -> f32x4 {
            const bool ok = av[mt] && (kb * 16 + 4 * g4 < H);
            return ok ? *reinterpret_cast<const f32x4*>(ap[mt] + kb * 16) : zero;
        };
        auto load_b = [&](int nt, int kb) -> f32x4 {
            return bv[nt] ? *reinterpret_cast<const f32x4*>(bp[nt] + kb * 16) : zero;
        };
        if (kb0 < kb1) {
            f32x4 a_n[2], b_n[2];
            a_n[0] = load_a(0, kb0); a_n[1] = load_a(1, kb0);
            b_n[0] = load_b(0, kb0); b_n[1] = load_b(1, kb0);
            for (int kb = kb0; kb < kb1; ++kb) {
                const f32x4 a0 = a_n[0], a1 = a_n[1], b0 = b_n[0], b1 = b_n[1];
                if (kb + 1 < kb1) {
                    a_n[0] = load_a(0, kb + 1); a_n[1] = load_a(1, kb + 1);
                    b_n[0] = load_b(0, kb + 1); b_n[1] = load_b(1, kb + 1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b0[q], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b1[q], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b0[q], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b1[q], acc[1][1], 0, 0, 0);
                }
            }
        }
        // C layout of mfma_f32_16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wave][mt * 16 + g4 * 4 + q][nt * 16 + r] = acc[mt][nt][q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cidx = q * JT + u;
            pre[q] += (red[0][bl][cidx] + red[1][bl][cidx]) + (red[2][bl][cidx] + red[3][bl][cidx]);
        }
    }
    if (act) {
        const float ig = sigmoidf_(pre[0]);
        const float fg = sigmoidf_(pre[1]);
        const float gg = tanhf(pre[2]);
        const float og = sigmoidf_(pre[3]);
        const float cn = fg * cprev + ig * gg;
        const float h = og * tanhf(cn);
        gp[0] = ig;
        gp[H] = fg;
        gp[2 * H] = gg;
        gp[3 * H] = og;
        const long long o = (row0 + b) * ld_h + dir * H + j0 + u;
        A.c[o] = cn;
        A.hy[o] = h;
    }
}

struct LstmBwdArgs {
    const float* gates;
    const float* c;
    const float* dhy;
    const float* wt;
    float* dg;
    float* dcs;
    const int32_t* bs;
    const int64_t* offs;
    int T, H, ndir, step;
};

// One backward timestep.  grid = (ceil(H / 16), ceil(maxB / 16), ndir), 256 threads.
// Workgroup tile: 16 batch rows x 16 hidden units of dh_rec = dgates_{next} W_hh, K = 4H.
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const LstmBwdArgs A) {
    const int dir = blockIdx.z;
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * 16;
    const int t = dir == 0 ? A.T - 1 - A.step : A.step;
    const int nb = A.bs[t];
    if (m0 >= nb) return;
    const long long row0 = A.offs[t];
    const int tn = dir == 0 ? t + 1 : t - 1;      // processed by the previous launch
    const int tp = dir == 0 ? t - 1 : t + 1;      // the forward pass' predecessor (for c_{t-1})
    int nnext = 0, npv = 0;
    long long nrow0 = 0, prow0 = 0;
    if (tn >= 0 && tn < A.T) {
        nnext = min(A.bs[tn], nb);
        nrow0 = A.offs[tn];
    }
    if (tp >= 0 && tp < A.T) {
        npv = min(A.bs[tp], nb);
        prow0 = A.offs[tp];
    }
    const int H = A.H, G = 4 * H;
    const long long ld_g = (long long)A.ndir * G, ld_h = (long long)A.ndir * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, r = lane & 15;

    __shared__ float red[4][16][17];
    const bool has_rec = nnext > m0;
    const int bl = tid >> 4, jl = tid & 15;
    const int b = m0 + bl, j = n0 + jl;
    const bool act = b < nb && j < H;

    // epilogue operands first
    float dh = 0.f, dc = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cn = 0.f, cprev = 0.f;
    const long long oh = (row0 + b) * ld_h + dir * H + j;
    const long long og_ = (row0 + b) * ld_g + (long long)dir * G + j;
    float* dcs = A.dcs + ((long long)b * A.ndir + dir) * H + j;
    if (act) {
        dh = A.dhy[oh];
        dc = *dcs;
        ig = A.gates[og_];
        fg = A.gates[og_ + H];
        gg = A.gates[og_ + 2 * H];
        og = A.gates[og_ + 3 * H];
        cn = A.c[oh];
        if (b < npv) cprev = A.c[(prow0 + b) * ld_h + dir * H + j];
    }

    if (has_rec) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nblk = G >> 4;
        const int per = (nblk + 3) >> 2;
        const int kb0 = wave * per;
        const int kb1 = min(nblk, kb0 + per);
        const bool av = m0 + r < nnext;
        const bool bv = n0 + r < H;
        const float* ap = A.dg + (nrow0 + (av ? m0 + r : 0)) * ld_g + (long long)dir * G + 4 * g4;
        const float* bp = A.wt + ((long long)dir * H + (bv ? n0 + r : 0)) * G + 4 * g4;
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kb0 < kb1) {
            f32x4 a_n = av ? *reinterpret_cast<const f32x4*>(ap + kb0 * 16) : zero;
            f32x4 b_n = bv ? *reinterpret_cast<const f32x4*>(bp + kb0 * 16) : zero;
            for (int kb = kb0; kb < kb1; ++kb) {
                const f32x4 a = a_n, bb = b_n;
                if (kb + 1 < kb1) {
                    a_n = av ? *reinterpret_cast<const f32x4*>(ap + (kb + 1) * 16) : zero;
                    b_n = bv ? *reinterpret_cast<const f32x4*>(bp + (kb + 1) * 16) : zero;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], bb[q], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave][g4 * 4 + q][r] = acc[q];
        __syncthreads();
        if (b < nnext) dh += (red[0][bl][jl] + red[1][bl][jl]) + (red[2][bl][jl] + red[3][bl][jl]);
    }
    if (act) {
        const float tc = tanhf(cn);
        const float d_o = dh * tc;
        dc += dh * og * (1.f - tc * tc);
        const float d_i = dc * gg;
        const float d_g = dc * ig;
        const float d_f = dc * cprev;
        *dcs = dc * fg;
        float* dgp = A.dg + og_;
        dgp[0] = d_i * ig * (1.f - ig);
        dgp[H] = d_f * fg * (1.f - fg);
        dgp[2 * H] = d_g * (1.f - gg * gg);
        dgp[3 * H] = d_o * og * (1.f - og);
    }
}

}  // namespace ptmi

using namespace ptmi;

extern "C" {

int ptmi_lstm_forward(float* gates, float* hy, float* c, const float* w_hh_pad, const int32_t* batch_sizes,
                      const int64_t* offsets, int32_t T, int32_t max_batch, int32_t H, int32_t KP,
                      int32_t ndir, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!gates || !hy || !c || !w_hh_pad || !batch_sizes || !offsets, PTMI_E_INVALID);
    PTMI_RETURN_IF(T < 0 || max_batch < 1 || H < 1 || (ndir != 1 && ndir != 2), PTMI_E_INVALID);
    PTMI_RETURN_IF(H % 4 != 0 || KP % 16 != 0 || KP < H, PTMI_E_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    LstmArgs A{gates, hy, c, w_hh_pad, batch_sizes, offsets, T, H, KP, ndir, 0};
    const dim3 grid((unsigned)((H + 7) / 8), (unsigned)ndir, (unsigned)((max_batch + 31) / 32));
    for (int s = 0; s < T; ++s) {
        A.step = s;
        hipLaunchKernelGGL(lstm_fwd_step_kernel<8>, grid, dim3(256), 0, st, A);
    }
    return launch_status();
}

int ptmi_lstm_backward(const float* gates, const float* c, const float* dhy, const float* w_hh_t, float* dgates,
                       float* dc_state, const int32_t* batch_sizes, const int64_t* offsets, int32_t T,
                       int32_t max_batch, int32_t H, int32_t ndir, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!gates || !c || !dhy || !w_hh_t || !dgates || !dc_state || !batch_sizes || !offsets,
                   PTMI_E_INVALID);
    PTMI_RETURN_IF(T < 0 || max_batch < 1 || H < 1 || (ndir != 1 && ndir != 2), PTMI_E_INVALID);
    PTMI_RETURN_IF(H % 4 != 0, PTMI_E_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dc_state, 0, sizeof(float) * (size_t)max_batch * ndir * H, st);
    if (e != hipSuccess) return (int)e;
    LstmBwdArgs A{gates, c, dhy, w_hh_t, dgates, dc_state, batch_sizes, offsets, T, H, ndir, 0};
    const dim3 grid((unsigned)((H + 15) / 16), (unsigned)((max_batch + 15) / 16), (unsigned)ndir);
    for (int s = 0; s < T; ++s) {
        A.step = s;
        hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(256), 0, st, A);
    }
    return launch_status();
}

}  // extern "C"
