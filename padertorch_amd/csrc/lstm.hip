// Packed-sequence (B)LSTM recurrence for gfx950 (MI355X): forward and backward-through-time.
//
// Replaces the recurrent part of torch.nn.LSTM on a PackedSequence as used by
// padertorch/contrib/examples/source_separation/pit/model.py:60-66,97 and contrib/tcl/dc.py:32-34,61
// (PyTorch semantics: gate order i,f,g,o; two bias vectors; zero initial state).
//
// The input projections X W_ih^T + b (60 % of the LSTM FLOPs) are ONE dense GEMM per layer for
// both directions and stay on the BLAS library.  What is hand-written here is the part that is
// sequential in time:
//     gates = gx[t] + h_{t-1} W_hh^T   (exact fp32 on the matrix cores: v_mfma_f32_16x16x4_f32,
//                                        K split over the wavefronts of a workgroup, LDS reduce)
//     i,f,o = sigmoid, g = tanh, c_t = f c_{t-1} + i g, h_t = o tanh(c_t)      (fused epilogue)
// and the mirrored step of the backward pass (dh_rec = dgates_{t+1} W_hh, then the gate
// derivatives), both directions in one launch.  Two execution forms with the same results:
//   * lstm_{fwd,bwd}_step_kernel: one launch per timestep (MIOpen spends 4 launches, 2 GEMMs + 2
//     pointwise, per step and direction here); no residency requirement, also captured as hipGraphs;
//   * lstm_{fwd,bwd}_persistent_kernel (default): ONE launch per layer and pass.  The workgroup's slice
//     of W_hh lives in registers for all T steps; steps are chained by write-through stores into a
//     tile-major hand-off copy (one contiguous KB per operand load), a drain, and per-chain arrival
//     counters.  All workgroups must be co-resident (host-checked); every spin is bounded.
// Environment: PTMI_LSTM_F32 (these exact-fp32 kernels instead of csrc/lstm_split.hip: the A/B reference),
// PTMI_LSTM_DBG (timing ablations - 16 no poll, 32 no drain, 64 no MFMA, 128 no operand loads, 256 no look-ahead loads:
// results void), PTMI_LSTM_MAX_POLLS (watchdog budget).  Round 2's tile / placement knobs are gone with round 3's prune.
//
// Layouts (rows = packed time-major rows of the PackedSequence, row(t, b) = offs[t] + b):
//   gx / gates / dgates  [rows][ndir][4][H]   (pre-activations in, activations out: in place)
//   hy, c, dhy           [rows][ndir][H]
//   w_hh_pad             [ndir][4H][KP]       KP = H rounded up to 16, zero padded
//   w_hh_t               [ndir][H][4H]
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "lstm_common.h"

// fragments of the backward K slice in flight per wavefront (8-wavefront kernel, see HALF there)
#ifndef PTMI_BWD_CA
#define PTMI_BWD_CA 5
#endif

// fragments of the forward K slice in flight per wavefront (us per step at B = 32 / 16 / 1: 5: 4.45 / 3.72 / 3.45,
// 3: 4.44 / 3.62 / 3.54, 2: 4.32 / 3.54 / 3.43, 1: 4.41 / 3.59 / 3.37)
#ifndef PTMI_FWD_CA
#define PTMI_FWD_CA 2
#endif

namespace ptmi {

// Per-direction bookkeeping of ONE timestep, computed on the host and passed as kernel arguments
// (a scalar load of batch_sizes[t] / offsets[t] from memory costs a cold round trip per launch).
struct StepMeta {
    long long row0[2];   // first packed row of this step's time index
    long long prow0[2];  // first packed row of the neighbouring time index (see kernels)
    long long qrow0[2];  // backward only: first row of the forward-sense predecessor
    int nb[2];           // active sequences at this time index
    int nprev[2];        // of those, how many are also active at the neighbouring time index
    int npv[2];          // backward only: how many have a forward-sense predecessor
};

struct LstmArgs {
    float* gx;
    float* hy;
    float* c;
    const float* w;
    int H, KP, ndir, dbg;
    StepMeta m;
    const float* c0;   // [ndir, max_batch, H] initial cell state of every sequence, or null (= 0)
    int max_batch;
};

// One forward timestep.  grid = (ceil(H / JT), ndir, ceil(maxB / 32)), NW * 64 threads.
// Workgroup tile: 32 batch rows x (4 gates x JT hidden units) = 32 x NC outputs, NC = 4 JT (16 or 32).
// K (= H, padded to KP) is split over the NW wavefronts; every wavefront issues the float4 loads of
// its whole K slice (CH 16-wide blocks) up front so that the L2 latency is paid once, not per block.
template <int JT, int NW, int CH>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_step_kernel(const LstmArgs A) {
    constexpr int NC = 4 * JT;          // gate columns per workgroup
    constexpr int NT = NC / 16;         // MFMA tiles along the gate columns
    static_assert(NC % 16 == 0, "gate columns must fill MFMA tiles");
    const int dir = blockIdx.y;
    const int j0 = blockIdx.x * JT;
    const int m0 = blockIdx.z * 32;
    const int nb = A.m.nb[dir];
    if (m0 >= nb) return;
    const long long row0 = A.m.row0[dir];
    const int nprev = A.m.nprev[dir];
    const long long prow0 = A.m.prow0[dir];
    const int H = A.H, G = 4 * H;
    const long long ld_g = (long long)A.ndir * G, ld_h = (long long)A.ndir * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, r = lane & 15;

    __shared__ float red[NW][32][NC + 1];
    const bool has_rec = nprev > m0 && !(A.dbg & 1);
    const int mtiles = (min(nprev, m0 + 32) - m0 + 15) >> 4;   // 1 or 2 live row tiles

    // epilogue operands first: their latency hides behind the GEMM
    const int bl = tid / JT, u = tid - bl * JT;
    const int b = m0 + bl;
    const bool act = tid < 32 * JT && b < nb && j0 + u < H;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    float cprev = 0.f;
    float* gp = A.gx + (row0 + b) * ld_g + (long long)dir * G + j0 + u;
    if (act) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = gp[q * H];
        if (b < nprev) cprev = A.c[(prow0 + b) * ld_h + dir * H + j0 + u];
        else if (A.c0) cprev = A.c0[((long long)dir * A.max_batch + b) * H + j0 + u];   // first step of sequence b
    }

    if (has_rec) {
        f32x4 acc[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nblk = A.KP >> 4;
        const int per = (nblk + NW - 1) / NW;
        const int kb0 = wave * per;
        const int kb1 = min(nblk, kb0 + per);
        const float* ap[2];
        bool av[2];
        const float* bp[NT];
        bool bv[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int i = m0 + mt * 16 + r;
            av[mt] = i < nprev;
            ap[mt] = A.hy + (prow0 + (av[mt] ? i : 0)) * ld_h + dir * H + 4 * g4;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cidx = nt * 16 + r;
            const int gate = cidx / JT, uu = cidx - gate * JT;
            bv[nt] = j0 + uu < H;
            bp[nt] = A.w + ((long long)dir * G + gate * H + (bv[nt] ? j0 + uu : 0)) * A.KP + 4 * g4;
        }
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = kb0; kc < kb1; kc += CH) {
            f32x4 a[CH][2], bq[CH][NT];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int kb = kc + i;
                const bool in = kb < kb1;
                const bool kin = in && (kb * 16 + 4 * g4 < H);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    a[i][mt] = (kin && av[mt]) ? *reinterpret_cast<const f32x4*>(ap[mt] + kb * 16) : zero;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bq[i][nt] = (in && bv[nt]) ? *reinterpret_cast<const f32x4*>(bp[nt] + kb * 16) : zero;
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (kc + i < kb1 && !(A.dbg & 2)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][0][q], bq[i][nt][q], acc[0][nt], 0, 0, 0);
                            if (mtiles > 1 || (A.dbg & 8))
                                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][1][q], bq[i][nt][q], acc[1][nt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // C layout of mfma_f32_16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wave][mt * 16 + g4 * 4 + q][nt * 16 + r] = acc[mt][nt][q];
        __syncthreads();
        if (tid < 32 * JT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cidx = q * JT + u;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) s += red[w][bl][cidx];
                pre[q] += s;
            }
        }
    }
    if (act) {
        const float ig = sigmoidf_(pre[0]);
        const float fg = sigmoidf_(pre[1]);
        const float gg = tanhf_(pre[2]);
        const float og = sigmoidf_(pre[3]);
        const float cn = fg * cprev + ig * gg;
        const float h = og * tanhf_(cn);
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
    int H, ndir;
    StepMeta m;
    const float* c0;
    int max_batch;
};

// One backward timestep.  grid = (ceil(H / 16), ceil(maxB / 16), ndir), NW * 64 threads.
// Workgroup tile: 16 batch rows x 16 hidden units of dh_rec = dgates_{next} W_hh, K = 4H split
// over NW wavefronts (each issues its whole K slice of float4 loads up front).
template <int NW, int CH>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_step_kernel(const LstmBwdArgs A) {
    const int dir = blockIdx.z;
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * 16;
    const int nb = A.m.nb[dir];
    if (m0 >= nb) return;
    const long long row0 = A.m.row0[dir];
    // "next" = the time index the previous launch processed (its dgates feed dh_rec);
    // "pv"   = the forward pass' predecessor (for c_{t-1})
    const int nnext = A.m.nprev[dir], npv = A.m.npv[dir];
    const long long nrow0 = A.m.prow0[dir], prow0 = A.m.qrow0[dir];
    const int H = A.H, G = 4 * H;
    const long long ld_g = (long long)A.ndir * G, ld_h = (long long)A.ndir * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, r = lane & 15;

    __shared__ float red[NW][16][17];
    const bool has_rec = nnext > m0;
    const int bl = (tid >> 4) & 15, jl = tid & 15;
    const int b = m0 + bl, j = n0 + jl;
    const bool act = tid < 256 && b < nb && j < H;

    // epilogue operands first
    float dh = 0.f, dc = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cn = 0.f, cprev = 0.f;
    const long long oh = (row0 + b) * ld_h + dir * H + j;
    const long long og_ = (row0 + b) * ld_g + (long long)dir * G + j;
    float* dcs = A.dcs + ((long long)b * A.ndir + dir) * H + j;
    if (act) {
        dh = A.dhy[oh];
        if (b < nnext) dc = *dcs;      // rows without a successor step start from dc = 0 (no memset needed)
        ig = A.gates[og_];
        fg = A.gates[og_ + H];
        gg = A.gates[og_ + 2 * H];
        og = A.gates[og_ + 3 * H];
        cn = A.c[oh];
        if (b < npv) cprev = A.c[(prow0 + b) * ld_h + dir * H + j];
        else if (A.c0) cprev = A.c0[((long long)dir * A.max_batch + b) * H + j];
    }

    if (has_rec) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nblk = G >> 4;
        const int per = (nblk + NW - 1) / NW;
        const int kb0 = wave * per;
        const int kb1 = min(nblk, kb0 + per);
        const bool av = m0 + r < nnext;
        const bool bv = n0 + r < H;
        const float* ap = A.dg + (nrow0 + (av ? m0 + r : 0)) * ld_g + (long long)dir * G + 4 * g4;
        const float* bp = A.wt + ((long long)dir * H + (bv ? n0 + r : 0)) * G + 4 * g4;
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = kb0; kc < kb1; kc += CH) {
            f32x4 a[CH], bq[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const bool in = kc + i < kb1;
                a[i] = (in && av) ? *reinterpret_cast<const f32x4*>(ap + (kc + i) * 16) : zero;
                bq[i] = (in && bv) ? *reinterpret_cast<const f32x4*>(bp + (kc + i) * 16) : zero;
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][q], bq[i][q], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave][g4 * 4 + q][r] = acc[q];
        __syncthreads();
        if (tid < 256 && b < nnext) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[w][bl][jl];
            dh += s;
        }
    }
    if (act) {
        const float tc = tanhf_(cn);
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

// (Rounds 1-4 kept a persistent EXACT-fp32 pair of kernels here - lstm_fwd_persistent_kernel / lstm_bwd_persistent_kernel, the flag
//  protocol on v_mfma_f32_16x16x4_f32 - reachable only through PTMI_LSTM_F32=1 once the split kernels of lstm_split.hip covered every
//  resident configuration.  Removed in round 5: PTMI_LSTM_F32=1 now selects the step-per-launch kernels above, which are exact fp32 too.)

static void neighbour(const int32_t* bs, const int64_t* offs, int T, int t, int tn, int* n, long long* row) {
    *n = 0;
    *row = 0;
    if (tn >= 0 && tn < T) {
        *n = bs[tn] < bs[t] ? bs[tn] : bs[t];
        *row = offs[tn];
    }
}

// Enqueue the T forward step kernels on `st` (eagerly, or into a stream capture).
static int enqueue_forward(float* gates, float* hy, float* c, const float* c0, const float* w_hh_pad, const int32_t* batch_sizes,
                           const int64_t* offsets, int T, int max_batch, int H, int KP, int ndir, hipStream_t st) {
    const char* dbg_env = getenv("PTMI_LSTM_DBG");
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    LstmArgs A{gates, hy, c, w_hh_pad, H, KP, ndir, dbg, {}, c0, max_batch};
    // JT = 8 (32 gate columns per workgroup) reads h_{t-1} from half as many workgroups as JT = 4;
    // measured faster at H = 600 (9.4 vs 10.3 us per step, B = 32) because the step is bound by
    // operand traffic, not by the matrix cores.  JT = 4 only when H is too small to fill the chip.
    const unsigned mz = (unsigned)((max_batch + 31) / 32);
    bool small_tiles = (long long)((H + 7) / 8) * ndir * mz < 64;
    if (dbg & 4) small_tiles = !small_tiles;
    for (int s = 0; s < T; ++s) {
        for (int d = 0; d < ndir; ++d) {
            const int t = d == 0 ? s : T - 1 - s;          // direction 1 walks time backwards
            A.m.nb[d] = batch_sizes[t];
            A.m.row0[d] = offsets[t];
            neighbour(batch_sizes, offsets, T, t, d == 0 ? t - 1 : t + 1, &A.m.nprev[d], &A.m.prow0[d]);
        }
        if (small_tiles)
            hipLaunchKernelGGL((lstm_fwd_step_kernel<4, 8, 5>), dim3((unsigned)((H + 3) / 4), (unsigned)ndir, mz),
                               dim3(512), 0, st, A);
        else
            hipLaunchKernelGGL((lstm_fwd_step_kernel<8, 8, 5>), dim3((unsigned)((H + 7) / 8), (unsigned)ndir, mz),
                               dim3(512), 0, st, A);
    }
    return launch_status();
}

static int enqueue_backward(const float* gates, const float* c, const float* c0, const float* dhy, const float* w_hh_t,
                            float* dgates, float* dc_state, const int32_t* batch_sizes, const int64_t* offsets,
                            int T, int max_batch, int H, int ndir, hipStream_t st) {
    LstmBwdArgs A{gates, c, dhy, w_hh_t, dgates, dc_state, H, ndir, {}, c0, max_batch};
    const dim3 grid((unsigned)((H + 15) / 16), (unsigned)((max_batch + 15) / 16), (unsigned)ndir);
    for (int s = 0; s < T; ++s) {
        for (int d = 0; d < ndir; ++d) {
            const int t = d == 0 ? T - 1 - s : s;          // reverse of the forward order
            A.m.nb[d] = batch_sizes[t];
            A.m.row0[d] = offsets[t];
            neighbour(batch_sizes, offsets, T, t, d == 0 ? t + 1 : t - 1, &A.m.nprev[d], &A.m.prow0[d]);
            neighbour(batch_sizes, offsets, T, t, d == 0 ? t - 1 : t + 1, &A.m.npv[d], &A.m.qrow0[d]);
        }
        hipLaunchKernelGGL((lstm_bwd_step_kernel<16, 10>), grid, dim3(1024), 0, st, A);
    }
    return launch_status();
}

}  // namespace ptmi

using namespace ptmi;

// compute units of the current device (the residency limits of the persistent kernels derive from it)
static int cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// Per-device word that counts timed-out persistent launches (set once by the host binding; see ptmi_lstm_set_error_sink).
static unsigned* g_error_sink[64] = {};
static unsigned* error_sink() {
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? g_error_sink[dev] : nullptr;
}

static int lstm_backward_persistent_impl(const float* gates, const float* c, const float* c0, const float* dhy,
                                         const float* w_hh_t, float* dgates, uint16_t* dgates_t, const int32_t* batch_sizes_dev,
                                         const int64_t* offsets_dev, const uint64_t* step_masks, uint32_t* flags, float* dc_carry,
                                         int32_t T, int32_t max_batch, int64_t rows, int32_t H, int32_t ndir, int32_t s_begin,
                                         int32_t s_end, int32_t prefilled, ptmi_stream_t stream, const float* dc_n = nullptr);

extern "C" {

int ptmi_lstm_forward(float* gates, float* hy, float* c, const float* c0, const float* w_hh_pad, const int32_t* batch_sizes,
                      const int64_t* offsets, int32_t T, int32_t max_batch, int32_t H, int32_t KP,
                      int32_t ndir, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!gates || !hy || !c || !w_hh_pad || !batch_sizes || !offsets, PTMI_E_INVALID);
    PTMI_RETURN_IF(T < 0 || max_batch < 1 || H < 1 || (ndir != 1 && ndir != 2), PTMI_E_INVALID);
    PTMI_RETURN_IF(H % 4 != 0 || KP % 16 != 0 || KP < H, PTMI_E_UNSUPPORTED);
    return enqueue_forward(gates, hy, c, c0, w_hh_pad, batch_sizes, offsets, T, max_batch, H, KP, ndir,
                           static_cast<hipStream_t>(stream));
}

int ptmi_lstm_backward(const float* gates, const float* c, const float* c0, const float* dhy, const float* w_hh_t, float* dgates,
                       float* dc_state, const int32_t* batch_sizes, const int64_t* offsets, int32_t T,
                       int32_t max_batch, int32_t H, int32_t ndir, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!gates || !c || !dhy || !w_hh_t || !dgates || !dc_state || !batch_sizes || !offsets,
                   PTMI_E_INVALID);
    PTMI_RETURN_IF(T < 0 || max_batch < 1 || H < 1 || (ndir != 1 && ndir != 2), PTMI_E_INVALID);
    PTMI_RETURN_IF(H % 4 != 0, PTMI_E_UNSUPPORTED);
    return enqueue_backward(gates, c, c0, dhy, w_hh_t, dgates, dc_state, batch_sizes, offsets, T, max_batch, H, ndir,
                            static_cast<hipStream_t>(stream));
}

int64_t ptmi_lstm_flags_elems(int32_t T, int32_t ndir, int32_t max_batch) {
    (void)T;
    return (int64_t)ndir * ((max_batch + 15) / 16) * kSlots + 8;   // one chain per 16-row tile + error words
}

// tile-major hand-off copy: [T][16-row tiles][ndir][cols / 16] tiles of 16 x 16 floats
static int64_t lstm_tile_elems(int32_t T, int32_t ndir, int32_t max_batch, int32_t cols) {
    return (int64_t)T * ((max_batch + 15) / 16) * ndir * cols * 16;
}

int64_t ptmi_lstm_scratch_elems(int32_t T, int32_t ndir, int32_t max_batch, int32_t H, int32_t backward) {
    // [tile-major hand-off copy (columns rounded up to 32) | backward: bias gradient [ndir][4H] + 8 words (word 0: max
    // |dgates| as float bits) | hand-off slots | 8 error words]
    return lstm_tile_elems(T, ndir, max_batch, backward ? (4 * H + 31) / 32 * 32 : (H + 31) / 32 * 32) +
           (backward ? (int64_t)ndir * 4 * H + 8 : 0) + ptmi_lstm_flags_elems(T, ndir, max_batch);
}

int ptmi_lstm_set_error_sink(uint32_t* word) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    PTMI_RETURN_IF(dev < 0 || dev >= 64, PTMI_E_UNSUPPORTED);
    g_error_sink[dev] = word;
    return PTMI_OK;
}

int ptmi_lstm_split_enabled(void) { return getenv("PTMI_LSTM_F32") ? 0 : 1; }

static bool fwd_uses_daf(int max_batch, int H, int ndir);
static void fwd_tile_shape(int max_batch, int H, int ndir, bool split, int* jt_out, int* mtl_out);

int ptmi_lstm_forward_fills(int32_t T, int32_t ndir, int32_t max_batch, int32_t H) {
    if (T < 1 || max_batch < 1 || H < 1 || H % 4 != 0 || (ndir != 1 && ndir != 2) || !ptmi_lstm_split_enabled()) return 0;
    const int G32 = (4 * H + 31) / 32 * 32;
    if ((G32 / 32 + 7) / 8 > 10 || !bwd_daf_applies() || !fwd_uses_daf(max_batch, H, ndir)) return 0;
    // one forward launch (all row tiles resident at once), a wavefront without elements in its workgroups
    int jt, mtl;
    fwd_tile_shape(max_batch, H, ndir, true, &jt, &mtl);
    const int ntiles = (max_batch + 16 * mtl - 1) / (16 * mtl);
    const int jx = (H + jt - 1) / jt;
    return ((long long)jx * ndir * ntiles <= cu_count() && 16 * mtl * jt <= 7 * 64) ? 2 : 0;      // 2: planes AND the words behind them
}

int ptmi_lstm_scratch_prefill(uint32_t* scratch, int32_t T, int32_t ndir, int32_t max_batch, int32_t H, int32_t backward,
                              ptmi_stream_t stream) {
    PTMI_RETURN_IF(!scratch || T < 1 || max_batch < 1 || H < 1 || (ndir != 1 && ndir != 2), PTMI_E_INVALID);
    if (H % 4 != 0 || !ptmi_lstm_split_enabled()) return 0;
    int cols;
    if (backward) {
        const int G32 = (4 * H + 31) / 32 * 32;
        if ((G32 / 32 + 7) / 8 > 10 || !bwd_daf_applies()) return 0;
        cols = G32;
    } else {
        if (!fwd_uses_daf(max_batch, H, ndir)) return 0;
        cols = (H + 31) / 32 * 32;
    }
    const int rc = daf_prefill(scratch, (size_t)lstm_tile_elems(T, ndir, max_batch, cols), static_cast<hipStream_t>(stream));
    return rc ? rc : 1;
}

int32_t ptmi_lstm_handoff_cols(int32_t H, int32_t backward) {
    if (H < 1 || H % 4 != 0 || !ptmi_lstm_split_enabled()) return 0;
    if (backward) {
        const int G32 = (4 * H + 31) / 32 * 32;
        return (G32 / 32 + 7) / 8 <= 10 ? G32 : 0;           // the rule of ptmi_lstm_backward_persistent_range
    }
    const int KP32 = (H + 31) / 32 * 32;
    return (KP32 / 32 + 7) / 8 <= 3 ? KP32 : 0;              // the rule of ptmi_lstm_forward_persistent
}

// Workgroup tile of the persistent forward launch (jt hidden units x 16 mtl rows) for a batch / layer size.
static void fwd_tile_shape(int max_batch, int H, int ndir, bool split, int* jt_out, int* mtl_out) {
    int jt = max_batch <= 16 ? 8 : 12, mtl = max_batch <= 32 ? 1 : 2;
    // split kernels, one 16-row tile per workgroup: 16 units (38 instead of 50 workgroups per chain at H = 600) measured
    // 3.19 against 3.27 us per step with the fragment-order hand-off copy, and leaves 48 more CUs to other queues
    // (flag-protocol kernels only; with the data-as-flag hand-off 12 units measure 2.48 against 2.55)
    if (jt == 12 && (long long)((H + 11) / 12) * ndir > cu_count()) jt = 16;      // wide tiles: one workgroup per CU
    if (jt == 16 && (long long)((H + 15) / 16) * ndir > cu_count()) jt = 8;
    *jt_out = jt;
    *mtl_out = mtl;
}

// Does the persistent forward launch of this configuration hand its rows on by the data-as-flag protocol (planes pre-filled
// with the pattern)?  One answer for all launches of a call (later launches have at most as many row tiles as the first).
static bool fwd_uses_daf(int max_batch, int H, int ndir) {
    const int KP32 = (H + 31) / 32 * 32;
    const bool split = ptmi_lstm_split_enabled() && (KP32 / 32 + 7) / 8 <= 3;
    if (!split) return false;
    int jt, mtl;
    fwd_tile_shape(max_batch, H, ndir, split, &jt, &mtl);
    const bool wide = jt >= 12;
    const int ntiles = (max_batch + 16 * mtl - 1) / (16 * mtl);
    const int jx = wide ? (H + jt - 1) / jt : (H + 7) / 8;
    const int cus = cu_count();
    const int cap = wide ? cus : cus * 7 / 4;
    if ((long long)jx * ndir > cap) return false;
    const int per_launch = std::min(ntiles, cap / (jx * ndir));
    return fwd_daf_applies(jt, mtl == 1, (long long)jx * ndir * per_launch <= cus);
}

int ptmi_lstm_forward_persistent(float* gates, float* hy, float* c, const float* c0, const float* w_hh_pad,
                                 const uint32_t* w_hh_amax, const int32_t* batch_sizes_dev, const int64_t* offsets_dev,
                                 uint32_t* flags, int32_t T, int32_t max_batch, int64_t rows, int32_t H, int32_t KP,
                                 int32_t ndir, int32_t prefilled, uint32_t* backward_scratch, ptmi_stream_t stream) {
    return ptmi_lstm_forward_persistent_slots(gates, hy, c, c0, w_hh_pad, w_hh_amax, batch_sizes_dev, offsets_dev, nullptr, flags, T,
                                              max_batch, rows, H, KP, ndir, prefilled, backward_scratch, stream);
}

int ptmi_lstm_forward_persistent_slots(float* gates, float* hy, float* c, const float* c0, const float* w_hh_pad,
                                       const uint32_t* w_hh_amax, const int32_t* batch_sizes_dev, const int64_t* offsets_dev,
                                       const uint64_t* step_masks, uint32_t* flags, int32_t T, int32_t max_batch, int64_t rows,
                                       int32_t H, int32_t KP, int32_t ndir, int32_t prefilled, uint32_t* backward_scratch,
                                       ptmi_stream_t stream) {
    PTMI_RETURN_IF(!gates || !hy || !c || !w_hh_pad || !batch_sizes_dev || !offsets_dev || !flags, PTMI_E_INVALID);
    // row slots: every (time index, slot) row exists in the buffers; at most 64 slots (one mask word per step and kind)
    PTMI_RETURN_IF(step_masks && (rows != (int64_t)T * max_batch || max_batch > 64 || c0), PTMI_E_UNSUPPORTED);
    PTMI_RETURN_IF(T < 1 || max_batch < 1 || H < 1 || (ndir != 1 && ndir != 2) || rows < 1, PTMI_E_INVALID);
    PTMI_RETURN_IF(H % 4 != 0 || KP % 16 != 0 || KP < H, PTMI_E_UNSUPPORTED);
    constexpr int NW = 8;
    // the resident W slice must fit 3 k blocks of 32 per wavefront; all workgroups must be co-resident
    const int KP32 = (H + 31) / 32 * 32;
    const bool split = ptmi_lstm_split_enabled() && (KP32 / 32 + NW - 1) / NW <= 3;      // split kernels: <= 3 k blocks of 32 per wavefront
    PTMI_RETURN_IF(!split, PTMI_E_UNSUPPORTED);                     // the caller falls back to the step-per-launch kernels
    // 16-row workgroups (two interleaved chains per 32 rows) while all of them stay co-resident
    // (512 threads at <= 128 VGPRs: 2 per CU; keep a margin below 256 x 2), else 32-row workgroups
    const int jx8 = (H + 7) / 8;
    const long long hy_bytes = rows * ndir * H * 4;
    PTMI_RETURN_IF(hy_bytes > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    // Workgroup tile (rows x hidden units), measured forward us per step at H = 600, T = 253:
    //   B <= 16: 16 x 8 (4.9);  B <= 32: two independent chains of 16 x 12 on separate CUs (200 workgroups:
    //   5.4; 16 x 16 on 152: 5.8; 32 x 8: 6.6; 16 x 8 with 300 workgroups sharing CUs: 8.3);
    //   B > 32: 32 x 12 (B = 64: 32 x 16 8.8, 32 x 8 11.6).
    int jt, mtl;
    fwd_tile_shape(max_batch, H, ndir, split, &jt, &mtl);
    const bool small = mtl == 1, wide = jt >= 12;
    const int ntiles = (max_batch + 16 * mtl - 1) / (16 * mtl);
    const int jx = wide ? (H + jt - 1) / jt : jx8;
    const int cus = cu_count();
    const int cap = wide ? cus : cus * 7 / 4;     // 12/16-unit tiles need the whole register file: one workgroup per CU
    PTMI_RETURN_IF((long long)jx * ndir > cap || jx > kSlots, PTMI_E_UNSUPPORTED);
    // row tiles are independent recurrences: a batch whose tiles do not all fit runs as several launches
    const int per_launch = std::min(ntiles, cap / (jx * ndir));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // scratch = [tile-major hy | arrival counters | 8 error words]; only the counters need zeroing
    PTMI_RETURN_IF(KP != (H + 15) / 16 * 16, PTMI_E_UNSUPPORTED);
    float* const hyt = reinterpret_cast<float*>(flags);
    flags += lstm_tile_elems(T, ndir, max_batch, KP32);
    const bool daf = split && fwd_uses_daf(max_batch, H, ndir);
    PTMI_RETURN_IF(!daf, PTMI_E_UNSUPPORTED);          // no data-as-flag instantiation for this tile shape: the step-per-launch kernels
    PTMI_RETURN_IF(step_masks && !daf, PTMI_E_UNSUPPORTED);          // the data-as-flag kernels carry the row masks
    if (daf && !prefilled) {        // every 16-bit value of the planes = 0xFFFF (no value can be), the counters behind them zero: one launch
        int fe = daf_prefill_and_zero(hyt, (size_t)lstm_tile_elems(T, ndir, max_batch, KP32), flags, (size_t)ptmi_lstm_flags_elems(T, ndir, max_batch), st);
        if (fe) return fe;
    } else {
        hipError_t e = zero_words_async(flags, (size_t)ptmi_lstm_flags_elems(T, ndir, max_batch), st);
        if (e != hipSuccess) return (int)e;
    }
    LstmPersistArgs A{gates, hy, c, w_hh_pad, batch_sizes_dev, offsets_dev, flags, T, H, KP, ndir,
                      (unsigned)jx, getenv("PTMI_LSTM_MAX_POLLS") ? (unsigned)atoi(getenv("PTMI_LSTM_MAX_POLLS")) : 1u << 22, (int)hy_bytes,
                      (unsigned)(ptmi_lstm_flags_elems(T, ndir, max_batch) - 8),
                      getenv("PTMI_LSTM_DBG") ? atoi(getenv("PTMI_LSTM_DBG")) : 0, 0, ntiles, c0, max_batch, hyt, (max_batch + 15) / 16,
                      w_hh_amax, KP32};
    A.err_sink = error_sink();
    A.uniform = (rows == (int64_t)T * max_batch) ? 1 : 0;      // batch sizes never grow: equal lengths
    A.masks = reinterpret_cast<const unsigned long long*>(step_masks);
    if (backward_scratch && ptmi_lstm_forward_fills(T, ndir, max_batch, H)) {     // this layer's backward planes get their pattern here
        const int G32 = (4 * H + 31) / 32 * 32;
        A.fill_ptr = reinterpret_cast<uint4*>(backward_scratch);
        A.fill_n16 = (unsigned long long)lstm_tile_elems(T, ndir, max_batch, G32) / 4;
        // the words behind the planes: [ndir][4H] bias sums, 8 words, arrival slots + error words - a multiple of 4 words (kSlots is)
        A.zero_n16 = (unsigned long long)((long long)ndir * 4 * H + 8 + ptmi_lstm_flags_elems(T, ndir, max_batch)) / 4;
    }
    for (int t0 = 0; t0 < ntiles; t0 += per_launch) {
        A.tile0 = t0;
        const int nt = std::min(per_launch, ntiles - t0);
        const dim3 grid((unsigned)jx, (unsigned)ndir, (unsigned)nt);
        const bool one_per_cu = (long long)jx * ndir * nt <= cus;
        // one workgroup per CU: the workgroups of a chain (direction x row tile) on 8 / chains neighbouring XCDs, as in the
        // backward kernel (a chain's hand-off rows and slots then live in the L2s of those XCDs only)
        const int chains = ndir * nt;
        A.span = (one_per_cu && chains <= 8 && 8 % chains == 0 && (jx + 8 / chains - 1) / (8 / chains) * 8 <= cus) ? 8 / chains : 0;
        A.nx = jx;
        A.nt = nt;
        const dim3 grid1(A.span ? (unsigned)((jx + A.span - 1) / A.span * 8) : 0u);
        int rc = launch_fwd_split(A, jt, small, one_per_cu, A.span ? grid1 : grid, st, daf);
        if (rc) return rc;
    }
    return PTMI_OK;
}

int ptmi_lstm_backward_persistent(const float* gates, const float* c, const float* c0, const float* dhy, const float* w_hh_t,
                                  float* dgates, const int32_t* batch_sizes_dev, const int64_t* offsets_dev,
                                  uint32_t* flags, int32_t T, int32_t max_batch, int64_t rows, int32_t H,
                                  int32_t ndir, int32_t prefilled, ptmi_stream_t stream) {
    return ptmi_lstm_backward_persistent_range(gates, c, c0, dhy, w_hh_t, dgates, batch_sizes_dev, offsets_dev, flags, nullptr, T,
                                               max_batch, rows, H, ndir, 0, T, prefilled, stream);
}

int ptmi_lstm_backward_persistent_range(const float* gates, const float* c, const float* c0, const float* dhy,
                                        const float* w_hh_t, float* dgates, const int32_t* batch_sizes_dev,
                                        const int64_t* offsets_dev, uint32_t* flags, float* dc_carry, int32_t T,
                                        int32_t max_batch, int64_t rows, int32_t H, int32_t ndir, int32_t s_begin, int32_t s_end,
                                        int32_t prefilled, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!dgates, PTMI_E_INVALID);
    return ptmi_lstm_backward_persistent_planes(gates, c, c0, dhy, w_hh_t, dgates, nullptr, batch_sizes_dev, offsets_dev, flags, dc_carry,
                                                T, max_batch, rows, H, ndir, s_begin, s_end, prefilled, stream);
}

int32_t ptmi_lstm_backward_planes_ok(int32_t T, int32_t ndir, int32_t max_batch, int64_t rows, int32_t H) {
    if (T < 1 || max_batch < 1 || H < 1 || H % 4 != 0 || (ndir != 1 && ndir != 2) || !ptmi_lstm_split_enabled()) return 0;
    if (rows != (int64_t)T * max_batch || max_batch % 16 != 0) return 0;             // equal lengths, whole 16-row tiles
    const int G32 = (4 * H + 31) / 32 * 32;
    if ((G32 / 32 + 7) / 8 > 10 || !bwd_daf_applies()) return 0;                     // the data-as-flag split kernels run
    const int nx = (H + 15) / 16;
    return ((long long)nx * ndir <= cu_count() - 16 && nx <= kSlots) ? 1 : 0;
}

// the half k block behind the last packed row (rows % 32 == 16) of every column tile and plane: zero
__global__ void zero_tp_tail_kernel(uint4* planes, long long tiles, int kb_total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // (tile, plane, chunk 32..63)
    if (i >= tiles * 64) return;
    const long long tile = i >> 6;
    const int plane = (int)(i >> 5) & 1, ch = 32 + (int)(i & 31);
    planes[((tile * kb_total + kb_total - 1) * 2 + plane) * 64 + ch] = make_uint4(0u, 0u, 0u, 0u);
}

int ptmi_lstm_backward_persistent_planes(const float* gates, const float* c, const float* c0, const float* dhy,
                                         const float* w_hh_t, float* dgates, uint16_t* dgates_t, const int32_t* batch_sizes_dev,
                                         const int64_t* offsets_dev, uint32_t* flags, float* dc_carry, int32_t T,
                                         int32_t max_batch, int64_t rows, int32_t H, int32_t ndir, int32_t s_begin, int32_t s_end,
                                         int32_t prefilled, ptmi_stream_t stream) {
    return lstm_backward_persistent_impl(gates, c, c0, dhy, w_hh_t, dgates, dgates_t, batch_sizes_dev, offsets_dev, nullptr, flags, dc_carry,
                                         T, max_batch, rows, H, ndir, s_begin, s_end, prefilled, stream);
}

int ptmi_lstm_backward_persistent_states(const float* gates, const float* c, const float* c0, const float* dhy, const float* dc_n,
                                         const float* w_hh_t, float* dgates, const int32_t* batch_sizes_dev,
                                         const int64_t* offsets_dev, uint32_t* flags, float* dc_0, int32_t T, int32_t max_batch,
                                         int64_t rows, int32_t H, int32_t ndir, int32_t prefilled, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!dgates, PTMI_E_INVALID);
    return lstm_backward_persistent_impl(gates, c, c0, dhy, w_hh_t, dgates, nullptr, batch_sizes_dev, offsets_dev, nullptr, flags, dc_0, T,
                                         max_batch, rows, H, ndir, 0, T, prefilled, stream, dc_n);
}

int ptmi_lstm_backward_persistent_slots(const float* gates, const float* c, const float* dhy, const float* w_hh_t, float* dgates,
                                        uint16_t* dgates_t, const int32_t* batch_sizes_dev, const int64_t* offsets_dev,
                                        const uint64_t* step_masks, uint32_t* flags, int32_t T, int32_t max_batch, int64_t rows,
                                        int32_t H, int32_t ndir, int32_t prefilled, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!step_masks || (!dgates && !dgates_t), PTMI_E_INVALID);
    PTMI_RETURN_IF(rows != (int64_t)T * max_batch || max_batch > 64, PTMI_E_UNSUPPORTED);
    return lstm_backward_persistent_impl(gates, c, nullptr, dhy, w_hh_t, dgates, dgates_t, batch_sizes_dev, offsets_dev, step_masks, flags,
                                         nullptr, T, max_batch, rows, H, ndir, 0, T, prefilled, stream);
}

}  // extern "C"

static int lstm_backward_persistent_impl(const float* gates, const float* c, const float* c0, const float* dhy,
                                         const float* w_hh_t, float* dgates, uint16_t* dgates_t, const int32_t* batch_sizes_dev,
                                         const int64_t* offsets_dev, const uint64_t* step_masks, uint32_t* flags, float* dc_carry,
                                         int32_t T, int32_t max_batch, int64_t rows, int32_t H, int32_t ndir, int32_t s_begin,
                                         int32_t s_end, int32_t prefilled, ptmi_stream_t stream, const float* dc_n) {
    PTMI_RETURN_IF(!gates || !c || !dhy || !w_hh_t || (!dgates && !dgates_t) || !batch_sizes_dev || !offsets_dev || !flags,
                   PTMI_E_INVALID);
    PTMI_RETURN_IF(dgates_t && (reinterpret_cast<uintptr_t>(dgates_t) & 15) != 0, PTMI_E_INVALID);
    PTMI_RETURN_IF(T < 1 || max_batch < 1 || H < 1 || (ndir != 1 && ndir != 2) || rows < 1, PTMI_E_INVALID);
    PTMI_RETURN_IF(s_begin < 0 || s_end > T || s_begin >= s_end, PTMI_E_INVALID);
    const bool whole = s_begin == 0 && s_end == T;
    PTMI_RETURN_IF(!whole && !dc_carry, PTMI_E_INVALID);
    PTMI_RETURN_IF(H % 4 != 0, PTMI_E_UNSUPPORTED);
    const int G32 = (4 * H + 31) / 32 * 32;
    const bool split = ptmi_lstm_split_enabled() && (G32 / 32 + 7) / 8 <= 10;           // split kernel: <= 10 k blocks of 32 per wavefront
    PTMI_RETURN_IF(!split, PTMI_E_UNSUPPORTED);                     // the caller falls back to the step-per-launch kernels
    const int resident = cu_count() - 16;     // one workgroup per CU, with a margin
    // One workgroup per CU must be resident (at most 240 per launch).  Row tiles are independent recurrences,
    // so a batch whose tiles do not fit at once runs as several launches over groups of tiles; before that,
    // 16-row chains become 32-row chains (8-wavefront workgroups, MTL = 2: one launch up to batch 64 at
    // H = 600; 7.5 us per step instead of 2 x 5.0).
    const int nx = (H + 15) / 16, nt16 = (max_batch + 15) / 16;
    PTMI_RETURN_IF((long long)nx * ndir > resident || nx > kSlots, PTMI_E_UNSUPPORTED);
    int mtl = nt16 > resident / (nx * ndir) ? 2 : 1;
    const int ntiles = (nt16 + mtl - 1) / mtl;
    const int per_launch = std::min(ntiles, resident / (nx * ndir));
    const long long dg_bytes = rows * ndir * 4 * H * 4;
    PTMI_RETURN_IF(dg_bytes > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // scratch = [tile-major dgates | bias gradient [ndir][4H] | arrival counters | 8 error words];
    // the bias gradient and the counters are zeroed here (one memset)
    float* const dgt = reinterpret_cast<float*>(flags);
    flags += lstm_tile_elems(T, ndir, max_batch, G32);
    float* const dbias = reinterpret_cast<float*>(flags);
    if (s_begin == 0) {         // a later range continues on the first one's counters, bias sums and maximum
        const size_t nz = (size_t)(ndir * 4 * H + 8 + ptmi_lstm_flags_elems(T, ndir, max_batch));
        if (split && bwd_daf_applies() && !prefilled) {       // data-as-flag hand-off: the planes start as the fill pattern (same launch)
            int fe = daf_prefill_and_zero(dgt, (size_t)lstm_tile_elems(T, ndir, max_batch, G32), flags, nz, st);
            if (fe) return fe;
        } else if (prefilled != 2) {       // (2: the forward launch has zeroed these words with the pattern fill - ptmi_lstm_forward_fills)
            hipError_t e = zero_words_async(flags, nz, st);
            if (e != hipSuccess) return (int)e;
        }
    }
    flags += ndir * 4 * H;
    uint32_t* const dg_amax = flags;
    flags += 8;
    LstmPersistBwdArgs A{gates, c, dhy, w_hh_t, dgates, batch_sizes_dev, offsets_dev, flags, T, H, ndir,
                         (unsigned)nx, getenv("PTMI_LSTM_MAX_POLLS") ? (unsigned)atoi(getenv("PTMI_LSTM_MAX_POLLS")) : 1u << 22, (int)dg_bytes,
                         (unsigned)(ptmi_lstm_flags_elems(T, ndir, max_batch) - 8), 0, ntiles,
                         getenv("PTMI_LSTM_DBG") ? atoi(getenv("PTMI_LSTM_DBG")) : 0, c0, max_batch, 0, 0, 0, dgt, nt16, dbias,
                         split ? dg_amax : nullptr, G32};
    A.err_sink = error_sink();
    A.uniform = (rows == (int64_t)T * max_batch) ? 1 : 0;
    A.masks = reinterpret_cast<const unsigned long long*>(step_masks);
    PTMI_RETURN_IF(step_masks && !(split && bwd_daf_applies()), PTMI_E_UNSUPPORTED);
    PTMI_RETURN_IF(dc_n && (!split || step_masks), PTMI_E_UNSUPPORTED);
    A.dcn = dc_n;
    A.s_begin = s_begin;
    A.s_end = s_end;
    A.dc_carry = dc_carry;
    if (dgates_t) {
        PTMI_RETURN_IF(!split || !ptmi_lstm_backward_planes_ok(T, ndir, max_batch, rows, H), PTMI_E_UNSUPPORTED);
        // the planes hold the rows of THIS launch's step range: (s_end - s_begin) * max_batch packed rows per direction, from
        // time index T - s_end on (forward direction: processed last to first) / s_begin on (reverse direction)
        const int64_t range_rows = (int64_t)(s_end - s_begin) * max_batch;
        A.dgtp = reinterpret_cast<uint4*>(dgates_t);
        A.tp_kb = (int)((range_rows + 31) / 32);
        A.tp_dir_stride = (long long)(4 * H / 16) * A.tp_kb * 2 * 64;
        A.tp_row0[0] = (long long)(T - s_end) * max_batch;
        A.tp_row0[1] = (long long)s_begin * max_batch;
        if (range_rows % 32 != 0) {
            const long long tiles = (long long)ndir * (4 * H / 16);
            hipLaunchKernelGGL(zero_tp_tail_kernel, dim3((unsigned)((tiles * 64 + 255) / 256)), dim3(256), 0, st, A.dgtp, tiles, A.tp_kb);
            int rc = launch_status();
            if (rc) return rc;
        }
    }
    for (int t0 = 0; t0 < ntiles; t0 += per_launch) {
        A.tile0 = t0;
        const int nt = std::min(per_launch, ntiles - t0);
        const int chains = nt * ndir;
        A.nx = nx;
        A.nt = nt;
        A.span = (chains <= 8 && 8 % chains == 0) ? 8 / chains : 0;
        const unsigned nwg = A.span ? (unsigned)((nx + A.span - 1) / A.span * 8) : (unsigned)(nx * chains);
        int rc = launch_bwd_split(A, mtl, nwg, st);
        if (rc) return rc;
    }
    return PTMI_OK;
}
