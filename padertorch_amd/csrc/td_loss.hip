// Time-domain regression losses (padertorch/ops/losses/regression.py:47-378: mse / log-mse /
// log1p-mse / SDR / SI-SDR / source-aggregated SDR) and their use under pit_loss
// (ops/losses/source_separation.py:34-124; TasNet loss, contrib/.../tasnet/model.py:154-176).
//
// Every one of these losses between estimate row i and target row j is a closed form of five sums
// over time:  See_i = sum e_i^2,  Stt_j = sum t_j^2,  Set_ij = sum e_i t_j,  Se_i = sum e_i,
// St_j = sum t_j  (||e - t||^2 = See - 2 Set + Stt; SI-SDR: alpha = Set / Stt, ...).  So ONE streaming
// pass over the 2K rows of an example yields everything all K! permutations of all loss variants
// need (HBM bound: 2*K*T*4 bytes per example, read once), and the backward pass is one more streaming
// pass  d e_i = A_i e_i + sum_j B_ij t_j + C_i  whose K*(K+2) coefficients come from differentiating
// the closed forms (done on the host side on [B, K*K+4K] tensors).
//
// Products of two fp32 values are exact in fp64 and the sums are kept in fp64 (per thread -> wave ->
// workgroup -> fixed-order reduction over chunks: bitwise reproducible), so cancellation in
// See - Set^2/Stt (high SI-SDR) costs nothing; the fp64 vector rate is far above what 8 TB/s of
// fp32 input needs (K^2 + 4K fused multiply-adds per 2K loaded values).
#include "common.h"

namespace ptmi {

constexpr int kTdMaxK = 8;

struct TdArgs {
    const float* est;
    const float* tgt;
    const int32_t* lengths;   // [B] valid samples per example, or null (= T)
    long long T;
    long long eb, ek, tb, tk;   // strides in elements; time is contiguous
    long long chunk;            // samples per workgroup (multiple of 4)
    int K, nchunks, vec;        // vec: every row start is 16-byte aligned
    double* ws;                 // [B][nchunks][NS]
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// stats layout per example: Set[K][K] | See[K] | Stt[K] | Se[K] | St[K]
template <int K>
__global__ __launch_bounds__(256) void td_stats_kernel(const TdArgs A) {
    constexpr int NS = K * K + 4 * K;
    const int b = blockIdx.y, c = blockIdx.x;
    const long long len = A.lengths ? min((long long)A.lengths[b], A.T) : A.T;
    const long long t0 = (long long)c * A.chunk;
    const long long t1 = min(t0 + A.chunk, len);
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;
    const float* __restrict__ e = A.est + (long long)b * A.eb;
    const float* __restrict__ t = A.tgt + (long long)b * A.tb;
    auto add = [&](const float (&ev)[K], const float (&tv)[K]) {
        double ed[K], td[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ed[k] = (double)ev[k];
            td[k] = (double)tv[k];
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
#pragma unroll
            for (int j = 0; j < K; ++j) acc[i * K + j] = fma(ed[i], td[j], acc[i * K + j]);
            acc[K * K + i] = fma(ed[i], ed[i], acc[K * K + i]);
            acc[K * K + K + i] = fma(td[i], td[i], acc[K * K + K + i]);
            acc[K * K + 2 * K + i] += ed[i];
            acc[K * K + 3 * K + i] += td[i];
        }
    };
    if (A.vec) {
        for (long long i = t0 + 4 * threadIdx.x; i < t1; i += 4 * 256) {
            if (i + 4 <= t1) {
                float4 ev4[K], tv4[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    ev4[k] = *reinterpret_cast<const float4*>(e + k * A.ek + i);
                    tv4[k] = *reinterpret_cast<const float4*>(t + k * A.tk + i);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float ev[K], tv[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        ev[k] = reinterpret_cast<const float*>(&ev4[k])[q];
                        tv[k] = reinterpret_cast<const float*>(&tv4[k])[q];
                    }
                    add(ev, tv);
                }
            } else {   // ragged tail of the example: element-wise
                for (long long ii = i; ii < t1; ++ii) {
                    float ev[K], tv[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        ev[k] = e[k * A.ek + ii];
                        tv[k] = t[k * A.tk + ii];
                    }
                    add(ev, tv);
                }
            }
        }
    } else {
        for (long long i = t0 + threadIdx.x; i < t1; i += 256) {
            float ev[K], tv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                ev[k] = e[k * A.ek + i];
                tv[k] = t[k * A.tk + i];
            }
            add(ev, tv);
        }
    }
    __shared__ double red[4][NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double v = wave_sum(acc[s]);
        if (lane == 0) red[wave][s] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS)
        A.ws[((long long)b * A.nchunks + c) * NS + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// K in 5..8: one workgroup per (chunk, example, estimate row i): Set[i][:], See_i, Se_i and, for
// i == 0, Stt / St.  Target rows are re-read K times (rare configuration).
__global__ __launch_bounds__(256) void td_stats_row_kernel(const TdArgs A) {
    const int K = A.K, NS = K * K + 4 * K;
    const int b = blockIdx.y, c = blockIdx.x, i = blockIdx.z;
    const long long len = A.lengths ? min((long long)A.lengths[b], A.T) : A.T;
    const long long t0 = (long long)c * A.chunk;
    const long long t1 = min(t0 + A.chunk, len);
    double set[kTdMaxK], stt[kTdMaxK], st[kTdMaxK], see = 0.0, se = 0.0;
#pragma unroll
    for (int j = 0; j < kTdMaxK; ++j) set[j] = stt[j] = st[j] = 0.0;
    const float* __restrict__ e = A.est + (long long)b * A.eb + i * A.ek;
    const float* __restrict__ t = A.tgt + (long long)b * A.tb;
    for (long long x = t0 + threadIdx.x; x < t1; x += 256) {
        const double ed = (double)e[x];
        see = fma(ed, ed, see);
        se += ed;
#pragma unroll
        for (int j = 0; j < kTdMaxK; ++j) {
            if (j < K) {
                const double td = (double)t[j * A.tk + x];
                set[j] = fma(ed, td, set[j]);
                stt[j] = fma(td, td, stt[j]);
                st[j] += td;
            }
        }
    }
    __shared__ double red[4][3 * kTdMaxK + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < kTdMaxK; ++j) {
        const double a = wave_sum(set[j]), bb = wave_sum(stt[j]), cc = wave_sum(st[j]);
        if (lane == 0) {
            red[wave][j] = a;
            red[wave][kTdMaxK + j] = bb;
            red[wave][2 * kTdMaxK + j] = cc;
        }
    }
    {
        const double a = wave_sum(see), bb = wave_sum(se);
        if (lane == 0) {
            red[wave][3 * kTdMaxK] = a;
            red[wave][3 * kTdMaxK + 1] = bb;
        }
    }
    __syncthreads();
    double* w = A.ws + ((long long)b * A.nchunks + c) * NS;
    const int x = threadIdx.x;
    auto tot = [&](int s) { return ((red[0][s] + red[1][s]) + red[2][s]) + red[3][s]; };
    if (x < K) {
        w[i * K + x] = tot(x);
        if (i == 0) {
            w[K * K + K + x] = tot(kTdMaxK + x);
            w[K * K + 3 * K + x] = tot(2 * kTdMaxK + x);
        }
    }
    if (x == 0) {
        w[K * K + i] = tot(3 * kTdMaxK);
        w[K * K + 2 * K + i] = tot(3 * kTdMaxK + 1);
    }
}

// stats[b, s] = sum_c ws[b, c, s] in chunk order (deterministic).
__global__ void td_reduce_kernel(const double* __restrict__ ws, double* __restrict__ stats, int nchunks, int NS) {
    const int b = blockIdx.x;
    for (int s = threadIdx.x; s < NS; s += blockDim.x) {
        double v = 0.0;
        for (int c = 0; c < nchunks; ++c) v += ws[((long long)b * nchunks + c) * NS + s];
        stats[(long long)b * NS + s] = v;
    }
}

struct TdLinArgs {
    const float* x;
    const float* y;
    const int32_t* lengths;
    const float* A;    // [B][K]
    const float* Bc;   // [B][K][K]
    const float* C;    // [B][K]
    float* out;
    long long T;
    long long xb, xk, yb, yk, ob, ok;
    long long chunk;
    int vec;
};

// out[b, i, t] = A[b,i] x[b,i,t] + sum_j Bc[b,i,j] y[b,j,t] + C[b,i]   (t < length_b, else 0)
template <int K>
__global__ __launch_bounds__(256) void td_lincomb_kernel(const TdLinArgs P) {
    const int b = blockIdx.y;
    const long long len = P.lengths ? min((long long)P.lengths[b], P.T) : P.T;
    const long long t0 = (long long)blockIdx.x * P.chunk;
    const long long t1 = min(t0 + P.chunk, P.T);
    float a[K], cc[K], bc[K][K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        a[i] = P.A[(long long)b * K + i];
        cc[i] = P.C[(long long)b * K + i];
#pragma unroll
        for (int j = 0; j < K; ++j) bc[i][j] = P.Bc[((long long)b * K + i) * K + j];
    }
    const float* __restrict__ x = P.x + (long long)b * P.xb;
    const float* __restrict__ y = P.y + (long long)b * P.yb;
    float* __restrict__ o = P.out + (long long)b * P.ob;
    if (P.vec) {
        for (long long t = t0 + 4 * threadIdx.x; t + 4 <= t1; t += 4 * 256) {
            float4 xv[K], yv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                xv[k] = *reinterpret_cast<const float4*>(x + k * P.xk + t);
                yv[k] = *reinterpret_cast<const float4*>(y + k * P.yk + t);
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = fmaf(a[i], reinterpret_cast<const float*>(&xv[i])[q], cc[i]);
#pragma unroll
                    for (int j = 0; j < K; ++j) v = fmaf(bc[i][j], reinterpret_cast<const float*>(&yv[j])[q], v);
                    r[q] = (t + q < len) ? v : 0.f;
                }
                *reinterpret_cast<float4*>(o + i * P.ok + t) = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
        // the (< 4 sample) tail of the row when T % 4 != 0
        const long long tail0 = t1 - ((t1 - t0) & 3);
        if (threadIdx.x < t1 - tail0) {
            const long long t = tail0 + threadIdx.x;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float v = fmaf(a[i], x[i * P.xk + t], cc[i]);
#pragma unroll
                for (int j = 0; j < K; ++j) v = fmaf(bc[i][j], y[j * P.yk + t], v);
                o[i * P.ok + t] = t < len ? v : 0.f;
            }
        }
    } else {
        for (long long t = t0 + threadIdx.x; t < t1; t += 256) {
            float yv[K];
#pragma unroll
            for (int j = 0; j < K; ++j) yv[j] = y[j * P.yk + t];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float v = fmaf(a[i], x[i * P.xk + t], cc[i]);
#pragma unroll
                for (int j = 0; j < K; ++j) v = fmaf(bc[i][j], yv[j], v);
                o[i * P.ok + t] = t < len ? v : 0.f;
            }
        }
    }
}

static long long pick_chunk(long long batch, long long T) {
    // ~2048 workgroups per call when the input allows it; 1024..65536 samples each
    long long chunk = (batch * T + 2047) / 2048;
    chunk = (chunk + 1023) / 1024 * 1024;
    if (chunk < 1024) chunk = 1024;
    if (chunk > 65536) chunk = 65536;
    return chunk;
}

static bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

template <int K>
static int launch_lincomb(const TdLinArgs& P, long long batch, hipStream_t st) {
    const unsigned nchunks = (unsigned)((P.T + P.chunk - 1) / P.chunk);
    hipLaunchKernelGGL(td_lincomb_kernel<K>, dim3(nchunks, (unsigned)batch), dim3(256), 0, st, P);
    return launch_status();
}

}  // namespace ptmi

using namespace ptmi;

extern "C" {

int64_t ptmi_td_stats_elems(int32_t K) { return K < 1 ? PTMI_E_INVALID : (int64_t)K * K + 4 * K; }

int64_t ptmi_td_workspace_elems(int64_t batch, int32_t K, int64_t T) {
    if (batch < 1 || K < 1 || T < 1) return PTMI_E_INVALID;
    const long long chunk = pick_chunk(batch, T);
    return batch * ((T + chunk - 1) / chunk) * ((int64_t)K * K + 4 * K);
}

int ptmi_td_pair_stats(const float* est, const float* tgt, const int32_t* lengths, int64_t batch, int32_t K,
                       int64_t T, const int64_t* strides, double* workspace, double* stats,
                       ptmi_stream_t stream) {
    PTMI_RETURN_IF(!est || !tgt || !strides || !workspace || !stats, PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 1 || K < 1 || T < 1, PTMI_E_INVALID);
    PTMI_RETURN_IF(K > kTdMaxK || batch > 65535, PTMI_E_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    TdArgs A{};
    A.est = est;
    A.tgt = tgt;
    A.lengths = lengths;
    A.T = T;
    A.eb = strides[0];
    A.ek = strides[1];
    A.tb = strides[2];
    A.tk = strides[3];
    A.chunk = pick_chunk(batch, T);
    A.K = K;
    A.nchunks = (int)((T + A.chunk - 1) / A.chunk);
    A.vec = T >= 4 && aligned16(est) && aligned16(tgt) && A.eb % 4 == 0 && A.ek % 4 == 0 && A.tb % 4 == 0 &&
            A.tk % 4 == 0;
    A.ws = workspace;
    const dim3 grid((unsigned)A.nchunks, (unsigned)batch);
    switch (K) {
        case 1: hipLaunchKernelGGL(td_stats_kernel<1>, grid, dim3(256), 0, st, A); break;
        case 2: hipLaunchKernelGGL(td_stats_kernel<2>, grid, dim3(256), 0, st, A); break;
        case 3: hipLaunchKernelGGL(td_stats_kernel<3>, grid, dim3(256), 0, st, A); break;
        case 4: hipLaunchKernelGGL(td_stats_kernel<4>, grid, dim3(256), 0, st, A); break;
        default:
            hipLaunchKernelGGL(td_stats_row_kernel, dim3((unsigned)A.nchunks, (unsigned)batch, (unsigned)K),
                               dim3(256), 0, st, A);
    }
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(td_reduce_kernel, dim3((unsigned)batch), dim3(64), 0, st, workspace, stats, A.nchunks,
                       K * K + 4 * K);
    return launch_status();
}

int ptmi_td_lincomb(const float* x, const float* y, const int32_t* lengths, const float* coef_a,
                    const float* coef_b, const float* coef_c, int64_t batch, int32_t K, int64_t T,
                    const int64_t* strides, float* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!x || !y || !coef_a || !coef_b || !coef_c || !strides || !out, PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 1 || K < 1 || T < 1, PTMI_E_INVALID);
    PTMI_RETURN_IF(K > kTdMaxK || batch > 65535, PTMI_E_UNSUPPORTED);
    TdLinArgs P{};
    P.x = x;
    P.y = y;
    P.lengths = lengths;
    P.A = coef_a;
    P.Bc = coef_b;
    P.C = coef_c;
    P.out = out;
    P.T = T;
    P.xb = strides[0];
    P.xk = strides[1];
    P.yb = strides[2];
    P.yk = strides[3];
    P.ob = strides[4];
    P.ok = strides[5];
    P.chunk = pick_chunk(batch, T);
    P.vec = aligned16(x) && aligned16(y) && aligned16(out) && P.xb % 4 == 0 && P.xk % 4 == 0 && P.yb % 4 == 0 &&
            P.yk % 4 == 0 && P.ob % 4 == 0 && P.ok % 4 == 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (K) {
        case 1: return launch_lincomb<1>(P, batch, st);
        case 2: return launch_lincomb<2>(P, batch, st);
        case 3: return launch_lincomb<3>(P, batch, st);
        case 4: return launch_lincomb<4>(P, batch, st);
        case 5: return launch_lincomb<5>(P, batch, st);
        case 6: return launch_lincomb<6>(P, batch, st);
        case 7: return launch_lincomb<7>(P, batch, st);
        default: return launch_lincomb<8>(P, batch, st);
    }
}

}  // extern "C"
