// Masked normalisation (padertorch/modules/normalization.py:189-246 forward through
// mask_and_compute_stats :497-512, hand-written backward :322-411; running-statistics path :218-246).
//
// x is a contiguous tensor of rank <= 5 described by its data_format; statistics are taken over a
// subset of the axes ("stat groups" = one (mean, power, n) triple per index of the remaining axes),
// gamma / beta live on the independent axes, and positions t >= sequence_lengths[b] are masked out.
// Three kernels cover forward, backward and the running-statistics mode:
//   norm_reduce_kernel  masked sums per group over the reduced axes (three integrand sets: statistics,
//                       backward statistics, gamma/beta gradients), fp64 accumulation, deterministic
//                       two-stage reduction.  Lanes run along the innermost axis: along the reduction
//                       when that axis is reduced, along 64 neighbouring groups when it is kept, so
//                       global reads are coalesced either way;
//   norm_apply_kernel   y = mask * ((x - mean) * rstd * gamma + beta);
//   norm_bwd_kernel     dx = mask * (ghat * rstd + c1[g] * xc + c0[g]).
// All are one pass over the tensor: HBM bound.
#include "common.h"

namespace ptmi {

constexpr int kNormRank = 5;

struct NormGeo {
    int rank;
    int size[kNormRank];
    long long stride[kNormRank];   // contiguous element strides
    long long sgs[kNormRank];      // stride of the axis in the stat-group index (0: statistics axis)
    long long igs[kNormRank];      // stride of the axis in the gamma / beta index (0: broadcast)
    int bdim, tdim;                // batch / sequence axis or -1
    // the reduction at hand: axes ordered outermost .. innermost
    int nkept, nred;
    int kept[kNormRank], red[kNormRank];
    long long n_groups, n_red;
    long long ogs[kNormRank];      // stride of the axis in the OUTPUT group index of this reduction
    unsigned fdm[kNormRank], fds1[kNormRank], fds2[kNormRank];   // division by size[d] as multiply + shifts
};

// n / size[d] and n % size[d] for 32-bit n without a hardware divide (Granlund-Montgomery round-up
// method: q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(m, n); exact for every n < 2^32).
template <typename IdxT>
__device__ __forceinline__ void divmod(const NormGeo& g, int d, IdxT n, IdxT& q, int& r);
template <>
__device__ __forceinline__ void divmod<unsigned>(const NormGeo& g, int d, unsigned n, unsigned& q, int& r) {
    const unsigned t = __umulhi(g.fdm[d], n);
    q = (t + ((n - t) >> g.fds1[d])) >> g.fds2[d];
    r = (int)(n - q * (unsigned)g.size[d]);
}
template <>
__device__ __forceinline__ void divmod<long long>(const NormGeo& g, int d, long long n, long long& q, int& r) {
    q = n / g.size[d];
    r = (int)(n - q * g.size[d]);
}

struct NormRedArgs {
    const float* x;
    const float* gy;
    const int32_t* lengths;
    const float* mean;    // [stat groups]
    const float* rstd;    // [stat groups]
    const float* gamma;   // [indep groups] or null
    double* ws;           // [nchunks][n_groups][3]
    NormGeo g;
    int mode;             // 0 statistics, 1 backward statistics, 2 gamma / beta gradients
    int shift;
    int gt;               // groups per workgroup: 1 (innermost axis reduced) or 64 (innermost kept)
    long long chunk;      // reduced elements per workgroup
    int nchunks;
};

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// IdxT = unsigned (tensors below 2^31 elements: 32-bit index decode) or long long
template <typename IdxT>
__global__ __launch_bounds__(256) void norm_reduce_kernel(const NormRedArgs A) {
    const NormGeo& g = A.g;
    const int gt = A.gt;
    const int rows = 256 / gt;
    const int col = threadIdx.x % gt, row = threadIdx.x / gt;
    const long long grp = (long long)blockIdx.x * gt + col;
    const bool gvalid = grp < g.n_groups;
    // kept multi-index of this thread's group
    int idx[kNormRank] = {0, 0, 0, 0, 0};
    long long base = 0;
    {
        IdxT rem = (IdxT)(gvalid ? grp : 0);
        for (int i = g.nkept - 1; i >= 0; --i) {
            const int d = g.kept[i];
            IdxT q;
            divmod<IdxT>(g, d, rem, q, idx[d]);
            rem = q;
            base += idx[d] * g.stride[d];
        }
    }
    const long long r0 = (long long)blockIdx.y * A.chunk;
    const long long r1 = min(r0 + A.chunk, g.n_red);
    double s0 = 0., s1 = 0., s2 = 0.;
    for (long long r = r0 + row; r < r1 && gvalid; r += rows) {
        IdxT rem = (IdxT)r;
        long long off = base;
        for (int i = g.nred - 1; i >= 0; --i) {
            const int d = g.red[i];
            IdxT q;
            divmod<IdxT>(g, d, rem, q, idx[d]);
            rem = q;
            off += idx[d] * g.stride[d];
        }
        bool m = true;
        if (A.lengths && g.bdim >= 0 && g.tdim >= 0) m = idx[g.tdim] < A.lengths[idx[g.bdim]];
        if (!m) continue;
        const float xv = A.x[off];
        if (A.mode == 0) {
            s0 += xv;
            s1 += (double)xv * xv;
            s2 += 1.;
        } else {
            long long sg = 0, ig = 0;
            for (int d = 0; d < g.rank; ++d) {
                sg += idx[d] * g.sgs[d];
                ig += idx[d] * g.igs[d];
            }
            const float gyv = A.gy[off];
            const float mu = A.shift ? A.mean[sg] : 0.f;
            if (A.mode == 1) {
                const float ghat = A.gamma ? gyv * A.gamma[ig] : gyv;
                const float xc = xv - mu;
                s0 += ghat;
                s1 += (double)ghat * xc;
                s2 += xc;
            } else {
                const float xhat = (xv - mu) * A.rstd[sg];
                s0 += (double)gyv * xhat;
                s1 += gyv;
            }
        }
    }
    __shared__ double red[3][256];
    double* out = A.ws + ((long long)blockIdx.y * g.n_groups) * 3;
    if (gt == 1) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        s0 = wsum(s0);
        s1 = wsum(s1);
        s2 = wsum(s2);
        if (lane == 0) {
            red[0][wave] = s0;
            red[1][wave] = s1;
            red[2][wave] = s2;
        }
        __syncthreads();
        if (threadIdx.x < 3 && gvalid)
            out[grp * 3 + threadIdx.x] =
                ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
    } else {
        red[0][threadIdx.x] = s0;
        red[1][threadIdx.x] = s1;
        red[2][threadIdx.x] = s2;
        __syncthreads();
        if (row == 0 && gvalid) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double v = 0.;
                for (int rr = 0; rr < rows; ++rr) v += red[k][rr * gt + col];
                out[grp * 3 + k] = v;
            }
        }
    }
}

// out[i] = sum_c ws[c][i], always in the same order (deterministic).  Few chunks: one thread per output
// (coalesced across outputs); many chunks: one wavefront per output, lanes over chunks + shuffle tree
// (a serial walk over hundreds of chunks would be one dependent load latency per chunk).
__global__ void norm_reduce2_kernel(const double* __restrict__ ws, double* __restrict__ out, long long n, int nchunks) {
    if (nchunks <= 8) {
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n) return;
        double v = 0.;
        for (int c = 0; c < nchunks; ++c) v += ws[(long long)c * n + i];
        out[i] = v;
    } else {
        const int lane = threadIdx.x & 63;
        const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (i >= n) return;
        double v = 0.;
        for (int c = lane; c < nchunks; c += 64) v += ws[(long long)c * n + i];
        v = wsum(v);
        if (lane == 0) out[i] = v;
    }
}

struct NormEwArgs {
    const float* x;
    const float* gy;
    const int32_t* lengths;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    const float* c0;   // backward: per stat group additive term
    const float* c1;   // backward: per stat group factor of xc
    float* out;
    NormGeo g;
    long long total;
    int shift, scale, backward;
};

template <typename IdxT>
__global__ __launch_bounds__(256) void norm_elementwise_kernel(const NormEwArgs A) {
    const NormGeo& g = A.g;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < A.total; i += (long long)gridDim.x * 256) {
        IdxT rem = (IdxT)i;
        long long sg = 0, ig = 0;
        int ib = 0, it = 0;
        for (int d = g.rank - 1; d >= 0; --d) {
            IdxT q;
            int id;
            divmod<IdxT>(g, d, rem, q, id);
            rem = q;
            sg += id * g.sgs[d];
            ig += id * g.igs[d];
            if (d == g.bdim) ib = id;
            if (d == g.tdim) it = id;
        }
        bool m = true;
        if (A.lengths && g.bdim >= 0 && g.tdim >= 0) m = it < A.lengths[ib];
        float r = 0.f;
        if (m) {
            const float xv = A.x[i];
            const float mu = A.shift ? A.mean[sg] : 0.f;
            const float rs = A.scale ? A.rstd[sg] : 1.f;
            if (!A.backward) {
                r = (xv - mu) * rs;
                if (A.gamma) r *= A.gamma[ig];
                if (A.beta) r += A.beta[ig];
            } else {
                const float ghat = A.gamma ? A.gy[i] * A.gamma[ig] : A.gy[i];
                r = ghat * rs + A.c1[sg] * (xv - mu) + A.c0[sg];
            }
        }
        A.out[i] = r;
    }
}

static bool geo_from(const ptmi_norm_geom* in, int which, NormGeo* g) {
    if (!in || in->rank < 1 || in->rank > kNormRank) return false;
    g->rank = in->rank;
    long long stride = 1;
    for (int d = in->rank - 1; d >= 0; --d) {
        if (in->size[d] < 1 || in->size[d] > 0x7fffffff) return false;
        g->size[d] = (int)in->size[d];
        g->stride[d] = stride;
        stride *= in->size[d];
        int l = 0;
        while ((1ull << l) < (unsigned long long)in->size[d]) ++l;
        g->fdm[d] = (unsigned)((((1ull << 32) * ((1ull << l) - (unsigned long long)in->size[d])) / (unsigned long long)in->size[d]) + 1);
        g->fds1[d] = l < 1 ? l : 1;
        g->fds2[d] = l < 1 ? 0 : l - 1;
    }
    for (int d = 0; d < kNormRank; ++d) {
        g->sgs[d] = d < in->rank ? in->stat_group_stride[d] : 0;
        g->igs[d] = d < in->rank ? in->indep_stride[d] : 0;
        g->ogs[d] = 0;
    }
    g->bdim = in->batch_dim;
    g->tdim = in->seq_dim;
    // which: 0 = reduce over the statistics axes (groups = stat groups), 1 = reduce over the axes
    // gamma / beta are broadcast along (groups = independent groups)
    g->nkept = g->nred = 0;
    g->n_groups = g->n_red = 1;
    for (int d = 0; d < in->rank; ++d) {
        const long long gs = which == 0 ? in->stat_group_stride[d] : in->indep_stride[d];
        const bool kept = gs != 0 || (in->size[d] == 1);
        if (kept) {
            g->kept[g->nkept++] = d;
            g->n_groups *= g->size[d];
        } else {
            g->red[g->nred++] = d;
            g->n_red *= g->size[d];
        }
    }
    return true;
}

}  // namespace ptmi

using namespace ptmi;

extern "C" {

// Chunks of the reduced range: enough workgroups to fill the chip (the loops are latency bound: one
// load per thread and iteration), bounded by the traffic of the fp64 partials.
static int group_tile(const NormGeo& g) {
    const bool inner_reduced = g.nred > 0 && g.red[g.nred - 1] == g.rank - 1 && g.size[g.rank - 1] > 1;
    return (inner_reduced || g.n_groups == 1) ? 1 : 64;
}

static int plan_chunks(const NormGeo& g, int gt, long long* chunk) {
    const long long tiles = (g.n_groups + gt - 1) / gt;
    const long long rows = 256 / gt;
    long long want = (4096 + tiles - 1) / tiles;
    const long long max_chunks = (g.n_red + rows * 2 - 1) / (rows * 2);     // >= 2 iterations per thread
    // the fp64 partials (24 B per group and chunk) stay below ~1/4 of the bytes the pass reads
    const long long ws_cap = std::max<long long>(1, g.n_red / 24);
    want = std::min(std::min(want, max_chunks), std::min<long long>(ws_cap, 1024));
    if (want < 1) want = 1;
    *chunk = (g.n_red + want - 1) / want;
    return (int)((g.n_red + *chunk - 1) / *chunk);
}

int64_t ptmi_norm_workspace_elems(const ptmi_norm_geom* geom, int32_t which) {
    NormGeo g;
    if (!geo_from(geom, which, &g)) return PTMI_E_INVALID;
    long long chunk;
    return 3 * g.n_groups * plan_chunks(g, group_tile(g), &chunk);
}

int ptmi_norm_reduce(int32_t mode, const float* x, const float* gy, const int32_t* lengths, const float* mean,
                     const float* rstd, const float* gamma, const ptmi_norm_geom* geom, int32_t shift,
                     double* workspace, double* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!x || !geom || !workspace || !out || mode < 0 || mode > 2, PTMI_E_INVALID);
    PTMI_RETURN_IF(mode > 0 && !gy, PTMI_E_INVALID);
    PTMI_RETURN_IF(mode == 2 && !rstd, PTMI_E_INVALID);
    PTMI_RETURN_IF(mode > 0 && shift && !mean, PTMI_E_INVALID);
    NormRedArgs A{};
    PTMI_RETURN_IF(!geo_from(geom, mode == 2 ? 1 : 0, &A.g), PTMI_E_INVALID);
    A.x = x;
    A.gy = gy;
    A.lengths = lengths;
    A.mean = mean;
    A.rstd = rstd;
    A.gamma = gamma;
    A.ws = workspace;
    A.mode = mode;
    A.shift = shift;
    const NormGeo& g = A.g;
    // lanes along the innermost axis: reduced -> one group per workgroup; kept -> 64 groups per workgroup
    A.gt = group_tile(g);
    const long long tiles = (g.n_groups + A.gt - 1) / A.gt;
    A.nchunks = plan_chunks(g, A.gt, &A.chunk);
    PTMI_RETURN_IF(tiles > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g.n_groups * g.n_red < 0x7fffffffLL)
        hipLaunchKernelGGL(norm_reduce_kernel<unsigned>, dim3((unsigned)tiles, (unsigned)A.nchunks), dim3(256), 0, st, A);
    else
        hipLaunchKernelGGL(norm_reduce_kernel<long long>, dim3((unsigned)tiles, (unsigned)A.nchunks), dim3(256), 0, st, A);
    int rc = launch_status();
    if (rc) return rc;
    const long long n = 3 * g.n_groups;
    const long long blocks2 = A.nchunks <= 8 ? (n + 255) / 256 : (n + 3) / 4;
    hipLaunchKernelGGL(norm_reduce2_kernel, dim3((unsigned)blocks2), dim3(256), 0, st, workspace, out, n, A.nchunks);
    return launch_status();
}

int ptmi_norm_elementwise(int32_t backward, const float* x, const float* gy, const int32_t* lengths,
                          const float* mean, const float* rstd, const float* gamma, const float* beta,
                          const float* c0, const float* c1, const ptmi_norm_geom* geom, int32_t shift,
                          int32_t scale, float* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!x || !geom || !out, PTMI_E_INVALID);
    PTMI_RETURN_IF(backward && (!gy || !c0 || !c1), PTMI_E_INVALID);
    PTMI_RETURN_IF((shift && !mean) || (scale && !rstd), PTMI_E_INVALID);
    NormEwArgs A{};
    PTMI_RETURN_IF(!geo_from(geom, 0, &A.g), PTMI_E_INVALID);
    A.x = x;
    A.gy = gy;
    A.lengths = lengths;
    A.mean = mean;
    A.rstd = rstd;
    A.gamma = gamma;
    A.beta = beta;
    A.c0 = c0;
    A.c1 = c1;
    A.out = out;
    A.total = A.g.n_groups * A.g.n_red;
    A.shift = shift;
    A.scale = scale;
    A.backward = backward;
    const long long blocks = std::min<long long>((A.total + 255) / 256, 256LL * 16);
    if (blocks <= 0) return PTMI_OK;
    if (A.total < 0x7fffffffLL)
        hipLaunchKernelGGL(norm_elementwise_kernel<unsigned>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), A);
    else
        hipLaunchKernelGGL(norm_elementwise_kernel<long long>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), A);
    return launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Unit-norm embeddings: y[n, e, f] = x[n, e, f] / max(||x[n, :, f]||_2, eps), the
// torch.nn.functional.normalize(h, dim=-2) of padertorch/contrib/tcl/dc.py:70 (Hershey 2016), forward
// and backward as ONE pass each over the [N, E, F] tensor (F contiguous).  torch runs it as a norm
// reduction, a clamp and a broadcast division (and five element-wise kernels backward), each a full
// HBM round trip over the 660 MB embedding of the BASELINE deep-clustering batch.
// One thread owns one column f (four when F % 4 == 0) of one n: E strided loads into registers, the sum of
// squares, the scaled stores - HBM traffic = one read + one write (backward: two reads + one write).
namespace ptmi {

// W = columns per thread: 4 (b128 accesses) when F % 4 == 0, else 1 (rows of odd length are not 16-byte
// aligned; one column per lane keeps every wavefront access one contiguous 256 B segment).
template <int EMAX, int W>
__global__ __launch_bounds__(256) void unit_norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ inv, long long N, int E, int F, float eps) {
    const int fq = F / W;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * fq) return;
    const long long n = gid / fq;
    const int f0 = (int)(gid - n * fq) * W;
    const float* xp = x + n * (long long)E * F + f0;
    float* yp = y + n * (long long)E * F + f0;
    float v[EMAX][W];
    float ss[W];
#pragma unroll
    for (int q = 0; q < W; ++q) ss[q] = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        // no branch around a load (rows past E re-read row E - 1 and count as 0): with one, the compiler
        // waits for every load before it issues the next
        const long long eo = (long long)(e < E ? e : E - 1) * F;
        if (W == 4) {
            const float4 t = *reinterpret_cast<const float4*>(xp + eo);
            v[e][0] = t.x; v[e][W > 1 ? 1 : 0] = t.y; v[e][W > 2 ? 2 : 0] = t.z; v[e][W > 3 ? 3 : 0] = t.w;
        } else {
            v[e][0] = xp[eo];
        }
    }
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
#pragma unroll
        for (int q = 0; q < W; ++q) ss[q] += e < E ? v[e][q] * v[e][q] : 0.f;
    }
    float r[W];
#pragma unroll
    for (int q = 0; q < W; ++q) {
        const float nq = sqrtf(ss[q]);
        r[q] = 1.f / (nq < eps ? eps : nq);          // clamp_min like torch.nn.functional.normalize: a NaN norm stays NaN (fmaxf would return eps)
        inv[n * F + f0 + q] = r[q];
    }
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        if (e < E) {
            if (W == 4)
                *reinterpret_cast<float4*>(yp + (long long)e * F) =
                    make_float4(v[e][0] * r[0], v[e][W > 1 ? 1 : 0] * r[W > 1 ? 1 : 0], v[e][W > 2 ? 2 : 0] * r[W > 2 ? 2 : 0],
                                v[e][W > 3 ? 3 : 0] * r[W > 3 ? 3 : 0]);
            else
                yp[(long long)e * F] = v[e][0] * r[0];
        }
    }
}

// dx = inv (g - y <g, y>) where the norm was not clamped, inv g where it was (the clamp has no gradient)
template <int EMAX, int W>
__global__ __launch_bounds__(256) void unit_norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                            const float* __restrict__ inv, float* __restrict__ dx,
                                                            long long N, int E, int F, float eps) {
    const int fq = F / W;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * fq) return;
    const long long n = gid / fq;
    const int f0 = (int)(gid - n * fq) * W;
    const long long base = n * (long long)E * F + f0;
    float gv[EMAX][W], yv[EMAX][W];
    float dot[W], r[W];
#pragma unroll
    for (int q = 0; q < W; ++q) {
        dot[q] = 0.f;
        r[q] = inv[n * F + f0 + q];
    }
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const long long eo = base + (long long)(e < E ? e : E - 1) * F;       // branch-free, see the forward kernel
        if (W == 4) {
            const float4 a = *reinterpret_cast<const float4*>(g + eo);
            const float4 c = *reinterpret_cast<const float4*>(y + eo);
            gv[e][0] = a.x; gv[e][W > 1 ? 1 : 0] = a.y; gv[e][W > 2 ? 2 : 0] = a.z; gv[e][W > 3 ? 3 : 0] = a.w;
            yv[e][0] = c.x; yv[e][W > 1 ? 1 : 0] = c.y; yv[e][W > 2 ? 2 : 0] = c.z; yv[e][W > 3 ? 3 : 0] = c.w;
        } else {
            gv[e][0] = g[eo];
            yv[e][0] = y[eo];
        }
    }
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
#pragma unroll
        for (int q = 0; q < W; ++q) dot[q] += e < E ? gv[e][q] * yv[e][q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < W; ++q)
        if (r[q] * eps >= 1.f) dot[q] = 0.f;            // norm <= eps: y = x / eps, no d norm term
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        if (e < E) {
            float o[W];
#pragma unroll
            for (int q = 0; q < W; ++q) o[q] = r[q] * (gv[e][q] - yv[e][q] * dot[q]);
            if (W == 4)
                *reinterpret_cast<float4*>(dx + base + (long long)e * F) =
                    make_float4(o[0], o[W > 1 ? 1 : 0], o[W > 2 ? 2 : 0], o[W > 3 ? 3 : 0]);
            else
                dx[base + (long long)e * F] = o[0];
        }
    }
}

// E > 32 (no register tile of the whole vector): the same arithmetic in the same order with the E strided values read twice - once for
// the sum of squares (the dot product), once for the scaled store; one column per thread.  The second read comes from the caches.
__global__ __launch_bounds__(256) void unit_norm_fwd_wide_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv,
                                                                 long long N, int E, int F, float eps) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * F) return;
    const long long n = gid / F;
    const int f = (int)(gid - n * F);
    const float* xp = x + n * (long long)E * F + f;
    float* yp = y + n * (long long)E * F + f;
    float ss = 0.f;
    for (int e = 0; e < E; ++e) {
        const float v = xp[(long long)e * F];
        ss += v * v;
    }
    const float nq = sqrtf(ss);
    const float r = 1.f / (nq < eps ? eps : nq);
    inv[n * F + f] = r;
    for (int e = 0; e < E; ++e) yp[(long long)e * F] = xp[(long long)e * F] * r;
}

__global__ __launch_bounds__(256) void unit_norm_bwd_wide_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                 const float* __restrict__ inv, float* __restrict__ dx, long long N, int E,
                                                                 int F, float eps) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * F) return;
    const long long n = gid / F;
    const int f = (int)(gid - n * F);
    const long long base = n * (long long)E * F + f;
    const float r = inv[n * F + f];
    float dot = 0.f;
    for (int e = 0; e < E; ++e) dot += g[base + (long long)e * F] * y[base + (long long)e * F];
    if (r * eps >= 1.f) dot = 0.f;
    for (int e = 0; e < E; ++e) dx[base + (long long)e * F] = r * (g[base + (long long)e * F] - y[base + (long long)e * F] * dot);
}

}  // namespace ptmi

extern "C" {

int ptmi_unit_norm_forward(const float* x, float* y, float* inv_norm, int64_t N, int32_t E, int32_t F, float eps,
                           ptmi_stream_t stream) {
    PTMI_RETURN_IF(N < 0 || E < 1 || F < 1 || !(eps > 0.f), PTMI_E_INVALID);
    if (N == 0) return PTMI_OK;
    PTMI_RETURN_IF(!x || !y || !inv_norm, PTMI_E_INVALID);
    if (E > 32) {
        const long long wb = (N * F + 255) / 256;
        PTMI_RETURN_IF(wb > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
        hipLaunchKernelGGL(ptmi::unit_norm_fwd_wide_kernel, dim3((unsigned)wb), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, inv_norm,
                           (long long)N, E, F, eps);
        return ptmi::launch_status();
    }
    const int W = (F & 3) == 0 ? 4 : 1;
    const long long blocks = (N * (F / W) + 255) / 256;
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define PTMI_UN_FWD(EM, WW) hipLaunchKernelGGL((ptmi::unit_norm_fwd_kernel<EM, WW>), grid, block, 0, st, x, y, inv_norm, (long long)N, E, F, eps)
    if (W == 4) {
        if (E <= 8) PTMI_UN_FWD(8, 4); else if (E <= 20) PTMI_UN_FWD(20, 4); else PTMI_UN_FWD(32, 4);
    } else {
        if (E <= 8) PTMI_UN_FWD(8, 1); else if (E <= 20) PTMI_UN_FWD(20, 1); else PTMI_UN_FWD(32, 1);
    }
#undef PTMI_UN_FWD
    return ptmi::launch_status();
}

int ptmi_unit_norm_backward(const float* gy, const float* y, const float* inv_norm, float* dx, int64_t N, int32_t E,
                            int32_t F, float eps, ptmi_stream_t stream) {
    PTMI_RETURN_IF(N < 0 || E < 1 || F < 1 || !(eps > 0.f), PTMI_E_INVALID);
    if (N == 0) return PTMI_OK;
    PTMI_RETURN_IF(!gy || !y || !inv_norm || !dx, PTMI_E_INVALID);
    if (E > 32) {
        const long long wb = (N * F + 255) / 256;
        PTMI_RETURN_IF(wb > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
        hipLaunchKernelGGL(ptmi::unit_norm_bwd_wide_kernel, dim3((unsigned)wb), dim3(256), 0, static_cast<hipStream_t>(stream), gy, y, inv_norm,
                           dx, (long long)N, E, F, eps);
        return ptmi::launch_status();
    }
    const int W = (F & 3) == 0 ? 4 : 1;
    const long long blocks = (N * (F / W) + 255) / 256;
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define PTMI_UN_BWD(EM, WW) hipLaunchKernelGGL((ptmi::unit_norm_bwd_kernel<EM, WW>), grid, block, 0, st, gy, y, inv_norm, dx, (long long)N, E, F, eps)
    if (W == 4) {
        if (E <= 8) PTMI_UN_BWD(8, 4); else if (E <= 20) PTMI_UN_BWD(20, 4); else PTMI_UN_BWD(32, 4);
    } else {
        if (E <= 8) PTMI_UN_BWD(8, 1); else if (E <= 20) PTMI_UN_BWD(20, 1); else PTMI_UN_BWD(32, 1);
    }
#undef PTMI_UN_BWD
    return ptmi::launch_status();
}

}  // extern "C"
