// Deep-clustering loss (Hershey 2016) for gfx950: single-pass Gram matrix on the matrix cores.
//
// Replaces padertorch/ops/losses/source_separation.py:13-31 (three einsum GEMMs over an (N, E) and an
// (N, K) matrix that the caller first re-lays-out with 't e f -> (t f) e', contrib/tcl/dc.py:73-84):
//     loss = (|X'X|_F^2 - 2 |X'T|_F^2 + |T'T|_F^2) / N^2,   N = T * F rows per example.
// With V = [X | T] (N x D, D = E + K) all three products are blocks of the ONE Gram matrix V'V.
// The kernel streams V once (HBM bound: D * 4 B per row), in whatever layout the caller has
// (strided (t, c, f) addressing: the model's (T, E, F) embedding is consumed in place), stages one
// tile of rows in LDS and accumulates V'V with v_mfma_f32_32x32x2_f32 (exact fp32; both operands of
// the symmetric product are the SAME register).  Per-workgroup partial Grams are reduced in fp64 in a
// fixed order (bitwise reproducible).  Backward: dX = 4/N^2 (X (X'X) - T (T'X)) as a streaming kernel.
#include <cstdlib>

#include "common.h"

namespace ptmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDcMaxD = 32;      // E + K <= 32 (one 32x32 MFMA tile)
constexpr int kDcTile = 256;     // rows (n) per LDS tile
constexpr int kDcPitch = 258;    // LDS pitch (floats): even, conflict-free ds_read_b64 by 32 columns
constexpr int kDcTilesPerWg = 8;

struct DcArgs {
    const float* x;
    const float* t;
    const int32_t* row_frames;
    long long T;                 // time steps (padded length)
    long long xs[4];             // x element (b, t, e, f) at b*xs[0] + t*xs[1] + e*xs[2] + f*xs[3]
    long long ts[4];             // t element (b, t, k, f) likewise
    int E, K, F;                 // F = inner extent per time step (rows n = t*F + f)
    int nchunks;                 // workgroups per example
    int dbg;                     // timing-ablation bits of round 1 (1: no MFMA, 2: no loads, 4: no LDS staging; 8: generic kernels); always 0 now
    unsigned x_span, t_span;     // bytes one example of x / t spans (0: not known to fit the fast kernels)
};

// Stage rows [n0, n0 + kDcTile) of example b (row n = (t, f) = (n / F, n % F)) of V^T into LDS:
// lds[c][i] = V[n0 + i][c]; rows past N_b and columns past D are zero.
__device__ __forceinline__ void dc_stage(float* lds, const DcArgs& A, int b, long long n0, long long N_b,
                                         int tid) {
    const int D = A.E + A.K;
    const float* xb = A.x + b * A.xs[0];
    const float* tb = A.t + b * A.ts[0];
    const bool along_rows = A.xs[3] == 1;     // inner index contiguous: threads run along the rows
    for (int idx = tid; idx < D * kDcTile; idx += 256) {
        int c, i;
        if (along_rows) {
            c = idx / kDcTile;
            i = idx - c * kDcTile;
        } else {                              // column index contiguous (plain (N, E) rows)
            i = idx / D;
            c = idx - i * D;
        }
        const long long n = n0 + i;
        float v = 0.f;
        if (n < N_b) {
            const long long tt = n / A.F;
            const long long f = n - tt * A.F;
            v = c < A.E ? xb[tt * A.xs[1] + c * A.xs[2] + f * A.xs[3]]
                        : tb[tt * A.ts[1] + (c - A.E) * A.ts[2] + f * A.ts[3]];
        }
        lds[c * kDcPitch + i] = v;
    }
    for (int idx = tid; idx < (kDcMaxD - D) * kDcTile; idx += 256) {
        const int c = D + idx / kDcTile, i = idx % kDcTile;
        lds[c * kDcPitch + i] = 0.f;
    }
}

// Inner-contiguous layouts (xs[3] == 1: the model's (t, e, f) embedding): thread i owns row n0 + i of a
// tile for ALL columns, so (t, f) = divmod(n, F) is computed once per row and the D loads of a row are
// issued back to back; consecutive threads read consecutive f (coalesced per column).
struct DcRow {
    long long ox, ot;
    bool valid, in_range;     // row of the example / row of the padded [T, F] extent
};

__device__ __forceinline__ DcRow dc_row(const DcArgs& A, long long n, long long N_b) {
    DcRow r;
    r.valid = n < N_b;
    r.in_range = n < A.T * A.F;
    const unsigned nn = (unsigned)(r.in_range ? n : 0);   // T * F < 2^31 (host checked)
    const unsigned tt = nn / (unsigned)A.F;
    const unsigned f = nn - tt * (unsigned)A.F;
    r.ox = (long long)tt * A.xs[1] + (long long)f * A.xs[3];
    r.ot = (long long)tt * A.ts[1] + (long long)f * A.ts[3];
    return r;
}

__device__ __forceinline__ void dc_load_row(float (&v)[kDcMaxD], const DcArgs& A, const float* xb, const float* tb,
                                            const DcRow& r) {
#pragma unroll
    for (int c = 0; c < kDcMaxD; ++c) {
        float val = 0.f;
        if (c < A.E) {
            if (r.valid) val = xb[r.ox + c * A.xs[2]];
        } else if (c < A.E + A.K) {
            if (r.valid) val = tb[r.ot + (c - A.E) * A.ts[2]];
        }
        v[c] = val;
    }
}

// partial[b, chunk, 32, 32] (fp32) = V'V over the chunk's tiles
__global__ __launch_bounds__(256) void dc_gram_kernel(const DcArgs A, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float lds[kDcMaxD * kDcPitch];
    __shared__ float red[4][32][33];
    const int b = blockIdx.x / A.nchunks;
    const int chunk = blockIdx.x - b * A.nchunks;
    const long long T_b = A.row_frames ? (long long)A.row_frames[b] : A.T;
    const long long N_b = T_b * A.F;
    const long long ntiles = (N_b + kDcTile - 1) / kDcTile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bool along_rows = A.xs[3] == 1 && A.ts[3] == 1;
    const float* xb = A.x + b * A.xs[0];
    const float* tb = A.t + b * A.ts[0];
    float v[kDcMaxD];
    const long long tile0 = (long long)chunk * kDcTilesPerWg;
    if (A.dbg & 2)
        for (int c2 = 0; c2 < kDcMaxD; ++c2) v[c2] = (float)(tid + c2);
    if (along_rows && tile0 < ntiles && !(A.dbg & 2)) dc_load_row(v, A, xb, tb, dc_row(A, tile0 * kDcTile + tid, N_b));
    for (int it = 0; it < kDcTilesPerWg; ++it) {
        const long long tile = tile0 + it;
        if (tile >= ntiles) break;                      // uniform
        __syncthreads();                                // previous tile consumed
        if (along_rows) {
            if (!(A.dbg & 4)) {
#pragma unroll
                for (int c = 0; c < kDcMaxD; ++c) lds[c * kDcPitch + tid] = v[c];
            }
        } else {
            dc_stage(lds, A, b, tile * kDcTile, N_b, tid);
        }
        __syncthreads();
        // the next tile's rows travel from HBM while this tile runs through the matrix cores
        if (along_rows && it + 1 < kDcTilesPerWg && tile + 1 < ntiles && !(A.dbg & 2))
            dc_load_row(v, A, xb, tb, dc_row(A, (tile + 1) * kDcTile + tid, N_b));
        // wave w takes rows [64 w, 64 w + 64) of the tile; lane (c, h) rows 64 w + 32 h + 2 j + {0, 1}
        const float* col = lds + c * kDcPitch + 64 * wave + 32 * h;
        if (!(A.dbg & 1)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 v2 = *reinterpret_cast<const float2*>(col + 2 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v2.x, v2.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v2.y, v2.y, acc, 0, 0, 0);
            }
        }
    }
    // C layout of mfma_f32_32x32x2: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][c] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        const int i = idx >> 5, j = idx & 31;
        partial[((long long)blockIdx.x << 10) + idx] = (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]);
    }
}

// gram[b, 32, 32] (fp64) = sum over the example's chunks in order; ex_loss[b]; loss = batch mean
__global__ void dc_reduce_kernel(const float* __restrict__ partial, double* __restrict__ gram,
                                 float* __restrict__ ex_loss, int nchunks, int E, int K, int F,
                                 const int32_t* row_frames, long long T) {
    const int b = blockIdx.x;
    const long long T_b = row_frames ? (long long)row_frames[b] : T;
    const long long used = ((T_b * F + kDcTile - 1) / kDcTile + kDcTilesPerWg - 1) / kDcTilesPerWg;
    __shared__ double part[256];
    double mine = 0.0;
    for (int idx = threadIdx.x; idx < 1024; idx += blockDim.x) {
        // same order of additions as a plain loop over the chunks (bitwise reproducible), but 8 loads in
        // flight: one chunk at a time is one dependent L2 / HBM round trip per chunk (63 of them: 100 us)
        double s = 0.0;
        const long long n = used < nchunks ? used : nchunks;
        const float* src = partial + (((long long)b * nchunks) << 10) + idx;
        for (long long c = 0; c < n; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[((c + u < n ? c + u : n - 1)) << 10];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c + u < n) s += (double)v[u];
        }
        gram[((long long)b << 10) + idx] = s;
        const int i = idx >> 5, j = idx & 31;
        const int D = E + K;
        if (i < D && j < D) {
            const double w = (i < E) == (j < E) ? 1.0 : -1.0;     // xx and tt blocks +, xt and tx blocks -
            mine += w * s * s;
        }
    }
    part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)blockDim.x; ++i) s += part[i];
        const double N = (double)T_b * F;
        ex_loss[b] = (float)(s / (N * N));
    }
}

__global__ void dc_mean_kernel(const float* __restrict__ ex_loss, float* __restrict__ loss, int batch) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < batch; ++b) s += (double)ex_loss[b];
        loss[0] = (float)(s / batch);
    }
}

struct DcBwdArgs {
    DcArgs a;
    const double* gram;
    const float* gscale;
    float* dx;
    long long batch;
};

// dx[b, t, e, f] = gscale / batch * 4 / N_b^2 * sum_i V[n, i] C[i, e],  C = [X'X ; -T'X]
__global__ __launch_bounds__(256) void dc_backward_kernel(const DcBwdArgs B) {
    const DcArgs& A = B.a;
    __shared__ __attribute__((aligned(16))) float lds[kDcMaxD * kDcPitch];
    __shared__ float Cm[kDcMaxD][kDcMaxD + 1];
    const int b = blockIdx.x / A.nchunks;
    const int chunk = blockIdx.x - b * A.nchunks;
    const long long T_b = A.row_frames ? (long long)A.row_frames[b] : A.T;
    const long long N_b = T_b * A.F;
    const long long ntiles = (N_b + kDcTile - 1) / kDcTile;
    const int tid = threadIdx.x;
    const int D = A.E + A.K;
    const double N = (double)T_b * A.F;
    const float coef = T_b > 0 ? B.gscale[0] * (float)(4.0 / (N * N * (double)B.batch)) : 0.f;
    for (int idx = tid; idx < kDcMaxD * (kDcMaxD + 1); idx += 256) (&Cm[0][0])[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < D * A.E; idx += 256) {
        const int i = idx / A.E, e = idx - i * A.E;
        const double g = B.gram[((long long)b << 10) + i * 32 + e];
        Cm[i][e] = (float)(i < A.E ? g : -g) * coef;
    }
    __syncthreads();                                    // Cm is complete
    if (A.xs[3] == 1 && A.ts[3] == 1) {
        // register path: thread i owns row n0 + i: D loads, E dot products against C (LDS broadcast
        // reads), E stores; consecutive threads touch consecutive f (coalesced per column)
        const float* xb = A.x + b * A.xs[0];
        const float* tb = A.t + b * A.ts[0];
        float* dxb = B.dx + b * A.xs[0];
        const long long tile0 = (long long)chunk * kDcTilesPerWg;
        const long long ntiles_all = (A.T * A.F + kDcTile - 1) / kDcTile;   // incl. the padded frames: zeros
        float v[kDcMaxD];
        DcRow r = dc_row(A, tile0 * kDcTile + tid, N_b);
        if (tile0 < ntiles_all) dc_load_row(v, A, xb, tb, r);
        for (int it = 0; it < kDcTilesPerWg; ++it) {
            const long long tile = tile0 + it;
            if (tile >= ntiles_all) break;
            float cur[kDcMaxD];
#pragma unroll
            for (int q = 0; q < kDcMaxD; ++q) cur[q] = v[q];
            const DcRow rc = r;
            if (it + 1 < kDcTilesPerWg && tile + 1 < ntiles_all) {
                r = dc_row(A, (tile + 1) * kDcTile + tid, N_b);
                dc_load_row(v, A, xb, tb, r);
            }
            if (rc.in_range) {                  // rows past the example's length get zeros (v = 0 there)
                for (int e = 0; e < A.E; ++e) {
                    float sum = 0.f;
#pragma unroll
                    for (int q = 0; q < kDcMaxD; ++q) sum = fmaf(cur[q], Cm[q][e], sum);   // rows q >= D of Cm are zero
                    dxb[rc.ox + e * A.xs[2]] = sum;
                }
            }
        }
        return;
    }
    for (int it = 0; it < kDcTilesPerWg; ++it) {
        const long long tile = (long long)chunk * kDcTilesPerWg + it;
        if (tile >= ntiles) break;
        const long long n0 = tile * kDcTile;
        const int nf = (int)min((long long)kDcTile, N_b - n0);
        __syncthreads();
        dc_stage(lds, A, b, n0, N_b, tid);
        __syncthreads();
        float* dxb = B.dx + b * A.xs[0];
        const bool along_rows = A.xs[3] == 1;
        for (int idx = tid; idx < A.E * nf; idx += 256) {
            int e, i;
            if (along_rows) {
                e = idx / nf;
                i = idx - e * nf;
            } else {
                i = idx / A.E;
                e = idx - i * A.E;
            }
            float s = 0.f;
            for (int q = 0; q < D; ++q) s += lds[q * kDcPitch + i] * Cm[q][e];
            const long long n = n0 + i;
            const long long tt = n / A.F;
            const long long f = n - tt * A.F;
            dxb[tt * A.xs[1] + e * A.xs[2] + f * A.xs[3]] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Fast forms for the model's layout (inner index contiguous, E <= 24, K <= 8, an example within 1 GiB).
// Same arithmetic as dc_gram_kernel / dc_backward_kernel; what differs is how a row gets into registers:
// BUFFER loads without a single branch.  Rows past the example get an out-of-range voffset, columns past
// E (K) an out-of-range soffset, and the hardware returns 0 for both.  With an `if` per column (the
// generic kernels) the compiler's merged wait counts put an s_waitcnt vmcnt(0) behind every load: 32
// serial HBM round trips per tile, 1.0-1.5 TB/s.
constexpr int kDcKX = 8;
constexpr unsigned kDcOobRow = 0x80000000u, kDcOobCol = 0x40000000u;

template <int EX>
struct DcFastRow {
    float x[EX];
    float t[kDcKX];
};

template <int EX>
__device__ __forceinline__ void dc_fast_load(DcFastRow<EX>& v, const DcArgs& A, __amdgpu_buffer_rsrc_t rx,
                                             __amdgpu_buffer_rsrc_t rt, const DcRow& r) {
    const unsigned vx = r.valid ? (unsigned)(r.ox * 4) : kDcOobRow;
    const unsigned vt = r.valid ? (unsigned)(r.ot * 4) : kDcOobRow;
#pragma unroll
    for (int c = 0; c < EX; ++c)
        v.x[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                               rx, vx, c < A.E ? (unsigned)(c * A.xs[2] * 4) : kDcOobCol, 0));
#pragma unroll
    for (int k = 0; k < kDcKX; ++k)
        v.t[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                               rt, vt, k < A.K ? (unsigned)(k * A.ts[2] * 4) : kDcOobCol, 0));
}

template <int EX>
__global__ __launch_bounds__(256) void dc_gram_fast_kernel(const DcArgs A, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float lds[kDcMaxD * kDcPitch];
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(lds);      // reuses the tile after the last MFMA (4 WGs / CU)
    const int b = blockIdx.x / A.nchunks;
    const int chunk = blockIdx.x - b * A.nchunks;
    const long long T_b = A.row_frames ? (long long)A.row_frames[b] : A.T;
    const long long N_b = T_b * A.F;
    const long long ntiles = (N_b + kDcTile - 1) / kDcTile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x + b * A.xs[0]), 0, A.x_span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.t + b * A.ts[0]), 0, A.t_span, 0x00020000);
    for (int idx = tid; idx < kDcMaxD * kDcPitch; idx += 256) lds[idx] = 0.f;      // rows nobody stages stay 0
    DcFastRow<EX> v;
    const long long tile0 = (long long)chunk * kDcTilesPerWg;
    if (tile0 < ntiles) dc_fast_load<EX>(v, A, rx, rt, dc_row(A, tile0 * kDcTile + tid, N_b));
    for (int it = 0; it < kDcTilesPerWg; ++it) {
        const long long tile = tile0 + it;
        if (tile >= ntiles) break;                      // uniform
        __syncthreads();                                // previous tile consumed
#pragma unroll
        for (int q = 0; q < EX; ++q) lds[q * kDcPitch + tid] = v.x[q];              // columns E..EX-1 are zeros
#pragma unroll
        for (int k = 0; k < kDcKX; ++k)
            if (A.E + k < kDcMaxD) lds[(A.E + k) * kDcPitch + tid] = v.t[k];        // after x: row E + k belongs to t
        __syncthreads();
        if (it + 1 < kDcTilesPerWg && tile + 1 < ntiles)
            dc_fast_load<EX>(v, A, rx, rt, dc_row(A, (tile + 1) * kDcTile + tid, N_b));
        const float* col = lds + c * kDcPitch + 64 * wave + 32 * h;
        if (!(A.dbg & 1)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 v2 = *reinterpret_cast<const float2*>(col + 2 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v2.x, v2.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v2.y, v2.y, acc, 0, 0, 0);
            }
        }
    }
    __syncthreads();                                    // every wavefront is done with the last tile
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][c] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        const int i = idx >> 5, j = idx & 31;
        partial[((long long)blockIdx.x << 10) + idx] = (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]);
    }
}

template <int EX>
__global__ __launch_bounds__(256) void dc_backward_fast_kernel(const DcBwdArgs B) {
    const DcArgs& A = B.a;
    __shared__ float Cm[kDcMaxD + kDcKX][kDcMaxD + 1];
    const int b = blockIdx.x / A.nchunks;
    const int chunk = blockIdx.x - b * A.nchunks;
    const long long T_b = A.row_frames ? (long long)A.row_frames[b] : A.T;
    const long long N_b = T_b * A.F;
    const int tid = threadIdx.x;
    const int D = A.E + A.K;
    const double N = (double)T_b * A.F;
    const float coef = T_b > 0 ? B.gscale[0] * (float)(4.0 / (N * N * (double)B.batch)) : 0.f;
    for (int idx = tid; idx < (kDcMaxD + kDcKX) * (kDcMaxD + 1); idx += 256) (&Cm[0][0])[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < D * A.E; idx += 256) {
        const int i = idx / A.E, e = idx - i * A.E;
        const double g = B.gram[((long long)b << 10) + i * 32 + e];
        Cm[i][e] = (float)(i < A.E ? g : -g) * coef;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x + b * A.xs[0]), 0, A.x_span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.t + b * A.ts[0]), 0, A.t_span, 0x00020000);
    float* dxb = B.dx + b * A.xs[0];
    const long long tile0 = (long long)chunk * kDcTilesPerWg;
    const long long ntiles_all = (A.T * A.F + kDcTile - 1) / kDcTile;   // incl. the padded frames: zeros
    DcFastRow<EX> v;
    DcRow r = dc_row(A, tile0 * kDcTile + tid, N_b);
    if (tile0 < ntiles_all) dc_fast_load<EX>(v, A, rx, rt, r);
    for (int it = 0; it < kDcTilesPerWg; ++it) {
        const long long tile = tile0 + it;
        if (tile >= ntiles_all) break;
        const DcFastRow<EX> cur = v;
        const DcRow rc = r;
        if (it + 1 < kDcTilesPerWg && tile + 1 < ntiles_all) {
            r = dc_row(A, (tile + 1) * kDcTile + tid, N_b);
            dc_fast_load<EX>(v, A, rx, rt, r);
        }
        if (rc.in_range) {                  // rows past the example's length get zeros (v = 0 there)
            for (int e = 0; e < A.E; ++e) {
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < EX; ++q) sum = fmaf(cur.x[q], Cm[q][e], sum);          // x[q] = 0 for q >= E
#pragma unroll
                for (int k = 0; k < kDcKX; ++k) sum = fmaf(cur.t[k], Cm[A.E + k][e], sum);  // t[k] = 0, Cm row 0 for k >= K
                dxb[rc.ox + e * A.xs[2]] = sum;
            }
        }
    }
}

static int dc_fill(DcArgs& A, const float* x, const float* t, int64_t T, const int64_t* strides, int32_t E,
                   int32_t K, int32_t F, const int32_t* row_frames) {
    if (!x || !t || !strides || E < 1 || K < 1 || F < 1 || T < 0) return PTMI_E_INVALID;
    if (E + K > kDcMaxD || T * F >= 0x7fffffffLL) return PTMI_E_UNSUPPORTED;
    A.x = x;
    A.t = t;
    A.row_frames = row_frames;
    A.T = T;
    for (int i = 0; i < 4; ++i) {
        A.xs[i] = strides[i];
        A.ts[i] = strides[4 + i];
    }
    A.E = E;
    A.K = K;
    A.F = F;
    A.nchunks = (int)(((T * F + kDcTile - 1) / kDcTile + kDcTilesPerWg - 1) / kDcTilesPerWg);
    if (A.nchunks < 1) A.nchunks = 1;
    A.dbg = 0;
    // fast kernels: inner-contiguous, non-negative strides, one example within 1 GiB, E <= 24, K <= 8
    A.x_span = A.t_span = 0;
    bool fast = T > 0 && A.xs[3] == 1 && A.ts[3] == 1 && E <= 24 && K <= kDcKX && !(A.dbg & 8);
    for (int i = 1; i < 3; ++i) fast = fast && A.xs[i] >= 0 && A.ts[i] >= 0;
    if (fast) {
        const long long xsp = ((T - 1) * A.xs[1] + (E - 1) * A.xs[2] + F) * 4;
        const long long tsp = ((T - 1) * A.ts[1] + (K - 1) * A.ts[2] + F) * 4;
        if (xsp <= 0x40000000LL && tsp <= 0x40000000LL) {
            A.x_span = (unsigned)xsp;
            A.t_span = (unsigned)tsp;
        }
    }
    return PTMI_OK;
}

}  // namespace ptmi

using namespace ptmi;

extern "C" {

int64_t ptmi_dc_workspace_elems(int64_t batch, int64_t T, int32_t F) {
    int64_t nchunks = ((T * F + kDcTile - 1) / kDcTile + kDcTilesPerWg - 1) / kDcTilesPerWg;
    if (nchunks < 1) nchunks = 1;
    return batch * nchunks * 1024;      // float32 elements
}

int ptmi_dc_loss_forward(const float* x, const float* t, int64_t batch, int64_t T, const int64_t* strides,
                         int32_t E, int32_t K, int32_t F, const int32_t* row_frames, float* workspace,
                         double* gram, float* ex_loss, float* loss, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!workspace || !gram || !ex_loss || !loss || batch < 1, PTMI_E_INVALID);
    DcArgs A{};
    int rc = dc_fill(A, x, t, T, strides, E, K, F, row_frames);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long blocks = (long long)batch * A.nchunks;
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    if (A.x_span && E <= 8) hipLaunchKernelGGL(dc_gram_fast_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, A, workspace);
    else if (A.x_span && E <= 16) hipLaunchKernelGGL(dc_gram_fast_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, A, workspace);
    else if (A.x_span) hipLaunchKernelGGL(dc_gram_fast_kernel<24>, dim3((unsigned)blocks), dim3(256), 0, st, A, workspace);
    else hipLaunchKernelGGL(dc_gram_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A, workspace);
    rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(dc_reduce_kernel, dim3((unsigned)batch), dim3(256), 0, st, workspace, gram, ex_loss,
                       A.nchunks, (int)E, (int)K, (int)F, row_frames, (long long)T);
    hipLaunchKernelGGL(dc_mean_kernel, dim3(1), dim3(64), 0, st, ex_loss, loss, (int)batch);
    return launch_status();
}

int ptmi_dc_loss_backward(const float* x, const float* t, const double* gram, const float* gscale,
                          int64_t batch, int64_t T, const int64_t* strides, int32_t E, int32_t K, int32_t F,
                          const int32_t* row_frames, float* dx, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!gram || !gscale || !dx || batch < 1, PTMI_E_INVALID);
    DcBwdArgs B{};
    int rc = dc_fill(B.a, x, t, T, strides, E, K, F, row_frames);
    if (rc) return rc;
    B.gram = gram;
    B.gscale = gscale;
    B.dx = dx;
    B.batch = batch;
    const long long blocks = (long long)batch * B.a.nchunks;
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B.a.x_span && E <= 8) hipLaunchKernelGGL(dc_backward_fast_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, B);
    else if (B.a.x_span && E <= 16) hipLaunchKernelGGL(dc_backward_fast_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, B);
    else if (B.a.x_span) hipLaunchKernelGGL(dc_backward_fast_kernel<24>, dim3((unsigned)blocks), dim3(256), 0, st, B);
    else hipLaunchKernelGGL(dc_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, st, B);
    return launch_status();
}

}  // extern "C"
