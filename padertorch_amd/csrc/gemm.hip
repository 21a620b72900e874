// Dense fp32 GEMM on the fp16 matrix cores of gfx950 (MI355X): split-operand ("3 x fp16") products.
//
// Replaces the library GEMMs of the PIT / deep-clustering hot path: the LSTM input projections
// X [W_ih_f; W_ih_r]^T + b (torch.nn.LSTM inside padertorch/contrib/examples/source_separation/pit/model.py:60-66,97),
// the dense layers (:98-104), and their input / weight gradients in the backward pass.
//
// fp32 MFMA runs at 1/16 of the fp16 / bf16 rate on this part and there is no TF32.  Every fp32 operand value v
// is therefore used as  v * s = hi + lo  with  hi = fp16(v * s),  lo = fp16(v * s - hi)  (the subtraction is
// exact in fp32; s = 2^(13 - exponent(max |v|)) per operand tensor keeps hi inside fp16's range and lo normal for
// every value within 2^-18 of the largest one; smaller values keep an absolute error of 2^-25 / s), and a product
// a b is accumulated as  a_hi b_hi + a_hi b_lo + a_lo b_hi  in ONE fp32 MFMA accumulator
// (v_mfma_f32_32x32x16_f16; fp16 x fp16 products are exact in fp32).  The dropped term a_lo b_lo and the
// rounding of lo are <= 2^-21 relative: measured against fp64 the result is as close as the exact-fp32 MFMA
// chain (scripts/mb/split_mfma_accuracy.hip: max error / sum|a b| 0.5-2e-7 vs 0.9-4e-7 for v_mfma_f32_16x16x4_f32).
// PRODUCTS = 1 is the plain reduced-precision mode (operands rounded to bf16, one product): BASELINE's "bf16" run.
//
// The operands stay fp32 in HBM: the split happens in registers on the way into LDS (global -> VGPR ->
// v_cvt_pk_f16_f32 / v_sub / v_cvt_pk_f16_f32 -> ds_write_b64), so there are no fp16 copies to keep coherent and both
// storage orders of an operand are handled by the staging pattern (the "rows-contiguous" order is transposed
// 4 x 4 in registers):
//     C[M, N] (+)= alpha * A B  (+ bias[N]),   A given as [M][K] (k contiguous) or [K][M] (m contiguous),
//                                              B given as [N][K] (k contiguous) or [K][N] (n contiguous).
// Workgroup tile 128 x 128 x 32, 4 wavefronts of 64 x 64 (2 x 2 MFMA tiles of 32 x 32), LDS rows of 32 halfs padded
// to 40 (80 B: ds_read_b128 of 16 different rows and the transposed ds_write_b64 are both conflict free), two LDS
// stages and two register stages: tile t+2 is in flight from memory while tile t is multiplied; the split + LDS writes of tile t+1
// follow the MFMAs of tile t.  Split K (weight gradients: few output tiles, long K) writes one slab per K range and sums
// the slabs in a fixed order in a second kernel (bitwise reproducible, no atomics).
#include <stdlib.h>

#include "common.h"

namespace ptmi {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;          // [N] or null
    const unsigned* amax_a;     // float bits of max |A| (device), or null: scale 1
    const unsigned* amax_b;
    int M, N, K;
    long long lda, ldb, ldc;
    int accumulate;             // C += instead of C =
    int ksplit;                 // K range per blockIdx.z (multiple of 32)
    float* workspace;           // split K: [splits][M][N] partial results
    long long a_bytes, b_bytes; // extents of the operand buffers (buffer-load bounds)
    int dbg;                    // PTMI_GEMM_DBG timing ablations (results void): 1 no global loads after the first tile, 2 no LDS
                                // writes after the first tile, 4 no MFMAs, 8 no fragment reads
};

constexpr int BM = 128, BN = 128, BK = 32, PITCH = 40;      // PITCH in halfs
constexpr int PLANE = BM * PITCH;                            // halfs per operand plane

// s = 2^(13 - e) for max|v| = m 2^e, 1 <= m < 2 (zero / denormal maximum: 1)
__device__ __forceinline__ float operand_scale(const unsigned* amax_bits) {
    if (!amax_bits) return 1.f;
    const unsigned e = (*amax_bits >> 23) & 0xffu;
    if (e == 0u || e == 0xffu) return 1.f;
    return __uint_as_float((unsigned)(127 + 13 + 127 - (int)e) << 23);
}

// v (already scaled) -> hi, lo fp16 pairs
template <int PRODUCTS>
__device__ __forceinline__ void split4(const f32x4 v, unsigned (&hi)[2], unsigned (&lo)[2]) {
    if (PRODUCTS == 3) {
        // round to nearest (v_cvt_pk_f16_f32): |lo| <= 2^-11 |hi|, lo's own rounding <= 2^-22 |v|
        const h16x2 h0 = __builtin_convertvector(f32x2{v[0], v[1]}, h16x2);
        const h16x2 h1 = __builtin_convertvector(f32x2{v[2], v[3]}, h16x2);
        const h16x2 l0 = __builtin_convertvector(f32x2{v[0] - (float)h0[0], v[1] - (float)h0[1]}, h16x2);
        const h16x2 l1 = __builtin_convertvector(f32x2{v[2] - (float)h1[0], v[3] - (float)h1[1]}, h16x2);
        hi[0] = __builtin_bit_cast(unsigned, h0);
        hi[1] = __builtin_bit_cast(unsigned, h1);
        lo[0] = __builtin_bit_cast(unsigned, l0);
        lo[1] = __builtin_bit_cast(unsigned, l1);
    } else {
        typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
        const b16x2 h0 = {(__bf16)v[0], (__bf16)v[1]};
        const b16x2 h1 = {(__bf16)v[2], (__bf16)v[3]};
        hi[0] = __builtin_bit_cast(unsigned, h0);
        hi[1] = __builtin_bit_cast(unsigned, h1);
        lo[0] = lo[1] = 0u;
    }
}

// One operand tile (128 rows x 32 k) from global memory into registers, branch free: BUFFER loads whose byte offset is
// replaced by an out-of-range value where the element does not exist (the hardware returns 0) - a branch or a select
// on the loaded value would make the compiler wait for every load separately.
//   KMAJOR: the operand is [rows][K]: thread (r0 = tid >> 3, c4 = tid & 7) takes k = 4 c4 .. 4 c4 + 3 of rows r0 + 32 i
//   else:   the operand is [K][rows]: thread (kq = tid & 7, rq = tid >> 3) takes the 4 x 4 block k = 4 kq + i, rows 4 rq ..
// FAST: every float4 is aligned and either fully inside or fully outside K / rows (checked on the host).
constexpr unsigned kOutOfRange = 0x80000000u;

// K-major staging: row (inside each block of 32 rows) of thread group q = tid >> 3.  The two rows of a 16-lane
// ds_write_b64 group are 4 rows = 80 dwords = 16 banks (mod 32) apart: their 16-dword k spans do not collide.
__device__ __forceinline__ int krow(int tid) {
    const int q = tid >> 3;
    return (q & 1) * 4 + ((q >> 1) & 3) + (q >> 3) * 8;
}

template <bool KMAJOR>
struct TileLoader {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned off[4];          // byte offsets of this thread's four float4 at k0 = 0 (kOutOfRange: row does not exist)
    int kthread;              // k of the thread's first element relative to the tile's k0
    long long kstride;        // bytes per unit of k0

    __device__ __forceinline__ TileLoader(const float* P, long long ld, int row0, int nrows, long long bytes, int tid) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (int)bytes, 0x00020000);
        if (KMAJOR) {
            const int c4 = tid & 7, r0 = krow(tid);
            kthread = 4 * c4;
            kstride = 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + r0 + 32 * i;
                off[i] = row < nrows ? (unsigned)(((long long)row * ld + kthread) * 4) : kOutOfRange;
            }
        } else {
            const int kq = tid & 7, rq = tid >> 3;
            kthread = 4 * kq;
            kstride = ld * 4;
            const int row = row0 + 4 * rq;
#pragma unroll
            for (int i = 0; i < 4; ++i) off[i] = row < nrows ? (unsigned)(((long long)(kthread + i) * ld + row) * 4) : kOutOfRange;
        }
    }

    template <bool FAST>
    __device__ __forceinline__ void load(int k0, int kend, int row0, int nrows, int tid, f32x4 (&v)[4]) const {
        const unsigned soff = (unsigned)(k0 * kstride);
        if (KMAJOR) {
            if (FAST) {
                const bool in = k0 + kthread < kend;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, in ? off[i] : kOutOfRange, soff, 0));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                rsrc, k0 + kthread + q < kend ? off[i] + 4u * q : kOutOfRange, soff, 0));
            }
        } else {
            if (FAST) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsrc, k0 + kthread + i < kend ? off[i] : kOutOfRange, soff, 0));
            } else {
                const int row = row0 + 4 * (tid >> 3);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                rsrc, (k0 + kthread + i < kend && row + q < nrows) ? off[i] + 4u * q : kOutOfRange,
                                                                soff, 0));
            }
        }
    }
};

// registers -> (hi, lo) planes of one LDS stage
template <bool KMAJOR, int PRODUCTS>
__device__ __forceinline__ void store_tile(_Float16* hi_plane, _Float16* lo_plane, int tid, const f32x4 (&v)[4], float scale) {
    if (KMAJOR) {
        const int c4 = tid & 7, r0 = krow(tid);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h[2], l[2];
            split4<PRODUCTS>(v[i] * scale, h, l);
            const int off = (r0 + 32 * i) * PITCH + 4 * c4;
            *reinterpret_cast<uint2*>(hi_plane + off) = make_uint2(h[0], h[1]);
            if (PRODUCTS == 3) *reinterpret_cast<uint2*>(lo_plane + off) = make_uint2(l[0], l[1]);
        }
    } else {
        const int kq = tid & 7, rq = tid >> 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                   // row 4 rq + j: k = 4 kq .. 4 kq + 3 = component j of the four loads
            unsigned h[2], l[2];
            const f32x4 t = {v[0][j], v[1][j], v[2][j], v[3][j]};
            split4<PRODUCTS>(t * scale, h, l);
            const int off = (4 * rq + j) * PITCH + 4 * kq;
            *reinterpret_cast<uint2*>(hi_plane + off) = make_uint2(h[0], h[1]);
            if (PRODUCTS == 3) *reinterpret_cast<uint2*>(lo_plane + off) = make_uint2(l[0], l[1]);
        }
    }
}

template <int PRODUCTS>
__device__ __forceinline__ f32x16 mma(const uint4 a, const uint4 b, const f32x16 c) {
    if (PRODUCTS == 3)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
}

template <bool A_KMAJOR, bool B_KMAJOR, bool FAST, int PRODUCTS>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* const lds = reinterpret_cast<_Float16*>(smem);          // [stage][A hi | A lo | B hi | B lo][128][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup -> output tile, XCD aware (dispatch puts workgroup b on XCD b % 8, each XCD has its own 4 MB L2): the
    // tiles are ordered in bands of 8 tile rows, column by column inside a band, and every XCD takes one contiguous
    // eighth of that order, so the ~64 workgroups an XCD runs at a time form an 8 x 8 block of tiles that share 8 A
    // panels and 8 B panels through that XCD's L2 (plain row-major order: every workgroup of an XCD streams its own
    // panels from the Infinity Cache).  Speed only: any placement computes the same tiles.
    const int ntn = (G.N + BN - 1) / BN, ntm = (G.M + BM - 1) / BM;
    const int total = ntm * ntn, chunk = (total + 7) >> 3;
    const int L = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || L >= total) return;
    const int band = L / (8 * ntn), rem = L - band * 8 * ntn;
    const int band_rows = min(8, ntm - 8 * band);
    const int tn = rem / band_rows, tm = 8 * band + rem - tn * band_rows;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * G.ksplit, kend = min(G.K, kbeg + G.ksplit);
    const float sa = operand_scale(G.amax_a), sb = operand_scale(G.amax_b);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    const int frag = (lane & 31) * PITCH + (lane >> 5) * 8;          // this lane's (row, k group) inside a 32-row tile
    const TileLoader<A_KMAJOR> LA(G.A, G.lda, m0, G.M, G.a_bytes, tid);
    const TileLoader<B_KMAJOR> LB(G.B, G.ldb, n0, G.N, G.b_bytes, tid);

    // Two register stages: tile t+2 is requested from memory at the top of iteration t, tile t+1 (requested one
    // iteration earlier) is split and written to the other LDS stage after the MFMAs of tile t.
    f32x4 va0[4], vb0[4], va1[4], vb1[4];
    auto body = [&](int t, int stage, f32x4 (&la)[4], f32x4 (&lb)[4], f32x4 (&ca)[4], f32x4 (&cb)[4]) {
        const _Float16* cur = lds + stage * 4 * PLANE;
        _Float16* nxt = lds + (stage ^ 1) * 4 * PLANE;
        // no branches in here (one scheduling region): loads past the end of K return zeros, the last iteration's LDS
        // writes go to the stage nobody reads any more
        if (!(G.dbg & 1)) {
            LA.template load<FAST>(kbeg + (t + 2) * BK, kend, m0, G.M, tid, la);
            LB.template load<FAST>(kbeg + (t + 2) * BK, kend, n0, G.N, tid, lb);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 ah[2], al[2], bh[2], bl[2];
            if (G.dbg & 8) {
#pragma unroll
                for (int i = 0; i < 2; ++i) ah[i] = al[i] = bh[i] = bl[i] = make_uint4(t, tid, kk, i);
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (wm * 64 + i * 32) * PITCH + frag + kk * 16;
                ah[i] = *reinterpret_cast<const uint4*>(cur + o);
                if (PRODUCTS == 3) al[i] = *reinterpret_cast<const uint4*>(cur + PLANE + o);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (wn * 64 + j * 32) * PITCH + frag + kk * 16;
                bh[j] = *reinterpret_cast<const uint4*>(cur + 2 * PLANE + o);
                if (PRODUCTS == 3) bl[j] = *reinterpret_cast<const uint4*>(cur + 3 * PLANE + o);
            }
            }
            if (G.dbg & 4) {
                asm volatile("" ::"v"(ah[0].x), "v"(al[1].y), "v"(bh[0].z), "v"(bl[1].w));
                continue;
            }
            // small terms first, the hi x hi term last
            if (PRODUCTS == 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mma<PRODUCTS>(al[i], bh[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mma<PRODUCTS>(ah[i], bl[j], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma<PRODUCTS>(ah[i], bh[j], acc[i][j]);
        }
        if (!(G.dbg & 2)) {
            store_tile<A_KMAJOR, PRODUCTS>(nxt, nxt + PLANE, tid, ca, sa);
            store_tile<B_KMAJOR, PRODUCTS>(nxt + 2 * PLANE, nxt + 3 * PLANE, tid, cb, sb);
        }
        // Issue order for the scheduler: the matrix pipe and the vector ALU are separate, and one 32 x 32 x 16 MFMA keeps
        // its pipe busy for 32 cycles = about 6 VALU issue slots of the same wavefront.  The split of tile t+1 (about 150
        // VALU instructions per wavefront and k-step) and its LDS writes are therefore dealt out between the MFMAs
        // of tile t instead of running behind them (measured: SQ_ACTIVE_INST_VALU 16 % of the wave cycles, issue
        // stalls 45 %, when the two blocks followed each other).
        __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);                 // VMEM reads: next-but-one tile
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x100, PRODUCTS == 3 ? 8 : 4, 0);      // DS reads: this k16 step's fragments
#pragma unroll
            for (int m = 0; m < (PRODUCTS == 3 ? 12 : 4); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, PRODUCTS == 3 ? 6 : 12, 0);   // VALU under it
                if (PRODUCTS != 3 || m % 3 != 2) __builtin_amdgcn_sched_group_barrier(0x200, PRODUCTS == 3 ? 1 : 2, 0);   // DS write
            }
        }
        __syncthreads();
    };

    LA.template load<FAST>(kbeg, kend, m0, G.M, tid, va0);
    LB.template load<FAST>(kbeg, kend, n0, G.N, tid, vb0);
    LA.template load<FAST>(kbeg + BK, kend, m0, G.M, tid, va1);
    LB.template load<FAST>(kbeg + BK, kend, n0, G.N, tid, vb1);
    store_tile<A_KMAJOR, PRODUCTS>(lds, lds + PLANE, tid, va0, sa);
    store_tile<B_KMAJOR, PRODUCTS>(lds + 2 * PLANE, lds + 3 * PLANE, tid, vb0, sb);
    __syncthreads();
    for (int t = 0; t < nk; t += 2) {
        body(t, 0, va0, vb0, va1, vb1);                   // requests tile t+2 into the stage-0 registers, splits tile t+1
        if (t + 1 < nk) body(t + 1, 1, va1, vb1, va0, vb0);
    }

    // C layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const float alpha = 1.f / (sa * sb);
    const bool slab = gridDim.z > 1;            // split K: this range's partial result goes to its own slab
    float* const Cz = slab ? G.workspace + (long long)blockIdx.z * G.M * G.N : G.C;
    const long long ldc = slab ? G.N : G.ldc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= G.N) continue;
        const float bv = (G.bias && !slab) ? G.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (row < G.M) {
                    float* c = Cz + (long long)row * ldc + col;
                    const float v = acc[i][j][q] * alpha + bv;
                    if (G.accumulate && !slab) *c += v;
                    else *c = v;
                }
            }
        }
    }
}

// Wave-specialised form (the default): 8 wavefronts per workgroup, one CONSUMER (fragment reads + MFMAs of tile t) and one
// PRODUCER (global loads of tile t+2, split + LDS writes of tile t+1) on every SIMD.  The matrix pipe and the vector ALU /
// LDS / memory paths are separate units, but two workgroups of the kernel above fall into lock step on a CU (timing
// ablations: skeleton 103 + loads 51 + split and LDS writes 77 + fragment reads 55 + MFMAs 140 ~ the measured 384 us of the
// projection GEMM: nothing overlapped); with fixed roles the two kinds of work always belong to different wavefronts of
// the same SIMD and overlap by construction.  One barrier per k-step hands LDS stage (t+1) & 1 from the producers to the
// consumers and stage t & 1 back.
template <bool A_KMAJOR, bool B_KMAJOR, bool FAST, int PRODUCTS>
__global__ __launch_bounds__(512, 4) void gemm_split_ws_kernel(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* const lds = reinterpret_cast<_Float16*>(smem);          // [stage][A hi | A lo | B hi | B lo][128][PITCH]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int ntn = (G.N + BN - 1) / BN, ntm = (G.M + BM - 1) / BM;
    const int total = ntm * ntn, chunk = (total + 7) >> 3;
    const int L = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || L >= total) return;
    const int band = L / (8 * ntn), rem = L - band * 8 * ntn;
    const int band_rows = min(8, ntm - 8 * band);
    const int tn = rem / band_rows, tm = 8 * band + rem - tn * band_rows;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * G.ksplit, kend = min(G.K, kbeg + G.ksplit);
    const int nk = (kend - kbeg + BK - 1) / BK;

    if (producer) {
        const int ptid = threadIdx.x - 256;
        const float sa = operand_scale(G.amax_a), sb = operand_scale(G.amax_b);
        const TileLoader<A_KMAJOR> LA(G.A, G.lda, m0, G.M, G.a_bytes, ptid);
        const TileLoader<B_KMAJOR> LB(G.B, G.ldb, n0, G.N, G.b_bytes, ptid);
        f32x4 va0[4], vb0[4], va1[4], vb1[4];
        LA.template load<FAST>(kbeg, kend, m0, G.M, ptid, va0);
        LB.template load<FAST>(kbeg, kend, n0, G.N, ptid, vb0);
        LA.template load<FAST>(kbeg + BK, kend, m0, G.M, ptid, va1);
        LB.template load<FAST>(kbeg + BK, kend, n0, G.N, ptid, vb1);
        store_tile<A_KMAJOR, PRODUCTS>(lds, lds + PLANE, ptid, va0, sa);
        store_tile<B_KMAJOR, PRODUCTS>(lds + 2 * PLANE, lds + 3 * PLANE, ptid, vb0, sb);
        __syncthreads();
        // iteration t: request tile t+2 into the registers tile t came through, split tile t+1 into the other LDS stage
        auto body = [&](int t, int stage, f32x4 (&la)[4], f32x4 (&lb)[4], f32x4 (&ca)[4], f32x4 (&cb)[4]) {
            _Float16* nxt = lds + (stage ^ 1) * 4 * PLANE;
            LA.template load<FAST>(kbeg + (t + 2) * BK, kend, m0, G.M, ptid, la);
            LB.template load<FAST>(kbeg + (t + 2) * BK, kend, n0, G.N, ptid, lb);
            store_tile<A_KMAJOR, PRODUCTS>(nxt, nxt + PLANE, ptid, ca, sa);
            store_tile<B_KMAJOR, PRODUCTS>(nxt + 2 * PLANE, nxt + 3 * PLANE, ptid, cb, sb);
            __syncthreads();
        };
        for (int t = 0; t < nk; t += 2) {
            body(t, 0, va0, vb0, va1, vb1);
            if (t + 1 < nk) body(t + 1, 1, va1, vb1, va0, vb0);
        }
        return;
    }

    // consumers: wavefront (wm, wn) owns the 64 x 64 block of the tile
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int frag = (lane & 31) * PITCH + (lane >> 5) * 8;
    __syncthreads();                                   // stage 0 is filled
    __builtin_amdgcn_s_setprio(1);
    for (int t = 0; t < nk; ++t) {
        const _Float16* cur = lds + (t & 1) * 4 * PLANE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (wm * 64 + i * 32) * PITCH + frag + kk * 16;
                ah[i] = *reinterpret_cast<const uint4*>(cur + o);
                if (PRODUCTS == 3) al[i] = *reinterpret_cast<const uint4*>(cur + PLANE + o);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (wn * 64 + j * 32) * PITCH + frag + kk * 16;
                bh[j] = *reinterpret_cast<const uint4*>(cur + 2 * PLANE + o);
                if (PRODUCTS == 3) bl[j] = *reinterpret_cast<const uint4*>(cur + 3 * PLANE + o);
            }
            if (PRODUCTS == 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mma<PRODUCTS>(al[i], bh[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mma<PRODUCTS>(ah[i], bl[j], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma<PRODUCTS>(ah[i], bh[j], acc[i][j]);
        }
        __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);

    const float alpha = 1.f / (operand_scale(G.amax_a) * operand_scale(G.amax_b));
    const bool slab = gridDim.z > 1;
    float* const Cz = slab ? G.workspace + (long long)blockIdx.z * G.M * G.N : G.C;
    const long long ldc = slab ? G.N : G.ldc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= G.N) continue;
        const float bv = (G.bias && !slab) ? G.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (row < G.M) {
                    float* c = Cz + (long long)row * ldc + col;
                    const float v = acc[i][j][q] * alpha + bv;
                    if (G.accumulate && !slab) *c += v;
                    else *c = v;
                }
            }
        }
    }
}

// split K, second pass: C (+)= sum over the slabs in slab order (fixed order: bitwise reproducible) + bias
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ C,
                                                          long long ldc, const float* __restrict__ bias, int M, int N, int accumulate) {
    const long long total = (long long)M * N;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / N), c = (int)(i - (long long)r * N);
        float sum = 0.f;
        for (int z = 0; z < splits; ++z) sum += ws[(long long)z * total + i];
        if (bias) sum += bias[c];
        float* o = C + (long long)r * ldc + c;
        *o = accumulate ? *o + sum : sum;
    }
}

// max |x| as float bits (non-negative floats order like unsigned integers): one atomicMax per workgroup
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long rows, long long cols, long long ld,
                                                     unsigned* __restrict__ out) {
    __shared__ unsigned red[4];
    const long long n = rows * cols;
    unsigned m = 0u;
    if (ld == cols && (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        // four independent 16-byte loads per lane and iteration (few workgroups - see ptmi_absmax - so each has to keep
        // enough bytes in flight itself)
        const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        long long i = blockIdx.x * 256ll + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const f32x4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                m = max(max(m, __float_as_uint(v0[q]) & 0x7fffffffu),
                        max(max(__float_as_uint(v1[q]) & 0x7fffffffu, __float_as_uint(v2[q]) & 0x7fffffffu), __float_as_uint(v3[q]) & 0x7fffffffu));
        }
        for (; i < n4; i += stride) {
            const f32x4 v = x4[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) m = max(m, __float_as_uint(v[q]) & 0x7fffffffu);
        }
    } else {
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const long long r = i / cols, c = i - r * cols;
            m = max(m, __float_as_uint(x[r * ld + c]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m >= 0x7f800000u) m = 0x7f7fffffu;          // inf / nan: the GEMM result will be non-finite anyway
        atomicMax(out, m);
    }
}

template <bool AK, bool BK_, int PRODUCTS>
static int launch_gemm(const GemmArgs& G, bool fast, dim3 grid, hipStream_t st) {
    const size_t lds = (size_t)2 * 4 * PLANE * sizeof(_Float16);      // 81920 B: two workgroups per CU
    static const bool ws = !(getenv("PTMI_GEMM_WS") && atoi(getenv("PTMI_GEMM_WS")) == 0);
    if (ws && fast) {
        auto k = gemm_split_ws_kernel<AK, BK_, true, PRODUCTS>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, grid, dim3(512), lds, st, G);
        return launch_status();
    }
    if (ws) {
        auto k = gemm_split_ws_kernel<AK, BK_, false, PRODUCTS>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, grid, dim3(512), lds, st, G);
        return launch_status();
    }
    if (fast) {
        auto k = gemm_split_kernel<AK, BK_, true, PRODUCTS>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, G);
    } else {
        auto k = gemm_split_kernel<AK, BK_, false, PRODUCTS>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, G);
    }
    return launch_status();
}

}  // namespace ptmi

using namespace ptmi;

extern "C" {

int ptmi_absmax(const float* x, int64_t rows, int64_t cols, int64_t ld, uint32_t* out_bits, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!x || !out_bits || rows < 0 || cols < 0 || ld < cols, PTMI_E_INVALID);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = zero_words_async(out_bits, 1, st);
    if (e != hipSuccess) return (int)e;
    if (rows * cols == 0) return PTMI_OK;
    // every workgroup ends with one atomicMax on the same word: ~12 ns each at the L2 (2048 workgroups spent 25 us there,
    // whatever the size of the matrix), so at most 512 of them
    const long long work = (rows * cols + 4095) / 4096;
    const unsigned grid = (unsigned)std::min<long long>(std::max<long long>(work, 1), 512);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, st, x, (long long)rows, (long long)cols, (long long)ld, out_bits);
    return launch_status();
}

int ptmi_gemm_split(const float* a, int32_t a_kmajor, int64_t lda, const uint32_t* amax_a, const float* b, int32_t b_kmajor,
                    int64_t ldb, const uint32_t* amax_b, const float* bias, float* c, int64_t ldc, int32_t m, int32_t n,
                    int32_t k, int32_t accumulate, int32_t products, int32_t split_k, float* workspace, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!a || !b || !c || m < 0 || n < 0 || k < 0, PTMI_E_INVALID);
    PTMI_RETURN_IF(products != 1 && products != 3, PTMI_E_UNSUPPORTED);
    PTMI_RETURN_IF(lda < (a_kmajor ? k : m) || ldb < (b_kmajor ? k : n) || ldc < n, PTMI_E_INVALID);
    if (m == 0 || n == 0) return PTMI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (k == 0) {           // empty sum: C = bias (or unchanged when accumulating)
        PTMI_RETURN_IF(!accumulate, PTMI_E_UNSUPPORTED);
        return PTMI_OK;
    }
    // float4 granularity: base 16-byte aligned, leading dimension a multiple of 4, and the contiguous extent a multiple of 4
    // (K of a k-major operand: surplus elements would enter the sums) - or, for a rows-contiguous operand, a leading
    // dimension that covers the extent rounded up to 4 (the surplus elements exist in memory and only feed output
    // rows / columns past M / N, which are never stored)
    auto aligned = [](const void* p, int64_t ld, int64_t inner, bool kmajor) {
        return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && ((inner & 3) == 0 || (!kmajor && ld >= ((inner + 3) & ~3LL)));
    };
    const bool fast = aligned(a, lda, a_kmajor ? k : m, a_kmajor) && aligned(b, ldb, b_kmajor ? k : n, b_kmajor);
    int splits = std::max(1, std::min<int>(split_k, (k + BK - 1) / BK));
    int ksplit = ((k + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (k + ksplit - 1) / ksplit;
    PTMI_RETURN_IF(splits > 1 && !workspace, PTMI_E_INVALID);
    // bytes from the operand's base to the end of its last row (32-bit buffer offsets; bit 31 marks "no such element")
    const long long a_bytes = ((long long)((a_kmajor ? m : k) - 1) * lda + (a_kmajor ? k : m)) * 4;
    const long long b_bytes = ((long long)((b_kmajor ? n : k) - 1) * ldb + (b_kmajor ? k : n)) * 4;
    PTMI_RETURN_IF(a_bytes >= 0x7fffffffLL || b_bytes >= 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    GemmArgs G{a, b, c, bias, amax_a, amax_b, m, n, k, lda, ldb, ldc, accumulate ? 1 : 0, ksplit, workspace, a_bytes, b_bytes,
               getenv("PTMI_GEMM_DBG") ? atoi(getenv("PTMI_GEMM_DBG")) : 0};
    const int tiles = ((m + BM - 1) / BM) * ((n + BN - 1) / BN);
    const dim3 grid((unsigned)((tiles + 7) / 8 * 8), 1u, (unsigned)splits);
    const int sel = (a_kmajor ? 2 : 0) | (b_kmajor ? 1 : 0);
    int rc;
    if (products == 3) {
        switch (sel) {
            case 3: rc = launch_gemm<true, true, 3>(G, fast, grid, st); break;
            case 2: rc = launch_gemm<true, false, 3>(G, fast, grid, st); break;
            case 1: rc = launch_gemm<false, true, 3>(G, fast, grid, st); break;
            default: rc = launch_gemm<false, false, 3>(G, fast, grid, st); break;
        }
    } else {
        switch (sel) {
            case 3: rc = launch_gemm<true, true, 1>(G, fast, grid, st); break;
            case 2: rc = launch_gemm<true, false, 1>(G, fast, grid, st); break;
            case 1: rc = launch_gemm<false, true, 1>(G, fast, grid, st); break;
            default: rc = launch_gemm<false, false, 1>(G, fast, grid, st); break;
        }
    }
    if (rc != PTMI_OK || splits == 1) return rc;
    const long long total = (long long)m * n;
    const unsigned rgrid = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3(rgrid), dim3(256), 0, st, workspace, splits, c, (long long)ldc, bias, m, n,
                       accumulate ? 1 : 0);
    return launch_status();
}

int64_t ptmi_gemm_workspace_elems(int32_t m, int32_t n, int32_t k, int32_t split_k) {
    const int splits = std::max(1, std::min<int>(split_k, (k + BK - 1) / BK));
    return splits > 1 ? (int64_t)splits * m * n : 0;
}

}  // extern "C"
