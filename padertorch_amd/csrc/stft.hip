// STFT / iSTFT / fused PIT feature front-end for gfx950 (MI355X).
//
// Replaces padertorch/ops/_stft.py:103-174 (STFT.__call__), :176-263 (STFT.inverse) and
// padertorch/contrib/examples/source_separation/pit/data.py:49-77 (pre_batch_transform).
//
// Design:
//  * one 256-thread workgroup owns FPB consecutive frames of one batch row; the samples those
//    frames cover ((FPB-1)*shift + L floats) are staged ONCE into LDS with coalesced dword loads
//    (fading / frame padding = predicated zero fill, bit-exact framing, no padded copy in HBM);
//  * a real FFT of `size` points is a complex FFT of M = size/2 = R1*R2 points on packed
//    even/odd samples, done as two in-register FFTs (R1 then R2 points per lane, LPF = max(R1,R2)
//    lanes per frame, 64/LPF frames per wavefront) with ONE transposition through LDS
//    (pitch R2+1 -> conflict free ds_write_b64/ds_read_b64), window and inter-stage twiddles
//    held in VGPRs;
//  * the hermitian split (and |.| / cos-phase epilogues) reads the natural-order spectrum from
//    LDS and each wavefront streams its frames' rows out as one contiguous run of HBM stores.
//  * the inverse runs the same machinery backwards and overlap-adds from LDS; halo frames are
//    recomputed (ceil(L/shift)-1 per workgroup) so the result is deterministic (no atomics).
// The kernels are HBM-bound (2568 algorithmic bytes per frame at size 512 / shift 128).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "fft_regs.h"

namespace ptmi {

// ------------------------------------------------------------------------------------------------
template <int R1_, int R2_>
struct Plan {
    static constexpr int R1 = R1_, R2 = R2_;
    static constexpr int M = R1 * R2;      // complex FFT length
    static constexpr int F = M + 1;        // one-sided bins
    static constexpr int SIZE = 2 * M;     // real FFT length
    static constexpr int LPF = R1 > R2 ? R1 : R2;  // lanes per frame
    static constexpr int FPW = 64 / LPF;           // frames per wavefront
    static constexpr int FPB = 4 * FPW;            // frames per 256-thread workgroup
    static constexpr int P = R2 + 1;               // transposition pitch (complex)
    static constexpr int FS0 = (R1 * P > F ? R1 * P : F);
    static constexpr int FS = (FS0 + 1) & ~1;      // per-frame LDS region (complex), even
    static constexpr int NIT = (FPW * F + 63) / 64;  // epilogue items per lane
    // inverse kernel: waves per SIMD the register budget is sized for (its LDS footprint grows with M)
    static constexpr int INV_WAVES = M <= 256 ? 3 : (M <= 512 ? 2 : 1);
};

struct FwdArgs {
    const float* x;
    const float* s;  // features kernel: sources [batch, K, row_stride]
    const int32_t* row_samples;
    const float* window;
    const cpx* twiddle;
    float* out;      // stft: spectrum; features: Y_abs
    float* X_abs;
    float* cos_pd;
    long long x_row_stride;
    long long num_samples;
    long long out_frames;
    long long batch;
    int nchunks;
    int layout;
    int K;
    int dbg;         // PTMI_STFT_DBG ablation bits (1: no stores, 2: no FFT, 4: no loads)
    bool aligned2;   // every frame start is 8-byte aligned -> float2 sample loads
    float edge_scale;
    Geo g;
    // fused (log-)mel epilogue (stft_fwd_kernel<PL, true>): band-compressed filterbank in 16-byte
    // aligned groups of 8 bins
    const int32_t* mel_lo;    // [M] first bin of filter m's first group / 4
    const int32_t* mel_cnt;   // [M] number of 8-bin groups
    const int32_t* mel_off;   // [M] offset of its weights in mel_w / 4
    const float* mel_w;       // [nnz] weights, zero-padded to whole groups
    int mel_M, mel_nnz, mel_power, mel_log;
    float mel_eps;
    // features kernel: the model's first-layer input written on the way out (ptmi_pit_features_packed): log1p(|Y|) as
    // PackedSequence rows (time-major: row = lp_offs[t] + b, or t * batch + b when lp_offs is null) in fp32 and as fp16 (hi, lo)
    // planes of 2^9 log1p(|Y|) in MFMA-fragment order (csrc/gemm_planes.hip: operand A of the first input projection)
    float* lp_out;
    _Float16* lp_planes;
    const long long* lp_offs;
    int lp_kb;
};

// W_size^j for 0 <= j < size from the half-circle table (j = 0..M): W^(j) = -W^(j-M) for j > M.
template <int M>
__device__ __forceinline__ cpx tw_full(const cpx* tw, int j) {
    if (j > M) {
        const cpx v = tw[j - M];
        return cpx{-v.x, -v.y};
    }
    return tw[j];
}

// LDS carve-up shared by the forward kernels (all offsets multiples of 16 bytes):
//   buf  [FPB][FS] cpx   per-frame transposition / natural-order spectrum
//   tws  [F+1]     cpx   split twiddles W_size^k, k = 0..M
//   tw1t [R1][LPF] cpx   inter-stage twiddles W_M^(l*k1), lane-contiguous
//   win  [SIZE]    float analysis window, zero beyond window_length
//   sig  [...]     float staged samples of the FPB frames, zero-filled tail up to SIZE
template <class PL>
struct FwdLds {
    cpx* buf;
    cpx* tws;
    cpx* tw1t;
    float* win;
    float* sig;
    __device__ __forceinline__ explicit FwdLds(char* smem) {
        buf = reinterpret_cast<cpx*>(smem);
        tws = buf + PL::FPB * PL::FS;
        tw1t = tws + PL::F + 1;
        win = reinterpret_cast<float*>(tw1t + PL::R1 * PL::LPF);
        sig = win + PL::SIZE;
    }
    static size_t bytes(int shift) {
        const size_t chunk = (size_t)(PL::FPB - 1) * shift + PL::SIZE;
        return sizeof(cpx) * ((size_t)PL::FPB * PL::FS + PL::F + 1 + PL::R1 * PL::LPF) +
               sizeof(float) * (PL::SIZE + chunk + 2 * PL::LPF + 8);
    }
};

// Fill the constant tables (once per workgroup, coalesced; the tables are L2 resident).
template <class PL>
__device__ __forceinline__ void load_tables(const FwdLds<PL>& S, const float* __restrict__ window,
                                            const cpx* __restrict__ twiddle, int L, int tid) {
    for (int i = tid; i <= PL::M; i += 256) S.tws[i] = twiddle[i];
    for (int i = tid; i < PL::SIZE; i += 256) S.win[i] = (i < L) ? window[i] : 0.f;
    for (int i = tid; i < PL::R1 * PL::LPF; i += 256) {
        const int k1 = i / PL::LPF, l = i - k1 * PL::LPF;
        S.tw1t[i] = tw_full<PL::M>(twiddle, (2 * l * k1) % PL::SIZE);   // W_M^(l*k1) = W_size^(2*l*k1)
    }
}

// Two-step complex FFT of one frame.  On entry lane l (< R2) holds a[i1] = in[R2*i1 + l].
// On exit fbuf[k], k < M, holds the transform in natural order.  Contains block-wide barriers.
template <class PL, bool INV>
__device__ __forceinline__ void fft_to_lds(cpx (&a)[PL::R1], cpx* fbuf, int l, const cpx* tw1t) {
    constexpr int R1 = PL::R1, R2 = PL::R2, P = PL::P;
    if (l < R2) {
        fft_dif<R1, INV>(a);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            const cpx v = a[bitrev<R1>(k1)];
            const cpx w = tw1t[k1 * PL::LPF + l];
            fbuf[k1 * P + l] = INV ? cmulc(v, w) : cmul(v, w);
        }
    }
    __syncthreads();
    cpx c[R2];
    if (l < R1) {
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) c[n2] = fbuf[l * P + n2];
    }
    __syncthreads();
    if (l < R1) {
        fft_dif<R2, INV>(c);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) fbuf[l + R1 * k2] = c[bitrev<R2>(k2)];
    }
    __syncthreads();
}

// Stage the samples of frames [t0, t0+FPB) of one row into LDS: zero outside [0, n_b) (fading and
// frame padding) and zero up to SIZE past the last frame start (so the unpredicated window loads
// below never see uninitialised LDS).  Row offsets fit in 32 bits (checked on the host).
template <class PL>
__device__ __forceinline__ void stage_signal(float* sig, const float* __restrict__ xrow, int n_b,
                                             int t0, const Geo& g, int tid) {
    const int chunk = (PL::FPB - 1) * g.shift + PL::SIZE;
    const int v0 = t0 * g.shift - g.pad_left;
    for (int i = tid; i < chunk; i += 256) {
        const int xi = v0 + i;
        float v = 0.f;
        if (xi >= 0 && xi < n_b) v = xrow[xi];
        sig[i] = v;
    }
}

// Load the windowed, even/odd packed samples of this lane's frame into registers
// (a[n1] = x_w[2 (R2 n1 + l)] + i x_w[2 (R2 n1 + l) + 1]).  No predicates: win is 0 beyond L.
template <class PL>
__device__ __forceinline__ void load_windowed(cpx (&a)[PL::R1], const float* sf, const float* win, int l) {
#pragma unroll
    for (int n1 = 0; n1 < PL::R1; ++n1) {
        const int k = 2 * (PL::R2 * n1 + l);
        a[n1] = cpx{sf[k] * win[k], sf[k + 1] * win[k + 1]};
    }
}

// Hermitian split: bin k of the size-point real FFT from the natural-order M-point complex FFT.
template <class PL>
__device__ __forceinline__ cpx split_bin(const cpx* zb, const cpx* tws, int k) {
    constexpr int M = PL::M;
    const cpx z1 = zb[k & (M - 1)];
    const cpx z2 = zb[(M - k) & (M - 1)];
    const cpx w = tws[k];
    const float ex = 0.5f * (z1.x + z2.x), ey = 0.5f * (z1.y - z2.y);  // E = (Z[k] + conj Z[M-k]) / 2
    const float dx = 0.5f * (z1.x - z2.x), dy = 0.5f * (z1.y + z2.y);  // D = (Z[k] - conj Z[M-k]) / 2
    // X = E + (-i D) * w,  -i D = (dy, -dx)
    return cpx{ex + dy * w.x + dx * w.y, ey + dy * w.y - dx * w.x};
}

// ------------------------------------------------------------------------------------------------
// Forward kernels, wave-private pipelines: every wavefront owns FPW frames of ONE signal from the
// global loads to the stores (no workgroup barrier on the way), so the wavefronts of a CU drift
// apart and overlap each other's load / FFT / store phases.
__device__ __forceinline__ void wave_sync() {
    // LDS hand-off between lanes of ONE wavefront: DS operations of a wavefront execute in order,
    // so only the compiler must be kept from reordering across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tables shared by the workgroup + per-wavefront spectrum buffers (+ features: phasors of Y)
template <class PL>
struct WaveLds {
    cpx* tws;    // [F+1]      split twiddles W_size^k
    cpx* tw1t;   // [R1][LPF]  inter-stage twiddles
    float* win;  // [SIZE]     window, zero beyond window_length
    cpx* buf;    // [nwaves][FPW][FS]
    cpx* yph;    // [nyph][FPW][YS]  unit phasors of the mixture spectrum (features: one per wavefront)
    static constexpr int YS = (PL::F + 1) & ~1;
    __device__ __forceinline__ WaveLds(char* smem, int nwaves) {
        tws = reinterpret_cast<cpx*>(smem);
        tw1t = tws + PL::F + 1;
        win = reinterpret_cast<float*>(tw1t + PL::R1 * PL::LPF);
        buf = reinterpret_cast<cpx*>(win + PL::SIZE);
        yph = buf + nwaves * PL::FPW * PL::FS;
    }
    __host__ __device__ static size_t bytes(int nwaves, int nyph) {
        return sizeof(cpx) * (PL::F + 1 + PL::R1 * PL::LPF + (size_t)nwaves * PL::FPW * PL::FS +
                              (size_t)nyph * PL::FPW * YS) + sizeof(float) * PL::SIZE;
    }
};

template <class PL>
__device__ __forceinline__ void load_tables_w(const WaveLds<PL>& S, const float* __restrict__ window,
                                              const cpx* __restrict__ twiddle, int L, int tid, int nthr) {
    for (int i = tid; i <= PL::M; i += nthr) S.tws[i] = twiddle[i];
    for (int i = tid; i < PL::SIZE; i += nthr) S.win[i] = (i < L) ? window[i] : 0.f;
    for (int i = tid; i < PL::R1 * PL::LPF; i += nthr) {
        const int k1 = i / PL::LPF, l = i - k1 * PL::LPF;
        S.tw1t[i] = tw_full<PL::M>(twiddle, (2 * l * k1) % PL::SIZE);
    }
}

// Windowed even/odd packed samples of one frame straight from global memory (L1/L2 absorb the
// size/shift-fold overlap between the frames of a wavefront).  Fading / frame padding is a SELECT
// on unconditionally issued loads from clamped addresses: a branch around each load would make the
// compiler wait for every load separately (16 dependent round trips instead of one).
template <class PL>
__device__ __forceinline__ void load_windowed_global(cpx (&a)[PL::R1], const float* __restrict__ xrow, int n_b,
                                                     int x0, int L, bool aligned, const float* win, int l) {
    const int last = n_b > 0 ? n_b - 1 : 0;
    if (aligned && n_b >= 2) {
        // x0 and k are even, so (xi, xi+1) is an 8-byte aligned pair; only the pair that straddles
        // the end of an odd-length row needs the separately loaded last sample.
        const int cmax = (n_b - 2) & ~1;
        const float xl = xrow[last];
        float2 v[PL::R1];
#pragma unroll
        for (int n1 = 0; n1 < PL::R1; ++n1) {
            const int xi = x0 + 2 * (PL::R2 * n1 + l);
            const int c = min(max(xi, 0), cmax);
            v[n1] = *reinterpret_cast<const float2*>(xrow + c);
        }
#pragma unroll
        for (int n1 = 0; n1 < PL::R1; ++n1) {
            const int k = 2 * (PL::R2 * n1 + l);
            const int xi = x0 + k;
            float e = (xi >= 0 && xi < n_b) ? v[n1].x : 0.f;
            const float o = (xi + 1 >= 0 && xi + 1 < n_b) ? v[n1].y : 0.f;
            if (xi == last && xi > cmax) e = xl;
            a[n1] = cpx{e * win[k], o * win[k + 1]};
        }
    } else {
        float ve[PL::R1], vo[PL::R1];
#pragma unroll
        for (int n1 = 0; n1 < PL::R1; ++n1) {
            const int xi = x0 + 2 * (PL::R2 * n1 + l);
            ve[n1] = xrow[min(max(xi, 0), last)];
            vo[n1] = xrow[min(max(xi + 1, 0), last)];
        }
#pragma unroll
        for (int n1 = 0; n1 < PL::R1; ++n1) {
            const int k = 2 * (PL::R2 * n1 + l);
            const int xi = x0 + k;
            const float e = (n_b > 0 && xi >= 0 && xi < n_b) ? ve[n1] : 0.f;
            const float o = (n_b > 0 && xi + 1 >= 0 && xi + 1 < n_b) ? vo[n1] : 0.f;
            a[n1] = cpx{e * win[k], o * win[k + 1]};
        }
    }
}

// Interior frames (every sample of the wavefront's frames lies inside the row, 8-byte aligned):
// immediate-offset float2 loads and float2 window reads, no clamps / selects.
template <class PL>
__device__ __forceinline__ void load_windowed_interior(cpx (&a)[PL::R1], const float* __restrict__ xf,
                                                       const float* win, int l) {
    const float2* __restrict__ px = reinterpret_cast<const float2*>(xf) + l;
    const float2* pw = reinterpret_cast<const float2*>(win) + l;
    float2 v[PL::R1];
#pragma unroll
    for (int n1 = 0; n1 < PL::R1; ++n1) v[n1] = px[PL::R2 * n1];
#pragma unroll
    for (int n1 = 0; n1 < PL::R1; ++n1) {
        const float2 w = pw[PL::R2 * n1];
        a[n1] = cpx{v[n1].x * w.x, v[n1].y * w.y};
    }
}

template <class PL>
__device__ __forceinline__ void load_frames(cpx (&a)[PL::R1], const float* __restrict__ xrow, int n_b, int tw0,
                                            int fl, int l, const Geo& g, bool aligned, const float* win) {
    const int x_first = tw0 * g.shift - g.pad_left;
    const int x_last = (tw0 + PL::FPW - 1) * g.shift - g.pad_left + PL::SIZE;
    if (aligned && x_first >= 0 && x_last <= n_b) {          // wave-uniform
        if (l < PL::R2) load_windowed_interior<PL>(a, xrow + x_first + fl * g.shift, win, l);
    } else {
        load_windowed_global<PL>(a, xrow, n_b, x_first + fl * g.shift, g.L, aligned, win, l);
    }
}

// Two-step complex FFT of the FPW frames of ONE wavefront (wave-level hand-offs only).
template <class PL, bool INV>
__device__ __forceinline__ void fft_to_lds_wave(cpx (&a)[PL::R1], cpx* fbuf, int l, const cpx* tw1t) {
    constexpr int R1 = PL::R1, R2 = PL::R2, P = PL::P;
    if (l < R2) {
        fft_dif<R1, INV>(a);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            const cpx v = a[bitrev<R1>(k1)];
            const cpx w = tw1t[k1 * PL::LPF + l];
            fbuf[k1 * P + l] = INV ? cmulc(v, w) : cmul(v, w);
        }
    }
    wave_sync();
    cpx c[R2];
    if (l < R1) {
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) c[n2] = fbuf[l * P + n2];
    }
    wave_sync();
    if (l < R1) {
        fft_dif<R2, INV>(c);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) fbuf[l + R1 * k2] = c[bitrev<R2>(k2)];
    }
    wave_sync();
}

// Hermitian split of the bin pair (k, M-k) from Z[k], Z[M-k]:  with E = (Z[k] + conj Z[M-k])/2 and
// Q = W^k (Z[k] - conj Z[M-k]) / (2i):  X[k] = E + Q,  X[M-k] = conj(E - Q).
__device__ __forceinline__ void split_vals(cpx z1, cpx z2, cpx w, cpx& Xk, cpx& Xm);

template <class PL>
__device__ __forceinline__ void split_pair(const cpx* zb, const cpx* tws, int k, cpx& Xk, cpx& Xm) {
    constexpr int M = PL::M;
    split_vals(zb[k & (M - 1)], zb[(M - k) & (M - 1)], tws[k], Xk, Xm);
}

__device__ __forceinline__ void split_vals(cpx z1, cpx z2, cpx w, cpx& Xk, cpx& Xm) {
    const cpx e2 = add_conj(z1, z2);                    // 2 E
    const cpx d = sub_conj(z1, z2) * cpx{0.5f, 0.5f};   // D
    cpx t, q;                                           // Q = (-i D) * w
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(t) : "v"(d), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(q) : "v"(d), "v"(w), "v"(t));
    const cpx h1 = cpx{0.5f, 0.5f}, h2 = cpx{0.5f, -0.5f};
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(Xk) : "v"(e2), "s"(h1), "v"(q));                   // E + Q
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1]" : "=v"(Xm) : "v"(e2), "s"(h2), "v"(q));    // conj(E - Q)
}

// MEL = false: spectrum out.  MEL = true: |X|^power -> band-compressed mel filterbank -> log(. + eps)
// out [rows, frames, mel_M] (contrib/je/modules/features.py:171-176, 297-330): the spectrum never
// leaves LDS, HBM traffic per frame drops from 512 + 2056 B to 512 + 4 * mel_M B.
// (the fused (log-)mel instantiation of the 512-point plan needs 169 registers: at three workgroups per CU - a budget of 168 - it spilled ONE
//  into scratch; it runs at two, the plain instantiation keeps three)
template <class PL, bool MEL>
__global__ __launch_bounds__(256, (MEL && PL::INV_WAVES > 2) ? 2 : PL::INV_WAVES) void stft_fwd_kernel(const FwdArgs A) {
    constexpr int M = PL::M, F = PL::F, FPW = PL::FPW, FS = PL::FS, LPF = PL::LPF;
    constexpr int NP = M / 2 + 1;   // bin pairs (k, M-k) per frame
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WaveLds<PL> S(smem, 4);
    // mel tables behind the wave buffers: lo | cnt | off [mel_M] int32, weights [mel_nnz] float
    float* mw = reinterpret_cast<float*>(smem + ((WaveLds<PL>::bytes(4, 0) + 15) & ~(size_t)15));   // 16-byte aligned
    int32_t* mlo = reinterpret_cast<int32_t*>(mw + A.mel_nnz);
    int32_t* mcnt = mlo + A.mel_M;
    int32_t* moff = mcnt + A.mel_M;
    if (MEL) {
        for (int i = threadIdx.x; i < A.mel_M; i += 256) {
            mlo[i] = A.mel_lo[i];
            mcnt[i] = A.mel_cnt[i];
            moff[i] = A.mel_off[i];
        }
        for (int i = threadIdx.x; i < A.mel_nnz; i += 256) mw[i] = A.mel_w[i];
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int fl = lane / LPF, l = lane - fl * LPF;
    cpx* wbuf = S.buf + wave * FPW * FS;
    const float es = A.edge_scale;

    load_tables_w<PL>(S, A.window, A.twiddle, A.g.L, tid, 256);
    __syncthreads();

    // persistent wavefronts: work item = FPW consecutive frames of one row; nchunks items per row.
    // The samples of the NEXT item are fetched into registers before the current item's FFT and
    // epilogue (interior items: one immediate-offset float2 load per FFT input), so that HBM
    // latency hides behind compute instead of heading every item.
    struct Item {
        int b, tw0, n_b;
        const float* xrow;
        bool interior;
    };
    const unsigned items = (unsigned)(A.batch * A.nchunks);
    auto decode = [&](unsigned item) {
        Item it;
        it.b = (int)(item / (unsigned)A.nchunks);
        it.tw0 = (int)(item - (unsigned)it.b * (unsigned)A.nchunks) * FPW;
        it.n_b = A.row_samples ? A.row_samples[it.b] : (int)A.num_samples;
        it.xrow = A.x + (long long)it.b * A.x_row_stride;
        const int x_first = it.tw0 * A.g.shift - A.g.pad_left;
        const int x_last = (it.tw0 + FPW - 1) * A.g.shift - A.g.pad_left + PL::SIZE;
        it.interior = A.aligned2 && x_first >= 0 && x_last <= it.n_b && !(A.dbg & 4);   // wave-uniform
        return it;
    };
    float2 pre[PL::R1];
    auto issue = [&](const Item& it) {
        const float2* __restrict__ px =
            reinterpret_cast<const float2*>(it.xrow + it.tw0 * A.g.shift - A.g.pad_left + fl * A.g.shift) + l;
        if (l < PL::R2) {
#pragma unroll
            for (int n1 = 0; n1 < PL::R1; ++n1) pre[n1] = px[PL::R2 * n1];
        }
    };
    const unsigned stride = gridDim.x * 4;
    unsigned item = blockIdx.x * 4 + wave;
    Item cur = decode(item < items ? item : 0);
    bool cur_pre = item < items && cur.interior;
    if (cur_pre) issue(cur);
    for (; item < items; item += stride) {
        const int b = cur.b, tw0 = cur.tw0, n_b = cur.n_b;
        const int frames_b = (int)row_frames_of(A.g, n_b);
        cpx a[PL::R1];
        if (cur_pre) {
            const cpx* pw = reinterpret_cast<const cpx*>(S.win) + l;
            if (l < PL::R2) {
#pragma unroll
                for (int n1 = 0; n1 < PL::R1; ++n1) a[n1] = cpx{pre[n1].x, pre[n1].y} * pw[PL::R2 * n1];
            }
        } else if (!(A.dbg & 4)) {
            load_frames<PL>(a, cur.xrow, n_b, tw0, fl, l, A.g, A.aligned2, S.win);
        } else {
            for (int i = 0; i < PL::R1; ++i) a[i] = cpx{(float)(lane + i), 1.f};
        }
        {
            const unsigned nxt = item + stride;
            cur = decode(nxt < items ? nxt : 0);
            cur_pre = nxt < items && cur.interior;
            if (cur_pre) issue(cur);
        }
        if (!(A.dbg & 2))
            fft_to_lds_wave<PL, false>(a, wbuf + fl * FS, l, S.tw1t);
        else
            for (int i = 0; i < PL::R1; ++i) wbuf[fl * FS + i * LPF + l] = a[i];

        const int nfr = min(FPW, (int)A.out_frames - tw0);
        if (MEL) {
            // |X[k]|^power of all pairs into registers, then written DENSELY (frame f at float offset
            // f * 2 FS, 16-byte aligned rows) over the spectrum; zero for frames past the row's end
            constexpr int NITP = (FPW * NP + 63) / 64;
            float pk[NITP], pm[NITP];
#pragma unroll
            for (int it = 0; it < NITP; ++it) {
                const int p = min(lane + 64 * it, FPW * NP - 1);
                const int f = p / NP, k = p - f * NP;
                cpx Xk, Xm;
                split_pair<PL>(wbuf + f * FS, S.tws, k, Xk, Xm);
                pk[it] = Xk.x * Xk.x + Xk.y * Xk.y;
                pm[it] = Xm.x * Xm.x + Xm.y * Xm.y;
                if (A.mel_power == 1) {
                    pk[it] = sqrtf(pk[it]);
                    pm[it] = sqrtf(pm[it]);
                }
                if (tw0 + f >= frames_b) pk[it] = pm[it] = 0.f;
            }
            wave_sync();
            float* P = reinterpret_cast<float*>(wbuf);
#pragma unroll
            for (int it = 0; it < NITP; ++it) {
                const int p = lane + 64 * it;
                if (p < FPW * NP) {
                    const int f = p / NP, k = p - f * NP, km = M - k;
                    P[f * 2 * FS + k] = pk[it];
                    if (km != k) P[f * 2 * FS + km] = pm[it];
                }
            }
            constexpr int PADN = ((F + 3) & ~3) + 8 - F;      // the 8-bin groups may reach past bin M: zeros
            for (int i = lane; i < FPW * PADN; i += 64) P[(i / PADN) * 2 * FS + F + i % PADN] = 0.f;
            wave_sync();
            // one lane per (frame, filter): the filter's band in 16-byte aligned groups of 8 bins
            // (weights zero-padded), two b128 reads of the spectrum and two of the weights per group
            float* __restrict__ orow = A.out + ((long long)b * A.out_frames + tw0) * A.mel_M;
            const int nout = nfr * A.mel_M;
            const float4* P4 = reinterpret_cast<const float4*>(P);
            const float4* W4 = reinterpret_cast<const float4*>(mw);
            for (int o = lane; o < nout; o += 64) {
                const int f = o / A.mel_M, m = o - f * A.mel_M;
                const float4* pw = P4 + f * (FS / 2) + mlo[m];
                const float4* w = W4 + moff[m];
                const int cnt = mcnt[m];
                float acc = 0.f;
                for (int i = 0; i < cnt; ++i) {
                    const float4 p0 = pw[2 * i], p1 = pw[2 * i + 1], w0 = w[2 * i], w1 = w[2 * i + 1];
                    acc = fmaf(p0.x, w0.x, acc);
                    acc = fmaf(p0.y, w0.y, acc);
                    acc = fmaf(p0.z, w0.z, acc);
                    acc = fmaf(p0.w, w0.w, acc);
                    acc = fmaf(p1.x, w1.x, acc);
                    acc = fmaf(p1.y, w1.y, acc);
                    acc = fmaf(p1.z, w1.z, acc);
                    acc = fmaf(p1.w, w1.w, acc);
                }
                orow[o] = A.mel_log ? logf(acc + A.mel_eps) : acc;
            }
            wave_sync();   // the next item's transposition reuses wbuf
            continue;
        }
        float* __restrict__ orow = A.out + ((long long)b * A.out_frames + tw0) * (2 * F);
        const bool fast = nfr == FPW && tw0 + FPW <= frames_b && es == 1.f &&
                          A.layout == PTMI_LAYOUT_INTERLEAVED && !(A.dbg & 1);
        if (fast) {
            // all frames live: pairs (k, M-k), k = 0..M/2-1, then the self-paired middle bins
            float2* __restrict__ o2 = reinterpret_cast<float2*>(orow);
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
#pragma unroll
                for (int k0 = 0; k0 < M / 2; k0 += 64) {
                    const int k = k0 + lane;
                    if (M / 2 >= 64 || k < M / 2) {
                        cpx Xk, Xm;
                        split_pair<PL>(wbuf + f * FS, S.tws, k, Xk, Xm);
                        o2[f * F + k] = make_float2(Xk.x, Xk.y);
                        o2[f * F + M - k] = make_float2(Xm.x, Xm.y);
                    }
                }
            }
            if (lane < FPW) {
                cpx Xk, Xm;
                split_pair<PL>(wbuf + lane * FS, S.tws, M / 2, Xk, Xm);
                o2[lane * F + M / 2] = make_float2(Xk.x, Xk.y);
            }
        } else {
            for (int p = lane; p < nfr * NP; p += 64) {
                const int f = p / NP, k = p - f * NP;
                cpx Xk, Xm;
                split_pair<PL>(wbuf + f * FS, S.tws, k, Xk, Xm);
                if (k == 0) {   // DC and Nyquist
                    Xk = cpx{Xk.x * es, es == 1.f ? Xk.y : 0.f};
                    Xm = cpx{Xm.x * es, es == 1.f ? Xm.y : 0.f};
                }
                if (tw0 + f >= frames_b) Xk = Xm = cpx{0.f, 0.f};
                const int km = M - k;
                if ((A.dbg & 1) && Xk.x != 123456.f) continue;
                if (A.layout == PTMI_LAYOUT_INTERLEAVED) {
                    float2* o2 = reinterpret_cast<float2*>(orow) + f * F;
                    o2[k] = make_float2(Xk.x, Xk.y);
                    if (km != k) o2[km] = make_float2(Xm.x, Xm.y);
                } else {
                    float* o1 = orow + f * 2 * F;
                    o1[k] = Xk.x;
                    o1[F + k] = Xk.y;
                    if (km != k) {
                        o1[km] = Xm.x;
                        o1[F + km] = Xm.y;
                    }
                }
            }
        }
        wave_sync();   // the next item's transposition reuses wbuf
    }
}

// ------------------------------------------------------------------------------------------------
// Fused PIT front-end: Y_abs [B,T,F], X_abs / cos_phase_difference [B,T,K,F].
// One wavefront owns FPW frames of one example for ALL K+1 signals: it transforms the mixture first
// and keeps the unit phasors of Y in its private LDS slice, then each source, forming
// cos(angle(Y) - angle(X)) = Re(y_hat conj x_hat)  (angle(0) := 0 like np.angle) on the way out.
__device__ __forceinline__ void mag_phasor(cpx X, float& mag, cpx& ph) {
    const float p = X.x * X.x + X.y * X.y;
    const float r = __builtin_amdgcn_rsqf(p);            // 1/|X|; inf at 0 and for a DENORMAL p (the hardware flushes the operand)
    mag = p * r;
    ph = cpx{X.x * r, X.y * r};
    // (a NaN bin - a NaN sample in the frame - fails the comparison and stays NaN in the magnitude and in the phasor like np.abs /
    //  np.angle of the reference's features (pit/data.py:67-75), so that the loss and with it the Trainer's non-finite check see it)
    if (__builtin_expect(p < 1.1754944e-38f, 0)) {
        // zero or below the normal range (rare: a frame whose only samples meet window taps of ~1e-17 - a Blackman window's first tap
        // - gives |X|^2 ~ 1e-38; `p * rsq(p)` was inf there, round 6): the same arithmetic on 2^64 X
        const cpx Xs{X.x * 0x1p64f, X.y * 0x1p64f};
        const float ps = Xs.x * Xs.x + Xs.y * Xs.y;
        const float rs = __builtin_amdgcn_rsqf(ps);
        const bool zero = ps < 1.1754944e-38f;           // |X| < 2^-127: zero like the exact zero (angle(0) := 0)
        mag = zero ? 0.f : ps * rs * 0x1p-64f;
        ph = zero ? cpx{1.f, 0.f} : cpx{Xs.x * rs, Xs.y * rs};
    }
}

template <class PL>
__global__ __launch_bounds__(256, 2) void pit_features_kernel(const FwdArgs A) {
    constexpr int M = PL::M, F = PL::F, FPW = PL::FPW, FS = PL::FS, LPF = PL::LPF;
    constexpr int YS = WaveLds<PL>::YS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WaveLds<PL> S(smem, 4);
    const int nsig = A.s ? A.K + 1 : 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int fl = lane / LPF, l = lane - fl * LPF;
    cpx* wbuf = S.buf + wave * FPW * FS;
    cpx* yph = S.yph + wave * FPW * YS;

    load_tables_w<PL>(S, A.window, A.twiddle, A.g.L, tid, 256);
    __syncthreads();

    // persistent wavefronts: work item = FPW frames of one example; the wavefront transforms the
    // mixture first (magnitudes out, unit phasors kept in its private LDS slice), then every source
    // (magnitudes and cos of the phase difference out).  No workgroup barrier on the way; the samples
    // of the NEXT signal are fetched into registers before the current one is transformed.
    struct Task {
        int b, tw0, n_b;
        const float* xrow;
        bool interior;
    };
    const unsigned items = (unsigned)(A.batch * A.nchunks);
    const unsigned stride = gridDim.x * 4;
    auto decode = [&](unsigned item, int q) {
        Task I;
        // frame-group major, example minor: neighbouring wavefronts hold the SAME frames of neighbouring examples, whose packed rows
        // (t * batch + b) and 16-byte plane chunks (16 rows of a tile side by side) then reach the L2 together and leave it as whole
        // lines (example major, rounds 2 - 4: 16-byte partial lines once the planes outgrow the L2 - B = 64: +43 us for the packed outputs)
        const unsigned chunk = item / (unsigned)A.batch;
        I.b = (int)(item - chunk * (unsigned)A.batch);
        I.tw0 = (int)chunk * FPW;
        I.n_b = A.row_samples ? A.row_samples[I.b] : (int)A.num_samples;
        I.xrow = (q == 0) ? A.x + (long long)I.b * A.x_row_stride
                          : A.s + ((long long)I.b * A.K + (q - 1)) * A.x_row_stride;
        const int x_first = I.tw0 * A.g.shift - A.g.pad_left;
        const int x_last = (I.tw0 + FPW - 1) * A.g.shift - A.g.pad_left + PL::SIZE;
        I.interior = A.aligned2 && x_first >= 0 && x_last <= I.n_b && !(A.dbg & 4);   // wave-uniform
        return I;
    };
    float2 pre[PL::R1];
    auto issue = [&](const Task& I) {
        const float2* __restrict__ px =
            reinterpret_cast<const float2*>(I.xrow + I.tw0 * A.g.shift - A.g.pad_left + fl * A.g.shift) + l;
        if (l < PL::R2) {
#pragma unroll
            for (int n1 = 0; n1 < PL::R1; ++n1) pre[n1] = px[PL::R2 * n1];
        }
    };
    unsigned item = blockIdx.x * 4 + wave;
    Task cur = decode(item < items ? item : 0, 0);
    bool cur_pre = item < items && cur.interior;
    if (cur_pre) issue(cur);
    for (; item < items; item += stride) {
        for (int q = 0; q < nsig; ++q) {
            const int b = cur.b, tw0 = cur.tw0;
            const int frames_b = (int)row_frames_of(A.g, cur.n_b);
            const int nfr = min(FPW, (int)A.out_frames - tw0);
            {
                cpx a[PL::R1];
                if (cur_pre) {
                    const cpx* pw = reinterpret_cast<const cpx*>(S.win) + l;
                    if (l < PL::R2) {
#pragma unroll
                        for (int n1 = 0; n1 < PL::R1; ++n1) a[n1] = cpx{pre[n1].x, pre[n1].y} * pw[PL::R2 * n1];
                    }
                } else if (!(A.dbg & 4)) {
                    load_frames<PL>(a, cur.xrow, cur.n_b, tw0, fl, l, A.g, A.aligned2, S.win);
                } else {
                    for (int i = 0; i < PL::R1; ++i) a[i] = cpx{(float)(lane + i), 1.f};
                }
                {   // next signal of this item, or the mixture of the wavefront's next item
                    const bool last = q + 1 == nsig;
                    const unsigned nitem = last ? item + stride : item;
                    cur = decode(nitem < items ? nitem : 0, last ? 0 : q + 1);
                    cur_pre = nitem < items && cur.interior;
                    if (cur_pre) issue(cur);
                }
                if (!(A.dbg & 2)) fft_to_lds_wave<PL, false>(a, wbuf + fl * FS, l, S.tw1t);
            }
            const int fstride = (q == 0) ? F : A.K * F;
            const long long o0 = (q == 0) ? ((long long)b * A.out_frames + tw0) * F
                                          : (((long long)b * A.out_frames + tw0) * A.K + (q - 1)) * F;
            float* __restrict__ mag_out = (q == 0 ? A.out : A.X_abs) + o0;
            float* __restrict__ cp = A.cos_pd + o0;      // q > 0 only
            {
                // Bins OWNED per lane (round 5).  The kernel is bound by vector-instruction issue, and more than half of its instructions
                // were this epilogue's: with a flat (frame, bin pair) index per lane (rounds 2 - 4) every pair paid a division by NP,
                // 64-bit store addresses and a per-lane validity select; same arithmetic per bin, outputs bit-identical, 6 - 12 % less
                // time (profiles/r5_graph_head.txt).  A lane owns bin pair (k, M - k), k = kb + (lane mod KPL), of frame
                // f0 + lane / KPL (KPL = min(64, M / 2): for M >= 128 one frame at a time, f wave-uniform): the twiddle is read once per
                // k and serves all frames, row bases and the validity are scalar, stores take a scalar base + 32-bit lane offset.
                // Bin M / 2 (its own partner) is the last pass, one lane per frame.
                constexpr int HM = M / 2, KPL = HM < 64 ? HM : 64, FG = 64 / KPL;
                const int fo = FG > 1 ? lane / KPL : 0, kl = FG > 1 ? lane - fo * KPL : lane;
                auto one_pair = [&](int f, int k, bool two, cpx w) {
                    const int km = M - k;
                    const cpx z1 = wbuf[f * FS + (k & (M - 1))], z2 = wbuf[f * FS + (km & (M - 1))];
                    cpx yk{0.f, 0.f}, ym{0.f, 0.f};
                    if (q > 0) {
                        yk = yph[f * YS + k];
                        ym = yph[f * YS + km];
                    }
                    const bool valid = tw0 + f < frames_b;
                    cpx Xk, Xm, pk, pm;
                    float mk, mm;
                    split_vals(z1, z2, w, Xk, Xm);
                    if (!valid) Xk = Xm = cpx{0.f, 0.f};
                    mag_phasor(Xk, mk, pk);
                    mag_phasor(Xm, mm, pm);
                    float* mrow = mag_out + f * fstride;
                    mrow[k] = mk;
                    if (two) mrow[km] = mm;
                    if (q == 0) {
                        yph[f * YS + k] = pk;
                        if (two) yph[f * YS + km] = pm;
                        if (A.lp_out && valid) {
                            // pit/model.py:91-94: pack_sequence -> log1p, and the fp16 planes the first projection multiplies.
                            // log1p on the hardware log2 (v_log_f32): the library log1pf costs ~35 instructions per value; 1 + |Y| rounds
                            // with an absolute error <= 6e-8, far inside the tolerance the features are compared at (atol 1e-6).  The
                            // values are parked in the spectrum slots this lane has just consumed (k and M - k; slot M is nobody's):
                            // the planes are written in 16-byte chunks of 8 bins by the pass behind this loop
                            const int t = tw0 + f;
                            const long long prow = (A.lp_offs ? A.lp_offs[t] : (long long)t * A.batch) + b;
                            const float lk = __log2f(1.f + mk) * 0.69314718f, lm = __log2f(1.f + mm) * 0.69314718f;
                            float* lrow = A.lp_out + prow * F;
                            lrow[k] = lk;
                            if (two) lrow[km] = lm;
                            if (A.lp_planes) {
                                wbuf[f * FS + k].x = lk;
                                if (two) wbuf[f * FS + km].x = lm;
                            }
                        }
                    } else {
                        float* crow = cp + f * fstride;
                        const cpx ck = yk * pk, cm = ym * pm;
                        crow[k] = valid ? ck.x + ck.y : 0.f;
                        if (two) crow[km] = valid ? cm.x + cm.y : 0.f;
                    }
                };
                for (int kb = 0; kb < HM; kb += KPL) {
                    const int k = kb + kl;
                    const cpx w = S.tws[k];
#pragma unroll
                    for (int f0 = 0; f0 < FPW; f0 += FG) {
                        const int f = f0 + fo;
                        if (f < nfr) one_pair(f, k, true, w);
                    }
                }
                if (lane < nfr) one_pair(lane, HM, false, S.tws[HM]);
            }
            if (q == 0 && A.lp_planes) {
                // fp16 (hi, lo) planes of 2^9 log1p|Y| in MFMA-fragment order: chunk (bins 8 c .. 8 c + 7 of one packed row) = one
                // 16-byte store per plane; bins past F are zero (csrc/gemm_planes.hip reads whole 32-wide blocks)
                wave_sync();
                // (round 6: the chunk count is the plan's - F is a template constant, the host passes lp_kb = (F + 31) / 32 -, so the
                //  split of a chunk index into (frame, chunk) is a multiplication; full chunks take no per-bin validity select; the
                //  conversions are written on 2-vectors: v_pk_mul_f32 / v_cvt_pk_f16_f32 / v_pk_add_f32 on gfx950, bit for bit the
                //  scalar round-to-nearest conversions)
                constexpr int KBC = (F + 31) / 32, chunks = KBC * 4, FULL = F / 8;
                typedef float f2v __attribute__((ext_vector_type(2)));
                typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                typedef _Float16 h8v __attribute__((ext_vector_type(8)));
                for (int c = lane; c < nfr * chunks; c += 64) {
                    const int f = c / chunks, ch = c - f * chunks, t = tw0 + f;
                    if (t >= frames_b) continue;
                    const long long prow = (A.lp_offs ? A.lp_offs[t] : (long long)t * A.batch) + b;
                    const float* src = reinterpret_cast<const float*>(wbuf + f * FS + ch * 8);          // the parked values: every second float
                    f2v sv[4];
                    if (ch < FULL) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) sv[e] = f2v{src[4 * e], src[4 * e + 2]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            sv[e] = f2v{ch * 8 + 2 * e < F ? src[4 * e] : 0.f, ch * 8 + 2 * e + 1 < F ? src[4 * e + 2] : 0.f};
                    }
                    h8v hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f2v v = sv[e] * 512.f;
                        const h2v h = __builtin_convertvector(v, h2v);
                        const h2v l = __builtin_convertvector(v - __builtin_convertvector(h, f2v), h2v);
                        hi[2 * e] = h[0];
                        hi[2 * e + 1] = h[1];
                        lo[2 * e] = l[0];
                        lo[2 * e + 1] = l[1];
                    }
                    _Float16* o = A.lp_planes + (((prow >> 4) * KBC + (ch >> 2)) * 2) * 512 + ((ch & 3) * 16 + (prow & 15)) * 8;
                    *reinterpret_cast<h8v*>(o) = hi;
                    *reinterpret_cast<h8v*>(o + 512) = lo;
                }
            }
            wave_sync();   // the next transposition reuses wbuf; yph is read by the following signals
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct InvArgs {
    const float* spec;
    const int32_t* row_frames;
    const float* syn_window;
    const cpx* twiddle;
    float* out;
    long long num_frames;
    long long out_samples;
    long long out_row_stride;
    long long cut_left;
    long long batch;
    int nchunks;    // runs per row
    int layout;
    int halo;       // ceil(L / shift) - 1 frames whose tails reach into a run
    int run_hops;   // hops (= frames) whose output a run owns
    int groups;     // groups of FPW frames a run walks (halo + run_hops + tail flush)
    int dbg;        // PTMI_STFT_DBG ablation bits (1: no stores, 2: no FFT, 4: no loads, 8: no OLA terms)
    float edge_scale;
    Geo g;
};

// Inverse STFT as wave-private persistent pipelines.  Work item = a RUN of consecutive frames of one
// row, owned by ONE wavefront from the spectrum loads to the sample stores: the frames are taken in
// groups of FPW, each group is inverse-transformed in LDS (the synthesis window is applied by the
// last FFT stage) and overlap-added into a small wave-private ring that carries the (L - shift)
// samples still awaiting later frames; the first FPW*shift samples of every group are final and
// stream out.  A run starts `halo` = ceil(L/shift) - 1 frames early (their tails reach into the
// run's first hop), so every output sample is summed by exactly one wavefront in a fixed order
// (deterministic, no atomics).  Spectra of interior groups are fetched one group AHEAD into
// registers (FPW consecutive rows = one contiguous chunk), hiding HBM latency behind FFT + OLA.
template <class PL>
__device__ __forceinline__ void ifft_windowed_to_lds_wave(cpx (&a)[PL::R1], cpx* fbuf, int l,
                                                          const cpx* tw1t, const cpx* wsyn2) {
    constexpr int R1 = PL::R1, R2 = PL::R2, P = PL::P;
    if (l < R2) {
        fft_dif<R1, true>(a);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) fbuf[k1 * P + l] = cmulc(a[bitrev<R1>(k1)], tw1t[k1 * PL::LPF + l]);
    }
    wave_sync();
    cpx c[R2];
    if (l < R1) {
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) c[n2] = fbuf[l * P + n2];
    }
    wave_sync();
    if (l < R1) {
        fft_dif<R2, true>(c);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {   // z[n] = x[2n] + i x[2n+1], times the window pair
            fbuf[l + R1 * k2] = c[bitrev<R2>(k2)] * wsyn2[l + R1 * k2];
        }
    }
    wave_sync();
}

template <class PL>
__global__ __launch_bounds__(256, PL::INV_WAVES) void istft_kernel(const InvArgs A) {
    constexpr int M = PL::M, F = PL::F, FPW = PL::FPW, FS = PL::FS, LPF = PL::LPF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx* tws = reinterpret_cast<cpx*>(smem);
    cpx* tw1t = tws + F + 1;
    float* wsyn = reinterpret_cast<float*>(tw1t + PL::R1 * LPF);
    cpx* bufs = reinterpret_cast<cpx*>(wsyn + PL::SIZE);
    const int shift = A.g.shift, L = A.g.L;
    const int plen = max(L - shift, 0);                 // samples carried between groups
    const int plen_pad = max((plen + 3) & ~3, 4);
    float* rings = reinterpret_cast<float*>(bufs + 4 * FPW * FS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane / LPF, l = lane - fl * LPF;
    cpx* wbuf = bufs + wave * FPW * FS;
    float* ring0 = rings + wave * 2 * plen_pad;

    for (int i = tid; i <= M; i += 256) tws[i] = A.twiddle[i];
    for (int i = tid; i < PL::SIZE; i += 256) wsyn[i] = (i < L) ? A.syn_window[i] : 0.f;
    for (int i = tid; i < PL::R1 * LPF; i += 256) {
        const int k1 = i / LPF, ll = i - k1 * LPF;
        tw1t[i] = tw_full<M>(A.twiddle, (2 * ll * k1) % PL::SIZE);
    }
    __syncthreads();

    const int span = (FPW - 1) * shift + L;             // samples a group of FPW frames touches
    const int adv = FPW * shift;                        // of which this many become final
    const int pend = max(span, adv);                    // (shift > L leaves silent gaps to write)
    constexpr int nit = (FPW * F + 63) / 64;
    const bool interleaved = A.layout == PTMI_LAYOUT_INTERLEAVED;
    const unsigned items = (unsigned)(A.batch * A.nchunks);
    for (unsigned item = blockIdx.x * 4 + wave; item < items; item += gridDim.x * 4) {
        const int b = (int)(item / (unsigned)A.nchunks);
        const int run = (int)(item - (unsigned)b * (unsigned)A.nchunks);
        const long long T_b = A.row_frames ? (long long)A.row_frames[b] : A.num_frames;
        const long long own0 = (long long)run * A.run_hops * shift;      // owned OLA range starts here
        const long long tfirst = (long long)run * A.run_hops - A.halo;   // first frame of the first group
        // owned and wanted positions, relative to own0:  [rel_lo, rel_hi)
        const long long lo64 = A.cut_left - own0, hi64 = A.cut_left + A.out_samples - own0;
        const int own_len = A.run_hops * shift;
        const int rel_lo = (int)min(max(lo64, 0LL), (long long)own_len);
        const int rel_hi = (A.dbg & 1) ? 0 : (int)min(max(hi64, 0LL), (long long)own_len);
        float* outb = A.out + (long long)b * A.out_row_stride + (own0 - A.cut_left);
        const float* rowbase = A.spec + (long long)b * A.num_frames * 2 * F;
        for (int i = lane; i < plen_pad; i += 64) ring0[i] = 0.f;        // the ring half group 0 reads

        cpx pre[nit];
        auto issue = [&](long long tg) {   // FPW consecutive interleaved rows = one contiguous chunk
            const float2* base = reinterpret_cast<const float2*>(rowbase) + tg * F;
#pragma unroll
            for (int it = 0; it < nit; ++it) {
                const float2 v = base[min(lane + 64 * it, FPW * F - 1)];
                pre[it] = cpx{v.x, v.y};
            }
        };
        // wave-uniform group classes: interior groups are prefetched, edge groups load in place
        auto is_live = [&](long long tg) { return tg < T_b && tg + FPW > 0; };
        auto is_interior = [&](long long tg) {
            return interleaved && tg >= 0 && tg + FPW <= T_b && !(A.dbg & 4);
        };
        if (is_interior(tfirst)) issue(tfirst);

        for (int g = 0; g < A.groups; ++g) {
            const long long tg = tfirst + (long long)g * FPW;
            float* rin = ring0 + (g & 1) * plen_pad;
            float* rout = ring0 + ((g + 1) & 1) * plen_pad;
            const bool live = is_live(tg);
            // (a) raw one-sided spectra, frame f at wbuf[f * F ...] (dense: the FFT re-lays them out)
            if (is_interior(tg)) {
#pragma unroll
                for (int it = 0; it < nit; ++it)
                    if (it + 1 < nit || lane + 64 * it < FPW * F) wbuf[lane + 64 * it] = pre[it];
            } else if (live && !(A.dbg & 4)) {
                const long long tmax = T_b > 0 ? T_b - 1 : 0;
#pragma unroll 4
                for (int it = 0; it < nit; ++it) {
                    const int idx = min(lane + 64 * it, FPW * F - 1);
                    const int f = idx / F, k = idx - f * F;
                    const long long t = tg + f;
                    const long long r = min(max(t, 0LL), tmax);
                    cpx X;
                    if (interleaved) {
                        const float2 v = *reinterpret_cast<const float2*>(rowbase + (r * F + k) * 2);
                        X = cpx{v.x, v.y};
                    } else {
                        X = cpx{rowbase[r * 2 * F + k], rowbase[r * 2 * F + F + k]};
                    }
                    if (t < 0 || t >= T_b) X = cpx{0.f, 0.f};
                    if (lane + 64 * it < FPW * F) wbuf[idx] = X;
                }
            }
            if (g + 1 < A.groups && is_interior(tg + FPW)) issue(tg + FPW);
            wave_sync();
            // (b) Z'[k] = (X[k] + conj X[M-k]) + i e^{+i pi k / M} (X[k] - conj X[M-k]),  k = R2*i1 + l
            if (live && !(A.dbg & 2)) {
                cpx a[PL::R1];
                const cpx* raw = wbuf + fl * F;
#pragma unroll
                for (int i1 = 0; i1 < PL::R1; ++i1) {
                    const int k = PL::R2 * i1 + l;
                    cpx v = cpx{0.f, 0.f};
                    if (l < PL::R2) {
                        cpx x1 = raw[k], x2 = raw[M - k];
                        const cpx w = tws[k];
                        if (i1 == 0 && l == 0) {
                            // DC / Nyquist: imaginary parts never reach the output (_stft.py:37-40)
                            x1 = cpx{x1.x * A.edge_scale, 0.f};
                            x2 = cpx{x2.x * A.edge_scale, 0.f};
                        }
                        // S = X[k] + conj X[M-k],  D = X[k] - conj X[M-k],  Z' = S + i conj(w) D
                        v = add_i(add_conj(x1, x2), cmulc(sub_conj(x1, x2), w));
                    }
                    a[i1] = v;
                }
                wave_sync();   // every lane has consumed raw before the transposition overwrites it
                ifft_windowed_to_lds_wave<PL>(a, wbuf + fl * FS, l, tw1t, reinterpret_cast<const cpx*>(wsyn));
            }
            // (c) overlap-add: wbuf viewed as floats holds windowed frame f at [f * 2 FS, f * 2 FS + size).
            // Frame f reaches group position p with its sample j = p - f * shift if 0 <= j < L.  Positions
            // are taken in batches of U per lane: first ALL LDS reads of a batch (unconditional: clamped
            // index + select, so they issue back to back), then its stores.
            const float* fr = reinterpret_cast<const float*>(wbuf);
            const int rel0 = (g * FPW - A.halo) * shift;     // group position 0 relative to own0
            const bool terms = live && !(A.dbg & 8);
            constexpr int U = 8;
            for (int p0 = lane; p0 < pend; p0 += 64 * U) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + 64 * u;
                    const float carried = rin[min(p, plen_pad - 1)];
                    v[u] = (p < plen) ? carried : 0.f;
                }
                if (terms) {
                    float x[U][FPW];
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int f = 0; f < FPW; ++f) {
                            const int j = p0 + 64 * u - f * shift;
                            x[u][f] = fr[f * 2 * FS + ((unsigned)j < (unsigned)L ? j : 0)];
                        }
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int f = 0; f < FPW; ++f) {
                            const int j = p0 + 64 * u - f * shift;
                            v[u] += ((unsigned)j < (unsigned)L) ? x[u][f] : 0.f;
                        }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + 64 * u;
                    if (p < adv) {
                        const int rel = rel0 + p;
                        if (rel >= rel_lo && rel < rel_hi) outb[rel] = v[u];
                    } else if (p - adv < plen) {
                        rout[p - adv] = v[u];
                    }
                }
            }
            wave_sync();   // span - adv == plen: rout is completely rewritten; wbuf is free again
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic (any even size) fallbacks: direct DFT, O(F*L) per frame.  Correctness path for sizes
// that are not 64..2048 powers of two (e.g. the doctests' STFT(100..200, ...)).
__device__ __forceinline__ cpx tw_any(const cpx* tw, int size, long long j) {
    const int M = size >> 1;
    int r = (int)(j % size);
    if (r <= M) return tw[r];
    const cpx v = tw[size - r];   // W^(size-r) = conj(W^r)
    return cpx{v.x, -v.y};
}

__global__ __launch_bounds__(256) void stft_generic_kernel(const FwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xw = reinterpret_cast<float*>(smem);
    const int F = A.g.size / 2 + 1, M = A.g.size / 2;
    const long long b = blockIdx.x / A.out_frames;
    const long long t = blockIdx.x - b * A.out_frames;
    const long long n_b = A.row_samples ? (long long)A.row_samples[b] : A.num_samples;
    const long long frames_b = row_frames_of(A.g, n_b);
    const float* xrow = A.x + b * A.x_row_stride;
    for (int j = threadIdx.x; j < A.g.L; j += 256) {
        const long long xi = t * A.g.shift - A.g.pad_left + j;
        xw[j] = (xi >= 0 && xi < n_b) ? xrow[xi] * A.window[j] : 0.f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < F; k += 256) {
        float re = 0.f, im = 0.f;
        if (t < frames_b) {
            for (int j = 0; j < A.g.L; ++j) {
                const cpx w = tw_any(A.twiddle, A.g.size, (long long)k * j);
                re += xw[j] * w.x;
                im += xw[j] * w.y;
            }
            if (k == 0 || k == M) {
                re *= A.edge_scale;
                im = A.edge_scale == 1.f ? im : 0.f;
            }
        }
        float* orow = A.out + (b * A.out_frames + t) * 2 * F;
        if (A.layout == PTMI_LAYOUT_INTERLEAVED) {
            orow[2 * k] = re;
            orow[2 * k + 1] = im;
        } else {
            orow[k] = re;
            orow[F + k] = im;
        }
    }
}

// The fused PIT features for any even size (direct DFT; same outputs as pit_features_kernel incl. the packed log1p rows and planes):
// one workgroup per (example, frame), the mixture first (unit phasors kept in LDS), then every source.
__global__ __launch_bounds__(256) void pit_features_generic_kernel(const FwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = A.g.size / 2 + 1;
    cpx* yph = reinterpret_cast<cpx*>(smem);                    // [F]
    float* xw = reinterpret_cast<float*>(yph + F);              // [L]
    const long long b = blockIdx.x / A.out_frames;
    const long long t = blockIdx.x - b * A.out_frames;
    const long long n_b = A.row_samples ? (long long)A.row_samples[b] : A.num_samples;
    const bool valid = t < row_frames_of(A.g, n_b);
    const int nsig = A.s ? A.K + 1 : 1;
    for (int q = 0; q < nsig; ++q) {
        const float* xrow = q == 0 ? A.x + b * A.x_row_stride : A.s + (b * A.K + (q - 1)) * A.x_row_stride;
        __syncthreads();            // (the previous signal's samples have been consumed)
        for (int j = threadIdx.x; j < A.g.L; j += 256) {
            const long long xi = t * A.g.shift - A.g.pad_left + j;
            xw[j] = (xi >= 0 && xi < n_b) ? xrow[xi] * A.window[j] : 0.f;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < F; k += 256) {
            cpx X{0.f, 0.f};
            if (valid) {
                for (int j = 0; j < A.g.L; ++j) {
                    const cpx w = tw_any(A.twiddle, A.g.size, (long long)k * j);
                    X.x += xw[j] * w.x;
                    X.y += xw[j] * w.y;
                }
            }
            float mag;
            cpx ph;
            mag_phasor(X, mag, ph);
            if (q == 0) {
                A.out[(b * A.out_frames + t) * F + k] = mag;
                yph[k] = ph;
                if (A.lp_out && valid) {
                    const long long prow = (A.lp_offs ? A.lp_offs[t] : t * A.batch) + b;
                    const float lv = __log2f(1.f + mag) * 0.69314718f;
                    A.lp_out[prow * F + k] = lv;
                    if (A.lp_planes) {
                        const float sv = lv * 512.f;
                        const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
                        _Float16* o = A.lp_planes + (((prow >> 4) * A.lp_kb + (k >> 5)) * 2) * 512 + (((k >> 3) & 3) * 16 + (prow & 15)) * 8 + (k & 7);
                        o[0] = hi;
                        o[512] = lo;
                    }
                }
            } else {
                const long long o = ((b * A.out_frames + t) * A.K + (q - 1)) * F + k;
                A.X_abs[o] = mag;
                const cpx c = yph[k] * ph;          // cos(angle(Y) - angle(X)) = Re(y_hat conj x_hat)
                A.cos_pd[o] = valid ? c.x + c.y : 0.f;
            }
        }
        if (q == 0 && A.lp_out && A.lp_planes && valid) {       // bins F .. 32 lp_kb - 1 of this row: zero (the GEMM reads whole k blocks)
            const long long prow = (A.lp_offs ? A.lp_offs[t] : t * A.batch) + b;
            for (int k = F + threadIdx.x; k < A.lp_kb * 32; k += 256) {
                _Float16* o = A.lp_planes + (((prow >> 4) * A.lp_kb + (k >> 5)) * 2) * 512 + (((k >> 3) & 3) * 16 + (prow & 15)) * 8 + (k & 7);
                o[0] = (_Float16)0.f;
                o[512] = (_Float16)0.f;
            }
        }
    }
}

__global__ __launch_bounds__(256) void istft_generic_kernel(const InvArgs A) {
    const int F = A.g.size / 2 + 1, M = A.g.size / 2;
    const long long per_row = (A.out_samples + 255) / 256;
    const long long b = blockIdx.x / per_row;
    const long long n = (blockIdx.x - b * per_row) * 256 + threadIdx.x;
    if (n >= A.out_samples) return;
    const long long T_b = A.row_frames ? (long long)A.row_frames[b] : A.num_frames;
    const long long o = n + A.cut_left;
    long long tlo = (o - A.g.L + 1 <= 0) ? 0 : (o - A.g.L + A.g.shift) / A.g.shift;
    long long thi = o / A.g.shift;
    if (thi > T_b - 1) thi = T_b - 1;
    float acc = 0.f;
    for (long long t = tlo; t <= thi; ++t) {
        const int j = (int)(o - t * A.g.shift);
        const float* row = A.spec + (b * A.num_frames + t) * 2 * F;
        float v = 0.f;
        for (int k = 0; k <= M; ++k) {
            float re, im;
            if (A.layout == PTMI_LAYOUT_INTERLEAVED) {
                re = row[2 * k];
                im = row[2 * k + 1];
            } else {
                re = row[k];
                im = row[F + k];
            }
            const cpx w = tw_any(A.twiddle, A.g.size, (long long)k * j);  // (cos, -sin)(2 pi k j / size)
            if (k == 0 || k == M) {
                v += A.edge_scale * re * w.x;
            } else {
                v += 2.f * (re * w.x + im * w.y);   // 2 Re(X e^{+i theta}) = 2 (re cos - im sin), w.y = -sin
            }
        }
        acc += v * A.syn_window[j];
    }
    A.out[b * A.out_row_stride + n] = acc;
}

// ------------------------------------------------------------------------------------------------


constexpr size_t kMaxSmem = 64 * 1024;  // keep >= 2 workgroups per CU (160 KiB LDS)

// Workgroups of `Kernel` that stay resident per CU (registers AND LDS), cached per thread.
template <auto Kernel>
static int blocks_per_cu(int threads, size_t smem) {
    thread_local size_t key_smem = ~(size_t)0;
    thread_local int key_threads = 0, blocks = 0;
    if (blocks < 1 || key_smem != smem || key_threads != threads) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, Kernel, threads, smem) != hipSuccess || blocks < 1)
            blocks = 1;
        key_smem = smem;
        key_threads = threads;
    }
    return blocks;
}

template <class PL>
static int launch_fwd(FwdArgs& A, long long batch, bool features, hipStream_t st) {
    const char* dbg_env = getenv("PTMI_STFT_DBG");
    A.dbg = dbg_env ? atoi(dbg_env) : 0;
    A.aligned2 = (A.x_row_stride % 2 == 0) && (A.g.shift % 2 == 0) && (A.g.pad_left % 2 == 0) &&
                 (reinterpret_cast<uintptr_t>(A.x) % 8 == 0) &&
                 (!A.s || reinterpret_cast<uintptr_t>(A.s) % 8 == 0);
    if (!features) {
        const bool mel = A.mel_w != nullptr;
        const size_t smem = WaveLds<PL>::bytes(4, 0) + (mel ? 16 + sizeof(int32_t) * 3 * A.mel_M + sizeof(float) * A.mel_nnz : 0);
        if (smem > kMaxSmem) return PTMI_E_UNSUPPORTED;
        A.batch = batch;
        A.nchunks = (int)((A.out_frames + PL::FPW - 1) / PL::FPW);      // work items per row
        const long long items = batch * A.nchunks;
        if (items <= 0) return PTMI_OK;
        if (items > 0x7fffffffLL) return PTMI_E_UNSUPPORTED;
        // persistent grid: as many workgroups as stay resident (LDS bound), never more than needed
        if (mel) {
            const long long resident = 256LL * blocks_per_cu<stft_fwd_kernel<PL, true>>(256, smem);
            hipLaunchKernelGGL((stft_fwd_kernel<PL, true>), dim3((unsigned)std::min((items + 3) / 4, resident)),
                               dim3(256), smem, st, A);
        } else {
            const long long resident = 256LL * blocks_per_cu<stft_fwd_kernel<PL, false>>(256, smem);
            hipLaunchKernelGGL((stft_fwd_kernel<PL, false>), dim3((unsigned)std::min((items + 3) / 4, resident)),
                               dim3(256), smem, st, A);
        }
        return launch_status();
    }
    const size_t smem = WaveLds<PL>::bytes(4, 4);
    A.batch = batch;
    A.nchunks = (int)((A.out_frames + PL::FPW - 1) / PL::FPW);          // work items per example
    const long long items = batch * A.nchunks;
    if (items <= 0) return PTMI_OK;
    if (items > 0x7fffffffLL) return PTMI_E_UNSUPPORTED;
    if (smem > 160 * 1024) return PTMI_E_UNSUPPORTED;
    if (smem > kMaxSmem) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pit_features_kernel<PL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    const long long resident = 256LL * blocks_per_cu<pit_features_kernel<PL>>(256, smem);
    const long long blocks = std::min((items + 3) / 4, resident);
    hipLaunchKernelGGL(pit_features_kernel<PL>, dim3((unsigned)blocks), dim3(256), smem, st, A);
    return launch_status();
}

static int dispatch_fwd(FwdArgs& A, long long batch, bool features, hipStream_t st) {
    if (A.g.L > A.g.size) return PTMI_E_INVALID;
    switch (A.g.size) {
        case 64: return launch_fwd<Plan<4, 8>>(A, batch, features, st);
        case 128: return launch_fwd<Plan<8, 8>>(A, batch, features, st);
        case 256: return launch_fwd<Plan<8, 16>>(A, batch, features, st);
        case 512: return launch_fwd<Plan<16, 16>>(A, batch, features, st);
        case 1024: return launch_fwd<Plan<16, 32>>(A, batch, features, st);
        case 2048: return launch_fwd<Plan<32, 32>>(A, batch, features, st);
        default: return PTMI_E_UNSUPPORTED;
    }
}

template <class PL>
static int launch_inv(InvArgs& A, long long batch, hipStream_t st) {
    const int shift = A.g.shift, L = A.g.L;
    A.halo = (L + shift - 1) / shift - 1;
    const int plen_pad = std::max((std::max(L - shift, 0) + 3) & ~3, 4);
    const size_t smem = sizeof(cpx) * (PL::F + 1 + PL::R1 * PL::LPF + (size_t)4 * PL::FPW * PL::FS) +
                        sizeof(float) * (PL::SIZE + (size_t)4 * 2 * plen_pad);
    if (smem > 160 * 1024) return PTMI_E_UNSUPPORTED;
    if (smem > kMaxSmem &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(istft_kernel<PL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return PTMI_E_UNSUPPORTED;
    const int per_cu = blocks_per_cu<istft_kernel<PL>>(256, smem);
    const long long resident_waves = 256LL * per_cu * 4;
    // a run owns `run_hops` hops; it walks halo frames before them (their tails reach into the run)
    // and is done when its last owned hop is final (frames past the row's end read as zeros, which
    // flushes the tail).  Long runs amortise the halo; short ones are chosen when the whole call
    // would otherwise not fill the chip.
    const long long total = A.cut_left + A.out_samples;                 // OLA samples needed
    const int min_hops = std::max(2 * PL::FPW, 2 * A.halo);
    // the run length with the smallest (rounds of the resident wavefronts) x (frames a run walks = its hops + the halo): round 6 -
    // until then 64 hops unless the call did not fill the chip; 1536 x 64000 samples took 4 rounds of 67 frames where 1 round of
    // 259 does (-3 %)
    int run_hops = min_hops;
    long long best = -1;
    for (int rh = min_hops; rh <= 1024; rh *= 2) {
        const long long chunks = (total + (long long)rh * shift - 1) / ((long long)rh * shift);
        const long long rounds = (batch * chunks + resident_waves - 1) / resident_waves;
        const long long cost = rounds * (rh + A.halo);
        if (best < 0 || cost < best) {
            best = cost;
            run_hops = rh;
        }
        if (chunks == 1) break;
    }
    A.run_hops = (run_hops + PL::FPW - 1) / PL::FPW * PL::FPW;
    A.groups = (A.halo + A.run_hops + PL::FPW - 1) / PL::FPW;
    const char* dbg_env = getenv("PTMI_STFT_DBG");
    A.dbg = dbg_env ? atoi(dbg_env) : 0;
    A.batch = batch;
    A.nchunks = (int)((total + (long long)A.run_hops * shift - 1) / ((long long)A.run_hops * shift));
    const long long items = batch * A.nchunks;
    if (items <= 0) return PTMI_OK;
    if (items > 0x7fffffffLL) return PTMI_E_UNSUPPORTED;
    const long long blocks = std::min((items + 3) / 4, 256LL * per_cu);
    hipLaunchKernelGGL(istft_kernel<PL>, dim3((unsigned)blocks), dim3(256), smem, st, A);
    return launch_status();
}

static int dispatch_inv(InvArgs& A, long long batch, hipStream_t st) {
    if (A.g.L > A.g.size) return PTMI_E_INVALID;
    switch (A.g.size) {
        case 64: return launch_inv<Plan<4, 8>>(A, batch, st);
        case 128: return launch_inv<Plan<8, 8>>(A, batch, st);
        case 256: return launch_inv<Plan<8, 16>>(A, batch, st);
        case 512: return launch_inv<Plan<16, 16>>(A, batch, st);
        case 1024: return launch_inv<Plan<16, 32>>(A, batch, st);
        case 2048: return launch_inv<Plan<32, 32>>(A, batch, st);
        default: return PTMI_E_UNSUPPORTED;
    }
}

// Mel filterbank (+ log) of a given spectrogram [N, F] -> [N, M] (MelTransform.forward,
// contrib/je/modules/features.py:297-330): 8 rows per workgroup staged in LDS, one thread per output.
struct MelArgs {
    const float* spec;
    float* out;
    const int32_t* lo;
    const int32_t* cnt;
    const int32_t* off;
    const float* w;
    long long N;
    int F, M, nnz, log_;
    float eps;
};

__global__ __launch_bounds__(256) void mel_apply_kernel(const MelArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = 8;
    const int PS = ((A.F + 3) & ~3) + 8;                     // padded row: the 8-bin groups may reach past F
    float* rows = reinterpret_cast<float*>(smem);            // [R][PS]
    float* mw = rows + R * PS;
    int32_t* mlo = reinterpret_cast<int32_t*>(mw + A.nnz);
    int32_t* mcnt = mlo + A.M;
    int32_t* moff = mcnt + A.M;
    for (int i = threadIdx.x; i < A.M; i += 256) {
        mlo[i] = A.lo[i];
        mcnt[i] = A.cnt[i];
        moff[i] = A.off[i];
    }
    for (int i = threadIdx.x; i < A.nnz; i += 256) mw[i] = A.w[i];
    for (int i = threadIdx.x; i < R * (PS - A.F); i += 256) rows[(i / (PS - A.F)) * PS + A.F + i % (PS - A.F)] = 0.f;
    const float4* W4 = reinterpret_cast<const float4*>(mw);
    for (long long n0 = (long long)blockIdx.x * R; n0 < A.N; n0 += (long long)gridDim.x * R) {
        const int nr = (int)min((long long)R, A.N - n0);
        __syncthreads();
        for (int i = threadIdx.x; i < nr * A.F; i += 256) rows[(i / A.F) * PS + i % A.F] = A.spec[n0 * A.F + i];
        __syncthreads();
        for (int o = threadIdx.x; o < nr * A.M; o += 256) {
            const int r = o / A.M, m = o - r * A.M;
            const float4* pw = reinterpret_cast<const float4*>(rows + r * PS) + mlo[m];
            const float4* w = W4 + moff[m];
            float acc = 0.f;
            for (int i = 0; i < mcnt[m]; ++i) {
                const float4 p0 = pw[2 * i], p1 = pw[2 * i + 1], w0 = w[2 * i], w1 = w[2 * i + 1];
                acc = fmaf(p0.x, w0.x, acc);
                acc = fmaf(p0.y, w0.y, acc);
                acc = fmaf(p0.z, w0.z, acc);
                acc = fmaf(p0.w, w0.w, acc);
                acc = fmaf(p1.x, w1.x, acc);
                acc = fmaf(p1.y, w1.y, acc);
                acc = fmaf(p1.z, w1.z, acc);
                acc = fmaf(p1.w, w1.w, acc);
            }
            A.out[n0 * A.M + o] = A.log_ ? logf(acc + A.eps) : acc;
        }
    }
}

static bool geom_ok(const ptmi_stft_geom* g) {
    return g && g->size >= 2 && (g->size % 2 == 0) && g->shift >= 1 && g->window_length >= 1 &&
           g->pad_left >= 0 && g->pad_right >= 0;
}

}  // namespace ptmi

using namespace ptmi;

extern "C" {

int64_t ptmi_stft_num_frames(const ptmi_stft_geom* g, int64_t num_samples) {
    if (!geom_ok(g)) return PTMI_E_INVALID;
    return row_frames_of(to_geo(g), num_samples);
}

int64_t ptmi_istft_num_samples(const ptmi_stft_geom* g, int64_t num_frames) {
    if (!geom_ok(g)) return PTMI_E_INVALID;
    // (frames-1)*shift + L minus the fading cut: int(pw) left, ceil(pw) right (_stft.py:257-262)
    return (num_frames - 1) * g->shift + g->window_length - g->pad_left - g->pad_right;
}

int ptmi_stft_forward(const float* x, int64_t batch, int64_t x_row_stride, int64_t num_samples,
                      const int32_t* row_samples, const float* window, const float* twiddle,
                      const ptmi_stft_geom* g, int64_t out_frames, int32_t layout, float edge_scale,
                      float* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!geom_ok(g) || !x || !window || !twiddle || !out, PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 0 || out_frames < 0 || (layout != 0 && layout != 1), PTMI_E_INVALID);
    PTMI_RETURN_IF(g->window_length > g->size, PTMI_E_INVALID);
    PTMI_RETURN_IF(num_samples > 0x7ff00000LL || out_frames > 0x7ff00000LL / (g->size + 2), PTMI_E_UNSUPPORTED);
    if (batch == 0 || out_frames == 0) return PTMI_OK;
    FwdArgs A{};
    A.x = x;
    A.row_samples = row_samples;
    A.window = window;
    A.twiddle = reinterpret_cast<const cpx*>(twiddle);
    A.out = out;
    A.x_row_stride = x_row_stride;
    A.num_samples = num_samples;
    A.out_frames = out_frames;
    A.layout = layout;
    A.edge_scale = edge_scale;
    A.g = to_geo(g);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = dispatch_fwd(A, batch, false, st);
    if (rc != PTMI_E_UNSUPPORTED) return rc;
    const long long blocks = (long long)batch * out_frames;
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    const size_t smem = sizeof(float) * (size_t)g->window_length;
    PTMI_RETURN_IF(smem > kMaxSmem, PTMI_E_UNSUPPORTED);
    hipLaunchKernelGGL(stft_generic_kernel, dim3((unsigned)blocks), dim3(256), smem, st, A);
    return launch_status();
}

int ptmi_stft_logmel(const float* x, int64_t batch, int64_t x_row_stride, int64_t num_samples,
                     const int32_t* row_samples, const float* window, const float* twiddle,
                     const ptmi_stft_geom* g, int64_t out_frames, const int32_t* mel_lo, const int32_t* mel_cnt,
                     const int32_t* mel_off, const float* mel_w, int32_t mel_M, int32_t mel_nnz, int32_t power,
                     int32_t log_, float eps, float* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!geom_ok(g) || !x || !window || !twiddle || !out, PTMI_E_INVALID);
    PTMI_RETURN_IF(!mel_lo || !mel_cnt || !mel_off || !mel_w || mel_M < 1 || mel_nnz < 1, PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 0 || out_frames < 0 || (power != 1 && power != 2), PTMI_E_INVALID);
    PTMI_RETURN_IF(g->window_length > g->size, PTMI_E_INVALID);
    PTMI_RETURN_IF(num_samples > 0x7ff00000LL || out_frames > 0x7ff00000LL / (g->size + 2), PTMI_E_UNSUPPORTED);
    if (batch == 0 || out_frames == 0) return PTMI_OK;
    FwdArgs A{};
    A.x = x;
    A.row_samples = row_samples;
    A.window = window;
    A.twiddle = reinterpret_cast<const cpx*>(twiddle);
    A.out = out;
    A.x_row_stride = x_row_stride;
    A.num_samples = num_samples;
    A.out_frames = out_frames;
    A.layout = PTMI_LAYOUT_INTERLEAVED;
    A.edge_scale = 1.f;
    A.g = to_geo(g);
    A.mel_lo = mel_lo;
    A.mel_cnt = mel_cnt;
    A.mel_off = mel_off;
    A.mel_w = mel_w;
    A.mel_M = mel_M;
    A.mel_nnz = mel_nnz;
    A.mel_power = power;
    A.mel_log = log_;
    A.mel_eps = eps;
    return dispatch_fwd(A, batch, false, static_cast<hipStream_t>(stream));   // power-of-two sizes 64..2048
}

int ptmi_mel_apply(const float* spec, int64_t N, int32_t F, const int32_t* mel_lo, const int32_t* mel_cnt,
                   const int32_t* mel_off, const float* mel_w, int32_t mel_M, int32_t mel_nnz, int32_t log_,
                   float eps, float* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!spec || !out || !mel_lo || !mel_cnt || !mel_off || !mel_w, PTMI_E_INVALID);
    PTMI_RETURN_IF(N < 0 || F < 1 || mel_M < 1 || mel_nnz < 1, PTMI_E_INVALID);
    if (N == 0) return PTMI_OK;
    const size_t smem = sizeof(float) * (8 * ((((size_t)F + 3) & ~(size_t)3) + 8) + mel_nnz) + sizeof(int32_t) * 3 * (size_t)mel_M;
    PTMI_RETURN_IF(smem > kMaxSmem, PTMI_E_UNSUPPORTED);
    MelArgs A{spec, out, mel_lo, mel_cnt, mel_off, mel_w, N, F, mel_M, mel_nnz, log_, eps};
    const long long blocks = std::min<long long>((N + 7) / 8, 256LL * 8);
    hipLaunchKernelGGL(mel_apply_kernel, dim3((unsigned)blocks), dim3(256), smem, static_cast<hipStream_t>(stream), A);
    return launch_status();
}

int ptmi_istft_forward(const float* spec, int64_t batch, int64_t num_frames, const int32_t* row_frames,
                       const float* syn_window, const float* twiddle, const ptmi_stft_geom* g,
                       int32_t layout, float edge_scale, int64_t cut_left, int64_t out_samples,
                       int64_t out_row_stride, float* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!geom_ok(g) || !spec || !syn_window || !twiddle || !out, PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 0 || num_frames < 0 || out_samples < 0 || (layout != 0 && layout != 1),
                   PTMI_E_INVALID);
    PTMI_RETURN_IF(g->window_length > g->size, PTMI_E_INVALID);
    if (batch == 0 || out_samples == 0) return PTMI_OK;
    InvArgs A{};
    A.spec = spec;
    A.row_frames = row_frames;
    A.syn_window = syn_window;
    A.twiddle = reinterpret_cast<const cpx*>(twiddle);
    A.out = out;
    A.num_frames = num_frames;
    A.out_samples = out_samples;
    A.out_row_stride = out_row_stride;
    A.cut_left = cut_left;
    A.layout = layout;
    A.edge_scale = edge_scale;
    A.g = to_geo(g);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = dispatch_inv(A, batch, st);
    if (rc != PTMI_E_UNSUPPORTED) return rc;
    const long long blocks = (long long)batch * ((out_samples + 255) / 256);
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    hipLaunchKernelGGL(istft_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A);
    return launch_status();
}

int ptmi_pit_features(const float* y, const float* s, int64_t batch, int32_t K, int64_t row_stride,
                      int64_t num_samples, const int32_t* row_samples, const float* window,
                      const float* twiddle, const ptmi_stft_geom* g, int64_t out_frames, float* Y_abs,
                      float* X_abs, float* cos_pd, ptmi_stream_t stream) {
    return ptmi_pit_features_packed(y, s, batch, K, row_stride, num_samples, row_samples, window, twiddle, g, out_frames, Y_abs, X_abs,
                                    cos_pd, nullptr, nullptr, nullptr, stream);
}

int ptmi_pit_features_packed(const float* y, const float* s, int64_t batch, int32_t K, int64_t row_stride,
                             int64_t num_samples, const int32_t* row_samples, const float* window,
                             const float* twiddle, const ptmi_stft_geom* g, int64_t out_frames, float* Y_abs,
                             float* X_abs, float* cos_pd, float* log1p_packed, uint16_t* log1p_planes,
                             const int64_t* packed_offsets, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!geom_ok(g) || !y || !window || !twiddle || !Y_abs, PTMI_E_INVALID);
    PTMI_RETURN_IF(log1p_planes && !log1p_packed, PTMI_E_INVALID);
    PTMI_RETURN_IF(log1p_planes && (reinterpret_cast<uintptr_t>(log1p_planes) & 15) != 0, PTMI_E_INVALID);
    PTMI_RETURN_IF(s && (!X_abs || !cos_pd || K < 1), PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 0 || out_frames < 0, PTMI_E_INVALID);
    PTMI_RETURN_IF(num_samples > 0x7ff00000LL || out_frames > 0x7ff00000LL / (g->size + 2), PTMI_E_UNSUPPORTED);
    if (batch == 0 || out_frames == 0) return PTMI_OK;
    FwdArgs A{};
    A.x = y;
    A.s = s;
    A.K = K;
    A.row_samples = row_samples;
    A.window = window;
    A.twiddle = reinterpret_cast<const cpx*>(twiddle);
    A.out = Y_abs;
    A.X_abs = X_abs;
    A.cos_pd = cos_pd;
    A.x_row_stride = row_stride;
    A.num_samples = num_samples;
    A.out_frames = out_frames;
    A.edge_scale = 1.f;
    A.g = to_geo(g);
    A.lp_out = log1p_packed;
    A.lp_planes = reinterpret_cast<_Float16*>(log1p_planes);
    A.lp_offs = reinterpret_cast<const long long*>(packed_offsets);
    A.lp_kb = (g->size / 2 + 1 + 31) / 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = dispatch_fwd(A, batch, true, st);
    if (rc != PTMI_E_UNSUPPORTED) return rc;
    // any other even size (paderbox.stft takes any: pit/data.py:52-53): direct DFT, one workgroup per frame
    A.batch = batch;
    const long long blocks = (long long)batch * out_frames;
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    const size_t smem = sizeof(cpx) * (size_t)(g->size / 2 + 1) + sizeof(float) * (size_t)g->window_length;
    PTMI_RETURN_IF(smem > kMaxSmem, PTMI_E_UNSUPPORTED);
    hipLaunchKernelGGL(pit_features_generic_kernel, dim3((unsigned)blocks), dim3(256), smem, st, A);
    return launch_status();
}

}  // extern "C"
