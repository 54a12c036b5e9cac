// wgrad.hip -- weight gradients of the dense / first / depthwise convs and the depthwise input gradient.
//
// Dense weight gradient dW[co][ci][tap] = sum_rows dy[row][co] * x[in(row, tap)][ci] is a GEMM whose contraction runs
// over the ROWS of two channels-last tensors.  v_mfma_f32_32x32x2_f32 wants, per lane, ONE A element (row m = lane & 31,
// k = lane >> 5) and ONE B element (column n = lane & 31, same k): with k = activation row and m / n = channel, a lane's
// operand is a single element of a channels-last row and a wave's load is two coalesced 128-byte row segments -- no
// transposition through LDS.  bf16 activations are widened on load; accumulation and the gradient are fp32.
// Row chunks are spread over the grid (split-K) and combined with fp32 atomics into the zero-initialised gradient.
//
// The input gradients of the dense convs need no kernel of their own: a 1x1x1 conv's dgrad is pasn_conv3d_fwd with the
// transposed weight (strided ones followed by pasn_scatter_strided).
#include "common.h"

namespace pasn {

constexpr int WG_U = 8;  // row pairs in flight per wave

template <typename T>
__device__ __forceinline__ float ld_f(const T* p) {
    return (float)*p;
}

// x: [N][Ti][Hi][Wi][Cin_p], dy: [N][To][Ho][Wo][Cout_p], dw: fp32 [Cout][Cin][taps]
template <typename T, bool PW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                         pasn_conv_desc d, int ci_tiles, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, k = lane >> 5;
    const int taps = d.kt * d.kh * d.kw;
    int tile = blockIdx.y;
    const int tap = tile % taps;
    tile /= taps;
    const int ci_t = tile % ci_tiles, co_t = tile / ci_tiles;
    const int co = co_t * 32 + m, ci = ci_t * 32 + m;
    const bool a_ok = co < d.Cout_p, b_ok = ci < d.Cin_p;
    const int coc = a_ok ? co : 0, cic = b_ok ? ci : 0;
    const long R = (long)d.N * d.To * d.Ho * d.Wo;
    const long r0 = ((long)blockIdx.x * 4 + wave) * rows_per_wave;
    const long r1 = min(R, r0 + rows_per_wave);
    const int tt = tap / (d.kh * d.kw), th = (tap / d.kw) % d.kh, tw = tap % d.kw;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (long rb = r0; rb < r1; rb += 2 * WG_U) {
        T ra[WG_U], rx[WG_U];
        float ma[WG_U], mx[WG_U];
#pragma unroll
        for (int u = 0; u < WG_U; ++u) {
            const long r = rb + 2 * u + k;
            const bool rok = r < r1;
            const long rc = rok ? r : r0;
            long in_row = rc;
            bool vok = rok;
            if (!PW) {
                const int wo = (int)(rc % d.Wo);
                long q = rc / d.Wo;
                const int ho = (int)(q % d.Ho);
                q /= d.Ho;
                const int to = (int)(q % d.To), n = (int)(q / d.To);
                const int ti = to * d.st - d.pt + tt, hi = ho * d.sh - d.ph + th, wi = wo * d.sw - d.pw + tw;
                const bool in = ti >= 0 && ti < d.Ti && hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi;
                vok = rok && in;
                in_row = in ? (((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi : 0;
            }
            ra[u] = dy[rc * d.Cout_p + coc];
            rx[u] = x[in_row * d.Cin_p + cic];
            ma[u] = (rok && a_ok) ? 1.0f : 0.0f;
            mx[u] = (vok && b_ok) ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < WG_U; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ra[u] * ma[u], (float)rx[u] * mx[u], acc, 0, 0, 0);
    }
    // acc element `reg` of this lane: row (= co offset) acc_row(reg, k), column (= ci offset) m
    const int cig = ci_t * 32 + m;
    if (cig < d.Cin) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int cog = co_t * 32 + acc_row(reg, k);
            if (cog < d.Cout) unsafeAtomicAdd(dw + ((size_t)cog * d.Cin + cig) * taps + tap, acc[reg]);
        }
    }
}

// First conv (planar input x [N][3][T][Hi][Wi], window (1,kh,kw)): dw fp32 [Cout][3*kh*kw]
template <typename TIN, typename T>
__global__ __launch_bounds__(256) void first_conv_wgrad_kernel(const TIN* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                               pasn_conv_desc d, int col_tiles, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, k = lane >> 5;
    const int cols = 3 * d.kh * d.kw;
    const int col_t = blockIdx.y % col_tiles, co_t = blockIdx.y / col_tiles;
    const int co = co_t * 32 + m, col = col_t * 32 + m;
    const bool a_ok = co < d.Cout_p, b_ok = col < cols;
    const int coc = a_ok ? co : 0;
    const int ci = b_ok ? col / (d.kh * d.kw) : 0, th = b_ok ? (col / d.kw) % d.kh : 0, tw = b_ok ? col % d.kw : 0;
    const long R = (long)d.N * d.To * d.Ho * d.Wo;
    const long r0 = ((long)blockIdx.x * 4 + wave) * rows_per_wave;
    const long r1 = min(R, r0 + rows_per_wave);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (long rb = r0; rb < r1; rb += 2 * WG_U) {
        T ra[WG_U];
        TIN rx[WG_U];
        float ma[WG_U], mx[WG_U];
#pragma unroll
        for (int u = 0; u < WG_U; ++u) {
            const long r = rb + 2 * u + k;
            const bool rok = r < r1;
            const long rc = rok ? r : r0;
            const int wo = (int)(rc % d.Wo);
            long q = rc / d.Wo;
            const int ho = (int)(q % d.Ho);
            q /= d.Ho;
            const int to = (int)(q % d.To), n = (int)(q / d.To);
            const int hi = ho * d.sh - d.ph + th, wi = wo * d.sw - d.pw + tw;
            const bool in = hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi;
            const long off = in ? ((((long)n * 3 + ci) * d.Ti + to) * d.Hi + hi) * d.Wi + wi : 0;
            ra[u] = dy[rc * d.Cout_p + coc];
            rx[u] = x[off];
            ma[u] = (rok && a_ok) ? 1.0f : 0.0f;
            mx[u] = (rok && in && b_ok) ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < WG_U; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ra[u] * ma[u], (float)rx[u] * mx[u], acc, 0, 0, 0);
    }
    if (col < cols) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int cog = co_t * 32 + acc_row(reg, k);
            if (cog < d.Cout) unsafeAtomicAdd(dw + (size_t)cog * cols + col, acc[reg]);
        }
    }
}

// ---- depthwise input gradient (any window / stride): dx[n,ti,hi,wi,c] = sum_taps dy[n,to,ho,wo,c] * w[tap][c] -----------
template <typename T>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                                       pasn_conv_desc d) {
    const int CG = d.Cin_p / 8;
    const size_t total = (size_t)d.N * d.Ti * d.Hi * d.Wi * CG;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        const size_t row = i / CG;
        const int wi = (int)(row % d.Wi);
        size_t q = row / d.Wi;
        const int hi = (int)(q % d.Hi);
        q /= d.Hi;
        const int ti = (int)(q % d.Ti), n = (int)(q / d.Ti);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        for (int a = 0; a < d.kt; ++a) {
            const int tn = ti + d.pt - a;
            if (tn < 0 || tn % d.st) continue;
            const int to = tn / d.st;
            if (to >= d.To) continue;
            for (int b = 0; b < d.kh; ++b) {
                const int hn = hi + d.ph - b;
                if (hn < 0 || hn % d.sh) continue;
                const int ho = hn / d.sh;
                if (ho >= d.Ho) continue;
                for (int c = 0; c < d.kw; ++c) {
                    const int wn = wi + d.pw - c;
                    if (wn < 0 || wn % d.sw) continue;
                    const int wo = wn / d.sw;
                    if (wo >= d.Wo) continue;
                    float g[8], wv[8];
                    load8(dy + ((((size_t)n * d.To + to) * d.Ho + ho) * d.Wo + wo) * d.Cout_p + cg * 8, g);
                    load8(w + (size_t)((a * d.kh + b) * d.kw + c) * d.Cout_p + cg * 8, wv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(g[j], wv[j], acc[j]);
                }
            }
        }
        store8(dx + row * d.Cin_p + cg * 8, acc);
    }
}

// ---- depthwise weight gradient: partial[chunk][tap][Cp] over output-row chunks, one temporal tap plane per blockIdx.z ----
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                               pasn_conv_desc d, int CG, int CGb, long rows_per_chunk) {
    __shared__ float red[256 * 8];
    const int cg = threadIdx.x % CGb, rl = threadIdx.x / CGb, RL = 256 / CGb;
    const int a = blockIdx.z;  // temporal tap
    const int KP = d.kh * d.kw;  // <= 9
    float acc[9][8];
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[p][j] = 0.0f;
    const long R = (long)d.N * d.To * d.Ho * d.Wo;
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    if (cg < CG) {
        for (long r = r0 + rl; r < r1; r += RL) {
            const int wo = (int)(r % d.Wo);
            long q = r / d.Wo;
            const int ho = (int)(q % d.Ho);
            q /= d.Ho;
            const int to = (int)(q % d.To), n = (int)(q / d.To);
            const int ti = to * d.st - d.pt + a;
            if (ti < 0 || ti >= d.Ti) continue;
            float g[8];
            load8(dy + r * d.Cout_p + cg * 8, g);
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                if (p < KP) {
                    const int hi = ho * d.sh - d.ph + p / d.kw, wi = wo * d.sw - d.pw + p % d.kw;
                    if (hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi) {
                        float v[8];
                        load8(x + ((((size_t)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.Cin_p + cg * 8, v);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[p][j] = fmaf(g[j], v[j], acc[p][j]);
                    }
                }
            }
        }
    }
    const int taps = d.kt * KP;
    float* out = partial + ((size_t)blockIdx.x * taps + (size_t)a * KP) * d.Cout_p;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        if (p < KP) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[p][j];
            __syncthreads();
            for (int t = threadIdx.x; t < CGb * 8; t += 256) {
                const int g2 = t >> 3, j = t & 7;
                if (g2 < CG) {
                    float s = 0.0f;
                    for (int q = 0; q < RL; ++q) s += red[(q * CGb + g2) * 8 + j];
                    out[(size_t)p * d.Cout_p + g2 * 8 + j] = s;
                }
            }
        }
    }
}

__global__ void dw_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int chunks, int taps, int C, int Cp);
bool pw_wgrad_bf16(const void* x, const void* dy, float* dw, const pasn_conv_desc& d, hipStream_t s);
size_t dw_wgrad_strip_floats(const pasn_conv_desc& d);
bool dw_wgrad_strip(const void* x, const void* dy, float* ws, float* dw, const pasn_conv_desc& d, int dtype, hipStream_t s);

static int wgrad_rows_per_wave(long R, int tiles) {
    long waves_per_tile = std::max<long>(1, std::min<long>(std::min(8192, 2048 + 65536 / std::max(1, tiles)) / std::max(1, tiles), R / 128));
    waves_per_tile = (waves_per_tile + 3) / 4 * 4;
    long rpw = (R + waves_per_tile - 1) / waves_per_tile;
    rpw = (rpw + 2 * WG_U - 1) / (2 * WG_U) * (2 * WG_U);
    return (int)rpw;
}

static long dw_wgrad_rows_per_chunk(const pasn_conv_desc& d) {
    const long R = (long)d.N * d.To * d.Ho * d.Wo;
    int CG = d.Cout_p / 8, b = 1;
    while (b < CG) b <<= 1;
    const int RL = 256 / b;
    const long chunks = std::max<long>(1, std::min<long>(2048, R / ((long)RL * 8)));
    return (R + chunks - 1) / chunks;
}

}  // namespace pasn

using namespace pasn;

extern "C" size_t pasn_conv3d_wgrad_workspace_bytes(const pasn_conv_desc* d, int dtype) {
    return d ? wgrad_halo_workspace_bytes(*d, dtype) : 0;
}

extern "C" int pasn_conv3d_wgrad_ws(const void* x, const void* dy, float* dw, const pasn_conv_desc* d, int dtype, void* ws, void* stream) {
    PASN_REQUIRE(x && dy && dw && d, "null pointer");
    PASN_REQUIRE(d->Cin_p % 8 == 0 && d->Cout_p % 8 == 0 && d->Cin <= d->Cin_p && d->Cout <= d->Cout_p, "bad channel extents");
    if (ws && wgrad_halo(x, dy, dw, ws, *d, dtype, (hipStream_t)stream)) return check_launch("conv3d_wgrad_halo");
    return pasn_conv3d_wgrad(x, dy, dw, d, dtype, stream);
}

extern "C" int pasn_conv3d_wgrad(const void* x, const void* dy, float* dw, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(x && dy && dw && d, "null pointer");
    PASN_REQUIRE(d->Cin_p % 8 == 0 && d->Cout_p % 8 == 0 && d->Cin <= d->Cin_p && d->Cout <= d->Cout_p, "bad channel extents");
    const int taps = d->kt * d->kh * d->kw;
    if (dtype == PASN_BF16 && !tune("PASN_NO_WGRAD_LDS") && pw_wgrad_bf16(x, dy, dw, *d, (hipStream_t)stream))
        return check_launch("conv3d_wgrad");
    const int co_tiles = ceil_div(d->Cout, 32), ci_tiles = ceil_div(d->Cin, 32);
    const long tiles = (long)co_tiles * ci_tiles * taps;
    PASN_REQUIRE(tiles <= 65535, "too many weight tiles for one launch");
    const long R = (long)d->N * d->To * d->Ho * d->Wo;
    const int rpw = wgrad_rows_per_wave(R, (int)tiles);
    const dim3 grid(ceil_div(R, (long)rpw * 4), (unsigned)tiles);
    const bool pw = taps == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0;
    hipStream_t s = (hipStream_t)stream;
#define WG(T, P) hipLaunchKernelGGL((conv_wgrad_kernel<T, P>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, dw, *d, ci_tiles, rpw)
    if (dtype == PASN_BF16) {
        if (pw) WG(__bf16, true);
        else WG(__bf16, false);
    } else {
        if (pw) WG(float, true);
        else WG(float, false);
    }
#undef WG
    return check_launch("conv3d_wgrad");
}

// im2col of the planar clip for the first conv's weight gradient: X[row][col] (bf16, 32-column groups), col = (ci, r, s); then the
// gradient is the pointwise GEMM dW[co][col] = sum_rows dy[row][co] X[row][col] on the LDS-transposed bf16 MFMA kernel.  The
// gather costs one pass over the clip's windows instead of one scattered 2-byte load per MFMA operand element.
template <typename TIN>
__global__ __launch_bounds__(256) void first_conv_im2col_kernel(const TIN* __restrict__ x, __bf16* __restrict__ X, pasn_conv_desc d, int colp) {
    const int groups = colp / 8, cols = 3 * d.kh * d.kw;
    const long R = (long)d.N * d.To * d.Ho * d.Wo, total = R * groups;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int g = (int)(idx % groups);
        const long row = idx / groups;
        const int wo = (int)(row % d.Wo);
        long q = row / d.Wo;
        const int ho = (int)(q % d.Ho);
        q /= d.Ho;
        const int to = (int)(q % d.To), n = (int)(q / d.To);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = g * 8 + j;
            const int ci = col / (d.kh * d.kw), r = (col / d.kw) % d.kh, sx = col % d.kw;
            const int hi = ho * d.sh - d.ph + r, wi = wo * d.sw - d.pw + sx;
            const bool ok = col < cols && hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi;
            v[j] = ok ? (float)x[((((long)n * 3 + ci) * d.Ti + to) * d.Hi + hi) * d.Wi + wi] : 0.0f;
        }
        store8(X + row * colp + g * 8, v);
    }
}

extern "C" size_t pasn_first_conv_wgrad_workspace_bytes(const pasn_conv_desc* d, int dtype) {
    if (!d || dtype != PASN_BF16 || tune("PASN_NO_FIRST_IM2COL")) return 0;
    const int colp = (3 * d->kh * d->kw + 7) / 8 * 8;
    if (colp > 512) return 0;
    return (size_t)d->N * d->To * d->Ho * d->Wo * colp * sizeof(__bf16);
}

extern "C" int pasn_first_conv_wgrad(const void* x, const void* dy, float* dw, const pasn_conv_desc* d, int in_dtype, int dtype,
                                     void* ws, void* stream) {
    PASN_REQUIRE(x && dy && dw && d, "null pointer");
    PASN_REQUIRE(d->kt == 1 && d->st == 1 && d->pt == 0 && d->Cin == 3, "first conv is (1,kh,kw) over 3 planar channels");
    const int cols = 3 * d->kh * d->kw;
    hipStream_t s = (hipStream_t)stream;
    if (ws && pasn_first_conv_wgrad_workspace_bytes(d, dtype)) {
        const int colp = (cols + 7) / 8 * 8;
        const long R = (long)d->N * d->To * d->Ho * d->Wo, total = R * (colp / 8);
        const int nb = (int)std::min<long>((total + 255) / 256, 1 << 20);
        if (in_dtype == PASN_BF16)
            hipLaunchKernelGGL(first_conv_im2col_kernel<__bf16>, dim3(nb), dim3(256), 0, s, (const __bf16*)x, (__bf16*)ws, *d, colp);
        else
            hipLaunchKernelGGL(first_conv_im2col_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, (__bf16*)ws, *d, colp);
        pasn_conv_desc g = *d;  // the equivalent pointwise problem over the im2col rows
        g.Ti = d->To; g.Hi = d->Ho; g.Wi = d->Wo;
        g.Cin = cols; g.Cin_p = colp;
        g.kt = g.kh = g.kw = 1; g.st = g.sh = g.sw = 1; g.pt = g.ph = g.pw = 0;
        if (pw_wgrad_bf16(ws, dy, dw, g, s)) return check_launch("first_conv_wgrad");
    }
    const int co_tiles = ceil_div(d->Cout, 32), col_tiles = ceil_div(cols, 32);
    const long R = (long)d->N * d->To * d->Ho * d->Wo;
    const int rpw = wgrad_rows_per_wave(R, co_tiles * col_tiles);
    const dim3 grid(ceil_div(R, (long)rpw * 4), co_tiles * col_tiles);
#define FW(TI, T) hipLaunchKernelGGL((first_conv_wgrad_kernel<TI, T>), grid, dim3(256), 0, s, (const TI*)x, (const T*)dy, dw, *d, col_tiles, rpw)
    if (in_dtype == PASN_BF16 && dtype == PASN_BF16) FW(__bf16, __bf16);
    else if (in_dtype == PASN_F32 && dtype == PASN_BF16) FW(float, __bf16);
    else if (in_dtype == PASN_BF16) FW(__bf16, float);
    else FW(float, float);
#undef FW
    return check_launch("first_conv_wgrad");
}

// 3x3x3, stride (1,2,2), pad 1 (the first block of every X3D stage): a thread owns a 2x2 input patch.  Even rows / columns
// see only the centre tap, odd ones the two outer taps, so the four pixels need dy[to][i..i+1][j..j+1] for the three temporal
// taps -- 12 loads and 27 FMAs per channel for 4 outputs, no divergent tap loop.
template <int CH, typename T>
__device__ __forceinline__ void dg_load(const T* p, float (&v)[CH]) {
    if constexpr (CH == 8) load8(p, v);
    else load4(p, v);
}
template <int CH, typename T>
__device__ __forceinline__ void dg_store(T* p, const float (&v)[CH]) {
    if constexpr (CH == 8) store8(p, v);
    else store4(p, v);
}

template <typename T, int CH>  // CH channels per thread: 4 keeps the patch + taps + gradients at ~130 registers (8: 330, one wave per SIMD)
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                                          pasn_conv_desc d) {
    // the 27 x Cp taps in LDS, staged once per block (the first version read its 27 weight vectors per patch from global memory: 864
    // bytes of weights for 192 bytes of gradients per thread)
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [27][Cp]
    for (int i = threadIdx.x * 4; i < 27 * d.Cout_p; i += 256 * 4) *reinterpret_cast<f32x4*>(wl + i) = *reinterpret_cast<const f32x4*>(w + i);
    __syncthreads();
    const int CG = d.Cin_p / CH, Hh = (d.Hi + 1) / 2, Wh = (d.Wi + 1) / 2;
    const size_t total = (size_t)d.N * d.Ti * Hh * Wh * CG;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int cg = (int)(idx % CG);
        size_t q = idx / CG;
        const int j = (int)(q % Wh);
        q /= Wh;
        const int i = (int)(q % Hh);
        q /= Hh;
        const int ti = (int)(q % d.Ti), n = (int)(q / d.Ti);
        float o[2][2][CH];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < CH; ++e) o[a][b][e] = 0.0f;
        // all twelve gradient loads first (clamped addresses), masks afterwards
        float g[3][2][2][CH];
        unsigned okbits = 0;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int to = ti + 1 - kt;
            const bool tok = to >= 0 && to < d.To;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int ho = i + a, wo = j + b;
                    const bool ok = tok && ho < d.Ho && wo < d.Wo;
                    dg_load<CH>(dy + ((((size_t)n * d.To + (tok ? to : 0)) * d.Ho + (ok ? ho : 0)) * d.Wo + (ok ? wo : 0)) * d.Cout_p + cg * CH, g[kt][a][b]);
                    okbits |= (ok ? 1u : 0u) << (kt * 4 + a * 2 + b);
                }
        }
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if (!((okbits >> (kt * 4 + a * 2 + b)) & 1u)) {
#pragma unroll
                        for (int e = 0; e < CH; ++e) g[kt][a][b][e] = 0.0f;
                    }
            const float* wk = wl + kt * 9 * d.Cout_p + cg * CH;
            float wv[9][CH];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) dg_load<CH>(wk + tp * d.Cout_p, wv[tp]);
            // input (2i+a, 2j+b) <- output (ho, wo) through tap (kh, kw) with 2*ho - 1 + kh = 2i + a
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                o[0][0][e] = fmaf(g[kt][0][0][e], wv[4][e], o[0][0][e]);                                                      // (1,1)
                o[0][1][e] = fmaf(g[kt][0][0][e], wv[5][e], fmaf(g[kt][0][1][e], wv[3][e], o[0][1][e]));                      // (1,2) from j, (1,0) from j+1
                o[1][0][e] = fmaf(g[kt][0][0][e], wv[7][e], fmaf(g[kt][1][0][e], wv[1][e], o[1][0][e]));                      // (2,1) from i, (0,1) from i+1
                o[1][1][e] = fmaf(g[kt][0][0][e], wv[8][e], fmaf(g[kt][0][1][e], wv[6][e],
                             fmaf(g[kt][1][0][e], wv[2][e], fmaf(g[kt][1][1][e], wv[0][e], o[1][1][e]))));
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int hi = 2 * i + a, wi = 2 * j + b;
                if (hi < d.Hi && wi < d.Wi) dg_store<CH>(dx + ((((size_t)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.Cin_p + cg * CH, o[a][b]);
            }
    }
}

extern "C" int pasn_dwconv3d_dgrad(const void* dy, const float* w, void* dx, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(dy && w && dx && d, "null pointer");
    PASN_REQUIRE(d->Cin_p == d->Cout_p && d->Cin_p % 8 == 0, "depthwise conv keeps the channel stride");
    if (d->kt == 3 && d->kh == 3 && d->kw == 3 && d->st == 1 && d->sh == 2 && d->sw == 2 && d->pt == 1 && d->ph == 1 && d->pw == 1 &&
        d->Cout_p <= 512 && !tune("PASN_NO_DGRAD_S2")) {
        const size_t items = (size_t)d->N * d->Ti * ((d->Hi + 1) / 2) * ((d->Wi + 1) / 2) * (d->Cin_p / 4);
        const int nb = (int)std::min<size_t>((items + 255) / 256, 4096);  // grid-stride: the weight staging amortises over many patches
        const size_t wlds = (size_t)27 * d->Cout_p * sizeof(float);  // <= 55 KB (Cout_p <= 512 checked above)
        if (dtype == PASN_BF16)
            hipLaunchKernelGGL((dw_dgrad_s2_kernel<__bf16, 4>), dim3(nb), dim3(256), wlds, (hipStream_t)stream, (const __bf16*)dy, w, (__bf16*)dx, *d);
        else
            hipLaunchKernelGGL((dw_dgrad_s2_kernel<float, 4>), dim3(nb), dim3(256), wlds, (hipStream_t)stream, (const float*)dy, w, (float*)dx, *d);
        return check_launch("dwconv3d_dgrad");
    }
    const size_t total = (size_t)d->N * d->Ti * d->Hi * d->Wi * (d->Cin_p / 8);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 1 << 20);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16) hipLaunchKernelGGL(dw_dgrad_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)dy, w, (__bf16*)dx, *d);
    else hipLaunchKernelGGL(dw_dgrad_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)dy, w, (float*)dx, *d);
    return check_launch("dwconv3d_dgrad");
}

extern "C" size_t pasn_dwconv3d_wgrad_workspace_floats(const pasn_conv_desc* d) {
    if (!d || d->Cout_p <= 0 || d->Cout_p % 8 || d->Cout_p > 2048) return 0;
    const size_t fast = tune("PASN_NO_DWWG_STRIP") ? 0 : dw_wgrad_strip_floats(*d);
    if (fast) return fast;
    const long R = (long)d->N * d->To * d->Ho * d->Wo;
    const long rpc = dw_wgrad_rows_per_chunk(*d);
    const long chunks = (R + rpc - 1) / rpc;
    return (size_t)chunks * d->kt * d->kh * d->kw * d->Cout_p;
}

extern "C" int pasn_dwconv3d_wgrad(const void* x, const void* dy, float* ws, float* dw, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(x && dy && ws && dw && d, "null pointer");
    PASN_REQUIRE(d->Cin_p == d->Cout_p && d->Cin_p % 8 == 0 && d->Cout_p <= 2048, "depthwise conv keeps the channel stride (<= 2048)");
    PASN_REQUIRE(d->kh * d->kw <= 9, "spatial window above 3x3 is not covered");
    if (!tune("PASN_NO_DWWG_STRIP") && dw_wgrad_strip(x, dy, ws, dw, *d, dtype, (hipStream_t)stream)) return check_launch("dwconv3d_wgrad");
    const long R = (long)d->N * d->To * d->Ho * d->Wo;
    const long rpc = dw_wgrad_rows_per_chunk(*d);
    const int chunks = (int)((R + rpc - 1) / rpc);
    int CG = d->Cout_p / 8, CGb = 1;
    while (CGb < CG) CGb <<= 1;
    const int taps = d->kt * d->kh * d->kw;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(chunks, 1, d->kt);
    if (dtype == PASN_BF16)
        hipLaunchKernelGGL(dw_wgrad_partial_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)x, (const __bf16*)dy, ws, *d, CG, CGb, rpc);
    else
        hipLaunchKernelGGL(dw_wgrad_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dy, ws, *d, CG, CGb, rpc);
    hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(ceil_div((long)taps * d->Cout_p, 64)), dim3(256), 0, s, ws, dw, chunks, taps, d->Cout,
                       d->Cout_p);
    return check_launch("dwconv3d_wgrad");
}

// =====================================================================================================================
// bf16 fast paths
// =====================================================================================================================
namespace pasn {

// ---- pointwise (1x1x1, any stride) weight gradient on v_mfma_f32_32x32x16_bf16 ------------------------------------------
// The contraction index (the activation ROW) is the slow index of both operands, the MFMA wants 8 consecutive k per lane.
// A thread therefore loads an 8-row x 8-channel patch (eight 16-byte loads), transposes it in registers (32 v_perm_b32) and
// stores, per channel, the 8 consecutive rows as ONE 16-byte LDS write: LDS holds At[channel][KT rows] / Bt[channel][KT rows],
// and a fragment read is a conflict-free ds_read_b128 (row pitch KT*2 + 16 bytes).  Four waves share the staged rows; each
// owns up to TPW 32x32 (co, ci) tiles.  Split-K over the grid, fp32 atomics into the zeroed gradient.
template <int KT>
struct WgLds {
    static constexpr int PITCH = KT * 2 + 16;  // bytes per channel row
    static constexpr int SLOTS = KT / 8;       // 16-byte slots (8 rows of a channel) per channel row
    static_assert((SLOTS & (SLOTS - 1)) == 0, "the slot rotation below wraps with a mask");
};
// Slot rotation (round 5).  A staging thread writes the 8 rows of channels 8 cg .. 8 cg + 7 as eight 16-byte LDS writes, and the 8 lanes a
// ds_write_b128 serves together hold 8 CONSECUTIVE channel groups of one row octet: 8 * PITCH bytes apart = a multiple of 128 bytes whatever the
// pitch -- one bank group, 8-way conflicts on every staging write (SQ counters, tools/pmc_lds_audit.sh: 24 LDS cycles per LDS instruction, 78 %
// of them conflicts, in every pointwise weight-gradient kernel).  Rotating a channel row's slots by its channel group, slot' = (slot + cg) mod
// SLOTS, spreads the 8 lanes over 8 slots; a fragment read (32 channel rows of one slot) adds the row's group the same way: 17 slots of pitch x
// channel row + rotation stays conflict-free except for one pair of lanes per group.
__device__ __forceinline__ int wg_slot(int slot, int cg, int slots) { return (slot + cg) & (slots - 1); }

__device__ __forceinline__ void transpose8x8_bf16(const uint4 (&in)[8], uint4 (&out)[8]) {
    // in[r] = 8 channels of row r (2 per dword); out[c] = 8 rows of channel c (2 per dword)
    const unsigned* I = reinterpret_cast<const unsigned*>(in);
    unsigned* O = reinterpret_cast<unsigned*>(out);
#pragma unroll
    for (int q = 0; q < 4; ++q)        // channel pair (2q, 2q+1)
#pragma unroll
        for (int p = 0; p < 4; ++p) {  // row pair (2p, 2p+1)
            const unsigned lo = I[(2 * p) * 4 + q], hi = I[(2 * p + 1) * 4 + q];
            O[(2 * q) * 4 + p] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);      // low halves  -> channel 2q
            O[(2 * q + 1) * 4 + p] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);  // high halves -> channel 2q+1
        }
}

template <int KT, int TPW>
__global__ __launch_bounds__(256) void pw_wgrad_bf16_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy, float* __restrict__ dw,
                                                            pasn_conv_desc d, int co_tiles, int ci_tiles, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int PITCH = WgLds<KT>::PITCH;
    const int CGo = d.Cout_p / 8, CGi = d.Cin_p / 8;
    unsigned char* At = lds;                                   // [co_tiles*32][PITCH]
    unsigned char* Bt = lds + (size_t)co_tiles * 32 * PITCH;   // [ci_tiles*32][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const long R = (long)d.N * d.To * d.Ho * d.Wo;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    // rows of x are gathered through the conv's window map when it is not the identity (strided 1x1x1 convs, windowed convs:
    // one tap per blockIdx.z, rows that fall into the padding read as zero)
    const int taps = d.kt * d.kh * d.kw, tap = blockIdx.z;
    const int tt = tap / (d.kh * d.kw), th = (tap / d.kw) % d.kh, tw = tap % d.kw;
    const bool strided = d.st != 1 || d.sh != 1 || d.sw != 1 || taps != 1 || d.pt || d.ph || d.pw;
    const int ntiles = co_tiles * ci_tiles;
    // this wave's tiles: (blockIdx.y * 4 + wave) * TPW + j; out-of-range ones alias tile 0 and are dropped at the end
    int t_co[TPW], t_ci[TPW];
    bool t_ok[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = (blockIdx.y * 4 + wave) * TPW + j;
        t_ok[j] = t < ntiles;
        const int tc = t_ok[j] ? t : 0;
        t_co[j] = tc / ci_tiles;
        t_ci[j] = tc % ci_tiles;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    // zero the LDS rows of padded channels once (channels >= C*_p of the last tile are never written by the staging)
    for (int i = tid * 16; i < (co_tiles + ci_tiles) * 32 * PITCH; i += 256 * 16) *reinterpret_cast<uint4*>(lds + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int units = (KT / 8) * (CGo + CGi);  // 8-row x 8-channel patches per staged K tile (at most 2 per thread, checked on the host)
    // the 16 row loads of a thread are issued as one batch (raw registers), transposed and staged afterwards
    uint4 pre[2][8];
    unsigned okbits = 0;  // validity of the 16 prefetched rows; applied when they are staged (a select right after a load would
                          // make hipcc wait for that load on the spot and serialise the batch)
    auto fetch = [&](long rb) {
        okbits = 0;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int u = tid + v * 256;
            const bool live = u < units;
            const int uc = live ? u : 0;
            const int g = uc % (CGo + CGi), r8 = uc / (CGo + CGi);
            const bool is_a = g < CGo;
            const int cg = is_a ? g : g - CGo;
            const int cp = is_a ? d.Cout_p : d.Cin_p;
            const __bf16* src = is_a ? dy : x;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long r = rb + r8 * 8 + i;
                bool ok = live && r < r1;
                long row = ok ? r : r0;
                if (!is_a && strided) {
                    const int wo = (int)(row % d.Wo);
                    long q = row / d.Wo;
                    const int ho = (int)(q % d.Ho);
                    q /= d.Ho;
                    const int to = (int)(q % d.To), n = (int)(q / d.To);
                    const int ti = to * d.st - d.pt + tt, hi = ho * d.sh - d.ph + th, wi = wo * d.sw - d.pw + tw;
                    const bool in = ti >= 0 && ti < d.Ti && hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi;
                    ok = ok && in;
                    row = in ? (((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi : 0;
                }
                pre[v][i] = *reinterpret_cast<const uint4*>(src + row * cp + cg * 8);
                okbits |= (ok ? 1u : 0u) << (v * 8 + i);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int u = tid + v * 256;
            if (u < units) {
                const int g = u % (CGo + CGi), r8 = u / (CGo + CGi);
                const bool is_a = g < CGo;
                const int cg = is_a ? g : g - CGo;
                uint4 out[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (!((okbits >> (v * 8 + i)) & 1u)) pre[v][i] = make_uint4(0, 0, 0, 0);
                transpose8x8_bf16(pre[v], out);
                unsigned char* dst = (is_a ? At : Bt) + (size_t)(cg * 8) * PITCH + wg_slot(r8, cg, WgLds<KT>::SLOTS) * 16;
#pragma unroll
                for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(dst + c * PITCH) = out[c];
            }
        }
    };
    // (Issuing the next tile's loads before this tile's MFMAs -- a register software pipeline -- was measured 5 % SLOWER over the
    // 61 layers: the kernel is bound by HBM on the large layers and by its few blocks on the small ones, not by load latency.)
    for (long rb = r0; rb < r1; rb += KT) {
        fetch(rb);
        stage();
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KT / 16; ++kk) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(At + (size_t)(t_co[j] * 32 + m) * PITCH + wg_slot(kk * 2 + h, t_co[j] * 4 + (m >> 3), WgLds<KT>::SLOTS) * 16);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bt + (size_t)(t_ci[j] * 32 + m) * PITCH + wg_slot(kk * 2 + h, t_ci[j] * 4 + (m >> 3), WgLds<KT>::SLOTS) * 16);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int cig = t_ci[j] * 32 + m;
        if (t_ok[j] && cig < d.Cin) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int cog = t_co[j] * 32 + acc_row(reg, h);
                if (cog < d.Cout) unsafeAtomicAdd(dw + ((size_t)cog * d.Cin + cig) * taps + tap, acc[j][reg]);
            }
        }
    }
}

template <int KT, int TPW>
static void launch_pw_wgrad_bf16(const void* x, const void* dy, float* dw, const pasn_conv_desc& d, int co_tiles, int ci_tiles, int gy,
                                 hipStream_t s) {
    const long R = (long)d.N * d.To * d.Ho * d.Wo;
    // split-K partitions: enough blocks to fill the chip, but every partition ends in ntiles*1024 atomics on the same addresses
    const long ntiles = (long)co_tiles * ci_tiles;
    const int taps = d.kt * d.kh * d.kw;
    const long want_blocks = std::max<long>(1, std::min<long>(2048 / ((long)gy * taps) + 1, std::max<long>(16, 3000 / ntiles)));
    long rpb = std::max<long>(KT, (R + want_blocks - 1) / want_blocks);
    rpb = (rpb + KT - 1) / KT * KT;
    const dim3 grid(ceil_div(R, rpb), gy, taps);
    const size_t lds = (size_t)(co_tiles + ci_tiles) * 32 * WgLds<KT>::PITCH;
    hipLaunchKernelGGL((pw_wgrad_bf16_kernel<KT, TPW>), grid, dim3(256), lds, s, (const __bf16*)x, (const __bf16*)dy, dw, d, co_tiles, ci_tiles,
                       (int)rpb);
}

// ---- the same for WIDE stride-1 pointwise layers (X3D stages 4-5: 216 <-> 96, 432 <-> 192 channels) -----------------------------
// pw_wgrad_bf16_kernel stages ALL channels of dy and x per 32-row step in every block and then lets blockIdx.y pick 16 of the (co, ci)
// tiles: on the 432 x 192 layers six y-blocks each load, transpose and store the same 624 channels for 8 MFMAs per wave and step, one
// exposed L2 round trip per step, 210 blocks for 256 CUs -- 71 us for 31 MB (0.44 TB/s), 69 us for 63 MB on the 216 x 96 layers; 34
// launches, 2.4 of the step's 32 ms.  Here a block owns a 2 x 2 group of tiles (one per wave) and stages ONLY the 64 + 64 channels those
// need, 128 rows per step: one 8 x 8 patch per thread and step, 8 MFMAs per wave between barriers, 35 KB of LDS (several blocks per CU,
// so one block's round trip hides under another's MFMAs), and 4 x 1024 atomics per block instead of 16 x 1024.
constexpr int WT_KT = 128, WT_PITCH = WT_KT * 2 + 16;

// COT x CIT tiles per WAVE (round 4): a block owns (2 COT) x (2 CIT) tiles.  (1, 1) stages 32 KB per 128-row step for 8 MFMAs per wave -- more than a
// CU takes from L2 in the time; (1, 2) / (2, 1) stage 48 KB for 16 (one shared fragment read per two MFMAs) and halve the groups along the wide side.
template <int COT, int CIT>
__global__ __launch_bounds__(256) void pw_wgrad_tile_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy, float* __restrict__ dw,
                                                            pasn_conv_desc d, int co_groups, int ci_groups, int rows_per_block, int gy2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int CA = 64 * COT, CB = 64 * CIT;       // channels of dy / x a block stages
    constexpr int GA = CA / 8, GB = CB / 8, NG = GA + GB;  // 8-channel groups
    constexpr int NP = (16 * NG + 255) / 256;         // 8-row x 8-channel patches per thread and step
    unsigned char* At = lds;                          // [CA co channels][WT_PITCH]
    unsigned char* Bt = lds + (size_t)CA * WT_PITCH;  // [CB ci channels][WT_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const long R = (long)d.N * d.To * d.Ho * d.Wo;
    // Block -> (row range, tile group).  The gy tile groups of one row range read the same rows (each its own channels of them): as
    // blockIdx.y of a 2-D grid they ran a whole sweep over the rows apart, and every group fetched its rows from the fabric again -- 327 MB for
    // a 125 MB layer at 216 <-> 96 channels, 3.5 TB/s of fabric traffic for 1.3 TB/s of algorithmic bytes.  1-D grid (gy2 > 0): workgroups
    // b, b + 8, b + 16, ... share an XCD (round-robin dispatch), so the j-th block of XCD b % 8 takes tile group j % gy of row range
    // (j / gy) * 8 + b % 8: the groups of a row range run back to back on ONE L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if (gy2 > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        by = j % gy2;
        bx = (j / gy2) * 8 + xcd;
        if ((long)bx * rows_per_block >= R) return;  // (the grid is padded to whole groups of 8 row ranges)
    }
    const long r0 = (long)bx * rows_per_block, r1 = min(R, r0 + rows_per_block);
    const int cop = by / ci_groups, cip = by % ci_groups;
    const int co0 = cop * CA, ci0 = cip * CB;  // first channel of this block's co / ci tiles
    // staging roles: NP patches of 8 rows x 8 channels per thread and step; patch p = tid + 256 k -> channel group p % NG (fastest: a row's
    // groups are contiguous in memory), row octet p / NG
    const __bf16* src[NP];
    unsigned char* dst[NP];
    int cpp[NP], r8[NP];
    bool pok[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = tid + 256 * k;
        const int g = p % NG;
        r8[k] = min(p / NG, 15);
        const bool is_a = g < GA;
        const int cg = is_a ? g : g - GA;
        cpp[k] = is_a ? d.Cout_p : d.Cin_p;
        const int ch = (is_a ? co0 : ci0) + cg * 8;
        pok[k] = p < 16 * NG && ch < cpp[k];  // a group past the last tile, or no patch: zeros / nothing
        src[k] = (is_a ? dy : x) + (pok[k] ? ch : 0);
        dst[k] = p < 16 * NG ? (is_a ? At : Bt) + (size_t)(cg * 8) * WT_PITCH + wg_slot(r8[k], cg, WT_KT / 8) * 16 : nullptr;
    }
    f32x16 acc[COT][CIT];
#pragma unroll
    for (int a = 0; a < COT; ++a)
#pragma unroll
        for (int b = 0; b < CIT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    const int tco = (wave >> 1) * COT, tci = (wave & 1) * CIT;  // this wave's first tile inside the block's group
    // The rows of step s + 1 are requested BEFORE the MFMAs of step s (register double buffer): a block's steps no longer each expose a
    // memory round trip, so fewer, longer blocks do the job -- and every block ends in fp32 atomics on the same small matrix.
    uint4 pre[NP][8];
    unsigned okbits[NP];  // applied after ALL loads are issued (a select next to a load serialises the batch)
    auto request = [&](long rb) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            okbits[k] = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long r = rb + r8[k] * 8 + i;
                const bool ok = pok[k] && r < r1;
                pre[k][i] = *reinterpret_cast<const uint4*>(src[k] + (ok ? r : r0) * cpp[k]);
                okbits[k] |= (ok ? 1u : 0u) << i;
            }
        }
    };
    if (r0 < r1) request(r0);
    for (long rb = r0; rb < r1; rb += WT_KT) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (!((okbits[k] >> i) & 1u)) pre[k][i] = make_uint4(0, 0, 0, 0);
            uint4 out[8];
            transpose8x8_bf16(pre[k], out);
            if (dst[k] != nullptr) {
#pragma unroll
                for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(dst[k] + c * WT_PITCH) = out[c];
            }
        }
        if (rb + WT_KT < r1) request(rb + WT_KT);  // in flight under the barrier and the MFMAs below
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < WT_KT / 16; ++kk) {
            bf16x8 a[COT], b[CIT];
#pragma unroll
            for (int i = 0; i < COT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(At + (size_t)((tco + i) * 32 + m) * WT_PITCH + wg_slot(kk * 2 + h, (tco + i) * 4 + (m >> 3), WT_KT / 8) * 16);
#pragma unroll
            for (int j = 0; j < CIT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Bt + (size_t)((tci + j) * 32 + m) * WT_PITCH + wg_slot(kk * 2 + h, (tci + j) * 4 + (m >> 3), WT_KT / 8) * 16);
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < CIT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < CIT; ++j) {
            const int cig = ci0 + (tci + j) * 32 + m;
            if (cig < d.Cin) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cog = co0 + (tco + i) * 32 + acc_row(reg, h);
                    if (cog < d.Cout) unsafeAtomicAdd(dw + (size_t)cog * d.Cin + cig, acc[i][j][reg]);
                }
            }
        }
}

static bool pw_wgrad_tile(const void* x, const void* dy, float* dw, const pasn_conv_desc& d, hipStream_t s) {
    if (const char* e = tune("PASN_NO_WGRAD_TILE"))
        if (e[0] == '1') return false;
    const bool pointwise = d.kt * d.kh * d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0;
    const int co_tiles = ceil_div(d.Cout_p, 32), ci_tiles = ceil_div(d.Cin_p, 32);
    if (!pointwise || co_tiles + ci_tiles <= 6) return false;  // narrow layers: every block stages all channels anyway
    // tiles per wave along the wide side (PASN_WGT_WIDE=0: one tile per wave everywhere, the kernel of rounds 2-3)
    const bool wide = !tune_is("PASN_WGT_WIDE", '0');
    const int cot = wide && co_tiles >= 2 * ci_tiles ? 2 : 1, cit = wide && cot == 1 && ci_tiles >= 2 * co_tiles ? 2 : 1;
    const int co_groups = ceil_div(co_tiles, 2 * cot), ci_groups = ceil_div(ci_tiles, 2 * cit);
    const long R = (long)d.N * d.To * d.Ho * d.Wo;
    const int gy = co_groups * ci_groups;
    // row partitions: about four blocks per CU in flight, at least two 128-row steps each
    // (two tiles per wave: the same ~13 steps per block, i.e. half the blocks -- 57-59 us at 216 <-> 96 against 64-69 with 1024: r04_pwwg_xcd.txt)
    const long target = tune_dev("PASN_WGT_BLOCKS") ? atol(tune_dev("PASN_WGT_BLOCKS")) : (cot * cit == 2 ? 512 : 1024);
    long parts = std::max<long>(1, std::min<long>(target / gy + 1, R / (2 * WT_KT)));
    long rpb = (ceil_div(R, parts) + WT_KT - 1) / WT_KT * WT_KT;
    const long gx = ceil_div(R, rpb);
    const bool xcd = !tune_is("PASN_WGT_XCD", '0');
    const dim3 grid = xcd ? dim3((unsigned)(ceil_div(gx, 8L) * 8 * gy)) : dim3((unsigned)gx, gy);
    const size_t lds = (size_t)64 * (cot + cit) * WT_PITCH;
#define WGT(A, B)                                                                                                                       \
    hipLaunchKernelGGL((pw_wgrad_tile_kernel<A, B>), grid, dim3(256), lds, s, (const __bf16*)x, (const __bf16*)dy, dw, d, co_groups, ci_groups, \
                       (int)rpb, xcd ? gy : 0)
    if (cot == 2) WGT(2, 1);
    else if (cit == 2) WGT(1, 2);
    else WGT(1, 1);
#undef WGT
    return true;
}

// returns false when the geometry is outside the fast path
bool pw_wgrad_bf16(const void* x, const void* dy, float* dw, const pasn_conv_desc& d, hipStream_t s) {
    if (d.kt * d.kh * d.kw > 64) return false;
    if (pw_wgrad_tile(x, dy, dw, d, s)) return true;
    const int co_tiles = ceil_div(d.Cout_p, 32), ci_tiles = ceil_div(d.Cin_p, 32);
    const int ntiles = co_tiles * ci_tiles;
    const bool small = (co_tiles + ci_tiles) <= 6;  // few channels: stage more rows per step so every thread has a patch to move
    const int KT = small ? 128 : 32;
    if ((size_t)(co_tiles + ci_tiles) * 32 * (KT * 2 + 16) > 64 * 1024) return false;
    if ((KT / 8) * (d.Cout_p / 8 + d.Cin_p / 8) > 512) return false;  // the kernel's register pipeline holds 2 patches per thread
    int tpw = ceil_div(ntiles, 4);
    tpw = tpw <= 1 ? 1 : tpw <= 2 ? 2 : tpw <= 4 ? 4 : 8;
    const int tpw_cap = tune_dev("PASN_WG_TPW") ? atoi(tune_dev("PASN_WG_TPW")) : 4;  // 8 tiles per wave (occupancy 1) measured 7 % slower
    tpw = std::min(tpw, std::max(1, tpw_cap));
    const int gy = ceil_div(ntiles, 4 * tpw);
#define PW(K, T) launch_pw_wgrad_bf16<K, T>(x, dy, dw, d, co_tiles, ci_tiles, gy, s)
    if (small) {
        if (tpw == 1) PW(128, 1);
        else if (tpw == 2) PW(128, 2);
        else PW(128, 4);
    } else {
        if (tpw == 1) PW(32, 1);
        else if (tpw == 2) PW(32, 2);
        else if (tpw == 4) PW(32, 4);
        else PW(32, 8);
    }
#undef PW
    return true;
}

}  // namespace pasn

// ---- depthwise 3x3 (spatial) weight gradient, strip form -------------------------------------------------------------------
// item = (channel group, strip of WT outputs along w, HR consecutive output rows of one (n, to) plane); one temporal tap per
// blockIdx.z.  Per output row a thread loads the WT gradients and the three (WT-1)*SW+3 wide input rows once and feeds all nine
// spatial taps from registers (the row-per-thread kernel above re-loads every input pixel nine times and pays an integer
// division per row); the item decomposition is done once.  Partials per block, fixed-order combine by dw_wgrad_reduce_kernel.
namespace pasn {

template <int CH, typename T>
__device__ __forceinline__ void loadc(const T* p, float (&v)[CH]) {
    if constexpr (CH == 8) load8(p, v);
    else load4(p, v);
}

// CH channels per thread (4: half the registers of 8, twice the resident waves -- the kernel is bound by load latency, not by
// bytes per load instruction); CGb = lanes per position (power of two >= Cp / CH)
// NA = temporal taps handled by one thread: 1 (one tap per blockIdx.z: x and dy stream from HBM once per tap) or 3 (all three in
// one pass: a third of the HBM traffic, three times the accumulators)
template <typename T, int SW, int WT, int CH, int NA>
__global__ __launch_bounds__(256) void dw_wgrad_strip_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                             pasn_conv_desc d, int CG, int CGb, int strips, int HR, int hgroups, long items) {
    __shared__ float red[256 * CH];
    constexpr int IW = (WT - 1) * SW + 3;
    const int cg = threadIdx.x % CGb, pl = threadIdx.x / CGb, PL = 256 / CGb;
    float acc[NA * 9][CH];
#pragma unroll
    for (int p = 0; p < NA * 9; ++p)
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[p][j] = 0.0f;
    // over (n, to, hgroup, strip); a block keeps accumulating over several item groups before its one partial is written
    for (long item = (long)blockIdx.x * PL + pl; cg < CG && item < items; item += (long)gridDim.x * PL) {
        const int strip = (int)(item % strips);
        long q = item / strips;
        const int hg = (int)(q % hgroups);
        q /= hgroups;
        const int to = (int)(q % d.To), n = (int)(q / d.To);
        const int wo0 = strip * WT, wi0 = wo0 * SW - 1;
        const T* gp = dy + (((size_t)n * d.To + to) * d.Ho) * d.Wo * d.Cout_p + cg * CH;
        const int h1 = min(d.Ho, (hg + 1) * HR);
        {
            for (int ho = hg * HR; ho < h1; ++ho) {
                // every load of the row first (clamped addresses), the zeroing of out-of-image values afterwards: a select right after
                // a load makes hipcc wait for that load on the spot -- 18 dependent round trips per output row in the first version
                float g[WT][CH];
                unsigned gok = 0;
#pragma unroll
                for (int j = 0; j < WT; ++j) {
                    const int wo = wo0 + j;
                    const bool ok = wo < d.Wo;
                    loadc<CH>(gp + ((size_t)ho * d.Wo + (ok ? wo : 0)) * d.Cout_p, g[j]);
                    gok |= (ok ? 1u : 0u) << j;
                }
#pragma unroll
                for (int ai = 0; ai < NA; ++ai) {
                const int a = NA == 1 ? (int)blockIdx.z : ai;
                const int ti = to * d.st - d.pt + a;
                const bool tok = ti >= 0 && ti < d.Ti;
                if (NA == 1 && !tok) continue;
                const T* xp = x + (((size_t)n * d.Ti + (tok ? ti : 0)) * d.Hi) * d.Wi * d.Cin_p + cg * CH;
                float xr[3][IW][CH];
                unsigned xok = 0;
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) {
                    const int hi = ho * SW - 1 + dh;
                    const bool hok = tok && hi >= 0 && hi < d.Hi;
#pragma unroll
                    for (int i = 0; i < IW; ++i) {
                        const int wi = wi0 + i;
                        const bool ok = hok && wi >= 0 && wi < d.Wi;
                        loadc<CH>(xp + ((size_t)(hok ? hi : 0) * d.Wi + (ok ? wi : 0)) * d.Cin_p, xr[dh][i]);
                        xok |= (ok ? 1u : 0u) << (dh * IW + i);
                    }
                }
                if (ai == 0) {
#pragma unroll
                    for (int j = 0; j < WT; ++j)
                        if (!((gok >> j) & 1u)) {
#pragma unroll
                            for (int e = 0; e < CH; ++e) g[j][e] = 0.0f;
                        }
                }
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) {
#pragma unroll
                    for (int i = 0; i < IW; ++i)
                        if (!((xok >> (dh * IW + i)) & 1u)) {
#pragma unroll
                            for (int e = 0; e < CH; ++e) xr[dh][i][e] = 0.0f;
                        }
#pragma unroll
                    for (int dw_ = 0; dw_ < 3; ++dw_)
#pragma unroll
                        for (int j = 0; j < WT; ++j)
#pragma unroll
                            for (int e = 0; e < CH; ++e)
                                acc[ai * 9 + dh * 3 + dw_][e] = fmaf(g[j][e], xr[dh][j * SW + dw_][e], acc[ai * 9 + dh * 3 + dw_][e]);
                }
                }
            }
        }
    }
    const int taps = d.kt * 9;
    float* out = partial + ((size_t)blockIdx.x * taps + (size_t)(NA == 1 ? blockIdx.z : 0) * 9) * d.Cout_p;
#pragma unroll
    for (int p = 0; p < NA * 9; ++p) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CH; ++j) red[threadIdx.x * CH + j] = acc[p][j];
        __syncthreads();
        for (int t = threadIdx.x; t < CGb * CH; t += 256) {
            const int g2 = t / CH, j = t % CH;
            if (g2 < CG) {
                float s = 0.0f;
                for (int q2 = 0; q2 < PL; ++q2) s += red[(q2 * CGb + g2) * CH + j];
                out[(size_t)p * d.Cout_p + g2 * CH + j] = s;
            }
        }
    }
}

// ---- depthwise 3x3x3 weight gradient, T-marching form (stride (1,s,s), pad 1) -------------------------------------------------------
// The strip kernel above runs one temporal tap per blockIdx.z: every (frame, row) of x and dy is loaded and converted three times, 18
// loads for 108 FMAs.  It is VALU-issue bound at ~1 TB/s (the forward stencil, with the same 27 FMAs per element, runs at 2.4).  Here a
// thread owns (4 channels, a strip of WT outputs, one output row) and MARCHES ALONG T: the three rows of input frame ti are loaded and
// converted ONCE and meet the gradients of output frames ti+1, ti, ti-1 (temporal taps 0, 1, 2), which sit in a three-frame register
// ring -- 18 loads for 324 FMAs, 27 x 4 accumulators.
template <int SW, int WT>
__global__ __launch_bounds__(256, 2) void dw_wgrad_march_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy, float* __restrict__ partial,
                                                             pasn_conv_desc d, int CG, int CGb, int strips, long items) {
    typedef __bf16 T;
    constexpr int CH = 4;
    __shared__ float red[256 * CH];
    constexpr int IW = (WT - 1) * SW + 3;
    const int cg = threadIdx.x % CGb, pl = threadIdx.x / CGb, PL = 256 / CGb;
    float acc[27][CH];
#pragma unroll
    for (int p = 0; p < 27; ++p)
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[p][j] = 0.0f;
    for (long item = (long)blockIdx.x * PL + pl; cg < CG && item < items; item += (long)gridDim.x * PL) {  // (n, ho, strip)
        const int strip = (int)(item % strips);
        const long q = item / strips;
        const int ho = (int)(q % d.Ho), n = (int)(q / d.Ho);
        const int wo0 = strip * WT, wi0 = wo0 * SW - 1;
        // gradients of this strip in frame `to` (clamped addresses; columns past the row are masked after the loads)
        const T* gp = dy + (((size_t)n * d.To) * d.Ho + ho) * d.Wo * d.Cout_p + cg * CH;
        const size_t gframe = (size_t)d.Ho * d.Wo * d.Cout_p;
        unsigned gmask = 0;
        int goff[WT];
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const bool ok = wo0 + j < d.Wo;
            goff[j] = (ok ? wo0 + j : 0) * d.Cout_p;
            gmask |= (ok ? 1u : 0u) << j;
        }
        // input rows hi = ho*SW - 1 + dh, columns wi0 .. wi0 + IW - 1 (clamped; masked after the loads)
        const T* xp = x + ((size_t)n * d.Ti) * d.Hi * d.Wi * d.Cin_p + cg * CH;
        const size_t xframe = (size_t)d.Hi * d.Wi * d.Cin_p;
        unsigned xmask = 0;
        int xoff[3][IW];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = ho * SW - 1 + dh;
            const bool hok = hi >= 0 && hi < d.Hi;
#pragma unroll
            for (int i = 0; i < IW; ++i) {
                const int wi = wi0 + i;
                const bool ok = hok && wi >= 0 && wi < d.Wi;
                xoff[dh][i] = ((hok ? hi : 0) * d.Wi + (ok ? wi : 0)) * d.Cin_p;
                xmask |= (ok ? 1u : 0u) << (dh * IW + i);
            }
        }
        float g0[WT][CH], g1[WT][CH], g2[WT][CH];  // gradients of output frames ti-1, ti, ti+1
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            loadc<CH>(gp + goff[j], g1[j]);
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                g0[j][e] = 0.0f;
                if (!((gmask >> j) & 1u)) g1[j][e] = 0.0f;
            }
        }
#pragma unroll 1
        for (int ti = 0; ti < d.Ti; ++ti) {
            // this frame's loads first (raw 8-byte words: 30 registers instead of 60 converted ones): the three input rows and the
            // gradients of frame ti + 1
            uint2 raw[3][IW];
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int i = 0; i < IW; ++i) raw[dh][i] = *reinterpret_cast<const uint2*>(xp + (size_t)ti * xframe + xoff[dh][i]);
            const bool next = ti + 1 < d.To;
#pragma unroll
            for (int j = 0; j < WT; ++j) loadc<CH>(gp + (size_t)(next ? ti + 1 : ti) * gframe + goff[j], g2[j]);
#pragma unroll
            for (int j = 0; j < WT; ++j)
#pragma unroll
                for (int e = 0; e < CH; ++e)
                    if (!next || !((gmask >> j) & 1u)) g2[j][e] = 0.0f;
            // temporal tap a pairs input frame ti with output frame ti - a + 1: a = 0 -> g2, 1 -> g1, 2 -> g0
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                float xr[IW][CH];  // one row converted at a time
#pragma unroll
                for (int i = 0; i < IW; ++i) {
                    const bool ok = (xmask >> (dh * IW + i)) & 1u;
                    const unsigned lo = ok ? raw[dh][i].x : 0u, hi = ok ? raw[dh][i].y : 0u;
                    xr[i][0] = __uint_as_float(lo << 16);
                    xr[i][1] = __uint_as_float(lo & 0xffff0000u);
                    xr[i][2] = __uint_as_float(hi << 16);
                    xr[i][3] = __uint_as_float(hi & 0xffff0000u);
                }
#pragma unroll
                for (int dw_ = 0; dw_ < 3; ++dw_)
#pragma unroll
                    for (int j = 0; j < WT; ++j)
#pragma unroll
                        for (int e = 0; e < CH; ++e) {
                            const float xv = xr[j * SW + dw_][e];
                            acc[0 * 9 + dh * 3 + dw_][e] = fmaf(g2[j][e], xv, acc[0 * 9 + dh * 3 + dw_][e]);
                            acc[1 * 9 + dh * 3 + dw_][e] = fmaf(g1[j][e], xv, acc[1 * 9 + dh * 3 + dw_][e]);
                            acc[2 * 9 + dh * 3 + dw_][e] = fmaf(g0[j][e], xv, acc[2 * 9 + dh * 3 + dw_][e]);
                        }
            }
#pragma unroll
            for (int j = 0; j < WT; ++j)
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    g0[j][e] = g1[j][e];
                    g1[j][e] = g2[j][e];
                }
        }
    }
    float* out = partial + (size_t)blockIdx.x * 27 * d.Cout_p;
#pragma unroll
    for (int p = 0; p < 27; ++p) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CH; ++j) red[threadIdx.x * CH + j] = acc[p][j];
        __syncthreads();
        for (int t = threadIdx.x; t < CGb * CH; t += 256) {
            const int g2i = t / CH, j = t % CH;
            if (g2i < CG) {
                float s = 0.0f;
                for (int q2 = 0; q2 < PL; ++q2) s += red[(q2 * CGb + g2i) * CH + j];
                out[(size_t)p * d.Cout_p + g2i * CH + j] = s;
            }
        }
    }
}

// ---- T-marching form, second cut (round 4) --------------------------------------------------------------------------------------------
// The kernel above asks for a step's 18 rows at the top of the step and waits for all of them (two resident waves per SIMD at 256 VGPRs
// cannot cover it): 1.1-1.7 TB/s, a quarter of its vector-issue bound.  Here
//  * a thread owns CH = 4 (or 2: 4-byte loads, half the accumulators, 4 waves per SIMD) channels of a strip of WT = 2 outputs,
//  * every row's registers are re-requested for the NEXT frame right after their conversion, ahead of the step's 27 x WT packed FMAs, and the
//    gradients one frame further ahead: a step never waits for a load it asked for in the same step,
//  * loads go through buffer descriptors: the frame offset is a scalar operand, positions outside the plane carry an out-of-range offset and
//    read as zero (no select per loaded value, no 64-bit address arithmetic per step),
//  * lanes map to (item, channel group) without padding the group count to a power of two.
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned WG_OOB = 0x80000000u;

template <int CW>
__device__ __forceinline__ void wg_load(unsigned (&r)[CW], __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    if constexpr (CW == 1) {
        r[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0);
    } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0);
        r[0] = v[0];
        r[1] = v[1];
    }
}
template <int CW>
__device__ __forceinline__ void wg_cvt(const unsigned (&r)[CW], wg_f32x2 (&v)[CW]) {
#pragma unroll
    for (int c = 0; c < CW; ++c) v[c] = wg_f32x2{__uint_as_float(r[c] << 16), __uint_as_float(r[c] & 0xffff0000u)};
}

template <int SW, int WT, int CH>
__global__ __launch_bounds__(256) void dw_wgrad_march2_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy, float* __restrict__ partial,
                                                              pasn_conv_desc d, int CG, int PL, int strips, long items) {
    constexpr int IW = (WT - 1) * SW + 3, CW = CH / 2;
    __shared__ float red[256 * CH];
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    const bool live = pl < PL;
    wg_f32x2 acc[27][CW];
#pragma unroll
    for (int p = 0; p < 27; ++p)
#pragma unroll
        for (int c = 0; c < CW; ++c) acc[p][c] = wg_f32x2{0.0f, 0.0f};
    const int Cp = d.Cout_p;
    const unsigned xframe = (unsigned)d.Hi * d.Wi * Cp * 2u, gframe = (unsigned)d.Ho * d.Wo * Cp * 2u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x), 0, (unsigned)d.N * d.Ti * xframe, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(dy), 0, (unsigned)d.N * d.To * gframe, 0x00020000);
    for (long item = (long)blockIdx.x * PL + pl; live && item < items; item += (long)gridDim.x * PL) {  // (n, ho, strip)
        const int strip = (int)(item % strips);
        const long q = item / strips;
        const int ho = (int)(q % d.Ho), n = (int)(q / d.Ho);
        const int wo0 = strip * WT, wi0 = wo0 * SW - 1;
        unsigned gv[WT], xv[3][IW];  // byte offsets inside frame 0 of clip n; outside the plane: out of range (reads as zero)
#pragma unroll
        for (int j = 0; j < WT; ++j)
            gv[j] = wo0 + j < d.Wo ? (unsigned)n * d.To * gframe + (unsigned)((ho * d.Wo + wo0 + j) * Cp + cg * CH) * 2u : WG_OOB;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = ho * SW - 1 + dh;
            const bool hok = hi >= 0 && hi < d.Hi;
#pragma unroll
            for (int i = 0; i < IW; ++i) {
                const int wi = wi0 + i;
                xv[dh][i] = (hok && wi >= 0 && wi < d.Wi) ? (unsigned)n * d.Ti * xframe + (unsigned)((hi * d.Wi + wi) * Cp + cg * CH) * 2u : WG_OOB;
            }
        }
        wg_f32x2 g0[WT][CW], g1[WT][CW];  // gradients of output frames ti - 1, ti
        unsigned gn[WT][CW], raw[3][IW][CW];
#pragma unroll
        for (int j = 0; j < WT; ++j) wg_load<CW>(gn[j], grs, gv[j], 0u);
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            wg_cvt<CW>(gn[j], g1[j]);
#pragma unroll
            for (int c = 0; c < CW; ++c) g0[j][c] = wg_f32x2{0.0f, 0.0f};
        }
#pragma unroll
        for (int j = 0; j < WT; ++j) wg_load<CW>(gn[j], grs, d.To > 1 ? gv[j] : WG_OOB, d.To > 1 ? gframe : 0u);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int i = 0; i < IW; ++i) wg_load<CW>(raw[dh][i], xrs, xv[dh][i], 0u);
#pragma unroll 1
        for (int ti = 0; ti < d.Ti; ++ti) {
            wg_f32x2 g2[WT][CW];  // gradient of output frame ti + 1 (zero past the clip)
#pragma unroll
            for (int j = 0; j < WT; ++j) wg_cvt<CW>(gn[j], g2[j]);
            {
                const bool more = ti + 2 < d.To;
                const unsigned so = (unsigned)min(ti + 2, d.To - 1) * gframe;
#pragma unroll
                for (int j = 0; j < WT; ++j) wg_load<CW>(gn[j], grs, more ? gv[j] : WG_OOB, so);
            }
            wg_f32x2 xc[3][IW][CW];
            const unsigned sx = (unsigned)min(ti + 1, d.Ti - 1) * xframe;  // (the last step's request is not used)
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
#pragma unroll
                for (int i = 0; i < IW; ++i) wg_cvt<CW>(raw[dh][i], xc[dh][i]);
#pragma unroll
                for (int i = 0; i < IW; ++i) wg_load<CW>(raw[dh][i], xrs, xv[dh][i], sx);
            }
            // temporal tap a pairs input frame ti with output frame ti - a + 1: a = 0 -> g2, 1 -> g1, 2 -> g0
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw_ = 0; dw_ < 3; ++dw_)
#pragma unroll
                    for (int j = 0; j < WT; ++j)
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            const wg_f32x2 xvv = xc[dh][j * SW + dw_][c];
                            acc[0 * 9 + dh * 3 + dw_][c] = __builtin_elementwise_fma(g2[j][c], xvv, acc[0 * 9 + dh * 3 + dw_][c]);
                            acc[1 * 9 + dh * 3 + dw_][c] = __builtin_elementwise_fma(g1[j][c], xvv, acc[1 * 9 + dh * 3 + dw_][c]);
                            acc[2 * 9 + dh * 3 + dw_][c] = __builtin_elementwise_fma(g0[j][c], xvv, acc[2 * 9 + dh * 3 + dw_][c]);
                        }
#pragma unroll
            for (int j = 0; j < WT; ++j)
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    g0[j][c] = g1[j][c];
                    g1[j][c] = g2[j][c];
                }
        }
    }
    float* out = partial + (size_t)blockIdx.x * 27 * Cp;
#pragma unroll
    for (int p = 0; p < 27; ++p) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            red[threadIdx.x * CH + 2 * c] = acc[p][c][0];
            red[threadIdx.x * CH + 2 * c + 1] = acc[p][c][1];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < CG * CH; t += 256) {
            const int g2i = t / CH, j = t % CH;
            float s = 0.0f;
            for (int q2 = 0; q2 < PL; ++q2) s += red[(q2 * CG + g2i) * CH + j];
            out[(size_t)p * Cp + g2i * CH + j] = s;
        }
    }
}

// dw[c][tap] = sum_chunks partial[chunk][tap*Cp + c]: 64 columns x 4 parts per block, parts combined in a fixed order
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int chunks, int taps, int C,
                                                              int Cp) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int L = taps * Cp;
    float s = 0.0f;
    if (col < L) {
#pragma unroll 8
        for (int ch = part; ch < chunks; ch += 4) s += partial[(size_t)ch * L + col];
    }
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && col < L) {
        const float tsum = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const int tap = col / Cp, c = col % Cp;
        if (c < C) dw[(size_t)c * taps + tap] = tsum;
    }
}

// Temporal-only depthwise conv (kh = kw = 1, stride 1: the X3D stem's (5,1,1) conv): a thread owns 8 channels of one plane
// position and walks the frames; no divisions in the loop, the kt input frames of a step are the previous step's plus one.
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_temporal_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                                pasn_conv_desc d, int CG, int CGb, long items) {
    __shared__ float red[256 * 8];
    constexpr int KMAX = 5;
    const int cg = threadIdx.x % CGb, pl = threadIdx.x / CGb, PL = 256 / CGb;
    const long HW = (long)d.Hi * d.Wi;
    float acc[KMAX][8];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.0f;
    for (long item = (long)blockIdx.x * PL + pl; cg < CG && item < items; item += (long)gridDim.x * PL) {
        const long n = item / HW, pos = item % HW;
        const T* xp = x + ((size_t)n * d.Ti * HW + pos) * d.Cin_p + cg * 8;
        const T* gp = dy + ((size_t)n * d.To * HW + pos) * d.Cout_p + cg * 8;
        for (int t = 0; t < d.To; ++t) {
            float g[8];
            load8(gp + (size_t)t * HW * d.Cout_p, g);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int ti = t - d.pt + k;
                if (k < d.kt && ti >= 0 && ti < d.Ti) {
                    float v[8];
                    load8(xp + (size_t)ti * HW * d.Cin_p, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(g[j], v[j], acc[k][j]);
                }
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * d.kt * d.Cout_p;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < d.kt) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[k][j];
            __syncthreads();
            for (int t = threadIdx.x; t < CGb * 8; t += 256) {
                const int g2 = t >> 3, j = t & 7;
                if (g2 < CG) {
                    float sum = 0.0f;
                    for (int q2 = 0; q2 < PL; ++q2) sum += red[(q2 * CGb + g2) * 8 + j];
                    out[(size_t)k * d.Cout_p + g2 * 8 + j] = sum;
                }
            }
        }
    }
}

static bool dw_temporal_ok(const pasn_conv_desc& d) {
    return d.kh == 1 && d.kw == 1 && d.kt <= 5 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && d.Ti == d.To &&
           d.Cout_p % 8 == 0 && d.Cout_p <= 2048;
}
static long dw_temporal_blocks(const pasn_conv_desc& d) {
    int CG = d.Cout_p / 8, b = 1;
    while (b < CG) b <<= 1;
    const long items = (long)d.N * d.Hi * d.Wi;
    return std::min<long>((items + 256 / b - 1) / (256 / b), 2048);
}

struct DwWgGeom {
    int ok, SW, WT, strips, HR, hgroups, CG, CGb, PL;
    long items, blocks;
};

static DwWgGeom dw_wgrad_strip_geom(const pasn_conv_desc& d) {
    DwWgGeom g{};
    if (d.kh != 3 || d.kw != 3 || d.ph != 1 || d.pw != 1 || d.sh != d.sw || (d.sh != 1 && d.sh != 2)) return g;
    if (d.Cout_p % 8 || d.Cout_p > 2048) return g;
    g.SW = d.sh;
    g.WT = g.SW == 1 ? 3 : 2;
    g.strips = ceil_div(d.Wo, g.WT);
    g.CG = d.Cout_p / 4;  // 4 channels per thread
    if (g.CG > 256) return g;
    g.CGb = 1;
    while (g.CGb < g.CG) g.CGb <<= 1;
    g.PL = 256 / g.CGb;
    // rows per item: enough items to fill the chip a few times over, few enough partial blocks to keep the combine small
    const long planes = (long)d.N * d.To;
    int HR = d.Ho;
    while (HR > 1 && planes * ceil_div(d.Ho, HR) * g.strips * g.CG < 400000) HR = (HR + 1) / 2;
    g.HR = HR;
    g.hgroups = ceil_div(d.Ho, HR);
    g.items = planes * g.hgroups * g.strips;
    // at most 768 blocks per temporal tap: each block loops over its item groups, so the partial buffer (and the combine
    // pass over it) stays small however many items there are
    g.blocks = std::min<long>((g.items + g.PL - 1) / g.PL, 768);
    g.ok = 1;
    return g;
}

}  // namespace pasn

namespace pasn {

// T-marching form: 3x3x3, temporal stride 1 and pad 1, frames kept (To == Ti)
static bool dw_wgrad_march_ok(const pasn_conv_desc& d) {
    if (const char* e = tune("PASN_NO_DWWG_MARCH"))
        if (e[0] == '1') return false;
    return d.kt == 3 && d.st == 1 && d.pt == 1 && d.To == d.Ti && d.kh == 3 && d.kw == 3 && d.ph == 1 && d.pw == 1;
}
static long dw_wgrad_march_blocks(const pasn_conv_desc& d, const DwWgGeom& g) {
    const long items = (long)d.N * d.Ho * g.strips;
    return std::min<long>((items + g.PL - 1) / g.PL, 1024);
}

struct DwWgMarch2 {
    int ok, SW, WT, CH, CG, PL, strips;
    long items, blocks;
};
// second cut of the marching kernel (PASN_DWWG_MARCH2=0: the first one).  Measured at the X3D-S shapes (tools/dwwg_bench.py,
// profiles/r04_dwwg_sweep.txt): 4 channels per thread and at most 512 blocks (= partial rows for the combine pass) is the best or within
// 2 % of the best arm at every shape; 2 channels / strips of 3 / 256-1024 blocks are switches
static DwWgMarch2 dw_wgrad_march2_geom(const pasn_conv_desc& d) {
    DwWgMarch2 g{};
    if (tune_is("PASN_DWWG_MARCH2", '0') || !dw_wgrad_march_ok(d) || d.sh != d.sw || (d.sh != 1 && d.sh != 2)) return g;
    g.CH = tune_is("PASN_DWWG_CH", '2') ? 2 : 4;
    if (d.Cout_p % g.CH || d.Cout_p / g.CH > 256) return g;
    // 32-bit byte offsets into either tensor
    if ((double)d.N * d.Ti * d.Hi * d.Wi * d.Cin_p * 2.0 >= 2147483648.0 || (double)d.N * d.To * d.Ho * d.Wo * d.Cout_p * 2.0 >= 2147483648.0) return g;
    g.SW = d.sh;
    g.WT = g.SW == 1 && tune_is("PASN_DWWG_WT", '3') ? 3 : 2;
    g.strips = ceil_div(d.Wo, g.WT);
    g.CG = d.Cout_p / g.CH;
    g.PL = 256 / g.CG;
    g.items = (long)d.N * d.Ho * g.strips;
    int cap = 512;
    if (const char* e = tune("PASN_DWWG_BLOCKS")) cap = std::max(64, atoi(e));
    g.blocks = std::min<long>((g.items + g.PL - 1) / g.PL, cap);
    g.ok = 1;
    return g;
}

size_t dw_wgrad_strip_floats(const pasn_conv_desc& d) {
    if (dw_temporal_ok(d)) return (size_t)dw_temporal_blocks(d) * d.kt * d.Cout_p;
    const DwWgGeom g = dw_wgrad_strip_geom(d);
    if (g.ok && dw_wgrad_march_ok(d)) {  // bf16 takes a marching kernel, fp32 the strip kernel: room for either
        const DwWgMarch2 m2 = dw_wgrad_march2_geom(d);
        return (size_t)std::max<long>(std::max<long>(dw_wgrad_march_blocks(d, g), g.blocks), m2.ok ? m2.blocks : 0) * 27 * d.Cout_p;
    }
    return g.ok ? (size_t)g.blocks * d.kt * 9 * d.Cout_p : 0;
}

bool dw_wgrad_strip(const void* x, const void* dy, float* ws, float* dw, const pasn_conv_desc& d, int dtype, hipStream_t s) {
    if (dw_temporal_ok(d)) {
        int CG = d.Cout_p / 8, CGb = 1;
        while (CGb < CG) CGb <<= 1;
        const long items = (long)d.N * d.Hi * d.Wi, blocks = dw_temporal_blocks(d);
        if (dtype == PASN_BF16)
            hipLaunchKernelGGL(dw_wgrad_temporal_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)x, (const __bf16*)dy, ws, d, CG,
                               CGb, items);
        else
            hipLaunchKernelGGL(dw_wgrad_temporal_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, (const float*)dy, ws, d, CG, CGb,
                               items);
        hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(ceil_div((long)d.kt * d.Cout_p, 64)), dim3(256), 0, s, ws, dw, (int)blocks, d.kt, d.Cout,
                           d.Cout_p);
        return true;
    }
    const DwWgGeom g = dw_wgrad_strip_geom(d);
    if (!g.ok) return false;
    if (const DwWgMarch2 m = dw_wgrad_march2_geom(d); dtype == PASN_BF16 && m.ok) {
#define DWM2(SWv, WTv, CHv)                                                                                                                      \
    hipLaunchKernelGGL((dw_wgrad_march2_kernel<SWv, WTv, CHv>), dim3((unsigned)m.blocks), dim3(256), 0, s, (const __bf16*)x, (const __bf16*)dy, ws, d, \
                       m.CG, m.PL, m.strips, m.items)
        if (m.SW == 2) {
            if (m.CH == 2) DWM2(2, 2, 2);
            else DWM2(2, 2, 4);
        } else if (m.WT == 3) {
            if (m.CH == 2) DWM2(1, 3, 2);
            else DWM2(1, 3, 4);
        } else {
            if (m.CH == 2) DWM2(1, 2, 2);
            else DWM2(1, 2, 4);
        }
#undef DWM2
        hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(ceil_div((long)27 * d.Cout_p, 64)), dim3(256), 0, s, ws, dw, (int)m.blocks, 27, d.Cout, d.Cout_p);
        return true;
    }
    if (dtype == PASN_BF16 && dw_wgrad_march_ok(d)) {
        const long items = (long)d.N * d.Ho * g.strips, blocks = dw_wgrad_march_blocks(d, g);
#define DWM(SWv, WTv) \
    hipLaunchKernelGGL((dw_wgrad_march_kernel<SWv, WTv>), dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)x, (const __bf16*)dy, ws, d, g.CG, g.CGb, g.strips, items)
        if (g.SW == 1) DWM(1, 3);
        else DWM(2, 2);
#undef DWM
        hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(ceil_div((long)27 * d.Cout_p, 64)), dim3(256), 0, s, ws, dw, (int)blocks, 27, d.Cout, d.Cout_p);
        return true;
    }
    const bool fuse3 = tune("PASN_DWWG_FUSED") ? atoi(tune("PASN_DWWG_FUSED")) != 0 : false;
    const bool na3 = fuse3 && d.kt == 3;
    const dim3 grid((unsigned)g.blocks, 1, na3 ? 1 : d.kt);
#define DWS(T, SWv, WTv)                                                                                                              \
    if (na3)                                                                                                                          \
        hipLaunchKernelGGL((dw_wgrad_strip_kernel<T, SWv, WTv, 4, 3>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, ws, d, g.CG, g.CGb, \
                           g.strips, g.HR, g.hgroups, g.items);                                                                      \
    else                                                                                                                              \
        hipLaunchKernelGGL((dw_wgrad_strip_kernel<T, SWv, WTv, 4, 1>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, ws, d, g.CG, g.CGb, g.strips, \
                       g.HR, g.hgroups, g.items)
    if (dtype == PASN_BF16) {
        if (g.SW == 1) DWS(__bf16, 1, 3);
        else DWS(__bf16, 2, 2);
    } else {
        if (g.SW == 1) DWS(float, 1, 3);
        else DWS(float, 2, 2);
    }
#undef DWS
    const int taps = d.kt * 9;
    hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(ceil_div((long)taps * d.Cout_p, 64)), dim3(256), 0, s, ws, dw, (int)g.blocks, taps, d.Cout,
                       d.Cout_p);
    return true;
}

}  // namespace pasn
