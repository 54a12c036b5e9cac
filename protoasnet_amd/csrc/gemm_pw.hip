// Pointwise (1x1x1, stride 1) convolution with LARGE K / N (X3D stages 4-5, head B's five convs): a classic LDS-tiled
// MFMA GEMM   Y[m][n] = act(scale[n] * sum_k X'[m][k] W[n][k] + bias[n] [+ R[m][n]])   with a fused epilogue.
//
// The small-K kernel (pwconv.hip) keeps the whole weight matrix in registers; here K*N is up to 432x432, so the reuse
// has to come from a block tile: BM = 128 rows x BN = 128 output channels, K walked in BK = 32 slices through
// double-buffered LDS.  Per slice every thread requests its two 16-byte pieces of the X and W tiles for slice s+1
// BEFORE the MFMAs of slice s (register staging, write after the barrier), so global latency hides under the matrix
// work.  4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA 32x32 tiles; A operand = weights (row = channel), B operand =
// activations (column = position), so the accumulator has the position on the lane and 4 consecutive channels per
// quad.  Epilogue: scale/bias in fp32 -> block LDS image [128 rows][128 ch] -> whole-row 16-byte-per-lane residual read
// + activation + store (fully coalesced; the accumulator layout alone would give 8-byte stores at an 864-byte stride).
// LDS rows are padded to an odd number of 16-byte slots: conflict-free ds_read_b128 fragment reads.
#include "common.h"

namespace pasn {

constexpr int G_BM = 128, G_BN = 128, G_BK = 32;

template <typename T>
__global__ __launch_bounds__(256) void gemm_pw_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                      const float* __restrict__ scale, const float* __restrict__ bias,
                                                      const T* __restrict__ res, const float* __restrict__ gate,
                                                      T* __restrict__ y, long M, int S, int Cin_p, int Cout, int Cout_p,
                                                      int w_kc, int act, int in_swish) {
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;            // elements per 16-byte piece
    constexpr int KSTEP = Traits<T>::KSTEP;
    constexpr int ROW = G_BK + CH;               // padded LDS row (elements): G_BK*es/16 slots (even) + 1
    constexpr int PIECES = G_BK / CH;            // 16-byte pieces per tile row
    constexpr int PPT = G_BM * PIECES / 256;     // pieces per thread per tile (2 for bf16, 4 for fp32)
    constexpr int OROW = G_BN + CH;              // output image row (elements)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);                      // [2][G_BM][ROW]
    T* ws = xs + 2 * G_BM * ROW;                             // [2][G_BN][ROW]
    T* os = reinterpret_cast<T*>(smem);                      // [G_BM][OROW]  (aliases the tiles after the K loop)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;                 // wave's 64-row / 64-channel quadrant
    const long m0 = (long)blockIdx.x * G_BM;
    const int n0 = blockIdx.y * G_BN;
    const bool xform = (gate != nullptr) || (in_swish != 0);
    const int nk = (w_kc + G_BK - 1) / G_BK;

    uint4 xr[PPT], wr[PPT];
    auto fetch = [&](int kt) {  // global -> registers for K slice kt
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = tid + i * 256;
            const int row = p / PIECES, pc = p - row * PIECES;
            const int k = kt * G_BK + pc * CH;
            const long m = m0 + row;
            xr[i] = (m < M && k < Cin_p) ? *reinterpret_cast<const uint4*>(x + m * Cin_p + k) : make_uint4(0, 0, 0, 0);
            const int n = n0 + row;  // weight rows are zero padded to a multiple of 128
            wr[i] = (k < w_kc) ? *reinterpret_cast<const uint4*>(w + (long)n * w_kc + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto stash = [&](int kt, int buf) {  // registers -> LDS (applying the fused input transform to X once)
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = tid + i * 256;
            const int row = p / PIECES, pc = p - row * PIECES;
            uint4 xv = xr[i];
            if (xform) {
                const int k = kt * G_BK + pc * CH;
                const long m = m0 + row;
                if (m < M && k < Cin_p) {
                    T* e = reinterpret_cast<T*>(&xv);
                    const float* gp = gate ? gate + (m / S) * Cin_p + k : nullptr;
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        float v = (float)e[j];
                        if (gp) v *= gp[j];
                        if (in_swish) v = v * sigmoidf_(v);
                        e[j] = (T)v;
                    }
                }
            }
            *reinterpret_cast<uint4*>(xs + ((size_t)buf * G_BM + row) * ROW + pc * CH) = xv;
            *reinterpret_cast<uint4*>(ws + ((size_t)buf * G_BN + row) * ROW + pc * CH) = wr[i];
        }
    };

    f32x16 acc[2][2];  // [channel tile][row tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    fetch(0);
    stash(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch(kt + 1);  // in flight under the MFMAs below
        const T* xb = xs + ((size_t)buf * G_BM + wm * 64 + c) * ROW + h * CH;
        const T* wb = ws + ((size_t)buf * G_BN + wn * 64 + c) * ROW + h * CH;
#pragma unroll
        for (int ks = 0; ks < G_BK / KSTEP; ++ks) {
            frag a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = load_frag<T>(wb + (size_t)i * 32 * ROW + ks * KSTEP);
                b[i] = load_frag<T>(xb + (size_t)i * 32 * ROW + ks * KSTEP);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma32(acc[i][j], a[i], b[j]);
        }
        if (kt + 1 < nk) stash(kt + 1, buf ^ 1);  // the other buffer was last read in iteration kt-1 (barrier below)
        __syncthreads();
    }

    // ---- epilogue: scale/bias -> LDS image -> coalesced residual + activation + store ----------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wn * 64 + i * 32 + 8 * g + 4 * h;  // channel inside the block tile
                const int n = n0 + col;
                float o[4], sc[4] = {1.0f, 1.0f, 1.0f, 1.0f}, bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (scale) load4(scale + n, sc);
                if (bias) load4(bias + n, bs);
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = acc[i][j][4 * g + q] * sc[q] + bs[q];
                store4(os + (size_t)(wm * 64 + j * 32 + c) * OROW + col, o);
            }
    __syncthreads();
    {
        const int width = min(G_BN, Cout_p - n0);  // channels of this block that exist (multiple of 8)
        const int cgs = width / 8;
        for (int p = tid; p < G_BM * cgs; p += 256) {
            const int row = p / cgs, cg = p - row * cgs;
            const long m = m0 + row;
            if (m >= M) continue;
            float v[8];
            load8(os + (size_t)row * OROW + cg * 8, v);
            const int n = n0 + cg * 8;
            if (res) {
                float r[8];
                load8(res + m * Cout_p + n, r);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += r[q];
            }
            act_vec(v, act);
            mask_tail(v, Cout - n);
            store8(y + m * Cout_p + n, v);
        }
    }
}

bool gemm_pw_applicable(const pasn_conv_desc& d, int dtype) {
    const bool pointwise = d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 &&
                           d.ph == 0 && d.pw == 0;
    if (!pointwise) return false;
    if (const char* e = getenv("PASN_NO_GEMM"))
        if (e[0] == '1') return false;
    (void)dtype;
    return d.w_rows % G_BN == 0 && d.Cin_p >= 64;  // below that the register-resident kernel (pwconv.hip) wins
}

template <typename T>
int launch_gemm_pw(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                   void* y, const pasn_conv_desc& d, hipStream_t s) {
    constexpr int CH = Traits<T>::CH;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int S = d.To * d.Ho * d.Wo;
    const size_t tiles = (size_t)2 * (G_BM + G_BN) * (G_BK + CH) * sizeof(T);
    const size_t image = (size_t)G_BM * (G_BN + CH) * sizeof(T);
    const size_t lds = tiles > image ? tiles : image;
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pw_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  96 * 1024);
        attr = true;
    }
    const dim3 grid(ceil_div(M, G_BM), ceil_div(d.Cout_p, G_BN)), block(256);
    hipLaunchKernelGGL((gemm_pw_kernel<T>), grid, block, lds, s, (const T*)x, (const T*)w, scale, bias, (const T*)res, gate,
                       (T*)y, M, S, d.Cin_p, d.Cout, d.Cout_p, d.w_kc, d.act, d.in_swish);
    return check_launch("gemm_pw_kernel");
}

template int launch_gemm_pw<float>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                   const pasn_conv_desc&, hipStream_t);
template int launch_gemm_pw<__bf16>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                    const pasn_conv_desc&, hipStream_t);

}  // namespace pasn
