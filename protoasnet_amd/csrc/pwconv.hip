// Pointwise (1x1x1, stride 1) convolution = row-streaming GEMM  Y[m][co] = act(scale*sum_ci W[co][ci] X'[m][ci] + bias [+ R[m][co]])
// over channels-last rows, built for memory-level parallelism (every X3D expand / project conv and the five convs of
// head B; PMC showed the fragment-from-global conv kernel ~70 % of its wave-cycles in s_waitcnt on dependent
// k-step loads while moving only ~1.3 TB/s).
//
//   phase 1  the block's TM consecutive rows of X are ONE contiguous byte range: all 256 threads stream it into LDS
//            with independent 16-byte loads (whole cache lines, many in flight), applying the optional fused input
//            transform x' = swish(x * gate[n][ci]) once per element on the way;
//   phase 2  MFMA straight from LDS (B operand = activation rows, ds_read_b128, odd 16-byte-slot row stride =>
//            conflict-free) against weight fragments from L2 (A operand), output channels in chunks of <= 128;
//   phase 3  the accumulators (position on the lane, 4 consecutive channels per register quad) are staged through
//            LDS so that the residual read and the store are whole-row 16-byte-per-lane streams as well.
#include "common.h"

namespace pasn {

template <typename T>
__global__ __launch_bounds__(256) void pwconv_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ bias,
                                                     const T* __restrict__ res, const float* __restrict__ gate,
                                                     T* __restrict__ y, long M, int S, int Cin_p, int Cout, int Cout_p,
                                                     int w_kc, int act, int in_swish, int TM, int xrow, int co_chunk) {
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;
    constexpr int KSTEP = Traits<T>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);  // [TM][xrow]   (xrow >= w_kc, tail zero)
    const int orow = co_chunk + 8;
    T* os = xs + (size_t)TM * xrow;      // [TM][orow]   output staging of one channel chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const long m0 = (long)blockIdx.x * TM;

    // ---- phase 1: rows -> LDS (8-channel groups; groups >= Cin_p/8 are the zero K-padding) --------------------
    {
        const int cgs_in = Cin_p / 8, cgs_row = w_kc / 8;  // the MFMA reads k in [0, w_kc); the row pad beyond is never read
        const bool xform = (gate != nullptr) || (in_swish != 0);
        for (int i = threadIdx.x; i < TM * cgs_row; i += 256) {
            const int pl = i / cgs_row, cg = i - pl * cgs_row;
            const long m = m0 + pl;
            float v[8];
            if (m < M && cg < cgs_in) {
                load8(x + m * Cin_p + cg * 8, v);
                if (xform) {
                    float g[8];
                    if (gate) load8(gate + (m / S) * Cin_p + cg * 8, g);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float t = gate ? v[j] * g[j] : v[j];
                        v[j] = in_swish ? t * sigmoidf_(t) : t;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.0f;
            }
            store8(xs + (size_t)pl * xrow + cg * 8, v);
        }
    }
    __syncthreads();

    // ---- phase 2/3 per output-channel chunk ---------------------------------------------------------------------------
    const int WM = TM / 32;            // position tiles (4, 2 or 1); the 4 waves split as WM x (4/WM)
    const int ptile = wave % WM, cosplit = wave / WM, nsplit = 4 / WM;
    const int ksteps = w_kc / KSTEP;
    const T* xrow_p = xs + (size_t)(ptile * 32 + c) * xrow + h * CH;
    for (int co0 = 0; co0 < Cout_p; co0 += co_chunk) {
        const int width = min(co_chunk, Cout_p - co0);  // channels of this chunk (multiple of 8)
        const int tiles = (width + 31) / 32;
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
        const T* wp[4];
        bool on[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tj = cosplit + j * nsplit;
            on[j] = tj < tiles;  // wave-uniform
            wp[j] = w + (long)(co0 + (on[j] ? tj : 0) * 32 + c) * w_kc + h * CH;
        }
        for (int ks = 0; ks < ksteps; ++ks) {
            const frag bf = load_frag<T>(xrow_p + ks * KSTEP);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (on[j]) {
                    const frag af = load_frag<T>(wp[j] + ks * KSTEP);
                    mma32(acc[j], af, bf);
                }
            }
        }
        // accumulators -> staging rows (scale/bias applied in fp32, activation after the residual in the copy-out)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!on[j]) continue;
            const int tj = cosplit + j * nsplit;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = tj * 32 + 8 * g + 4 * h;  // channel inside the chunk
                if (col >= width) continue;
                const int co = co0 + col;
                float o[4], sc[4] = {1.0f, 1.0f, 1.0f, 1.0f}, bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (scale) load4(scale + co, sc);  // one 16-byte load per quad, not four dword gathers
                if (bias) load4(bias + co, bs);
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = acc[j][4 * g + q] * sc[q] + bs[q];
                store4(os + (size_t)(ptile * 32 + c) * orow + col, o);
            }
        }
        __syncthreads();
        {
            const int cgs_out = width / 8;
            for (int i = threadIdx.x; i < TM * cgs_out; i += 256) {
                const int pl = i / cgs_out, cg = i - pl * cgs_out;
                const long m = m0 + pl;
                if (m >= M) continue;
                float v[8];
                load8(os + (size_t)pl * orow + cg * 8, v);
                const int co = co0 + cg * 8;
                if (res) {
                    float r[8];
                    load8(res + m * Cout_p + co, r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += r[j];
                }
                act_vec(v, act);
                mask_tail(v, Cout - co);
                store8(y + m * Cout_p + co, v);
            }
        }
        __syncthreads();
    }
}

// Geometry shared by the launcher and the variant query.
PwGeom pw_geom(const pasn_conv_desc& d, int dtype) {
    PwGeom g = {0, 0, 0, 0};
    const bool pointwise = d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 &&
                           d.ph == 0 && d.pw == 0;
    if (!pointwise) return g;
    if (const char* e = getenv("PASN_NO_PWCONV"))
        if (e[0] == '1') return g;
    const int es = dtype == PASN_BF16 ? 2 : 4;
    const int pad = 16 / es;  // one 16-byte slot
    g.xrow = ((d.w_kc * es / 16) % 2 == 0) ? d.w_kc + pad : d.w_kc;  // odd number of 16-byte slots per row
    g.co_chunk = d.Cout_p >= 128 ? 128 : (d.Cout_p + 31) / 32 * 32;
    // largest row tile whose LDS footprint still lets two blocks share a CU (160 KB); 0 = use the generic kernel
    for (int tm = 128; tm >= 32; tm >>= 1) {
        const size_t bytes = (size_t)tm * (g.xrow + g.co_chunk + 8) * es;
        if (bytes <= 80 * 1024) {
            g.TM = tm;
            g.lds = (int)bytes;
            break;
        }
    }
    return g;
}

template <typename T>
int launch_pwconv(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                  void* y, const pasn_conv_desc& d, const PwGeom& g, hipStream_t s) {
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int S = d.To * d.Ho * d.Wo;
    static bool attr_set = false;
    if (!attr_set) {  // allow > 64 KB of dynamic LDS (the kernel never asks for more than 96 KB)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((pwconv_kernel<T>), dim3(ceil_div(M, g.TM)), dim3(256), (size_t)g.lds, s, (const T*)x, (const T*)w,
                       scale, bias, (const T*)res, gate, (T*)y, M, S, d.Cin_p, d.Cout, d.Cout_p, d.w_kc, d.act, d.in_swish,
                       g.TM, g.xrow, g.co_chunk);
    return check_launch("pwconv_kernel");
}

template int launch_pwconv<float>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                  const pasn_conv_desc&, const PwGeom&, hipStream_t);
template int launch_pwconv<__bf16>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                   const pasn_conv_desc&, const PwGeom&, hipStream_t);

}  // namespace pasn
