// Pointwise (1x1x1, stride 1) convolution for SMALL K and N (the byte-heavy X3D stage-2/3 expand / project convs):
//   Y[m][co] = act(scale[co] * sum_ci W[co][ci] X'[m][ci] + bias[co] [+ R[m][co]]),   X' = X or swish(X * gate[n]).
//
// These layers move gigabytes with ~20 FLOP/byte, so the job is to keep HBM busy, not the matrix cores.  PMC on the
// one-tile-per-wave kernel showed ~70 % of wave cycles in s_waitcnt and ~1.3 TB/s: by Little's law the few KB a wave
// had in flight -- and only during its load phase -- cannot cover HBM latency.  This kernel is PERSISTENT:
//   * a wave loads its weight fragments (NT x KS MFMA A operands, <= 64 VGPRs) and the block its scale/bias ONCE;
//   * it then walks 32-row tiles (tile += #waves); the NEXT tile's rows (the MFMA B operand, 16 bytes per lane straight
//     from global: every row is one contiguous channels-last record) are requested BEFORE the current tile's MFMAs,
//     the current tile's residual rows before its MFMAs, so loads are always in flight under compute and stores;
//   * epilogue from the accumulator: position on the lane, 4 consecutive channels per register quad -> 8/16-byte stores.
// An LDS-staged variant (coalesced whole-row loads and stores through LDS) was built first and measured SLOWER
// (3 barriers per 128-row tile; 260 us of pure per-block latency on the 6.4 M-row layer): profiles/README.md.
#include "common.h"

namespace pasn {

template <typename T, int KS, int NT, bool RES>
__global__ __launch_bounds__(256) void pwconv_persist_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ bias,
                                                             const T* __restrict__ res, const float* __restrict__ gate,
                                                             T* __restrict__ y, long M, int S, int Cin_p, int Cout, int Cout_p,
                                                             int w_kc, int act, int in_swish) {
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;
    constexpr int KSTEP = Traits<T>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sbl = reinterpret_cast<float*>(smem);                  // [2][NT*32] scale | bias of the channel chunk
    T* stage = reinterpret_cast<T*>(smem + 2 * NT * 32 * 4);       // [4 waves][32 rows][Cout_p] output images
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int co_base = blockIdx.y * (NT * 32);
    for (int i = threadIdx.x; i < NT * 32; i += 256) {
        sbl[i] = scale ? scale[co_base + i] : 1.0f;
        sbl[NT * 32 + i] = bias ? bias[co_base + i] : 0.0f;
    }
    __syncthreads();

    const int ks_real = w_kc / KSTEP;  // <= KS; the extra template steps are zero
    frag A[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            A[nt][ks] = ks < ks_real ? load_frag<T>(w + (long)(co_base + nt * 32 + c) * w_kc + ks * KSTEP + h * CH) : zero_frag<T>();

    const long ntiles = (M + 31) / 32;
    const long stride = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    const bool xform = (gate != nullptr) || (in_swish != 0);

    frag Bc[KS], Bn[KS];
    auto load_rows = [&](long t, frag (&B)[KS]) {
        const long m = t * 32 + c;
        const T* xp = x + m * Cin_p + h * CH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            B[ks] = (m < M && ks * KSTEP + h * CH < Cin_p) ? load_frag<T>(xp + ks * KSTEP) : zero_frag<T>();
    };
    if (tile < ntiles) load_rows(tile, Bc);

    for (; tile < ntiles; tile += stride) {
        const long m = tile * 32 + c;
        const bool mv = m < M;
        if (tile + stride < ntiles) load_rows(tile + stride, Bn);  // next tile's rows: in flight during everything below
        // this tile's residual rows, requested before the MFMAs that hide them
        float rv[RES ? NT : 1][4][4];  // compile-time: expand convs (no residual) do not pay 16*NT registers
        if (RES) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co_base + nt * 32 + 8 * g + 4 * h;
                    if (mv && co < Cout_p) {
                        load4(res + m * Cout_p + co, rv[nt][g]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) rv[nt][g][j] = 0.0f;
                    }
                }
        }
        if (xform) {  // x' = swish(x * gate[n][ci]), rounded back to the MFMA input type
            const float* gp = gate ? gate + (mv ? m / S : 0) * Cin_p + h * CH : nullptr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks * KSTEP + h * CH < Cin_p) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        float v = (float)Bc[ks][j];
                        if (gp) v *= gp[ks * KSTEP + j];
                        if (in_swish) v = v * sigmoidf_(v);
                        Bc[ks][j] = (T)v;
                    }
                }
            }
        }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mma32(acc[nt], A[nt][ks], Bc[ks]);
        // Epilogue.  The accumulator layout (position on the lane, 4 consecutive channels per quad) would give 8-byte
        // stores scattered at the row stride -- measured at ~1.7 TB/s of writes.  The 32 rows of a tile are ONE contiguous
        // global range, so bounce the finished tile through a WAVE-PRIVATE LDS image of exactly that range (no block
        // barrier: a wave's DS operations execute in order) and write it back as 16 bytes per lane, fully coalesced.
        T* st = stage + (size_t)wave * 32 * Cout_p;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = nt * 32 + 8 * g + 4 * h;
                if (col >= Cout_p) continue;
                float o[4], sc[4], bs[4];
                load4(sbl + col, sc);
                load4(sbl + NT * 32 + col, bs);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[nt][4 * g + j] * sc[j] + bs[j];
                if (RES) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += rv[nt][g][j];
                }
                act_vec(o, act);
                mask_tail(o, Cout - col);
                store4(st + (size_t)c * Cout_p + col, o);
            }
        __builtin_amdgcn_wave_barrier();
        {
            constexpr int CPL = 16 / (int)sizeof(T);              // elements per 16-byte piece
            const long base = tile * 32 * (long)Cout_p;           // first element of the tile in y
            const long lim = M * (long)Cout_p;                    // rows beyond M do not exist
            const int pieces = 32 * Cout_p / CPL;
            for (int q = lane; q < pieces; q += 64) {
                if (base + (long)q * CPL < lim)
                    *reinterpret_cast<uint4*>(y + base + (long)q * CPL) = *reinterpret_cast<const uint4*>(st + (size_t)q * CPL);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the image is reused by this wave's next tile
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Bc[ks] = Bn[ks];
    }
}

// Geometry shared by the launcher and the variant query.  KS = template k-steps (2, 4, 8), NT = 32-channel tiles per wave.
PwGeom pw_geom(const pasn_conv_desc& d, int dtype) {
    PwGeom g = {0, 0, 0, 0};
    const bool pointwise = d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 &&
                           d.ph == 0 && d.pw == 0;
    if (!pointwise) return g;
    if (const char* e = getenv("PASN_NO_PWCONV"))
        if (e[0] == '1') return g;
    const int kstep = dtype == PASN_BF16 ? 16 : 8;
    const int ks = d.w_kc / kstep;
    const int KS = ks <= 2 ? 2 : ks <= 4 ? 4 : ks <= 8 ? 8 : 0;
    const int tiles = ceil_div(d.Cout_p, 32);
    const int NT = tiles == 1 ? 1 : tiles == 2 ? 2 : 4;
    if (KS == 0 || tiles > 4 || KS * NT > 16) return g;  // weights must fit 64 VGPRs and one channel chunk
    g.TM = KS;        // (fields reused: TM = KS, xrow = NT)
    g.xrow = NT;
    g.co_chunk = NT * 32;
    g.lds = 0;
    return g;
}

template <typename T>
int launch_pwconv(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                  void* y, const pasn_conv_desc& d, const PwGeom& g, hipStream_t s) {
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int S = d.To * d.Ho * d.Wo;
    const long ntiles = (M + 31) / 32;
    long blocks = (ntiles + 3) / 4;
    // persistent: as many blocks as stay resident (NT = 4 instances hold ~200 VGPRs -> 2 waves per SIMD), each wave
    // strides over the row tiles
    const long cap = 256L * (g.xrow <= 2 ? 4 : 2);
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks, 1), block(256);
#define PASN_PW2(KS_, NT_, RES_)                                                                                                   \
    do {                                                                                                                      \
        if (lds > 64 * 1024) {                                                                                                \
            static bool attr = false;                                                                                         \
            if (!attr) {                                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_persist_kernel<T, KS_, NT_, RES_>),           \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);                             \
                attr = true;                                                                                                  \
            }                                                                                                                 \
        }                                                                                                                     \
        hipLaunchKernelGGL((pwconv_persist_kernel<T, KS_, NT_, RES_>), grid, block, lds, s, (const T*)x, (const T*)w, scale, bias,   \
                           (const T*)res, gate, (T*)y, M, S, d.Cin_p, d.Cout, d.Cout_p, d.w_kc, d.act, d.in_swish);           \
    } while (0)
#define PASN_PW(KS_, NT_)                 \
    do {                                  \
        if (res) PASN_PW2(KS_, NT_, true); \
        else PASN_PW2(KS_, NT_, false);    \
    } while (0)
    const int KS = g.TM, NT = g.xrow;
    const size_t lds = (size_t)2 * NT * 32 * 4 + (size_t)4 * 32 * d.Cout_p * sizeof(T);  // <= 64.5 KB (fp32, 128 channels)
    if (KS == 2) {
        if (NT == 1) PASN_PW(2, 1); else if (NT == 2) PASN_PW(2, 2); else PASN_PW(2, 4);
    } else if (KS == 4) {
        if (NT == 1) PASN_PW(4, 1); else if (NT == 2) PASN_PW(4, 2); else PASN_PW(4, 4);
    } else {
        if (NT == 1) PASN_PW(8, 1); else PASN_PW(8, 2);
    }
#undef PASN_PW
#undef PASN_PW2
    return check_launch("pwconv_persist_kernel");
}

template int launch_pwconv<float>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                  const pasn_conv_desc&, const PwGeom&, hipStream_t);
template int launch_pwconv<__bf16>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                   const pasn_conv_desc&, const PwGeom&, hipStream_t);

}  // namespace pasn
