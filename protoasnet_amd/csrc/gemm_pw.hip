// Dense convolution as an LDS-tiled implicit GEMM on MFMA with a fused epilogue:
//     Y[m][n] = act(scale[n] * sum_{tap,ci} X'[in(m,tap)][ci] * W[n][tap][ci] + bias[n] [+ R[m][n]])
// m = output position (n,t,h,w flattened), K = taps x (per-tap padded) input channels.
// Used for (a) pointwise convs with large K / N (X3D stages 4-5, head B) -- compile-time fast path PW = true -- and
// (b) every windowed dense conv of the R(2+1)D-18 and ResNet-18 trunks ((1,3,3), (3,1,1), 3x3, strided 1x1).
//
// The small-K pointwise kernel (pwconv.hip) keeps the whole weight matrix in registers; here the reuse comes from a block
// tile: BM = 128 positions x BN = 128 output channels, K walked in BK = 32 slices through double-buffered LDS.  Per
// slice every thread requests its 16-byte pieces of the X and W tiles for slice s+1 BEFORE the MFMAs of slice s (register
// staging, LDS write after the barrier), so global latency hides under the matrix work.  A 16-byte piece = 8 (bf16) /
// 4 (fp32) consecutive input channels of ONE tap (channel strides are multiples of 8), so the im2col gather is a plain
// vector load from the tap's neighbour position, or zeros outside the image / in the channel padding.
// 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA 32x32 tiles; A operand = weights (row = channel), B operand = activations
// (column = position): the accumulator has the position on the lane and 4 consecutive channels per quad.
// Epilogue: scale/bias in fp32 -> block LDS image [128 rows][128 ch] -> whole-row 16-byte-per-lane residual read +
// activation + store (fully coalesced).  LDS rows are padded to an odd number of 16-byte slots (conflict-free b128 reads).
#include "common.h"

namespace pasn {

constexpr int G_BK = 32;

// BN = 128: block tile 128 positions x 128 channels, waves 2 x 2.  BN = 64: 256 positions x 64 channels, waves 4 x 1 -- for layers with
// at most 64 output channels (half of the wide tile's MFMA work was spent on zero rows there).
template <typename T, bool PW, int G_BN>
__global__ __launch_bounds__(256) void gemm_conv_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ bias,
                                                        const T* __restrict__ res, const float* __restrict__ gate,
                                                        T* __restrict__ y, pasn_conv_desc d) {
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;            // elements per 16-byte piece
    constexpr int KSTEP = Traits<T>::KSTEP;
    constexpr int ROW = G_BK + CH;               // padded LDS row (elements)
    constexpr int PIECES = G_BK / CH;            // 16-byte pieces per tile row
    constexpr int G_BM = G_BN == 128 ? 128 : 256;
    constexpr int WM = G_BM / 64;                // waves along the positions
    constexpr int PPT = G_BM * PIECES / 256;     // X pieces per thread per tile
    constexpr int PPW = G_BN * PIECES / 256;     // W pieces per thread per tile (>= 1)
    constexpr int RSTEP = 256 / PIECES;          // tile rows between a thread's consecutive pieces
    constexpr int OROW = G_BN + CH;              // output image row (elements)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);                      // [2][G_BM][ROW]
    T* ws = xs + 2 * G_BM * ROW;                             // [2][G_BN][ROW]
    T* os = reinterpret_cast<T*>(smem);                      // [G_BM][OROW]  (aliases the tiles after the K loop)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;                // wave's 64-position / 64-channel quadrant
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int S = d.To * d.Ho * d.Wo;
    const long m0 = (long)blockIdx.x * G_BM;
    const int n0 = blockIdx.y * G_BN;
    const int Cin_p = d.Cin_p, Cout_p = d.Cout_p, kc = d.w_kc;
    const int taps = d.kt * d.kh * d.kw;
    const int Ktot = taps * kc;
    const int nk = (Ktot + G_BK - 1) / G_BK;
    const bool xform = (gate != nullptr) || (d.in_swish != 0);

    // this thread's tile rows (fixed over the K loop): row = prow0 + i * RSTEP, piece column pc
    const int pc = tid % PIECES, prow0 = tid / PIECES;
    // Per row, ONCE: the window origin as an element offset (x CHANNEL stride) and a bit mask of the taps that fall inside the image.
    // Per K slice only a slice-uniform (tap, ci) offset is added and one mask bit is tested.  (PMC on the first version, R(2+1)D's
    // 64 -> 144 (1,3,3) layer: 139 VALU instructions per slice and wave next to 8 MFMAs -- VALU 60 % busy, matrix pipe 26 %: the
    // per-slice bounds compares, 64-bit offset chains, register-stage moves and masks were the kernel.)
    long rm[PPT], robase[PPT];
    unsigned tapmask[PPT];
    bool rv[PPT];
    const int taps_ = d.kt * d.kh * d.kw;
    const bool small_window = taps_ <= 32;  // the 32-bit tap mask covers every window of the trunks (27 taps at most)
    int rt0[PPT], rh0[PPT], rw0[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        rm[i] = m0 + prow0 + i * RSTEP;
        rv[i] = rm[i] < M;
        robase[i] = 0;
        tapmask[i] = 0;
        rt0[i] = rh0[i] = rw0[i] = 0;
        if (!PW) {
            const long mm = rv[i] ? rm[i] : 0;
            const int rw = (int)(mm % d.Wo);
            long r = mm / d.Wo;
            const int rh = (int)(r % d.Ho);
            r /= d.Ho;
            const int rt = (int)(r % d.To);
            const int rn = (int)(r / d.To);
            rt0[i] = rt * d.st - d.pt;
            rh0[i] = rh * d.sh - d.ph;
            rw0[i] = rw * d.sw - d.pw;
            robase[i] = ((((long)rn * d.Ti + rt0[i]) * d.Hi + rh0[i]) * d.Wi + rw0[i]) * Cin_p;
            if (small_window && rv[i]) {
                int tp = 0;
                for (int a = 0; a < d.kt; ++a)
                    for (int b2 = 0; b2 < d.kh; ++b2)
                        for (int e = 0; e < d.kw; ++e, ++tp)
                            if ((unsigned)(rt0[i] + a) < (unsigned)d.Ti && (unsigned)(rh0[i] + b2) < (unsigned)d.Hi &&
                                (unsigned)(rw0[i] + e) < (unsigned)d.Wi)
                                tapmask[i] |= 1u << tp;
            }
        }
    }
    // this thread's piece of slice kt is k = kt * G_BK + pc * CH = tap * kc + ci: decoded once, then advanced per slice
    int f_tap = 0, f_ci = pc * CH, f_da = 0, f_db = 0, f_de = 0;
    if (!PW) {
        f_tap = f_ci / kc;
        f_ci -= f_tap * kc;
        f_da = f_tap / (d.kh * d.kw);
        const int r2 = f_tap - f_da * d.kh * d.kw;
        f_db = r2 / d.kw;
        f_de = r2 - f_db * d.kw;
    }

    // Two register stages: B = the slice just requested, A = the one stashed next (moved B -> A once per slice; unrolling the K loop
    // x2 to make the stages compile-time names doubled the accumulator registers -- the MFMA loop must stay ONE basic block).
    uint4 xrA[PPT], wrA[PPW], xrB[PPT], wrB[PPW];
    unsigned okA = 0, okB = 0;  // bit i: row i of the stage is inside the image / the K range
    // ISSUE ONLY: raw, unconditional loads from clamped addresses.  The masks (image border / K tail) are applied in
    // stash(), after the MFMAs of the current slice: a select on the loaded value right here makes hipcc wait
    // (vmcnt(0)) for the prefetch BEFORE the MFMAs it was meant to hide under.
    auto fetch = [&](int kt, uint4 (&xr)[PPT], uint4 (&wr)[PPW], unsigned& okbits) {  // called with kt = 0, 1, 2, ... in order
        const int k = kt * G_BK + pc * CH;  // flattened K index of this thread's piece
        const int ci = PW ? k : f_ci;
        const bool kvalid = (PW ? k < Ktot : f_tap < taps) && ci < Cin_p;
        const int tapoff = PW ? ci : (((f_da * d.Hi) + f_db) * d.Wi + f_de) * Cin_p + ci;  // slice-uniform per thread
        okbits = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            bool ok = rv[i] && kvalid;
            long off;
            if (PW) {
                off = rm[i] * Cin_p + tapoff;
            } else {
                if (small_window) ok = ok && ((tapmask[i] >> (f_tap & 31)) & 1u);
                else ok = ok && (unsigned)(rt0[i] + f_da) < (unsigned)d.Ti && (unsigned)(rh0[i] + f_db) < (unsigned)d.Hi &&
                          (unsigned)(rw0[i] + f_de) < (unsigned)d.Wi;
                off = robase[i] + tapoff;
            }
            okbits |= (ok ? 1u : 0u) << i;
            xr[i] = *reinterpret_cast<const uint4*>(x + (ok ? off : 0));
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int n = n0 + prow0 + i * RSTEP;  // weight rows are zero padded to a multiple of 128
            wr[i] = *reinterpret_cast<const uint4*>(w + (long)n * Ktot + (k < Ktot ? k : 0));
        }
        if (!PW) {  // advance (tap, ci) by one slice
            f_ci += G_BK;
            while (f_ci >= kc) {
                f_ci -= kc;
                ++f_tap;
                if (++f_de == d.kw) {
                    f_de = 0;
                    if (++f_db == d.kh) {
                        f_db = 0;
                        ++f_da;
                    }
                }
            }
        }
    };
    auto stash = [&](int kt, int buf, const uint4 (&xr)[PPT], const uint4 (&wr)[PPW], unsigned okbits) {  // registers -> LDS
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int row = prow0 + i * RSTEP;
            uint4 xv = xr[i];
            const bool ok = (okbits >> i) & 1u;
            xv.x = ok ? xv.x : 0u;
            xv.y = ok ? xv.y : 0u;
            xv.z = ok ? xv.z : 0u;
            xv.w = ok ? xv.w : 0u;
            if (PW && xform) {
                const int k = kt * G_BK + pc * CH;
                if (rv[i] && k < Cin_p) {
                    T* e = reinterpret_cast<T*>(&xv);
                    const float* gp = gate ? gate + (rm[i] / S) * Cin_p + k : nullptr;
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        float v = (float)e[j];
                        if (gp) v *= gp[j];
                        if (d.in_swish) v = v * sigmoidf_(v);
                        e[j] = (T)v;
                    }
                }
            }
            *reinterpret_cast<uint4*>(xs + ((size_t)buf * G_BM + row) * ROW + pc * CH) = xv;
        }
        const bool kin = kt * G_BK + pc * CH < Ktot;  // only the last slice can run past the K extent
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            uint4 wv = wr[i];
            if (kt == nk - 1) {
                wv.x = kin ? wv.x : 0u;
                wv.y = kin ? wv.y : 0u;
                wv.z = kin ? wv.z : 0u;
                wv.w = kin ? wv.w : 0u;
            }
            *reinterpret_cast<uint4*>(ws + ((size_t)buf * G_BN + prow0 + i * RSTEP) * ROW + pc * CH) = wv;
        }
    };

    f32x16 acc[2][2];  // [channel tile][position tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    auto mma_slice = [&](int buf) {
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
    };

    auto shift = [&]() {
#pragma unroll
        for (int i = 0; i < PPT; ++i) xrA[i] = xrB[i];
#pragma unroll
        for (int i = 0; i < PPW; ++i) wrA[i] = wrB[i];
        okA = okB;
    };
    fetch(0, xrB, wrB, okB);
    shift();
    if (nk > 1) fetch(1, xrB, wrB, okB);
    stash(0, 0, xrA, wrA, okA);  // waits for slice 0 only (slice 1 was issued after it)
    shift();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 2 < nk) fetch(kt + 2, xrB, wrB, okB);  // slices kt+1 (stage A) and kt+2 (stage B) are in flight under the MFMAs
        mma_slice(buf);
        if (kt + 1 < nk) stash(kt + 1, buf ^ 1, xrA, wrA, okA);  // the other buffer was last read in iteration kt-1 (barrier below)
        shift();
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
            act_vec(v, d.act);
            mask_tail(v, d.Cout - n);
            store8(y + m * Cout_p + n, v);
        }
    }
}

static bool is_pointwise(const pasn_conv_desc& d) {
    return d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0;
}

bool gemm_pw_applicable(const pasn_conv_desc& d, int dtype) {
    if (const char* e = tune("PASN_NO_GEMM"))
        if (e[0] == '1') return false;
    (void)dtype;
    if (d.w_rows % 128 != 0 || d.w_kc % 8 != 0) return false;
    if (is_pointwise(d)) return d.Cin_p >= 64;  // below that the register-resident kernel (pwconv.hip) wins
    if (d.in_swish) return false;               // the fused input transform exists on the pointwise paths only
    return d.Cin_p * d.kt * d.kh * d.kw >= 64;  // windowed dense convs with a real K
}

template <typename T>
int launch_gemm_pw(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                   void* y, const pasn_conv_desc& d, hipStream_t s) {
    constexpr int CH = Traits<T>::CH;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    // narrow tile only when ALL the output channels fit it (Cout <= 64: 216 -> 199 us on R(2+1)D's 144 -> 64 layers).  Splitting
    // 144 / 288 / 576 channels into 64-wide tiles instead of a half-empty last 128-wide one was measured 10-19 % SLOWER: the kernel
    // is bound by staging the activation tile (re-read once per channel tile), not by the MFMAs spent on zero rows.
    const bool no_narrow = tune("PASN_NO_GEMM_BN64") != nullptr;
    const bool narrow = !no_narrow && d.Cout_p <= 64;
    const int BN = narrow ? 64 : 128, BM = narrow ? 256 : 128;
    const size_t tiles = (size_t)2 * (BM + BN) * (G_BK + CH) * sizeof(T);
    const size_t image = (size_t)BM * (BN + CH) * sizeof(T);
    const size_t lds = tiles > image ? tiles : image;
    const dim3 grid(ceil_div(M, BM), ceil_div(d.Cout_p, BN)), block(256);
    const bool pw = is_pointwise(d);
    if (!pw) PASN_REQUIRE(gate == nullptr, "the SE gate transform is only fused into pointwise convs");
#define PASN_GC(PW_, BN_)                                                                                                  \
    do {                                                                                                                    \
        if (lds > 64 * 1024) PASN_MAX_LDS(96 * 1024, gemm_conv_kernel<T, PW_, BN_>);                                      \
        hipLaunchKernelGGL((gemm_conv_kernel<T, PW_, BN_>), grid, block, lds, s, (const T*)x, (const T*)w, scale, bias,     \
                           (const T*)res, gate, (T*)y, d);                                                                  \
    } while (0)
    if (pw) {
        if (narrow) PASN_GC(true, 64); else PASN_GC(true, 128);
    } else {
        if (narrow) PASN_GC(false, 64); else PASN_GC(false, 128);
    }
#undef PASN_GC
    return check_launch("gemm_conv_kernel");
}

template int launch_gemm_pw<float>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                   const pasn_conv_desc&, hipStream_t);
template int launch_gemm_pw<__bf16>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                    const pasn_conv_desc&, hipStream_t);

}  // namespace pasn
