// Pointwise conv for wide layers (Cin_p >= 64: X3D stages 4-5, the prototype-head convs): X tiles shared through LDS,
// one 32-channel output tile per wave with its weights in registers, latency hidden by occupancy.
//
// These GEMMs are skinny -- M = 25k..100k positions, K and N a few hundred channels -- so they are bound by streaming
// X in and Y out, not by the matrix cores.  The 128 x 128 LDS-tiled kernel (gemm_pw.hip) walks K in 32-wide slices with
// a barrier each: 16 KB in flight per block, a serial prologue and epilogue per tile, 0.8-2.4 TB/s measured.  Here:
//   * a block owns 64 positions x 4 channel tiles (one per wave; blockIdx walks (position tile, channel-tile group) with
//     the groups of one position tile adjacent, so X comes from HBM once and from L2 for the other groups);
//   * EVERY independent global load is issued up front, in the order it is consumed: the 16-byte pieces of the whole-K X
//     tile, the gate rows, this wave's weight fragments for the whole K extent (KS registers-resident fragments, as in
//     pwconv.hip), the residual rows.  One memory round trip per block instead of a chain of them -- two earlier versions
//     of this file show what the chain costs: weight fragments streamed from L2 with a 4-step prefetch stalled every
//     4 k-steps (~1 us of L2 latency against 0.1 us of MFMA), and a persistent variant that prefetched the next tile into
//     registers ran at 254 VGPRs, where hipcc parks the accumulators in AGPRs and moves them around every MFMA;
//   * the fused input transform x' = swish(x * gate) is applied once per element on the way into LDS (gate rows from LDS);
//   * K loop: A from registers, B from the LDS tile (row stride an odd number of 16-byte slots: conflict-free);
//   * epilogue: scale/bias -> wave-private fp32 LDS image -> row-wise residual + activation + 16-byte stores.
// No software pipeline across tiles: 90-200 VGPRs and 34-78 KB of LDS leave 2-4 blocks per CU in different phases.
#include "common.h"

namespace pasn {

// Strided 1x1x1 convs (the X3D / R(2+1)D shortcut convs): output position m reads input row rowmap(m).
struct XtStride {
    int on;                 // 0: rows are contiguous (stride 1)
    int To, Ho, Wo, Ti, Hi, Wi, st, sh, sw;
};
__device__ __forceinline__ long xt_row(const XtStride& g, long m) {
    if (!g.on) return m;
    const int wo = (int)(m % g.Wo);
    long r = m / g.Wo;
    const int ho = (int)(r % g.Ho);
    r /= g.Ho;
    const int to = (int)(r % g.To);
    const long n = r / g.To;
    return ((n * g.Ti + (long)to * g.st) * g.Hi + (long)ho * g.sh) * g.Wi + (long)wo * g.sw;
}

constexpr int XT_BM = 64;        // positions per tile
constexpr int XT_SROW = 36;      // fp32 row stride of the 32 x 32 epilogue image (9 slots: odd)
constexpr int XT_MAXP = 14;      // 16-byte pieces of a tile per thread (64 rows x 448 bf16 / 224 fp32 columns)

// minimum waves per SIMD the allocator must keep: the grid is many uniform small blocks, so resident slots decide the
// number of "rounds" (KS = 14 with residual came out ONE register over the 3-wave limit without this)
template <typename T, int KS, bool XF, bool RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KS <= 8 ? 4 : (KS <= 14 ? 3 : 2))))
void pwconv_xtile_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                                           const T* __restrict__ res, const float* __restrict__ gate,
                                                           T* __restrict__ y, long M, int S, int Cin_p, int Cout, int Cout_p,
                                                           int w_kc, int act, int in_swish, int gy, int w_frag, XtStride geo) {
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;
    constexpr int KSTEP = Traits<T>::KSTEP;
    constexpr int PPR = KS * 2;                // 16-byte pieces per LDS row: exactly KS k-steps, zero beyond Cin_p
    constexpr int NP = PPR * XT_BM / 256;      // pieces per thread (KS / 2)
    constexpr int KP = KS * KSTEP + CH;        // LDS row stride: 2 KS + 1 slots (odd: conflict-free b128 reads)
    static_assert(KS % 2 == 0, "KS must be even");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);  // [64][KP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    float* fbase = reinterpret_cast<float*>(smem + (size_t)XT_BM * KP * sizeof(T));
    float* scr = fbase + wave * 32 * XT_SROW;  // [4 waves][32][36] epilogue images
    float* gl = fbase + 4 * 32 * XT_SROW;      // [2 clips][w_kc] gate rows of this tile (XF only)
    const int by = blockIdx.x % gy;
    const long tile = blockIdx.x / gy;
    const int ctiles = (Cout_p + 31) / 32;
    const int ct = by * 4 + wave;  // this wave's channel tile (may not exist: the wave then only helps staging)
    const bool live = ct < ctiles;
    const int co = ct * 32;
    const long m0 = tile * XT_BM;
    const long n0 = m0 / S;
    const int r0 = (int)(m0 - n0 * S);
    const int nks = w_kc / KSTEP;  // <= KS

    // ---- all independent loads, in consumption order: X pieces, gate rows, weights, residual rows ---------------------
    uint4 xr[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        // ISSUE ONLY (clamped address, no use of the value here): a select on the loaded value inside this loop makes
        // hipcc wait for each load before issuing the next -- 7-14 sequential round trips, measured 8-10 us per block
        const int p = tid + 256 * u, prow = p / PPR, pcol = p - prow * PPR;  // compile-time divisor
        const bool ok = m0 + prow < M && pcol * CH < Cin_p;
        xr[u] = *reinterpret_cast<const uint4*>(x + (ok ? xt_row(geo, m0 + prow) * Cin_p + pcol * CH : 0));
    }
    float gv[XF ? 4 : 1];
    if (XF && gate) {  // gate rows of the (at most two: S >= 64) clips this tile touches; 2 * w_kc <= 1024 floats
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, q = i / w_kc, k = i - q * w_kc;
            const bool ok = i < 2 * w_kc && k < Cin_p && (n0 + q) * (long)S < M;
            const float g = gate[ok ? (n0 + q) * Cin_p + k : 0];
            gv[u] = ok ? g : 0.0f;
        }
    }
    frag A[KS];  // stationary weights: rows co + c, the whole K extent
    // fragment-major: (tile ct, step ks) is 64 lanes x 16 bytes, contiguous; row-major: a 32-row gather.
    // Unconditional loads from clamped (existing) fragments; steps beyond w_kc are zeroed after the staging barrier.
    // Wide layers (KS > 16) request only the first half here and the rest once the staging registers `xr` are dead
    // (both halves live with xr is > 256 registers = one wave per SIMD); the second half lands under the first MFMAs.
    constexpr int KA = KS > 16 ? KS / 2 : KS;
    const int ctc = live ? ct : ctiles - 1;
    const T* abase = w_frag ? w + ((long)ctc * nks * 64 + lane) * CH : w + (long)(ctc * 32 + c) * w_kc + h * CH;
    const int astep = w_frag ? 64 * CH : KSTEP;
#pragma unroll
    for (int ks = 0; ks < KA; ++ks) A[ks] = load_frag<T>(abase + (size_t)(ks < nks ? ks : nks - 1) * astep);
    uint4 rr[RES ? 2 : 1][2][sizeof(T) == 2 ? 1 : 2];  // compile-time: expand convs (no residual) pay no registers
    if (RES && live) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = lane + 64 * i;
                const long m = m0 + j * 32 + (q >> 2);
                const int n = co + (q & 3) * 8;
                const bool ok = m < M && n < Cout_p;
                const uint4* rp = reinterpret_cast<const uint4*>(res + (ok ? m * Cout_p + n : 0));
#pragma unroll
                for (int e = 0; e < (sizeof(T) == 2 ? 1 : 2); ++e) rr[j][i][e] = rp[e];
            }
    }

    // ---- stage: gate rows -> LDS, then the X tile (transform applied once per element) --------------------------------
    if (XF) {  // (no gate: rows of 1.0 -- the transform below has no per-element condition)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tid + 256 * u < 2 * w_kc) gl[tid + 256 * u] = gate ? gv[u] : 1.0f;
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int p = tid + 256 * u, prow = p / PPR, pcol = p - prow * PPR;
        const bool ok = m0 + prow < M && pcol * CH < Cin_p;
        uint4 xv = xr[u];
        xv.x = ok ? xv.x : 0u;
        xv.y = ok ? xv.y : 0u;
        xv.z = ok ? xv.z : 0u;
        xv.w = ok ? xv.w : 0u;
        if (XF && pcol * CH < Cin_p) {  // x' = swish(x * gate[clip][ci]), rounded back to the MFMA input type
            T* e = reinterpret_cast<T*>(&xv);
            // the piece's gate values as whole 16-byte reads, all before the arithmetic: `if (gp) f *= gp[j]` compiled to one predicated
            // ds_read_b32 + s_waitcnt per ELEMENT (NP x CH serialised LDS round trips per thread: the gated instances ran 13-17 us longer
            // than the plain ones)
            const float* gp = gl + ((r0 + prow >= S) ? w_kc : 0) + pcol * CH;
            float gq[CH];
#pragma unroll
            for (int j = 0; j < CH; j += 4) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(gp + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) gq[j + i] = q4[i];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) gq[j] *= (float)e[j];
            if (in_swish) {  // one wave-uniform branch around the piece (inside the element loop it compiles to a select per element)
#pragma unroll
                for (int j = 0; j < CH; ++j) gq[j] = gq[j] * sigmoidf_(gq[j]);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) e[j] = (T)gq[j];
        }
        *reinterpret_cast<uint4*>(xs + (size_t)prow * KP + pcol * CH) = xv;
    }
#pragma unroll
    for (int ks = KA; ks < KS; ++ks) A[ks] = load_frag<T>(abase + (size_t)(ks < nks ? ks : nks - 1) * astep);
    __syncthreads();
    if (!live) return;

#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        if (ks >= nks) A[ks] = zero_frag<T>();  // wave-uniform; only the template steps beyond w_kc
    // scale / bias of this wave's channels: requested now, they land under the K loop
    float sc[4][4], bs[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = co + 8 * g + 4 * h;
        const bool ok = live && n < Cout_p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[g][q] = 1.0f;
            bs[g][q] = 0.0f;
        }
        if (ok && scale) load4(scale + n, sc[g]);
        if (ok && bias) load4(bias + n, bs[g]);
    }
    // ---- K loop: A from registers, B from the LDS tile -----------------------------------------------------------------
    const T* xb0 = xs + (size_t)c * KP + h * CH;
    const T* xb1 = xb0 + (size_t)32 * KP;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        // NO guard on ks: a wave-uniform branch per step makes every step its own basic block, and hipcc then moves the
        // whole accumulator between AGPRs and VGPRs around every MFMA (measured: 13 us per tile instead of < 1).  The
        // steps beyond w_kc multiply zero weights with the zero-padded tile columns.
        const frag b0 = load_frag<T>(xb0 + (size_t)ks * KSTEP);
        const frag b1 = load_frag<T>(xb1 + (size_t)ks * KSTEP);
        mma32(acc[0], A[ks], b0);
        mma32(acc[1], A[ks], b1);
    }
    // ---- epilogue: the accumulator has the position on the lane and 4 consecutive channels per quad; bounce each 32 x 32
    // tile through the wave's private fp32 image so that residual adds and stores are whole 8-channel pieces per lane
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = acc[j][4 * g + q] * sc[g][q] + bs[g][q];
            *reinterpret_cast<f32x4*>(scr + c * XT_SROW + 8 * g + 4 * h) = o;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = lane + 64 * i;
            const int row = q >> 2, cg = q & 3;
            const long m = m0 + j * 32 + row;
            const int n = co + cg * 8;
            if (m < M && n < Cout_p) {
                float v[8];
                load8(scr + row * XT_SROW + cg * 8, v);
                if (RES) {
                    float r[8];
                    raw_to_f8<T>(rr[j][i], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                act_vec(v, act);
                mask_tail(v, Cout - n);
                store8(y + m * Cout_p + n, v);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the image is reused by the second position tile
    }
}

static bool xt_pointwise(const pasn_conv_desc& d) {  // any stride: a strided 1x1x1 conv is a row gather
    return d.kt == 1 && d.kh == 1 && d.kw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0;
}
static bool xt_strided(const pasn_conv_desc& d) { return d.st != 1 || d.sh != 1 || d.sw != 1; }

// Template k-steps covering w_kc (the LDS tile is zero padded to exactly KS steps); 0 = too wide for the registers.
static int xt_ks(const pasn_conv_desc& d, int dtype) {
    const int nks = d.w_kc / (dtype == PASN_BF16 ? 16 : 8);
    if (dtype == PASN_BF16) {
        const int opts[] = {2, 4, 6, 8, 12, 14, 16, 28};
        for (int o : opts)
            if (nks <= o) return o;
        return 0;
    }
    return nks <= 16 ? 16 : nks <= 28 ? 28 : 0;
}
static size_t xt_lds(const pasn_conv_desc& d, int dtype) {
    const int ks = xt_ks(d, dtype), kstep = dtype == PASN_BF16 ? 16 : 8, ch = dtype == PASN_BF16 ? 8 : 4;
    return (size_t)XT_BM * (ks * kstep + ch) * (dtype == PASN_BF16 ? 2 : 4) + (size_t)(4 * 32 * XT_SROW + 2 * d.w_kc) * sizeof(float);
}

int pw_xtile_ks(const pasn_conv_desc& d, int dtype) { return xt_ks(d, dtype); }

bool pw_xtile_applicable(const pasn_conv_desc& d, int dtype) {
    if (const char* e = tune("PASN_NO_XTILE"))
        if (e[0] == '1') return false;
    // narrow stride-1 layers belong to pwconv.hip (weights for ALL channels in registers); strided ones have no such kernel
    const int mink = xt_strided(d) ? 16 : (tune_dev("PASN_XT_MINK") ? atoi(tune_dev("PASN_XT_MINK")) : 32);
    if (!xt_pointwise(d) || d.Cin_p < mink) return false;
    if (xt_strided(d) && d.in_swish) return false;  // narrower layers: pwconv.hip (weights for ALL channels in registers)
    const int ch = dtype == PASN_BF16 ? 8 : 4;
    if (d.w_kc % (2 * ch) != 0 || d.w_kc < d.Cin_p || d.w_rows < ((d.Cout_p + 31) / 32) * 32) return false;
    if (2 * d.w_kc > 1024) return false;       // gate staging slots
    if (xt_ks(d, dtype) == 0) return false;     // the stationary weights must fit the registers
    if ((long)d.To * d.Ho * d.Wo < XT_BM) return false;         // a tile may touch at most two clips (gate rows in LDS)
    return xt_lds(d, dtype) <= 80 * 1024;                       // two blocks per CU
}

template <typename T>
int launch_pw_xtile(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                    void* y, const pasn_conv_desc& d, hipStream_t s) {
    const int dtype = sizeof(T) == 2 ? PASN_BF16 : PASN_F32;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int S = d.To * d.Ho * d.Wo;
    const size_t lds = xt_lds(d, dtype);
    const int ntiles = ceil_div(M, XT_BM);
    const int gy = ceil_div((d.Cout_p + 31) / 32, 4);  // groups of 4 channel tiles (one per wave)
    const dim3 grid((unsigned)ntiles * gy), block(256);  // block id = position tile * gy + channel group: groups of a tile adjacent
    const bool xf = gate != nullptr || d.in_swish != 0;
    const int ks = xt_ks(d, dtype);
    XtStride geo = {xt_strided(d) ? 1 : 0, d.To, d.Ho, d.Wo, d.Ti, d.Hi, d.Wi, d.st, d.sh, d.sw};
    if (geo.on) PASN_REQUIRE(gate == nullptr, "the SE gate transform is only fused into stride-1 pointwise convs");
#define PASN_XT2(KS_, XF_, RES_)                                                                                                  \
    do {                                                                                                                    \
        PASN_MAX_LDS(96 * 1024, pwconv_xtile_kernel<T, KS_, XF_, RES_>);                                                  \
        hipLaunchKernelGGL((pwconv_xtile_kernel<T, KS_, XF_, RES_>), grid, block, lds, s, (const T*)x, (const T*)w, scale, bias, \
                           (const T*)res, gate, (T*)y, M, S, d.Cin_p, d.Cout, d.Cout_p, d.w_kc, d.act, d.in_swish, gy, d.w_frag, geo); \
    } while (0)
#define PASN_XT(KS_, XF_)                  \
    do {                                   \
        if (res) PASN_XT2(KS_, XF_, true); \
        else PASN_XT2(KS_, XF_, false);    \
    } while (0)
#define PASN_XT_KS(XF_)                            \
    if (sizeof(T) == 2) {                          \
        switch (ks) {                              \
            case 2: PASN_XT(2, XF_); break;        \
            case 4: PASN_XT(4, XF_); break;        \
            case 6: PASN_XT(6, XF_); break;        \
            case 8: PASN_XT(8, XF_); break;        \
            case 12: PASN_XT(12, XF_); break;      \
            case 14: PASN_XT(14, XF_); break;      \
            case 16: PASN_XT(16, XF_); break;      \
            default: PASN_XT(28, XF_); break;      \
        }                                          \
    } else {                                       \
        switch (ks) {                              \
            case 16: PASN_XT(16, XF_); break;      \
            default: PASN_XT(28, XF_); break;      \
        }                                          \
    }
    if (xf) {
        PASN_XT_KS(true)
    } else {
        PASN_XT_KS(false)
    }
#undef PASN_XT_KS
#undef PASN_XT
#undef PASN_XT2
    return check_launch("pwconv_xtile_kernel");
}

template int launch_pw_xtile<float>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                    const pasn_conv_desc&, hipStream_t);
template int launch_pw_xtile<__bf16>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                     const pasn_conv_desc&, hipStream_t);

}  // namespace pasn
