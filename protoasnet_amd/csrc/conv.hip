// Trunk kernels for gfx950: first-layer conv (planar clip -> channels-last), dense conv as implicit GEMM on
// MFMA, depthwise stencil with fused squeeze-excite partial sums, squeeze-excite gate, max pooling.
//
// Layout: activations are channels-last [N][T][H][W][Cp], Cp % 8 == 0, channels >= C are zero.  A wave's
// 64 lanes read 16-byte channel chunks of consecutive positions, so every global access is a whole
// number of 16-byte pieces of contiguous rows (Cp*elem bytes per position, positions contiguous).
#include "common.h"

namespace pasn {

// =================================================================================================
// first conv: planar (N,3,T,H,W) -> channels-last, window (1,kh,kw).  One thread = one output position x
// all COP output channels; the 3*kh*kw x COP weight matrix sits in LDS and is read as broadcasts.
// =================================================================================================
template <typename TIN, typename TOUT, int COP>
__global__ __launch_bounds__(256) void first_conv_kernel(const TIN* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ scale, const float* __restrict__ bias,
                                                         TOUT* __restrict__ y, pasn_conv_desc d, float in_a, float in_b) {
    // d.Cin planar input channels: 3 (the reference dataloader's clip) or 1 (a grey clip whose three channels would be identical:
    // the caller passes weights summed over the input channels).  Every loaded value goes through x' = x * in_a + in_b (the device
    // side of as_dataloader.py:180-182's normalisation; 1, 0 = none); zero padding pads the NORMALISED tensor, as the reference does.
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cin*kh*kw][COP]
    const int taps = d.Cin * d.kh * d.kw;
    for (int i = threadIdx.x; i < taps * COP; i += blockDim.x) wl[i] = w[i];
    __syncthreads();

    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int wo = (int)(m % d.Wo);
    long r = m / d.Wo;
    const int ho = (int)(r % d.Ho);
    r /= d.Ho;
    const int t = (int)(r % d.To);
    const int n = (int)(r / d.To);

    float acc[COP];
#pragma unroll
    for (int c = 0; c < COP; ++c) acc[c] = 0.0f;

    const long plane = (long)d.Hi * d.Wi;
    for (int ci = 0; ci < d.Cin; ++ci) {
        const TIN* xp = x + (((long)n * d.Cin + ci) * d.Ti + t) * plane;
        for (int kr = 0; kr < d.kh; ++kr) {
            const int hi = ho * d.sh - d.ph + kr;
            if (hi < 0 || hi >= d.Hi) continue;
            for (int ks = 0; ks < d.kw; ++ks) {
                const int wi = wo * d.sw - d.pw + ks;
                if (wi < 0 || wi >= d.Wi) continue;
                const float xv = fmaf((float)xp[(long)hi * d.Wi + wi], in_a, in_b);
                const float* wr = wl + ((ci * d.kh + kr) * d.kw + ks) * COP;
#pragma unroll
                for (int c = 0; c < COP; c += 4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
                    acc[c + 0] = fmaf(xv, wv[0], acc[c + 0]);
                    acc[c + 1] = fmaf(xv, wv[1], acc[c + 1]);
                    acc[c + 2] = fmaf(xv, wv[2], acc[c + 2]);
                    acc[c + 3] = fmaf(xv, wv[3], acc[c + 3]);
                }
            }
        }
    }
    TOUT* yp = y + m * COP;
#pragma unroll
    for (int c = 0; c < COP; c += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = acc[c + j] * scale[c + j] + bias[c + j];
        act_vec(v, d.act);
        mask_tail(v, d.Cout - c);
        store8(yp + c, v);
    }
}

template <typename TIN, typename TOUT>
static int launch_first_conv(const void* x, const float* w, const float* scale, const float* bias, void* y,
                             const pasn_conv_desc& d, float in_a, float in_b, hipStream_t s) {
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const dim3 grid(ceil_div(M, 256)), block(256);
    const size_t lds = (size_t)d.Cin * d.kh * d.kw * d.Cout_p * sizeof(float);
#define PASN_FC(COP)                                                                                          \
    case COP:                                                                                                  \
        hipLaunchKernelGGL((first_conv_kernel<TIN, TOUT, COP>), grid, block, lds, s, (const TIN*)x, w, scale, \
                           bias, (TOUT*)y, d, in_a, in_b);                                                     \
        break;
    switch (d.Cout_p) {
        PASN_FC(8)
        PASN_FC(16)
        PASN_FC(24)
        PASN_FC(32)
        PASN_FC(48)
        PASN_FC(64)
        default:
            set_error("pasn_first_conv_fwd: Cout_p must be one of 8,16,24,32,48,64");
            return PASN_ERR_UNSUPPORTED;
    }
#undef PASN_FC
    return check_launch("first_conv_kernel");
}

// =================================================================================================
// dense conv as implicit GEMM on MFMA.
//   D[co][pos] = sum_{tap, ci} W[co][tap][ci] * X[inpos(pos, tap)][ci]
// A operand = packed weights (row = output channel), B operand = activations (column = output position),
// both fetched straight from global/L2 in fragment shape: lane (r, h) reads the 16 bytes that hold k values
// k0 + h*CH .. of row/column r.  One wave owns MT x 32 positions and NT x 32 output channels; a block is 4
// waves = 4*MT*32 consecutive positions; blockIdx.y walks output-channel chunks of NT*32.
// Epilogue: the accumulator keeps the position on the lane and 4 consecutive output channels per register
// quad, so each quad becomes one 8-byte (bf16) / 16-byte (fp32) channels-last store.
// =================================================================================================
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag gate_swish(typename Traits<T>::frag f, const float* gate, bool swish) {
    constexpr int CH = Traits<T>::CH;
    typename Traits<T>::frag o;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        float v = (float)f[j];
        if (gate) v *= gate[j];
        if (swish) v = v * sigmoidf_(v);
        o[j] = (T)v;
    }
    return o;
}

template <typename T, int NT, int MT>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                          const float* __restrict__ scale, const float* __restrict__ bias,
                                                          const T* __restrict__ res, const float* __restrict__ gate,
                                                          T* __restrict__ y, pasn_conv_desc d) {
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;
    constexpr int KSTEP = Traits<T>::KSTEP;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const long mbase = ((long)blockIdx.x * 4 + wave) * (32 * MT);
    if (mbase >= M) return;  // wave-uniform
    const int co_base = blockIdx.y * (32 * NT);
    const int taps = d.kt * d.kh * d.kw;
    const int ksteps = d.w_kc / KSTEP;
    const long wrow_stride = (long)taps * d.w_kc;

    long m[MT];
    int pn[MT], pt[MT], ph[MT], pw[MT];
    bool mv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        m[mt] = mbase + mt * 32 + c;
        mv[mt] = m[mt] < M;
        const long mm = mv[mt] ? m[mt] : 0;
        pw[mt] = (int)(mm % d.Wo);
        long r = mm / d.Wo;
        ph[mt] = (int)(r % d.Ho);
        r /= d.Ho;
        pt[mt] = (int)(r % d.To);
        pn[mt] = (int)(r / d.To);
    }

    f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.0f;

    const T* wp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wp[nt] = w + (long)(co_base + nt * 32 + c) * wrow_stride + h * CH;

    const bool xform = (gate != nullptr) || (d.in_swish != 0);
    int tap = 0;
    for (int a = 0; a < d.kt; ++a) {
        for (int b = 0; b < d.kh; ++b) {
            for (int e = 0; e < d.kw; ++e, ++tap) {
                const T* xp[MT];
                const float* gp[MT];
                bool v[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int ti = pt[mt] * d.st - d.pt + a;
                    const int hi = ph[mt] * d.sh - d.ph + b;
                    const int wi = pw[mt] * d.sw - d.pw + e;
                    v[mt] = mv[mt] && ti >= 0 && ti < d.Ti && hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi;
                    const long off = (((long)pn[mt] * d.Ti + ti) * d.Hi + hi) * d.Wi + wi;
                    xp[mt] = x + (v[mt] ? off : 0) * d.Cin_p + h * CH;
                    gp[mt] = gate ? gate + (long)pn[mt] * d.Cin_p + h * CH : nullptr;
                }
                const long wtap = (long)tap * d.w_kc;
                for (int ks = 0; ks < ksteps; ++ks) {
                    const int k0 = ks * KSTEP;
                    const bool cv = (k0 + h * CH) < d.Cin_p;  // second lane half may hang over the channel pad
                    frag bf[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (v[mt] && cv) {
                            bf[mt] = load_frag<T>(xp[mt] + k0);
                            if (xform) bf[mt] = gate_swish<T>(bf[mt], gp[mt] ? gp[mt] + k0 : nullptr, d.in_swish != 0);
                        } else {
                            bf[mt] = zero_frag<T>();
                        }
                    }
                    frag af[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) af[nt] = load_frag<T>(wp[nt] + wtap + k0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) mma32(acc[nt][mt], af[nt], bf[mt]);
                }
            }
        }
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!mv[mt]) continue;
        T* yp = y + m[mt] * d.Cout_p;
        const T* rp = res ? res + m[mt] * d.Cout_p : nullptr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co_base + nt * 32 + 8 * g + 4 * h;
                if (co >= d.Cout_p) continue;
                float o[4];
                float rv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sc[4] = {1.0f, 1.0f, 1.0f, 1.0f}, bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (rp) load4(rp + co, rv);
                if (scale) load4(scale + co, sc);  // one 16-byte load per quad, not four dword gathers
                if (bias) load4(bias + co, bs);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[nt][mt][4 * g + j] * sc[j] + bs[j] + rv[j];
                act_vec(o, d.act);
                mask_tail(o, d.Cout - co);
                store4(yp + co, o);
            }
        }
    }
}

// Tile choice (also reported to the host by pasn_conv3d_variant so benchmarks can name the kernel instance).
// NT: output-channel tiles per wave (weights are padded to 128 rows so any NT <= 4 stays in bounds).
// MT: position tiles per wave.  Big position counts reuse each weight fragment twice -- but only while the
// accumulators leave room for >= 3 waves per SIMD (NT*MT*16 accumulator registers; NT=4,MT=2 would drop to one
// wave and cannot hide HBM latency).
static void conv_variant(const pasn_conv_desc& d, int& NT, int& MT) {
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int tiles = ceil_div(d.Cout_p, 32);
    NT = tiles >= 4 ? 4 : tiles;
    if (tiles > 4 && tiles % 4 != 0 && tiles % 3 == 0) NT = 3;
    MT = (M >= 256L * 1024 && NT <= 2) ? 2 : 1;
}

template <typename T>
static int launch_conv3d(const void* x, const void* w, const float* scale, const float* bias, const void* res,
                         const float* gate, void* y, const pasn_conv_desc& d, hipStream_t s) {
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int tiles = ceil_div(d.Cout_p, 32);
    int NT, MT;
    conv_variant(d, NT, MT);
    const dim3 grid(ceil_div(M, 4 * 32 * MT), ceil_div(tiles, NT)), block(256);
    PASN_REQUIRE((long)grid.y * NT * 32 <= d.w_rows, "packed weight has too few rows for the chosen tiling");
#define PASN_CV(NT_, MT_)                                                                                     \
    hipLaunchKernelGGL((conv3d_mfma_kernel<T, NT_, MT_>), grid, block, 0, s, (const T*)x, (const T*)w, scale, \
                       bias, (const T*)res, gate, (T*)y, d)
    if (MT == 2) {
        if (NT == 1) PASN_CV(1, 2); else PASN_CV(2, 2);
    } else {
        switch (NT) {
            case 1: PASN_CV(1, 1); break;
            case 2: PASN_CV(2, 1); break;
            case 3: PASN_CV(3, 1); break;
            default: PASN_CV(4, 1); break;
        }
    }
#undef PASN_CV
    return check_launch("conv3d_mfma_kernel");
}

// =================================================================================================
// depthwise stencil, channels-last.  Thread (cx, py): channel group cx (8 channels = one 16/32-byte
// chunk), position slot py; a block walks DW_POS consecutive positions of ONE clip.  Optional fused
// squeeze-excite partial sums (per block, fixed order => deterministic).
// =================================================================================================
constexpr int DW_POS = 128;  // positions per block

template <typename T>
__global__ __launch_bounds__(256) void dwconv3d_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ bias,
                                                       T* __restrict__ y, float* __restrict__ pool, pasn_conv_desc d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [taps][Cp] weights, then [PY][Cp] pool scratch
    const int taps = d.kt * d.kh * d.kw;
    const int Cp = d.Cout_p;
    float* wl = lds;
    float* red = lds + taps * Cp;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int nthreads = blockDim.x * blockDim.y;
    for (int i = tid; i < taps * Cp; i += nthreads) wl[i] = w[i];
    __syncthreads();

    const int cg = threadIdx.x;
    const bool cvalid = cg * 8 < Cp;
    const int n = blockIdx.y;
    const int S = d.To * d.Ho * d.Wo;
    const int p0 = blockIdx.x * DW_POS;
    float sc[8], bs[8], psum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = cvalid ? scale[cg * 8 + j] : 0.0f;
        bs[j] = cvalid ? bias[cg * 8 + j] : 0.0f;
        psum[j] = 0.0f;
    }
    if (cvalid) {
        for (int p = p0 + threadIdx.y; p < p0 + DW_POS && p < S; p += blockDim.y) {
            const int wo = p % d.Wo;
            int r = p / d.Wo;
            const int ho = r % d.Ho;
            const int to = r / d.Ho;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
            int tap = 0;
            for (int a = 0; a < d.kt; ++a) {
                const int ti = to * d.st - d.pt + a;
                for (int b = 0; b < d.kh; ++b) {
                    const int hi = ho * d.sh - d.ph + b;
                    for (int e = 0; e < d.kw; ++e, ++tap) {
                        const int wi = wo * d.sw - d.pw + e;
                        if (ti < 0 || ti >= d.Ti || hi < 0 || hi >= d.Hi || wi < 0 || wi >= d.Wi) continue;
                        const long off = ((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.Cin_p + cg * 8;
                        float xv[8], wv[8];
                        load8(x + off, xv);
                        load8(wl + tap * Cp + cg * 8, wv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], wv[j], acc[j]);
                    }
                }
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = acc[j] * sc[j] + bs[j];
                psum[j] += o[j];
            }
            act_vec(o, d.act);
            mask_tail(o, d.Cout - cg * 8);
            store8(y + ((long)n * S + p) * Cp + cg * 8, o);
        }
    }
    if (pool) {  // block-uniform
        if (cvalid) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[threadIdx.y * Cp + cg * 8 + j] = psum[j];
        }
        __syncthreads();
        for (int ch = tid; ch < Cp; ch += nthreads) {
            float s = 0.0f;
            for (int q = 0; q < (int)blockDim.y; ++q) s += red[q * Cp + ch];
            pool[((long)n * gridDim.x + blockIdx.x) * Cp + ch] = s;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Strip variant (the one the X3D trunk runs): a thread owns WT consecutive outputs along W of one (n, to, ho)
// row for one 8-channel group and keeps their WT x 8 fp32 accumulators in registers.  For every (kt, kh) it
// walks the (WT-1)*SW + KW input columns of that row once; each loaded 16-byte chunk is converted once and
// feeds every output whose window covers it (a register sliding window: 9*(WT+2)/WT loads per output instead
// of 27 for the 3x3x3 stride-1 stencil), and the KW x 8 weights of the row are read from LDS once per strip.
// Lanes are laid out channel-group fastest, so a wave's loads are runs of whole Cp-wide position rows.
// Block = R strips x CG channel groups (R = 256 / CG) of ONE clip; SE partial sums are reduced over the R
// strips in fixed order.
// -------------------------------------------------------------------------------------------------
template <typename T, int WT, int KW, int SW>
__global__ __launch_bounds__(256) void dwconv3d_strip_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ bias,
                                                             T* __restrict__ y, float* __restrict__ pool, pasn_conv_desc d,
                                                             int CG, int R, int strips) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [taps][Cp] weights, then [R][Cp] pool scratch
    constexpr int NC = (WT - 1) * SW + KW;                       // input columns a strip touches
    const int taps = d.kt * d.kh * KW;
    const int Cp = d.Cout_p;
    float* wl = lds;
    float* red = lds + taps * Cp;
    for (int i = threadIdx.x; i < taps * Cp; i += blockDim.x) wl[i] = w[i];
    __syncthreads();

    const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
    // XCD-aware block order (speed only): workgroups are dealt round-robin to the 8 XCDs, each with a private L2.
    // Rows (to, ho) and their (kt, kh) neighbours are read by adjacent blocks, so give every XCD one contiguous
    // run of logical blocks (whole clips) instead of every 8th block.  Bijective for any grid size.
    const int bpc = gridDim.x / d.N;  // blocks per clip
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int n = lb / bpc, bx = lb % bpc;
    const int rows_total = d.To * d.Ho * strips;
    float psum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) psum[j] = 0.0f;
    // Persistent per clip: a block stages the window weights (up to 46 KB) ONCE and walks chunks of R row strips
    // (chunk += blocks per clip), so neighbouring chunks (shared input rows) run on neighbouring blocks of one XCD and the
    // squeeze-excite partial sums shrink to one row per block (<= 32 per clip instead of one per chunk).
    const int nchunks = (rows_total + R - 1) / R;
    for (int chunk = bx; chunk < nchunks; chunk += bpc) {
        const int item = chunk * R + r;
        if (item >= rows_total) continue;
        const int strip = item % strips;
        const int ho = (item / strips) % d.Ho;
        const int to = item / (strips * d.Ho);
        const int wo0 = strip * WT;
        const int wi0 = wo0 * SW - d.pw;
        float acc[WT][8];
#pragma unroll
        for (int o = 0; o < WT; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[o][j] = 0.0f;
        // One "step" = one (kt, kh) input row of the window.  The row of step s+1 is requested (raw 16-byte words,
        // zeros where the window leaves the image) BEFORE the row of step s is converted and consumed, so a wave
        // always has a full row of loads in flight under its FMAs instead of 9 dependent load->wait->compute rounds.
        constexpr int NV = (8 * sizeof(T)) / 16;
        uint4 cur[NC][NV], nxt[NC][NV];
        bool curv = false, nxtv = false;
        // at most the first / last column of a strip can leave the image: pad <= 1, no ragged strip, <= 1 column overhang
        const bool edge1 = d.pw <= 1 && d.Wo % WT == 0 && ((d.Wo - 1) * SW + KW - 1 - d.pw - (d.Wi - 1)) <= 1;
        auto fetch = [&](int step, uint4 (&buf)[NC][NV]) -> bool {
            const int a = step / d.kh, b = step - a * d.kh;
            const int ti = to * d.st - d.pt + a, hi = ho * d.sh - d.ph + b;
            if (ti < 0 || ti >= d.Ti || hi < 0 || hi >= d.Hi) return false;
            const T* xrow = x + ((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi) * d.Cin_p + cg * 8;
            // Loads are UNCONDITIONAL (address clamped into the row) and columns outside the image are zeroed with
            // selects afterwards: `ok ? load : 0` made hipcc wrap every column in its own exec-masked branch, which
            // serialised the 9-17 loads of a row.  With `edge1` (pad <= 1 and no ragged last strip) only the first and
            // the last column of a strip can leave the image, so the interior columns carry no select at all.
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int wi = wi0 + c;
                const int wc = min(max(wi, 0), d.Wi - 1);
                const uint4* src = reinterpret_cast<const uint4*>(xrow + (long)wc * d.Cin_p);
#pragma unroll
                for (int q = 0; q < NV; ++q) buf[c][q] = src[q];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (edge1 && c != 0 && c != NC - 1) continue;
                const int wi = wi0 + c;
                const bool ok = wi >= 0 && wi < d.Wi;
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    buf[c][q].x = ok ? buf[c][q].x : 0u;
                    buf[c][q].y = ok ? buf[c][q].y : 0u;
                    buf[c][q].z = ok ? buf[c][q].z : 0u;
                    buf[c][q].w = ok ? buf[c][q].w : 0u;
                }
            }
            return true;
        };
        const int nsteps = d.kt * d.kh;
        curv = fetch(0, cur);
        for (int step = 0; step < nsteps; ++step) {
            if (step + 1 < nsteps) nxtv = fetch(step + 1, nxt);
            if (curv) {
                float wv[KW][8];
#pragma unroll
                for (int e = 0; e < KW; ++e) load8(wl + (step * KW + e) * Cp + cg * 8, wv[e]);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float xv[8];
                    raw_to_f8<T>(cur[c], xv);
#pragma unroll
                    for (int e = 0; e < KW; ++e) {
                        if ((c - e) >= 0 && (c - e) % SW == 0 && (c - e) / SW < WT) {  // resolved at compile time
                            const int o = (c - e) / SW;
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[o][j] = fmaf(xv[j], wv[e][j], acc[o][j]);
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < NV; ++q) cur[c][q] = nxt[c][q];
            curv = nxtv;
        }
        float sc[8], bs[8];
        load8(scale + cg * 8, sc);
        load8(bias + cg * 8, bs);
        T* yrow = y + ((((long)n * d.To + to) * d.Ho + ho) * d.Wo) * Cp + cg * 8;
#pragma unroll
        for (int o = 0; o < WT; ++o) {
            const int wo = wo0 + o;
            if (wo >= d.Wo) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = acc[o][j] * sc[j] + bs[j];
                psum[j] += v[j];
            }
            act_vec(v, d.act);
            mask_tail(v, d.Cout - cg * 8);
            store8(yrow + (long)wo * Cp, v);
        }
    }
    if (pool) {  // block-uniform
#pragma unroll
        for (int j = 0; j < 8; ++j) red[r * Cp + cg * 8 + j] = psum[j];
        __syncthreads();
        for (int ch = threadIdx.x; ch < Cp; ch += blockDim.x) {
            float s = 0.0f;
            for (int q = 0; q < R; ++q) s += red[q * Cp + ch];
            pool[((long)n * bpc + bx) * Cp + ch] = s;
        }
    }
}

static void dw_block_shape(int Cp, int& bx, int& by) {
    const int cgs = Cp / 8;
    bx = 1;
    while (bx < cgs) bx <<= 1;
    if (bx > 256) bx = 256;
    by = 256 / bx;
}

// Strip geometry; WT = 0 means "use the generic kernel" (window / stride outside the specialised set).
struct DwGeom {
    int WT, CG, R, strips, blocks;
};
static DwGeom dw_geom(const pasn_conv_desc& d) {
    DwGeom g = {0, d.Cout_p / 8, 0, 0, 0};
    const bool special = (d.kw == 3 && (d.sw == 1 || d.sw == 2)) || (d.kw == 1 && d.sw == 1);
    if (special && g.CG <= 256) {
        // stride 2 touches 2*WT+1 columns per row: WT = 4 keeps the double-buffered row (prefetch) within the register
        // budget (measured 370 vs 450 us on the 54-channel 112^2 layer); stride 1 prefers long strips (fewer reloads)
        g.WT = d.sw == 2 ? 4 : (d.Wo % 7 == 0) ? 7 : 8;
        if (const char* e = tune_dev("PASN_DW_WT")) {  // tuning knob: 4, 7 or 8
            const int v = atoi(e);
            if (v == 4 || v == 7 || v == 8) g.WT = v;
        }
        g.R = 256 / g.CG;
        g.strips = ceil_div(d.Wo, g.WT);
        const int nchunks = ceil_div((long)d.To * d.Ho * g.strips, g.R);
        g.blocks = nchunks < 32 ? nchunks : 32;  // blocks per clip (persistent over the chunks); also the SE partial count
    } else {
        g.blocks = ceil_div((long)d.To * d.Ho * d.Wo, DW_POS);
    }
    return g;
}

template <typename T>
static int launch_dwconv3d(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool,
                           const pasn_conv_desc& d, hipStream_t s) {
    PASN_REQUIRE(d.Cout_p / 8 <= 256, "depthwise conv supports at most 2048 channels");
    const int taps = d.kt * d.kh * d.kw;
    const DwGeom g = dw_geom(d);
    if (g.WT == 0) {
        int bx, by;
        dw_block_shape(d.Cout_p, bx, by);
        const dim3 grid(g.blocks, d.N), block(bx, by);
        const size_t lds = (size_t)(taps + by) * d.Cout_p * sizeof(float);
        PASN_REQUIRE(lds <= 64 * 1024, "depthwise conv window x channels too large for the LDS weight tile");
        hipLaunchKernelGGL((dwconv3d_kernel<T>), grid, block, lds, s, (const T*)x, w, scale, bias, (T*)y, pool, d);
        return check_launch("dwconv3d_kernel");
    }
    const dim3 grid(g.blocks * d.N), block(g.CG * g.R);
    const size_t lds = (size_t)(taps + g.R) * d.Cout_p * sizeof(float);
    PASN_REQUIRE(lds <= 64 * 1024, "depthwise conv window x channels too large for the LDS weight tile");
#define PASN_DW(WT_, KW_, SW_)                                                                                       \
    hipLaunchKernelGGL((dwconv3d_strip_kernel<T, WT_, KW_, SW_>), grid, block, lds, s, (const T*)x, w, scale, bias, \
                       (T*)y, pool, d, g.CG, g.R, g.strips)
#define PASN_DW_WT(KW_, SW_)                   \
    switch (g.WT) {                            \
        case 4: PASN_DW(4, KW_, SW_); break;   \
        case 7: PASN_DW(7, KW_, SW_); break;   \
        default: PASN_DW(8, KW_, SW_); break;  \
    }
    if (d.kw == 1) {
        PASN_DW_WT(1, 1)
    } else if (d.sw == 1) {
        PASN_DW_WT(3, 1)
    } else {
        PASN_DW_WT(3, 2)
    }
#undef PASN_DW_WT
#undef PASN_DW
    return check_launch("dwconv3d_strip_kernel");
}

// =================================================================================================
// squeeze-excite gate: one block per clip.
// =================================================================================================
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ pool, int pool_blocks, float inv_positions,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      float* __restrict__ gate, int C, int Cp, int Cse) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [Cp] mean, [Cse] hidden, [4][Cp] partial sums
    float* mean = sm;
    float* hid = sm + Cp;
    float* part = sm + Cp + Cse;
    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // mean over positions: wave q-slices the block partials (lanes along channels: coalesced), fixed order throughout
    for (int ch = lane; ch < Cp; ch += 64) {
        float s = 0.0f;
        const float* pp = pool + (long)n * pool_blocks * Cp + ch;
        // 8 independent partial sums: 8 loads in flight per lane instead of one dependent chain (this tiny kernel is
        // pure latency: ~44 sequential L2 round trips per lane cost 27 us per launch); the order stays fixed
        float p8[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int q = wave;
        for (; q + 28 < pool_blocks; q += 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) p8[k] += pp[(long)(q + 4 * k) * Cp];
        }
        for (int k = 0; q < pool_blocks; q += 4, ++k) p8[k] += pp[(long)q * Cp];
        s = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
        part[wave * Cp + ch] = s;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < Cp; ch += blockDim.x)
        mean[ch] = ((part[ch] + part[Cp + ch]) + (part[2 * Cp + ch] + part[3 * Cp + ch])) * inv_positions;
    __syncthreads();
    // fc1 + ReLU: a wave per hidden unit, lanes along the C-long dot product.  Eight units per wave are computed together
    // and the C loop is unrolled so all of a wave's weight loads are independent and in flight at once: this kernel is pure
    // latency (one block per clip), and the rolled version paid Cse/4 dependent L2 round trips (36 us at Cse = 32).
    constexpr int JU = 8;
    for (int j0 = wave; j0 < Cse; j0 += 4 * JU) {
        float s[JU];
#pragma unroll
        for (int u = 0; u < JU; ++u) s[u] = 0.0f;
        for (int c0 = 0; c0 < C; c0 += 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = c0 + lane + 64 * k;
                const bool ok = ch < C;
                const float m = ok ? mean[ch] : 0.0f;
#pragma unroll
                for (int u = 0; u < JU; ++u) {
                    const int j = j0 + 4 * u;
                    const bool use = ok && j < Cse;  // unconditional load from a clamped index, select after (no branches)
                    const float wv = w1[use ? (long)j * C + ch : 0];
                    s[u] = fmaf(use ? wv : 0.0f, m, s[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < JU; ++u) {
            float t = s[u];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
            const int j = j0 + 4 * u;
            if (lane == 0 && j < Cse) hid[j] = fmaxf(t + b1[j], 0.0f);
        }
    }
    __syncthreads();
    // fc2 + sigmoid: a thread per channel; its Cse-long weight row is contiguous -> independent 16-byte loads
    for (int ch = threadIdx.x; ch < Cp; ch += blockDim.x) {
        float g = 0.0f;
        if (ch < C) {
            float s = b2[ch];
            const float* wr = w2 + (long)ch * Cse;
            if ((Cse & 3) == 0) {
                for (int j0 = 0; j0 < Cse; j0 += 32) {
                    f32x4 wv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        wv[u] = *reinterpret_cast<const f32x4*>(wr + (j0 + 4 * u < Cse ? j0 + 4 * u : 0));
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + 4 * u < Cse) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) s = fmaf(wv[u][e], hid[j0 + 4 * u + e], s);
                        }
                }
            } else {
                for (int j = 0; j < Cse; ++j) s = fmaf(wr[j], hid[j], s);
            }
            g = sigmoidf_(s);
        }
        gate[(long)n * Cp + ch] = g;
    }
}

// =================================================================================================
// max pooling, channels-last, 8-channel groups.
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void maxpool3d_kernel(const T* __restrict__ x, T* __restrict__ y, pasn_conv_desc d) {
    const int cgs = d.Cout_p / 8;
    const long total = (long)d.N * d.To * d.Ho * d.Wo * cgs;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % cgs);
    long m = idx / cgs;
    const int wo = (int)(m % d.Wo);
    long r = m / d.Wo;
    const int ho = (int)(r % d.Ho);
    r /= d.Ho;
    const int to = (int)(r % d.To);
    const int n = (int)(r / d.To);
    float best[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) best[j] = -INFINITY;
    for (int a = 0; a < d.kt; ++a) {
        const int ti = to * d.st - d.pt + a;
        if (ti < 0 || ti >= d.Ti) continue;
        for (int b = 0; b < d.kh; ++b) {
            const int hi = ho * d.sh - d.ph + b;
            if (hi < 0 || hi >= d.Hi) continue;
            for (int e = 0; e < d.kw; ++e) {
                const int wi = wo * d.sw - d.pw + e;
                if (wi < 0 || wi >= d.Wi) continue;
                float v[8];
                load8(x + ((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.Cin_p + cg * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) best[j] = fmaxf(best[j], v[j]);
            }
        }
    }
    store8(y + m * d.Cout_p + cg * 8, best);
}

}  // namespace pasn

using namespace pasn;

static bool conv_desc_ok(const pasn_conv_desc* d) {
    return d && d->N > 0 && d->Ti > 0 && d->Hi > 0 && d->Wi > 0 && d->To > 0 && d->Ho > 0 && d->Wo > 0 && d->Cin > 0 &&
           d->Cout > 0 && d->Cin_p >= d->Cin && d->Cout_p >= d->Cout && d->Cin_p % 8 == 0 && d->Cout_p % 8 == 0 && d->kt > 0 &&
           d->kh > 0 && d->kw > 0 && d->st > 0 && d->sh > 0 && d->sw > 0;
}

// Which pointwise kernel: the persistent register-resident one (pwconv.hip) wins on gated / swish-input layers and on a
// single channel tile; with >= 2 channel tiles and no input transform the X-tile kernel is faster (measured on the X3D-S
// stage-3 layers: 48->108 29 vs 37 us, 108->48 33 vs 36.5 us).
static bool prefer_xtile(const pasn_conv_desc& d, int dtype, bool has_gate) {
    if (!pw_xtile_applicable(d, dtype)) return false;
    if (has_gate || d.in_swish) {  // PASN_XTILE_GATED=1: the X-tile kernel on gated layers too.  Re-measured after its gate reads became whole
        // pieces (round 2): still behind the persistent kernel on the three gated 108 -> 48 layers (9.00 k vs 9.04 k clips/s end to end)
        const char* e = tune("PASN_XTILE_GATED");
        if (!(e && e[0] == '1')) return false;
    }
    if (d.st != 1 || d.sh != 1 || d.sw != 1) return true;  // strided 1x1x1 (shortcut convs): the only specialised kernel
    return (d.Cout_p + 31) / 32 >= 2;
}

static int first_conv_dispatch(const void* x, const float* w, const float* scale, const float* bias, void* y, const pasn_conv_desc* d,
                               int in_dtype, int out_dtype, float in_a, float in_b, void* stream) {
    PASN_REQUIRE(x && w && scale && bias && y && d, "null pointer");
    PASN_REQUIRE(d->N > 0 && (d->Cin == 3 || d->Cin == 1) && d->kt == 1 && d->st == 1 && d->pt == 0 && d->To == d->Ti,
                 "first conv is (1,kh,kw) over 3 (or 1, grey) planar channels");
    PASN_REQUIRE(d->Cout_p % 8 == 0 && d->Cout_p >= d->Cout, "Cout_p must be a multiple of 8");
    PASN_REQUIRE((size_t)d->Cin * d->kh * d->kw * d->Cout_p * 4 <= 64 * 1024, "first conv weights exceed the LDS tile");
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == PASN_F32 && out_dtype == PASN_F32) return launch_first_conv<float, float>(x, w, scale, bias, y, *d, in_a, in_b, s);
    if (in_dtype == PASN_F32 && out_dtype == PASN_BF16) return launch_first_conv<float, __bf16>(x, w, scale, bias, y, *d, in_a, in_b, s);
    if (in_dtype == PASN_BF16 && out_dtype == PASN_BF16) return launch_first_conv<__bf16, __bf16>(x, w, scale, bias, y, *d, in_a, in_b, s);
    if (in_dtype == PASN_BF16 && out_dtype == PASN_F32) return launch_first_conv<__bf16, float>(x, w, scale, bias, y, *d, in_a, in_b, s);
    if (in_dtype == PASN_U8 && out_dtype == PASN_F32) return launch_first_conv<unsigned char, float>(x, w, scale, bias, y, *d, in_a, in_b, s);
    if (in_dtype == PASN_U8 && out_dtype == PASN_BF16) return launch_first_conv<unsigned char, __bf16>(x, w, scale, bias, y, *d, in_a, in_b, s);
    set_error("pasn_first_conv_fwd: unknown dtype");
    return PASN_ERR_ARG;
}

extern "C" int pasn_first_conv_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y,
                                   const pasn_conv_desc* d, int in_dtype, int out_dtype, void* stream) {
    PASN_REQUIRE(d && d->Cin == 3, "pasn_first_conv_fwd reads 3 planar channels (pasn_first_conv_gray_fwd reads one)");
    return first_conv_dispatch(x, w, scale, bias, y, d, in_dtype, out_dtype, 1.0f, 0.0f, stream);
}

extern "C" int pasn_first_conv_mfma_slot(const pasn_conv_desc* d, int in_dtype, int out_dtype) {
    if (!d || (in_dtype != PASN_F32 && in_dtype != PASN_BF16 && in_dtype != PASN_U8)) return -1;
    return first_conv_mfma_slot(*d, out_dtype);
}

extern "C" int pasn_first_conv_mfma_fwd(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc* d,
                                        int in_dtype, float in_a, float in_b, void* stream) {
    PASN_REQUIRE(x && wq && scale && bias && y && d, "null argument");
    const int o = first_conv_mfma_slot(*d, PASN_BF16);
    PASN_REQUIRE(o >= 0, "pasn_first_conv_mfma_fwd: layer not covered (ask pasn_first_conv_mfma_slot)");
    return launch_first_conv_mfma(x, wq, scale, bias, y, *d, in_dtype, in_a, in_b, o, (hipStream_t)stream);
}

extern "C" int pasn_first_conv_gray_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y,
                                        const pasn_conv_desc* d, int in_dtype, int out_dtype, float in_a, float in_b, void* stream) {
    PASN_REQUIRE(d && d->Cin == 1, "pasn_first_conv_gray_fwd reads ONE planar channel");
    return first_conv_dispatch(x, w, scale, bias, y, d, in_dtype, out_dtype, in_a, in_b, stream);
}

extern "C" int pasn_conv3d_fwd(const void* x, const void* w, const float* scale, const float* bias, const void* residual,
                               const float* gate, void* y, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(x && w && y, "null pointer");  // scale may be NULL (= 1), bias may be NULL (= 0)
    PASN_REQUIRE(conv_desc_ok(d), "bad geometry (channel strides must be multiples of 8)");
    const int kstep = dtype == PASN_BF16 ? 16 : 8;
    PASN_REQUIRE(d->w_kc >= d->Cin_p && d->w_kc % kstep == 0, "w_kc must cover Cin_p and be a multiple of the MFMA k-step");
    PASN_REQUIRE(d->w_rows % 128 == 0 && d->w_rows >= d->Cout_p, "w_rows must be a multiple of 128 covering Cout_p");
    hipStream_t s = (hipStream_t)stream;
    PASN_REQUIRE(dtype == PASN_F32 || dtype == PASN_BF16, "unknown dtype");
    if (const TcGeom tg = tconv_geom(*d, dtype, gate != nullptr); tg.ok)  // temporal (3,1,1) conv, weight-stationary + T-marching (bf16, fragment-major weights)
        return launch_tconv_ws(x, w, scale, bias, residual, y, *d, tg, s);
    if (const WsGeom wg = pw_ws_geom(*d, dtype, gate != nullptr, residual != nullptr); wg.ok)  // weight-stationary persistent blocks (bf16, fragment-major weights)
        return launch_pw_ws(x, w, scale, bias, residual, gate, y, *d, wg, s);
    if (pw_tiny_applicable(*d, dtype, gate != nullptr))  // fp32, few positions (the image heads): one wave per 32 x 32 output tile
        return launch_pw_tiny(x, w, scale, bias, residual, y, *d, dtype, s);
    const bool xt_first = prefer_xtile(*d, dtype, gate != nullptr);
    const PwGeom pg = xt_first ? PwGeom{0, 0, 0, 0} : pw_geom(*d, dtype);  // 1x1x1 stride-1 convs: the row-streaming kernel
    PASN_REQUIRE(d->w_frag == 0 || (pw_xtile_applicable(*d, dtype) && !pg.TM),
                 "fragment-major weights are only read by the pwconv_xtile kernel (variant >= 2500)");
    if (pg.TM) {
        if (dtype == PASN_F32) return launch_pwconv<float>(x, w, scale, bias, residual, gate, y, *d, pg, s);
        return launch_pwconv<__bf16>(x, w, scale, bias, residual, gate, y, *d, pg, s);
    }
    if (pw_xtile_applicable(*d, dtype)) {  // wide pointwise layers: whole-K X tiles in LDS, weights streamed from L2
        if (dtype == PASN_F32) return launch_pw_xtile<float>(x, w, scale, bias, residual, gate, y, *d, s);
        return launch_pw_xtile<__bf16>(x, w, scale, bias, residual, gate, y, *d, s);
    }
    if (const int nt = (gate == nullptr && d->w_frag == 0) ? igemm_nt(*d, dtype) : 0)  // windowed dense convs, bf16: direct-to-LDS implicit GEMM
        return launch_igemm(x, w, scale, bias, residual, y, *d, nt, s);
    if (gemm_pw_applicable(*d, dtype)) {  // large K / N pointwise: LDS-tiled GEMM
        if (dtype == PASN_F32) return launch_gemm_pw<float>(x, w, scale, bias, residual, gate, y, *d, s);
        return launch_gemm_pw<__bf16>(x, w, scale, bias, residual, gate, y, *d, s);
    }
    if (dtype == PASN_F32) return launch_conv3d<float>(x, w, scale, bias, residual, gate, y, *d, s);
    return launch_conv3d<__bf16>(x, w, scale, bias, residual, gate, y, *d, s);
}

// 0 = not covered, 1 = pwconv_xpair_kernel (one block per 64-position tile), 2 = the weight-stationary persistent kernel in pair mode.
// flags: bit 0 = a gate tensor (or the squeeze-excite operands) will be passed.
extern "C" int pasn_conv3d_pair_variant(const pasn_conv_desc* d1, const pasn_conv_desc* d2, int dtype, int flags) {
    if (!conv_desc_ok(d1) || !conv_desc_ok(d2)) return 0;
    pasn_conv_desc f1 = *d1, f2 = *d2;
    f1.w_frag = f2.w_frag = 1;
    if (pw_ws_geom(f1, dtype, (flags & 1) != 0, true, false, &f2).ok) return 2;
    return pw_xpair_ks(*d1, *d2, dtype, nullptr) != 0 ? 1 : 0;
}

// The chained pair with the first conv's squeeze-excite gate computed in the launch's prologue (pasn_conv3d_se_fwd + pasn_conv3d_pair_fwd in one).
extern "C" int pasn_conv3d_pair_se_supported(const pasn_conv_desc* d1, const pasn_conv_desc* d2, int dtype, int Cse) {
    if (!conv_desc_ok(d1) || !conv_desc_ok(d2) || Cse <= 0 || Cse > 32 || Cse % 4 != 0 || d1->Cin > 512) return 0;
    if (const char* e = tune("PASN_NO_SE_PROLOGUE"))
        if (e[0] == '1') return 0;
    pasn_conv_desc f1 = *d1, f2 = *d2;
    f1.w_frag = f2.w_frag = 1;
    return pw_ws_geom(f1, dtype, true, true, true, &f2).ok;
}

extern "C" int pasn_conv3d_pair_se_fwd(const void* x, const void* w1, const float* scale1, const float* bias1, const void* residual,
                                       const float* pool_partial, int pool_blocks, int positions, const float* fc1_w, const float* fc1_b,
                                       const float* fc2_w, const float* fc2_b, int Cse, void* y1, const pasn_conv_desc* d1, const void* w2,
                                       const float* scale2, const float* bias2, void* y2, const pasn_conv_desc* d2, int dtype, void* stream) {
    PASN_REQUIRE(x && w1 && w2 && y1 && y2 && residual && pool_partial && fc1_w && fc1_b && fc2_w && fc2_b, "null pointer");
    PASN_REQUIRE(conv_desc_ok(d1) && conv_desc_ok(d2) && pool_blocks > 0 && positions > 0, "bad geometry");
    PASN_REQUIRE(dtype == PASN_BF16 && d1->w_frag == 1 && d2->w_frag == 1, "bf16 with fragment-major weights only");
    PASN_REQUIRE(pasn_conv3d_pair_se_supported(d1, d2, dtype, Cse), "pair not covered (pasn_conv3d_pair_se_supported returns 0)");
    const WsGeom wg = pw_ws_geom(*d1, dtype, true, true, true, d2);
    const WsSe se = {pool_partial, pool_blocks, 1.0f / (float)positions, fc1_w, fc1_b, fc2_w, fc2_b, d1->Cin, Cse};
    const WsPair p2 = {(const __bf16*)w2, scale2, bias2, (__bf16*)y2, d2->Cout, d2->Cout_p, d2->w_kc / 16, d2->act};
    return launch_pw_ws(x, w1, scale1, bias1, residual, nullptr, y1, *d1, wg, (hipStream_t)stream, &se, &p2);
}

extern "C" int pasn_conv3d_pair_supported(const pasn_conv_desc* d1, const pasn_conv_desc* d2, int dtype) {
    return pasn_conv3d_pair_variant(d1, d2, dtype, 0) != 0 || pasn_conv3d_pair_variant(d1, d2, dtype, 1) != 0;
}

extern "C" int pasn_conv3d_pair_fwd(const void* x, const void* w1, const float* scale1, const float* bias1, const void* residual,
                                    const float* gate, void* y1, const pasn_conv_desc* d1, const void* w2, const float* scale2,
                                    const float* bias2, void* y2, const pasn_conv_desc* d2, int dtype, void* stream) {
    PASN_REQUIRE(x && w1 && w2 && y1 && y2 && residual, "null pointer");
    PASN_REQUIRE(conv_desc_ok(d1) && conv_desc_ok(d2), "bad geometry");
    PASN_REQUIRE(dtype == PASN_BF16 && d1->w_frag == 1 && d2->w_frag == 1, "bf16 with fragment-major weights only");
    if (const WsGeom wg = pw_ws_geom(*d1, dtype, gate != nullptr, true, false, d2); wg.ok) {  // persistent blocks, both weight sets in registers
        const WsPair p2 = {(const __bf16*)w2, scale2, bias2, (__bf16*)y2, d2->Cout, d2->Cout_p, d2->w_kc / 16, d2->act};
        return launch_pw_ws(x, w1, scale1, bias1, residual, gate, y1, *d1, wg, (hipStream_t)stream, nullptr, &p2);
    }
    PASN_REQUIRE(pw_xpair_ks(*d1, *d2, dtype, nullptr) != 0, "pair not covered (pasn_conv3d_pair_supported returns 0)");
    return launch_pw_xpair(x, w1, scale1, bias1, residual, gate, y1, *d1, w2, scale2, bias2, y2, *d2, (hipStream_t)stream);
}

extern "C" int pasn_conv3d_se_supported(const pasn_conv_desc* d, int dtype, int Cse, int has_residual) {
    if (!conv_desc_ok(d) || Cse <= 0 || Cse > 32 || Cse % 4 != 0 || d->Cin > 512) return 0;
    if (const char* e = tune("PASN_NO_SE_PROLOGUE"))
        if (e[0] == '1') return 0;
    pasn_conv_desc df = *d;
    df.w_frag = 1;
    return pw_ws_geom(df, dtype, true, has_residual != 0, true).ok;
}

extern "C" int pasn_conv3d_se_fwd(const void* x, const void* w, const float* scale, const float* bias, const void* residual,
                                  const float* pool_partial, int pool_blocks, int positions, const float* fc1_w, const float* fc1_b,
                                  const float* fc2_w, const float* fc2_b, int Cse, void* y, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(x && w && y && pool_partial && fc1_w && fc1_b && fc2_w && fc2_b, "null pointer");
    PASN_REQUIRE(conv_desc_ok(d) && pool_blocks > 0 && positions > 0, "bad geometry");
    PASN_REQUIRE(d->w_frag == 1 && pasn_conv3d_se_supported(d, dtype, Cse, residual != nullptr), "layer not covered (pasn_conv3d_se_supported returns 0)");
    const WsGeom wg = pw_ws_geom(*d, dtype, true, residual != nullptr, true);
    const WsSe se = {pool_partial, pool_blocks, 1.0f / (float)positions, fc1_w, fc1_b, fc2_w, fc2_b, d->Cin, Cse};
    return launch_pw_ws(x, w, scale, bias, residual, nullptr, y, *d, wg, (hipStream_t)stream, &se);
}

extern "C" int pasn_conv3d_short_supported(const pasn_conv_desc* d, const pasn_conv_desc* d2, int dtype) {
    if (!conv_desc_ok(d) || !conv_desc_ok(d2)) return 0;
    return pw_short_ks2(*d, *d2, dtype) != 0;
}

extern "C" int pasn_conv3d_short_fwd(const void* x, const void* w, const float* scale, const float* bias, const float* gate, const void* x2,
                                     const void* w2, const float* scale2, void* y, const pasn_conv_desc* d, const pasn_conv_desc* d2,
                                     int dtype, void* stream) {
    PASN_REQUIRE(x && w && x2 && w2 && y, "null pointer");
    PASN_REQUIRE(conv_desc_ok(d) && conv_desc_ok(d2), "bad geometry");
    PASN_REQUIRE(pw_short_ks2(*d, *d2, dtype) != 0, "layer pair not covered (pasn_conv3d_short_supported returns 0)");
    PASN_REQUIRE(d->w_frag == 0, "row-major weights");
    const PwShort sc = {x2, w2, scale2, d2->Cin_p, d2->w_kc, d2->Ho, d2->Wo, d2->Hi, d2->Wi, d2->sh, d2->sw};
    return launch_pwconv<__bf16>(x, w, scale, bias, nullptr, gate, y, *d, pw_geom(*d, dtype), (hipStream_t)stream, &sc);
}

extern "C" int pasn_conv3d_variant(const pasn_conv_desc* d, int dtype, int flags) {
    if (!conv_desc_ok(d)) return 0;
    const int has_gate = flags & 1, has_res = (flags >> 1) & 1;  // bit 0: an SE gate tensor is passed, bit 1: a residual is passed
    {  // asked before the host has packed the weights: the weight-stationary kernel reads them fragment-major like the X-tile kernel
        pasn_conv_desc df = *d;
        df.w_frag = 1;
        if (const TcGeom tg = tconv_geom(df, dtype, has_gate != 0); tg.ok) return 9000 + tg.KSF;  // tconv_ws_kernel<KSF, residual?> (fragment-major weights)
        if (const int v = pw_ws_variant(df, dtype, has_gate != 0, has_res != 0)) return v;
    }
    if (pw_tiny_applicable(*d, dtype, has_gate != 0)) return 2002;  // pwconv_tiny_f32_kernel
    const PwGeom pg = prefer_xtile(*d, dtype, has_gate != 0) ? PwGeom{0, 0, 0, 0} : pw_geom(*d, dtype);
    if (pg.TM) return 1000 + pg.TM * 10 + pg.xrow;  // pwconv_persist_kernel<dtype, KS, NT>
    if (pw_xtile_applicable(*d, dtype))               // pwconv_xtile_kernel<dtype, input transform?>
        return 2500 + 2 * pw_xtile_ks(*d, dtype) + ((d->in_swish != 0) ? 1 : 0);
    if (const int nt = has_gate ? 0 : igemm_nt(*d, dtype)) return 6000 + nt;  // igemm_glds_kernel<NT>
    if (gemm_pw_applicable(*d, dtype))                // gemm_conv_kernel<dtype, pointwise?>
        return 2000 + ((d->kt * d->kh * d->kw == 1 && d->st * d->sh * d->sw == 1) ? 0 : 1);
    int NT, MT;
    conv_variant(*d, NT, MT);
    return NT * 10 + MT;
}

extern "C" int pasn_dwconv3d_variant(const pasn_conv_desc* d, int dtype) {
    if (!d || d->Cout_p <= 0 || d->Cout_p % 8 != 0) return 0;
    if (dw_temporal_applicable(*d, dtype)) return 70000 + d->kt;  // dwconv_t_kernel<dtype, KT> (without pool partial rows; with them the generic kernels)
    if (dw_tz_geom(*d, dtype).ok) return 60001;  // dwconv3d_tz_kernel (stride 1, planes 9 .. 14 wide)
    if (const DwMfmaGeom mf = dw_mfma_geom(*d, dtype); mf.ok) return 50001;  // dwconv3d_mfma_kernel (stride 1)
    const DwMarchGeom m = dw_march_geom(*d, dtype);
    if (m.WT) return 3000 + m.WT * 10 + d->sw;  // dwconv3d_march_kernel<SW, WT>
    const DwGeom g = dw_geom(*d);
    return g.WT ? g.WT * 100 + d->kw * 10 + d->sw : 0;
}

extern "C" int pasn_dwconv3d_pool_blocks(const pasn_conv_desc* d, int dtype) {
    if (!d || d->Cout_p <= 0 || d->Cout_p % 8 != 0) return 0;
    if (const DtGeom tz = dw_tz_geom(*d, dtype); tz.ok) return tz.nT;
    const DwMfmaGeom mf = dw_mfma_geom(*d, dtype);
    if (mf.ok) return mf.chunks;
    const DwMarchGeom m = dw_march_geom(*d, dtype);
    if (m.WT) return m.bpc;
    return dw_geom(*d).blocks;
}

// Partial rows written by pasn_dwconv3d_se_fwd (always the T-marching VALU stencil, whatever pasn_dwconv3d_fwd would take for the layer).
extern "C" int pasn_dwconv3d_se_pool_blocks(const pasn_conv_desc* d, int dtype) {
    if (!d || d->Cout_p <= 0 || d->Cout_p % 8 != 0) return 0;
    return dw_march_geom(*d, dtype).bpc;
}

extern "C" int pasn_dwconv3d_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y,
                                 float* pool_partial, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(x && w && scale && bias && y, "null pointer");
    PASN_REQUIRE(conv_desc_ok(d), "bad geometry (channel strides must be multiples of 8)");
    PASN_REQUIRE(d->Cin == d->Cout && d->Cin_p == d->Cout_p, "depthwise conv keeps the channel count");
    hipStream_t s = (hipStream_t)stream;
    if (!pool_partial && dw_temporal_applicable(*d, dtype)) return launch_dw_temporal(x, w, scale, bias, y, *d, dtype, s);  // (kt,1,1): T-marching register ring
    if (const DtGeom tz = dw_tz_geom(*d, dtype); tz.ok) return launch_dw_tz(x, w, scale, bias, y, pool_partial, *d, tz, s);
    const DwMfmaGeom mf = dw_mfma_geom(*d, dtype);
    if (mf.ok) return launch_dw_mfma(x, w, scale, bias, y, pool_partial, *d, mf, s);
    const DwMarchGeom m = dw_march_geom(*d, dtype);
    if (m.WT) return launch_dw_march(x, w, scale, bias, y, pool_partial, *d, m, s);
    if (dtype == PASN_F32) return launch_dwconv3d<float>(x, w, scale, bias, y, pool_partial, *d, s);
    if (dtype == PASN_BF16) return launch_dwconv3d<__bf16>(x, w, scale, bias, y, pool_partial, *d, s);
    set_error("pasn_dwconv3d_fwd: unknown dtype");
    return PASN_ERR_ARG;
}

// First block of an X3D stage, front half: expand conv + BN + ReLU -> stride-(1,2,2) depthwise stencil + BN (+ act, + SE pool partial rows)
// in ONE launch (x3d_expdw.hip).  de = the 1x1x1 conv (fragment-major weights), d = the depthwise conv on its output.
extern "C" int pasn_x3d_expdw_supported(const pasn_conv_desc* de, const pasn_conv_desc* d, int dtype) {
    if (!conv_desc_ok(de) || !conv_desc_ok(d)) return 0;
    pasn_conv_desc f = *de;
    f.w_frag = 1;
    return xe_geom(f, *d, dtype).ok;
}
extern "C" int pasn_x3d_expdw_pool_blocks(const pasn_conv_desc* de, const pasn_conv_desc* d, int dtype) {
    if (!conv_desc_ok(de) || !conv_desc_ok(d)) return 0;
    pasn_conv_desc f = *de;
    f.w_frag = 1;
    const XeGeom g = xe_geom(f, *d, dtype);
    return g.ok ? g.chunks : 0;
}
extern "C" int pasn_x3d_expdw_variant(const pasn_conv_desc* de, const pasn_conv_desc* d, int dtype) {
    if (!conv_desc_ok(de) || !conv_desc_ok(d)) return -1;
    pasn_conv_desc f = *de;
    f.w_frag = 1;
    const XeGeom g = xe_geom(f, *d, dtype);
    return g.ok ? (g.tz ? 1 : 0) : -1;
}
extern "C" int pasn_x3d_expdw_fwd(const void* x, const void* wa, const float* scale_a, const float* bias_a, const float* w, const float* scale,
                                  const float* bias, void* y, float* pool_partial, const pasn_conv_desc* de, const pasn_conv_desc* d, int dtype,
                                  void* stream) {
    PASN_REQUIRE(x && wa && bias_a && w && scale && bias && y, "null pointer");  // scale_a may be NULL: norm_a's scale folded into wa by the caller
    PASN_REQUIRE(conv_desc_ok(de) && conv_desc_ok(d), "bad geometry (channel strides must be multiples of 8)");
    PASN_REQUIRE(dtype == PASN_BF16 && de->w_frag == 1, "bf16 with fragment-major expand weights only");
    const XeGeom g = xe_geom(*de, *d, dtype);
    PASN_REQUIRE(g.ok, "layer pair not covered (pasn_x3d_expdw_supported returns 0)");
    return launch_x3d_expdw(x, wa, scale_a, bias_a, w, scale, bias, y, pool_partial, *de, *d, g, (hipStream_t)stream);
}

// Depthwise stencil + squeeze-excite gate in ONE launch (the clip's last-arriving block computes the gate); only where the T-marching
// stencil covers the layer -- pasn_dwconv3d_se_supported says so, the caller otherwise issues pasn_dwconv3d_fwd + pasn_se_gate_fwd.
extern "C" int pasn_dwconv3d_se_supported(const pasn_conv_desc* d, int dtype, int Cse) {
    if (!d || d->Cout_p <= 0 || d->Cout_p % 8 != 0 || Cse <= 0) return 0;
    if (const char* e = tune("PASN_NO_SE_FUSE"))
        if (e[0] == '1') return 0;
    if (dw_mfma_geom(*d, dtype).ok) return 0;    // the matrix-core stencil + the stand-alone gate beat the fused VALU launch (8669 vs 8623 clips/s)
    const DwMarchGeom m = dw_march_geom(*d, dtype);
    // The last block of a clip computes the gate alone, at the END of the launch: atomic + acquire, the partial rows (agent-scope loads:
    // memory-side round trips), two FCs -- an exposed tail of 5-30 us that grows with the channel count (432 channels: +29 us per launch,
    // 216: +4..9 us) against ~9 us for the stand-alone gate launch it replaces.  Measured end to end (32 x 16 x 224 x 224, one box): fused
    // everywhere 8084 clips/s, fused up to 256 channels 8218, up to 128 channels 8250, nowhere 8193.  A version of the tail with every
    // load batched up front (fc rows in registers) moved 432 channels to +16 us and the narrow stages to +6..9 us: no better.
    const int max_c = tune("PASN_SE_FUSE_MAXC") ? atoi(tune("PASN_SE_FUSE_MAXC")) : 128;
    return m.WT != 0 && d->Cout_p <= max_c && m.R * d->Cout_p >= d->Cout_p + Cse + 8;  // the gate's LDS scratch is the pool scratch
}

extern "C" int pasn_dwconv3d_se_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool_partial,
                                    const pasn_conv_desc* d, int dtype, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                                    const float* fc2_b, int Cse, float* gate, int32_t* counter, void* stream) {
    PASN_REQUIRE(x && w && scale && bias && y && pool_partial && fc1_w && fc1_b && fc2_w && fc2_b && gate && counter, "null pointer");
    PASN_REQUIRE(conv_desc_ok(d) && d->Cin == d->Cout && d->Cin_p == d->Cout_p, "depthwise conv keeps the channel count");
    PASN_REQUIRE(pasn_dwconv3d_se_supported(d, dtype, Cse), "layer not covered by the fused stencil + gate launch");
    const DwMarchGeom m = dw_march_geom(*d, dtype);
    const DwSeArgs se = {fc1_w, fc1_b, fc2_w, fc2_b, gate, counter, Cse};
    return launch_dw_march(x, w, scale, bias, y, pool_partial, *d, m, (hipStream_t)stream, se);
}

// Fast path of the gate (C <= 512, Cse <= 32, Cse % 4 == 0 -- every X3D width): the kernel is three dependent round trips
// (pool partials -> fc1 weights -> fc2 weights) of a few KB each, one block per clip.  Here BOTH weight matrices are
// requested into registers at kernel entry, before the pool reduction, so the three trips overlap; arithmetic and
// summation order are those of se_gate_kernel (bit-identical gates).
__global__ __launch_bounds__(256) void se_gate_fast_kernel(const float* __restrict__ pool, int pool_blocks, float inv_positions,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           float* __restrict__ gate, int C, int Cp, int Cse) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [Cp] mean, [Cse] hidden, [4][Cp] partial sums
    float* mean = sm;
    float* hid = sm + Cp;
    float* part = sm + Cp + Cse;
    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int JU = 8;
    // fc1 weights of this wave's hidden units j = wave + 4u: rows j, columns lane + 64 k (k < 8 covers C <= 512)
    float w1r[JU][8];
#pragma unroll
    for (int u = 0; u < JU; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = wave + 4 * u, ch = lane + 64 * k;
            w1r[u][k] = w1[(j < Cse && ch < C) ? (long)j * C + ch : 0];
        }
    // fc2 rows of this thread's channels ch = tid + 256 r (r < 2 covers C <= 512), Cse <= 32 floats each
    f32x4 w2r[2][8];
    float b2r[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ch = threadIdx.x + 256 * r;
        b2r[r] = b2[ch < C ? ch : 0];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            w2r[r][u] = *reinterpret_cast<const f32x4*>(w2 + ((ch < C && 4 * u < Cse) ? (long)ch * Cse + 4 * u : 0));
    }
    float b1r[JU];
#pragma unroll
    for (int u = 0; u < JU; ++u) b1r[u] = b1[wave + 4 * u < Cse ? wave + 4 * u : 0];

    // mean over positions: thread = (channel quad, parity of the partial-row index); 16-byte loads, 8 in flight, summed in
    // index order (fixed: bitwise reproducible).  The rolled per-64-channel loop of se_gate_kernel is one round trip per
    // iteration -- 7 on the 432-channel stage, most of that kernel's 20 us.
    {
        const int quad = threadIdx.x & 127, par = threadIdx.x >> 7;
        if (quad * 4 < Cp) {
            const float* pp = pool + (long)n * pool_blocks * Cp + quad * 4;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            int q = par;
            for (; q + 14 < pool_blocks; q += 16) {
                f32x4 t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(pp + (long)(q + 2 * k) * Cp);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += t[k];
            }
            if (q < pool_blocks) {  // the rest (fewer than eight rows of this parity): in flight together as well, added in the same order
                const int last = q + 2 * ((pool_blocks - 1 - q) / 2);
                f32x4 t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(pp + (long)min(q + 2 * k, last) * Cp);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (q + 2 * k < pool_blocks) acc += t[k];
            }
            *reinterpret_cast<f32x4*>(part + par * Cp + quad * 4) = acc;
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < Cp; ch += blockDim.x) mean[ch] = (part[ch] + part[Cp + ch]) * inv_positions;
    __syncthreads();
    {
        float s[JU];
#pragma unroll
        for (int u = 0; u < JU; ++u) s[u] = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = lane + 64 * k;
            const float m = ch < C ? mean[ch] : 0.0f;
#pragma unroll
            for (int u = 0; u < JU; ++u) s[u] = fmaf((ch < C && wave + 4 * u < Cse) ? w1r[u][k] : 0.0f, m, s[u]);
        }
#pragma unroll
        for (int u = 0; u < JU; ++u) {
            float t = s[u];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
            const int j = wave + 4 * u;
            if (lane == 0 && j < Cse) hid[j] = fmaxf(t + b1r[u], 0.0f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ch = threadIdx.x + 256 * r;
        if (ch < Cp) {
            float g = 0.0f;
            if (ch < C) {
                float sacc = b2r[r];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (4 * u < Cse) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) sacc = fmaf(w2r[r][u][e], hid[4 * u + e], sacc);
                    }
                g = sigmoidf_(sacc);
            }
            gate[(long)n * Cp + ch] = g;
        }
    }
}

extern "C" int pasn_se_gate_fwd(const float* pool_partial, int pool_blocks, int positions, const float* w1, const float* b1,
                                const float* w2, const float* b2, float* gate, int N, int C, int Cp, int Cse, void* stream) {
    PASN_REQUIRE(pool_partial && w1 && b1 && w2 && b2 && gate, "null pointer");
    PASN_REQUIRE(N > 0 && C > 0 && Cp >= C && Cse > 0 && pool_blocks > 0 && positions > 0, "bad sizes");
    const size_t lds = (size_t)(5 * Cp + Cse) * sizeof(float);
    if (Cp <= 512 && Cse <= 32 && Cse % 4 == 0)
        hipLaunchKernelGGL(se_gate_fast_kernel, dim3(N), dim3(256), lds, (hipStream_t)stream, pool_partial, pool_blocks,
                           1.0f / (float)positions, w1, b1, w2, b2, gate, C, Cp, Cse);
    else
        hipLaunchKernelGGL(se_gate_kernel, dim3(N), dim3(256), lds, (hipStream_t)stream, pool_partial, pool_blocks,
                           1.0f / (float)positions, w1, b1, w2, b2, gate, C, Cp, Cse);
    return check_launch("se_gate_kernel");
}

extern "C" int pasn_maxpool3d_fwd(const void* x, void* y, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(x && y, "null pointer");
    PASN_REQUIRE(conv_desc_ok(d) && d->Cin_p == d->Cout_p, "bad geometry");
    const long total = (long)d->N * d->To * d->Ho * d->Wo * (d->Cout_p / 8);
    const dim3 grid(ceil_div(total, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_F32)
        hipLaunchKernelGGL((maxpool3d_kernel<float>), grid, block, 0, s, (const float*)x, (float*)y, *d);
    else if (dtype == PASN_BF16)
        hipLaunchKernelGGL((maxpool3d_kernel<__bf16>), grid, block, 0, s, (const __bf16*)x, (__bf16*)y, *d);
    else {
        set_error("pasn_maxpool3d_fwd: unknown dtype");
        return PASN_ERR_ARG;
    }
    return check_launch("maxpool3d_kernel");
}
