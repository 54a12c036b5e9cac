// Fused X3D bottleneck front half:  1x1x1 expand conv + BN + ReLU  ->  depthwise 3x3x3 (stride (1,s,s)) + BN
// (+ Swish, + squeeze-excite partial sums), one launch, the 2.25x-wide expanded activation never touches HBM.
//
// Why: unfused, the expanded tensor is written once and read back once -- 42 % of all trunk bytes at the
// benchmark shape -- and PMC showed the unfused kernels latency-bound on small per-block tiles (~70 % of wave
// cycles in s_waitcnt).  Here a block owns a (TH x TW) output tile x a chunk of CC inner channels of ONE clip and
// MARCHES ALONG T:
//
//     x plane (t+2)  --coalesced 16-byte loads, issued before the stencil of plane t-->  registers
//     stencil(t)      reads the three resident expanded planes t-1, t, t+1 from an LDS ring (bf16/fp32 rows
//                     [pos][CC]); each thread owns WT consecutive outputs along W x 8 channels (register sliding
//                     window, fp32 accumulators), BN_b / Swish / SE partial sums in the epilogue
//     registers -> LDS (x tile),  MFMA expand of plane t+2: E[cc][pos] = relu(sa * sum_k Wa[cc][k] x[pos][k] + ba),
//                     forced to 0 outside the image (the stencil zero-pads the EXPANDED activation) -> ring slot
//
// so every input plane is fetched once per block (spatial halo only), the HBM latency of the next plane hides
// under the VALU work of the current one, and the only global traffic is x in / depthwise output out.
// Squeeze-excite sums stay per-block partials in fixed order (bitwise reproducible).
#include "common.h"

namespace pasn {

constexpr int XD_XPT = 4;  // max 16/32-byte x chunks a thread holds in flight per plane (register budget)

struct XdGeom {
    int ok;
    int TH, TW, PH, PW, PP, PPt;        // output tile, input (halo) tile, positions, positions rounded to 32
    int CC, CCt, chunks;                // inner channels per block (multiple of 8), rounded to 32, chunk count
    int tilesH, tilesW, strips, threads;
    int xs_row, wa_row, ring_row;       // LDS row strides in elements
    int off_ring, off_xs, off_wa, off_wdw, off_sab, off_sbb, lds;
};

static inline int odd_slots(int elems, int es) {  // row stride with an odd number of 16-byte slots (conflict-free b128 column reads)
    const int per = 16 / es;
    int slots = (elems + per - 1) / per;
    if (slots % 2 == 0) ++slots;
    return slots * per;
}

XdGeom xd_geom(const pasn_conv_desc& d, int dtype, int WT) {
    XdGeom g = {};
    const bool shape_ok = d.kt == 3 && d.kh == 3 && d.kw == 3 && d.st == 1 && d.sh == d.sw && (d.sw == 1 || d.sw == 2) &&
                          d.pt == 1 && d.ph == 1 && d.pw == 1 && d.To == d.Ti && d.w_kc >= d.Cin_p;
    if (!shape_ok) return g;
    const int es = dtype == PASN_BF16 ? 2 : 4;
    const int s = d.sw;
    const int budget = 150 * 1024;
    g.TW = d.Wo < 28 ? d.Wo : 28;
    g.tilesW = ceil_div(d.Wo, g.TW);
    g.strips = ceil_div(g.TW, WT);
    g.PW = (g.TW - 1) * s + 3;
    g.xs_row = odd_slots(d.w_kc, es);
    g.wa_row = g.xs_row;
    // Search (output rows per tile) x (channel chunks) for the cheapest configuration that fits LDS.  Cost model:
    // the stencil dominates, so weight thread utilisation and CU fill fully and the spatial halo (extra expand
    // MFMAs + x traffic) by half.  Chunking the inner channels costs only a re-read of the narrow x tile.
    const int th_try[] = {28, 16, 14, 8, 7, 4, 2, 1};
    double best = 1e30;
    for (int chunks = 1; chunks <= 16; ++chunks) {
        const int CC = (ceil_div(d.Cout_p, chunks) + 7) / 8 * 8;
        if (chunks > 1 && ceil_div(d.Cout_p, CC) != chunks) continue;
        for (int ti = 0; ti < 8; ++ti) {
            const int TH = th_try[ti];
            if (TH > d.Ho && TH != 1) continue;
            const int PH = (TH - 1) * s + 3;
            const int PP = PH * g.PW, PPt = (PP + 31) / 32 * 32, CCt = (CC + 31) / 32 * 32;
            int off = 0;
            const int off_ring = off;
            const int ring_row = odd_slots(CC, es);  // odd 16-byte-slot stride: neighbouring strips never alias banks
            off += (3 * PP * ring_row * es + 15) / 16 * 16;
            const int off_xs = off;
            off += 2 * PPt * g.xs_row * es;  // double buffered: plane pl lands in tile pl & 1
            const int off_wa = off;
            off += CCt * g.wa_row * es;
            const int off_wdw = off;
            off += 27 * CC * 4;
            const int off_sab = off;
            off += 2 * CCt * 4;
            const int off_sbb = off;
            off += 2 * CC * 4;
            const int cgc = CC / 8;
            const int threads = (TH * g.strips * cgc > 256 || PP * (d.Cin_p / 8) > XD_XPT * 256) ? 512 : 256;
            if (off > budget) continue;
            if (PP * (d.Cin_p / 8) > XD_XPT * threads) continue;   // x tile must fit the per-thread prefetch registers
            if (cgc > threads) continue;
            const int RT = threads / cgc, work = TH * g.strips;    // row-strip slots vs row-strips per plane
            const double util = (double)work * cgc / ((double)ceil_div(work, RT) * threads);
            const long blocks = (long)d.N * ceil_div(d.Ho, TH) * g.tilesW * chunks;
            const int per_cu = off > 76 * 1024 ? 1 : 2;
            const double fill = (double)blocks / ((double)ceil_div(blocks, 256L * per_cu) * 256 * per_cu);
            const double halo = (double)(PH * g.PW) / ((double)TH * s * g.TW * s);
            double cost = (0.5 + 0.5 * halo) / (util * fill);
            if (const char* e = getenv("PASN_XD_TH")) cost += (atoi(e) == TH) ? -1e6 : 0.0;          // tuning overrides
            if (const char* e = getenv("PASN_XD_CHUNKS")) cost += (atoi(e) == chunks) ? -1e3 : 0.0;
            if (cost >= best) continue;
            best = cost;
            g.ok = 1;
            g.TH = TH; g.PH = PH; g.PP = PP; g.PPt = PPt; g.CC = CC; g.CCt = CCt; g.chunks = chunks;
            g.tilesH = ceil_div(d.Ho, TH);
            g.threads = threads;
            g.ring_row = ring_row;
            g.off_ring = off_ring; g.off_xs = off_xs; g.off_wa = off_wa; g.off_wdw = off_wdw; g.off_sab = off_sab; g.off_sbb = off_sbb;
            g.lds = off;
        }
    }
    return g;
}

template <typename T>
struct Raw;  // one 8-channel chunk as raw 16-byte words
template <>
struct Raw<__bf16> {
    uint4 v[1];
};
template <>
struct Raw<float> {
    uint4 v[2];
};

template <typename T, int SW, int WT>
__global__ __launch_bounds__(512) void x3d_expand_dw_kernel(const T* __restrict__ x, const T* __restrict__ wa,
                                                            const float* __restrict__ sa, const float* __restrict__ ba,
                                                            const float* __restrict__ wb, const float* __restrict__ sb,
                                                            const float* __restrict__ bb, T* __restrict__ y,
                                                            float* __restrict__ pool, pasn_conv_desc d, XdGeom g) {
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;
    constexpr int KSTEP = Traits<T>::KSTEP;
    constexpr int NC = (WT - 1) * SW + 3;  // input columns one strip touches
    constexpr int NV = sizeof(Raw<T>) / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* ring = reinterpret_cast<T*>(smem + g.off_ring);       // [3][PP][RR]
    T* xs0 = reinterpret_cast<T*>(smem + g.off_xs);          // [2][PPt][xs_row]
    T* was = reinterpret_cast<T*>(smem + g.off_wa);          // [CCt][wa_row]
    float* wdw = reinterpret_cast<float*>(smem + g.off_wdw); // [27][CC]
    float* sab = reinterpret_cast<float*>(smem + g.off_sab); // [2][CCt]   BN after the expand conv
    float* sbb = reinterpret_cast<float*>(smem + g.off_sbb); // [2][CC]    BN after the depthwise conv

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    const int T_ = d.Ti, Hi = d.Hi, Wi = d.Wi, Ho = d.Ho, Wo = d.Wo, Cin_p = d.Cin_p, Ci_p = d.Cout_p;
    const int PP = g.PP, PW = g.PW, CC = g.CC, CCt = g.CCt, RR = g.ring_row;

    // ---- which tile am I (XCD-aware order: an XCD gets whole clips) ---------------------------------------------------
    const int per_clip = g.tilesH * g.tilesW * g.chunks;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int n = lb / per_clip;
    int rem = lb - n * per_clip;
    const int chunk = rem % g.chunks;  // chunk fastest: the blocks that share an x tile run back to back
    rem /= g.chunks;
    const int tw = rem % g.tilesW, th = rem / g.tilesW;
    const int c0 = chunk * CC;
    const int cch = min(CC, Ci_p - c0);  // channels really present in this chunk (multiple of 8)
    const int cgc = cch / 8;
    const int ho0 = th * g.TH, wo0 = tw * g.TW;
    const int hi0 = ho0 * SW - 1, wi0 = wo0 * SW - 1;

    // ---- constants -> LDS: depthwise weights, BN_a scale/bias, expand-weight rows of this chunk; xs zeroed once --------
    for (int i = tid; i < 27 * CC; i += nthr) {
        const int tap = i / CC, ch = i - tap * CC;
        wdw[i] = ch < cch ? wb[(long)tap * Ci_p + c0 + ch] : 0.0f;
    }
    for (int i = tid; i < CCt; i += nthr) {
        sab[i] = i < cch ? sa[c0 + i] : 0.0f;
        sab[CCt + i] = i < cch ? ba[c0 + i] : 0.0f;
    }
    for (int i = tid; i < CC; i += nthr) {
        sbb[i] = i < cch ? sb[c0 + i] : 0.0f;
        sbb[CC + i] = i < cch ? bb[c0 + i] : 0.0f;
    }
    {
        const int kc8 = d.w_kc / 8;
        for (int i = tid; i < CCt * kc8; i += nthr) {
            const int r = i / kc8, k8 = i - r * kc8;
            Raw<T> v;
            if (r < cch) {
                const uint4* src = reinterpret_cast<const uint4*>(wa + (long)(c0 + r) * d.w_kc + k8 * 8);
#pragma unroll
                for (int q = 0; q < NV; ++q) v.v[q] = src[q];
            } else {
#pragma unroll
                for (int q = 0; q < NV; ++q) v.v[q] = make_uint4(0, 0, 0, 0);
            }
            uint4* dst = reinterpret_cast<uint4*>(was + (size_t)r * g.wa_row + k8 * 8);
#pragma unroll
            for (int q = 0; q < NV; ++q) dst[q] = v.v[q];
        }
        // zero the whole x tile once: its K padding [Cin_p, w_kc) is never written again and must not hold NaNs
        const int words = 2 * g.PPt * g.xs_row * (int)sizeof(T) / 16;
        uint4* z = reinterpret_cast<uint4*>(xs0);
        for (int i = tid; i < words; i += nthr) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // ---- x plane tile: global -> registers (asynchronous), registers -> LDS -------------------------------------------------
    const int cgin = Cin_p / 8;
    const int xtotal = PP * cgin;
    Raw<T> xr[XD_XPT];
    auto load_x = [&](int p) {
#pragma unroll
        for (int k = 0; k < XD_XPT; ++k) {
            const int idx = tid + k * nthr;
            if (idx < xtotal) {
                const int pos = idx / cgin, cg = idx - pos * cgin;
                const int pr = pos / PW, pc = pos - pr * PW;
                const int hi = hi0 + pr, wi = wi0 + pc;
                if (hi >= 0 && hi < Hi && wi >= 0 && wi < Wi) {
                    const uint4* src =
                        reinterpret_cast<const uint4*>(x + ((((long)n * T_ + p) * Hi + hi) * Wi + wi) * Cin_p + cg * 8);
#pragma unroll
                    for (int q = 0; q < NV; ++q) xr[k].v[q] = src[q];
                } else {
#pragma unroll
                    for (int q = 0; q < NV; ++q) xr[k].v[q] = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };
    auto store_x = [&](T* xs) {
#pragma unroll
        for (int k = 0; k < XD_XPT; ++k) {
            const int idx = tid + k * nthr;
            if (idx < xtotal) {
                const int pos = idx / cgin, cg = idx - pos * cgin;
                uint4* dst = reinterpret_cast<uint4*>(xs + (size_t)pos * g.xs_row + cg * 8);
#pragma unroll
                for (int q = 0; q < NV; ++q) dst[q] = xr[k].v[q];
            }
        }
    };

    // ---- MFMA expand of the x tile in LDS into ring slot `slot` ----------------------------------------------------------------
    const int ksteps = d.w_kc / KSTEP;
    auto expand = [&](const T* xs, int slot) {
        T* rs = ring + (size_t)slot * PP * RR;
        const int c = lane & 31, h = lane >> 5;
        for (int pt = wave; pt < g.PPt / 32; pt += nwaves) {
            const int pos = pt * 32 + c;
            const bool pv = pos < PP;
            const int pr = pos / PW, pc = pos - pr * PW;
            const int hi = hi0 + pr, wi = wi0 + pc;
            const bool inimg = pv && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
            const T* bp = xs + (size_t)pos * g.xs_row + h * CH;
            for (int ct = 0; ct < CCt / 32; ++ct) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
                const T* ap = was + (size_t)(ct * 32 + c) * g.wa_row + h * CH;
                for (int ks = 0; ks < ksteps; ++ks) mma32(acc, load_frag<T>(ap + ks * KSTEP), load_frag<T>(bp + ks * KSTEP));
                if (pv) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int chl = ct * 32 + 8 * g4 + 4 * h;
                        if (chl < cch) {
                            float o[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float v = acc[4 * g4 + q] * sab[chl + q] + sab[CCt + chl + q];
                                o[q] = inimg ? fmaxf(v, 0.0f) : 0.0f;
                            }
                            store4(rs + (size_t)pos * RR + chl, o);
                        }
                    }
                }
            }
        }
    };

    // ---- depthwise stencil of output plane t from ring planes t-1, t, t+1 ---------------------------------------------------------
    const int RT = nthr / cgc;  // row-strip slots; thread = (rt, cg), cg fastest
    const bool st_on = tid < RT * cgc;
    const int scg = tid % cgc, srt = tid / cgc;
    float psum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) psum[j] = 0.0f;
    const int wlim = min(Wo, wo0 + g.TW);
    auto stencil = [&](int t) {
        if (!st_on) return;
        for (int q = srt; q < g.TH * g.strips; q += RT) {
            const int r = q / g.strips, s = q - r * g.strips;
            const int ho = ho0 + r, wob = wo0 + s * WT;
            if (ho >= Ho || wob >= wlim) continue;
            float acc[WT][8];
#pragma unroll
            for (int o = 0; o < WT; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] = 0.0f;
            for (int a = 0; a < 3; ++a) {
                const int p = t - 1 + a;
                if (p < 0 || p >= T_) continue;
                const T* rs = ring + (size_t)(p % 3) * PP * RR;
#pragma unroll 1
                for (int b = 0; b < 3; ++b) {  // not unrolled: keeps one row of inputs + 3x8 weights live, no spills
                    float wv[3][8];
#pragma unroll
                    for (int e = 0; e < 3; ++e) load8(wdw + ((a * 3 + b) * 3 + e) * CC + scg * 8, wv[e]);
                    const T* rowp = rs + (size_t)((r * SW + b) * PW + s * WT * SW) * RR + scg * 8;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        float xv[8];
                        load8(rowp + (size_t)c * RR, xv);
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            if ((c - e) >= 0 && (c - e) % SW == 0 && (c - e) / SW < WT) {  // compile-time
                                const int o = (c - e) / SW;
#pragma unroll
                                for (int j = 0; j < 8; ++j) acc[o][j] = fmaf(xv[j], wv[e][j], acc[o][j]);
                            }
                        }
                    }
                }
            }
            float bsc[8], bbs[8];
            load8(sbb + scg * 8, bsc);
            load8(sbb + CC + scg * 8, bbs);
            T* yrow = y + ((((long)n * T_ + t) * Ho + ho) * Wo) * Ci_p + c0 + scg * 8;
#pragma unroll
            for (int o = 0; o < WT; ++o) {
                const int wo = wob + o;
                if (wo >= wlim) continue;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = acc[o][j] * bsc[j] + bbs[j];
                    psum[j] += v[j];
                }
                act_vec(v, d.act);
                mask_tail(v, d.Cout - (c0 + scg * 8));
                store8(yrow + (long)wo * Ci_p, v);
            }
        }
    };

    // ---- march along T (one call site per phase: iterations tt = -2, -1 only fill the ring) --------------------------------------
    for (int tt = -2; tt < T_; ++tt) {
        const int pl = tt + 2;  // plane fetched + expanded in this iteration
        const bool more = pl < T_;
        T* xs = xs0 + (size_t)(pl & 1) * g.PPt * g.xs_row;
        if (more) load_x(pl);   // in flight while the stencil runs
        if (tt >= 0) stencil(tt);
        if (more) store_x(xs);  // tile pl & 1 was last read by the expand two iterations ago
        __syncthreads();        // x tile complete; plane tt-1's ring slot is free
        if (more) {
            expand(xs, pl % 3);
            __syncthreads();
        }
    }

    // ---- squeeze-excite partial sums: fixed-order reduction over the row-strip slots ----------------------------------------------------
    if (pool) {
        float* red = reinterpret_cast<float*>(ring);  // the ring is dead now (all threads passed the last barrier)
        if (st_on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[srt * CC + scg * 8 + j] = psum[j];
        }
        __syncthreads();
        const int pool_blocks = g.tilesH * g.tilesW;
        for (int ch = tid; ch < cch; ch += nthr) {
            float s = 0.0f;
            for (int q = 0; q < RT; ++q) s += red[q * CC + ch];
            pool[((long)n * pool_blocks + th * g.tilesW + tw) * Ci_p + c0 + ch] = s;
        }
    }
}

template <typename T>
static int launch_xd(const void* x, const void* wa, const float* sa, const float* ba, const float* wb, const float* sb,
                     const float* bb, void* y, float* pool, const pasn_conv_desc& d, const XdGeom& g, int WT, hipStream_t s) {
    const dim3 grid(d.N * g.tilesH * g.tilesW * g.chunks), block(g.threads);
#define PASN_XD(SW_, WT_)                                                                                                   \
    do {                                                                                                                    \
        PASN_MAX_LDS(160 * 1024, x3d_expand_dw_kernel<T, SW_, WT_>);                                                      \
        hipLaunchKernelGGL((x3d_expand_dw_kernel<T, SW_, WT_>), grid, block, (size_t)g.lds, s, (const T*)x, (const T*)wa, sa, \
                           ba, wb, sb, bb, (T*)y, pool, d, g);                                                              \
    } while (0)
    if (d.sw == 1) {
        if (WT == 7) PASN_XD(1, 7); else PASN_XD(1, 4);
    } else {
        if (WT == 7) PASN_XD(2, 7); else PASN_XD(2, 4);
    }
#undef PASN_XD
    return check_launch("x3d_expand_dw_kernel");
}

static int xd_wt() {
    if (const char* e = getenv("PASN_XD_WT"))
        if (atoi(e) == 7) return 7;
    return 4;
}

}  // namespace pasn

using namespace pasn;

static bool xd_desc_ok(const pasn_conv_desc* d) {
    return d && d->N > 0 && d->Ti > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && d->Cin > 0 && d->Cout > 0 &&
           d->Cin_p >= d->Cin && d->Cout_p >= d->Cout && d->Cin_p % 8 == 0 && d->Cout_p % 8 == 0;
}

extern "C" int pasn_x3d_expand_dw_variant(const pasn_conv_desc* d, int dtype) {
    if (!xd_desc_ok(d) || (dtype != PASN_F32 && dtype != PASN_BF16)) return 0;
    if (x3d_front_geom(*d, dtype).ok) return 1;  // x3d_front_kernel (7x7 planes, PASN_FRONT=1)
    const char* e = getenv("PASN_FUSED");
    if (!e || e[0] != '1') return 0;
    return xd_geom(*d, dtype, xd_wt()).ok ? 2 : 0;  // x3d_expand_dw_kernel (opt-in)
}

extern "C" int pasn_x3d_expand_dw_pool_blocks(const pasn_conv_desc* d, int dtype) {
    if (!xd_desc_ok(d) || (dtype != PASN_F32 && dtype != PASN_BF16)) return 0;
    {
        const XfrontGeom f = x3d_front_geom(*d, dtype);
        if (f.ok) return f.nT;
    }
    // Opt-in until it beats the unfused pair everywhere (round-1 measurements: profiles/README.md).
    const char* e = getenv("PASN_FUSED");
    if (!e || e[0] != '1') return 0;
    const XdGeom g = xd_geom(*d, dtype, xd_wt());
    if (getenv("PASN_XD_VERBOSE"))
        fprintf(stderr, "[x3d_expand_dw] %d->%d s%d in %dx%dx%d: ok=%d TH=%d TW=%d PP=%d CC=%d chunks=%d tiles=%dx%d threads=%d lds=%d blocks=%d\n",
                d->Cin, d->Cout, d->sw, d->Ti, d->Hi, d->Wi, g.ok, g.TH, g.TW, g.PP, g.CC, g.chunks, g.tilesH, g.tilesW,
                g.threads, g.lds, d->N * g.tilesH * g.tilesW * g.chunks);
    return g.ok ? g.tilesH * g.tilesW : 0;
}

extern "C" int pasn_x3d_expand_dw_fwd(const void* x, const void* wa, const float* sa, const float* ba, const float* wb,
                                      const float* sb, const float* bb, void* y, float* pool_partial, const pasn_conv_desc* d,
                                      int dtype, void* stream) {
    PASN_REQUIRE(x && wa && sa && ba && wb && sb && bb && y, "null pointer");
    PASN_REQUIRE(xd_desc_ok(d), "bad geometry (channel strides must be multiples of 8)");
    PASN_REQUIRE(dtype == PASN_F32 || dtype == PASN_BF16, "unknown dtype");
    const int kstep = dtype == PASN_BF16 ? 16 : 8;
    PASN_REQUIRE(d->w_kc >= d->Cin_p && d->w_kc % kstep == 0 && d->w_rows >= d->Cout_p, "packed expand weight does not cover the geometry");
    {
        const XfrontGeom f = x3d_front_geom(*d, dtype);
        if (f.ok) return launch_x3d_front(x, wa, sa, ba, wb, sb, bb, y, pool_partial, *d, f, (hipStream_t)stream);
    }
    const int WT = xd_wt();
    const XdGeom g = xd_geom(*d, dtype, WT);
    if (!g.ok) {
        set_error("pasn_x3d_expand_dw_fwd: geometry not supported by the fused kernel (pool_blocks query returns 0)");
        return PASN_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_F32) return launch_xd<float>(x, wa, sa, ba, wb, sb, bb, y, pool_partial, *d, g, WT, s);
    return launch_xd<__bf16>(x, wa, sa, ba, wb, sb, bb, y, pool_partial, *d, g, WT, s);
}
