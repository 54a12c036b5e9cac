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
//
// KS2 > 0: the block's strided 1x1x1 SHORTCUT conv + its norm rides in the same launch (first block of an X3D stage):
//   Y = act(scale * (W . X') + scale2 * (W2 . X2[rowmap(m)]) + bias),  bias = both norms' shifts summed by the host,
// a second accumulator set fed from the block INPUT at the strided position of every output row (second weight set in registers,
// its rows prefetched like X).  The shortcut tensor (written by one launch, read back as the residual by the next) never exists.
#include "common.h"
#include <type_traits>

namespace pasn {

// act_vec with the activation optionally compiled in (ACTC = -1: the run-time switch)
template <int ACTC, int N>
__device__ __forceinline__ void act_vec_c(float (&v)[N], int act) {
    if constexpr (ACTC == PASN_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = relu_f32(v[j]);
    } else {
        act_vec(v, act);
    }
}

// 4 consecutive channels of a residual row, kept as loaded until the epilogue
template <typename T>
struct RawQuad;
template <>
struct RawQuad<__bf16> {
    using type = uint2;
    static __device__ __forceinline__ void to_f4(const uint2& r, float (&v)[4]) {
        v[0] = __uint_as_float(r.x << 16);
        v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16);
        v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
};
template <>
struct RawQuad<float> {
    using type = uint4;
    static __device__ __forceinline__ void to_f4(const uint4& r, float (&v)[4]) {
        v[0] = __uint_as_float(r.x);
        v[1] = __uint_as_float(r.y);
        v[2] = __uint_as_float(r.z);
        v[3] = __uint_as_float(r.w);
    }
};

template <typename T, int KS, int NT, bool RES, int KS2 = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((KS == 2 && NT == 4 && !RES) ? 3 : 1))) void pwconv_persist_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ bias,
                                                             const T* __restrict__ res, const float* __restrict__ gate,
                                                             T* __restrict__ y, long M, int S, int Cin_p, int Cout, int Cout_p,
                                                             int w_kc, int act, int in_swish, int gate_rows, PwShort sc2) {
    static_assert(KS2 == 0 || !RES, "the fused shortcut replaces the residual operand");
    constexpr int SB = KS2 ? 3 : 2;  // scale | bias (| scale2) arrays in LDS
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;
    constexpr int KSTEP = Traits<T>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sbl = reinterpret_cast<float*>(smem);                  // [SB][NT*32] scale | bias (| shortcut scale) of the channel chunk
    T* stage = reinterpret_cast<T*>(smem + SB * NT * 32 * 4);      // [4 waves][32 rows][Cout_p] output images
    // [gate_rows][Cin_p] the WHOLE squeeze-excite gate tensor (a few KB), staged once per block: the transform then reads
    // LDS instead of issuing 8 dependent dword loads per k-step (64 L2 round trips per tile in the first version)
    float* gl = reinterpret_cast<float*>(smem + SB * NT * 32 * 4 + (size_t)4 * 32 * Cout_p * sizeof(T));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int co_base = blockIdx.y * (NT * 32);
    for (int i = threadIdx.x; i < NT * 32; i += 256) {
        sbl[i] = scale ? scale[co_base + i] : 1.0f;
        sbl[NT * 32 + i] = bias ? bias[co_base + i] : 0.0f;
        if (KS2) sbl[2 * NT * 32 + i] = sc2.scale2 ? sc2.scale2[co_base + i] : 1.0f;
    }
    if (gate_rows)
        for (int i = threadIdx.x; i < gate_rows * Cin_p; i += 256) gl[i] = gate[i];
    __syncthreads();

    const int ks_real = w_kc / KSTEP;  // <= KS; the extra template steps are zero
    frag A[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            A[nt][ks] = ks < ks_real ? load_frag<T>(w + (long)(co_base + nt * 32 + c) * w_kc + ks * KSTEP + h * CH) : zero_frag<T>();

    // shortcut weights: a second stationary set (KS2 fragments per channel tile)
    frag A2[KS2 ? NT : 1][KS2 ? KS2 : 1];
    const T* x2 = reinterpret_cast<const T*>(sc2.x2);
    if (KS2) {
        const T* w2 = reinterpret_cast<const T*>(sc2.w2);
        const int ks2_real = sc2.w_kc2 / KSTEP;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
                A2[nt][ks] = ks < ks2_real ? load_frag<T>(w2 + (long)(co_base + nt * 32 + c) * sc2.w_kc2 + ks * KSTEP + h * CH) : zero_frag<T>();
    }

    const long ntiles = (M + 31) / 32;
    const long stride = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    const bool xform = (gate != nullptr) || (in_swish != 0);
    // clip of this lane's row, kept incrementally (one 64-bit division here instead of one per tile: the gated layers
    // do 4-8 MFMAs per tile, a software division is of the same order)
    long cn = 0;   // clip index of row m = tile * 32 + c
    long crem = 0;  // m - cn * S
    if (gate) {
        const long m_first = tile * 32 + c;
        cn = m_first / S;
        crem = m_first - cn * S;
    }
    const long row_step = stride * 32;

    frag Bc[KS], Bn[KS];
    // ISSUE ONLY: unconditional loads from clamped (existing) addresses; rows beyond M / columns beyond Cin_p are zeroed
    // when the fragments are consumed.  (`cond ? load : 0` puts every load in its own branch, and any use of the value
    // next to the load makes hipcc wait for it before issuing the next one.)
    auto load_rows = [&](long t, frag (&B)[KS]) {
        const long m = min(t * 32 + c, M - 1);
        const T* xp = x + m * Cin_p;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * KSTEP + h * CH;
            B[ks] = load_frag<T>(xp + (k < Cin_p ? k : 0));
        }
    };
    // shortcut rows: this lane's output position as (wo, ho, frame index n To + to), advanced by the tile stride with carries (one
    // division chain here instead of three per tile)
    frag B2c[KS2 ? KS2 : 1], B2n[KS2 ? KS2 : 1];
    int pwo = 0, pho = 0;
    long ptn = 0;
    int dwo = 0, dho = 0;
    long dtn = 0;
    if (KS2) {
        const long m_first = tile * 32 + c;
        pwo = (int)(m_first % sc2.Wo);
        const long r = m_first / sc2.Wo;
        pho = (int)(r % sc2.Ho);
        ptn = r / sc2.Ho;
        const long rs = stride * 32;
        dwo = (int)(rs % sc2.Wo);
        const long r2 = rs / sc2.Wo;
        dho = (int)(r2 % sc2.Ho);
        dtn = r2 / sc2.Ho;
    }
    const long rows_in = KS2 ? (M / ((long)sc2.Ho * sc2.Wo)) * sc2.Hi * sc2.Wi : 0;  // M = frames x Ho x Wo
    auto load_rows2 = [&](frag (&B)[KS2 ? KS2 : 1]) {  // rows of the position held in (pwo, pho, ptn); clamped like load_rows
        long row = (ptn * sc2.Hi + (long)pho * sc2.sh) * sc2.Wi + (long)pwo * sc2.sw;
        row = row < rows_in ? row : rows_in - 1;
        const T* xp = x2 + row * sc2.Cin2_p;
#pragma unroll
        for (int ks = 0; ks < (KS2 ? KS2 : 1); ++ks) {
            const int k = ks * KSTEP + h * CH;
            B[ks] = load_frag<T>(xp + (k < sc2.Cin2_p ? k : 0));
        }
    };
    auto advance2 = [&]() {
        pwo += dwo;
        const int c1 = pwo >= sc2.Wo ? 1 : 0;
        pwo -= c1 ? sc2.Wo : 0;
        pho += dho + c1;
        const int c2 = pho >= sc2.Ho ? 1 : 0;
        pho -= c2 ? sc2.Ho : 0;
        ptn += dtn + c2;
    };
    if (tile < ntiles) {
        load_rows(tile, Bc);
        if (KS2) {
            load_rows2(B2c);
            advance2();  // (pwo, pho, ptn) now names the NEXT tile's position
        }
    }

    for (; tile < ntiles; tile += stride) {
        const long m = tile * 32 + c;
        const bool mv = m < M;
        if (tile + stride < ntiles) {
            load_rows(tile + stride, Bn);  // next tile's rows: in flight during everything below
            if (KS2) {
                load_rows2(B2n);
                advance2();
            }
        }
        // this tile's residual rows, requested before the MFMAs that hide them
        // raw (unconverted), unconditional, clamped: see load_rows.  Compile-time RES: expand convs pay no registers.
        // bf16: the tile's 32 residual rows are ONE contiguous range, like the output: whole 16-byte pieces, lane-contiguous (the quad
        // loads below are 8 bytes per lane at the row stride: 32 partial lines per wave-load, what held the residual instances at
        // 3.2-4.3 TB/s); added in the copy-out loop on the rounded pre-activation sums, as igemm_epilogue does
        constexpr bool RCOPY = RES && sizeof(T) == 2;
        constexpr int RPL = 2 * NT;  // 16-byte pieces per lane: 32 rows x NT*32 channels / 8 / 64
        uint4 rp[RCOPY ? RPL : 1];
        if (RCOPY) {
            const long base = tile * 32 * (long)Cout_p, lim = M * (long)Cout_p;
            const int pieces = 32 * Cout_p / 8;
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const int q = lane + 64 * u;
                const bool ok = q < pieces && base + (long)q * 8 < lim;
                rp[u] = *reinterpret_cast<const uint4*>(res + (ok ? base + (long)q * 8 : 0));
            }
        }
        typename RawQuad<T>::type rq[(RES && !RCOPY) ? NT : 1][4];
        if (RES && !RCOPY) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co_base + nt * 32 + 8 * g + 4 * h;
                    const bool ok = mv && co < Cout_p;
                    rq[nt][g] = *reinterpret_cast<const typename RawQuad<T>::type*>(res + (ok ? m * Cout_p + co : 0));
                }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)  // zero what load_rows clamped
            if (!(mv && ks * KSTEP + h * CH < Cin_p)) Bc[ks] = zero_frag<T>();
        if (KS2) {
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
                if (!(mv && ks * KSTEP + h * CH < sc2.Cin2_p)) B2c[ks] = zero_frag<T>();
        }
        if (xform) {  // x' = swish(x * gate[n][ci]), rounded back to the MFMA input type
            const long n = mv ? cn : 0;
            const float* gp = gate ? (gate_rows ? gl : gate) + n * Cin_p + h * CH : nullptr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks * KSTEP + h * CH < Cin_p) {
                    float gv[CH];
#pragma unroll
                    for (int j = 0; j < CH; j += 4) {
                        const f32x4 q = gp ? *reinterpret_cast<const f32x4*>(gp + ks * KSTEP + j) : f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                        gv[j] = q[0];
                        gv[j + 1] = q[1];
                        gv[j + 2] = q[2];
                        gv[j + 3] = q[3];
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j) gv[j] *= (float)Bc[ks][j];
                    if (in_swish) {  // one wave-uniform branch around the piece (inside the element loop it compiles to a select per element)
#pragma unroll
                        for (int j = 0; j < CH; ++j) gv[j] = gv[j] * sigmoidf_(gv[j]);
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j) Bc[ks][j] = (T)gv[j];
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
        f32x16 acc2[KS2 ? NT : 1];
        if (KS2) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[nt][i] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma32(acc2[nt], A2[nt][ks], B2c[ks]);
        }
        // Epilogue.  The accumulator layout (position on the lane, 4 consecutive channels per quad) would give 8-byte
        // stores scattered at the row stride -- measured at ~1.7 TB/s of writes.  The 32 rows of a tile are ONE contiguous
        // global range, so bounce the finished tile through a WAVE-PRIVATE LDS image of exactly that range (no block
        // barrier: a wave's DS operations execute in order) and write it back as 16 bytes per lane, fully coalesced.
        // The activation is dispatched ONCE per tile (ReLU -- every X3D / ResNet pointwise unit -- compiled in): `act_vec(o, act)` inside the
        // unrolled (channel tile, group) loops was a ladder of scalar compares and branches per group, 4 NT of them per tile.
        auto epilogue = [&](auto actc) {
            constexpr int ACTC = decltype(actc)::value;
        T* st = stage + (size_t)wave * 32 * Cout_p;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = nt * 32 + 8 * g + 4 * h;
                if (nt * 32 + 8 * g >= Cout_p) continue;  // wave-uniform (Cout_p is a multiple of 8: both lane halves of a group exist or neither)
                float o[4], sc[4], bs[4];
                load4(sbl + col, sc);
                load4(sbl + NT * 32 + col, bs);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[nt][4 * g + j] * sc[j] + bs[j];
                if (KS2) {
                    float s2[4];
                    load4(sbl + 2 * NT * 32 + col, s2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(acc2[nt][4 * g + j], s2[j], o[j]);
                }
                if (RES && !RCOPY) {
                    float r4[4];
                    RawQuad<T>::to_f4(rq[nt][g], r4);
                    const bool ok = mv && co_base + col < Cout_p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += ok ? r4[j] : 0.0f;
                }
                if (!RCOPY) {
                    act_vec_c<ACTC>(o, act);
                    // only the group that holds padded channels pays the selects (wave-uniform test: as a per-lane one the compiler hoists a
                    // predicate per group out of the tile loop -- 30-150 spilled SGPRs per instance, read back lane by lane every tile)
                    if (nt * 32 + 8 * g + 8 > Cout) {
                        int left = Cout - col;
                        asm volatile("" : "+v"(left));  // computed here, on the path that needs it (not hoisted per group and kept in SGPRs)
                        mask_tail(o, left);
                    }
                }
                store4(st + (size_t)c * Cout_p + col, o);
            }
        __builtin_amdgcn_wave_barrier();
        {
            constexpr int CPL = 16 / (int)sizeof(T);              // elements per 16-byte piece
            const long base = tile * 32 * (long)Cout_p;           // first element of the tile in y
            const long lim = M * (long)Cout_p;                    // rows beyond M do not exist
            const int pieces = 32 * Cout_p / CPL;
            if (RCOPY) {
#pragma unroll
                for (int u = 0; u < RPL; ++u) {
                    const int q = lane + 64 * u;
                    if (q < pieces && base + (long)q * CPL < lim) {
                        uint4 iv[1] = {*reinterpret_cast<const uint4*>(st + (size_t)q * CPL)}, rv[1] = {rp[u]};
                        float v[8], r8[8];
                        raw_to_f8<__bf16>(iv, v);
                        raw_to_f8<__bf16>(rv, r8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r8[e];
                        act_vec_c<ACTC>(v, act);
                        if (Cout != Cout_p) {  // block-uniform: the project convs' channel counts (24 / 48 / 96 / 192) have no padding
                            int left = Cout - (q * 8) % Cout_p;  // channels left from the piece's first one
                            asm volatile("" : "+v"(left));
                            mask_tail(v, left);
                        }
                        store8(reinterpret_cast<__bf16*>(y) + base + (long)q * CPL, v);
                    }
                }
            } else {
                for (int q = lane; q < pieces; q += 64) {
                    if (base + (long)q * CPL < lim)
                        *reinterpret_cast<uint4*>(y + base + (long)q * CPL) = *reinterpret_cast<const uint4*>(st + (size_t)q * CPL);
                }
            }
        }
        };
        if (act == PASN_ACT_RELU) epilogue(std::integral_constant<int, PASN_ACT_RELU>{});
        else epilogue(std::integral_constant<int, -1>{});
        __builtin_amdgcn_wave_barrier();  // the image is reused by this wave's next tile
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Bc[ks] = Bn[ks];
        if (KS2) {
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) B2c[ks] = B2n[ks];
        }
        if (gate) {  // advance the clip index to the next tile's row
            crem += row_step;
            while (crem >= S) {
                crem -= S;
                ++cn;
            }
        }
    }
}

// Geometry shared by the launcher and the variant query.  KS = template k-steps (2, 4, 8), NT = 32-channel tiles per wave.
PwGeom pw_geom(const pasn_conv_desc& d, int dtype) {
    PwGeom g = {0, 0, 0, 0};
    const bool pointwise = d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 &&
                           d.ph == 0 && d.pw == 0;
    if (!pointwise) return g;
    if (const char* e = tune("PASN_NO_PWCONV"))
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

// Shortcut-fused instances (bf16): the first project convs of X3D stages 2 and 3 -- (KS, NT, KS2) = (4, 1, 2): 54 -> 24 + 24 -> 24 stride 2;
// (8, 2, 4): 108 -> 48 + 48 -> 48 stride 2.  0 = not covered.
int pw_short_ks2(const pasn_conv_desc& d, const pasn_conv_desc& d2, int dtype) {
    if (dtype != PASN_BF16) return 0;
    if (const char* e = tune("PASN_NO_SHORTFUSE"))
        if (e[0] == '1') return 0;
    const PwGeom g = pw_geom(d, dtype);
    if (!g.TM) return 0;
    const bool geo = d2.kt == 1 && d2.kh == 1 && d2.kw == 1 && d2.pt == 0 && d2.ph == 0 && d2.pw == 0 && d2.st == 1 && d2.Ti == d2.To &&
                     d2.N == d.N && d2.To == d.To && d2.Ho == d.Ho && d2.Wo == d.Wo && d2.Cout_p == d.Cout_p && d2.Cout == d.Cout &&
                     d2.w_frag == 0 && d2.w_kc % 16 == 0 && d2.w_kc >= d2.Cin_p && d2.w_rows >= d.w_rows && d2.in_swish == 0 &&
                     (d2.Hi - 1) / d2.sh + 1 == d2.Ho && (d2.Wi - 1) / d2.sw + 1 == d2.Wo;
    if (!geo) return 0;
    const int ks2 = d2.w_kc / 16;
    if (g.TM == 4 && g.xrow == 1 && ks2 <= 2) return 2;
    if (g.TM == 8 && g.xrow == 2 && ks2 <= 4) return ks2 <= 2 ? 2 : 4;
    return 0;
}

template <typename T>
int launch_pwconv(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                  void* y, const pasn_conv_desc& d, const PwGeom& g, hipStream_t s, const PwShort* sc) {
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int S = d.To * d.Ho * d.Wo;
    const long ntiles = (M + 31) / 32;
    const long want_blocks = (ntiles + 3) / 4;
    // persistent: exactly as many blocks as stay RESIDENT for this instance and LDS size (asked from the runtime once per
    // instance: a fixed guess put twice the resident number on the gated NT = 2 instance, whose second half then redid the
    // weight / gate staging in a second round); each wave strides over the row tiles
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const dim3 block(256);
#define PASN_PW2(KS_, NT_, RES_) PASN_PW3(KS_, NT_, RES_, 0, PwShort{})
#define PASN_PW3(KS_, NT_, RES_, KS2_, SC_)                                                                                        \
    do {                                                                                                                      \
        if (lds > 64 * 1024) PASN_MAX_LDS(80 * 1024, pwconv_persist_kernel<T, KS_, NT_, RES_, KS2_>); /* BEFORE the occupancy query */ \
        static std::atomic<long> cached_{-1}; /* (device << 40 | lds << 8 | per_cu) of the last query of this instance */          \
        const long key_ = ((long)dev << 40) | ((long)lds << 8);                                                               \
        long c_ = cached_.load(std::memory_order_relaxed);                                                                    \
        int per_cu = (c_ >= 0 && (c_ & ~0xffL) == key_) ? (int)(c_ & 0xff) : 0;                                               \
        if (per_cu == 0) {                                                                                                    \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pwconv_persist_kernel<T, KS_, NT_, RES_, KS2_>, 256, lds) !=  \
                    hipSuccess || per_cu < 1)                                                                                 \
                per_cu = 2;                                                                                                   \
            cached_.store(key_ | (per_cu & 0xff), std::memory_order_relaxed);                                                 \
        }                                                                                                                     \
        long blocks = want_blocks < (long)n_cu * per_cu ? want_blocks : (long)n_cu * per_cu;                                  \
        const dim3 grid((unsigned)blocks, 1);                                                                                 \
        hipLaunchKernelGGL((pwconv_persist_kernel<T, KS_, NT_, RES_, KS2_>), grid, block, lds, s, (const T*)x, (const T*)w, scale, bias,   \
                           (const T*)res, gate, (T*)y, M, S, d.Cin_p, d.Cout, d.Cout_p, d.w_kc, d.act, d.in_swish, gate_rows, SC_); \
    } while (0)
#define PASN_PW(KS_, NT_)                 \
    do {                                  \
        if (res) PASN_PW2(KS_, NT_, true); \
        else PASN_PW2(KS_, NT_, false);    \
    } while (0)
    const int KS = g.TM, NT = g.xrow;
    size_t lds = (size_t)(sc ? 3 : 2) * NT * 32 * 4 + (size_t)4 * 32 * d.Cout_p * sizeof(T);  // <= 64.5 KB (fp32, 128 channels)
    int gate_rows = 0;  // the whole gate tensor rides in LDS when it is small (it is: N x Cin_p floats)
    if (gate && (size_t)d.N * d.Cin_p * 4 <= 32 * 1024 && lds + (size_t)d.N * d.Cin_p * 4 <= 80 * 1024) {
        gate_rows = d.N;
        lds += (size_t)d.N * d.Cin_p * 4;
    }
    if (sc) {  // shortcut-fused instances (bf16 only: pw_short_ks2)
        if constexpr (sizeof(T) == 2) {
            PASN_REQUIRE(res == nullptr, "the fused shortcut replaces the residual");
            if (KS == 4 && NT == 1) PASN_PW3(4, 1, false, 2, *sc);
            else if (KS == 8 && NT == 2 && sc->w_kc2 <= 32) PASN_PW3(8, 2, false, 2, *sc);
            else if (KS == 8 && NT == 2) PASN_PW3(8, 2, false, 4, *sc);
            else PASN_REQUIRE(false, "no shortcut-fused instance for this layer");
        } else {
            PASN_REQUIRE(false, "the shortcut-fused pointwise conv is bf16 only");
        }
    } else if (KS == 2) {
        if (NT == 1) PASN_PW(2, 1); else if (NT == 2) PASN_PW(2, 2); else PASN_PW(2, 4);
    } else if (KS == 4) {
        if (NT == 1) PASN_PW(4, 1); else if (NT == 2) PASN_PW(4, 2); else PASN_PW(4, 4);
    } else {
        if (NT == 1) PASN_PW(8, 1); else PASN_PW(8, 2);
    }
#undef PASN_PW
#undef PASN_PW2
#undef PASN_PW3
    return check_launch("pwconv_persist_kernel");
}

template int launch_pwconv<float>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                  const pasn_conv_desc&, const PwGeom&, hipStream_t, const PwShort*);
template int launch_pwconv<__bf16>(const void*, const void*, const float*, const float*, const void*, const float*, void*,
                                   const pasn_conv_desc&, const PwGeom&, hipStream_t, const PwShort*);

}  // namespace pasn
