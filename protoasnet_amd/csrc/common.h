// Shared device helpers for the gfx950 kernels: dtype traits, 16-byte vector access, MFMA wrappers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>

#include "../../include/protoasnet_amd.h"
#include "tuning.h"

namespace pasn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ---- host-side error plumbing -------------------------------------------------------------------
void set_error(const std::string& msg);
int check_launch(const char* what);
#define PASN_REQUIRE(cond, msg)                                        \
    do {                                                               \
        if (!(cond)) {                                                 \
            ::pasn::set_error(std::string(__func__) + ": " + (msg));   \
            return PASN_ERR_ARG;                                       \
        }                                                              \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Raise a kernel instance's dynamic-LDS limit once per (call site, DEVICE).  hipFuncAttributeMaxDynamicSharedMemorySize is a per-device
// attribute: a process-wide "done" flag left every launch needing > 64 KB failing on a second GPU driven by the same process.  The
// flag is an atomic bit mask indexed by the current device, so concurrent callers are fine too.  Usage: PASN_MAX_LDS(bytes, kernel<...>).
#define PASN_MAX_LDS(bytes, ...)                                                                                              \
    do {                                                                                                                      \
        static std::atomic<unsigned long long> pasn_lds_done_{0ull};                                                          \
        int pasn_lds_dev_ = 0;                                                                                                \
        (void)hipGetDevice(&pasn_lds_dev_);                                                                                   \
        const unsigned long long pasn_lds_bit_ = 1ull << (pasn_lds_dev_ & 63);                                                \
        if (!(pasn_lds_done_.load(std::memory_order_acquire) & pasn_lds_bit_)) {                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (bytes));                                                                               \
            pasn_lds_done_.fetch_or(pasn_lds_bit_, std::memory_order_release);                                                \
        }                                                                                                                     \
    } while (0)

// pointwise-conv kernel (pwconv.hip): row tile, LDS row stride, channel chunk, LDS bytes; TM == 0 -> not applicable
struct PwGeom {
    int TM, xrow, co_chunk, lds;
};
PwGeom pw_geom(const pasn_conv_desc& d, int dtype);
// pwconv.hip: the fused strided shortcut conv's operands and geometry: output row (n, to, ho, wo) reads x2 row ((n To + to) Hi + ho sh) Wi + wo sw
struct PwShort {
    const void* x2;
    const void* w2;        // [rows][w_kc2], k-contiguous like w
    const float* scale2;   // [rows] or NULL (= 1)
    int Cin2_p, w_kc2, Ho, Wo, Hi, Wi, sh, sw;
};

int pw_short_ks2(const pasn_conv_desc& d, const pasn_conv_desc& d2, int dtype);  // template k-steps of the fused shortcut, 0 = not covered
template <typename T>
int launch_pwconv(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                  void* y, const pasn_conv_desc& d, const PwGeom& g, hipStream_t s, const PwShort* sc = nullptr);

// X-stationary pointwise conv for wide layers (pwconv_xtile.hip)
bool pw_xtile_applicable(const pasn_conv_desc& d, int dtype);
int pw_xtile_ks(const pasn_conv_desc& d, int dtype);  // template k-steps of the instance picked
template <typename T>
int launch_pw_xtile(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                    void* y, const pasn_conv_desc& d, hipStream_t s);
// pwconv_ws.hip: weight-stationary pointwise conv (bf16, fragment-major weights): persistent blocks, LDS-DMA stage ring; ok = 0: not covered
struct WsGeom {
    int ok, KS, KS2, MT, CT, PT, NW, gy, NS;  // (KS2: second conv of a chained pair, 0 = single conv) template k-steps, 32-position sub-tiles per wave, waves along channels / positions, channel groups, stages
    int rpb, nslots, abl;                  // rows (positions) per block, blocks per channel group, timing ablations (PASN_WS_ABL)
    int xreg, greg, rreg, stage_bytes, lds_bytes;  // stage regions (X tile, gate rows, residual tile), whole KiB each
};
// squeeze-excite operands of the gate computed in the kernel's prologue (pool == NULL: off): partial rows [N][pool_blocks][Cin_p] of the stencil,
// 1 / positions, fc1 [cse][C] + bias, fc2 [C][cse] + bias
struct WsSe {
    const float* pool;
    int pool_blocks;
    float inv_positions;
    const float *w1, *b1, *w2, *b2;
    int C, cse;
};
// second conv of a chained pair (w2 == NULL: single conv): fragment-major weights, fp32 scale / bias (NULL = 1 / 0), output, channel counts, k-steps, activation
struct WsPair {
    const __bf16* w2;
    const float *scale2, *bias2;
    __bf16* y2;
    int Cout2, Cout2_p, nks2, act2;
};
WsGeom pw_ws_geom(const pasn_conv_desc& d, int dtype, bool has_gate, bool has_res, bool se_prologue = false, const pasn_conv_desc* d2 = nullptr);
int pw_ws_variant(const pasn_conv_desc& d, int dtype, bool has_gate, bool has_res);  // 7000 + KS * 10 + MT, or 0
int launch_pw_ws(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate, void* y,
                 const pasn_conv_desc& d, const WsGeom& g, hipStream_t s, const WsSe* se = nullptr, const WsPair* pair = nullptr);
// pwconv_xpair.hip: project conv of block i chained with the expand conv of block i+1 (bf16); 0 = not covered
int pw_xpair_ks(const pasn_conv_desc& d1, const pasn_conv_desc& d2, int dtype, int* ks2_out);
int launch_pw_xpair(const void* x, const void* w1, const float* s1, const float* b1, const void* res, const float* gate, void* y1,
                    const pasn_conv_desc& d1, const void* w2, const float* s2, const float* b2, void* y2,
                    const pasn_conv_desc& d2, hipStream_t s);
// LDS-tiled MFMA GEMM for large-K pointwise convs (gemm_pw.hip)
bool gemm_pw_applicable(const pasn_conv_desc& d, int dtype);
// dwmarch.hip: T-marching depthwise 3x3x3 stencil (bf16).  WT = 0: geometry / dtype not covered.
struct DwMarchGeom {
    int WT, CG, R, strips, Tc, bpc;  // outputs per strip, channel groups, items per block, strips per row, T chunk, blocks per clip
};
DwMarchGeom dw_march_geom(const pasn_conv_desc& d, int dtype);
bool dw_march_red_ok(const pasn_conv_desc& d, const DwMarchGeom& g);  // the dgrad + backward-sums instances cover this geometry
// fused squeeze-excite gate (gate == nullptr: off): fc1 [cse][C] + bias, fc2 [C][cse] + bias, gate out [N][Cp], counter [N] ints (zero)
struct DwRedArgs;
struct DwSeArgs {
    const float *w1, *b1, *w2, *b2;
    float* gate;
    int* counter;
    int cse;
};
int launch_dw_march(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool,
                    const pasn_conv_desc& d, const DwMarchGeom& g, hipStream_t s, const DwSeArgs& se = DwSeArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0},
                    int stats = 0,  // stats: `pool` is the batch-statistics partial buffer [N][bpc][2][Cp] (sum, sum of squares)
                    const struct DwRedArgs* red = nullptr,   // red: the launch is a dgrad whose outputs also feed the producer unit's backward sums
                    const float* shift = nullptr);           // stats without red: per-channel shift of the moments (NULL: 0)
// backward sums of the unit that PRODUCED the stencil's input (training): y = that unit's raw conv output (same geometry as the stencil's
// output), stat = its (mean, invstd, sc, sh) table, act = its activation; `pool` then receives [N][bpc][2][Cp] = (sum d', sum d' yhat)
struct DwRedArgs {
    const void* y;
    const float* stat;
    int act;
};
// dwmfma.hip: the stride-1 depthwise 3x3x3 stencil on the matrix cores (block-diagonal bf16 weight operands, LDS-DMA frame ring, T-marching); ok = 0: not covered
struct DwMfmaGeom {
    int ok, CT, CQ;            // channel tiles of 16, quads of 4 tiles (one block owns a quad)
    int BH, BW, RPT, RTH, RTW;  // outputs per region, output rows per 16-lane position tile, regions per frame
    int RP, NI;                // staged positions per frame, 1-KiB DMA instructions per frame
    int Tc, nT, upb, chunks, bpc;  // T chunk (+count), units (T chunk x region) per block, SE partial rows per clip, blocks per clip
    int abl;                   // timing ablations (PASN_DWMFMA_ABL; results are wrong when set)
};
DwMfmaGeom dw_mfma_geom(const pasn_conv_desc& d, int dtype);
int launch_dw_mfma(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool, const pasn_conv_desc& d,
                   const DwMfmaGeom& g, hipStream_t s, int stats = 0, const float* shift = nullptr);  // stats: pool = [N][chunks][2][Cp]: (sum, sum of squares) of (raw output - shift[c])
// dwtemporal.hip (round 5): depthwise (kt,1,1) conv, kt = 3 / 5, stride 1: thread = (position, 8 channels) marching along T with a register ring
bool dw_temporal_applicable(const pasn_conv_desc& d, int dtype);
int launch_dw_temporal(const void* x, const float* w, const float* scale, const float* bias, void* y, const pasn_conv_desc& d, int dtype, hipStream_t s);
// tconv_ws.hip (round 5): temporal (3,1,1) stride-1 conv, weight-stationary and T-marching (fragment-major weights, K = (dt, channel)); ok = 0: not covered
struct TcGeom {
    int ok, KSF, CT, ptiles, bm, lds;  // k-steps per frame (Cin_p / 16), channel tiles of 32, tiles per frame and their positions (<= 64), dynamic LDS
};
TcGeom tconv_geom(const pasn_conv_desc& d, int dtype, bool has_gate);
int launch_tconv_ws(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y, const pasn_conv_desc& d,
                    const TcGeom& g, hipStream_t s);
// dw_tz.hip (round 5): the stride-1 depthwise 3x3x3 stencil of planes at most 14 x 14 in Toeplitz form on a channel-planar LDS image (channels-last input transposed by
// ds_read_b64_tr_b16); ok = 0: not covered
struct DtGeom {
    int ok, CG;                // 16-channel groups
    int Tc, nT, lds;           // T chunk (+count = SE partial rows per clip = blocks per clip and channel group), dynamic LDS
    int abl;                   // tuning builds: 1 + the block whose wave 0 leaves shader-clock stamps (PASN_TZ_STAMPS)
};
DtGeom dw_tz_geom(const pasn_conv_desc& d, int dtype);
int launch_dw_tz(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool, const pasn_conv_desc& d, const DtGeom& g,
                 hipStream_t s);
// x3d_expdw.hip: expand conv (1x1x1 + BN + ReLU) -> depthwise 3x3x3 stride-(1,s,s) stencil, s = 1 or 2, (+BN, act, SE partial sums) in one launch, both on
// the matrix cores, the expanded activation only ever in LDS (the first block of an X3D stage); ok = 0: not covered
struct XeGeom {
    int ok, SS, KS, XS, xtb, lds;   // stride, expand k-steps in registers, 16-byte slots per staged x position (odd), bytes per x tile, dynamic LDS
    int CQ, RTH, RTW;               // 64-channel quads, regions (3 x 14 outputs) per frame
    int Tc, nT, upb, chunks, bpc;   // T chunk (+count), units per block, SE partial rows per clip, blocks per clip
    int abl;                        // timing ablations (PASN_EXPDW_ABL; results are wrong when set)
    int fuse;                       // steady-state step as one scheduling region (PASN_EXPDW_FUSE=0: expand, then stencil)
    // x3d_expdw_tz.hip (round 5): the Toeplitz formulation on a channel-planar image, stride 1; tz = 1: that kernel runs, the fields above are unused
    int tz, tzCG, tzRTH, tzRTW;     // 16-channel groups, regions (8 x 14 outputs) per frame
    int tzTc, tznT, tzChunks;       // T chunk (+count), SE partial rows per clip (= units per clip)
    int tzXS, tzLds;                // 16-byte slots per staged x position, dynamic LDS
};
XeGeom xe_geom(const pasn_conv_desc& de, const pasn_conv_desc& d, int dtype);
void xe_geom_tz(XeGeom& g, const pasn_conv_desc& de, const pasn_conv_desc& d);
int launch_x3d_expdw_tz(const void* x, const void* wa, const float* ba, const float* w, const float* scale, const float* bias, void* y, float* pool,
                        const pasn_conv_desc& de, const pasn_conv_desc& d, const XeGeom& g, hipStream_t s);
int launch_x3d_expdw(const void* x, const void* wa, const float* sa, const float* ba, const float* w, const float* scale, const float* bias,
                     void* y, float* pool, const pasn_conv_desc& de, const pasn_conv_desc& d, const XeGeom& g, hipStream_t s);
// x3d_edp.hip: a whole X3D block of the 7 x 7 stage (expand -> stencil -> project [-> next expand]) in one launch, both wide tensors in LDS; ok = 0: not covered
struct EdpGeom {
    int ok, KSA, KSC, CTA, CTC, CTN, NQ;                       // k-steps / 32-channel tiles of the expand, project and next expand convs; quads of the inner width
    int fimg_off, dch_off, tab_off, cst_off, lds_bytes;       // LDS layout: [x image] [frame images | block-output image] [stencil-output chunk] [row table] [tables]
    int tiles, tpb, grid, stamps;
};
EdpGeom edp_geom(const pasn_conv_desc& da, const pasn_conv_desc& dd, const pasn_conv_desc& dc, const pasn_conv_desc* dn, int dtype);
// x3d_pe.hip: project conv (+ squeeze-excite gate in the prologue) chained with the next block's expand conv for the 432-channel X3D stage,
// weights streamed per tile (nothing stationary); ok = 0: not covered
struct PeGeom {
    int ok, KSC, KSA, CTC, CTA;                               // (even) k-steps and 32-channel output tiles of the two convs
    int R, RTn, DPL, XPL;                                     // rows per tile, 32-row tiles, 16-byte slots per row of the two images
    int xt_off, gate_off, tab_off, cst_off, lds_bytes;        // LDS layout
    int tiles, tpb, grid, se, stamps;
};
PeGeom pe_geom(const pasn_conv_desc& d1, const pasn_conv_desc& d2, int dtype, bool se, int cse);
// pwconv_tiny.hip: fp32 / bf16 pointwise conv on few positions (the image heads): one wave per 32 x 32 output tile, operands straight from global
bool pw_tiny_applicable(const pasn_conv_desc& d, int dtype, bool has_gate);
int launch_pw_tiny(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y, const pasn_conv_desc& d,
                   int dtype, hipStream_t s);
// igemm.hip: windowed dense convs (bf16) as an implicit GEMM with direct-to-LDS staging; NT = 0: not covered
int igemm_nt(const pasn_conv_desc& d, int dtype);
// first_conv_mfma.hip: the 7x7 stride-2 stems on the matrix cores (bf16 out); slot < 0: not covered
int first_conv_mfma_slot(const pasn_conv_desc& d, int out_dtype);
int launch_first_conv_mfma(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc& d,
                           int in_dtype, float in_a, float in_b, int o, hipStream_t s);
// stem_mfma.hip: the X3D stem (conv_xy + conv_t + BN + ReLU) as one MFMA map, bf16 out
int x3d_stem_mfma_supported(const pasn_conv_desc& d, int out_dtype);
int launch_x3d_stem_mfma(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc& d, int in_dtype,
                         float in_a, float in_b, hipStream_t s);
// wgrad_halo.hip: weight gradient of the stride-1 "same" (1,3,3) / (3,1,1) convs (bf16) through a partial buffer; 0 bytes = not covered
size_t wgrad_halo_workspace_bytes(const pasn_conv_desc& d, int dtype);
bool wgrad_halo(const void* x, const void* dy, float* dw, void* ws, const pasn_conv_desc& d, int dtype, hipStream_t s);
// igemm_halo.hip: the same for stride-1 "same" (1,k,k) / (3,1,1) layers with the activation halo tile kept in LDS across the taps
int igemm_halo_mode(const pasn_conv_desc& d);
bool igemm_halo_fits(const pasn_conv_desc& d, int mode, int nt, int mt);
int launch_igemm_halo(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y,
                      const pasn_conv_desc& d, int mode, int nt, int mt, hipStream_t s);
int launch_igemm(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y,
                 const pasn_conv_desc& d, int nt, hipStream_t s);
template <typename T>
int launch_gemm_pw(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate,
                   void* y, const pasn_conv_desc& d, hipStream_t s);

// head_chain.hip: head B with the intermediate maps resident in LDS (bf16, D = 256)
bool xproto_chain_supported(const pasn_xproto_desc& d, int dtype);
int xproto_chain_tiles(const pasn_xproto_desc& d);
int launch_xproto_chain(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                        const void* w4, const float* b4, const void* w5, float* occ, float* slabs, const pasn_xproto_desc& d, hipStream_t s);

// ---- dtype traits: one MFMA "k-chunk" is the 16 bytes a lane feeds to the matrix core ---------------
template <typename T>
struct Traits;
template <>
struct Traits<float> {
    static constexpr int CH = 4;      // elements per 16-byte lane chunk
    static constexpr int KSTEP = 8;   // K covered by the two lane halves of one fragment pair
    using frag = f32x4;
};
template <>
struct Traits<__bf16> {
    static constexpr int CH = 8;
    static constexpr int KSTEP = 16;
    using frag = bf16x8;
};

// D(32x32) += A(32xKSTEP) * B(KSTEPx32).  Lane l = (r = l & 31, h = l >> 5) supplies, for A row r and for
// B column r, the CH consecutive k values  k0 + h*CH .. k0 + h*CH + CH-1.
// bf16: one v_mfma_f32_32x32x16_bf16.  fp32: four v_mfma_f32_32x32x2_f32; MFMA j contracts the k pair
// {j, CH + j}; any k order is fine as long as A and B agree, which they do by construction.
__device__ __forceinline__ void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}
// Accumulator element `reg` of lane l sits at column (l & 31), row acc_row(reg, l >> 5).
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

template <typename T>
__device__ __forceinline__ typename Traits<T>::frag zero_frag() {
    typename Traits<T>::frag z;
#pragma unroll
    for (int j = 0; j < Traits<T>::CH; ++j) z[j] = (T)0.0f;
    return z;
}
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag load_frag(const T* p) {
    return *reinterpret_cast<const typename Traits<T>::frag*>(p);
}

// ---- 8-channel groups (the unit of the stencil / pooling kernels) ---------------------------------------
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = a[j];
        v[4 + j] = b[j];
    }
}
__device__ __forceinline__ void load8(const __bf16* p, float (&v)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = v[j];
        b[j] = v[4 + j];
    }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(__bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = a;
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = v[j];
    *reinterpret_cast<f32x4*>(p) = a;
}
__device__ __forceinline__ void store4(__bf16* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x4*>(p) = a;
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = a[j];
}
__device__ __forceinline__ void load4(const __bf16* p, float (&v)[4]) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)a[j];
}

// 8 channels held as raw 16-byte words (register prefetch buffers) -> fp32.  bf16 -> fp32 is a 16-bit shift / mask.
template <typename T>
__device__ __forceinline__ void raw_to_f8(const uint4 (&r)[(8 * sizeof(T)) / 16], float (&v)[8]);
template <>
__device__ __forceinline__ void raw_to_f8<__bf16>(const uint4 (&r)[1], float (&v)[8]) {
    const unsigned w[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <>
__device__ __forceinline__ void raw_to_f8<float>(const uint4 (&r)[2], float (&v)[8]) {
    v[0] = __uint_as_float(r[0].x); v[1] = __uint_as_float(r[0].y); v[2] = __uint_as_float(r[0].z); v[3] = __uint_as_float(r[0].w);
    v[4] = __uint_as_float(r[1].x); v[5] = __uint_as_float(r[1].y); v[6] = __uint_as_float(r[1].z); v[7] = __uint_as_float(r[1].w);
}

// Sum of `rows` floats pp[b * stride], b = 0 .. rows - 1, IN INDEX ORDER (bitwise reproducible), with the loads of eight rows in flight
// at a time whatever `rows` is.  (A `for (; b < rows; ++b) sum += pp[...]` tail is one dependent L2 round trip per row: the squeeze-excite
// gates of the 14 x 14 and 7 x 7 stages have 4 and 2 partial rows per clip and read them one after the other -- ~1 us each, per clip.)
__device__ __forceinline__ float sum_rows_in_order(const float* pp, int rows, long stride, float sum = 0.0f) {
    for (int b = 0; b < rows; b += 8) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = pp[(long)min(b + e, rows - 1) * stride];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (b + e < rows) sum += t[e];
    }
    return sum;
}

// Logical block id for physical workgroup `bid` of an `nwg`-block 1-D grid such that the blocks sharing an XCD
// (bid % 8, observed round-robin dispatch) form ONE contiguous logical range.  Placement affects speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each); an IEEE division here costs ~10 VALU instructions per element
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// ReLU as a signed-integer max on the bit pattern (a float is negative iff its bits are a negative int32): ONE v_max_i32.  fmaxf costs
// two vector instructions wherever the compiler cannot prove its operand free of signalling NaNs (a value that went through a lane swap or
// a run-time activation switch: v_max_f32 v, v, v to quieten it first) -- 8 of ~50 per 8 outputs in epilogues bound by vector issue.
// -0 -> +0; a NaN with the sign bit clear passes through (as torch.relu propagates it), one with the sign bit set becomes 0.
__device__ __forceinline__ float relu_f32(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case PASN_ACT_RELU: return relu_f32(v);
        case PASN_ACT_SIGMOID: return sigmoidf_(v);
        case PASN_ACT_SWISH: return v * sigmoidf_(v);
        case PASN_ACT_ABS: return fabsf(v);
        default: return v;
    }
}

// derivative of the activation at pre-activation u (training backward passes)
__device__ __forceinline__ float act_grad(float u, int act) {
    switch (act) {
        case PASN_ACT_RELU: return u > 0.0f ? 1.0f : 0.0f;
        case PASN_ACT_SIGMOID: {
            const float s = sigmoidf_(u);
            return s * (1.0f - s);
        }
        case PASN_ACT_SWISH: {
            const float s = sigmoidf_(u);
            return s * (1.0f + u * (1.0f - s));
        }
        case PASN_ACT_ABS: return u > 0.0f ? 1.0f : (u < 0.0f ? -1.0f : 0.0f);
        default: return 1.0f;
    }
}

// dv[j] *= act'(u[j]) for a whole register vector behind ONE wave-uniform switch (act_grad per element compiles to a ladder of scalar
// compares and branches per ELEMENT: 348 branches in bn_bwd_apply_kernel's loop body)
template <int N>
__device__ __forceinline__ void act_grad_mul(float (&dv)[N], const float (&u)[N], int act) {
    switch (act) {
        case PASN_ACT_RELU:
#pragma unroll
            for (int j = 0; j < N; ++j) dv[j] = u[j] > 0.0f ? dv[j] : 0.0f;
            break;
        case PASN_ACT_SIGMOID:
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float s = sigmoidf_(u[j]);
                dv[j] *= s * (1.0f - s);
            }
            break;
        case PASN_ACT_SWISH:
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float s = sigmoidf_(u[j]);
                dv[j] *= s * (1.0f + u[j] * (1.0f - s));
            }
            break;
        case PASN_ACT_ABS:
#pragma unroll
            for (int j = 0; j < N; ++j) dv[j] *= u[j] > 0.0f ? 1.0f : (u[j] < 0.0f ? -1.0f : 0.0f);
            break;
        default:
            break;
    }
}

// Activation of a whole register vector behind ONE wave-uniform switch.  (hipcc does not hoist the per-element
// switch of apply_act out of unrolled epilogue loops: every element got its own scalar branch ladder.)
template <int N>
__device__ __forceinline__ void act_vec(float (&v)[N], int act) {
    switch (act) {
        case PASN_ACT_RELU:
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = relu_f32(v[j]);
            break;
        case PASN_ACT_SIGMOID:
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = sigmoidf_(v[j]);
            break;
        case PASN_ACT_SWISH:
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = v[j] * sigmoidf_(v[j]);
            break;
        case PASN_ACT_ABS:
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = fabsf(v[j]);
            break;
        default:
            break;
    }
}
// Zero the elements of a channel vector that lie at or beyond the real channel count (`nvalid` = C - first channel).
template <int N>
__device__ __forceinline__ void mask_tail(float (&v)[N], int nvalid) {
    if (nvalid < N) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j >= nvalid) v[j] = 0.0f;
    }
}

}  // namespace pasn
