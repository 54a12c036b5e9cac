// Head B as ONE launch per tile of positions (bf16, D = 256): add-on (2 convs) + occurrence module (3 convs) + the tile's share of
// the occurrence-weighted pooling, with every intermediate map kept in LDS.
//
// Reference: Video_XProtoNet.forward (Video_XProtoNet.py:66-98): x -> add_on_layers -> f; x -> occurrence_module -> |.| -> occ;
// features_extracted[n][p][d] = sum_s occ[n][p][s] f[n][d][s]; cosine similarity and last layer follow (xproto_finish_kernel).
//
// head_xproto.hip runs this as 5 pointwise-conv launches + pooling + finish: at the headline shape (32 x 784 positions, 192 trunk
// channels) that is 7 launches of 5-27 us (~100 us) for 10 GFLOP and ~30 MB -- every launch at its fixed cost (ramp, weight fetch,
// drain), and f / f1 / o1 / o2 / occ each written to and read back from memory.  Here a block owns R <= 104 consecutive positions of
// ONE clip (S = 784: 8 tiles of 98 -> 256 blocks, one per CU; trunks with up to 256 channels -- R(2+1)D-18[:-3] -- take 96-row
// tiles, see HcGeom) and walks the chain with the tile resident:
//   x tile  --c1--> f1 --c2--> f^T            (add-on;   ^T = stored [channel][position] for the pooling's K = position)
//   x tile  --c3--> o1 --c4--> o2 --c5--> occ^T (occurrence module, |.| fused)
//   slab[p][d] = occ^T[p][:] . f^T[d][:]       (one more MFMA pass; the slabs of a clip's tiles are summed in fixed order by the finish kernel)
// Each wave owns one 32-channel tile of the stage's output for all four 32-position sub-tiles (c4: 4 tiles x 2 halves, c5: 2 x 4), its
// weight fragments (fragment-major, whole K) are fetched straight into registers while the previous stage's epilogue runs, activations
// are read from LDS as 16-byte k-contiguous pieces (odd row strides: conflict-free ds_read_b128).  Row-major outputs (f1, o1, o2) use
// MFMA(A = weights, B = activations): positions on the lanes, v_permlane32_swap gives a lane 8 consecutive channels -> one 16-byte LDS
// write.  The transposed outputs (f^T, occ^T) use MFMA(A = activations, B = weights): channels on the lanes, the same swap gives 8
// consecutive POSITIONS of one channel -> one 16-byte write into the [channel][position] image.  Rounding points are those of the
// separate launches (every map rounded to bf16 between convs; fp32 accumulation; exact bf16 x bf16 products in the pooling).
#include "common.h"

namespace pasn {

typedef __attribute__((address_space(3))) void* hc_lds_ptr_t;

// Tile geometry per instance: KS1 = k-steps of the trunk channels, ROWS = staged positions per tile (a whole number of 8-position pieces).
//   <12, 104>: trunk stride <= 192 (X3D): x 41.6 KB | f1 / o1 / occ^T 54.9 KB | f^T 61.4 KB = 158 KB; 104 rows = 3.25 sub-tiles of 32 (S = 784 -> 8
//              tiles of 98 per clip = one block per CU at 32 clips)
//   <16, 96>:  trunk stride <= 256 (R(2+1)D-18[:-3]): x 50.7 | 50.7 | 53.2 = 155 KB; 96 rows = exactly 3 sub-tiles
template <int KS1, int ROWS>
struct HcGeom {
    static constexpr int PIECES = ROWS / 8;
    static constexpr int NST = (ROWS + 31) / 32;          // 32-position sub-tiles that hold any staged row
    static constexpr int XS = 2 * KS1 + 1;                // 16-byte slots per x row (odd)
    static constexpr int B1S = 33;                        // f1 / o1 rows: 256 channels + pad slot
    static constexpr int O2S = 17;                        // o2 rows: 128 channels + pad slot
    static constexpr int TS = PIECES + 1 + (PIECES & 1);  // transposed rows: the pieces + pad, odd
    static constexpr int XR_BYTES = ROWS * XS * 16;       // x tile; later o2
    static constexpr int B1_BYTES = ROWS * B1S * 16;      // f1, then o1, then occ^T
    static constexpr int B2_BYTES = 256 * TS * 16;        // f^T
    static constexpr int LDS = XR_BYTES + B1_BYTES + B2_BYTES;
    static_assert(ROWS % 8 == 0 && XR_BYTES >= ROWS * O2S * 16 && B1_BYTES >= 64 * TS * 16 && LDS <= 160 * 1024, "regions must fit");
};
constexpr unsigned HC_OOB = 0x80000000u;

struct HcArgs {
    const __bf16 *x, *w1, *w2, *w3, *w4, *w5;  // fragment-major weights: add_on 0 / 2, occurrence_module 0 / 2 / 4
    const float *b1, *b2, *b3, *b4;            // their biases (occurrence_module.4 has none)
    float *slabs, *occ;
    int N, S, Cbp, nks1, P, G, R, full;
};

__device__ __forceinline__ void hc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fragments K0 .. K1 - 1 of channel tile ct (whole-K fragment-major rows of nks steps); steps beyond nks are zero fragments
template <int KS, int K0 = 0, int K1 = KS>
__device__ __forceinline__ void hc_load_w(bf16x8 (&W)[KS], const __bf16* __restrict__ w, int ct, int nks, int lane) {
    const __bf16* b = w + ((long)ct * nks * 64 + lane) * 8;
#pragma unroll
    for (int ks = K0; ks < K1; ++ks) W[ks] = load_frag<__bf16>(b + (size_t)(ks < nks ? ks : nks - 1) * 512);
#pragma unroll
    for (int ks = K0; ks < K1; ++ks)
        if (ks >= nks) W[ks] = zero_frag<__bf16>();  // wave-uniform
}

// acc[st] += W x tile(rows[st]) over KS k-steps; SWAP: the activations are the A operand (positions on the accumulator rows)
template <int KS, int NST, bool SWAP>
__device__ __forceinline__ void hc_mma(f32x16 (&acc)[NST], const bf16x8 (&W)[KS], const char* in, int stride, const int (&rows)[NST], int h) {
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[st][e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 b[NST];
#pragma unroll
        for (int st = 0; st < NST; ++st) b[st] = *reinterpret_cast<const bf16x8*>(in + (rows[st] * stride + 2 * ks + h) * 16);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            if (SWAP) acc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[st], W[ks], acc[st], 0, 0, 0);
            else acc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[ks], b[st], acc[st], 0, 0, 0);
        }
    }
}

// registers 8 pr + q and 8 pr + 4 + q of the two half-waves -> 8 consecutive accumulator ROWS 16 pr + 8 h .. + 7 of this lane's column
__device__ __forceinline__ void hc_swap8(const f32x16& a, int pr, float (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * pr + q]), __float_as_uint(a[8 * pr + 4 + q]), false, false);
        v[q] = __uint_as_float(sw[0]);
        v[4 + q] = __uint_as_float(sw[1]);
    }
}

// row-major epilogue (A = weights): out[position][channel], bias + ReLU, rounded to bf16
template <int NST, int ROWS>
__device__ __forceinline__ void hc_epi_rows(const f32x16 (&acc)[NST], const float* __restrict__ bias, int ct, int st0, char* out, int stride,
                                            int c, int h) {
    float bs[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) load8(bias + ct * 32 + 16 * pr + 8 * h, bs[pr]);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int row = 32 * (st0 + st) + c;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float v[8];
            hc_swap8(acc[st], pr, v);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaxf(v[e] + bs[pr][e], 0.0f);
            if (row < ROWS) *reinterpret_cast<bf16x8*>(out + (row * stride + 4 * ct + 2 * pr + h) * 16) = o;
        }
    }
}

// transposed epilogue (A = activations): out[channel][position]; ABS: |.| and zeros at positions >= valid (the occurrence map)
template <int NST, bool ABS, int PIECES, int TS>
__device__ __forceinline__ void hc_epi_cols(const f32x16 (&acc)[NST], float bias, int ct, int st0, char* out, int valid, int c, int h) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int piece = 4 * (st0 + st) + 2 * pr + h;
            float v[8];
            hc_swap8(acc[st], pr, v);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = v[e] + bias;
                if (ABS) f = piece * 8 + e < valid ? fabsf(f) : 0.0f;
                o[e] = (__bf16)f;
            }
            if (piece < PIECES) *reinterpret_cast<bf16x8*>(out + ((ct * 32 + c) * TS + piece) * 16) = o;
        }
    }
}

template <int KS1, int ROWS>
__global__ __launch_bounds__(512) void xproto_chain_kernel(HcArgs a) {
    using G = HcGeom<KS1, ROWS>;
    constexpr int HC_ROWS = ROWS, HC_PIECES = G::PIECES, HC_KS1 = KS1, HC_XS = G::XS, HC_B1S = G::B1S, HC_O2S = G::O2S, HC_TS = G::TS, NST = G::NST;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const XR = smem;
    char* const B1 = smem + G::XR_BYTES;
    char* const B2 = B1 + G::B1_BYTES;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x / a.G, g = blockIdx.x - n * a.G;
    const int s0 = g * a.R;
    const int valid = min(a.R, a.S - s0);  // >= 1 (host: (G - 1) R < S)
    const unsigned m0 = (unsigned)n * (unsigned)a.S + (unsigned)s0;
    const bool full = a.full != 0;

    {  // x tile by LDS-DMA: slot s of the image = (row s / XS, piece s % XS); pieces beyond the channels / rows beyond the tile arrive as zeros
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (unsigned)a.N * (unsigned)a.S * (unsigned)a.Cbp * 2u, 0x00020000);
        const int ppr = a.Cbp >> 3;
        constexpr int nix = (HC_ROWS * HC_XS + 63) / 64;  // the last instruction may spill zero slots into B1 (written later)
        for (int j = wave; j < nix; j += 8) {
            const int s = j * 64 + lane;
            const int r = s / HC_XS, p = s - r * HC_XS;
            const unsigned off = (r < valid && p < ppr) ? (m0 + (unsigned)r) * (unsigned)a.Cbp * 2u + (unsigned)p * 16u : HC_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (hc_lds_ptr_t)(XR + j * 1024), 16, (int)off, 0, 0, 0);
        }
    }
    int rows4[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) rows4[st] = min(32 * st + c, HC_ROWS - 1);

    f32x16 acc[NST];
    bf16x8 W3[HC_KS1];
    if (full) {  // kernel-uniform
        bf16x8 W1[HC_KS1];
        hc_load_w<HC_KS1>(W1, a.w1, wave, a.nks1, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        hc_barrier();
        // (half of the next stage's weight fragments are requested BEFORE this stage's MFMAs, the rest behind them: all of them up front
        // does not fit the registers -- 19 spilled, 28.8 us -- and all of them behind leaves an L2 round trip per stage exposed)
        // ---- c1: x -> f1 = relu(.), rows of B1 ----
        bf16x8 W2[16];
        hc_load_w<16, 0, 8>(W2, a.w2, wave, 16, lane);
        hc_mma<HC_KS1, NST, false>(acc, W1, XR, HC_XS, rows4, h);
        hc_load_w<16, 8, 16>(W2, a.w2, wave, 16, lane);
        hc_epi_rows<NST, HC_ROWS>(acc, a.b1, wave, 0, B1, HC_B1S, c, h);
        hc_barrier();
        // ---- c2: f1 -> f^T (no activation), B2 ----
        hc_load_w<HC_KS1, 0, 6>(W3, a.w3, wave, a.nks1, lane);
        const float bias2 = a.b2[wave * 32 + c];
        hc_mma<16, NST, true>(acc, W2, B1, HC_B1S, rows4, h);
        hc_load_w<HC_KS1, 6, HC_KS1>(W3, a.w3, wave, a.nks1, lane);
        hc_epi_cols<NST, false, HC_PIECES, HC_TS>(acc, bias2, wave, 0, B2, HC_ROWS, c, h);
        hc_barrier();  // everyone is past its reads of f1
    } else {
        hc_load_w<HC_KS1>(W3, a.w3, wave, a.nks1, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        hc_barrier();
    }
    // ---- c3: x -> o1 = relu(.), rows of B1 ----
    const int ct4 = wave & 3, sp4 = wave >> 2;
    bf16x8 W4[16];
    hc_load_w<16, 0, 8>(W4, a.w4, ct4, 16, lane);
    hc_mma<HC_KS1, NST, false>(acc, W3, XR, HC_XS, rows4, h);
    hc_load_w<16, 8, 16>(W4, a.w4, ct4, 16, lane);
    hc_epi_rows<NST, HC_ROWS>(acc, a.b3, wave, 0, B1, HC_B1S, c, h);
    hc_barrier();  // o1 complete; everyone is past its reads of the x tile
    // ---- c4: o1 -> o2 = relu(.), rows of XR: wave = (channel tile, half of the positions) ----
    {
        f32x16 acc2[2];
        const int rows2[2] = {min(64 * sp4 + c, HC_ROWS - 1), min(64 * sp4 + 32 + c, HC_ROWS - 1)};
        const int pt5 = wave & 1, st5 = wave >> 1;
        bf16x8 W5[8];
        hc_load_w<8>(W5, a.w5, pt5, 8, lane);
        hc_mma<16, 2, false>(acc2, W4, B1, HC_B1S, rows2, h);
        hc_epi_rows<2, HC_ROWS>(acc2, a.b4, ct4, 2 * sp4, XR, HC_O2S, c, h);
        hc_barrier();  // o2 complete; everyone is past its reads of o1
        // ---- c5: o2 -> occ^T = |.|, rows of B1: wave = (prototype tile, 32-position sub-tile) ----
        f32x16 acc1[1];
        const int rows1[1] = {min(32 * st5 + c, HC_ROWS - 1)};
        hc_mma<8, 1, true>(acc1, W5, XR, HC_O2S, rows1, h);
        hc_epi_cols<1, true, HC_PIECES, HC_TS>(acc1, 0.0f, pt5, st5, B1, valid, c, h);
    }
    hc_barrier();
    // ---- occurrence map rows of this tile: occ[n][p][s0 + s] (fp32 of the bf16 map, as the separate launches store it) ----
    for (int idx = threadIdx.x; idx < a.P * valid; idx += 512) {
        const int p = idx / valid, s = idx - p * valid;
        a.occ[((long)n * a.P + p) * a.S + s0 + s] = (float)*reinterpret_cast<const __bf16*>(B1 + p * HC_TS * 16 + s * 2);
    }
    if (!full) return;
    // ---- pooling: slab[p][d] = sum_s occ^T[p][s] f^T[d][s]; wave = 32 feature columns, both prototype tiles ----
    f32x16 accp[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int e = 0; e < 16; ++e) accp[pt][e] = 0.0f;
    const bool two = a.P > 32;
    constexpr int PKS = (HC_PIECES + 1) / 2;  // k-steps of 16 positions
#pragma unroll
    for (int ks = 0; ks < PKS; ++ks) {
        const bool dead = 2 * ks + h >= HC_PIECES;  // 104 rows = 6.5 k-steps: the last half piece does not exist, both operands are zero there
        const int slot = dead ? 0 : 2 * ks + h;
        bf16x8 fb = *reinterpret_cast<const bf16x8*>(B2 + ((wave * 32 + c) * HC_TS + slot) * 16);
        bf16x8 o0 = *reinterpret_cast<const bf16x8*>(B1 + (c * HC_TS + slot) * 16);
        bf16x8 o1 = *reinterpret_cast<const bf16x8*>(B1 + ((32 + c) * HC_TS + slot) * 16);
        if (2 * ks + 1 >= HC_PIECES) {
            typedef unsigned hc_u32x4 __attribute__((ext_vector_type(4)));
            hc_u32x4 z = {0u, 0u, 0u, 0u};
            fb = dead ? __builtin_bit_cast(bf16x8, z) : fb;
            o0 = dead ? __builtin_bit_cast(bf16x8, z) : o0;
            o1 = dead ? __builtin_bit_cast(bf16x8, z) : o1;
        }
        accp[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o0, fb, accp[0], 0, 0, 0);
        if (two) accp[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o1, fb, accp[1], 0, 0, 0);
    }
    float* slab = a.slabs + ((long)n * a.G + g) * a.P * 256 + wave * 32 + c;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = 32 * pt + acc_row(r, h);
            if (p < a.P) slab[(long)p * 256] = accp[pt][r];
        }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------------
static int hc_rows(const pasn_xproto_desc& d) { return d.Cbp <= 192 ? 104 : 96; }

bool xproto_chain_supported(const pasn_xproto_desc& d, int dtype) {
    if (const char* e = tune("PASN_HEAD_CHAIN"))
        if (e[0] == '0') return false;
    if (dtype != PASN_BF16) return false;
    if (d.D != 256 || d.Dp != 256 || d.Hd != 128 || d.Hp != 128) return false;
    if (d.P < 1 || d.P > 64) return false;
    if (d.Cbp % 8 != 0 || d.Cbp > 256 || d.Cbp < 8) return false;
    if ((long)d.N * d.S * d.Cbp * 2 >= (1L << 31)) return false;
    return d.N > 0 && d.S > 0;
}
int xproto_chain_tiles(const pasn_xproto_desc& d) { return ceil_div(d.S, hc_rows(d)); }

int launch_xproto_chain(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                        const void* w4, const float* b4, const void* w5, float* occ, float* slabs, const pasn_xproto_desc& d, hipStream_t s) {
    HcArgs a;
    a.x = (const __bf16*)x;
    a.w1 = (const __bf16*)w1; a.w2 = (const __bf16*)w2; a.w3 = (const __bf16*)w3; a.w4 = (const __bf16*)w4; a.w5 = (const __bf16*)w5;
    a.b1 = b1; a.b2 = b2; a.b3 = b3; a.b4 = b4;
    a.slabs = slabs;
    a.occ = occ;
    a.N = d.N; a.S = d.S; a.Cbp = d.Cbp;
    a.nks1 = ceil_div(d.Cbp, 16);
    a.P = d.P;
    a.G = xproto_chain_tiles(d);
    a.R = ceil_div(d.S, a.G);
    a.full = d.mode == 0 ? 1 : 0;
    const dim3 grid((unsigned)(d.N * a.G)), block(512);
    if (d.Cbp <= 192) {
        constexpr size_t lds = HcGeom<12, 104>::LDS;
        PASN_MAX_LDS(160 * 1024, xproto_chain_kernel<12, 104>);
        hipLaunchKernelGGL((xproto_chain_kernel<12, 104>), grid, block, lds, s, a);
    } else {
        constexpr size_t lds = HcGeom<16, 96>::LDS;
        PASN_MAX_LDS(160 * 1024, xproto_chain_kernel<16, 96>);
        hipLaunchKernelGGL((xproto_chain_kernel<16, 96>), grid, block, lds, s, a);
    }
    return check_launch("xproto_chain_kernel");
}

}  // namespace pasn
