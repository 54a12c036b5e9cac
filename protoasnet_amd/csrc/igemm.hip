// Windowed dense convolution (bf16) as an implicit GEMM with DIRECT-TO-LDS staging -- the (1,3,3) / (3,1,1) / 3x3 / strided 1x1 convs of the
// R(2+1)D-18 and ResNet-18 trunks (reference resnet_features.py:49-66,202-213,307-327; torchvision Conv2Plus1D / BasicBlock).
//
// Round 1's kernel (gemm_pw.hip: 128 x 128 x 32 tile, register staging) ran R(2+1)D's 64 -> 144 (1,3,3) layer at 326 TF/s = 13 % of the
// bf16 MFMA peak; its PMC profile (profiles/r01i_pmc_gemm_conv.txt): matrix pipe 26 % busy, VALU issue 60 % (139 VALU instructions per
// 32-wide K slice next to 8 MFMAs: per-piece bounds masks, 64-bit offset chains, register-stage moves, selects and ds_writes of the
// stash), 33 M LDS bank-conflict cycles of 95 M, and the activation tile re-read once per 128-channel tile (144 channels = two tiles,
// the second 7/8 empty).  This kernel:
//   * stages both tiles with global_load_lds_dwordx4 (LDS-DMA): no staging registers, no selects, no ds_write; a piece outside the image
//     or the K range is fetched from a 256-byte page of zeros instead (one select on the ADDRESS);
//   * LDS rows are 64 bytes (BK = 32) in a lane-linear image as the DMA requires; the logical 16-byte piece q of row r sits in slot
//     q ^ ((r >> 2) & 3) -- conflict-free for the ds_read_b128 lane groups, and the same for every row group a lane serves, so a lane
//     always fetches one fixed piece index (one (tap, channel) decode per K slice and lane);
//   * the block tile is 256 positions x NT*32 channels with a wave owning 64 positions x ALL the block's channels (NT = 5 covers 144 or
//     160 channels, NT = 4: 128, NT = 2: 64): the activation tile is read once per 160 / 128 channels, 2 + NT fragment reads feed 2*NT
//     MFMAs;
//   * two LDS stages, the next slice's DMA issued before the current slice's MFMAs, one barrier per slice, two blocks per CU.
// A = weights (row = output channel), B = activations (column = position): the accumulator has the position on the lane, which the
// epilogue turns into whole-row 16-byte stores through a wave-private LDS image (scale / bias, residual, activation fused).
#include "common.h"
#include "igemm_epilogue.h"

namespace pasn {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;


constexpr int IG_BK = 32;

// NT channel tiles x MT position tiles (of 32) per WAVE; the block is 4 waves along the positions: BM = 128 * MT, BN = 32 * NT.
template <int NT, int MT>
__global__ __launch_bounds__(256, 2) void igemm_glds_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            const __bf16* __restrict__ res, __bf16* __restrict__ y, pasn_conv_desc d,
                                                            int scb_off) {
    constexpr int BN = NT * 32, IG_BM = 128 * MT, XG = 2 * MT;  // XG: 16-row DMA groups of the activation tile per wave
    constexpr int XBYTES = IG_BM * 64, WBYTES = BN * 64, STAGE = XBYTES + WBYTES;
    constexpr int WGROUPS = BN / 16;               // 16-row DMA groups of the weight tile
    constexpr int OROW = BN + 8;                   // epilogue image row (elements)
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2][STAGE]; the epilogue images alias it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    // consecutive position tiles on ONE XCD: the temporal / spatial taps of neighbouring tiles re-read the same rows, and only tiles that
    // share an L2 can hit on them (round-robin placement sent every row to three XCDs: 632 MB fetched for a 231 MB input, PMC)
    const long m0 = (long)(d.w_frag == 7 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x)) * IG_BM;
    const int n0 = blockIdx.y * BN;
    const int Cin_p = d.Cin_p, Cout_p = d.Cout_p, kc = d.w_kc;
    const int taps = d.kt * d.kh * d.kw;
    const int Ktot = taps * kc;
    const int nk = (Ktot + IG_BK - 1) / IG_BK;

    // ---- DMA roles: a wave-instruction fills 16 rows x 4 slots; lane -> (row l >> 2, slot l & 3), logical piece q (fixed per lane) ----
    const int q = (lane & 3) ^ ((lane >> 4) & 3);
    const int rsub = lane >> 2;
    int xoff[XG];           // window origin of this lane's activation rows (element offset), rows (wave*XG + i)*16 + rsub
    unsigned tapmask[XG];   // taps of that window inside the image (bit per tap; <= 27 taps)
#pragma unroll
    for (int i = 0; i < XG; ++i) {
        const long m = m0 + (wave * XG + i) * 16 + rsub;
        xoff[i] = 0;
        tapmask[i] = 0;
        if (m < M) {
            const int ow = (int)(m % d.Wo);
            long r = m / d.Wo;
            const int oh = (int)(r % d.Ho);
            r /= d.Ho;
            const int ot = (int)(r % d.To);
            const int on = (int)(r / d.To);
            const int t0 = ot * d.st - d.pt, h0 = oh * d.sh - d.ph, w0 = ow * d.sw - d.pw;
            xoff[i] = (((on * d.Ti + t0) * d.Hi + h0) * d.Wi + w0) * Cin_p;  // host guarantees the tensor fits 31 bits of elements
            int tp = 0;
            for (int a = 0; a < d.kt; ++a)
                for (int b2 = 0; b2 < d.kh; ++b2)
                    for (int e = 0; e < d.kw; ++e, ++tp)
                        if ((unsigned)(t0 + a) < (unsigned)d.Ti && (unsigned)(h0 + b2) < (unsigned)d.Hi && (unsigned)(w0 + e) < (unsigned)d.Wi)
                            tapmask[i] |= 1u << tp;
        }
    }
    int woff[3];           // this lane's weight rows (element offset of the row start), groups wave, wave + 4, wave + 8
    bool wrow_ok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int n = n0 + (wave + 4 * i) * 16 + rsub;
        wrow_ok[i] = n < d.w_rows;
        woff[i] = n * Ktot;
    }
    // (tap, channel) of this lane's piece in slice 0; advanced by one slice per issue
    int f_ci = q * 8, f_tap = 0, f_da = 0, f_db = 0, f_de = 0;
    {
        f_tap = f_ci / kc;
        f_ci -= f_tap * kc;
        f_da = f_tap / (d.kh * d.kw);
        const int r2 = f_tap - f_da * d.kh * d.kw;
        f_db = r2 / d.kw;
        f_de = r2 - f_db * d.kw;
    }
    // descriptors over the whole activation / weight tensors (igemm_nt guarantees both below 2^31 elements: 32-bit byte offsets)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(x), 0, (unsigned)min((long)d.N * d.Ti * d.Hi * d.Wi * Cin_p * 2, 0xffffffe0L), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(w), 0, (unsigned)min((long)d.w_rows * Ktot * 2, 0xffffffe0L), 0x00020000);

    auto issue = [&](int kt, int buf) {  // called with kt = 0, 1, 2, ... in order
        char* xb = smem + buf * STAGE;
        char* wb = xb + XBYTES;
        const int k = kt * IG_BK + q * 8;
        const bool kvalid = f_tap < taps && f_ci < Cin_p;
        const int tapoff = ((f_da * d.Hi + f_db) * d.Wi + f_de) * Cin_p + f_ci;
#pragma unroll
        for (int i = 0; i < XG; ++i) {
            const bool ok = kvalid && ((tapmask[i] >> (f_tap & 31)) & 1u);
            // buffer addressing: a masked piece carries an out-of-range offset and the hardware zero-fills its LDS cell
            const unsigned vo = ok ? (unsigned)(xoff[i] + tapoff) * 2u : 0xfffffff0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(xb + (wave * XG + i) * 1024), 16, (int)vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (wave + 4 * i < WGROUPS) {  // wave-uniform
                const bool ok = wrow_ok[i] && k < Ktot;
                const unsigned vo = ok ? (unsigned)(woff[i] + k) * 2u : 0xfffffff0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(wb + (wave + 4 * i) * 1024), 16, (int)vo, 0, 0, 0);
            }
        }
        f_ci += IG_BK;  // advance (tap, ci) by one slice
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
    };

    f32x16 acc[NT][MT];  // [channel tile][position tile]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment addresses: row stride 64 bytes, logical piece p of row r in slot p ^ ((r >> 2) & 3); (r >> 2) & 3 == (c >> 2) & 3 for every tile
    const int sw = (c >> 2) & 3;
    const int xrow = (wave * MT * 32 + c) * 64, wrow = c * 64;
    auto mma_slice = [&](int buf) {
        const char* xb = smem + buf * STAGE;
        const char* wb = xb + XBYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ((2 * ks + h) ^ sw) * 16;
            bf16x8 b[MT], a[NT];
#pragma unroll
            for (int j = 0; j < MT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(xb + xrow + j * 32 * 64 + slot);
#pragma unroll
            for (int i = 0; i < NT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wb + wrow + i * 32 * 64 + slot);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) mma32(acc[i][j], a[i], b[j]);
        }
    };

    float* const scb = reinterpret_cast<float*>(smem + scb_off);  // scale | bias of this block's channels, beyond tiles and epilogue image
    igemm_stage_scale_bias<BN>(scb, scale, bias, n0, d.w_rows, tid);
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);  // the other stage was last read in iteration kt - 1 (barrier below)
        mma_slice(buf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of slice kt + 1 has landed ...
        __syncthreads();                                   // ... and so has everyone's; nobody still reads stage `buf`
    }

    // ---- epilogue: scale / bias -> wave-private LDS image of 32 positions x BN channels -> residual + activation + whole-row stores ----
    const int width = min(BN, Cout_p - n0);  // channels of this block that exist (multiple of 8)
    const int cgs = width / 8;
    auto tile_rows = [&](int j, long& mbase, int& nvalid) {
        mbase = m0 + wave * MT * 32 + j * 32;
        nvalid = (int)max(0L, min((long)32, M - mbase));
    };
    // (the LDS-image epilogue is no longer compiled into these kernels: a second epilogue body behind a run-time flag cost them ~2 %,
    // profiles/README entries 92 / 94; the MFMA stems still use it)
    igemm_epilogue_direct<NT, MT>(acc, scb, res, y, n0, cgs, d, lane, tile_rows);
}


// Instance for this layer: NT channel tiles per block in the low decimal digit, MT position tiles per wave in the next; 0 = not this kernel.
int igemm_nt(const pasn_conv_desc& d, int dtype) {
    if (dtype != PASN_BF16) return 0;
    if (const char* e = tune("PASN_NO_IGEMM"))
        if (e[0] == '1') return 0;
    const bool pointwise = d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1;
    if (pointwise || d.in_swish) return 0;                 // the pointwise kernels keep those
    const int taps = d.kt * d.kh * d.kw;
    if (taps == 1) return 0;                               // strided 1x1x1 shortcuts stay on the x-tile kernel
    if (taps > 32 || d.w_kc % 16 != 0 || d.w_kc < d.Cin_p || d.Cin_p * taps < 64) return 0;
    if ((long)d.N * d.Ti * d.Hi * d.Wi * d.Cin_p >= (1L << 31) || (long)d.w_rows * taps * d.w_kc >= (1L << 31)) return 0;  // 32-bit offsets
    if (const char* e = tune_dev("PASN_IGEMM_NT")) return atoi(e);
    // cover the channels with as few, as full blocks as possible: 144 -> one block of 160, 288 -> two of 160, 576 -> four of 160 (640);
    // 64 -> 64; everything else in 128s
    const int c = d.Cout_p;
    int nt;
    if (c <= 64) nt = 2;
    else {
        const int n5 = ceil_div(c, 160), n4 = ceil_div(c, 128);
        nt = (n5 * 160 <= n4 * 128 || n5 < n4) ? 5 : 4;
    }
    // positions per block: as many as the registers allow (the work between two barriers grows with MT), as long as the grid still
    // covers the chip twice (two blocks per CU)
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int gy = ceil_div(c, nt * 32);
    int mt = 2;  // (NT = 2 with MT = 4 was measured SLOWER, 284 vs 223 us on 144 -> 64 (3,1,1): 9 DMA issues per 16 MFMAs; the LDS-DMA
                 // issue cost, ~60-180 cycles per wave-instruction, then outweighs the matrix work of the slice)
    while (mt > 1 && ceil_div(M, 128L * mt) * gy < 512) mt >>= 1;
    // stride-1 "same" (1,k,k) / (3,1,1) layers: the halo-tile kernel (igemm_halo.hip) -- hundreds digit = 1 spatial taps, 2 temporal taps
    const int mode = igemm_halo_mode(d);
    if (mode && igemm_halo_fits(d, mode, nt, mt)) return mode * 100 + mt * 10 + nt;
    return mt * 10 + nt;
}

int launch_igemm(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y,
                 const pasn_conv_desc& d, int inst, hipStream_t s) {
    if (inst >= 100) return launch_igemm_halo(x, w, scale, bias, res, y, d, inst / 100, inst % 10, (inst / 10) % 10, s);
    const int nt = inst % 10, mt = inst / 10 ? inst / 10 : 2;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const dim3 grid(ceil_div(M, 128L * mt), ceil_div(d.Cout_p, nt * 32)), block(256);
    pasn_conv_desc dk = d;
    if (const char* e = tune_dev("PASN_IGEMM_RR"))
        if (e[0] == '1') dk.w_frag = 7;  // A/B switch: round-robin tile placement
#define PASN_IG(NT_, MT_)                                                                                                     \
    if (nt == NT_ && mt == MT_) {                                                                                             \
        const size_t tiles = (size_t)2 * (128 * MT_ * 64 + NT_ * 32 * 64), image = (size_t)4 * 32 * (NT_ * 32 + 8) * 2;       \
        const size_t scb_off = tiles > image ? tiles : image, lds = scb_off + (size_t)NT_ * 32 * 8;                           \
        if (lds > 64 * 1024) PASN_MAX_LDS(96 * 1024, igemm_glds_kernel<NT_, MT_>);                                            \
        hipLaunchKernelGGL((igemm_glds_kernel<NT_, MT_>), grid, block, lds, s, (const __bf16*)x, (const __bf16*)w, scale, bias, \
                           (const __bf16*)res, (__bf16*)y, dk, (int)scb_off);                        \
        return check_launch("igemm_glds_kernel");                                                                             \
    }
    PASN_IG(2, 4)
    PASN_IG(2, 2)
    PASN_IG(2, 1)
    PASN_IG(4, 2)
    PASN_IG(4, 1)
    PASN_IG(5, 2)
    PASN_IG(5, 1)
#undef PASN_IG
    set_error("launch_igemm: no such instance");
    return PASN_ERR_ARG;
}

}  // namespace pasn
