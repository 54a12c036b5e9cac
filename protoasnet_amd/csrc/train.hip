// train.hip -- the elementwise / reduction half of the training path: batch-statistics norm, activation forward and
// backward, squeeze-excite backward, strided scatter.  All HBM-bound passes over channels-last rows [N][S][Cp];
// a thread owns 8 channels of a row, a block's threads tile (row lane, channel group) so that a wave reads whole rows
// back to back, and every reduction is two-stage with a fixed summation order (bitwise reproducible gradients).
//
// Reference semantics: torch.nn.BatchNorm{2,3}d in train mode (biased batch variance for normalisation, unbiased for
// the running estimate, momentum 0.1) as instantiated at resnet_features.py:140,180 and by every trunk the reference
// trains (agents call model.train(): Video_XProtoNet_e2e.py:118); autograd's derivative of the same expressions.
#include "common.h"

namespace pasn {

constexpr int PASN_MAX_GROUPS = 4;  // statistics groups of one pass (bn_finalize_kernel)

struct RowGeom {
    int CG, LPR, RL, chunks, rows_per_chunk;
};

// 256 threads = RL row lanes x LPR lanes per row; lane cg of a row owns channels 8 cg .. 8 cg + 7 (lanes cg >= CG and the threads beyond
// RL * LPR idle).  LPR = the next power of two >= CG.  PASN_TRAIN_ROWS_CONTIG=1: LPR = CG -- a block's live threads read RL whole rows as
// ONE contiguous span and 240-255 of 256 threads are live at the X3D widths instead of 75-88 %; measured +-0 on the step (the passes are
// not short of requests in flight), and it re-partitions every sum, so the default keeps the summation order of rounds 1-3.
static RowGeom row_geom(int N, int S, int Cp) {
    RowGeom g;
    g.CG = Cp / 8;
    g.LPR = 1;
    while (g.LPR < g.CG) g.LPR <<= 1;
    if (tune_is("PASN_TRAIN_ROWS_CONTIG", '1')) g.LPR = g.CG;
    g.RL = 256 / g.LPR;
    int blocks = 1024;
    if (const char* e = tune("PASN_TRAIN_BLOCKS")) blocks = std::max(256, atoi(e));
    const int want = std::max(1, blocks / std::max(1, N));
    const int maxc = std::max(1, S / (g.RL * 16));  // small tensors: fewer partials, the finalize pass is pure latency
    g.chunks = std::min(want, maxc);
    g.rows_per_chunk = ceil_div(S, g.chunks);
    g.chunks = ceil_div(S, g.rows_per_chunk);
    return g;
}

// Rows in flight per thread: every pass below requests ROWS_U rows of each operand, then computes, then stores.  (A plain loop, even
// unrolled, waited for each row's loads before the next row's went out wherever the pass writes in place or branches on the activation:
// two or three 16-byte loads in flight per thread, 2.8-4 TB/s.)
constexpr int ROWS_U = 4;
template <typename T>
struct Raw8 {
    uint4 r[(8 * sizeof(T)) / 16];
};
template <typename T>
__device__ __forceinline__ Raw8<T> load_raw8(const T* p) {
    Raw8<T> q;
#pragma unroll
    for (int i = 0; i < (int)((8 * sizeof(T)) / 16); ++i) q.r[i] = reinterpret_cast<const uint4*>(p)[i];
    return q;
}

// Rows through a buffer descriptor that ends where the block's chunk ends (num_records = r1 rows): a row past the chunk reads as zeros and its
// store is dropped by the hardware -- no clamped row index, no 64-bit address per row, no select per element to keep such a row out of the sums
// (these passes are bound by vector issue).  Offsets are 32-bit bytes inside ONE clip (the launchers check S * Cp * sizeof(T) < 2^31).
typedef unsigned rb_u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const T* clip, int rows, int Cp) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(clip), 0, (unsigned)rows * (unsigned)Cp * (unsigned)sizeof(T), 0x00020000);
}
template <typename T>
__device__ __forceinline__ Raw8<T> load_raw8(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    Raw8<T> q;
#pragma unroll
    for (int i = 0; i < (int)((8 * sizeof(T)) / 16); ++i) {
        const rb_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16 * i), 0, 0);
        q.r[i] = uint4{v[0], v[1], v[2], v[3]};
    }
    return q;
}
__device__ __forceinline__ void store8(__amdgpu_buffer_rsrc_t rs, unsigned off, const float (&v)[8], const __bf16*) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rb_u32x4, a), rs, (int)off, 0, 0);
}
__device__ __forceinline__ void store8(__amdgpu_buffer_rsrc_t rs, unsigned off, const float (&v)[8], const float*) {
    __builtin_amdgcn_raw_buffer_store_b128(rb_u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rs, (int)off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(rb_u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, rs, (int)off + 16, 0, 0);
}

// Sum the per-thread 8-channel accumulators `acc[W][8]` over the row lanes of the block (fixed order) and store them
// at out[w * Cp + cg*8 + j].
template <int W>
__device__ __forceinline__ void block_reduce_rows(float (&acc)[W][8], float* red, float* out, int Cp, int CG, int LPR) {
    const int tid = threadIdx.x, RL = 256 / LPR;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[w][j];
        __syncthreads();
        for (int t = tid; t < LPR * 8; t += 256) {
            const int cg = t >> 3, j = t & 7;
            if (cg < CG) {
                float s = 0.0f;
                for (int rl = 0; rl < RL; ++rl) s += red[(rl * LPR + cg) * 8 + j];
                out[(size_t)w * Cp + cg * 8 + j] = s;
            }
        }
    }
}

// ---- batch statistics ---------------------------------------------------------------------------------------------
// ws[n][chunk][2][Cp]: sum(y - k), sum((y - k)^2) with the shift k[c] = y[0][0][c] (keeps the variance free of
// cancellation when |mean| >> std).
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const T* __restrict__ y, float* __restrict__ ws, int S, int Cp,
                                                               int CG, int LPR, int rows_per_chunk, int chunks) {
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, ch = blockIdx.x;
    const int cg = threadIdx.x % LPR, rl = threadIdx.x / LPR, RL = 256 / LPR;
    float acc[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.0f;
    if (cg < CG && rl < RL) {
        float k[8];
        load8(y + cg * 8, k);
        const int r0 = ch * rows_per_chunk, r1 = min(S, r0 + rows_per_chunk);
        const T* base = y + (size_t)n * S * Cp + cg * 8;
        for (int r = r0 + rl; r < r1; r += ROWS_U * RL) {
            Raw8<T> yv[ROWS_U];
#pragma unroll
            for (int u = 0; u < ROWS_U; ++u) yv[u] = load_raw8(base + (size_t)min(r + u * RL, r1 - 1) * Cp);
#pragma unroll
            for (int u = 0; u < ROWS_U; ++u) {
                float v[8];
                raw_to_f8<T>(yv[u].r, v);
                const bool ok = r + u * RL < r1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dlt = ok ? v[j] - k[j] : 0.0f;
                    acc[0][j] += dlt;
                    acc[1][j] = fmaf(dlt, dlt, acc[1][j]);
                }
            }
        }
    }
    block_reduce_rows<2>(acc, red, ws + ((size_t)n * chunks + ch) * 2 * Cp, Cp, CG, LPR);
}

// One block per 16 channels.  Totals: each of the 16 "parts" sums every 16th partial (independent loads, one LDS combine in a
// fixed order).  Per-clip pool (squeeze-excite only): a part owns whole clips, no cross-thread combine at all.
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ ws, const T* __restrict__ y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* rmean, float* rvar,  // (no restrict: `shift` may alias rmean)
                                                          float momentum, float eps, float* __restrict__ stat, float* __restrict__ pool_u,
                                                          int N, int S, int C, int Cp, int chunks, const float* shift, int gdiv) {
    // GROUPS (round 4): clips n / gdiv form one statistics group -- the two trunk passes of the reference's loss recipe (the clips and their
    // warped copies) run as ONE 2N-clip pass, each half with its own batch statistics: stat[g][4][Cp], the running estimates updated group by
    // group in order, exactly as two N-clip passes leave them.  gdiv = N: one group, the layout of rounds 1-3.
    __shared__ float red[2][16][16];
    __shared__ float scsh[2][PASN_MAX_GROUPS][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool live = c < Cp;
    const int G = N / gdiv;
    // the shift the partials were taken with: y[0][0][c] (bn_stats_partial_kernel), or `shift` (the stencils' fused statistics: the running
    // mean as it stood BEFORE this update -- read here ahead of the barrier, written below behind it), or 0
    float k = !live ? 0.0f : y ? (float)y[c] : (shift && c < C) ? shift[c] : 0.0f;  // `shift` holds C floats (padded channels: 0)
    asm volatile("" : "+v"(k));  // the value is taken HERE: `shift` may be the running mean this kernel updates behind the barrier
    for (int g = 0; g < G; ++g) {
        float a1 = 0.0f, a2 = 0.0f;
        if (live) {
            const int lo = g * gdiv * chunks, hi = lo + gdiv * chunks;
#pragma unroll 8
            for (int i = lo + part; i < hi; i += 16) {
                const float* p = ws + (size_t)i * 2 * Cp + c;
                a1 += p[0];
                a2 += p[Cp];
            }
        }
        if (g) __syncthreads();  // the previous group's totals have been read
        red[0][part][cl] = a1;
        red[1][part][cl] = a2;
        __syncthreads();
        if (part == 0) {
            float t1 = 0.0f, t2 = 0.0f;
            for (int q = 0; q < 16; ++q) {
                t1 += red[0][q][cl];
                t2 += red[1][q][cl];
            }
            float mean = 0.0f, invstd = 0.0f, sc = 0.0f, sh = 0.0f;
            if (live && c < C) {
                const float R = (float)gdiv * (float)S;
                const float m = t1 / R;
                const float var = fmaxf(t2 / R - m * m, 0.0f);
                mean = k + m;
                invstd = 1.0f / sqrtf(var + eps);
                const float gm = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
                sc = gm * invstd;
                sh = bt - mean * sc;
                if (rmean) {
                    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
                    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var * (R / fmaxf(R - 1.0f, 1.0f));
                }
            }
            if (live) {
                float* st = stat + (size_t)g * 4 * Cp;
                st[c] = mean;
                st[Cp + c] = invstd;
                st[2 * Cp + c] = sc;
                st[3 * Cp + c] = sh;
            }
            scsh[0][g][cl] = sc;
            scsh[1][g][cl] = sh;
        }
    }
    if (pool_u == nullptr) return;
    __syncthreads();
    if (!live) return;
    for (int n = part; n < N; n += 16) {
        const float sc = scsh[0][n / gdiv][cl], sh = scsh[1][n / gdiv][cl];
        float b1 = 0.0f;
        const float* p = ws + (size_t)n * chunks * 2 * Cp + c;
#pragma unroll 4
        for (int ch = 0; ch < chunks; ++ch) b1 += p[(size_t)ch * 2 * Cp];
        pool_u[(size_t)n * Cp + c] = sc * (k + b1 / (float)S) + sh;
    }
}

// ---- a = act((y*sc + sh + residual) * gate) ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void affine_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ stat, const T* __restrict__ res,
                                                             const float* __restrict__ gate, T* __restrict__ a, int S, int Cp, int CG,
                                                             int LPR, int rows_per_chunk, int act, int gdiv) {
    const int n = blockIdx.y, ch = blockIdx.x;
    const int cg = threadIdx.x % LPR, rl = threadIdx.x / LPR, RL = 256 / LPR;
    if (cg >= CG || rl >= RL) return;
    stat += (size_t)(n / gdiv) * 4 * Cp;  // this clip's statistics group
    float sc[8], sh[8], g[8];
    load8(stat + 2 * Cp + cg * 8, sc);
    load8(stat + 3 * Cp + cg * 8, sh);
    if (gate) load8(gate + (size_t)n * Cp + cg * 8, g);
    const int r0 = ch * rows_per_chunk, r1 = min(S, r0 + rows_per_chunk);
    const size_t clip = (size_t)n * S * Cp;
    const __amdgpu_buffer_rsrc_t yrs = row_rsrc(y + clip, r1, Cp), qrs = row_rsrc(res ? res + clip : y + clip, r1, Cp), ars = row_rsrc(a + clip, r1, Cp);
    const unsigned rowb = (unsigned)Cp * (unsigned)sizeof(T), cgo = (unsigned)cg * 8u * (unsigned)sizeof(T);
    for (int r = r0 + rl; r < r1; r += ROWS_U * RL) {
        unsigned o[ROWS_U];
        Raw8<T> yv[ROWS_U], qv[ROWS_U];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) {
            o[u] = (unsigned)(r + u * RL) * rowb + cgo;  // past the chunk: out of range -- zeros in, nothing out
            yv[u] = load_raw8<T>(yrs, o[u]);
        }
        if (res) {
#pragma unroll
            for (int u = 0; u < ROWS_U; ++u) qv[u] = load_raw8<T>(qrs, o[u]);
        }
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) {
            float v[8];
            raw_to_f8<T>(yv[u].r, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
            if (res) {
                float q[8];
                raw_to_f8<T>(qv[u].r, q);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += q[j];
            }
            if (gate) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= g[j];
            }
            act_vec(v, act);
            store8(ars, o[u], v, (const T*)nullptr);
        }
    }
}

// ---- backward pass over one unit (in place on d) --------------------------------------------------------------------
// mode 0:  d' = d * act'(y*sc + sh + residual)                 partials: sum d', sum d' * yhat      (yhat = (y-mean)*invstd)
// mode 1:  d' = d * act'((y*sc + sh) * gate)                   partials: sum d' * (y*sc + sh)       (per clip: grad of the gate)
// mode 2:  d' = d * gate + add[n][c]                           partials: sum d', sum d' * yhat
template <typename T, int MODE>
__global__ __launch_bounds__(256) void grad_pass_kernel(T* __restrict__ d, const T* __restrict__ y, const float* __restrict__ stat,
                                                        const T* __restrict__ res, const float* __restrict__ gate, const float* __restrict__ add,
                                                        float* __restrict__ ws, int S, int Cp, int CG, int LPR, int rows_per_chunk,
                                                        int chunks, int act, int write_back, int gdiv) {
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, ch = blockIdx.x;
    stat += (size_t)(n / gdiv) * 4 * Cp;  // this clip's statistics group
    const int cg = threadIdx.x % LPR, rl = threadIdx.x / LPR, RL = 256 / LPR;
    constexpr int W = MODE == 4 ? 3 : 2;
    constexpr int U = ROWS_U;
    float acc[W][8];
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[w][j] = 0.0f;
    if (cg < CG && rl < RL) {
        float mean[8], invstd[8], sc[8], sh[8], g[8], ad[8];
        load8(stat + cg * 8, mean);
        load8(stat + Cp + cg * 8, invstd);
        load8(stat + 2 * Cp + cg * 8, sc);
        load8(stat + 3 * Cp + cg * 8, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            g[j] = 1.0f;
            ad[j] = 0.0f;
        }
        if (gate) load8(gate + (size_t)n * Cp + cg * 8, g);
        if (add) load8(add + (size_t)n * Cp + cg * 8, ad);
        const int r0 = ch * rows_per_chunk, r1 = min(S, r0 + rows_per_chunk);
        const size_t clip = (size_t)n * S * Cp;
        const __amdgpu_buffer_rsrc_t yrs = row_rsrc(y + clip, r1, Cp), drs = row_rsrc(d + clip, r1, Cp),
                                     qrs = row_rsrc((MODE == 0 && res) ? res + clip : y + clip, r1, Cp);
        const unsigned rowb = (unsigned)Cp * (unsigned)sizeof(T), cgo = (unsigned)cg * 8u * (unsigned)sizeof(T);
        for (int r = r0 + rl; r < r1; r += U * RL) {
            unsigned o[U];
            Raw8<T> yv[U], dr[U], qv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                o[u] = (unsigned)(r + u * RL) * rowb + cgo;  // past the chunk: out of range -- zeros in, nothing out
                yv[u] = load_raw8<T>(yrs, o[u]);
                dr[u] = load_raw8<T>(drs, o[u]);
            }
            if (MODE == 0 && res) {
#pragma unroll
                for (int u = 0; u < U; ++u) qv[u] = load_raw8<T>(qrs, o[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = r + u * RL < r1;
                float v[8], dv[8];
                raw_to_f8<T>(yv[u].r, v);
                raw_to_f8<T>(dr[u].r, dv);
                if (MODE == 0) {
                    float uu[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) uu[j] = fmaf(v[j], sc[j], sh[j]);
                    if (res) {
                        float q[8];
                        raw_to_f8<T>(qv[u].r, q);
#pragma unroll
                        for (int j = 0; j < 8; ++j) uu[j] += q[j];
                    }
                    act_grad_mul(dv, uu, act);
                } else if (MODE == 1) {
                    float uu[8], ug[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        uu[j] = fmaf(v[j], sc[j], sh[j]);
                        ug[j] = uu[j] * g[j];
                    }
                    act_grad_mul(dv, ug, act);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[0][j] = fmaf(dv[j], uu[j], acc[0][j]);  // (a row past the chunk: d = 0)
                } else if (MODE == 4) {
                    // as mode 1, but the sums the squeeze-excite unit's norm needs are taken here, per clip: (sum d', sum d' yhat, sum yhat).
                    // d'' = d' gate + add[n] is affine in d' per clip, so sum d'' and sum d'' yhat follow without a second pass over (d, y)
                    // (pasn_se_gate_bwd_stat), and sum d' u = gamma sum d' yhat + beta sum d' (the gate's gradient) as well
                    float ug[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) ug[j] = fmaf(v[j], sc[j], sh[j]) * g[j];
                    act_grad_mul(dv, ug, act);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float yc = v[j] - mean[j];  // (x invstd once, behind the loop)
                        acc[0][j] += dv[j];               // (a row past the chunk: d = 0, so d' = 0)
                        acc[1][j] = fmaf(dv[j], yc, acc[1][j]);
                        acc[W - 1][j] += ok ? yc : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) dv[j] = ok ? fmaf(dv[j], g[j], ad[j]) : 0.0f;  // (add[n][c] != 0: a row past the chunk must stay out of the sums)
                }
                if (MODE != 1 && MODE != 4) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        acc[0][j] += dv[j];  // (a row past the chunk: d' = 0)
                        acc[1][j] = fmaf(dv[j], v[j] - mean[j], acc[1][j]);  // sum d' (y - mean); x invstd once, behind the loop
                    }
                }
                if (write_back) store8(drs, o[u], dv, (const T*)nullptr);
            }
        }
        // yhat = (y - mean) invstd: the factor invstd[c] is common to a thread's whole sum (the kernels are bound by vector issue, not bytes: 70-78 %
        // of the SIMD cycles carry a vector instruction; one multiply per element and sum less)
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[1][j] *= invstd[j];
                if (MODE == 4) acc[W - 1][j] *= invstd[j];
            }
        }
    }
    block_reduce_rows<W>(acc, red, ws + ((size_t)n * chunks + ch) * W * Cp, Cp, CG, LPR);
}

// totals of the mode 0 / 2 partials -> coef[g][2][Cp] = (sum d'/R, sum d' yhat / R) per statistics group, dgamma, dbeta (either may be NULL)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ ws, float* __restrict__ coef, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int N, int S, int C, int Cp, int chunks, int gdiv) {
    // per statistics group g (clips n / gdiv): coef[g][2][Cp]; dgamma / dbeta are the parameter's: summed over the groups in order
    __shared__ float red[2][16][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool live = c < Cp;
    const int G = N / gdiv;
    float dg = 0.0f, db = 0.0f;
    for (int g = 0; g < G; ++g) {
        float a1 = 0.0f, a2 = 0.0f;
        if (live) {
            const int lo = g * gdiv * chunks, hi = lo + gdiv * chunks;
#pragma unroll 8
            for (int i = lo + part; i < hi; i += 16) {
                const float* p = ws + (size_t)i * 2 * Cp + c;
                a1 += p[0];
                a2 += p[Cp];
            }
        }
        if (g) __syncthreads();
        red[0][part][cl] = a1;
        red[1][part][cl] = a2;
        __syncthreads();
        if (part == 0 && live) {
            float b1 = 0.0f, b2 = 0.0f;
            for (int q = 0; q < 16; ++q) {
                b1 += red[0][q][cl];
                b2 += red[1][q][cl];
            }
            const float R = (float)gdiv * (float)S;
            coef[(size_t)g * 2 * Cp + c] = b1 / R;
            coef[(size_t)g * 2 * Cp + Cp + c] = b2 / R;
            dg += b2;
            db += b1;
        }
    }
    if (part == 0 && live && c < C) {
        if (dgamma) dgamma[c] = dg;
        if (dbeta) dbeta[c] = db;
    }
}

// dy = sc * (d' - m1 - yhat * m2),  d' = d (act == NONE: already differentiated by the reduce pass) or d * act'(y*sc + sh)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ d, const T* __restrict__ y, const float* __restrict__ stat,
                                                           const float* __restrict__ coef, T* __restrict__ dy, int S, int Cp, int CG, int LPR,
                                                           int rows_per_chunk, int act, int gdiv) {
    const int n = blockIdx.y, ch = blockIdx.x;
    const int cg = threadIdx.x % LPR, rl = threadIdx.x / LPR, RL = 256 / LPR;
    if (cg >= CG || rl >= RL) return;
    stat += (size_t)(n / gdiv) * 4 * Cp;  // this clip's statistics group
    coef += (size_t)(n / gdiv) * 2 * Cp;
    constexpr int U = ROWS_U;
    float mean[8], invstd[8], sc[8], sh[8], m1[8], m2[8];
    load8(stat + cg * 8, mean);
    load8(stat + Cp + cg * 8, invstd);
    load8(stat + 2 * Cp + cg * 8, sc);
    load8(stat + 3 * Cp + cg * 8, sh);
    load8(coef + cg * 8, m1);
    load8(coef + Cp + cg * 8, m2);
    // dy = sc (d' - m1 - (y - mean) invstd m2) = sc d' + bq (y - mean) + cq: one subtraction and two FMAs per element
    float bq[8], cq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bq[j] = -sc[j] * invstd[j] * m2[j];
        cq[j] = -sc[j] * m1[j];
    }
    const int r0 = ch * rows_per_chunk, r1 = min(S, r0 + rows_per_chunk);
    const size_t clip = (size_t)n * S * Cp;
    const __amdgpu_buffer_rsrc_t yrs = row_rsrc(y + clip, r1, Cp), drs = row_rsrc(d + clip, r1, Cp), ors = row_rsrc(dy + clip, r1, Cp);
    const unsigned rowb = (unsigned)Cp * (unsigned)sizeof(T), cgo = (unsigned)cg * 8u * (unsigned)sizeof(T);
    for (int r = r0 + rl; r < r1; r += U * RL) {
        unsigned o[U];
        Raw8<T> yv[U], dr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o[u] = (unsigned)(r + u * RL) * rowb + cgo;  // past the chunk: out of range -- zeros in, nothing out
            yv[u] = load_raw8<T>(yrs, o[u]);
            dr[u] = load_raw8<T>(drs, o[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float v[8], dv[8];
            raw_to_f8<T>(yv[u].r, v);
            raw_to_f8<T>(dr[u].r, dv);
            if (act != PASN_ACT_NONE) {
                float uu[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) uu[j] = fmaf(v[j], sc[j], sh[j]);
                act_grad_mul(dv, uu, act);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) dv[j] = fmaf(sc[j], dv[j], fmaf(bq[j], v[j] - mean[j], cq[j]));
            store8(ors, o[u], dv, (const T*)nullptr);
        }
    }
}

// ---- squeeze-excite backward (one block per clip) ---------------------------------------------------------------------
// gate = sigmoid(w2 . relu(w1 . pool + b1) + b2).  In: dgate partials (mode 1), pool_u.  Out: add[n][c] = dpool[c] / S and
// this clip's parameter-gradient contributions pn[n][...] = (dw1 [Cse][C], db1 [Cse], dw2 [C][Cse], db2 [C]).
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ ws, int chunks, const float* __restrict__ pool_u,
                                                         const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ add, float* __restrict__ pn, int S, int C,
                                                         int Cp, int Cse, const float* __restrict__ stat, float* __restrict__ ws3, int gdiv) {
    extern __shared__ float sm[];
    if (stat) stat += (size_t)(blockIdx.x / gdiv) * 4 * Cp;  // this clip's statistics group
    float* pool = sm;             // [C]
    float* ds = pool + C;         // [C]
    float* h = ds + C;            // [Cse]
    float* dh = h + Cse;          // [Cse]
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) pool[c] = pool_u[(size_t)n * Cp + c];
    __syncthreads();
    // hidden units: 8 lanes share one dot product over the channels (shuffle combine in a fixed order)
    for (int j0 = 0; j0 < Cse; j0 += 32) {
        const int j = j0 + (tid >> 3), sub = tid & 7;
        float a = 0.0f;
        if (j < Cse)
#pragma unroll 8
            for (int c = sub; c < C; c += 8) a = fmaf(w1[(size_t)j * C + c], pool[c], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        if (j < Cse && sub == 0) h[j] = fmaxf(a + b1[j], 0.0f);
    }
    __syncthreads();
    float* p = pn + (size_t)n * (2 * (size_t)C * Cse + Cse + C);
    float* p_w1 = p;
    float* p_b1 = p_w1 + (size_t)Cse * C;
    float* p_w2 = p_b1 + Cse;
    float* p_b2 = p_w2 + (size_t)C * Cse;
    for (int c = tid; c < C; c += 256) {
        float dg = 0.0f;
        if (stat) {
            // mode-4 partials [n][chunk][3][Cp]: clip totals (fixed order), left in chunk 0 for se_bn_coef_kernel; the gate's gradient
            // sum d' u with u = gamma yhat + beta
            float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            // batches of 8 chunks: 24 independent loads in flight, then the adds in chunk order (a rolled load -> add loop is one L2 round
            // trip per chunk: ~1 us x 32 chunks per launch, 15 launches per step)
            for (int ch0 = 0; ch0 < chunks; ch0 += 8) {
                float t1[8], t2[8], t3[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* q = ws3 + ((size_t)n * chunks + min(ch0 + u, chunks - 1)) * 3 * Cp + c;
                    t1[u] = q[0];
                    t2[u] = q[Cp];
                    t3[u] = q[2 * Cp];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (ch0 + u < chunks) {
                        a1 += t1[u];
                        a2 += t2[u];
                        a3 += t3[u];
                    }
            }
            float* q0 = ws3 + (size_t)n * chunks * 3 * Cp + c;
            q0[0] = a1;
            q0[Cp] = a2;
            q0[2 * Cp] = a3;
            const float scv = stat[2 * Cp + c], gamma = scv / stat[Cp + c], beta = fmaf(stat[c], scv, stat[3 * Cp + c]);
            dg = fmaf(gamma, a2, beta * a1);
        } else {
            for (int ch0 = 0; ch0 < chunks; ch0 += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ws[((size_t)n * chunks + min(ch0 + u, chunks - 1)) * 2 * Cp + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) dg += ch0 + u < chunks ? t[u] : 0.0f;
            }
        }
        float a = b2[c];
#pragma unroll 8
        for (int j = 0; j < Cse; ++j) a = fmaf(w2[(size_t)c * Cse + j], h[j], a);
        const float g = 1.0f / (1.0f + expf(-a));
        const float d = dg * g * (1.0f - g);
        ds[c] = d;
        p_b2[c] = d;
#pragma unroll 8
        for (int j = 0; j < Cse; ++j) p_w2[(size_t)c * Cse + j] = d * h[j];
    }
    __syncthreads();
    for (int j0 = 0; j0 < Cse; j0 += 32) {
        const int j = j0 + (tid >> 3), sub = tid & 7;
        float a = 0.0f;
        if (j < Cse)
#pragma unroll 8
            for (int c = sub; c < C; c += 8) a = fmaf(w2[(size_t)c * Cse + j], ds[c], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        if (j < Cse && sub == 0) {
            a = h[j] > 0.0f ? a : 0.0f;
            dh[j] = a;
            p_b1[j] = a;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float a = 0.0f;
#pragma unroll 8
        for (int j = 0; j < Cse; ++j) {
            a = fmaf(w1[(size_t)j * C + c], dh[j], a);
            p_w1[(size_t)j * C + c] = dh[j] * pool[c];
        }
        add[(size_t)n * Cp + c] = a / (float)S;
    }
    for (int c = C + tid; c < Cp; c += 256) add[(size_t)n * Cp + c] = 0.0f;
}

// The per-clip parameter-gradient contributions pn[n][dw1 | db1 | dw2 | db2] summed over the clips (fixed order), each
// segment into its own gradient tensor -- one launch for the four.
__global__ __launch_bounds__(256) void se_sum_over_clips_kernel(const float* __restrict__ pn, float* __restrict__ dw1, float* __restrict__ db1,
                                                                float* __restrict__ dw2, float* __restrict__ db2, int N, int C, int Cse) {
    const size_t o_b1 = (size_t)Cse * C, o_w2 = o_b1 + Cse, o_b2 = o_w2 + (size_t)C * Cse, len = o_b2 + C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    float s = 0.0f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) s += pn[(size_t)n * len + i];
    if (i < o_b1) dw1[i] = s;
    else if (i < o_w2) db1[i - o_b1] = s;
    else if (i < o_b2) dw2[i - o_w2] = s;
    else db2[i - o_b2] = s;
}

// ---- strided scatter: dst[n][t*st][h*sh][w*sw][:] (+)= src[n][t][h][w][:], every other element 0 (or kept) -------------
template <typename T>
__global__ __launch_bounds__(256) void scatter_strided_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int To, int Ho, int Wo,
                                                              int Ti, int Hi, int Wi, int st, int sh, int sw, int Cp, int accumulate) {
    const int CG = Cp / 8;
    const size_t total = (size_t)N * Ti * Hi * Wi * CG;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        size_t row = i / CG;
        const int w = (int)(row % Wi);
        size_t q = row / Wi;
        const int h = (int)(q % Hi);
        q /= Hi;
        const int t = (int)(q % Ti), n = (int)(q / Ti);
        const bool on = (t % st == 0) && (h % sh == 0) && (w % sw == 0) && (t / st < To) && (h / sh < Ho) && (w / sw < Wo);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
        T* o = dst + row * Cp + cg * 8;
        if (on) {
            load8(src + ((((size_t)n * To + t / st) * Ho + h / sh) * Wo + w / sw) * Cp + cg * 8, v);
            if (accumulate) {
                float e[8];
                load8(o, e);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += e[j];
            }
            store8(o, v);
        } else if (!accumulate) {
            store8(o, v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(T* __restrict__ a, const T* __restrict__ b, size_t groups) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < groups; i += (size_t)gridDim.x * 256) {
        float x[8], y[8];
        load8(a + i * 8, x);
        load8(b + i * 8, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        store8(a + i * 8, x);
    }
}

// ---- max-pool backward (gather form, no atomics): an input element receives dy of every window whose FIRST maximum (scan
// order t, h, w -- the index torch's forward records) it is ------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, pasn_conv_desc d) {
    const int CG = d.Cin_p / 8;
    const size_t total = (size_t)d.N * d.Ti * d.Hi * d.Wi * CG;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        const size_t row = i / CG;
        const int wi = (int)(row % d.Wi);
        size_t q = row / d.Wi;
        const int hi = (int)(q % d.Hi);
        q /= d.Hi;
        const int ti = (int)(q % d.Ti), n = (int)(q / d.Ti);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        const T* xn = x + (size_t)n * d.Ti * d.Hi * d.Wi * d.Cin_p + cg * 8;
        for (int to = max(0, (ti + d.pt - d.kt + d.st) / d.st); to <= min(d.To - 1, (ti + d.pt) / d.st); ++to)
            for (int ho = max(0, (hi + d.ph - d.kh + d.sh) / d.sh); ho <= min(d.Ho - 1, (hi + d.ph) / d.sh); ++ho)
                for (int wo = max(0, (wi + d.pw - d.kw + d.sw) / d.sw); wo <= min(d.Wo - 1, (wi + d.pw) / d.sw); ++wo) {
                    float best[8];
                    int bidx[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        best[j] = -INFINITY;
                        bidx[j] = -1;
                    }
                    for (int a = 0; a < d.kt; ++a)
                        for (int b = 0; b < d.kh; ++b)
                            for (int c = 0; c < d.kw; ++c) {
                                const int t2 = to * d.st - d.pt + a, h2 = ho * d.sh - d.ph + b, w2 = wo * d.sw - d.pw + c;
                                if (t2 < 0 || t2 >= d.Ti || h2 < 0 || h2 >= d.Hi || w2 < 0 || w2 >= d.Wi) continue;
                                const int lin = (t2 * d.Hi + h2) * d.Wi + w2;
                                float v[8];
                                load8(xn + (size_t)lin * d.Cin_p, v);
#pragma unroll
                                for (int j = 0; j < 8; ++j)
                                    if (v[j] > best[j] || (v[j] != v[j] && bidx[j] < 0)) {
                                        best[j] = v[j];
                                        bidx[j] = lin;
                                    }
                            }
                    const int me = (ti * d.Hi + hi) * d.Wi + wi;
                    float g[8];
                    load8(dy + ((((size_t)n * d.To + to) * d.Ho + ho) * d.Wo + wo) * d.Cout_p + cg * 8, g);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (bidx[j] == me) acc[j] += g[j];
                }
        store8(dx + row * d.Cin_p + cg * 8, acc);
    }
}

}  // namespace pasn

using namespace pasn;

#define ROWS_ARGS_OK(N, S, C, Cp)                                                                         \
    PASN_REQUIRE((N) > 0 && (S) > 0 && (C) > 0 && (Cp) >= (C) && (Cp) % 8 == 0, "bad tensor extents");   \
    PASN_REQUIRE((Cp) <= 2048, "channel stride above 2048 is not covered");                                     \
    PASN_REQUIRE((long)(S) * (Cp) * 4 < 2147483648L, "one clip's rows must stay below 2 GiB (32-bit offsets inside a clip)")

#define GROUPS_OK(N, groups) PASN_REQUIRE((groups) >= 1 && (groups) <= PASN_MAX_GROUPS && (N) % (groups) == 0, "statistics groups must divide the batch (1 .. 4)")

extern "C" int pasn_train_chunks(int N, int S, int Cp) {
    if (N <= 0 || S <= 0 || Cp <= 0 || Cp % 8 || Cp > 2048) return 0;
    return row_geom(N, S, Cp).chunks;
}

extern "C" int pasn_bn_stats_fwd(const void* y, float* ws, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float momentum, float eps, float* stat, float* pool_u, int N, int S, int C,
                                 int Cp, int dtype, void* stream) {
    return pasn_bn_stats_fwd_g(y, ws, gamma, beta, running_mean, running_var, momentum, eps, stat, pool_u, N, S, C, Cp, dtype, 1, stream);
}

// ... with `groups` statistics groups of N / groups consecutive clips each: stat[groups][4][Cp]; running estimates updated group by group
extern "C" int pasn_bn_stats_fwd_g(const void* y, float* ws, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float* stat, float* pool_u, int N, int S, int C,
                                   int Cp, int dtype, int groups, void* stream) {
    ROWS_ARGS_OK(N, S, C, Cp);
    GROUPS_OK(N, groups);
    PASN_REQUIRE(y && ws && stat, "null pointer");
    PASN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean / running_var go together");
    const RowGeom g = row_geom(N, S, Cp);
    const dim3 grid(g.chunks, N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16) {
        hipLaunchKernelGGL(bn_stats_partial_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)y, ws, S, Cp, g.CG, g.LPR, g.rows_per_chunk, g.chunks);
        hipLaunchKernelGGL(bn_finalize_kernel<__bf16>, dim3(ceil_div(Cp, 16)), dim3(256), 0, s, ws, (const __bf16*)y, gamma, beta, running_mean,
                           running_var, momentum, eps, stat, pool_u, N, S, C, Cp, g.chunks, (const float*)nullptr, N / groups);
    } else {
        hipLaunchKernelGGL(bn_stats_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)y, ws, S, Cp, g.CG, g.LPR, g.rows_per_chunk, g.chunks);
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(ceil_div(Cp, 16)), dim3(256), 0, s, ws, (const float*)y, gamma, beta, running_mean,
                           running_var, momentum, eps, stat, pool_u, N, S, C, Cp, g.chunks, (const float*)nullptr, N / groups);
    }
    return check_launch("bn_stats_fwd");
}

// Depthwise stencil + batch statistics in ONE pass over y (the X3D conv_b units): the T-marching stencil accumulates sum / sum of squares
// of its fp32 outputs per block, bn_finalize_kernel turns the [N][rows][2][Cp] partials into the unit's statistics table.  Saves the
// read of y by pasn_bn_stats_fwd (27 launches, ~1 ms of the X3D-S step).  Unshifted second moments: var = E[y^2] - mean^2 in fp32 over
// per-block fp32 partials -- a conv output without bias has |mean| ~ std.
extern "C" int pasn_dwconv3d_stats_rows(const pasn_conv_desc* d, int dtype) {
    if (!d || d->Cout_p <= 0 || d->Cout_p % 8 != 0 || d->Cout_p > 2048) return 0;
    if (const char* e = tune("PASN_NO_DW_STATS"))
        if (e[0] == '1') return 0;
    // stride-1 layers: the matrix-core stencil with the statistics epilogue (one partial row pair per (clip, chunk)); PASN_DW_STATS_MFMA=0: the
    // VALU stencil everywhere, as before round 3
    if (!(tune("PASN_DW_STATS_MFMA") && tune("PASN_DW_STATS_MFMA")[0] == '0'))
        if (const DwMfmaGeom mf = dw_mfma_geom(*d, dtype); mf.ok && !mf.abl) return mf.chunks;
    const DwMarchGeom m = dw_march_geom(*d, dtype);
    return m.WT ? m.bpc : 0;
}

extern "C" int pasn_dwconv3d_stats_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y, float* ws, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps, float* stat, float* pool_u,
                                       const pasn_conv_desc* d, int dtype, void* stream) {
    return pasn_dwconv3d_stats_fwd_g(x, w, scale, bias, y, ws, gamma, beta, running_mean, running_var, momentum, eps, stat, pool_u, d, dtype, 1, stream);
}

extern "C" int pasn_dwconv3d_stats_fwd_g(const void* x, const float* w, const float* scale, const float* bias, void* y, float* ws, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float eps, float* stat, float* pool_u,
                                         const pasn_conv_desc* d, int dtype, int groups, void* stream) {
    PASN_REQUIRE(x && w && scale && bias && y && ws && stat && d, "null pointer");
    GROUPS_OK(d->N, groups);
    PASN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean / running_var go together");
    PASN_REQUIRE(d->act == PASN_ACT_NONE, "the training stencil writes the raw conv output");
    const int rows = pasn_dwconv3d_stats_rows(d, dtype);
    PASN_REQUIRE(rows > 0, "layer not covered (pasn_dwconv3d_stats_rows returns 0)");
    hipStream_t s = (hipStream_t)stream;
    const int Cp = d->Cout_p, S = d->To * d->Ho * d->Wo;
    int rc;
    const DwMfmaGeom mf = dw_mfma_geom(*d, dtype);
    if (mf.ok && !mf.abl && mf.chunks == rows && !(tune("PASN_DW_STATS_MFMA") && tune("PASN_DW_STATS_MFMA")[0] == '0')) {
        rc = launch_dw_mfma(x, w, scale, bias, y, ws, *d, mf, s, 1, running_mean);  // moments shifted by the running mean (as it stands now:
    } else {                                                                           // the finalize kernel reads it before updating it)
        const DwMarchGeom m = dw_march_geom(*d, dtype);
        rc = launch_dw_march(x, w, scale, bias, y, ws, *d, m, s, DwSeArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}, 1, nullptr, running_mean);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel<__bf16>, dim3(ceil_div(Cp, 16)), dim3(256), 0, s, ws, (const __bf16*)nullptr, gamma, beta, running_mean,
                       running_var, momentum, eps, stat, pool_u, d->N, S, d->Cout, Cp, rows, (const float*)running_mean, d->N / groups);
    return check_launch("dwconv3d_stats_fwd");
}

// Input gradient of a stride-1 "same" depthwise conv (the forward stencil with reversed taps) AND the backward sums of the unit that
// produced the conv's input, in one pass: replaces pasn_dwconv3d_fwd (as dgrad) + pasn_unit_bwd_reduce(mode 3) of the X3D conv_a units.
extern "C" int pasn_dwconv3d_dgrad_reduce_rows(const pasn_conv_desc* d, int dtype) {
    if (!d || d->sw != 1 || d->sh != 1) return 0;
    // OPT-IN (PASN_DW_DGRAD_REDUCE=1): measured, the fusion does not pay -- the stencil is bound by vector-instruction issue, and the ~25
    // extra instructions per output vector cost what the separate pass did (X3D-S: 22 launches 2.31 ms fused vs 1.37 + 0.87 ms)
    const char* on = tune("PASN_DW_DGRAD_REDUCE");
    if (!on || on[0] != '1') return 0;
    if (pasn_dwconv3d_stats_rows(d, dtype) == 0) return 0;
    const DwMarchGeom m = dw_march_geom(*d, dtype);
    return dw_march_red_ok(*d, m) ? m.bpc : 0;
}

extern "C" int pasn_dwconv3d_dgrad_reduce(const void* dy, const float* w_flipped, const float* scale, const float* bias, void* dx,
                                          const void* y_prev, const float* stat_prev, int act_prev, float* ws, float* coef, float* dgamma,
                                          float* dbeta, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(dy && w_flipped && scale && bias && dx && y_prev && stat_prev && ws && coef && d, "null pointer");
    PASN_REQUIRE(d->act == PASN_ACT_NONE, "a gradient pass has no activation");
    const int rows = pasn_dwconv3d_dgrad_reduce_rows(d, dtype);
    PASN_REQUIRE(rows > 0, "layer not covered (pasn_dwconv3d_dgrad_reduce_rows returns 0)");
    hipStream_t s = (hipStream_t)stream;
    const int Cp = d->Cout_p, S = d->To * d->Ho * d->Wo;
    const DwMarchGeom m = dw_march_geom(*d, dtype);
    const DwRedArgs rd = {y_prev, stat_prev, act_prev};
    const int rc = launch_dw_march(dy, w_flipped, scale, bias, dx, ws, *d, m, s, DwSeArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}, 1, &rd);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(Cp, 16)), dim3(256), 0, s, ws, coef, dgamma, dbeta, d->N, S, d->Cout, Cp, rows, d->N);
    return check_launch("dwconv3d_dgrad_reduce");
}

extern "C" int pasn_affine_act_fwd(const void* y, const float* stat, const void* residual, const float* gate, void* a, int N, int S,
                                   int C, int Cp, int act, int dtype, void* stream) {
    return pasn_affine_act_fwd_g(y, stat, residual, gate, a, N, S, C, Cp, act, dtype, 1, stream);
}

extern "C" int pasn_affine_act_fwd_g(const void* y, const float* stat, const void* residual, const float* gate, void* a, int N, int S,
                                     int C, int Cp, int act, int dtype, int groups, void* stream) {
    ROWS_ARGS_OK(N, S, C, Cp);
    GROUPS_OK(N, groups);
    PASN_REQUIRE(y && stat && a, "null pointer");
    const RowGeom g = row_geom(N, S, Cp);
    const dim3 grid(g.chunks, N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16)
        hipLaunchKernelGGL(affine_act_fwd_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)y, stat, (const __bf16*)residual, gate, (__bf16*)a,
                           S, Cp, g.CG, g.LPR, g.rows_per_chunk, act, N / groups);
    else
        hipLaunchKernelGGL(affine_act_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)y, stat, (const float*)residual, gate, (float*)a, S,
                           Cp, g.CG, g.LPR, g.rows_per_chunk, act, N / groups);
    return check_launch("affine_act_fwd");
}

template <typename T>
static void launch_grad_pass(int mode, void* d, const void* y, const float* stat, const void* res, const float* gate, const float* add,
                             float* ws, int N, int S, int Cp, const RowGeom& g, int act, hipStream_t s, int gdiv) {
    const dim3 grid(g.chunks, N);
#define GP(M)                                                                                                                      \
    hipLaunchKernelGGL((grad_pass_kernel<T, M>), grid, dim3(256), 0, s, (T*)d, (const T*)y, stat, (const T*)res, gate, add, ws, S, Cp, g.CG, \
                       g.LPR, g.rows_per_chunk, g.chunks, act, (int)(mode != 3), gdiv)
    if (mode == 0 || mode == 3) GP(0);
    else if (mode == 1) GP(1);
    else if (mode == 4) GP(4);
    else GP(2);
#undef GP
}

extern "C" int pasn_unit_bwd_reduce(int mode, void* d, const void* y, const float* stat, const void* residual, const float* gate,
                                    const float* add, float* ws, float* coef, float* dgamma, float* dbeta, int N, int S, int C, int Cp,
                                    int act, int dtype, void* stream) {
    return pasn_unit_bwd_reduce_g(mode, d, y, stat, residual, gate, add, ws, coef, dgamma, dbeta, N, S, C, Cp, act, dtype, 1, stream);
}

extern "C" int pasn_unit_bwd_reduce_g(int mode, void* d, const void* y, const float* stat, const void* residual, const float* gate,
                                      const float* add, float* ws, float* coef, float* dgamma, float* dbeta, int N, int S, int C, int Cp,
                                      int act, int dtype, int groups, void* stream) {
    ROWS_ARGS_OK(N, S, C, Cp);
    GROUPS_OK(N, groups);
    PASN_REQUIRE(mode >= 0 && mode <= 4, "mode must be 0 .. 4");
    PASN_REQUIRE(d && y && stat && ws, "null pointer");
    PASN_REQUIRE(mode == 1 || mode == 4 || coef, "coef is required for modes 0, 2 and 3");
    PASN_REQUIRE(mode == 0 || mode == 3 || gate, "modes 1, 2 and 4 need the gate");
    PASN_REQUIRE(mode != 3 || residual == nullptr, "mode 3 leaves d untouched: a residual branch needs the differentiated d of mode 0");
    const RowGeom g = row_geom(N, S, Cp);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16) launch_grad_pass<__bf16>(mode, d, y, stat, residual, gate, add, ws, N, S, Cp, g, act, s, N / groups);
    else launch_grad_pass<float>(mode, d, y, stat, residual, gate, add, ws, N, S, Cp, g, act, s, N / groups);
    if (mode != 1 && mode != 4)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(Cp, 16)), dim3(256), 0, s, ws, coef, dgamma, dbeta, N, S, C, Cp, g.chunks, N / groups);
    return check_launch("unit_bwd_reduce");
}

extern "C" int pasn_bn_bwd_apply(const void* d, const void* y, const float* stat, const float* coef, void* dy, int N, int S, int C, int Cp,
                                 int act, int dtype, void* stream) {
    return pasn_bn_bwd_apply_g(d, y, stat, coef, dy, N, S, C, Cp, act, dtype, 1, stream);
}

extern "C" int pasn_bn_bwd_apply_g(const void* d, const void* y, const float* stat, const float* coef, void* dy, int N, int S, int C, int Cp,
                                   int act, int dtype, int groups, void* stream) {
    ROWS_ARGS_OK(N, S, C, Cp);
    GROUPS_OK(N, groups);
    PASN_REQUIRE(d && y && stat && coef && dy, "null pointer");
    const RowGeom g = row_geom(N, S, Cp);
    const dim3 grid(g.chunks, N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)d, (const __bf16*)y, stat, coef, (__bf16*)dy, S, Cp,
                           g.CG, g.LPR, g.rows_per_chunk, act, N / groups);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)d, (const float*)y, stat, coef, (float*)dy, S, Cp, g.CG,
                           g.LPR, g.rows_per_chunk, act, N / groups);
    return check_launch("bn_bwd_apply");
}

extern "C" size_t pasn_se_bwd_workspace_floats(int N, int C, int Cse) { return (size_t)N * (2 * (size_t)C * Cse + Cse + C); }

extern "C" int pasn_se_gate_bwd(const float* ws, const float* pool_u, const float* w1, const float* b1, const float* w2, const float* b2,
                                float* add, float* pn, float* dw1, float* db1, float* dw2, float* db2, int N, int S, int C, int Cp, int Cse,
                                void* stream) {
    ROWS_ARGS_OK(N, S, C, Cp);
    PASN_REQUIRE(Cse > 0 && ws && pool_u && w1 && b1 && w2 && b2 && add && pn && dw1 && db1 && dw2 && db2, "null pointer");
    const size_t lds = (2 * (size_t)C + 2 * Cse) * sizeof(float);
    PASN_REQUIRE(lds <= 64 * 1024, "squeeze-excite width above the LDS budget");
    const RowGeom g = row_geom(N, S, Cp);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(N), dim3(256), lds, s, ws, g.chunks, pool_u, w1, b1, w2, b2, add, pn, S, C, Cp, Cse, (const float*)nullptr,
                       (float*)nullptr, N);
    const size_t len = 2 * (size_t)C * Cse + Cse + C;
    hipLaunchKernelGGL(se_sum_over_clips_kernel, dim3(ceil_div((long)len, 256)), dim3(256), 0, s, pn, dw1, db1, dw2, db2, N, C, Cse);
    return check_launch("se_gate_bwd");
}

// coef / dgamma / dbeta of a squeeze-excite unit's norm from the per-clip totals (A1, A2, A3) = (sum d', sum d' yhat, sum yhat) that
// se_mlp_bwd_kernel left in chunk 0 of the mode-4 partials:  sum d'' = sum_n gate A1 + S add,  sum d'' yhat = sum_n gate A2 + add A3.
__global__ __launch_bounds__(256) void se_bn_coef_kernel(const float* __restrict__ ws3, int chunks, const float* __restrict__ gate,
                                                         const float* __restrict__ add, float* __restrict__ coef, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int N, int S, int C, int Cp, int gdiv) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Cp) return;
    float dg = 0.0f, db = 0.0f;
    for (int lo = 0; lo < N; lo += gdiv) {  // one statistics group after the other
        const int hi = lo + gdiv;
        float b1 = 0.0f, b2 = 0.0f;
        for (int n0 = lo; n0 < hi; n0 += 8) {  // 8 clips' loads in flight, the sums in clip order
            float q0[8], q1[8], q2[8], g[8], a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = min(n0 + u, hi - 1);
                const float* q = ws3 + (size_t)n * chunks * 3 * Cp + c;
                q0[u] = q[0];
                q1[u] = q[Cp];
                q2[u] = q[2 * Cp];
                g[u] = gate[(size_t)n * Cp + c];
                a[u] = add[(size_t)n * Cp + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (n0 + u < hi) {
                    b1 += fmaf(g[u], q0[u], (float)S * a[u]);
                    b2 += fmaf(g[u], q1[u], a[u] * q2[u]);
                }
        }
        const float R = (float)gdiv * (float)S;
        float* cf = coef + (size_t)(lo / gdiv) * 2 * Cp;
        cf[c] = b1 / R;
        cf[Cp + c] = b2 / R;
        dg += b2;
        db += b1;
    }
    if (c < C) {
        if (dgamma) dgamma[c] = dg;
        if (dbeta) dbeta[c] = db;
    }
}

// dy = sc * (d'' - m1 - yhat * m2) with d'' = d' * gate[n][c] + add[n][c] formed on the fly (d' as left by mode 4)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_se_kernel(const T* __restrict__ d, const T* __restrict__ y, const float* __restrict__ stat,
                                                              const float* __restrict__ coef, const float* __restrict__ gate,
                                                              const float* __restrict__ add, T* __restrict__ dy, int S, int Cp, int CG, int LPR,
                                                              int rows_per_chunk, int gdiv) {
    const int n = blockIdx.y, ch = blockIdx.x;
    const int cg = threadIdx.x % LPR, rl = threadIdx.x / LPR, RL = 256 / LPR;
    if (cg >= CG || rl >= RL) return;
    stat += (size_t)(n / gdiv) * 4 * Cp;  // this clip's statistics group
    coef += (size_t)(n / gdiv) * 2 * Cp;
    constexpr int U = ROWS_U;
    float mean[8], invstd[8], sc[8], m1[8], m2[8], g[8], ad[8];
    load8(stat + cg * 8, mean);
    load8(stat + Cp + cg * 8, invstd);
    load8(stat + 2 * Cp + cg * 8, sc);
    load8(coef + cg * 8, m1);
    load8(coef + Cp + cg * 8, m2);
    load8(gate + (size_t)n * Cp + cg * 8, g);
    load8(add + (size_t)n * Cp + cg * 8, ad);
    // dy = sc (d' gate + add - m1 - (y - mean) invstd m2) = sg d' + bq (y - mean) + cq
    float sg[8], bq[8], cq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sg[j] = sc[j] * g[j];
        bq[j] = -sc[j] * invstd[j] * m2[j];
        cq[j] = sc[j] * (ad[j] - m1[j]);
    }
    const int r0 = ch * rows_per_chunk, r1 = min(S, r0 + rows_per_chunk);
    const size_t clip = (size_t)n * S * Cp;
    const __amdgpu_buffer_rsrc_t yrs = row_rsrc(y + clip, r1, Cp), drs = row_rsrc(d + clip, r1, Cp), ors = row_rsrc(dy + clip, r1, Cp);
    const unsigned rowb = (unsigned)Cp * (unsigned)sizeof(T), cgo = (unsigned)cg * 8u * (unsigned)sizeof(T);
    for (int r = r0 + rl; r < r1; r += U * RL) {
        unsigned o[U];
        Raw8<T> yv[U], dr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o[u] = (unsigned)(r + u * RL) * rowb + cgo;  // past the chunk: out of range -- zeros in, nothing out
            yv[u] = load_raw8<T>(yrs, o[u]);
            dr[u] = load_raw8<T>(drs, o[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float v[8], dv[8];
            raw_to_f8<T>(yv[u].r, v);
            raw_to_f8<T>(dr[u].r, dv);
#pragma unroll
            for (int j = 0; j < 8; ++j) dv[j] = fmaf(sg[j], dv[j], fmaf(bq[j], v[j] - mean[j], cq[j]));
            store8(ors, o[u], dv, (const T*)nullptr);
        }
    }
}

extern "C" int pasn_se_gate_bwd_stat(float* ws3, const float* pool_u, const float* stat, const float* gate, const float* w1, const float* b1,
                                     const float* w2, const float* b2, float* add, float* pn, float* dw1, float* db1, float* dw2, float* db2,
                                     float* coef, float* dgamma, float* dbeta, int N, int S, int C, int Cp, int Cse, void* stream) {
    return pasn_se_gate_bwd_stat_g(ws3, pool_u, stat, gate, w1, b1, w2, b2, add, pn, dw1, db1, dw2, db2, coef, dgamma, dbeta, N, S, C, Cp, Cse, 1, stream);
}

extern "C" int pasn_se_gate_bwd_stat_g(float* ws3, const float* pool_u, const float* stat, const float* gate, const float* w1, const float* b1,
                                       const float* w2, const float* b2, float* add, float* pn, float* dw1, float* db1, float* dw2, float* db2,
                                       float* coef, float* dgamma, float* dbeta, int N, int S, int C, int Cp, int Cse, int groups, void* stream) {
    ROWS_ARGS_OK(N, S, C, Cp);
    GROUPS_OK(N, groups);
    PASN_REQUIRE(Cse > 0 && ws3 && pool_u && stat && gate && w1 && b1 && w2 && b2 && add && pn && dw1 && db1 && dw2 && db2 && coef, "null pointer");
    const size_t lds = (2 * (size_t)C + 2 * Cse) * sizeof(float);
    PASN_REQUIRE(lds <= 64 * 1024, "squeeze-excite width above the LDS budget");
    const RowGeom g = row_geom(N, S, Cp);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(N), dim3(256), lds, s, (const float*)nullptr, g.chunks, pool_u, w1, b1, w2, b2, add, pn, S, C, Cp, Cse,
                       stat, ws3, N / groups);
    const size_t len = 2 * (size_t)C * Cse + Cse + C;
    hipLaunchKernelGGL(se_sum_over_clips_kernel, dim3(ceil_div((long)len, 256)), dim3(256), 0, s, pn, dw1, db1, dw2, db2, N, C, Cse);
    hipLaunchKernelGGL(se_bn_coef_kernel, dim3(ceil_div(Cp, 256)), dim3(256), 0, s, ws3, g.chunks, gate, add, coef, dgamma, dbeta, N, S, C, Cp, N / groups);
    return check_launch("se_gate_bwd_stat");
}

extern "C" int pasn_bn_bwd_apply_se(const void* d, const void* y, const float* stat, const float* coef, const float* gate, const float* add,
                                    void* dy, int N, int S, int C, int Cp, int dtype, void* stream) {
    return pasn_bn_bwd_apply_se_g(d, y, stat, coef, gate, add, dy, N, S, C, Cp, dtype, 1, stream);
}

extern "C" int pasn_bn_bwd_apply_se_g(const void* d, const void* y, const float* stat, const float* coef, const float* gate, const float* add,
                                      void* dy, int N, int S, int C, int Cp, int dtype, int groups, void* stream) {
    ROWS_ARGS_OK(N, S, C, Cp);
    GROUPS_OK(N, groups);
    PASN_REQUIRE(d && y && stat && coef && gate && add && dy, "null pointer");
    const RowGeom g = row_geom(N, S, Cp);
    const dim3 grid(g.chunks, N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_se_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)d, (const __bf16*)y, stat, coef, gate, add, (__bf16*)dy,
                           S, Cp, g.CG, g.LPR, g.rows_per_chunk, N / groups);
    else
        hipLaunchKernelGGL(bn_bwd_apply_se_kernel<float>, grid, dim3(256), 0, s, (const float*)d, (const float*)y, stat, coef, gate, add, (float*)dy, S,
                           Cp, g.CG, g.LPR, g.rows_per_chunk, N / groups);
    return check_launch("bn_bwd_apply_se");
}

extern "C" int pasn_scatter_strided(const void* src, void* dst, const pasn_conv_desc* d, int accumulate, int dtype, void* stream) {
    PASN_REQUIRE(src && dst && d, "null pointer");
    // src: [N][To][Ho][Wo][Cin_p] (compact), dst: [N][Ti][Hi][Wi][Cin_p]; element (t,h,w) of src lands on (t*st, h*sh, w*sw)
    PASN_REQUIRE(d->Cin_p % 8 == 0 && d->st > 0 && d->sh > 0 && d->sw > 0, "bad descriptor");
    PASN_REQUIRE((d->To - 1) * d->st < d->Ti && (d->Ho - 1) * d->sh < d->Hi && (d->Wo - 1) * d->sw < d->Wi, "source does not fit the target");
    const size_t total = (size_t)d->N * d->Ti * d->Hi * d->Wi * (d->Cin_p / 8);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16)
        hipLaunchKernelGGL(scatter_strided_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)src, (__bf16*)dst, d->N, d->To, d->Ho, d->Wo,
                           d->Ti, d->Hi, d->Wi, d->st, d->sh, d->sw, d->Cin_p, accumulate);
    else
        hipLaunchKernelGGL(scatter_strided_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)src, (float*)dst, d->N, d->To, d->Ho, d->Wo,
                           d->Ti, d->Hi, d->Wi, d->st, d->sh, d->sw, d->Cin_p, accumulate);
    return check_launch("scatter_strided");
}

extern "C" int pasn_add_inplace(void* a, const void* b, size_t elements, int dtype, void* stream) {
    PASN_REQUIRE(a && b && elements % 8 == 0, "element count must be a multiple of 8");
    const size_t groups = elements / 8;
    const int blocks = (int)std::min<size_t>((groups + 255) / 256, 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16) hipLaunchKernelGGL(add_inplace_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (__bf16*)a, (const __bf16*)b, groups);
    else hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)a, (const float*)b, groups);
    return check_launch("add_inplace");
}

extern "C" int pasn_maxpool3d_bwd(const void* x, const void* dy, void* dx, const pasn_conv_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(x && dy && dx && d, "null pointer");
    PASN_REQUIRE(d->Cin_p == d->Cout_p && d->Cin_p % 8 == 0, "pooling keeps the channel stride");
    const size_t total = (size_t)d->N * d->Ti * d->Hi * d->Wi * (d->Cin_p / 8);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 1 << 20);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, (const __bf16*)dy, (__bf16*)dx, *d);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)dx, *d);
    return check_launch("maxpool3d_bwd");
}
