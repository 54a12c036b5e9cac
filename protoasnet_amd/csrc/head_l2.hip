// Head A (PPNet): squared-L2 distance map, global min / argmin, log activation, last layer -- one launch.
//
//   dist[n][p][s] = relu( ||z_s||^2 - 2 z_s.p + ||p||^2 )            (reference ProtoPNet.py:189-207)
//
// x.p is the only dense contraction and runs on MFMA (A = feature tile, B = prototype tile); ||z_s||^2 -- the reference's
// ones-convolution, computed there P times -- falls out of the feature fragments that feed the MFMA, ||p||^2 is computed once
// per block while the prototypes are staged.
//
// Round 2 (round 1: 76 us for 8 images of 7x7x512 -- every lane fetched its prototype fragment from global memory with eight
// predicated scalar loads per k-step, a dependent L2 round trip 64 times per tile and prototype tile):
//   * the prototypes are staged ONCE per block into LDS in the compute dtype, rows padded by one 16-byte slot (a 1024-byte row
//     stride would put all 32 rows of a fragment read on the same banks), in chunks of 64 prototypes;
//   * the k-loop holds no global dependence but the feature fragments, which are independent 16-byte loads issued 8 k-steps deep;
//   * both prototype tiles of a chunk share each feature fragment (z is read once per chunk, not once per 32 prototypes).
// One block per image; a wave per 32-position tile (7x7 maps: two waves busy -- the launch is latency, not bandwidth).
#include "common.h"

namespace pasn {

__device__ __forceinline__ void lex_min(float& v, int& i, float ov, int oi) {
    if (ov < v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

constexpr int L2H_PCH = 64;  // prototypes per LDS chunk (two 32-row MFMA tiles)

template <typename T>
__global__ __launch_bounds__(256) void l2_head_kernel(const T* __restrict__ z, const float* __restrict__ protos,
                                                      const float* __restrict__ fc_w, float* __restrict__ dist,
                                                      float* __restrict__ min_dist, int32_t* __restrict__ argmin,
                                                      float* __restrict__ logits, int S, int D, int Dp, int P, int K,
                                                      int activation, float eps) {
    using frag = typename Traits<T>::frag;
    constexpr int CH = Traits<T>::CH;
    constexpr int KSTEP = Traits<T>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Dk = (D + KSTEP - 1) / KSTEP * KSTEP;
    const int rowe = Dk + CH;                           // LDS row stride in elements: + one 16-byte slot
    const int pall = (P + 31) / 32 * 32;
    float* wmin = sm;                                   // [4][64]
    int* widx = reinterpret_cast<int*>(sm + 256);       // [4][64]
    float* p2s = sm + 512;                              // [64]
    float* mins = sm + 576;                             // [pall]
    T* pl = reinterpret_cast<T*>(sm + 576 + pall);      // [64][rowe]

    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int stiles = (S + 31) / 32;
    const int ksteps = Dk / KSTEP;

    for (int p0 = 0; p0 < P; p0 += L2H_PCH) {
        const int rows = min(L2H_PCH, pall - p0);  // 32 or 64
        const int ptl = rows / 32;
        // ---- stage this chunk's prototypes (zero rows / columns beyond P / D) and their squared norms -------------------------
        __syncthreads();  // the previous chunk's fragments are no longer read
        // a wave takes rows wave, wave + 4, ...; EIGHT rows' 16-byte loads are in flight per lane and step (a wave walking its rows one
        // at a time paid one L2 round trip per row -- 16 of them, most of the launch), no integer division, packed LDS stores
        if ((D & 3) == 0) {
            const int d4 = D >> 2;
            for (int rb = wave; rb < rows; rb += 32) {
                for (int g = lane; g < d4; g += 64) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = rb + 4 * u;
                        const bool ok = r < rows && (p0 + r) < P;
                        v[u] = *reinterpret_cast<const f32x4*>(protos + (ok ? (long)(p0 + r) * D + 4 * g : 0));
                        if (!ok) v[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = rb + 4 * u;
                        if (r < rows) {
                            float t[4] = {v[u][0], v[u][1], v[u][2], v[u][3]};
                            store4(pl + r * rowe + 4 * g, t);
                        }
                    }
                }
            }
            for (int i = threadIdx.x; i < rows * (Dk - D); i += 256)  // columns D .. Dk of every row (the row pad slot is never read)
                pl[(i / (Dk - D)) * rowe + D + i % (Dk - D)] = (T)0.0f;
        } else {
            const int total = rows * rowe;
            for (int i = threadIdx.x; i < total; i += 256) {
                const int r = i / rowe, k = i - r * rowe;
                pl[i] = (T)(((p0 + r) < P && k < D) ? protos[(long)(p0 + r) * D + k] : 0.0f);
            }
        }
        __syncthreads();
        {  // squared norms of what the MFMA will see: four threads per row, independent 16-byte LDS reads, two exchange steps
            const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
            float ss = 0.0f;
            if (r < rows) {
                for (int k = q * CH; k < Dk; k += 4 * CH) {
                    const frag f = *reinterpret_cast<const frag*>(pl + r * rowe + k);
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const float vr = (float)f[j];
                        ss = fmaf(vr, vr, ss);
                    }
                }
            }
            ss += __shfl_xor(ss, 1);
            ss += __shfl_xor(ss, 2);
            if (q == 0 && r < rows) p2s[r] = ss;
        }
        __syncthreads();

        float rmin[2] = {INFINITY, INFINITY};
        int ridx[2] = {0x7fffffff, 0x7fffffff};
        for (int st = wave; st < stiles; st += 4) {
            const int s = st * 32 + c;
            const bool sv = s < S;
            const T* zrow = z + ((long)n * S + (sv ? s : 0)) * Dp;
            const T* b0p = pl + c * rowe + h * CH;
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
            float x2 = 0.0f;
            for (int ks0 = 0; ks0 < ksteps; ks0 += 8) {
                frag a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {  // raw, unconditional, address-clamped loads: all eight in flight before the first MFMA
                    // (a predicated load is a branch + a wait per load: eight sequential L2 round trips per group)
                    a[u] = load_frag<T>(zrow + min((ks0 + u) * KSTEP + h * CH, Dp - CH));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {  // mask what lies beyond the row (or the image) afterwards
                    const int k = (ks0 + u) * KSTEP + h * CH;
                    if (!(sv && ks0 + u < ksteps && k < Dp)) a[u] = zero_frag<T>();
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (ks0 + u < ksteps) {  // block-uniform
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            const float v = (float)a[u][j];
                            x2 = fmaf(v, v, x2);
                        }
                        const frag b0 = *reinterpret_cast<const frag*>(b0p + (ks0 + u) * KSTEP);
                        mma32(acc[0], a[u], b0);
                        if (ptl == 2) {
                            const frag b1 = *reinterpret_cast<const frag*>(b0p + 32 * rowe + (ks0 + u) * KSTEP);
                            mma32(acc[1], a[u], b1);
                        }
                    }
                }
            }
            x2 += __shfl_xor(x2, 32);  // ||z_s||^2 of position st*32 + c, on lanes c and c + 32
            float xs[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) xs[i] = __shfl(x2, acc_row(i, h));  // ... of the 16 positions this lane holds
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t < ptl) {  // block-uniform
                    const float p2v = p2s[t * 32 + c];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {  // increasing i = increasing position: strict '<' keeps the first minimum
                        const int si = st * 32 + acc_row(i, h);
                        float dv = fmaxf(xs[i] + (-2.0f * acc[t][i] + p2v), 0.0f);
                        acc[t][i] = dv;
                        dv = si < S ? dv : INFINITY;
                        const bool better = dv < rmin[t];
                        rmin[t] = better ? dv : rmin[t];
                        ridx[t] = better ? si : ridx[t];
                    }
                }
            }
            if (dist) {  // uniform; the push / prototype_distances path only: lane = prototype, register = position
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int pr = p0 + t * 32 + c;
                    if (t < ptl && pr < P) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int si = st * 32 + acc_row(i, h);
                            if (si < S) dist[((long)n * P + pr) * S + si] = acc[t][i];
                        }
                    }
                }
            }
        }
        // combine the two lane halves (rows 4h + ...), first index on ties, then the four waves through LDS
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float ov = __shfl_xor(rmin[t], 32);
            const int oi = __shfl_xor(ridx[t], 32);
            const bool take = ov < rmin[t] || (ov == rmin[t] && oi < ridx[t]);
            rmin[t] = take ? ov : rmin[t];
            ridx[t] = take ? oi : ridx[t];
        }
        if (h == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wmin[wave * 64 + t * 32 + c] = rmin[t];
                widx[wave * 64 + t * 32 + c] = ridx[t];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            const int r = threadIdx.x;
            float v = wmin[r];
            int i = widx[r];
#pragma unroll
            for (int q = 1; q < 4; ++q) lex_min(v, i, wmin[q * 64 + r], widx[q * 64 + r]);
            mins[p0 + r] = v;
            if (p0 + r < P) {
                min_dist[(long)n * P + p0 + r] = v;
                if (argmin) argmin[(long)n * P + p0 + r] = i;
            }
        }
    }
    __syncthreads();
    // prototype activation + last layer (no bias)
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float s = 0.0f;
        for (int p = 0; p < P; ++p) {
            const float dv = mins[p];
            const float a = activation == 0 ? logf((dv + 1.0f) / (dv + eps)) : -dv;
            s = fmaf(a, fc_w[(long)k * P + p], s);
        }
        logits[(long)n * K + k] = s;
    }
}

}  // namespace pasn

using namespace pasn;

extern "C" int pasn_l2_head_fwd(const void* z, const float* protos, const float* fc_w, float* dist, float* min_dist,
                                int32_t* argmin, float* logits, int N, int S, int D, int Dp, int P, int K, int dtype,
                                int activation, float eps, void* stream) {
    PASN_REQUIRE(z && protos && fc_w && min_dist && logits, "null pointer");
    PASN_REQUIRE(N > 0 && S > 0 && D > 0 && P > 0 && K > 0, "empty problem");
    PASN_REQUIRE(Dp >= D && Dp % 8 == 0, "Dp must be a multiple of 8 covering D");
    PASN_REQUIRE(activation == 0 || activation == 1, "activation must be 0 (log) or 1 (linear)");
    PASN_REQUIRE(dtype == PASN_F32 || dtype == PASN_BF16, "unknown dtype");
    const int pall = (P + 31) / 32 * 32;
    const int kstep = dtype == PASN_BF16 ? 16 : 8, ch = kstep / 2, es = dtype == PASN_BF16 ? 2 : 4;
    const int Dk = (D + kstep - 1) / kstep * kstep;
    const size_t lds = (size_t)(576 + pall) * sizeof(float) + (size_t)L2H_PCH * (Dk + ch) * es;
    PASN_REQUIRE(lds <= 160 * 1024, "prototype dimension too large for the LDS tile (64 prototypes x D)");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_F32) {
        if (lds > 64 * 1024) PASN_MAX_LDS(160 * 1024, l2_head_kernel<float>);
        hipLaunchKernelGGL((l2_head_kernel<float>), dim3(N), dim3(256), lds, s, (const float*)z, protos, fc_w, dist,
                           min_dist, argmin, logits, S, D, Dp, P, K, activation, eps);
    } else if (dtype == PASN_BF16) {
        if (lds > 64 * 1024) PASN_MAX_LDS(160 * 1024, l2_head_kernel<__bf16>);
        hipLaunchKernelGGL((l2_head_kernel<__bf16>), dim3(N), dim3(256), lds, s, (const __bf16*)z, protos, fc_w, dist,
                           min_dist, argmin, logits, S, D, Dp, P, K, activation, eps);
    } else {
        set_error("pasn_l2_head_fwd: unknown dtype");
        return PASN_ERR_ARG;
    }
    return check_launch("l2_head_kernel");
}
