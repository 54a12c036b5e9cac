// Head A (PPNet): squared-L2 distance map, global min / argmin, log activation, last layer -- one launch.
//
//   dist[n][p][s] = relu( ||z_s||^2 - 2 z_s.p + ||p||^2 )            (reference ProtoPNet.py:189-207)
//
// x.p is the only dense contraction and runs on MFMA (A = prototype tile, B = feature tile, both in
// fragment shape); ||z_s||^2 -- the reference's ones-convolution, computed there P times -- and ||p||^2 fall
// out of the very fragments that feed the MFMA (each lane squares the 16 bytes it loaded; the two lane
// halves are combined with one cross-half shuffle).  One block per image, a wave per 32-position tile.
#include "common.h"

namespace pasn {

__device__ __forceinline__ void lex_min(float& v, int& i, float ov, int oi) {
    if (ov < v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

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
    float* wmin = sm;                                   // [4][32]
    int* widx = reinterpret_cast<int*>(sm + 128);       // [4][32]
    float* mins = sm + 256;                             // [ptiles*32]

    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int stiles = (S + 31) / 32, ptiles = (P + 31) / 32;
    const int Dk = (D + KSTEP - 1) / KSTEP * KSTEP;

    for (int pt = 0; pt < ptiles; ++pt) {
        const int p0 = pt * 32;
        float rmin[16];
        int ridx[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            rmin[i] = INFINITY;
            ridx[i] = 0x7fffffff;
        }
        for (int st = wave; st < stiles; st += 4) {
            const int s = st * 32 + c;
            const bool sv = s < S;
            const T* zp = z + ((long)n * S + (sv ? s : 0)) * Dp;
            const bool pv = (p0 + c) < P;
            const float* pp = protos + (long)(pv ? p0 + c : 0) * D;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            float x2 = 0.0f, p2 = 0.0f;
            for (int k0 = 0; k0 < Dk; k0 += KSTEP) {
                const int k = k0 + h * CH;
                frag a, b;
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float v = (pv && (k + j) < D) ? pp[k + j] : 0.0f;
                    a[j] = (T)v;
                    const float vr = (float)a[j];
                    p2 = fmaf(vr, vr, p2);
                }
                if (sv && k < Dp) {
                    b = load_frag<T>(zp + k);
                } else {
                    b = zero_frag<T>();
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float v = (float)b[j];
                    x2 = fmaf(v, v, x2);
                }
                mma32(acc, a, b);
            }
            x2 += __shfl_xor(x2, 32);
            p2 += __shfl_xor(p2, 32);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = acc_row(i, h);
                const float p2r = __shfl(p2, rr);
                const float dv = fmaxf(x2 + (-2.0f * acc[i] + p2r), 0.0f);
                if (sv && (p0 + rr) < P) {
                    if (dist) dist[((long)n * P + p0 + rr) * S + s] = dv;
                    if (dv < rmin[i]) {  // s grows along a lane's tiles: strict '<' keeps the first minimum
                        rmin[i] = dv;
                        ridx[i] = s;
                    }
                }
            }
        }
        // min over the 32 positions held by the lanes of each half, first index on ties
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(rmin[i], off);
                const int oi = __shfl_xor(ridx[i], off);
                lex_min(rmin[i], ridx[i], ov, oi);
            }
        }
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = acc_row(i, h);
                wmin[wave * 32 + rr] = rmin[i];
                widx[wave * 32 + rr] = ridx[i];
            }
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int r = threadIdx.x;
            float v = wmin[r];
            int i = widx[r];
#pragma unroll
            for (int q = 1; q < 4; ++q) lex_min(v, i, wmin[q * 32 + r], widx[q * 32 + r]);
            mins[p0 + r] = v;
            if (p0 + r < P) {
                min_dist[(long)n * P + p0 + r] = v;
                if (argmin) argmin[(long)n * P + p0 + r] = i;
            }
        }
        __syncthreads();
    }
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
    const int ptiles = (P + 31) / 32;
    const size_t lds = (size_t)(256 + ptiles * 32) * sizeof(float);
    PASN_REQUIRE(lds <= 64 * 1024, "too many prototypes for one block");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_F32)
        hipLaunchKernelGGL((l2_head_kernel<float>), dim3(N), dim3(256), lds, s, (const float*)z, protos, fc_w, dist,
                           min_dist, argmin, logits, S, D, Dp, P, K, activation, eps);
    else if (dtype == PASN_BF16)
        hipLaunchKernelGGL((l2_head_kernel<__bf16>), dim3(N), dim3(256), lds, s, (const __bf16*)z, protos, fc_w, dist,
                           min_dist, argmin, logits, S, D, Dp, P, K, activation, eps);
    else {
        set_error("pasn_l2_head_fwd: unknown dtype");
        return PASN_ERR_ARG;
    }
    return check_launch("l2_head_kernel");
}
