// warp.hip -- the affine warp of TransformLoss (src/loss/loss.py:257-320): rotation by `angle` about the image centre, isotropic
// `scale`, no translation / shear, bilinear interpolation, fill 0 -- torchvision.transforms.functional.affine on tensors, i.e.
// (torchvision 0.14 functional_tensor: _get_inverse_affine_matrix + _gen_affine_grid + grid_sample(align_corners=False) with an
// appended ones channel for the fill):
//     centred output pixel (xc, yc) = (j - W/2 + 0.5, i - H/2 + 0.5)
//     source  (xs, ys) = ( cos a * xc + sin a * yc, -sin a * xc + cos a * yc) / scale,   pixel coords (xs + W/2 - 0.5, ys + H/2 - 0.5)
//     out = bilinear(img; zeros outside) * bilinear(ones; zeros outside)
// Applied to every H x W plane of a planar tensor (the clip: N*3*T planes; the occurrence maps: N*P*T' planes), the same
// transform for all planes, as the reference reshapes (N, D, T, H, W) to (N*T, D, H, W) before the call.  HBM-bound gather.
#include "common.h"

namespace pasn {

struct WarpGeom {
    float c, s, inv_scale;
};

__device__ __forceinline__ void warp_taps(int i, int j, int H, int W, const WarpGeom g, int (&xi)[2], int (&yi)[2], float (&wx)[2], float (&wy)[2],
                                          float& mask) {
    const float xc = (float)j - 0.5f * W + 0.5f, yc = (float)i - 0.5f * H + 0.5f;
    const float xs = (g.c * xc + g.s * yc) * g.inv_scale + 0.5f * W - 0.5f;
    const float ys = (-g.s * xc + g.c * yc) * g.inv_scale + 0.5f * H - 0.5f;
    const float x0 = floorf(xs), y0 = floorf(ys);
    xi[0] = (int)x0;
    xi[1] = xi[0] + 1;
    yi[0] = (int)y0;
    yi[1] = yi[0] + 1;
    wx[1] = xs - x0;
    wx[0] = 1.0f - wx[1];
    wy[1] = ys - y0;
    wy[0] = 1.0f - wy[1];
    mask = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (yi[a] >= 0 && yi[a] < H && xi[b] >= 0 && xi[b] < W) mask += wy[a] * wx[b];
}

template <typename T>
__global__ __launch_bounds__(256) void affine_warp_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long planes, int H, int W, WarpGeom g) {
    const long total = planes * H * W;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx % W), i = (int)((idx / W) % H);
        const long m = idx / ((long)H * W);
        int xi[2], yi[2];
        float wx[2], wy[2], mask;
        warp_taps(i, j, H, W, g, xi, yi, wx, wy, mask);
        const T* p = x + m * H * W;
        float v = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (yi[a] >= 0 && yi[a] < H && xi[b] >= 0 && xi[b] < W) v = fmaf(wy[a] * wx[b], (float)p[(long)yi[a] * W + xi[b]], v);
        y[idx] = (T)(v * mask);
    }
}

// adjoint: dx[m][src] += w * mask * dy[m][i][j]   (fp32; dx zeroed by the caller)
__global__ __launch_bounds__(256) void affine_warp_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long planes, int H, int W,
                                                              WarpGeom g) {
    const long total = planes * H * W;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx % W), i = (int)((idx / W) % H);
        const long m = idx / ((long)H * W);
        int xi[2], yi[2];
        float wx[2], wy[2], mask;
        warp_taps(i, j, H, W, g, xi, yi, wx, wy, mask);
        const float d = dy[idx] * mask;
        float* p = dx + m * H * W;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (yi[a] >= 0 && yi[a] < H && xi[b] >= 0 && xi[b] < W) unsafeAtomicAdd(p + (long)yi[a] * W + xi[b], wy[a] * wx[b] * d);
    }
}

static WarpGeom warp_geom(float angle_deg, float scale) {
    const double r = (double)angle_deg * 3.14159265358979323846 / 180.0;
    return WarpGeom{(float)cos(r), (float)sin(r), 1.0f / scale};
}

}  // namespace pasn

using namespace pasn;

extern "C" int pasn_affine_warp_fwd(const void* x, void* y, long planes, int H, int W, float angle_deg, float scale, int dtype, void* stream) {
    PASN_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && scale > 0.0f, "bad arguments");
    const long total = planes * H * W;
    const int blocks = (int)std::min<long>((total + 255) / 256, 1 << 20);
    const WarpGeom g = warp_geom(angle_deg, scale);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16) hipLaunchKernelGGL(affine_warp_fwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, (__bf16*)y, planes, H, W, g);
    else hipLaunchKernelGGL(affine_warp_fwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)y, planes, H, W, g);
    return check_launch("affine_warp_fwd");
}

extern "C" int pasn_affine_warp_bwd(const float* dy, float* dx, long planes, int H, int W, float angle_deg, float scale, void* stream) {
    PASN_REQUIRE(dy && dx && planes > 0 && H > 0 && W > 0 && scale > 0.0f, "bad arguments");
    const long total = planes * H * W;
    const int blocks = (int)std::min<long>((total + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(affine_warp_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, dx, planes, H, W, warp_geom(angle_deg, scale));
    return check_launch("affine_warp_bwd");
}
