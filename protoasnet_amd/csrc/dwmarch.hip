// T-marching depthwise 3x3x3 stencil (stride (1,s,s), pad 1), bf16 activations -- the X3D conv_b of every block.
//
// Why: PMC on the strip kernel (tools/pmc_dw.sh) shows it ISSUE-bound, not bandwidth-bound: 360 VALU instructions per
// 8-channel output vector, of which only 108 are the packed FMAs; the rest is the bf16->fp32 conversion of every input
// vector once per (kt) it is used in (93), register double-buffer copies (46), address arithmetic and edge selects.  On
// the small-spatial stages it is latency-bound instead: 9 dependent load rounds per thread and a 46 KB weight stage per
// block for ~1 strip of work.
//
// Here a thread owns WT consecutive outputs along W of one (ho) row for one 8-channel group and MARCHES ALONG T with
// three accumulator sets in flight (outputs t-1, t, t+1).  Every input row (ti, hi) is loaded and converted ONCE and
// feeds all three kt taps: loads, conversions and address arithmetic per output drop 3x, and there is no double-buffer
// copy (the raw row registers are refilled for the next row right after conversion, so the loads fly under ~430 packed
// FMAs).  T is split into chunks (halo frames recomputed) only as far as needed to keep >= 2 waves per SIMD.
//
// Weights live in LDS as two 16-byte planes per tap ([tap][half][CG][4] fp32) so a wave's ds_read_b128 walks consecutive
// slots (conflict-free; lanes of the same channel group broadcast).
#include "common.h"

namespace pasn {

constexpr int DWM_CGS = 64;   // LDS slot stride of the weight planes = max channel groups (512 channels)
constexpr int DWM_ROWS = 29;  // 27 taps + scale + bias
constexpr int DWM_RED_ROWS = 4;  // RED instances: + the producer unit's (mean, invstd, sc, sh)
static size_t dwm_lds_bytes(int R, int Cp, bool red = false) {
    return (size_t)((DWM_ROWS + (red ? DWM_RED_ROWS : 0)) * 2 * DWM_CGS * 4 + R * Cp) * sizeof(float);
}

template <int SW, int WT, bool RED = false>
__global__ __launch_bounds__(256, 2) void dwconv3d_march_kernel(const __bf16* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ bias,
                                                             __bf16* __restrict__ y, float* __restrict__ pool, pasn_conv_desc d,
                                                             int CG, int R, int strips, int Tc, int bpc, DwSeArgs se, int stats,
                                                             DwRedArgs rd, const float* __restrict__ shift) {
    // [27 taps + scale + bias][2 halves][DWM_CGS slots][4] fp32 (fixed slot stride: every tap is an immediate ds_read
    // offset), then [R][Cp] pool scratch
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NC = (WT - 1) * SW + 3;  // input columns a strip touches
    const int Cp = d.Cout_p;
    constexpr int ROWS = DWM_ROWS + (RED ? DWM_RED_ROWS : 0);
    float* red = lds + ROWS * 2 * DWM_CGS * 4;
    // stage weights | scale | bias: batches of 8 independent 16-byte loads, THEN the LDS writes (a rolled load -> ds_write loop
    // is one L2 round trip per iteration: 14 of them per block on the 432-channel layers, ~10 us of a 35 us launch)
    {
        const int total = ROWS * 2 * CG;
        for (int i0 = threadIdx.x; i0 < total; i0 += 8 * blockDim.x) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * (int)blockDim.x, total - 1);
                const int row = i / (2 * CG), rem = i - row * 2 * CG, half = rem / CG, g = rem - half * CG;
                const float* src = row < 27 ? w + (long)row * Cp : (row == 27 ? scale : row == 28 ? bias : rd.stat + (long)(row - 29) * Cp);
                v[u] = *reinterpret_cast<const f32x4*>(src + g * 8 + half * 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * (int)blockDim.x;
                if (i < total) {
                    const int row = i / (2 * CG), rem = i - row * 2 * CG, half = rem / CG, g = rem - half * CG;
                    *reinterpret_cast<f32x4*>(lds + ((row * 2 + half) * DWM_CGS + g) * 4) = v[u];
                }
            }
        }
    }
    __syncthreads();

    const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);  // whole clips per XCD: halo rows of neighbouring items share an L2
    const int n = lb / bpc, bx = lb % bpc;
    const int nT = (d.To + Tc - 1) / Tc;
    const int items = nT * d.Ho * strips;
    const int item = bx * R + r;
    float psum[8], psq[8];  // psq: second moments for the training path's batch statistics (stats != 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) psum[j] = psq[j] = 0.0f;
    // batch statistics (stats, not RED): moments of (y - k) with a per-channel shift k known before the launch (the running mean; NULL: 0):
    // sum (y - k)^2 does not cancel against the squared mean when |mean| >> std.  The first moment is sum y - count k, taken at the end.
    float kshift[8];
    int kcount = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) kshift[j] = (!RED && stats && shift && cg * 8 + j < d.Cout) ? shift[cg * 8 + j] : 0.0f;  // `shift` holds Cout floats

    if (item < items) {
        const int strip = item % strips;
        const int ho = (item / strips) % d.Ho;
        const int t0 = (item / (strips * d.Ho)) * Tc;
        const int t1 = min(t0 + Tc, d.To);
        const int wo0 = strip * WT;
        const int wi0 = wo0 * SW - 1;
        const int Ti = d.Ti, Hi = d.Hi, Wi = d.Wi;
        // per-thread column byte offsets (clamped into the row) and validity bits; the row base is added per row
        unsigned colofs[NC];
        unsigned colok = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int wi = wi0 + c;
            colofs[c] = (unsigned)(min(max(wi, 0), Wi - 1) * Cp * 2 + cg * 16);
            colok |= (wi >= 0 && wi < Wi) ? (1u << c) : 0u;
        }
        // only the first / last column of a strip can leave the image when there is no ragged strip
        const bool edge1 = d.Wo % WT == 0 && ((d.Wo - 1) * SW + 1 - (Wi - 1)) <= 1;
        const char* xb = reinterpret_cast<const char*>(x);
        const unsigned rowbytes = (unsigned)(Wi * Cp * 2);
        uint4 raw[NC];
        auto issue = [&](int ti, int kh) {  // unconditional loads from a clamped (always valid) row
            const int tic = min(max(ti, 0), Ti - 1);
            const int hic = min(max(ho * SW - 1 + kh, 0), Hi - 1);
            const unsigned base = (unsigned)((n * Ti + tic) * Hi + hic) * rowbytes;
#pragma unroll
            for (int c = 0; c < NC; ++c) raw[c] = *reinterpret_cast<const uint4*>(xb + (size_t)(base + colofs[c]));
        };

        // Three accumulator sets, one per output frame in flight.  Their ROLES rotate (the set that held output ti-1 is
        // emitted, zeroed and becomes output ti+2's), so the frame loop is unrolled x3 with the sets passed by name:
        // no register moves (the rolled version copied 2 x WT x 8 registers per frame).
        float A0[WT][8], A1[WT][8], A2[WT][8];
#pragma unroll
        for (int o = 0; o < WT; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) A0[o][j] = A1[o][j] = A2[o][j] = 0.0f;

        // one input frame ti: P = output ti-1 (kt = 2), C = output ti (kt = 1), N = output ti+1 (kt = 0)
        auto frame = [&](int ti, float (&P)[WT][8], float (&C)[WT][8], float (&N)[WT][8]) {
            const bool tv = ti >= 0 && ti < Ti;
            // opaque per-frame zero: keeps the 54 weight reads of a frame INSIDE the loop (hoisted, they are 216 registers)
            int zo = 0;
            asm volatile("" : "+v"(zo));
            const float* wl = lds + cg * 4 + zo;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hi = ho * SW - 1 + kh;
                const bool rv = tv && hi >= 0 && hi < Hi;
                // convert the row once (columns outside the image become zeros)
                float xr[NC][8];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    uint4 v[1] = {raw[c]};
                    if (!(edge1 && c != 0 && c != NC - 1)) {
                        const bool ok = (colok >> c) & 1u;
                        v[0].x = ok ? v[0].x : 0u;
                        v[0].y = ok ? v[0].y : 0u;
                        v[0].z = ok ? v[0].z : 0u;
                        v[0].w = ok ? v[0].w : 0u;
                    }
                    raw_to_f8<__bf16>(v, xr[c]);
                }
                // refill the raw registers with the next row; it lands under this row's FMAs
                __builtin_amdgcn_sched_barrier(0);  // the refill must not be hoisted above the conversion (a second raw row live)
                if (kh < 2) issue(ti, kh + 1);
                else issue(ti + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (rv) {
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        __builtin_amdgcn_sched_barrier(0);  // keep one kt group (24 weight registers) live at a time
                        float wv[3][8];
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const int tap = (kt * 3 + kh) * 3 + e;
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(wl + (tap * 2 + 0) * DWM_CGS * 4);
                            const f32x4 hh = *reinterpret_cast<const f32x4*>(wl + (tap * 2 + 1) * DWM_CGS * 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                wv[e][j] = lo[j];
                                wv[e][4 + j] = hh[j];
                            }
                        }
                        float (&T_)[WT][8] = kt == 0 ? N : kt == 1 ? C : P;
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                if ((c - e) >= 0 && (c - e) % SW == 0 && (c - e) / SW < WT) {  // resolved at compile time
                                    const int o = (c - e) / SW;
#pragma unroll
                                    for (int j = 0; j < 8; ++j) T_[o][j] = fmaf(xr[c][j], wv[e][j], T_[o][j]);
                                }
                    }
                }
            }
            // output frame ti-1 has now seen frames ti-2, ti-1, ti
            const int to = ti - 1;
            if (to >= t0 && to < t1) {
                float sc[8], bs[8];
                {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(wl + (27 * 2 + 0) * DWM_CGS * 4);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(wl + (27 * 2 + 1) * DWM_CGS * 4);
                    const f32x4 e = *reinterpret_cast<const f32x4*>(wl + (28 * 2 + 0) * DWM_CGS * 4);
                    const f32x4 f = *reinterpret_cast<const f32x4*>(wl + (28 * 2 + 1) * DWM_CGS * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sc[j] = a[j];
                        sc[4 + j] = b[j];
                        bs[j] = e[j];
                        bs[4 + j] = f[j];
                    }
                }
                __bf16* yrow = y + ((((long)n * d.To + to) * d.Ho + ho) * d.Wo) * Cp + cg * 8;
#pragma unroll
                for (int o = 0; o < WT; ++o) {
                    const int wo = wo0 + o;
                    if (wo >= d.Wo) continue;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = P[o][j] * sc[j] + bs[j];
                        psum[j] += v[j];
                    }
                    if (RED) {
                        // this launch is a dgrad: v is the gradient w.r.t. the producer unit's activation output at this position.  Its
                        // backward sums (sum d', sum d' yhat with d' = v act'(y sc + sh), yhat = (y - mean) invstd) are taken here, from
                        // the fp32 v, instead of by a separate pass over (d, y)
                        float yv[8];
                        load8(reinterpret_cast<const __bf16*>(rd.y) + (yrow - y) + (long)wo * Cp, yv);
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {  // 4 channels at a time: 16 table registers live instead of 32
                            const f32x4 mean = *reinterpret_cast<const f32x4*>(wl + ((29 + 0) * 2 + hf) * DWM_CGS * 4);
                            const f32x4 istd = *reinterpret_cast<const f32x4*>(wl + ((29 + 1) * 2 + hf) * DWM_CGS * 4);
                            const f32x4 rsc = *reinterpret_cast<const f32x4*>(wl + ((29 + 2) * 2 + hf) * DWM_CGS * 4);
                            const f32x4 rsh = *reinterpret_cast<const f32x4*>(wl + ((29 + 3) * 2 + hf) * DWM_CGS * 4);
#pragma unroll
                            for (int j4 = 0; j4 < 4; ++j4) {
                                const int j = hf * 4 + j4;
                                const float dp = v[j] * act_grad(fmaf(yv[j], rsc[j4], rsh[j4]), rd.act);
                                psum[j] += dp - v[j];  // psum already holds v (the SE-pool convention above); the sum wanted here is of d'
                                psq[j] = fmaf(dp, (yv[j] - mean[j4]) * istd[j4], psq[j]);
                            }
                        }
                    } else if (stats) {  // block-uniform
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float dv = v[j] - kshift[j];
                            psq[j] = fmaf(dv, dv, psq[j]);
                        }
                        ++kcount;
                    }
                    act_vec(v, d.act);
                    if (d.Cout - cg * 8 < 8) mask_tail(v, d.Cout - cg * 8);  // only the last channel group has padding
                    store8(yrow + (long)wo * Cp, v);
                }
            }
#pragma unroll
            for (int o = 0; o < WT; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) P[o][j] = 0.0f;  // becomes the set of output frame ti+2
        };

        issue(t0 - 1, 0);
#pragma unroll 1
        for (int ti = t0 - 1; ti <= t1; ti += 3) {
            frame(ti, A0, A1, A2);
            if (ti + 1 <= t1) frame(ti + 1, A1, A2, A0);
            if (ti + 2 <= t1) frame(ti + 2, A2, A0, A1);
        }
    }
    if (pool && stats) {
        // training path: this block's row of the batch-statistics partials, ws[n][bx][2][Cp] = (sum (y - k), sum (y - k)^2) of the fp32
        // outputs (the layout bn_finalize_kernel reads, with the same shift k), reduced over the R items in fixed order
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[r * Cp + cg * 8 + j] = pass ? psq[j] : (RED ? psum[j] : psum[j] - (float)kcount * kshift[j]);
            __syncthreads();
            for (int ch = threadIdx.x; ch < Cp; ch += blockDim.x) {
                float s = 0.0f;
                for (int q = 0; q < R; ++q) s += red[q * Cp + ch];
                pool[(((long)n * bpc + bx) * 2 + pass) * Cp + ch] = s;
            }
        }
    } else if (pool) {  // block-uniform; squeeze-excite partial sums reduced over the R items in fixed order
#pragma unroll
        for (int j = 0; j < 8; ++j) red[r * Cp + cg * 8 + j] = psum[j];
        __syncthreads();
        for (int ch = threadIdx.x; ch < Cp; ch += blockDim.x) {
            float s = 0.0f;
            for (int q = 0; q < R; ++q) s += red[q * Cp + ch];
            // write-through (sc1) store: with the fused gate below another workgroup reads this row inside the launch
            __hip_atomic_store(pool + ((long)n * bpc + bx) * Cp + ch, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (se.gate) {
            // ---- fused squeeze-excite gate: the LAST block of a clip to arrive computes the clip's gate (15 stand-alone 9-13 us
            // launches per X3D-S forward otherwise).  Placement-independent hand-off (cdna_hip_programming.md Guideline 16, counter
            // form): partial rows stored write-through, every storing wave drains its stores, the block's barrier, ONE lane adds to
            // the clip's agent-scope counter; the block that draws bpc - 1 acquires and reads all rows with agent-scope loads.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(red);  // the partial-sum scratch is free again
            if (threadIdx.x == 0) {
                const int old = __hip_atomic_fetch_add(se.counter + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == bpc - 1;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(se.counter + n, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                }
                *flag = last;
            }
            __syncthreads();
            const int last = *flag;
            __syncthreads();  // everyone has read the flag before the scratch is reused
            if (last) {       // block-uniform
                float* mean = red;
                float* hid = red + Cp;
                const int C = d.Cout, Cse = se.cse;
                const float inv_positions = 1.0f / (float)(d.To * d.Ho * d.Wo);
                for (int ch = threadIdx.x; ch < Cp; ch += blockDim.x) {  // fixed order over the clip's partial rows: bitwise reproducible
                    float s = 0.0f;
                    const float* pp = pool + (long)n * bpc * Cp + ch;
                    for (int q0 = 0; q0 < bpc; q0 += 8) {
                        float t[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            t[u] = __hip_atomic_load(pp + (long)min(q0 + u, bpc - 1) * Cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int u = 0; u < 8; ++u) s += (q0 + u < bpc) ? t[u] : 0.0f;
                    }
                    mean[ch] = s * inv_positions;
                }
                __syncthreads();
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;  // full waves only
                if (wave < nw) {
                    for (int j = wave; j < Cse; j += nw) {
                        float t = 0.0f;
                        for (int ch = lane; ch < C; ch += 64) t = fmaf(se.w1[(long)j * C + ch], mean[ch], t);
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
                        if (lane == 0) hid[j] = fmaxf(t + se.b1[j], 0.0f);
                    }
                }
                __syncthreads();
                for (int ch = threadIdx.x; ch < Cp; ch += blockDim.x) {
                    float g = 0.0f;
                    if (ch < C) {
                        float sacc = se.b2[ch];
                        for (int j = 0; j < Cse; ++j) sacc = fmaf(se.w2[(long)ch * Cse + j], hid[j], sacc);
                        g = sigmoidf_(sacc);
                    }
                    se.gate[(long)n * Cp + ch] = g;
                }
            }
        }
    }
}

// Geometry: WT = 0 means "not this kernel".
DwMarchGeom dw_march_geom(const pasn_conv_desc& d, int dtype) {
    DwMarchGeom g = {0, 0, 0, 0, 0, 0};
    if (dtype != PASN_BF16) return g;
    if (const char* e = tune("PASN_NO_DWMARCH"))
        if (e[0] == '1') return g;
    const bool shape = d.kt == 3 && d.kh == 3 && d.kw == 3 && d.st == 1 && d.sh == d.sw && (d.sw == 1 || d.sw == 2) && d.pt == 1 &&
                       d.ph == 1 && d.pw == 1 && d.To == d.Ti && d.Cin_p == d.Cout_p && d.Cout_p % 8 == 0 && d.Cout_p / 8 <= DWM_CGS;
    if (!shape) return g;
    if (d.Ho != (d.Hi + 2 - 3) / d.sh + 1 || d.Wo != (d.Wi + 2 - 3) / d.sw + 1) return g;
    // 32-bit byte offsets inside the kernel
    if ((long)d.N * d.Ti * d.Hi * d.Wi * d.Cin_p * 2 >= (1L << 32)) return g;
    g.CG = d.Cout_p / 8;
    g.R = 256 / g.CG;
    // Pick (WT, Tc) with a small cost model.  One work item per thread makes the grid a fixed number of blocks; at 2
    // resident blocks per CU the kernel runs in ceil(blocks / 512) "rounds", and a half-empty last round is pure loss
    // (measured: 704 blocks = 2 rounds for 1.4 rounds of work).  Per-thread cost = frames x instructions per frame.
    const int force_wt = tune("PASN_DWM_WT") ? atoi(tune("PASN_DWM_WT")) : 0;
    const int force_tc = tune("PASN_DWM_TC") ? atoi(tune("PASN_DWM_TC")) : 0;
    double best = 1e30;
    for (int wt = 2; wt <= 3; ++wt) {
        if (force_wt && wt != force_wt) continue;
        if (wt == 3 && d.sw == 2) continue;  // 7 input columns: the stride-2 WT = 3 instance spills 30 registers
        const int nc = (wt - 1) * d.sw + 3;
        const double per_frame = 27.0 * wt * 4 + 3.0 * nc * 8 + 54 + 3.0 * nc * 2 + 30.0 * wt + 60;
        for (int tc = force_tc ? (force_tc < d.To ? force_tc : d.To) : d.To; tc >= 1; tc = (tc + 1) / 2) {
            const int nT = ceil_div(d.To, tc), strips = ceil_div(d.Wo, wt);
            const long blocks = (long)d.N * ceil_div((long)nT * d.Ho * strips, g.R);
            const double rounds = (double)ceil_div(blocks, 512);
            const double frames = tc + (nT > 1 ? 2.0 * (nT - 1) / nT : 0.0);
            const double t = rounds * frames * per_frame;
            if (t < best) {
                best = t;
                g.WT = wt;
                g.Tc = tc;
            }
            if (force_tc || tc <= 4) break;  // forced: that value only; otherwise To, To/2, ... down to ~4
        }
    }
    if (g.WT == 0) return DwMarchGeom{0, 0, 0, 0, 0, 0};
    g.strips = ceil_div(d.Wo, g.WT);
    g.bpc = ceil_div((long)ceil_div(d.To, g.Tc) * d.Ho * g.strips, g.R);
    if (dwm_lds_bytes(g.R, d.Cout_p) > 72 * 1024) return DwMarchGeom{0, 0, 0, 0, 0, 0};
    return g;
}

// the RED instances stage four more table rows: still two blocks per CU?
bool dw_march_red_ok(const pasn_conv_desc& d, const DwMarchGeom& g) {
    return g.WT != 0 && d.sw == 1 && d.sh == 1 && dwm_lds_bytes(g.R, d.Cout_p, true) <= 80 * 1024;
}

int launch_dw_march(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool,
                    const pasn_conv_desc& d, const DwMarchGeom& g, hipStream_t s, const DwSeArgs& se, int stats, const DwRedArgs* red,
                    const float* shift) {
    const dim3 grid(g.bpc * d.N), block(g.CG * g.R);
    const size_t lds = dwm_lds_bytes(g.R, d.Cout_p, red != nullptr);
    const DwRedArgs rd = red ? *red : DwRedArgs{nullptr, nullptr, 0};
#define PASN_DWM(SW_, WT_, RED_)                                                                                             \
    hipLaunchKernelGGL((dwconv3d_march_kernel<SW_, WT_, RED_>), grid, block, lds, s, (const __bf16*)x, w, scale, bias, \
                       (__bf16*)y, pool, d, g.CG, g.R, g.strips, g.Tc, g.bpc, se, stats, rd, shift)
    if (red) {  // dgrad of a stride-1 "same" conv + the producer unit's backward sums (stats layout of `pool`)
        if (!dw_march_red_ok(d, g) || !pool || !stats) {
            set_error("launch_dw_march: the fused backward sums need a stride-1 launch with a statistics buffer");
            return PASN_ERR_ARG;
        }
        if (g.WT == 3) PASN_DWM(1, 3, true);
        else PASN_DWM(1, 2, true);
    } else if (d.sw == 1 && g.WT == 3) PASN_DWM(1, 3, false);
    else if (d.sw == 1) PASN_DWM(1, 2, false);
    else if (g.WT == 3) PASN_DWM(2, 3, false);
    else PASN_DWM(2, 2, false);
#undef PASN_DWM
    return check_launch("dwconv3d_march_kernel");
}

}  // namespace pasn
