// The registry of tuning switches and the snapshot of the environment they are read from (see tuning.h).
#include "tuning.h"

#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>

#include "common.h"

extern char** environ;

namespace pasn {

struct TuneEntry {
    const char* name;
    const char* cls;  // "route" | "geom" | "dev"
    const char* doc;
};

// clang-format off
static const TuneEntry kRegistry[] = {
    // ---- route: documented switches of the product build ---------------------------------------------------------------------------
    {"PASN_DWMFMA",           "route", "0: depthwise 3x3x3 stride-1 stencil on the VALU kernels instead of the matrix-core kernel (dwmfma.hip)"},
    {"PASN_DWMFMA_MAXW",      "route", "matrix-core stencil only on planes at most this wide"},
    {"PASN_DW_STATS_MFMA",    "route", "0: training forward takes the stencil's batch statistics on the VALU kernel"},
    {"PASN_DW_DGRAD_REDUCE",  "route", "0: depthwise dgrad without the fused backward sums of the producer unit"},
    {"PASN_DWWG_FUSED",       "route", "1: depthwise weight gradient with the three kt taps in one launch"},
    {"PASN_EXPDW",            "route", "0: no fused expand-conv + stencil launch (x3d_expdw.hip)"},
    {"PASN_DW_TEMPORAL",      "route", "0: depthwise (kt,1,1) convs on the generic strip kernel instead of the T-marching kernel (dwtemporal.hip)"},
    {"PASN_DW_TZ",            "route", "0: the stride-1 stencils of planes 9 .. 14 wide on dwconv3d_mfma_kernel instead of the Toeplitz kernel (dw_tz.hip); 1: the Toeplitz kernel also where its blocks do not fit one round (more than 512)"},
    {"PASN_EXPDW_TZ",         "route", "0: stride-1 fused expand + stencil launches on the block-diagonal kernel (x3d_expdw.hip) instead of the Toeplitz kernel (x3d_expdw_tz.hip)"},
    {"PASN_EXPDW_S1",         "route", "stride-1 use of the fused expand + stencil launch: 0 off, 1 the SE blocks of wide planes only"},
    {"PASN_EXPDW_FUSE",       "route", "0: fused expand + stencil step as two scheduling regions (expand, then stencil)"},
    {"PASN_EXPDW_FOLD",       "route", "0: norm_a's scale applied in the fused launch's epilogue instead of folded into the expand weights (host side, plan.py)"},
    {"PASN_HEAD_CHAIN",       "route", "0: head B as seven launches instead of the chained launch (head_chain.hip)"},
    {"PASN_NO_DGRAD_S2",      "route", "stride-2 depthwise dgrad through the generic kernel"},
    {"PASN_NO_DWMARCH",       "route", "no T-marching VALU stencil (dwmarch.hip)"},
    {"PASN_NO_DWWG_MARCH",    "route", "depthwise weight gradient without the T-marching kernel"},
    {"PASN_NO_DWWG_STRIP",    "route", "depthwise weight gradient without the strip kernel"},
    {"PASN_NO_DW_STATS",      "route", "training forward: stencil and batch statistics as two calls"},
    {"PASN_NO_FC_MFMA",       "route", "7x7 stems on the VALU kernel instead of first_conv_mfma.hip"},
    {"PASN_NO_FIRST_IM2COL",  "route", "first-layer weight gradient without the im2col path"},
    {"PASN_NO_GEMM",          "route", "no LDS-tiled GEMM for pointwise convs (gemm_pw.hip)"},
    {"PASN_NO_GEMM_BN64",     "route", "no 64-column instance of the LDS-tiled GEMM"},
    {"PASN_TCONV",            "route", "0: temporal (3,1,1) convs on the implicit-GEMM kernels instead of the weight-stationary T-marching kernel (tconv_ws.hip)"},
    {"PASN_NO_HALO",          "route", "no halo-tile implicit GEMM (igemm_halo.hip)"},
    {"PASN_NO_IGEMM",         "route", "no direct-to-LDS implicit GEMM (igemm.hip)"},
    {"PASN_NO_PACK",          "route", "1: training weights packed by torch ops instead of the one-launch pack kernel (host side, train.py)"},
    {"PASN_WGT_WIDE",         "geom",  "0: the wide pointwise weight gradient with one 32 x 32 tile per wave everywhere (no 1 x 2 / 2 x 1 tiles per wave along the wide side)"},
    {"PASN_WGT_XCD",          "geom",  "0: the wide pointwise weight gradient's tile groups as blockIdx.y (a sweep apart) instead of back to back on one XCD"},
    {"PASN_DWWG_MARCH2",      "route", "0: depthwise weight gradient by round 2's marching kernel instead of round 4's (wgrad.hip)"},
    {"PASN_DWWG_CH",          "geom",  "channels per thread of the depthwise weight-gradient march (4 default, 2)"},
    {"PASN_DWWG_WT",          "geom",  "outputs per strip of the depthwise weight-gradient march at stride 1 (2 default, 3)"},
    {"PASN_DWWG_BLOCKS",      "geom",  "cap on the blocks (= partial rows) of the depthwise weight-gradient march (default 512)"},
    {"PASN_TRAIN_BLOCKS",     "geom",  "blocks an elementwise training pass is cut into (default 1024; the partial-sum workspaces scale with it)"},
    {"PASN_TRAIN_ROWS_CONTIG", "geom", "1: the elementwise training passes map Cp / 8 lanes to a row (contiguous spans, fewer idle lanes) instead of the next power of two"},
    {"PASN_NO_TRAIN_PAIR",    "route", "1: the loss recipe's two trunk passes of a training step as two passes instead of one paired pass with two statistics groups (host side, trainer.py)"},
    {"PASN_TRAIN_STREAMS",    "route", "1: weight-gradient launches of the training step on the main stream instead of a second one (host side, train.py)"},
    {"PASN_TRAIN_SIDE_DEPTH", "geom",  "side-stream weight-gradient launches outstanding before the main stream waits (default 2)"},
    {"PASN_NO_EDP",           "route", "1: no whole-block launch for the 7 x 7 stage (x3d_edp.hip)"},
    {"PASN_NO_PE",            "route", "1: no streamed project + expand pair launch for the 432-channel stage (x3d_pe.hip)"},
    {"PASN_NO_PWCONV",        "route", "no register-resident persistent pointwise conv (pwconv.hip)"},
    {"PASN_NO_PWTINY",        "route", "no one-wave-per-tile pointwise conv for small maps (pwconv_tiny.hip)"},
    {"PASN_NO_SE_ANALYTIC",   "route", "training: squeeze-excite backward without the analytic pooled-gradient path (host side, train.py)"},
    {"PASN_NO_SE_FUSE",       "route", "squeeze-excite gate never inside the stencil launch"},
    {"PASN_NO_SE_PROLOGUE",   "route", "squeeze-excite gate never in the project conv's prologue"},
    {"PASN_NO_SHORTFUSE",     "route", "strided shortcut conv as its own launch"},
    {"PASN_NO_STEM",          "route", "X3D stem as two launches"},
    {"PASN_NO_STEM_MFMA",     "route", "X3D stem on the VALU kernel instead of stem_mfma.hip"},
    {"PASN_NO_TAIL_MFMA",     "route", "training head tail without the matrix-core pooling"},
    {"PASN_NO_WGRAD_GATHER",  "route", "windowed weight gradient without the gather kernel"},
    {"PASN_NO_WGRAD_HALO",    "route", "windowed weight gradient without the halo kernel (wgrad_halo.hip)"},
    {"PASN_NO_WGRAD_LDS",     "route", "pointwise weight gradient without the LDS kernel"},
    {"PASN_NO_WGRAD_TILE",    "route", "windowed weight gradient without the tiled kernel"},
    {"PASN_NO_XPAIR",         "route", "project conv never chained with the next expand conv (pwconv_xpair.hip)"},
    {"PASN_NO_XTILE",         "route", "no X-stationary pointwise conv (pwconv_xtile.hip)"},
    {"PASN_POOL_VALU",        "route", "1: head-B pooling on the VALU kernel"},
    {"PASN_SE_FUSE_MAXC",     "route", "largest channel count whose squeeze-excite gate rides in the stencil launch (default 128)"},
    {"PASN_WGRAD_DET",        "route", "1: windowed weight gradient through fixed-order partial buffers only"},
    {"PASN_WS",               "route", "0: no weight-stationary pointwise conv (pwconv_ws.hip); 'all': every covered layer"},
    {"PASN_WSPAIR",           "route", "0: no chained pairs in the weight-stationary kernel"},
    {"PASN_WS_GATED",         "route", "0: gated (SE / Swish input) layers not on the weight-stationary kernel"},
    {"PASN_XPAIR_ALL",        "route", "1: chain every covered project / expand pair, also the narrow stage-3 ones"},
    {"PASN_XTILE_GATED",      "route", "0: gated layers not on the X-stationary kernel"},
    // ---- geom: geometry overrides the parity tests sweep (product build) -----------------------------------------------------------
    {"PASN_DWMFMA_TC",        "geom",  "matrix-core stencil: forced T chunk"},
    {"PASN_DWMFMA_UPB",       "geom",  "matrix-core stencil: forced units per block"},
    {"PASN_DWM_TC",           "geom",  "VALU marching stencil: forced T chunk"},
    {"PASN_DWM_WT",           "geom",  "VALU marching stencil: forced outputs per strip"},
    {"PASN_EXPDW_TC",         "geom",  "fused expand + stencil: forced T chunk"},
    {"PASN_EXPDW_UPB",        "geom",  "fused expand + stencil: forced units per block"},
    {"PASN_WS_BPC",           "geom",  "weight-stationary conv: forced blocks per channel group"},
    {"PASN_WS_MT",            "geom",  "weight-stationary conv: forced 32-row sub-tiles per wave"},
    {"PASN_WS_NS",            "geom",  "weight-stationary conv: forced stage count"},
    {"PASN_WS_PT",            "geom",  "weight-stationary conv: forced waves along positions"},
    {"PASN_WS_MINK",          "geom",  "weight-stationary conv: smallest padded K it takes (default 48)"},
    {"PASN_PE_MT",            "geom",  "streamed project + expand pair: 32-row tiles per unit (2 or 4, default 4)"},
    // ---- dev: only with -DPASN_TUNING (timing ablations give WRONG results) ----------------------------------------------------------
    {"PASN_DWMFMA_ABL",       "dev",   "matrix-core stencil timing ablations (bit mask)"},
    {"PASN_EXPDW_ABL",        "dev",   "fused expand + stencil timing ablations (bit mask)"},
    {"PASN_HALO_ABL",         "dev",   "halo implicit GEMM timing ablations"},
    {"PASN_WS_ABL",           "dev",   "weight-stationary conv timing ablations (needs -DPASN_WS_ABLATE too)"},
    {"PASN_EDP_STAMPS",       "dev",   "whole-block launch of the 7 x 7 stage: in-kernel phase stamps (tools/edp_bench.py)"},
    {"PASN_TZ_STAMPS",        "dev",   "Toeplitz expand + stencil launch: in-kernel phase stamps (tools/tz_bench.py)"},
    {"PASN_PE_STAMPS",        "dev",   "streamed project + expand pair: in-kernel phase stamps (tools/pe_bench.py)"},
    {"PASN_DW_WT",            "dev",   "strip stencil: outputs per thread (4, 7, 8)"},
    {"PASN_HALO_SP",          "dev",   "halo implicit GEMM: slice pipeline on / off"},
    {"PASN_IGEMM_NT",         "dev",   "implicit GEMM: forced channel tiles per block"},
    {"PASN_IGEMM_RR",         "dev",   "implicit GEMM: round-robin tile order"},
    {"PASN_WGT_BLOCKS",       "dev",   "tiled weight gradient: target block count"},
    {"PASN_WG_TPW",           "dev",   "tiled weight gradient: tiles per wave cap"},
    {"PASN_WS_HELP",          "dev",   "weight-stationary conv: 0 = no helper waves for the input transform"},
    {"PASN_WS_ROWS",          "dev",   "weight-stationary conv: minimum rows per block"},
    {"PASN_XT_MINK",          "dev",   "X-stationary conv: smallest padded K it takes"},
};
// clang-format on

struct Snapshot {
    std::map<std::string, std::string> set;   // registered names that are set
    std::string unknown;                      // PASN_* names in the environment that the library does not know (space separated)
};

static std::mutex g_mu;
static Snapshot* g_snap = nullptr;  // replaced, never freed while readers may hold pointers into it (reload is a test / tool operation)

static bool host_side_name(const char* n) {  // read by the Python package / the tools, not by this library
    static const char* const kHost[] = {"PASN_LIB_PATH", "PASN_EXTRA_HIPCC_FLAGS", "PASN_NATIVE_RCCL", "PASN_BENCH_BACKEND", "PASN_PARITY_LOG", "PASN_TB_TORCHPROF", "PASN_KB_ACT", "PASN_HC_ABL"};
    for (const char* h : kHost)
        if (!strcmp(h, n)) return true;
    return false;
}

static Snapshot* take_snapshot() {
    Snapshot* s = new Snapshot;
    for (char** e = environ; e && *e; ++e) {
        if (strncmp(*e, "PASN_", 5)) continue;
        const char* eq = strchr(*e, '=');
        if (!eq) continue;
        const std::string name(*e, eq - *e);
        bool known = false;
        for (const TuneEntry& t : kRegistry)
            if (name == t.name) {
                known = true;
#ifndef PASN_TUNING
                if (!strcmp(t.cls, "dev")) break;  // compiled out: never enters the snapshot
#endif
                s->set[name] = eq + 1;
                break;
            }
        if (!known && !host_side_name(name.c_str())) s->unknown += (s->unknown.empty() ? "" : " ") + name;
    }
    return s;
}

static Snapshot* snapshot() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_snap) g_snap = take_snapshot();
    return g_snap;
}

const char* tune(const char* name) {
    Snapshot* s = snapshot();
    if (s->set.empty()) return nullptr;
    const auto it = s->set.find(name);
    return it == s->set.end() ? nullptr : it->second.c_str();
}

}  // namespace pasn

using namespace pasn;

extern "C" void pasn_tuning_reload(void) {
    Snapshot* s = take_snapshot();
    std::lock_guard<std::mutex> lk(g_mu);
    g_snap = s;  // (the old snapshot is leaked on purpose: a concurrent reader may still hold a value pointer; reload is rare)
}

extern "C" const char* pasn_tuning_get(const char* name) { return name ? tune(name) : nullptr; }

// "NAME=VALUE" lines of the switches in force, then one "unknown: ..." line if the environment holds PASN_* names the library ignores,
// then -- with `with_registry` -- one "name<TAB>class<TAB>meaning" line per registry entry.  Returns the length needed (excluding the NUL).
extern "C" int pasn_tuning_report(char* buf, int cap, int with_registry) {
    Snapshot* s = snapshot();
    std::string out;
    for (const auto& kv : s->set) out += kv.first + "=" + kv.second + "\n";
    if (!s->unknown.empty()) out += "unknown: " + s->unknown + "\n";
    if (with_registry)
        for (const TuneEntry& t : kRegistry) out += std::string(t.name) + "\t" + t.cls + "\t" + t.doc + "\n";
    if (buf && cap > 0) {
        const int n = (int)out.size() < cap - 1 ? (int)out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int)out.size();
}
