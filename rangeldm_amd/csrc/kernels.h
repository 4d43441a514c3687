// Kernel parameter blocks and launchers of librangeldm_hip (gfx950).  Activations: bf16 channels-last [B][W][H][C].
#pragma once
#include "common.h"

namespace rldm {

// ---------------------------------------------------------------------------------------------------------------
// Fused circular implicit-GEMM convolution (conv_igemm.hip)
//   y = conv_k( pad( silu?( GN?( cat[x0, x1] ) ) ) ) + bias (+ temb[row]) (+ res)
// geometry: virtual input v(w', h') = x(w'/up, h'/up) of size (Win*up, Hin*up); output pixel (w, h) reads
//   v(wrap(w*stride + i - pad_lo), h*stride + j - pad_lo) for taps i (azimuth), j (beams); h out of range -> 0.
// Reference arithmetic: ldm/utils.py:40-55,107-116; vae/sgm/modules/diffusionmodules/model.py:93-125,164-172.
// ---------------------------------------------------------------------------------------------------------------
// A normalised (+ activated) copy of a conv's output, written by the PRODUCER's epilogue for one consuming GroupNorm
// (conv_small.hip, tiles that own a whole image: the statistics of its 32 channels are complete inside the workgroup).
// y = silu?( gamma * (x - mean_g) * rstd_g + beta ) on the bf16-rounded output, stored at channel offset of the consumer's
// (possibly concatenated) input tensor.
struct NormView {
    bf16_t* y;              // consumer's pre-activated input [B][W][H][ld], already offset to this producer's first channel
    const float* gamma;     // consumer's GroupNorm affine, offset the same way
    const float* beta;
    int ld;
    int cpg_shift;          // log2(channels per group) of the CONSUMER's norm (<= 5: a 32-channel tile holds whole groups)
    float inv_n;            // 1 / (pixels per image * channels per group)
    float eps;
    int silu;
};

// The scheduler step fused into conv_out's fp32 NCHW epilogue (conv_igemm.hip): the thread that produces eps[i] also holds the
// index of x[i], so x_prev[i] = step(x[i], eps[i], noise[i]) costs one more load and store instead of a launch of its own
// (DDIMScheduler.step / DDPMScheduler.step; same arithmetic as sched_step_kernel).
// One scheduler step without its noise term.  mode = (ddpm ? 1 : 0) | prediction << 1, prediction 0 epsilon | 1 v_prediction | 2 sample
// (diffusers DDIMScheduler.step / DDPMScheduler.step [3P]; the `prediction_type` branches ahead of the update).
// c0 = sqrt(alpha_prod_t), c1 = sqrt(beta_prod_t); DDIM: c2 = sqrt(alpha_prod_prev), c3 = direction coefficient (multiplies the
// predicted epsilon); DDPM: c2 / c3 = the posterior mean's coefficients of x0 / x_t.  The epsilon branch is the expression the
// kernels have always used (bit-identical).
__device__ __forceinline__ float sched_prev(const int mode, const float c0, const float c1, const float c2, const float c3, const float x,
                                            const float e) {
    const int pred = mode >> 1;
    float x0, pe;
    if (pred == 0) {
        x0 = (x - c1 * e) / c0;
        pe = e;
    } else if (pred == 1) {
        x0 = c0 * x - c1 * e;
        pe = c0 * e + c1 * x;
    } else {
        x0 = e;
        pe = (x - c0 * e) / c1;
    }
    return (mode & 1) == 0 ? c2 * x0 + c3 * pe : c2 * x0 + c3 * x;
}

struct SchedFuse {
    const float* coef_table;  // device [steps][5]; null: not fused
    const int* step_ptr;
    const float* x;
    const float* noise;       // [steps][noise_step_stride] or null
    long long noise_step_stride;
    float* x_prev;
    int mode;                 // 0 ddim, 1 ddpm; | prediction type << 1 (sched_prev)
    // the NEXT step's conv_in input (pack_input's bf16 [B][W][H][pack_ld] tensor): the thread that holds x_prev[i] stores its bf16 image
    // too, so a step's first launch is conv_in instead of a pack launch (the pos-encoding / condition channels never change); or null
    bf16_t* pack;
    int pack_ld;
};

struct ConvParams {
    // main phase: TAPS taps per CK-channel chunk over cat[x0, x1], GroupNorm (+SiLU) applied on the way into LDS
    const bf16_t* x0;
    const bf16_t* x1;       // second tensor of a channel concat (may be null)
    int C0, C1;             // channels of x0 / x1; C0 and C0 + C1 are multiples of the kernel's CK
    // residual phase: one centre tap per chunk over cat[r0, r1] (raw, at OUTPUT resolution) with the 1x1 shortcut
    // weights (or an identity) appended to the weight stream; R0 + R1 == 0: none
    const bf16_t* r0;
    const bf16_t* r1;
    int R0, R1;
    int B, Win, Hin;
    int up;                 // 1 | 2 (nearest upsample folded into indexing)
    int stride;             // 1 | 2
    int pad_lo;             // 1: symmetric pad 1, 0: end-only pad (or 1x1)
    int Wout, Hout;
    int TW, TH;             // output-pixel tile handled by one block (TW*TH <= BM)
    int colb;               // halo column pitch in bytes (conv_halo_col_bytes)
    int tiles_h, tiles_img; // Hout / TH, (Wout / TW) * tiles_h
    int th_shift;           // log2(TH) (TH is a power of two)
    int magic_thv;          // ceil(2^20 / halo rows): slot / THv == (slot * magic) >> 20
    int magic_cpg;          // ceil(2^20 / channels per group)
    // GroupNorm prologue (null st0 -> none): per-channel partial (sum, sumsq) [B][P][C] written by the producers of x0/x1
    const float2* st0;
    const float2* st1;
    int P0, P1;
    int gn_groups;
    const float* gn_gamma;
    const float* gn_beta;
    float gn_eps;
    int silu;
    // weights: bf16, packed [ntile_n][Cin/CK * taps + (R0+R1)/CK][BN][CK + 8 pad]
    const bf16_t* wpk;
    int N;                  // real output channels
    int ntile_n;
    const float* bias;      // padded to ntile_n * BN
    // accumulator init: bias + temb row
    const float* temb;      // null or [rows][temb_ld], row = step*rows_per_step + (per_sample ? b : 0)
    int temb_ld;
    const int* step_ptr;    // device int (null -> 0)
    int temb_rows_per_step;
    int temb_per_sample;
    // outputs
    bf16_t* y;              // bf16 channels-last [B][Wout][Hout][y_ld]
    int y_ld;
    float2* y_stats;        // null or [B][tiles per image][N] partial (sum, sumsq) of the stored bf16 values
    float* y_nchw;          // if set: fp32 NCHW output [B][N][Wout][Hout] instead of y
    // split-K over channel chunks
    int ksplit;             // >= 1
    float* slab;            // [tiles][ksplit][BM*BN] fp32 (ksplit > 1)
    int* ticket;            // [tiles] zero-initialised arrival counters (re-armed by the last arriver)
    int lds_total;           // dynamic LDS bytes of the launch (set by launch_conv)
    float gn_inv_n;          // conv_small.hip: 1 / (pixels per image * channels per group)
    const bf16_t* res;       // conv_small.hip: identity residual [B][Wout][Hout][N] added in the epilogue (or null)
    unsigned long long* ts;  // tuning: s_memtime stamps of blocks 0..3, wave 0 ([4][64]) or null
    int dbg;                // tuning ablations (rldm_debug_set_flags): 1 skip stores, 2 skip main loop, 4 skip GN finalize
    SchedFuse sch;          // conv_igemm.hip, y_nchw outputs: the sampler's scheduler step in the epilogue
    int* step_inc;          // conv_igemm.hip: the sampler's device step index, advanced by the step's FIRST launch (conv_in, which does not read it) or null
    int nviews;             // conv_small.hip, image-owning tiles: normalised copies of the output for up to 3 consumers
    NormView nv[3];
    int st_inst;            // conv_stream.hip: 0 the 8-wave instance the tile implies; round 4's 4-wave workgroups on 16 x 8 tiles, two
                            // resident per CU: 1 = 128 pixels x 128 channels, 2 = 128 pixels x 64 channels x 2 k-groups
    int exp;                // ... round-4 experiment switches (rldm_debug_set_flags2 >> 8; conv_stream_body.h)
    int* cu_lock;           // ... [4096] zero-initialised per-CU locks (exp & 2)
};

struct ConvTile {
    int BM, BN, CK, taps;
};
// row stride (bytes) of one LDS / packed-weight row for a CK
inline int conv_row_bytes(int CK) { return CK * 2 + 16; }
size_t conv_lds_bytes(const ConvTile& t, const ConvParams& p);
bool conv_tile_supported(const ConvTile& t);
int conv_max_halo_slots(const ConvTile& t);
int conv_halo_col_bytes(const ConvTile& t, int TH, int stride);
int conv_tile_threads(const ConvTile& t);
int launch_conv(const ConvTile& t, const ConvParams& p, hipStream_t stream);

// Activation-stationary stride-1 variant for 64-pixel tiles (conv_small.hip): 3x3 over one pre-activated input tensor
// (C1 == 0, no GroupNorm prologue) or 1x1 with the GroupNorm affine folded in; an identity residual is added in the
// epilogue (ConvParams::res).  wpk is the fragment-ordered image
// [N/32][k-groups][9*Cin/16/KG + R/16/KG][64 lanes][8 bf16] (ConvLayer::get_fragpacked), colb from conv_small_col_bytes,
// BN in {32, 64}, KG = conv_small_kgroups(BN).
int conv_small_kgroups(int BN);
int conv_small_col_bytes(int Cin, int TH, int taps);
size_t conv_small_lds_bytes(const ConvParams& p, int taps, int BN);
bool conv_small_supported(const ConvParams& p, int taps, int BN);
int launch_conv_small(const ConvParams& p, int taps, int BN, hipStream_t stream);

// Weight-streaming 3x3 / stride 1 variant (conv_stream.hip): 256 pixels (32 x 8) x 128 channels per workgroup for the
// full-resolution levels, 128 pixels (16 x 8) x 64 channels with 4 k-groups for the 128x8 level; 64-channel chunks.  Same
// ConvParams (colb = conv_halo_col_bytes of a CK = 64 tile); wpk is the fragment-ordered image
// [N/32][KG][(Cin/64 * 9 + R/64) * 4/KG k-steps][64 lanes][8 bf16] (ConvLayer::get_streampacked).
int conv_stream_bn(const ConvParams& p);          // 128 (TW = 32) | 64 (TW = 16: the 4-k-group instance)
int conv_stream_kgroups(const ConvParams& p);
size_t conv_stream_lds_bytes(const ConvParams& p);
bool conv_stream_supported(const ConvParams& p, int taps);
int launch_conv_stream(const ConvParams& p, hipStream_t stream);

// conv_regw.hip (round 4): 64 -> 64 channel 3x3 / stride 1 convs over 16 x 8 tiles with the weights resident in registers and a persistent
// tile loop per workgroup (the VAE decoder's full-resolution level).  `p` as for conv_stream (st_inst 2's tile); the output statistics are
// ONE partial per workgroup: y_stats is [B][conv_regw_wg_per_image(p)][N].
size_t conv_regw_lds_bytes();
int conv_regw_wg_per_image(const ConvParams& p);
int conv_regw_partials(const ConvParams& p);
bool conv_regw_supported(const ConvParams& p);
int launch_conv_regw(const ConvParams& p, hipStream_t stream);
// conv_regw.hip, conv_c16_kernel (round 4): the network's input layer -- 3x3 / stride 1 over 16 (padded) input channels, 128 | N, 16 x 8 tiles;
// weights [N / 32][9 taps][64 lanes][8 bf16] (ConvLayer::get_c16packed); advances ConvParams::step_inc like the generic kernel
size_t conv_c16_lds_bytes();
bool conv_c16_supported(const ConvParams& p);
int launch_conv_c16(const ConvParams& p, hipStream_t stream);
// conv_regw.hip, conv_o4_kernel (round 4): the UNet's output layer -- GroupNorm + SiLU -> 3x3 over 128 channels -> <= 4 channels, fp32 NCHW +
// ConvParams::sch (the scheduler step); weights stream-packed with 2 k-groups (ConvLayer::get_streampacked(128, 2))
size_t conv_o4_lds_bytes();
bool conv_o4_supported(const ConvParams& p);
int launch_conv_o4(const ConvParams& p, hipStream_t stream);
// conv_regw.hip, conv_ds2_kernel (round 4): 3x3 / stride 2 / pad 1 over 256 channels, no norm, on 32 x 1 output tiles x 32 channels with the 8
// waves as k-groups (the UNet's 64x4 -> 32x2 down-sampler); weights stream-packed with 2 k-groups; one statistics partial per tile
size_t conv_ds2_lds_bytes();
bool conv_ds2_supported(const ConvParams& p);
int launch_conv_ds2(const ConvParams& p, hipStream_t stream);

// Persistent trunk launch (trunk.hip): consecutive conv_small launches whose tile owns a whole image (<= 64 pixels, 32-channel
// tiles) as the phases of ONE launch; the N / 32 workgroups of an image hand their outputs to each other through the L2 of the
// XCD they share.  `kind`: 0 = 3x3 over 256 channels, 1 = 3x3 over 512, 2 = 1x1 over 256 (conv_small instances <1,2,9,2>,
// <1,4,9,2>, <1,2,1,2>).
// A phase record is 64 dwords: ONE vector load per wave (lane l holds word l), requested a phase ahead, and v_readlane puts the
// fields into SGPRs -- the place the kernel-argument copy of a stand-alone launch lives in.
enum TrunkWord {
    TW_X0 = 0, TW_R0 = 2, TW_R1 = 4, TW_WPK = 6, TW_BIAS = 8, TW_Y = 10, TW_YSTATS = 12, TW_RES = 14,    // 64-bit pointers
    TW_R0C = 16, TW_R1C, TW_WIN, TW_HIN, TW_WOUT, TW_HOUT, TW_TW, TW_TH, TW_COLB, TW_THSHIFT, TW_N, TW_YLD, TW_NVIEWS,
    TW_KIND, TW_G, TW_NMINE, TW_TEMBOFF,
    TW_NV0 = 34,            // 2 views x 11 words: y (2), gamma (2), beta (2), ld, cpg_shift, inv_n, eps, silu
    TW_NVSTRIDE = 11,
    // phases of a MULTI-TILE cluster (kind >= 8: an image is several 64-pixel tiles x 64-channel tiles; no views) keep the
    // consumer-side GroupNorm of their input in the words the views would occupy
    TW_ST0 = 34, TW_GAMMA = 36, TW_BETA = 38,                   // 64-bit pointers
    TW_P0 = 40, TW_GROUPS, TW_MAGIC_CPG, TW_INVN, TW_EPS, TW_SILU, TW_TILES_H, TW_TILES_IMG,
    // conv_stream phases (kind 15: the full-resolution levels as clusters of 16 pixel tiles / 8 pixel tiles x 2 channel tiles)
    TW_X1 = 48, TW_ST1 = 50,                                    // 64-bit pointers
    TW_C0 = 52, TW_C1, TW_P1, TW_MAGIC_THV, TW_UP,
    TW_SUB,                 // conv_stream phases of variant 4: 1 = the sub-pixel form of nearest x2 + 3x3 (rank = input tile * 4 + parity, 128 output channels),
                            // 2 = tiles as tall as the image (st_inst 7)
    TW_WBYTES = 58,         // bytes of the phase's packed weights (TW_WPK ...): what the PREVIOUS phase touches, one dword per 128-byte line,
                            // so that they wait in the XCD's L2 (round 5: trunk_warm_next; 0: nothing to warm)
    TW_WORDS = 64
};
// phase kinds: 0..2 / 4..6 image-owning conv_small tiles (64 / 32 pixels), 3 attention over a pre-normalised x,
// 9..11 3x3 conv over 256 / 384 / 512 channels on 64-pixel x 64-channel tiles of a multi-tile image (8: 128 channels, not instantiated), 12 its 1x1 over 256,
// 13 attention with the GroupNorm fold inside (two query tiles per wave)
// 14 GroupNorm (+ SiLU) of a concatenated input as a phase of its own (norm.hip's gn_apply_kernel; record: x0 / x1 in TW_X0 / TW_R0,
// their channels in TW_R0C / TW_R1C, statistics in TW_ST0 / TW_RES with TW_P0 / TW_TILES_H partials, pixels per image in TW_WIN)
// (round 5) 16..18: the image-owning 64-pixel kinds 0..2 on 16-channel tiles (conv_small_body's H16 instances)
enum TrunkKind { TK_H16 = 16, TK_ATTN = 3, TK_CL_3x3_128 = 8, TK_CL_3x3_256, TK_CL_3x3_384, TK_CL_3x3_512, TK_CL_1x1_256, TK_ATTN_FOLD, TK_GN_APPLY, TK_STREAM };
struct TrunkPhase {
    unsigned w[TW_WORDS];
};
struct TrunkParams {
    const TrunkPhase* phases;   // device
    int nphases;
    int B, ranks;               // images; workgroups per image (its cluster): channel tiles x pixel tiles
    int ntile_n, nwn;           // channel tiles per image and 32-channel tiles per workgroup (N / 32 and 1 for image-owning tiles)
    int variant;                // kernel: 0 image-owning conv_small tiles, 1 multi-tile conv_small clusters, 2 conv_stream<256 px, 128 ch>,
                                // 3 conv_stream<128 px, 64 ch> (each set of instances has its own register allocation),
                                // 4 conv_stream<128 px, 128 ch> on 4 waves: 32 workgroups per image, two per CU (round 4)
    int skew;                   // variant 4: start delay of the second image group, x 1024 cycles
    unsigned* counters;         // device [B][32] zero-initialised: [0] arrivals (monotonic), [1] rank 0's XCC id + 1, [3] launches so far
    int* error;                 // device flag: 1 a bounded wait gave up, 2 a cluster is spread over several XCDs
    const float* temb;          // the plan's time-embedding table (PlanIO), set per launch
    const int* step_ptr;
    int temb_rows_per_step, temb_per_sample, temb_ld;
    unsigned long long* ts;     // ABLATE builds: [phase][16] s_memtime stamps of workgroup 0 (rldm_debug_timestamps buffer) or null
};
int launch_trunk(const TrunkParams& tp, size_t lds, hipStream_t stream);
int trunk_max_resident(int variant, size_t lds);   // workgroups of that kernel per CU by the runtime's occupancy query (< 0: query failed)

// ---------------------------------------------------------------------------------------------------------------
// Per-channel statistics (norm.hip) for tensors that did not come out of a conv epilogue (tests, external inputs):
// deterministic partial (sum, sumsq) per (b, pixel chunk p, channel).
// ---------------------------------------------------------------------------------------------------------------
struct GnStatsParams {
    const bf16_t* x;        // [B][npix][C]
    int C;
    int B, npix;            // pixels per image
    int P;                  // pixel chunks per image (grid.x)
    float2* part;           // [B][P][C] per-channel (sum, sumsq)
};
int launch_gn_stats(const GnStatsParams& p, hipStream_t stream);

// [B][P][C] partials -> [B][2][C]: the per-image double-precision total as float head + tail (see gn_fold_kernel).
struct GnFoldParams {
    const float2* part;     // [B][P][C]
    float2* out;            // [B][2][C]
    int B, P, C;
};
int launch_gn_fold(const GnFoldParams& p, hipStream_t stream);

// y = silu?( GroupNorm( cat[x0, x1] ) ) as one bf16 tensor [B][npix][C0 + C1], from the producers' per-channel partials.
// Used in front of conv_small.hip where every 32/64-channel tile of a conv would otherwise redo the whole activation.
struct GnApplyParams {
    const bf16_t* x0;
    const bf16_t* x1;
    int C0, C1;
    const float2* st0;
    const float2* st1;
    int P0, P1;
    int B, npix;
    int groups;
    const float* gamma;
    const float* beta;
    float eps;
    int silu;
    bf16_t* y;
    float inv_n;            // 1 / (npix * channels per group) (set by launch_gn_apply)
    // two outputs instead of one (ysplit > 0, a multiple of 8): channels [0, ysplit) -> y [B][npix][ysplit], the rest -> y1
    // [B][npix][Cin - ysplit] -- a concatenation wider than a conv kernel takes is convolved half by half (runtime.hip)
    bf16_t* y1;
    int ysplit;
};
int launch_gn_apply(const GnApplyParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// Multi-head self-attention with head_dim 8 (attention.hip).  qkv: [B][L][3C] (q | k | v, q pre-scaled by
// log2(e)/sqrt(8) through the packed weights), out: [B][L][C].
// ---------------------------------------------------------------------------------------------------------------
struct AttnParams {
    const bf16_t* qkv;
    bf16_t* out;
    int B, L, C;
};
int launch_attention(const AttnParams& p, hipStream_t stream);

// Fused GroupNorm -> q/k/v projection -> attention (attention.hip): x is the block input with its statistics partials;
// wfrag = per-head MFMA A fragments [heads][C/16][64 lanes][8 bf16] (rows 0-7 q pre-scaled by log2(e)/sqrt(8), 8-15 k,
// 16-23 v, 24-31 zero), bias [heads][32] fp32.  out: [B][L][C].
struct AttnQkvParams {
    const bf16_t* x;
    const float2* st;
    int P;
    const float* gamma;
    const float* beta;
    float eps;
    int groups;
    float inv_n;
    int magic_cpg;
    const bf16_t* wfrag;
    const float* bias;
    bf16_t* out;
    int B, L, C;
    unsigned long long* ts;     // ABLATE builds: phase stamps of workgroup 0 + [start, end] of every workgroup, launches with L == ts_L
    int ts_L;
    // fused output projection (round 3; null proj_w: off): y = to_out(attention) + bias + res and y's GroupNorm statistics, computed
    // by the SAME launch once all heads of the image have published their rows (a cluster seam like trunk.hip's: the image's
    // workgroups share an XCD).  proj_w: MFMA A fragments [C/32][C/16][64 lanes][8 bf16]; proj_stats: [B][L/64][C] partials.
    const bf16_t* proj_w;
    const float* proj_bias;
    const bf16_t* proj_res;
    bf16_t* proj_y;
    float2* proj_stats;
    unsigned* proj_counter;     // [B][32] arrival counters (monotonic; [3] = launches so far), zeroed at allocation
    int* proj_error;            // the plan's self-check word (1: a wait gave up, 2: an image's workgroups on several XCDs)
};
int launch_attention_qkv(const AttnQkvParams& p, hipStream_t stream);
// the second-generation launch's geometry for (B, L, C): heads per workgroup and waves; non-zero: it declines this shape
int attention_qkv2_geometry(int B, int L, int C, int* HG, int* waves);
// ... and whether that launch can carry the output projection (workgroups of an image = 64-pixel blocks, all co-resident)
bool attention_proj_fusable(int B, int L, int C, int cus);
int launch_attention_proj(const AttnQkvParams& p, hipStream_t stream);    // the same tail as a launch of its own (proj_counter unused)
// weight fragments per wave a phase of the persistent launch requests for the NEXT phase (trunk_seam.h; the plan builder's TW_G)
#ifndef RLDM_TRUNK_PREFETCH
#define RLDM_TRUNK_PREFETCH 18     /* (round 3: 12 -> 18 = the whole ring of a 3x3 / 256-channel phase: +0.5 %, trunk<0> 201 -> 225 VGPRs) */
#endif
// Timing-only ablations of the round-3 headroom study: `make TAG=x EXTRA=-DRLDM_EXP_NORES=1` etc. build a library
// whose RESULTS ARE WRONG and whose speed bounds what the removed piece can be worth (tools/ab_libs.sh, tools/fwd_time.py).  All 0 in
// every shipped build.  NORES: conv_stream skips its residual phase; NONORM: ... the GroupNorm + SiLU arithmetic of its staging (raw
// copy); NOSTATS: ... the statistics partial loads in front of its fold (mean 0 / variance 1 instead); NOWAIT: the persistent launches' cluster waits do not poll.
#ifndef RLDM_EXP_NORES
#define RLDM_EXP_NORES 0
#endif
#ifndef RLDM_EXP_NONORM
#define RLDM_EXP_NONORM 0
#endif
#ifndef RLDM_EXP_NOSTATS
#define RLDM_EXP_NOSTATS 0
#endif
#ifndef RLDM_EXP_NOWAIT
#define RLDM_EXP_NOWAIT 0
#endif
#ifndef RLDM_STREAM_PF2
#define RLDM_STREAM_PF2 0          /* conv_stream's 4-k-group instance: halo chunks requested two chunks ahead (round 4 experiment; measured
                                      SLOWER, 128x8 convs 13.4 -> 15.1 us: +16 VGPRs = 15 spills; the waits were not the loads') */
#endif
#ifndef RLDM_STREAM_PF2_MI2
#define RLDM_STREAM_PF2_MI2 1      /* ... the same for the 64-pixel instance (a chunk is 18 k-steps of 64 cycles; the second set costs 8 registers) */
#endif
#ifndef RLDM_STREAM_ILV
#define RLDM_STREAM_ILV 1          /* conv_stream K loop: pixel-fragment reads interleaved with the step's MFMAs (round 4; 0 = behind them) */
#endif
#ifndef RLDM_RES_DEPTH
#define RLDM_RES_DEPTH 1           /* residual chunks of a conv_stream tile in flight in registers (conv_stream_body.h); 2 and 3 measured the same */
#endif
// LDS of the second-generation fused attention body (attention_body.h) for HG heads per workgroup on `waves` waves: K rows, V^T,
// the GroupNorm affine + scratch, the heads' W' fragments and biases, and one 32-row x 144-byte x staging tile per wave
constexpr int kAttnXRowBytes = 128 + 16, kAttnXStageBytes = 32 * kAttnXRowBytes;
inline size_t attention_qkv2_lds_bytes(int L, int C, int HG, int waves) {
    const size_t Lp = (size_t)(L + 31) / 32 * 32;
    const size_t a = (size_t)2 * C * 8, b2 = (size_t)HG * (C / 16) * 64 * 4;
    return HG * Lp * 16 + (size_t)HG * 10 * (Lp + 8) * 2 + 16 + (size_t)C * 8 + (a > b2 ? a : b2) + (size_t)HG * C * 64 +
           (size_t)HG * 128 + (size_t)waves * kAttnXStageBytes + 128;
}

// ---------------------------------------------------------------------------------------------------------------
// Small kernels (elementwise.hip)
// ---------------------------------------------------------------------------------------------------------------
// conv_in input: [B][W][H][Cpad] bf16 from fp32 NCHW sources: x (cx ch, * scale), pos-encoding channel, cond (cc ch)
struct PackInputParams {
    const float* x; int cx; float scale;
    int pos_encoding;
    const float* cond; int cc;
    int B, W, H, Cpad;
    bf16_t* out;
    int* step_inc;            // sampler: the device step index is advanced here, by the first launch of a step (or null)
};
int launch_pack_input(const PackInputParams& p, hipStream_t stream);
int launch_nchw_f32_to_nhwc_bf16(const float* src, bf16_t* dst, int B, int C, int W, int H, int Cpad, hipStream_t s);
int launch_nhwc_bf16_to_nchw_f32(const bf16_t* src, float* dst, int B, int C, int W, int H, int ld, hipStream_t s);

// time embedding + all per-resnet projections for `rows` timesteps:
//   e=[cos(t f), sin(t f)] (dim0) -> Linear(dim0, D) -> SiLU -> Linear(D, D) -> SiLU -> Linear(D, total) (+bias)
struct TembParams {
    const float* t;          // [rows] timesteps as float (device)
    int rows, dim0, D, total;
    const float* w1; const float* b1;   // [D][dim0]
    const float* w2; const float* b2;   // [D][D]
    const float* wp; const float* bp;   // [total][D] concatenated time_emb_proj
    float* out;              // [rows][total]
    float* scratch;          // [2][rows][D] (the two hidden layers)
    int flip_sin_to_cos, freq_shift;   // diffusers Timesteps arguments: UNet2DModel (1, 0); sgm get_timestep_embedding (0, 1)
};
int launch_temb(const TembParams& p, hipStream_t stream);

// scheduler steps; coef on host (baked) or read from device table row *step_ptr
struct SchedParams {
    int mode;                 // 0 ddim, 1 ddpm; | prediction type << 1 (sched_prev)
    float coef[5];
    const float* coef_table;  // device [steps][5] or null
    const int* step_ptr;
    const float* eps; const float* x; const float* noise;   // noise may be null
    long long noise_step_stride;                            // elements between steps in `noise` (table mode)
    float* x_prev;
    long long n;
};
int launch_sched_step(const SchedParams& p, hipStream_t stream);
int launch_add_noise(const float* x0, const float* noise, const float* sa, const float* sb, int B, long long per,
                     float* out, hipStream_t stream);
int launch_diag_gaussian(const float* moments, const float* noise, float scale, int B, int z, int spatial, float* out,
                         hipStream_t stream);
int launch_step_counter(int* step_ptr, int set_to, int increment, hipStream_t stream);
int launch_thrash(const void* buf, size_t bytes, float* sink, hipStream_t stream);   // tuning aid: sweep a buffer through the caches
int launch_trunk_check(const int* err, float* a, long long na, float* b, long long nb, hipStream_t stream);   // NaN-poison on a tripped self-check
int launch_stamp(unsigned long long* slot, hipStream_t stream);   // *slot = wall_clock64() (100 MHz)
int launch_scale_f32(const float* src, float* dst, float scale, long long n, hipStream_t stream);

}  // namespace rldm
