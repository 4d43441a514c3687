/*
 * rangeldm_hip.h -- C ABI of librangeldm_hip.so: the MI355X (gfx950) implementation of the RangeLDM denoising
 * hot path (UNet2DModel forward, DDPM/DDIM scheduler step, AutoencoderKL encode/decode, whole sampling loop).
 *
 * The reference has no FFI: the path sits behind Python duck-typing.  Each entry point below names
 * the reference call it replaces (file:line under the reference tree).  The python modules under rangeldm_amd/ are the thin ctypes shim
 * that re-presents these as `unet(x, t).sample`, `scheduler.step(...).prev_sample`, `vae.decode(z).sample` and
 * `pipe(batch_size=..., num_inference_steps=...)`; INTEGRATION.md shows the binding a reference maintainer adds.
 *
 * Conventions
 *   - Every function returns 0 on success, non-zero on failure; the message is in rldm_last_error() (thread-local).
 *     Nothing throws across the ABI.
 *   - Tensors at the boundary are the reference's: fp32, NCHW with dim2 = W (azimuth), dim3 = H (beams)
 *     (ldm/dataset.py:228-233), contiguous, resident in device (HBM) memory, owned by the caller.  Internally
 *     activations are bf16 channels-last [B][W][H][C] with fp32 accumulation; weights are owned by the library.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all work is enqueued on it (the
 *     sampler uses an internal stream fenced by events against `stream`).  Single caller thread per handle.
 */
#ifndef RANGELDM_HIP_H
#define RANGELDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLDM_MAX_LEVELS 8

typedef struct rldm_unet rldm_unet;       /* UNet2DModel replacement   */
typedef struct rldm_vae rldm_vae;         /* AutoencoderKL replacement */
typedef struct rldm_sampler rldm_sampler; /* Pipeline.__call__ loop    */

/* UNet2DModel(**model_config): ldm/train_unconditional.py:237-242, ldm/configs/RangeLDM.yaml:17-24.
 * Library defaults the reference relies on are explicit fields. */
typedef struct rldm_unet_config {
    int32_t sample_w, sample_h;                 /* sample_size = (W, H)                                  */
    int32_t in_channels, out_channels;
    int32_t layers_per_block;
    int32_t num_levels;
    int32_t block_out_channels[RLDM_MAX_LEVELS];
    int32_t down_attn[RLDM_MAX_LEVELS];         /* 1: AttnDownBlock2D, 0: DownBlock2D                    */
    int32_t up_attn[RLDM_MAX_LEVELS];           /* 1: AttnUpBlock2D,   0: UpBlock2D                      */
    int32_t attention_head_dim;                 /* 8 (only value supported by the d=8 attention kernel)  */
    int32_t norm_num_groups;                    /* 32                                                    */
    float norm_eps;                             /* 1e-5                                                  */
    int32_t mid_attention;                      /* add_attention                                         */
    int32_t flip_sin_to_cos;                    /* Timesteps(..., flip_sin_to_cos): 1 for UNet2DModel    */
    int32_t freq_shift;                         /* Timesteps(..., downscale_freq_shift): 0; (0, 1) is the
                                                   sinusoid of vae/sgm/modules/diffusionmodules/model.py:28-46 */
} rldm_unet_config;

/* sgm Encoder/Decoder kwargs: vae/configs/kitti360.yaml:30-62 (== AutoencoderKL after ldm/convert_vae.py:123-189). */
typedef struct rldm_vae_config {
    int32_t in_channels, out_channels;
    int32_t ch;
    int32_t num_levels;
    int32_t ch_mult[RLDM_MAX_LEVELS];
    int32_t num_res_blocks;
    int32_t z_channels;
    int32_t double_z;
    int32_t norm_num_groups;
    float norm_eps;                             /* 1e-6 */
    float scaling_factor;                       /* 0.18215, ldm/convert_vae.py:159-168 */
} rldm_vae_config;

const char* rldm_last_error(void);
/* 0 if a gfx950 device is usable; fills name (may be NULL). */
int rldm_device_info(char* name, size_t name_len, int* compute_units);

/* ---- UNet2DModel ------------------------------------------------------------------------------------------- */
/* replaces UNet2DModel(**cfg) + replace_down/replace_conv surgery: ldm/inference.py:84-85,102-104 */
int rldm_unet_create(const rldm_unet_config* cfg, rldm_unet** out);
void rldm_unet_destroy(rldm_unet* m);
/* replaces load_state_dict / safetensors.load_model (ldm/inference.py:120): one call per diffusers key,
 * `data` = HOST fp32, `numel` elements in the reference's (C_out, C_in, kh, kw) / (out, in) order. */
int rldm_unet_set_param(rldm_unet* m, const char* name, const float* data, int64_t numel);
/* checks every key was supplied, packs bf16 MFMA-ordered weights to HBM.  May be called again after further
 * rldm_unet_set_param calls (load_state_dict on a live model, e.g. periodic EMA evaluation): it frees and rebuilds the
 * device weights and bumps the model's generation; every rldm_sampler built on the model notices at its next
 * rldm_sample and re-plans / re-captures its graphs (between set_param and finalize the model refuses to run). */
int rldm_unet_finalize(rldm_unet* m);
/* replaces `unet(sample, timestep).sample` (ldm/pipelines.py:103,239,360,500; train: ldm/train_unconditional.py:512).
 * sample: device fp32 [B, in_channels, W, H]; timesteps: HOST int64, nt == 1 (broadcast) or nt == B;
 * out: device fp32 [B, out_channels, W, H]. */
int rldm_unet_forward(rldm_unet* m, const float* sample, const int64_t* timesteps, int nt, int B, float* out,
                      void* stream);

/* ---- AutoencoderKL ------------------------------------------------------------------------------------------ */
int rldm_vae_create(const rldm_vae_config* cfg, rldm_vae** out);     /* ldm/inference.py:86-96 */
void rldm_vae_destroy(rldm_vae* m);
int rldm_vae_set_param(rldm_vae* m, const char* name, const float* data, int64_t numel);  /* ldm/inference.py:97 */
int rldm_vae_finalize(rldm_vae* m);
/* replaces `vae.decode(z).sample` (ldm/pipelines.py:367,507).  z: device fp32 [B, z_channels, W/f, H/f] (already
 * divided by scaling_factor by the caller, as the reference does at :365); image: device fp32 [B, out_ch, W, H]. */
int rldm_vae_decode(rldm_vae* m, const float* z, int B, int latent_w, int latent_h, float* image, void* stream);
/* replaces `vae.encode(x)` up to the moments (ldm/train_unconditional.py:480, ldm/pipelines.py:408).
 * x: device fp32 [B, in_ch, W, H]; moments: device fp32 [B, 2*z, W/f, H/f] = [mean | logvar]. */
int rldm_vae_encode(rldm_vae* m, const float* x, int B, int w, int h, float* moments, void* stream);
/* replaces DiagonalGaussianDistribution.sample (vae/sgm/modules/distributions/distributions.py:24-41):
 * out = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale.  All device fp32; n = B*z*w*h elements of `out`. */
int rldm_diag_gaussian_sample(const float* moments, const float* noise, float scale, int B, int z, int spatial,
                              float* out, void* stream);

/* ---- scheduler steps (elementwise; coefficients computed by the host shim exactly as diffusers does) -------- */
/* replaces DDIMScheduler.step (ldm/pipelines.py:244-246), SURVEY.md B.2:
 *   x0 = (x - sqrt_beta_t*eps)/sqrt_alpha_t ; prev = sqrt_alpha_prev*x0 + dir_coef*eps + sigma*noise
 * coef = {sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, dir_coef, sigma}; noise may be NULL when sigma == 0. */
int rldm_sched_ddim_step(const float coef[5], const float* eps, const float* x, const float* noise, float* x_prev,
                         int64_t n, void* stream);
/* replaces DDPMScheduler.step (ldm/pipelines.py:106,362), SURVEY.md B.3:
 *   x0 = (x - sqrt_beta_t*eps)/sqrt_alpha_t ; prev = c_x0*x0 + c_xt*x + sigma*noise
 * coef = {sqrt_alpha_t, sqrt_beta_t, c_x0, c_xt, sigma}. */
int rldm_sched_ddpm_step(const float coef[5], const float* eps, const float* x, const float* noise, float* x_prev,
                         int64_t n, void* stream);
/* The same two steps for any `prediction_type` of the scheduler config (ldm/train_unconditional.py:345-352 passes it through;
 * :505-510 trains epsilon or v_prediction): what the network output means --
 *   RLDM_PRED_EPSILON  x0 = (x - sqrt_beta_t*out)/sqrt_alpha_t,      eps = out                 (the two entry points above)
 *   RLDM_PRED_V        x0 = sqrt_alpha_t*x - sqrt_beta_t*out,        eps = sqrt_alpha_t*out + sqrt_beta_t*x
 *   RLDM_PRED_SAMPLE   x0 = out,                                     eps = (x - sqrt_alpha_t*out)/sqrt_beta_t
 * then DDIM: prev = coef[2]*x0 + coef[3]*eps + coef[4]*noise; DDPM: prev = coef[2]*x0 + coef[3]*x + coef[4]*noise (coef as above).
 * sampler_mode: RLDM_SAMPLER_DDIM | RLDM_SAMPLER_DDPM. */
#define RLDM_PRED_EPSILON 0
#define RLDM_PRED_V 1
#define RLDM_PRED_SAMPLE 2
int rldm_sched_step(int sampler_mode, int prediction_type, const float coef[5], const float* model_output, const float* x,
                    const float* noise, float* x_prev, int64_t n, void* stream);
/* replaces DDPMScheduler.add_noise (ldm/train_unconditional.py:498): out = sa[b]*x0 + sb[b]*noise (sa, sb HOST [B]).
 * DDPMScheduler.get_velocity (ldm/train_unconditional.py:508) is the same map: v = sa[b]*noise - sb[b]*x0, i.e. this entry point
 * with (x0, noise) swapped and sb negated (rangeldm_amd/schedulers.py get_velocity). */
int rldm_sched_add_noise(const float* x0, const float* noise, const float* sqrt_alpha, const float* sqrt_beta, int B,
                         int64_t per_sample, float* out, void* stream);

/* ---- whole sampling loop (HIP-graph captured) --------------------------------------------------------------- */
#define RLDM_SAMPLER_DDIM 0   /* eta = 0 DDIM  (DDIMPipelineRange, ldm/pipelines.py:144-258; BASELINE metric) */
#define RLDM_SAMPLER_DDPM 1   /* strided ancestral DDPM (LDMPipelineRange as shipped, ldm/pipelines.py:282-383) */

typedef struct rldm_sampler_config {
    int32_t batch;            /* per-GPU batch                                                                */
    int32_t num_steps;        /* num_inference_steps                                                          */
    int32_t mode;             /* RLDM_SAMPLER_*                                                               */
    int32_t pos_encoding;     /* extra constant channel: 1 at azimuth 0 (ldm/pipelines.py:229-232,346-349)     */
    int32_t cond_channels;    /* channels of the per-step concatenated condition (ldm/pipelines.py:498), or 0  */
    /* per-step scheduler coefficients, HOST, [num_steps][5] in the layout of rldm_sched_{ddim,ddpm}_step      */
    const float* coef;
    /* timesteps, HOST int64 [num_steps] (scheduler.timesteps)                                                 */
    const int64_t* timesteps;
    /* routing options of THIS sampler's plans: the bits of rldm_debug_set_flags, scoped to the sampler (0: defaults).
     * 1 << 24 = every layer a launch of its own -- what a host sets for a sampler it knows will share the GPU.         */
    int32_t plan_flags;
    /* RLDM_PRED_*: what the UNet's output means to the scheduler step (scheduler.config.prediction_type)               */
    int32_t prediction_type;
} rldm_sampler_config;

/* replaces Pipeline.__init__ + the per-call setup of ldm/pipelines.py:329-349; vae may be NULL (pixel-space RangeDM) */
int rldm_sampler_create(rldm_unet* unet, rldm_vae* vae, const rldm_sampler_config* cfg, rldm_sampler** out);
void rldm_sampler_destroy(rldm_sampler* s);
/* replaces the loop + decode of ldm/pipelines.py:353-367 (:496-507 with cond, :234-246 without VAE).
 * x_T: device fp32 [B, out_ch, W, H]; step_noise: device fp32 [num_steps, B, out_ch, W, H] or NULL (DDIM);
 * cond: device fp32 [B, cond_channels, W, H] or NULL; images: device fp32 [B, 2, 4W, 4H] (or the final x_0 when
 * the sampler has no VAE); latents_out: optional device fp32 [B, out_ch, W, H] receiving the final latent. */
int rldm_sample(rldm_sampler* s, const float* x_T, const float* step_noise, const float* cond, float* images,
                float* latents_out, void* stream);
/* The reference's contract is "a correct tensor or an exception" (ldm/pipelines.py:218-222,463-464).  rldm_sample is asynchronous,
 * so the exception half is this call: it waits for the last rldm_sample of `s` and returns 0 when its outputs are valid.  Non-zero
 * = the self-check of the call's persistent launches tripped (1: a wait inside a workgroup cluster gave up, i.e. the GPU was shared
 * with other work; 2: a cluster was spread over several XCDs); the outputs of that call were NaN-marked ON THE DEVICE by the call's
 * last launch (so an unchecked consumer cannot mistake them for images), rldm_last_error() explains, and the sampler has already
 * rebuilt its plans as one launch per layer: calling rldm_sample again gives valid images.  The Python shim calls it in every
 * pipeline __call__ and raises RuntimeError. */
int rldm_sampler_status(rldm_sampler* s);

/* ---- multi-GPU exchange steps: RCCL over xGMI on the caller's stream (rangeldm_amd/csrc/collective.hip) --
 * One process per GPU.  Rank 0 makes the id and hands its RLDM_UNIQUE_ID_BYTES bytes to the other ranks by any side channel
 * (MPI, a file, torch.distributed's store); every rank then calls rldm_comm_create with its current HIP device set.  RCCL is
 * bound at run time (dlopen): the copy already loaded in the process (PyTorch's) if there is one, else RLDM_RCCL_LIB, else
 * the system librccl. */
#define RLDM_UNIQUE_ID_BYTES 128
typedef struct rldm_comm rldm_comm;
/* binds RCCL and nothing else: what every rank but 0 calls to learn whether it CAN take part (rldm_comm_unique_id would also open
 * ncclGetUniqueId's listening socket and root thread, which only the rank whose id is used should own) */
int rldm_comm_bind(void);
int rldm_comm_unique_id(void* id_out, size_t cap);          /* rank 0 only */
int rldm_comm_create(const void* unique_id, int rank, int world, rldm_comm** out);     /* collective over all ranks */
void rldm_comm_destroy(rldm_comm* c);
int rldm_comm_info(const rldm_comm* c, int* rank, int* world, char* rccl_origin, size_t cap);
/* replaces the per-rank file writing of ldm/inference.py:159-183 as the hand-over of a sample-sharded batch: every rank
 * contributes `count` floats (its finished (B_local, 2, W, H) images, or the x_0 latents) and receives all ranks' buffers in
 * rank order in `all` (world * count floats).  Stream-ordered behind rldm_sample when given the same stream. */
int rldm_allgather_images(rldm_comm* c, const float* local, float* all, int64_t count, void* stream);
/* replaces DDP's gradient exchange (accelerate.prepare(model), ldm/train_unconditional.py:402-404; backward :545): in-place
 * sum (average != 0: mean) over the ranks of `count` floats -- one contiguous bucket of the flat gradient buffer. */
int rldm_allreduce_grads(rldm_comm* c, float* grads, int64_t count, int average, void* stream);

/* ---- range image <-> point cloud (rangeldm_amd/csrc/lidar.hip) ------------------------- */
typedef struct rldm_lidar rldm_lidar;     /* point_cloud_to_range_image replacement, ldm/dataset.py:135-294 */

/* point_cloud_to_range_image.__init__ (ldm/dataset.py:136-154); per-beam tables of the sensor subclasses
 * (ldm/kitti360_range_image.py:19-48, ldm/nuscenes_range_image.py:20-35) are passed to rldm_lidar_create. */
typedef struct rldm_lidar_config {
    int32_t beams;                      /* H = len(incl)                                            */
    int32_t width;                      /* azimuth bins of the projection (`width`, 1024)           */
    int32_t mode;                       /* 0: (r - mean) / std, 1: log2(r + 1) / 6, 2: 1 / r        */
    float mean, std;                    /* 20, 40                                                   */
    float range_fill, intensity_fill;   /* range_fill_value = [100, 0]                              */
    int32_t grid[3];                    /* grid_sizes = [D, H, W] of the BEV volume ([1, 1024, 1024]) */
    float pc_range[6];                  /* [-25.6, -25.6, -3, 25.6, 25.6, 1]                        */
    int32_t normalize_volume_densities; /* log(density + 1)                                         */
} rldm_lidar_config;

int rldm_lidar_create(const rldm_lidar_config* cfg, const float* incl /*host [beams]*/, const float* height /*host*/,
                      rldm_lidar** out);
void rldm_lidar_destroy(rldm_lidar* l);
/* to_pc_torch (ldm/dataset.py:228-278): range_images device fp32 (B, C, W, beams) -> points device fp32
 * [B][W*beams][C > 1 ? 4 : 3] = (x, y, z[, remission]); point index = w * beams + h.  The input is not modified. */
int rldm_lidar_to_points(rldm_lidar* l, const float* range_images, int B, int C, int W, float* points, void* stream);
/* to_voxel (ldm/dataset.py:280-294 over _splat_points_to_volumes :13-132): -> voxel device fp32 (B, 2*D, H, W):
 * D planes of [log(1 +)] vote density, then D planes of density-normalised remission.  C >= 2. */
int rldm_lidar_to_voxel(rldm_lidar* l, const float* range_images, int B, int C, int W, float* voxel, void* stream);
/* `pc[np.linalg.norm(pc[:, :3], 2, axis=1) < max_depth]` per image, order preserved (ldm/inference.py:177-179):
 * points [B][N][cols] -> out [B][N][cols] (first counts[b] rows valid), counts device int32 [B]. cols = 3 or 4. */
int rldm_lidar_filter_points(rldm_lidar* l, const float* points, int B, int N, int cols, float max_depth, float* out,
                             int32_t* counts, void* stream);
/* `(x[b].permute(2, 1, 0).clip(0, 1) * 255).astype(uint8)[:, :, channel]` (ldm/inference.py:180-183):
 * src device fp32 (B, C, W, H) -> dst device bytes [B][H][W] (the pixels of the 8-bit range / BEV PNG). */
int rldm_render_u8(const float* src, int B, int C, int W, int H, int channel, uint8_t* dst, void* stream);
/* point_cloud_to_range_image.__call__ + process_miss_value + normalize + the (2,1,0) permute of
 * RangeDataset.__getitem__ (ldm/dataset.py:159-226, 320-333): one sweep `points` device fp32 [n_points][stride]
 * (x, y, z, intensity, ...) -> image device fp32 (2, width, beams), mask / car_window_mask device bytes
 * (width, beams).  rows: device int32 [n_points] beam index per return, or NULL = nearest inclination
 * (ldm/kitti360_range_image.py:51-61).  min_depth > 0 drops returns with |xyz| <= min_depth
 * (ldm/nuscenes_range_image.py:37-41).  `points` is not modified. */
int rldm_lidar_project(rldm_lidar* l, const float* points, int n_points, int stride, const int32_t* rows, float min_depth,
                       float* image, uint8_t* mask, uint8_t* car_window_mask, void* stream);

/* ---- BEV-histogram evaluation (rangeldm_amd/csrc/metrics.hip) ------------------------------ */
/* load_point_cloud_xyz depth mask (metrics/metrics/histogram/mmd.py:39-44: min_depth < |xyz| < max_depth) +
 * point_cloud_to_histogram(field_size, bins, pc)[0] (histogram.py:4-18, np.histogramdd over +-field_size/2) for a ragged
 * batch: sample s = points[offsets[s] .. offsets[s+1]) (device fp32 [n][stride], device int32 [num_samples + 1])
 * -> hist device uint32 [num_samples][bins][bins] (x bin major).  Counts are exact. */
int rldm_bev_histogram(const float* points, const int32_t* offsets, int num_samples, int stride, float field_size, int bins,
                       float min_depth, float max_depth, uint32_t* hist, void* stream);
/* jsd_2d(sum(hx) / total, sum(hy) / total) (metrics/metrics/histogram/jsd.py:14-16,90-101; scipy jensenshannon, base e).
 * hx / hy device uint32 [n][bins][bins]; *jsd is a HOST double (the call synchronises the stream). */
int rldm_hist_jsd(const uint32_t* hx, int nx, const uint32_t* hy, int ny, int bins, double* jsd, void* stream);
/* np.linalg.norm(x_i / sum(x_i) - y_j / sum(y_j), 2) ** 2 for every pair -- the SPECTRAL norm the `gaussian` kernel of
 * metrics/metrics/histogram/dist_helper.py:84-104 takes of two 2-D pmfs.  lambda device fp32 [nx][ny]; symmetric = 1
 * (hx == hy): only j > i is written, the rest is 0.  bins <= 104, multiple of 4. */
int rldm_hist_spectral_sq(const uint32_t* hx, int nx, const uint32_t* hy, int ny, int bins, int symmetric, float* lambda,
                          void* stream);
/* compute_mmd(samples1, samples2, gaussian, is_hist=True) (dist_helper.py:156-172) with sigma (0.5):
 * out4 HOST doubles = {s1, s2, cross, s1 + s2 - 2 cross} (the call synchronises the stream). */
int rldm_hist_mmd(const uint32_t* hx, int nx, const uint32_t* hy, int ny, int bins, float sigma, double* out4, void* stream);

/* ---- UNet training step (rangeldm_amd/csrc/train.hip; ldm/train_unconditional.py:466-558) ----
 * Op-level entry points driven by rangeldm_amd/training.py (the autograd tape is host-side).  Every tensor is device
 * fp32, activations / gradients channels-last [B][W][H][C] (W wraps, H zero-pads); GEMM operands are rounded to bf16 into
 * the MFMA with fp32 accumulation (the reference's `mixed_precision: bf16`). */
typedef struct rldm_train_conv_desc {
    int32_t B, Win, Hin, Cin;   /* input tensor                                                                    */
    int32_t N;                  /* output channels                                                                 */
    int32_t taps;               /* 9: 3x3 pad 1 (ldm/utils.py:40-55), 1: 1x1 / Linear                             */
    int32_t stride;             /* 1 | 2 (Downsample2D, ldm/utils.py:107-116)                                      */
    int32_t mode;               /* 0 plain, 1 nearest-x2 of the input first (Upsample2D), 2 zero insertion (data
                                   gradient of a stride-2 conv); output = (Win << (mode != 0)) / stride            */
} rldm_train_conv_desc;
/* y = conv(x) + bias + rowadd[b] + res (each optional); w_packed = bf16 [N][taps][ceil16(Cin)] from
 * rldm_train_pack_weights (forward copy, or the transposed copy with Cin/N swapped for the data gradient). */
int rldm_train_conv(const rldm_train_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* rowadd,
                    int rowadd_ld, const float* res, float* y, int accumulate, void* stream);
/* How many K splits rldm_train_conv uses for this shape (1: none).  A split launch adds its partial tiles to y atomically
 * after zero-filling it; with accumulate != 0 it adds onto what y holds, so a caller that owns pre-zeroed outputs (one fill
 * for all of a step's split launches) passes accumulate = 1 and saves the per-launch fill. */
int rldm_train_conv_splits(const rldm_train_conv_desc* d, int rowadd_ld);
/* dw[N][Cin][taps] += sum over pixels of dy (x) x  (torch weight layout; dw must be zeroed by the caller). */
int rldm_train_wgrad(const rldm_train_conv_desc* d, const float* dy, const float* x, float* dw, void* stream);
/* The same plus the bias / per-image row gradients of rldm_train_colsum from the same pass over dy (rows / total may be NULL);
 * inside the all-taps kernel the sums are taken from the staged bf16 tile. */
int rldm_train_wgrad_bias(const rldm_train_conv_desc* d, const float* dy, const float* x, float* dw, float* rows, int rows_ld,
                          int rows_accumulate, float* total, void* stream);
/* rows[b][n] (+)= sum over image b's pixels of dy[p][n] (time-embedding row gradient); total[n] += over all images (bias). */
int rldm_train_colsum(const float* dy, int B, int npix, int N, float* rows, int rows_ld, int rows_accumulate, float* total,
                      void* stream);
/* GroupNorm (+ SiLU): stats [B][groups][2] = (mean, rstd) are written by forward and read by backward. */
int rldm_train_gn_forward(const float* x, int B, int npix, int C, int groups, float eps, const float* gamma, const float* beta,
                          int silu, float* stats, float* y, void* stream);
int rldm_train_gn_backward(const float* x, const float* dy, const float* stats, int B, int npix, int C, int groups,
                           const float* gamma, const float* beta, int silu, float* scratch /*[B][groups][2]*/, float* dx,
                           int accumulate, float* dgamma, float* dbeta, void* stream);
/* ---- fused tape (round 5): GroupNorm never runs as a tensor pass of its own ----------------------------------------------
 * A tensor's GroupNorm statistics travel as per-(image, channel) pairs cs [B][C][2] = (sum, sum of squares), accumulated
 * (atomically, into a buffer the caller zeroed) by the epilogue of the conv that produced the tensor.  Consumers rebuild
 * act(GroupNorm(x)) from x + cs while they stage x -- the reference's `F.silu(norm(x))` in front of every ResnetBlock2D conv
 * and `group_norm(x)` in front of to_q/k/v (sgm model.py:93-125, diffusers ResnetBlock2D / Attention) -- and a concatenated
 * input (`torch.cat([h, skip], 1)` of the up blocks) is read from its two sources in place.  Backward: the data-gradient conv
 * whose output is d act(GN(g)) turns it into dz = dy act'(z) in its epilogue and accumulates gs [B][C][2] = (sum dz, sum dz
 * xhat); rldm_train_gn_backward_apply finishes dx (+ residual gradient, split over the two sources) and d gamma / d beta. */
typedef struct rldm_train_fuse {
    /* input side (conv, wgrad: the `x` operand) */
    const float* x1;            /* second source: channels [C0, Cin) of the input (NULL: `x` holds all Cin)                */
    int32_t C0;                 /* channels of `x` when x1 != NULL (a multiple of 64 -- 32 when Cin % 64 != 0)             */
    const float* cs0;           /* (sum, sumsq) pairs of x / x1; cs0 == NULL: the input is used as it is                   */
    const float* cs1;
    const float* gamma;         /* GroupNorm affine over the Cin channels                                                  */
    const float* beta;
    int32_t silu, groups;
    float eps;
    /* output side (conv only; at most one of the two) */
    float* cs_out;              /* += (sum, sumsq) of y per (image, channel): [B][N][2]                                    */
    const float* g0;            /* data gradient: y = d act(GN(cat(g0, g1))); stored as dz, gs_out [B][N][2] += sums       */
    const float* g1;
    int32_t G0;                 /* channels of g0 when g1 != NULL                                                          */
    const float* gcs0;          /* (sum, sumsq) pairs of g0 / g1                                                           */
    const float* gcs1;
    const float* ggamma;
    const float* gbeta;
    int32_t gsilu, ggroups;
    float geps;
    float* gs_out;
} rldm_train_fuse;
/* rldm_train_conv with the above folded in.  prezeroed: as `accumulate` of rldm_train_conv for split launches (y is a zeroed
 * buffer of the caller); a fused epilogue never adds onto an existing y.  _ok: 1 if the shape has a fused instance. */
int rldm_train_conv_fused_ok(const rldm_train_conv_desc* d, const rldm_train_fuse* f, int rowadd_ld);
int rldm_train_conv_fused(const rldm_train_conv_desc* d, const rldm_train_fuse* f, const float* x, const void* w_packed,
                          const float* bias, const float* rowadd, int rowadd_ld, const float* res, float* y, int prezeroed,
                          void* stream);
/* rldm_train_wgrad_bias with x = act(GN(cat(x, x1))) rebuilt while staging (input side of `f` only). */
int rldm_train_wgrad_fused_ok(const rldm_train_conv_desc* d, const rldm_train_fuse* f);
int rldm_train_wgrad_fused(const rldm_train_conv_desc* d, const rldm_train_fuse* f, const float* dy, const float* x, float* dw,
                           float* rows, int rows_ld, int rows_accumulate, float* total, void* stream);
/* The all-taps weight-gradient kernel leaves partial tiles that a reduction adds into dw.  With deferral on, that reduction is not
 * launched on its own: it rides (as extra workgroups) on the next rldm_train_conv / _conv_fused launch on the same stream -- the
 * data gradient of the same layer, which does not depend on it -- or is launched by whatever comes first of: the next weight-gradient
 * call, rldm_train_flush_reduce, rldm_train_defer_reduce(0).  Call rldm_train_flush_reduce before anything else reads dw. */
int rldm_train_defer_reduce(int on);
int rldm_train_flush_reduce(void);
/* (round 6) Grouped weight gradients.  The weight gradients of a step (`loss.backward()`, ldm/train_unconditional.py:545) depend on
 * nothing behind them in the backward pass and nothing in it depends on them.  With grouping on, rldm_train_wgrad_bias / _wgrad_fused
 * calls that the all-taps kernel covers are QUEUED (their dy / x / statistics operands must stay alive and unmodified), and
 * rldm_train_wgrad_group_flush -- or group(0), or a queued call on another stream -- runs the queue as a handful of launches that each
 * compute up to 22 layers (block id -> layer, tile, K slice; the K slices of a tile are summed in slice order by the tile's last
 * arriver: no reduction launches, bit-reproducible gradients).  dw / rows / total are final only after the flush.  One caller thread. */
int rldm_train_wgrad_group(int on);
int rldm_train_wgrad_group_flush(void);
int rldm_train_wgrad_group_pending(void);   /* queued layers (tests) */
/* cs [B][C][2] += (sum, sumsq) per (image, channel) of x [B][npix][C]: for tensors no fused conv produced. */
int rldm_train_chan_stats(const float* x, int B, int npix, int C, float* cs, void* stream);
/* dx = rstd (gamma dz - mean_g(gamma dz) - xhat mean_g(gamma dz xhat)) + res, the GroupNorm over cat(x0, x1) [C0 | C - C0
 * channels]; written (accumulate == 0) or added to dx0 / dx1; dgamma[c] += sum_b gs[b][c].y, dbeta[c] += sum_b gs[b][c].x
 * (both may be NULL).  res: [B][npix][C] or NULL. */
int rldm_train_gn_backward_apply(const float* dz, const float* x0, const float* x1, int C0, const float* cs0, const float* cs1,
                                 const float* gs, int B, int npix, int C, int groups, float eps, const float* gamma,
                                 const float* res, float* dx0, int accumulate0, float* dx1, int accumulate1, float* dgamma,
                                 float* dbeta, void* stream);
/* Linear layers on B <= 16 rows (TimestepEmbedding MLP, ResnetBlock2D.time_emb_proj): y[b][n] (+)= sum_k x[b][k]
 * W[n][k] + bias[n] with W the packed bf16 copy [N][ceil16(K)] (forward copy; the transposed copy gives the data gradient);
 * x / y rows may be slices of wider matrices (ldx / ldy in floats).  _wgrad: dw [N][K] fp32 += dy^T x, dbias[n] += sum_b dy. */
int rldm_train_linear_rows(const float* x, int ldx, const void* w_packed, int K, const float* bias, float* y, int ldy, int B, int N,
                           int accumulate, void* stream);
int rldm_train_linear_rows_wgrad(const float* dy, int ldy, const float* x, int ldx, int B, int N, int K, float* dw, float* dbias,
                                 void* stream);
/* softmax(q k^T / sqrt(8)) v per head of 8 channels; q, k, v, o [B][L][C]; lse / delta [B][C/8][L]. */
int rldm_train_attention_forward(const float* q, const float* k, const float* v, int B, int L, int C, float* o, float* lse,
                                 void* stream);
int rldm_train_attention_backward(const float* q, const float* k, const float* v, const float* o, const float* dO, const float* lse,
                                  int B, int L, int C, float* delta, float* dq, float* dk, float* dv, void* stream);
/* The same with q, k, v the three thirds of ONE projection output qkv [B][L][3C] (to_q / to_k / to_v fused into one 1x1 conv
 * with 3C outputs) and dq, dk, dv the thirds of dqkv [B][L][3C]; o, dO [B][L][C]. */
int rldm_train_attention_qkv_forward(const float* qkv, int B, int L, int C, float* o, float* lse, void* stream);
int rldm_train_attention_qkv_backward(const float* qkv, const float* o, const float* dO, const float* lse, int B, int L, int C,
                                      float* delta, float* dqkv, void* stream);
int rldm_train_add(const float* a, const float* b, float* y, int64_t n, void* stream);
int rldm_train_copy_channels(const float* src, int src_ld, int src_off, float* dst, int dst_ld, int dst_off, int ncopy,
                             int64_t npix, int accumulate, void* stream);
int rldm_train_sum2x2(const float* du, int B, int W, int H, int C, float* dx, void* stream);       /* nearest-x2 backward */
int rldm_train_silu(const float* x, const float* dy, float* y, int64_t n, int backward, int accumulate, void* stream);
int rldm_train_timestep_embedding(const int64_t* timesteps /*device*/, int B, int dim, float* out, void* stream);
/* (B, C, W, H) fp32 + optional pos-encoding channel (ldm/train_unconditional.py:455-463,500-501) -> [B][W][H][C(+1)] */
int rldm_train_pack_input(const float* x, int B, int C, int W, int H, int pos_encoding, float* y, void* stream);
int rldm_train_unpack_output(const float* x, int B, int C, int W, int H, float* y, void* stream);
/* F.mse_loss(model_output, target) with optional per-sample weights (min-SNR, :529-543): pred [B][W][H][C], target
 * (B, C, W, H); dpred [B][W][H][C]; *loss device double. */
int rldm_train_mse(const float* pred, const float* target, const float* weight, int B, int C, int W, int H, float* dpred,
                   double* loss, void* stream);
int rldm_train_sqnorm(const float* g, int64_t n, double* out /*device*/, void* stream);
typedef struct rldm_adamw_config {
    float lr, beta1, beta2, eps, weight_decay;  /* torch.optim.AdamW, ldm/train_unconditional.py:357-363           */
    float max_grad_norm;                        /* clip_grad_norm_ (:548); <= 0: off; uses *sqnorm                 */
    float ema_decay;                            /* EMAModel.step (:556); used when ema != NULL                     */
    int32_t step;                               /* 1-based optimizer step (bias corrections)                       */
} rldm_adamw_config;
int rldm_train_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema, const double* sqnorm,
                     int64_t n, const rldm_adamw_config* c, void* stream);
/* The same step with its per-step scalars read from device memory -- dyn[4] = (lr, 1 - beta1^step, 1 - beta2^step, ema decay),
 * written by rldm_train_hyper_step -- so a captured step graph can be replayed; zero_grads != 0 also clears `grads`
 * (optimizer.zero_grad(), ldm/train_unconditional.py:551). c->lr / c->step / c->ema_decay are ignored. */
int rldm_train_adamw_dyn(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema, const double* sqnorm,
                         int64_t n, const rldm_adamw_config* c, const float* dyn, int zero_grads, void* stream);
typedef struct rldm_hyper_config {
    float lr, beta1, beta2;                     /* base learning rate, AdamW betas                                  */
    float ema_max_decay, ema_inv_gamma, ema_power; /* EMAModel(use_ema_warmup=True) (:320-329)                      */
    int64_t lr_warmup_steps, total_steps;       /* get_scheduler("cosine", ...) (:394-399)                          */
} rldm_hyper_config;
/* step = ++*step_counter (device int64); dyn[4] <- the scalars of optimizer step `step` (lr of step - 1 scheduler steps). */
int rldm_train_hyper_step(int64_t* step_counter, const rldm_hyper_config* c, float* dyn, void* stream);
/* master fp32 [N][Cin][taps] -> bf16 [N][taps][ceil16(Cin)] (forward) and bf16 [Cin][taps][ceil16(N)] (flipped / transposed:
 * data gradient; may be NULL). */
int rldm_train_pack_weights(const float* w, int N, int Cin, int taps, void* w_forward, void* w_transposed, void* stream);
/* the same for every layer of the model in one launch: descs = DEVICE array sorted by `first` (cumulative count of
 * max(forward, transposed) copy elements), params = the flat fp32 parameter buffer. */
typedef struct rldm_pack_desc {
    int64_t first;          /* cumulative element index of this layer in the launch                               */
    int64_t param_offset;   /* offset of the layer's [N][Cin][taps] weight in the flat parameter buffer (elements) */
    void* w_forward;        /* bf16 [N][taps][ceil16(Cin)]                                                         */
    void* w_transposed;     /* bf16 [Cin][taps][ceil16(N)] or NULL                                                 */
    int32_t N, Cin, taps, pad_;
} rldm_pack_desc;
int rldm_train_pack_weights_all(const float* params, const rldm_pack_desc* descs, int num_layers, int64_t total, void* stream);
/* the same as a tiled transpose (coalesced reads and writes): descs[i].first = cumulative count of 64 x 64 (N, Cin) tiles,
 * total_tiles = their sum = the grid. */
int rldm_train_pack_weights_tiled(const float* params, const rldm_pack_desc* descs, int num_layers, int64_t total_tiles, void* stream);

/* ---- introspection used by bench.py / tests ----------------------------------------------------------------- */
/* algorithmic FLOPs (2*MACs of conv/linear/QK^T/PV) of one UNet forward / VAE decode / encode for batch B */
double rldm_unet_flops(rldm_unet* m, int B);
double rldm_vae_decode_flops(rldm_vae* m, int B, int latent_w, int latent_h);
/* number of kernel launches in one UNet forward plan (for the launch-overhead budget in DESIGN.md) */
int rldm_unet_num_launches(rldm_unet* m, int B);

/* Instrumented pass used by bench.py for the roofline: runs ONE UNet step (+ one VAE decode) eagerly on the sampler's
 * stream with a HIP event pair around every kernel launch and writes JSON
 *   {"unet_step": {kernel: {launches, ms, flops, bytes}, ...}, "vae_decode": {...}, "plan_flags": F, "fell_back": bool, ...}
 * (flops/bytes = algorithmic work of those launches; plan_flags = the routing bits in effect for this sampler's plans,
 * fell_back = the library added bits of its own, i.e. a persistent launch failed its self-check earlier) into json_out. */
int rldm_sampler_profile(rldm_sampler* s, const float* x_T, char* json_out, size_t cap);

/* low-level op entry points (used by the parity tests to check each kernel in isolation) */
typedef struct rldm_conv_desc {
    int32_t B, Cin0, Cin1, Win, Hin;  /* inputs x0 [B][Win][Hin][Cin0] (+ x1 [..][Cin1] concatenated), bf16 NHWC */
    int32_t Cout;
    int32_t ksize;                    /* 1 or 3                                                                */
    int32_t stride;                   /* 1 or 2                                                                */
    int32_t pad_mode;                 /* 0: symmetric pad 1 (wrap W / zero H); 1: end-only pad (VAE downsample) */
    int32_t upsample;                 /* 1: nearest x2 folded into the input indexing                          */
    int32_t gn;                       /* 1: GroupNorm(32) prologue on the (concatenated) input                  */
    int32_t silu;                     /* 1: SiLU after the norm                                                */
    float eps;
} rldm_conv_desc;
/* One fused conv: y = conv(silu(GN(cat[x0,x1]))) + bias + temb[b] + res.  All device pointers; x0/x1/res/y are
 * fp32 NCHW here (converted on device) so tests can feed reference tensors; weight (Cout, Cin, k, k), gamma/beta,
 * bias, temb ([B][Cout] or NULL) are HOST fp32. */
int rldm_test_conv(const rldm_conv_desc* d, const float* x0, const float* x1, const float* weight, const float* bias,
                   const float* gamma, const float* beta, const float* temb, const float* res, float* y, void* stream);
/* kernel-tuning aids (tools/bench_conv.py): time the fused conv kernel alone on synthetic data with HIP events on
 * `stream` (avg_us per launch over `iters` back-to-back launches; with_res = channels of the fused residual phase:
 * 0 none, Cout identity, else a synthetic 1x1 shortcut); force the pixel/channel tile and split-K (0 = automatic). */
int rldm_bench_conv(const rldm_conv_desc* d, int with_res, int with_temb, int warmup, int iters, float* avg_us,
                    char* kernel_name, size_t name_cap, void* stream);
int rldm_debug_force_tile(int BM, int BN, int ksplit);
/* self-check word of the persistent trunk launches (trunk.hip) of the batch-B plan of a UNet: 0 fine or no trunk, 1 a bounded
 * wait gave up, 2 a cluster of workgroups was spread over several XCDs; synchronises the device */
int rldm_unet_trunk_status(rldm_unet* m, int B);
/* routing options of the rldm_unet_forward plans of ONE model (the bits of rldm_debug_set_flags, scoped; drops its cached plans) */
int rldm_unet_set_plan_flags(rldm_unet* m, int flags);
/* tests: the next rldm_sample of `s` behaves as if a cluster wait had given up in the middle of the run (code 1 or 2) */
int rldm_debug_inject_trunk_error(rldm_sampler* s, int code);
int rldm_debug_timestamps(unsigned long long* host_out);   /* NULL: enable; else read back [4][64] s_memtime stamps */
int rldm_debug_block_times(unsigned long long* host_out, int nblocks);   /* ABLATE builds: [start, end] (100 MHz) of every workgroup of the last conv_stream launch */
/* routing / ablation switches, PROCESS-WIDE, read when a plan is built (RLDM_DBG_FLAGS seeds them): for tuning runs and tests.  A host
 * that wants one sampler / model routed differently uses rldm_sampler_config::plan_flags / rldm_unet_set_plan_flags (same bits, scoped),
 * and the library's own fall-backs are scoped the same way.  The ones a maintainer may need
 * (INTEGRATION.md section 5): 1 << 24 every layer a launch of its own with the persistent launches' tiles (identical results),
 * 1 << 25 ... with the default tiles, 1 << 26 no multi-tile clusters (the 64x4 / 256x16 levels as launches: required when the GPU
 * is shared with other streams), 512 / 1 << 28 only the conv_small / conv_stream clusters off, 1 << 27 gn_apply stays a launch,
 * 1 << 30 the 128x8 conv pairs as 2-phase launches, 64 keeps the sampler's pack_input launch, 1 << 23 keeps the scheduler step
 * a launch, 1 << 20 every GroupNorm on the consumer side.  0 restores the defaults. */
int rldm_debug_set_flags(int flags);
/* second word of the same kind (RLDM_DBG_FLAGS2 seeds it), round 4: 1 / 2 / 4 keep the 8-wave conv_stream workgroups at the
 * full-resolution levels / the 128x8 level / the VAE's 64-channel level (default: 4-wave workgroups, two resident per CU), 8 keeps the
 * 4-wave full-resolution convs launches of their own, 16 / 128 keep the round-3 tiles of the 128x8 level / the generic kernel for the first
 * down-sampler, 32 / 64 route to the specialised-wave experiment / the 64-pixel tile wherever it fits (tests); 1 << 24 keeps the VAE
 * decoder's 64 -> 64 convs and conv_out on the per-tile kernels (default: conv_regw.hip -- weights resident in registers, a run of tiles per
 * workgroup), 1 << 25 caps that kernel's grid at 8 runs (tests: runs of several tiles on small images), 1 << 26 makes a test / bench conv of
 * <= 4 output channels an fp32-NCHW output layer like the decoder's conv_out (rldm_test_conv, rldm_bench_conv); 1 << 27 keeps nearest x2 +
 * 3x3 convs as a 3x3 over the up-sampled halo (default: the sub-pixel form -- four 2x2 convs over the input with summed weights), 1 << 28
 * keeps the 256x16 level's sub-pixel up-sampler a launch of its own (default: a phase of that level's persistent launch), 1 << 29 keeps the
 * 16 x 8 tiles with a staged halo ring at the 16- / 8-beam levels (default: tiles as tall as the image). */
int rldm_debug_set_flags2(int flags);
/* in-graph timeline of the UNet ops of the sampler's step graph (debug flag 8192 set before rldm_sampler_create) */
int rldm_debug_graph_trace(unsigned long long* stamps, int cap, char* names, size_t names_cap);   /* kernel ablation switches, see ConvParams::dbg */
/* statistics side-output of the conv epilogue (feeds the next GroupNorm): stats device fp32 [B][Cout][2] = per-image
 * (sum, sum of squares) of the bf16 outputs of conv(x0); plain single-input conv only. */
int rldm_test_conv_stats(const rldm_conv_desc* d, const float* x0, const float* weight, const float* bias, float* stats,
                         void* stream);
/* multi-head (d=8) self-attention core: qkv device fp32 [B][L][3C] -> out device fp32 [B][L][C] */
int rldm_test_attention(const float* qkv, int B, int L, int C, float* out, void* stream);
/* the launch every attention block of the UNet actually runs -- GroupNorm -> to_q / to_k / to_v -> softmax(q k^T/sqrt 8) v
 * (diffusers Attention before to_out; reference analogue vae/sgm/modules/attention.py:194-284 behind a GroupNorm):
 * x device fp32 [B][L][C] token-major, gamma / beta host [C], wqkv host [3C][C] (q | k | v rows), bqkv host [3C]
 * -> out device fp32 [B][L][C], heads concatenated. */
int rldm_test_attention_qkv(const float* x, int B, int L, int C, int groups, float eps, const float* gamma, const float* beta,
                            const float* wqkv, const float* bqkv, float* out, void* stream);
/* HIP-event time of that launch alone on synthetic data (tools/bench_attn.py) */
int rldm_bench_attention_qkv(int B, int L, int C, int warmup, int iters, float* avg_us, void* stream);

/* ---- calibration of the box (bench.py `calibration`; the metric's definition) -----------------------------------
 * What this GPU does right now, measured in the benchmark's own process so that a throughput line can be read against it:
 * mfma_tflops / mfma_clock_mhz = a pure v_mfma_f32_32x32x16_bf16 loop on every SIMD (HIP events) and the shader clock while it runs
 * (s_memtime ticks per s_memrealtime tick x 100 MHz); copy_gbs = a 1 GiB device-to-device copy, bytes read + written per second.
 * Synchronises the stream; allocates and frees 2 GiB. */
int rldm_calibrate(double* mfma_tflops, double* mfma_clock_mhz, double* copy_gbs, void* stream);
/* Clock stamps around any work on `stream`: RLDM_CALIB_STAMP_BLOCKS workgroups (block ids go round the XCDs) each write
 * slots[4 * block + {0, 1, 2, 3}] = {XCC id, s_memtime, s_memrealtime, 1}.  Two stamps of the same XCC id: (d s_memtime / d s_memrealtime)
 * x 100 MHz = the mean shader clock over what ran between them.  slots: device, 4 * RLDM_CALIB_STAMP_BLOCKS uint64. */
#define RLDM_CALIB_STAMP_BLOCKS 16
int rldm_calib_clock_stamp(unsigned long long* slots, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RANGELDM_HIP_H */
