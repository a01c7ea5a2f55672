/* eve_hip.h -- C ABI of libeve_hip.so: the MI355X (gfx950) kernels behind the EVE hot path
 * (EyeNet encoder + RefineNet point-of-gaze refiner, /root/reference/src/models/).
 *
 * The reference has no native layer (it is 100% PyTorch); what each entry point replaces is
 * therefore an ATen op site inside the reference's Python modules, cited per function as
 * `file:line` relative to /root/reference.  A binding (ctypes / cffi / a torch extension) passes
 *   - raw DEVICE pointers (activations NHWC, element type selected by `dtype`),
 *   - explicit shapes,
 *   - the HIP stream to launch on (a hipStream_t passed as void*; NULL = the null stream).
 * No entry point allocates, synchronises the device, or keeps state between calls (the one process-wide
 * object is the read-mostly kernel-selection table, eve_dispatch_config below).  Every function
 * returns 0 on success and a non-zero code on failure; eve_last_error() returns a thread-local
 * message for the last failure.  The Python side (eve_amd/_lib.py) raises RuntimeError from it.
 *
 * Layout conventions
 *   activations   [N][H][W][C]   (NHWC), C a multiple of 4 (f32) / 8 (bf16): 16-byte channel vectors
 *   conv weights  forward : [Cout][KH][KW][Cin]  ("OHWI", K = KH*KW*Cin contiguous per output channel)
 *                 dgrad   : [Cin][KH][KW][Cout]  ("IHWO")
 *   statistics    float [N][C][2] = (mean, rstd) per instance-norm plane
 *   weight / bias / affine gradients are always float and are ACCUMULATED into (caller zeroes them)
 */
#ifndef EVE_HIP_H_
#define EVE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVE_ABI_VERSION 10

typedef void* eve_stream_t; /* hipStream_t */

enum { EVE_DT_F32 = 0, EVE_DT_BF16 = 1, EVE_DT_F16 = 2 };   /* F16: IEEE half, the second 16-bit instantiation */
enum {
    EVE_ACT_NONE = 0,
    EVE_ACT_RELU = 1,    /* nn.ReLU          (eye_net trunk; refine_net.py:38 encoder blocks)   */
    EVE_ACT_LEAKY = 2,   /* nn.LeakyReLU(.01) (refine_net.py:108,113,221)                        */
    EVE_ACT_SELU = 3,    /* nn.SELU          (eye_net.py:54,77,83,89)                            */
    EVE_ACT_TANH = 4,    /* nn.Tanh / torch.tanh (eye_net.py:85; common.py:351,380,413)          */
    EVE_ACT_SIGMOID = 5  /* nn.Sigmoid / torch.sigmoid (refine_net.py:223; common.py:377-379,410) */
};

int eve_abi_version(void);
/* Symbol (without namespace / signature) of the kernel the last conv / stem call of this thread launched; lets a
 * profiler attribute a timed launch to the row of the same name in a rocprofv3 kernel summary.            */
const char* eve_last_kernel(void);
const char* eve_last_error(void);
/* Kernel selection (ABI v6).  Which kernel an entry point dispatches for a given shape is a pure function of the shape and
 * of THIS structure.  It is filled ONCE, when the library is loaded: built-in defaults, overridden by the EVE_* environment
 * variables named per field (tuning experiments; nothing on a call path reads the environment).  A test harness prints
 * eve_get_dispatch_config() and asserts it equals eve_get_default_dispatch_config() before it makes a parity claim
 * (tests/conftest.py), and forces a variant -- e.g. the four-wave convolution next to the eight-wave one -- with
 * eve_set_dispatch_config(), never through the environment of a running process.                                        */
typedef struct eve_dispatch_config {
    int struct_bytes;              /* sizeof(eve_dispatch_config): ABI guard                                              */
    int conv_impl_v1;              /* EVE_CONV_IMPL=v1      0   first-generation register-staged igemm / wgrad kernels     */
    int conv_tile_big;             /* EVE_CONV_TILE=2       0   256 x 128 per-tap tiles                                    */
    int conv_halo;                 /* EVE_CONV_HALO         1   halo-resident 3x3 kernels (conv_fast.h)                    */
    int conv_ws64;                 /* EVE_CONV_WS64         1   conv3x3_ws64_kernel for 32 x 32 x 64 -> 64                 */
    int conv_wg8;                  /* EVE_CONV_WG8          1   eight-wave kernels (0 off, 3: not for 16 x 16 x 128)       */
    int conv_wg8_min_tiles;        /* EVE_CONV_WG8_MIN_TILES 224  fewer tiles: four-wave kernels                          */
    int conv_wg8_s2_min_tiles;     /* EVE_CONV_WG8_S2_MIN_TILES 48  the same floor for the stride-2 forward / data grad.  */
    int halo_persist;              /* EVE_HALO_PERSIST      1   persistent tile stream for <= 64 input channels            */
    int wgrad_target_wgs;          /* EVE_WGRAD_TARGET_WGS  0   (0: 256 x resident workgroups per CU)                      */
    int wgrad_min_rows;            /* EVE_WGRAD_MIN_ROWS    1536 shortest pixel range of a weight-gradient split          */
    int wgrad_halo;                /* EVE_WGRAD_HALO        1   band-resident weight gradients (wgrad_halo.h)              */
    int wgrad_wg8;                 /* EVE_WGRAD_WG8         1   256 x 256 eight-wave weight gradient (wgrad_wg8.h)         */
    int wg64_th, wg64_nreg;        /* EVE_WG64_TH / _NREG   0 0 (0: largest band / region count that fits LDS)             */
    int wg64_fixed;                /* EVE_WG64_FIXED        1   unrolled wgrad_halo64_kernel for 32-wide planes            */
    int in_split;                  /* EVE_IN_SPLIT          1   64 Ki-element InstanceNorm planes as two channel halves    */
    int in_min_threads;            /* EVE_IN_MIN_THREADS    512                                                            */
    int in_stats_one_pass;         /* EVE_IN_STATS_ONE_PASS 1   shifted-moment statistics for planes beyond L2             */
    int stem_split;                /* EVE_STEM_SPLIT        1   two waves per image in the fused stem at small batches     */
    int in_trunk_kernels;          /* EVE_IN_TRUNK          1   branch-free InstanceNorm kernels for the ResNet trunk's cases */
    int stem_fused_wgrad;          /* EVE_STEM_FUSED_WGRAD  1   stem backward + weight gradient in one launch (eve_stem_bwd_wgrad) */
    int stem_fwd_pairs;            /* EVE_STEM_FWD_PAIRS    1   fused stem forward with two waves per image (32 channels each)       */
    int conv1x1_stream;            /* EVE_CONV1X1_STREAM    1   1x1 convolutions between 16..128 channels on the streaming kernel (no LDS)    */
    int conv3x3_stream;            /* EVE_CONV3X3_STREAM    1   3x3 / stride 1 between 16..64 channels on 64 / 128-wide images: row-streaming kernel */
    int in_big_planes;             /* EVE_IN_BIG_PLANES     1   register-resident InstanceNorm (no affine) for planes beyond 8 192 vectors, dealt by channels */
    int cgru_seq_max_b;            /* EVE_CGRU_SEQ_MAX_B    384 16-bit conv-GRU clip scans: one sequence per workgroup (cgru_scan1.hip) up to this many sequences, three per workgroup (cgru_scan.hip) beyond */
    int cgru_scan;                 /* EVE_CGRU_SCAN         1   conv-RNN bottleneck as ONE clip-long launch per direction (0: per-frame launches; bit 1 cleared = 2: forward scan only, per-frame backward) */
    int small_linear;              /* EVE_SMALL_LINEAR      1   float32 nn.Linear of the tail on the small-tile FMA kernels (linear_small.hip)  */
    int tail_loss_node;            /* EVE_TAIL_LOSS_NODE    1   EyeNet tail + losses as one autograd node (ops.EyeTailLossFn)                   */
    int bucket_elems;              /* EVE_BUCKET_ELEMS      4194304  floats per data-parallel gradient bucket (parallel.GradSync)              */
    int gate_wait_polls;           /* EVE_GATE_WAIT_POLLS   1<<21    polls (64 x 64 clocks apart) before a stream gate gives up and poisons the step */
    long long wgrad_halo_min_m;    /* EVE_WGRAD_HALO_MIN_M  1<<20 pixels from which the band-resident weight gradient runs */
} eve_dispatch_config;
int eve_get_dispatch_config(eve_dispatch_config* out);           /* what the entry points use now                         */
int eve_get_default_dispatch_config(eve_dispatch_config* out);   /* the built-in defaults (environment ignored)           */
int eve_set_dispatch_config(const eve_dispatch_config* cfg);     /* explicit override (tests, tuning); struct_bytes checked */
/* Device scratch is CALLER-OWNED and passed PER CALL (ABI v6; v5 kept one process-global pointer): the entry points that can
 * use scratch take `workspace` (16-byte aligned device pointer, or NULL) and `workspace_bytes`, use it on the stream of that
 * call only, and fall back to their scratch-free form when it is missing or too small.  Users: the split-K weight gradient of
 * the 256..512-channel layers (per-split partial filters with plain stores + a summing launch instead of 16 M float atomics)
 * and the data gradient of the stride-2 3x3 layers (filters re-packed per call for conv3x3_wg8_kernel<.., NT>).           */

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA.  Replaces every nn.Conv2d / nn.Linear forward on the path:
 * torchvision ResNet convs built at src/models/eye_net.py:48-50 and run at :106; nn.Linear at
 * eye_net.py:51-56,81-92; nn.Conv2d at src/models/refine_net.py:48,52,61,214,217,221-222 and
 * src/models/common.py:338,362,395-398.  A Linear is the KH=KW=1, IH=IW=1 case.
 * ------------------------------------------------------------------------------------------------ */
typedef struct eve_conv_desc {
    int dtype;               /* EVE_DT_* of x / w / y                                   */
    int N, IH, IW, Cin;      /* x  : [N][IH][IW][Cin]                                   */
    int OH, OW, Cout;        /* y  : [N][OH][OW][Cout]                                  */
    int KH, KW, stride, pad; /* cross-correlation, zero padding                         */
} eve_conv_desc;

/* y = act(conv(x', w) + bias),  x' = pro_act(x * scale[n,c] + shift[n,c]) when in_scale_shift != NULL
 * (float [N][Cin][2]; padding stays zero AFTER the transform), else x' = x.  bias may be NULL.
 * epi_act = EVE_ACT_* [| EVE_EPI_ACCUMULATE]: with the flag the result is ADDED to what y holds (y += act(...)) in the
 * kernel epilogue -- `layers(x) + skip_layer(x)` of RefineNet's BasicBlock (refine_net.py:64-67) is the 1x1 skip
 * convolution accumulating into the 3x3 branch's output, no add launch.                                              */
#define EVE_EPI_ACCUMULATE 0x100
int eve_conv2d_fwd(const eve_conv_desc* d, const void* x, const void* w_ohwi, const float* bias,
                   int epi_act, const float* in_scale_shift, int pro_act, void* y,
                   eve_stream_t stream);
/* The same, and the InstanceNorm2d statistics of the output -- mean_rstd [N][Cout][2] = (mean, 1 / sqrt(biased variance + eps)) per
 * plane, taken on the stored (rounded) values -- from the convolution's own epilogue when the dispatched kernel walks whole
 * images (the row-streaming 3x3 kernel; ABI v7).  *stats_written = 1 if filled, 0 if the shape took another kernel (run
 * eve_instnorm_stats then).  refine_net.py:45-53: every convolution of a pre-activation block is followed by InstanceNorm2d. */
int eve_conv2d_fwd_stats(const eve_conv_desc* d, const void* x, const void* w_ohwi, const float* bias, int epi_act, void* y,
                         float* mean_rstd, float eps, int* stats_written, eve_stream_t stream);
/* dx = conv_transpose(dy, w): gradient w.r.t. the conv INPUT.  w_ihwo is [Cin][KH][KW][Cout].    */
int eve_conv2d_dgrad(const eve_conv_desc* d, const void* dy, const void* w_ihwo, void* dx,
                     void* workspace, unsigned long long workspace_bytes, eve_stream_t stream);
/* dx += the same data gradient (dx already holds the gradient of the block's other branch: autograd's add at
 * the residual fork of torchvision BasicBlock, fused into the epilogue).                              */
int eve_conv2d_dgrad_acc(const eve_conv_desc* d, const void* dy, const void* w_ihwo, void* dx,
                     eve_stream_t stream);
/* dw[Cout][KH][KW][Cin] (float, accumulated) += sum_m dy[m][co] * x'[m][(kh,kw,ci)]              */
int eve_conv2d_wgrad(const eve_conv_desc* d, const void* x, const void* dy,
                     const float* in_scale_shift, int pro_act, float* dw_ohwi,
                     void* workspace, unsigned long long workspace_bytes, eve_stream_t stream);
/* ... together with db[Cout] (float, accumulated) += sum_m dy[m][co] in the same pass over dy: autograd of a biased
 * nn.Conv2d (refine_net.py's U-Net and conv-GRU convolutions all carry one).                                   */
int eve_conv2d_wgrad_bias(const eve_conv_desc* d, const void* x, const void* dy, float* dw_ohwi, float* db,
                          void* workspace, unsigned long long workspace_bytes, eve_stream_t stream);
/* ResNet stem (torchvision ResNet.conv1: 7x7 / stride 2 / pad 3, 3 -> 64, no bias; eye_net.py:48-50), bf16:
 * eve_stem_pack_input writes x_padded [N][IH+6][IW+8][4] bf16 (channels 0..C-1, zero 4th channel and borders)
 * from the float NCHW patch; eve_stem7x7s2_fwd computes y [N][IH/2][IW/2][64] from it and the OHWI weights
 * with Cin padded to 8 ([64][7][7][8] bf16).  IH even, IW a multiple of 128.                           */
int eve_stem_pack_input(int dtype /* EVE_DT_BF16 | EVE_DT_F16 */, int N, int C, int IH, int IW, const float* src_nchw, void* x_padded,
                        eve_stream_t stream);
int eve_stem7x7s2_fwd(int dtype, int N, int IH, int IW, const void* x_padded, const void* w_ohwi8, void* y,
                      eve_stream_t stream);
/* Decoded uint8 frames on the device (datasources/eve_sequences.py:196-211 does this on the host and ships floats):
 * dst[n][c][y][x] (float) = src[n][y][x][c] * scale (+ shift): eye patches scale 2/255, shift -1; screen frames
 * scale 1/255, no shift.  One rounded multiply and one rounded add, bit-identical to the numpy expressions.      */
int eve_frames_u8_to_nchw(long long N, int H, int W, int C, const uint8_t* src_nhwc, float scale, float shift,
                          int has_shift, float* dst_nchw, eve_stream_t stream);
/* ... and the same values straight into the stem kernels' packed input x_padded [N][IH+6][IW+8][4] bf16 (what
 * eve_stem_pack_input builds from the float NCHW tensor).                                                      */
int eve_frames_u8_to_stem(int dtype, long long N, int C, int IH, int IW, const uint8_t* src_nhwc, float scale, float shift,
                          void* x_padded, eve_stream_t stream);
/* The whole stem in one launch: conv1 -> bn1 (InstanceNorm2d, no affine) -> relu -> maxpool 3x3/2 pad 1
 * (torchvision ResNet._forward_impl as built by eye_net.py:48-50).  The 64-channel convolution output is never
 * written: y_pool [N][IH/4][IW/4][64] bf16, idx (window position kh*3+kw of the arg-max, same shape, uint8) and
 * mean_rstd [N][64][2] are the only outputs.  IW == 128, IH a multiple of 4.                              */
int eve_stem_fwd_fused(int dtype, int N, int IH, int IW, const void* x_padded, const void* w_ohwi8, float eps,
                       void* y_pool, uint8_t* idx, float* mean_rstd, eve_stream_t stream);
/* Its backward up to the convolution output: dx [N][IH/2][IW/2][64] bf16 = d(conv1 out) from dy_pool, recomputing
 * the convolution from x_padded (autograd of bn1/relu/maxpool in eye_net.py:106); feed dx to eve_conv2d_wgrad. */
/* Round 4: backward AND weight gradient of the fused stem in ONE launch -- dw [64][7][8][4] float (accumulated; filter column
 * 7 and channel 3 do not exist) straight from d(y_pool): every recomputed convolution row's d(conv1 out) stays in LDS and is
 * multiplied there with the input rows the recomputation has staged anyway; the 1 GB d(conv1 out) tensor of
 * eve_stem_bwd_dx + eve_stem_wgrad is never written or read.  Same arithmetic per channel (float summation order aside).    */
int eve_stem_bwd_wgrad(int dtype, int N, int IH, int IW, const void* x_padded, const void* w_ohwi8, const float* mean_rstd,
                       const void* dy_pool, const void* dy_pool2, const void* y_pool, const uint8_t* idx, float* dw,
                       void* workspace, unsigned long long workspace_bytes, eve_stream_t stream);
/* `workspace`: at least eve_stem_bwd_wgrad_workspace(dtype, N, IH) bytes, 16-byte aligned, caller-owned, used on `stream` by this
 * call only: the two InstanceNorm plane sums and the masked, summed gradient are formed by a streaming pass of their own
 * (stem_grad_prep_kernel) and the fused kernel reads one pooled tensor and the arg-max codes, once.  REQUIRED since ABI v9 (v8
 * fell back to a one-launch form that read the three pooled tensors twice and spilled registers): NULL / too small is an error. */
unsigned long long eve_stem_bwd_wgrad_workspace(int dtype, int N, int IH);
/* ... and the stem's weight gradient from the same packed patches: dw [64][7][8][4] float (accumulated; filter
 * column 7 and channel 3 do not exist and are ignored by the caller).  Replaces autograd of conv1 (eye_net.py:106). */
int eve_stem_wgrad(int dtype, int N, int IH, int IW, const void* x_padded, const void* dconv, float* dw, eve_stream_t stream);
int eve_stem_bwd_dx(int dtype, int N, int IH, int IW, const void* x_padded, const void* w_ohwi8, const float* mean_rstd,
                    const void* dy_pool, const void* dy_pool2 /* nullable second summand */, const void* y_pool, const uint8_t* idx, void* dx, eve_stream_t stream);
/* Small float32 linear layers (nn.Linear of the EyeNet tail, eye_net.py:52-90: fc, fc_common, GRU input
 * projection, gaze / pupil heads) with M rows and K, N <= 4096:
 *   fwd:   y[M][N]   = act(x[M][K] . w_in_out[K][N] + bias)
 *   dgrad: dx[M][K]  = (dy * act'(y))[M][N] . w_out_in[N][K]          (y may be NULL when act is NONE)
 *   wgrad: dw[N][K] += (dy * act'(y))^T . x ;  db[N] += column sums    (db nullable; float atomics)          */
int eve_linear_fwd(int M, int K, int N, const float* x, const float* w_in_out, const float* bias, int act,
                   float* y, eve_stream_t stream);
int eve_linear_dgrad(int M, int K, int N, const float* dy, const float* y, int act, const float* w_out_in,
                     float* dx, eve_stream_t stream);
int eve_linear_wgrad(int M, int K, int N, const float* dy, const float* y, int act, const float* x,
                     float* dw_out_in, float* db, eve_stream_t stream);
/* ... with row strides (a column range of a wider matrix on either side), a bias shorter than N (padded heads) and an
 * accumulating data gradient (dx += ...): what lets the tail run without cat / pad / slice / add launches (round 4).            */
int eve_linear_fwd_ex(int M, int K, int N, const float* x, int ldx, const float* w_in_out, const float* bias, int n_bias,
                      int act, float* y, int ldy, eve_stream_t stream);
int eve_linear_dgrad_ex(int M, int K, int N, const float* dy, int lddy, const float* y, int act, const float* w_out_in,
                        float* dx, int lddx, int accumulate, eve_stream_t stream);
/* gaze [M][2] = pi/2 * g2[:, :2] and pupil [M] = p2[:, 0] of the heads' 4-wide last layers (eye_net.py:139-146), and back:
 * d_g2 / d_p2 [2*BT][4] from the loss kernel's per-side unit gradients (rows: left clips, then right), scaled by
 * coeff * *g_full (device scalar, NULL = 1).                                                                                  */
/* cat[m][col .. col+3] = (h[m][0], h[m][1], 0, 0), rows 0 .. BT-1 from h_left [BT][2], BT .. 2BT-1 from h_right (eye_net.py:113-114's
 * torch.cat of fc's output and the head pose, written behind the 128 columns eve_linear_fwd_ex filled; ld, col multiples of 4).   */
int eve_tail_head_pose(int BT, const float* h_left, const float* h_right, float* cat, int ld, int col, eve_stream_t stream);
int eve_tail_outputs_fwd(int M, const float* g2, const float* p2, float* gaze, float* pupil, eve_stream_t stream);
int eve_tail_outputs_bwd(int BT, const float* dg_l, const float* dg_r, const float* dp_l, const float* dp_r, const float* g_full,
                         float coeff_ang, float coeff_l1, float* d_g2, float* d_p2, eve_stream_t stream);
/* The EyeNet train-step losses and their gradients in one launch (losses/angular.py:33-38, losses/l1.py,
 * losses/base_loss_with_validity.py:64-73; weighted sum of eve.py:234-265).  Every pointer argument is an array of
 * two device pointers {left, right}: g_pred/g_tgt [B][T][2] (pitch, yaw), p_pred/p_tgt [B][T], validity bytes [B][T].
 * terms[5] (zeroed by the caller) += {ang_l, l1_l, ang_r, l1_r, coeff_ang*(ang_l+ang_r) + coeff_l1*(l1_l+l1_r)};
 * dg/dp receive d(term of that side)/d(prediction).                                                     */
int eve_eye_losses(int B, int T, const float* const* g_pred, const float* const* g_tgt, const uint8_t* const* g_val,
                   const float* const* p_pred, const float* const* p_tgt, const uint8_t* const* p_val,
                   float coeff_ang, float coeff_l1, float* terms, float* const* dg, float* const* dp,
                   eve_stream_t stream);
/* Every validity-masked term of EVE.calculate_losses_and_metrics over [B][T][D <= 3] predictions (eve.py:286-439: the gaze /
 * PoG / pupil losses and metrics; src/losses/{mse,euclidean,l1,angular}.py with the clip reduction of
 * base_loss_with_validity.py:64-73) in ONE launch (ABI v7).  terms[i]: pred, tgt [B][T][D] float, valid [B][T] bytes, kind
 * 0 MSE | 1 Euclidean distance | 2 L1 | 3 angular error in degrees (D = 2: pitch, yaw), dpred = NULL or [B][T][D] <-
 * d term_i / d pred for a unit upstream gradient (kinds 0, 2, 3).  out[i] = term i.  n <= EVE_VEC_TERMS_MAX.                */
#define EVE_VEC_TERMS_MAX 32
typedef struct eve_vec_term { const float* pred; const float* tgt; const unsigned char* valid; float* dpred; int D; int kind; } eve_vec_term;
int eve_vector_terms(const eve_vec_term* terms, int n, int B, int T, float* out, eve_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * Gaze geometry, heat-maps and soft-argmax: the per-frame glue of EVE.forward either side of the two networks
 * (eve.py:114-166, 545-601).  Flat batches of N = B*T frames, float32, row-major small matrices.
 * ------------------------------------------------------------------------------------------------ */
/* to_screen_coordinates (models/common.py:157-187), optionally preceded by apply_offset_augmentation (:190-229) when
 * kappa/head_R are given.  g [N][2] (pitch, yaw), origin [N][3] mm (camera frame), R [N][3][3], inv_cam [N][4][4],
 * ppm [N][2] pixels per mm.  Outputs: g_out [N][2] (the augmented angles; optional without kappa), pog_mm [N][2],
 * pog_px [N][2] clamped to [0, screen], jac [N][6][2] = d(g_out, pog_mm, pog_px) / d(pitch, yaw) for the backward. */
int eve_gaze_to_pog(long long N, const float* g, const float* origin, const float* R, const float* inv_cam,
                    const float* ppm, const float* head_R, const float* kappa, float screen_w, float screen_h,
                    float* g_out, float* pog_mm, float* pog_px, float* jac, eve_stream_t stream);
/* dg [N][2] = jac^T (dg_out, dmm, dpx); any of the three incoming gradients may be null.                       */
int eve_gaze_to_pog_bwd(long long N, const float* jac, const float* dg_out, const float* dmm, const float* dpx,
                        float* dg, eve_stream_t stream);
/* calculate_combined_gaze_direction (common.py:136-154): g [N][2] from the mean eye origin to a screen point.     */
int eve_combined_gaze(long long N, const float* origin, const float* pog_mm, const float* R, const float* cam,
                      float* g, eve_stream_t stream);
/* batch_make_heatmaps (common.py:236-255): out [N][H][W] = 1e-8 + exp(-|p - c|^2 / (2 sigma^2)), c = centre_px * (W/screen_w,
 * H/screen_h); multiplied by validity[n] (bytes) when given (the label maps of eve.py:503-520).                  */
int eve_make_heatmaps(long long N, int H, int W, const float* centres_px, const uint8_t* validity, float sigma,
                      float screen_w, float screen_h, float* out, eve_stream_t stream);
int eve_make_heatmaps_bwd(long long N, int H, int W, const float* centres_px, float sigma, float screen_w, float screen_h,
                          const float* dout, float* dcentres, eve_stream_t stream);
/* soft_argmax (common.py:304-333): pog_px [N][2] = clamp(screen * E_softmax(100 h)[(x/(W-1), y/(H-1))]); stats [N][4] =
 * (lx, ly, max, sum exp) feed the backward.                                                                    */
int eve_soft_argmax_fwd(long long N, int H, int W, const float* heat, float screen_w, float screen_h, float* pog_px,
                        float* stats, eve_stream_t stream);
int eve_soft_argmax_bwd(long long N, int H, int W, const float* heat, const float* stats, const float* dpog,
                        float screen_w, float screen_h, float* dheat, eve_stream_t stream);
/* db[C] (float, accumulated) += sum over the M = N*OH*OW rows of dy[M][C]                         */
int eve_bias_grad(int dtype, long long M, int C, const void* dy, float* db, eve_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Instance normalisation (+ optional affine, residual add, activation).  Replaces
 * nn.InstanceNorm2d inside the ResNet (norm_layer at eye_net.py:50: eps 1e-5, biased variance, no
 * affine, no running stats) with the following ReLU / residual add of BasicBlock.forward, and the
 * affine InstanceNorm2d + activation pairs of refine_net.py:46-47,50-51,59-60,215-216.
 * ------------------------------------------------------------------------------------------------ */
/* mean_rstd[n][c] = (mean, 1/sqrt(var_biased + eps)) over the HW plane                            */
int eve_instnorm_stats(int dtype, int N, int HW, int C, const void* x, float eps, float* mean_rstd,
                       eve_stream_t stream);
/* y = act(gamma[c] * (x - mean) * rstd + beta[c] + res);  gamma/beta/res may be NULL              */
int eve_instnorm_act_fwd(int dtype, int N, int HW, int C, const void* x, const float* mean_rstd,
                         const float* gamma, const float* beta, const void* res, int act, void* y,
                         eve_stream_t stream);
/* g = dy * act'(y);  dx = gamma*rstd*(g - mean_hw(g) - xhat*mean_hw(g*xhat));  dres = g (if != NULL);
 * sums[n][c] = (sum_hw g, sum_hw g*xhat)  (reduce over n for dbeta / dgamma).
 * y may be NULL when there was no residual: act'(.) is then recomputed from x with the forward's own scale / shift
 * (beta is needed for that when gamma is given) -- one tensor less to read in both passes.                    */
int eve_instnorm_act_bwd(int dtype, int N, int HW, int C, const void* dy, const void* y,
                         const void* x, const float* mean_rstd, const float* gamma, const float* beta, int act,
                         void* dx, void* dres, float* sums, const void* dx_add, eve_stream_t stream);
/* dx_add (nullable, like x): a second gradient of the normalised tensor's INPUT, added to dx in the epilogue -- the identity-skip
 * blocks of RefineNet (refine_net.py:35-67: x feeds `layers` and the block's final add) then need no gradient-fork add launch. */

/* Two affine + activation heads over ONE normalised input: RefineNet's pre-activation BasicBlock feeds the block input to
 * `layers` (refine_net.py:46-47) and to `skip_layer` (:59-60), each starting with InstanceNorm2d(affine) -> activation.
 * y_h = act(gamma_h * (x - mean) * rstd + beta_h);  x is read once.  y_a / y_b are [N][HW][ldy] tensors (16-byte aligned).
 * The input may be the channel-concatenation of TWO sources (the decoder's torch.cat([upsampled, encoder_output]),
 * refine_net.py:125-126): x [N][HW][C] owns channels [0, C) of the heads, x2 [N][HW][C2] (nullable) channels [C, C + C2);
 * InstanceNorm statistics are per channel, so each source is normalised on its own (mean_rstd / mean_rstd2 from
 * eve_instnorm_stats) straight into the concatenated layout and the concatenation is never materialised un-normalised.
 * gamma_h / beta_h: C + C2 floats.  ldy >= C + C2.  y_b / gamma_b / beta_b may be NULL (one head).                     */
int eve_instnorm_act2_fwd(int dtype, int N, int HW, int C, const void* x, const float* mean_rstd,
                          const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                          int act, void* y_a, void* y_b, int ldy, int C2, const void* x2, const float* mean_rstd2,
                          eve_stream_t stream);
/* g_h = dy_h * act'(y_h) (y_h recomputed from x);  dx = sum_h gamma_h*rstd*(g_h - mean_hw(g_h) - xhat*mean_hw(g_h*xhat)):
 * the gradient sum of the fork is formed in registers.  dy_a / dy_b: [N][HW][lddy] gradients of the heads; dx / dx2: dense
 * gradients of the sources.  sums_h[n][c] = (sum_hw g_h, sum_hw g_h*xhat) for the C + C2 channels.
 * dy_b (with gamma_b, beta_b, sums_b) and the second source (x2, mean_rstd2, dx2) may be NULL.                         */
int eve_instnorm_act2_bwd(int dtype, int N, int HW, int C, const void* dy_a, const void* dy_b, int lddy,
                          const void* x, const float* mean_rstd, const float* gamma_a, const float* beta_a,
                          const float* gamma_b, const float* beta_b, int act, void* dx, float* sums_a,
                          float* sums_b, int C2, const void* x2, const float* mean_rstd2, void* dx2, eve_stream_t stream);

/* Single-pass variants for planes that fit one workgroup's registers (HW*C <= 64 Ki elements): statistics and
 * apply in one launch (also writes mean_rstd), and the whole backward in one read of dy / y / x.
 * Same arithmetic as the three entry points above.  Return -1 (no error text) when the plane does not fit;
 * the caller then uses the multi-pass entry points.                                                 */
/* sign_mask (nullable, N*HW*C/vec bytes, vec = 16 / sizeof(element)): bit e of byte v = (element e of 16-byte vector v
 * of y is > 0) -- all a ReLU backward needs of y, at 1/16 of its bytes (the trunk's block-output InstanceNorm).   */
int eve_instnorm_fwd_fused(int dtype, int N, int HW, int C, const void* x, const float* gamma,
                           const float* beta, const void* res, int act, float eps, void* y,
                           float* mean_rstd, unsigned char* sign_mask, eve_stream_t stream);
/* dy2 (nullable): a second summand of the incoming gradient, added on load -- the residual fork of a ResNet
 * block delivers d(block input) as two tensors and the sum is never materialised.                      */
/* sign_mask (nullable; act == EVE_ACT_RELU): the forward's mask, read INSTEAD of y.                          */
/* beta (nullable): with gamma and no y, act'(.) is recomputed from x with the forward's own scale / shift
 * (act(x * rstd*gamma + (beta - mean*rstd*gamma))) -- no residual: one tensor less to read and to keep.       */
int eve_instnorm_bwd_fused(int dtype, int N, int HW, int C, const void* dy, const void* dy2, const void* y,
                           const void* x, const float* mean_rstd, const float* gamma, const float* beta, int act,
                           void* dx, void* dres, float* sums, const unsigned char* sign_mask, const void* dx_add, eve_stream_t stream);

/* out[j] = sum_r in[r][j] (float32, fixed summation order): the batch reduction of the per-plane partials `sums` above into
 * d(beta) / d(gamma) of an affine InstanceNorm2d (refine_net.py:46,50,59,215).                                          */
int eve_sum_rows(int rows, int cols, const float* in, float* out, eve_stream_t stream);
/* in [rows][C][2] (per-plane (d beta, d gamma) partials of an affine InstanceNorm backward) -> out0[c] += sum_r in[r][c][0],
 * out1[c] += sum_r in[r][c][1], fixed order (ABI v7): the two parameter gradients land in the caller's gradient buffer.  */
int eve_sum_rows_pairs(int rows, int C, const float* in, float* out0, float* out1, eve_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise activation gradient: dx = dy * act'(y)  (for Linear/conv epilogue activations).
 * ------------------------------------------------------------------------------------------------ */
int eve_act_bwd(int dtype, long long n, const void* dy, const void* y, int act, void* dx,
                eve_stream_t stream);
/* out = a + b (same dtype), used for gradient fan-in of the residual / skip branches.             */
int eve_add(int dtype, long long n, const void* a, const void* b, void* out, eve_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pooling / resampling.  nn.MaxPool2d(3,2,1) and AdaptiveAvgPool2d(1) of the torchvision ResNet
 * (eye_net.py:48,106); nn.AdaptiveMaxPool2d (refine_net.py:93) and nn.Upsample(bilinear,
 * align_corners=False) (refine_net.py:101).
 * ------------------------------------------------------------------------------------------------ */
/* 3x3 / stride 2 / pad 1 max-pool; idx[n][oh][ow][c] (uint8) = kh*3+kw of the first maximum       */
int eve_maxpool3x3s2_fwd(int dtype, int N, int IH, int IW, int C, const void* x, void* y,
                         uint8_t* idx, eve_stream_t stream);
int eve_maxpool3x3s2_bwd(int dtype, int N, int IH, int IW, int C, const void* dy,
                         const uint8_t* idx, void* dx, eve_stream_t stream);
/* ResNet stem tail fused: y = maxpool3x3s2(relu(IN(x))) with no affine, given mean_rstd of x (statistics
 * from eve_instnorm_stats); the normalised full-resolution tensor is never written.  idx as above.
 * Backward: dx = d(loss)/dx through pool, ReLU and the instance norm, from the pooled tensors and x.  */
int eve_in_relu_maxpool_fwd(int dtype, int N, int IH, int IW, int C, const void* x,
                            const float* mean_rstd, void* y, uint8_t* idx, eve_stream_t stream);
int eve_in_relu_maxpool_bwd(int dtype, int N, int IH, int IW, int C, const void* dy_pool,
                            const void* y_pool, const uint8_t* idx, const void* x,
                            const float* mean_rstd, void* dx, eve_stream_t stream);
/* mean over the HW plane: y[N][C]; and its gradient dx[n][hw][c] = dy[n][c] / HW                  */
int eve_avgpool_fwd(int dtype, int N, int HW, int C, const void* x, void* y, eve_stream_t stream);
int eve_avgpool_bwd(int dtype, int N, int HW, int C, const void* dy, void* dx, eve_stream_t stream);
/* ABI v10: the same with the pooled [N][C] side in FLOAT32 memory (values of format `dtype`, widened): EyeNet's trunk -> tail
 * hand-over (eye_net.py:52-56: avgpool -> flatten -> fc) without the two cast launches; bit-identical to pool + cast.          */
int eve_avgpool_fwd_f32(int dtype, int N, int HW, int C, const void* x, float* y, eve_stream_t stream);
int eve_avgpool_bwd_f32(int dtype, int N, int HW, int C, const float* dy, void* dx, eve_stream_t stream);
/* adaptive max-pool, window i = [floor(i*I/O), ceil((i+1)*I/O)); idx = flat ih*IW+iw (int32)       */
int eve_adaptive_maxpool_fwd(int dtype, int N, int IH, int IW, int OH, int OW, int C, const void* x,
                             void* y, int32_t* idx, eve_stream_t stream);
/* add (nullable, [N][IH][IW][C]): a second gradient of the pooled tensor's INPUT, added in the epilogue -- the encoder output
 * feeds the pool AND the decoder's skip connection (refine_net.py:103-126), autograd's fork add is then not launched.            */
int eve_adaptive_maxpool_bwd(int dtype, int N, int IH, int IW, int OH, int OW, int C,
                             const void* dy, const int32_t* idx, const void* add, void* dx, eve_stream_t stream);
/* bilinear resize, align_corners=False, and its adjoint                                            */
int eve_bilinear_fwd(int dtype, int N, int IH, int IW, int OH, int OW, int C, const void* x, void* y,
                     eve_stream_t stream);
int eve_bilinear_bwd(int dtype, int N, int IH, int IW, int OH, int OW, int C, const void* dy,
                     void* dx, eve_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Layout / dtype plumbing at the module boundary (the reference hands NCHW float tensors:
 * eye_net.py:100-103, refine_net.py:239-248).
 * ------------------------------------------------------------------------------------------------ */
/* dst[n][h][w][c < Cpad] = c < C ? src[n][c][h][w] : 0                                             */
int eve_nchw_to_nhwc(int dtype_dst, int N, int C, int H, int W, int Cpad, const float* src_nchw,
                     void* dst_nhwc, eve_stream_t stream);
int eve_nhwc_to_nchw(int dtype_src, int N, int C, int H, int W, int Cpad, const void* src_nhwc,
                     float* dst_nchw, eve_stream_t stream);
/* float <-> dtype casts of flat buffers (weights to the compute dtype)                             */
int eve_cast(int dtype_src, int dtype_dst, long long n, const void* src, void* dst,
             eve_stream_t stream);
/* OHWI float master weights -> OHWI and IHWO copies in the compute dtype (one pass)                */
int eve_pack_weights(int dtype_dst, int Cout, int taps, int Cin, const float* w_ohwi, void* dst_ohwi,
                     void* dst_ihwo, eve_stream_t stream);
/* The same for up to EVE_PACK_BATCH_MAX weights in one launch (a model's conv weights after an optimiser step). */
#define EVE_PACK_BATCH_MAX 48
typedef struct eve_pack_item {
    const float* w_ohwi; void* dst_ohwi; void* dst_ihwo;      /* either destination may be NULL */
    int Cout, taps, Cin;
    int src_Cout, src_Cin;   /* ABI v10: the SOURCE is [src_Cout][taps][src_Cin] (0 = Cout / Cin); the destinations' extra output /
                              * input channels are written as zeros -- conv1's 3 -> 4/8 input channels, the 130 -> 132 wide
                              * fc_common.0, the 2- and 1-wide heads padded to 4 were a fill + a copy launch each, every step */
} eve_pack_item;
int eve_pack_weights_batch(int dtype_dst, int count, const eve_pack_item* items /* host array */, eve_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Recurrent cells.
 * GRU scan: torch.nn.GRUCell applied over T steps (eye_net.py:69,125; state hand-over :116-133).
 *   gi  : float [S][T][3H]  = W_ih x_t + b_ih for every step (batched GEMM done by the caller)
 *   whh : float [3H][H] (the forward scan takes its transpose whh_t [H][3H]); bhh float [3H];
 *   h0 float [S][H] or NULL (= zeros, eye_net.py:120-122)
 *   hs  : float [S][T][H] all hidden states;  work: float [S][T][3H] (r, z, n) + [S][T][H] (hn) saved
 * ------------------------------------------------------------------------------------------------ */
int eve_gru_scan_fwd(int S, int T, int H, const float* gi, const float* whh_t, const float* bhh,
                     const float* h0, float* hs, float* gates, float* hn_pre, eve_stream_t stream);
/* given dhs [S][T][H] (gradient on every h_t): dgi [S][T][3H], dgh [S][T][3H] (for dW_hh/db_hh by
 * GEMM), dh0 [S][H] (may be NULL)                                                                  */
int eve_gru_scan_bwd(int S, int T, int H, const float* dhs, const float* whh, const float* h0,
                     const float* hs, const float* gates, const float* hn_pre, float* dgi, float* dgh,
                     float* dh0, eve_stream_t stream);
/* nn.RNNCell (tanh) and nn.LSTMCell over T -- the other recurrent variants of eye_net.py:60-67.  gi = W_ih x + b_ih
 * for all steps ([S][T][G*H], G = 1 / 4 gate blocks in torch order i,f,g,o), whh_t = W_hh^T [H][G*H], whh = W_hh.
 * The backward returns dpre = d(pre-activation) [S][T][G*H] (the gradient of gi, and the operand of dW_hh, db_hh). */
int eve_rnn_scan_fwd(int S, int T, int H, const float* gi, const float* whh_t, const float* bhh, const float* h0,
                     float* hs, eve_stream_t stream);
int eve_rnn_scan_bwd(int S, int T, int H, const float* dhs, const float* whh, const float* hs, float* dpre,
                     float* dh0, eve_stream_t stream);
int eve_lstm_scan_fwd(int S, int T, int H, const float* gi, const float* whh_t, const float* bhh, const float* h0,
                      const float* c0, float* hs, float* cs, float* gates, eve_stream_t stream);
int eve_lstm_scan_bwd(int S, int T, int H, const float* dhs, const float* dcs, const float* whh, const float* c0,
                      const float* hs, const float* cs, const float* gates, float* dpre, float* dh0, float* dc0,
                      eve_stream_t stream);

/* conv-GRU gate math of CGRUCell.forward (common.py:409-414), NHWC, C = hidden size:
 *   step 1: (r, u) = sigmoid(g1[..., 0:C], g1[..., C:2C]);  rh = r * h
 *   step 2: o = tanh(g2);  h' = (1 - u) * o + u * h                                                */
/* CGRUCell over all T frames of a clip in ONE persistent launch (the 5x8x64 bottleneck of refine_net.py:132-176):
 * hidden state resident in LDS, both gate GEMMs on MFMA with the filter banks streamed through an LDS-DMA ring,
 * sigmoid / tanh / blend as epilogues.  dtype bf16 / f16: cgru_scan.hip (three sequences per workgroup, 16-bit MFMA);
 * dtype f32 (ABI v7): cell_scan_f32.hip (one workgroup per sequence, float state, v_mfma_f32_16x16x4_f32 -- the parity
 * mode's 1e-4 rad runs on one launch per clip as well).  xs [B][T][5][8][64]; h0 [B][5][8][64] or NULL; w1 = gates_1 OHWI
 * [128][3][3][128] (inputs x|h), w2 = gate_2 OHWI [64][3][3][128] (inputs r*h|x); outputs: hs [B][T][5][8][64]
 * (state, caller's order) and, TIME-major [T][B][5][8][.] for the frame-reversed backward: hs_tm, ru (both sigmoid
 * gates, 128 ch), rh (r*h), og (tanh gate).                                                               */
int eve_cgru_scan_fwd(int dtype, int B, int T, const void* xs, const void* h0, const void* w1, const float* b1, const void* w2,
                      const float* b2, void* hs, void* hs_tm, void* ru, void* rh, void* og, eve_stream_t stream);
/* Backward of eve_cgru_scan_fwd in ONE persistent launch (bf16 / f16 / f32; common.py:400-415 differentiated): frames last to first, the
 * gradient into the previous hidden state carried in registers.  Time-major inputs [T][B][5][8][.]: dhs_tm (d hs), and the
 * forward's ru / og / hs_tm; h0 or NULL; w1t = gates_1 bank IHWO [128][3][3][128], w2t = gate_2 bank IHWO [128][3][3][64].
 * Outputs (time-major): dg1_all [..][128], dg2_all [..][64] (pre-activation gradients: the operands of the batched weight /
 * bias gradients), dxs_tm [..][64]; dh0 [B][5][8][64] or NULL.                                                             */
int eve_cgru_scan_bwd(int dtype, int B, int T, const void* dhs_tm, const void* ru, const void* og, const void* hs_tm, const void* h0,
                      const void* w1t, const void* w2t, void* dg1_all, void* dg2_all, void* dxs_tm, void* dh0,
                      eve_stream_t stream);
/* CRNNCell (common.py:331-352: h_t = tanh(conv3x3([x_t | h_{t-1}]) + b)) over all T frames of a clip in ONE launch, float32
 * (ABI v7; 16-bit callers convert the 5x8x64 bottleneck tensors).  xs [B][T][5][8][64]; h0 [B][5][8][64] or NULL; w OHWI
 * [64][3][3][128]; bias [64].  Outputs hs [B][T][5][8][64] and its time-major copy hs_tm [T][B][5][8][64].                 */
int eve_crnn_scan_fwd(int B, int T, const float* xs, const float* h0, const float* w, const float* bias, float* hs,
                      float* hs_tm, eve_stream_t stream);
/* Backward of eve_crnn_scan_fwd, one launch: time-major dhs_tm / hs_tm; wt = the bank IHWO [128][3][3][64].  Outputs
 * (time-major) dpre_all [T][B][5][8][64] = gradient of the pre-activation (operand of the batched weight / bias gradient),
 * dxs_tm; dh0 [B][5][8][64] or NULL.                                                                                       */
int eve_crnn_scan_bwd(int B, int T, const float* dhs_tm, const float* hs_tm, const float* wt, float* dpre_all, float* dxs_tm,
                      float* dh0, eve_stream_t stream);
/* CLSTMCell (common.py:355-385; gates in / forget / out / cell) over all T frames of a clip in ONE launch, float32, forward
 * only -- the reference's Bottleneck stores the (h, c) tuple and never feeds it on (refine_net.py:168-174), so no gradient
 * reaches the cell.  w OHWI [256][3][3][128]; bias [256]; h0 / c0 or NULL.  Outputs hs, cs [B][T][5][8][64].               */
int eve_clstm_scan_fwd(int B, int T, const float* xs, const float* h0, const float* c0, const float* w, const float* bias,
                       float* hs, float* cs, eve_stream_t stream);
int eve_cgru_gates1(int dtype, long long P, int C, const void* g1, const void* h, void* ru, void* rh,
                    eve_stream_t stream);
int eve_cgru_gates2(int dtype, long long P, int C, const void* g2, const void* ru, const void* h,
                    void* o, void* hnew, eve_stream_t stream);
/* backward of step 2: given dh' -> dg2 (pre-tanh, [P][C]); dru [P][2C] = (0, gradient on the
 * post-sigmoid update gate); dh [P][C] = direct path u * dh'                                        */
int eve_cgru_gates2_bwd(int dtype, long long P, int C, const void* dhnew, const void* ru,
                        const void* h, const void* o, void* dg2, void* dru, void* dh,
                        eve_stream_t stream);
/* backward of step 1: given d(rh) [P][C], dru [P][2C] and saved ru, h -> dg1 (pre-sigmoid, [P][2C])
 * and dh [P][C] = d(rh) * r                                                                         */
int eve_cgru_gates1_bwd(int dtype, long long P, int C, const void* drh, const void* dru,
                        const void* ru, const void* h, void* dg1, void* dh, eve_stream_t stream);

/* conv-LSTM gate math of CLSTMCell.forward (common.py:376-385), forward only: gates [P][4C] in the
 * reference's chunk order (in, forget, out, cell); c' = sig(f)*c + sig(i)*tanh(g); h' = sig(o)*tanh(c').
 * The reference never back-propagates through it (refine_net.py:168-174 drops tuple states).         */
int eve_clstm_gates_fwd(int dtype, long long P, int C, const void* gates, const void* c_prev, void* h,
                        void* c, eve_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RefineNet output head and the heat-map losses (csrc/heatmap_loss.hip).
 * ------------------------------------------------------------------------------------------------ */
/* nn.Sigmoid of `final` (src/models/refine_net.py:221-223) evaluated in float: logits = channel 0 of the last 1x1
 * convolution's NHWC output [pixels][Cpad] (compute dtype) -> out float [pixels] (= heatmap_final [N][1][H][W]).   */
int eve_heatmap_head_fwd(int dtype, long long pixels, int Cpad, const void* logits, float* out, eve_stream_t stream);
/* dlogits[pixel][c] = c == 0 ? dy * y * (1 - y) : 0, compute dtype                                                   */
int eve_heatmap_head_bwd(int dtype, long long pixels, int Cpad, const float* dy, const float* y, void* dlogits,
                         eve_stream_t stream);
/* loss_ce_heatmap_* (kind 0: F.binary_cross_entropy per frame, src/losses/cross_entropy.py:27-35) or
 * loss_mse_heatmap_final (kind 1, src/losses/mse.py) with the validity reduction of
 * src/losses/base_loss_with_validity.py:64-73.  pred, gt float [B][T][HW]; validity uint8 [B][T]; outputs:
 * per_map float [B*T] (per-frame means), loss float [1], w float [B*T] = d loss / d per_map.                        */
int eve_heatmap_loss_fwd(int kind, int B, int T, int HW, const float* pred, const float* gt, const uint8_t* validity,
                         float* per_map, float* loss, float* w, eve_stream_t stream);
/* dpred = upstream[0] * w[frame] / HW * d(element loss)/d pred  (upstream: device scalar, no host sync)             */
int eve_heatmap_loss_bwd(int kind, int BT, int HW, const float* pred, const float* gt, const float* w,
                         const float* upstream, float* dpred, eve_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Round 4: the EyeNet tail (src/models/eye_net.py:109-146): all its weight / bias gradients in one launch:
 * dW[N][K] += (dY * act'(Y))^T . [X | X2], db[N] += column sums.                                                      */
#define EVE_WGRAD_BATCH_MAX 12
typedef struct eve_wgrad_problem {
    const float* dY;         /* [M][N]                                                                                        */
    const float* Y;          /* [M][N] or NULL (act == EVE_ACT_NONE)                                                          */
    const float* X;          /* [M][K1]                                                                                       */
    const float* X2;         /* [M][K2] or NULL: the input is the concatenation (K1 + K2 <= K; the rest of K is zero padding) */
    float* dW;               /* [N][K] accumulated                                                                            */
    float* db;               /* [N] accumulated, or NULL                                                                      */
    int M, N, K, K1, K2, act, rows_per_split /* set by the library */;
    int ldY;                 /* row stride of dY and Y in floats (0 = N): the first N columns of a wider matrix                  */
    int ldX, ldW;            /* row strides of X (0 = K1) and dW (0 = K): K1 leading columns of a wider X, an unpadded dW        */
    int x_shift_T;           /* > 0: X row m is replaced by row m - 1, zero where m % x_shift_T == 0 (the previous hidden state   */
    int reserved;            /*      of a scan over sequences of x_shift_T steps: dW_hh = dpre^T . h_prev)                        */
} eve_wgrad_problem;
typedef struct eve_wgrad_batch { int n; int first_block[EVE_WGRAD_BATCH_MAX]; eve_wgrad_problem p[EVE_WGRAD_BATCH_MAX]; } eve_wgrad_batch;
int eve_linear_wgrad_batch(const eve_wgrad_problem* problems, int n, eve_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser step over flat float buffers.  torch.optim.Adam with coupled L2 weight decay
 * (src/train.py:49-55) after nn.utils.clip_grad_norm_ (src/core/training.py:492-498).
 * ------------------------------------------------------------------------------------------------ */
/* Stream gates (ABI v7) -- how a hipGraph replay of forward + backward releases each gradient bucket's all-reduce on the
 * communication stream the moment the bucket's last gradient has been written, with the RCCL calls left OUTSIDE the graph
 * (north_star: "RCCL all-reduce of gradients ... overlapped with backward"; the reference's backward -> clip -> step sequence is
 * src/core/training.py:489-502).  eve_gate_signal: a one-thread kernel (capturable: it becomes a node of the graph behind the
 * weight-gradient kernel that completes the bucket) that releases and increments *flag.  eve_gate_wait: a one-wave kernel on the
 * OTHER stream that polls *flag until it has reached `value` (the replay count) -- or *value_ref when value_ref != NULL: a
 * device word, which is what lets the wait itself be a node of the graph (the replay's first node counts the replays there, and
 * the graph's clip + Adam nodes sit behind gate-waits for the "bucket reduced" words the communication stream signals) --
 * bounded: after max_polls polls (0: eve_dispatch_config.gate_wait_polls, ~seconds) it increments *timeouts, writes +inf to
 * *poison (when not NULL) and returns, so a missing signal cannot hang the device -- and cannot be trained on either (ABI v9):
 * `poison` is a float the caller keeps INSIDE the last gradient bucket it all-reduces, so after the collectives every rank
 * holds a non-zero value there and eve_adam_step(.., poison) skips the update on every rank alike (the all-reduce that ran on
 * the half-written bucket is discarded with it).  flag / timeouts / value_ref: device words, zeroed by the caller.           */
int eve_gate_signal(unsigned* flag, eve_stream_t stream);
int eve_gate_wait(const unsigned* flag, unsigned value, const unsigned* value_ref, unsigned* timeouts, float* poison,
                  unsigned max_polls, eve_stream_t stream);

/* out[0] += sum g^2 (caller zeroes out[0]; take sqrt on the host or in eve_adam_step).  Fixed summation order: the
 * result is bit-reproducible, so data-parallel replicas clip by the identical factor.  workspace: EVE_SUMSQ_WORKSPACE
 * floats of scratch owned by the caller.                                                            */
#define EVE_SUMSQ_WORKSPACE 1024
int eve_sumsq(long long n, const float* g, float* out, float* workspace, eve_stream_t stream);
/* Optimiser state that must live on the device (a replayed hipGraph cannot change kernel arguments, and the decision to skip
 * a step is taken on the device): 48 bytes, caller-owned, zero-initialised except loss_scale (1 = no scaling).            */
typedef struct eve_adam_guard {
    int step;             /* optimiser steps TAKEN so far = Adam's bias-correction exponent                                  */
    int skipped_total;    /* calls that left weights and moments untouched because the gradient norm was not finite         */
    int skipped_run;      /* ... consecutive ones (two in a row halve loss_scale)                                           */
    int good_run;         /* consecutive taken steps since the loss scale last changed (2 000 double it)                     */
    float loss_scale;     /* the factor the caller multiplied the loss by before backward; divided out of the gradient here */
    float applied;        /* 1 if the last call updated the weights, 0 if it skipped                                        */
    float clip;           /* factor the last taken step applied to the stored gradient (1/loss_scale and the norm clip)     */
    float bc1, bc2_sqrt;  /* bias corrections of the last taken step                                                        */
    int skipped_gate;     /* calls skipped because *poison != 0: a gradient bucket's stream gate timed out (ABI v9); also in skipped_total */
    float reserved[2];
} eve_adam_guard;
/* clip factor c = min(1, max_norm / (sqrt(*sumsq) * gscale' + 1e-6)) if sumsq != NULL and max_norm > 0, else 1;
 * g' = c * gscale' * g + wd * p;  m,v Adam moments;  p -= lr * mhat / (sqrt(vhat) + eps).
 * guard == NULL: gscale' = gscale, bias-correction step t = `step` (host value), no overflow handling.
 * guard != NULL: gscale' = gscale / guard->loss_scale, t = ++guard->step; with check_finite a non-finite *sumsq (an
 *   overflowed float16 gradient) SKIPS the step: weights, moments and guard->step stay, guard->skipped_* count it and the
 *   loss scale backs off (see eve_adam_guard).  The learning rate is `lr`, or *lr_dev (device float) when lr_dev != NULL: the
 *   LR schedule of src/core/training.py:382-418,436-442 writes *lr_dev before each (possibly replayed) step.
 *   poison != NULL (guard required): *poison != 0 (or NaN) -- a stream gate of this step's gradient exchange timed out,
 *   eve_gate_wait -- SKIPS the step like a non-finite norm does, counted in guard->skipped_gate; the loss scale is left alone. */
int eve_adam_step(long long n, float* p, const float* g, float* m, float* v, const float* sumsq,
                  float max_norm, float gscale, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, eve_adam_guard* guard, int check_finite, const float* lr_dev,
                  const float* poison, eve_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EVE_HIP_H_ */
