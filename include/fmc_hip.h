/*
 * fmc_hip.h -- C ABI of libfmc_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * denoising hot path of FMC (FudanCVL/SynFMC).
 *
 * This is the drop-in boundary.  The reference is pure Python and has
 * no FFI of its own; every entry point below replaces a chain of implicit PyTorch/cuDNN/cuBLAS
 * calls at the cited reference site (`path:line` relative to the reference root).  The
 * reference-side binding a maintainer would add is a ctypes stub: see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named `h_*`;
 *   - the caller owns every buffer; kernels never allocate, free or synchronise;
 *   - all work is enqueued on `stream` (a hipStream_t; NULL = the default stream);
 *   - every function returns 0 on success or a negative FMC_E_* code; `fmc_last_error()`
 *     returns a thread-local, human readable description of the last failure;
 *   - `dtype` selects the storage type of activations: FMC_BF16 (bf16 storage, fp32
 *     accumulate / statistics) or FMC_F32 (fp32 storage; the attention kernels then run the
 *     matrix products as split-bf16 x3 MFMA so that results agree with an fp32 reference to
 *     ~1e-5 -- this is the parity mode, not the fast mode);
 *   - activations are channels-last: an image batch is `[N, H*W, C]` with C contiguous.
 */
#ifndef FMC_HIP_H
#define FMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMC_VERSION 100 /* 0.1.0 */

enum { FMC_BF16 = 0, FMC_F32 = 1 };

enum {
    FMC_OK = 0,
    FMC_E_SHAPE = -1,  /* unsupported / inconsistent shape      -> Python ValueError          */
    FMC_E_DTYPE = -2,  /* unknown dtype code                    -> Python TypeError           */
    FMC_E_ALIGN = -3,  /* pointer or stride not 16-byte aligned -> Python ValueError          */
    FMC_E_LAUNCH = -4, /* hipLaunchKernel failed                -> Python RuntimeError        */
    FMC_E_NULL = -5    /* NULL where a pointer is required      -> Python ValueError          */
};

int fmc_version(void);
const char* fmc_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), channels-last.
 * Replaces nn.GroupNorm + SiLU in diffusers' ResnetBlock2D (norm1/norm2; ctor args at
 * fmc/models/unet_blocks.py:306-317), `conv_norm_out` + `conv_act` (fmc/models/unet.py:284-285,
 * 749-752), InflatedGroupNorm of the motion modules (fmc/models/resnet.py:27-37, no activation)
 * and the `norm` of Transformer2DModel (eps 1e-6).
 *   x, y        : [N, HW, C]  (dtype), y may alias x
 *   gamma, beta : [C] fp32
 *   stats       : [N, G, 2] fp32 out (mean, rstd) -- kept for the backward; must not be NULL
 *   workspace   : fp32 scratch of fmc_groupnorm_workspace_bytes(N, C, G) bytes
 *   act         : 0 = none, 1 = SiLU
 *   x2, C1      : two-source input, or NULL / 0.  With x2 the normalised tensor is the channel concat of x [N, HW, C1]
 *                 and x2 [N, HW, C - C1] (C1 % 8 == 0); y is the contiguous [N, HW, C].  The up blocks' `torch.cat([
 *                 hidden, skip], dim=1)` (unet_blocks.py:683,798) is then never materialised.
 * Requires C % 8 == 0 and C % G == 0.
 * ------------------------------------------------------------------------------------------- */
int64_t fmc_groupnorm_workspace_bytes(int N, int C, int G);
int fmc_groupnorm_silu_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats,
                           void* workspace, int N, int HW, int C, int G, float eps, int act, int dtype,
                           const void* x2, int C1, void* stream);
/* dX of the above (frozen gamma/beta: the only case on the FMC training path).
 *   dy, x, dx : [N, HW, C]; stats from the forward; act as in the forward. */
int fmc_groupnorm_silu_bwd(const void* dy, const void* x, void* dx, const float* gamma, const float* beta,
                           const float* stats, void* workspace, int N, int HW, int C, int G, int act,
                           int dtype, void* stream);

/* LayerNorm over the last dim of a token matrix [M, C] (eps 1e-5 everywhere on the path:
 * BasicTransformerBlock.norm1-3, TemporalTransformerBlock.norms / ff_norm,
 * fmc/models/motion_module.py:282-285).  Optional fused positional-encoding add for the temporal
 * blocks (`pos_encoder(norm(x))`, motion_module.py:355-356): if `pe` != NULL, token row r gets
 * pe[(r / pe_inner) % pe_frames] added AFTER the normalisation (rows are `[(b f), hw]` ordered:
 * pe_inner = hw, pe_frames = F).   gamma/beta/pe: fp32. */
int fmc_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, const float* pe,
                      int64_t M, int C, float eps, int pe_inner, int pe_frames, int dtype, void* stream);
/* The same with the residual add in front: sum_out = x + addend (rounded to the storage type), y = LayerNorm(sum_out) (+ pe).  For
 * `h = attn(...) + h` followed by the next norm (diffusers BasicTransformerBlock, fmc/models/motion_module.py:282-300) where the
 * projection ran on the vendor-library arm and would leave the add to an elementwise launch.  C in {320, 640, 1280}. */
int fmc_layernorm_add_fwd(const void* x, const void* addend, void* sum_out, void* y, const float* gamma, const float* beta,
                          const float* pe, int64_t M, int C, float eps, int pe_inner, int pe_frames, int dtype, void* stream);

/* GEGLU gate of diffusers' FeedForward (`a * gelu_erf(g)` with a,g = chunk(proj(x), 2);
 * fmc/models/motion_module.py:284 and BasicTransformerBlock.ff):  x [M, 2*Cff] -> y [M, Cff]. */
int fmc_geglu_fwd(const void* x, void* y, int64_t M, int Cff, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spatial attention: softmax(Q K^T * scale) V per (batch, head), flash style (no S x S tensor).
 * Replaces head_to_batch_dim + baddbmm + softmax + bmm + batch_to_head_dim in
 * fmc/models/attention_processor.py:61-67 / 148-154 for `attn1` (self, S_kv = S_q) and `attn2`
 * (text cross attention, S_kv = 77).
 *   q : [B, Sq, H*D]  rows `q_row_stride` elements apart, batches `q_batch_stride` apart
 *   k, v : [Bkv, Skv, H*D] likewise; batch b of q reads kv batch  b / kv_batch_div
 *          (kv_batch_div = F lets all frames of a clip share one text K/V)
 *   o : [B, Sq, H*D] with its own strides
 *   lse : optional [B, H, Sq] fp32 (natural-log-sum-exp of the scaled scores), NULL to skip
 *   D must be a multiple of 8 and <= 160.  All strides in ELEMENTS, multiples of 8.
 * ------------------------------------------------------------------------------------------- */
int fmc_spatial_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
                         int Sq, int Skv, int D, int64_t q_batch_stride, int64_t q_row_stride,
                         int64_t kv_batch_stride, int64_t kv_row_stride, int64_t o_batch_stride,
                         int64_t o_row_stride, int kv_batch_div, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Temporal attention: for every pixel and head, softmax(Q K^T * scale) V over the F frames.
 * Replaces the same call chain for the motion-module / camera-encoder attention blocks
 * (fmc/models/attention_processor.py:271-281 and :61-67 reached from
 * fmc/models/motion_module.py:365-389).
 * Token (clip n, frame f, pixel p) lives at  base + n*clip_stride + f*frame_stride + p*pix_stride
 * (elements); the H*D channels of a token are contiguous.  The reference layout `(b h w) f c` is
 * clip_stride = HW*F*C, pix_stride = F*C, frame_stride = C; the native channels-last layout
 * `[(b f), hw, c]` is clip_stride = F*HW*C, frame_stride = HW*C, pix_stride = C -- no transposing
 * copy in either case.  q/k/v share one stride triple (they are slices of one fused QKV row),
 * o has its own.   F in {16, 32};  D % 8 == 0, D <= 160;  (H*D) % 8 == 0.
 * ------------------------------------------------------------------------------------------- */
int fmc_temporal_attn_fwd(const void* q, const void* k, const void* v, void* o, int n_clips, int n_pix,
                          int F, int H, int D, int64_t clip_stride, int64_t frame_stride,
                          int64_t pix_stride, int64_t o_clip_stride, int64_t o_frame_stride,
                          int64_t o_pix_stride, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pluecker-ray embedding.  Replaces `ray_condition` (fmc/data/dataset.py:930-972) + the permutes
 * of `to_plucker_embedding` (train_cam_obj_ctrl.py:80-91, :833), which run on the CPU every step.
 *   K   : [B*F, 4] fp32 (fx, fy, cx, cy)      c2w : [B*F, c2w_rows(3|4), 4] fp32
 *   layout 0: out [B, F, H, W, 6]   (what ray_condition returns)
 *   layout 1: out [B, 6, F, H, W]   (what the trainers feed the pose encoder)
 *   layout 2: out [B*F, H/8, W/8, 384] channels-last, PixelUnshuffle(8) already applied
 *             (channel = c*64 + dy*8 + dx; input of CameraPoseEncoder.encoder_conv_in,
 *              fmc/models/pose_adaptor.py:228-232)
 *   out dtype: FMC_F32 or FMC_BF16.
 * ------------------------------------------------------------------------------------------- */
int fmc_plucker_fwd(const float* K, const float* c2w, void* out, int B, int F, int H, int W, int c2w_rows,
                    int layout, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * OMC rasteriser.  Replaces the Python loops of get_traj_features_v2 (fmc/util.py:158-201).
 *   poses : [BF, n_obj, 12] fp32      masks : [BF, n_obj, H, W] fp32 (Gaussian circle masks)
 *   layout 0: feat [BF, 13, H, W], mask_out [BF, 1, H, W]           (the Adapter's reference inputs)
 *   layout 2: feat [BF, H/8, W/8, 832] channels-last + PixelUnshuffle(8); mask_out [BF, H, W]
 *   Per pixel the LAST object with mask > 0 wins; feat = (pose*mask, mask) * mask.
 * ------------------------------------------------------------------------------------------- */
int fmc_omc_rasterize_fwd(const float* poses, const float* masks, void* feat, float* mask_out, int BF,
                          int n_obj, int H, int W, int layout, int dtype, void* stream);

/* Gaussian circle masks: the analytic part of the reference's sphere-mask synthesis (fmc/data/dataset.py:5365-5380;
 * cv2.minEnclosingCircle stays on the host and supplies the circles).  circles [N,3] fp32 = (cx, cy, radius) in
 * pixels, radius > 0; out [N,H,W] fp32 = [(x-int(cx))^2 + (y-int(cy))^2 <= int(r)^2] * g / max(g),
 * g = exp(-0.5 (d / (r/2))^2) with d the distance to the float centre.  Feeds fmc_omc_rasterize_fwd. */
int fmc_gaussian_circle_mask_fwd(const float* circles, float* out, int N, int H, int W, void* stream);

/* y[n,i,j,:] = x[n,i,j,:] * mask_in[n, si(i), sj(j)] with PyTorch's nearest rule
 * si(i) = min(floor(i * (float)Hin/h), Hin-1); also emits the resampled mask for the next level
 * (the reference cascades: fmc/adapter.py:175-177).  Used for the backward too (dX = mask * dY).
 *   x, y : [N, h*w, C] (dtype)   mask_in : [N, Hin, Win] fp32   mask_out : [N, h, w] fp32 or NULL */
int fmc_mask_modulate_fwd(const void* x, const float* mask_in, void* y, float* mask_out, int N, int h, int w,
                          int C, int Hin, int Win, int dtype, void* stream);

/* OMC injection `hidden + traj_features[idx]` (fmc/modified_modules.py:115-117,172-174) including
 * the classifier-free-guidance zero half (pipeline_animation_cm_om.py:671-676): the first
 * `skip_elems` elements of h get nothing added (just copied when out != h).
 *   h, out : [n_elems] (dtype), out may alias h;   t : [n_elems - skip_elems] (dtype) */
int fmc_feature_add_fwd(const void* h, const void* t, void* out, int64_t n_elems, int64_t skip_elems,
                        int dtype, void* stream);

/* CFG combine + DDIM (eta = 0) update, fused (pipeline_animation_cm_om.py:711-720; diffusers
 * DDIMScheduler.step):  eps = eps_u + g (eps_c - eps_u);  x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);
 * x' = sqrt(a_prev) x0 + sqrt(1-a_prev) eps.   eps_uc: [2, n] (dtype) (uncond, cond), or [1, n]
 * when guidance <= 1 (then pass has_uncond = 0).   x, x_out: [n] fp32 latents. */
int fmc_cfg_ddim_step(const void* eps_uc, const float* x, float* x_out, int64_t n, int has_uncond,
                      float guidance, float alpha_t, float alpha_prev, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogue:  out = alpha * (x @ w^T + bias) + residual      (epilogue 0)
 *                                      out = (x @ wa^T + ba) * gelu_erf(x @ wg^T + bg)  (epilogue 1, GEGLU)
 * Replaces nn.Linear of diffusers' Attention / FeedForward / Transformer2D 1x1 proj (call sites
 * fmc/models/attention_processor.py:50-69,255-283, fmc/models/motion_module.py:219,228,284) together with the
 * `+ hidden_states` residual that follows (motion_module.py:289-297) and the Camera-Adapter axpy
 * `qkv_merge(h + pose) * scale + h` (attention_processor.py:257).
 *   x [M, K] rows `ldx` apart, w [N, K] contiguous, bias [N] or NULL, residual [M, N] rows `ldres` apart or NULL,
 *   out [M, N] (epilogue 0) or [M, N/2] (epilogue 1), rows `ldo` apart.  All bf16, fp32 accumulate.
 *   epilogue 1 expects w / bias pre-interleaved per 64 rows: rows [64t, 64t+32) = value rows [32t, 32t+32) of the
 *   GEGLU projection, rows [64t+32, 64t+64) = the matching gate rows.
 *   Requires K % 64 == 0, N % 8 == 0 (N % 64 == 0 for epilogue 1), strides % 8 == 0.
 *   x2 != NULL: two-source A operand -- columns [0, k_split) of every row come from x, [k_split, K) from x2 (rows
 *   `ldx2` apart, k_split % 64 == 0): the 1x1 shortcut conv of an up-block ResNet reads (hidden, skip) without a concat.
 *   residual2 != NULL (needs residual; same row stride; epilogue 0): a second tensor added like residual.  The Camera
 *   Adapter's `qkv_merge(h + pose) * s + h` is linear in pose, and pose is constant over the denoising steps of a clip:
 *   with residual2 = s * (pose @ w^T + bias) computed once per clip the per-step call is alpha = s, no bias,
 *   residual = h, and the `h + pose` pass of attention_processor.py:257 disappears.
 *   tile: workgroup tile geometry, 0 = pick by shape, 1 = 128x128 (4 waves), 2 = 256x128 (8), 3 = 256x256 (16) with
 *   64-deep k-tiles in a 2-stage LDS ring; 4..6 = the same three with 32-deep k-tiles (half the LDS, twice the
 *   workgroups per CU); 7 = 256x128, 64-deep, 3 stages; 8 / 9 / 10 = 256x256 / 128x128 / 256x128, 32-deep, 4 stages
 *   (deeper rings keep more bytes in flight per CU); 11 = 128x320 (10 waves; spans N = 320 / 640 / 960 without padded
 *   columns); 12 = 128x320, 32-deep, 4 stages; 13 / 14 = the 8-phase 256x256 kernel (8 waves of 128x64, the two wave rows
 *   one barrier apart so each SIMD always has one wave on the matrix pipe and one loading; operands by half-tile
 *   `buffer_load ... lds` with a counted vmcnt, 4 / 5 half-tiles ahead; falls back to 3 for a two-source A operand or
 *   operands beyond 2 GiB); 15 = the persistent kernel of the K = 320 token projections (epilogue 0, N % 320 == 0, M % 64 == 0, one
 *   residual at most: 5 waves keep the weights of a 320-column block in registers for the life of the workgroup, 64-row A tiles
 *   arrive by LDS-DMA two tiles ahead, the residual rows are DMA'd into the output staging tile; the bias enters the accumulator
 *   first, so it agrees with the other arms to the last bit or two, not bit for bit; anything else falls back to 5);
 *   16 = 160 x 320 tiles (8 waves of 80 x 80 on the 16x16x32 MFMA, 32-deep sub-tiles in a 5-buffer LDS-DMA ring, one phase per sub-tile;
 *   N % 320 == 0, otherwise falls back to 13; token projections with more tiles than CUs and M % 160 == 0 run its persistent form, in which
 *   the next tile's operands stream in under the epilogue; GEGLU wants weight rows ordered [8 value | 8 gate] per 16);
 *   18 = tile 16 on a weight the caller PRE-PACKED tile-major in the kernel's own sub-tile order -- projections `[N / 320][K / 32][320][32]`,
 *   3x3 filters `[Cout / 320][Cin / 64][9 taps][2][320][32]` (sub-tile = (64-channel chunk, tap, 32-channel half)): every W piece of a sub-tile is
 *   one contiguous KiB; linear LDS-DMA requests move 30-50 % more bytes per CU than 16 rows x 64 B (tools/ubench/dma_mfma); bit-identical
 *   results, -1 .. -5 % per launch.  No fall-back exists for such a weight: N % 320 != 0, a two-source operand or stream-K are FMC_E_SHAPE.  The
 *   `w_tilemajor` flag of the tile-16-only entry points below (fmc_*_gn, fmc_linear_bf16_ln / _lnc / _ffblk) says the same about their weight.
 *   17 = the persistent form on 256 x 320 tiles (5 operand requests per 40 MFMAs and wave instead of 4 per 25), GEGLU epilogue only,
 *   M % 256 == 0 and more tiles than CUs -- anything else falls back to 16; same weight row order and bit-identical results.
 *   19 / 20 = 64 x 128 tiles of the ring kernel for small M (a wave = 32 x 64; 64-deep k-tiles in a 3-stage ring / 32-deep in a 4-stage ring: 2 / 3 workgroups per CU),
 *   21 / 22 = 128 x 128 / 64 x 256 on 8 such waves (two workgroups = 16 waves per CU);
 *   fmc_linear_bf16 without GEGLU only, bit-identical to tiles 1 / 4; split-K, stream-K, two residuals and fp32 storage fall back to tile 1.
 *   Every arm computes the same function -- bit for bit among the plain-grid arms of one k-tile depth
 *   (the 32-deep arms, split-K and stream-K add the same products in another order) -- so callers may time them and keep
 *   the fastest.
 *   split_k > 1 (epilogue 0 only): the k-tiles of every output tile are dealt to split_k workgroups that write fp32
 *   partial sums to `workspace` (>= split_k * M * N * 4 bytes, 16-byte aligned, caller-owned scratch); a second kernel
 *   sums them in a fixed order and applies the epilogue.  For M*N too small to fill 256 CUs (the 5x8 level).
 *   split_k == -3 / split_k <= -16 (arms 13 / 14 only; other arms treat it as -1): the K-LOCKSTEP split.  Every tile's reduction is cut into
 *   S chunks (S = -split_k - 16, or chosen by a cost model for -3); unit (chunk, tile) u runs on workgroup (u mod CUs) of a persistent pass in
 *   such an order that the CUs of one XCD work on the same k-chunk of the same filter column block at the same time (the weight leaves the
 *   Infinity Cache once per XCD and round, not once per tile: 775 -> ~90 MB per launch on the 10x16-level convolutions); fp32 partials in
 *   accumulator layout, one slot per unit (workspace >= 4096 + tiles * S * 256 KiB), summed in chunk order by a bandwidth-shaped finishing
 *   kernel that applies the plain epilogue (GEGLU: by the 8-phase kernel's own finishing launch).  Deterministic; same workspace as -1.
 *   split_k == -2 (arms 13 / 14 only; other arms treat it as -1): the tiles of the whole rounds (tiles / CUs * CUs of them) run on the
 *   plain grid and only the last partial round goes through the two stream-K launches below -- same workspace, same results as -1.
 *   split_k == -1: stream-K.  (On the 8-phase arms 13 / 14: TWO launches -- a persistent pass, one workgroup per CU, every
 *   segment's fp32 accumulators into a workspace slot in accumulator layout, then one workgroup per output tile that sums
 *   the slots covering its tile and runs the epilogue; workspace >= 4096 + CUs * (ceil(tiles / CUs) + 2) * 256 KiB, no
 *   flags.  What follows describes the other arms.)  One persistent workgroup per CU slot; the (tile, k-tile) iteration space is cut into equal
 *   contiguous ranges so every CU does the same number of k-tiles whatever tiles / CUs is; a tile cut by a range
 *   boundary is finished by the workgroup holding its first k-tiles, the others hand over fp32 partials through the
 *   workspace (sc1 accesses + flags, no fences; deterministic summation order).  `workspace` = [4096 bytes of flags,
 *   ZERO on entry and handed back zero | >= workgroups * tile_rows * tile_cols * 4 bytes]; too small a problem or
 *   workspace silently runs the plain grid.  Both epilogues.
 * ------------------------------------------------------------------------------------------- */
int fmc_linear_bf16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int N,
                    int K, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, int epilogue, int tile, int split_k,
                    void* workspace, int64_t workspace_bytes, const void* x2, int64_t ldx2, int k_split,
                    const void* residual2, void* stream);

/* Implicit-GEMM 3x3 convolution (stride 1, pad 1) on channels-last bf16 images with the ResNet-block epilogue:
 *   out[i,y,x,:] = conv(x)[i,y,x,:] + bias + temb[i,:] + residual[i,y,x,:]
 * Replaces conv1 / conv2 of diffusers' ResnetBlock2D (ctor args fmc/models/unet_blocks.py:306-317), including
 * `+ time_emb_proj(silu(temb))[:, :, None, None]` and the `input_tensor + hidden_states` residual, and the conv of
 * Upsample2D (unet_blocks.py:625).
 *   x [n_img, H, W, Cin], w [Cout, 3, 3, Cin] (= the filter in torch.channels_last memory format),
 *   bias [Cout] | NULL, residual / out [n_img, H, W, Cout].  Cin % 64 == 0, Cout % 8 == 0.
 *   temb | NULL: image i adds row (i / temb_img_div) of temb, rows `temb_row_stride` elements apart -- [n_img, Cout]
 *   contiguous is (Cout, 1); the U-Net passes a column slice of ONE projection of all 22 ResNet blocks' time embeddings,
 *   [clips, sum Cout], with temb_img_div = frames, so the reference's per-frame repeat never materialises.
 *   upsample2x != 0: x is [n_img, H/2, W/2, Cin] and the convolution reads it through a nearest-neighbour 2x upsample
 *   (diffusers Upsample2D, unet_blocks.py:625: `F.interpolate(scale_factor=2)` then conv) -- the 4x larger tensor is
 *   never written; H, W stay the OUTPUT size.  upsample2x == 2: stride-2 convolution (diffusers Downsample2D,
 *   unet_blocks.py:350: 3x3, stride 2, pad 1): x is [n_img, 2H, 2W, Cin], H, W the OUTPUT size.
 *   tile / split_k / workspace: as for fmc_linear_bf16 (M = n_img*H*W, N = Cout, K = 9*Cin). */
int fmc_conv3x3_bf16(const void* x, const void* w, const void* bias, const void* temb, const void* residual, void* out,
                     int n_img, int H, int W, int Cin, int Cout, int64_t temb_row_stride, int temb_img_div,
                     int upsample2x, int tile, int split_k, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm statistics out of the producing GEMM / conv epilogue (reference sites: the GroupNorms of diffusers'
 * ResnetBlock2D / Transformer2DModel and of the motion module, fmc/models/unet_blocks.py:306-317, fmc/models/motion_module.py:177).
 * fmc_linear_bf16_gn / fmc_conv3x3_bf16_gn = fmc_linear_bf16 (epilogue 0) / fmc_conv3x3_bf16 on tile 16 (160 x 320 tiles; N resp. Cout % 320
 *   == 0, pixels per image gn_hw resp. H W % 160 == 0, M % gn_hw == 0) that ALSO write, per (image, 160-row tile of the image, group of
 *   N / 32 channels), the pair (sum, sum of squares) of the rounded bf16 outputs: gn_partials [M / gn_hw][gn_hw / 160][32][2] fp32.
 * fmc_groupnorm_apply_fwd = the second pass of fmc_groupnorm_silu_fwd alone: combines `partials [N][part_splits][G][2]` (fp64, fixed
 *   order), writes mean / rstd to stats and y = act(GroupNorm(x)).  The first pass (one more read of x) is not launched.
 * ------------------------------------------------------------------------------------------- */
int fmc_linear_bf16_gn(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int N, int K,
                       int64_t ldx, int64_t ldres, int64_t ldo, float alpha, const void* residual2, float* gn_partials, int gn_hw,
                       int w_tilemajor, void* stream);
/* fmc_linear_bf16_ln = fmc_linear_bf16 (epilogue 0, tile 16, persistent form) for N == 320 that ALSO writes the CONSUMER's LayerNorm of the
 *   rows it produces (diffusers BasicTransformerBlock norm1 / norm2 / norm3 behind proj_in / attn1 / attn2, and the motion module's norms
 *   behind proj_in / the attention blocks: fmc/models/motion_module.py:282-288,355): a 160 x 320 tile holds whole rows, so
 *   ln_out[m, :] = (out[m, :] - mean) * rstd * ln_gamma + ln_beta (+ ln_pe[(m / ln_pe_inner) % ln_pe_frames, :]) is computed from the
 *   bf16-rounded rows in the staging tile (two-pass statistics, as fmc_layernorm_fwd) and out is never read again by a LayerNorm launch.
 *   Needs bf16, N == 320, M % 160 == 0, M / 160 > CUs, ln_pe_inner % 160 == 0 (a tile lies in one frame); ln_out contiguous [M, 320];
 *   gamma / beta / pe fp32.  Anything else is FMC_E_SHAPE (callers fall back to fmc_linear_bf16 + fmc_layernorm_fwd). */
int fmc_linear_bf16_ln(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int N, int K,
                       int64_t ldx, int64_t ldres, int64_t ldo, float alpha, const void* residual2, void* ln_out, const float* ln_gamma,
                       const float* ln_beta, float ln_eps, const float* ln_pe, int ln_pe_inner, int ln_pe_frames, float* ln_stats, int w_tilemajor,
                       void* stream);
/* ... or, with ln_out == NULL and ln_stats != NULL, only the rows' statistics: ln_stats[M][2] = (mean, rstd) fp32 -- for a consumer that is a
 * GEMM and applies the LayerNorm itself:
 * fmc_linear_bf16_lnc = LayerNorm(x) @ w^T + b computed WITHOUT materialising LayerNorm(x): with w_gamma = w diag(gamma) (bf16),
 *   ln_c[n] = sum_k w_gamma[n, k] and ln_bias = w beta + b (fp32 [N]),   out[m, n] = rstd[m] (x[m, :] . w_gamma[n, :] - mean[m] ln_c[n]) + ln_bias[n]
 *   in the epilogue of tile 16's persistent form (epilogue 0, or 1 = GEGLU on [8 value | 8 gate]-ordered rows, ln_c / ln_bias in the same
 *   order); the products are exact in fp32 (bf16 x bf16), so the only new rounding is that of w gamma to bf16, in place of LayerNorm(x) to
 *   bf16.  Call sites: attn1 / attn2.to_q / the GEGLU projection behind norm1 / norm2 / norm3 of diffusers' BasicTransformerBlock and behind
 *   the motion module's ff_norm (fmc/models/motion_module.py:295-299).  Needs bf16, N % 320 == 0, M % 160 == 0, more 160 x 320 tiles than CUs. */
int fmc_linear_bf16_lnc(const void* x, const void* w_gamma, void* out, int64_t M, int N, int K, int64_t ldx, int64_t ldo, int epilogue,
                        const float* ln_stats, const float* ln_c, const float* ln_bias, int w_tilemajor, void* stream);
/* The feed-forward's intermediate in TILE-MAJOR order (diffusers FeedForward: GEGLU projection -> Linear, fmc call sites as for
 * fmc_linear_bf16): `[M / 160][C / 32][160 rows][32]` instead of `[M][C]`.  The GEGLU projection (epilogue 1, out_blocked = 1) writes its
 * gated 160 x 160 tile as five contiguous 10-KiB blocks, and the second GEMM (epilogue 0, x_blocked = 1) requests each 32-deep A sub-tile as
 * ONE contiguous block: linear 1-KiB LDS-DMA requests and whole DRAM pages instead of 160 row segments of 64 bytes.  The tensor is private to
 * the pair; results are bit-identical to the row-major path.  Tile 16's persistent form only (M % 160 == 0, N % 320 == 0, more tiles than CUs);
 * ln_stats / ln_c / ln_bias != NULL: the GEGLU projection also applies its input's LayerNorm (fmc_linear_bf16_lnc; bias must be NULL then). */
int fmc_linear_bf16_ffblk(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int N, int K,
                          int64_t ldres, float alpha, int epilogue, int x_blocked, int out_blocked, const float* ln_stats, const float* ln_c,
                          const float* ln_bias, int w_tilemajor, void* stream);
/* The GroupNorm in front of a transformer's proj_in folded INTO the projection (diffusers Transformer2DModel.norm -> proj_in, fmc/models/unet_blocks.py:323-333;
 * TemporalTransformer3DModel.norm -> proj_in, fmc/models/motion_module.py:124-129: no activation between the two):
 *   proj(GN(x))[m, :] = W'_img x[m, :] + bias'_img,   W'_img = W diag(rstd[img, g(c)] gamma[c]),   bias'_img = bias + W beta - W'_img mean[img, g(.)]
 * fmc_groupnorm_fold_linear: partials [n_img, part_splits, G, 2] (sum, sum of squares per image, split and group, as fmc_groupnorm_apply_fwd takes them:
 *   the producer's epilogue wrote them) -> w_out bf16 [n_img][N][C] (each image's matrix tile-major [N / 320][C / 32][320][32] if w_tilemajor) and
 *   bias_out fp32 [n_img][N], computed from the ROUNDED W'_img so that the mean term cancels against W'_img x exactly.
 * fmc_linear_bf16_imgw: out[m, :] = w_img[m / img_rows] x[m, :] + bias_img[m / img_rows] on tile 16's persistent form (M % 160 == 0, img_rows % 160 == 0,
 *   N % 320 == 0, at least as many tiles as CUs); ln_stats != NULL (N == 320): also the rows' LayerNorm (mean, rstd) -> ln_stats[M][2], as fmc_linear_bf16_ln.
 * The normalised tensor is neither written nor read: one HBM pass over x less per transformer. */
int fmc_groupnorm_fold_linear(const float* partials, int part_splits, const float* gamma, const float* beta, const void* w, const void* bias,
                              void* w_out, float* bias_out, int n_img, int HW, int C, int G, int N, float eps, int w_tilemajor, void* stream);
int fmc_linear_bf16_imgw(const void* x, const void* w_img, const float* bias_img, void* out, int64_t M, int N, int K, int64_t ldx, int64_t ldo,
                         int img_rows, float* ln_stats, float ln_eps, int w_tilemajor, void* stream);
/* The feed-forward's output projection and the transformer's proj_out as ONE product (diffusers BasicTransformerBlock `ff(norm3(h)) + h` followed by
 * Transformer2DModel.proj_out + residual, fmc/models/unet_blocks.py:323-333; the motion module's `ff(ff_norm(h)) + h` followed by
 * TemporalTransformer3DModel.proj_out + residual, fmc/models/motion_module.py:130-134,295-299):
 *   proj_out(ff2(g) + b2 + h) + bp + x  =  [g | h] [Wp W2 | Wp]^T + (Wp b2 + bp) + x
 * x_blocked = the gated intermediate g, tile-major [M / 160][k_split / 32][160][32] (fmc_geglu*_ln_bf16 / fmc_geglu_pipe_ln_bf16 with out_blocked);
 * x2 = h, row-major [M, K - k_split] with rows ldx2 apart; w = the folded weight [N, K] (the caller folds it once per weight version, in fp32, and
 * rounds once), tile-major if w_tilemajor; gn_partials (may be NULL) as fmc_linear_bf16_gn.  One launch and two HBM passes less than the pair; the
 * block's output before proj_out is never rounded to bf16.  Tile 16's persistent form only: M % 160 == 0, N % 320 == 0, at least as many tiles as CUs,
 * k_split % 64 == 0. */
int fmc_linear_bf16_fftail(const void* x_blocked, const void* x2, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                           int N, int K, int k_split, int64_t ldx2, int64_t ldres, float* gn_partials, int gn_hw, int w_tilemajor, void* stream);
int fmc_conv3x3_bf16_gn(const void* x, const void* w, const void* bias, const void* temb, const void* residual, void* out, int n_img,
                        int H, int W, int Cin, int Cout, int64_t temb_row_stride, int temb_img_div, int upsample2x,
                        float* gn_partials, int w_tilemajor, void* stream);
int fmc_groupnorm_apply_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats, const float* partials,
                            int part_splits, int N, int HW, int C, int G, float eps, int act, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3 convolution with the input halo tile resident in LDS and GroupNorm + SiLU applied while it is staged (SURVEY.md section 8 f1;
 * csrc/conv_halo.hip).  Replaces `conv(nonlinearity(norm(x)))` of diffusers' ResnetBlock2D (ctor args fmc/models/unet_blocks.py:306-317:
 * norm1 -> SiLU -> conv1 (+ time_emb_proj) and norm2 -> SiLU -> conv2 (+ input_tensor)), `InflatedConv3d` / `InflatedGroupNorm`
 * (fmc/models/resnet.py:16-37) and Upsample2D's conv (unet_blocks.py:625) -- the normalised tensor is never written.
 *   x [n_img, Hs, Ws, Cin1] (+ x2 [n_img, Hs, Ws, Cin - Cin1] | NULL: the up blocks' `cat([hidden, skip], 1)`, unet_blocks.py:683,798, read in place);
 *   Hs, Ws = H, W, or H / 2, W / 2 with upsample2x (nearest 2x upsample folded into the halo addressing); H, W = OUTPUT size, W % 32 == 0;
 *   w_packed = fmc_conv3x3_halo_pack_weight(filter [Cout][3][3][Cin]): Cin (both sources) % 64 == 0, Cout % 160 == 0;
 *   bias [Cout] | NULL, temb / temb_row_stride / temb_img_div and residual [n_img, H, W, Cout] | NULL as for fmc_conv3x3_bf16;
 *   gn_coef [n_img, Cin, 2] fp32 | NULL: the operand is act(x * gn_coef[i, c, 0] + gn_coef[i, c, 1]) rounded to bf16 (act = SiLU when gn_act),
 *     zero outside the image (the reference pads the NORMALISED tensor); produced by fmc_groupnorm_coef;
 *   gn_partials [n_img, fmc_conv3x3_halo_tiles_per_image(H, W), 32, 2] fp32 | NULL: (sum, sum of squares) of the bf16-rounded outputs per image,
 *     pixel tile and group of Cout / 32 channels -- the statistics pass of the GroupNorm that consumes `out` (fmc_groupnorm_coef /
 *     fmc_groupnorm_apply_fwd take them as `partials` with part_splits = tiles per image).
 * fmc_groupnorm_coef: partial sums [N, part_splits, G, 2] -> coef [N, C, 2] = (rstd gamma_c, beta_c - mean rstd gamma_c) and, when stats != NULL,
 *   stats [N, G, 2] = (mean, rstd); HW = pixels per image (count per group = HW * C / G); fp64 combination, deterministic. */
/* ---------------------------------------------------------------------------------------------
 * Token GEMM for the small-M projections of the 10x16 / 5x8 levels (csrc/gemm4.hip, round 5): nn.Linear of diffusers' Attention / Transformer2D / the
 * motion module there (fmc/models/attention_processor.py:50-69,255-283, fmc/models/motion_module.py:219,228,284) with what follows it,
 *   out = alpha * (x W^T + bias) + residual + residual2          (bf16; any of bias / residual / residual2 NULL; residual2 needs residual),
 * on 160 x 160 tiles, 4 waves, software-pipelined fragment reads and operand requests.  x rows ldx apart, residual(s) ldres, out ldo; W [N, K] row-major.
 * K % 64 == 0, N % 8 == 0, strides % 8 == 0, operands < 2 GiB; edge tiles are masked.  An arm of `hip_ops.linear`'s per-shape choice (700). */
int fmc_linear4_supported(int64_t M, int N, int K, int64_t ldx);
int fmc_linear4_bf16(const void* x, const void* w, const void* bias, const void* residual, const void* residual2, void* out, int64_t M, int N, int K,
                     int64_t ldx, int64_t ldres, int64_t ldo, float alpha, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The vendor arm of the token projections, called directly (csrc/vendor_gemm.hip, round 5): out = x W^T + bias + residual as ONE hipBLASLt launch
 * (D = A B + beta C with the bias epilogue) -- nn.Linear + the residual add of attention to_out / the feed-forward output / proj_out
 * (fmc/models/attention_processor.py:69, fmc/models/motion_module.py:228,299) where the per-shape choice is the library.  torch's F.linear cannot pass C and
 * bias together and followed the GEMM with an elementwise add.  bf16; bias [N] | NULL; residual rows ldres apart | NULL; x rows ldx, out rows ldo apart;
 * W [N, K] row-major.  `algo` indexes the heuristic's candidate list, 0 <= algo < fmc_vendor_linear_candidates(...) (0 = the library's first choice).
 * Ownership (round 6): nothing here allocates device memory.  fmc_vendor_init() creates the library handle of the CURRENT device (idempotent),
 * fmc_vendor_destroy() drops it together with the device's cached plans; every other entry point fails with FMC_E_NULL before init.  Split-K / stream-K
 * candidates need scratch: the caller passes `workspace` (16-byte aligned, >= fmc_vendor_workspace_bytes() = the size candidates are planned for; one buffer
 * per stream that issues these calls, NULL / 0 allowed for candidates that need none).  The first call for a problem queries the heuristic (host work):
 * make it outside stream capture.  fmc_vendor_version() = hipblasLtGetVersion (candidate indices are only meaningful within one library version). */
int fmc_vendor_init(void);
int fmc_vendor_destroy(void);
int fmc_vendor_version(void);
int64_t fmc_vendor_workspace_bytes(void);
int fmc_vendor_linear_candidates(int64_t M, int N, int K, int64_t ldx, int64_t ldres, int64_t ldo, int has_bias, int has_residual);
int fmc_vendor_linear_bf16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int N, int K,
                           int64_t ldx, int64_t ldres, int64_t ldo, int algo, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head attention for the steps either side of the denoising loop (csrc/attn_generic.hip, round 6): softmax(q k^T scale + mask) v
 * for the shapes the hot-path kernels above do not take -- the VAE mid block's single head of width 512 (diffusers AutoencoderKL; reached from
 * `decode_latents`, fmc/pipelines/pipeline_animation_cm_om.py:465-478, and `vae.encode`, train_cam_obj_ctrl.py:786) and CLIP's causal text self-attention
 * (`_encode_prompt`, pipeline_animation_cm_om.py:480-568).  q [B, Sq, H D], k / v [B, Skv, H D] with batch / row strides in elements (slices of a fused
 * projection are fine), o [B, Sq, H D]; D in {32, 64, 128, 256, 512}; `causal`: key j attends to query i only when j <= i; `key_keep` [B][Skv] bytes
 * (1 = attend) or NULL; dtype FMC_BF16 or FMC_F32 (split-bf16 x3 products).  A fully masked row yields zeros. */
int fmc_attention_supported(int D);
int fmc_attention_fwd(const void* q, const void* k, const void* v, void* o, const unsigned char* key_keep, int B, int H, int Sq, int Skv, int D,
                      int64_t q_batch_stride, int64_t q_row_stride, int64_t kv_batch_stride, int64_t kv_row_stride, int64_t o_batch_stride,
                      int64_t o_row_stride, float scale, int causal, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm + GEGLU projection with the gate software-pipelined under the matrix work (csrc/geglu_pipe.hip, round 6): diffusers FeedForward's
 * `GEGLU(norm(h))` (fmc/models/motion_module.py:295-299; BasicTransformerBlock.norm3 / ff) at C = 320 | 640 -- same function as fmc_geglu320_ln_bf16 /
 * fmc_geglu640_ln_bf16: out [M, cff] = (n W_v^T + b_v) * gelu(n W_g^T + b_g), n = LayerNorm(h).  One wave per SIMD keeps two accumulator sets and gates
 * chunk c - 1 in the shadow of chunk c's MFMAs; no staging tile, no barrier after the LayerNorm.  variant 0: 80-row tiles, 4 waves x 32 gated columns
 * (C = 320 | 640, M % 80 == 0); variant 1: 160-row tiles, 8 waves x 16 gated columns (C = 320, M % 160 == 0: every weight fragment serves twice the rows).
 * w_packed: `hip_ops.pack_geglu_frag(w, G)`, G = 32 (variant 0) | 16 (variant 1): [cff / G column groups][C / 32 k-steps][G / 8 blocks: value blocks, then gate
 * blocks][lane][8]; bias [2 cff] (value | gate) or NULL; cff % 128 == 0; out_blocked: tile-major [M / 160][cff / 32][160][32] (the operand layout of the
 * feed-forward's second GEMM; M % 160 == 0). */
int fmc_geglu_pipe_supported(int64_t M, int cff, int C, int variant);
int fmc_geglu_pipe_ln_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_packed, const void* bias,
                           int64_t M, int cff, int C, int out_blocked, int variant, void* stream);

/* The statistics pass of fmc_groupnorm_silu_fwd alone (x read once, nothing written but the sums): partials [N][splits][G][2] fp32 with
 * splits = fmc_groupnorm_partial_splits(HW, C); x2 / C1: two-source channel concat as for fmc_groupnorm_silu_fwd. */
int fmc_groupnorm_partial_splits(int HW, int C);
int fmc_groupnorm_partials(const void* x, const void* x2, int C1, float* partials, int N, int HW, int C, int G, int dtype, void* stream);
int fmc_conv3x3_halo_supported(int n_img, int H, int W, int Cin, int Cin1, int Cout, int upsample2x);
int64_t fmc_conv3x3_halo_packed_bytes(int Cin, int Cout);
int fmc_conv3x3_halo_pack_weight(const void* w, void* dst, int Cin, int Cout, void* stream);
int fmc_conv3x3_halo_tiles_per_image(int H, int W);
int fmc_conv3x3_halo_bf16(const void* x, const void* x2, int Cin1, const void* w_packed, const void* bias, const void* temb, const void* residual,
                          void* out, int n_img, int H, int W, int Cin, int Cout, int64_t temb_row_stride, int temb_img_div, int upsample2x,
                          const float* gn_coef, int gn_act, float* gn_partials, void* stream);
/* The same convolution for the SMALL feature maps (csrc/conv_halo4.hip): image width a multiple of 8; the image is cut into row blocks of 10 x 16
 * pixels (W % 16 == 0) or 5 x 8; a tile is 320 pixels = 2 / 8 row blocks x 80 output channels, 4 waves with software-pipelined fragment reads.  Arguments as
 * fmc_conv3x3_halo_bf16 without the GroupNorm operand path; Cout % 80 == 0; w_packed = fmc_conv3x3_halo4_pack_weight(filter);
 * gn_partials [n_img, fmc_conv3x3_halo4_row_blocks_per_image(H, W), 32, 2].  fmc_conv3x3_halo4_tiles: workgroups an un-split launch has.
 * split_k > 1 (5x8-pixel images: 64 tiles on 256 CUs otherwise): the 64-channel chunks of the reduction are dealt to split_k workgroups per tile, fp32
 * partials in `workspace` (>= split_k * n_img * H * W * Cout * 4 bytes), summed in a fixed order by a second launch that runs the epilogue.
 * wide != 0 (W % 16 == 0, Cout % 160 == 0; its own packed filter): 8 waves, two per SIMD, 320 pixels x 160 channels per tile -- the same software-pipelined
 * loop where that many tiles fill the chip (the 40x64 / 20x32 levels). */
int fmc_conv3x3_halo4_supported(int n_img, int H, int W, int Cin, int Cin1, int Cout, int upsample2x, int wide);
int fmc_conv3x3_halo4_pack_weight(const void* w, void* dst, int Cin, int Cout, int wide, void* stream);
int fmc_conv3x3_halo4_row_blocks_per_image(int H, int W);
int fmc_conv3x3_halo4_tiles(int n_img, int H, int W, int Cout, int wide);
int fmc_conv3x3_halo4_bf16(const void* x, const void* x2, int Cin1, const void* w_packed, const void* bias, const void* temb, const void* residual,
                           void* out, int n_img, int H, int W, int Cin, int Cout, int64_t temb_row_stride, int temb_img_div, int upsample2x,
                           float* gn_partials, int split_k, void* workspace, int64_t workspace_bytes, int wide, void* stream);
int fmc_groupnorm_coef(const float* partials, int part_splits, const float* gamma, const float* beta, float* coef, float* stats, int N, int HW,
                       int C, int G, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp32-storage ("parity") mode of the two GEMMs above: split-bf16 x3 on the SAME kernels.
 * The reference's CPU path is fp32 (north_star: outputs within 1e-3 rel-inf of it); the bf16 product path can only be held
 * to bf16's own rounding against it.  To check the hand-written tile maps, operand loaders (token / implicit 3x3 conv /
 * upsample / stride 2), fragment layouts, split-K order and epilogue arithmetic at fp32 accuracy, every fp32 operand is
 * split into bf16 `hi` and bf16 `lo = bf16(x - hi)` (x = hi + lo to ~2^-17 relative) and laid out along the reduction:
 *     activation rows  [hi | hi | lo]      weight rows  [hi | lo | hi]       (three blocks of K columns each)
 * so that ONE bf16 MFMA GEMM over K3 = 3 K accumulates hi hi' + hi lo' + lo hi' in fp32 -- the fp32 product up to the
 * dropped lo lo' term (2^-18).  Same call sites as fmc_linear_bf16 / fmc_conv3x3_bf16 when the model is held in fp32
 * (fmc/models/attention_processor.py:50-69,255-283, motion_module.py:219,228,284, unet_blocks.py:306-317,625).
 *
 * fmc_split_bf16x3: src fp32 [rows, C] (rows `ld_src` floats apart) -> dst bf16 [rows, 3 K] (rows `ld_dst` apart), columns
 *   [col0, col0 + C) of each of the three K-blocks; role 0 = activation pattern, 1 = weight pattern.  col0 / K let a
 *   two-source operand (`x2` of fmc_linear_bf16) be written into one buffer.  For the convolution the rows are pixels
 *   (C = K = Cin: x3 is [n, H, W, 3 Cin]) resp. (Cout, tap) pairs (w3 is [Cout, 3, 3, 3 Cin]).  C, K, col0 % 8 == 0.
 * fmc_linear_x3_f32 / fmc_conv3x3_x3_f32: as fmc_linear_bf16 / fmc_conv3x3_bf16 on such operands (K3 = 3 K, Cin3 = 3 Cin),
 *   with bias / temb / residual(s) / out in FP32 (strides in floats, % 4 == 0) and the whole epilogue (bias, alpha, temb,
 *   residuals, exact-erf GEGLU) in fp32 straight from the accumulator registers.  tile: arms 1..14 (15 falls back to 5);
 *   split_k >= 1 only (no stream-K); no two-source operand (split both sources into one x3).
 * ------------------------------------------------------------------------------------------- */
int fmc_split_bf16x3(const float* src, void* dst, int64_t rows, int C, int64_t ld_src, int64_t ld_dst, int col0, int K,
                     int role, void* stream);
int fmc_linear_x3_f32(const void* x3, const void* w3, const float* bias, const float* residual, float* out, int64_t M, int N,
                      int K3, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, int epilogue, int tile, int split_k,
                      void* workspace, int64_t workspace_bytes, const float* residual2, void* stream);
int fmc_conv3x3_x3_f32(const void* x3, const void* w3, const float* bias, const float* temb, const float* residual, float* out,
                       int n_img, int H, int W, int Cin3, int Cout, int64_t temb_row_stride, int temb_img_div,
                       int upsample2x, int tile, int split_k, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward entry points (training stages 2/3 of the reference: the U-Net is frozen but the activation gradient
 * flows through every layer back to the OMC / CMC injection points, train_cam_obj_ctrl.py:917-929,
 * train_cam_ctrl.py:626-650; in the reference this is PyTorch autograd through the same call sites as the forwards).
 * ------------------------------------------------------------------------------------------- */
/* dX (and optionally dgamma / dbeta, fp32 [C], ACCUMULATED into: zero them first) of fmc_layernorm_fwd. */
int fmc_layernorm_bwd(const void* dy, const void* x, const float* gamma, void* dx, float* dgamma, float* dbeta,
                      int64_t M, int C, float eps, int dtype, void* stream);
/* The same two backward passes with an ADDEND (shape of dx, may be NULL): dx = addend + dX(norm).  In `h + f(norm(h))` -- every residual
 * connection of the U-Net (diffusers ResnetBlock2D / BasicTransformerBlock, fmc/models/motion_module.py:282-300) -- h receives a gradient
 * along the skip and one through the norm; autograd sums them with an elementwise kernel per connection (303 launches per training step,
 * train_cam_obj_ctrl.py:915); a norm node that owns both uses of h takes the skip gradient here instead. */
int fmc_layernorm_bwd_add(const void* dy, const void* x, const float* gamma, void* dx, float* dgamma, float* dbeta, const void* addend,
                          int64_t M, int C, float eps, int dtype, void* stream);
int fmc_groupnorm_silu_bwd_add(const void* dy, const void* x, void* dx, const float* gamma, const float* beta, const float* stats,
                               void* workspace, int N, int HW, int C, int G, int act, const void* addend, int dtype, void* stream);
/* dX [M, 2*Cff] of fmc_geglu_fwd from dy [M, Cff] and the forward input x. */
int fmc_geglu_bwd(const void* dy, const void* x, void* dx, int64_t M, int Cff, int dtype, void* stream);
/* dQ, dK, dV of fmc_spatial_attn_fwd.  o / lse are the forward's outputs, d_o has o's strides, dvec is a
 * [B, H, Sq] fp32 scratch (receives rowsum(dO .* O)).  dk / dv are [B / kv_batch_div, Skv, H*D]: frames sharing one
 * text K/V are summed inside the kernel.  dk and dv may be NULL together (frozen text projections: the key/value
 * side needs no gradient); the dK/dV kernel is then not launched. */
int fmc_spatial_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                         float* dvec, void* dq, void* dk, void* dv, int B, int H, int Sq, int Skv, int D,
                         int64_t q_batch_stride, int64_t q_row_stride, int64_t kv_batch_stride, int64_t kv_row_stride,
                         int64_t o_batch_stride, int64_t o_row_stride, int64_t dq_batch_stride, int64_t dq_row_stride,
                         int64_t dkv_batch_stride, int64_t dkv_row_stride, int kv_batch_div, float scale, int dtype,
                         void* stream);
/* dQ, dK, dV of fmc_temporal_attn_fwd (probabilities are recomputed; nothing saved by the forward).
 * q/k/v share (clip, frame, pix) strides, d_o has its own, dq/dk/dv share a third triple. */
int fmc_temporal_attn_bwd(const void* q, const void* k, const void* v, const void* d_o, void* dq, void* dk, void* dv,
                          int n_clips, int n_pix, int F, int H, int D, int64_t clip_stride, int64_t frame_stride,
                          int64_t pix_stride, int64_t do_clip_stride, int64_t do_frame_stride, int64_t do_pix_stride,
                          int64_t dq_clip_stride, int64_t dq_frame_stride, int64_t dq_pix_stride, float scale,
                          int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp8 temporal attention (BASELINE.json configs[4]).  Same chain as fmc_temporal_attn_fwd (attention_processor.py:267-281
 * reached from motion_module.py:365-389), with Q, K, V stored as OCP e4m3 bytes + one fp32 scale per tensor
 * (value = byte * scale).
 *
 * fmc_linear_fp8_qkv: the fused q | k | v projection `x @ w^T` (x [M, K] bf16 rows ldx apart, w [N = 3C, K] bf16, no bias:
 *   to_q / to_k / to_v have none) on the 8-phase bf16 MFMA kernel with an e4m3 epilogue: column block b = n / C is written as
 *   sat(acc * inv_scales[b]) to out_fp8 [M, N] bytes.  inv_scales: 3 device floats (1 / scale).  amax_bits (3 device words,
 *   may be NULL): running max |acc| per block as float bit patterns (atomicMax) -- what a delayed-scaling policy derives the
 *   next call's scales from.
 * fmc_temporal_attn_fp8_fwd: S^T = K Q^T on v_mfma_f32_16x16x32_fp8_fp8, softmax in fp32, P V on the bf16 MFMA with V
 *   converted e4m3 -> bf16 (exact); o is bf16.  q/k/v strides in bytes (multiples of 16), o strides in elements; scales = 3
 *   device floats {scale_q, scale_k, scale_v}.  F in {16, 32}.
 * fmc_temporal_attn_fp8_bwd: fmc_temporal_attn_bwd with e4m3 q, k, v (dequantised while staged; d_o, dq, dk, dv bf16).
 * fmc_fp8_scales_roll: the delayed-scaling update, one tiny launch: scale[b] = max(margin * amax[b] / 448, 1e-12),
 *   inv_scale[b] = 1 / scale[b], amax[b] = 0 for the three blocks (all fp32 device words; amax as written by the atomicMax of
 *   non-negative float bit patterns).
 * ------------------------------------------------------------------------------------------- */
int fmc_fp8_scales_roll(void* amax, void* scale, void* inv_scale, float margin, void* stream);
int fmc_linear_fp8_qkv(const void* x, const void* w, void* out_fp8, int64_t M, int N, int K, int64_t ldx,
                       const void* inv_scales, void* amax_bits, void* stream);
int fmc_temporal_attn_fp8_fwd(const void* q, const void* k, const void* v, void* o, const void* scales, int n_clips,
                              int n_pix, int F, int H, int D, int64_t clip_stride, int64_t frame_stride,
                              int64_t pix_stride, int64_t o_clip_stride, int64_t o_frame_stride, int64_t o_pix_stride,
                              float scale, void* stream);
int fmc_temporal_attn_fp8_bwd(const void* q, const void* k, const void* v, const void* scales, const void* d_o, void* dq,
                              void* dk, void* dv, int n_clips, int n_pix, int F, int H, int D, int64_t clip_stride,
                              int64_t frame_stride, int64_t pix_stride, int64_t do_clip_stride, int64_t do_frame_stride,
                              int64_t do_pix_stride, int64_t dq_clip_stride, int64_t dq_frame_stride,
                              int64_t dq_pix_stride, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layout pass of the 3x3-convolution WEIGHT gradient (the trainable convs of the OMC Adapter / camera encoder;
 * `loss.backward()` at train_cam_obj_ctrl.py:915).  dW[co][dy][dx][ci] = sum_pixels dY[pixel, co] * X[pixel + tap, ci] is a GEMM
 * whose reduction index is the pixel; this pass writes a [n, H, W, C] bf16 tensor channel-major over a zero-padded pixel index
 * p = (img * (H + 2) + y + 1) * Wp + x + 1, Wp = round_up(W + 2, 8):
 *     dst[s][c][guard + p] = src[img][y][x + s - shifts / 2]   (0 in the padding and up to row_len),
 * shifts = 3 copies (dx = -1, 0, +1) for X, 1 for dY.  dW of one kernel row dy is then ONE
 * fmc_linear_bf16(x = dst_X viewed as [3 Cin, row_len] + guard + (dy - 1) * Wp, w = dst_dY [Cout, K], K = row_len of dY)
 * with split_k over the pixels: out[dx * Cin + ci][co].  (`hip_ops.conv3x3_weight_grad`)
 * ------------------------------------------------------------------------------------------- */
int fmc_nhwc_to_cmajor_padded(const void* src, void* dst, int n_img, int H, int W, int C, int64_t row_len, int guard,
                              int shifts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The fused temporal attention block of the motion modules at the 40x64 level (round 4; csrc/temporal_block.hip).
 * Replaces, for one attention block of `TemporalTransformerBlock.forward` (fmc/models/motion_module.py:287-300) as driven by
 * `TemporalSelfAttention.forward` (:349-389) and `PoseAdaptorAttnProcessor.forward` / `AttnProcessor.__call__`
 * (fmc/models/attention_processor.py:202-293 / :20-82):
 *     n = LayerNorm(h) + pe[frame];   m = merge_scale * (n W_m^T) + pose_term + n   (only with w_merge_tm);   q | k | v = m W_qkv^T;
 *     o = softmax(q k^T * scale) v per (pixel, head) over the frames;   out = o W_out^T + b_out + h
 * in ONE launch: a 160-row tile (10 pixels x 16 frames) stays in LDS from h to out, q / k / v / scores / o live in registers.
 *   h, out, pose_term: bf16 [n_clips, frames, hw, channels] (channels-last video tokens; out may not alias h);
 *   ln_gamma fp32 [C]; ln_bpe fp32 [frames][C] = LayerNorm beta + positional-encoding row (pe zero when the block has none);
 *   w_merge_tm / w_out_tm: the [C, C] weights tile-major [C / 32][C][32] (`hip_ops._w_tilemajor`); pose_term = merge_scale * (W_m pose + b_m);
 *   w_qkv_packed: the fused [3 C, C] projection in MFMA-fragment order per head (`hip_ops.pack_temporal_qkv`);
 *   ln_stats (optional): fp32 [rows][2] = (mean, rstd) of every out row with eps ln_stats_eps, for a consumer GEMM that applies the next
 *   LayerNorm itself (fmc_linear_bf16_lnc).
 * Shapes: frames == 16, channels == 320, heads == 8, hw % 10 == 0 (the 40x64 level of the 16x320x512 configurations); anything else is
 * FMC_E_SHAPE -- callers keep the un-fused chain there.  bf16 only.  MFMA bound: 2 rows C (C [merge] + 4 C) + 4 rows F C flops.
 * ------------------------------------------------------------------------------------------- */
int fmc_temporal_block_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_bpe, float ln_eps, const void* w_merge_tm,
                            const void* pose_term, float merge_scale, const void* w_qkv_packed, const void* w_out_tm, const void* b_out,
                            float* ln_stats, float ln_stats_eps, int n_clips, int frames, int hw, int channels, int heads, float scale,
                            void* stream);
/* ---- fused text cross-attention block of the spatial transformer at the 20x32 level (round 4) ----------------------------------------------------
 * Replaces, for `attn2` of diffusers' BasicTransformerBlock as the reference U-Net builds it (fmc/models/unet_blocks.py:323-333; Attention /
 * AttnProcessor: fmc/models/attention_processor.py:20-82, :148-154 with LoRA):
 *     out = to_out(softmax(to_q(LayerNorm(h)) k^T * scale) v) + b_out + h        (k | v = the text's fused to_k / to_v projection)
 * in ONE launch (was LayerNorm + to_q GEMM + cross-attention kernel + to_out GEMM): the skeleton of the C = 640 temporal block (80-row tiles resident
 * in LDS, weights in fragment order straight into registers, wave = head), k / v fragments of the <= 80 text tokens held in registers.
 *   h, out: bf16 [n_images][hw][640] tokens, hw % 80 == 0; ln_gamma fp32 [640]; ln_bpe fp32 [16][640], every row = the LayerNorm beta;
 *   w_q_packed / w_out_frag: `hip_ops.pack_w_frag80` of the [640, 640] weights; kvfrag: fmc_xattn_pack_kv of the text k | v (one per text row);
 *   images_per_text: images that share a text row (frames of a clip).
 * fmc_xattn_pack_kv: kv bf16 [batch][S <= 80][1280] (batch stride ld_batch elements) -> bf16 [batch][8 heads][12800] MFMA fragments. */
int fmc_xattn_block640_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_bpe, float ln_eps, const void* w_q_packed,
                            const void* kvfrag, const void* w_out_frag, const void* b_out, int n_images, int hw, int n_keys, int images_per_text,
                            float scale, void* stream);
int fmc_xattn_pack_kv(const void* kv, void* out, int batch, int S, int64_t ld_batch, void* stream);
/* The same block at the 40x64 level (C = 320, 8 heads x 40) on the skeleton of fmc_temporal_block_bf16 (160-row tiles, persistent): hw % 160 == 0;
 * w_q_packed = `hip_ops.pack_xattn_q40` (per head [10 k-steps][q0 | q1 | (q tail, zeros)][lane][8]), w_out_tm tile-major (`_w_tilemajor`),
 * kvfrag = fmc_xattn_pack_kv40 (kv bf16 [batch][S <= 80][640] -> [batch][8][7680]); ln_stats (optional): (mean, rstd) of every out row. */
int fmc_xattn_block320_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_bpe, float ln_eps, const void* w_q_packed,
                            const void* kvfrag, const void* w_out_tm, const void* b_out, float* ln_stats, float ln_stats_eps, int n_images, int hw,
                            int n_keys, int images_per_text, float scale, void* stream);
int fmc_xattn_pack_kv40(const void* kv, void* out, int batch, int S, int64_t ld_batch, void* stream);
/* ---- LayerNorm + GEGLU projection of a feed-forward at the 20x32 level, A operand resident (round 4) ----------------------------------------------
 * out[M][cff] = (n W_v^T + b_v) * gelu(n W_g^T + b_g), n = LayerNorm(h): `GEGLU.forward` of diffusers' FeedForward behind norm3 / ff_norm
 * (fmc/models/motion_module.py:295-299; BasicTransformerBlock).  A workgroup keeps its 80 normalised rows in LDS and walks the cff / 320 column chunks
 * with the chunk's weight rows streamed in MFMA-fragment order (`hip_ops.pack_geglu_frag80`) -- replaces fmc_layernorm_fwd + fmc_linear_bf16(GEGLU).
 *   h bf16 [M][640], M % 80 == 0; out bf16 [M][cff] row-major, cff % 320 == 0; bias bf16 [2 cff] (value | gate) or NULL.
 *   out_blocked != 0 (M % 160 == 0): out is written tile-major, [M / 160][cff / 32][160][32] -- the layout fmc_linear_bf16_ffblk reads with x_blocked
 *   (the intermediate is private to the feed-forward; its second GEMM then requests contiguous 10-KiB operand blocks). */
int fmc_geglu640_ln_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_packed, const void* bias,
                         int64_t M, int cff, int out_blocked, void* stream);
/* The same at the 40x64 level: h bf16 [M][320], cff % 160 == 0; 4 waves and 77 KiB of LDS per workgroup, two workgroups per CU. */
int fmc_geglu320_ln_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_packed, const void* bias,
                         int64_t M, int cff, int out_blocked, void* stream);
/* Diagnostic: `buf` = device buffer of [workgroups][4][8] int64 that receives s_memrealtime stamps (100 MHz) of wave 0 at the phase boundaries of
 * its first four tiles (tools/scratch/r04/probe_tb.py); NULL switches the stamps off (default). */
int fmc_temporal_block_set_debug(void* buf);

#ifdef __cplusplus
}
#endif
#endif /* FMC_HIP_H */
