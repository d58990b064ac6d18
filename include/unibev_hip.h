/*
 * unibev_hip.h — C ABI of libunibev_hip.so: the MI355X (gfx950) kernels behind UniBEV's BEV-encoder
 * hot path.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless the parameter name ends in `_host`.
 *   - Tensors are dense, row-major, in the shapes written next to each parameter.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous
 *     and never synchronise the device.
 *   - The caller owns every buffer.  "accumulated" outputs must be zeroed by the caller.
 *   - Return value: 0 on success, a negative ubv_status otherwise; ubv_last_error() returns a
 *     thread-local message for the last failure.  No call aborts the process.
 *   - dtype codes (ubv_dtype) describe the element type of `value`-like tensors; sampling
 *     locations, attention weights, reference points and all gradients w.r.t. them are float32.
 *
 * Each entry point names the reference interface it replaces.  Reference paths are relative to
 * /root/reference/projects/UniBEV/unibev_plugin/ ("[ext]" = the un-vendored mmcv-full 1.3.17 /
 * mmdet3d 0.18.1 symbol reached from that call site).
 */
#ifndef UNIBEV_HIP_H_
#define UNIBEV_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { UBV_F32 = 0, UBV_F16 = 1, UBV_BF16 = 2 } ubv_dtype;

typedef enum {
  UBV_OK = 0,
  UBV_ERR_INVALID = -1,      /* bad argument / unsupported shape */
  UBV_ERR_LAUNCH = -2,       /* HIP runtime error at launch */
  UBV_ERR_UNSUPPORTED = -3   /* valid request this build has no kernel for */
} ubv_status;

int ubv_version(void);
const char* ubv_last_error(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char* ubv_arch(void);

/* Test aid (no reference counterpart): launches a kernel that writes `pattern` to 64 KB of LDS in 2048 blocks, so that
 * the LDS a following kernel is handed holds those bits (e.g. 0x7fc00000, a NaN) instead of whatever ran before.  Used by
 * tests/test_lift_gpu.py to show that no kernel consumes LDS it did not fill. */
int ubv_debug_fill_lds(uint32_t pattern, void* stream);

/* Test aid (no reference counterpart): a synthetic co-runner for the two-stream hazard study (tools/ab/lift_concurrent.py,
 * profiles/r05_pk_mfma_hazard.txt).  kind 0 back-to-back bf16 MFMAs; 1 LDS stores + barriers + b128 reads + MFMAs (the
 * skeleton of ubv_gemm_nt without global memory); 2 the same without MFMAs; 3 streaming reads of src[n_floats];
 * 4 ds_read_b64_tr_b16 loop; 5 scalar-f32 VALU loop; 6 v_cvt_pk_bf16_f32 loop; 7 MFMAs on 8 independent accumulators;
 * 8 streaming copy of src's first half onto its second; 9 v_pk_fma_f32 loop; 10 v_pk_fma_f32 between MFMAs; 11 scalar v_fma_f32
 * between MFMAs.  blocks x 256 threads, lds_bytes of dynamic LDS (>= 40 KB). */
int ubv_debug_aggressor(int kind, int iters, int blocks, int lds_bytes, const float* src, int64_t n_floats, float* sink,
                        void* stream);

/* Test aid: shader-clock stamps of the weight-stationary GEMM's tile loop (csrc/gemm_ws.hip, launches made with
 * UBV_WS_ABL=16): out_host [2 blocks][16 tiles][8 stamps]. */
int ubv_debug_ws_timing(uint64_t* out_host);

/* Study / test aid: which weight-gradient kernel ubv_gemm_wgrad* (`dense`) and ubv_spconv_wgrad_pairs (`sparse`) launch
 * from now on.  0: the 4-wave kernel, two blocks per CU.  4 or 8: the wave-specialised kernel (csrc/gemm_wgrad_ws.inl:
 * that many producer waves + 4 MFMA waves, one block per CU).  -1: leave as it is.  Start values: dense 0
 * (UBV_WGRAD_WS=1 with UBV_WGRAD_PW=4|8 changes it), sparse 0 (UBV_SPCONV_WGRAD_WS=0|4|8).  Same results either way;
 * ubv_gemm_wgrad_splits follows `dense`, so ask it again after a change.  Not thread-safe. */
int ubv_debug_set_wgrad_ws(int dense, int sparse);

/* Optional per-kernel timing: while enabled, every kernel of the sampling family is bracketed by
 * HIP events on its launch stream.  ubv_profile_read() synchronises on them and writes one line per
 * kernel name: "name<TAB>launches<TAB>total_ms<TAB>algorithmic_bytes_per_launch\n"; returns the
 * number of bytes the full text needs.  ubv_profile_enable(0|1|2) also clears the records; 2 = the scopes around
 * whole operators only ("bev_lift_fwd<…", "…_op<…"): the per-kernel scopes record their events between an
 * operator's launches, so an operator's own duration is read from a level-2 pass. */
int ubv_profile_enable(int on);
int64_t ubv_profile_read(char* out, int64_t capacity);

/* ------------------------------------------------------------------------------------------------
 * k1 — multi-scale deformable attention sampling.
 * Replaces [ext] mmcv `_ext.ms_deform_attn_forward` / `ms_deform_attn_backward`, loaded at
 * models/modules/spatial_cross_attention_img.py:19-20, spatial_cross_attention_pts.py:19-20,
 * decoder.py:29-30 and called through MultiScaleDeformableAttnFunction.apply at
 * spatial_cross_attention_img.py:433-435, spatial_cross_attention_pts.py:440-442, decoder.py:325-327.
 *
 *   value            [B, S, H, Dh]        dtype
 *   spatial_shapes   [L, 2] int64 (h, w)  (device, as in mmcv)
 *   level_start      [L]    int64         (device)
 *   sampling_loc     [B, Nq, H, L, P, 2]  f32, (x, y) normalised to [0, 1]
 *   attn_weight      [B, Nq, H, L, P]     f32
 *   out              [B, Nq, H*Dh]        dtype
 * out[b,q,h*Dh+c] = sum_{l,p} w * bilinear(value_l[b,:,h,c], x = loc_x*W_l - 0.5, y = loc_y*H_l - 0.5),
 * zero padding per corner (== F.grid_sample(bilinear, zeros, align_corners=False) on 2*loc-1).
 * `im2col_step` of the mmcv signature is accepted and ignored (no batch chunking is needed).
 */
int ubv_ms_deform_attn_forward(const void* value, const int64_t* spatial_shapes,
                               const int64_t* level_start, const float* sampling_loc,
                               const float* attn_weight, void* out, int B, int S, int H, int Dh,
                               int L, int Nq, int P, int dtype, int im2col_step, void* stream);

/*   grad_out          [B, Nq, H*Dh]       dtype
 *   grad_value        [B, S, H, Dh]       f32, ACCUMULATED (caller zeroes; cast to dtype by caller)
 *   grad_sampling_loc [B, Nq, H, L, P, 2] f32, written
 *   grad_attn_weight  [B, Nq, H, L, P]    f32, written
 */
int ubv_ms_deform_attn_backward(const void* value, const int64_t* spatial_shapes,
                                const int64_t* level_start, const float* sampling_loc,
                                const float* attn_weight, const void* grad_out, float* grad_value,
                                float* grad_sampling_loc, float* grad_attn_weight, int B, int S,
                                int H, int Dh, int L, int Nq, int P, int dtype, int im2col_step,
                                void* stream);

/* The same backward for ONE level whose shape (fh, fw) the HOST knows (mmcv's signature hands spatial_shapes over as a
 * device tensor; [ext] mmcv ms_deform_attn_cuda_backward reads it on the host, ops/csrc/pytorch/cuda/ms_deform_attn_cuda.cu).
 * grad_value is computed by OWNER TILES — the sampling points are binned by the 8x8-pixel tile they touch and each tile
 * accumulates its points on the matrix cores and stores its pixels once — instead of one f32 atomic per (corner,
 * channel): grad_value is WRITTEN (no zeroing needed), bit-reproducible up to the arrival order inside a tile.
 * grad_sampling_loc / grad_attn_weight as above.  Shapes: L == 1, fh * fw == S, Dh in {16, 32}, P in {4, 8}, H * Dh /
 * (16 bytes of channels) dividing 64; anything else returns UBV_ERR_UNSUPPORTED (-3) and the caller keeps
 * ubv_ms_deform_attn_backward.  `workspace`: ubv_ms_deform_attn_backward_workspace(...) bytes of device scratch (0: shape
 * not covered). */
int64_t ubv_ms_deform_attn_backward_workspace(int B, int fh, int fw, int H, int Dh, int Nq, int P, int dtype);
int ubv_ms_deform_attn_backward_planned(const void* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                        const float* sampling_loc, const float* attn_weight, const void* grad_out,
                                        float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int B,
                                        int S, int H, int Dh, int L, int Nq, int P, int dtype, int fh, int fw,
                                        void* workspace, int64_t workspace_bytes, void* stream);

/* The operator with a QUERY-GRID hint (round 5): the caller states that the Nq queries are a qgrid_h x qgrid_w grid in
 * row-major order — true at the reference's BEV call sites (models/modules/spatial_cross_attention_pts.py:439-442: the
 * queries are the bev_h x bev_w BEV grid; the encoder's self-attention likewise) — and that the single level is fh x fw.
 * mmcv's signature cannot say either; with them the operator runs on the TILE plan of the fused lifting kernels
 * (csrc/bev_lift_tile.hip: block = 8x8 query tile x head, one lane per sampling point, the tile's pixel box of the value
 * map in LDS; backward: d(locations), d(weights) from the same block, grad_value by owner tiles, no f32 atomics).
 * Covered: f32, L == 1, H == 8, Dh == 32, P in {4, 8}, qgrid_h * qgrid_w == Nq (ubv_ms_deform_attn_grid_supported).
 * forward_grid falls back to ubv_ms_deform_attn_forward for anything else; backward_grid returns UBV_ERR_UNSUPPORTED and
 * the caller takes ubv_ms_deform_attn_backward[_planned].  grad_value is WRITTEN; workspace:
 * ubv_ms_deform_attn_backward_grid_workspace(...) bytes. */
int ubv_ms_deform_attn_grid_supported(int H, int Dh, int L, int P, int dtype, int fh, int fw, int Nq, int qgrid_h,
                                      int qgrid_w);
int ubv_ms_deform_attn_forward_grid(const void* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                    const float* sampling_loc, const float* attn_weight, void* out, int B, int S, int H,
                                    int Dh, int L, int Nq, int P, int dtype, int fh, int fw, int qgrid_h, int qgrid_w,
                                    void* stream);
int64_t ubv_ms_deform_attn_backward_grid_workspace(int B, int fh, int fw, int H, int Dh, int Nq, int P, int dtype,
                                                   int qgrid_h, int qgrid_w);
int ubv_ms_deform_attn_backward_grid(const void* value, const float* sampling_loc, const float* attn_weight,
                                     const void* grad_out, float* grad_value, float* grad_sampling_loc,
                                     float* grad_attn_weight, int B, int S, int H, int Dh, int L, int Nq, int P, int dtype,
                                     int fh, int fw, int qgrid_h, int qgrid_w, void* workspace, int64_t workspace_bytes,
                                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused BEV query lifting (one level): offsets -> locations, logits -> softmax, sampling,
 * per-camera accumulation and the camera mean, in one kernel; nothing but the GEMM outputs is
 * materialised.  Replaces the tensor arithmetic of
 *   MSDeformableAttention3DImg.forward   models/modules/spatial_cross_attention_img.py:385-435
 *   MSDeformableAttention3DPts.forward   models/modules/spatial_cross_attention_pts.py:383-442
 *   [ext] MultiScaleDeformableAttention (self-attn slot; vendored copy decoder.py:299-327)
 * and, for num_cams > 1, the re-batch / scatter-add / count-divide of
 *   SpatialCrossAttentionImg.forward     models/modules/spatial_cross_attention_img.py:141-212.
 *
 *   value     [B*Nc, S = fh*fw, H, Dh]  dtype   (batch-major, camera-minor)
 *   offsets   row (b,q) at element (b*Nq+q)*off_stride : [H, P, 2] raw Linear output
 *   logits    row (b,q) at element (b*Nq+q)*log_stride : [H, P]    raw Linear output
 *   offlog_dtype  element type of offsets / logits (and of their gradients in backward): UBV_F32,
 *             or `dtype` itself — under autocast the Linear emits 16-bit values and the kernels
 *             read them directly (bit-identical to up-casting first).  Rows 16-byte aligned.
 *   ref       [Nc, B, Nq, Z, 2] f32 reference points; flat point p uses anchor p % Z
 *             (quirk q3, spatial_cross_attention_img.py:407-419)
 *   vis0      [Nc, Nq] uint8 or NULL: camera i contributes to query q iff vis0[i,q] != 0
 *             (visibility of BATCH ELEMENT 0, quirk q1, :141-152)
 *   count     [B, Nq] f32 or NULL: the result is divided by count[b,q] (quirk q2, :209-212)
 *   out       [B, Nq, H*Dh] dtype
 *   loc = ref + offsets / (fw, fh);  w = softmax_P(logits)
 * qgrid_w/qgrid_h: if Nq == qgrid_w*qgrid_h the queries are walked in 8x8 tiles of that grid
 * (L2 locality); pass 0 for raster order.
 * Supported shapes: see ubv_bev_lift_supported() (Dh in {16, 32}, P in {4, 8}).
 * workspace: optional scratch of ubv_bev_lift_forward_workspace(...) bytes.  When it is given and
 * the shape qualifies (16-bit dtype, Dh = 32, P = 8, fh*fw <= 192: the 8x22 camera maps), the
 * gather runs on the matrix cores as out^T = V^T . A^T with a per-wave coefficient matrix built in
 * LDS, reading a fragment-ordered copy of `value` written into the scratch; otherwise (or with
 * NULL) the gather kernel runs.  Both give the same result up to the rounding of the
 * coefficients to `dtype`.
 */
int64_t ubv_bev_lift_forward_workspace(int B, int Nc, int fh, int fw, int H, int Dh, int P, int dtype);
int ubv_bev_lift_forward(const void* value, const void* offsets, int64_t off_stride,
                         const void* logits, int64_t log_stride, int offlog_dtype,
                         const float* ref, const uint8_t* vis0, const float* count, void* out, int B, int Nc, int fh,
                         int fw, int H, int Dh, int Nq, int P, int Z, int qgrid_w, int qgrid_h,
                         int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/*   grad_out     [B, Nq, H*Dh]       dtype
 *   grad_value   [B*Nc, S, H, Dh]    f32, WRITTEN (previous content ignored; zeroed internally
 *                                    only on the paths that accumulate)
 *   grad_value_lowp  [B*Nc, S, H, Dh] dtype or NULL (16-bit dtypes only): when given, the final
 *                grad_value is written HERE, rounded once from the f32 accumulation, and
 *                grad_value is only the f32 accumulation scratch (content unspecified on return).
 *   grad_offsets row (b,q) at element (b*Nq+q)*goff_stride : [H, P, 2] offlog_dtype, written
 *   grad_logits  row (b,q) at element (b*Nq+q)*glog_stride : [H, P]    offlog_dtype, written
 *   ref_is_grid  1 iff Nc == 1 and the Nq queries form the qgrid_w x qgrid_h BEV grid with
 *                reference points near their cell centres (BEV self-attention and SCA-pts).  Selects
 *                the GRID plan: sampling points binned by 8x8-pixel owner tile, every tile summed
 *                by one wave (MFMA) and stored once — no global atomics, no zeroing; exact for any
 *                offsets (bucket overflow goes through an atomic side path).  0 is always safe.
 *   slot_center  accepted for ABI stability, ignored (an earlier plan used it as a search hint).
 *   workspace    scratch of at least ubv_bev_lift_backward_workspace(...) bytes (may be NULL when
 *                that is 0): per-tile buckets of point records / per-camera visible-query lists
 *                and partial maps.
 *   visible_lists  NULL, or the per-camera visible-query lists of ubv_compact_visible(vis0): the
 *                CAMERA plan walks them; they depend on vis0 alone, so one compaction per forward
 *                pass serves every layer's backward (NULL: compacted here, 14 us per call).
 */
int64_t ubv_bev_lift_backward_workspace(int B, int Nc, int fh, int fw, int H, int Dh, int Nq, int P,
                                        int qgrid_w, int qgrid_h, int ref_is_grid);

/* Ordered compaction of each camera's visible queries: lists[cam*Nq + i] = i-th query (ascending)
 * with vis0[cam, q] != 0 (all queries when vis0 is NULL), lists[Nc*Nq + cam] = their number.
 * `lists` holds ubv_visible_lists_elems(Nc, Nq) int32.  Replaces the per-camera
 * `nonzero()` index lists of spatial_cross_attention_img.py:141-152 (six host syncs there). */
int64_t ubv_visible_lists_elems(int Nc, int Nq);
int ubv_compact_visible(const uint8_t* vis0, int Nc, int Nq, int32_t* lists, void* stream);
/* The same lists for queries that are a BEV grid qgrid_w wide (Nq = qgrid_w x rows, both multiples of 8): the visible
 * queries of a camera come TILE BY TILE (8 x 8 tiles of the grid, row-major inside a tile) instead of ascending, so a
 * batch of consecutive entries is a compact patch of the ground and of the camera's map (fewer touched pixel blocks
 * per batch in the CAMERA backward).  qgrid_w = 0, or a grid that is not whole tiles: ascending order. */
int ubv_compact_visible_grid(const uint8_t* vis0, int Nc, int Nq, int qgrid_w, int32_t* lists, void* stream);

int ubv_bev_lift_backward(const void* value, const void* offsets, int64_t off_stride,
                          const void* logits, int64_t log_stride, int offlog_dtype,
                          const float* ref, const uint8_t* vis0, const float* count,
                          const float* slot_center, const void* grad_out, float* grad_value,
                          void* grad_value_lowp, void* grad_offsets, int64_t goff_stride,
                          void* grad_logits, int64_t glog_stride, int B, int Nc, int fh, int fw,
                          int H, int Dh, int Nq, int P, int Z, int qgrid_w, int qgrid_h,
                          int ref_is_grid, int dtype, const int32_t* visible_lists, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* 1 if ubv_bev_lift_* has a kernel for this shape, else 0 (callers then compose k1). */
int ubv_bev_lift_supported(int H, int Dh, int P, int dtype);

/* ------------------------------------------------------------------------------------------------
 * Camera projection of the pillar reference points + visibility.
 * Replaces ImgEncoder.point_sampling, models/modules/encoder_unibev_detr_img.py:112-187, and the
 * `count` / per-camera index bookkeeping of spatial_cross_attention_img.py:141-153, 209-211.
 *
 *   lidar2img [B, Nc, 4, 4] f32
 *   xs [bev_w], ys [bev_h], zs [D] f32: normalised pillar coordinates exactly as
 *             get_reference_points builds them (encoder_unibev_detr_img.py:68-73)
 *   pc_range_host[6], img_h/img_w: img_metas[0]['img_shape'][0] (quirk q5, :166-167)
 *   ref_cam  [Nc, B, Nq, D, 2] f32 out      bev_mask [Nc, B, Nq, D] uint8 out
 *   vis0     [Nc, Nq] uint8 out (any_D bev_mask[:, 0])    count [B, Nq] f32 out
 *            count = clamp(sum_cams any_D(bev_mask), min=1)
 */
int ubv_point_sampling(const float* lidar2img, const float* xs, const float* ys, const float* zs,
                       const float* pc_range_host, float img_h, float img_w, float* ref_cam,
                       uint8_t* bev_mask, uint8_t* vis0, float* count, int B, int Nc, int bev_h,
                       int bev_w, int D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Feature flattening + embedding add.
 * Replaces UniBEVTransformer._pre_process_img_feats / _pre_process_pts_feats,
 * models/modules/transformer_fusion.py:230-278.
 *   in   [N, C, HW] dtype (N = bs*num_cams for images, bs for LiDAR BEV features)
 *   embA [groups, C] f32 or NULL, row (n % groups)   (cams_embeds, groups = num_cams)
 *   embB [C] f32 or NULL                              (level embedding of this level)
 *   out  [N, HW, C] dtype = in^T + embA + embB
 */
int ubv_flatten_embed_forward(const void* in, const float* embA, int groups, const float* embB,
                              void* out, int N, int C, int HW, int dtype, void* stream);
/*   grad_out [N, HW, C] dtype -> grad_in [N, C, HW] dtype (written);
 *   grad_emb [N, C] f32 ACCUMULATED (caller zeroes): per-n column sums of grad_out; the caller
 *   reduces over n for the parameter gradients. */
int ubv_flatten_embed_backward(const void* grad_out, void* grad_in, float* grad_emb, int N, int C,
                               int HW, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BEV fusion epilogue.  Replaces channel_feature_norm / spatial_feature_norm /
 * multi_modal_fusion and the final permute, models/modules/transformer_fusion.py:280-413, 549.
 *   img, pts [B, Nq, C] dtype (either may be NULL = missing modality, treated as zeros)
 *   cw_img, cw_pts [C] f32 per-channel factors (CNW softmax x modality flag; host-composed)
 *   sw_img, sw_pts [Nq] f32 per-query factors or NULL (SpatialNormWeights)
 *   cat == 0: out [Nq, B, C]   = cw_img*sw_img*img + cw_pts*sw_pts*pts
 *   cat == 1: out [Nq, B, 2C]  = [cw_img*sw_img*img , cw_pts*sw_pts*pts]
 */
int ubv_bev_fuse_forward(const void* img, const void* pts, const float* cw_img,
                         const float* cw_pts, const float* sw_img, const float* sw_pts, void* out,
                         int B, int Nq, int C, int cat, int dtype, void* stream);
/*   grad_out as `out`; grad_img/grad_pts [B,Nq,C] dtype (NULL to skip);
 *   grad_cw [2, C] f32 and grad_sw [2, Nq] f32 ACCUMULATED (NULL to skip). */
int ubv_bev_fuse_backward(const void* grad_out, const void* img, const void* pts,
                          const float* cw_img, const float* cw_pts, const float* sw_img,
                          const float* sw_pts, void* grad_img, void* grad_pts, float* grad_cw,
                          float* grad_sw, int B, int Nq, int C, int cat, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused residual + dropout + LayerNorm:  y = LN(identity + dropout(x)) * gamma + beta.
 * Replaces the tail `self.dropout(out) + identity` of every attention / FFN of an encoder layer
 * (models/modules/decoder.py:338, spatial_cross_attention_img.py:215, spatial_cross_attention_pts.py:206,
 * [ext] mmcv FFN) together with the following 'norm' of BaseTransformerLayer's operation_order
 * (encoder_unibev_detr_img.py:434-436).
 *   x        [R, C] dtype           identity, y, grad_y, grad_identity [R, C] stream_dtype
 *   stream_dtype  element type of the residual stream: UBV_F32, or `dtype` itself (the reference's
 *            fp16 mode keeps the stream in half, mmcv wrap_fp16_model); mean / variance / the
 *            normalisation are computed in f32 either way and y is rounded once
 *   gamma, beta [C] f32             mean, rstd [R] f32 (saved for backward)
 *   p        dropout probability (0 = eval); the keep mask is a stateless hash of (seed, element),
 *            so backward regenerates it from the same seed and nothing is stored
 *   seed_dev NULL, or a device pointer to a per-step base that is ADDED to `seed` by the kernel:
 *            a captured HIP graph bakes `seed` in, the base is advanced on the device between
 *            replays (forward and backward of one step must see the same value)
 *   grad_gamma, grad_beta [C] f32 ACCUMULATED (caller zeroes);  grad_x [R, C] dtype
 *   grad_x_colsum [C] f32 or NULL, ACCUMULATED: column sums of grad_x as stored — the bias
 *            gradient of the Linear that produced x (its backward then need not re-read grad_x)
 *   ordered_workspace  NULL: the column sums (grad_gamma, grad_beta, grad_x_colsum) are accumulated with f32 atomics, one
 *            per column per block — their last bits depend on the order the blocks retire in.  Otherwise a scratch
 *            buffer of ubv_add_dropout_layernorm_backward_workspace(C) bytes (contents irrelevant, used on `stream`):
 *            the blocks' partial sums are written there and added in block order by a second small launch —
 *            bit-reproducible 1-D parameter gradients.
 *   bcast_rows  0, or the number of rows x and identity really hold: they REPEAT over the R rows (row r reads row
 *            r % bcast_rows; R % bcast_rows == 0) — the first encoder layer's self-attention, whose queries are one
 *            table for every sample (encoder_unibev_detr_img.py: bev_query.unsqueeze(1).repeat(1, bs, 1)); y, mean,
 *            rstd and the dropout mask stay per row; the backward writes grad_x and grad_identity as [bcast_rows, C],
 *            SUMMED over the repeats.  Rows of 16, 32 or 64 16-byte lanes only (f32: C = 64, 128, 256).
 *   C % 4 == 0, C <= 1024.
 */
int64_t ubv_add_dropout_layernorm_backward_workspace(int C);
int ubv_add_dropout_layernorm_forward(const void* x, const void* identity, const float* gamma,
                                      const float* beta, void* y, float* mean, float* rstd,
                                      int64_t R, int64_t bcast_rows, int C, float eps, float p, uint64_t seed,
                                      const uint64_t* seed_dev, int dtype, int stream_dtype,
                                      void* stream);
int ubv_add_dropout_layernorm_backward(const void* grad_y, const void* x, const void* identity,
                                       const float* gamma, const float* mean, const float* rstd,
                                       void* grad_x, void* grad_identity, float* grad_gamma,
                                       float* grad_beta, float* grad_x_colsum, int64_t R, int64_t bcast_rows,
                                       int C, float p, uint64_t seed, const uint64_t* seed_dev, int dtype,
                                       int stream_dtype, void* ordered_workspace, void* stream);

/* FFN activation of the encoder layers, y = dropout(relu(x)) in one pass ([ext] mmcv FFN:
 * Sequential(Linear, ReLU, Dropout(ffn_drop)); configs/unibev: feedforward_channels=512,
 * ffn_dropout=0.1).  Same stateless keep mask as above; backward needs only y
 * (grad_x = grad_y / (1-p) where y != 0).  n elements of dtype, n a multiple of 16 bytes' worth. */
int ubv_relu_dropout_forward(const void* x, void* y, int64_t n, float p, uint64_t seed,
                             const uint64_t* seed_dev, int dtype, void* stream);
int ubv_relu_dropout_backward(const void* grad_y, const void* y, void* grad_x, int64_t n, float p,
                              int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * y[M, N] = x[M, K] . w[N, K]^T (+ bias[N]), row-major, all operands of `dtype`, f32 accumulation:
 * the forward of the encoder's Linear layers ([ext] torch.nn.functional.linear behind value_proj,
 * sampling_offsets / attention_weights, output_proj and the FFN under models/modules).  The GEMM
 * is hipBLASLt's; this entry point keeps its descriptor, layouts and algorithm per (M, N, K, dtype,
 * bias) so that a call costs one hipblasLtMatmul on the host (the framework path rebuilds them and
 * re-queries the heuristic on every call).  bias may be NULL.  workspace: ubv_linear_workspace()
 * bytes.  Returns UBV_ERR_UNSUPPORTED if hipBLASLt offers no algorithm for the shape.
 */
int64_t ubv_linear_workspace(void);
int ubv_linear_forward(const void* x, const void* w, const void* bias, void* y, int64_t M, int N,
                       int K, int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Hand-written MFMA GEMM of the same Linear layers:
 *   y[M, N] = x[M, K] . w[N, K]^T (+ bias[N]) (+ residual[M, N])
 * forward ([ext] torch.nn.functional.linear), and with w := the transposed weight the input gradient
 * dX = dY . W with the residual branch's gradient added in the epilogue.
 *   dtype UBV_F32: x, y, residual f32; the weight comes SPLIT into bf16 halves w_hi + w_lo (both
 *     [N, K], ubv_split_weight) and the product runs on the matrix cores as x_hi w_hi + x_hi w_lo +
 *     x_lo w_hi with f32 accumulation: ~2^-17 per product instead of f32's 2^-24, the BEV features
 *     of the full-size fixture stay at 3e-4 of the reference's (bar 1e-3) and the GEMM becomes
 *     HBM-bound instead of bound by the f32 MFMA rate.
 *   dtype UBV_F16 / UBV_BF16: x, w_hi, y, residual in that type, w_lo NULL, one product.
 *   bias f32 [N] or NULL; residual may alias y; ldx / ldw / ldy row strides in elements.
 * Needs K % 32 == 0, N % 32 == 0, 16-byte aligned rows; otherwise UBV_ERR_UNSUPPORTED (callers fall
 * back to ubv_linear_forward / the framework GEMM).
 */
int ubv_gemm_nt(const void* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw,
                const float* bias, const void* residual, void* y, int64_t ldy, int64_t M, int N, int K,
                int dtype, void* stream);
/* The fused value_proj | sampling_offsets | attention_weights GEMM of the BEV self-attention ([ext] mmcv
 * MultiScaleDeformableAttention.forward, vendored copy P/models/modules/decoder.py:283-312: three Linears on the same
 * query) and its backward, f32 data.  ubv_gemm_nt_dual is ubv_gemm_nt with optional extras: columns k >= k_split of X
 * come from x2 [M, K - k_split] (the input gradient of the concatenated weight from the two output gradients), columns
 * n >= n_split of Y go to y2 [M, N - n_split] (value to one tensor, offsets | logits to another; n_split a multiple of
 * 128, N need not fill the last column tile) and the row-periodic term of ubv_gemm_nt_rowbias then applies to y2 only.
 * ubv_gemm_wgrad_dual is ubv_gemm_wgrad with grad_out given as [M, n_split] + [M, N - n_split]. */
int ubv_gemm_nt_dual(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int k_split, const void* w_hi,
                     const void* w_lo, int64_t ldw, const float* bias, const void* residual, const void* row_bias,
                     int64_t row_period, int64_t row_ld, void* y, int64_t ldy, void* y2, int64_t ldy2, int n_split,
                     int64_t M, int N, int K, void* stream);
int ubv_gemm_wgrad_dual(const void* grad_out, const void* grad_out2, int n_split, const void* x, float* partials,
                        float* grad_wb, int64_t M, int N, int K, int splits, void* stream);

/* ubv_gemm_nt with a ROW-PERIODIC additive term: Y[m, :] = X[m, :] . W^T + bias + row_bias[m % row_period, :]
 * (row_bias [row_period, N] in Y's type, leading dimension row_ld).  Replaces `query + query_pos` in front of the
 * sampling_offsets / attention_weights Linears of the BEV self-attention ([ext] mmcv MultiScaleDeformableAttention
 * .forward: `query = query + query_pos`, vendored copy P/models/modules/decoder.py:283-284): query_pos is the same
 * for every sample of the batch, so (query + pos) . W^T = query . W^T + (pos . W^T)[q] with the second term computed
 * once per step for all layers. */
int ubv_gemm_nt_rowbias(const void* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw,
                        const float* bias, const void* row_bias, int64_t row_period, int64_t row_ld, void* y,
                        int64_t ldy, int64_t M, int N, int K, int dtype, void* stream);

/* The two GEMMs around the FFN activation ([ext] mmcv FFN: Sequential(Linear, ReLU, Dropout) -> Linear,
 * mmcv/cnn/bricks/transformer.py; configured at projects/UniBEV/configs/unibev/
 * unibev_nus_LC_cnw_256_modality_dropout.py:281-291, ffn_dropout 0.1) with the activation in the epilogue:
 *   act 1: y = dropout(relu(x w^T + bias), p)      — the keep mask of ubv_relu_dropout_forward for the same
 *          (seed, seed_dev) and the element's index in the contiguous [M, N] output (ldy == N);
 *   act 2: y = (x w^T) * 1/(1-p) where mask[m][n] != 0, else 0  — the input gradient of the Linear that
 *          FOLLOWS the activation (x = its grad_out, w = its transposed weight), already multiplied by the
 *          activation's derivative; mask [M, ldy] is the activation's saved output, in `dtype`.
 * Same shapes, dtypes and return codes as ubv_gemm_nt. */
int ubv_gemm_nt_act(const void* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw,
                    const float* bias, void* y, int64_t ldy, int64_t M, int N, int K, int dtype,
                    int act, const void* mask, float p, uint64_t seed, const uint64_t* seed_dev, void* stream);
/* Weight gradient of the same layers ([ext] torch.nn.Linear backward):
 *   dW[N, K] = sum_m grad_out[m, n] x[m, k],  db[n] = sum_m grad_out[m, n]
 * as a split-K MFMA product over `splits` slabs of rows (ubv_gemm_wgrad_splits picks the count that
 * fills the chip).  grad_out [M, N] and x [M, K] share `dtype` (f32: split-bf16 products as above).
 * partials [splits, N*K + N] f32: scratch (slab s holds its dW part followed by its N bias sums);
 * grad_wb [N*K + N] f32, WRITTEN: dW row-major, then db — the slabs summed by a second launch.
 * Rows are read 16 bytes at a time: N and K multiples of 4 (f32) / 8 (16-bit), buffers 16-byte aligned,
 * else UBV_ERR_UNSUPPORTED. */
int ubv_gemm_wgrad_splits(int64_t M, int N, int K);
int ubv_gemm_wgrad(const void* grad_out, const void* x, float* partials, float* grad_wb, int64_t M, int N,
                   int K, int splits, int dtype, void* stream);
/* f32 w [N, K] -> bf16 halves w_hi, w_lo [N, K] and (unless NULL) their transposes wt_hi, wt_lo
 * [K, N] for the input-gradient GEMM.  One launch per Linear per step. */
int ubv_split_weight(const float* w, int N, int K, void* w_hi, void* w_lo, void* wt_hi, void* wt_lo,
                     void* stream);
/* The same for n parameters in one launch (the weights of every Linear of a pass).  Entry i is a parameter
 * [rows, cols]; wh / wl point at its first row inside a (possibly concatenated) [N_total, cols] pair of halves, wth / wtl
 * (both NULL or both set) at its first COLUMN inside the [cols, N_total] transposed halves with leading dimension ld_t. */
int ubv_split_weights_batched(int n, const float* const* w, const int* rows, const int* cols, void* const* wh,
                              void* const* wl, void* const* wth, void* const* wtl, const int* ld_t, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Reductions behind the gradients of the encoder's Linear layers (value_proj, sampling_offsets,
 * attention_weights, output_proj, FFN; [ext] torch.nn.Linear backward in the reference), one launch
 * per Linear:
 *   grad_bias[n] += sum_m grad_out[m, n]     grad_out [rows, N] dtype (NULL to skip),
 *                                            grad_bias [N] f32 ACCUMULATED (caller zeroes)
 *   grad_weight[i] = sum_s partials[s, i]    partials [S, NK] dtype: the split-K slices of the
 *                                            weight-gradient GEMM (NULL to skip), grad_weight [NK] f32
 * N and NK multiples of 16 bytes' worth of elements, N <= 256 * that.
 */
/* out[i] = a[i] + b[i], f32, n elements, 16-byte aligned operands (out may alias a or b).  The sum of two gradients of
 * one tensor — what [ext] torch's autograd engine does with a framework add kernel when a parameter has two consumers
 * (here: a self-attention's sampling_offsets / attention_weights weights, used by the layer's GEMM and by the folded
 * positional term, encoder_unibev_detr_img.py:413-420 `query + query_pos`) — as a kernel of this library, for use inside
 * the two-stream window of the encoders (unibev_amd/debug.py). */
int ubv_add2_f32(const float* a, const float* b, float* out, int64_t n, void* stream);

/* out[r * ld + c] = sum over s < S of slices[s][r * cols + c], f32: the batch sum of S contiguous [rows, cols] blocks
 * written as a COLUMN SLICE (row stride ld >= cols) of a wider matrix.  The gradient of a term shared by the samples of a
 * batch — the positional part of `(query + query_pos) . W^T` (encoder_unibev_detr_img.py:413-420), folded into one GEMM
 * over the positional table for all layers — lands directly in its layer's columns of the fold's gradient matrix, so
 * that the fold's backward is ONE input-gradient GEMM and ONE weight-gradient pass with no framework cat in front.
 * cols, ld multiples of 4; 16-byte aligned operands. */
int ubv_slice_sum_f32(const float* slices, int S, int64_t rows, int cols, float* out, int64_t ld, void* stream);

int ubv_linear_grad_reduce(const void* grad_out, int64_t rows, int N, float* grad_bias,
                           const void* partials, int S, int64_t NK, float* grad_weight, int dtype,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * LiDAR front end.
 * Replaces [ext] mmdet3d ops built from `pts_voxel_layer` and called at
 * models/detectors/unibev_detector.py:163-167 (Voxelization -> hard_voxelize, deterministic),
 * :117 (HardSimpleVFE), and the SparseConvTensor.dense() scatter at the tail of SparseEncoder
 * (configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:194-208).
 */

/* Bytes of scratch ubv_hard_voxelize needs. */
int64_t ubv_hard_voxelize_workspace(int N, int max_points, int max_voxels);

/* points [N, F] f32 (x, y, z first).  Voxel coordinate per axis c = floor((p - min) / size) in
 * f32, point dropped unless 0 <= c < grid; voxels are numbered by first appearance in input order;
 * at most max_points points per voxel (input order), at most max_voxels voxels.
 *   voxels      [max_voxels, max_points, F] f32 out, zero padded
 *   coors       [max_voxels, 3] int32 out, (z, y, x)
 *   num_points  [max_voxels] int32 out
 *   voxel_num   [1] int32 out (device): number of voxels produced
 *   voxel_size_host[3], range_host[6]: host arrays; grid = round((max-min)/size)
 */
int ubv_hard_voxelize(const float* points, float* voxels, int32_t* coors, int32_t* num_points,
                      int32_t* voxel_num, void* workspace, int64_t workspace_bytes, int N, int F,
                      const float* voxel_size_host, const float* range_host, int max_points,
                      int max_voxels, void* stream);

/* A BATCH of clouds in one launch chain ([ext] mmdet3d runs Voxelization once per sample from a Python loop,
 * models/detectors/unibev_detector.py:163-167): cloud b = points_host[b] [n_host[b], F] (host arrays of device pointers
 * / counts, B <= 16).  Every sample is voxelized exactly as by ubv_hard_voxelize (same numbering, same caps, bit for
 * bit); outputs are per-sample slabs voxels [B, max_voxels, max_points, F], coors [B, max_voxels, 3], num_points
 * [B, max_voxels], voxel_num [B] (device).  workspace: ubv_hard_voxelize_batch_workspace(B, max_b n, ...) bytes. */
int64_t ubv_hard_voxelize_batch_workspace(int B, int n_max, int max_points, int max_voxels);
int ubv_hard_voxelize_batch(const float* const* points_host, const int* n_host, int B, float* voxels, int32_t* coors,
                            int32_t* num_points, int32_t* voxel_num, void* workspace, int64_t workspace_bytes, int F,
                            const float* voxel_size_host, const float* range_host, int max_points, int max_voxels,
                            void* stream);
/* The same chain with [ext] HardSimpleVFE on the way (models/detectors/unibev_detector.py:117, `pts_voxel_encoder`):
 * mean [B, max_voxels, F] f32 = the sum of each voxel's stored points / num_points (rows past voxel_num zero), written
 * by the gather launch — bit-identical to ubv_voxel_mean on the voxel slab. */
int ubv_hard_voxelize_batch_vfe(const float* const* points_host, const int* n_host, int B, float* voxels, int32_t* coors,
                                int32_t* num_points, int32_t* voxel_num, float* mean, void* workspace,
                                int64_t workspace_bytes, int F, const float* voxel_size_host, const float* range_host,
                                int max_points, int max_voxels, void* stream);

/* [ext] mmdet3d dynamic_voxelize: coors [N, 3] int32 (z, y, x), or (-1,-1,-1) when outside. */
int ubv_dynamic_voxelize(const float* points, int32_t* coors, int N, int F,
                         const float* voxel_size_host, const float* range_host, void* stream);

/* [ext] mmdet3d `dynamic_point_to_voxel_forward(feats, coors, reduce_type)` — DynamicScatter, the
 * "dynamic_scatter" of BASELINE.json's north_star (the caller that would
 * reach it is the dynamic branch of UniBEV.voxelize, models/detectors/unibev_detector.py:151-175;
 * no shipped config selects it).
 *   feats  [N, C] f32     coors [N, D] int32, D in 1..4 (z, y, x) or (batch, z, y, x);
 *                         coordinates < 2^21 (D <= 3) / 2^15 (D = 4)
 *   reduce_type  0 sum, 1 mean, 2 max
 * A point with any negative coordinate is dropped (map -1).  Voxels are the unique coordinate rows
 * in ascending lexicographic order (torch.unique_dim of the published op).  Outputs have capacity N:
 *   voxel_feats [N, C] f32, voxel_coors [N, D] int32, voxel_points_count [N] int32 — rows
 *   >= voxel_num[0] are unspecified;  point2voxel_map [N] int32;
 *   voxel_num [2] int32 (device): [0] = number of voxels M, [1] = number of valid points.
 * Deterministic: every voxel is reduced in input order (the published kernel adds atomically).
 * Nothing is read back to the host.  workspace: ubv_dynamic_scatter_workspace(N) bytes. */
int64_t ubv_dynamic_scatter_workspace(int N);
int ubv_dynamic_point_to_voxel_forward(const float* feats, const int32_t* coors, int N, int C, int D,
                                       int reduce_type, float* voxel_feats, int32_t* voxel_coors,
                                       int32_t* point2voxel_map, int32_t* voxel_points_count,
                                       int32_t* voxel_num, void* workspace, int64_t workspace_bytes,
                                       void* stream);

/* [ext] HardSimpleVFE: mean [M, F] = voxels[:, :, :F].sum(1) / num_points.  M is read from the
 * device counter `voxel_num` (rows >= *voxel_num are written as zeros); max_voxels bounds the launch. */
int ubv_voxel_mean(const float* voxels, const int32_t* num_points, const int32_t* voxel_num,
                   float* mean, int max_voxels, int max_points, int F, void* stream);

/* SparseConvTensor.dense(): dense [B, C, D, Hs, Ws] f32 (caller zeroes) <- feats [M, C] at
 * coors [M, 4] int32 (batch, z, y, x).  M read from device counter m_dev if non-NULL else m. */
int ubv_sparse_to_dense(const float* feats, const int32_t* coors, const int32_t* m_dev, int m,
                        float* dense, int B, int C, int D, int Hs, int Ws, void* stream);

/* ---- sparse 3-D convolutions of the LiDAR middle encoder -----------------
 * [ext] mmdet3d 0.18.1 SparseEncoder -> spconv SubMConv3d / SparseConv3d (un-vendored; configured at
 * projects/UniBEV/configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:194-208).  These entry points
 * replace spconv's ops.get_indice_pairs (the rulebook) and ops.indice_conv / indice_conv_backward (gather -
 * GEMM - scatter).  Voxel coordinates are int32 rows (batch, z, y, x), mmdet3d's `coors`.
 *
 * Rulebook = a dense neighbour map nbr[k][row] (int32, -1 = inactive; k = (kz * ky_size + ky) * kx_size + kx)
 * looked up in an open-addressing hash of the active voxels' keys:
 *   ubv_spconv_table_slots(n)        power-of-two slot count for n voxels
 *   ubv_spconv_hash_build            coors [n, 4] of a (D, H, W) map -> table_keys [slots] / table_vals [slots]
 *   ubv_spconv_neighbors             rows of `row_dims` looked up in the table of `target_dims`:
 *        transposed = 0: target = row * stride - pad + k      (output rows -> inputs; SubM: stride 1, pad k/2)
 *        transposed = 1: target = (row + pad - k) / stride where divisible (input rows -> outputs: the map
 *                        of the input gradient)
 *   ubv_spconv_output_sites          the output set of a strided convolution, built on the device: inputs mark one bit
 *                                    per output cell they reach, a popcount scan ranks the bits, out_coors [cap, 4]
 *                                    come out in ascending key order and count_dev[0] holds their number.  The input
 *                                    count is read from n_dev when given (else n), so the layers of an encoder chain
 *                                    back to back and the host reads all counts in one copy.  bitmap:
 *                                    ubv_spconv_sites_words(B, out_dims) int32, sums: words / 256 + 2 int32 (scratch).
 *   ubv_spconv_pairs                 compacted rulebook of a neighbour map (the weight gradient's form): per offset the
 *                                    rows with a neighbour, in row order — out_rows[k][i] = row, in_rows[k][i] =
 *                                    nbr[k][row] for i < counts[k] (entries past the count are not written);
 *                                    chunk_sums: kvol * ubv_spconv_pairs_chunks(rows) int32 of scratch.
 * Product:
 *   ubv_spconv_gather_mma            out[row, :] = sum_k feats[nbr[k][row], :] . w[k]^T on the matrix cores.
 *        feats [*, Cin] (dtype), w_hi (and w_lo for f32: split-bf16 halves) [kvol, 32*ceil(Cout/32), Cin] in the
 *        16-bit operand type, rows past Cout zero; out [rows, Cout] (dtype), WRITTEN.  Cin % 16 == 0,
 *        Cout <= 128.  The same call computes the input gradient (feats = grad_out, nbr = the transposed map,
 *        w = the [k][Cin][Cout] weight as stored). */
int64_t ubv_spconv_table_slots(int64_t n);
/* Weight gradient: grad_w[k][co][ci] = sum_rows grad_out[row][co] * feats[nbr[k][row]][ci] — the split-K MFMA
 * weight-gradient kernel of the Linear layers with gathered rows, one launch for all kernel offsets.
 * partials [splits, kvol, Cout*Cin + Cout] f32 scratch (ubv_spconv_wgrad_splits picks `splits`);
 * grad_w [kvol, Cout*Cin + Cout] f32 WRITTEN, each block's Cout trailing entries unused.  Cout, Cin <= 128,
 * multiples of 4 (f32) / 8 (16-bit). */
int ubv_spconv_wgrad_splits(int64_t rows, int kvol);
int ubv_spconv_wgrad(const void* grad_out, const void* feats, const int32_t* nbr, int64_t ld, int64_t rows,
                     float* partials, float* grad_w, int Cout, int Cin, int kvol, int splits, int dtype, void* stream);

/* The same weight gradient over COMPACTED (output row, input row) pairs — spconv's rulebook form (spconv 1.x
 * `indice_pairs` / `indice_pair_num`, built by get_indice_pairs; mmdet3d/ops/spconv/functional.py
 * SparseConvFunction.backward -> indice_conv_backward multiplies only the pairs of each offset).  Per offset k the
 * first counts[k] entries of out_rows[k][.] / in_rows[k][.] (leading dimension ld) name the rows of grad_out and
 * feats to multiply; slabs past the count store zero tiles at once.  On LiDAR clouds ~70 % of the (row, offset)
 * slots of a submanifold convolution have no neighbour: this form skips them (9.8 -> 3.x ms per encoder backward). */
int ubv_spconv_wgrad_pairs(const void* grad_out, const void* feats, const int32_t* in_rows, const int32_t* out_rows,
                           const int32_t* counts, int64_t ld, int64_t rows, float* partials, float* grad_w, int Cout,
                           int Cin, int kvol, int splits, int dtype, void* stream);

/* BatchNorm1d (+ ReLU) over the rows of a sparse feature matrix x [N, C] — the norm / activation between the sparse
 * convolutions ([ext] mmdet3d make_sparse_convmodule / SparseBasicBlock: BatchNorm1d(eps 1e-3, momentum 0.01), ReLU;
 * torch.nn.functional.batch_norm semantics: biased variance for the normalisation, unbiased for running_var).
 * training != 0: batch statistics (mean / rstd [C] are WRITTEN, running_* updated when given); else mean / rstd are
 * INPUTS (the caller derives them from the running statistics).  y = relu?(gamma * (x - mean) * rstd + beta).
 * partial: ubv_rows_bn_partial_elems(C) floats of scratch.  Deterministic (fixed-order two-level sums).
 * backward: grad_x, grad_gamma [C], grad_beta [C] written; relu != 0 masks grad_y where the forward output was 0.
 * residual [N, C] (optional): y = relu?(bn(x) + residual) — the tail of mmdet3d's SparseBasicBlock
 * (out = relu(bn2(conv2(.)) + identity)); its backward takes the stored output y_out (the ReLU mask) and also writes
 * grad_residual = the masked grad_y. */
int64_t ubv_rows_bn_partial_elems(int C);
int ubv_rows_bn_forward(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float* mean, float* rstd, float* partial, void* y, int64_t N, int C,
                        float eps, float momentum, int relu, int training, int dtype, void* stream);
int ubv_rows_bn_backward(const void* x, const void* grad_y, const void* y_out, const float* gamma, const float* beta,
                         const float* mean, const float* rstd, float* partial, float* grad_gamma, float* grad_beta,
                         void* grad_x, void* grad_residual, int64_t N, int C, int relu, int dtype, void* stream);
int ubv_spconv_hash_build(const int32_t* coors, int64_t n, int D, int H, int W, int64_t* table_keys,
                          int32_t* table_vals, int64_t slots, void* stream);
int ubv_spconv_neighbors(const int32_t* coors, int64_t rows, int B, const int* row_dims, const int* target_dims,
                         const int* ksize, const int* stride, const int* pad, int transposed,
                         const int64_t* table_keys, const int32_t* table_vals, int64_t slots, int32_t* nbr,
                         int64_t ld, void* stream);
int64_t ubv_spconv_sites_words(int B, const int* out_dims);
int ubv_spconv_output_sites(const int32_t* coors, const int32_t* n_dev, int64_t n, int B, const int* in_dims,
                            const int* out_dims, const int* ksize, const int* stride, const int* pad, int32_t* bitmap,
                            int64_t words, int32_t* sums, int32_t* out_coors, int64_t cap, int32_t* count_dev,
                            void* stream);
int64_t ubv_spconv_pairs_chunks(int64_t rows);
int ubv_spconv_pairs(const int32_t* nbr, int64_t ld, int64_t rows, int kvol, int32_t* chunk_sums, int32_t* out_rows,
                     int32_t* in_rows, int32_t* counts, void* stream);
/* w [kvol][Cin][Cout] (spconv's stored layout, dtype) -> ubv_spconv_gather_mma's operand [kvol][32*ceil(rows/32)][K]:
 * transpose != 0 (forward): rows = Cout, K = Cin; transpose == 0 (input gradient): rows = Cin, K = Cout, offsets mirrored
 * when flip != 0 (submanifold layers).  f32: bf16 halves w_hi + w_lo; 16-bit: w_hi only (w_lo NULL). */
int ubv_spconv_weight_operand(const void* w, int kvol, int Cin, int Cout, int transpose, int flip, int dtype,
                              void* w_hi, void* w_lo, void* stream);
int ubv_spconv_gather_mma(const void* feats, const int32_t* nbr, int64_t ld, int64_t rows, const void* w_hi,
                          const void* w_lo, void* out, int Cin, int Cout, int kvol, int dtype, void* stream);

/* ---- optimizer step over flat f32 buffers -----------------------------------------------------------
 * [ext] mmcv OptimizerHook(grad_clip = dict(max_norm = 35, norm_type = 2)) -> torch.nn.utils.clip_grad_norm_, and
 * torch.optim.AdamW (reference: projects/UniBEV/configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:455-462).
 *   ubv_sumsq_f32     out[0] = sum x[i]^2 (deterministic two-stage reduction; workspace of
 *                     ubv_sumsq_workspace() bytes)
 *   ubv_adamw_flat    *step += 1 (device-side int64), then AdamW on p / m / v [n] from g [n]; when `sumsq` is given
 *                     the gradient is first scaled by min(1, max_norm / (sqrt(*sumsq) + 1e-6)) — nothing read back.
 *   ubv_adamw_flat_groups   the same step with torch.optim param_groups (what mmcv's DefaultOptimizerConstructor makes
 *                     of the config's paramwise_cfg = dict(custom_keys = {'img_backbone': dict(lr_mult = 0.1)}),
 *                     ...cnw_256_modality_dropout.py:455-462): range k of the buffers is [group_end[k-1], group_end[k])
 *                     with its own lr and weight decay (HOST arrays of n_groups entries, ends ascending, the last one
 *                     = n, at most ubv_adamw_flat_max_groups() ranges); one step counter and one clip coefficient. */
int64_t ubv_sumsq_workspace(void);
int ubv_sumsq_f32(const float* x, int64_t n, float* out, void* workspace, void* stream);
int ubv_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int64_t* step, const float* sumsq, float max_norm, void* stream);
int ubv_adamw_flat_max_groups(void);
int ubv_adamw_flat_groups(float* p, const float* g, float* m, float* v, int64_t n, int n_groups, const int64_t* group_end,
                          const float* group_lr, const float* group_weight_decay, float beta1, float beta2, float eps,
                          int64_t* step, const float* sumsq, float max_norm, void* stream);

/* ---- GridMask (the device-side image augmentation) --------------------------
 * Reference: models/utils/grid_mask.py:70-123 (GridMask.forward; built at models/detectors/unibev_detector.py:75):
 * y = x * mask over `planes` = n*c images of h x w, mask = the centre crop of a (1.5 h x 1.5 w) grid of stripes
 * of width l and period d starting at st_h / st_w (rows when use_h, columns when use_w), inverted when mode == 1.
 * The five integers are the reference's np.random draws (the host keeps its draw order); no rotation (the
 * detector's rotate = 1 always draws 0).  x may equal y. */
int ubv_grid_mask(const void* x, void* y, int64_t planes, int h, int w, int d, int l, int st_h, int st_w,
                  int use_h, int use_w, int mode, int dtype, void* stream);

/* ---- Modulated deformable convolution, DCNv2 (ResNet-101 stages 3-4) -------------
 * Reference: [ext] mmcv `modulated_deform_conv_forward / _backward` (ops/csrc/pytorch/modulated_deform_conv.cpp,
 * kernels in common/cuda/modulated_deform_conv_cuda_kernel.cuh) under `ModulatedDeformConv2dPack`, selected by
 * `dcn=dict(type='DCNv2', deform_groups=1)` at configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:229-236.
 * Layouts differ from mmcv's on purpose (channels-last, tap-major columns):
 *   x        [N, H, W, C]                      channels-last feature map
 *   offset   [N, dg*2*kh*kw, Ho, Wo]           as the offset convolution emits it: (dy, dx) pairs per tap
 *   mask     [N, dg*kh*kw, Ho, Wo]             modulation (after the sigmoid)
 *   columns  [N*Ho*Wo, kh*kw*C]                column k*C + c = mask * bilinear sample of channel c at tap k
 * The convolution is then y[N*Ho*Wo, Cout] = columns . Wp^T with Wp = weight.permute(0,2,3,1) ([Cout, kh*kw*C],
 * ubv_gemm_nt), born channels-last.  Backward: grad_columns = grad_y . Wp (ubv_gemm_nt), ubv_dcn_col2im produces
 * grad_x (f32, ZEROED by the caller: owner tiles + atomics for far offsets) and grad_offset / grad_mask (f32),
 * grad_weight = ubv_gemm_wgrad(grad_y, columns).  C / dg must be a multiple of 4 (f32) or 8 (16-bit). */
int ubv_dcn_im2col(const void* x, const void* offset, const void* mask, void* columns, int N, int H, int W, int C,
                   int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                   int deform_groups, int dtype, void* stream);
int ubv_dcn_col2im(const void* grad_columns, const void* x, const void* offset, const void* mask, float* grad_x,
                   float* grad_offset, float* grad_mask, int N, int H, int W, int C, int Ho, int Wo, int kh, int kw,
                   int sh, int sw, int ph, int pw, int dh, int dw, int deform_groups, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIBEV_HIP_H_ */
