/* vmapstep.h - C ABI of the MI355X-native per-object training step (libvmapstep.so).
 *
 * The reference (kxhit/vMAP) has no plugin/FFI interface: the hot path is inlined in train.py.  The
 * de-facto operator boundary this library replaces is
 *
 *   state      utils.py:30-34   update_vmap -> stacked parameters (leading object dimension)
 *   forward    train.py:293-294 vmap(pe_model)(...), vmap(fc_model)(...)
 *   loss       train.py:303-306 loss.step_batch_loss(...)                (loss.py:5-62)
 *   backward   train.py:324     batch_loss.backward()
 *   optimiser  train.py:325     optimiser.step()  (torch.optim.AdamW built at train.py:67)
 *   step loop  train.py:270-277 data_idx = slice(i*R, (i+1)*R) over the per-frame sample tensors
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors in the Python host code);
 *     the library never allocates, frees or synchronises: it only enqueues kernels on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream of the CURRENT device).  A call runs on the device
 *     that owns `stream`, whatever device is current on the calling thread (which is restored on return);
 *   - strides are in ELEMENTS; tensors may be the non-contiguous slices train.py:271-277 produces;
 *   - return value 0 = ok, negative = error; vmapstep_last_error() describes the last failure of the
 *     calling thread; no C++ exception crosses the ABI;
 *   - data-dependent decisions the reference takes with host syncs (render_rays.py:68-73 "any object has
 *     an empty mask", :88-90 "loss explode -> exit(-1)") are reported through `flags`, on the device.
 */
#ifndef VMAPSTEP_H
#define VMAPSTEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMAPSTEP_ABI_VERSION 7
#define VMAPSTEP_NUM_FC 14 /* field-MLP tensors per object, nn.Module.parameters() order (model.py:28-49) */

#define VMAPSTEP_OK 0
#define VMAPSTEP_ERR_ARGUMENT (-1)    /* null pointer / inconsistent shape */
#define VMAPSTEP_ERR_UNSUPPORTED (-2) /* shape the device kernels do not implement */
#define VMAPSTEP_ERR_WORKSPACE (-3)   /* workspace too small / misaligned */
#define VMAPSTEP_ERR_DEVICE (-4)      /* HIP runtime error while enqueueing */

/* flags[4] written per step */
#define VMAPSTEP_FLAG_DROP_DEPTH 0   /* render_rays.py:68-73 fired for the depth term (whole batch) */
#define VMAPSTEP_FLAG_DROP_COLOUR 1
#define VMAPSTEP_FLAG_DROP_OPACITY 2
#define VMAPSTEP_FLAG_EXPLODE 3      /* render_rays.py:88-90: a per-object loss term exceeded 1e5 */

/* Measurement / test overrides of the automatic launch plan.  Passed PER CALL through vmapstep_shape::tuning (NULL or
 * all-zero = automatic); the library keeps no tuning state of its own, so two operators in one process (two streams,
 * two devices, two threads) cannot influence each other.  The plan - and with it the workspace layout - depends on these
 * values: pass the same tuning to vmapstep_workspace_bytes and to every call that uses that workspace. */
#define VMAPSTEP_KERNEL_AUTO 0    /* hidden 32: step_main_s32; hidden 64: step_main_wp, hidden 128: step_main_ws (all: bf16 matrix pipe, split operands,
                                     float32-equivalent forward); other widths: the exact-fp32 kernels below               */
#define VMAPSTEP_KERNEL_GEN 1     /* hidden 64..256: step_main_gen (one wave per 32-point tile)                      */
#define VMAPSTEP_KERNEL_WIDE4 2   /* hidden 128/256: step_main_wide<4> (one tile per workgroup, four waves per tile) */
#define VMAPSTEP_KERNEL_H32_F32 4 /* hidden 32: step_main_h32 on the exact-fp32 matrix instruction instead of the default
                                     step_main_s32 (bf16 matrix pipe, split operands, float32-equivalent forward)      */
#define VMAPSTEP_KERNEL_WS1 5     /* hidden 64 / 128: step_main_ws (one wave per output block; the default at hidden 128)            */
#define VMAPSTEP_KERNEL_S32_BWD6 8 /* (7 was a measurement prototype of ABI v4 and stays invalid)
                                    * hidden 32, float32 weights: step_main_s32 with the SIX-product backward (hi.lo + lo.hi + mid.mid on top of
                                    * the default's three: the backward is then float32-equivalent, ~2^-24, like the forward; ~+15 % kernel time).
                                    * The reference is float32 end to end (train.py:64-66): this is the strictly comparable form of the default kernel */
#define VMAPSTEP_KERNEL_WP 6      /* hidden 64 / 128: step_main_wp (two waves per block, partial sums exchanged through LDS; the
                                     default at hidden 64)                                                               */
/* (3 and 7 were measurement prototypes of earlier versions - step_main_wide<2>, 16-point forward tiles - and are refused.) */
typedef struct vmapstep_tuning {
    int32_t workgroups_per_object; /* 0 = automatic (256 / n_obj, at most one per ray group)                      */
    int32_t kernel;                /* VMAPSTEP_KERNEL_*                                                           */
    int32_t generic_finalize;      /* 1: step_finalize instead of the table-driven step_finalize_h32 (A/B parity); */
                                   /* hidden 64 / 128 / 256: step_finalize_ws always with a thread per quad and    */
                                   /* row group, never its one-thread-per-quad form (A/B: same bits either way)    */
    int32_t ws_flags;              /* step_main_ws, measurement: bit 0 = never use single-tile rounds, bit 1 =     */
                                   /* always three-tile rounds (hidden 128), bit 2 = never three-tile rounds (A/B) */
} vmapstep_tuning;

typedef struct vmapstep_shape {
    int32_t n_obj;   /* objects in the stack                         (len(obj_dict))        */
    int32_t rays;    /* rays per object per step, R                  (cfg.n_per_optim)      */
    int32_t samples; /* samples per ray, S = n_bins_cam2surface+n_bins                      */
    int32_t hidden;  /* hidden width H                               (hidden_feature_size)  */
    int32_t weight_dtype; /* VMAPSTEP_WEIGHTS_F32 or VMAPSTEP_WEIGHTS_BF16 (see below)              */
    int32_t reserved;     /* 0                                                                       */
    const vmapstep_tuning* tuning; /* NULL = automatic                                               */
} vmapstep_shape;

/* weight_dtype: the reference is fp32 only (AMP = False, train.py:64).  BF16 = BASELINE configs[3]/[4] "bf16 weights +
 * fp32 accumulate": master parameters and optimiser state stay fp32, the values the kernels compute from (the packed
 * parameter image) are the masters rounded to bfloat16 (round-to-nearest-even); products and sums are fp32.  Parity is
 * defined against the fp32 oracle evaluated on the rounded weights. */
#define VMAPSTEP_WEIGHTS_F32 0
#define VMAPSTEP_WEIGHTS_BF16 1

/* One stacked tensor: object k starts at ptr + k * obj_stride (elements); each object's block is dense. */
typedef struct vmapstep_tensor {
    float* ptr;
    int64_t obj_stride;
} vmapstep_tensor;

/* The 15 trainable stacked tensors (or same-shaped gradients): fc_param of utils.py:31 + B_layer.weight. */
typedef struct vmapstep_params {
    vmapstep_tensor fc[VMAPSTEP_NUM_FC]; /* [n,H,87] [n,H] [n,H,H] [n,H] [n,H,H+87] [n,H] [n,H,H] [n,H]
                                            [n,1,H] [n,1] [n,H,H+42] [n,H] [n,3,H] [n,3]             */
    vmapstep_tensor pe_B;                /* [n,21,3]  embedding.py:75-76 */
} vmapstep_params;

/* One step's ray batch: the six tensors of train.py:271-277 (strides in elements, outermost first). */
typedef struct vmapstep_batch {
    const float* pcs;          int64_t pcs_stride[4];        /* [n,R,S,3]  sample points, object frame  */
    const float* z;            int64_t z_stride[3];          /* [n,R,S]    sample depths                */
    const float* gt_depth;     int64_t gt_depth_stride[2];   /* [n,R]                                   */
    const float* gt_rgb;       int64_t gt_rgb_stride[3];     /* [n,R,3]    already divided by 255       */
    const uint8_t* sem;        int64_t sem_stride[2];        /* [n,R]      0 other, 1 this, 2 unknown   */
    const uint8_t* depth_mask; int64_t depth_mask_stride[2]; /* [n,R]      bool as bytes                */
    /* ABI v7 - the sampler -> step hand-off as RAYS instead of points (second half; vmap.py:452-457): when
     * pcs == NULL the kernels rebuild sample point s of ray r of object k as (ray_o + ray_d * z[k][r][s]) - center[k], every
     * operation rounded on its own - the arithmetic of vmap.py:452-454 and of vmapstep_sample_frame, so the result is bit-identical
     * to reading the points it replaces; 24 + 4 S bytes per ray instead of 16 S (64 instead of 160 at S = 10).                     */
    const float* ray_o;        int64_t ray_o_stride[3];      /* [n,R,3]    ray origin, world frame (the keyframe's camera centre) */
    const float* ray_d;        int64_t ray_d_stride[3];      /* [n,R,3]    ray direction, world frame                             */
    const float* center;       int64_t center_stride;        /* [n,3]      obj_center (vmap.py:454); NULL = zeros; elements between objects */
} vmapstep_batch;

typedef struct vmapstep_outputs {
    float* loss;          /* [n_steps]     required: batch loss (loss.py:60)                      */
    int32_t* flags;       /* [n_steps][4]  required                                               */
    float* render_depth;  /* [n,R]    optional (NULL = skip): loss.py:27, last step only          */
    float* render_color;  /* [n,R,3]  optional: loss.py:30                                        */
    float* opacity;       /* [n,R]    optional: loss.py:31                                        */
    float* var;           /* [n,R]    optional: loss.py:28-29                                     */
    float* loss_terms;    /* [n,4]    optional, last step only: per object the depth, colour and opacity terms BEFORE their
                             weights (render_rays.py:87's three reductions) and l_batch = depth + colour_scaling * colour +
                             opacity_scaling * opacity (loss.py:59).  A ray-sharded caller sums them over ranks together
                             with the gradients and hands the sums to vmapstep_adamw_apply.                              */
} vmapstep_outputs;

/* torch.optim.AdamW hyper-parameters + state for the fused update (train.py:67, :325). */
typedef struct vmapstep_adamw {
    float lr, beta1, beta2, eps, weight_decay;
    int32_t step;        /* number of updates already applied to this stack (0 for a fresh update_vmap) */
    float* exp_avg;      /* [n][padded_params] first moments  (see vmapstep_param_layout)               */
    float* exp_avg_sq;   /* [n][padded_params] second moments                                           */
    /* Optional, for callers that REPLAY a captured frame call (hipGraph: kernel arguments are frozen, so the step count
     * cannot come from `step`): a device table bias_table[t] = { lr / (1 - beta1^(t+1)), sqrt(1 - beta2^(t+1)) } for
     * t = 0 .. table_len - 1 (later steps use the last entry: build it until both factors stop changing in float32) and a
     * device int32 step_counter[2] = { updates applied before the call, steps of the previous call not yet folded in }.
     * vmapstep_train_steps then takes the step count from the device (its first launch folds [1] into [0] and sets [1] =
     * n_steps) and ignores `step`.  NULL = the host-side count above.  Not accepted by the prepared / apply entry points. */
    const float* bias_table;
    int32_t table_len;
    int32_t* step_counter;
} vmapstep_adamw;

const char* vmapstep_last_error(void);
int vmapstep_abi_version(void);

/* sizes[15]: element count per object of the 14 field tensors then B_layer.weight; their sum -> *params;
 * *padded_params = row length of the optimiser-state slabs. */
int vmapstep_param_layout(int32_t hidden, int64_t sizes[VMAPSTEP_NUM_FC + 1], int64_t* params, int64_t* padded_params);

/* Bytes of scratch the calls below need for `shape` and up to `max_steps` steps per call (256-byte aligned). */
int vmapstep_workspace_bytes(const vmapstep_shape* shape, int32_t max_steps, size_t* bytes);

/* The launch plan the library derives from `shape` (and its tuning): which fused-step kernel runs and how the batch is cut
 * into workgroups.  Diagnostics (ABI v6): logging, benchmarks and tests ask instead of re-deriving the rules; no device needed. */
typedef struct vmapstep_plan_info {
    char kernel[48];               /* e.g. "step_main_s32", "step_main_ws<4>", "step_main_wp<2>", "step_main_gen"            */
    int32_t rays_per_round;        /* rays a workgroup takes per round (pass)                                                 */
    int32_t rounds_per_object;     /* ceil(rays / rays_per_round)                                                             */
    int32_t workgroups_per_object; /* = partial-gradient rows per object the finalize sums                                    */
    int32_t tiles_per_round;       /* step_main_ws: 32-point tiles per round (1, 2 or 3); other kernels: 0                    */
    int32_t waves_per_workgroup;
    int32_t single_round;          /* 1: every workgroup runs exactly one round (the specialised kernel forms apply)          */
} vmapstep_plan_info;
int vmapstep_describe_plan(const vmapstep_shape* shape, int32_t max_steps, vmapstep_plan_info* info);

/* One step of train.py:293-306 + :324: loss and the gradients of all 15 stacked tensors (written to `grads`,
 * which has the layout of `params`; replaces loss.backward() populating p.grad). */
int vmapstep_fwd_bwd(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                     const vmapstep_batch* batch, float color_scaling, float opacity_scaling,
                     const vmapstep_params* grads, const vmapstep_outputs* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Forward + loss only (no gradients): rendered depth / colour / opacity / variance and the batch loss. */
int vmapstep_render(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                    const vmapstep_batch* batch, float color_scaling, float opacity_scaling,
                    const vmapstep_outputs* out, void* workspace, size_t workspace_bytes, void* stream);

/* The step loop of train.py:270-326 for one frame: for i in [0, n_steps): batch rays [i*ray_step, i*ray_step+R)
 * of the per-frame tensors described by `frame` -> forward, loss, backward, fused AdamW update of `params` in
 * place (opt->step is the count BEFORE the first of these steps; the caller adds n_steps afterwards).
 * `grads` may be NULL (gradients are then consumed by the fused update only); if given it receives the
 * gradients of the LAST step. out->loss / out->flags receive one entry per step. */
int vmapstep_train_steps(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                         const vmapstep_batch* frame, int64_t ray_step, int32_t n_steps,
                         float color_scaling, float opacity_scaling, const vmapstep_adamw* opt,
                         const vmapstep_params* grads, const vmapstep_outputs* out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Multi-GPU split of vmapstep_train_steps (objects sharded over ranks).  The only quantity of the
 * step that couples objects is the batch-wide "any object has an empty mask" switch (render_rays.py:68-73):
 *   1. vmapstep_prepare(...)        per-step mask statistics + switches of THIS rank's objects -> workspace;
 *   2. the caller max-reduces the int32[n_steps][4] array at (char*)workspace + *flags_offset across ranks
 *      (one tiny collective per frame, outside the per-step path);
 *   3. vmapstep_train_steps_prepared(...) runs the step loop on the prepared (and reduced) workspace.
 * Single-GPU callers just use vmapstep_train_steps. */
int vmapstep_prepare(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_batch* frame,
                     int64_t ray_step, int32_t n_steps, void* workspace, size_t workspace_bytes,
                     size_t* flags_offset, void* stream);
/* Byte offset in the workspace of the float[n_steps][n_obj][4] mask counts vmapstep_prepare wrote
 * (N_depth&obj, N_obj, N_sem, 0).  A RAY-sharded caller (the shared background model, train.py:308-316, trained
 * data-parallel) sum-reduces them over ranks and rewrites the switches (count == 0) before the prepared calls. */
int vmapstep_workspace_counts_offset(const vmapstep_shape* shape, int32_t max_steps, size_t* counts_offset);
/* vmapstep_fwd_bwd for step `step_index` of a prepared frame: `batch` is that step's ray batch (the caller applies the
 * slice), the mask counts / switches are the (possibly reduced) ones vmapstep_prepare wrote for that step, and the
 * packed parameter image is the one vmapstep_prepare built and vmapstep_adamw_apply keeps current. */
int vmapstep_fwd_bwd_prepared(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                              const vmapstep_batch* batch, int32_t step_index, float color_scaling, float opacity_scaling,
                              const vmapstep_params* grads, const vmapstep_outputs* out,
                              void* workspace, size_t workspace_bytes, void* stream);
/* torch.optim.AdamW's update of `params` from EXTERNALLY provided gradients (train.py:325 for a model whose gradients
 * were summed over ranks by the caller: the shared background model, train.py:308-316): `grad_slab` holds, per object,
 * the gradients in flat parameter order (the 14 field tensors then B_layer.weight), one row of grad_stride =
 * padded_params floats (vmapstep_param_layout) per object, 16-byte aligned.  Also rewrites the packed parameter image in
 * `workspace`, so that the next vmapstep_fwd_bwd_prepared of the frame reads the updated weights.  opt->step = updates
 * already applied.
 * `loss_terms` (optional, [n][4] as written by vmapstep_outputs::loss_terms and summed over ranks): the same launch then
 * also produces the step's GLOBAL result - out->loss[0] = sum of l_batch over objects (loss.py:60) and out->flags[0..3] =
 * the (already reduced) empty-mask switches of step `step_index` of the prepared frame + the "loss explode" test of
 * render_rays.py:88-90 on the summed terms - so a ray-sharded step needs no launch besides forward/backward, the
 * collective and this one. */
int vmapstep_adamw_apply(const vmapstep_shape* shape, const vmapstep_params* params, const float* grad_slab,
                         int64_t grad_stride, const vmapstep_adamw* opt, const float* loss_terms, int32_t step_index,
                         float color_scaling, float opacity_scaling, const vmapstep_outputs* out,
                         void* workspace, size_t workspace_bytes, void* stream);
int vmapstep_train_steps_prepared(const vmapstep_shape* shape, const vmapstep_params* params,
                                  const vmapstep_tensor* pe_scale, const vmapstep_batch* frame, int64_t ray_step,
                                  int32_t n_steps, float color_scaling, float opacity_scaling,
                                  const vmapstep_adamw* opt, const vmapstep_params* grads,
                                  const vmapstep_outputs* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- batched depth-guided sampler --------------------------------------------------------
 * Replaces the per-object loop train.py:208-218 -> vmap.py:319-459 (get_training_samples / sample_3d_points) and the
 * torch.stack of train.py:255-260: ONE launch writes the six per-frame tensors of ALL objects, contiguous
 * [n, F*P, ...], ready for vmapstep_train_steps.  Random numbers come from a counter-based Philox4x32-10 keyed by
 * (seed, frame_counter, object, ray): parity with the reference's torch generator is statistical; with
 * `test_randoms` the per-ray numbers are supplied by the caller and the result is deterministic (used by the tests
 * against the reference's own sampler). */
typedef struct vmapstep_sample_object {      /* one entry per object, array lives in DEVICE memory */
    const uint8_t* rgbs;       /* [K][W][H][4] RGB + pixel state (vmap.py:143-156)  */
    const float* depth;        /* [K][W][H]                                          */
    const float* t_wc;         /* [K][4][4]                                          */
    const float* bbox;         /* [K][4] u lo, u hi, v lo, v hi                      */
    int32_t n_keyframes;
    int32_t last2[2];          /* the two latest keyframe slots (vmap.py:329-331)    */
    float center[3];           /* obj_center                                         */
    int32_t obj_id;            /* shared-store mode: instance id of this object      */
    /* Shared frame store (one copy of every frame for all objects instead of vmap.py:143-176's per-object buffers):
     * slots == NULL: rgbs/depth/t_wc are this object's own [K] buffers.  slots != NULL: they are the store's arrays,
     * keyframe k of the object is store slot slots[k] (< 256), byte 3 of a pixel is unused and the pixel state comes
     * from the store's instance image `inst` (train.py:128-130). */
    const int32_t* slots;      /* [K] or NULL                                        */
    const int32_t* inst;       /* [C][W][H] or NULL                                  */
} vmapstep_sample_object;

typedef struct vmapstep_sample_cfg {
    int32_t width, height;             /* W, H of the keyframe images                                  */
    int32_t frames, samples_per_frame; /* F = n_iter_per_frame * win_size, P = n_samples_per_frame     */
    int32_t n_bins_cam2surface, n_bins;
    float fx, fy, cx, cy;
    float min_depth, surface_eps, stop_eps;
} vmapstep_sample_cfg;

typedef struct vmapstep_sample_randoms {     /* test mode, device pointers, any may be NULL */
    const int32_t* kf_ids;     /* [n][F]            */
    const float* u_w;          /* [n][F*P]          */
    const float* u_h;          /* [n][F*P]          */
    const float* u_z;          /* [n][F*P][S]       */
    const float* g_z;          /* [n][F*P][n_bins]  */
} vmapstep_sample_randoms;

/* `workspace` (optional, >= vmapstep_sample_workspace_bytes(n_obj), 4-byte aligned): with it the frame is sampled by as many
 * workgroups per object as fill the chip (two launches: the objects' maximum sampled depths - the one quantity that couples an
 * object's rays, vmap.py:391 - joined by an order-independent atomic max, then the samples); without it one workgroup per
 * object does everything.  Same pixels, same samples, same bits either way. */
int vmapstep_sample_workspace_bytes(int32_t n_obj, size_t* bytes);
int vmapstep_sample_frame(const vmapstep_sample_cfg* cfg, const vmapstep_sample_object* objects_device, int32_t n_obj,
                          float* pcs, float* z, float* gt_depth, float* gt_rgb, uint8_t* sem, uint8_t* depth_mask,
                          uint64_t seed, uint32_t frame_counter, const vmapstep_sample_randoms* test_randoms,
                          void* workspace, size_t workspace_bytes, void* stream);
/* ABI v7 - the same sampler handing the frame over as RAYS (second half: "or just o, dir, z = 64 B/ray instead of
 * 160"): ray_o / ray_d [n, F*P, 3] (world-frame origin and direction of every ray, vmap.py:31-41, :507-516) and center [n, 3] (the objects'
 * obj_center; optional) instead of the points tensor; the step kernels rebuild pcs = (ray_o + ray_d * z) - center themselves
 * (vmapstep_batch::ray_o / ray_d / center with pcs == NULL), bit-identical to the points this call would have written.  `pcs` is optional
 * here (non-NULL: written as well - tests).  Same pixels, samples, random numbers as vmapstep_sample_frame. */
int vmapstep_sample_frame_rays(const vmapstep_sample_cfg* cfg, const vmapstep_sample_object* objects_device, int32_t n_obj,
                               float* ray_o, float* ray_d, float* center, float* pcs,
                               float* z, float* gt_depth, float* gt_rgb, uint8_t* sem, uint8_t* depth_mask,
                               uint64_t seed, uint32_t frame_counter, const vmapstep_sample_randoms* test_randoms,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ---- inference query -------------------------------------------------------------------
 * Occupancy sigmoid(alpha) and colour of ONE object's field (object `obj_index` of the stacked tensors) at `n_points`
 * arbitrary points of the object frame: what Trainer.eval_points (trainer.py:77-95) computes chunk by chunk for mesh
 * extraction.  Any supported width (32: LDS-resident weights; 64..256: weights streamed from the L2-resident image).
 * workspace >= vmapstep_query_workspace_bytes(hidden). */
int vmapstep_query_workspace_bytes(int32_t hidden, size_t* bytes);
int vmapstep_query_points(int32_t hidden, const vmapstep_params* params, const vmapstep_tensor* pe_scale, int32_t obj_index,
                          const float* points, int64_t n_points, const int64_t points_stride[2],
                          float* occupancy, float* color, void* workspace, size_t workspace_bytes, void* stream);

/* Measurement hook: vmapstep_train_steps with every launch of the dominant kernel timed in the real step sequence (prep,
 * then main / finalize alternating); waits for the device and returns average durations in milliseconds:
 * main_kernel_ms[0] = the dispatch's own begin -> end timestamps (events attached to the launch with hipExtLaunchKernel:
 * what a rocprofv3 kernel trace of the same run reports per dispatch, and what bench.py's roofline uses),
 * main_kernel_ms[1] = a pair of stream events recorded around the launch (includes the cost of the events themselves). */
int vmapstep_profile_train_steps(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                                 const vmapstep_batch* frame, int64_t ray_step, int32_t n_steps,
                                 float color_scaling, float opacity_scaling, const vmapstep_adamw* opt,
                                 const vmapstep_outputs* outputs, void* workspace, size_t workspace_bytes, void* stream,
                                 float main_kernel_ms[2]);

/* Measurement hook: step_prep once, then the dominant kernel (step_main, forward+backward) `reps` times back to
 * back on `stream` with nothing in between, so that events recorded around the call give its average launch
 * duration (bench.py's roofline figure).  Writes only to the workspace. */
int vmapstep_profile_main_kernel(const vmapstep_shape* shape, const vmapstep_params* params,
                                 const vmapstep_tensor* pe_scale, const vmapstep_batch* batch, int32_t reps,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostics: one forward+backward launch with in-kernel shader-clock stamps. timing receives
 * [workgroups][4 waves][16 marks] uint32 (s_memtime low word) and *n_workgroups the launch's grid size;
 * timing_elems is the capacity of the buffer in uint32 elements. */
int vmapstep_profile_phases(const vmapstep_shape* shape, const vmapstep_params* params,
                            const vmapstep_tensor* pe_scale, const vmapstep_batch* batch,
                            uint32_t* timing, size_t timing_elems, int32_t* n_workgroups,
                            void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VMAPSTEP_H */
