/*
 * dmf.h — C ABI of the MI355X-native dual-modal fusion hot path (libdmf_hip.so).
 *
 * The reference (salalalala23/Dual-modal-fusion) is pure Python and defines NO FFI: its plug-in boundary
 * is the model loader `importlib.import_module('model.' + net_name).Net(args=cfg)`
 * (solver/mainsolver.py:31-34) whose product is called as `net(ms, pan)` (mainsolver.py:52) and trained by
 * `CrossEntropyLoss` + `loss.backward()` + `Adam.step()` (mainsolver.py:49-55).  This library sits one level
 * below that boundary: dual-modal-fusion_amd/model/gmfnet.py::Net binds these entry points with ctypes
 * (INTEGRATION.md shows the stub).  Each entry point names the reference call site it replaces.
 *
 * Conventions: every function returns 0 on success, non-zero on error (text via dmf_last_error());
 * all pointers except `shape`/`in` are DEVICE pointers owned by the caller; no allocation and no host
 * synchronisation inside; `stream` is a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef DMF_H
#define DMF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMF_VERSION 300   /* 0.3.0 (round 3): dmf_train_plan_steps, dmf_forward_ce, tagged-word exchange (dmf_xgmi_sizes grew), one patch kernel; 0.2.0: dmf_input.half, unit-gradient step, loss scaler, SGD / RMSprop steps */
#define DMF_KMAX 64       /* max number of logits (Categories_Number, utils/config.py:25) */

/* Network / patch geometry (oracle/gmfnet_ref.py::arch_from_cfg). */
typedef struct dmf_shape {
  int32_t C;    /* primary bands           (DATA_DICT[city].size[2], config.yml:77-80)          */
  int32_t C2;   /* auxiliary bands                                                               */
  int32_t P;    /* patch_size              (config.yml:27)                                       */
  int32_t S;    /* aux / primary resolution ratio (reference hard-codes 4, dataset.py:166)       */
  int32_t F;    /* feature width                                                                 */
  int32_t G;    /* spectral groups                                                               */
  int32_t H;    /* hidden width of the head                                                      */
  int32_t K;    /* Categories_Number (logits), <= DMF_KMAX                                       */
  int32_t attention;   /* 0 = late fusion only, 1 = cross-modal attention block                  */
  int32_t heads;       /* attention heads (config.yml:70), head_dim = E / heads                  */
  int32_t E;           /* attention embed dim (config.yml:69)                                    */
  int32_t reserved;
} dmf_shape;

/* Where a batch of patches comes from. */
typedef struct dmf_input {
  int32_t mode;        /* 0: materialised patches (the reference dataloader's tensors, dataset.py:168-185)
                          1: gather from the resident padded scene by pixel coordinates            */
  int32_t B;           /* patches in this batch                                                    */
  /* mode 0 */
  const float* a;      /* [B, C, P, P]        band-major patches                                   */
  const float* b;      /* [B, C2, S*P, S*P]                                                        */
  /* mode 1 */
  const float* sceneA; /* [Hp, Wp, C]   padded, normalised primary scene, pixel-major (function.py:99-117) */
  const float* sceneB; /* [HpB, WpB, C2] padded aux scene                                          */
  const int32_t* xy;   /* [B, 2] top-left pixel (x = row, y = col) of each patch (dataset.py:171-172) */
  int32_t Wp;          /* row pitch of sceneA in pixels                                            */
  int32_t WpB;         /* row pitch of sceneB in pixels                                            */
  /* optional (mode 1): device int holding the index of the current batch inside a pre-uploaded epoch plan;
     when non-NULL, patch b reads xy[(*cursor)*B + b] and labels[(*cursor)*B + b].  Lets a captured
     hipGraph of many steps be replayed without host-side pointer updates. */
  const int32_t* cursor;
  /* mode 1: sceneA holds IEEE fp16 [Hp, Wp, C] instead of fp32 (half the gather bytes).  The first 1x1 conv then runs on
     fp16 operands (window and weights rounded to nearest even) with fp32 accumulation; everything behind it, gradients
     included, stays fp32 (oracle: cfg['gmf']['half'] = 1).  mode 0 with half != 0: the fp32 patches `a` are rounded to
     fp16 as they are staged.  Shapes: dmf_half_supported. */
  int32_t half;
  int32_t reserved;
} dmf_input;

int32_t dmf_version(void);
const char* dmf_last_error(void);

/* 0 if a compiled kernel instance exists for this shape. */
int32_t dmf_shape_supported(const dmf_shape* shape);

/* Which patch kernel dmf_forward (mode 0), dmf_train_fwd_bwd (1) or dmf_backward_dlogits (2) launches for this shape:
 * 2 = the wave-per-channel-segment kernel (csrc/dmf_patch_v2.hip), 0 = no instance.  (1 was round 1's generic kernel,
 * retired in round 3.)  Reporting only (bench.py names the kernel it times). */
int32_t dmf_patch_variant(const dmf_shape* shape, int32_t mode);

/* Flat parameter vector theta (fp32).  Tensor order and offsets (in floats):
 *   0 spec_a.weight [F, C/G]   1 spec_a.bias [F]   2 spat_a.weight [F, 9]   3 spat_a.bias [F]
 *   4 lift_b.weight [F, C2*S*S] 5 lift_b.bias [F]  6 spat_b.weight [F, 9]   7 spat_b.bias [F]
 *   8 fc1.weight [H, 2F]       9 fc1.bias [H]     10 fc2.weight [K, H]     11 fc2.bias [K]
 *  12 attn_wq [E, F]  13 attn_wk [E, F]  14 attn_wv [E, F]  15 attn_wo [F, E]   (attention only)
 * offsets[i] = start of tensor i, offsets[16] = total count. */
int32_t dmf_param_layout(const dmf_shape* shape, int64_t offsets[17]);

/* Bytes of scratch the train/backward entry points need for a batch of B patches. */
int64_t dmf_workspace_bytes(const dmf_shape* shape, int32_t B);

/* `pool_w` [P*P] is the network's fixed pooling profile (buffer `pool_w` of the Net).
 *
 * Replaces `output = self.cur_model(data1, data2)` in eval (mainsolver.py:109,169,180) and
 * `pred = output.data.max(1)` (mainsolver.py:139,170).  logits [B, K]; pred [B] int32 may be NULL. */
int32_t dmf_forward(const dmf_shape* shape, const dmf_input* in, const float* theta, const float* pool_w,
                    float* logits, int32_t* pred, void* stream);
/* The same launch with the per-patch cross-entropy against labels [B] int32 written to loss [B] — the validation pass,
 * `loss = self.loss(output, target.long())` under no_grad (mainsolver.py:62-76), by the training kernel's own formula.
 * Fails (use dmf_forward_attn + the caller's loss) for the attention network. */
int32_t dmf_forward_ce(const dmf_shape* shape, const dmf_input* in, const float* theta, const float* pool_w,
                       const int32_t* labels, float* logits, float* loss, int32_t* pred, void* stream);

/* Forward of the network WITH the cross-modal attention block (shape->attention == 1; BASELINE configs[2]):
 * the conv stages emit bf16 token maps, then a matrix-core kernel (bf16 MFMA operands, fp32 accumulate) does the
 * projections, Q K^T, softmax, P V, the output projection, the pooling correction and the head.
 * `workspace` needs dmf_attn_workspace_bytes(shape, B) bytes.  Training: dmf_train_attn_fwd_bwd below. */
int64_t dmf_attn_workspace_bytes(const dmf_shape* shape, int32_t B);
int32_t dmf_forward_attn(const dmf_shape* shape, const dmf_input* in, const float* theta, const float* pool_w,
                         void* workspace, float* logits, int32_t* pred, void* stream);

/* Replaces forward + `self.loss(output, target.long())` + `loss.backward()` (mainsolver.py:52-54) in ONE
 * launch.  labels [B] int32.  loss_scale multiplies dlogits (1/B for CrossEntropyLoss' mean reduction,
 * utils/utils.py:29).  Writes logits [B, K] and per-patch CE loss [B] (unscaled); leaves per-block weight-
 * gradient slabs and head vectors in `workspace` for dmf_grad_reduce*. */
int32_t dmf_train_fwd_bwd(const dmf_shape* shape, const dmf_input* in, const float* theta, const float* pool_w,
                          const int32_t* labels, float loss_scale,
                          float* logits, float* loss, void* workspace, int32_t* adam_step_dev, void* stream);
/* adam_step_dev (may be NULL): device int incremented by one per call (the optimiser step count a later
 * dmf_grad_reduce_adam / dmf_adam_step on the same stream reads instead of its host `step` argument). */

/* The same step for the attention network (shape->attention == 1; BASELINE configs[2]): forward, loss, backward of
 * head, attention block and conv stages in three launches (token kernel, attention fwd+bwd kernel, conv backward
 * from dense gradient maps).  Exactly one of labels (fused cross-entropy, as dmf_train_fwd_bwd) and dlogits
 * (caller-supplied dL/dlogits, as dmf_backward_dlogits; logits are still written) is non-NULL; loss may be NULL.
 * `workspace` as for dmf_train_fwd_bwd (dmf_workspace_bytes counts the attention slabs when shape->attention),
 * `attn_workspace` = dmf_attn_train_workspace_bytes(shape, B) bytes.  Gradients leave through dmf_grad_reduce*. */
int64_t dmf_attn_train_workspace_bytes(const dmf_shape* shape, int32_t B);
int32_t dmf_train_attn_fwd_bwd(const dmf_shape* shape, const dmf_input* in, const float* theta, const float* pool_w,
                               const int32_t* labels, const float* dlogits, float loss_scale,
                               float* logits, float* loss, void* workspace, void* attn_workspace,
                               int32_t* adam_step_dev, void* stream);

/* The train step for a loss that couples the whole batch (qua_loss: tostagesolver.py:275-277), in two launches around the
 * caller's loss kernel and WITHOUT a second pass over the patches:
 *   dmf_forward_unit   forward (logits [B, K]) + the conv backward for a UNIT gradient on every pooled feature; leaves the
 *                      head vectors and one row of unit gradients per patch in `workspace`
 *   (loss kernel)      dL/dlogits [B, K] from the logits of the whole batch (dmf_qua_loss)
 *   dmf_backward_unit  dh, dz per patch from dL/dlogits, slab rows = sum of dz x unit rows; then dmf_grad_reduce* as usual
 * Exact, not an approximation: the net is piecewise linear and a feature channel reaches the head only through its two
 * pooled scalars.  Returns non-zero for shapes without such a kernel (dmf_unit_supported; use dmf_forward +
 * dmf_backward_dlogits there).  adam_step_dev as in dmf_train_fwd_bwd. */
int32_t dmf_unit_supported(const dmf_shape* shape);
int32_t dmf_forward_unit(const dmf_shape* shape, const dmf_input* in, const float* theta, const float* pool_w,
                         float* logits, void* workspace, int32_t* adam_step_dev, void* stream);
int32_t dmf_backward_unit(const dmf_shape* shape, int32_t B, const float* theta, const float* dlogits, void* workspace,
                          void* stream);

/* 0 if the fp16-scene kernels (dmf_input.half) exist for this shape. */
int32_t dmf_half_supported(const dmf_shape* shape);

/* ---- dynamic loss scaling: the role of `torch.cuda.amp.GradScaler` (tostagesolver.py:83-84 creates two, :98 / :119
 * `scaler.scale(loss).backward(); scaler.step(opt); scaler.update()`), device-resident so that a captured graph can
 * carry it.  state = DMF_SCALER_FLOATS floats on the device: [0] scale, [1] growth tracker, [2] found_inf of the step in
 * flight, [3] number of skipped steps so far, [4..] internal.
 *   dmf_scaler_init            state <- (init_scale, 0, 0, 0)                       GradScaler(init_scale=65536.)
 *   dmf_train_fwd_bwd_scaled   dmf_train_fwd_bwd with dL/dlogits multiplied by loss_scale * state[0]   scaler.scale(loss)
 *   dmf_qua_loss_scaled        the same for the stage-2 loss kernel (dlogits multiplied by grad_scale * state[0])
 *   dmf_grad_reduce_scaled     dmf_grad_reduce with the unscale (grad = sum / state[0]) and the non-finite check folded in,
 *                              plus the bookkeeping of dmf_grad_reduce_adam (loss_hist[cursor] = mean loss, cursor += 1);
 *                              follow with dmf_unscale_adam(unscaled = 1).  Single GPU: data parallel must check AFTER
 *                              the all-reduce (dmf_grad_reduce, all-reduce, dmf_unscale_adam(unscaled = 0))
 *   dmf_unscale_adam           grad *= grad_scale / state[0] (skipped when unscaled != 0);
 *                              any non-finite element -> the WHOLE step is skipped (no Adam state
 *                              change, adam_step_dev taken back), scale *= backoff_factor, tracker = 0; else Adam as
 *                              dmf_adam_step and tracker += 1, at growth_interval: scale *= growth_factor, tracker = 0
 *                              (scaler.unscale_ + scaler.step + scaler.update).  grad is the flat gradient of
 *                              dmf_grad_reduce (after the all-reduce when data parallel); adam_step_dev is required. */
#define DMF_SCALER_FLOATS 8
int32_t dmf_scaler_init(float* state, float init_scale, void* stream);
int32_t dmf_train_fwd_bwd_scaled(const dmf_shape* shape, const dmf_input* in, const float* theta, const float* pool_w,
                                 const int32_t* labels, float loss_scale, const float* scaler_state,
                                 float* logits, float* loss, void* workspace, int32_t* adam_step_dev, void* stream);
int32_t dmf_unscale_adam(float* theta, float* grad, float* m, float* v, int64_t n,
                         float lr, float beta1, float beta2, float eps, float grad_scale,
                         float* scaler_state, float growth_factor, float backoff_factor, int32_t growth_interval,
                         int32_t unscaled, int32_t* adam_step_dev, int32_t* cursor_dev, void* stream);
int32_t dmf_grad_reduce_scaled(const dmf_shape* shape, int32_t B, const void* workspace, float* grad, float* scaler_state,
                               int32_t* cursor_dev, const float* loss, float* loss_hist, void* stream);

/* Backward for a caller-supplied dL/dlogits [B, K] (the autograd path: torch computes the loss). */
int32_t dmf_backward_dlogits(const dmf_shape* shape, const dmf_input* in, const float* theta, const float* pool_w,
                             const float* dlogits, void* workspace, void* stream);

/* Workspace -> flat gradient [n_params] (deterministic fixed-order sums). */
int32_t dmf_grad_reduce(const dmf_shape* shape, int32_t B, const void* workspace, float* grad, void* stream);

/* Replaces `self.optimizer.step()` for torch.optim.Adam(lr) defaults (utils/utils.py:10-12;
 * betas 0.9/0.999, eps 1e-8, no weight decay).  step = 1-based step count.  grad_scale multiplies grad
 * first (1/world_size after an all-reduce(sum)). */
int32_t dmf_adam_step(float* theta, const float* grad, float* m, float* v, int64_t n,
                      float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale,
                      const int32_t* adam_step_dev, int32_t* cursor_dev, void* stream);
/* adam_step_dev (may be NULL): when given, the step count is read from the device instead of `step`.
 * cursor_dev (may be NULL): device int advanced by one (the epoch-plan cursor of dmf_input). */

/* The reference's other two optimisers (utils/utils.py:13-16), on the flat gradient of dmf_grad_reduce:
 * `torch.optim.SGD(params, lr, momentum)` — momentum_buf [n] may be NULL when momentum == 0; step (or *step_dev) == 1 marks the
 * first step, where torch initialises the buffer with the gradient — and `torch.optim.RMSprop(params, lr, alpha)` (eps 1e-8
 * is torch's default; square_avg [n] starts at zero).  grad_scale, cursor_dev as in dmf_adam_step. */
int32_t dmf_sgd_step(float* theta, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                     int32_t step, float grad_scale, const int32_t* step_dev, int32_t* cursor_dev, void* stream);
int32_t dmf_rmsprop_step(float* theta, const float* grad, float* square_avg, int64_t n, float lr, float alpha, float eps,
                         float grad_scale, int32_t* cursor_dev, void* stream);

/* dmf_grad_reduce + dmf_adam_step in one launch (single-GPU step). grad may be NULL. */
int32_t dmf_grad_reduce_adam(const dmf_shape* shape, int32_t B, const void* workspace,
                             float* theta, float* m, float* v, float* grad,
                             float lr, float beta1, float beta2, float eps, int32_t step,
                             const int32_t* adam_step_dev, int32_t* cursor_dev,
                             const float* loss, float* loss_hist, void* stream);
/* loss / loss_hist (may be NULL): when both are given, loss_hist[cursor] = mean(loss[0..B)) (fixed-order sum),
 * i.e. the value the reference prints per step (`loss.item()`, mainsolver.py:58) without a host sync. */

/* The reference's whole inner loop `for batch in loader: zero_grad; forward; CE; backward; Adam.step` (mainsolver.py:49-55)
 * over n_steps consecutive batches of a resident plan, as ONE call: step k = dmf_train_fwd_bwd on the B patches at
 * in->xy + 2 B k with labels + B k, then dmf_grad_reduce_adam (device step count, cursor, loss history) — 2 n_steps launches
 * enqueued by one C loop, nothing else between them.  in->mode must be 1 and in->cursor NULL; adam_step_dev and cursor_dev are
 * required (device ints: steps taken so far, next row of loss_hist).  Late-fusion network, ADAM, one GPU. */
int32_t dmf_train_plan_steps(const dmf_shape* shape, const dmf_input* in, float* theta, const float* pool_w,
                             const int32_t* labels, float loss_scale, float* logits, float* loss, void* workspace,
                             float* m, float* v, float lr, float beta1, float beta2, float eps,
                             int32_t* adam_step_dev, int32_t* cursor_dev, float* loss_hist, int32_t n_steps, void* stream);

/* ---- stage 2 of the two-stage path (solver/tostagesolver.py:259-346) ---------------------------------------
 * The stage-2 net takes ONE input, the four streams (ms, pan, ms_gan, pan_gan) stacked on the batch axis
 * (`torch.concat([data1..data4])`, tostagesolver.py:272), and is trained with `qua_loss`. */
typedef struct dmf_qua_params { float alpha, beta, gamma, epsilon, tao; } dmf_qua_params;   /* cfg['dqtl'][...] */
/* Replaces `self.loss(output, bs, target, cfg)` + the logits part of `loss.backward()` (tostagesolver.py:275-277,
 * train/loss_function.py:57-76).  logits [4*bs, K]; labels [bs] int32 (the reference passes float class ids);
 * cursor (may be NULL) as in dmf_input: labels[(*cursor)*bs + i], loss_hist[*cursor].  loss [1] and loss_hist may be
 * NULL; dlogits [4*bs, K] may be NULL (loss only: the validation loop, tostagesolver.py:293-295) and is multiplied
 * by grad_scale. */
int32_t dmf_qua_loss(const float* logits, int32_t bs, int32_t K, const int32_t* labels, const int32_t* cursor,
                     const dmf_qua_params* params, float grad_scale, float* loss, float* loss_hist, float* dlogits,
                     void* stream);
int32_t dmf_qua_loss_scaled(const float* logits, int32_t bs, int32_t K, const int32_t* labels, const int32_t* cursor,
                            const dmf_qua_params* params, float grad_scale, const float* scaler_state,
                            float* loss, float* loss_hist, float* dlogits, void* stream);
/* Replaces `(output[:bs] + output[bs:2*bs]).softmax(dim=-1).data.max(1)[1]` (tostagesolver.py:337,366,378). */
int32_t dmf_pair_argmax(const float* logits, int32_t bs, int32_t K, int32_t* pred, void* stream);
/* Auxiliary input of the single-stream net: per-pixel mean over bands, ((x0+x1)+x2)+... then / C.
 * layout 0: pixel-major scene [n_pix, C] (n_img = 1) -> out [n_pix]; 1: band-major patches [n_img, C, n_pix] -> out [n_img, n_pix]. */
int32_t dmf_band_mean(const float* x, int32_t layout, int64_t n_img, int64_t n_pix, int32_t C, float* out, void* stream);

/* ---- data-parallel gradient exchange over xGMI ------------------------------------
 * The reference has no multi-GPU path (BaseSolver builds one loader on one device, basesolver.py:86-105); the
 * coupling between data-parallel ranks is the parameter update of mainsolver.py:54-55 only.  One-shot exchange
 * for the sub-MB gradient: every rank writes its local gradient into an inbox slot in EVERY rank's uncached device
 * buffer over xGMI — each value together with its sequence number in ONE 8-byte store, so the value is its own
 * arrival mark — and each rank then polls its own inbox and adds the values in rank order (same bits on every
 * rank, no atomics), inside the gradient-reduce + Adam launch.  Buffers are shared
 * between the per-GPU processes with HIP IPC handles; the caller moves the 64-byte handles between ranks (any
 * host channel: torch.distributed all_gather_object, a file, a pipe).  Everything is device-side, so a whole
 * data-parallel step can be captured in a hipGraph. */
#define DMF_XGMI_MAX_RANKS 16
typedef struct dmf_xgmi_comm {
  int32_t world, rank;
  int64_t capacity;       /* floats per exchange this communicator was sized for (>= n_params)             */
  int32_t timeout_ms;     /* a rank that waits longer for a peer sets status = 1 and stops waiting          */
  int32_t seq_bias;       /* added to the device step count to form the exchange sequence number; the host  */
                          /* raises it whenever it rewinds the device step count (graph warm-up)            */
  void* data[DMF_XGMI_MAX_RANKS];    /* data[r]: rank r's inbox (own allocation or IPC mapping)             */
  void* flags[DMF_XGMI_MAX_RANKS];   /* flags[r]: rank r's status block (only the owner's is used)          */
} dmf_xgmi_comm;
int32_t dmf_xgmi_sizes(int64_t capacity, int32_t world, int64_t* data_bytes, int64_t* flag_bytes);
int32_t dmf_xgmi_alloc(int64_t bytes, void** ptr);          /* uncached device memory, zero-filled (host sync) */
int32_t dmf_xgmi_free(void* ptr);
int32_t dmf_xgmi_export(void* ptr, uint8_t handle[64]);     /* hipIpcGetMemHandle                              */
int32_t dmf_xgmi_open(const uint8_t handle[64], void** ptr);/* hipIpcOpenMemHandle (another process' buffer)   */
int32_t dmf_xgmi_close(void* ptr);
int32_t dmf_xgmi_status(const dmf_xgmi_comm* comm, int32_t* status);   /* host sync; 0 ok, 1 a wait timed out   */
/* buf[0..n) <- sum over ranks in rank order (n <= capacity).  seq: 1, 2, 3 ... per call, same on all ranks. */
int32_t dmf_xgmi_allreduce(const dmf_xgmi_comm* comm, float* buf, int64_t n, int32_t seq, void* stream);
/* dmf_grad_reduce + exchange + dmf_adam_step(grad_scale) in one launch: the data-parallel form of
 * dmf_grad_reduce_adam.  adam_step_dev is required (the sequence number is *adam_step_dev + seq_bias). */
int32_t dmf_grad_reduce_xgmi_adam(const dmf_shape* shape, int32_t B, const void* workspace,
                                  float* theta, float* m, float* v, const dmf_xgmi_comm* comm,
                                  float lr, float beta1, float beta2, float eps, float grad_scale,
                                  const int32_t* adam_step_dev, int32_t* cursor_dev,
                                  const float* loss, float* loss_hist, void* stream);

/* Replaces the per-sample `.item()` loop `test_matrix[pred][target] += 1` (mainsolver.py:140-141):
 * matrix [K, K] int64, rows = prediction. */
int32_t dmf_confusion_accum(const int32_t* pred, const int32_t* target, int32_t B, int32_t K,
                            int64_t* matrix, void* stream);

/* Replaces `label_np[x][y] = pred` (mainsolver.py:171-173,182-183): map [H, W] int32. */
int32_t dmf_labelmap_write(const int32_t* pred, const int32_t* xy, int32_t B, int32_t W, int32_t* map, void* stream);

/* Replaces image_convert/IHS.py:14-19 `pan2ms` (2x2 mean pool + 2x2 polyphase split) for out [H, W, 4]
 * from pan [>=4H, >=4W] with row pitch `pitch` floats; computed in fp64 like the reference. */
int32_t dmf_pan2ms(const double* pan, int32_t pitch, int32_t H, int32_t W, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMF_H */
