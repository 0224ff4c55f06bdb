/*
 * var_hip.h -- C ABI of libvar_hip.so, the MI355X (gfx950) implementation of the
 * VAR contrastive-pretext hot path of PeixinC/VoiceControlledRobot-VAR.
 *
 * The reference has no FFI: its seam is the class-valued config attribute
 * `config.pretextModel = VARPretextNet`
 * (Envs/pybullet/arms/tasks/fourInARow/config.py:30) and the torch calls made by
 * VAR_Pretext.trainRepresentation (VAR/pretext_VAR.py:55-70).  Each entry point
 * below names the reference code it stands in for.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. tensor.data_ptr());
 *     the library owns only its context workspace;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); entries are
 *     stream-ordered, never synchronise, never allocate (var_plan excepted) and are
 *     therefore capturable into a HIP graph;
 *   - return 0 on success, a negative VAR_ERR_* otherwise; var_last_error() gives text;
 *   - layouts are the reference's: images NCHW (u8 or f32), MFCC (B,1,100,40) f32,
 *     embeddings (B,3) row-major, parameters = the 26 state_dict() tensors of the Kuka
 *     VARPretextNet (models/pretext/arm_pretext_model.py:39-56) back to back in
 *     registration order, each in its PyTorch layout (OIHW / (out,in)): 213478 floats;
 *   - one context per (process, device); a context is not re-entrant.
 */
#ifndef VAR_HIP_H
#define VAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAR_OK 0
#define VAR_ERR_ARG (-1)        /* bad argument (null pointer, unsupported size) */
#define VAR_ERR_HIP (-2)        /* a HIP runtime call failed */
#define VAR_ERR_PLAN (-3)       /* var_plan() not called or too small for this call */
#define VAR_ERR_STATE (-4)      /* backward without a saved forward, etc. */

#define VAR_N_PARAMS 213478
#define VAR_EMB_DIM 3
#define VAR_MFCC_FRAMES 100
#define VAR_MFCC_COEFFS 40

typedef struct var_ctx var_ctx;

/* Library / context -------------------------------------------------------- */

/* Create a context on `device_id` (replaces `model.to(device)`, pretext.py:306). */
int var_init(int device_id, var_ctx** out);
int var_destroy(var_ctx* ctx);
const char* var_last_error(var_ctx* ctx);   /* ctx may be NULL: last init error */
int var_param_count(void);                  /* == VAR_N_PARAMS */

/* Size the context workspace (activations, gradient scratch, split-K slabs) for up
 * to `max_batch` triplets of img_hw x img_hw images (84 or 96).  The only entry that
 * allocates; call it outside the step loop / graph capture. */
int var_plan(var_ctx* ctx, int max_batch, int img_hw);

/* Generation counters.  var_plan_generation: bumped by every re-plan (var_plan / var_ithor_plan growing the
 * workspace); a superseded workspace stays allocated until var_destroy, so HIP graphs captured against it keep
 * replaying on valid memory.  var_saved_generation: id (> 0) of the forward whose activations the workspace
 * currently holds for var_arm_encoder_bwd, 0 = none -- a host that interleaves forwards of several models
 * (autograd) compares it with the id it noted after its own forward before calling the backward. */
int var_plan_generation(var_ctx* ctx);
int var_saved_generation(var_ctx* ctx);

/* Kernel-side weight images, one per MODEL.  The kernels read conv filters re-laid as [tap][cin][cout] and
 * [tap][cout][cin] ("packed image", 1.2 MB) next to the parameter arena.  A host with several models on one
 * device -- the reference's RL stage keeps a frozen copy of the encoder beside the policy
 * (Envs/vec_env/vec_pretext_normalize.py:82-94) -- creates one image per model and binds it before that model's
 * calls; var_weights_bind(ctx, NULL) selects the context's own default image.  Binding is host state read at
 * launch time (like hipSetDevice): a captured graph keeps the image that was bound during capture.
 * var_pack_weights re-derives the BOUND image from `params` and records that arena; it must follow every change
 * of the parameters made outside this library (load_state_dict); var_adam_step* keep the bound image current by
 * themselves.  Every var_arm_* entry checks that the bound image was packed from the `params` it is given and
 * returns VAR_ERR_STATE otherwise. */
typedef struct var_weights var_weights;
int var_weights_create(var_ctx* ctx, var_weights** out);
int var_weights_destroy(var_ctx* ctx, var_weights* w);
int var_weights_bind(var_ctx* ctx, var_weights* w);
int var_pack_weights(var_ctx* ctx, void* stream, const float* params);

/* Encoder ------------------------------------------------------------------
 * PretextNetBase.VAR_forward (models/pretext/pretext_base.py:10-41) for the Kuka
 * VARPretextNet: imgBranch -> imgTriplet -> F.normalize, soundCNN -> soundTriplet ->
 * F.normalize for the positive and the negative clip.
 *   image            (B,C>=3,H,H) u8 or f32, first 3 channels used (pretext_base.py:22);
 *                    u8 images are divided by 255 as dataset.py:67-68 does; may be NULL
 *   image_bstride    elements between consecutive images (C*H*H)
 *   mfcc_pos/neg     (B,1,100,40) f32, either may be NULL
 *   outputs          any may be NULL: image_feat/pos_feat/neg_feat (B,3),
 *                    image_raw (B,576) = image_feat_raw, pos_raw (B,160) = pos_sound_raw
 *   save_for_bwd     1: keep activations in the workspace for var_arm_encoder_bwd; 0: do not; 2: do not, and a batch of
 *                    <= 64 images takes the kernels that minimise the latency of a small batch (the RL stage's 8 envs;
 *                    Envs/vec_env/vec_pretext_normalize.py:82-101) -- equal to the training forward within rounding, not bitwise
 */
int var_arm_encoder_fwd(var_ctx* ctx, void* stream, const float* params,
                        const void* image, int image_is_u8, long image_bstride,
                        const float* mfcc_pos, const float* mfcc_neg, int B, int H,
                        float* image_feat, float* pos_feat, float* neg_feat,
                        float* image_raw, float* pos_raw, int save_for_bwd);

/* out[r] = sum_k a[r][k] * b[r][k], rows x dim f32 (dim <= 64): the intrinsic reward <image_feat, goal_sound_feat> of
 * Envs/vec_env/vec_pretext_normalize.py:96-101 (`calcReward`: torch.sum(a * b, dim=1)) as ONE launch on `stream`; the sum
 * runs over k in index order in fp32. */
int var_row_dot(var_ctx* ctx, void* stream, const float* a, const float* b, int rows, int dim, float* out);
/* Arms the NEXT var_arm_encoder_fwd with an image and at most 32 rows (the RL stage's envs): its image-head launch also leaves
 * reward_out[b] = <image_feat[b], goal_feat[b]> (3 floats per row, var_row_dot's sum) -- the intrinsic reward of
 * VAR/pretext_base.py's calcReward without a launch of its own (each launch is ~5 us on that path).  goal_feat must be complete
 * on the forward's stream (the cached goal embedding of the later steps of an episode).  Disarmed by that forward, or by
 * (NULL, NULL); forwards of more rows ignore it. */
int var_set_reward_dot(var_ctx* ctx, const float* goal_feat, float* reward_out);

/* autograd backward of the encoder (loss.backward(), VAR/pretext_VAR.py:68) from the
 * gradients of the three embeddings; writes d(loss)/d(param) for all 213478
 * parameters into `grads` (overwrites; arena layout).  Any g_* may be NULL (= zeros). */
int var_arm_encoder_bwd(var_ctx* ctx, void* stream, const float* params,
                        const float* g_image_feat, const float* g_pos_feat, const float* g_neg_feat,
                        float* grads);

/* torch.nn.TripletMarginLoss(margin, p=2, eps=1e-6, reduction='mean')
 * (VAR/pretext_VAR.py:38,64) forward + backward in one launch.
 * loss_out[0] = sum_i hinge_i * inv_count;  g* = d(loss_out)/d(a|p|n).
 * inv_count = 1/B for the reference's mean; 1/B_global under data parallelism. */
int var_triplet_fwd_bwd(var_ctx* ctx, void* stream, const float* a, const float* p, const float* n,
                        int B, float margin, float inv_count,
                        float* loss_out, float* ga, float* gp, float* gn);

/* One fused training-step body: zero_grad -> model(image,pos,neg) -> triplet loss ->
 * backward (VAR/pretext_VAR.py:56-68), everything up to but excluding optimizer.step().
 * grads (arena) and loss_out[0] are overwritten.  feats_out: NULL, or 9*B floats = three (B,3) blocks
 * [image_feat | pos_feat | neg_feat]. */
int var_arm_loss_grad(var_ctx* ctx, void* stream, const float* params,
                      const void* image, int image_is_u8, long image_bstride,
                      const float* mfcc_pos, const float* mfcc_neg, int B, int H,
                      float margin, float inv_count,
                      float* grads, float* loss_out, float* feats_out);

/* The same with the data-loader work of dataset.py:64-89 / Envs/audioLoader.py:147-157 folded in:
 * the batch is gathered by index from a dataset resident in HBM and the MFCC front-end runs
 * inside the step (by default with the rest of the sound branch on the library's side stream; see var_set_streams).
 *   image        dataset images (N,C>=3,H,H) u8|f32; sample b uses row image_index[b] (NULL: row b)
 *   pcm          dataset clips, rows of pcm_stride int16 samples; clip_index (2B) = [pos | neg] rows
 *                (NULL: rows 0..2B-1); lens (2B) valid samples per clip, 0 = the "empty" class whose
 *                MFCC is all zeros (dataset.py:37-38) */
int var_arm_loss_grad_pcm(var_ctx* ctx, void* stream, const float* params,
                          const void* image, int image_is_u8, long image_bstride, const int* image_index,
                          const int16_t* pcm, int pcm_stride, const int* clip_index, const int* lens,
                          int B, int H, float margin, float inv_count,
                          float* grads, float* loss_out, float* feats_out);

/* var_arm_loss_grad with the image batch gathered by index (sample b = row image_index[b] of the HBM-resident
 * dataset, as in var_arm_loss_grad_pcm) and the MFCC features given.  The data-parallel replayed step uses it: the
 * front-end of step k+1 (var_mfcc) is then enqueued between the gradient all-reduce of step k and its optimiser
 * step, so that the collective's latency hides behind work that does not depend on the weights. */
int var_arm_loss_grad_gather(var_ctx* ctx, void* stream, const float* params,
                             const void* image, int image_is_u8, long image_bstride, const int* image_index,
                             const float* mfcc_pos, const float* mfcc_neg, int B, int H,
                             float margin, float inv_count,
                             float* grads, float* loss_out, float* feats_out);

/* torch.optim.Adam(lr, betas, eps, weight_decay) .step() (VAR/pretext_VAR.py:33-35,69)
 * on flat arenas; `step` is the 1-based step count.  When n == VAR_N_PARAMS and
 * params is the model arena the packed weight images are refreshed as well. */
int var_adam_step(var_ctx* ctx, void* stream, float* params, const float* grads, float* exp_avg,
                  float* exp_avg_sq, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step);

/* The same step with the step count (incremented here) and the learning rate read from DEVICE
 * memory, so that the launch can be captured once into a HIP graph and replayed every step
 * (MultiStepLR then updates *lr_dev between replays). */
int var_adam_step_dev(var_ctx* ctx, void* stream, float* params, const float* grads, float* exp_avg,
                      float* exp_avg_sq, long n, const float* lr_dev, float beta1, float beta2, float eps,
                      float weight_decay, int* step_dev);

/* var_adam_step_dev plus the data-loader cursor of a graph-replayed epoch: index_table (n_rows x row_ints int32,
 * e.g. [image_index | clip_index | lens] per step, built once per epoch on the device -- the reference's
 * DataLoader(shuffle=True), VAR/pretext_VAR.py:26-31,55) is walked on the DEVICE: at the end of the step row
 * (*cursor_dev + 1) mod n_rows is copied into index_row (the buffer the captured var_arm_loss_grad_pcm reads) and
 * *cursor_dev is advanced, so a replay needs no host-side copy.  One kernel launch does the Adam update, the
 * re-pack of the weight images, the step count and the row fetch.  ahead != 0: index_row holds TWO rows,
 * [row cursor+1 | row cursor+2] -- the data-parallel pipeline computes the MFCC features of step k+1 before the
 * optimiser step of step k, so its front-end reads the second copy while the gradient pass reads the first. */
int var_adam_step_graph(var_ctx* ctx, void* stream, float* params, const float* grads, float* exp_avg,
                        float* exp_avg_sq, long n, const float* lr_dev, float beta1, float beta2, float eps,
                        float weight_decay, int* step_dev, const int* index_table, int row_ints, int n_rows,
                        int* cursor_dev, int* index_row, int ahead);

/* Audio front-end: Envs/audioLoader.py:147-157 (torchaudio MFCC branch) + :241-252
 * (processSoundFeat).  pcm: rows of `pcm_stride` int16 samples; output clip i reads row
 * clip_index[i] (NULL: row i) and lens[i] valid samples (<= pcm_stride; 0 = "empty" class =>
 * zeros); out: (nclips, 1, out_frames, 40) f32, frames beyond 1 + len/160 are zero
 * (MFCC-domain padding), frames beyond out_frames are dropped. */
int var_mfcc(var_ctx* ctx, void* stream, const int16_t* pcm, const int* lens, const int* clip_index,
             int nclips, int pcm_stride, int out_frames, float* out);

/* The same front-end with the STFT parameters of the dataset a clip comes from (Envs/audioLoader.py:23-31, param_dict:
 * nFFT / int(windowLenTime * fs) / int(windowStepTime * fs)): 512 / 400 / 160 for GoogleCommand, FSC, ESC50, Spatial,
 * Synthetic -- var_mfcc's kernel -- and 1024 / 800 / 640 for NSynth and UrbanSound, or any power-of-two n_fft in
 * 64..2048 with win_length <= n_fft; T = 1 + len / hop_length frames.  The first call for a new configuration builds
 * its tables (it allocates: make it once outside graph capture). */
int var_mfcc_ex(var_ctx* ctx, void* stream, const int16_t* pcm, const int* lens, const int* clip_index,
                int nclips, int pcm_stride, int out_frames, int n_fft, int win_length, int hop_length, float* out);

/* iTHOR model ------------------------------------------------------------------------------
 * The second VARPretextNet of the reference (models/pretext/ai2thor_pretext_model.py:5-58, config 4 of
 * BASELINE.json): stride-1 3x3 convolutions with 2x2 max pools on the image, three wide stride-2 convolutions
 * and a bidirectional GRU(448 -> 512) on the (1,600,40) sound features, Linear heads, F.normalize.  Parameters
 * are its 36 state_dict() tensors back to back in registration order (imgBranch.{0,2,5,8,11,14}, rnn.*_l0,
 * rnn.*_l0_reverse, cnn.{0,2,4}, imgTriplet.{0,2}, soundTriplet.{0,2,4}), each in its PyTorch layout:
 * var_ithor_param_count() = 3849126 floats.  The entries mirror the Kuka ones above (same argument meaning);
 * image_raw is (B,1152), pos_raw (B,1024); img_hw is any side that ends in a 3x3 map (96, 84).
 * var_adam_step / var_triplet_fwd_bwd are shared with the Kuka model. */
int var_ithor_param_count(void);
int var_ithor_plan(var_ctx* ctx, int max_batch, int img_hw);
/* Operand precision of every product of the iTHOR model: 0 = fp32 (default; the parity path), 1 = bf16 operands
 * (round to nearest even) with fp32 accumulation on v_mfma_f32_32x32x16_bf16 -- BASELINE config 4's stated precision;
 * parameters, gradients, the optimiser state and every activation the model returns stay fp32.  The sound CNN's
 * intermediate maps then live as bf16 images only; 2 = as 1, and the fp32 copies of those maps (conv 1 / conv 2 outputs and
 * the gradient wrt conv 2's output: 1.3 GB of stores per step at batch 256 that nothing but var_debug_buffer reads) are
 * written too -- what the layer-wise parity tests use.  -1 = query.  Returns the previous setting (or a negative error
 * code); call after var_ithor_plan. */
int var_ithor_set_bf16(var_ctx* ctx, int on);
/* bf16 mode only: run the 73 time steps of each GRU pass as ONE persistent launch (1, the default) in which the workgroups
 * of a 64-clip slice hand the recurrent state to each other through memory, or as one launch per time step (0).  Both
 * give bit-identical results.  The persistent form needs its whole grid (32 workgroups per 64 clips) resident at once:
 * it is skipped by itself when the grid exceeds the device's CU count, and every wait in it is bounded -- if a wait
 * expires (e.g. another process holds part of the GPU) the launch ends and the step's embeddings / gradient are
 * overwritten with NaN rather than left partially updated.  The time-out word of the CURRENT step also guards the
 * optimiser: var_adam_step / var_adam_step_dev over this model's arena (n == var_ithor_param_count()) then leave
 * parameters, moments and step count untouched, so a transient time-out costs one step, not the run.  The next
 * var_ithor_* forward clears the current word by itself and files the event in two sticky words (count, last code).
 * var_ithor_gru_status copies the status (blocking): the current step's code (1 + t forward, 101 + t backward) if it
 * timed out, else 0x40000000 | code of the last earlier time-out, else 0.  The residency test uses the occupancy the
 * runtime grants the two sequence kernels (hipOccupancyMaxActiveBlocksPerMultiprocessor), not just the CU count.
 * -1 = query; returns the previous setting; setting a form (0 / 1) also clears every status word (after a
 * hipDeviceSynchronize-like wait on the null stream). */
int var_ithor_set_gru_sequence(var_ctx* ctx, int on);
int var_ithor_gru_status(var_ctx* ctx, unsigned* word);
/* Data parallelism: the time-out word above is per rank, but the poisoned gradient of the rank that timed out is summed into
 * every rank's buffer by the all-reduce.  var_ithor_guard_loss names a device float that var_adam_step / var_adam_step_dev over
 * this model's arena read at launch time as a second guard: not finite = leave parameters, moments and step count alone.
 * Point it at the loss slot that travels with the gradient through the all-reduce (the timed-out rank's loss is NaN, so the
 * sum is NaN on every rank) and all replicas skip the same step.  NULL removes the guard.  The pointer is read when an Adam
 * launch is enqueued (a captured launch keeps the one it was captured with). */
int var_ithor_guard_loss(var_ctx* ctx, const float* loss_dev);
int var_ithor_encoder_fwd(var_ctx* ctx, void* stream, const float* params,
                          const void* image, int image_is_u8, long image_bstride,
                          const float* snd_pos, const float* snd_neg, int B, int H,
                          float* image_feat, float* pos_feat, float* neg_feat,
                          float* image_raw, float* pos_raw, int save_for_bwd);
int var_ithor_saved_generation(var_ctx* ctx);     /* as var_saved_generation, for the iTHOR workspace */
int var_ithor_encoder_bwd(var_ctx* ctx, void* stream, const float* params,
                          const float* g_image_feat, const float* g_pos_feat, const float* g_neg_feat,
                          float* grads);
int var_ithor_loss_grad(var_ctx* ctx, void* stream, const float* params,
                        const void* image, int image_is_u8, long image_bstride,
                        const float* snd_pos, const float* snd_neg, int B, int H,
                        float margin, float inv_count,
                        float* grads, float* loss_out, float* feats_out);

/* In-batch-negatives contrastive head (BASELINE.json configs[2]; an EXTENSION without a reference counterpart --
 * the reference trains with the explicit-negative triplet loss above).  anchor (B,3) = the local image embeddings,
 * cand (M,3) = the candidate sound embeddings of the GLOBAL batch (every rank's [positives ; negatives],
 * var_allgather_emb), target[i] = column of sample i's positive.
 *   L = inv_count * sum_i [ logsumexp_j(-d_ij / tau) + d_{i,target[i]} / tau ],  d_ij = ||a_i - c_j + 1e-6||_2
 * loss_out[0] = this rank's share of L; g_anchor (B,3) = dL/da; g_cand (M,3) = the gradient wrt every candidate from
 * this rank's rows (sum over the ranks -- var_allreduce_grads on it -- and keep your own rows).  scratch: 2*B floats.
 * One wavefront per row / per candidate, wave-shuffle softmax, no atomics. */
int var_inbatch_loss_fwd_bwd(var_ctx* ctx, void* stream, const float* anchor, const float* cand, const int* target,
                             int B, int M, float tau, float inv_count, float* scratch, float* loss_out,
                             float* g_anchor, float* g_cand);

/* Collectives ----------------------------------------------------------------------------------
 * For hosts without torch.distributed: one RCCL communicator per context.  Rank 0 obtains a 128-byte unique id
 * (var_comm_unique_id), the host ships it to the other ranks over its own channel, every rank calls var_comm_init.
 * var_allreduce_grads: in-place sum over the ranks of the flat gradient arena (append the loss as one more float and
 * it travels in the same message) -- the ONE exchange of the data-parallel step, between var_*_loss_grad
 * (inv_count = 1/B_global) and var_adam_step.  var_allgather_emb: global[r*n_local ...] = rank r's `local`
 * (embeddings for an in-batch-negatives loss; BASELINE config 3's extension).  Stream-ordered like every other
 * entry.  librccl.so is opened on first use (dlopen), not at load time. */
int var_comm_unique_id(var_ctx* ctx, void* id128);
int var_comm_init(var_ctx* ctx, int rank, int nranks, const void* id128);
int var_comm_destroy(var_ctx* ctx);
int var_allreduce_grads(var_ctx* ctx, void* stream, float* flat_grad, long n);
int var_allgather_emb(var_ctx* ctx, void* stream, const float* local, float* global, long n_local);

/* RL actor-critic forward ---------------------------------------------------------------
 * Policy.act up to the sampling (models/ppo/model.py:57-69): armNet_VAR.forward (models/RL/arm_RL_model.py:99-134, the
 * 96x96 image branch, recurrent: one GRU(128 -> 512) step from rnn_hxs * masks, models/ppo/model.py:118-121) and the
 * mean layer of DiagGaussian (models/ppo/distributions.py:65-84).  Kuka configuration: representationDim 3,
 * robotStateDim 2, RLRecurrentInputSize 128, RLRecurrentSize 512, RLActionHiddenSize 128, actionDim 2
 * (fourInARow/config.py:67-106, kuka/env_config.py:37).  params = Policy.state_dict() back to back in registration
 * order (base.gru.*, base.imgCNN.{0,2,5,7,10,12,15,17}, base.motorMlp, cnnMlp, imgMotorMlp, imgMotorMlp2, soundMlp,
 * fusionMlp, mlp_all, actor, critic, critic_linear, dist.fc_mean, dist.logstd._bias): var_armnet_param_count() floats.
 *   image (B,3,96,96) u8 (divided by 255) or f32; image_feat (B,3), robot_pose (B,2), goal_sound_feat (B,3),
 *   rnn_hxs (B,512), masks (B,1)  ->  value (B,1), actor_features (B,128), action_mean (B,2, may be NULL),
 *   rnn_hxs_out (B,512).  Sampling / log-probabilities (a handful of flops) stay with the caller.
 * Kernel paths (same results within 2e-5): B <= 64 images run the convolutions on LDS-band kernels with the pools fused, the
 * filters re-packed inside the first launch of every call (parameters updated in place between two calls are picked up);
 * B <= 8 rows (the RL stage's envs) run the 22 Linear layers + GRU step as one persistent launch whose workgroups hand their
 * vectors over as (value, epoch) pairs -- it needs its 128 workgroups co-resident; if they are not, every wait times out
 * (bounded: a workgroup gives up all its later waits at once after its first expired one) and the outputs are NaN.  The
 * reference's Policy.act (models/ppo/model.py:57-69) has no failure mode, so this one is reported: var_armnet_status copies
 * (blocking) 1 if the most recent chain launch timed out, 0x40000001 if an earlier one did since the last
 * var_armnet_clear_status, else 0; a timed-out launch leaves nothing behind -- the next one is clean.  rnn_hxs_out must not
 * overlap rnn_hxs (VAR_ERR_ARG: the chain reads the old state from all workgroups of its GRU stage while one writes the new
 * one).  Captured into a HIP graph the call replays as 9 kernel nodes. */
int var_armnet_param_count(void);
int var_armnet_plan(var_ctx* ctx, int max_batch);
int var_armnet_forward(var_ctx* ctx, void* stream, const float* params, const void* image, int image_is_u8,
                       long image_bstride, const float* image_feat, const float* robot_pose,
                       const float* goal_sound_feat, const float* rnn_hxs, const float* masks, int B,
                       float* value, float* actor_features, float* action_mean, float* rnn_hxs_out);
int var_armnet_status(var_ctx* ctx, unsigned* word);
int var_armnet_clear_status(var_ctx* ctx);

/* The iTHOR/FSC audio front-end: python_speech_features.mfcc as called at Envs/audioLoader.py:158-161 (pre-emphasis
 * .97, 400/160 frames with a zero-padded tail, np.hamming, |rfft_512|^2/512, 40 triangles, log, orthonormal DCT-II,
 * lifter 22, coefficient 0 = log frame energy; int16 samples NOT normalised) + processSoundFeat (:241-252).
 * Arguments as var_mfcc; T = 1 + ceil((len-400)/160) frames per clip (1 if len <= 400), out (nclips,1,out_frames,40)
 * f32 (the library computes float64; this kernel float32).  The tables are built on the first call (the only
 * call of this entry that allocates: make it once outside graph capture). */
int var_mfcc_psf(var_ctx* ctx, void* stream, const int16_t* pcm, const int* lens, const int* clip_index,
                 int nclips, int pcm_stride, int out_frames, float* out);

/* Measurement and testing hooks -------------------------------------------------
 * var_profile_select: record HIP events, on the launch stream, around every launch of one
 * kernel family (tag in [0, var_profile_tag_count()), -1 = off); var_profile_read returns the
 * summed durations and the launch count since the select (it synchronises on the events).
 * var_set_streams: which parts of a step leave the caller's stream (bit 0: sound CNN forward, bit 1: sound
 * CNN backward, bit 4: with bit 0, MFCC stays on the caller's stream, bit 5: the one-launch image forward of image-only
 * calls also beside a sound branch -- slower there, kept for timing, bit 6: the training step's forward and backward
 * hand over between their two streams with graph edges again instead of device-side flags, see var_join_status); -1 restores the default (3).
 * 0 puts every kernel on the caller's stream (per-kernel timing).  Returns the old mask.
 * var_join_status: in a training step recorded under stream capture (var_arm_loss_grad* with all three branches, two
 * streams; launched eagerly the step keeps its stream edges) the backward's first kernels
 * do not wait for the other stream's forward through the streams (a barrier packet in a replayed graph costs ~10 us there):
 * the last workgroup of each branch's last forward kernel counts a flag up and polls the other branch's before it ends; a
 * poll gives up after 5 ms -- the other branch never ran: a fault, the step's numbers are undefined -- and counts itself.
 * *timeouts = that count since var_init (synchronous copy; the reference's loss.backward(), VAR/pretext_VAR.py:68, has no
 * such failure mode, so it is reported).
 * var_debug_buffer: address/length of a named workspace buffer ("act1".."act5", "gact1"..,
 * "sact1".."sact4", "gsact1".., "emb", "gemb", "wpack"; "ithor_s1".."ithor_s3", "ithor_gs1".."ithor_gs3") for
 * layer-wise parity tests.
 * var_debug_ithor_dense: one dense product of the iTHOR model's bf16 mode through the kernel its schedule would pick,
 * C[m + n*M] (+= when `add`) = sum_k A(m,k) B(k,n), A(m,k) = a[m*K + k] if a_kfast else a[k*M + m], B likewise with
 * b[n*K + k] | b[k*N + n]; nsplit > 1 writes split-K slabs C + s*M*N instead (device pointers, fp32).  Returns 1 when
 * the staged bf16 kernel (dense_bf16.h) ran, 0 when the shapes fell back to the gather-GEMM, < 0 on error.  `add` bit 1:
 * both operands are first copied to bf16 and handed over as copies, as the model's schedule does for its big products
 * (nsplit must be 1); then 2 is returned when the resident-panel kernel (K = 448, both operands k-fast) took it. */
int var_profile_tag_count(void);
const char* var_profile_tag_name(int tag);
int var_profile_select(var_ctx* ctx, int tag);
int var_profile_read(var_ctx* ctx, float* total_ms, int* count);
int var_set_streams(var_ctx* ctx, int mask);
int var_join_status(var_ctx* ctx, unsigned* timeouts);
int var_debug_buffer(var_ctx* ctx, const char* name, void** ptr, long* nfloats);
/* tests: the next persistent GRU forward is launched one workgroup short per hand-off group, so that no group can
 * complete -- what the resident part of a grid sees when the rest is not: every wait must expire (about 0.3 s), the
 * launch must end, var_ithor_gru_status must read non-zero and the step's outputs must be NaN. */
int var_debug_ithor_gru_drop_workgroup(var_ctx* ctx);
/* tests: the next small-batch chain launch of var_armnet_forward runs one workgroup short: the vectors that workgroup owes never
 * arrive, every consumer's wait must expire (about 0.3 s), the launch must end with NaN outputs, var_armnet_status must read 1,
 * and the launch after that must be clean (status 0x40000001 until cleared). */
int var_debug_armnet_drop_workgroup(var_ctx* ctx);
int var_debug_ithor_dense(var_ctx* ctx, void* stream, int a_kfast, int b_kfast, const float* a, const float* b,
                          float* c, int M, int N, int K, int nsplit, int add);

#ifdef __cplusplus
}
#endif
#endif /* VAR_HIP_H */
