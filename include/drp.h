/*
 * drp.h -- C ABI of the MI355X-native particle-GNN rollout + sampling-MPC engine.
 *
 * The reference (WangYixuan12/dyn-res-pile-manip) has no FFI for this path: its
 * boundary is a Python call surface.  This header is the
 * C-ABI a Python/ctypes (or any other) host binds to get the same operations;
 * every entry point cites the reference function it replaces (paths relative to
 * the reference repo).  Plain pointers and sizes only; all float data is fp32
 * row-major; host buffers are caller-owned; device workspaces are owned by the
 * context and re-used while the shapes fit.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DRP_E* code otherwise, and
 *     never throws; drp_last_error() gives the message of the last failure.
 *   - one context per GPU, one HIP stream per context.  Functions taking host
 *     buffers synchronise before returning; the drp_mpc_* family only enqueues
 *     work on the context's stream (drp_sync() waits).
 *   - B = n_sample * n_batch rows; row = sample * n_batch + batch
 *     (planners.py:336-339).  K = 10 in-edges per receiver at most
 *     (model/gnn_dyn.py:231).  F = 64 features (nf_effect).
 */
#ifndef DRP_H
#define DRP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRP_K 10
#define DRP_F 64
#define DRP_N_WEIGHTS 38403   /* floats in the state_dict, SURVEY.md 8 a16 */

enum {
    DRP_OK = 0,
    DRP_EINVAL = -1,    /* bad argument / shape */
    DRP_ESTATE = -2,    /* call order: weights / camera / goal / state not set */
    DRP_EHIP = -3,      /* HIP runtime error */
    DRP_ENOMEM = -4,
    DRP_ECOMM = -5,     /* RCCL error */
    DRP_ERANGE = -6     /* weights or inputs outside the range the split-fp16 relation encoder of
                           DRP_ENGINE_FUSED / _SPLIT is scaled for (its hidden activations travel as two fp16
                           pieces times an exact power of two chosen from the weights): nothing was computed;
                           DRP_ENGINE_MFMA / _VALU have no such limit */
};

/* which kernels compute the MLPs */
enum {
    DRP_ENGINE_VALU = 0,  /* fp32 VALU reference kernels */
    DRP_ENGINE_MFMA = 1,  /* fp32 MFMA (v_mfma_f32_32x32x2_f32) kernels */
    DRP_ENGINE_SPLIT = 2, /* as MFMA, relation encoder on split-fp16 (two terms, 3 MFMAs per product) MFMA, fp32 accumulate */
    DRP_ENGINE_FUSED = 3  /* as SPLIT, encoder recomputed inside each aggregate: the edge-constant
                             buffer is never materialised */
};

typedef struct drp_ctx drp_ctx;

/* ---- life cycle ------------------------------------------------------------------ */
int drp_create(int device, drp_ctx** out);
void drp_destroy(drp_ctx* ctx);
const char* drp_last_error(const drp_ctx* ctx);     /* ctx may be NULL: last create error */
int drp_sync(drp_ctx* ctx);
int drp_set_engine(drp_ctx* ctx, int engine);
int drp_device_info(drp_ctx* ctx, char* name, size_t name_len, int* n_cu, size_t* hbm_bytes);

/* ---- model constants ----------------------------------------------------------- */
/* PropNetDiffDenModel.load_state_dict (visualize_mpc.py:36-41): the 38 403 floats of
 * the state_dict, concatenated in its own key order, torch Linear
 * layout [out,in].  adj_thresh = config train.particle.adj_thresh
 * (model/gnn_dyn.py:206). */
int drp_load_weights(drp_ctx* ctx, const float* blob, size_t n_floats, float adj_thresh);

/* PlannerGD.world2cam (planners.py:192-209): m34 = first three rows of
 * inv(inv(cam_extrinsic) diag(1,-1,-1,1)) in fp32, global_scale from the config;
 * intr = env.get_cam_params() = [fx,fy,cx,cy] (env/flex_env.py:1135-1142). */
int drp_set_camera(drp_ctx* ctx, const float m34[12], float global_scale, const float intr[4]);

/* config_reward_ptcl's constants (env/flex_rewards.py:172-177, planners.py:620-624):
 * field = goal - distanceTransform(goal < 0.5), shifted to min 0, [h,w];
 * goal_coor [m,2] = (col,row) goal pixels. */
int drp_set_goal(drp_ctx* ctx, const float* field, int h, int w, const float* goal_coor, int m);

/* ---- goal pre-processing on the device (row f3) ------------------------------------------------
 * cv2.distanceTransform(src, cv2.DIST_L2, 5) (env/flex_rewards.py:174; utils.py:553,572,603):
 * distance of every non-zero pixel of src [h,w] to the nearest zero pixel.
 * DRP_DT_CV5: OpenCV's 5x5 fixed-point chamfer (weights 1, 1.4, 2.1969), same integers as its
 * two raster passes.  DRP_DT_EXACT: exact Euclidean distance (sqrt of the integer squared
 * distance, in float64, rounded to float32). */
#define DRP_DT_CV5 0
#define DRP_DT_EXACT 1
int drp_distance_transform(drp_ctx* ctx, const uint8_t* src, int h, int w, int mode, float* dist_out);

/* Everything config_reward_ptcl and the planner derive from the goal image, in one call and
 * kept on the device (replaces drp_set_goal + host work): obs_goal [h,w] = the goal distance
 * image the caller passes to trajectory_optimization_ptcl_multi_traj (planners.py:567).
 *   field     = obs_goal - distanceTransform(obs_goal < 0.5); field -= min   (env/flex_rewards.py:172-177)
 *   goal_coor = fps_np(flip((obs_goal < 0.5).nonzero()), min(max_goal_pts, count), fps_init)
 *               (planners.py:620-624; max_goal_pts = 5 * particle_num there)
 * field_out [h,w] / goal_coor_out [m,2] / m_out are optional copies for the caller. */
int drp_set_goal_image(drp_ctx* ctx, const float* obs_goal, int h, int w, int mode, int max_goal_pts,
                       int fps_init, float* field_out, float* goal_coor_out, int* m_out);

/* ---- single operations on host buffers (unit parity with the reference) --------
 * These stage their inputs in the buffers the drp_mpc_* / drp_gd_* sessions keep their state in: calling one of
 * them ends a running session (its next call returns DRP_ESTATE; begin again). */
/* PlannerGD.gen_s_delta (planners.py:211-257). s_cur [B,N,3], action [B,4] -> [B,N,3] */
int drp_gen_s_delta(drp_ctx* ctx, const float* s_cur, const float* action, int B, int N,
                    float* s_delta_out);

/* The graph of predict_one_step (model/gnn_dyn.py:223-251) as receiver-major lists:
 * nbr_idx [B,N,10] int16 (ascending sender, -1 padded), nbr_cnt [B,N] uint8. */
int drp_build_graph(drp_ctx* ctx, const float* s_cur, const float* s_delta, int B, int N,
                    int16_t* nbr_idx_out, uint8_t* nbr_cnt_out);

/* PropNetDiffDenModel.predict_one_step (model/gnn_dyn.py:209-254).
 * a_cur [B,N], s_cur/s_delta [B,N,3], dens [B] -> s_pred [B,N,3] */
int drp_step(drp_ctx* ctx, const float* a_cur, const float* s_cur, const float* s_delta,
             const float* dens, int B, int N, float* s_pred_out);

/* PropModuleDiffDen.forward (model/gnn_dyn.py:147-198) with the relations given as
 * lists instead of dense one-hot Rr/Rs. */
int drp_forward(drp_ctx* ctx, const float* a_cur, const float* s_cur, const float* s_delta,
                const float* dens, const int16_t* nbr_idx, const uint8_t* nbr_cnt, int B, int N,
                float* s_pred_out);

/* PlannerGD.ptcl_model_rollout (planners.py:302-370): s0 [nb,N,3], attr [nb,N],
 * dens [nb], actions [B,H,4] -> states_out [B,H,N,3] (nullable).  If reward_out is
 * not NULL it receives config_reward_ptcl of every step, [B,H] (what
 * ptcl_evaluate_traj computes, planners.py:414-422); needs drp_set_goal. */
int drp_rollout(drp_ctx* ctx, const float* s0, const float* attr, const float* dens, int nb,
                int N, const float* actions, int B, int H, float* states_out, float* reward_out);

/* config_reward_ptcl (env/flex_rewards.py:156-214) downstream of the distance
 * transform.  state [Bp,N,3] -> reward [Bp]. */
int drp_reward(drp_ctx* ctx, const float* state, int Bp, int N, int normalize, float* reward_out);

/* ---- device-resident sampling MPC (MPPI) ------------------------------------------
 * One iteration = sample_action_sequences (planners.py:69-190) -> ptcl_model_rollout
 * -> final-step reward -> optimize_action (planners.py:549-561), all on the stream.
 * The sample axis may be sharded over ranks: each rank runs n_sample_local samples
 * and the softmax-weighted mean is combined from per-rank partials. */
typedef struct drp_mpc_params {
    int n_batch;          /* initial-state columns (particle re-samplings) */
    int n_particles;
    int n_sample;         /* samples on THIS rank */
    int n_look_ahead;     /* H */
    double sigma;         /* mpc.sigma * global_scale / 12 (planners.py:116) */
    double beta_filter;   /* mpc.mppi.beta_filter (planners.py:93) */
    double reward_weight; /* mpc.mppi.reward_weight (planners.py:553) */
    float act_lo[4];      /* clip box (planners.py:152-155) */
    float act_hi[4];
    uint64_t seed;        /* Philox key */
    uint64_t sample_offset; /* first global sample index of this rank (Philox counter) */
    int noise_type;       /* DRP_NOISE_*: the sampler's noise_type argument (planners.py:75,116-135,169-175) */
    int reserved;         /* 0 */
} drp_mpc_params;
#define DRP_NOISE_NORMAL 0      /* N(0, sigma)                                                  */
#define DRP_NOISE_UNIFORM 1     /* U(-sigma, sigma); the caller passes sigma = 2 global_scale / 12 */
#define DRP_NOISE_TOTAL_RAND 2  /* no residual: every push drawn uniformly from the clip box   */

int drp_mpc_begin(drp_ctx* ctx, const drp_mpc_params* p, const float* s0, const float* attr,
                  const float* dens, const double* nominal /* [H,4] */);
/* noise: NULL -> device Philox draws; else host [n_sample,H,4] draws: standard normal (DRP_NOISE_NORMAL),
 * U(-1,1) (DRP_NOISE_UNIFORM) or U[0,1) (DRP_NOISE_TOTAL_RAND). */
int drp_mpc_sample(drp_ctx* ctx, const float* noise, uint64_t iteration);
int drp_mpc_set_actions(drp_ctx* ctx, const float* actions /* [B,H,4] */);
int drp_mpc_rollout(drp_ctx* ctx, int reward_all_steps);
/* per-rank partials of the softmax mean over column `col`'s samples:
 * out[0]=m, out[1]=Z, out[2..2+4H)=A, then sum r, sum r^2, max r, argmax (as double). */
int drp_mpc_partials(drp_ctx* ctx, double* out /* [6+4H], nullable */);
/* combine `n_ranks` partial records (as returned above, concatenated) into the new
 * nominal sequence; uploads it as the next iteration's nominal. */
int drp_mpc_update(drp_ctx* ctx, const double* partials, int n_ranks, double* nominal_out);
/* partials -> (RCCL all-gather if a communicator is attached) -> update, no host hop */
int drp_mpc_update_device(drp_ctx* ctx);
/* Elite (cross-entropy-method style) update -- NOT in the reference (planners.py has no CEM; SURVEY.md section 8e
 * lists it as the other form of the planner's one exchange): the new nominal sequence is the mean of the k best
 * samples' sequences, higher final-step reward first, ties to the lower global sample index.
 * drp_mpc_elite: this rank's k best as records [reward, global sample index, act[4H]] (2 + 4H doubles each, best
 * first; reward = -inf and index = -1 pad a rank with fewer than k samples).
 * drp_mpc_update_elite: combine n_ranks x k records (host transport) into the nominal sequence.
 * drp_mpc_update_elite_device: the rank's statistics record (as drp_mpc_update_device) and its k elite records
 * travel as ONE message, [6 + 4H | k (2 + 4H)] doubles, in ONE RCCL all-gather per iteration when a communicator is
 * attached, and both combines read the gathered messages in place; no host hop.  1 <= k <= 1024, up to 9 000
 * samples per rank. */
int drp_mpc_elite(drp_ctx* ctx, int k, double* out /* [k][2+4H], nullable */);
int drp_mpc_update_elite(drp_ctx* ctx, const double* records, int n_ranks, int k, double* nominal_out /* [H][4], nullable */);
int drp_mpc_update_elite_device(drp_ctx* ctx, int k);
int drp_mpc_get(drp_ctx* ctx, float* actions /*[B,H,4]*/, float* rewards /*[B] final*/,
                float* rewards_all /*[B,H]*/, float* states /*[B,H,N,3]*/, double* nominal);
/* The pushes and final rewards of the iteration just enqueued, without a host wait: slot (0 or 1) takes them into pinned
 * memory behind the iteration's kernels -- before the next drp_mpc_sample overwrites the pushes --; drp_mpc_wait(slot)
 * blocks until they are there and copies them out (each pointer nullable).  The planner's loop enqueues iteration i + 1
 * before it waits for iteration i.  A slot must be waited for before it is used again; drp_mpc_begin and the one-shot
 * calls drop what is in flight. */
int drp_mpc_fetch_async(drp_ctx* ctx, int slot);
int drp_mpc_wait(drp_ctx* ctx, int slot, float* actions /*[B,H,4]*/, float* rewards /*[B] final*/);

/* fps_np (utils.py:451-466): farthest-point subsample of pts [n,dim] (dim 2 or 3) to k points
 * starting from init_idx; idx_out [k] are indices into pts, max_dist_out the largest distance
 * of any point to the chosen set.  Used for the goal pixels (planners.py:620-624). */
int drp_fps(drp_ctx* ctx, const float* pts, int n, int dim, int k, int init_idx, int32_t* idx_out,
            float* max_dist_out);

/* ---- training on the same kernels (row f4; train/train_gnn_dyn.py:159-214) ---------------------
 * One iteration of the reference's training loop body for one collated batch
 * (train/train_gnn_dyn.py:20-45 collate_fn: samples zero-padded to the batch's largest
 * particle count N):
 *   s_cur = states[:, 0]; for t < n_rollout: s_pred = predict_one_step(attrs[:, 0], s_cur,
 *   states_delta[:, t], particle_dens); loss += sum_b mse(s_pred[b, :n_b], states[b, t+1, :n_b]);
 *   s_cur = s_pred;  loss /= n_rollout * B;  loss.backward();  Adam(lr, betas=(beta1, .999)).step()
 * states [B, n_rollout+1, N, 3], states_delta [B, n_rollout, N, 3], attrs [B, n_rollout+1, N],
 * particle_nums [B], particle_dens [B].  loss_out receives the loss (before the update),
 * grad_out (nullable, 38 403 floats in state_dict order) the gradient of every parameter.
 * The arrays are copied before the call returns (one upload), and the call returns when the iteration is complete on
 * the device (its only host wait): the caller's buffers are free at once, drp_get_weights serves the updated weights. */
#define DRP_TRAIN_EVAL 0     /* loss only ('valid' phase, torch.set_grad_enabled(False)) */
#define DRP_TRAIN_GRAD 1     /* loss + gradients, weights untouched */
#define DRP_TRAIN_UPDATE 2   /* loss + gradients + one Adam step on the context's weights */
int drp_train_begin(drp_ctx* ctx, int n_rollout, double lr, double beta1);
int drp_train_step(drp_ctx* ctx, const float* states, const float* states_delta, const float* attrs,
                   const int32_t* particle_nums, const float* particle_dens, int B, int N, int mode,
                   double* loss_out, float* grad_out);
int drp_train_set_lr(drp_ctx* ctx, double lr);
/* model.state_dict() (train/train_gnn_dyn.py:226,244): the current weights, drp_load_weights layout */
int drp_get_weights(drp_ctx* ctx, float* blob_out, size_t n_floats);

/* ---- particle extraction from the depth image (row f2; env/flex_env.py:933-951) --------------
 * The reference runs this chain on the host between every pair of planner calls, 30 times per
 * observation (batch_size=30, env/flex_env.py:1020,1093).  All clouds are float64 [n,3] in the
 * camera frame, exactly as the reference's numpy arrays. */

/* utils.py:491-506 depth2fgpcd(depth, mask, cam_params): row-major foreground pixels
 * (mask != 0 and depth > 0) -> points ((x-cx)*d/fx, (y-cy)*d/fy, d).  mask NULL selects the
 * rule of env/flex_env.py:945, depth < 0.599/0.8 in float32.  cam = {fx, fy, cx, cy}.
 * pcd_out may be NULL (count only); DRP_EINVAL if cap < n (n_out is still set). */
int drp_depth2fgpcd(drp_ctx* ctx, const float* depth, const uint8_t* mask, int h, int w, const double cam[4],
                    double* pcd_out, int cap, int* n_out);

/* utils.py:533-544 downsample_pcd(pcd, voxel_size) = open3d voxel_down_sample: one point per
 * occupied voxel (index floor((p - (min_bound - voxel/2)) / voxel)), the mean of its points
 * summed in index order; voxels are emitted in ascending (ix, iy, iz) (open3d's order is that
 * of an unordered_map, i.e. unspecified). */
int drp_downsample_pcd(drp_ctx* ctx, const double* pcd, int n, double voxel, double* out, int cap, int* m_out);

/* utils.py:423-436 fps(pcd, particle_num, init_idx) for `batch` independent starts:
 * dgl.geometry.farthest_point_sampler on the float32 copy of the cloud (squared float32
 * distances, first maximum), pts_out[batch, npoints, 3] float32 = the chosen points,
 * r_out[batch] = max_i min_j |pcd_i - pts_j| in float64.  init_idx NULL: start b is drawn
 * from `seed` (dgl draws it from torch's global generator). */
int drp_fps_pcd(drp_ctx* ctx, const double* pcd, int n, int npoints, int batch, const int32_t* init_idx,
                uint64_t seed, float* pts_out, double* r_out);

/* utils.py:438-449 fps_rad(pcd, radius): farthest-point sampling of the float64 cloud from
 * `init_idx` (the reference draws it from numpy's global generator) until every point is within
 * `radius` of a sample -- the dataset's sampler (dataset/dataset_gnn_dyn.py:99).  idx_out holds
 * up to `cap` indices into pcd, count_out how many were chosen (cap reached: sampling stops). */
int drp_fps_rad(drp_ctx* ctx, const double* pcd, int n, double radius, int init_idx, int cap, int32_t* idx_out,
                int* count_out);

/* utils.py:468-477 recenter(pcd, sampled_pcd, r): out[b, j] = float32 mean of the cloud points
 * with |p - sampled[b, j]| < r[b] (NaN when there is none, as numpy's mean of an empty set). */
int drp_recenter(drp_ctx* ctx, const double* pcd, int n, const float* sampled, int npoints, int batch,
                 const double* r, float* out);

/* env/flex_env.py:933-951 FlexEnv.obs2ptcl_fixed_num_batch(obs, particle_num, batch_size), the
 * whole chain on the device with one upload and one download: depth_raw = obs[..., -1] [h, w]
 * (world units), depth = depth_raw / global_scale, foreground depth < 0.599/0.8, voxel 0.01,
 * recentering radius min(0.02, 0.5 * particle_r).  ptcl_out[batch, npoints, 3] float64 (float32
 * values, as the reference's array), r_out[batch] = particle_r (dens = 1 / r^2,
 * env/flex_env.py:1022).  n_fg / n_down (nullable) return the cloud sizes. */
int drp_obs2ptcl(drp_ctx* ctx, const float* depth_raw, int h, int w, float global_scale, const double cam[4],
                 int npoints, int batch, const int32_t* init_idx, uint64_t seed, double* ptcl_out, double* r_out,
                 int* n_fg, int* n_down);

/* ---- gradient-descent planner (the reference's live mpc_type 'GD') ----------------------------
 * One iteration of planners.py:682-764: rollout -> final-step reward -> loss = -sum(reward)
 * -> d loss / d pushes by reverse mode (through every step of the horizon) -> Adam(lr) step
 * -> clip box.  actions [B,H,4] with B = traj_num * n_batch rows (row = traj * n_batch + batch).
 * The forward pass writes its tape on the fused engine; when drp_set_engine has selected an fp32 engine, or the weights /
 * inputs are outside the split-fp16 relation encoder's range, on the fp32 matrix engine instead (slower, same gradients
 * to fp32 rounding): drp_gd_* and drp_train_step never return DRP_ERANGE -- env/flex_env.py:973-976 accepts no other
 * planner than this one. */
int drp_gd_begin(drp_ctx* ctx, const float* s0, const float* attr, const float* dens, int nb, int N,
                 const float* actions, int B, int H, double lr, const float act_lo[4], const float act_hi[4]);
/* forward + backward only: rewards [B], d loss / d actions [B,H,4], d loss / d state_pred
 * [B,H,N,3] (each nullable) */
int drp_gd_grad(drp_ctx* ctx, float* rewards_out, float* grad_act_out, float* grad_state_out);
/* one full iteration (forward, backward, Adam, clip); rewards of the iterate BEFORE the update */
int drp_gd_step(drp_ctx* ctx, float* rewards_out);
int drp_gd_get(drp_ctx* ctx, float* actions_out);
/* The same iteration without a host wait: slot (0 .. DRP_GD_SLOTS - 1) takes the iteration's rewards (of the iterate
 * before the update) and the updated pushes in pinned memory, written by the iteration's own kernels (no copy on the
 * stream); drp_gd_wait(slot) blocks until they are there and copies them out (each pointer nullable).  The planner's loop
 * keeps a few iterations enqueued ahead of the one it waits for, so the per-iteration bookkeeping of planners.py:721-738
 * runs beside the device instead of between its iterations.
 * A slot must be waited for before it is used again; drp_gd_begin and the one-shot calls drop what is in flight. */
#define DRP_GD_SLOTS 8
int drp_gd_step_async(drp_ctx* ctx, int slot);
int drp_gd_wait(drp_ctx* ctx, int slot, float* rewards_out, float* actions_out);

/* ---- multi-GPU (RCCL over xGMI) ------------------------------------------------------
 * The library does not link librccl: the first of these calls binds, at run time, $DRP_RCCL_LIB, else the librccl that
 * sits NEXT TO THE HIP RUNTIME THIS LIBRARY RUNS ON (PyTorch's bundled pair when the host imported torch first, /opt/rocm's
 * when torch came later or not at all: a wheel imported afterwards brings a second HIP runtime into the process, and its
 * RCCL would talk to that one), else a librccl the process has already mapped, else /opt/rocm's.  drp_comm_info reports
 * which one.
 * Hang guard: while a communicator is attached, every host wait of the context (drp_sync, drp_mpc_wait, drp_gd_wait,
 * the waits inside the blocking calls) polls the stream, the communicator's asynchronous error and a deadline
 * (env DRP_COMM_TIMEOUT_S, default 60; DRP_COMM_INIT_TIMEOUT_S, default 300, for drp_comm_init).  On error or
 * timeout the communicator is aborted (ncclCommAbort) and the call returns DRP_ECOMM.  The failure is STICKY: until
 * drp_comm_destroy (continue alone) or a fresh drp_comm_init, every entry point that would have combined the ranks' shards
 * (drp_mpc_update_device, drp_mpc_update_elite_device, drp_comm_allgather) returns DRP_ECOMM too -- none of them quietly
 * carries on with this rank's data.  What needs no other rank (rollouts, rewards, fetches) keeps working.  The helper
 * threads behind an abort or an init are given a bounded time to finish by drp_comm_destroy, drp_destroy and at exit. */
int drp_comm_unique_id(char* id128);                       /* ncclGetUniqueId */
/* ncclCommInitRank.  A ncclUniqueId serves one communicator per rank: a second drp_comm_init of the same (id, rank) in
 * this process returns DRP_ECOMM instead of never returning (several contexts of one process, one per GPU, share an id).
 * A rank whose peers do not arrive within DRP_COMM_INIT_TIMEOUT_S gets DRP_ECOMM; a communicator that comes up after that
 * is aborted by the helper that was waiting for it. */
int drp_comm_init(drp_ctx* ctx, const char* id128, int rank, int n_ranks);
int drp_comm_destroy(drp_ctx* ctx);
/* n_ranks = ncclCommCount and rank = ncclCommUserRank of the attached communicator (0 / -1 without one),
 * version = ncclGetVersion of the bound library, path = the file it was loaded from (each pointer nullable). */
int drp_comm_info(drp_ctx* ctx, int* n_ranks, int* rank, int* version, char* path, size_t path_len);
/* All-gather of one host buffer per rank over the context's communicator (upload, ncclAllGather, download):
 * recv [n_ranks][bytes] in rank order.  Without a communicator (or one rank) it copies send to recv.  The planner
 * mirror uses it for its per-iteration bookkeeping record (per-column best reward / index / pushes,
 * planners.py:721-727) when the sample axis is sharded; the update itself stays on the device. */
int drp_comm_allgather(drp_ctx* ctx, const void* send, size_t bytes, void* recv);

/* ---- measurement / debugging ----------------------------------------------------- */
/* HIP-event timing of one kernel class on the context's stream.  name: "graph",
 * "node_encode", "edge_encode", "project", "aggregate", "update", "predict", "reward",
 * "mppi", "prop" (the fused propagation-step kernel of DRP_ENGINE_FUSED).  drp_probe_read returns total ms and launches since drp_probe_begin. */
int drp_probe_begin(drp_ctx* ctx, const char* kernel_class);
int drp_probe_read(drp_ctx* ctx, double* total_ms, long* launches);
/* What the propagation kernels (km_prop, km_prop3, km_rollout) EXECUTED since drp_probe_begin(ctx, "prop+work") -- the "prop"
 * probe with the kernels' own counters switched on (two atomic adds per tile: they cost the 300-particle launch 8 %, so a
 * timed region runs under the plain "prop" probe and the counters over an iteration of their own): out[0] slot iterations that ran the relation
 * encoder's chain (78 16-bit MFMAs each), [1] slot iterations served by the edge-chain cache (none), [2] / [3] tiles of
 * propagation steps that are not / are the last (144 / 96), [4] particle-encoder tiles inside the launch (204),
 * [5] the 16-bit MFMAs (32x32x16, 32 768 FLOP each) those add up to; [6] shader-clock cycles (s_memtime) and [7] 100 MHz
 * ticks (s_memrealtime) between entry and exit of the counted launches, summed over their workgroups: 100 * [6] / [7] is the
 * shader clock in MHz the kernel ran at.  bench.py's roofline numerator and its sclk_mhz_under_load. */
int drp_probe_work(drp_ctx* ctx, unsigned long long out[8]);
/* Which kernel variant served the launches since drp_dispatch_reset (or drp_create): graph build (k_graph, k_graph_q4,
 * k_graph_strips_q<128|256>, k_graph_cells, k_graph_rev, inside km_rollout), propagation kernel with its template flags
 * (km_prop<last|mid,tape,pair,work>, km_prop3<tape|plain,pair,cache|cache+rows,work>, km_rollout<pair|tile32,...>), the
 * stage kernels of the fp32 engines, reverse-mode variant (kmb_rows_bwd, kmb_step_bwd, the stage kernels), training and
 * pre-processing kernels -- names joined by ';' into out (truncated to out_len), return value = the full length.
 * drp_dispatch_variants lists every name the library can report (default_only != 0: without those that need an environment
 * switch or the counting probe).  The shapes and thresholds that select a variant are measured constants
 * (csrc/capi_ctx.h, capi_pipeline.h); tests/test_gpu_fuzz_oracle.py checks every default variant against the oracle through these. */
int drp_dispatch_reset(drp_ctx* ctx);
long drp_last_dispatch(drp_ctx* ctx, char* out, size_t out_len);
long drp_dispatch_variants(int default_only, char* out, size_t out_len);
/* The split-fp16 relation encoder's range shift for the loaded weights: hidden activations travel as two fp16 pieces
 * times 2^-shift; bound = the proven largest activation for the envelope |attr| <= 2, |s_r - s_s| <= 1.5, density <= 10 000;
 * wmax = the largest |weight| packed as fp16; ok = 0 when no shift can carry the weights (every call of the fused / split
 * engine then returns DRP_ERANGE; the gradient-descent planner and the trainer write their tape on the fp32 engine).
 * Each pointer nullable. */
int drp_range_info(drp_ctx* ctx, int* shift, double* bound, double* wmax, int* ok);
/* hold the context's stream for ms (<= 10 000) milliseconds -- what a collective waiting for a dead peer looks like to
 * the host; the tests of the hang guard use it */
int drp_debug_stall(drp_ctx* ctx, int ms);
/* copy an intermediate device buffer to the host: "s_delta","nbr_idx","nbr_cnt",
 * "particle_encode"(eff0),"c_node","c_edge","proj","agg","effect"; the weight blob "w_raw" and its packed copies
 * "w_valu","w_mfma","w_mfma_bwd","w_split","w_split6" (byte sizes: the returned value). returns bytes. */
long drp_debug_fetch(drp_ctx* ctx, const char* name, void* out, size_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DRP_H */
