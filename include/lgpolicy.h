/* lgpolicy.h — C ABI of the rollout-collection side of the training loop: the
 * actor / critic MLP forward passes and action sampling of rsl_rl's PPO.act, and RolloutStorage.compute_returns.
 * Library: extended_legged_gym_amd/csrc/liblgstep.so (same library as lgstep.h).  Device pointers unless marked HOST;
 * every call is asynchronous on the caller's hipStream_t; 0 / negative status as in lgstep.h.
 *
 * Reference (vendored rsl_rl): modules/actor_critic.py:38-66 (nn.Sequential of Linear + activation), :120-136
 * (update_distribution / act / get_actions_log_prob / act_inference / evaluate), algorithms/ppo.py:147-159 (PPO.act),
 * storage/rollout_storage.py:145-167 (compute_returns). */
#ifndef LGPOLICY_H
#define LGPOLICY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LG_MLP_MAX_LAYERS 8
enum lg_activation { LG_ACT_ELU = 0, LG_ACT_RELU = 1, LG_ACT_TANH = 2, LG_ACT_LRELU = 3 /* slope 0.01 */, LG_ACT_SELU = 4 };

typedef struct lg_mlp lg_mlp;

/* One nn.Sequential(Linear, act, ..., Linear) (actor_critic.py:42-66).  dims[0 .. num_layers]: input width, hidden widths,
 * output width; weights[l] is Linear.weight of layer l, (dims[l+1], dims[l]) row-major, biases[l] (dims[l+1]) — HOST
 * pointers, re-tiled for the matrix cores and uploaded.  Widths up to 512 per hidden layer. */
lg_mlp* lg_mlp_create(int32_t num_layers, const int32_t* dims, const float* const* weights, const float* const* biases,
                      int32_t activation, int device_id);
void lg_mlp_destroy(lg_mlp* mlp);
const char* lg_mlp_last_error(lg_mlp* mlp);

/* y (n, dims[L]) = mlp(x (n, dims[0])): all layers in one launch, activations stay in LDS, fp32 MFMA
 * (actor(observations) / critic(observations), actor_critic.py:129-136). */
int lg_mlp_forward(lg_mlp* mlp, const float* x, int64_t n, float* y, void* stream);

/* PPO.act (ppo.py:147-159) for a feed-forward ActorCritic: one launch evaluates both networks;
 *   action_mean = actor(obs); actions = action_mean + std * z, z ~ N(0,1) from Philox4x32-10 keyed by (seed, call, row);
 *   actions_log_prob = sum_a [-(a - mean)^2 / (2 std^2) - log std - log sqrt(2 pi)]; values = critic(critic_obs).
 * std: (num_actions) device vector (the `std` parameter, or exp(log_std)).  Outputs: actions, action_mean (n, A),
 * actions_log_prob (n), values (n, critic out width).  deterministic != 0: actions = action_mean (act_inference). */
int lg_policy_act(lg_mlp* actor, lg_mlp* critic, const float* obs, const float* critic_obs, int64_t n, const float* std,
                  uint64_t seed, uint64_t call, int32_t deterministic, float* actions, float* action_mean,
                  float* actions_log_prob, float* values, void* stream);

/* RolloutStorage.compute_returns (rollout_storage.py:145-167): GAE over T transitions of n envs, all (T, n) row-major f32
 * (dones: 0 / 1 as f32), last_values (n); writes returns and advantages (T, n); normalize != 0: advantages =
 * (adv - mean) / (std + 1e-8) with the unbiased std over all T*n entries (torch.std). */
int lg_compute_returns(const float* rewards, const float* dones, const float* values, const float* last_values, int32_t T,
                       int64_t n, float gamma, float lam, int32_t normalize, float* returns, float* advantages, void* stream);

/* The collection loop of OnPolicyRunner.learn (runners/on_policy_runner.py:395-445) for a feed-forward policy, without
 * returning to the host between steps: for t in [0, T):
 *     observations[t] = env obs;  PPO.act (ppo.py:147-159) -> actions[t], values[t], actions_log_prob[t], mu[t], sigma[t];
 *     one lg_step of the env with actions[t];   PPO.process_env_step (ppo.py:161-183): rewards[t] = rew + gamma * values[t] * time_outs,
 *     dones[t] = reset_buf
 * then last_values = critic(env obs) (ppo.py:186-190) and, when returns / advantages are given, compute_returns.
 * Rows are laid out as RolloutStorage holds them (rollout_storage.py:47-76), (T, n, .) row-major f32 (dones as 0 / 1).
 * Sampling call t uses Philox call number first_call + t, so the loop draws exactly what T separate lg_policy_act calls
 * with calls first_call .. first_call + T - 1 draw.  The policy sees the env's obs_buf as both actor and critic input
 * (no privileged observations).  Everything is enqueued on `stream`; nothing is synchronised. */
struct lg_ctx;
typedef struct lg_rollout {
  float* observations;       /* (T, n, num_obs) */
  float* actions;            /* (T, n, A) */
  float* rewards;            /* (T, n) */
  float* dones;              /* (T, n) */
  float* values;             /* (T, n) */
  float* actions_log_prob;   /* (T, n) */
  float* mu;                 /* (T, n, A) */
  float* sigma;              /* (T, n, A) */
  float* last_values;        /* (n) */
  float* returns;            /* (T, n) or NULL */
  float* advantages;         /* (T, n) or NULL */
} lg_rollout;
int lg_collect_rollout(struct lg_ctx* env, lg_mlp* actor, lg_mlp* critic, const float* std, uint64_t seed, uint64_t first_call,
                       int32_t T, float gamma, float lam, int32_t normalize_advantage, const lg_rollout* out, void* stream);

/* ---- the sampling planner's arithmetic around rollout_batch.
 * The reference's planner envs (envs/batch_rollout/robot_traj_grad_sampling.py:210-280) hand `rollout_batch` as a callback to the
 * optimiser of the external package `traj_sampling` (imported at :18, not in the reference tree, no pinned version): per diffusion step
 * it perturbs the node trajectories, interpolates nodes -> dense plans, rolls the plans out, and re-weights the samples.  Restated here
 * from the published algorithm that package implements (DIAL-MPC: Xue et al., "Full-Order Sampling-Based MPC for Torque-Level
 * Locomotion Control via Diffusion-Style Annealing", 2024, with the MPPI update of its reference code; config names:
 * robot_traj_grad_sampling_config.py:44-71 -- num_samples, temp_sample, horizon_samples, horizon_nodes, update_method "mppi"):
 *
 * lg_plan_from_nodes: plans[i, h, a] = sum_k phi[h, k] * nodes[i, k, a]     (n, K, A) -> (n, H, A); phi (H, K) holds the interpolation
 *     weights of the node -> sample-time spline (a linear operator: the host builds it once, linear or cubic);
 * lg_mppi_update: for main env m with sample rows i in [m R, (m + 1) R):
 *     r_i = mean_h rewards[i, h];   z_i = (r_i - mean_i r) / std_i r   (population std; all-equal rewards: z = 0);
 *     weights[i] = softmax_i(z_i / temperature);   new_nodes[m, k, a] = sum_i weights[i] nodes[i, k, a].
 * All pointers are device pointers, row-major f32; asynchronous on `stream`. */
int lg_plan_from_nodes(const float* nodes, const float* phi, int64_t n, int32_t K, int32_t H, int32_t A, float* plans, void* stream);
int lg_mppi_update(const float* rewards, const float* nodes, int32_t num_main, int32_t R, int32_t H, int32_t K, int32_t A,
                   float temperature, float* new_nodes, float* weights, void* stream);


/* The diffusion passes of one control step WITHOUT a return to the host between them (round 6; SURVEY s8(f) rank 4 "sampler glue fused":
 * `optimize_all_trajectories`, robot_traj_grad_sampling.py:226-247, which in the reference is a Python loop of sample -> node2u -> rollout_batch -> softmax per pass).
 *
 * lg_mppi_sample_plans: the samples of one pass.  Row i = m R + s of the launch (main env m, sample s):
 *     nodes[i, k, a] = mean[m, k, a] + sigma_scale * sigma_nodes[k] * z(i, k A + a),   z = 0 for s = 0 (sample 0 is the mean itself), else N(0, 1):
 *         Philox4x32-10 with counter (i, call_lo, (k A + a) >> 1, call_hi) and key (seed_lo, seed_hi), Box-Muller on its first two words
 *         (u1 = max(u01(o0), 2^-24), u2 = u01(o1); cosine for even k A + a, sine for odd) -- the generator of lg_policy_act;
 *     plans[i, h, a] = sum_k phi[h, k] * nodes[i, k, a]                        (what lg_plan_from_nodes computes from the same nodes).
 * lg_planner_diffuse: n_diffuse passes of { lg_mppi_sample_plans (sigma_scale = traj_diffuse_factor^pass, call = call0 + pass) -> lg_rollout_batch on `ctx`
 *     (sync main -> rollout, H rollout steps, sync) -> lg_mppi_update -> mean }, enqueued by ONE call; `mean` (M, K, A) is updated in place, `weights` (M, R) and
 *     `rewards` (M R, H) hold the last pass's.  nodes / plans / rewards are caller-provided workspaces of (M R, K, A), (M R, H, A), (M R, H) floats.
 *     env_ids: the n = M R rollout envs of `ctx` in row order (RobotBatchRollout.rollout_env_indices); A must be the robot's DOF count. */
struct lg_ctx;
int lg_mppi_sample_plans(const float* mean, const float* sigma_nodes, float sigma_scale, const float* phi, int32_t num_main, int32_t R, int32_t K, int32_t H,
                         int32_t A, uint64_t seed, uint64_t call, float* nodes, float* plans, void* stream);
int lg_planner_diffuse(struct lg_ctx* ctx, float* mean, const float* sigma_nodes, const float* phi, int32_t num_main, int32_t R, int32_t K, int32_t H, int32_t A,
                       int32_t n_diffuse, float traj_diffuse_factor, float temperature, uint64_t seed, uint64_t call0, const int32_t* env_ids,
                       int32_t rollouts_per_main, float pos_drift, float* nodes, float* plans, float* rewards, float* weights, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LGPOLICY_H */
