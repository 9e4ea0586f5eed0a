// lg_instance.h — one kernel instance per supported topology (legs x joints per leg).
//
// lg_step.hip is compiled once per robot topology (-DLG_LEGS=4: one DPP quad per env, 16 envs per wave; -DLG_LEGS=6: eight lanes per env,
// 8 envs per wave): lanes per env, DOF / body / feet extents and every LDS layout that follows from them are compile-time constants of the
// instance, so the four-legged kernels are exactly what they were before the hexapod existed.  Each instance lives in its own namespace
// (lg4 / lg6: same-named kernels and host helpers must not be merged by the linker) and exports its entry points under a prefixed C name
// (lg4_lg_step, lg6_lg_step, ...); lg_dispatch.cpp defines the ABI's names (include/lgstep.h) on top of them, choosing by
// lg_robot_model.num_legs at lg_create and by the context's first member afterwards.
#pragma once
#ifndef LG_LEGS
#define LG_LEGS 4
#endif
// joints per leg of the instance: 3 for the four- and six-legged robots; the two-legged instance is Cassie's open 2 x 6 chain (cassie.urdf:315-416; the
// knee-spring joints that would close a loop are commented out in the reference's file)
#ifndef LG_JOINTS
#if LG_LEGS == 2
#define LG_JOINTS 6
#else
#define LG_JOINTS 3
#endif
#endif
#define LG_CAT2(a, b) a##b
#define LG_CAT(a, b) LG_CAT2(a, b)
#define LG_NS LG_CAT(lg, LG_LEGS)                           // lg4 / lg6
#define LG_ENTRY(name) LG_CAT(LG_CAT(LG_NS, _), name)       // lg4_lg_step ...

// entry points implemented once per instance: X(return type, name, (parameters), (arguments)); the first parameter is the context
#define LG_INSTANCE_ENTRIES(X)                                                                                                                        \
  X(int, lg_get_tensor, (lg_ctx * c, int id, void** dptr, int64_t shape[4], int32_t* ndim, int32_t* dtype), (c, id, dptr, shape, ndim, dtype))        \
  X(int, lg_step, (lg_ctx * c, const float* actions, void* stream), (c, actions, stream))                                                             \
  X(int, lg_step_physics, (lg_ctx * c, const float* actions, void* stream), (c, actions, stream))                                                     \
  X(int, lg_step_transition, (lg_ctx * c, const float* actions, float* next_obs, const float* values, float gamma, float* rewards, float* dones, void* stream), \
    (c, actions, next_obs, values, gamma, rewards, dones, stream))                                                                                    \
  X(int, lg_step_subset, (lg_ctx * c, const float* actions, const int32_t* env_ids, int32_t n, int32_t rollout_mode, void* stream),                   \
    (c, actions, env_ids, n, rollout_mode, stream))                                                                                                   \
  X(int, lg_set_reward_terms, (lg_ctx * c, int32_t num_terms, const int32_t* term_ids, const float* scales, void* stream), (c, num_terms, term_ids, scales, stream)) \
  X(int, lg_set_async_gait, (lg_ctx * c, const float weights[3], float foot_z_align, void* stream), (c, weights, foot_z_align, stream))               \
  X(int, lg_step_subset_physics, (lg_ctx * c, const float* actions, const int32_t* env_ids, int32_t n, void* stream), (c, actions, env_ids, n, stream)) \
  X(int, lg_post_physics_subset, (lg_ctx * c, const int32_t* env_ids, int32_t n, int32_t rollout_mode, void* stream), (c, env_ids, n, rollout_mode, stream)) \
  X(int, lg_sync_main_to_rollout, (lg_ctx * c, int32_t rollouts_per_main, float pos_drift, void* stream), (c, rollouts_per_main, pos_drift, stream))   \
  X(int, lg_rollout_batch, (lg_ctx * c, const float* all_us, int32_t horizon, const int32_t* env_ids, int32_t n, int32_t rollouts_per_main, float pos_drift, float* rewards, void* stream), \
    (c, all_us, horizon, env_ids, n, rollouts_per_main, pos_drift, rewards, stream))                                                                  \
  X(int, lg_compute_torques, (lg_ctx * c, const float* actions, void* stream), (c, actions, stream))                                                  \
  X(int, lg_simulate, (lg_ctx * c, void* stream), (c, stream))                                                                                        \
  X(int, lg_post_physics_step, (lg_ctx * c, void* stream), (c, stream))                                                                               \
  X(int, lg_reset_idx, (lg_ctx * c, const int32_t* env_ids, int32_t n, int32_t update_curriculum, void* stream), (c, env_ids, n, update_curriculum, stream)) \
  X(int, lg_set_state_indexed, (lg_ctx * c, const float* root_states, const float* dof_state, const int32_t* env_ids, int32_t n, void* stream),       \
    (c, root_states, dof_state, env_ids, n, stream))                                                                                                  \
  X(int, lg_gather_step_rows, (lg_ctx * c, const int32_t* env_ids, int32_t n, float* obs_out, float* rew_out, uint8_t* reset_out, uint8_t* time_out_out, void* stream), \
    (c, env_ids, n, obs_out, rew_out, reset_out, time_out_out, stream))                                                                               \
  X(int, lg_step_subset_rows, (lg_ctx * c, const float* actions, const int32_t* env_ids, int32_t n, int32_t rollout_mode, float* obs_out, float* rew_out, uint8_t* reset_out, uint8_t* time_out_out, void* stream), \
    (c, actions, env_ids, n, rollout_mode, obs_out, rew_out, reset_out, time_out_out, stream))                                                        \
  X(int, lg_set_extra_obs, (lg_ctx * c, const float* dptr), (c, dptr))                                                                                \
  X(int, lg_set_extra_termination, (lg_ctx * c, const uint8_t* dptr), (c, dptr))                                                                     \
  X(int, lg_profile_begin, (lg_ctx * c, int32_t max_samples, int32_t stride), (c, max_samples, stride))                                               \
  X(int, lg_profile_end, (lg_ctx * c, float mean_ms[3], int32_t* nsamples), (c, mean_ms, nsamples))                                                   \
  X(int, lg_debug_read_stamps, (lg_ctx * c, unsigned long long out[64]), (c, out))

#ifdef LG_INSTANCE_TU     // inside lg_step.hip: the definitions below the ABI's names get the instance's prefix
#define lg_abi_sizes LG_ENTRY(lg_abi_sizes)
#define lg_arena_bytes LG_ENTRY(lg_arena_bytes)
#define lg_last_error LG_ENTRY(lg_last_error)
#define lg_destroy LG_ENTRY(lg_destroy)
#define lg_create LG_ENTRY(lg_create)
#define lg_get_tensor LG_ENTRY(lg_get_tensor)
#define lg_step LG_ENTRY(lg_step)
#define lg_step_physics LG_ENTRY(lg_step_physics)
#define lg_step_transition LG_ENTRY(lg_step_transition)
#define lg_step_subset LG_ENTRY(lg_step_subset)
#define lg_set_reward_terms LG_ENTRY(lg_set_reward_terms)
#define lg_set_async_gait LG_ENTRY(lg_set_async_gait)
#define lg_step_subset_physics LG_ENTRY(lg_step_subset_physics)
#define lg_post_physics_subset LG_ENTRY(lg_post_physics_subset)
#define lg_sync_main_to_rollout LG_ENTRY(lg_sync_main_to_rollout)
#define lg_rollout_batch LG_ENTRY(lg_rollout_batch)
#define lg_compute_torques LG_ENTRY(lg_compute_torques)
#define lg_simulate LG_ENTRY(lg_simulate)
#define lg_post_physics_step LG_ENTRY(lg_post_physics_step)
#define lg_reset_idx LG_ENTRY(lg_reset_idx)
#define lg_set_state_indexed LG_ENTRY(lg_set_state_indexed)
#define lg_gather_step_rows LG_ENTRY(lg_gather_step_rows)
#define lg_step_subset_rows LG_ENTRY(lg_step_subset_rows)
#define lg_set_extra_obs LG_ENTRY(lg_set_extra_obs)
#define lg_set_extra_termination LG_ENTRY(lg_set_extra_termination)
#define lg_profile_begin LG_ENTRY(lg_profile_begin)
#define lg_profile_end LG_ENTRY(lg_profile_end)
#define lg_debug_read_stamps LG_ENTRY(lg_debug_read_stamps)
#endif
