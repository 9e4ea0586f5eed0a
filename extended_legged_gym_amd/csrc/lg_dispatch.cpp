// lg_dispatch.cpp — the ABI's entry points (include/lgstep.h) on top of the per-topology kernel instances (lg_instance.h): lg_create picks the
// instance from lg_robot_model.num_legs, every later call reads it from the context's first member.
#include <stdint.h>
#include <stddef.h>
#include "../../include/lgstep.h"
#include "lg_instance.h"

namespace {
thread_local int g_last_instance = 4;                         // which instance holds the reason of the last failed lg_create / lg_arena_bytes
inline int legs_of(const lg_ctx* c) { return *reinterpret_cast<const int32_t*>(c); }
const char* const kBadLegs = "lg_robot_model.num_legs must be 4, 6 (three joints per leg) or 2 (six joints per leg): the library holds one kernel instance per supported topology";
inline bool known(int legs) { return legs == 4 || legs == 6 || legs == 2; }
thread_local const char* g_dispatch_err = nullptr;
}

extern "C" {
#define DECL(ret, name, params, args) ret lg4_##name params; ret lg6_##name params; ret lg2_##name params;
LG_INSTANCE_ENTRIES(DECL)
#undef DECL
void lg4_lg_abi_sizes(int32_t out[4]);
size_t lg4_lg_arena_bytes(const lg_config*, const lg_robot_model*, const lg_terrain*);
size_t lg6_lg_arena_bytes(const lg_config*, const lg_robot_model*, const lg_terrain*);
size_t lg2_lg_arena_bytes(const lg_config*, const lg_robot_model*, const lg_terrain*);
lg_ctx* lg2_lg_create(const lg_config*, const lg_robot_model*, const lg_terrain*, int, void*);
const char* lg2_lg_last_error(lg_ctx*);
void lg2_lg_destroy(lg_ctx*);
lg_ctx* lg4_lg_create(const lg_config*, const lg_robot_model*, const lg_terrain*, int, void*);
lg_ctx* lg6_lg_create(const lg_config*, const lg_robot_model*, const lg_terrain*, int, void*);
const char* lg4_lg_last_error(lg_ctx*);
const char* lg6_lg_last_error(lg_ctx*);
void lg4_lg_destroy(lg_ctx*);
void lg6_lg_destroy(lg_ctx*);

#define DEF(ret, name, params, args) \
  ret name params { if (!c) return LG_ERR_INVALID; const int L_ = legs_of(c); return L_ == 6 ? lg6_##name args : (L_ == 2 ? lg2_##name args : lg4_##name args); }
LG_INSTANCE_ENTRIES(DEF)
#undef DEF

int lg_step_rollout(lg_ctx* c, const float* actions, const int32_t* env_ids, int32_t n, float* obs_out, float* rew_out, uint8_t* reset_out, uint8_t* time_out_out, void* stream) {
  return lg_step_subset_rows(c, actions, env_ids, n, 1, obs_out, rew_out, reset_out, time_out_out, stream);
}

void lg_abi_sizes(int32_t out[4]) { lg4_lg_abi_sizes(out); }    // (the structs are the same for every instance)

size_t lg_arena_bytes(const lg_config* cfg, const lg_robot_model* model, const lg_terrain* terrain) {
  g_dispatch_err = nullptr;
  if (!model || !known(model->num_legs)) { g_dispatch_err = kBadLegs; return 0; }
  g_last_instance = model->num_legs;
  return model->num_legs == 6 ? lg6_lg_arena_bytes(cfg, model, terrain) : (model->num_legs == 2 ? lg2_lg_arena_bytes(cfg, model, terrain) : lg4_lg_arena_bytes(cfg, model, terrain));
}

lg_ctx* lg_create(const lg_config* cfg, const lg_robot_model* model, const lg_terrain* terrain, int device_id, void* arena) {
  g_dispatch_err = nullptr;
  if (!model || !known(model->num_legs)) { g_dispatch_err = kBadLegs; return nullptr; }
  g_last_instance = model->num_legs;
  return model->num_legs == 6 ? lg6_lg_create(cfg, model, terrain, device_id, arena)
                              : (model->num_legs == 2 ? lg2_lg_create(cfg, model, terrain, device_id, arena) : lg4_lg_create(cfg, model, terrain, device_id, arena));
}

const char* lg_last_error(lg_ctx* c) {
  if (!c) return g_dispatch_err ? g_dispatch_err : (g_last_instance == 6 ? lg6_lg_last_error(nullptr) : (g_last_instance == 2 ? lg2_lg_last_error(nullptr) : lg4_lg_last_error(nullptr)));
  return legs_of(c) == 6 ? lg6_lg_last_error(c) : (legs_of(c) == 2 ? lg2_lg_last_error(c) : lg4_lg_last_error(c));
}

void lg_destroy(lg_ctx* c) {
  if (!c) return;
  if (legs_of(c) == 6) lg6_lg_destroy(c); else if (legs_of(c) == 2) lg2_lg_destroy(c); else lg4_lg_destroy(c);
}
}  // extern "C"
