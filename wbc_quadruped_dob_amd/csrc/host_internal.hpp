// Shared by the host translation units of the library (wbc_api.cpp, wbc_multi.cpp): the thread-local error string.
#pragma once
#include <cstddef>
#include <string>
#include "../../include/wbc_hip.h"
namespace wbc {
int fail(int code, const std::string& msg);   // records msg for wbc_last_error(), returns code
// the argument checks of wbc_step_batch (rollout = true: those of wbc_rollout_batch), without touching the device
int check_step_args(const wbc_solver* s, size_t N, const wbc_batch_in* in, const wbc_batch_out* out,
                    const wbc_observer_state* obs, bool rollout);
int check_params_public(const wbc_params* p);   // wbc_solver_set_params' checks, no side effect
}
