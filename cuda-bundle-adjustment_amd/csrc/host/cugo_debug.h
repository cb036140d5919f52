/* Diagnosis entry points of libcugo_hip_hooks.so (make HOOKS=1) — NOT part of the product ABI (include/cugo_hip.h):
 * the product library neither exports them nor carries the code behind them.  tools/autopsy.py, tools/inject_*.py and
 * tools/delay_check.py load the hooks library (CUGO_LIB=.../libcugo_hip_hooks.so).  DESIGN.md section 2. */
#ifndef CUGO_DEBUG_H
#define CUGO_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* with CUGO_DEBUG_KEEP=1 a solver keeps device copies of the fronts, W, L21 and the solution after each of its first 16
 * factor_solve calls; this writes those of the solver that ran last (its graph still open) to dir/call<k>.bin:
 * int64[8] header {fronts, W, L21, x permuted, x: doubles}, then the arrays. */
int cugo_debug_dump(const char* dir, int* n_calls);
/* the solver that ran last becomes "the reference" (which = 1 below; its graph has to stay open); one slot of either
 * solver (which = 0: the one that ran last) to a file of the same layout; a named plan array of either solver */
int cugo_debug_pin_reference(void);
int cugo_debug_dump_call(int which, int call, const char* path);
int cugo_debug_plan_array(int which, const char* name, const int32_t** out);
#ifdef __cplusplus
}
#endif
#endif
