/* hip_stub.c -- TEST INFRASTRUCTURE. The sanitizer build of the host code has no GPU runtime: the HIP shim's
 * entry points that smolscale_api.c binds are stubbed to fail, which is what a machine without a GPU reports.
 * (The planner entry points are the real ones from smol_plan.c.) */
#include "smol_hip.h"

int smol_hip_device_count (void) { return 0; }
int smol_hip_plan_create (const SmolHipJob *job, int device, SmolHipPlan **plan_out) { (void) job; (void) device; (void) plan_out; return SMOL_HIP_E_NO_DEVICE; }
void smol_hip_plan_destroy (SmolHipPlan *plan) { (void) plan; }
int smol_hip_scale_rows_host_multi (SmolHipPlan *const *plans, uint32_t n_plans, const void *pixels_in_host, uint32_t rowstride_in,
                                    void *dest_host, uint32_t rowstride_out, uint32_t first_row, uint32_t n_rows)
{ (void) plans; (void) n_plans; (void) pixels_in_host; (void) rowstride_in; (void) dest_host; (void) rowstride_out; (void) first_row; (void) n_rows; return SMOL_HIP_E_NO_DEVICE; }
