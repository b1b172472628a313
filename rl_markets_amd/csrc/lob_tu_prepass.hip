// Translation unit of the once-per-episode kernels (lob_launch.h): the synthetic generator and the record repacker, reset_kernel
// with the market pre-pass, its resumption for long streams, finalize_kernel.  gfx950 only; no CPU execution path.
#define LOB_TU_SPLIT 1
#define LOB_TU_PREPASS 1
#include <hip/hip_runtime.h>

#include "lob_internal.h"
#include "lob_kernels.h"

void lobk_gen_events(hipStream_t st, const lob_gen_params& g, int D, int T, u64 first_book, int B, uint32_t* out) {
    hipLaunchKernelGGL(gen_events_kernel, dim3((B + 255) / 256), dim3(256), 0, st, g, D, T, first_book, B, out);
}
void lobk_repack(hipStream_t st, const uint32_t* src, int D, int T, size_t n_records, uint32_t* dst) {
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, st, src, D, T, n_records, dst);
}

void lobk_reset(hipStream_t st, int lanes, bool t2, bool roles, const DevParams* Pd, const DevState& S) {
#define LOB_RESET_LAUNCH(L, TM) hipLaunchKernelGGL((reset_kernel<L, TM>), dim3((S.B + L - 1) / L), dim3(L), 0, st, Pd, S)
#ifdef LOB_EXPERIMENTS
    if (roles && lanes == 64) {  // the pre-pass on two waves per 64 books (LOB_PREPASS_ROLES=1: measured slower, NOTES.md)
        if (t2) hipLaunchKernelGGL(reset2_kernel<2>, dim3((S.B + 63) / 64), dim3(128), 0, st, Pd, S);
        else hipLaunchKernelGGL(reset2_kernel<LOB_MAX_TRADES>, dim3((S.B + 63) / 64), dim3(128), 0, st, Pd, S);
        return;
    }
    if (lanes == 32) { if (t2) LOB_RESET_LAUNCH(32, 2); else LOB_RESET_LAUNCH(32, LOB_MAX_TRADES); return; }
    if (lanes == 16) { if (t2) LOB_RESET_LAUNCH(16, 2); else LOB_RESET_LAUNCH(16, LOB_MAX_TRADES); return; }
#endif
    (void)lanes; (void)roles;
    if (t2) LOB_RESET_LAUNCH(64, 2); else LOB_RESET_LAUNCH(64, LOB_MAX_TRADES);
#undef LOB_RESET_LAUNCH
}

void lobk_prepass_extend(hipStream_t st, bool t2, bool roles, const DevParams* Pd, const DevState& S) {
#ifdef LOB_EXPERIMENTS
    if (roles) {
        if (t2) hipLaunchKernelGGL(prepass_extend2_kernel<2>, dim3((S.B + 63) / 64), dim3(128), 0, st, Pd, S);
        else hipLaunchKernelGGL(prepass_extend2_kernel<LOB_MAX_TRADES>, dim3((S.B + 63) / 64), dim3(128), 0, st, Pd, S);
        return;
    }
#endif
    (void)roles;
    if (t2) hipLaunchKernelGGL(prepass_extend_kernel<2>, dim3((S.B + 63) / 64), dim3(64), 0, st, Pd, S);
    else hipLaunchKernelGGL(prepass_extend_kernel<LOB_MAX_TRADES>, dim3((S.B + 63) / 64), dim3(64), 0, st, Pd, S);
}

void lobk_finalize(hipStream_t st, bool t2, const DevParams* Pd, const DevState& S) {
    if (t2) hipLaunchKernelGGL(finalize_kernel<2>, dim3((S.B + 255) / 256), dim3(256), 0, st, Pd, S);
    else hipLaunchKernelGGL(finalize_kernel<LOB_MAX_TRADES>, dim3((S.B + 255) / 256), dim3(256), 0, st, Pd, S);
}
