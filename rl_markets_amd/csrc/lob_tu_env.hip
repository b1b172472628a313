// Translation unit of the environment-step kernels (lob_launch.h): env_kernel, env_step_kernel, the ClearInventory / state /
// dump kernels.  gfx950 only; there is no CPU execution path in this file.
#define LOB_TU_SPLIT 1
#define LOB_TU_ENV 1
#include <hip/hip_runtime.h>

#include "lob_internal.h"
#include "lob_fast.h"
#include "lob_kernels.h"
#include "lob_envstep.h"

int lobk_experiments() {
#ifdef LOB_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

// env_kernel with 64 books per wave, or 16 when the batch is too small to give every SIMD a wave (the caller's choice)
void lobk_env(hipStream_t st, int lanes, bool t2, const DevParams* Pd, const DevState& S, const i32* actions, int count_updates, int b0, int nb, int sid, int par) {
#define LOB_ENV_LAUNCH(L, TM) hipLaunchKernelGGL((env_kernel<L, TM>), dim3((nb + L - 1) / L), dim3(L), 0, st, Pd, S, actions, count_updates, b0, nb, sid, par)
#ifdef LOB_EXPERIMENTS
    // learner / backtester steps with the event loop compacted across 256-book blocks (opt-in: measured no faster than
    // env_kernel<64> because a step is a chain of dependent look-ups per event and the block keeps its LDS until its unluckiest
    // book is done); 32 books per wave
    if (!actions && lanes == 256) {
        hipLaunchKernelGGL(env_compact_kernel, dim3((nb + LOB_ENVC_BLOCK - 1) / LOB_ENVC_BLOCK), dim3(LOB_ENVC_BLOCK), 0, st, Pd, S, count_updates, b0, nb, sid, par);
        return;
    }
    if (lanes == 32) { if (t2) LOB_ENV_LAUNCH(32, 2); else LOB_ENV_LAUNCH(32, LOB_MAX_TRADES); return; }
#endif
    // (the merged trade list of a pass in 2 register slots instead of LOB_MAX_TRADES when the records have two trade slots)
    if (lanes == 16) { if (t2) LOB_ENV_LAUNCH(16, 2); else LOB_ENV_LAUNCH(16, LOB_MAX_TRADES); }
    else { if (t2) LOB_ENV_LAUNCH(64, 2); else LOB_ENV_LAUNCH(64, LOB_MAX_TRADES); }
#undef LOB_ENV_LAUNCH
}

void lobk_env_mode(hipStream_t st, bool t2, int mode, const DevParams* Pd, const DevState& S, int nb, int sid, int par, const EnvFuse& F) {
    const dim3 grid((nb + 63) / 64), block(64);
    if (mode == 1) {
        if (t2) hipLaunchKernelGGL((env_kernel<64, 2, 1>), grid, block, 0, st, Pd, S, (const i32*)nullptr, 1, 0, nb, sid, par, F);
        else hipLaunchKernelGGL((env_kernel<64, LOB_MAX_TRADES, 1>), grid, block, 0, st, Pd, S, (const i32*)nullptr, 1, 0, nb, sid, par, F);
    } else {
        if (t2) hipLaunchKernelGGL((env_kernel<64, 2, 2>), grid, block, 0, st, Pd, S, (const i32*)nullptr, 1, 0, nb, sid, par, F);
        else hipLaunchKernelGGL((env_kernel<64, LOB_MAX_TRADES, 2>), grid, block, 0, st, Pd, S, (const i32*)nullptr, 1, 0, nb, sid, par, F);
    }
}

void lobk_env_step(hipStream_t st, bool inline_general, bool dq, bool half_waves, bool lanes16, const DevParams* Pd, const DevState& S, int nb, int sid, int par,
                   const EnvFuse& F, const uint32_t* rnd) {
    const dim3 grid((nb + 63) / 64), block(64);
    if (lanes16 && !dq) {
        const dim3 g16((nb + LOB_ENV16_BOOKS - 1) / LOB_ENV16_BOOKS);
        if (inline_general) hipLaunchKernelGGL(env_step16_kernel<true>, g16, block, 0, st, Pd, S.self, sid, par, F, rnd);
        else hipLaunchKernelGGL(env_step16_kernel<false>, g16, block, 0, st, Pd, S.self, sid, par, F, rnd);
        return;
    }
#define LOB_S_ARG S.self   // (the state's device-resident copy: lob_engine.hip sync_state)
#ifdef LOB_EXPERIMENTS
    if (half_waves && !dq && !inline_general) {  // LOB_ENV_STEP_LANES=32: two half-full waves per SIMD (measured slower, NOTES.md)
        hipLaunchKernelGGL((env_step_kernel<false, false, 32>), dim3((nb + 31) / 32), block, 0, st, Pd, LOB_S_ARG, sid, par, F, rnd);
        return;
    }
#endif
    (void)half_waves;
    if (inline_general && dq) hipLaunchKernelGGL((env_step_kernel<true, true>), grid, block, 0, st, Pd, LOB_S_ARG, sid, par, F, rnd);
    else if (dq) hipLaunchKernelGGL((env_step_kernel<false, true>), grid, block, 0, st, Pd, LOB_S_ARG, sid, par, F, rnd);
    else if (inline_general) hipLaunchKernelGGL(env_step_kernel<true>, grid, block, 0, st, Pd, LOB_S_ARG, sid, par, F, rnd);
    else hipLaunchKernelGGL(env_step_kernel<false>, grid, block, 0, st, Pd, LOB_S_ARG, sid, par, F, rnd);
#undef LOB_S_ARG
}

void lobk_clear_inventory(hipStream_t st, const DevParams* Pd, const DevState& S) {
    hipLaunchKernelGGL(clear_inventory_kernel, dim3((S.B + 255) / 256), dim3(256), 0, st, Pd, S);
}
void lobk_get_state(hipStream_t st, const DevParams* Pd, const DevState& S, f32* out, f64* reward) {
    hipLaunchKernelGGL(get_state_kernel, dim3((S.B + 255) / 256), dim3(256), 0, st, Pd, S, out, reward);
}
void lobk_dump(hipStream_t st, const DevParams* Pd, const DevState& S, int first, int n, lob_book_dump* out) {
    hipLaunchKernelGGL(dump_kernel, dim3((n + 63) / 64), dim3(64), 0, st, Pd, S, first, n, out);
}
