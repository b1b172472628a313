// Launch entry points of the kernel families that are compiled in translation units of their own (the engine library is four
// .hip files built in parallel: lob_engine.hip -- the C ABI's host side and the update / memo / trace kernels --, lob_tu_env.hip,
// lob_tu_prepass.hip, lob_tu_learn.hip).  Plain host functions: which instantiation runs is decided here, by the same rules
// lob_engine.hip used when it held the launches itself.  Kernels measured and lost (NOTES.md "Round 4") are only compiled with
// -DLOB_EXPERIMENTS (tools/exp_variants.sh); a product build answers LOB_EXPERIMENTS-only requests with the product kernel.
#ifndef LOB_LAUNCH_H
#define LOB_LAUNCH_H

#include <hip/hip_runtime.h>

#include "lob_state.h"

struct EnvFuse {  // env_kernel MODE 1 / 2, env_step_kernel (lob_kernels.h)
    const i32* list;
    const i32* list_n;
    int lpar, sid_prev;
    u64 ver;
};

// 1 in a -DLOB_EXPERIMENTS build (lob_experiments_enabled(), a diagnostic export: the tests of the opt-in variants skip without it)
int lobk_experiments();

// ---- lob_tu_env.hip ----
// env_kernel<lanes, TM, 0>: `lanes` 16 | 64 (32, and 256 = env_compact_kernel: experiments)
void lobk_env(hipStream_t st, int lanes, bool t2, const DevParams* Pd, const DevState& S, const i32* actions, int count_updates, int b0, int nb, int sid, int par);
// env_kernel<64, TM, mode>: 1 = action selection fused (books without a list go on the work list), 2 = the work list's books
void lobk_env_mode(hipStream_t st, bool t2, int mode, const DevParams* Pd, const DevState& S, int nb, int sid, int par, const EnvFuse& F);
// env_step_kernel<inline_general, dq> (two trade slots); half_waves: <false, false, 32> (experiments)
// `lanes16`: env_step16_kernel<inline_general> -- a book's levels across 16 lanes, four books per wave (small batches; one weight vector)
void lobk_env_step(hipStream_t st, bool inline_general, bool dq, bool half_waves, bool lanes16, const DevParams* Pd, const DevState& S, int nb, int sid, int par,
                   const EnvFuse& F, const uint32_t* rnd);
void lobk_clear_inventory(hipStream_t st, const DevParams* Pd, const DevState& S);
void lobk_get_state(hipStream_t st, const DevParams* Pd, const DevState& S, f32* out, f64* reward);
void lobk_dump(hipStream_t st, const DevParams* Pd, const DevState& S, int first, int n, lob_book_dump* out);

// ---- lob_tu_prepass.hip ----
void lobk_gen_events(hipStream_t st, const lob_gen_params& g, int D, int T, u64 first_book, int B, uint32_t* out);
void lobk_repack(hipStream_t st, const uint32_t* src, int D, int T, size_t n_records, uint32_t* dst);
// reset_kernel<lanes, TM> (`roles`: reset2_kernel, `lanes` 16 | 32: experiments)
void lobk_reset(hipStream_t st, int lanes, bool t2, bool roles, const DevParams* Pd, const DevState& S);
void lobk_prepass_extend(hipStream_t st, bool t2, bool roles, const DevParams* Pd, const DevState& S);
void lobk_finalize(hipStream_t st, bool t2, const DevParams* Pd, const DevState& S);

// ---- lob_tu_learn.hip ----
// learn_q_pair_kernel / learn_q_lane_kernel<algo, vt, tr>: vt = 8 when the state has eight variables (else 0)
void lobk_learn_q(hipStream_t st, bool pair, int algo, bool v8, bool tr, int grid, size_t lds, const DevParams* Pd, const DevState& S, const uint32_t* rnd,
                  int lpar, u64 ver, int sid, int acc_fuse);
void lobk_learn_q_fast(hipStream_t st, int algo, int grid, size_t lds, const DevParams* Pd, const DevState& S, const uint32_t* rnd, int lpar, u64 ver);
// hipFuncAttributeMaxDynamicSharedMemorySize of every instantiation above
hipError_t lobk_learn_set_lds(int fast_lds, int lane_lds, int pair_lds);

#endif
