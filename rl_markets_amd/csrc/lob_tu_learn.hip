// Translation unit of the fast learner kernels (lob_launch.h): Q(s', .) + TD error with a lane, two lanes or a wave per book
// (learn_q_lane_kernel, learn_q_pair_kernel, learn_q_fast_kernel; lob_fast.h).  gfx950 only; no CPU execution path.
#define LOB_TU_SPLIT 1
#define LOB_TU_LEARN 1
#include <hip/hip_runtime.h>

#include "lob_internal.h"
#include "lob_fast.h"

#define LOB_QL_ARGS dim3(grid), dim3(pair ? LOB_QP_BLOCK : LOB_QL_BLOCK), lds, st, Pd, S.self, rnd, lpar, ver, sid, acc_fuse
#define LOB_QP_ARGS LOB_QL_ARGS
#define LOB_QL_ONE(A, VT, TR)                                                        \
    do {                                                                             \
        if (pair) hipLaunchKernelGGL((learn_q_pair_kernel<A, VT, TR>), LOB_QP_ARGS); \
        else hipLaunchKernelGGL((learn_q_lane_kernel<A, VT, TR>), LOB_QL_ARGS);      \
    } while (0)
#define LOB_QL_VT(A, TR) do { if (v8) LOB_QL_ONE(A, 8, TR); else LOB_QL_ONE(A, 0, TR); } while (0)

void lobk_learn_q(hipStream_t st, bool pair, int algo, bool v8, bool tr, int grid, size_t lds, const DevParams* Pd, const DevState& S, const uint32_t* rnd,
                  int lpar, u64 ver, int sid, int acc_fuse) {
    if (algo == LOB_ALGO_DOUBLE_Q) LOB_QL_VT(LOB_ALGO_DOUBLE_Q, true);   // (only with the fused Watkins trace step: lob_create)
    else if (algo == LOB_ALGO_QLAMBDA && tr) LOB_QL_VT(LOB_ALGO_QLAMBDA, true);
    else if (algo == LOB_ALGO_QLAMBDA) LOB_QL_VT(LOB_ALGO_QLAMBDA, false);
    else LOB_QL_VT(LOB_ALGO_SARSA, false);
}

void lobk_learn_q_fast(hipStream_t st, int algo, int grid, size_t lds, const DevParams* Pd, const DevState& S, const uint32_t* rnd, int lpar, u64 ver) {
    if (algo == LOB_ALGO_QLAMBDA) hipLaunchKernelGGL((learn_q_fast_kernel<LOB_ALGO_QLAMBDA, LOB_FAST_NB>), dim3(grid), dim3(LOB_FAST_BLOCK), lds, st, Pd, S.self, rnd, lpar, ver);
    else hipLaunchKernelGGL((learn_q_fast_kernel<LOB_ALGO_SARSA, LOB_FAST_NB>), dim3(grid), dim3(LOB_FAST_BLOCK), lds, st, Pd, S.self, rnd, lpar, ver);
}

hipError_t lobk_learn_set_lds(int fast_lds, int lane_lds, int pair_lds) {
    hipError_t er = hipSuccess;
#define LOB_SET(K, BYTES) if (er == hipSuccess) er = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES)
    LOB_SET((learn_q_fast_kernel<LOB_ALGO_SARSA, LOB_FAST_NB>), fast_lds);
    LOB_SET((learn_q_fast_kernel<LOB_ALGO_QLAMBDA, LOB_FAST_NB>), fast_lds);
    LOB_SET((learn_q_lane_kernel<LOB_ALGO_SARSA, 0, false>), lane_lds);
    LOB_SET((learn_q_lane_kernel<LOB_ALGO_QLAMBDA, 0, false>), lane_lds);
    LOB_SET((learn_q_lane_kernel<LOB_ALGO_SARSA, 8, false>), lane_lds);
    LOB_SET((learn_q_lane_kernel<LOB_ALGO_QLAMBDA, 8, false>), lane_lds);
    LOB_SET((learn_q_lane_kernel<LOB_ALGO_QLAMBDA, 0, true>), lane_lds);
    LOB_SET((learn_q_lane_kernel<LOB_ALGO_QLAMBDA, 8, true>), lane_lds);
    LOB_SET((learn_q_lane_kernel<LOB_ALGO_DOUBLE_Q, 0, true>), lane_lds);
    LOB_SET((learn_q_lane_kernel<LOB_ALGO_DOUBLE_Q, 8, true>), lane_lds);
    LOB_SET((learn_q_pair_kernel<LOB_ALGO_SARSA, 0, false>), pair_lds);
    LOB_SET((learn_q_pair_kernel<LOB_ALGO_QLAMBDA, 0, false>), pair_lds);
    LOB_SET((learn_q_pair_kernel<LOB_ALGO_SARSA, 8, false>), pair_lds);
    LOB_SET((learn_q_pair_kernel<LOB_ALGO_QLAMBDA, 8, false>), pair_lds);
    LOB_SET((learn_q_pair_kernel<LOB_ALGO_QLAMBDA, 0, true>), pair_lds);
    LOB_SET((learn_q_pair_kernel<LOB_ALGO_QLAMBDA, 8, true>), pair_lds);
    LOB_SET((learn_q_pair_kernel<LOB_ALGO_DOUBLE_Q, 0, true>), pair_lds);
    LOB_SET((learn_q_pair_kernel<LOB_ALGO_DOUBLE_Q, 8, true>), pair_lds);
#undef LOB_SET
    return er;
}
