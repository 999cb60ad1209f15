// tf_env_kernels.hip - the fused TriFinger step kernel (roles: tf_roles.h) and its launcher for ONE (EXT, WIDE) pair.
//
// Compiled nine times (Makefile: -DTF_EXT=0|1|2 -DTF_WIDE=0|1, and -DTF_EXT=0|1|2 -DTF_WIDE=2): EXT 0 the headline kernels, 1 the extended domain
// randomisation, 2 the general box object; WIDE 0 the 128-register instantiation (four workgroups per CU), 1 the 256-register one for populations of at
// most 32768 envs, 2 the 256-register one with four helper wavefronts per workgroup (one workgroup per CU: at most 16384 envs; the launches that
// simulate only - the others are served by the WIDE = 1 unit).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tf_roles.h"
#include "tf_launch.h"

#ifndef TF_EXT
#error "compile with -DTF_EXT=0|1|2 -DTF_WIDE=0|1|2"
#endif

// One launch = one or more hooks of the reference step (MODE) for every env of the handle.
// WIDE = false: 128 registers, 4 workgroups per CU (4 wavefronts per SIMD) - populations that fill the chip; WIDE = true: 256 registers, no spills,
// nothing parked in LDS between substeps, the cube role's contact-space records in registers - populations of at most 32768 envs, which never put
// more than two workgroups on a CU, so the occupancy the narrow build buys is not used (tf_create picks).  Same arithmetic.
// HELP: the WIDE kernel in workgroups of eight wavefronts - 0..2 fingers, 3 cube, 4..7 helpers (tf_roles.h: helper_role) - for populations that leave a CU to
// one workgroup: the second wavefront slot of every SIMD, empty otherwise, carries the finger-finger rows (middle-distal: 4..6, distal pass: 7).  Same arithmetic again.
template <int A, bool IS_RESET, bool ASYM, int MODE, int EXT, bool WIDE, bool HELP = false>
__global__ void __launch_bounds__(HELP ? NT_HELP : NT, HELP ? 1 : (WIDE ? 2 : 4)) k_env(const DevParams* __restrict__ Pp, const StepArgs sa, const float* __restrict__ action) {
    __shared__ __attribute__((aligned(16))) float lds[((EXT == 2) ? (WIDE ? LDS_SLOTS_BOX_WIDE : LDS_SLOTS_BOX) : (HELP ? LDS_SLOTS_HELP : LDS_SLOTS)) * WAVE];
    const DevParams& P = *Pp;
    {   // Warm the scalar cache with the parameter block (one dword per 64-byte line) BEFORE the state loads of every workgroup of the
        // launch saturate the L2: the model constants the free motion needs then come out of the constant cache instead of queueing
        // behind that burst.
        const unsigned* pw = reinterpret_cast<const unsigned*>(Pp);
        unsigned touch = 0u;
#pragma unroll
        for (unsigned k = 0; k < sizeof(DevParams) / 64u; ++k) touch |= pw[16u * k];
        asm volatile("" ::"s"(touch));
    }
    Ctx cx;
    cx.tid = (int)threadIdx.x;
    cx.lane = (int)threadIdx.x & (WAVE - 1);
#if defined(TF_ROLE_ROT)         // developer variant: which wavefront of the workgroup takes which role rotates with the workgroup index
    cx.role = __builtin_amdgcn_readfirstlane((((int)threadIdx.x >> 6) + ((int)blockIdx.x >> TF_ROLE_ROT)) & 3);
#else
    cx.role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#endif
    cx.wave_first = (int)blockIdx.x * WAVE;
    const int i_raw = cx.wave_first + cx.lane;
    cx.valid = i_raw < P.N;
    cx.i = cx.valid ? i_raw : (P.N - 1);
    cx.n_valid = (P.N - cx.wave_first < WAVE) ? (P.N - cx.wave_first) : WAVE;
#if defined(TF_ONLY_FINGER)      // developer builds for per-role resource analysis (make resource-usage-roles)
    finger_role<A, IS_RESET, ASYM, MODE, EXT, WIDE>(P, sa, action, lds, cx);
#elif defined(TF_ONLY_CUBE)
    cube_role<A, IS_RESET, ASYM, MODE, EXT, WIDE>(P, sa, action, lds, cx);
#else
    if (cx.role == 3) cube_role<A, IS_RESET, ASYM, MODE, EXT, WIDE, HELP>(P, sa, action, lds, cx);
    else if (HELP && cx.role > 3) helper_role<ASYM, MODE, EXT>(P, sa, lds, cx);
    else finger_role<A, IS_RESET, ASYM, MODE, EXT, WIDE, HELP>(P, sa, action, lds, cx);
#endif
}


template <int MODE, bool IS_RESET>
static void go(const EnvLaunch& a) {
    constexpr int EXT = TF_EXT;
    constexpr bool WIDE = TF_WIDE != 0, HELP = TF_WIDE == 2;
    dim3 grid(a.grid), block(HELP ? NT_HELP : NT);
    if (a.action_dim == 9) {
        if (a.asym) hipLaunchKernelGGL((k_env<9, IS_RESET, true, MODE, EXT, WIDE, HELP>), grid, block, 0, a.stream, a.d_params, a.sa, a.action);
        else hipLaunchKernelGGL((k_env<9, IS_RESET, false, MODE, EXT, WIDE, HELP>), grid, block, 0, a.stream, a.d_params, a.sa, a.action);
    } else {
#if !defined(TF_DEV_MIN)
        if (a.asym) hipLaunchKernelGGL((k_env<18, IS_RESET, true, MODE, EXT, WIDE, HELP>), grid, block, 0, a.stream, a.d_params, a.sa, a.action);
        else hipLaunchKernelGGL((k_env<18, IS_RESET, false, MODE, EXT, WIDE, HELP>), grid, block, 0, a.stream, a.d_params, a.sa, a.action);
#endif
    }
}

#define TF_CAT3_(a, b, c) a##b##_##c
#define TF_CAT3(a, b, c) TF_CAT3_(a, b, c)
void TF_CAT3(tf_launch_env_, TF_EXT, TF_WIDE)(int lm, const EnvLaunch& a) {
    switch (lm) {
    case TF_LM_STEP: go<M_FUSED_STEP, false>(a); break;
    case TF_LM_STEP_RAND: go<M_FUSED_STEP_RAND, false>(a); break;
    case TF_LM_RESET: go<M_FUSED_RESET, true>(a); break;
#if !defined(TF_DEV_MIN)      // developer builds carry the fused launches only
    case TF_LM_SIM: go<M_SIM, false>(a); break;
#if TF_WIDE != 2              // (the helper unit carries the launches that simulate; the host sends the others to the WIDE = 1 unit)
    case TF_LM_RESETS: go<M_RESETS, false>(a); break;
    case TF_LM_TORQUE: go<M_TORQUE, false>(a); break;
    case TF_LM_POST: go<M_POST, false>(a); break;
    case TF_LM_FINISH: go<M_FINISH, false>(a); break;
#endif
#endif
    default: break;
    }
}

int TF_CAT3(tf_occupancy_env_, TF_EXT, TF_WIDE)(int action_dim, bool asym) {
    constexpr int EXT = TF_EXT;
    constexpr bool WIDE = TF_WIDE != 0, HELP = TF_WIDE == 2;
    int n = -1;
    hipError_t e = hipSuccess;
    if (action_dim == 9) {
        if (asym) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_env<9, false, true, M_FUSED_STEP_RAND, EXT, WIDE, HELP>, HELP ? NT_HELP : NT, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_env<9, false, false, M_FUSED_STEP_RAND, EXT, WIDE, HELP>, HELP ? NT_HELP : NT, 0);
    } else {
#if !defined(TF_DEV_MIN)
        if (asym) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_env<18, false, true, M_FUSED_STEP_RAND, EXT, WIDE, HELP>, HELP ? NT_HELP : NT, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_env<18, false, false, M_FUSED_STEP_RAND, EXT, WIDE, HELP>, HELP ? NT_HELP : NT, 0);
#endif
    }
    return e == hipSuccess ? n : -1;
}
