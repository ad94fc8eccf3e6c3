// ct_fresh.h -- loads of device-side control words.
//
// Words one kernel writes and a LATER launch reads at a wave-uniform address (a cell's coordinates updated in place round after round, a `done`
// flag, a list length, an overflow latch) are read with agent-scope loads.  As plain loads the compiler turns them into scalar loads, and
// ct_correct.hip's centre-of-mass kernel then saw the coordinates of TWO rounds ago now and then -- only while kernels of another stream (the
// U-Net, a GEMM) were running, 10-50 % of the calls on every box tried, never on an idle GPU (scripts/probe/corr_beside_unet.py;
// FrameChain.run_sequence is what runs the correction and the watershed beside a U-Net).  Which cache kept the old word was not established
// (the scalar cache is the suspect: per-thread vector loads of the same array were never stale); the sc1 loads below do not depend on the answer.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float fresh_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int fresh_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int fresh_u32(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
