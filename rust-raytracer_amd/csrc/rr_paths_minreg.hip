// librr_hip.so, second translation unit: the builds of k_wf_paths for scenes with exactly one simple BVH subtree (the Cornell
// box with the Suzanne mesh, the random spheres), compiled with the instruction scheduler that minimises register use
// (-mllvm -amdgpu-sched-strategy=iterative-minreg: the build sets it for this file only).  Same source, same bits; at 5 waves per
// SIMD, 96 registers, the default scheduler's order of these builds spills 32 registers (the lean build: 14), this one's 24
// (4).  Measured, this file's flag on the whole library: Suzanne 5 670 -> 5 969 Msamples/s, random spheres 4 378 -> 4 448
// (1000 spp: 6 310 -> 6 376), final_scene 6 548 -> 6 510 (its builds stay in rr_hip.hip), Cornell box and the 1.1 M-primitive
// mesh the same.
#include <hip/hip_runtime.h>

#include "rr_device.h"

using namespace rrdev;

#define RR_WF_TEMPLATES_ONLY  // (the kernels of the header that are no templates are rr_hip.hip's)
#include "rr_wavefront.h"

// The host-side handle (what hipLaunchKernel takes) of the build for a scene with one simple subtree under the reference's
// semantics (lean: rr_hip.hip, rr_context::lean_scene): rr_hip.hip launches it like its own builds.
using PathsKernel = void (*)(DevScene, WfArgs, float4*, unsigned int*, unsigned long long*);
__attribute__((visibility("hidden"))) PathsKernel rr_paths_kernel_one_subtree(bool lean) {
    return lean ? k_wf_paths<COUNT_OFF, true, true, false, true, RR_RENDER_WAVES_PER_SIMD + 1, true>
                : k_wf_paths<COUNT_OFF, true, true, false, true, RR_RENDER_WAVES_PER_SIMD + 1, false>;
}
