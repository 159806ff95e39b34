// Library management entry points of the C ABI.
#include "common.h"
#include "physdock_hip.h"

PD_EXPORT int pd_abi_version(void) { return PD_ABI_VERSION; }
PD_EXPORT int pd_gemm_args_size(void) { return (int)sizeof(pd_gemm_args); }
PD_EXPORT int pd_attn_args_size(void) { return (int)sizeof(pd_attn_args); }

// hipGraph helpers: the sampler's step loop is host-deterministic (the schedule, the
// noise on/off switch and the physics branch of reference model.py:213,223,252 are known
// before the loop starts), so the whole loop is captured once and replayed.
PD_EXPORT int pd_graph_begin(void* stream) {
    return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}
PD_EXPORT int pd_graph_end(void* stream, void** exec_out) {
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) return PD_ERR_LAUNCH;
    hipGraphExec_t e = nullptr;
    hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (err != hipSuccess) return PD_ERR_LAUNCH;
    *exec_out = (void*)e;
    return PD_OK;
}
PD_EXPORT int pd_graph_launch(void* exec, void* stream) {
    return hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream) == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}
PD_EXPORT int pd_graph_destroy(void* exec) {
    return hipGraphExecDestroy((hipGraphExec_t)exec) == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}
