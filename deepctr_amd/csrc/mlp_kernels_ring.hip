// dctr_mlp_fwd / dctr_embed_mlp_fwd kernel for 16 batch rows per workgroup with the DNN's weights streamed ONCE per workgroup through an LDS-DMA
// ring shared by its eight waves (mlp_device.h: mlp_ring_kernel) — the small-launch form: launches below 64 rows per CU.
#include "mlp_device.h"

namespace dctr_mlp {

int launch_rt1_ring(const MlpParams& p, const FusedGather& fg, unsigned blocks, size_t lds, int ring_off, hipStream_t stream) {
    static thread_local size_t granted[DCTR_MAX_DEVICES] = {0};
    hipError_t e = dctr_grant_lds((const void*)mlp_ring_kernel<0>, lds, granted);
    DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_fwd(ring): cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
    DCTR_LAUNCH(mlp_ring_kernel<0>, dim3(blocks), dim3(NTHR), lds, stream, p, fg, ring_off);
    return dctr_launch_status("dctr_mlp_fwd(ring)");
}

}  // namespace dctr_mlp
