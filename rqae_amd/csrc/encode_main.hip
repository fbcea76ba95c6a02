// encode_main.hip -- translation unit of the main path: rqae_encode_lds_kernel for d = 2304 (c = 5, 4), packed-VALU
// arm, with and without the reconstruction and with the Gemma norm folded in.  Built with
// -mllvm -amdgpu-sched-strategy=max-ilp (__graft_entry__.py): +2-4 % on these kernels; the same flag slows the
// d = 256 shape by 20 % and makes the generic kernels spill, so everything else lives in rqae_hip.hip.
#define RQAE_TU_MAIN 1
#include "encode_lds.hpp"

extern "C" __attribute__((visibility("hidden"))) int rqae_main_tu_launch(int c, const void *enc_params, void *stream)
{
    const EncParams &p = *static_cast<const EncParams *>(enc_params);
    if (c == 5) return launch_encode_lds<9, 5>(p, static_cast<hipStream_t>(stream), false);
    if (c == 4) return launch_encode_lds<9, 4>(p, static_cast<hipStream_t>(stream), false);
    return fail("encode: no main-path kernel for c=%d", c);
}

