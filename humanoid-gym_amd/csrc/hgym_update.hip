// hgym_update.hip -- mlp_fb_kernel (hgym_fused.hpp: the update's forward + PPO loss + dZ chain of a 64-row tile) in a translation unit,
// i.e. a device code object, of its own.  Its two instantiations are 250 KB of code; inside hgym_net.hip they put that file's code
// object at 868 KB of the 960 KiB build.py allows (a device code object beyond ~1 MiB made 8-process runs on one GPU abort at random in
// round 4: DESIGN.md section 7).  Host code reaches the kernel through launch_mlp_fb only.
#include "hgym_fused.hpp"

namespace hgym {

int32_t launch_mlp_fb(const FwdArgs& fb, const FbLoss& fl, bool shadow, int tiles, int nets, size_t lds, hipStream_t s) {
    const int32_t rc = shadow ? ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_fb_kernel<true>), lds, "mlp_fb_kernel<shadow>")
                              : ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_fb_kernel<false>), lds, "mlp_fb_kernel");
    if (rc) return rc;
    if (shadow) hipLaunchKernelGGL(mlp_fb_kernel<true>, dim3(tiles, nets), dim3(1024), lds, s, fb, fl);
    else hipLaunchKernelGGL(mlp_fb_kernel<false>, dim3(tiles, nets), dim3(1024), lds, s, fb, fl);
    return HGYM_OK;
}

}  // namespace hgym
