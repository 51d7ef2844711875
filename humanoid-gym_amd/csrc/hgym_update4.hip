// hgym_update4.hip -- mlp_fb4_kernel (hgym_fb4.hpp: the update's forward + PPO loss + dZ chain on 128-row tiles, eight compute + four service
// wavefronts) in a translation unit, i.e. a device code object, of its own (build.py: every code object below 960 KiB).  Host code reaches the
// kernel through fb4_supported / launch_mlp_fb4 only.
#define FB3_NO_KERNEL 1
#include "hgym_fb4.hpp"

namespace hgym {

bool fb3_supported(const FwdArgs& a, int nets);      // the same shape pair (hgym_update3.hip)

bool fb4_supported(const FwdArgs& a, int nets) {
    if (!fb3_supported(a, nets)) return false;
    for (int i = 0; i < nets; ++i)
        if (fb4_lds_bytes(a.net[a.net0 + i]) > 160 * 1024) return false;
    return true;
}

// B rows (padded to 64 by the caller's buffers) in tiles of 128
int32_t launch_mlp_fb4(const FwdArgs& fb, const FbLoss& fl, int nets, hipStream_t s) {
    size_t lds = 0;
    for (int i = 0; i < nets; ++i) lds = std::max(lds, (size_t)fb4_lds_bytes(fb.net[fb.net0 + i]));
    const int32_t rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_fb4_kernel), lds, "mlp_fb4_kernel");
    if (rc) return rc;
    const int tiles = (((fb.M + 63) / 64) * 64 + FB4_BM - 1) / FB4_BM;
    hipLaunchKernelGGL(mlp_fb4_kernel, dim3(tiles, nets), dim3(FB3_THREADS), lds, s, fb, fl);
    return HGYM_OK;
}

}  // namespace hgym
