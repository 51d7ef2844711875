// hgym_update3.hip -- mlp_fb3_kernel (hgym_fb3.hpp: the update's forward + PPO loss + dZ chain of a 64-row tile on eight compute and four
// service wavefronts) in a translation unit, i.e. a device code object, of its own (build.py: every code object below 960 KiB).  Host code
// reaches the kernel through fb3_supported / launch_mlp_fb3 only.
#include "hgym_fb3.hpp"

namespace hgym {

// The shapes the kernel is instantiated for: per net (first hidden width 512, six 128-column input chunks) or (768, two chunks), second /
// third hidden widths 256 / 128 (what the strip maps assume), a head of one 16-column block, 12 actions; inputs from the bf16 shadow.
bool fb3_supported(const FwdArgs& a, int nets) {
    if (nets != 2 || a.A != 12) return false;
    for (int i = 0; i < nets; ++i) {
        const FusedNet& n = a.net[a.net0 + i];
        const int nb = n.layer[0].NB, nc = n.layer[0].KB / 4;
        if (!n.xb || n.layer[0].KB % 4) return false;
        if (!((nb == 32 && nc == 6) || (nb == 48 && nc == 2))) return false;
        if (n.layer[1].N != 256 || n.layer[2].N != 128 || n.layer[3].NB != 1) return false;
        if (n.layer[3].N != (i == 0 ? 12 : 1)) return false;
    }
    return true;
}

int32_t launch_mlp_fb3(const FwdArgs& fb, const FbLoss& fl, int tiles, int nets, size_t lds, hipStream_t s) {
    const int32_t rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_fb3_kernel), lds, "mlp_fb3_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(mlp_fb3_kernel, dim3(tiles, nets), dim3(FB3_THREADS), lds, s, fb, fl);
    return HGYM_OK;
}

}  // namespace hgym
