// hgym_fb2.hip -- mlp_fb2_kernel (hgym_fb2.hpp) in a translation unit, i.e. a DEVICE CODE OBJECT, of its own.
//
// Round 4 found that with a device code object beyond ~1 MiB in the library, runs of eight processes on one GPU abort at random with
// HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION -- whether or not anything of it is ever launched (tests/test_dist_gpu.py, 8 ranks; bisected over
// commits and library variants: hgym_net's code object at 0.87 / 0.92 / 0.99 MB: 0 failures in 5-8 runs each; at 1.15 MB (the first
// mlp_fb2_kernel, 281 KB, inside hgym_net.hip) 3-4 of 6; at 1.19 MB (the update kernel split four ways) 4 of 4).  build.py therefore
// checks every code object against 960 KiB, and this kernel (125 KB) does not ride in hgym_net's.
#include "hgym_fb2.hpp"

namespace hgym {

int32_t launch_fb2(const FwdArgs& a, const FbLoss& L, const Fb2Sched& sch, int tiles, int nets, hipStream_t s) {
    size_t lds = 0;
    for (int i = 0; i < nets; ++i) lds = std::max(lds, (size_t)fb2_lds_bytes(a.net[a.net0 + i]));
    HG_REQUIRE(lds <= 160 * 1024, HGYM_E_UNSUPPORTED, "mlp_fb2_kernel needs %zu bytes of LDS", lds);
    auto* const k = &mlp_fb2_kernel<FB2_NCH_A, FB2_STREAM_A, FB2_NCH_C, FB2_STREAM_C>;
    const int32_t rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, "mlp_fb2_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(tiles, nets), dim3(FB2_NW * 64), lds, s, a, L, sch);
    HG_CHECK_LAUNCH("mlp_fb2_kernel");
    return HGYM_OK;
}

}  // namespace hgym
