// hgym_fb2_api.hpp -- what hgym_net.hip needs of the 128-row update kernel (hgym_fb2.hpp / hgym_fb2.hip): the tile schedule record, the one
// (actor, critic) shape pair the kernel is instantiated for, and the launch function.
#pragma once
#include "hgym_fused.hpp"

namespace hgym {

struct Fb2Sched {
    int nb;      // row blocks of 16 in the (64-padded) batch
    int T;       // tiles per net: tile i owns row blocks [i * nb / T, (i + 1) * nb / T), at most 8 of them
};
// (first hidden width / 256, input wider than two 128-column chunks) of actor and critic: XBot-L's
constexpr int FB2_NCH_A = 2, FB2_NCH_C = 3;
constexpr bool FB2_STREAM_A = true, FB2_STREAM_C = false;

// reserves the kernel's dynamic LDS, launches grid (tiles, nets) on s; a.dbg / a.nets set by the caller
int32_t launch_fb2(const FwdArgs& a, const FbLoss& L, const Fb2Sched& sch, int tiles, int nets, hipStream_t s);

}  // namespace hgym
