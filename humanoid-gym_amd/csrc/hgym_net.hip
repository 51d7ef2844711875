// placeholder, replaced by the real implementation
#include "hgym_common.hpp"
using namespace hgym;
extern "C" {
int64_t hgym_net_param_count(const HgymNetConfig*) { return -1; }
int64_t hgym_net_workspace_bytes(const HgymNetConfig*) { return -1; }
int32_t hgym_net_sync_shadow(const HgymNetConfig*, const HgymNet*, void*) { HG_FAIL(HGYM_E_UNSUPPORTED, "not built"); }
int32_t hgym_mlp_forward(const HgymNetConfig*, const HgymNet*, int32_t, int32_t, const float*, int64_t, float*, void*) { HG_FAIL(HGYM_E_UNSUPPORTED, "not built"); }
int32_t hgym_policy_act(const HgymNetConfig*, const HgymNet*, int32_t, const float*, const float*, const float*, uint64_t, const int64_t*, float*, float*, float*, float*, float*, void*) { HG_FAIL(HGYM_E_UNSUPPORTED, "not built"); }
int32_t hgym_ppo_grad(const HgymNetConfig*, const HgymPPOConfig*, const HgymNet*, const HgymBatch*, void*) { HG_FAIL(HGYM_E_UNSUPPORTED, "not built"); }
int32_t hgym_ppo_apply(const HgymNetConfig*, const HgymPPOConfig*, const HgymNet*, void*) { HG_FAIL(HGYM_E_UNSUPPORTED, "not built"); }
}
