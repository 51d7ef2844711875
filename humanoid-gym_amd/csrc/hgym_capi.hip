// hgym_capi.hip -- library-level entry points (version, error string, device probe).
#include "hgym_common.hpp"

namespace hgym {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int device_cus();
}  // namespace hgym

extern "C" {
int32_t hgym_version(void) { return HGYM_VERSION; }
const char* hgym_last_error(void) { return hgym::last_error_buf(); }
int64_t hgym_sizeof(const char* name) {
    if (!name) return -1;
#define HG_SZ(T) if (!strcmp(name, #T)) return (int64_t)sizeof(T)
    HG_SZ(HgymEnvConfig); HG_SZ(HgymStrided); HG_SZ(HgymSimTensors); HG_SZ(HgymEnvState); HG_SZ(HgymEnvOut);
    HG_SZ(HgymEnvNoise); HG_SZ(HgymNetConfig); HG_SZ(HgymPPOConfig); HG_SZ(HgymNet); HG_SZ(HgymBatch);
#undef HG_SZ
    return -1;
}
int32_t hgym_device_cus(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    return hgym::device_cus();
}
}
