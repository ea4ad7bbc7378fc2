// ALS / WRMF (als_implicit.pyx:211-352) -- placeholder until the MFMA solve lands.
#include "mfx_common.h"
extern "C" int mfx_als_wrmf(int, int, double *, double *, const int32_t *, int64_t, const int32_t *,
                            const int32_t *, int64_t, const int32_t *, int, int, int, double)
{
    return mfx::fail(MFX_E_STATE, "mfx_als_wrmf: not built yet");
}
