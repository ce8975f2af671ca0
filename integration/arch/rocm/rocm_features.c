/* arch/rocm/rocm_features.c -- feature probe of the arch/rocm backend.
 * Pattern: x86_check_features (arch/x86/x86_features.c:69-117) filling struct cpu_features (cpu_features.h:23-37). */
#ifdef ZNG_ROCM_STANDALONE_CHECK
#  include "zlibng_min.h"
#else
#  include "zbuild.h"
#endif
#include "zng_rocm.h"
#include "rocm_functions.h"

void Z_INTERNAL rocm_check_features(struct rocm_cpu_features *features) {
    /* never aborts: no device, or a device that is not gfx950, just leaves the CPU tiers in place */
    features->has_gfx950 = zng_rocm_device_count() > 0 && zng_rocm_init(-1) == ZNG_ROCM_OK;
}
