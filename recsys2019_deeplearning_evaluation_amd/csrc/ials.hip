// ials.hip -- IALS solve step on MI355X (gfx950).  (placeholder: entry points report UNSUPPORTED until the kernels land)
#include "common.h"
using namespace mi355rec;
struct mi355rec_ials { int unused; };
#define IALS_TODO() guarded([&] { fail(MI355REC_E_UNSUPPORTED, "IALS device path not built yet"); })
extern "C" int mi355rec_ials_create(mi355rec_ials_t *, int32_t, int32_t, int32_t, double, const int32_t *, const int32_t *, const float *, const double *, const double *) { return IALS_TODO(); }
extern "C" int mi355rec_ials_run_epochs(mi355rec_ials_t, int32_t) { return IALS_TODO(); }
extern "C" int mi355rec_ials_user_half(mi355rec_ials_t, int32_t, int32_t) { return IALS_TODO(); }
extern "C" int mi355rec_ials_item_half(mi355rec_ials_t, int32_t, int32_t) { return IALS_TODO(); }
extern "C" int mi355rec_ials_device_factors(mi355rec_ials_t, double **, double **) { return IALS_TODO(); }
extern "C" int mi355rec_ials_sync(mi355rec_ials_t) { return IALS_TODO(); }
extern "C" int mi355rec_ials_get_factors(mi355rec_ials_t, double *, double *) { return IALS_TODO(); }
extern "C" int mi355rec_ials_get_stats(mi355rec_ials_t, mi355rec_stats *) { return IALS_TODO(); }
extern "C" void mi355rec_ials_destroy(mi355rec_ials_t) {}
