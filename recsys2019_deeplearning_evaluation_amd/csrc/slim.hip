// slim.hip -- SLIM-BPR epoch on MI355X (gfx950).  (placeholder: entry points report UNSUPPORTED until the kernels land)
#include "common.h"
using namespace mi355rec;
struct mi355rec_slim { int unused; };
#define SLIM_TODO() guarded([&] { fail(MI355REC_E_UNSUPPORTED, "SLIM-BPR device path not built yet"); })
extern "C" int mi355rec_slim_create(mi355rec_slim_t *, const mi355rec_slim_config *, int32_t, int32_t, const int32_t *, const int32_t *) { return SLIM_TODO(); }
extern "C" int mi355rec_slim_run_epochs(mi355rec_slim_t, int32_t) { return SLIM_TODO(); }
extern "C" int mi355rec_slim_run_samples(mi355rec_slim_t, const int32_t *, const int32_t *, const int32_t *, int64_t) { return SLIM_TODO(); }
extern "C" int mi355rec_slim_get_S_topk(mi355rec_slim_t, int32_t, int32_t *, float *) { return SLIM_TODO(); }
extern "C" int mi355rec_slim_get_S_dense(mi355rec_slim_t, float *) { return SLIM_TODO(); }
extern "C" int mi355rec_slim_get_stats(mi355rec_slim_t, mi355rec_stats *) { return SLIM_TODO(); }
extern "C" void mi355rec_slim_destroy(mi355rec_slim_t) {}
