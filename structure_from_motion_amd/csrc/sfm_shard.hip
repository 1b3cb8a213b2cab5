// Cross-shard model selection for hypothesis-sharded RANSAC: every rank (GPU) selects over its
// own block of hypotheses and publishes ONE 40-byte sfm_select_result; after a single all-gather of those records
// each rank folds them with the rule below and so arrives at the same global winner with no further exchange.
//
// The rule is the sequential one of the reference (lib/ransac/ransac.py:83-86: a model replaces the incumbent only
// when its error is STRICTLY lower, so among equal errors the earliest iteration wins) applied across shards:
// minimum error key first, then minimum global hypothesis index.  Flag statistics (degenerate samples,
// lib/epipolar/eight_point.py:415-421) combine as min(first index) / sum(count), so that every rank takes the
// same raise-or-skip decision the single-GPU path takes (ransac.py:65 lets the fitter's exception abort the call).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sfm_common.h"

namespace {

using sfmhost::check_launch;
using sfmhost::fail;
using sfmhost::grid_for;

constexpr uint64_t kNoModelKey = 0x7FFFFFFFFFFFFFFFull;

// gathered: [world][batch] (rank-major, the layout an all-gather of each rank's [batch] records produces).
__host__ __device__ inline void fold_records(const sfm_select_result* gathered, int64_t world, int64_t batch, int64_t b,
                                             sfm_select_result* global, int64_t* best_h, sfm_select_result* single) {
    sfm_select_result g;
    g.key = kNoModelKey;
    g.best_h = -1;
    g.best_err = INFINITY;
    g.first_flagged = INT64_MAX;
    g.n_flagged = 0;
    g.best_cnt = 0;
    int64_t flagged = 0;
    for (int64_t r = 0; r < world; ++r) {
        const sfm_select_result c = gathered[r * batch + b];
        if (c.best_h >= 0 && c.key != kNoModelKey &&
            (c.key < g.key || (c.key == g.key && (g.best_h < 0 || c.best_h < g.best_h)))) {
            g.key = c.key;
            g.best_h = c.best_h;
            g.best_err = c.best_err;
            g.best_cnt = c.best_cnt;
        }
        if (c.first_flagged < g.first_flagged) g.first_flagged = c.first_flagged;
        flagged += c.n_flagged > 0 ? c.n_flagged : 0;
    }
    g.n_flagged = flagged > 0x7FFFFFFF ? 0x7FFFFFFF : (int32_t)flagged;  // saturate: only "> 0" and the order of magnitude matter
    if (global) global[b] = g;
    if (best_h) best_h[b] = g.best_h;
    if (single) {  // the same record addressed at a one-hypothesis E / S array (the locally re-derived winner)
        sfm_select_result s = g;
        s.best_h = g.best_h >= 0 ? 0 : -1;
        single[b] = s;
    }
}

__global__ void fold_records_kernel(const sfm_select_result* __restrict__ gathered, int64_t world, int64_t batch,
                                    sfm_select_result* __restrict__ global, int64_t* __restrict__ best_h,
                                    sfm_select_result* __restrict__ single) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) fold_records(gathered, world, batch, b, global, best_h, single);
}

int check_args(const char* fn, const void* gathered, int64_t world, int64_t batch) {
    if (world < 1 || batch < 0) return fail(SFM_EINVAL, fn);
    if (batch > 0 && !gathered) return fail(SFM_EINVAL, fn);
    return SFM_OK;
}

}  // namespace

extern "C" {

int sfm_fold_select_records(const sfm_select_result* gathered, int64_t world, int64_t batch,
                            sfm_select_result* global, int64_t* best_h, sfm_select_result* single, void* stream) {
    if (check_args("sfm_fold_select_records: need world >= 1, batch >= 0 and a record array", gathered, world, batch) != SFM_OK)
        return SFM_EINVAL;
    if (batch == 0) return SFM_OK;
    SFM_REQUIRE_GRID("sfm_fold_select_records", batch, 64, 64);
    hipLaunchKernelGGL(fold_records_kernel, dim3(grid_for(batch, 64)), dim3(64), 0, (hipStream_t)stream, gathered, world,
                       batch, global, best_h, single);
    return check_launch("fold_records_kernel");
}

int sfm_fold_select_records_host(const sfm_select_result* gathered, int64_t world, int64_t batch,
                                 sfm_select_result* global, int64_t* best_h, sfm_select_result* single) {
    if (check_args("sfm_fold_select_records_host: need world >= 1, batch >= 0 and a record array", gathered, world, batch) != SFM_OK)
        return SFM_EINVAL;
    for (int64_t b = 0; b < batch; ++b) fold_records(gathered, world, batch, b, global, best_h, single);
    return SFM_OK;
}

}  // extern "C"
