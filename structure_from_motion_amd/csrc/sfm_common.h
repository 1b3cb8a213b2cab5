// Shared host/device helpers of the libsfm_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sfm_hip.h"

namespace sfmhost {

// thread-local last-error text shared by all translation units (defined in sfm_kernels.hip)
char* error_buffer();
constexpr int kErrorBytes = 512;

inline int fail(int code, const char* msg) {
    snprintf(error_buffer(), kErrorBytes, "%s", msg);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(error_buffer(), kErrorBytes, "%s: %s", what, hipGetErrorString(err));
        return SFM_EHIP;
    }
    return SFM_OK;
}

// Limits of ONE launch (HIP on gfx950): at most 2^31-1 blocks and 2^32-1 threads along x, 65535 blocks along y/z.
// Kernels without a grid-stride loop need every work item covered by a block of their own: their entry points
// refuse sizes beyond that with SFM_EINVAL (SFM_REQUIRE_GRID) instead of silently covering a prefix.
// `per_block` work items per block of `threads` threads.
inline bool grid_fits(int64_t work, int64_t per_block, int64_t threads, int64_t y = 1, int64_t z = 1) {
    if (work < 0 || per_block <= 0 || threads <= 0) return false;
    if (work > (int64_t)1 << 60) return false;
    const int64_t g = (work + per_block - 1) / per_block;
    return g <= 0x7FFFFFFFLL && g * threads <= 0xFFFFFFFFLL && y <= 65535 && z <= 65535;
}

// blocks that cover `work` items exactly once (the caller has checked grid_fits)
inline unsigned grid_for(int64_t work, int64_t block) {
    const int64_t g = (work + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g);
}

// capped grid for kernels that walk their items with a grid-stride loop
inline unsigned grid_stride(int64_t work, int64_t block, int64_t cap) {
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// Second launch of a fused small pass (defined in sfm_score.hip, called by sfm_ransac_pass_small in sfm_kernels.hip):
// SED scoring of all hypotheses with selection and inlier mask done by the block that finishes last.
struct SmallPass {
    const double* corr;
    int64_t n;
    const double* E;
    const int32_t* S;
    const int32_t* flags;
    int64_t h_count;
    double thr, min_extra;
    int aggregation;
    int64_t h_offset;
    int32_t* cnt;
    double* s1;
    double* s2;
    sfm_select_result* result;
    uint8_t* mask;  // may be NULL
    unsigned char* workspace;
    hipStream_t stream;
    const sfm_score_options* options;   // launch options of the scoring launch (NULL: the process-wide defaults)
};
double small_pass_a_scale(double thr);   // factor the prepared a-side coordinates carry for this threshold
// where the fit launch leaves the scoring order of the pass, or NULL when the pass is too small for an order to matter
int32_t* small_pass_order(unsigned char* workspace, int64_t n, int64_t h_count);
int launch_small_score(const SmallPass& pass);

// Scoring of a fused LARGE pass (sfm_ransac_pass_large, sfm_kernels.hip): sfm_score_sed's launches for one pair, except that a
// range-split launch of the matrix-pipe kernel leaves its ranges' partials to be folded inside the pass's selection launch.
struct LargeScore {
    int units;                 // ranges of the points (<= 1: cnt / s1 / s2 are final)
    unsigned char* split;      // their partials: sfm_score_ws.h, [range][hypothesis]
    const unsigned char* fix;  // the hypotheses' sample corrections
};
struct LargePass {
    const double* corr;
    int64_t n;
    const double* E;
    const int32_t* S;
    int64_t h_count;
    double thr;
    int32_t* cnt;
    double* s1;
    double* s2;
    unsigned char* workspace;
    int64_t workspace_bytes;
    unsigned* select_state;    // 16 words the scoring launches zero for the selection launch (NULL: none)
    hipStream_t stream;
    const sfm_score_options* options;   // launch options (NULL: the process-wide defaults)
    int64_t batch = 1;          // image pairs (sfm_ransac_pass_batch; every array with a leading pair dimension)
    bool tables_ready = false;  // launch_large_setup + the fit launch of the pass have prepared maxima, zeroing and both operand tables
};
int launch_large_score(const LargePass& pass, LargeScore* folded_later);
// First launch of a fused pass whose scoring call will take the matrix-pipe kernel: per-block partial maxima of the points and
// every zeroing the pass needs (matrix_setup_kernel).  Fills `tables` with what the fit launch of the pass needs to write both
// operand tables and the sample corrections itself (the MatrixPrep argument of fit_eight_point_kernel); tables->matrix = false
// (and nothing launched) where the scoring call takes another kernel.
struct MatrixTables {
    bool matrix;
    const float4* partial;
    int partials;
    double a_scale;
    uint4* hyp_table;
    unsigned char* fix;
    uint4* table;
    int step_blocks;   // blocks of four steps that write the point table inside the fit launch (one pair); 0: a launch of its own (batches)
};
int launch_large_setup(const LargePass& pass, MatrixTables* tables);
bool score_options_valid(const sfm_score_options* options);   // NULL (the process-wide defaults) is valid

}  // namespace sfmhost

#define SFM_REQUIRE_GRID(fn, work, per_block, ...)                                                     \
    do {                                                                                               \
        if (!sfmhost::grid_fits((work), (per_block), __VA_ARGS__))                                       \
            return sfmhost::fail(SFM_EINVAL, fn ": size exceeds what one launch covers (2^31-1 blocks, " \
                                                "2^32-1 threads in x; 65535 in y/z)");               \
    } while (0)

constexpr int kWave = 64;

// One correspondence in K-normalised coordinates: 32 bytes, read as two 16-byte loads.
struct alignas(32) Corr {
    double xa, ya, xb, yb;
};
