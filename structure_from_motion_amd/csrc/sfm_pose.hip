// Batched, device-resident pose selection and triangulation for many image pairs (BASELINE.json
// config "256 image pairs x 10k correspondences: E-estimation + cheirality + triangulation end-to-end
// on GPU").  These entry points chain after sfm_select_best / sfm_inlier_mask without a host round
// trip; together they are the device form of reference eight_point.py:181-242 (_recover_r_t) and
// triangulation.py:42-62 (triangulate_points) applied to the RANSAC inliers, as apps/sfm.py:110-186 does.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sfm_common.h"
#include "sfm_math.h"

namespace {

using sfmhost::check_launch;
using sfmhost::fail;
using sfmhost::grid_for;

// Cheirality test (eight_point.py:449-488) of every correspondence of every pair under its 4 candidate
// poses.  Points whose inlier mask is 0 are reported as not passing — and cost nothing: each wave first compacts
// the inliers of its 512-point chunk (ballot + prefix count into an LDS list), then runs the DLT solves only on
// full 64-lane groups of inliers.  With the usual 30-40 % inliers that is ~3 passes per chunk instead of 8 per pose.
constexpr int kChunkPoints = 512;

__global__ __launch_bounds__(256) void cheirality_batched_kernel(
    const Corr* __restrict__ corr, int64_t n, const double* __restrict__ pose_rt,
    const uint8_t* __restrict__ mask, double distance_threshold, uint8_t* __restrict__ pass) {
    __shared__ int32_t list[256 / kWave][kChunkPoints];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int64_t b = blockIdx.y;
    const int64_t base = ((int64_t)blockIdx.x * (256 / kWave) + wave) * kChunkPoints;
    if (base >= n) return;  // whole wave; no block-level barrier below
    const uint8_t* m = mask != nullptr ? mask + b * n : nullptr;
    uint8_t* out = pass + b * 4 * n;
    // The four candidates come in two antipodal pairs, (R, t) and (R, -t) (eight_point.py:210-212; sfm_decompose_essential writes
    // them so): with P1 = [I | 0] the DLT null vector of (R, -t) is that of (R, t) with its last component negated — X' = -X —
    // and R X' - t = -(R X + t), so both depths change sign and the norm stays: ONE solve decides both poses.  Taken only when
    // the second pose of a pair IS the first with t negated, bit for bit (block-uniform); any other pose table is solved pose
    // by pose.  One pair of poses per block in z (more waves in flight; the compaction is cheap enough to repeat).
    const int pose_begin = 2 * (int)blockIdx.z;
    const double* rt0 = pose_rt + (b * 4 + pose_begin) * 12;
    const double* rt1 = rt0 + 12;
    bool antipodal = true;
#pragma unroll
    for (int k = 0; k < 12; ++k) antipodal = antipodal && (k < 9 ? rt1[k] == rt0[k] : rt1[k] == -rt0[k]);
    int total = 0;  // wave-uniform
    for (int s = 0; s < kChunkPoints; s += kWave) {
        const int64_t i = base + s + lane;
        const bool inside = i < n;
        const bool act = inside && (m == nullptr || m[i] != 0);
        if (inside && !act) {
            out[pose_begin * n + i] = 0;
            out[(pose_begin + 1) * n + i] = 0;
        }
        const unsigned long long votes = __ballot(act);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(votes >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)votes, 0));
        if (act) list[wave][total + before] = s + lane;
        total += (int)__popcll(votes);
    }
    const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int j = 0; j < total; j += kWave) {
        const bool active = j + lane < total;
        // tail lanes redo the group's first point so the wave-uniform Jacobi loops see valid data
        const int64_t i = base + list[wave][active ? j + lane : j];
        const Corr p = corr[b * n + i];
        for (int k = 0; k < (antipodal ? 1 : 2); ++k) {
            const double* rt = k == 0 ? rt0 : rt1;
            double P2[12];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                P2[r * 4 + 0] = rt[r * 3 + 0];
                P2[r * 4 + 1] = rt[r * 3 + 1];
                P2[r * 4 + 2] = rt[r * 3 + 2];
                P2[r * 4 + 3] = rt[9 + r];
            }
            double X[3];
            sfm::triangulate_dlt(P1, P2, p.xa, p.ya, p.xb, p.yb, X);
            const double z2 = ((P2[8] * X[0] + P2[9] * X[1]) + P2[10] * X[2]) + P2[11];
            const double norm = sqrt((X[0] * X[0] + X[1] * X[1]) + X[2] * X[2]);
            const bool ok = (X[2] >= -1e-8) && (z2 >= -1e-8) && (norm <= distance_threshold);
            if (active) out[(pose_begin + k) * n + i] = ok ? 1 : 0;
            if (antipodal) {   // the mirrored pose: -X, -z2, the same norm (NaN fails both, as it does when solved)
                const bool mirrored = (-X[2] >= -1e-8) && (-z2 >= -1e-8) && (norm <= distance_threshold);
                if (active) out[(pose_begin + 1) * n + i] = mirrored ? 1 : 0;
            }
        }
    }
}

// Pose vote (eight_point.py:213-237): votes[p] = number of passing correspondences, not counting the one
// at position 0 of the list handed to the reference (np.count_nonzero of the *index* array) — here the
// correspondence `skip_index[b]`; best = first maximum, -1 if every vote is zero.
__global__ __launch_bounds__(256) void pose_vote_kernel(const uint8_t* __restrict__ pass, int64_t n,
                                                        const int32_t* __restrict__ skip_index,
                                                        int32_t* __restrict__ votes,
                                                        int32_t* __restrict__ best) {
    const int64_t b = blockIdx.x;
    const int64_t skip = skip_index != nullptr ? (int64_t)skip_index[b] : -1;
    int cnt[4] = {0, 0, 0, 0};
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int use = (i != skip) ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) cnt[p] += use & (int)pass[(b * 4 + p) * n + i];
    }
    __shared__ int partial[4][4];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int total = sfm::wave_sum(cnt[p]);
        if (lane == 0) partial[wave][p] = total;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int v[4];
        int arg = -1, top = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[p] = partial[0][p] + partial[1][p] + partial[2][p] + partial[3][p];
            votes[b * 4 + p] = v[p];
            if (v[p] > top) {  // strict: the first maximum wins (np.argmax)
                top = v[p];
                arg = p;
            }
        }
        best[b] = arg;
    }
}

// Triangulation (triangulation.py:42-62) of the correspondences that pass the cheirality test of the
// chosen pose: P1 = [K|0], P2 = [K|0] [R t; 0 1], pixel coordinates.
struct Intrinsics {
    double k[9];
};

__global__ __launch_bounds__(256) void triangulate_selected_kernel(
    const double2* __restrict__ pix_a, const double2* __restrict__ pix_b, int64_t n, Intrinsics K,
    const double* __restrict__ pose_rt, const int32_t* __restrict__ best, const uint8_t* __restrict__ pass,
    double* __restrict__ X, uint8_t* __restrict__ valid) {
    // same wave-level compaction as cheirality_batched_kernel: DLT solves only for the points that passed
    __shared__ int32_t list[256 / kWave][kChunkPoints];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int64_t b = blockIdx.y;
    const int64_t base = ((int64_t)blockIdx.x * (256 / kWave) + wave) * kChunkPoints;
    if (base >= n) return;
    const int pose = best[b];
    const bool have_pose = pose >= 0;
    const uint8_t* chosen = pass + (b * 4 + (have_pose ? pose : 0)) * n;
    int total = 0;
    for (int s = 0; s < kChunkPoints; s += kWave) {
        const int64_t i = base + s + lane;
        const bool inside = i < n;
        const bool keep = inside && have_pose && chosen[i] != 0;
        if (inside && !keep) {
            double* out = X + (b * n + i) * 3;
            out[0] = 0.0;
            out[1] = 0.0;
            out[2] = 0.0;
            valid[b * n + i] = 0;
        }
        const unsigned long long votes = __ballot(keep);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(votes >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)votes, 0));
        if (keep) list[wave][total + before] = s + lane;
        total += (int)__popcll(votes);
    }
    if (total == 0) return;
    const double* rt = pose_rt + (b * 4 + pose) * 12;
    double P1[12], P2[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            P1[r * 4 + c] = K.k[r * 3 + c];
            P2[r * 4 + c] = (K.k[r * 3 + 0] * rt[0 * 3 + c] + K.k[r * 3 + 1] * rt[1 * 3 + c]) + K.k[r * 3 + 2] * rt[2 * 3 + c];
        }
        P1[r * 4 + 3] = 0.0;
        P2[r * 4 + 3] = (K.k[r * 3 + 0] * rt[9 + 0] + K.k[r * 3 + 1] * rt[9 + 1]) + K.k[r * 3 + 2] * rt[9 + 2];
    }
    for (int j = 0; j < total; j += kWave) {
        const bool active = j + lane < total;
        const int64_t i = base + list[wave][active ? j + lane : j];
        const double2 a = pix_a[b * n + i];
        const double2 q = pix_b[b * n + i];
        double Xp[3];
        sfm::triangulate_dlt(P1, P2, a.x, a.y, q.x, q.y, Xp);
        if (active) {
            double* out = X + (b * n + i) * 3;
            out[0] = Xp[0];
            out[1] = Xp[1];
            out[2] = Xp[2];
            valid[b * n + i] = 1;
        }
    }
}

}  // namespace

extern "C" {

int sfm_cheirality_batched(const double* corr, int64_t n, int64_t batch, const double* pose_rt,
                           const uint8_t* mask, double distance_threshold, uint8_t* pass, void* stream) {
    if (n < 0 || batch < 0) return fail(SFM_EINVAL, "sfm_cheirality_batched: negative size");
    if (n == 0 || batch == 0) return SFM_OK;
    if (batch > 65535) return fail(SFM_EINVAL, "sfm_cheirality_batched: batch > 65535");
    if (!corr || !pose_rt || !pass) return fail(SFM_EINVAL, "sfm_cheirality_batched: null pointer");
    // one antipodal pair of poses per block in z (one pose per block until round 5: 5.08 ms per C5 batch vs 5.29 with all four
    // poses in one wave, profiles/r01/README.md; the pair shares its solve now)
    SFM_REQUIRE_GRID("sfm_cheirality_batched", n, kChunkPoints * (256 / kWave), 256, batch);
    hipLaunchKernelGGL(cheirality_batched_kernel, dim3(grid_for(n, kChunkPoints * (256 / kWave)), (unsigned)batch, 2),
                       dim3(256), 0, (hipStream_t)stream, (const Corr*)corr, n, pose_rt, mask, distance_threshold, pass);
    return check_launch("cheirality_batched_kernel");
}

int sfm_pose_vote(const uint8_t* pass, int64_t n, int64_t batch, const int32_t* skip_index, int32_t* votes,
                  int32_t* best, void* stream) {
    if (n < 0 || batch < 0) return fail(SFM_EINVAL, "sfm_pose_vote: negative size");
    if (batch == 0) return SFM_OK;
    if (!votes || !best || (n > 0 && !pass)) return fail(SFM_EINVAL, "sfm_pose_vote: null pointer");
    hipLaunchKernelGGL(pose_vote_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, pass, n,
                       skip_index, votes, best);
    return check_launch("pose_vote_kernel");
}

int sfm_triangulate_selected(const double* pix_a, const double* pix_b, int64_t n, int64_t batch,
                             const double* K, const double* pose_rt, const int32_t* best,
                             const uint8_t* pass, double* X, uint8_t* valid, void* stream) {
    if (n < 0 || batch < 0) return fail(SFM_EINVAL, "sfm_triangulate_selected: negative size");
    if (n == 0 || batch == 0) return SFM_OK;
    if (batch > 65535) return fail(SFM_EINVAL, "sfm_triangulate_selected: batch > 65535");
    if (!pix_a || !pix_b || !K || !pose_rt || !best || !pass || !X || !valid)
        return fail(SFM_EINVAL, "sfm_triangulate_selected: null pointer");
    Intrinsics intr;
    for (int j = 0; j < 9; ++j) intr.k[j] = K[j];  // host pointer, passed by value
    SFM_REQUIRE_GRID("sfm_triangulate_selected", n, kChunkPoints * (256 / kWave), 256, batch);
    hipLaunchKernelGGL(triangulate_selected_kernel, dim3(grid_for(n, kChunkPoints * (256 / kWave)), (unsigned)batch),
                       dim3(256), 0, (hipStream_t)stream, (const double2*)pix_a, (const double2*)pix_b, n, intr, pose_rt, best,
                       pass, X, valid);
    return check_launch("triangulate_selected_kernel");
}

}  // extern "C"
