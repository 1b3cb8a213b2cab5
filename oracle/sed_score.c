/* CPU ORACLE (test infrastructure, not product code): plain-C restatement of the reference's RANSAC
 * scoring loop, used (a) by tests as a second, independent checker of oracle/sfm_oracle.py and
 * (b) by bench.py as the timed `cpu_baseline` ("port").
 *
 * Follows reference lib/ransac/ransac.py:66-82 with lib/epipolar/epipolar_ransac.py:18-25 and
 * lib/epipolar/sed.py:7-30 as the scorer, on K-normalised coordinates (the normalisation of
 * eight_point.py:127-133 is hoisted out of the loop, it does not depend on the hypothesis).
 * Floating-point order is the one documented in oracle/sfm_oracle.py: left-to-right 3-term dot
 * products with separate multiply/add roundings (compiled with -ffp-contract=off), sums over the 8
 * sample points first and then the surviving non-sample points in index order.
 */
#include <stdint.h>
#include <omp.h>

static inline double sed_value(const double* e, double xa, double ya, double xb, double yb) {
    const double lb0 = (xb * e[0] + yb * e[3]) + e[6];
    const double lb1 = (xb * e[1] + yb * e[4]) + e[7];
    const double lb2 = (xb * e[2] + yb * e[5]) + e[8];
    const double r = (lb0 * xa + lb1 * ya) + lb2;
    const double la0 = (e[0] * xa + e[1] * ya) + e[2];
    const double la1 = (e[3] * xa + e[4] * ya) + e[5];
    const double da = la0 * la0 + la1 * la1;
    const double db = lb0 * lb0 + lb1 * lb1;
    return (1.0 / da + 1.0 / db) * (r * r);
}

/* corr [n][4] = {xa, ya, xb, yb}; E [h][9]; S [h][8]; outputs [h].  Returns the thread count used. */
int sfm_oracle_score(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count,
                     double thr, int32_t* cnt, double* s1, double* s2, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    int used = 1;
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static)
        for (int64_t h = 0; h < h_count; ++h) {
            const double* e = E + h * 9;
            const int32_t* smp = S + h * 8;
            double sum1 = 0.0, sum2 = 0.0;
            for (int k = 0; k < 8; ++k) {
                const double* p = corr + (int64_t)smp[k] * 4;
                const double sed = sed_value(e, p[0], p[1], p[2], p[3]);
                sum1 += sed;
                sum2 += sed * sed;
            }
            int32_t c = 0;
            for (int64_t i = 0; i < n; ++i) {
                const double* p = corr + i * 4;
                const double sed = sed_value(e, p[0], p[1], p[2], p[3]);
                if (sed <= thr) {
                    int in_sample = 0;
                    for (int k = 0; k < 8; ++k) in_sample |= (smp[k] == (int32_t)i);
                    if (!in_sample) {
                        ++c;
                        sum1 += sed;
                        sum2 += sed * sed;
                    }
                }
            }
            cnt[h] = c;
            s1[h] = sum1;
            s2[h] = sum2;
        }
    }
    return used;
}

/* SED of n correspondences under one E (value-level check of the scalar routine). */
void sfm_oracle_sed_values(const double* corr, int64_t n, const double* e, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = sed_value(e, corr[i * 4], corr[i * 4 + 1], corr[i * 4 + 2], corr[i * 4 + 3]);
}
