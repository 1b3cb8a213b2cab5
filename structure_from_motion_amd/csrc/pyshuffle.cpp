// HOST code: bit-exact replay of CPython 3.10's `random.shuffle` on an index permutation, so that a
// caller who did `random.seed(k)` gets the same hypothesis samples as the reference's RANSAC driver
// (reference lib/ransac/ransac.py:59-64: cumulative in-place shuffle, first 8 entries = sample).
//
// CPython pieces restated (Lib/random.py, Modules/_randommodule.c of 3.10):
//   shuffle(x):            for i in reversed(range(1, len(x))): j = randbelow(i + 1); swap(x[i], x[j])
//   randbelow(n):          k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
//   getrandbits(k <= 32):  genrand_uint32() >> (32 - k)
//   genrand_uint32():      MT19937 (Matsumoto & Nishimura) with the standard tempering
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/sfm_hip.h"

namespace {

constexpr int kN = 624, kM = 397;

struct Mt19937 {
    uint32_t* mt;
    int index;

    uint32_t next() {
        if (index >= kN) {
            static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
            int kk = 0;
            for (; kk < kN - kM; ++kk) {
                const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + kM] ^ (y >> 1) ^ mag01[y & 1u];
            }
            for (; kk < kN - 1; ++kk) {
                const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (kM - kN)] ^ (y >> 1) ^ mag01[y & 1u];
            }
            const uint32_t y = (mt[kN - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[kN - 1] = mt[kM - 1] ^ (y >> 1) ^ mag01[y & 1u];
            index = 0;
        }
        uint32_t y = mt[index++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }

    uint32_t below(uint32_t n) {  // n >= 1, n < 2^31
        const int k = 32 - __builtin_clz(n);
        uint32_t r = next() >> (32 - k);
        while (r >= n) r = next() >> (32 - k);
        return r;
    }
};

}  // namespace

extern "C" int sfm_pyshuffle_table(uint32_t* mt_state, int32_t* mt_index, int64_t n, int64_t iterations,
                                   int32_t* S_out, int32_t* perm_io, int64_t snapshot_iteration,
                                   int32_t* snapshot) {
    if (!mt_state || !mt_index || n < 0 || iterations < 0 || n > 0x7FFFFFFF) return SFM_EINVAL;
    if (*mt_index < 0 || *mt_index > kN) return SFM_EINVAL;
    if (iterations > 0 && !S_out) return SFM_EINVAL;
    Mt19937 gen{mt_state, *mt_index};
    std::vector<int32_t> local;
    int32_t* perm = perm_io;
    if (!perm) {
        local.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) local[(size_t)i] = (int32_t)i;
        perm = local.data();
    }
    const int64_t take = n < 8 ? n : 8;
    for (int64_t it = 0; it < iterations; ++it) {
        for (int64_t i = n - 1; i >= 1; --i) {
            const uint32_t j = gen.below((uint32_t)(i + 1));
            const int32_t tmp = perm[i];
            perm[i] = perm[j];
            perm[j] = tmp;
        }
        for (int64_t k = 0; k < 8; ++k) S_out[it * 8 + k] = k < take ? perm[k] : -1;
        if (it == snapshot_iteration && snapshot) {
            for (int64_t i = 0; i < n; ++i) snapshot[i] = perm[i];
        }
    }
    *mt_index = gen.index;
    return SFM_OK;
}
