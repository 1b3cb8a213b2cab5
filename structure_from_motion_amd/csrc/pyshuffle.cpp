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
    // output words of the current state block, tempered in one vectorisable sweep when the block is (re)generated: the
    // draw loop below then spends one load per draw on them instead of the eight dependent ALU operations of the tempering
    uint32_t tempered[kN];

    Mt19937(uint32_t* state, int position) : mt(state), index(position) {
        for (int k = position < 0 ? 0 : position; k < kN; ++k) tempered[k] = temper(mt[k]);
    }

    // regenerate the 624 state words (Modules/_randommodule.c genrand_uint32, the `mti >= N` branch)
    void regenerate() {
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
        for (int k = 0; k < kN; ++k) tempered[k] = temper(mt[k]);
        index = 0;
    }

    static inline uint32_t temper(uint32_t y) {
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }

    // One descending Fisher-Yates pass over perm[0..n) with CPython's draw sequence.  For all i whose i + 1 has the
    // same bit length k, randbelow(i + 1) takes getrandbits(k) = word >> (32 - k) and redraws while the value is
    // >= i + 1.  The redraw is the unpredictable branch of the literal form (taken up to half the time); here a
    // rejected draw is a swap of perm[i] with itself and leaves i where it was, so the inner loop has no
    // data-dependent branch: one state word, one compare, one conditional move and one (possibly idle) swap per draw.
    void shuffle(int32_t* perm, int64_t n) {
        int64_t i = n - 1;
        while (i >= 1) {
            const int k = 64 - __builtin_clzll((unsigned long long)(i + 1));
            const int shift = 32 - k;
            const int64_t lowest = ((int64_t)1 << (k - 1)) - 1 > 1 ? ((int64_t)1 << (k - 1)) - 1 : 1;  // same k down to here
            while (i >= lowest) {
                if (index >= kN) regenerate();
                const uint32_t* word = tempered + index;
                // every draw lowers i by at most one, so the next `safe` draws need no test of the segment's lower end
                const int64_t room = i - lowest + 1;
                const int safe = (int)(room < (int64_t)(kN - index) ? room : (int64_t)(kN - index));
                for (int used = 0; used < safe; ++used) {
                    const uint32_t r = word[used] >> shift;
                    const bool accept = (int64_t)r <= i;          // r < i + 1
                    const int64_t j = accept ? (int64_t)r : i;    // rejected: swap perm[i] with itself
                    const int32_t tmp = perm[i];
                    perm[i] = perm[j];
                    perm[j] = tmp;
                    i -= accept ? 1 : 0;
                }
                index += safe;
            }
        }
    }
};

}  // namespace

extern "C" int sfm_pyshuffle_table(uint32_t* mt_state, int32_t* mt_index, int64_t n, int64_t iterations,
                                   int32_t* S_out, int32_t* perm_io, int64_t snapshot_iteration,
                                   int32_t* snapshot) {
    if (!mt_state || !mt_index || n < 0 || iterations < 0 || n > 0x7FFFFFFF) return SFM_EINVAL;
    if (*mt_index < 0 || *mt_index > kN) return SFM_EINVAL;
    if (iterations > 0 && !S_out) return SFM_EINVAL;
    Mt19937 gen(mt_state, *mt_index);
    std::vector<int32_t> local;
    int32_t* perm = perm_io;
    if (!perm) {
        local.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) local[(size_t)i] = (int32_t)i;
        perm = local.data();
    }
    const int64_t take = n < 8 ? n : 8;
    for (int64_t it = 0; it < iterations; ++it) {
        gen.shuffle(perm, n);
        for (int64_t k = 0; k < 8; ++k) S_out[it * 8 + k] = k < take ? perm[k] : -1;
        if (it == snapshot_iteration && snapshot) {
            for (int64_t i = 0; i < n; ++i) snapshot[i] = perm[i];
        }
    }
    *mt_index = gen.index;
    return SFM_OK;
}
