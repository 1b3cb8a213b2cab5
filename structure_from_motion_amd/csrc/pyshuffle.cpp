// HOST code: bit-exact replay of CPython 3.10's `random.shuffle` on an index permutation, so that a
// caller who did `random.seed(k)` gets the same hypothesis samples as the reference's RANSAC driver
// (reference lib/ransac/ransac.py:59-64: cumulative in-place shuffle, first 8 entries = sample).
//
// CPython pieces restated (Lib/random.py, Modules/_randommodule.c of 3.10):
//   shuffle(x):            for i in reversed(range(1, len(x))): j = randbelow(i + 1); swap(x[i], x[j])
//   randbelow(n):          k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
//   getrandbits(k <= 32):  genrand_uint32() >> (32 - k)
//   genrand_uint32():      MT19937 (Matsumoto & Nishimura) with the standard tempering
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../include/sfm_hip.h"

namespace {

constexpr int kN = 624, kM = 397;

struct Mt19937 {
    uint32_t* mt;
    int index;
    // output words of the current state block, tempered in one vectorisable sweep when the block is (re)generated: the
    // draw loop below then spends one load per draw on them instead of the eight dependent ALU operations of the tempering
    uint32_t tempered[kN + 16];
    bool wide;   // AVX-512 available (and not switched off with SFM_PYSHUFFLE_WIDE=0)

    Mt19937(uint32_t* state, int position) : mt(state), index(position) {
        for (int k = position < 0 ? 0 : position; k < kN; ++k) tempered[k] = temper(mt[k]);
        for (int k = kN; k < kN + 16; ++k) tempered[k] = 0u;
        const char* choice = getenv("SFM_PYSHUFFLE_WIDE");
        wide = __builtin_cpu_supports("avx512f") && !(choice && choice[0] == '0');
    }

    // regenerate the 624 state words (Modules/_randommodule.c genrand_uint32, the `mti >= N` branch) and temper them.
    // Word kk is made from the OLD words kk, kk + 1 and the word 397 places on (mod 624: old for kk < 227, already new
    // after that, 227 places back), so sixteen consecutive words can be made at once; the tail of each range and the
    // wrap-around word are made one at a time.
    static inline uint32_t twist(uint32_t upper, uint32_t lower, uint32_t far) {
        const uint32_t y = (upper & 0x80000000u) | (lower & 0x7fffffffu);
        return far ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }

    void regenerate() {
        if (wide) {
            regenerate_wide();
        } else {
            for (int kk = 0; kk < kN - kM; ++kk) mt[kk] = twist(mt[kk], mt[kk + 1], mt[kk + kM]);
            for (int kk = kN - kM; kk < kN - 1; ++kk) mt[kk] = twist(mt[kk], mt[kk + 1], mt[kk + (kM - kN)]);
            mt[kN - 1] = twist(mt[kN - 1], mt[0], mt[kM - 1]);
            for (int k = 0; k < kN; ++k) tempered[k] = temper(mt[k]);
        }
        index = 0;
    }

    __attribute__((target("avx512f"))) inline void sixteen(int kk, int far) {
        const __m512i top = _mm512_set1_epi32((int)0x80000000u), low = _mm512_set1_epi32(0x7fffffff);
        const __m512i one = _mm512_set1_epi32(1), matrix = _mm512_set1_epi32((int)0x9908b0dfu);
        const __m512i y = _mm512_or_si512(_mm512_and_si512(_mm512_loadu_si512(mt + kk), top),
                                          _mm512_and_si512(_mm512_loadu_si512(mt + kk + 1), low));
        const __m512i odd = _mm512_and_si512(_mm512_sub_epi32(_mm512_setzero_si512(), _mm512_and_si512(y, one)), matrix);
        __m512i v = _mm512_xor_si512(_mm512_xor_si512(_mm512_loadu_si512(mt + far), _mm512_srli_epi32(y, 1)), odd);
        _mm512_storeu_si512(mt + kk, v);
        v = _mm512_xor_si512(v, _mm512_srli_epi32(v, 11));
        v = _mm512_xor_si512(v, _mm512_and_si512(_mm512_slli_epi32(v, 7), _mm512_set1_epi32((int)0x9d2c5680u)));
        v = _mm512_xor_si512(v, _mm512_and_si512(_mm512_slli_epi32(v, 15), _mm512_set1_epi32((int)0xefc60000u)));
        v = _mm512_xor_si512(v, _mm512_srli_epi32(v, 18));
        _mm512_storeu_si512(tempered + kk, v);
    }

    __attribute__((target("avx512f"))) void regenerate_wide() {
        int kk = 0;
        for (; kk + 16 <= kN - kM; kk += 16) sixteen(kk, kk + kM);            // 0 .. 223
        for (; kk < kN - kM; ++kk) {                                          // 224 .. 226
            mt[kk] = twist(mt[kk], mt[kk + 1], mt[kk + kM]);
            tempered[kk] = temper(mt[kk]);
        }
        for (; kk + 16 <= kN - 1; kk += 16) sixteen(kk, kk + (kM - kN));      // 227 .. 610
        for (; kk < kN - 1; ++kk) {                                           // 611 .. 622
            mt[kk] = twist(mt[kk], mt[kk + 1], mt[kk + (kM - kN)]);
            tempered[kk] = temper(mt[kk]);
        }
        mt[kN - 1] = twist(mt[kN - 1], mt[0], mt[kM - 1]);
        tempered[kN - 1] = temper(mt[kN - 1]);
        for (int k = kN; k < kN + 16; ++k) tempered[k] = 0u;                  // the padding the wide loads run into
    }

    static inline uint32_t temper(uint32_t y) {
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }

    // One descending Fisher-Yates pass over perm[0..n) with CPython's draw sequence, in two phases.
    //
    // draws(): the accepted values of the pass.  For all i whose i + 1 has the same bit length k, randbelow(i + 1) takes
    // getrandbits(k) = word >> (32 - k) and redraws while the value is >= i + 1.  The redraw is the unpredictable branch
    // of the literal form (taken up to half the time); here every value is written to the next slot of `drawn` and the
    // slot only advances when the value was accepted, so the loop has no data-dependent branch and does not touch the
    // permutation.  With AVX-512 sixteen words are classified at once: a value <= i - 15 is accepted whatever the
    // fifteen draws before it did (i drops by at most one per draw), a value > i is rejected, and the accepted ones are
    // packed with one compress; a block holding a value in between (16 * 15 / 2^k of them) is taken one draw at a time.
    //
    // apply(): the n - 1 swaps perm[i] <-> perm[drawn[n - 1 - i]], which is all that is left.
    //
    // `drawn` has room for n + 15 values (the packed store writes whole vectors).
    void draws(int64_t n, uint32_t* drawn) {
        int64_t i = n - 1;
        while (i >= 1) {
            const int k = 64 - __builtin_clzll((unsigned long long)(i + 1));
            const int shift = 32 - k;
            const int64_t lowest = ((int64_t)1 << (k - 1)) - 1 > 1 ? ((int64_t)1 << (k - 1)) - 1 : 1;  // same k down to here
            if (wide && lowest >= 511) i = draws_wide(i, lowest, shift, drawn);
            while (i >= lowest) {
                if (index >= kN) regenerate();
                const uint32_t* word = tempered + index;
                // every draw lowers i by at most one, so the next `safe` draws need no test of the segment's lower end
                const int64_t room = i - lowest + 1;
                const int safe = (int)(room < (int64_t)(kN - index) ? room : (int64_t)(kN - index));
                for (int used = 0; used < safe; ++used) {
                    const uint32_t r = word[used] >> shift;
                    const bool accept = (int64_t)r <= i;          // r < i + 1
                    *drawn = r;
                    drawn += accept ? 1 : 0;
                    i -= accept ? 1 : 0;
                }
                index += safe;
            }
        }
    }

    // The part of a segment [lowest, i] that is at least sixteen long; returns the i it stopped at, `drawn` advanced.
    // A block of sixteen words is classified against `bound`, the i the PREVIOUS block started from (i <= bound <=
    // i + 16), so that the compare does not wait for the previous block's count: a value <= bound - 31 is accepted
    // whatever the up to 31 draws since then did, a value > bound is rejected, and the few in between ("close", 16 * 31
    // / 2^k per block) are settled one by one, in lane order, against the exact i and the accepted lanes below them.
    __attribute__((target("avx512f,popcnt"))) int64_t draws_wide(int64_t i64, const int64_t lowest, const int shift,
                                                                 uint32_t*& drawn) {
        const __m128i count = _mm_cvtsi32_si128(shift);
        const int floor = (int)lowest + 15;   // a block of sixteen draws may start while i >= floor
        int i = (int)i64, bound = i, at = index;
        uint32_t* out = drawn;
        while (i >= floor) {
            if (at + 16 > kN) {   // the last words of a state block, one draw at a time (i stays inside the segment)
                for (; at < kN; ++at) {
                    const uint32_t value = tempered[at] >> shift;
                    const bool accept = (int)value <= i;   // value < 2^31: n is an int32
                    *out = value;
                    out += accept ? 1 : 0;
                    i -= accept ? 1 : 0;
                }
                regenerate();
                at = 0;
                bound = i;
                continue;
            }
            const __m512i r = _mm512_srl_epi32(_mm512_loadu_si512(tempered + at), count);
            unsigned accepted = _mm512_cmple_epi32_mask(r, _mm512_set1_epi32(bound - 31));
            unsigned close = _mm512_cmple_epi32_mask(r, _mm512_set1_epi32(bound)) & ~accepted;
            while (close) {
                const int lane = __builtin_ctz(close);
                close &= close - 1u;
                const int below = __builtin_popcount(accepted & ((1u << lane) - 1u));
                if ((int)(tempered[at + lane] >> shift) <= i - below) accepted |= 1u << lane;
            }
            _mm512_storeu_si512(out, _mm512_maskz_compress_epi32((__mmask16)accepted, r));
            const int taken = __builtin_popcount(accepted);
            out += taken;
            bound = i;
            i -= taken;
            at += 16;
        }
        index = at;
        drawn = out;
        return i;
    }

    static void apply(int32_t* perm, int64_t n, const uint32_t* drawn) {
        for (int64_t i = n - 1; i >= 1; --i) {
            const uint32_t j = *drawn++;
            const int32_t tmp = perm[i];
            perm[i] = perm[j];
            perm[j] = tmp;
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
    auto after_pass = [&](int64_t it) {
        for (int64_t k = 0; k < 8; ++k) S_out[it * 8 + k] = k < take ? perm[k] : -1;
        if (it == snapshot_iteration && snapshot) {
            for (int64_t i = 0; i < n; ++i) snapshot[i] = perm[i];
        }
    };
    // (making the draws on a second thread a few passes ahead of the swaps was measured and dropped: 50 vs 47 ms at
    // 5 000 x 10 000 on the GPU box's EPYC 9575F, the buffers crossing cores cost what the overlap saves)
    std::vector<uint32_t> drawn((size_t)n + 16);
    for (int64_t it = 0; it < iterations; ++it) {
        gen.draws(n, drawn.data());
        Mt19937::apply(perm, n, drawn.data());
        after_pass(it);
    }
    *mt_index = gen.index;
    return SFM_OK;
}
