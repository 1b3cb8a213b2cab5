// one call of the shuffle replay, timed: clang++ -O3 -std=c++17 -Iinclude tools/micro/shuffle_replay.cpp && ./a.out 5000 10000
#include "../../structure_from_motion_amd/csrc/pyshuffle.cpp"
#include <chrono>
int main(int argc, char** argv) {
    int64_t n = atoll(argv[1]), it = atoll(argv[2]);
    for (int rep = 0; rep < 3; ++rep) {
    std::vector<uint32_t> st(624); for (int k = 0; k < 624; ++k) st[k] = 1812433253u * k + 12345u;
    int32_t idx = 624; std::vector<int32_t> S(it * 8);
    auto t0 = std::chrono::steady_clock::now();
    sfm_pyshuffle_table(st.data(), &idx, n, it, S.data(), nullptr, -1, nullptr);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long long sum = 0; for (auto v : S) sum += v;
    printf("%.1f ms  %.2f ns/elem  sum %lld\n", s * 1e3, s / (n * it) * 1e9, sum);
    }
}
