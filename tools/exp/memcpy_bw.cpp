// Host memcpy bandwidth with T threads for a 1080p float4 frame (33 MB): g++ -O2 -pthread memcpy_bw.cpp -o /tmp/memcpy_bw
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
int main()
{
    const size_t n = 1920 * 1080 * 16;
    char* a = (char*)malloc(n); char* b = (char*)malloc(n);
    memset(a, 1, n); memset(b, 2, n);
    for (int T : {1, 2, 4, 8, 16}) {
        double best = 1e9;
        for (int rep = 0; rep < 10; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int k = 0; k < T; ++k) th.emplace_back([=] { size_t lo = n * k / T, hi = n * (k + 1) / T; memcpy(b + lo, a + lo, hi - lo); });
            for (auto& t : th) t.join();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        printf("threads %2d: %.3f ms  %.1f GB/s\n", T, best, n / best / 1e6);
    }
    return b[5] == 1 ? 0 : 1;
}
