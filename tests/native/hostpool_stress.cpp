// CPU stress test of veloci_amd/csrc/hostpool.cpp (built with -fsanitize=thread by tests/test_hostpool.py): many back-to-back jobs of changing
// size from two caller threads; every part of every job must run exactly once, and a job's function object must not be touched after run() returns.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>

#include "../../veloci_amd/csrc/hostpool.hpp"

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 20000;
    vq::HostPool pool(7);
    std::atomic<long> failures{0};
    auto driver = [&](unsigned seed) {
        for (int r = 0; r < rounds; ++r) {
            seed = seed * 1664525u + 1013904223u;
            const size_t parts = 1 + (seed >> 16) % 97;
            auto hits = std::make_unique<std::atomic<int>[]>(parts);
            for (size_t i = 0; i < parts; ++i) hits[i].store(0);
            long sum = 0;
            std::atomic<long> total{0};
            {
                std::function<void(size_t)> fn = [&](size_t p) {
                    hits[p].fetch_add(1);
                    total.fetch_add(long(p) + 1);
                };
                pool.run(parts, fn);
            }  // fn is gone: a straggler that still called it would be a use-after-scope
            for (size_t i = 0; i < parts; ++i) {
                if (hits[i].load() != 1) failures.fetch_add(1);
                sum += long(i) + 1;
            }
            if (total.load() != sum) failures.fetch_add(1);
            if ((seed & 0xFF) == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));  // let the workers fall asleep now and then
        }
    };
    std::thread a(driver, 1u), b(driver, 2u);
    a.join();
    b.join();
    std::printf("failures %ld\n", failures.load());
    return failures.load() ? 1 : 0;
}
