// Fork-join pool for the host side of a batch (request compilation): run(parts, fn) calls fn(0..parts-1) on the workers and the caller.
// No HIP in here: tests/native/hostpool_stress.cpp builds it with ThreadSanitizer on the CPU.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vq {

class HostPool {
public:
    explicit HostPool(size_t workers);
    ~HostPool();
    void run(size_t parts, const std::function<void(size_t)>& fn);  // fn(0..parts-1), the caller takes part; returns when all are done
private:
    void worker();
    std::vector<std::thread> threads_;
    std::mutex run_mu_;  // one run() at a time
    // Workers sleep on `generation_` (a futex word): run() publishes the job, bumps it and wakes them all with ONE syscall — no mutex for the
    // woken threads to queue on (a condition variable hands its mutex from thread to thread: 15 workers started ~0.1 ms late, as long as the
    // whole compile of a 1024-query batch should take).  Parts are claimed and counted off with atomics.
    std::atomic<uint32_t> generation_{0};
    std::atomic<uint32_t> done_word_{0};
    std::atomic<uint64_t> ticket_{~0ull};  // (generation of the open job << 32) | next unclaimed part; ~0: closed (hostpool.cpp)
    std::atomic<size_t> pending_{0};
    std::atomic<uint32_t> inside_{0};  // workers inside the claim loop
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t parts_ = 0;
    std::atomic<bool> stop_{false};
    std::mutex err_mu_;
    std::exception_ptr error_;  // the first exception thrown by a part of the running job (rethrown by run())
};

}  // namespace vq
