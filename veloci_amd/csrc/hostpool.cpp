#include "hostpool.hpp"

#include <chrono>
#include <climits>
#include <cstdlib>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace vq {

static void futex_wait(std::atomic<uint32_t>* word, uint32_t expected) {
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(word), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
static void futex_wake_all(std::atomic<uint32_t>* word) { syscall(SYS_futex, reinterpret_cast<uint32_t*>(word), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }

HostPool::HostPool(size_t workers) {
    for (size_t i = 0; i < workers; ++i) threads_.emplace_back([this] { worker(); });
}
HostPool::~HostPool() {
    stop_.store(true);
    generation_.fetch_add(1);
    futex_wake_all(&generation_);
    for (auto& t : threads_) t.join();
}
// ticket_ = (generation of the open job << 32) | next unclaimed part; kClosed while run() rewrites the job description.  A part is claimed by a
// compare-exchange that also checks the generation, and only between inside_++ / inside_--: run() closes the ticket, waits for inside_ == 0,
// and only then touches fn_ / parts_ — a worker that woke late for an earlier job can neither claim a part of the new one nor read a
// half-written description.
static constexpr uint64_t kClosed = ~0ull;
void HostPool::worker() {
    uint32_t seen = 0;
    while (true) {
        uint32_t g;
        // After a job the next one usually follows within a batch period: poll for it for a while before sleeping — a sleeping worker's core
        // has gone idle, and waking it takes about as long as the whole job (VQ_POOL_SPIN_US, default 0: measured in DESIGN.md)
        static const long spin_us = std::getenv("VQ_POOL_SPIN_US") ? std::atol(std::getenv("VQ_POOL_SPIN_US")) : 0;
        if (spin_us > 0 && seen != 0) {
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
            while (generation_.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < until)
                for (int k = 0; k < 64; ++k) __builtin_ia32_pause();
        }
        while ((g = generation_.load(std::memory_order_acquire)) == seen) futex_wait(&generation_, seen);
        if (stop_.load()) return;
        // (seq_cst on the inside_ / ticket_ pair, on both sides: the worker's "I am inside" must be visible before it reads the ticket, the
        //  runner's "closed" before it reads inside_ — a store followed by a load of another word needs more than release / acquire)
        inside_.fetch_add(1, std::memory_order_seq_cst);
        while (true) {
            uint64_t t = ticket_.load(std::memory_order_seq_cst);
            if (t == kClosed || uint32_t(t >> 32) != g) break;
            const size_t part = size_t(t & 0xFFFFFFFFull);
            if (part >= parts_) break;
            if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
            try {
                (*fn_)(part);
            } catch (...) {  // kept for the caller of run(): a worker must neither die with it nor leave the part uncounted
                std::lock_guard<std::mutex> g2(err_mu_);
                if (!error_) error_ = std::current_exception();
            }
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                done_word_.store(1, std::memory_order_release);
                futex_wake_all(&done_word_);
            }
        }
        inside_.fetch_sub(1, std::memory_order_acq_rel);
        seen = g;
    }
}
void HostPool::run(size_t parts, const std::function<void(size_t)>& fn) {
    if (parts == 0) return;
    std::lock_guard<std::mutex> one(run_mu_);
    // (the ticket is closed and inside_ == 0 here: constructor state, or the end of the previous run())
    fn_ = &fn;
    parts_ = parts;
    done_word_.store(0, std::memory_order_relaxed);
    pending_.store(parts, std::memory_order_relaxed);
    const uint32_t g = generation_.load(std::memory_order_relaxed) + 1u;
    ticket_.store(uint64_t(g) << 32, std::memory_order_release);
    generation_.store(g, std::memory_order_release);
    futex_wake_all(&generation_);
    while (true) {  // the caller works too
        uint64_t t = ticket_.load(std::memory_order_acquire);
        const size_t part = size_t(t & 0xFFFFFFFFull);
        if (part >= parts) break;
        if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
        try {
            fn(part);
        } catch (...) {
            std::lock_guard<std::mutex> g2(err_mu_);
            if (!error_) error_ = std::current_exception();
        }
        if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) done_word_.store(1, std::memory_order_release);
    }
    for (int spin = 0; done_word_.load(std::memory_order_acquire) == 0; ++spin) {
        if (spin < 4000) __builtin_ia32_pause();
        else futex_wait(&done_word_, 0);
    }
    ticket_.store(kClosed, std::memory_order_seq_cst);
    while (inside_.load(std::memory_order_seq_cst) != 0) std::this_thread::yield();  // nobody still looks at fn_ / parts_ (or holds &fn)
    fn_ = nullptr;
    if (error_) {  // the first exception of any part, rethrown on the calling thread once every part is accounted for
        std::exception_ptr e;
        {
            std::lock_guard<std::mutex> g2(err_mu_);
            std::swap(e, error_);
        }
        std::rethrow_exception(e);
    }
}

}  // namespace vq
