// Small persistent host thread pool for the data-parallel passes of the host tail
// (per-overlap trim / type / edge arithmetic on the survivors of containment removal).
#pragma once

#include <stdint.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rala_hip {

class HostPool {
public:
    explicit HostPool(unsigned n_threads) : stop_(false), epoch_(0), pending_(0) {
        n_ = std::max(1u, n_threads);
        for (unsigned t = 1; t < n_; ++t) workers_.emplace_back([this, t]() { loop(t); });
    }
    ~HostPool() {
        {
            std::unique_lock<std::mutex> lk(m_);
            stop_ = true;
            ++epoch_;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    unsigned size() const { return n_; }

    // f(chunk, begin, end) over [0, n) cut into size() contiguous chunks; returns when all ran
    template <class F>
    void chunks(size_t n, F f) {
        if (n_ == 1 || n < 4096) {
            f(0u, (size_t)0, n);
            return;
        }
        job_ = [&](unsigned t) {
            const size_t b = n * t / n_, e = n * (t + 1) / n_;
            if (b < e) f(t, b, e);
        };
        {
            std::unique_lock<std::mutex> lk(m_);
            pending_ = n_ - 1;
            ++epoch_;
        }
        cv_.notify_all();
        job_(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
    }

private:
    void loop(unsigned t) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
            }
            job_(t);
            std::unique_lock<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }

    unsigned n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_;
    uint64_t epoch_;
    unsigned pending_;
    std::function<void(unsigned)> job_;
};

}  // namespace rala_hip
