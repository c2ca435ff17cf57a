/*!
 * @file thread_pool.hpp
 *
 * @brief Task pool with the interface the callers of rvaser/rala use (vendor/thread_pool is an
 * un-vendored submodule there: createThreadPool + submit_task returning a std::future, call sites
 * src/graph.cpp:235,369 and the include of src/main.cpp:7).  In this build the per-pile fan-out
 * runs on the GPU; the pool serves host-side callers and lets programs written against the
 * reference's headers compile unchanged.
 */

#pragma once

#include <stdint.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace thread_pool {

class ThreadPool;
std::unique_ptr<ThreadPool> createThreadPool(uint32_t num_threads = std::thread::hardware_concurrency() / 2);

class ThreadPool {
public:
    ~ThreadPool() {
        {
            std::lock_guard<std::mutex> hold(lock_);
            closing_ = true;
        }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
    }

    uint32_t num_threads() const { return (uint32_t)workers_.size(); }

    template <class F, class... Args>
    auto submit_task(F&& f, Args&&... args) -> std::future<typename std::result_of<F(Args...)>::type> {
        using R = typename std::result_of<F(Args...)>::type;
        auto job = std::make_shared<std::packaged_task<R()>>(std::bind(std::forward<F>(f), std::forward<Args>(args)...));
        std::future<R> result = job->get_future();
        {
            std::lock_guard<std::mutex> hold(lock_);
            queue_.emplace_back([job]() { (*job)(); });
        }
        wake_.notify_one();
        return result;
    }

    friend std::unique_ptr<ThreadPool> createThreadPool(uint32_t num_threads);

private:
    explicit ThreadPool(uint32_t n) {
        for (uint32_t i = 0; i < n; ++i) workers_.emplace_back([this]() { work(); });
    }
    ThreadPool(const ThreadPool&) = delete;
    const ThreadPool& operator=(const ThreadPool&) = delete;

    void work() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> hold(lock_);
                wake_.wait(hold, [this]() { return closing_ || !queue_.empty(); });
                if (queue_.empty()) return;          // closing and drained
                job = std::move(queue_.front());
                queue_.pop_front();
            }
            job();
        }
    }

    std::vector<std::thread> workers_;
    std::deque<std::function<void()>> queue_;
    std::mutex lock_;
    std::condition_variable wake_;
    bool closing_ = false;
};

inline std::unique_ptr<ThreadPool> createThreadPool(uint32_t num_threads) {
    return std::unique_ptr<ThreadPool>(new ThreadPool(num_threads == 0 ? 1 : num_threads));
}

}  // namespace thread_pool
