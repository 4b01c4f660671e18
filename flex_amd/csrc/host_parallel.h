// host_parallel.h -- the host side's only threading primitive: run fn(0..nchunks-1) on a few std::threads.
// Callers make their result independent of the thread count (each chunk writes its own slots).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace flex {

inline int host_threads() {
    long t = 0;
    if (const char *e = std::getenv("FLEX_HOST_THREADS")) t = std::strtol(e, nullptr, 10);
    if (t <= 0) t = static_cast<long>(std::thread::hardware_concurrency());
    return static_cast<int>(std::clamp<long>(t, 1, 32));
}

template <typename F>
void parallel_chunks(int64_t nchunks, F &&fn) {
    const int nt = static_cast<int>(std::min<int64_t>(host_threads(), nchunks));
    if (nt <= 1) {
        for (int64_t c = 0; c < nchunks; ++c) fn(c);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> th;
    th.reserve(nt);
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&] {
            for (int64_t c; (c = next.fetch_add(1)) < nchunks;) fn(c);
        });
    for (auto &t : th) t.join();
}

// the same, fn(chunk, worker) with worker in [0, host_threads()): for per-worker scratch
template <typename F>
void parallel_chunks_tid(int64_t nchunks, F &&fn) {
    const int nt = static_cast<int>(std::min<int64_t>(host_threads(), nchunks));
    if (nt <= 1) {
        for (int64_t c = 0; c < nchunks; ++c) fn(c, 0);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> th;
    th.reserve(nt);
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (int64_t c; (c = next.fetch_add(1)) < nchunks;) fn(c, t);
        });
    for (auto &t : th) t.join();
}

// A pool for callers that issue MANY small parallel sections (the clustering runs one per batch of 4096 communities:
// spawning 32 threads each time cost as much as the work).  run(nchunks, fn(chunk, worker)) blocks until every chunk is done;
// the calling thread works too (worker 0).
class WorkerPool {
   public:
    explicit WorkerPool(int nthreads) : n_(std::max(1, nthreads)) {
        for (int t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    WorkerPool(const WorkerPool &) = delete;
    int size() const { return n_; }
    void run(int64_t nchunks, const std::function<void(int64_t, int)> &fn) {
        if (nchunks <= 0) return;
        if (n_ == 1 || nchunks == 1) {
            for (int64_t c = 0; c < nchunks; ++c) fn(c, 0);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            nchunks_ = nchunks;
            next_.store(0);
            busy_ = n_ - 1;
            ++gen_;
        }
        cv_.notify_all();
        for (int64_t c; (c = next_.fetch_add(1)) < nchunks;) fn(c, 0);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [this] { return busy_ == 0; });
        fn_ = nullptr;
    }

   private:
    void loop(int tid) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int64_t, int)> *fn;
            int64_t n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
                n = nchunks_;
            }
            for (int64_t c; (c = next_.fetch_add(1)) < n;) (*fn)(c, tid);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--busy_ == 0) done_cv_.notify_one();
            }
        }
    }
    const int n_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int64_t, int)> *fn_ = nullptr;
    int64_t nchunks_ = 0;
    std::atomic<int64_t> next_{0};
    int busy_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

}  // namespace flex
