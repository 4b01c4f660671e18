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
#include <exception>
#include <thread>
#include <vector>

namespace flex {

// How many threads a parallel section started by THIS thread may use: the caller's own setting for the duration of one
// ABI call (flex_plan_tuning.host_threads, HostThreadsScope), else the process-wide cap (flex_set_host_threads), else the
// core count; at most 32.  Results never depend on it.
inline std::atomic<int> &host_threads_cap() {
    static std::atomic<int> cap{0};
    return cap;
}
inline int &host_threads_here() {
    static thread_local int n = 0;
    return n;
}
inline int host_threads() {
    long t = host_threads_here();
    if (t <= 0) t = host_threads_cap().load(std::memory_order_relaxed);
    if (t <= 0) t = static_cast<long>(std::thread::hardware_concurrency());
    return static_cast<int>(std::clamp<long>(t, 1, 32));
}
struct HostThreadsScope {  // sets the calling thread's count for one ABI call
    int saved;
    explicit HostThreadsScope(int n) : saved(host_threads_here()) {
        if (n > 0) host_threads_here() = n;
    }
    ~HostThreadsScope() { host_threads_here() = saved; }
    HostThreadsScope(const HostThreadsScope &) = delete;
    HostThreadsScope &operator=(const HostThreadsScope &) = delete;
};

// The first exception thrown by any chunk of a parallel section (a std::bad_alloc in a worker would otherwise reach
// std::terminate): the section stops handing out chunks, every thread is joined, and the caller rethrows it.
class FirstError {
   public:
    void capture() noexcept {
        std::lock_guard<std::mutex> lk(m_);
        if (!e_) e_ = std::current_exception();
        set_.store(true, std::memory_order_release);
    }
    bool set() const noexcept { return set_.load(std::memory_order_acquire); }
    void rethrow() {
        if (e_) std::rethrow_exception(e_);
    }

   private:
    std::mutex m_;
    std::exception_ptr e_;
    std::atomic<bool> set_{false};
};

template <typename F>
void parallel_chunks(int64_t nchunks, F &&fn) {
    const int nt = static_cast<int>(std::min<int64_t>(host_threads(), nchunks));
    if (nt <= 1) {
        for (int64_t c = 0; c < nchunks; ++c) fn(c);
        return;
    }
    std::atomic<int64_t> next{0};
    FirstError err;
    auto body = [&] {
        try {
            for (int64_t c; !err.set() && (c = next.fetch_add(1)) < nchunks;) fn(c);
        } catch (...) {
            err.capture();
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt);
    try {
        for (int t = 1; t < nt; ++t) th.emplace_back(body);
    } catch (...) {  // thread creation failed: the ones that started finish the section with the caller
        err.capture();
    }
    body();
    for (auto &t : th) t.join();
    err.rethrow();
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
    FirstError err;
    auto body = [&](int t) {
        try {
            for (int64_t c; !err.set() && (c = next.fetch_add(1)) < nchunks;) fn(c, t);
        } catch (...) {
            err.capture();
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt);
    try {
        for (int t = 1; t < nt; ++t) th.emplace_back(body, t);
    } catch (...) {
        err.capture();
    }
    body(0);
    for (auto &t : th) t.join();
    err.rethrow();
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
        FirstError err;
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            err_ = &err;
            nchunks_ = nchunks;
            next_.store(0);
            busy_ = n_ - 1;
            ++gen_;
        }
        cv_.notify_all();
        try {
            for (int64_t c; !err.set() && (c = next_.fetch_add(1)) < nchunks;) fn(c, 0);
        } catch (...) {
            err.capture();
        }
        {  // the workers still hold &fn and &err: wait for every one of them, whatever happened above
            std::unique_lock<std::mutex> lk(m_);
            done_cv_.wait(lk, [this] { return busy_ == 0; });
            fn_ = nullptr;
            err_ = nullptr;
        }
        err.rethrow();
    }

   private:
    void loop(int tid) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int64_t, int)> *fn;
            FirstError *err;
            int64_t n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
                err = err_;
                n = nchunks_;
            }
            try {
                for (int64_t c; !err->set() && (c = next_.fetch_add(1)) < n;) (*fn)(c, tid);
            } catch (...) {
                err->capture();
            }
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
    FirstError *err_ = nullptr;
    int64_t nchunks_ = 0;
    std::atomic<int64_t> next_{0};
    int busy_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

}  // namespace flex
