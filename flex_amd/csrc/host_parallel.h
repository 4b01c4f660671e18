// host_parallel.h -- the host side's only threading primitive: run fn(0..nchunks-1) on a few std::threads.
// Callers make their result independent of the thread count (each chunk writes its own slots).
#pragma once
#include <algorithm>
#include <atomic>
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

}  // namespace flex
