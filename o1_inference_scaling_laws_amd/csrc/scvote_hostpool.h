// scvote_hostpool.h -- the threads of the HOST-mode ingestion pipeline, and nothing else.
//
// No HIP in this header: the worker pool that copies pageable caller memory into the pinned bounce slots is plain C++17, so that
// the one piece of the library with threads, a mutex and two condition variables can be compiled by gcc under
// -fsanitize=thread / address,undefined and hammered on the CPU (tests/hostpool_sanitize.cpp, run by the `not gpu` tests; GPU
// sanitizers are not available on the MI355X pool).  scvote.hip's HostPipe owns one CopyPool next to its HIP streams / events / slots.
#ifndef SCVOTE_HOSTPOOL_H
#define SCVOTE_HOSTPOOL_H

#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace scv {

struct CopyPool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::function<void()>> jobs;     // guarded by mu (the vector; a job itself runs unlocked)
    size_t next_job = 0, jobs_done = 0;          // guarded by mu
    bool stop = false;                           // guarded by mu
    int start_failures = 0;                      // calling thread only

    void worker_loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || next_job < jobs.size(); });
            if (stop) return;
            const size_t j = next_job++;
            // the job is MOVED out under the lock: run() clears `jobs` only after jobs_done == size, but taking the callable here
            // means no thread ever touches the vector's storage without the mutex
            std::function<void()> job = std::move(jobs[j]);
            lk.unlock();
            job();
            lk.lock();
            if (++jobs_done == jobs.size()) cv_done.notify_all();
        }
    }
    // Fewer threads than asked for is fine (run() makes the calling thread a worker too): a container at its thread limit raises
    // std::system_error from std::thread's constructor; the pool then runs with the workers it already has, possibly none.
    void start(int nthreads, bool fail_for_test = false) noexcept {
        for (int i = (int)workers.size(); i < nthreads; ++i) {
            try {
                if (fail_for_test) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "test hook");
                workers.emplace_back([this] { worker_loop(); });
            } catch (const std::exception&) {
                ++start_failures;
                break;
            }
        }
    }
    // run the pieces on the workers (the caller takes pieces too) and return when all are done
    void run(std::vector<std::function<void()>>&& pieces) {
        if (pieces.empty()) return;
        if (workers.empty()) { for (auto& f : pieces) f(); return; }
        std::unique_lock<std::mutex> lk(mu);
        jobs = std::move(pieces);
        next_job = 0; jobs_done = 0;
        cv_work.notify_all();
        while (next_job < jobs.size()) {                 // the calling thread is a worker too
            const size_t j = next_job++;
            std::function<void()> job = std::move(jobs[j]);
            lk.unlock();
            job();
            lk.lock();
            ++jobs_done;
        }
        cv_done.wait(lk, [&] { return jobs_done == jobs.size(); });
        jobs.clear();
        next_job = 0; jobs_done = 0;
    }
    void join_all() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
        workers.clear();
    }
};

// memcpy split into pieces for the pool (pieces of >= 1 MiB, 64-byte aligned cuts)
inline void add_copy_pieces(std::vector<std::function<void()>>& pieces, void* dst, const void* src, size_t bytes, int parts,
                            size_t min_piece = (size_t)1 << 20) {
    if (!bytes) return;
    if (parts < 1) parts = 1;
    size_t piece = (bytes + parts - 1) / parts;
    if (piece < min_piece) piece = min_piece;
    piece = (piece + 63) & ~(size_t)63;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = bytes - off < piece ? bytes - off : piece;
        char* d = static_cast<char*>(dst) + off;
        const char* c = static_cast<const char*>(src) + off;
        pieces.emplace_back([d, c, n] { memcpy(d, c, n); });
    }
}

}  // namespace scv

#endif  // SCVOTE_HOSTPOOL_H
