// Sanitizer harness for the threaded host code of libscvote (SURVEY.md section 5 "race detection / sanitizers"; VERDICT r5 missing #6).
//
// The only threads the library owns are the copy workers of the HOST-mode ingestion pipeline (o1.py:50-68 keeps the samples in host
// memory; csrc/scvote.hip HostPipe stages them through pinned bounce slots).  Their code is csrc/scvote_hostpool.h -- plain C++, no
// HIP -- so gcc can build it under -fsanitize=thread and -fsanitize=address,undefined and this driver can use it the way
// host_pipelined() does: per chunk a fresh vector of memcpy pieces over two alternating bounce slots, the calling thread working
// too, pools that got fewer threads than they asked for (or none), pools shut down idle / right after a run / never started.
// tests/test_host_sanitizers.py compiles and runs it (CPU only; GPU sanitizers are not available on the MI355X pool).
#include "../o1_inference_scaling_laws_amd/csrc/scvote_hostpool.h"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

static int check(bool ok, const char* what) {
    if (!ok) { fprintf(stderr, "FAIL: %s\n", what); return 1; }
    return 0;
}

// one HOST-mode call: `chunks` chunks, each copied from the caller's buffer into the slot idx % 2 by the pool, then "consumed"
static int pipelined_call(scv::CopyPool& pool, int parts, size_t chunk_bytes, int chunks, size_t min_piece) {
    std::vector<unsigned char> src((size_t)chunks * chunk_bytes), slot[2];
    slot[0].resize(chunk_bytes); slot[1].resize(chunk_bytes);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)(i * 131u + (i >> 8));
    uint64_t want = 0, got = 0;
    for (unsigned char c : src) want += c;
    for (int idx = 0; idx < chunks; ++idx) {
        std::vector<std::function<void()>> pieces;
        unsigned char* bb = slot[idx % 2].data();
        // two streams per chunk (votes and tokens in the real pipeline): two runs of pieces in the same job list
        const size_t half = chunk_bytes / 2;
        scv::add_copy_pieces(pieces, bb, src.data() + (size_t)idx * chunk_bytes, half, parts, min_piece);
        scv::add_copy_pieces(pieces, bb + half, src.data() + (size_t)idx * chunk_bytes + half, chunk_bytes - half, parts, min_piece);
        pool.run(std::move(pieces));
        for (size_t i = 0; i < chunk_bytes; ++i) got += bb[i];           // the "DMA" reads the slot after run() returned
    }
    return check(got == want, "bytes copied by the pool differ from the source");
}

int main() {
    int bad = 0;
    // 1. pools of 0 .. 7 workers, many short calls (the job list is rebuilt per chunk; workers park and wake between calls)
    for (int threads = 0; threads <= 7; ++threads) {
        scv::CopyPool pool;
        pool.start(threads);
        bad += check((int)pool.workers.size() == threads && pool.start_failures == 0, "start() did not give the threads asked for");
        for (int call = 0; call < 40; ++call) {
            const size_t chunk = 4096 + (rnd() % 60000);
            bad += pipelined_call(pool, threads + 1, chunk, 1 + (int)(rnd() % 5), 256 + (rnd() % 4096));
        }
        // a run with a single piece and a run with many more pieces than threads
        std::atomic<int> hits{0};
        std::vector<std::function<void()>> one; one.emplace_back([&] { hits++; });
        pool.run(std::move(one));
        std::vector<std::function<void()>> many;
        for (int i = 0; i < 500; ++i) many.emplace_back([&] { hits++; });
        pool.run(std::move(many));
        pool.run({});                                                     // empty: returns at once
        bad += check(hits.load() == 501, "not every piece ran exactly once");
        pool.join_all();
        bad += check(pool.workers.empty(), "join_all() left threads behind");
    }
    // 2. a container at its thread limit: the first creation fails -> the calling thread does everything, results unchanged
    {
        scv::CopyPool pool;
        pool.start(6, /*fail_for_test=*/true);
        bad += check(pool.workers.empty() && pool.start_failures == 1, "a failed start must leave no workers and count the failure");
        bad += pipelined_call(pool, 6, 100000, 3, 1024);
        pool.start(3);                                                   // a later call may get threads after all
        bad += check(pool.workers.size() == 3, "start() after a failure");
        bad += pipelined_call(pool, 6, 100000, 3, 1024);
        pool.join_all();
    }
    // 3. grow an existing pool between calls (option "copy_threads" raised), shut down right after a run, shut down never started
    {
        scv::CopyPool pool;
        pool.start(2);
        bad += pipelined_call(pool, 3, 50000, 2, 512);
        pool.start(5);
        bad += check(pool.workers.size() == 5, "start() must only add the missing threads");
        bad += pipelined_call(pool, 6, 50000, 4, 512);
        pool.join_all();
        scv::CopyPool idle;
        idle.join_all();
    }
    // 4. several pools at once (one per scv_ctx; distinct contexts may be used from distinct threads: include/scvote.h)
    {
        std::vector<std::thread> callers;
        std::atomic<int> failures{0};
        for (int t = 0; t < 4; ++t)
            callers.emplace_back([&failures, t] {
                scv::CopyPool pool;
                pool.start(1 + t % 3);
                for (int call = 0; call < 10; ++call) {
                    std::vector<unsigned char> src(30000 + 1000 * t, (unsigned char)(t + call)), dst(src.size());
                    std::vector<std::function<void()>> pieces;
                    scv::add_copy_pieces(pieces, dst.data(), src.data(), src.size(), 4, 1000);
                    pool.run(std::move(pieces));
                    if (dst != src) failures++;
                }
                pool.join_all();
            });
        for (auto& c : callers) c.join();
        bad += check(failures.load() == 0, "pools of distinct contexts disturbed each other");
    }
    if (bad) return 1;
    printf("hostpool ok\n");
    return 0;
}
