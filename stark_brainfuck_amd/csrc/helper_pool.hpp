// helper_pool.hpp -- a few parked host threads for work that is independent of the GPU round trip in progress (host only).
//
// One user so far: the Fiat-Shamir look-ahead of Fri.commit (refpickle.hpp, Transcript::Lookahead).  Every challenge of
// /root/reference/code/fri.py:120 hashes the WHOLE proof stream again (ip.py:21-25: shake_256(pickle.dumps(objects))), and the
// pickle's frame header carries the total length, so no two challenges share a hashed prefix: with tens of KB in front of the FRI
// roots that is ~40 us of SHAKE256 per round on the proving thread, twice the GPU time of a late round.  The prefixes of ALL
// rounds are known when the commit phase starts, so they are absorbed here, side by side, while the first rounds run.
//
// The pool is created on first use and never destroyed (threads parked on a condition variable cost nothing); a forked child
// gets a fresh one (pthread_atfork handlers keep the pool's guard consistent across the fork).  BFS_HELPER_THREADS (default 8) sets their number, 0 switches it off (callers fall back to doing the work themselves).
#pragma once
#include <pthread.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace bfs {

class HelperPool {
   public:
    // nullptr when helpers are switched off or cannot be started
    static HelperPool* get() {
        State& st = state();
        std::lock_guard<std::mutex> lock(st.guard);
        if (!st.atfork_registered) {
            // fork() while another thread holds `guard` would leave it locked for ever in the child: take it across the fork, and
            // let the child start from "no pool" (it has the object but none of its threads; the old one is leaked)
            pthread_atfork([] { state().guard.lock(); }, [] { state().guard.unlock(); },
                           [] { State& c = state(); c.pool = nullptr; c.owner = 0; c.guard.unlock(); });
            st.atfork_registered = true;
        }
        const pid_t me = getpid();
        if (st.pool == nullptr || st.owner != me) {
            int want = 8;
            if (const char* e = getenv("BFS_HELPER_THREADS")) want = atoi(e);
            const int cores = (int)std::thread::hardware_concurrency();
            if (cores > 0 && want > cores - 1) want = cores - 1;
            st.pool = nullptr;
            st.owner = me;
            if (want > 0) {
                HelperPool* p = new HelperPool();
                try {
                    for (int i = 0; i < want; ++i) p->threads_.emplace_back([p] { p->run(); });
                } catch (...) {
                }
                if (!p->threads_.empty()) st.pool = p;
            }
        }
        return st.pool;
    }
    size_t size() const { return threads_.size(); }
    // Jobs are OFFERS: a job must be written so that whoever needs its result can also do the work itself when no helper has got to
    // it yet (refpickle.hpp, Lookahead::Job::claim) -- nothing ever waits for a helper that has not started.
    void submit(std::vector<std::function<void()>> jobs) {       // in order; one wake-up for all of them
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (auto& j : jobs) jobs_.push_back(std::move(j));
        }
        cv_.notify_all();
    }

   private:
    struct State {
        std::mutex guard;
        HelperPool* pool = nullptr;
        pid_t owner = 0;
        bool atfork_registered = false;
    };
    static State& state() {
        static State* s = new State();       // never destroyed: the atfork handlers may run during process teardown
        return *s;
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> jobs_;
    std::vector<std::thread> threads_;
    void run() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [this] { return !jobs_.empty(); });
                job = std::move(jobs_.front());
                jobs_.pop_front();
            }
            job();
        }
    }
};

}  // namespace bfs
