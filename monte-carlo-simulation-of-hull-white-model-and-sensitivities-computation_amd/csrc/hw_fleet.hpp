// hw_fleet.hpp -- the host-side machinery of hw_comm_*: G persistent worker threads (one per device) and the two-phase
// job protocol that keeps a failing rank from leaving the others in a collective.  No HIP, no RCCL in here: the engine
// passes the device work in as callables, so the protocol runs on a CPU under ThreadSanitizer against a mock collective
// (tests/cpp/fleet_probe.cpp).
//
// Workers.  start(G, bind) creates G threads; worker g calls bind(g) once (the engine: hipSetDevice).  dispatch(job)
// hands every worker the same job and waits for all of them; a worker spins for ~100 us after a job before it sleeps on
// a condition variable, so back-to-back calls never pay a futex wake-up.  A job returns 0 or an error code and may leave
// a text; dispatch reports the FIRST failing rank.
//
// Two phases.  Inside a job every worker calls, in this order,
//     rc = <phase 1: everything that can fail on this device alone, no collective>
//     if (!fleet.rendezvous(rc)) { <drain this device>; return rc; }   // somebody failed: NOBODY enters the collective
//     <phase 2: the collective>                                       // every rank got through phase 1
//     on a failure in phase 2:  if (fleet.broken()) return 0;          // a victim of a peer's abort: the culprit reports
//                               if (fleet.mark_broken(rank, text)) <abort the collective for ALL ranks>;  return error;
// rendezvous() is a barrier over the G workers of the current job (every worker runs every job, so they all arrive).
// A failure after the rendezvous cannot be undone rank by rank: the first worker to call mark_broken() gets `true` and must
// release the ranks that may be blocked in the collective (ncclCommAbort); the fleet is then broken() for good.
#ifndef HW_FLEET_HPP
#define HW_FLEET_HPP

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hw {

class Fleet {
public:
    using Job = std::function<int(int rank, std::string* error_text)>;

    Fleet() = default;
    Fleet(const Fleet&) = delete;
    Fleet& operator=(const Fleet&) = delete;
    ~Fleet() { stop(); }

    void start(int n_workers, std::function<void(int)> bind)
    {
        G_ = n_workers;
        rc_.assign((size_t)G_, 0);
        err_.assign((size_t)G_, std::string());
        for (int g = 0; g < G_; g++) workers_.emplace_back([this, g, bind] { if (bind) bind(g); loop(g); });
    }

    void stop()
    {
        if (workers_.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
            generation_.fetch_add(1, std::memory_order_acq_rel);
        }
        cv_job_.notify_all();
        for (std::thread& t : workers_) t.join();
        workers_.clear();
    }

    int size() const { return G_; }

    // run job(g, &text) on every worker and wait for all of them; returns the first failing rank or -1
    int dispatch(const Job& job, int* rc_of_first = nullptr, std::string* text_of_first = nullptr)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &job;
            arrived_.store(0, std::memory_order_relaxed);           // (every worker of the previous job has returned)
            phase1_failed_.store(0, std::memory_order_relaxed);
            pending_.store(G_, std::memory_order_release);
            generation_.fetch_add(1, std::memory_order_acq_rel);
        }
        cv_job_.notify_all();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
            job_ = nullptr;
        }
        for (int g = 0; g < G_; g++)
            if (rc_[(size_t)g] != 0) {
                if (rc_of_first) *rc_of_first = rc_[(size_t)g];
                if (text_of_first) *text_of_first = err_[(size_t)g];
                return g;
            }
        return -1;
    }

    // barrier between the two phases of the current job; rc1 = this worker's phase-1 result.  true: everybody succeeded.
    bool rendezvous(int rc1)
    {
        if (rc1 != 0) phase1_failed_.fetch_add(1, std::memory_order_release);
        arrived_.fetch_add(1, std::memory_order_acq_rel);
        for (unsigned spins = 0; arrived_.load(std::memory_order_acquire) < G_; spins++) {
            cpu_relax();
            if ((spins & 1023) == 1023) std::this_thread::yield();
        }
        return phase1_failed_.load(std::memory_order_acquire) == 0;
    }

    bool mark_broken() { return !broken_.exchange(true, std::memory_order_acq_rel); }   // true for the first caller only
    // The same, remembering WHO broke the fleet: the first caller's rank and text are kept (culprit()), so that the error that
    // reaches the user is the culprit's and not that of a lower-ranked victim whose own calls failed because of the abort.
    bool mark_broken(int rank, const std::string& text)
    {
        std::lock_guard<std::mutex> lk(culprit_mu_);
        if (broken_.load(std::memory_order_acquire)) return false;
        culprit_rank_ = rank;
        culprit_text_ = text;
        broken_.store(true, std::memory_order_release);
        return true;
    }
    bool broken() const { return broken_.load(std::memory_order_acquire); }
    int culprit(std::string* text)
    {
        std::lock_guard<std::mutex> lk(culprit_mu_);
        if (text) *text = culprit_text_;
        return culprit_rank_;
    }

private:
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }

    void loop(int g)
    {
        uint64_t seen = 0;
        for (;;) {
            // hybrid wait: spin ~100 us (back-to-back calls never pay a futex wake-up), then sleep
            bool got = false;
            const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(100);
            while (std::chrono::steady_clock::now() < spin_until) {
                if (generation_.load(std::memory_order_acquire) != seen) { got = true; break; }
                cpu_relax();
            }
            const Job* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (!got) cv_job_.wait(lk, [&] { return quit_ || generation_.load(std::memory_order_acquire) != seen; });
                if (quit_) return;
                seen = generation_.load(std::memory_order_acquire);
                job = job_;                                          // (job_ and quit_ are written under the mutex)
            }
            std::string text;
            const int r = (*job)(g, &text);
            rc_[(size_t)g] = r;
            err_[(size_t)g] = r ? text : std::string();
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                cv_done_.notify_one();
            }
        }
    }

    int G_ = 0;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_job_, cv_done_;
    std::atomic<uint64_t> generation_{0};
    std::atomic<int> pending_{0}, arrived_{0}, phase1_failed_{0};
    std::atomic<bool> broken_{false};
    std::mutex culprit_mu_;
    int culprit_rank_ = -1;
    std::string culprit_text_;
    bool quit_ = false;
    const Job* job_ = nullptr;
    std::vector<int> rc_;
    std::vector<std::string> err_;
};

}  // namespace hw
#endif
