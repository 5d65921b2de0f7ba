// CPU check of hw_comm_*'s host machinery (csrc/hw_fleet.hpp) against a mock collective, meant to be built with
// -fsanitize=thread as well (tests/test_sanitizers.py, tests/test_host_cpp.py).
//
// The mock all-reduce behaves like the real one in the only way that matters here: a rank that enters it does not come
// back until ALL ranks of the fleet have entered it -- or until the collective is aborted.  Every job follows the engine's
// protocol (comm_run, csrc/hw_engine.hip): phase 1 (may fail on one rank), rendezvous, phase 2 (the collective; may fail on
// one rank, which then aborts it for everybody).  Checked over thousands of jobs with failures injected at random:
//   * no rank ever enters the collective unless every rank got through phase 1           (nobody waits for a missing rank)
//   * a phase-1 failure is reported with the failing rank and its text, and the NEXT job works       (fleet stays usable)
//   * a phase-2 failure aborts once, releases the ranks inside the collective, and the fleet stays broken
//   * every dispatch returns within a bound (a deadlock would trip the watchdog) and the sums are right
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include "hw_fleet.hpp"

struct MockCollective {
    int G;
    std::atomic<int> entered{0}, generation{0};
    std::atomic<bool> aborted{false};
    std::atomic<long> sum{0};
    std::atomic<long> entered_without_full_phase1{0};
    // returns false if released by an abort
    bool all_reduce(long contribution, long* result)
    {
        const int gen = generation.load(std::memory_order_acquire);
        sum.fetch_add(contribution, std::memory_order_acq_rel);
        if (entered.fetch_add(1, std::memory_order_acq_rel) + 1 == G) {            // last one in completes it
            *result = sum.load(std::memory_order_acquire);
            generation.fetch_add(1, std::memory_order_acq_rel);
            return true;
        }
        while (generation.load(std::memory_order_acquire) == gen) {
            if (aborted.load(std::memory_order_acquire)) return false;
            std::this_thread::yield();
        }
        *result = sum.load(std::memory_order_acquire);
        return true;
    }
    void reset() { entered.store(0); sum.store(0); }
};

int main(int argc, char** argv)
{
    const int n_jobs = argc > 1 ? atoi(argv[1]) : 3000;
    long checks = 0, phase1_failures = 0, wrong = 0, slow = 0;
    std::atomic<int> bound{0};
    for (int G : {1, 2, 3, 8}) {
        // ---- part A: failures before the collective only; the fleet must stay usable throughout
        {
            hw::Fleet fleet;
            fleet.start(G, [&](int) { bound.fetch_add(1); });
            MockCollective coll{G};
            std::mt19937 rng(99 + G);
            for (int j = 0; j < n_jobs; j++) {
                const int fail_rank = (rng() % 4 == 0) ? (int)(rng() % G) : -1;     // a quarter of the jobs lose one rank in phase 1
                const int fail_rank2 = (rng() % 16 == 0) ? (int)(rng() % G) : -1;    // ... sometimes two
                coll.reset();
                std::vector<long> got((size_t)G, -1);
                std::atomic<int> phase1_ok{0};
                const hw::Fleet::Job job = [&](int g, std::string* text) -> int {
                    int rc = 0;
                    if (g == fail_rank || g == fail_rank2) { rc = -2; *text = "phase 1 failed on rank " + std::to_string(g); }
                    else phase1_ok.fetch_add(1, std::memory_order_acq_rel);
                    if (!fleet.rendezvous(rc)) return rc;                            // nobody enters the collective
                    if (phase1_ok.load(std::memory_order_acquire) != G) coll.entered_without_full_phase1.fetch_add(1);
                    long r = 0;
                    if (!coll.all_reduce(g + 1, &r)) { *text = "released by an abort"; return -3; }
                    got[(size_t)g] = r;
                    return 0;
                };
                const auto t0 = std::chrono::steady_clock::now();
                int rc = 0;
                std::string text;
                const int first = fleet.dispatch(job, &rc, &text);
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) slow++;
                const int expect_first = (fail_rank < 0 && fail_rank2 < 0) ? -1
                                         : (fail_rank < 0 ? fail_rank2 : (fail_rank2 < 0 ? fail_rank : std::min(fail_rank, fail_rank2)));
                if (first != expect_first) wrong++;
                if (first >= 0) {
                    phase1_failures++;
                    if (rc != -2 || text != "phase 1 failed on rank " + std::to_string(first)) wrong++;
                    if (coll.entered.load() != 0) wrong++;                            // NOBODY entered
                } else {
                    for (int g = 0; g < G; g++) if (got[(size_t)g] != (long)G * (G + 1) / 2) wrong++;
                }
                checks++;
            }
            if (coll.entered_without_full_phase1.load()) wrong++;
            if (fleet.broken()) wrong++;
        }
        // ---- part B: a failure INSIDE the collective phase: one abort, everybody released, broken for good
        for (int victim = 0; victim < G; victim++) {
            hw::Fleet fleet;
            fleet.start(G, nullptr);
            MockCollective coll{G};
            std::atomic<int> aborts{0};
            const hw::Fleet::Job job = [&](int g, std::string* text) -> int {
                if (!fleet.rendezvous(0)) return -1;
                if (g == victim) {                                                    // fails instead of entering: the others would wait for ever
                    *text = "phase 2 failed on rank " + std::to_string(g);
                    if (fleet.mark_broken(g, *text)) { aborts.fetch_add(1); coll.aborted.store(true, std::memory_order_release); }
                    return -4;
                }
                long r = 0;
                if (!coll.all_reduce(1, &r)) {
                    // released by the failing rank's abort.  The real collective then makes THIS rank's calls fail too (its
                    // all-reduce kernel is aborted): the engine's rule is that a rank which finds the fleet broken reports
                    // nothing of its own -- the culprit's rank and text are the ones the caller gets (fleet.culprit()).
                    if (fleet.broken()) return 0;
                    *text = "collateral error on rank " + std::to_string(g);
                    return -5;
                }
                return 0;
            };
            const auto t0 = std::chrono::steady_clock::now();
            int rc = 0;
            std::string text;
            const int first = fleet.dispatch(job, &rc, &text);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) slow++;
            if (G == 1) { if (first != 0 || rc != -4) wrong++; }
            else if (first != victim || rc != -4 || aborts.load() != 1 || !fleet.broken()) wrong++;
            std::string why;
            if (G > 1 && (fleet.culprit(&why) != victim || why != "phase 2 failed on rank " + std::to_string(victim))) wrong++;
            if (fleet.mark_broken(0, "late")) wrong++;                                // only the first failure is the culprit
            checks++;
        }
        // ---- part C: back-to-back tiny jobs (the spin path) and jobs after pauses (the sleep path)
        {
            hw::Fleet fleet;
            fleet.start(G, nullptr);
            std::atomic<long> total{0};
            const hw::Fleet::Job job = [&](int g, std::string*) -> int { total.fetch_add(g + 1); return fleet.rendezvous(0) ? 0 : -1; };
            for (int j = 0; j < 2000; j++) if (fleet.dispatch(job) != -1) wrong++;
            for (int ms : {1, 5, 20}) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); if (fleet.dispatch(job) != -1) wrong++; }
            if (total.load() != 2003L * G * (G + 1) / 2) wrong++;
            checks += 2003;
        }
    }
    printf("{\"checks\": %ld, \"phase1_failures\": %ld, \"wrong\": %ld, \"slow_dispatches\": %ld, \"workers_bound\": %d}\n",
           checks, phase1_failures, wrong, slow, bound.load());
    return (wrong || slow) ? 1 : 0;
}
