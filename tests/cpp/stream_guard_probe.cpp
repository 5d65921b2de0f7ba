// CPU check of the context's stream guard (csrc/hw_stream_guard.hpp) against a mock runtime (tests/test_host_cpp.py).
//
// The mock models what HIP guarantees and nothing more: work enqueued on a stream runs after everything enqueued on that
// stream before; record(event, stream) captures what precedes that point of the stream; wait(stream, event) makes later
// work of the stream run after what the event captured.  Every stream carries its FRONTIER = the set of the context's
// work items that anything enqueued on it from now on is ordered behind.  Checked for 200 000 random operations over
// own / pipe-owned / caller streams that are created and DESTROYED at random:
//   ORDER     when a call enqueues work item w on stream s, s's frontier holds every earlier work item of the context
//             (checked against the latest 64: older ones precede those on whatever stream they ran);
//   LIFETIME  no record() / wait() ever names a stream that is not alive;
// and that round 2's lazy scheme (record on the remembered handle at the switch) violates LIFETIME on the same schedule.
// Mode 2 repeats mode 0 with record() FAILING on one call in fifty: the guard must drain that stream on the host (sync),
// forget it, and go on -- no later call refused, ORDER intact, nothing pending on a stream it can no longer name.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>
#include "hw_stream_guard.hpp"

// a set of work ids kept as "everything below lo" + the few ids above it (work ids only grow, and a correct guard keeps
// every frontier prefix-complete, so `extra` stays tiny and a record is O(1) instead of a copy of 10^5 ids)
struct Frontier {
    int lo = 0;
    std::set<int> extra;
    bool count(int w) const { return w < lo || extra.count(w) != 0; }
    void compact() { while (!extra.empty() && *extra.begin() <= lo) { if (*extra.begin() == lo) lo++; extra.erase(extra.begin()); } }
    void insert(int w) { if (w >= lo) { extra.insert(w); compact(); } }
    void merge(const Frontier& o)
    {
        if (o.lo > lo) { lo = o.lo; }
        for (int w : o.extra) if (w >= lo) extra.insert(w);
        compact();
    }
};

struct Mock {
    struct S { bool alive = false; Frontier frontier; };
    std::vector<S> streams;
    Frontier event;                            // what the tail event captured at its last record
    long dead_stream_uses = 0, records = 0, waits = 0, syncs = 0, failed_records = 0;
    int fail_one_in = 0;                       // > 0: record() fails on one call in that many
    std::mt19937 frng{777};
    int record(int s)
    {
        records++;
        if (!streams[s].alive) { dead_stream_uses++; return 0; }
        if (fail_one_in > 0 && (int)(frng() % (unsigned)fail_one_in) == 0) { failed_records++; return 7; }
        event = streams[s].frontier;
        return 0;
    }
    // host synchronisation with s: everything enqueued on s so far has COMPLETED, so whatever is enqueued anywhere from now on runs after it
    int sync(int s)
    {
        syncs++;
        if (!streams[s].alive) { dead_stream_uses++; return 0; }
        const Frontier done = streams[s].frontier;
        for (auto& t : streams) if (t.alive) t.frontier.merge(done);
        completed.merge(done);
        return 0;
    }
    Frontier completed;                        // work known to have completed (a stream created later starts behind it)
    int wait(int s) { waits++; if (!streams[s].alive) { dead_stream_uses++; return 0; } streams[s].frontier.merge(event); return 0; }
};

// round 2's guard: remember the handle, record on it when another stream shows up
struct LazyGuard {
    int last = -1;
    void enter(Mock& m, int st) { if (last >= 0 && last != st) { m.record(last); m.wait(st); } last = st; }
};

int main()
{
    std::mt19937 rng(12345);
    long order_violations = 0, lazy_dead = 0, calls = 0, foreign_calls = 0, switches = 0;
    for (int mode = 0; mode < 3; mode++) {               // 0: a plain context (own stream + callers' streams); 1: a pipe's context; 2: mode 0 with failing records
        Mock m, lazy_m;
        hw::StreamGuard<int> g;
        LazyGuard lazy;
        const int own = 0, n_pipe = 8;
        m.streams.resize(1 + n_pipe);
        for (auto& s : m.streams) s.alive = true;
        g.set_own_stream(own);
        if (mode == 1) g.set_all_streams_owned(true);
        if (mode == 2) m.fail_one_in = 50;
        lazy_m.streams = m.streams;
        std::vector<int> callers;                         // alive caller streams (mode 0)
        std::set<int> all_work;
        int next_work = 0, prev_stream = -1;
        for (int op = 0; op < 100000; op++) {
            const int r = (int)(rng() % 100);
            if (mode != 1 && r < 12) {                    // the caller creates a stream
                m.streams.push_back(Mock::S{true, m.completed}); lazy_m.streams.push_back(Mock::S{true, Frontier()});
                callers.push_back((int)m.streams.size() - 1);
                continue;
            }
            if (mode != 1 && r < 24 && !callers.empty()) { // ... or destroys one, whatever the context last ran on it
                const size_t k = rng() % callers.size();
                m.streams[callers[k]].alive = false; lazy_m.streams[callers[k]].alive = false;
                callers.erase(callers.begin() + (long)k);
                continue;
            }
            int st = own;
            if (mode == 1) st = 1 + (int)(rng() % n_pipe);
            else if (!callers.empty() && r < 80) st = callers[rng() % callers.size()];
            // one enqueue: enter, the work itself, leave
            if (g.enter(m, st) != 0) { printf("guard refused a call\n"); return 1; }
            for (int w : all_work) if (!m.streams[st].frontier.count(w)) { order_violations++; break; }
            m.streams[st].frontier.insert(next_work);
            all_work.insert(next_work++);
            if (g.leave(m, st) != 0) { printf("guard refused a seal\n"); return 1; }
            lazy.enter(lazy_m, st);
            calls++; foreign_calls += (mode != 1 && st != own); switches += (prev_stream >= 0 && prev_stream != st); prev_stream = st;
            if (all_work.size() > 64) all_work.erase(all_work.begin());   // the invariant is checked against the latest 64 items
        }
        // what a destructor waits for never names a dead stream either
        int which = -1;
        const auto pend = g.pending(&which);
        if (pend == hw::StreamGuard<int>::Pending::OwnedStream && !m.streams[which].alive) m.dead_stream_uses++;
        if (pend == hw::StreamGuard<int>::Pending::DeviceSync) { printf("the guard ended up needing a device synchronisation\n"); return 1; }
        printf("mode %d: %ld records (%ld failed -> %ld host syncs), %ld waits, %ld uses of a dead stream\n", mode, m.records, m.failed_records, m.syncs, m.waits, m.dead_stream_uses);
        if (mode == 2 && (m.failed_records < 100 || m.syncs != m.failed_records)) { printf("failure injection did not exercise the fallback\n"); return 1; }
        if (m.dead_stream_uses) { printf("LIFETIME violated\n"); return 1; }
        lazy_dead += lazy_m.dead_stream_uses;
        if (mode == 1 && m.records > switches) { printf("a pipe's context recorded more often than it switched streams\n"); return 1; }
    }
    printf("{\"calls\": %ld, \"calls_on_caller_streams\": %ld, \"switches\": %ld, \"order_violations\": %ld, \"lazy_scheme_dead_stream_uses\": %ld}\n",
           calls, foreign_calls, switches, order_violations, lazy_dead);
    return order_violations ? 1 : 0;
}
