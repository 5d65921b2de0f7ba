// hw_stream_guard.hpp -- the stream-order guard of a context's workspaces, as a state machine over abstract
// operations so that its two rules can be checked on a CPU with a mock (tests/cpp/stream_guard_probe.cpp):
//
//   ORDER     every enqueue of a context is ordered, on the device, behind everything the context enqueued before,
//             whatever streams the calls arrive on;
//   LIFETIME  a caller's stream handle is never handed to the runtime after the call that received it has returned
//             (the caller may destroy the stream at any time after that call).
//
// How: the stream of the latest enqueue is remembered BY VALUE only (`tail_`, compared, never used).  A call on a
// caller's stream ends with `record(tail event, stream)` while the stream is certainly alive (leave()); the next call
// on another stream starts with `wait(stream, tail event)` (enter()).  For streams the library owns -- the context's
// own, a pipe's candidates -- the record is deferred to the moment another stream actually shows up, so calls that
// stay on one owned stream pay nothing.  Round 2 recorded lazily on the REMEMBERED handle of a caller's stream: a
// use-after-free once the caller had destroyed it (GPUTEST_r02.json).
//
// Ops must provide   int record(Stream s)        record the tail event on s (everything enqueued on s so far)
//                    int wait(Stream s)          make s wait for the tail event as last recorded
//                    int sync(Stream s)          block the host until everything enqueued on s has completed
// returning 0 on success; any other value is passed through to the caller of enter()/leave().
//
// A record that FAILS on a caller's stream (leave()) must not leave the guard naming a stream it could never seal again
// (every later call on another stream would be refused, and a destructor would free the workspaces under work still in
// flight): the stream is drained on the host instead, while the handle is still valid, and the guard forgets it.
#ifndef HW_STREAM_GUARD_HPP
#define HW_STREAM_GUARD_HPP

namespace hw {

constexpr int kGuardInternalError = -9999;   // an unsealed caller stream at a switch: cannot happen if every enter() has its leave()

template <typename Stream>
class StreamGuard {
public:
    void set_own_stream(Stream s) { own_ = s; have_own_ = true; }
    // every stream handed in belongs to the library's caller-side owner (hw_pipe) and outlives the context
    void set_all_streams_owned(bool v) { all_owned_ = v; }
    bool owned(Stream s) const { return all_owned_ || (have_own_ && s == own_); }

    // start of an enqueue on st
    template <typename Ops>
    int enter(Ops& ops, Stream st)
    {
        if (have_tail_ && !(tail_ == st)) {
            bool drained = false;
            if (!recorded_) {
                if (!owned(tail_)) return kGuardInternalError;
                const int r = ops.record(tail_);             // an owned stream: alive by construction
                if (r) {                                     // the record failed: drain the owned stream on the host instead
                    const int sy = ops.sync(tail_);
                    if (sy) return sy;
                    drained = true;
                } else {
                    recorded_ = true;
                }
            }
            if (!drained) {
                const int w = ops.wait(st);
                if (w) return w;
            }
        }
        recorded_ = false;                                   // new work is about to follow on st: the event no longer covers it
        tail_ = st;
        have_tail_ = true;
        return 0;
    }

    // end of that enqueue (call it whatever the enqueue returned)
    template <typename Ops>
    int leave(Ops& ops, Stream st)
    {
        if (!have_tail_ || !(tail_ == st) || recorded_ || owned(st)) return 0;
        const int r = ops.record(st);                        // a caller's stream: NOW, while it is certainly alive
        if (r) {
            const int sy = ops.sync(st);                     // fall back: drain it on the host while the handle is valid ...
            if (sy) { poisoned_ = true; return sy; }         // (nothing more can be done for this stream: destroy must sync the device)
            have_tail_ = false;                              // ... and forget it: nothing of this context is in flight there any more
            recorded_ = false;
            return 0;
        }
        recorded_ = true;
        return 0;
    }

    // what a destructor / host synchronisation has to wait for besides the context's own stream
    enum class Pending { Nothing, TailEvent, OwnedStream, DeviceSync };
    Pending pending(Stream* which) const
    {
        if (poisoned_) return Pending::DeviceSync;           // a caller's stream could be neither sealed nor drained
        if (!have_tail_ || (have_own_ && tail_ == own_)) return Pending::Nothing;
        if (recorded_) return Pending::TailEvent;
        if (owned(tail_)) { *which = tail_; return Pending::OwnedStream; }
        return Pending::DeviceSync;                          // an unsealed caller stream (cannot happen if every enter() has its leave()): be conservative
    }

private:
    Stream own_{}, tail_{};
    bool have_own_ = false, all_owned_ = false, have_tail_ = false, recorded_ = false, poisoned_ = false;
};

}  // namespace hw
#endif
