// stream_audit.h -- STOCS_DEBUG_STREAMS: a host-side happens-before checker for the two-stream sections of the library.
//
// stocs_find_congruent_all and stocs_make_transforms run part of their work on the context's auxiliary stream.  Every buffer such a
// section touches must be ordered between the streams by an event edge: main -> aux before the first use on the auxiliary stream,
// aux -> main before the next use on the main stream (or before the arena is recycled).  A missing edge does not fail -- it reads
// stale data once in a while.  With STOCS_DEBUG_STREAMS=1 the entry points describe what they enqueue -- use(stream, buffer,
// read / write), record(event, stream), wait(stream, event), host_sync(stream) -- and this checker keeps a vector clock per stream,
// a clock snapshot per event and the last write / last reads of every buffer: a use that is not ordered behind the conflicting
// use of the other stream is reported (the call then returns STOCS_ERR_STATE naming buffer and kernel).  Nothing of this touches
// the device; the launches are the same with and without it (DESIGN.md 3 holds the audited table).
#ifndef STOCS_STREAM_AUDIT_H
#define STOCS_STREAM_AUDIT_H

#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

namespace stocs {

struct StreamAudit {
    enum { NS = 2 };                      // 0: the context's stream, 1: its auxiliary stream
    struct Clock { unsigned long long c[NS]; };
    struct Access { int stream; unsigned long long tick; const char* what; };
    struct BufState { bool has_write; Access write; Access reads[NS]; bool has_read[NS]; const char* name; };
    bool on;
    Clock vc[NS];
    std::map<const void*, Clock> events;
    std::map<const void*, BufState> bufs;
    std::vector<std::string> violations;

    StreamAudit() : on(false) { reset(); }
    void reset() {
        for (int s = 0; s < NS; ++s) for (int k = 0; k < NS; ++k) vc[s].c[k] = 0;
        events.clear(); bufs.clear(); violations.clear();
    }
    // the entry point starts with both streams idle (its own entry synchronisation): everything before is ordered before everything after
    void begin(bool enable) { on = enable; if (on) reset(); }
    bool ordered(int t, const Access& a) const { return a.stream == t || vc[t].c[a.stream] >= a.tick; }
    void complain(const char* kind, const char* name, const Access& earlier, int t, const char* what) {
        char msg[512];
        snprintf(msg, sizeof(msg), "%s on '%s': '%s' (stream %d) is not ordered behind '%s' (stream %d) by any event edge", kind, name, what, t, earlier.what, earlier.stream);
        violations.push_back(msg);
    }
    void use(int t, const void* buf, bool write, const char* name, const char* what) {
        if (!on || !buf) return;
        vc[t].c[t]++;
        BufState& b = bufs[buf];
        if (!b.name) { b.has_write = false; for (int s = 0; s < NS; ++s) b.has_read[s] = false; }
        b.name = name;
        if (b.has_write && !ordered(t, b.write)) complain(write ? "write after write" : "read after write", name, b.write, t, what);
        if (write) {
            for (int s = 0; s < NS; ++s) if (b.has_read[s] && !ordered(t, b.reads[s])) complain("write after read", name, b.reads[s], t, what);
            b.has_write = true; b.write.stream = t; b.write.tick = vc[t].c[t]; b.write.what = what;
            for (int s = 0; s < NS; ++s) b.has_read[s] = false;
        } else {
            b.has_read[t] = true; b.reads[t].stream = t; b.reads[t].tick = vc[t].c[t]; b.reads[t].what = what;
        }
    }
    void record(const void* ev, int s) { if (on) events[ev] = vc[s]; }
    void wait(int t, const void* ev) {
        if (!on) return;
        std::map<const void*, Clock>::const_iterator it = events.find(ev);
        if (it == events.end()) { violations.push_back("wait on an event that was never recorded in this call"); return; }
        for (int k = 0; k < NS; ++k) if (it->second.c[k] > vc[t].c[k]) vc[t].c[k] = it->second.c[k];
    }
    // the host waited for stream s (or for an event recorded on it): whatever is enqueued from now on, on either stream, comes after
    void host_sync(int s) { if (!on) return; for (int t = 0; t < NS; ++t) for (int k = 0; k < NS; ++k) if (vc[s].c[k] > vc[t].c[k]) vc[t].c[k] = vc[s].c[k]; }
    void host_sync_event(const void* ev) {
        if (!on) return;
        std::map<const void*, Clock>::const_iterator it = events.find(ev);
        if (it == events.end()) return;
        for (int t = 0; t < NS; ++t) for (int k = 0; k < NS; ++k) if (it->second.c[k] > vc[t].c[k]) vc[t].c[k] = it->second.c[k];
    }
    // the arena the buffers live in is about to be recycled: every use of every buffer must be complete (ordered before both streams)
    void retire_all(const char* what) {
        if (!on) return;
        for (std::map<const void*, BufState>::iterator it = bufs.begin(); it != bufs.end(); ++it) {
            BufState& b = it->second;
            for (int t = 0; t < NS; ++t) {
                if (b.has_write && !ordered(t, b.write)) complain("recycled while written", b.name, b.write, t, what);
                for (int s = 0; s < NS; ++s) if (b.has_read[s] && !ordered(t, b.reads[s])) complain("recycled while read", b.name, b.reads[s], t, what);
            }
        }
        bufs.clear();
    }
};

}  // namespace stocs
#endif
