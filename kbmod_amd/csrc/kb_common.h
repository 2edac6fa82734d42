// Internal helpers shared by the translation units of libkbmod_hip.so.
#ifndef KB_COMMON_H_
#define KB_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "kbmod_hip.h"

namespace kb {

// Arrays the library built itself (kb_build_psi_phi_*: it allocated them and is their only writer) are known to be
// unchanged between searches unless a library call wrote into them, so a search may keep the padded copy it made of one
// (search_kernels.hip) without the caller vouching for it (flag 256).  Every such array has a generation that is renewed by
// whatever writes it; a padded copy stands while the generation it was made from is still the array's.
// note_array_built: registers [p, p + bytes) with a fresh generation and drops whatever was remembered about blocks it
// overlaps; note_array_written: something stored into the range (library calls that write say so themselves, a caller that
// writes by its own means says so through kb_note_array_written); note_array_gone: the block was freed.
void note_array_built(const void* p, uint64_t bytes);
void note_array_written(const void* p);
void note_array_gone(const void* p);
uint64_t array_generation(const void* p);  // 0: not an array the library built
bool array_is_library_owned(const void* p);

// Thread-local message behind kb_last_error().
void set_error(const std::string& msg);
int fail(const std::string& msg);  // set_error + return 1

#define KB_HIP_TRY(expr)                                                                          \
    do {                                                                                          \
        hipError_t kb_err_ = (expr);                                                              \
        if (kb_err_ != hipSuccess) {                                                              \
            return kb::fail(std::string(#expr) + " failed: " + hipGetErrorString(kb_err_));       \
        }                                                                                         \
    } while (0)

constexpr int WAVE = 64;  // gfx950 wavefront

// Every entry point that launches work starts with this: it fails loudly without a device, makes sure the
// runtime is initialised in this process even when the call is the library's first, and drops a stale error
// another library may have left on the calling thread (the launch checks below read the thread's last error).
#define KB_REQUIRE_DEVICE(what)                                                              \
    do {                                                                                     \
        if (kb_device_count() == 0) return kb::fail(std::string("GPU is not available for ") + (what)); \
        (void)hipGetLastError();                                                             \
    } while (0)

// RAII pair of HIP events recorded on the launch stream (bench/roofline timing).
struct EventTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    hipStream_t stream;
    bool active;
    EventTimer(hipStream_t s, bool enable) : stream(s), active(enable) {
        if (active) {
            if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) active = false;
        }
    }
    void begin() {
        if (active) (void)hipEventRecord(start, stream);
    }
    float end() {  // synchronises on the stop event
        mark_end();
        return elapsed();
    }
    // The two halves of end(): record the stop event now, read the time later (after work that follows has been
    // enqueued), so that the host does not stand still in the middle of a sequence of launches.
    void mark_end() {
        if (active) (void)hipEventRecord(stop, stream);
    }
    float elapsed() {  // synchronises on the stop event
        if (!active) return 0.0f;
        float ms = 0.0f;
        (void)hipEventSynchronize(stop);
        (void)hipEventElapsedTime(&ms, start, stop);
        return ms;
    }
    ~EventTimer() {
        if (start) (void)hipEventDestroy(start);
        if (stop) (void)hipEventDestroy(stop);
    }
};

}  // namespace kb
#endif
