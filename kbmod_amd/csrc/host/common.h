// Host-side PODs, logging and timers of the kbmod_amd.search module.
//
// Mirrors (same names, fields, defaults and error behaviour) the reference's
//   common.h:24-161      constants, Trajectory, SearchParameters
//   logging.h:30-240     Logging registry / Logger
//   debug_timer.{h,cpp}  DebugTimer
// (paths relative to /root/reference/src/kbmod/search/).  The structs are
// layout-identical to the C-ABI types of include/kbmod_hip.h so that host
// vectors can be handed to libkbmod_hip.so without conversion.
#ifndef KBH_COMMON_H_
#define KBH_COMMON_H_

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "kbmod_hip.h"

namespace search {

#ifdef _OPENMP
constexpr bool HAVE_OMP = true;
#else
constexpr bool HAVE_OMP = false;
#endif
constexpr bool HAVE_HIP_LIB = true;  // exported as HAS_CUDA: callers gate the device path on it (run_search.py:459)

constexpr unsigned int MAX_NUM_IMAGES = KB_MAX_NUM_IMAGES;  // common.h:31 (200 there)
constexpr float NO_DATA = NAN;                               // common.h:35

enum StampType { STAMP_SUM = 0, STAMP_MEAN, STAMP_MEDIAN, STAMP_VAR_WEIGHTED };  // common.h:37

inline bool pixel_value_valid(float value) { return std::isfinite(value); }  // common.h:41

inline void assert_sizes_equal(size_t actual, size_t expected, std::string name) {  // common.h:44-49
    if (actual != expected) {
        throw std::runtime_error("Size mismatch error [" + name + "]. Expected " + std::to_string(expected) +
                                 ". Found " + std::to_string(actual));
    }
}

// Turns a non-zero C-ABI status into the reference's exception type.
inline void check_status(int status) {
    if (status != 0) throw std::runtime_error(kb_last_error());
}

// common.h:55-115
struct Trajectory {
    float vx = 0.0f;
    float vy = 0.0f;
    float lh = 0.0f;
    float flux = 0.0f;
    int x = 0;
    int y = 0;
    int obs_count = 0;

    // common.h:71-76: double expression, returned as float.
    inline float get_x_pos(double time, bool centered = true) const {
        return centered ? (x + time * vx + 0.5f) : (x + time * vx);
    }
    inline float get_y_pos(double time, bool centered = true) const {
        return centered ? (y + time * vy + 0.5f) : (y + time * vy);
    }
    inline int get_x_index(double time) const { return (int)floor(get_x_pos(time, true)); }
    inline int get_y_index(double time) const { return (int)floor(get_y_pos(time, true)); }

    void clear() { *this = Trajectory(); }
    const std::string to_string() const {
        return "lh: " + std::to_string(lh) + " flux: " + std::to_string(flux) + " x: " + std::to_string(x) +
               " y: " + std::to_string(y) + " vx: " + std::to_string(vx) + " vy: " + std::to_string(vy) +
               " obs_count: " + std::to_string(obs_count);
    }
    // every float field finite and a non-negative count (what TrajectoryList::assert_valid checks per entry)
    bool is_valid() const {
        for (float f : {vx, vy, lh, flux})
            if (!std::isfinite(f)) return false;
        return obs_count >= 0;
    }
    // the Python constructor's argument order (x, y, vx, vy, flux, lh, obs_count) differs from the field order
    static Trajectory make_trajectory(int x, int y, float vx, float vy, float flux, float lh, int obs_count) {
        Trajectory t;
        t.vx = vx, t.vy = vy, t.lh = lh, t.flux = flux;
        t.x = x, t.y = y, t.obs_count = obs_count;
        return t;
    }
};
static_assert(sizeof(Trajectory) == sizeof(kb_trajectory) && sizeof(Trajectory) == 28, "Trajectory must be 28 bytes");
static_assert(offsetof(Trajectory, lh) == offsetof(kb_trajectory, lh) &&
                      offsetof(Trajectory, x) == offsetof(kb_trajectory, x) &&
                      offsetof(Trajectory, obs_count) == offsetof(kb_trajectory, obs_count),
              "Trajectory layout must match the C ABI");

// common.h:119-161.  Layout-identical to kb_search_params (passed by value to the device library).
struct SearchParameters : public kb_search_params {
    SearchParameters() {
        min_observations = 0;
        min_lh = 0.0f;
        do_sigmag_filter = 0;
        sgl_L = 0.25f;
        sgl_H = 0.75f;
        sigmag_coeff = -1.0f;
        encode_num_bytes = -1;
        x_start_min = x_start_max = y_start_min = y_start_max = 0;
        results_per_pixel = 8;
        total_results = 0;
    }
    const std::string to_string() const {
        std::string output = ("Filtering Settings:\n  min_observations: " + std::to_string(min_observations) +
                              "\n  min_lh: " + std::to_string(min_lh));
        if (do_sigmag_filter) {
            output += ("\n  SigmaG: [" + std::to_string(sgl_L) + ", " + std::to_string(sgl_H) +
                       "] coeff=" + std::to_string(sigmag_coeff));
        } else {
            output += "\n  SigmaG: OFF";
        }
        output += "\nResults per pixel: " + std::to_string(results_per_pixel);
        output += "\nencode_num_bytes: " + std::to_string(encode_num_bytes);
        output += ("\nBounds X=[" + std::to_string(x_start_min) + ", " + std::to_string(x_start_max) + "] Y=[" +
                   std::to_string(y_start_min) + ", " + std::to_string(y_start_max) + "]");
        return output;
    }
};

// Row-major float32 image; the reference's `Image` (image_utils_cpp.h:12) without Eigen.
struct Image {
    int64_t rows = 0, cols = 0;
    std::vector<float> data;
    Image() {}
    Image(int64_t r, int64_t c) : rows(r), cols(c), data((size_t)(r * c), 0.0f) {}
    inline float& operator()(int64_t r, int64_t c) { return data[(size_t)(r * cols + c)]; }
    inline const float& operator()(int64_t r, int64_t c) const { return data[(size_t)(r * cols + c)]; }
};

}  // namespace search

namespace logging {

// logging.h:22-36
enum LogLevel { DEBUG = 10, INFO = 20, WARNING = 30, ERROR = 40, CRITICAL = 50 };
typedef std::unordered_map<std::string, std::string> sdict;

inline LogLevel level_from_string(const std::string& s) {
    if (s == "DEBUG") return DEBUG;
    if (s == "INFO") return INFO;
    if (s == "ERROR") return ERROR;
    if (s == "CRITICAL") return CRITICAL;
    return WARNING;
}

// logging.h:38-106.  `sink`, when set, forwards to a Python logger (logging.h:135-146).
class Logger {
public:
    std::string name;
    LogLevel level_threshold = WARNING;
    std::function<void(const std::string&, const std::string&)> sink;

    explicit Logger(const std::string& logger_name) : name(logger_name) {}
    void log(const std::string& level, const std::string& msg) {
        if (sink) {
            sink(level, msg);
        } else if (level_threshold <= level_from_string(level)) {
            std::cout << "[" << level << " " << name << "] " << msg << std::endl;
        }
    }
    void debug(const std::string& msg) { log("DEBUG", msg); }
    void info(const std::string& msg) { log("INFO", msg); }
    void warning(const std::string& msg) { log("WARNING", msg); }
    void error(const std::string& msg) { log("ERROR", msg); }
    void critical(const std::string& msg) { log("CRITICAL", msg); }
};

// logging.h:165-220
class Logging {
public:
    static Logging* logging() {
        static Logging instance;
        return &instance;
    }
    void setConfig(sdict config) { default_config = config; }
    sdict getConfig() { return default_config; }
    Logger* get(const std::string& name) {
        auto it = registry.find(name);
        if (it == registry.end()) {
            auto lg = std::make_unique<Logger>(name);
            auto lv = default_config.find("level");
            if (lv != default_config.end()) lg->level_threshold = level_from_string(lv->second);
            it = registry.emplace(name, std::move(lg)).first;
        }
        return it->second.get();
    }

private:
    Logging() : default_config{{"level", "WARNING"}} {}
    sdict default_config;
    std::unordered_map<std::string, std::unique_ptr<Logger>> registry;
};

inline Logger* getLogger(const std::string& name) { return Logging::logging()->get(name); }

}  // namespace logging

namespace search {

// debug_timer.cpp:13-54
class DebugTimer {
public:
    DebugTimer(std::string message, std::string name) : message_(message), logger_(logging::getLogger(name)) {
        start();
    }
    DebugTimer(std::string message, logging::Logger* logger) : message_(message), logger_(logger) { start(); }
    explicit DebugTimer(std::string message) : message_(message) {
        std::string m = message;
        for (char& ch : m)
            if (ch == ' ') ch = '.';
        logger_ = logging::getLogger("DebugTimer." + m);
        start();
    }
    void start() {
        running_ = true;
        t_start_ = std::chrono::system_clock::now();
        logger_->debug("Starting " + message_ + " timer.");
    }
    void stop() {
        t_end_ = std::chrono::system_clock::now();
        running_ = false;
        auto d = std::chrono::duration_cast<std::chrono::milliseconds>(t_end_ - t_start_);
        logger_->debug("Finished " + message_ + " in " + std::to_string(d.count() / 1000.0) + " seconds.");
    }
    double read() {
        std::chrono::milliseconds d;
        if (running_) {
            d = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - t_start_);
        } else {
            d = std::chrono::duration_cast<std::chrono::milliseconds>(t_end_ - t_start_);
        }
        double result = d.count() / 1000.0;
        logger_->debug("Step " + message_ + " is at " + std::to_string(result) + " seconds.");
        return result;
    }

private:
    std::chrono::time_point<std::chrono::system_clock> t_start_, t_end_;
    bool running_ = false;
    std::string message_;
    logging::Logger* logger_;
};

}  // namespace search
#endif
