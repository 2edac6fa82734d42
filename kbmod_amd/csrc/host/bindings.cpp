// pybind11 module `kbmod_amd.search`: the Python-visible surface of the
// reference's `kbmod.search` (bindings.cpp:20-41 and the per-file binding blocks:
// common.h:164-217, stack_search.cpp:341-389, psi_phi_array.cpp:417-465,
// trajectory_list.cpp:246-289, cpu_search_algorithms.cpp:128-131,
// image_utils_cpp.cpp:180-194, kernel_helpers.cpp:109-117, debug_timer.cpp:57-69,
// logging.h:223-237), over numpy buffers instead of Eigen.
#include <pybind11/numpy.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "common.h"
#include "image_utils.h"
#include "psi_phi_array.h"
#include "stack_search.h"
#include "device_stack.h"
#include "trajectory_list.h"

namespace py = pybind11;
using namespace search;

// Converting image argument (list elements of StackSearch / fill_psi_phi_array*:
// pybind11's Eigen caster converts dtype there, e.g. float64 PSFs).
using conv_array = py::array_t<float, py::array::c_style | py::array::forcecast>;
// Non-converting image argument (image utils are bound with noconvert).
using strict_array = py::array_t<float, py::array::c_style>;


static Image to_image(const py::array& arr) {
    if (arr.ndim() != 2) throw std::runtime_error("Expected a 2-dimensional array.");
    py::array_t<float, py::array::c_style | py::array::forcecast> a(arr);
    Image img(a.shape(0), a.shape(1));
    std::memcpy(img.data.data(), a.data(), img.data.size() * sizeof(float));
    return img;
}
static std::vector<Image> to_images(const std::vector<conv_array>& arrs) {
    std::vector<Image> out;
    out.reserve(arrs.size());
    for (const auto& a : arrs) out.push_back(to_image(a));
    return out;
}
static py::array_t<float> from_image(const Image& img) {
    py::array_t<float> out({(py::ssize_t)img.rows, (py::ssize_t)img.cols});
    if (!img.data.empty()) std::memcpy(out.mutable_data(), img.data.data(), img.data.size() * sizeof(float));
    return out;
}
static Image strict_image(const py::array& arr, const char* what) {
    // noconvert semantics: a non-float32 or non-2D argument is a TypeError.
    if (!py::isinstance<py::array_t<float>>(arr) || arr.ndim() != 2) {
        throw py::type_error(std::string(what) + " must be a 2-dimensional float32 numpy array");
    }
    return to_image(arr);
}

PYBIND11_MODULE(search, m) {
    m.doc() = "MI355X-native shift-and-stack search (kbmod.search-compatible surface)";
    m.attr("KB_NO_DATA") = py::float_(NO_DATA);
    m.attr("HAS_CUDA") = py::bool_(HAVE_HIP_LIB);  // a HIP build: the device path exists (run_search.py:459)
    m.attr("HAS_HIP") = py::bool_(HAVE_HIP_LIB);
    m.attr("HAS_OMP") = py::bool_(HAVE_OMP);
    m.def("omp_max_threads", []() {
#ifdef _OPENMP
        return omp_get_max_threads();
#else
        return 1;
#endif
    }, "Threads an OpenMP region of the host layer (search_cpu_only, ...) runs with.");
    m.attr("MAX_NUM_IMAGES") = py::int_(MAX_NUM_IMAGES);
    py::enum_<StampType>(m, "StampType")
            .value("STAMP_SUM", StampType::STAMP_SUM)
            .value("STAMP_MEAN", StampType::STAMP_MEAN)
            .value("STAMP_MEDIAN", StampType::STAMP_MEDIAN)
            .value("STAMP_VAR_WEIGHTED", StampType::STAMP_VAR_WEIGHTED)
            .export_values();

    // ---- logging.h:223-237 ----
    py::class_<logging::Logging, std::unique_ptr<logging::Logging, py::nodelete>>(m, "Logging")
            .def(py::init([]() { return std::unique_ptr<logging::Logging, py::nodelete>(logging::Logging::logging()); }))
            .def("setConfig", &logging::Logging::setConfig)
            .def_static("getLogger",
                        [](py::str name) -> py::object {
                            py::object pylogger = py::module_::import("logging").attr("getLogger")(name);
                            logging::Logger* lg = logging::getLogger(std::string(name));
                            lg->sink = [pylogger](const std::string& level, const std::string& msg) {
                                std::string l = level;
                                for (char& ch : l) ch = std::tolower(ch);
                                pylogger.attr(l.c_str())(msg);
                            };
                            return pylogger;
                        })
            .def_static("registerLogger", [](py::object pylogger) -> void {
                logging::Logger* lg = logging::getLogger(pylogger.attr("name").cast<std::string>());
                lg->sink = [pylogger](const std::string& level, const std::string& msg) {
                    std::string l = level;
                    for (char& ch : l) ch = std::tolower(ch);
                    pylogger.attr(l.c_str())(msg);
                };
            });

    // ---- common.h:164-198 ----
    py::class_<Trajectory>(m, "Trajectory")
            .def(py::init(&Trajectory::make_trajectory), py::arg("x") = 0, py::arg("y") = 0, py::arg("vx") = 0.0f,
                 py::arg("vy") = 0.0f, py::arg("flux") = 0.0f, py::arg("lh") = 0.0f, py::arg("obs_count") = 0)
            .def_readwrite("vx", &Trajectory::vx)
            .def_readwrite("vy", &Trajectory::vy)
            .def_readwrite("lh", &Trajectory::lh)
            .def_readwrite("flux", &Trajectory::flux)
            .def_readwrite("x", &Trajectory::x)
            .def_readwrite("y", &Trajectory::y)
            .def_readwrite("obs_count", &Trajectory::obs_count)
            .def("get_x_pos", &Trajectory::get_x_pos, py::arg("time"), py::arg("centered") = true,
                 "Predicted x position at `time` (days since the first epoch): x + vx * time, plus 0.5 when `centered` "
                 "(pixel centres).")
            .def("get_y_pos", &Trajectory::get_y_pos, py::arg("time"), py::arg("centered") = true,
                 "Predicted y position at `time`: y + vy * time, plus 0.5 when `centered`.")
            .def("get_x_index", &Trajectory::get_x_index,
                 "Pixel column the trajectory is in at `time`: floor of the centred x position.")
            .def("get_y_index", &Trajectory::get_y_index,
                 "Pixel row the trajectory is in at `time`: floor of the centred y position.")
            .def("is_valid", &Trajectory::is_valid,
                 "True when vx, vy, lh and flux are finite and obs_count is not negative.")
            .def("clear", &Trajectory::clear)
            .def("__repr__", [](const Trajectory& t) { return "Trajectory(" + t.to_string() + ")"; })
            .def("__str__", &Trajectory::to_string)
            .def(py::pickle(
                    [](const Trajectory& p) {
                        return py::make_tuple(p.vx, p.vy, p.lh, p.flux, p.x, p.y, p.obs_count);
                    },
                    [](py::tuple t) {
                        // The reference demands 8 entries while writing 7 (common.h:187-197),
                        // so its pickles cannot be read back; accept the 7 it writes.
                        if (t.size() != 7) throw std::runtime_error("Invalid state!");
                        Trajectory trj;
                        trj.vx = t[0].cast<float>();
                        trj.vy = t[1].cast<float>();
                        trj.lh = t[2].cast<float>();
                        trj.flux = t[3].cast<float>();
                        trj.x = t[4].cast<int>();
                        trj.y = t[5].cast<int>();
                        trj.obs_count = t[6].cast<int>();
                        return trj;
                    }));

    // ---- common.h:200-217 ----
    py::class_<SearchParameters>(m, "SearchParameters")
            .def(py::init<>())
            .def("__str__", &SearchParameters::to_string)
            .def_property(
                    "min_observations", [](const SearchParameters& p) { return p.min_observations; },
                    [](SearchParameters& p, int v) { p.min_observations = v; })
            .def_property(
                    "min_lh", [](const SearchParameters& p) { return p.min_lh; },
                    [](SearchParameters& p, float v) { p.min_lh = v; })
            .def_property(
                    "do_sigmag_filter", [](const SearchParameters& p) { return p.do_sigmag_filter != 0; },
                    [](SearchParameters& p, bool v) { p.do_sigmag_filter = v ? 1 : 0; })
            .def_property(
                    "sgl_L", [](const SearchParameters& p) { return p.sgl_L; },
                    [](SearchParameters& p, float v) { p.sgl_L = v; })
            .def_property(
                    "sgl_H", [](const SearchParameters& p) { return p.sgl_H; },
                    [](SearchParameters& p, float v) { p.sgl_H = v; })
            .def_property(
                    "sigmag_coeff", [](const SearchParameters& p) { return p.sigmag_coeff; },
                    [](SearchParameters& p, float v) { p.sigmag_coeff = v; })
            .def_property(
                    "encode_num_bytes", [](const SearchParameters& p) { return p.encode_num_bytes; },
                    [](SearchParameters& p, int v) { p.encode_num_bytes = v; })
            .def_property(
                    "x_start_min", [](const SearchParameters& p) { return p.x_start_min; },
                    [](SearchParameters& p, int v) { p.x_start_min = v; })
            .def_property(
                    "x_start_max", [](const SearchParameters& p) { return p.x_start_max; },
                    [](SearchParameters& p, int v) { p.x_start_max = v; })
            .def_property(
                    "y_start_min", [](const SearchParameters& p) { return p.y_start_min; },
                    [](SearchParameters& p, int v) { p.y_start_min = v; })
            .def_property(
                    "y_start_max", [](const SearchParameters& p) { return p.y_start_max; },
                    [](SearchParameters& p, int v) { p.y_start_max = v; })
            .def_property(
                    "results_per_pixel", [](const SearchParameters& p) { return p.results_per_pixel; },
                    [](SearchParameters& p, unsigned int v) { p.results_per_pixel = v; })
            .def_property(
                    "total_results", [](const SearchParameters& p) { return p.total_results; },
                    [](SearchParameters& p, unsigned long long v) { p.total_results = v; });

    // ---- psi_phi_array.cpp:417-465 ----
    py::class_<PsiPhi>(m, "PsiPhi")
            .def(py::init<>())
            .def_readwrite("psi", &PsiPhi::psi)
            .def_readwrite("phi", &PsiPhi::phi);

    py::class_<PsiPhiArray>(m, "PsiPhiArray")
            .def(py::init<>())
            .def_property_readonly("on_gpu", &PsiPhiArray::on_gpu)
            .def_property_readonly("num_bytes", &PsiPhiArray::get_num_bytes)
            .def_property_readonly("num_times", &PsiPhiArray::get_num_times)
            .def_property_readonly("width", &PsiPhiArray::get_width)
            .def_property_readonly("height", &PsiPhiArray::get_height)
            .def_property_readonly("pixels_per_image", &PsiPhiArray::get_pixels_per_image)
            .def_property_readonly("num_entries", &PsiPhiArray::get_num_entries)
            .def_property_readonly("total_array_size", &PsiPhiArray::get_total_array_size)
            .def_property_readonly("block_size", &PsiPhiArray::get_block_size)
            .def_property_readonly("psi_min_val", &PsiPhiArray::get_psi_min_val)
            .def_property_readonly("psi_max_val", &PsiPhiArray::get_psi_max_val)
            .def_property_readonly("psi_scale", &PsiPhiArray::get_psi_scale)
            .def_property_readonly("phi_min_val", &PsiPhiArray::get_phi_min_val)
            .def_property_readonly("phi_max_val", &PsiPhiArray::get_phi_max_val)
            .def_property_readonly("phi_scale", &PsiPhiArray::get_phi_scale)
            .def_property_readonly("cpu_array_allocated", &PsiPhiArray::cpu_array_allocated)
            .def_property_readonly("gpu_array_allocated", &PsiPhiArray::gpu_array_allocated)
            .def_property_readonly("device_resident", &PsiPhiArray::device_resident)
            .def("set_meta_data", &PsiPhiArray::set_meta_data,
                 "Set the encoding width (1, 2 or 4 bytes per value; -1 means 4), the number of epochs and the image "
                 "size, and derive the array sizes from them.  Clears any existing data.")
            .def("set_time_array", &PsiPhiArray::set_time_array,
                 "Copy the epoch times (one per image, zero-shifted days) into the array object; their number must "
                 "equal num_times.")
            .def("move_to_gpu", &PsiPhiArray::move_to_gpu,
                 "Make the data resident on the device (allocates and uploads what is not there yet); raises "
                 "RuntimeError without a GPU or when already there.")
            .def("clear", &PsiPhiArray::clear)
            .def("clear_from_gpu", &PsiPhiArray::clear_from_gpu,
                 "Release the device copy (and keep the host copy).")
            .def("read_psi_phi", &PsiPhiArray::read_psi_phi,
                 "(psi, phi) at epoch `time`, pixel (`row`, `col`), decoded to float; NaN for both when the position "
                 "is outside the image or the value is NO_DATA.")
            .def("read_time", &PsiPhiArray::read_time,
                 "The zero-shifted time of epoch `time_index`; raises for an index out of range.")
            // Raw encoded array as a flat numpy vector (parity tests compare it bit for bit).
            .def("encoded_array", [](PsiPhiArray& a) -> py::array {
                const void* p = a.host_ptr();
                const py::ssize_t n = (py::ssize_t)a.get_num_entries();
                if (a.get_num_bytes() == 1) {
                    py::array_t<uint8_t> out(n);
                    std::memcpy(out.mutable_data(), p, (size_t)n);
                    return out;
                } else if (a.get_num_bytes() == 2) {
                    py::array_t<uint16_t> out(n);
                    std::memcpy(out.mutable_data(), p, (size_t)n * 2);
                    return out;
                }
                py::array_t<float> out(n);
                std::memcpy(out.mutable_data(), p, (size_t)n * 4);
                return out;
            },
                 "The array as stored ([T][H][W][psi, phi] of uint8 / uint16 / float32), copied to the host if it "
                 "lives in HBM.");
    m.def("compute_scale_params_from_image_vect", [](const std::vector<conv_array>& imgs, int num_bytes) {
        return compute_scale_params_from_image_vect(to_images(imgs), num_bytes);
    },
                 "[min, max, scale] for encoding a list of images with `num_bytes` bytes per value: min / max over all "
                 "finite pixels, scale = max(max - min, 1e-6) / (2^(8 num_bytes) - 1).");
    m.def("decode_uint_scalar", &decode_uint_scalar,
                 "Float value of an encoded sample: NaN for code 0, else (code - 1) * scale + min_val.");
    m.def("encode_uint_scalar", &encode_uint_scalar,
                 "Code of a float value: 0 for non-finite input, else the value clamped to [min_val, max_val] shifted "
                 "to start at 1 (fractional; the array stores its truncation).");
    m.def("fill_psi_phi_array", [](PsiPhiArray& result_data, int num_bytes, const std::vector<conv_array>& psi_imgs,
                                   const std::vector<conv_array>& phi_imgs, const std::vector<double> zeroed_times) {
        fill_psi_phi_array(result_data, num_bytes, to_images(psi_imgs), to_images(phi_imgs), zeroed_times);
    },
                 "Fill `result_data` from per-epoch psi and phi images and their zero-shifted times, encoding with "
                 "`num_bytes` bytes per value (scale parameters taken over all images).  The data stays on the host.");
    m.def(
            "fill_psi_phi_array_from_image_arrays",
            [](PsiPhiArray& result_data, int num_bytes, const std::vector<conv_array>& sci,
               const std::vector<conv_array>& var, const std::vector<conv_array>& psfs, std::vector<double> times,
               bool force_cpu) {
                std::vector<Image> s = to_images(sci), v = to_images(var), p = to_images(psfs);
                fill_psi_phi_array_from_image_arrays(result_data, num_bytes, s, v, p, times, force_cpu);
            },
            py::arg("result_data"), py::arg("num_bytes"), py::arg("sci_imgs"), py::arg("var_imgs"),
            py::arg("psf_kernels"), py::arg("zeroed_times"), py::arg("force_cpu") = false,
                 "Build psi and phi from science / variance images and PSF kernels (generate_psi / generate_phi per "
                 "epoch) and fill `result_data` from them.");

    // ---- debug_timer.cpp:57-69 ----
    py::class_<DebugTimer>(m, "DebugTimer")
            .def(py::init<std::string, std::string>())
            .def(py::init<std::string>())
            .def(py::init([](std::string message, py::object logger) {
                std::string name = std::string(py::str(logger.attr("name")));
                return std::unique_ptr<DebugTimer>(new DebugTimer(message, name));
            }))
            .def("start", &DebugTimer::start)
            .def("stop", &DebugTimer::stop)
            .def("read", &DebugTimer::read);

    // ---- trajectory_list.cpp:246-289 ----
    py::class_<TrajectoryList>(m, "TrajectoryList")
            .def(py::init<int>())
            .def(py::init<std::vector<Trajectory>&>())
            .def_property_readonly("on_gpu", &TrajectoryList::on_gpu)
            .def("__len__", &TrajectoryList::get_size)
            .def("resize", &TrajectoryList::resize,
                 "Change the number of entries; new entries are zeroed trajectories.  Host data only.")
            .def("get_size", &TrajectoryList::get_size,
                 "Number of trajectories in the list.")
            .def("get_memory", &TrajectoryList::get_memory,
                 "Bytes the list occupies (28 per trajectory).")
            .def_static("estimate_memory", &TrajectoryList::estimate_memory,
                 "Bytes a list of `num_elements` trajectories would occupy.")
            .def("get_trajectory", &TrajectoryList::get_trajectory, py::return_value_policy::reference_internal,
                 "Reference to entry `index` (changes made through it are seen by the list); raises for an index out "
                 "of range or data on the device.")
            .def("reset_all", &TrajectoryList::reset_all,
                 "Zero every entry.")
            .def("set_trajectory", &TrajectoryList::set_trajectory,
                 "Overwrite entry `index` with a copy of `new_value`.")
            .def("set_trajectories", &TrajectoryList::set_trajectories,
                 "Replace the whole list by copies of the given trajectories.")
            .def("get_list", &TrajectoryList::get_list,
                 "The trajectories as a Python list (copies).")
            .def("get_batch", &TrajectoryList::get_batch,
                 "Copies of `count` entries starting at `start` (fewer at the end of the list).")
            .def("sort_by_likelihood", &TrajectoryList::sort_by_likelihood,
                 "Sort in place by likelihood, highest first.")
            .def("filter_by_likelihood", &TrajectoryList::filter_by_likelihood,
                 "Drop entries whose likelihood is below `min_likelihood`; the survivors end up sorted by likelihood.")
            .def("filter_by_obs_count", &TrajectoryList::filter_by_obs_count,
                 "Drop entries with fewer than `min_obs_count` observations.")
            .def("assert_valid", &TrajectoryList::assert_valid,
                 "Raise RuntimeError at the first entry that is not valid (see Trajectory.is_valid).")
            .def("move_to_cpu", &TrajectoryList::move_to_cpu,
                 "Bring the data back from the device and release the device buffer.")
            .def("move_to_gpu", &TrajectoryList::move_to_gpu,
                 "Make the data resident on the device (allocates and uploads what is not there yet); raises "
                 "RuntimeError without a GPU or when already there.")
            // Bulk transfer as an (N, 7) float64 table [x, y, vx, vy, lh, flux, obs_count].
            .def("to_numpy", [](TrajectoryList& l) {
                const std::vector<Trajectory>& v = l.get_list();
                py::array_t<double> out({(py::ssize_t)v.size(), (py::ssize_t)7});
                auto r = out.mutable_unchecked<2>();
                for (py::ssize_t i = 0; i < (py::ssize_t)v.size(); ++i) {
                    r(i, 0) = v[i].x;
                    r(i, 1) = v[i].y;
                    r(i, 2) = v[i].vx;
                    r(i, 3) = v[i].vy;
                    r(i, 4) = v[i].lh;
                    r(i, 5) = v[i].flux;
                    r(i, 6) = v[i].obs_count;
                }
                return out;
            });
    m.def("extract_all_trajectory_x", &extract_all_trajectory_x,
                 "x of every trajectory of a list, in order.");
    m.def("extract_all_trajectory_y", &extract_all_trajectory_y,
                 "y of every trajectory of a list, in order.");
    m.def("extract_all_trajectory_vx", &extract_all_trajectory_vx,
                 "vx of every trajectory of a list, in order.");
    m.def("extract_all_trajectory_vy", &extract_all_trajectory_vy,
                 "vy of every trajectory of a list, in order.");
    m.def("extract_all_trajectory_lh", &extract_all_trajectory_lh,
                 "Likelihood of every trajectory of a list, in order.");
    m.def("extract_all_trajectory_flux", &extract_all_trajectory_flux,
                 "Flux of every trajectory of a list, in order.");
    m.def("extract_all_trajectory_obs_count", &extract_all_trajectory_obs_count,
                 "Observation count of every trajectory of a list, in order.");

    m.def("merge_topk_host", [](py::array_t<uint8_t, py::array::c_style> raw, int n_lists, uint64_t n_pixels, int K) {
        if ((uint64_t)raw.size() != (uint64_t)n_lists * n_pixels * K * sizeof(Trajectory)) {
            throw std::runtime_error("merge_topk_host: buffer size does not match n_lists * n_pixels * K * 28");
        }
        std::vector<Trajectory> out =
                merge_topk_host(reinterpret_cast<const Trajectory*>(raw.data()), n_lists, n_pixels, K);
        py::array_t<uint8_t> res((py::ssize_t)(out.size() * sizeof(Trajectory)));
        if (!out.empty()) std::memcpy(res.mutable_data(), out.data(), out.size() * sizeof(Trajectory));
        return res;
    });

    m.def("merge_compact_host",
          [](py::array_t<uint8_t, py::array::c_style> raw, int n_lists, int K, int x_min, int x_max, int y_min, int y_max,
             const std::vector<Trajectory>& all_cands) {
              if (x_max <= x_min || y_max <= y_min) throw std::runtime_error("merge_compact_host: invalid search bounds");
              const uint64_t n_pixels = (uint64_t)(x_max - x_min) * (uint64_t)(y_max - y_min);
              if ((uint64_t)raw.size() != (uint64_t)n_lists * n_pixels * K * sizeof(kb_compact_result)) {
                  throw std::runtime_error("merge_compact_host: buffer size does not match n_lists * n_pixels * K * 16");
              }
              std::vector<Trajectory> out =
                      merge_compact_host(reinterpret_cast<const kb_compact_result*>(raw.data()), n_lists, n_pixels, K,
                                         x_max - x_min, x_min, y_min, all_cands.data(), all_cands.size());
              py::array_t<uint8_t> res((py::ssize_t)(out.size() * sizeof(Trajectory)));
              if (!out.empty()) std::memcpy(res.mutable_data(), out.data(), out.size() * sizeof(Trajectory));
              return res;
          });

    m.def("merge_compact_exact_host",
          [](py::array_t<uint8_t, py::array::c_style> raw, int n_lists, int list_len, int K, int x_min, int x_max, int y_min,
             int y_max, const std::vector<Trajectory>& all_cands) {
              if (x_max <= x_min || y_max <= y_min) throw std::runtime_error("merge_compact_exact_host: invalid search bounds");
              const uint64_t n_pixels = (uint64_t)(x_max - x_min) * (uint64_t)(y_max - y_min);
              if ((uint64_t)raw.size() != (uint64_t)n_lists * n_pixels * list_len * sizeof(kb_compact_result)) {
                  throw std::runtime_error("merge_compact_exact_host: buffer size does not match n_lists * n_pixels * list_len * 16");
              }
              std::vector<Trajectory> out = merge_compact_exact_host(
                      reinterpret_cast<const kb_compact_result*>(raw.data()), n_lists, n_pixels, list_len, K, x_max - x_min,
                      x_min, y_min, all_cands.data(), all_cands.size());
              py::array_t<uint8_t> res((py::ssize_t)(out.size() * sizeof(Trajectory)));
              if (!out.empty()) std::memcpy(res.mutable_data(), out.data(), out.size() * sizeof(Trajectory));
              return res;
          },
          "Host twin of kb_merge_compact_exact: per-device lists of `list_len` 16-byte records per pixel, built by stable\n"
          "insertion, merged into the K results per pixel a single device would produce (ties included).");

    m.def("merge_compact_repairable_host",
          [](py::array_t<uint8_t, py::array::c_style> raw, int n_lists, int K, int x_min, int x_max, int y_min, int y_max,
             const std::vector<Trajectory>& all_cands) {
              if (x_max <= x_min || y_max <= y_min) throw std::runtime_error("merge_compact_repairable_host: invalid search bounds");
              const uint64_t n_pixels = (uint64_t)(x_max - x_min) * (uint64_t)(y_max - y_min);
              if (K <= 0 || (uint64_t)raw.size() != (uint64_t)n_lists * n_pixels * (uint64_t)K * sizeof(kb_compact_result)) {
                  throw std::runtime_error("merge_compact_repairable_host: buffer size does not match n_lists * n_pixels * K * 16");
              }
              std::vector<uint32_t> hazards;
              std::vector<Trajectory> out = merge_compact_repairable_host(
                      reinterpret_cast<const kb_compact_result*>(raw.data()), n_lists, n_pixels, K, x_max - x_min, x_min, y_min,
                      all_cands.data(), all_cands.size(), hazards);
              py::array_t<uint8_t> res((py::ssize_t)(out.size() * sizeof(Trajectory)));
              if (!out.empty()) std::memcpy(res.mutable_data(), out.data(), out.size() * sizeof(Trajectory));
              py::array_t<uint32_t> hz((py::ssize_t)hazards.size());
              if (!hazards.empty()) std::memcpy(hz.mutable_data(), hazards.data(), hazards.size() * sizeof(uint32_t));
              return py::make_tuple(res, hz);
          },
          "Host twin of kb_merge_compact_repairable: per-device lists of K 16-byte records per pixel (the reference's insertion\n"
          "over each device's slice) -> (the K results per pixel a single device would produce wherever the lists decide it,\n"
          "the numbers of the pixels where they do not and a repair is due).");

    m.def("sparse_header_bytes", &sparse_header_bytes, "Bytes of the header of a sparse exchange list for n_pixels pixels.");
    m.def("sparsify_compact_host",
          [](py::array_t<uint8_t, py::array::c_style> raw, uint64_t n_pixels, int list_len, float min_lh) {
              if ((uint64_t)raw.size() != n_pixels * (uint64_t)list_len * sizeof(kb_compact_result)) {
                  throw std::runtime_error("sparsify_compact_host: buffer size does not match n_pixels * list_len * 16");
              }
              std::vector<uint8_t> header;
              std::vector<kb_compact_result> packed;
              sparsify_compact_host(reinterpret_cast<const kb_compact_result*>(raw.data()), n_pixels, list_len, min_lh, header,
                                    packed);
              py::array_t<uint8_t> h((py::ssize_t)header.size());
              std::memcpy(h.mutable_data(), header.data(), header.size());
              py::array_t<uint8_t> p((py::ssize_t)(packed.size() * sizeof(kb_compact_result)));
              if (!packed.empty()) std::memcpy(p.mutable_data(), packed.data(), packed.size() * sizeof(kb_compact_result));
              return py::make_tuple(h, p);
          },
          "Host twin of kb_sparsify_compact: dense per-pixel lists of 16-byte records -> (header bytes, packed record bytes);\n"
          "keeps the records with cand >= 0 and not lh < min_lh (the reference's post-filter, stack_search.cpp:266-270).");
    m.def("merge_sparse_exact_host",
          [](py::array_t<uint8_t, py::array::c_style> headers, uint64_t header_stride,
             const std::vector<py::array_t<uint8_t, py::array::c_style>>& packed, int list_len, int K, int x_min, int x_max,
             int y_min, int y_max, const std::vector<Trajectory>& all_cands) {
              if (x_max <= x_min || y_max <= y_min) throw std::runtime_error("merge_sparse_exact_host: invalid search bounds");
              const uint64_t n_pixels = (uint64_t)(x_max - x_min) * (uint64_t)(y_max - y_min);
              if ((uint64_t)headers.size() < header_stride * packed.size()) {
                  throw std::runtime_error("merge_sparse_exact_host: header buffer shorter than n_lists * header_stride");
              }
              std::vector<const kb_compact_result*> ptrs;
              for (size_t r = 0; r < packed.size(); ++r) {
                  uint64_t total = 0;
                  for (uint64_t pix = 0; pix < n_pixels; ++pix) total += headers.data()[r * header_stride + pix];
                  if ((uint64_t)packed[r].size() < total * sizeof(kb_compact_result)) {
                      throw std::runtime_error("merge_sparse_exact_host: list " + std::to_string(r) + " holds fewer records than its counts say");
                  }
                  ptrs.push_back(reinterpret_cast<const kb_compact_result*>(packed[r].data()));
              }
              std::vector<Trajectory> out = merge_sparse_exact_host(headers.data(), header_stride, ptrs, n_pixels, list_len, K,
                                                                    x_max - x_min, x_min, y_min, all_cands.data(), all_cands.size());
              py::array_t<uint8_t> res((py::ssize_t)(out.size() * sizeof(Trajectory)));
              if (!out.empty()) std::memcpy(res.mutable_data(), out.data(), out.size() * sizeof(Trajectory));
              return res;
          },
          "Host twin of kb_merge_sparse_exact: the tie-exact merge over sparse per-device lists (headers + packed records).");

    // ---- near-duplicate grid filter on the device (filters/clustering_grid.py:152-175) ----
    m.def(
            "grid_filter_indices",
            [](const std::vector<Trajectory>& trjs, double bin_width, double max_dt) {
                std::vector<uint32_t> kept(trjs.size());
                uint64_t n_kept = 0;
                if (!trjs.empty()) {
                    if (kb_device_count() == 0) throw std::runtime_error("GPU is not available for the grid filter.");
                    void *t_dev = nullptr, *k_dev = nullptr;
                    auto check = [&](int rc) {
                        if (rc != 0) {
                            if (t_dev) kb_free_gpu_block(t_dev);
                            if (k_dev) kb_free_gpu_block(k_dev);
                            throw std::runtime_error(kb_last_error());
                        }
                    };
                    check(kb_allocate_gpu_block(trjs.size() * sizeof(Trajectory), &t_dev));
                    check(kb_allocate_gpu_block(trjs.size() * sizeof(uint32_t), &k_dev));
                    check(kb_copy_block_to_gpu(trjs.data(), t_dev, trjs.size() * sizeof(Trajectory)));
                    check(kb_grid_filter(reinterpret_cast<const kb_trajectory*>(t_dev), trjs.size(), bin_width, max_dt,
                                         reinterpret_cast<uint32_t*>(k_dev), &n_kept, nullptr));
                    if (n_kept) check(kb_copy_block_to_cpu(kept.data(), k_dev, n_kept * sizeof(uint32_t)));
                    kb_free_gpu_block(t_dev);
                    kb_free_gpu_block(k_dev);
                } else {
                    // same argument checks as the device entry point
                    if (!(bin_width >= 1.0) || !(max_dt >= 0.0)) throw std::runtime_error("invalid bin width or max time");
                }
                kept.resize(n_kept);
                return kept;
            },
            py::arg("trajectories"), py::arg("bin_width") = 10.0, py::arg("max_dt") = 1.0);

    // ---- stamp coadds on the device (filters/stamp_filters.py:72-168, core/stamp_utils.py) ----
    py::class_<DeviceImageStack>(m, "DeviceImageStack")
            .def(py::init<DeviceImageStack::FloatArray, py::object>(), py::arg("sci"), py::arg("var") = py::none())
            .def_static(
                    "from_device",
                    [](uintptr_t sci_dev, uintptr_t var_dev, int T, int H, int W, py::object owner) {
                        return std::unique_ptr<DeviceImageStack>(new DeviceImageStack(sci_dev, var_dev, T, H, W, std::move(owner)));
                    },
                    py::arg("sci_dev"), py::arg("var_dev"), py::arg("num_times"), py::arg("height"), py::arg("width"),
                    py::arg("owner") = py::none(),
                    "Stacks [T][H][W] float32 that already are in the memory of the current device (addresses as integers; "
                    "var_dev 0 = none): used where they lie.  `owner` is kept alive with the stack.")
            .def_property_readonly("num_times", &DeviceImageStack::num_times)
            .def_property_readonly("height", &DeviceImageStack::height)
            .def_property_readonly("width", &DeviceImageStack::width)
            .def_property_readonly("has_variance", &DeviceImageStack::has_variance)
            .def("all_stamps", &DeviceImageStack::all_stamps, py::arg("xvals"), py::arg("yvals"), py::arg("radius"))
            .def("coadds", &DeviceImageStack::coadds, py::arg("xvals"), py::arg("yvals"), py::arg("to_include") = py::none(),
                 py::arg("radius") = 10, py::arg("coadd_types") = std::vector<std::string>{"mean"});

    // ---- batched sigma-G clipping on the device (filters/sigma_g_filter.py:114-168) ----
    m.def(
            "sigma_g_clip_matrix",
            [](py::array_t<float, py::array::c_style | py::array::forcecast> lh, float low_bnd, float high_bnd,
               float n_sigma, float coeff, bool clip_negative) {
                if (lh.ndim() != 2) throw std::runtime_error("sigma_g_clip_matrix: expected an N x T matrix");
                py::array_t<bool> valid({lh.shape(0), lh.shape(1)});
                if (kb_sigma_g_clip_matrix_host(lh.data(), (uint64_t)lh.shape(0), (int32_t)lh.shape(1), low_bnd, high_bnd,
                                                n_sigma, coeff, clip_negative ? 1 : 0,
                                                reinterpret_cast<uint8_t*>(valid.mutable_data())) != 0) {
                    throw std::runtime_error(kb_last_error());
                }
                return valid;
            },
            py::arg("lh"), py::arg("low_bnd") = 25.0f, py::arg("high_bnd") = 75.0f, py::arg("n_sigma") = 2.0f,
            py::arg("coeff") = 0.7413f, py::arg("clip_negative") = false);

    // ---- cpu_search_algorithms.cpp:128-131 ----
    m.def("evaluate_trajectory_cpu", &evaluate_trajectory_cpu,
                 "Fill lh, flux and obs_count of `candidate` from the host copy of `psi_phi`: psi and phi are summed "
                 "in epoch order over the pixels floor(x + vx t + 0.5), floor(y + vy t + 0.5) that hold data; lh = "
                 "psi_sum / sqrt(phi_sum), flux = psi_sum / phi_sum, both -1 when phi_sum <= 0.");
    m.def("search_cpu_only", &search_cpu_only,
                 "The host search: for every start pixel inside the bounds of `params`, evaluate all `candidates` (no "
                 "sigma-G clip) and keep the `results_per_pixel` most likely in `results` (slot layout ((y - y_min) * "
                 "width + (x - x_min)) * K + rank).");

    // ---- stack_search.cpp:341-389 ----
    py::class_<StackSearch>(m, "StackSearch")
            .def(py::init([](const std::vector<conv_array>& sci, const std::vector<conv_array>& var,
                             const std::vector<conv_array>& psfs, std::vector<double> times, int num_bytes) {
                     std::vector<Image> s = to_images(sci), v = to_images(var), p = to_images(psfs);
                     return std::unique_ptr<StackSearch>(new StackSearch(s, v, p, times, num_bytes));
                 }),
                 py::arg("sci_imgs"), py::arg("var_imgs"), py::arg("psf_kernels"), py::arg("zeroed_times"),
                 py::arg("num_bytes") = -1)
            // the ingest form: contiguous [T][H][W] float32 stacks straight to the device builder
            .def_static(
                    "from_image_stacks",
                    [](py::array_t<float, py::array::c_style | py::array::forcecast> sci,
                       py::array_t<float, py::array::c_style | py::array::forcecast> var,
                       const std::vector<conv_array>& psfs, std::vector<double> times, int num_bytes, bool separable_psf,
                       bool empty_footprint_is_zero, bool register_host_memory) {
                        if (sci.ndim() != 3 || var.ndim() != 3) {
                            throw std::runtime_error("from_image_stacks expects [T][H][W] arrays");
                        }
                        for (int d = 0; d < 3; ++d) {
                            if (sci.shape(d) != var.shape(d)) {
                                throw std::runtime_error("The science and variance stacks differ in shape.");
                            }
                        }
                        std::vector<Image> p = to_images(psfs);
                        const uint32_t flags = (separable_psf ? (uint32_t)KB_BUILD_SEPARABLE : 0u) |
                                               (empty_footprint_is_zero ? (uint32_t)KB_BUILD_EMPTY_IS_ZERO : 0u) |
                                               (register_host_memory ? (uint32_t)KB_BUILD_REGISTER_HOST : 0u);
                        return std::unique_ptr<StackSearch>(new StackSearch(sci.data(), var.data(), (unsigned)sci.shape(0),
                                                                            (unsigned)sci.shape(1), (unsigned)sci.shape(2), p,
                                                                            times, num_bytes, flags));
                    },
                    py::arg("sci_stack"), py::arg("var_stack"), py::arg("psf_kernels"), py::arg("zeroed_times"),
                    py::arg("num_bytes") = -1, py::arg("separable_psf") = false, py::arg("empty_footprint_is_zero") = false,
                    py::arg("register_host_memory") = true)
            // stacks that already live on the current device (kbmod_amd.fits_ingest decodes WorkUnit files there)
            .def_static(
                    "from_device_stacks",
                    [](uintptr_t sci_dev, uintptr_t var_dev, unsigned int T, unsigned int H, unsigned int W,
                       const std::vector<conv_array>& psfs, std::vector<double> times, int num_bytes, bool separable_psf,
                       bool empty_footprint_is_zero) {
                        std::vector<Image> p = to_images(psfs);
                        const uint32_t flags = (separable_psf ? (uint32_t)KB_BUILD_SEPARABLE : 0u) |
                                               (empty_footprint_is_zero ? (uint32_t)KB_BUILD_EMPTY_IS_ZERO : 0u);
                        return std::unique_ptr<StackSearch>(new StackSearch(
                                StackSearch::DeviceStacks{}, reinterpret_cast<const float*>(sci_dev),
                                reinterpret_cast<const float*>(var_dev), T, H, W, p, times, num_bytes, flags));
                    },
                    py::arg("sci_dev"), py::arg("var_dev"), py::arg("num_times"), py::arg("height"), py::arg("width"),
                    py::arg("psf_kernels"), py::arg("zeroed_times"), py::arg("num_bytes") = -1,
                    py::arg("separable_psf") = false, py::arg("empty_footprint_is_zero") = false,
                    "StackSearch over science / variance stacks [T][H][W] float32 that already are in the memory of the current "
                    "device (addresses as integers): psi/phi is built there, nothing is uploaded.  The stacks may be released "
                    "once the call returns.")
            .def("set_search_devices", &StackSearch::set_search_devices,
                 "Devices the GPU search fans out over: the candidate list is cut into contiguous slices, one host "
                 "thread per slice on its device, per-pixel lists merged on the current device.  For up to 16 results "
                 "per pixel the merge reproduces the single-device result exactly, ties included (2 K stable lists + "
                 "replay of the reference's insertion); for 17 .. 32 equal likelihoods go to the lower candidate index "
                 "(logged); above 32 the search stays on one device (logged).  Entries may repeat.")
            .def("get_search_devices", &StackSearch::get_search_devices,
                 "The device list set by set_search_devices (empty: the current device alone).")
            .def_property_readonly("num_images", &StackSearch::num_images)
            .def_property_readonly("height", &StackSearch::get_image_height)
            .def_property_readonly("width", &StackSearch::get_image_width)
            .def_property_readonly("zeroed_times", &StackSearch::get_zeroed_times)
            .def("search_all", &StackSearch::search_all,
                 "Search every start pixel inside the current bounds for every trajectory of `search_list` (x, y are "
                 "ignored, vx, vy used) and keep the `results_per_pixel` best per pixel; then drop results below "
                 "min_lh / min_obs and sort all by likelihood, highest first.  on_gpu=True runs the HIP kernels "
                 "(per-pixel lists built in candidate order by strict-greater insertion, in-search sigma-G clip when "
                 "enabled; raises RuntimeError without a GPU), False the host search (no sigma-G).  Results are read "
                 "with get_results().")
            .def("evaluate_single_trajectory", &StackSearch::evaluate_single_trajectory,
                 "Fill lh, flux and obs_count of `trj` IN PLACE from its x, y, vx, vy.  use_kernel=False: the host "
                 "evaluator (no sigma-G); True: the host instantiation of the device evaluator, sigma-G clip and "
                 "thresholds included (needs a GPU, as in the reference).")
            .def("search_linear_trajectory", &StackSearch::search_linear_trajectory,
                 "A new Trajectory at (x, y, vx, vy) evaluated as evaluate_single_trajectory does.")
            .def("set_min_obs", &StackSearch::set_min_obs,
                 "Per-pixel lists only take trajectories with at least this many valid observations; 0 .. number of "
                 "images.")
            .def("set_min_lh", &StackSearch::set_min_lh,
                 "Results with a likelihood below this are dropped after the search (and, with the sigma-G filter on, "
                 "before they enter a per-pixel list).")
            .def("set_results_per_pixel", &StackSearch::set_results_per_pixel,
                 "How many results each start pixel keeps (K > 0; default 8).")
            .def("disable_gpu_sigmag_filter", &StackSearch::disable_gpu_sigmag_filter,
                 "Turn the in-search sigma-G clip off.")
            .def("enable_gpu_sigmag_filter", &StackSearch::enable_gpu_sigmag_filter,
                 "Turn the in-search sigma-G clip on: `percentiles` = two quantiles in (0, 1), low < high, of the "
                 "per-epoch psi / phi ratios; `sigmag_coeff` > 0 converts their spread into a standard deviation "
                 "(0.7413 for [0.25, 0.75]); samples further than two of those from the median are left out of the "
                 "sums.  `min_lh` is the likelihood a trajectory must keep after the clip.")
            .def("set_start_bounds_x", &StackSearch::set_start_bounds_x,
                 "Start pixels cover x in [x_min, x_max) (may extend beyond the image); x_min < x_max.")
            .def("set_start_bounds_y", &StackSearch::set_start_bounds_y,
                 "Start pixels cover y in [y_min, y_max); y_min < y_max.")
            .def("get_num_images", &StackSearch::num_images,
                 "Number of epochs.")
            .def("get_image_width", &StackSearch::get_image_width,
                 "Image width in pixels.")
            .def("get_image_height", &StackSearch::get_image_height,
                 "Image height in pixels.")
            .def("get_all_psi_phi_curves",
                 [](StackSearch& s, const std::vector<Trajectory>& t) { return from_image(s.get_all_psi_phi_curves(t)); },
                 "For every trajectory the psi values at its T predicted pixels followed by the T phi values (N x 2T "
                 "float32; 0 where there is no data).  Runs on the device when the array is resident there.")
            .def("preload_psi_phi_array", &StackSearch::preload_psi_phi_array,
                 "Keep the psi/phi array resident in HBM across searches (search_all otherwise releases its logical "
                 "hold after each search).")
            .def("unload_psi_phi_array", &StackSearch::unload_psi_phi_array,
                 "End the residency started by preload_psi_phi_array.")
            .def("psi_phi_array_on_gpu", &StackSearch::psi_phi_array_on_gpu,
                 "Whether the array is marked resident on the device.")
            .def("get_number_total_results", &StackSearch::get_number_total_results,
                 "Number of results the last search kept.")
            .def("get_results", &StackSearch::get_results,
                 "Copies of `count` results starting at `start`, most likely first; raises for a negative start or "
                 "count.")
            .def("get_all_results", &StackSearch::get_all_results,
                 "All results of the last search as a list (copies).")
            .def("set_results", &StackSearch::set_results,
                 "Replace the stored results (used by tests and by callers that post-filter).")
            .def("clear_results", &StackSearch::clear_results,
                 "Forget the stored results.")
            .def("compute_max_results", &StackSearch::compute_max_results,
                 "Upper bound of results a search can return: start-area width x height x results_per_pixel.")
            // Extras (not in the reference): bulk results, the psi/phi store, kernel timing, debug flags.
            .def("get_psi_phi_array", &StackSearch::get_psi_phi_array, py::return_value_policy::reference_internal,
                 "The PsiPhiArray this search owns (reference).")
            .def("set_search_flags", &StackSearch::set_search_flags,
                 "Flags handed to kb_device_search_filter (include/kbmod_hip.h: kernel choice, tile height, decode "
                 "form); 0 lets the library choose.  None changes a result.")
            .def("results_to_numpy",
                 [](StackSearch& s) {
                     const std::vector<Trajectory>& v = s.get_all_results();
                     py::array_t<double> out({(py::ssize_t)v.size(), (py::ssize_t)7});
                     auto r = out.mutable_unchecked<2>();
                     for (py::ssize_t i = 0; i < (py::ssize_t)v.size(); ++i) {
                         r(i, 0) = v[i].x;
                         r(i, 1) = v[i].y;
                         r(i, 2) = v[i].vx;
                         r(i, 3) = v[i].vy;
                         r(i, 4) = v[i].lh;
                         r(i, 5) = v[i].flux;
                         r(i, 6) = v[i].obs_count;
                     }
                     return out;
                 },
                 "The stored results as an (N, 7) float64 array with columns x, y, vx, vy, lh, flux, obs_count.")
            .def("last_search_stats", [](StackSearch& s) {
                const kb_search_stats& st = s.last_search_stats();
                py::dict d;
                d["search_kernel_ms"] = st.search_kernel_ms;
                d["table_kernel_ms"] = st.table_kernel_ms;
                d["num_evals"] = st.num_evals;
                d["algorithmic_bytes"] = st.algorithmic_bytes;
                d["kernel_variant"] = st.kernel_variant;
                d["num_search_launches"] = st.num_search_launches;
                d["sigmag_work_items"] = st.sigmag_work_items;
                d["sigmag_trajectories"] = st.sigmag_trajectories;
                d["sigmag_literal"] = st.sigmag_literal;
                d["kernel_name"] = std::string(st.kernel_name);
                d["special_epochs"] = st.special_epochs;
                d["edge_count_tables"] = st.edge_count_tables;
                d["env_overrides"] = st.env_overrides;
                const auto& h = s.last_host_times();
                d["host_search_ms"] = h.search;
                d["host_filter_sort_ms"] = h.filter_sort;
                d["host_download_ms"] = h.download;
                d["host_validate_ms"] = h.validate;
                d["host_total_ms"] = h.total;
                return d;
            },
                 "Measurements of the last device search: kernel and table times (HIP events), evaluations, "
                 "algorithmic bytes, the kernel instance that ran, sigma-G work counters.");

    m.def("pixel_value_valid", &pixel_value_valid,
                 "True for a finite pixel value (NO_DATA is NaN).");

    // ---- kernel_helpers.cpp:109-117 ----
    m.def("kb_has_gpu", &has_gpu, "Check if GPU is available");
    m.def("sigmag_filtered_indices", &sigmaGFilteredIndices,
                 "Test hook of the in-search sigma-G clip: the indices of `values` that survive, in ascending value "
                 "order (percentiles sgl0 / sgl1, coefficient, width in sigma-G).");
    m.def("print_cuda_stats", &print_cuda_stats,
                 "Print device name, compute units and memory to stdout.");
    m.def("get_gpu_total_memory", &get_gpu_total_memory,
                 "Bytes of HBM on the current device (0 without a GPU).");
    m.def("get_gpu_free_memory", &get_gpu_free_memory,
                 "Bytes of HBM currently free (0 without a GPU).");
    m.def("stat_gpu_memory_mb", &stat_gpu_memory_mb,
                 "One line of text with free and total HBM in MB.");
    m.def("validate_gpu", &validate_gpu, py::arg("req_memory") = 0,
                 "True when a GPU is present and has at least `req_memory` bytes free.");

    // ---- image_utils_cpp.cpp:180-194 (noconvert arguments) ----
    m.def(
            "convolve_image_cpu",
            [](const py::array& image, const py::array& psf) {
                return from_image(convolve_image_cpu(strict_image(image, "image"), strict_image(psf, "psf")));
            },
            py::arg("image").noconvert(true), py::arg("psf").noconvert(true),
                 "Masked, renormalised correlation of a float32 image with a PSF kernel on the host: non-finite pixels "
                 "are skipped and pass through, the result is (sum of value x weight) x (kernel total) / (weight "
                 "seen); NaN where no tap counted.");
    m.def(
            "convolve_image_gpu",
            [](const py::array& image, const py::array& psf) {
                return from_image(convolve_image_gpu(strict_image(image, "image"), strict_image(psf, "psf")));
            },
            py::arg("image").noconvert(true), py::arg("psf").noconvert(true),
                 "The same correlation on the device (0.0 instead of NaN where no tap counted, as the reference's "
                 "device kernel); raises RuntimeError without a GPU.");
    m.def(
            "convolve_image",
            [](const py::array& image, const py::array& psf) {
                return from_image(convolve_image(strict_image(image, "image"), strict_image(psf, "psf")));
            },
            py::arg("image").noconvert(true), py::arg("psf").noconvert(true),
                 "convolve_image_gpu when a GPU is present, else convolve_image_cpu.");
    m.def(
            "square_psf_values",
            [](const py::array& psf) { return from_image(square_psf_values(strict_image(psf, "given_psf"))); },
            py::arg("given_psf").noconvert(true),
                 "Element-wise square of a PSF kernel (the kernel phi is built with).");
    m.def(
            "generate_psi",
            [](const py::array& sci, const py::array& var, const py::array& psf) {
                return from_image(generate_psi(strict_image(sci, "sci"), strict_image(var, "var"), strict_image(psf, "psf")));
            },
            py::arg("sci").noconvert(true), py::arg("var").noconvert(true), py::arg("psf").noconvert(true),
                 "psi image of one epoch: (science / variance) correlated with the PSF; NO_DATA where the variance is "
                 "non-finite or zero or the science pixel is non-finite.");
    m.def(
            "generate_phi",
            [](const py::array& var, const py::array& psf) {
                return from_image(generate_phi(strict_image(var, "var"), strict_image(psf, "psf")));
            },
            py::arg("var").noconvert(true), py::arg("psf").noconvert(true),
                 "phi image of one epoch: (1 / variance) correlated with the squared PSF; NO_DATA where the variance "
                 "is non-finite or zero.");
}
