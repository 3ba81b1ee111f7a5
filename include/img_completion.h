// img_completion.h -- the header the reference includes but does not ship
// (/root/reference/src/DC_lidar_only/img_completion.cpp:15, main.cpp:1, utils.cpp:3).
//
// Drop-in: with this file on the include path, and libdcmt_hip.so linked, the reference's
// mains (DC_lidar_only/main.cpp, DC_lidar_camera/main_lc.cpp, DC_stereo_lidar/main_sl*.cpp)
// compile unmodified apart from dropping their textual `#include ".../img_completion.cpp"` /
// `#include ".../img_completion_lc.cpp"` lines: the two functions below have the reference's
// exact signatures and semantics, and run on the GPU through the C ABI of dcmt.h.
//
//   void img_completion(const cv::Mat&, cv::Mat&, const bool&, const std::string&)
//        reference: src/DC_lidar_only/img_completion.cpp:17-20
//   void interpolate_with_superpixels(Slic&, const cv::Mat&, cv::Mat&, const std::string&, int)
//        reference: src/DC_lidar_camera/img_completion_lc.cpp:34-38   (define DCMT_WITH_SLIC
//        after including the reference's slic.h, or use the generic overload below)
//
// Header-only C++11.  Needs only cv::Mat from OpenCV (rows, cols, type(), step[0], ptr<float>(),
// create(), CV_32FC1): the tests compile it against a 60-line stand-in (tests/mock_opencv).
#ifndef IMG_COMPLETION_H
#define IMG_COMPLETION_H

#include <opencv2/opencv.hpp>

#include <cstdint>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "dcmt.h"

namespace dcmt_shim {

// One context per thread, grown on demand: the reference function is re-entrant and has no
// state, a dcmt_ctx must not be shared between threads.
struct ThreadCtx {
    dcmt_ctx* ctx = nullptr;
    int rows = 0, cols = 0;
    ~ThreadCtx() { dcmt_destroy(ctx); }
    dcmt_ctx* get(int r, int c)
    {
        if (!ctx || r > rows || c > cols) {
            dcmt_destroy(ctx);
            ctx = nullptr;
            rows = r > rows ? r : rows;
            cols = c > cols ? c : cols;
            const int st = dcmt_create(device(), rows, cols, 1, &ctx);
            if (st != DCMT_OK) throw std::runtime_error(std::string("dcmt_create: ") + dcmt_strerror(st));
        }
        return ctx;
    }
    static int& device() { static int d = 0; return d; }     // dcmt_shim::ThreadCtx::device() = k selects the GPU
};

// The reference prints while it works (img_completion.cpp:29 the dimensions, :50 "max range is", :161 one hole count per loop
// iteration; img_completion_lc.cpp:173 the hole counts only) and so does the drop-in; dcmt_shim::quiet() = true silences it.
inline bool& quiet() { static bool q = false; return q; }

inline ThreadCtx& thread_ctx()
{
    static thread_local ThreadCtx t;
    return t;
}

inline void check_input(const cv::Mat& m)
{
    // The reference reads at<float>(i,j) without checking: any other type is undefined behaviour
    // there; here it is an error.
    if (m.type() != CV_32FC1) throw std::runtime_error("img_completion: sparse image must be CV_32FC1");
    if (m.rows < 1 || m.cols < 1) throw std::runtime_error("img_completion: empty image");
}

inline int blur_from_string(const std::string& blur_type)
{
    if (blur_type == "gaussian") return DCMT_BLUR_GAUSSIAN;   // img_completion.cpp:176
    if (blur_type == "bilateral") return DCMT_BLUR_BILATERAL; // :172 -- cv::bilateralFilter in place throws; so do we
    return DCMT_BLUR_NONE;                                    // any other string: no blur
}

inline void raise(int st, const char* what)
{
    // The reference's failure mode is a cv::Exception out of an OpenCV call; the closest here.
    if (st != DCMT_OK) throw std::runtime_error(std::string(what) + ": " + dcmt_strerror(st));
}

// Generic form of interpolate_with_superpixels: labels[col][row] exactly as Slic::clusters
// (reference slic.cpp:21-30 builds it column-major), n_labels = slic.centers.size().
inline void interpolate_with_labels(const std::vector<std::vector<int> >& clusters, int n_labels,
                                    const cv::Mat& sparse_r_img, cv::Mat& dense_r_img,
                                    const std::string& /*blur_type: unused by the reference, img_completion_lc.cpp:183*/,
                                    int use_superpixel, const double* normalize_range = nullptr /* {alpha, beta}: see below */)
{
    check_input(sparse_r_img);
    const int rows = sparse_r_img.rows, cols = sparse_r_img.cols;
    std::vector<int32_t> lab((size_t)rows * cols, -1);
    if (use_superpixel) {
        if ((int)clusters.size() < cols) throw std::runtime_error("interpolate_with_superpixels: clusters smaller than the image");
        for (int j = 0; j < cols; ++j) {
            if ((int)clusters[j].size() < rows) throw std::runtime_error("interpolate_with_superpixels: clusters smaller than the image");
            for (int i = 0; i < rows; ++i) lab[(size_t)i * cols + j] = clusters[j][i];   // [col][row] -> row-major
        }
    }
    cv::Mat out;
    out.create(rows, cols, CV_32FC1);
    dcmt_params p;
    dcmt_default_params(&p);
    if (normalize_range) { p.flags |= DCMT_FLAG_NORMALIZE; p.norm_lo = (float)normalize_range[0]; p.norm_hi = (float)normalize_range[1]; }
    p.verbose = quiet() ? 0 : 2;                             // img_completion_lc.cpp:173
    const int st = dcmt_complete_labeled_f32(thread_ctx().get(rows, cols), sparse_r_img.ptr<float>(), sparse_r_img.step[0], 0,
                                             lab.data(), sizeof(int32_t) * (size_t)cols, 0, n_labels, out.ptr<float>(),
                                             out.step[0], 0, rows, cols, 1, &p, use_superpixel);
    raise(st, "interpolate_with_superpixels");
    dense_r_img = out;
}

// The stereo-lidar callers' two lines in one call (DC_stereo_lidar/main_sl.cpp:370 + :386):
//     cv::normalize(projected, normalized, alpha, beta, cv::NORM_MINMAX);  img_completion(normalized, dense, extr, blur);
// the min-max pass runs on the GPU in front of the cascade (DCMT_FLAG_NORMALIZE) and the normalised image is never
// materialised.  For the labeled variant (:523 + :540) pass normalize_range to interpolate_with_labels.
inline void img_completion_normalized(const cv::Mat& projected, cv::Mat& dense_r_img, double alpha, double beta,
                                      const bool& /*extr*/, const std::string& blur_type)
{
    check_input(projected);
    const int rows = projected.rows, cols = projected.cols;
    cv::Mat out;
    out.create(rows, cols, CV_32FC1);
    dcmt_params p;
    dcmt_default_params(&p);
    p.blur = blur_from_string(blur_type);
    p.flags |= DCMT_FLAG_NORMALIZE;
    p.norm_lo = (float)alpha;
    p.norm_hi = (float)beta;
    p.verbose = quiet() ? 0 : 2;                             // (the reference's "max range is" would be that of the normalised image: beta)
    if (!quiet()) std::cout << "NUMERO ROWS, COLS: " << rows << " " << cols << std::endl << "max range is" << (float)(alpha > beta ? alpha : beta) << std::endl;
    const int st = dcmt_complete_f32(thread_ctx().get(rows, cols), projected.ptr<float>(), projected.step[0], 0,
                                     out.ptr<float>(), out.step[0], 0, rows, cols, 1, &p);
    raise(st, "img_completion_normalized");
    dense_r_img = out;
}

// ---- the steps either side of the path, on cv::Mat / std::vector like the reference's own code -----------------------

// DC_stereo_lidar/main_sl.cpp:478-520: velodyne points ([n][4] floats: x, y, z, reflectance, the .bin payload) through
// T (4x4) and P (3x4), both ROW-major, into a fresh CV_32FC1 image of rows x cols (0 = no point), like projected_depths.
inline void project_points(const float* xyzi, int n_points, const float T[16], const float P[12], int rows, int cols,
                           cv::Mat& projected_depths)
{
    cv::Mat out;
    out.create(rows, cols, CV_32FC1);
    raise(dcmt_project_points(thread_ctx().get(rows, cols), xyzi, n_points, T, P, out.ptr<float>(), out.step[0], rows, cols),
          "project_points");
    projected_depths = out;
}

// Slic::generate_superpixels (DC_lidar_camera/slic.cpp:101-182) on the CV_8UC3 image the reference passes; fills `clusters`
// as Slic::clusters ([col][row]) and returns slic.centers.size().
inline int slic_labels(const cv::Mat& lab_image, int step, int nc, std::vector<std::vector<int> >& clusters)
{
    if (lab_image.type() != CV_8UC3 || lab_image.rows < 1 || lab_image.cols < 1) throw std::runtime_error("slic_labels: image must be CV_8UC3");
    const int rows = lab_image.rows, cols = lab_image.cols;
    std::vector<int32_t> lab((size_t)rows * cols);
    raise(dcmt_slic_labels(thread_ctx().get(rows, cols), lab_image.ptr<unsigned char>(), lab_image.step[0], rows, cols, step, nc,
                           lab.data(), nullptr), "slic_labels");
    clusters.assign(cols, std::vector<int>(rows));
    for (int j = 0; j < cols; ++j)
        for (int i = 0; i < rows; ++i) clusters[j][i] = lab[(size_t)i * cols + j];
    return dcmt_slic_num_centers(rows, cols, step);
}

// DC_stereo_lidar/main_sl.cpp:1165-1246: get_initial_disparity + calculateMeasuementDerivatives + optimize_IG +
// retrieve_optimized_depth on the dense depth (CV_32FC1) and the two grey images (CV_8UC1).
inline void stereo_refine(const cv::Mat& dense_depth, const cv::Mat& left_gray, const cv::Mat& right_gray, cv::Mat& optimized_depth)
{
    check_input(dense_depth);
    const int rows = dense_depth.rows, cols = dense_depth.cols;
    if (left_gray.type() != CV_8UC1 || right_gray.type() != CV_8UC1 || left_gray.rows != rows || right_gray.rows != rows ||
        left_gray.cols != cols || right_gray.cols != cols) throw std::runtime_error("stereo_refine: grey images must be CV_8UC1 of the depth's size");
    cv::Mat out;
    out.create(rows, cols, CV_32FC1);
    dcmt_stereo_params sp;
    dcmt_default_stereo_params(&sp);
    raise(dcmt_stereo_refine(thread_ctx().get(rows, cols), dense_depth.ptr<float>(), dense_depth.step[0], left_gray.ptr<unsigned char>(),
                             left_gray.step[0], right_gray.ptr<unsigned char>(), right_gray.step[0], out.ptr<float>(), out.step[0], rows, cols, &sp),
          "stereo_refine");
    optimized_depth = out;
}

}  // namespace dcmt_shim

// reference: src/DC_lidar_only/img_completion.cpp:17-20.  `extr` is accepted and ignored, as there.
inline void img_completion(const cv::Mat& sparse_r_img, cv::Mat& dense_r_img, const bool& /*extr*/, const std::string& blur_type)
{
    dcmt_shim::check_input(sparse_r_img);
    const int rows = sparse_r_img.rows, cols = sparse_r_img.cols;
    cv::Mat out;                              // the reference overwrites dense_r_img with a fresh clone (:27)
    out.create(rows, cols, CV_32FC1);
    dcmt_params p;
    dcmt_default_params(&p);
    p.blur = dcmt_shim::blur_from_string(blur_type);
    p.verbose = dcmt_shim::quiet() ? 0 : 1;   // :29, :50, :161 -- printed by the library (stdout)
    const int st = dcmt_complete_f32(dcmt_shim::thread_ctx().get(rows, cols), sparse_r_img.ptr<float>(), sparse_r_img.step[0], 0,
                                     out.ptr<float>(), out.step[0], 0, rows, cols, 1, &p);
    dcmt_shim::raise(st, "img_completion");
    dense_r_img = out;
}

#ifdef DCMT_WITH_SLIC
// reference: src/DC_lidar_camera/img_completion_lc.cpp:34-38 (include the reference's slic.h first)
inline void interpolate_with_superpixels(Slic& slic, const cv::Mat& sparse_r_img, cv::Mat& dense_r_img,
                                         const std::string& blur_type, int use_superpixel)
{
    dcmt_shim::interpolate_with_labels(slic.clusters, (int)slic.centers.size(), sparse_r_img, dense_r_img, blur_type, use_superpixel);
}
#endif

#endif  // IMG_COMPLETION_H
