// Exercises include/img_completion.h exactly as the reference's main.cpp calls it
// (src/DC_lidar_only/main.cpp:93: img_completion(sparse, dense, false, "gaussian")) and as
// main_lc.cpp:220 calls interpolate_with_superpixels.  Reads raw f32/int32 files written by the
// pytest driver, writes raw f32 results; the driver compares them with the oracle.
//   shim_test <rows> <cols> <in.f32> <out.f32> [labels.i32 n_labels out_lc.f32 [out_norm100.f32 out_lc_norm80.f32]]
// Stand-in for the reference's Slic (src/DC_lidar_camera/slic.h:30-40): only the two public members the reference's
// interpolate_with_superpixels reads -- clusters[col][row] (:83) and centers.size() (:78).  With it and DCMT_WITH_SLIC
// the header below compiles the reference's exact signature (img_completion_lc.cpp:34-38), called further down.
#include <vector>
class Slic {
public:
    std::vector<std::vector<int> > clusters;        // [col][row]
    std::vector<std::vector<double> > centers;      // one (L, a, b, x, y) row per centre
};
#define DCMT_WITH_SLIC
#include "img_completion.h"

#include <cstdio>
#include <cstdlib>

static bool read_all(const char* path, void* dst, size_t bytes)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    const size_t n = std::fread(dst, 1, bytes, f);
    std::fclose(f);
    return n == bytes;
}
static bool write_all(const char* path, const void* src, size_t bytes)
{
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const size_t n = std::fwrite(src, 1, bytes, f);
    std::fclose(f);
    return n == bytes;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const int rows = std::atoi(argv[1]), cols = std::atoi(argv[2]);
    // a strided input (row step > cols * 4), as a ROI of a bigger cv::Mat would be
    const size_t pad = 16, row_step = (size_t)(cols + pad) * sizeof(float);
    std::vector<float> storage((size_t)rows * (cols + pad), -7.0f), packed((size_t)rows * cols);
    if (!read_all(argv[3], packed.data(), packed.size() * 4)) return 3;
    for (int r = 0; r < rows; ++r) std::memcpy(&storage[(size_t)r * (cols + pad)], &packed[(size_t)r * cols], (size_t)cols * 4);
    cv::Mat sparse(rows, cols, CV_32FC1, storage.data(), row_step);
    cv::Mat dense;                                   // empty, as in the reference's main (main.cpp:89)
    img_completion(sparse, dense, false, "gaussian");
    if (dense.rows != rows || dense.cols != cols || dense.type() != CV_32FC1) return 4;
    std::vector<float> out((size_t)rows * cols);
    for (int r = 0; r < rows; ++r) std::memcpy(&out[(size_t)r * cols], dense.ptr<float>(r), (size_t)cols * 4);
    if (!write_all(argv[4], out.data(), out.size() * 4)) return 5;
    for (int r = 0; r < rows; ++r)                    // the input must be untouched (const&)
        if (std::memcmp(&storage[(size_t)r * (cols + pad)], &packed[(size_t)r * cols], (size_t)cols * 4) != 0) return 6;

    bool threw = false;                               // "bilateral" throws in the reference (in-place cv::bilateralFilter)
    try { img_completion(sparse, dense, false, "bilateral"); } catch (const std::exception&) { threw = true; }
    if (!threw) return 7;

    if (argc >= 8) {
        const int n_labels = std::atoi(argv[6]);
        std::vector<int32_t> lab((size_t)rows * cols);
        if (!read_all(argv[5], lab.data(), lab.size() * 4)) return 8;
        std::vector<std::vector<int> > clusters(cols, std::vector<int>(rows));   // [col][row], as Slic::clusters
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) clusters[c][r] = lab[(size_t)r * cols + c];
        cv::Mat dense_sp;
        dcmt_shim::interpolate_with_labels(clusters, n_labels, sparse, dense_sp, "gaussian", 1);
        for (int r = 0; r < rows; ++r) std::memcpy(&out[(size_t)r * cols], dense_sp.ptr<float>(r), (size_t)cols * 4);
        if (!write_all(argv[7], out.data(), out.size() * 4)) return 9;
        {
            // the reference's own overload, as main_lc.cpp:220 calls it: interpolate_with_superpixels(slic, sparse, dense_sp, "gaussian", 1)
            Slic slic;
            slic.clusters = clusters;
            slic.centers.assign((size_t)n_labels, std::vector<double>(5, 0.0));
            cv::Mat dense_ref;
            interpolate_with_superpixels(slic, sparse, dense_ref, "gaussian", 1);
            if (dense_ref.rows != rows || dense_ref.cols != cols) return 20;
            for (int r = 0; r < rows; ++r)
                if (std::memcmp(dense_ref.ptr<float>(r), dense_sp.ptr<float>(r), (size_t)cols * 4) != 0) return 21;
            // use_superpixel == 0 (img_completion_lc.cpp:59-64): the plain first stage, Gaussian unconditional
            cv::Mat dense_nosp, dense_plain;
            interpolate_with_superpixels(slic, sparse, dense_nosp, "none", 0);
            img_completion(sparse, dense_plain, false, "gaussian");
            for (int r = 0; r < rows; ++r)
                if (std::memcmp(dense_nosp.ptr<float>(r), dense_plain.ptr<float>(r), (size_t)cols * 4) != 0) return 22;
        }
        if (argc >= 10) {
            // the stereo-lidar callers: cv::normalize(..., 0, 100) + img_completion (main_sl.cpp:370, :386) and
            // cv::normalize(..., 0, 80) + interpolate_with_superpixels (:523, :540), each as one fused call
            cv::Mat dense_n;
            dcmt_shim::img_completion_normalized(sparse, dense_n, 0, 100, false, "gaussian");
            for (int r = 0; r < rows; ++r) std::memcpy(&out[(size_t)r * cols], dense_n.ptr<float>(r), (size_t)cols * 4);
            if (!write_all(argv[8], out.data(), out.size() * 4)) return 10;
            const double range[2] = {0, 80};
            dcmt_shim::interpolate_with_labels(clusters, n_labels, sparse, dense_n, "gaussian", 1, range);
            for (int r = 0; r < rows; ++r) std::memcpy(&out[(size_t)r * cols], dense_n.ptr<float>(r), (size_t)cols * 4);
            if (!write_all(argv[9], out.data(), out.size() * 4)) return 11;
        }
    }
    if (argc >= 17) {
        // the steps either side of the path through the shim: argv[10] points.f32 [n][4], argv[11] n, argv[12] out_proj.f32 (T, P = the
        // KITTI-like constants below); argv[13] lab.u8 [rows][cols][3] -> argv[14] out_labels.i32; argv[15] right.u8 (left = channel-free
        // copy of lab's first plane is not available here: the driver writes left.u8 next to it as argv[15] + ".left") -> argv[16] out_stereo.f32
        const int n_pts = std::atoi(argv[11]);
        std::vector<float> pts((size_t)n_pts * 4);
        if (!read_all(argv[10], pts.data(), pts.size() * 4)) return 12;
        const float T[16] = {7.533745e-03f, -9.999714e-01f, -6.166020e-04f, -4.069766e-03f, 1.480249e-02f, 7.280733e-04f, -9.998902e-01f, -7.631618e-02f,
                             9.998621e-01f, 7.523790e-03f, 1.480755e-02f, -2.717806e-01f, 0.f, 0.f, 0.f, 1.f};
        const float P[12] = {7.215377e+02f, 0.f, 6.095593e+02f, 4.485728e+01f, 0.f, 7.215377e+02f, 1.728540e+02f, 2.163791e-01f, 0.f, 0.f, 1.f, 2.745884e-03f};
        cv::Mat proj;
        dcmt_shim::project_points(pts.data(), n_pts, T, P, rows, cols, proj);
        for (int r = 0; r < rows; ++r) std::memcpy(&out[(size_t)r * cols], proj.ptr<float>(r), (size_t)cols * 4);
        if (!write_all(argv[12], out.data(), out.size() * 4)) return 13;

        cv::Mat lab_img(rows, cols, CV_8UC3);
        if (!read_all(argv[13], lab_img.ptr<unsigned char>(), (size_t)rows * cols * 3)) return 14;
        std::vector<std::vector<int> > cl;
        const int n_centers = dcmt_shim::slic_labels(lab_img, 18, 50, cl);
        std::vector<int32_t> lab_out((size_t)rows * cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) lab_out[(size_t)r * cols + c] = cl[c][r];
        if (n_centers <= 0 || !write_all(argv[14], lab_out.data(), lab_out.size() * 4)) return 15;

        cv::Mat lg(rows, cols, CV_8UC1), rg(rows, cols, CV_8UC1), refined;
        const std::string left_path = std::string(argv[15]) + ".left";
        if (!read_all(left_path.c_str(), lg.ptr<unsigned char>(), (size_t)rows * cols) || !read_all(argv[15], rg.ptr<unsigned char>(), (size_t)rows * cols)) return 16;
        dcmt_shim::stereo_refine(dense, lg, rg, refined);
        for (int r = 0; r < rows; ++r) std::memcpy(&out[(size_t)r * cols], refined.ptr<float>(r), (size_t)cols * 4);
        if (!write_all(argv[16], out.data(), out.size() * 4)) return 17;
    }
    std::printf("shim ok\n");
    return 0;
}
