// Minimal stand-in for the part of cv::Mat that include/img_completion.h touches, so the shim
// can be compiled and exercised in images without OpenCV.  Written from OpenCV's documented
// interface; used ONLY to build this repo's own shim + test, never any reference source.
#pragma once
#include <cstddef>
#include <cstring>
#include <memory>

#define CV_32FC1 5
#define CV_8UC1 0
#define CV_8UC3 16

namespace cv {
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step[2] = {0, 0};
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    // a view over caller memory with an explicit row step (bytes), like cv::Mat(rows, cols, type, data, step)
    Mat(int r, int c, int type, void* data, size_t row_step) : rows(r), cols(c), type_(type), data_((unsigned char*)data)
    {
        step[0] = row_step; step[1] = elem();
    }
    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == type_ && buf_) return;
        rows = r; cols = c; type_ = type;
        step[1] = elem(); step[0] = step[1] * (size_t)c;
        buf_.reset(new unsigned char[step[0] * (size_t)r], std::default_delete<unsigned char[]>());
        data_ = buf_.get();
    }
    int type() const { return type_; }
    bool empty() const { return rows == 0 || cols == 0; }
    template <typename T> T* ptr(int r = 0) { return (T*)(data_ + step[0] * (size_t)r); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data_ + step[0] * (size_t)r); }
    template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
private:
    size_t elem() const { return type_ == CV_32FC1 ? 4 : (type_ == CV_8UC3 ? 3 : 1); }
    int type_ = CV_32FC1;
    std::shared_ptr<unsigned char> buf_;     // ref-counted like cv::Mat
    unsigned char* data_ = nullptr;
};
}  // namespace cv
