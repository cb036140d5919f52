// Test-only stand-in for the two OpenCV classes the reference's sample uses to read its JSON
// (cv::FileStorage / cv::FileNode).  OpenCV is not in this image; tests/test_boundary.py only
// type-checks the reference's own samples/sample_ba_from_file/main.cpp against the mirrored cugo
// headers (g++ -fsyntax-only), so declarations are all that is needed here.
#pragma once
#include <string>
#include <vector>
#define CV_Assert(x) ((void)(x))
namespace cv
{
class FileNode
{
public:
    FileNode operator[](const char*) const;
    operator int() const;
    operator float() const;
    operator double() const;
    const FileNode* begin() const;
    const FileNode* end() const;
};
class FileStorage
{
public:
    enum { READ = 0 };
    FileStorage(const std::string&, int);
    bool isOpened() const;
    FileNode operator[](const char*) const;
};
} // namespace cv
