// Bundle adjustment from a graph file through the cugo C++ API on MI355X.
//
// Reads the JSON format of the reference's KITTI samples (ref: samples/sample_ba_from_file/
// main.cpp:89-160 — top-level fx, fy, cx, cy, bf; pose_vertices[{id, fixed, q[x,y,z,w], t[3]}];
// landmark_vertices[{id, fixed, Xw[3]}]; monocular_edges / stereo_edges[{vertexP, vertexL,
// measurement[2|3], information}]) and runs the same protocol: warm-up initialize()+optimize(1),
// then timed initialize()+optimize(10), printing chi2 per iteration.
//
//   sample_ba_from_file ba_kitti_00.json [iterations]
//
// Build (see __graft_entry__.build):
//   g++ -std=c++17 -O2 -I cuda-bundle-adjustment_amd/include samples/sample_ba_from_file.cpp \
//       -L cuda-bundle-adjustment_amd -lcugo_hip -Wl,-rpath,'$ORIGIN/../cuda-bundle-adjustment_amd'
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <variant>
#include <vector>

#include <ba_types.h>
#include <cuda_graph_optimisation.h>

// ---- a minimal JSON reader (objects, arrays, numbers, strings, true/false/null) -------------
struct Json;
using JsonArray = std::vector<Json>;
using JsonObject = std::map<std::string, Json>;
struct Json
{
    std::variant<std::nullptr_t, double, std::string, std::shared_ptr<JsonArray>, std::shared_ptr<JsonObject>> v;
    double num() const { return std::get<double>(v); }
    const JsonArray& arr() const { return *std::get<std::shared_ptr<JsonArray>>(v); }
    const JsonObject& obj() const { return *std::get<std::shared_ptr<JsonObject>>(v); }
    const Json& operator[](const std::string& k) const { return obj().at(k); }
    bool has(const std::string& k) const { return obj().count(k) != 0; }
};

class JsonParser
{
public:
    explicit JsonParser(const std::string& text) : s_(text) {}
    Json parse()
    {
        Json j = value();
        return j;
    }

private:
    const std::string& s_;
    size_t i_ = 0;
    void ws()
    {
        while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r'))
            i_++;
    }
    [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("json: ") + what); }
    Json value()
    {
        ws();
        if (i_ >= s_.size())
            fail("unexpected end");
        const char c = s_[i_];
        if (c == '{')
        {
            auto o = std::make_shared<JsonObject>();
            i_++;
            ws();
            if (s_[i_] == '}')
            {
                i_++;
                return Json{o};
            }
            for (;;)
            {
                ws();
                std::string k = str();
                ws();
                if (s_[i_++] != ':')
                    fail("expected ':'");
                (*o)[k] = value();
                ws();
                if (s_[i_] == ',')
                {
                    i_++;
                    continue;
                }
                if (s_[i_] == '}')
                {
                    i_++;
                    break;
                }
                fail("expected ',' or '}'");
            }
            return Json{o};
        }
        if (c == '[')
        {
            auto a = std::make_shared<JsonArray>();
            i_++;
            ws();
            if (s_[i_] == ']')
            {
                i_++;
                return Json{a};
            }
            for (;;)
            {
                a->push_back(value());
                ws();
                if (s_[i_] == ',')
                {
                    i_++;
                    continue;
                }
                if (s_[i_] == ']')
                {
                    i_++;
                    break;
                }
                fail("expected ',' or ']'");
            }
            return Json{a};
        }
        if (c == '"')
            return Json{str()};
        if (s_.compare(i_, 4, "true") == 0)
        {
            i_ += 4;
            return Json{1.0};
        }
        if (s_.compare(i_, 5, "false") == 0)
        {
            i_ += 5;
            return Json{0.0};
        }
        if (s_.compare(i_, 4, "null") == 0)
        {
            i_ += 4;
            return Json{nullptr};
        }
        char* end = nullptr;
        const double d = std::strtod(s_.c_str() + i_, &end);
        if (end == s_.c_str() + i_)
            fail("bad number");
        i_ = end - s_.c_str();
        return Json{d};
    }
    std::string str()
    {
        if (s_[i_] != '"')
            fail("expected string");
        std::string out;
        for (i_++; i_ < s_.size() && s_[i_] != '"'; i_++)
        {
            if (s_[i_] == '\\' && i_ + 1 < s_.size())
                i_++;
            out.push_back(s_[i_]);
        }
        i_++;
        return out;
    }
};

int main(int argc, char** argv)
{
    if (argc < 2)
    {
        std::fprintf(stderr, "usage: %s graph.json [iterations] [float32]\n", argv[0]);
        return 2;
    }
    const int iterations = argc > 2 ? std::atoi(argv[2]) : 10;
    // the reference selects 32-bit internal floats at build time (cmake -DUSE_FLOAT32=ON); here it
    // is an option of the optimiser
    const bool float32 = argc > 3 && std::string(argv[3]) == "float32";
    std::ifstream in(argv[1]);
    if (!in)
    {
        std::fprintf(stderr, "cannot open %s\n", argv[1]);
        return 2;
    }
    std::stringstream buf;
    buf << in.rdbuf();
    const std::string text = buf.str();
    const Json root = JsonParser(text).parse();

    cugo::GraphOptimisationOptions options;
    options.perEdgeInformation = true;
    options.perEdgeCamera = true;
    options.useFloat32 = float32;
    auto optimizer = std::make_unique<cugo::CudaGraphOptimisationImpl>(options);

    cugo::PoseVertexSet poses(false);
    cugo::LandmarkVertexSet landmarks(true);
    cugo::MonoEdgeSet monoEdges;
    cugo::StereoEdgeSet stereoEdges;
    std::deque<cugo::PoseVertex> poseStore;
    std::deque<cugo::LandmarkVertex> landmarkStore;
    std::deque<cugo::MonoEdge> monoStore;
    std::deque<cugo::StereoEdge> stereoStore;

    for (const Json& n : root["pose_vertices"].arr())
    {
        double q[4], t[3];
        for (int i = 0; i < 4; i++)
            q[i] = n["q"].arr()[i].num();
        for (int i = 0; i < 3; i++)
            t[i] = n["t"].arr()[i].num();
        poseStore.emplace_back((int)n["id"].num(), cugo::Se3D(q, t), n["fixed"].num() != 0);
        poses.addVertex(&poseStore.back());
    }
    for (const Json& n : root["landmark_vertices"].arr())
    {
        double x[3];
        for (int i = 0; i < 3; i++)
            x[i] = n["Xw"].arr()[i].num();
        landmarkStore.emplace_back((int)n["id"].num(), cugo::Vec3d(x), n["fixed"].num() != 0);
        landmarks.addVertex(&landmarkStore.back());
    }
    const cugo::Camera camera(root["fx"].num(), root["fy"].num(), root["cx"].num(), root["cy"].num(),
                              root["bf"].num());
    if (root.has("monocular_edges"))
        for (const Json& n : root["monocular_edges"].arr())
        {
            monoStore.emplace_back();
            cugo::MonoEdge& e = monoStore.back();
            e.setVertex(poses.getVertex((int)n["vertexP"].num()), 0);
            e.setVertex(landmarks.getVertex((int)n["vertexL"].num()), 1);
            e.setMeasurement(cugo::Vec2d(n["measurement"].arr()[0].num(), n["measurement"].arr()[1].num()));
            e.setInformation(n["information"].num());
            e.setCamera(camera);
            monoEdges.addEdge(&e);
        }
    if (root.has("stereo_edges"))
        for (const Json& n : root["stereo_edges"].arr())
        {
            stereoStore.emplace_back();
            cugo::StereoEdge& e = stereoStore.back();
            e.setVertex(poses.getVertex((int)n["vertexP"].num()), 0);
            e.setVertex(landmarks.getVertex((int)n["vertexL"].num()), 1);
            e.setMeasurement(cugo::Vec3d(n["measurement"].arr()[0].num(), n["measurement"].arr()[1].num(),
                                         n["measurement"].arr()[2].num()));
            e.setInformation(n["information"].num());
            e.setCamera(camera);
            stereoEdges.addEdge(&e);
        }

    optimizer->addVertexSet(&poses);
    optimizer->addVertexSet(&landmarks);
    optimizer->addEdgeSet(&monoEdges);
    optimizer->addEdgeSet(&stereoEdges);
    stereoEdges.setRobustKernel(cugo::RobustKernelType::None, 1.0);

    std::cout << "=== Graph size : \n";
    std::cout << "num poses      : " << optimizer->nVertices(0) << "\n";
    std::cout << "num landmarks  : " << optimizer->nVertices(1) << "\n";
    for (const auto* es : optimizer->getEdgeSets())
        std::cout << "num edges      : " << es->nedges() << "\n";

    // warm-up to avoid one-off overheads (mutates the estimates, exactly like the reference)
    optimizer->initialize();
    optimizer->optimize(1);

    const auto t0 = std::chrono::steady_clock::now();
    optimizer->initialize();
    optimizer->optimize(iterations);
    const auto t1 = std::chrono::steady_clock::now();

    std::cout << "=== Processing time : " << std::chrono::duration<double>(t1 - t0).count() << " [sec]\n";
    for (const auto& kv : optimizer->timeProfile())
        std::printf("%-32s : %8.3f [msec]\n", kv.first.c_str(), kv.second);
    std::cout << "=== Objective function value : \n";
    for (const auto& stat : optimizer->batchStatistics().get())
        std::printf("iter: %2d, chi2: %.1f\n", stat.iteration + 1, stat.chi2);
    return 0;
}
