// g2o-shaped graph objects of the cugo API: vertices, vertex sets, edges, edge sets.
// Class and method names follow the reference (ref: src/optimisable_graph.h:40-816,
// src/optimisable_graph.hpp) so ORB-SLAM2-style callers compile unchanged; the bodies are
// host-only containers — flattening to device arrays happens in the optimiser
// (csrc/host/graph_optimisation.cpp), nothing here touches the GPU runtime.
//
// Differences from the reference worth knowing:
//  * edge containers keep insertion order (the reference iterates an unordered_set of
//    pointers, so its device edge order — and fp64 atomic sums — vary run to run);
//  * robust-kernel parameters are per edge set, not a process-global device object.
#pragma once
#include <atomic>
#include <algorithm>
#include <array>
#include <cassert>
#include <cstdint>
#include <map>
#include <type_traits>
#include <utility>
#include <vector>

#include "cugo_types.h"

namespace cugo
{

class BaseEdge;
class BaseEdgeSet;

// The reference's edge container is `std::unordered_set<BaseEdge*>` (ref: src/optimisable_graph.h:34; handed out
// by Vertex::getEdges() :129 and EdgeSet::get() :753).  This one has the same member surface — find, count,
// insert -> pair<iterator, bool> (a set: a second insert of the same edge is refused), emplace, erase(key) -> count,
// erase(iterator) -> next, range insert, begin / end / cbegin / cend, size, empty, clear, reserve — so code written
// against the reference's type compiles and behaves the same, but it ITERATES IN INSERTION ORDER: the flattened
// edge order (and with it every floating-point sum) is then the same from run to run, which an unordered_set of
// pointers does not give.  Storage: the edges in a vector; from 32 elements on an open-addressing index (pointer ->
// position) answers find / count / the duplicate test of insert in O(1).  erase keeps the order (O(n) move) and
// re-indexes; the index is maintained by the mutating members only, so lookups are const in fact as well as in name.
class EdgeContainer
{
public:
    using key_type = BaseEdge*;
    using value_type = BaseEdge*;
    using size_type = std::size_t;
    using iterator = std::vector<BaseEdge*>::const_iterator; // (as in a set, elements are not assignable through iterators)
    using const_iterator = iterator;

    iterator begin() const noexcept { return v_.begin(); }
    iterator end() const noexcept { return v_.end(); }
    const_iterator cbegin() const noexcept { return v_.begin(); }
    const_iterator cend() const noexcept { return v_.end(); }
    size_type size() const noexcept { return v_.size(); }
    bool empty() const noexcept { return v_.empty(); }

    iterator find(BaseEdge* e) const
    {
        if (v_.size() < kIndexFrom)
            return std::find(v_.begin(), v_.end(), e);
        // (the index is kept by every mutating member: concurrent lookups on a const container are safe, as they
        // are on the reference's std::unordered_set)
        const std::size_t mask = tab_.size() - 1;
        for (std::size_t h = hash(e) & mask;; h = (h + 1) & mask)
        {
            const uint32_t q = tab_[h];
            if (q == 0)
                return v_.end();
            if (v_[q - 1] == e)
                return v_.begin() + (q - 1);
        }
    }
    size_type count(BaseEdge* e) const { return find(e) != v_.end() ? 1 : 0; }
    std::pair<iterator, bool> insert(BaseEdge* e)
    {
        const iterator it = find(e);
        if (it != v_.end())
            return {it, false};
        v_.push_back(e);
        if (v_.size() >= kIndexFrom)
        {
            if (tab_.empty() || 2 * v_.size() > tab_.size())
                rebuild();
            else
                put(e, (uint32_t)v_.size());
        }
        return {v_.end() - 1, true};
    }
    std::pair<iterator, bool> emplace(BaseEdge* e) { return insert(e); }
    template <typename It>
    void insert(It first, It last)
    {
        for (; first != last; ++first)
            insert(*first);
    }
    size_type erase(BaseEdge* e)
    {
        const iterator it = find(e);
        if (it == v_.end())
            return 0;
        erase(it);
        return 1;
    }
    iterator erase(iterator pos)
    {
        const std::size_t at = (std::size_t)(pos - v_.begin());
        v_.erase(pos); // (keeps the order: O(n), like the re-indexing of the positions behind it)
        if (v_.size() >= kIndexFrom)
            rebuild();
        else
            tab_.clear();
        return v_.begin() + at;
    }
    void clear() noexcept
    {
        v_.clear();
        tab_.clear();
    }
    void reserve(size_type n) { v_.reserve(n); }

private:
    static constexpr std::size_t kIndexFrom = 32;
    static std::size_t hash(const BaseEdge* e) noexcept
    {
        std::uint64_t x = (std::uint64_t) reinterpret_cast<std::uintptr_t>(e);
        x ^= x >> 33, x *= 0xff51afd7ed558ccdull, x ^= x >> 29;
        return (std::size_t)x;
    }
    void put(BaseEdge* e, uint32_t pos1)
    {
        const std::size_t mask = tab_.size() - 1;
        std::size_t h = hash(e) & mask;
        while (tab_[h] != 0)
            h = (h + 1) & mask;
        tab_[h] = pos1;
    }
    void rebuild()
    {
        std::size_t cap = 64;
        while (cap < 4 * v_.size()) // load factor between 1/4 and 1/2
            cap *= 2;
        tab_.assign(cap, 0);
        for (std::size_t i = 0; i < v_.size(); i++)
            put(v_[i], (uint32_t)(i + 1));
    }
    std::vector<BaseEdge*> v_;
    std::vector<uint32_t> tab_; // open addressing, linear probing: position + 1 of the edge, 0 = free; kept from 32 elements on
};

// ------------------------------------------------------------------ change tracking ----
// Extension (not in the reference): every vertex set and edge set counts the changes made to it or
// to its members — anything except a vertex ESTIMATE: adding / removing members, fixing a vertex,
// a new measurement / information / camera / vertex of an edge, (in)activating an edge, a set-wide
// information / camera / robust kernel / outlier threshold.  initialize() compares the counts with
// those of its last full flattening: if none moved, the graph on the device is still the graph in
// these objects and only the estimates are refreshed (csrc/host/graph_optimisation.cpp).  Members
// reach their set through an owner pointer that addVertex / addEdge set.
class ChangeCounted
{
public:
    // (relaxed atomic: members of one set may be mutated from several threads, as they could be before the counter existed)
    void touch() noexcept { changes_.fetch_add(1, std::memory_order_relaxed); }
    unsigned long long changeCount() const noexcept { return changes_.load(std::memory_order_relaxed); }

private:
    std::atomic<unsigned long long> changes_{0};
};
class BaseVertexSet;
class BaseEdgeSet;

// ------------------------------------------------------------------ vertices -----------
class BaseVertex
{
public:
    virtual ~BaseVertex() {}
    void setOwnerSet(BaseVertexSet* s) noexcept { owner_ = s; }
    BaseVertexSet* ownerSet() const noexcept { return owner_; }

protected:
    inline void touchOwner() noexcept; // defined behind BaseVertexSet
    BaseVertexSet* owner_ = nullptr;

public:
    virtual int getId() const noexcept = 0;
    virtual void setId(const int id) noexcept = 0;
    virtual EdgeContainer& getEdges() noexcept = 0;
    virtual void addEdge(BaseEdge* edge) = 0;
    virtual void removeEdge(BaseEdge* edge) = 0;
    virtual void setFixed(bool status) noexcept = 0;
    virtual bool isFixed() const noexcept = 0;
    virtual int getIndex() const noexcept = 0;
    virtual void setIndex(const int idx) noexcept = 0;
    virtual bool isMarginilised() const noexcept = 0;
    virtual void clearEdges() noexcept = 0;
};

// ref: Vertex<T, Marginilised> src/optimisable_graph.h:109-155
template <typename T, bool Marginilised>
class Vertex : public BaseVertex
{
public:
    using EstimateType = T;
    const bool marginilised = Marginilised;

    Vertex() {}
    Vertex(int id, const EstimateType& est, bool fixed = false) : estimate(est), fixed(fixed), id(id) {}
    Vertex(int id, bool fixed = false) : fixed(fixed), id(id) {}

    EstimateType& getEstimate() noexcept { return estimate; }
    const EstimateType& getEstimate() const noexcept { return estimate; }
    void setEstimate(const EstimateType& est) noexcept { estimate = est; }
    EdgeContainer& getEdges() noexcept override { return edges; }
    void addEdge(BaseEdge* e) override { edges.insert(e); }
    void removeEdge(BaseEdge* e) override { edges.erase(e); }
    void setFixed(bool s) noexcept override
    {
        if (s != fixed)
            touchOwner();
        fixed = s;
    }
    bool isFixed() const noexcept override { return fixed; }
    void setId(const int i) noexcept override
    {
        touchOwner();
        id = i;
    }
    int getId() const noexcept override { return id; }
    int getIndex() const noexcept override { return idx; }
    void setIndex(const int i) noexcept override { idx = i; }
    bool isMarginilised() const noexcept override { return Marginilised; }
    void clearEdges() noexcept override { edges.clear(); }

protected:
    EstimateType estimate{};
    bool fixed = false;
    int id = -1;
    int idx = -1; // device index, assigned by initialize(): free vertices first
    EdgeContainer edges;
};

using PoseVertex = Vertex<Se3D, false>;
using LandmarkVertex = Vertex<Vec3d, true>;

class BaseVertexSet : public ChangeCounted
{
public:
    virtual ~BaseVertexSet() {}
    virtual bool removeVertex(BaseVertex* v, BaseEdgeSet* edgeSet) = 0;
    virtual size_t size() const noexcept = 0;
    virtual size_t estimateDataSize() const noexcept = 0;
    virtual size_t getDeviceEstimateSize() noexcept = 0;
    virtual bool isMarginilised() const noexcept = 0;
    virtual int getActiveSize() const noexcept = 0;
    virtual void clearEstimates() noexcept = 0;
    virtual void clearVertices() noexcept = 0;
    // flattening hooks used by the optimiser (ascending id; free first, fixed after —
    // ref: generateEstimateData src/optimisable_graph.hpp:84-126)
    virtual int estimateDim() const noexcept = 0;
    virtual void assignIndices(int firstFree, int firstFixed, int& nFree, int& nFixed) = 0;
    virtual void gatherEstimates(double* out) const = 0;        // by index, estimateDim each
    virtual void scatterEstimates(const double* in) = 0;        // ref: finalise() hpp:137-154
    virtual int countFree() const noexcept = 0;
};

inline void BaseVertex::touchOwner() noexcept
{
    if (owner_)
        owner_->touch();
}

namespace detail
{
// the library's persistent host thread pool (csrc/host/thread_pool.cpp)
unsigned hostPoolThreads();
void hostPoolRun(unsigned chunks, void (*fn)(void*, unsigned), void* ctx);
} // namespace detail

// Runs f(begin, end, chunk) over [0, n) split into contiguous chunks on the library's host
// threads (one chunk, in the caller, for small n).  Returns the number of chunks (<= detail::hostPoolThreads()).  The
// vertex/edge walks of initialize()/optimize() are memory-latency bound pointer chasing: they
// scale with threads.
template <typename F>
inline unsigned parallelChunks(size_t n, F&& f)
{
    const unsigned nt = n < 32768 ? 1u : detail::hostPoolThreads();
    if (nt <= 1)
    {
        f((size_t)0, n, 0u);
        return 1;
    }
    struct Ctx
    {
        F* f;
        size_t n;
        unsigned nt;
    } ctx{&f, n, nt};
    detail::hostPoolRun(
        nt,
        [](void* p, unsigned c) {
            Ctx& x = *static_cast<Ctx*>(p);
            (*x.f)(x.n * c / x.nt, x.n * (c + 1) / x.nt, c);
        },
        &ctx);
    return nt;
}

// ref: VertexSet<T, E> src/optimisable_graph.h:224-332
template <typename T, typename E>
class VertexSet : public BaseVertexSet
{
public:
    using VertexType = T;
    using EstimateType = E;

    explicit VertexSet(bool marg) : marginilised(marg) {}

    void addVertex(T* vertex)
    {
        vertexMap.emplace(vertex->getId(), vertex);
        vertex->setOwnerSet(this);
        touch();
        byIdStale = true;
    }
    T* getVertex(const int id) const { return vertexMap.at(id); }
    bool removeVertex(BaseVertex* v, BaseEdgeSet* edgeSet) override;
    size_t size() const noexcept override { return vertexMap.size(); }
    bool isMarginilised() const noexcept override { return marginilised; }
    std::vector<T*>& get() noexcept { return vertices; } // index order after initialize()
    size_t estimateDataSize() const noexcept override { return vertices.size(); }
    size_t getDeviceEstimateSize() noexcept override { return vertices.size(); }
    int getActiveSize() const noexcept override { return activeSize; }
    void clearEstimates() noexcept override
    {
        vertices.clear();
        activeSize = 0;
    }
    void clearVertices() noexcept override
    {
        for (auto& kv : vertexMap)
            kv.second->setOwnerSet(nullptr);
        vertexMap.clear();
        touch();
        byIdStale = true;
    }

    int estimateDim() const noexcept override { return (int)(sizeof(E) / sizeof(double)); }
    int countFree() const noexcept override
    {
        const std::vector<T*>& ids = idOrder();
        std::vector<int> part(detail::hostPoolThreads(), 0);
        parallelChunks(ids.size(), [&](size_t a, size_t b, unsigned t) {
            int c = 0;
            for (size_t i = a; i < b; i++)
                c += !ids[i]->isFixed();
            part[t] = c;
        });
        int c = 0;
        for (int v : part)
            c += v;
        return c;
    }
    // free vertices first, fixed after, both in ascending id; chunks of the id order are
    // partitioned independently once the per-chunk free/fixed counts are known
    void assignIndices(int firstFree, int firstFixed, int& nFree, int& nFixed) override
    {
        const std::vector<T*>& ids = idOrder();
        const size_t total = ids.size();
        std::vector<int> cfree(detail::hostPoolThreads() + 1, 0), cfix(detail::hostPoolThreads() + 1, 0);
        const unsigned nt = parallelChunks(total, [&](size_t a, size_t b, unsigned t) {
            int c = 0;
            for (size_t i = a; i < b; i++)
                c += !ids[i]->isFixed();
            cfree[t + 1] = c;
            cfix[t + 1] = (int)(b - a) - c;
        });
        for (unsigned t = 0; t < nt; t++)
            cfree[t + 1] += cfree[t], cfix[t + 1] += cfix[t];
        const int nfree = cfree[nt];
        vertices.assign(total, nullptr);
        parallelChunks(total, [&](size_t a, size_t b, unsigned t) {
            int f = cfree[t], x = cfix[t];
            for (size_t i = a; i < b; i++)
            {
                T* v = ids[i];
                if (!v->isFixed())
                {
                    v->setIndex(firstFree + f);
                    vertices[f++] = v;
                }
                else
                {
                    v->setIndex(firstFixed + x);
                    vertices[nfree + x++] = v;
                }
            }
        });
        activeSize = nfree;
        nFree = nfree;
        nFixed = (int)total - nfree;
    }
    void gatherEstimates(double* out) const override
    {
        static_assert(std::is_trivially_copyable<E>::value, "estimate must be POD");
        const int d = (int)(sizeof(E) / sizeof(double));
        parallelChunks(vertices.size(), [&](size_t a, size_t b, unsigned) {
            for (size_t i = a; i < b; i++)
            {
                const T* v = vertices[i];
                const E& e = v->getEstimate();
                const double* src = reinterpret_cast<const double*>(&e);
                std::copy(src, src + d, out + (size_t)v->getIndex() * d);
            }
        });
    }
    void scatterEstimates(const double* in) override
    {
        const int d = (int)(sizeof(E) / sizeof(double));
        parallelChunks(vertices.size(), [&](size_t a, size_t b, unsigned) {
            for (size_t i = a; i < b; i++)
            {
                T* v = vertices[i];
                E e;
                std::copy(in + (size_t)v->getIndex() * d, in + (size_t)(v->getIndex() + 1) * d,
                          reinterpret_cast<double*>(&e));
                v->setEstimate(e);
            }
        });
    }

protected:
    // the vertices in ascending id as a flat array: walking the std::map (one cache miss per
    // node) is paid once after the set changed, not on every initialize()
    const std::vector<T*>& idOrder() const
    {
        if (byIdStale)
        {
            byId.clear();
            byId.reserve(vertexMap.size());
            for (const auto& kv : vertexMap)
                byId.push_back(kv.second);
            byIdStale = false;
        }
        return byId;
    }

    std::map<int, VertexType*> vertexMap; // ascending id
    bool marginilised;
    std::vector<VertexType*> vertices;
    mutable std::vector<VertexType*> byId;
    mutable bool byIdStale = true;
    int activeSize = 0;
};

using PoseVertexSet = VertexSet<PoseVertex, Se3D>;
using LandmarkVertexSet = VertexSet<LandmarkVertex, Vec3d>;

// ------------------------------------------------------------------ edges --------------
class BaseEdge
{
public:
    using Information = Scalar;
    virtual ~BaseEdge() {}
    void setOwnerSet(BaseEdgeSet* s) noexcept { owner_ = s; }
    BaseEdgeSet* ownerSet() const noexcept { return owner_; }

protected:
    inline void touchOwner() noexcept; // defined behind BaseEdgeSet
    BaseEdgeSet* owner_ = nullptr;

public:
    virtual BaseVertex* getVertex(const int index) = 0;
    virtual void setVertex(BaseVertex* vertex, const int index) = 0;
    virtual bool allVerticesFixed() const noexcept = 0;
    virtual bool anyVerticesNotFixed() const noexcept = 0;
    virtual bool allVerticesNotFixed() const noexcept = 0;
    virtual void* getMeasurement() noexcept { return nullptr; }
    // read-only views for the optimiser's flattening (the accessors of the reference's API hand out
    // mutable references / pointers, so they count as a change of the edge)
    virtual const void* measurementData() const noexcept { return nullptr; }
    virtual const Camera& cameraData() const noexcept = 0;
    virtual Information informationValue() const noexcept = 0;
    virtual int dim() const noexcept = 0;
    virtual void setInformation(const Information info) noexcept = 0;
    virtual Information getInformation() noexcept = 0;
    virtual void setCamera(const Camera& camera) noexcept = 0;
    virtual Camera& getCamera() noexcept = 0;
    virtual void inactivate() noexcept = 0;
    virtual void setActive() noexcept = 0;
    virtual bool isActive() const noexcept = 0;
};

// ref: Edge<DIM, E, VertexTypes...> src/optimisable_graph.h:416-503
template <int DIM, typename E, typename... VertexTypes>
class Edge : public BaseEdge
{
public:
    using Measurement = E;
    static constexpr auto VertexSize = sizeof...(VertexTypes);

    Edge() : measurement(Measurement()), info_(0), isActive_(true)
    {
        for (auto& v : vertices)
            v = nullptr;
    }

    BaseVertex* getVertex(const int index) override { return vertices[index]; }
    void setVertex(BaseVertex* vertex, const int index) override
    {
        touchOwner();
        vertices[index] = vertex;
    }
    bool allVerticesFixed() const noexcept override
    {
        for (auto* v : vertices)
            if (!v->isFixed())
                return false;
        return true;
    }
    bool anyVerticesNotFixed() const noexcept override { return !allVerticesFixed(); }
    bool allVerticesNotFixed() const noexcept override
    {
        for (auto* v : vertices)
            if (v->isFixed())
                return false;
        return true;
    }
    int dim() const noexcept override { return DIM; }
    void setMeasurement(const Measurement& m) noexcept
    {
        touchOwner();
        measurement = m;
    }
    void setInformation(const Information info) noexcept override
    {
        touchOwner();
        info_ = info;
    }
    Information getInformation() noexcept override { return info_; }
    Information informationValue() const noexcept override { return info_; }
    void setCamera(const Camera& camera) noexcept override
    {
        touchOwner();
        camera_ = camera;
    }
    Camera& getCamera() noexcept override
    {
        touchOwner(); // a mutable reference leaves the object: assume it is written through
        return camera_;
    }
    const Camera& cameraData() const noexcept override { return camera_; }
    void inactivate() noexcept override
    {
        touchOwner();
        isActive_ = false;
    }
    void setActive() noexcept override
    {
        touchOwner();
        isActive_ = true;
    }
    bool isActive() const noexcept override { return isActive_; }

protected:
    Measurement measurement;
    Information info_;
    Camera camera_;
    BaseVertex* vertices[VertexSize];
    bool isActive_;
};

class BaseEdgeSet : public ChangeCounted
{
public:
    using Information = Scalar;
    virtual ~BaseEdgeSet() {}
    virtual void addEdge(BaseEdge* edge) = 0;
    virtual void removeEdge(BaseEdge* edge) = 0;
    virtual size_t nedges() const noexcept = 0;
    virtual size_t nActiveEdges() const noexcept = 0;
    virtual const EdgeContainer& get() noexcept = 0;
    virtual const int dim() const noexcept = 0;
    virtual void clearEdges() noexcept = 0;
    virtual void setRobustKernel(const RobustKernelType type, Scalar delta) noexcept = 0;
    virtual RobustKernel& getRobustKernel() noexcept = 0;
    // read-only views for the optimiser's flattening (the reference's getters above and below hand
    // out mutable references and therefore count as a change of the set)
    virtual const RobustKernel& robustKernelData() const noexcept = 0;
    virtual const Camera& cameraData() const noexcept = 0;
    virtual Information informationValue() const noexcept = 0;
    virtual void setInformation(const Information info) noexcept = 0;
    virtual Information getInformation() noexcept = 0;
    virtual void setCamera(const Camera& camera) noexcept = 0;
    virtual Camera& getCamera() noexcept = 0;
    virtual Scalar getOutlierThreshold() const noexcept = 0;
    virtual uint32_t getOutlierCount() const noexcept = 0;
    virtual uint32_t getInlierCount() const noexcept = 0;
    virtual bool isDirty() const noexcept = 0;
    virtual void setDirtyState(bool state) noexcept = 0;
    // set by the optimiser at initialize(): number of edges with at least one free vertex
    virtual void setActiveEdgeCount(size_t n) noexcept = 0;
    // set by the optimiser: outliers found since initialize() (ref: currOutlierCount_)
    virtual void setOutlierCount(uint32_t n) noexcept = 0;
};

inline void BaseEdge::touchOwner() noexcept
{
    if (owner_)
        owner_->touch();
}

// ref: EdgeSet<DIM, E, VertexTypes...> src/optimisable_graph.h:688-816
template <int DIM, typename E, typename... VertexTypes>
class EdgeSet : public BaseEdgeSet
{
public:
    using MeasurementType = E;
    static constexpr auto VertexSize = sizeof...(VertexTypes);

    void addEdge(BaseEdge* edge) override
    {
        for (int i = 0; i < (int)VertexSize; ++i)
            edge->getVertex(i)->addEdge(edge);
        edges.insert(edge);
        edge->setOwnerSet(this);
        touch();
        isDirty_ = true;
    }
    void removeEdge(BaseEdge* edge) override
    {
        for (int i = 0; i < (int)VertexSize; ++i)
            edge->getVertex(i)->removeEdge(edge);
        edges.erase(edge);
        edge->setOwnerSet(nullptr);
        touch();
        isDirty_ = true;
    }
    size_t nedges() const noexcept override { return edges.size(); }
    size_t nActiveEdges() const noexcept override { return activeEdgeSize_; }
    const EdgeContainer& get() noexcept override { return edges; }
    const int dim() const noexcept override { return DIM; }
    void setRobustKernel(const RobustKernelType type, Scalar delta) noexcept override
    {
        touch();
        kernel.create(type, delta);
    }
    RobustKernel& getRobustKernel() noexcept override
    {
        touch();
        return kernel;
    }
    const RobustKernel& robustKernelData() const noexcept override { return kernel; }
    const Camera& cameraData() const noexcept override { return camera_; }
    Information informationValue() const noexcept override { return info_; }
    void clearEdges() noexcept override
    {
        for (BaseEdge* e : edges)
            e->setOwnerSet(nullptr);
        edges.clear();
        activeEdgeSize_ = 0;
        touch();
        isDirty_ = true;
    }
    void setInformation(const Information info) noexcept override
    {
        touch();
        info_ = info;
    }
    Information getInformation() noexcept override { return info_; }
    void setCamera(const Camera& camera) noexcept override
    {
        touch();
        camera_ = camera;
    }
    Camera& getCamera() noexcept override
    {
        touch();
        return camera_;
    }
    void setOutlierThreshold(const Scalar t) noexcept
    {
        touch();
        outlierThreshold = t;
    }
    Scalar getOutlierThreshold() const noexcept override { return outlierThreshold; }
    uint32_t getOutlierCount() const noexcept override { return currOutlierCount_; }
    uint32_t getInlierCount() const noexcept override { return (uint32_t)activeEdgeSize_ - currOutlierCount_; }
    bool isDirty() const noexcept override { return isDirty_; }
    void setDirtyState(bool state) noexcept override { isDirty_ = state; }
    void setActiveEdgeCount(size_t n) noexcept override { activeEdgeSize_ = n; }
    void setOutlierCount(uint32_t n) noexcept override { currOutlierCount_ = n; }

protected:
    EdgeContainer edges;
    size_t activeEdgeSize_ = 0;
    RobustKernel kernel;
    // chi2 threshold above which an edge is inactivated at the end of optimize(); 0 = disabled
    // (ref: EdgeSet::updateEdges, src/optimisable_graph.hpp:603-640)
    Scalar outlierThreshold = 0.0;
    uint32_t currOutlierCount_ = 0;
    Information info_ = 0.0;
    Camera camera_;
    bool isDirty_ = true;
};

template <typename T, typename E>
bool VertexSet<T, E>::removeVertex(BaseVertex* v, BaseEdgeSet* edgeSet)
{
    auto it = vertexMap.find(v->getId());
    if (it == vertexMap.end())
        return false;
    const std::vector<BaseEdge*> es(it->second->getEdges().begin(), it->second->getEdges().end());
    for (BaseEdge* e : es)
        edgeSet->removeEdge(e);
    it->second->setOwnerSet(nullptr);
    vertexMap.erase(it);
    touch();
    byIdStale = true;
    return true;
}

} // namespace cugo
