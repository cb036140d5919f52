// Bundle-adjustment edge types of the cugo API (ref: include/ba_types.h:34-169, 238-254).
// g2o mapping: MonoEdge <-> EdgeSE3ProjectXYZ, StereoEdge <-> EdgeStereoSE3ProjectXYZ;
// vertex 0 = pose (VertexSE3Expmap), vertex 1 = landmark (VertexSBAPointXYZ).
// The reference's per-set virtual computeError / constructQuadraticForm dispatch into
// gpu::computeActiveErrors_<M> / constructQuadraticForm_<M>; here all sets of an optimiser
// are flattened into ONE landmark-major edge array (mono and stereo distinguished by a flag
// bit), so the sets are plain containers.
#pragma once
#include "optimisable_graph.h"

namespace cugo
{

/** monocular observation: measurement (u, v) */
class CUGO_API MonoEdge : public Edge<2, Vec2d, PoseVertex, LandmarkVertex>
{
public:
    void* getMeasurement() noexcept override
    {
        touchOwner(); // mutable pointer: counts as a change (optimisable_graph.h, change tracking)
        return static_cast<void*>(measurement.data);
    }
    const void* measurementData() const noexcept override { return static_cast<const void*>(measurement.data); }
};

/** stereo observation: measurement (u, v, u_right) */
class CUGO_API StereoEdge : public Edge<3, Vec3d, PoseVertex, LandmarkVertex>
{
public:
    void* getMeasurement() noexcept override
    {
        touchOwner();
        return static_cast<void*>(measurement.data);
    }
    const void* measurementData() const noexcept override { return static_cast<const void*>(measurement.data); }
};

class CUGO_API MonoEdgeSet : public EdgeSet<2, Vec2d, PoseVertex, LandmarkVertex>
{
};

class CUGO_API StereoEdgeSet : public EdgeSet<3, Vec3d, PoseVertex, LandmarkVertex>
{
};

} // namespace cugo
