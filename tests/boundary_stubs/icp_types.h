// Test-only stand-in for the reference's include/icp_types.h (LiDAR point-to-line / point-to-plane
// edges: out of scope here).  The reference's sample constructs and deletes one
// cugo::PlaneEdgeSet without ever adding it to the optimiser; this is just enough for
// tests/test_boundary.py to type-check that sample against the mirrored BA headers.
#pragma once
namespace cugo
{
class PlaneEdgeSet
{
};
} // namespace cugo
