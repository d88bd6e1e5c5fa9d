// meshing.h — the surface as a triangle mesh: Mesh / DeviceMesh (ref: include/vulcan/mesh.h),
// Extractor (ref: include/vulcan/extractor.h:116-134) and Exporter (ref: include/vulcan/
// exporter.h). Upstream's BlockExtractor (extractor.h:46-114: one block per launch, blocking
// copies between its stages, no faces) has no counterpart: the whole volume is one call into
// the C ABI (vk_extract_mesh) whose only readback is the four totals.
#pragma once

#include <memory>
#include <string>
#include <vector>
#include <vk.h>
#include <vulcan/buffer.h>
#include <vulcan/matrix.h>

namespace vulcan
{

class Volume;

struct Mesh          // mesh.h:9-14
{
  std::vector<Vector3f> points;
  std::vector<Vector3i> faces;
};

struct DeviceMesh    // mesh.h:16-21
{
  Buffer<Vector3f> points;
  Buffer<Vector3i> faces;
};

class Extractor
{
  public:

    explicit Extractor(std::shared_ptr<const Volume> volume);

    std::shared_ptr<const Volume> GetVolume() const;

    // upstream walks the visible blocks ("TODO: replace with all allocated blocks",
    // extractor.cu:455-457); true walks every allocated block
    bool GetAllAllocated() const { return all_allocated_; }
    void SetAllAllocated(bool all) { all_allocated_ = all; }

    // false: vertices at edge midpoints (extractor.cu:361); true (default): where the linearly
    // interpolated distance is zero
    bool GetInterpolate() const { return interpolate_; }
    void SetInterpolate(bool on) { interpolate_ = on; }

    void Extract(DeviceMesh& mesh) const;

    void Extract(Mesh& mesh) const;

    // cubes left out by the last Extract because a vertex they need belongs to a block that is
    // not in the list (visible-list mode only)
    int GetSkippedCubes() const { return skipped_; }

  protected:

    // extractor.cu:700-716 bound: 3 points per voxel, 5 faces per cube of every listed block
    void ResizeMesh(DeviceMesh& mesh) const;

    std::shared_ptr<const Volume> volume_;
    bool all_allocated_;
    bool interpolate_;
    mutable int skipped_;
    mutable Buffer<unsigned char> workspace_;
    mutable Buffer<int> counts_;
};

class Exporter
{
  public:

    explicit Exporter(const std::string& file);

    const std::string& GetFile() const;

    // ASCII PLY, byte-compatible with src/exporter.cpp:19-71
    void Export(const Mesh& mesh) const;

  protected:

    std::string file_;
};

} // namespace vulcan
