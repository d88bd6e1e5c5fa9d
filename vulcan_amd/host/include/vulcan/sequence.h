// sequence.h — RGB-D sequences on disk: what stands in for the HAL camera the upstream app
// reads from (ref: apps/vulcan/vulcan.cu:181-232) and for Image::Load / Save through OpenCV
// (ref: include/vulcan/image.h:100-133,228-253). A sequence is a directory of
// depth_%06d.pgm (binary PGM, 16 bit, depth / depth_scale), color_%06d.ppm (binary PPM, 8 bit)
// and sequence.txt (size, intrinsics, depth scale, one row-major depth_to_world per frame).
#pragma once

#include <string>
#include <vector>
#include <vulcan/frame.h>

namespace vulcan
{

class SequenceWriter
{
  public:

    SequenceWriter(const std::string& directory, int width, int height, const Projection& depth_projection,
        const Projection& color_projection, float depth_scale = 0.001f);

    ~SequenceWriter();

    // writes the frame's depth (and colour, when present) image and records its pose
    void Append(const Frame& frame);

    void Close();

    int GetFrameCount() const { return count_; }

  protected:

    std::string directory_;
    std::vector<std::string> lines_;
    float depth_scale_;
    int count_;
    bool closed_;
};

class SequenceReader
{
  public:

    explicit SequenceReader(const std::string& directory);

    int GetFrameCount() const { return count_; }
    int GetWidth() const { return width_; }
    int GetHeight() const { return height_; }

    // depth (metres), colour ([0, 1], when the file exists), intrinsics and pose of frame i;
    // normals are left to Frame::ComputeNormals
    void Read(int index, Frame& frame) const;

  protected:

    std::string directory_;
    int width_, height_, count_;
    float depth_scale_;
    Projection depth_projection_, color_projection_;
    std::vector<Transform> poses_;
};

} // namespace vulcan
