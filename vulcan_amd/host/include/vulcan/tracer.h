// tracer.h — raycasts the volume into a depth / colour / normal frame
// (ref: include/vulcan/tracer.h). Trace() takes the fused bounds path of the C
// ABI; the protected ComputePatches / ComputeBounds / ComputePoints /
// ComputeNormals keep the reference's four-stage form.
#pragma once

#include <memory>
#include <vulcan/buffer.h>
#include <vulcan/matrix.h>

namespace vulcan
{

struct Frame;
class Volume;

struct Patch
{
  static const int max_size = 16;

  Vector2s origin;

  Vector2s size;

  Vector2f bounds;
};

class Tracer
{
  public:

    Tracer(std::shared_ptr<const Volume> volume);

    virtual ~Tracer() {}

    std::shared_ptr<const Volume> GetVolume() const;

    const Vector2f& GetDepthRange() const;

    void SetDepthRange(const Vector2f& range);

    void SetDepthRange(float min, float max);

    void Trace(Frame& frame);

  protected:

    void ComputePatches(const Frame& frame);

    void ComputeBounds(const Frame& frame);

    void ComputePoints(Frame& frame);

    void ComputeNormals(Frame& frame);

    void ResetBoundsBuffer();

    void ResetBufferSize();

    int GetBufferSize();

  private:

    void Initialize();

  protected:

    Buffer<Patch> patches_;

    Buffer<Vector2f> bounds_;   // bounds grid, followed by the fused pass's scratch

    Buffer<int> buffer_size_;

    std::shared_ptr<const Volume> volume_;

    Vector2f depth_range_;

    int bounds_width_;

    int bounds_height_;
};

} // namespace vulcan
