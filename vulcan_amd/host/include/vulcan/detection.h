// detection.h — box detector over a point cloud: keep the points inside a
// sphere / three intervals, drop those further than 1.5 sigma from the centroid,
// report the centroid of what is left.
//
// API parity: class Detector keeps the public and protected names of the
// reference's detector.h:10-72 (which remains as a forwarder). Upstream needs a
// cuBLAS handle and reads a count back after each of its two compactions; here
// the whole of Detect() is enqueued as one call (vk_detect) and the only
// readback is the 48-byte result record.
#pragma once

#include <vk.h>
#include <vulcan/buffer.h>
#include <vulcan/matrix.h>

namespace vulcan
{

class Detector
{
  public:
    Detector();
    virtual ~Detector();

    // points further than this from the origin are ignored (<= 0: no test)
    float GetRadius() const;
    void SetRadius(float radius);
    const Vector3f& GetOrigin() const;
    void SetOrigin(const Vector3f& origin);

    // accepted interval per axis; (lo > hi) leaves the axis open. As upstream,
    // ALL three intervals are compared with the point's x coordinate
    // (detector.cu:26-28) unless SetBoundsUseOwnAxis(true).
    const Vector2f& GetBounds(int axis) const;
    void SetBounds(int axis, const Vector2f& bounds);
    bool GetBoundsUseOwnAxis() const;
    void SetBoundsUseOwnAxis(bool enabled);

    // fewer surviving points than this => Detect returns NaNs
    int GetMinInlierCount() const;
    void SetMinInlierCount(int count);

    // position of the box (mean of |x|, |y|, |z| over the inliers, as upstream)
    Vector3f Detect(const Buffer<Vector3f>& points);

    // what the last Filter/Detect left on the device
    const vk_detect_state& GetState() const;
    const Buffer<Vector3f>& GetInliers() const;

  protected:
    void Filter(const Buffer<Vector3f>& points);
    bool BoxDetected() const;
    Vector3f GetValidPosition() const;
    Vector3f GetInvalidPosition() const;
    int GetBufferSize() const;

    void Prepare(const Buffer<Vector3f>& points);
    void ReadState();

    vk_detector params_;
    vk_detect_state result_;
    Vector3f origin_;
    Vector2f bounds_[3];
    Buffer<Vector3f> points_;            // inliers, in input order
    Buffer<vk_detect_state> state_;      // device copy of result_
    Buffer<unsigned char> workspace_;
};

} // namespace vulcan
