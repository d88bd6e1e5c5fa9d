// room_scene.h — the closed-form tracking scene of the demo loop (tests/scenes.py holds the same
// definition for the Python side): a 4.4 x 3 x 5.2 m box room with nine spheres on its floor,
// ceiling and walls, seen by a camera whose yaw swings +-24 degrees while its centre moves on a
// small closed curve. Three pairs of orthogonal planes plus curved objects make every one of the
// six pose parameters observable to point-to-plane ICP (the sphere-centred scene of the fusion
// benchmark is invariant under rotation and cannot be tracked). Ray-surface intersections are
// analytic (a slab test and one quadratic per sphere) in double precision, rounded once.
// No reference counterpart: the reference app reads a camera (apps/vulcan/vulcan.cu:181-296).
#pragma once

#include <cmath>
#include <vector>

#include <vulcan/vulcan.h>

namespace room
{

const double kHalf[3] = {2.2, 1.5, 2.6};                      // box half extents; +y is down
const double kSpheres[9][4] = {{1.2, 0.9, 1.6, 0.6}, {-1.4, 1.0, 1.2, 0.5}, {-1.0, 0.95, -1.7, 0.55},
    {1.5, 1.05, -1.3, 0.45}, {0.2, -0.9, 2.1, 0.5}, {-1.9, -0.2, -0.3, 0.5}, {1.9, 0.1, 0.2, 0.45},
    {0.1, 1.1, 2.2, 0.4}, {-0.3, 1.15, -2.2, 0.35}};

// depth-to-world pose of frame i
inline vulcan::Transform Pose(int i, int frames_per_cycle = 240, double yaw_amplitude_deg = 24.0, double pitch_deg = 12.0)
{
  const double phase = 2.0 * M_PI * i / frames_per_cycle;
  const double a = yaw_amplitude_deg * M_PI / 180.0 * std::sin(phase);
  const double p = pitch_deg * M_PI / 180.0;
  const vulcan::Transform yaw = vulcan::Transform::Rotate(float(std::cos(a / 2)), 0.0f, float(std::sin(a / 2)), 0.0f);
  const vulcan::Transform pitch = vulcan::Transform::Rotate(float(std::cos(p / 2)), float(std::sin(p / 2)), 0.0f, 0.0f);
  return vulcan::Transform::Translate(float(0.25 * std::sin(phase)), float(0.04 * std::sin(2.0 * phase)),
      float(0.25 * (std::cos(phase) - 1.0))) * yaw * pitch;
}

inline double Albedo(const double* x)
{
  return 0.5 + 0.15 * std::cos(5.0 * x[0] + 1.0) * std::cos(4.0 * x[2]) + 0.15 * std::cos(6.0 * x[1]);
}

// depth [w*h] and grey colour [w*h] (albedo x Light::GetShading of a lamp at `lamp`, camera frame,
// light.h:53-60; clamped into (.02, .98), the range LightIntegrator accepts) of the room from `pose`
inline void Render(const vulcan::Projection& k, const vulcan::Transform& pose, int w, int h, double intensity,
    const double* lamp, std::vector<float>& depth, std::vector<vulcan::Vector3f>& color)
{
  const vulcan::Matrix4f M = pose.GetMatrix();
  double R[3][3], pos[3];
  for (int r = 0; r < 3; ++r)
  {
    for (int c = 0; c < 3; ++c) R[r][c] = M(r, c);
    pos[r] = M(r, 3);
  }
  const vulcan::Vector2f f = k.GetFocalLength(), c0 = k.GetCenterPoint();
  depth.resize(size_t(w) * h);
  color.resize(size_t(w) * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
    {
      const double ray[3] = {(x + 0.5 - c0[0]) / f[0], (y + 0.5 - c0[1]) / f[1], 1.0};
      double d[3], n[3] = {0, 0, 0};
      for (int j = 0; j < 3; ++j) d[j] = R[j][0] * ray[0] + R[j][1] * ray[1] + R[j][2] * ray[2];
      double s = INFINITY;
      int axis = 0;
      for (int j = 0; j < 3; ++j)
      {
        if (d[j] == 0) continue;
        const double sj = ((d[j] > 0 ? 1.0 : -1.0) * kHalf[j] - pos[j]) / d[j];
        if (sj < s) { s = sj; axis = j; }
      }
      n[axis] = d[axis] > 0 ? -1.0 : 1.0;
      const double aa = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      for (const auto& sp : kSpheres)
      {
        const double oc[3] = {pos[0] - sp[0], pos[1] - sp[1], pos[2] - sp[2]};
        const double b = d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2];
        const double disc = b * b - aa * (oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - sp[3] * sp[3]);
        if (disc <= 0) continue;
        const double t = (-b - std::sqrt(disc)) / aa;
        if (t > 0 && t < s)
        {
          s = t;
          for (int j = 0; j < 3; ++j) n[j] = (oc[j] + t * d[j]) / sp[3];
        }
      }
      const double point[3] = {pos[0] + s * d[0], pos[1] + s * d[1], pos[2] + s * d[2]};
      double value = Albedo(point);
      if (intensity > 0)
      {
        double nc[3], to_light[3], dist2 = 0, cosine = 0;
        for (int j = 0; j < 3; ++j) nc[j] = R[0][j] * n[0] + R[1][j] * n[1] + R[2][j] * n[2];
        for (int j = 0; j < 3; ++j) { to_light[j] = lamp[j] - s * ray[j]; dist2 += to_light[j] * to_light[j]; }
        for (int j = 0; j < 3; ++j) cosine += nc[j] * to_light[j];
        cosine /= std::sqrt(dist2);
        value *= std::fmax(intensity * cosine / dist2, 0.0);
      }
      value = std::fmin(std::fmax(value, 0.03), 0.97);
      depth[size_t(y) * w + x] = float(s);
      color[size_t(y) * w + x] = vulcan::Vector3f(float(value), float(value), float(value));
    }
}

}  // namespace room
