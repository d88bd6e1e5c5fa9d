// transform.h — rigid transform as a 4x4 matrix plus its cached inverse
// (ref: include/vulcan/transform.h). The inverse is carried, never computed:
// products multiply the inverses in reverse order, Inverse() swaps the pair.
// ToVk()/FromVk() convert to the C ABI's vk_transform (identical layout).
#pragma once

#include <vk.h>
#include <vulcan/matrix.h>

namespace vulcan
{

class Transform
{
  public:

    Transform() : matrix_(Matrix4f::Identity()), inv_matrix_(Matrix4f::Identity()) {}

    const Matrix4f& GetMatrix() const { return matrix_; }

    const Matrix4f& GetInverseMatrix() const { return inv_matrix_; }

    Vector3f GetTranslation() const
    {
      return Vector3f(matrix_(0, 3), matrix_(1, 3), matrix_(2, 3));
    }

    // rows 0..2 of the matrix applied to (x, y, z, w); w is passed through
    Vector4f operator*(const Vector4f& p) const
    {
      const Matrix4f& A = matrix_;
      Vector4f result;
      for (int r = 0; r < 3; ++r)
        result[r] = A(r, 0) * p[0] + A(r, 1) * p[1] + A(r, 2) * p[2] + A(r, 3) * p[3];
      result[3] = p[3];
      return result;
    }

    Transform operator*(const Transform& rhs) const
    {
      return Transform(matrix_ * rhs.matrix_, rhs.inv_matrix_ * inv_matrix_);
    }

    Transform Inverse() const { return Transform(inv_matrix_, matrix_); }

    static Transform Rotate(const Matrix3f& R)
    {
      Matrix4f matrix = Matrix4f::Identity();
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) matrix(r, c) = R(r, c);
      return Transform(matrix, matrix.Transpose());
    }

    static Transform Rotate(const Vector4f& q) { return Rotate(q[0], q[1], q[2], q[3]); }

    // unit quaternion (w, x, y, z)
    static Transform Rotate(float w, float x, float y, float z)
    {
      Matrix4f matrix = Matrix4f::Zeros();
      matrix(0, 0) = 1 - 2 * (y * y + z * z);
      matrix(0, 1) = 2 * (x * y - w * z);
      matrix(0, 2) = 2 * (x * z + w * y);
      matrix(1, 0) = 2 * (x * y + w * z);
      matrix(1, 1) = 1 - 2 * (x * x + z * z);
      matrix(1, 2) = 2 * (y * z - w * x);
      matrix(2, 0) = 2 * (x * z - w * y);
      matrix(2, 1) = 2 * (y * z + w * x);
      matrix(2, 2) = 1 - 2 * (x * x + y * y);
      matrix(3, 3) = 1.0f;
      return Transform(matrix, matrix.Transpose());
    }

    static Transform Translate(const Vector3f& t) { return Translate(t[0], t[1], t[2]); }

    static Transform Translate(float x, float y, float z)
    {
      Matrix4f matrix = Matrix4f::Identity();
      Matrix4f inverse = Matrix4f::Identity();
      const float t[3] = { x, y, z };
      for (int r = 0; r < 3; ++r)
      {
        matrix(r, 3) = t[r];
        inverse(r, 3) = -t[r];
      }
      return Transform(matrix, inverse);
    }

    vk_transform ToVk() const
    {
      vk_transform out;
      for (int i = 0; i < 16; ++i)
      {
        out.m[i] = matrix_.GetData()[i];
        out.inv[i] = inv_matrix_.GetData()[i];
      }
      return out;
    }

    static Transform FromVk(const vk_transform& in)
    {
      Matrix4f matrix, inverse;
      for (int i = 0; i < 16; ++i)
      {
        matrix.GetData()[i] = in.m[i];
        inverse.GetData()[i] = in.inv[i];
      }
      return Transform(matrix, inverse);
    }

  protected:

    Transform(const Matrix4f& matrix, const Matrix4f& inv_matrix) :
      matrix_(matrix), inv_matrix_(inv_matrix)
    {
    }

  protected:

    Matrix4f matrix_;

    Matrix4f inv_matrix_;
};

} // namespace vulcan
