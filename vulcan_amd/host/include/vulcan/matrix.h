// matrix.h — fixed-size column-major matrices and vectors on the host.
// Same public surface and the same float evaluation order as the reference's
// Matrix<T,M,N> (ref: include/vulcan/matrix.h): element (r,c) lives at
// data_[c*M + r]; Dot() starts from 0 and adds in index order and returns
// float; operator/(s) multiplies by 1.0f/s; Normalize() multiplies by 1/Norm().
#pragma once

#include <iomanip>
#include <ostream>
#include <vulcan/exception.h>
#include <vulcan/math.h>

namespace vulcan
{

template <typename T, int M, int N>
class Matrix
{
  static_assert(M > 0 && N > 0, "invalid matrix size");
  static constexpr int kTotal = M * N;
  static constexpr bool kIsVector = (M == 1 || N == 1);

  public:

    Matrix() {}

    Matrix(T v0, T v1) : data_{v0, v1}
    {
      static_assert(kTotal == 2, "2D vector required");
    }

    Matrix(T v0, T v1, T v2) : data_{v0, v1, v2}
    {
      static_assert(kTotal == 3, "3D vector required");
    }

    Matrix(T v0, T v1, T v2, T v3) : data_{v0, v1, v2, v3}
    {
      static_assert(kTotal == 4 && kIsVector, "4D vector required");
    }

    // element-wise conversion from another scalar type
    template <typename U>
    explicit Matrix(const Matrix<U, M, N>& other)
    {
      for (int c = 0; c < N; ++c)
        for (int r = 0; r < M; ++r) (*this)(r, c) = T(other(r, c));
    }

    // vector resize: copy the common prefix, zero-fill the rest
    template <int P, int Q>
    explicit Matrix(const Matrix<T, P, Q>& other)
    {
      static_assert(kIsVector && (P == 1 || Q == 1), "vector required");
      const int common = (kTotal < P * Q) ? kTotal : P * Q;
      for (int i = 0; i < common; ++i) data_[i] = other[i];
      for (int i = common; i < kTotal; ++i) data_[i] = T(0);
    }

    // vector extension by one trailing element, e.g. Vector4f(Vector3f, w)
    template <int P, int Q>
    explicit Matrix(const Matrix<T, P, Q>& other, T last)
    {
      static_assert(kIsVector && (P == 1 || Q == 1), "vector required");
      static_assert(kTotal == P * Q + 1, "invalid vector length");
      for (int i = 0; i < P * Q; ++i) data_[i] = other[i];
      data_[kTotal - 1] = last;
    }

    int GetRows() const { return M; }

    int GetColumns() const { return N; }

    int GetTotal() const { return kTotal; }

    T SquaredNorm() const { return this->Dot(*this); }

    T Norm() const
    {
      const float squared = SquaredNorm();
      VULCAN_DEBUG(squared > 0);
      return sqrt(squared);
    }

    void Normalize()
    {
      const T scale = T(1) / Norm();
      for (T& v : data_) v *= scale;
    }

    Matrix Normalized() const
    {
      Matrix result(*this);
      result.Normalize();
      return result;
    }

    float Dot(const Matrix& rhs) const
    {
      static_assert(kIsVector, "vector required");
      float sum = 0;
      for (int i = 0; i < kTotal; ++i) sum += data_[i] * rhs.data_[i];
      return sum;
    }

    Matrix Cross(const Matrix& rhs) const
    {
      static_assert(kTotal == 3, "3D vector required");
      const T* a = data_;
      const T* b = rhs.data_;
      Matrix result;
      result[0] = (a[1] * b[2]) - (a[2] * b[1]);
      result[1] = (a[2] * b[0]) - (a[0] * b[2]);
      result[2] = (a[0] * b[1]) - (a[1] * b[0]);
      return result;
    }

    Matrix<T, N, M> Transpose() const
    {
      Matrix<T, N, M> result;
      for (int c = 0; c < N; ++c)
        for (int r = 0; r < M; ++r) result(c, r) = (*this)(r, c);
      return result;
    }

    Matrix& operator+=(const Matrix& rhs)
    {
      for (int i = 0; i < kTotal; ++i) data_[i] += rhs.data_[i];
      return *this;
    }

    Matrix& operator-=(const Matrix& rhs)
    {
      for (int i = 0; i < kTotal; ++i) data_[i] -= rhs.data_[i];
      return *this;
    }

    template <typename S> Matrix& operator+=(S scalar)
    {
      for (T& v : data_) v += scalar;
      return *this;
    }

    template <typename S> Matrix& operator*=(S scalar)
    {
      for (T& v : data_) v *= scalar;
      return *this;
    }

    template <typename S> Matrix& operator/=(S scalar)
    {
      VULCAN_DEBUG(scalar != S(0));
      const float reciprocal = 1.0f / scalar;
      return (*this) *= reciprocal;
    }

    const Matrix operator+(const Matrix& rhs) const { return Matrix(*this) += rhs; }

    const Matrix operator-(const Matrix& rhs) const { return Matrix(*this) -= rhs; }

    template <typename S> const Matrix operator+(S scalar) const { return Matrix(*this) += scalar; }

    template <typename S> const Matrix operator*(S scalar) const { return Matrix(*this) *= scalar; }

    template <typename S> const Matrix operator/(S scalar) const { return Matrix(*this) /= scalar; }

    // (M x N) * (N x P): each element accumulates from 0 over n ascending
    template <int P>
    const Matrix<T, M, P> operator*(const Matrix<T, N, P>& rhs) const
    {
      Matrix<T, M, P> result;
      for (int p = 0; p < P; ++p)
        for (int m = 0; m < M; ++m)
        {
          result(m, p) = 0;
          for (int n = 0; n < N; ++n) result(m, p) += (*this)(m, n) * rhs(n, p);
        }
      return result;
    }

    const T& operator()(int row, int col) const
    {
      VULCAN_DEBUG_MSG(row >= 0 && row < M && col >= 0 && col < N, "index out of bounds");
      return data_[col * M + row];
    }

    T& operator()(int row, int col)
    {
      VULCAN_DEBUG_MSG(row >= 0 && row < M && col >= 0 && col < N, "index out of bounds");
      return data_[col * M + row];
    }

    const T& operator[](int index) const
    {
      static_assert(kIsVector, "vector required");
      VULCAN_DEBUG_MSG(index >= 0 && index < kTotal, "index out of bounds");
      return data_[index];
    }

    T& operator[](int index)
    {
      static_assert(kIsVector, "vector required");
      VULCAN_DEBUG_MSG(index >= 0 && index < kTotal, "index out of bounds");
      return data_[index];
    }

    bool operator==(const Matrix& rhs) const
    {
      for (int i = 0; i < kTotal; ++i)
        if (data_[i] != rhs.data_[i]) return false;
      return true;
    }

    bool operator!=(const Matrix& rhs) const { return !(*this == rhs); }

    static Matrix Constant(T value)
    {
      Matrix result;
      for (T& v : result.data_) v = value;
      return result;
    }

    static Matrix Zeros() { return Constant(T(0)); }

    static Matrix Ones() { return Constant(T(1)); }

    static Matrix Identity()
    {
      static_assert(M == N, "square matrix required");
      Matrix result = Zeros();
      for (int i = 0; i < M; ++i) result(i, i) = T(1);
      return result;
    }

    const T* GetData() const { return data_; }

    T* GetData() { return data_; }

  protected:

    T data_[M * N];
};

template <typename S, typename T, int M, int N>
inline Matrix<T, M, N> operator+(S scalar, const Matrix<T, M, N>& matrix)
{
  return matrix + scalar;
}

template <typename T, int M, int N>
inline Matrix<T, M, N> operator*(float scalar, const Matrix<T, M, N>& matrix)
{
  return matrix * scalar;
}

template <typename T, int M, int N>
std::ostream& operator<<(std::ostream& out, const Matrix<T, M, N>& matrix)
{
  for (int r = 0; r < M; ++r)
  {
    for (int c = 0; c < N; ++c)
      out << std::setw(9) << matrix(r, c) << (c + 1 < N ? " " : "");
    if (r + 1 < M) out << std::endl;
  }
  return out;
}

template <typename T, int M> using Vector = Matrix<T, M, 1>;

typedef Vector<char, 2> Vector2c;
typedef Vector<char, 3> Vector3c;
typedef Vector<char, 4> Vector4c;
typedef Vector<int, 2> Vector2i;
typedef Vector<int, 3> Vector3i;
typedef Vector<int, 4> Vector4i;
typedef Vector<short, 2> Vector2s;
typedef Vector<short, 3> Vector3s;
typedef Vector<short, 4> Vector4s;
typedef Vector<float, 2> Vector2f;
typedef Vector<float, 3> Vector3f;
typedef Vector<float, 4> Vector4f;
typedef Vector<float, 6> Vector6f;
typedef Vector<double, 2> Vector2d;
typedef Vector<double, 3> Vector3d;
typedef Vector<double, 4> Vector4d;
typedef Vector<double, 6> Vector6d;
typedef Matrix<float, 2, 2> Matrix2f;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;

} // namespace vulcan
