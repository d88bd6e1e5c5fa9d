// math.h — scalar helpers with the reference's comparison order
// (ref: include/vulcan/math.h:9-32; `min(a,b)` returns a when b is NaN).
#pragma once

#include <cmath>

namespace vulcan
{

template <typename T> inline T min(T a, T b) { return (b < a) ? b : a; }

template <typename T> inline T max(T a, T b) { return (b > a) ? b : a; }

template <typename T> inline T clamp(T v, T lo, T hi) { return min(hi, max(lo, v)); }

template <typename T> inline T sqrt(T value) { return ::std::sqrt(value); }

template <typename T> inline bool isnan(T value) { return ::std::isnan(value); }

} // namespace vulcan
