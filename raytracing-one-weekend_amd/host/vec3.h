// vec3.h — the vector type of the host scene-build API.
// Mirrors the reference's `vec3 = color = point = glm::dvec3` (src/vec3.h:6-15):
// three doubles addressable as x,y,z or r,g,b, with the component-wise operators
// the scene scripts use.  glm is not a dependency of this build.
#pragma once
#include <cmath>

namespace rtweekend::detail {

struct vec3 {
  union { double x; double r; };
  union { double y; double g; };
  union { double z; double b; };
  constexpr vec3() : x(0), y(0), z(0) {}
  constexpr vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
using color = vec3;
using point = vec3;

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(double s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 p, vec3 q) {
  return {p.y * q.z - q.y * p.z, p.z * q.x - q.z * p.x, p.x * q.y - q.x * p.y};
}
inline double length(vec3 v) { return std::sqrt(dot(v, v)); }
inline vec3 normalize(vec3 v) { return v * (1.0 / std::sqrt(dot(v, v))); }

}  // namespace rtweekend::detail

namespace rtweekend {
using detail::color;
using detail::point;
using detail::vec3;
}  // namespace rtweekend
