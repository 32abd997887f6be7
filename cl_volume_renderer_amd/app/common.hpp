// common.hpp -- camera basis and launch-size helper the renderer's host path needs
// (mirror of the reference's app/common.hpp: Position3D :5-57, evenness :59-66).
//
// Numeric contract of Position3D(alpha, beta, gamma, base): the rotated vector is evaluated in
// double, stored to float, then normalised by a float length (sqrtf of a double sum) -- the kernel
// receives exactly these three floats as its camera direction (app/renderer.cpp:140-148).
#pragma once

#include <cmath>

struct Position3D {
  float val[3];

  Position3D(double x, double y, double z) : val{(float)x, (float)y, (float)z} {}

  // yaw alpha about y, pitch beta, roll gamma, applied to `base`
  Position3D(double alpha, double beta, double gamma, Position3D base) {
    const double ca = std::cos(alpha), sa = std::sin(alpha);
    const double cb = std::cos(beta), sb = std::sin(beta);
    const double cg = std::cos(gamma), sg = std::sin(gamma);
    const double row0[3] = {ca * cb, ca * sb - sa * cg, ca * sb * cg + sa * sg};
    const double row1[3] = {-sb, cb * sg, cb * cg};
    const double row2[3] = {sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg};
    val[0] = (float)(row0[0] * base.val[0] + row0[1] * base.val[1] + row0[2] * base.val[2]);
    val[1] = (float)(row1[0] * base.val[0] + row1[1] * base.val[1] + row1[2] * base.val[2]);
    val[2] = (float)(row2[0] * base.val[0] + row2[1] * base.val[1] + row2[2] * base.val[2]);
    normalize();
  }

  Position3D operator+(Position3D o) const { return {val[0] + o.val[0], val[1] + o.val[1], val[2] + o.val[2]}; }
  Position3D operator-(Position3D o) const { return {val[0] - o.val[0], val[1] - o.val[1], val[2] - o.val[2]}; }
  Position3D operator*(float s) const { return {val[0] * s, val[1] * s, val[2] * s}; }
  Position3D operator/(float s) const { return {val[0] / s, val[1] / s, val[2] / s}; }

  double length() const {
    return sqrtf((float)(std::pow((double)val[0], 2) + std::pow((double)val[1], 2) + std::pow((double)val[2], 2)));
  }
  void normalize() { *this = *this / (float)length(); }
};

// smallest multiple of l that is >= g (NDRange global sizes must be multiples of the local size)
inline unsigned int evenness(const unsigned int g, const unsigned int l) {
  const unsigned int r = g % l;
  return r == 0 ? g : g + (l - r);
}
