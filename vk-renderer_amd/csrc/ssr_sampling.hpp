// ssr_sampling.hpp — reflection-direction sampling shared by the stochastic SSR trace kernels
// (trace.comp / trace_indirect.comp :65-77,143-154 and brdf.glsl:135-155).
#pragma once
#include "vkr_device.hpp"

namespace vkr {

// trace.comp:143-154
VKR_DEV f3 get_tangent(f3 n) {
  float max_xy = vmax(fabsf(n.x), fabsf(n.y));
  f3 t = (max_xy < 0.00001f) ? mk3(1, 0, 0) : mk3(n.y, -n.x, 0);
  return normalize(t);
}

// brdf.glsl:135-155; cos/sin(phi), phi = 2*PI*U2, come with the Halton entry (vkr_halton23_fill:
// evaluated in double on the host and rounded once — they steer the march)
VKR_DEV f3 sampleGGXVNDF(f3 Ve, float alpha_x, float alpha_y, float U1, float cos_phi, float sin_phi) {
  f3 Vh = normalize(mk3(alpha_x * Ve.x, alpha_y * Ve.y, Ve.z));
  float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
  f3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrt_ieee(lensq)) : mk3(1, 0, 0);
  f3 T2 = cross(Vh, T1);
  float r = sqrt_ieee(U1);
  float t1 = r * cos_phi;
  float t2 = r * sin_phi;
  float s = 0.5f * (1.0f + Vh.z);
  t2 = (1.0f - s) * sqrt_ieee(1.0f - t1 * t1) + s * t2;
  f3 Nh = (t1 * T1 + t2 * T2) + sqrt_ieee(vmax(0.0f, (1.0f - t1 * t1) - t2 * t2)) * Vh;
  return normalize(mk3(alpha_x * Nh.x, alpha_y * Nh.y, vmax(0.0f, Nh.z)));
}

}  // namespace vkr
