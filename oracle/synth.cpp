// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// synth.cpp — CPU statement of the synthetic G-buffer generator.
// There is no reference program for this: it replaces the raster stage
// (scene_renderer.cpp:140-220, gbuf/opaque_taa.{vert,frag}) and writes the same
// attachments in the same formats (scene_renderer.cpp:13-43):
//   depth D24 = encode_depth(z_view) (gbuffer_encode.glsl:71-73), normal = encode_normal
//   (gbuffer_encode.glsl:17-27) as RG16_UNORM, albedo / material as RGBA8_SRGB,
//   velocity = 0.5*(ndc_prev - ndc_cur) as RG16F (opaque_taa.frag:45).
//
// Frozen scene (world up is +y on screen; camera model of main.cpp:293-294):
//   ground plane y = 0; back wall z = 12, x in [-12,12], y in [0,7];
//   6 x 4 spheres, radius 0.6, centres (-5+2i, 0.6, 3+2j); everything else is sky (d = 1).
#include "shader_common.hpp"

using namespace oracle;

namespace {

inline uint32_t pcg(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t seed) { return pcg(seed ^ pcg(a ^ pcg(b))); }
inline float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

struct ObjectMaterial { vec3 base; float roughness, metallic; };
inline ObjectMaterial object_material(uint32_t id, uint32_t seed) {
  ObjectMaterial m;
  m.base = vec3(0.3f + 0.7f * u01(hash3(id, 1u, seed)), 0.3f + 0.7f * u01(hash3(id, 2u, seed)),
                0.3f + 0.7f * u01(hash3(id, 3u, seed)));
  m.roughness = 0.1f + 0.8f * u01(hash3(id, 4u, seed));
  m.metallic = (hash3(id, 5u, seed) & 1u) ? 1.0f : 0.0f;
  return m;
}

const float WALL_Z = 12.0f, WALL_X = 12.0f, WALL_Y = 7.0f;
const float SPHERE_R = 0.6f;
const int SPHERES_X = 6, SPHERES_Z = 4;

}  // namespace

extern "C" int vkr_ref_synth_gbuffer(const vkr_img* depth, const vkr_img* normal, const vkr_img* albedo,
                                     const vkr_img* material, const vkr_img* velocity, const vkr_synth_params* params) {
  Image D(*depth);
  const bool depth_only = (params->flags & VKR_SYNTH_DEPTH_ONLY) != 0;
  if (!depth_only && (!normal || !albedo || !material || !velocity)) return 1;
  vkr_img dummy = *depth;
  Image N(depth_only ? dummy : *normal), A(depth_only ? dummy : *albedo), M(depth_only ? dummy : *material),
      V(depth_only ? dummy : *velocity);
  mat4 c2w, prev_mvp, mvp;
  std::memcpy(c2w.m, params->camera_to_world.m, 64);
  std::memcpy(prev_mvp.m, params->prev_mvp.m, 64);
  std::memcpy(mvp.m, params->mvp.m, 64);
  const float tg = tanf(params->fovy / 2.0f);
  const float znear = params->znear, zfar = params->zfar, aspect = params->aspect;
  const uint32_t seed = params->seed;
  const vec3 eye = (c2w * vec4(0, 0, 0, 1)).xyz();
  const int fw = D.fw(), fh = D.fh();

#pragma omp parallel for schedule(dynamic, 8)
  for (int ly = 0; ly < D.h(); ly++) {
    int gy = D.oy() + ly;
    for (int lx = 0; lx < D.w(); lx++) {
      int gx = D.ox() + lx;
      float u = ((float)gx + 0.5f) / (float)fw, v = ((float)gy + 0.5f) / (float)fh;
      float xd = 2.0f * u - 1.0f, yd = 2.0f * v - 1.0f;
      vec3 dir_v((xd * aspect) * tg, yd * tg, -1.0f);
      vec3 dir = (c2w * vec4(dir_v, 0.0f)).xyz();

      float best_t = zfar;  // hits at or beyond zfar are sky
      uint32_t id = 0xFFFFFFFFu;
      vec3 nrm(0, 0, -1);
      // ground
      if (dir.y < 0.0f) {
        float t = -eye.y / dir.y;
        if (t > znear && t < best_t) { best_t = t; id = 0u; nrm = vec3(0, 1, 0); }
      }
      // back wall
      if (dir.z > 0.0f) {
        float t = (WALL_Z - eye.z) / dir.z;
        if (t > znear && t < best_t) {
          float hx = eye.x + t * dir.x, hy = eye.y + t * dir.y;
          if (abs(hx) <= WALL_X && hy >= 0.0f && hy <= WALL_Y) { best_t = t; id = 1u; nrm = vec3(0, 0, -1); }
        }
      }
      // spheres
      float a = dot(dir, dir);
      for (int j = 0; j < SPHERES_Z; j++) {
        for (int i = 0; i < SPHERES_X; i++) {
          vec3 c(-5.0f + 2.0f * (float)i, SPHERE_R, 3.0f + 2.0f * (float)j);
          vec3 oc = eye - c;
          float b = dot(dir, oc);
          float c0 = dot(oc, oc) - SPHERE_R * SPHERE_R;
          float disc = b * b - a * c0;
          if (disc > 0.0f) {
            float t = (-b - sqrtf(disc)) / a;
            if (t > znear && t < best_t) {
              best_t = t;
              id = 2u + (uint32_t)(j * SPHERES_X + i);
              nrm = normalize((eye + t * dir) - c);
            }
          }
        }
      }

      if (id == 0xFFFFFFFFu) {  // sky
        D.store_u32(lx, ly, 0, 0xFFFFFFu);
        if (!depth_only) {
          vec2 en = encode_normal(vec3(0, 0, -1));
          N.store(gx, gy, vec4(en.x, en.y, 0, 0));
          A.store(gx, gy, vec4(0.45f, 0.65f, 0.9f, 1.0f));
          M.store(gx, gy, vec4(0.5f, 1.0f, 0.0f, 0.5f));
          V.store(gx, gy, vec4(0, 0, 0, 0));
        }
        continue;
      }
      float z_view = -best_t;
      float dz = encode_depth(z_view, znear, zfar);
      D.store_u32(lx, ly, 0, float_to_d24(dz));
      if (depth_only) continue;

      vec3 P = eye + best_t * dir;
      ObjectMaterial om = object_material(id, seed);
      int cx, cy;
      if (id == 0u) { cx = f2i(floorf(P.x)); cy = f2i(floorf(P.z)); }
      else if (id == 1u) { cx = f2i(floorf(P.x)); cy = f2i(floorf(P.y)); }
      else { vec2 e = encode_normal(nrm); cx = f2i(floorf(8.0f * e.x)); cy = f2i(floorf(8.0f * e.y)); }
      float checker = (hash3((uint32_t)cx, (uint32_t)cy, seed ^ id) & 1u) ? 1.0f : 0.5f;
      vec2 en = encode_normal(nrm);
      N.store(gx, gy, vec4(en.x, en.y, 0, 0));
      A.store(gx, gy, vec4(om.base * checker, 1.0f));
      float roughness = om.roughness;
      if (params->flags & VKR_SYNTH_TEXTURED_ROUGHNESS) {  // per-texel roughness (include/vkr_postfx.h)
        const float n = u01(hash3((uint32_t)gx, (uint32_t)gy, seed ^ 0x7E57u)) - 0.5f;
        roughness = roughness + 0.3f * n;
        roughness = roughness < 0.02f ? 0.02f : (roughness > 1.0f ? 1.0f : roughness);
      }
      M.store(gx, gy, vec4(0.5f, roughness, om.metallic, 0.5f));
      vec4 cp = prev_mvp * vec4(P, 1.0f), cc = mvp * vec4(P, 1.0f);
      vec2 vel(0.5f * (cp.x / cp.w - cc.x / cc.w), 0.5f * (cp.y / cp.w - cc.y / cc.w));
      V.store(gx, gy, vec4(vel.x, vel.y, 0, 0));
    }
  }
  return 0;
}
