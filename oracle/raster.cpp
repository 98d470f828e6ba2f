// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// raster.cpp — CPU restatement of the G-buffer raster stage: SceneRenderer::draw_taa
// (scene_renderer.cpp:140-220: clear, LESS_OR_EQUAL depth test, cull none — gpu/pipelines.hpp:113-128)
// with shaders/gbuf/opaque_taa.vert:35-45 and opaque_taa.frag:26-46, as a classic immediate-mode
// z-buffer rasterizer: draws and triangles in submission order, every covered pixel depth-tested and
// shaded on the spot (the product resolves a visibility buffer instead).
//
// Frozen raster rules (the fixed-function stage is implementation-defined in Vulkan): pixel centres
// at (x + 0.5, y + 0.5); vertices snapped to 1/256 pixel; top-left fill rule; near-plane clipping
// in clip space (z >= 0), depth clipping per fragment (0 <= z <= 1); D24 = rint(z * (2^24 - 1));
// perspective-correct attributes from screen-space barycentrics; implicit texture LOD from forward
// differences of uv; sRGB textures decoded before filtering; REPEAT addressing.
#include <vector>

#include "shader_common.hpp"

using namespace oracle;

namespace {

struct VsOut { vec4 position, pos_after, pos_before; vec3 normal; vec2 uv; };

mat4 mat_mul(const vkr_mat4& a, const vkr_mat4& b) {  // GLSL mat4 * mat4
  mat4 c;
  for (int col = 0; col < 4; col++)
    for (int row = 0; row < 4; row++) {
      float s = a.m[0 * 4 + row] * b.m[col * 4 + 0];
      for (int k = 1; k < 4; k++) s = s + a.m[k * 4 + row] * b.m[col * 4 + k];
      c.m[col * 4 + row] = s;
    }
  return c;
}

VsOut lerp(const VsOut& p, const VsOut& q, float t) {
  VsOut o;
  auto l = [&](float a, float b) { return a + t * (b - a); };
  o.position = vec4(l(p.position.x, q.position.x), l(p.position.y, q.position.y), l(p.position.z, q.position.z), l(p.position.w, q.position.w));
  o.pos_after = vec4(l(p.pos_after.x, q.pos_after.x), l(p.pos_after.y, q.pos_after.y), l(p.pos_after.z, q.pos_after.z), l(p.pos_after.w, q.pos_after.w));
  o.pos_before = vec4(l(p.pos_before.x, q.pos_before.x), l(p.pos_before.y, q.pos_before.y), l(p.pos_before.z, q.pos_before.z), l(p.pos_before.w, q.pos_before.w));
  o.normal = vec3(l(p.normal.x, q.normal.x), l(p.normal.y, q.normal.y), l(p.normal.z, q.normal.z));
  o.uv = vec2(l(p.uv.x, q.uv.x), l(p.uv.y, q.uv.y));
  return o;
}

typedef long long i64;
i64 edge_fn(i64 ax, i64 ay, i64 bx, i64 by, i64 px, i64 py) { return (bx - ax) * (py - ay) - (by - ay) * (px - ax); }
bool is_top_left(i64 ax, i64 ay, i64 bx, i64 by) {
  const i64 dx = bx - ax, dy = by - ay;
  return dy < 0 || (dy == 0 && dx > 0);
}

int wrap_repeat(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }

// texture(): REPEAT, bilinear inside a level, linear between levels
vec4 sample_level_repeat(const Image& t, int mip, vec2 uv) {
  const int w = t.fw(mip), h = t.fh(mip);
  const float x = cfma(uv.x, (float)w, -0.5f), y = cfma(uv.y, (float)h, -0.5f);
  const float x0f = floorf(x), y0f = floorf(y);
  const float fx = x - x0f, fy = y - y0f;
  const int x0 = wrap_repeat(f2i(x0f), w), y0 = wrap_repeat(f2i(y0f), h);
  const int x1 = wrap_repeat(x0 + 1, w), y1 = wrap_repeat(y0 + 1, h);
  return mix(mix(t.load_local(x0, y0, mip), t.load_local(x1, y0, mip), fx), mix(t.load_local(x0, y1, mip), t.load_local(x1, y1, mip), fx), fy);
}
vec4 sample_trilinear(const Image& t, vec2 uv, vec2 duvdx, vec2 duvdy) {
  const float w = (float)t.fw(0), h = (float)t.fh(0);
  // level pair from the exponent of rho^2 (exact), blend factor from log2f (smooth)
  const float rx2 = (duvdx.x * w) * (duvdx.x * w) + (duvdx.y * h) * (duvdx.y * h);
  const float ry2 = (duvdy.x * w) * (duvdy.x * w) + (duvdy.y * h) * (duvdy.y * h);
  const float r2 = max(rx2, ry2);
  int l0 = 0;
  float f = 0.0f;
  if (r2 > 1.0f && r2 < 3.0e38f) {
    l0 = ilogbf(r2) >> 1;
    f = clamp(0.5f * log2f(r2) - (float)l0, 0.0f, 1.0f);
  }
  if (l0 >= t.mips() - 1) { l0 = t.mips() - 1; f = 0.0f; }
  const int l1 = min(l0 + 1, t.mips() - 1);
  const vec4 a = sample_level_repeat(t, l0, uv);
  if (f == 0.0f || l1 == l0) return a;
  return mix(a, sample_level_repeat(t, l1, uv), f);
}

}  // namespace

extern "C" uint64_t vkr_ref_raster_scratch_bytes(uint32_t, uint32_t, uint32_t) { return 0; }

extern "C" int vkr_ref_raster_gbuffer(const vkr_raster_scene* scene, const vkr_gbuf_const* consts, const vkr_img* albedo,
                                      const vkr_img* normal, const vkr_img* material, const vkr_img* velocity, const vkr_img* depth,
                                      void*, uint64_t) {
  Image ALBEDO(*albedo), NORMAL(*normal), MATERIAL(*material), VELOCITY(*velocity), DEPTH(*depth);
  const int W = ALBEDO.fw(), H = ALBEDO.fh();
  std::vector<Image> textures;
  for (uint32_t i = 0; i < scene->texture_count; i++) textures.emplace_back(scene->textures[i]);
  // clear: colour 0, depth 1 (scene_renderer.cpp:180-181)
  std::vector<uint32_t> zbuf((size_t)W * H, 0x00FFFFFFu);
  for (int ly = 0; ly < ALBEDO.h(); ly++)
    for (int lx = 0; lx < ALBEDO.w(); lx++) {
      ALBEDO.store_u32(lx, ly, 0, 0u); NORMAL.store_u32(lx, ly, 0, 0u); MATERIAL.store_u32(lx, ly, 0, 0u);
      VELOCITY.store_u32(lx, ly, 0, 0u); DEPTH.store_u32(lx, ly, 0, 0x00FFFFFFu);
    }
  const float jx = consts->jitter[0], jy = consts->jitter[1];

  for (uint32_t di = 0; di < scene->draw_count; di++) {
    const vkr_raster_draw& dr = scene->draws[di];
    const vkr_raster_transform& tr = scene->transforms[dr.transform_index];
    const mat4 mvp = mat_mul(consts->view_projection, tr.model);
    const mat4 prev_mvp = mat_mul(consts->prev_view_projection, tr.model);
    mat4 normal_mat;
    std::memcpy(normal_mat.m, tr.normal.m, 64);
    for (uint32_t tri = 0; tri < dr.index_count / 3u; tri++) {
      // vertex shader (opaque_taa.vert:35-45)
      VsOut in[3];
      for (int k = 0; k < 3; k++) {
        const vkr_raster_vertex& v = scene->vertices[dr.vertex_offset + scene->indices[dr.index_offset + 3u * tri + (uint32_t)k]];
        in[k].normal = normalize((normal_mat * vec4(v.norm[0], v.norm[1], v.norm[2], 0.0f)).xyz());
        in[k].uv = vec2(v.uv[0], v.uv[1]);
        const vec4 out_vector = mvp * vec4(v.pos[0], v.pos[1], v.pos[2], 1.0f);
        in[k].position = vec4(out_vector.x + out_vector.w * jx, out_vector.y + out_vector.w * jy, out_vector.z, out_vector.w);
        in[k].pos_after = out_vector;
        in[k].pos_before = prev_mvp * vec4(v.pos[0], v.pos[1], v.pos[2], 1.0f);
      }
      // near-plane clip
      VsOut poly[4];
      int n = 0;
      for (int k = 0; k < 3; k++) {
        const VsOut& p = in[k];
        const VsOut& q = in[(k + 1) % 3];
        const bool pin = p.position.z >= 0.0f, qin = q.position.z >= 0.0f;
        if (pin) poly[n++] = p;
        if (pin != qin) {
          const VsOut& s = pin ? p : q;
          const VsOut& e = pin ? q : p;
          poly[n++] = lerp(s, e, s.position.z / (s.position.z - e.position.z));
        }
      }
      for (int sub = 0; sub + 2 < n; sub++) {
        VsOut v[3] = {poly[0], poly[1 + sub], poly[2 + sub]};
        i64 X[3], Y[3];
        float w[3], z[3];
        bool ok = true;
        for (int k = 0; k < 3 && ok; k++) {
          const vec4 p = v[k].position;
          if (!(p.w > 0.0f)) { ok = false; break; }
          const float xs = ((p.x / p.w) * 0.5f + 0.5f) * (float)W;
          const float ys = ((p.y / p.w) * 0.5f + 0.5f) * (float)H;
          if (!(fabsf(xs) <= 1048576.0f && fabsf(ys) <= 1048576.0f)) { ok = false; break; }
          X[k] = (i64)rintf(xs * 256.0f);
          Y[k] = (i64)rintf(ys * 256.0f);
          w[k] = p.w;
          z[k] = p.z / p.w;
        }
        if (!ok) continue;
        i64 area2 = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
        if (area2 == 0) continue;
        if (area2 < 0) {
          std::swap(v[1], v[2]); std::swap(X[1], X[2]); std::swap(Y[1], Y[2]); std::swap(w[1], w[2]); std::swap(z[1], z[2]);
          area2 = -area2;
        }
        const i64 minx = std::min(X[0], std::min(X[1], X[2])), maxx = std::max(X[0], std::max(X[1], X[2]));
        const i64 miny = std::min(Y[0], std::min(Y[1], Y[2])), maxy = std::max(Y[0], std::max(Y[1], Y[2]));
        const int x0 = (int)std::max<i64>((minx - 128) >> 8, 0), x1 = (int)std::min<i64>((maxx - 128) >> 8, W - 1);
        const int y0 = (int)std::max<i64>((miny - 128) >> 8, 0), y1 = (int)std::min<i64>((maxy - 128) >> 8, H - 1);
        const double inv = 1.0 / (double)area2;
        auto lambdas = [&](int px, int py, float l[3], i64 e[3]) {
          const i64 PX = ((i64)px << 8) + 128, PY = ((i64)py << 8) + 128;
          e[0] = edge_fn(X[1], Y[1], X[2], Y[2], PX, PY);
          e[1] = edge_fn(X[2], Y[2], X[0], Y[0], PX, PY);
          e[2] = edge_fn(X[0], Y[0], X[1], Y[1], PX, PY);
          for (int k = 0; k < 3; k++) l[k] = (float)((double)e[k] * inv);
        };
        auto persp = [&](const float l[3], float b[3]) {
          const float q0 = l[0] / w[0], q1 = l[1] / w[1], q2 = l[2] / w[2];
          const float s = (q0 + q1) + q2;
          b[0] = q0 / s; b[1] = q1 / s; b[2] = q2 / s;
        };
        for (int py = y0; py <= y1; py++) {
          for (int px = x0; px <= x1; px++) {
            float l[3];
            i64 e[3];
            lambdas(px, py, l, e);
            if (e[0] < 0 || e[1] < 0 || e[2] < 0) continue;
            if (e[0] == 0 && !is_top_left(X[1], Y[1], X[2], Y[2])) continue;
            if (e[1] == 0 && !is_top_left(X[2], Y[2], X[0], Y[0])) continue;
            if (e[2] == 0 && !is_top_left(X[0], Y[0], X[1], Y[1])) continue;
            const float zf = (l[0] * z[0] + l[1] * z[1]) + l[2] * z[2];
            if (!(zf >= 0.0f && zf <= 1.0f)) continue;
            const uint32_t d24 = (uint32_t)rintf(zf * 16777215.0f);
            uint32_t& stored = zbuf[(size_t)py * W + px];
            if (!(d24 <= stored)) continue;  // VK_COMPARE_OP_LESS_OR_EQUAL
            // fragment shader (opaque_taa.frag:26-46).  `discard` (:32-34) means the fragment writes nothing,
            // depth included, so the depth buffer is only updated after the alpha test.
            float b[3], lx[3], ly[3], bx[3], by[3];
            i64 tmp[3];
            persp(l, b);
            lambdas(px + 1, py, lx, tmp);
            lambdas(px, py + 1, ly, tmp);
            persp(lx, bx);
            persp(ly, by);
#define BARY(B, F) ((B[0] * v[0].F + B[1] * v[1].F) + B[2] * v[2].F)
            const vec3 in_normal(BARY(b, normal.x), BARY(b, normal.y), BARY(b, normal.z));
            const vec2 in_uv(BARY(b, uv.x), BARY(b, uv.y));
            const vec4 pa(BARY(b, pos_after.x), BARY(b, pos_after.y), BARY(b, pos_after.z), BARY(b, pos_after.w));
            const vec4 pb(BARY(b, pos_before.x), BARY(b, pos_before.y), BARY(b, pos_before.z), BARY(b, pos_before.w));
            const vec2 ddx = vec2(BARY(bx, uv.x), BARY(bx, uv.y)) - in_uv, ddy = vec2(BARY(by, uv.x), BARY(by, uv.y)) - in_uv;
#undef BARY
            vec4 out_albedo(0.5f, 0.5f, 0.5f, 1.0f);
            if (dr.albedo_index != 0xFFFFFFFFu) out_albedo = sample_trilinear(textures[dr.albedo_index], in_uv, ddx, ddy);
            if (out_albedo.w == 0.0f) continue;  // discard
            stored = d24;
            vec4 out_material(0.5f, 0.9f, 0.5f, 0.5f);
            if (dr.mr_index != 0xFFFFFFFFu) out_material = sample_trilinear(textures[dr.mr_index], in_uv, ddx, ddy);
            const vec2 en = encode_normal(in_normal);
            const vec2 vel(0.5f * (pb.x / pb.w - pa.x / pa.w), 0.5f * (pb.y / pb.w - pa.y / pa.w));
            ALBEDO.store(px, py, out_albedo);
            MATERIAL.store(px, py, out_material);
            NORMAL.store(px, py, vec4(en.x, en.y, 0.0f, 0.0f));
            VELOCITY.store(px, py, vec4(vel.x, vel.y, 0.0f, 0.0f));
            const int wx = px - DEPTH.ox(), wy = py - DEPTH.oy();
            if (wx >= 0 && wy >= 0 && wx < DEPTH.w() && wy < DEPTH.h()) DEPTH.store_u32(wx, wy, 0, d24);
          }
        }
      }
    }
  }
  return 0;
}
