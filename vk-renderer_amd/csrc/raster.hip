// raster.hip — program "gbuf_opaque_taa": the G-buffer raster stage of the reference
// (SceneRenderer::draw_taa, scene_renderer.cpp:140-220 + shaders/gbuf/opaque_taa.{vert,frag}) as a
// compute rasterizer.
//
//   k_raster_clear    visibility buffer := empty
//   k_raster_setup    one thread per triangle: vertex shader on its three corners, near-plane clip (up to
//                     two sub-triangles), 8-bit sub-pixel snap, orientation -> ScreenTri records in scratch
//                     in scratch; records whose bounding box exceeds 64 blocks of 8x8 pixels go to a work list
//   k_raster_small    one wave per triangle walks the (<= 64) blocks of its bounding box, one pixel per lane
//   k_raster_large    the listed triangles, their bounding boxes cut into chunks of 16 blocks that are dealt to all
//                     waves of the launch (a screen-filling quad does not serialise on a few waves)
//                     both: fill rule on exact 64-bit edge functions of 24.8 coordinates, D24 depth, atomicMin
//                     of (depth << 32 | ~record) — LESS_OR_EQUAL with later triangles winning ties
//                     (gpu/pipelines.hpp:128)
//   k_raster_resolve  one thread per pixel: the winning triangle's record, perspective-correct attributes,
//                     implicit LOD from forward differences, trilinear sRGB fetches, stores
//
// opaque_taa.frag:32-34 discards a fragment whose albedo alpha is 0: it then writes neither depth nor any
// attachment.  A visibility buffer commits coverage before shading, so the discard is evaluated at coverage
// time: for draws with an albedo texture the covering lane computes the fragment's uv, its forward-difference
// LOD and the filtered alpha (the very expression the resolve evaluates again) and skips the atomicMin when
// it is 0.  A caller that knows a texture has no alpha-0 texel in any mip level sets
// VKR_RASTER_DRAW_OPAQUE_ALBEDO on the draw and the test is skipped (the result cannot differ).
//
// Coverage and depth are integer-exact functions of the snapped vertices, so they are bit-equal to
// the oracle's immediate-mode rasterizer; colour / normal / velocity follow the frozen fp32 contract.
#include <vector>

#include "vkr_host.hpp"

namespace vkr {

#define RASTER_MAX_TEXTURES 32
#define RASTER_GUARD_PX 1048576.0f  // |screen coordinate| beyond this: the triangle is dropped (documented limit)

struct DrawDev {  // one draw call, matrices premultiplied on the host exactly as the vertex shader does
  Mat4 mvp, prev_mvp, normal_mat;
  uint32_t albedo_index, mr_index, index_offset, vertex_offset;
  uint32_t tri_base, tri_count, alpha_test, pad1;  // alpha_test: the discard of opaque_taa.frag:32 can fire for this draw
};

struct RasterArgs {
  const vkr_raster_vertex* vertices;
  const uint32_t* indices;
  const DrawDev* draws;
  const Pyramid* tex;  // [RASTER_MAX_TEXTURES] in scratch
  uint32_t draw_count;
  unsigned long long* vis;
  struct ScreenTri* setup;  // [2 * total triangles]: sub-triangle records written by k_raster_setup
  int width, height;  // framebuffer = whole frame
  float jitter_x, jitter_y;
};

struct VsOut { f4 position, pos_after, pos_before; f3 normal; f2 uv; };

// opaque_taa.vert:35-45
VKR_DEV VsOut vertex_shader(const RasterArgs& a, const DrawDev& d, uint32_t index) {
  const vkr_raster_vertex v = a.vertices[d.vertex_offset + a.indices[d.index_offset + index]];
  VsOut o;
  o.normal = normalize(xyz(mul(d.normal_mat, mk4(v.norm[0], v.norm[1], v.norm[2], 0.0f))));
  o.uv = mk2(v.uv[0], v.uv[1]);
  const f4 out_vector = mul(d.mvp, mk4(v.pos[0], v.pos[1], v.pos[2], 1.0f));
  o.position = mk4(out_vector.x + out_vector.w * a.jitter_x, out_vector.y + out_vector.w * a.jitter_y, out_vector.z, out_vector.w);
  o.pos_after = out_vector;
  o.pos_before = mul(d.prev_mvp, mk4(v.pos[0], v.pos[1], v.pos[2], 1.0f));
  return o;
}

VKR_DEV VsOut vs_lerp(const VsOut& p, const VsOut& q, float t) {  // p + t (q - p), every output
  VsOut o;
#define L1(F) o.F = p.F + t * (q.F - p.F)
  L1(position.x); L1(position.y); L1(position.z); L1(position.w);
  L1(pos_after.x); L1(pos_after.y); L1(pos_after.z); L1(pos_after.w);
  L1(pos_before.x); L1(pos_before.y); L1(pos_before.z); L1(pos_before.w);
  L1(normal.x); L1(normal.y); L1(normal.z);
  L1(uv.x); L1(uv.y);
#undef L1
  return o;
}

// A triangle ready for rasterisation: snapped screen positions (8 sub-pixel bits), 1/w-space data
struct ScreenTri {
  int x[3], y[3];  // 24.8 fixed point, |v| <= 2^28 (guard band)
  float w[3], z[3];  // clip w and z / w
  long long area2;
  double inv_area2;  // 1.0 / (double)area2, once per triangle: every pixel's barycentrics multiply by it
  VsOut v[3];
  uint32_t alpha_tex;  // albedo texture whose filtered alpha decides the discard, 0xFFFFFFFF: no test
  bool valid;
};

// differences of 24.8 coordinates fit 32 bits, their products need 64 (v_mad_i64_i32)
VKR_DEV long long edge_fn(int ax, int ay, int bx, int by, int px, int py) {
  return (long long)(bx - ax) * (long long)(py - ay) - (long long)(by - ay) * (long long)(px - ax);
}
// top-left rule for an edge a->b of a triangle with positive area2 under edge_fn (y down)
VKR_DEV bool is_top_left(int ax, int ay, int bx, int by) {
  const int dx = bx - ax, dy = by - ay;
  return dy < 0 || (dy == 0 && dx > 0);
}

// Vertex shader on the three corners + near-plane clip (z_clip >= 0) -> sub-triangle `sub` (0 or 1).
// Returns the number of sub-triangles the clipped polygon has (0, 1 or 2) in *count.
VKR_DEV ScreenTri setup_triangle(const RasterArgs& a, const DrawDev& d, uint32_t tri, int sub, int* count) {
  ScreenTri t;
  t.valid = false;
  t.alpha_tex = d.alpha_test ? d.albedo_index : 0xFFFFFFFFu;
  VsOut in[3], poly[4];
  for (int k = 0; k < 3; k++) in[k] = vertex_shader(a, d, 3u * tri + (uint32_t)k);
  int n = 0;
  for (int k = 0; k < 3; k++) {  // Sutherland-Hodgman against z >= 0
    const VsOut& p = in[k];
    const VsOut& q = in[(k + 1) % 3];
    const bool pin = p.position.z >= 0.0f, qin = q.position.z >= 0.0f;
    if (pin) poly[n++] = p;
    if (pin != qin) {
      // always interpolate from the inside vertex so both orientations of a shared edge agree
      const VsOut& s = pin ? p : q;
      const VsOut& e = pin ? q : p;
      poly[n++] = vs_lerp(s, e, s.position.z / (s.position.z - e.position.z));
    }
  }
  *count = n < 3 ? 0 : n - 2;
  if (sub >= *count) return t;
  t.v[0] = poly[0]; t.v[1] = poly[1 + sub]; t.v[2] = poly[2 + sub];
  for (int k = 0; k < 3; k++) {
    const f4 p = t.v[k].position;
    if (!(p.w > 0.0f)) return t;
    const float xs = ((p.x / p.w) * 0.5f + 0.5f) * (float)a.width;
    const float ys = ((p.y / p.w) * 0.5f + 0.5f) * (float)a.height;
    if (!(fabsf(xs) <= RASTER_GUARD_PX && fabsf(ys) <= RASTER_GUARD_PX)) return t;
    t.x[k] = (int)rintf(xs * 256.0f);
    t.y[k] = (int)rintf(ys * 256.0f);
    t.w[k] = p.w;
    t.z[k] = p.z / p.w;
  }
  t.area2 = edge_fn(t.x[0], t.y[0], t.x[1], t.y[1], t.x[2], t.y[2]);
  if (t.area2 == 0) return t;
  if (t.area2 < 0) {  // cull none: both windings are drawn; normalise the orientation
    const VsOut tv = t.v[1]; t.v[1] = t.v[2]; t.v[2] = tv;
    int tl = t.x[1]; t.x[1] = t.x[2]; t.x[2] = tl;
    tl = t.y[1]; t.y[1] = t.y[2]; t.y[2] = tl;
    float tf = t.w[1]; t.w[1] = t.w[2]; t.w[2] = tf;
    tf = t.z[1]; t.z[1] = t.z[2]; t.z[2] = tf;
    t.area2 = -t.area2;
  }
  t.inv_area2 = 1.0 / (double)t.area2;
  t.valid = true;
  return t;
}

// What coverage and depth need of a ScreenTri, copied into registers once per triangle: the rasterising waves issue
// atomics between their reads of the record, and the compiler must otherwise assume those change it and reload.
struct CoverTri {
  int x[3], y[3];
  float z[3];
  double inv_area2;
  VKR_DEV CoverTri() {}
  VKR_DEV explicit CoverTri(const ScreenTri& t) : x {t.x[0], t.x[1], t.x[2]}, y {t.y[0], t.y[1], t.y[2]}, z {t.z[0], t.z[1], t.z[2]}, inv_area2 {t.inv_area2} {}
};
// coverage + depth of pixel (px, py); lambda: screen-space barycentrics
template <class T> VKR_DEV bool cover(const T& t, int px, int py, float lambda[3], uint32_t* d24) {
  const int X = (px << 8) + 128, Y = (py << 8) + 128;
  const long long e0 = edge_fn(t.x[1], t.y[1], t.x[2], t.y[2], X, Y);
  const long long e1 = edge_fn(t.x[2], t.y[2], t.x[0], t.y[0], X, Y);
  const long long e2 = edge_fn(t.x[0], t.y[0], t.x[1], t.y[1], X, Y);
  if (e0 < 0 || e1 < 0 || e2 < 0) return false;
  if (e0 == 0 && !is_top_left(t.x[1], t.y[1], t.x[2], t.y[2])) return false;
  if (e1 == 0 && !is_top_left(t.x[2], t.y[2], t.x[0], t.y[0])) return false;
  if (e2 == 0 && !is_top_left(t.x[0], t.y[0], t.x[1], t.y[1])) return false;
  const double inv = t.inv_area2;
  lambda[0] = (float)((double)e0 * inv);
  lambda[1] = (float)((double)e1 * inv);
  lambda[2] = (float)((double)e2 * inv);
  const float depth = (lambda[0] * t.z[0] + lambda[1] * t.z[1]) + lambda[2] * t.z[2];
  if (!(depth >= 0.0f && depth <= 1.0f)) return false;  // depth clipping (far plane; near was clipped)
  *d24 = (uint32_t)rintf(depth * 16777215.0f);
  return true;
}
// barycentrics at an arbitrary (possibly uncovered) pixel, for the forward differences of uv
VKR_DEV void lambda_at(const ScreenTri& t, int px, int py, float lambda[3]) {
  const int X = (px << 8) + 128, Y = (py << 8) + 128;
  const double inv = t.inv_area2;
  lambda[0] = (float)((double)edge_fn(t.x[1], t.y[1], t.x[2], t.y[2], X, Y) * inv);
  lambda[1] = (float)((double)edge_fn(t.x[2], t.y[2], t.x[0], t.y[0], X, Y) * inv);
  lambda[2] = (float)((double)edge_fn(t.x[0], t.y[0], t.x[1], t.y[1], X, Y) * inv);
}
VKR_DEV void perspective(const ScreenTri& t, const float lambda[3], float b[3]) {
  const float q0 = lambda[0] / t.w[0], q1 = lambda[1] / t.w[1], q2 = lambda[2] / t.w[2];
  const float s = (q0 + q1) + q2;
  b[0] = q0 / s; b[1] = q1 / s; b[2] = q2 / s;
}
#define BARY(F) ((b[0] * t.v[0].F + b[1] * t.v[1].F) + b[2] * t.v[2].F)

VKR_DEV int wrap_repeat(int i, int n) {
  if ((n & (n - 1)) == 0) return i & (n - 1);  // power-of-two extent (every mip of the usual texture): no integer division
  const int m = i % n;
  return m < 0 ? m + n : m;
}
// texture(sampler2D, uv) of an RGBA8_SRGB mip chain: REPEAT, bilinear, linear between the two mips of `lod`
// `lut`: the sRGB decode table (srgb_lut_stage), in LDS where the caller has staged it
VKR_DEV f4 sample_level_repeat(const Tex& t, f2 uv, const float* lut) {
  const float x = cfma(uv.x, (float)t.fw, -0.5f), y = cfma(uv.y, (float)t.fh, -0.5f);
  const float x0f = floorf(x), y0f = floorf(y);
  const float fx = x - x0f, fy = y - y0f;
  const int x0 = wrap_repeat(f2i(x0f), t.fw), y0 = wrap_repeat(f2i(y0f), t.fh);
  const int x1 = wrap_repeat(x0 + 1, t.fw), y1 = wrap_repeat(y0 + 1, t.fh);
  auto dec = [&](int tx, int ty) {
    const uint32_t v = *texel_ptr<const uint32_t>(t, tx, ty);
    return mk4(lut[v & 0xFFu], lut[(v >> 8) & 0xFFu], lut[(v >> 16) & 0xFFu], unorm8_to_float(v >> 24));
  };
  return mix4(mix4(dec(x0, y0), dec(x1, y0), fx), mix4(dec(x0, y1), dec(x1, y1), fx), fy);
}
VKR_DEV f4 sample_trilinear(const Pyramid& p, f2 uv, f2 duvdx, f2 duvdy, const float* lut) {
  const float w = (float)p.mip[0].fw, h = (float)p.mip[0].fh;
  // rho^2 = max squared footprint; lod = log2(rho).  The level pair comes from the exponent of rho^2
  // (exact), only the blend factor from log2f (smooth) — a libm ulp must not flip the pair.
  const float rx2 = (duvdx.x * w) * (duvdx.x * w) + (duvdx.y * h) * (duvdx.y * h);
  const float ry2 = (duvdy.x * w) * (duvdy.x * w) + (duvdy.y * h) * (duvdy.y * h);
  const float r2 = vmax(rx2, ry2);
  int l0 = 0;
  float f = 0.0f;
  if (r2 > 1.0f && r2 < 3.0e38f) {
    l0 = ilogbf(r2) >> 1;  // floor(log2(rho))
    f = vclamp(0.5f * log2f(r2) - (float)l0, 0.0f, 1.0f);
  }
  if (l0 >= p.count - 1) { l0 = p.count - 1; f = 0.0f; }  // sampler LOD range [0, 10] and the chain length
  const int l1 = min(l0 + 1, p.count - 1);
  const f4 a = sample_level_repeat(p.mip[l0], uv, lut);
  if (f == 0.0f || l1 == l0) return a;
  return mix4(a, sample_level_repeat(p.mip[l1], uv, lut), f);
}


// uv and its forward differences at pixel (px, py) of triangle t (lambda: its screen-space barycentrics there):
// what the fragment shader's texture() calls see (implicit derivatives as differences to the right / lower pixel)
struct FragUv { f2 uv, ddx, ddy; float b[3]; };
VKR_DEV FragUv fragment_uv(const ScreenTri& t, int px, int py, const float lambda[3]) {
  FragUv f;
  perspective(t, lambda, f.b);
  const float* b = f.b;
  f.uv = mk2(BARY(uv.x), BARY(uv.y));
  float lx1[3], ly1[3], bx1[3], by1[3];
  lambda_at(t, px + 1, py, lx1);
  lambda_at(t, px, py + 1, ly1);
  perspective(t, lx1, bx1);
  perspective(t, ly1, by1);
  const f2 uvx = mk2((bx1[0] * t.v[0].uv.x + bx1[1] * t.v[1].uv.x) + bx1[2] * t.v[2].uv.x, (bx1[0] * t.v[0].uv.y + bx1[1] * t.v[1].uv.y) + bx1[2] * t.v[2].uv.y);
  const f2 uvy = mk2((by1[0] * t.v[0].uv.x + by1[1] * t.v[1].uv.x) + by1[2] * t.v[2].uv.x, (by1[0] * t.v[0].uv.y + by1[1] * t.v[1].uv.y) + by1[2] * t.v[2].uv.y);
  f.ddx = uvx - f.uv; f.ddy = uvy - f.uv;
  return f;
}
// the per-draw constants travel as kernel arguments (8 per launch) into the draw table in scratch: no host
// staging memory has to outlive the call and nothing synchronises
struct DrawChunk { DrawDev d[8]; };
__global__ void k_raster_store_draws(DrawChunk c, DrawDev* dst, uint32_t n) {
  if (threadIdx.x < n) dst[threadIdx.x] = c.d[threadIdx.x];
}
struct TexChunk { Pyramid p[4]; };
__global__ void k_raster_store_textures(TexChunk c, Pyramid* dst, uint32_t n) {
  if (threadIdx.x < n) dst[threadIdx.x] = c.p[threadIdx.x];
}

__global__ void k_raster_clear(unsigned long long* vis, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) vis[i] = ~0ull;
}

// pixel bounding box (centres that can be covered), clipped to the viewport; false when empty
template <class T> VKR_DEV bool tri_bbox(const T& t, int width, int height, int* x0, int* y0, int* x1, int* y1) {
  const int minx = min(t.x[0], min(t.x[1], t.x[2])), maxx = max(t.x[0], max(t.x[1], t.x[2]));
  const int miny = min(t.y[0], min(t.y[1], t.y[2])), maxy = max(t.y[0], max(t.y[1], t.y[2]));
  *x0 = max((minx - 128) >> 8, 0); *x1 = min((maxx - 128) >> 8, width - 1);
  *y0 = max((miny - 128) >> 8, 0); *y1 = min((maxy - 128) >> 8, height - 1);
  return *x0 <= *x1 && *y0 <= *y1;
}
#define RASTER_SMALL_BLOCKS 64  // sub-triangles whose bounding box has more 8x8 blocks go to the shared-work kernel
#define RASTER_LARGE_CHUNK 16   // blocks per work item of the shared-work kernel
#define RASTER_LARGE_GRID 2048  // its blocks of four waves: chunk c goes to wave c mod (4 x grid)
struct LargeEntry { uint32_t rec, first_chunk; };  // a listed sub-triangle and the index of its first chunk

// one thread per triangle; a large sub-triangle takes a list slot AND its range of chunks with one 64-bit atomicAdd on
// *large_state (entries << 32 | chunks), so the list is sorted by first_chunk
// the draw that owns global triangle `gtri` (draws are consecutive ranges [tri_base, tri_base + tri_count))
VKR_DEV uint32_t draw_of(const RasterArgs& a, uint32_t gtri) {
  uint32_t lo = 0, hi = a.draw_count - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (a.draws[mid].tri_base <= gtri) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// (every draw of the frame in one launch: a draw of a few thousand triangles does not fill the chip on its own)
__global__ __launch_bounds__(256) void k_raster_setup(RasterArgs a, uint32_t total_tris, unsigned long long* large_state, LargeEntry* large_list) {
  const uint32_t gtri = blockIdx.x * blockDim.x + threadIdx.x;
  if (gtri >= total_tris) return;
  const DrawDev d = a.draws[draw_of(a, gtri)];
  const uint32_t tri = gtri - d.tri_base;
  for (int sub = 0; sub < 2; sub++) {
    int count;
    ScreenTri t = setup_triangle(a, d, tri, sub, &count);
    if (sub >= count) t.valid = false;
    const uint32_t rec = (d.tri_base + tri) * 2u + (uint32_t)sub;
    a.setup[rec] = t;
    int x0, y0, x1, y1;
    if (t.valid && tri_bbox(t, a.width, a.height, &x0, &y0, &x1, &y1)) {
      const int nb = ((x1 >> 3) - (x0 >> 3) + 1) * ((y1 >> 3) - (y0 >> 3) + 1);
      if (nb > RASTER_SMALL_BLOCKS) {
        const uint32_t chunks = (uint32_t)(nb + RASTER_LARGE_CHUNK - 1) / RASTER_LARGE_CHUNK;
        const unsigned long long v = atomicAdd(large_state, (1ull << 32) | (unsigned long long)chunks);
        large_list[(uint32_t)(v >> 32)] = LargeEntry {rec, (uint32_t)v};
      }
    }
  }
}

// Largest value edge a->b takes over the pixel centres X in [X0, X1], Y in [Y0, Y1] (24.8): when it is negative no pixel of
// the block is inside the triangle (an edge function is linear, its maximum over a box sits at a corner)
VKR_DEV long long edge_max(int ax, int ay, int bx, int by, int X0, int Y0, int X1, int Y1) {
  const int dx = bx - ax, dy = by - ay;
  return (long long)dx * (long long)((dx > 0 ? Y1 : Y0) - ay) - (long long)dy * (long long)((dy > 0 ? X0 : X1) - ax);
}
// 8x8 pixel block `b` (row-major inside the bounding box) of record `rec`, one pixel per lane
VKR_DEV void raster_block(const RasterArgs& a, const CoverTri& t, uint32_t alpha_tex, uint32_t rec, int x0, int y0, int x1, int y1, int b, int lane) {
  const int bw = (x1 >> 3) - (x0 >> 3) + 1;
  const int bx0 = ((x0 >> 3) + b % bw) << 3, by0 = ((y0 >> 3) + b / bw) << 3;
  {  // the whole block outside one edge (half the blocks of a large triangle's bounding box): nothing to test per pixel
    const int X0 = (bx0 << 8) + 128, Y0 = (by0 << 8) + 128, X1 = X0 + 7 * 256, Y1 = Y0 + 7 * 256;
    if (edge_max(t.x[1], t.y[1], t.x[2], t.y[2], X0, Y0, X1, Y1) < 0 || edge_max(t.x[2], t.y[2], t.x[0], t.y[0], X0, Y0, X1, Y1) < 0 ||
        edge_max(t.x[0], t.y[0], t.x[1], t.y[1], X0, Y0, X1, Y1) < 0)
      return;
  }
  const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
  if (px < x0 || px > x1 || py < y0 || py > y1) return;
  float lambda[3];
  uint32_t d24;
  if (!cover(t, px, py, lambda, &d24)) return;
  if (alpha_tex != 0xFFFFFFFFu) {  // opaque_taa.frag:32-34: out_albedo.a == 0 -> discard (no depth, no colour)
    const FragUv f = fragment_uv(a.setup[rec], px, py, lambda);
    if (sample_trilinear(a.tex[alpha_tex], f.uv, f.ddx, f.ddy, (const float*)k_srgb_decode_bits).w == 0.0f) return;  // only alpha is used: the colour decodes fold away
  }
  atomicMin(&a.vis[(size_t)py * a.width + px], ((unsigned long long)d24 << 32) | (0xFFFFFFFFull - (unsigned long long)rec));
}

// small triangles: one wave per triangle walks its (at most 64) blocks
__global__ __launch_bounds__(256) void k_raster_small(RasterArgs a, uint32_t total_tris) {
  const uint32_t gtri = blockIdx.x * 4u + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (gtri >= total_tris) return;
  for (uint32_t sub = 0; sub < 2; sub++) {
    const uint32_t rec = gtri * 2u + sub;
    if (!a.setup[rec].valid) continue;
    const CoverTri t(a.setup[rec]);
    const uint32_t alpha_tex = a.setup[rec].alpha_tex;
    int x0, y0, x1, y1;
    if (!tri_bbox(t, a.width, a.height, &x0, &y0, &x1, &y1)) continue;
    const int nb = ((x1 >> 3) - (x0 >> 3) + 1) * ((y1 >> 3) - (y0 >> 3) + 1);
    if (nb > RASTER_SMALL_BLOCKS) continue;
    for (int b = 0; b < nb; b++) raster_block(a, t, alpha_tex, rec, x0, y0, x1, y1, b, lane);
  }
}

// large triangles: their bounding boxes are cut into chunks of RASTER_LARGE_CHUNK blocks, all chunks of all listed
// triangles are dealt round-robin to the waves of the launch (a screen-filling quad is 4096 chunks at 4K: every wave
// of the launch works on it, none serialises)
__global__ __launch_bounds__(256) void k_raster_large(RasterArgs a, const unsigned long long* large_state, const LargeEntry* large_list) {
  const unsigned long long st = *large_state;
  const uint32_t n = (uint32_t)(st >> 32), chunks = (uint32_t)st;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), waves = gridDim.x * 4u;
  for (uint32_t c = wave; c < chunks; c += waves) {
    uint32_t lo = 0, hi = n - 1;  // the last entry with first_chunk <= c
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (large_list[mid].first_chunk <= c) lo = mid; else hi = mid - 1;
    }
    const LargeEntry e = large_list[lo];
    const CoverTri t(a.setup[e.rec]);
    const uint32_t alpha_tex = a.setup[e.rec].alpha_tex;
    int x0, y0, x1, y1;
    if (!tri_bbox(t, a.width, a.height, &x0, &y0, &x1, &y1)) continue;
    const int nb = ((x1 >> 3) - (x0 >> 3) + 1) * ((y1 >> 3) - (y0 >> 3) + 1);
    const int b0 = (int)(c - e.first_chunk) * RASTER_LARGE_CHUNK, b1 = min(b0 + RASTER_LARGE_CHUNK, nb);
    for (int b = b0; b < b1; b++) raster_block(a, t, alpha_tex, e.rec, x0, y0, x1, y1, b, lane);
  }
}

struct ResolveArgs {
  RasterArgs r;
  Tex albedo, normal, material, velocity, depth;
};

__global__ __launch_bounds__(256) void k_raster_resolve(ResolveArgs a) {
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE], s_thresh[VKR_SRGB_LUT_SIZE];
  srgb_lut_stage(s_lut, threadIdx.y * blockDim.x + threadIdx.x, 256);
  srgb_thresh_stage(s_thresh, threadIdx.y * blockDim.x + threadIdx.x, 256);
  __syncthreads();
  const int lx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ly = blockIdx.y * blockDim.y + threadIdx.y;
  if (lx >= a.albedo.w || ly >= a.albedo.h) return;
  const int px = a.albedo.ox + lx, py = a.albedo.oy + ly;
  const unsigned long long key = a.r.vis[(size_t)py * a.r.width + px];
  uint32_t o_albedo = 0u, o_normal = 0u, o_material = 0u, o_velocity = 0u, o_depth = 0x00FFFFFFu;  // cleared attachments
  if (key != ~0ull) {
    const uint32_t gid2 = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
    const uint32_t gid = gid2 >> 1;
    const DrawDev& d = a.r.draws[draw_of(a.r, gid)];
    const ScreenTri& t = a.r.setup[gid2];
    float lambda[3];
    uint32_t d24 = 0;
    cover(t, px, py, lambda, &d24);
    const FragUv fu = fragment_uv(t, px, py, lambda);
    const float* b = fu.b;
    const f3 in_normal = mk3(BARY(normal.x), BARY(normal.y), BARY(normal.z));
    const f2 in_uv = fu.uv;
    const f4 pa = mk4(BARY(pos_after.x), BARY(pos_after.y), BARY(pos_after.z), BARY(pos_after.w));
    const f4 pb = mk4(BARY(pos_before.x), BARY(pos_before.y), BARY(pos_before.z), BARY(pos_before.w));
    const f2 ddx = fu.ddx, ddy = fu.ddy;
    // opaque_taa.frag:26-46
    f4 out_albedo = mk4(0.5f, 0.5f, 0.5f, 1.0f);
    if (d.albedo_index != 0xFFFFFFFFu) out_albedo = sample_trilinear(a.r.tex[d.albedo_index], in_uv, ddx, ddy, s_lut);
    f4 out_material = mk4(0.5f, 0.9f, 0.5f, 0.5f);
    if (d.mr_index != 0xFFFFFFFFu) out_material = sample_trilinear(a.r.tex[d.mr_index], in_uv, ddx, ddy, s_lut);
    const f2 en = encode_normal(in_normal);
    const f2 vel = mk2(0.5f * (pb.x / pb.w - pa.x / pa.w), 0.5f * (pb.y / pb.w - pa.y / pa.w));
    o_albedo = float_to_srgb8_lds(out_albedo.x, s_thresh) | (float_to_srgb8_lds(out_albedo.y, s_thresh) << 8) | (float_to_srgb8_lds(out_albedo.z, s_thresh) << 16) | (float_to_unorm8(out_albedo.w) << 24);
    o_material = float_to_srgb8_lds(out_material.x, s_thresh) | (float_to_srgb8_lds(out_material.y, s_thresh) << 8) | (float_to_srgb8_lds(out_material.z, s_thresh) << 16) | (float_to_unorm8(out_material.w) << 24);
    o_normal = float_to_unorm16(en.x) | (float_to_unorm16(en.y) << 16);
    o_velocity = float_to_half_bits(vel.x) | (float_to_half_bits(vel.y) << 16);
    o_depth = (uint32_t)(key >> 32);
  }
  *texel_ptr<uint32_t>(a.albedo, lx, ly) = o_albedo;
  *texel_ptr<uint32_t>(a.normal, lx, ly) = o_normal;
  *texel_ptr<uint32_t>(a.material, lx, ly) = o_material;
  *texel_ptr<uint32_t>(a.velocity, lx, ly) = o_velocity;
  *texel_ptr<uint32_t>(a.depth, lx, ly) = o_depth;
}

// c = a * b with GLSL's mat4 * mat4 (column-major, each element a dot product accumulated left to right)
static void mat_mul(Mat4& c, const vkr_mat4& a, const vkr_mat4& b) {
  for (int col = 0; col < 4; col++)
    for (int row = 0; row < 4; row++) {
      float s = a.m[0 * 4 + row] * b.m[col * 4 + 0];
      for (int k = 1; k < 4; k++) s = s + a.m[k * 4 + row] * b.m[col * 4 + k];
      c.m[col * 4 + row] = s;
    }
}

}  // namespace vkr

using namespace vkr;

static uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

extern "C" uint64_t vkr_raster_scratch_bytes(uint32_t width, uint32_t height, uint32_t triangle_count) {
  return align_up((uint64_t)width * height * 8u, 256) + align_up(sizeof(DrawDev) * 1024u, 256) + align_up(sizeof(Pyramid) * RASTER_MAX_TEXTURES, 256) +
         align_up(sizeof(ScreenTri) * 2u * (uint64_t)triangle_count, 256) + 256u + align_up(sizeof(LargeEntry) * 2u * (uint64_t)triangle_count, 256);
}

extern "C" int vkr_raster_gbuffer(const vkr_raster_scene* scene, const vkr_gbuf_const* consts, const vkr_img* albedo,
                                  const vkr_img* normal, const vkr_img* material, const vkr_img* velocity, const vkr_img* depth,
                                  void* scratch, uint64_t scratch_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!scene || !consts || !scratch) { set_error("gbuf_opaque_taa: NULL argument"); return VKR_ERR_NULL; }
  if (scene->draw_count > 1024 || scene->texture_count > RASTER_MAX_TEXTURES) {
    set_error("gbuf_opaque_taa: at most 1024 draws and %d textures", RASTER_MAX_TEXTURES);
    return VKR_ERR_EXTENT;
  }
  ResolveArgs ra;
  VKR_TRY(make_tex(albedo, 0, VKR_FMT_RGBA8_SRGB, "gbuf_opaque_taa.albedo", &ra.albedo));
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "gbuf_opaque_taa.normal", &ra.normal));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "gbuf_opaque_taa.material", &ra.material));
  VKR_TRY(make_tex(velocity, 0, VKR_FMT_RG16_SFLOAT, "gbuf_opaque_taa.velocity", &ra.velocity));
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "gbuf_opaque_taa.depth", &ra.depth));
  if (!same_window(ra.albedo, ra.normal) || !same_window(ra.albedo, ra.material) || !same_window(ra.albedo, ra.velocity) ||
      !same_window(ra.albedo, ra.depth)) {
    set_error("gbuf_opaque_taa: attachments differ in extent");
    return VKR_ERR_EXTENT;
  }
  const int W = ra.albedo.fw, H = ra.albedo.fh;
  uint64_t total_tris = 0;
  for (uint32_t i = 0; i < scene->draw_count; i++) total_tris += scene->draws[i].index_count / 3u;
  if (total_tris >= 0x7FFFFFFFull) { set_error("gbuf_opaque_taa: too many triangles"); return VKR_ERR_EXTENT; }
  if (scratch_bytes < vkr_raster_scratch_bytes((uint32_t)W, (uint32_t)H, (uint32_t)total_tris)) { set_error("gbuf_opaque_taa: scratch too small"); return VKR_ERR_EXTENT; }
  std::vector<Pyramid> tex(scene->texture_count);
  for (uint32_t i = 0; i < scene->texture_count; i++) {
    const vkr_img& t = scene->textures[i];
    if (t.mip_count < 1 || t.mip_count > VKR_MAX_MIPS) { set_error("gbuf_opaque_taa: texture %u: bad mip count", i); return VKR_ERR_MIPS; }
    tex[i].count = (int)t.mip_count;
    for (int m = 0; m < (int)t.mip_count; m++) VKR_TRY(make_tex(&t, m, VKR_FMT_RGBA8_SRGB, "gbuf_opaque_taa.texture", &tex[i].mip[m]));
    for (int m = (int)t.mip_count; m < 16; m++) tex[i].mip[m] = tex[i].mip[0];
  }
  // per-draw constants: view_projection * model exactly as opaque_taa.vert:39,44 multiplies them
  std::vector<DrawDev> draws(scene->draw_count);
  uint32_t tri_base = 0;
  for (uint32_t i = 0; i < scene->draw_count; i++) {
    const vkr_raster_draw& s = scene->draws[i];
    if (s.transform_index >= scene->transform_count || (s.albedo_index != 0xFFFFFFFFu && s.albedo_index >= scene->texture_count) ||
        (s.mr_index != 0xFFFFFFFFu && s.mr_index >= scene->texture_count) || s.index_offset + s.index_count > scene->index_count) {
      set_error("gbuf_opaque_taa: draw %u references data outside the scene", i);
      return VKR_ERR_EXTENT;
    }
    DrawDev& d = draws[i];
    mat_mul(d.mvp, consts->view_projection, scene->transforms[s.transform_index].model);
    mat_mul(d.prev_mvp, consts->prev_view_projection, scene->transforms[s.transform_index].model);
    load_mat(d.normal_mat, scene->transforms[s.transform_index].normal);
    d.albedo_index = s.albedo_index; d.mr_index = s.mr_index;
    d.index_offset = s.index_offset; d.vertex_offset = s.vertex_offset;
    d.tri_base = tri_base; d.tri_count = s.index_count / 3u;
    d.alpha_test = (s.albedo_index != 0xFFFFFFFFu && !(s.reserved & VKR_RASTER_DRAW_OPAQUE_ALBEDO)) ? 1u : 0u;
    d.pad1 = 0;
    tri_base += d.tri_count;
  }
  if (tri_base >= 0x7FFFFFFFu) { set_error("gbuf_opaque_taa: too many triangles"); return VKR_ERR_EXTENT; }
  RasterArgs r;
  r.vertices = scene->vertices; r.indices = scene->indices;
  r.vis = (unsigned long long*)scratch;
  r.draws = (const DrawDev*)((uint8_t*)scratch + align_up((uint64_t)W * H * 8u, 256));
  r.tex = (const Pyramid*)((uint8_t*)r.draws + align_up(sizeof(DrawDev) * 1024u, 256));
  r.setup = (ScreenTri*)((uint8_t*)r.tex + align_up(sizeof(Pyramid) * RASTER_MAX_TEXTURES, 256));
  unsigned long long* large_state = (unsigned long long*)((uint8_t*)r.setup + align_up(sizeof(ScreenTri) * 2u * total_tris, 256));
  LargeEntry* large_list = (LargeEntry*)((uint8_t*)large_state + 256);
  r.draw_count = scene->draw_count;
  r.width = W; r.height = H;
  r.jitter_x = consts->jitter[0]; r.jitter_y = consts->jitter[1];
  for (uint32_t i = 0; i < scene->draw_count; i += 8) {
    DrawChunk c;
    const uint32_t n = scene->draw_count - i < 8u ? scene->draw_count - i : 8u;
    for (uint32_t k = 0; k < n; k++) c.d[k] = draws[i + k];
    for (uint32_t k = n; k < 8; k++) c.d[k] = draws[i];
    hipLaunchKernelGGL(k_raster_store_draws, dim3(1), dim3(64), 0, stream, c, const_cast<DrawDev*>(r.draws) + i, n);
  }
  for (uint32_t i = 0; i < scene->texture_count; i += 4) {
    TexChunk c;
    const uint32_t n = scene->texture_count - i < 4u ? scene->texture_count - i : 4u;
    for (uint32_t k = 0; k < 4; k++) c.p[k] = tex[i + (k < n ? k : 0)];
    hipLaunchKernelGGL(k_raster_store_textures, dim3(1), dim3(64), 0, stream, c, const_cast<Pyramid*>(r.tex) + i, n);
  }
  const size_t npx = (size_t)W * H;
  hipLaunchKernelGGL(k_raster_clear, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, r.vis, npx);
  {
    hipError_t e = hipMemsetAsync(large_state, 0, 256, stream);
    if (e != hipSuccess) { set_error("gbuf_opaque_taa: %s", hipGetErrorString(e)); return (int)e; }
  }
  if (tri_base) {
    hipLaunchKernelGGL(k_raster_setup, dim3((tri_base + 255) / 256), dim3(256), 0, stream, r, tri_base, large_state, large_list);
    hipLaunchKernelGGL(k_raster_small, dim3((tri_base + 3) / 4), dim3(256), 0, stream, r, tri_base);
  }
  // every draw's large triangles in one launch: submission order is carried by the record index in the key
  hipLaunchKernelGGL(k_raster_large, dim3(RASTER_LARGE_GRID), dim3(256), 0, stream, r, large_state, large_list);
  ra.r = r;
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_raster_resolve, grid2d(ra.albedo.w, ra.albedo.h, block), block, 0, stream, ra);
  return launch_status("gbuf_opaque_taa");
}
