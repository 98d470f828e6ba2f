// hiz_march.hpp — the hierarchical-Z ray march shared by the stochastic SSR trace (trace.comp) and
// the simple mirror SSR (ssr/shader.frag): one step of screen_trace.glsl:17-100 on a pyramid whose
// per-level descriptors live in LDS.  Exact IEEE sequence (it decides hit / no-hit).
#pragma once
#include "vkr_host.hpp"

namespace vkr {

// State of one ray between steps of hierarchical_raymarch_find_hor (trace.comp:206-268).  `position`
// is always origin + current_t * direction and the mip resolution is screen_size * 2^-mip (exact
// power-of-two scalings), so (current_t, mip, i, h) is the whole mutable state.
struct RayConst {
  f3 origin, direction, inv_direction;
  f3 normal, view_vec;  // pixel_normal (w0 of the horizon test) and camera_start
};
struct RayState { float t, h; int mip, i; };
struct MarchEnv {
  const uint4* mip_table;  // LDS: {base lo, base hi, pitch, w | h << 16} per pyramid level
  int mip_count;
  f2 screen_size, screen_size_inv;
  f2 uv_offset_abs;
  Proj pr;
  float horizon_d2;  // smallest d2 with sqrtf(d2) >= 0.3f: |v| < 0.3 <=> dot(v,v) < horizon_d2
  int min_mip = 0;   // most_detailed_mip: the march ends when it would refine below it; uv_offset_abs carries its 2^min_mip
  // LOCAL marches (multi-GPU head launch): only levels < local_levels are in memory, and of each only frame rows
  // [win_table[l].x, + win_table[l].y) — mip_table[l] then describes that window image (its base is the window's first row)
  // but keeps the FRAME's extent in .w, which decides inside / outside the frame.  A step that needs a texel of the frame
  // that is not there leaves the ray untouched and reports MARCH_PARK.
  const uint2* win_table = nullptr;
  int local_levels = 0;
};
enum { MARCH_END = 0, MARCH_MORE = 1, MARCH_PARK = 2 };

// The horizon update of trace.comp:253-262: v = reconstruct_view_vec(uv, surface_z) - camera_start, h = max(h, cos) iff
// |v| < 0.3.  z stays on the exact quotient (div_normal): v is a small difference of two view-space positions, so a
// 2^-22 relative error of z (v_rcp_f32 instead of the refined quotient) is an error of |P| / |v| * 2^-22 — 1e-4 to 1e-3 —
// in the cosine, measured as 49 texels of the 4K `raw` image outside tolerance.  Only the final 1 / |v| is approximate.
VKR_DEV void horizon_gate(const MarchEnv& env, const RayConst& rc, RayState& st, f2 uv, float surface_z) {
  const f3 v = reconstruct_view_vec(uv, surface_z, env.pr) - rc.view_vec;
  const float d2 = dot(v, v);
  if (d2 < env.horizon_d2) {  // length(v) < 0.3, decided exactly on the squared length
    const float c = dot(rc.normal, v) * __builtin_amdgcn_rsqf(d2);
    asm("v_max_f32 %0, %1, %2" : "=v"(st.h) : "v"(st.h), "v"(c));  // fmaxf without the canonicalising self-max (a NaN cosine is dropped either way)
  }
}

// One step of the march; returns false when the ray is finished.  HORIZON / PIN_STEPS = 15 / max 80 is
// hierarchical_raymarch_find_hor (trace.comp:206-268); no horizon / PIN_STEPS = 0 is the generic
// hierarchical_raymarch (screen_trace.glsl:51-100).
template <bool HORIZON, int PIN_STEPS, bool LOCAL>
VKR_DEV int march_step_ex(const MarchEnv& env, const RayConst& rc, RayState& st, int max_steps) {
  const float scale = __builtin_ldexpf(1.0f, -st.mip), scale_inv = __builtin_ldexpf(1.0f, st.mip);
  const f2 res = mk2(env.screen_size.x * scale, env.screen_size.y * scale);
  const f2 res_inv = mk2(env.screen_size_inv.x * scale_inv, env.screen_size_inv.y * scale_inv);
  const f3 position = madd(rc.origin, st.t, rc.direction);
  const f2 mip_pos = res * xy(position);
  // texelFetch(depth_tex, ivec2(p), mip): beyond the last mip or outside the mip extent -> 0
  float surface_z = 0.0f;
  if ((unsigned)st.mip < (unsigned)env.mip_count) {
    const uint4 m = env.mip_table[st.mip];
    const int tx = f2i_index(mip_pos.x), ty = f2i_index(mip_pos.y);
    if ((uint32_t)tx < (m.w & 0xFFFFu) && (uint32_t)ty < (m.w >> 16)) {  // 0 <= t < extent as one unsigned compare per axis
      typedef const __attribute__((address_space(1))) uint32_t* gptr_t;  // a global, not flat, address
      uint32_t row = (uint32_t)ty;
      if (LOCAL) {  // a texel of the frame: is it here?
        if (st.mip >= env.local_levels) return MARCH_PARK;
        const uint2 w = env.win_table[st.mip];
        row -= w.x;
        if (row >= w.y) return MARCH_PARK;
      }
      const uint64_t addr = (((uint64_t)m.y << 32) | m.x) + (uint64_t)(__umul24(row, m.z) + (uint32_t)tx * 4u);  // 32-bit offset, see toff()
      surface_z = d24_to_float(*(gptr_t)addr);
    }
  }
  // advance_ray (screen_trace.glsl:17-45)
  const f2 uv_offset = mk2(rc.direction.x < 0.0f ? -env.uv_offset_abs.x : env.uv_offset_abs.x,
                           rc.direction.y < 0.0f ? -env.uv_offset_abs.y : env.uv_offset_abs.y);
  const f2 floor_offset = mk2(rc.direction.x < 0.0f ? 0.0f : 1.0f, rc.direction.y < 0.0f ? 0.0f : 1.0f);
  f2 xy_plane = mk2(floorf(mip_pos.x), floorf(mip_pos.y)) + floor_offset;
  xy_plane = mk2(cfma(xy_plane.x, res_inv.x, uv_offset.x), cfma(xy_plane.y, res_inv.y, uv_offset.y));
  f3 t = (mk3(xy_plane.x, xy_plane.y, surface_z) - rc.origin) * rc.inv_direction;
  t.z = rc.direction.z > 0.0f ? t.z : 3.402823466e+38f;
  const float t_min = vmin(vmin(t.x, t.y), t.z);
  const bool above_surface = surface_z > position.z;
  const bool skipped_tile = (t_min != t.z) && above_surface;
  st.t = above_surface ? t_min : st.t;
  // trace.comp:245-250: the first 15 steps stay on the finest mip
  if (st.i >= PIN_STEPS) st.mip += skipped_tile ? 1 : -1;
  ++st.i;
  // trace.comp:253-262: horizon tracking around the new position
  if (HORIZON && st.mip <= 1) horizon_gate(env, rc, st, madd(xy(rc.origin), st.t, xy(rc.direction)), surface_z);
  return (st.i < max_steps && st.mip >= env.min_mip) ? MARCH_MORE : MARCH_END;
}
template <bool HORIZON, int PIN_STEPS>
VKR_DEV bool march_step(const MarchEnv& env, const RayConst& rc, RayState& st, int max_steps) {
  return march_step_ex<HORIZON, PIN_STEPS, false>(env, rc, st, max_steps) == MARCH_MORE;
}


// One of the PINNED steps of hierarchical_raymarch_find_hor (trace.comp:245-250: while i < 15 the march stays on its
// mip): march_step<true, 15> for a ray with st.mip == 0 and st.i < 15, with everything that is then known folded in —
// the level scale is 2^0 (res = screen_size, the same floats), the mip does not change, the horizon update always
// runs (mip 0 <= 1), the ray cannot end.  fetch0(tx, ty) returns texel (tx, ty) of pyramid level 0, 0 outside it.
// LOCAL: fetch0(tx, ty, &z) returns false when the texel is a texel of the frame that is not in memory: the ray stays as it is and
// the step reports false.
template <bool LOCAL, class Fetch0>
VKR_DEV bool march_step_pinned0(const MarchEnv& env, const RayConst& rc, RayState& st, const Fetch0& fetch0) {
  const f3 position = madd(rc.origin, st.t, rc.direction);
  const f2 mip_pos = env.screen_size * xy(position);
  float surface_z;
  if (LOCAL) {
    if (!fetch0(f2i_index(mip_pos.x), f2i_index(mip_pos.y), &surface_z)) return false;
  } else {
    fetch0(f2i_index(mip_pos.x), f2i_index(mip_pos.y), &surface_z);
  }
  const f2 uv_offset = mk2(rc.direction.x < 0.0f ? -env.uv_offset_abs.x : env.uv_offset_abs.x,
                           rc.direction.y < 0.0f ? -env.uv_offset_abs.y : env.uv_offset_abs.y);
  const f2 floor_offset = mk2(rc.direction.x < 0.0f ? 0.0f : 1.0f, rc.direction.y < 0.0f ? 0.0f : 1.0f);
  f2 xy_plane = mk2(floorf(mip_pos.x), floorf(mip_pos.y)) + floor_offset;
  xy_plane = mk2(cfma(xy_plane.x, env.screen_size_inv.x, uv_offset.x), cfma(xy_plane.y, env.screen_size_inv.y, uv_offset.y));
  f3 t = (mk3(xy_plane.x, xy_plane.y, surface_z) - rc.origin) * rc.inv_direction;
  t.z = rc.direction.z > 0.0f ? t.z : 3.402823466e+38f;
  const float t_min = vmin(vmin(t.x, t.y), t.z);
  st.t = surface_z > position.z ? t_min : st.t;
  ++st.i;
  horizon_gate(env, rc, st, madd(xy(rc.origin), st.t, xy(rc.direction)), surface_z);
  return true;
}

// the LDS descriptor of one pyramid level
VKR_DEV uint4 mip_descriptor(const Tex& m) {
  const uint64_t base = (uint64_t)m.p;
  return make_uint4((uint32_t)base, (uint32_t)(base >> 32), (uint32_t)m.pitch, (uint32_t)m.w | ((uint32_t)m.h << 16));
}
// initial_advance_ray (screen_trace.glsl:8-15) on the resolution of most_detailed_mip (env.min_mip)
VKR_DEV float initial_advance(const MarchEnv& env, const RayConst& rc) {
  const f2 uv_offset = mk2(rc.direction.x < 0.0f ? -env.uv_offset_abs.x : env.uv_offset_abs.x,
                           rc.direction.y < 0.0f ? -env.uv_offset_abs.y : env.uv_offset_abs.y);
  const f2 floor_offset = mk2(rc.direction.x < 0.0f ? 0.0f : 1.0f, rc.direction.y < 0.0f ? 0.0f : 1.0f);
  const float scale = __builtin_ldexpf(1.0f, -env.min_mip), scale_inv = __builtin_ldexpf(1.0f, env.min_mip);
  const f2 res = mk2(env.screen_size.x * scale, env.screen_size.y * scale);
  const f2 res_inv = mk2(env.screen_size_inv.x * scale_inv, env.screen_size_inv.y * scale_inv);
  const f2 cur_pos = res * xy(rc.origin);
  f2 xy_plane = mk2(floorf(cur_pos.x), floorf(cur_pos.y)) + floor_offset;
  xy_plane = mk2(cfma(xy_plane.x, res_inv.x, uv_offset.x), cfma(xy_plane.y, res_inv.y, uv_offset.y));
  const f2 t = (xy_plane - xy(rc.origin)) * xy(rc.inv_direction);
  return vmin(t.x, t.y);
}
VKR_DEV f3 safe_inverse(f3 d) {  // screen_trace.glsl:54-57
  return mk3(d.x != 0.0f ? 1.0f / d.x : 3.402823466e+38f, d.y != 0.0f ? 1.0f / d.y : 3.402823466e+38f,
             d.z != 0.0f ? 1.0f / d.z : 3.402823466e+38f);
}
// smallest float x with sqrtf(x) >= 0.3f: |v| < 0.3f <=> dot(v,v) < x (correctly rounded sqrt is monotone)
inline float horizon_threshold_d2() {
  float x = 0.3f * 0.3f;
  while (sqrtf(x) >= 0.3f) x = nextafterf(x, 0.0f);
  while (sqrtf(x) < 0.3f) x = nextafterf(x, 1.0f);
  return x;
}

}  // namespace vkr
